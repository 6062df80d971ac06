# Round 6, call I: production build of gemm10 (plan picks it for K >= 6144 grids of >= 4 tiles per CU, one workgroup per CU): the
# GEMM tests, the cfg 3 tests, and the A/B of the plan's choice (0m16) against gemm8 forced (256m16) and the vendor library.
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( timeout 1200 python -m pytest tests/test_hip_kernels.py tests/test_hip_cfg3.py -m gpu -q -x 2>&1 | tail -4 )
( AB_SHAPES="32768x3072x12288,32768x3072x15360,278528x3072x12288,32768x12288x3072,8704x3072x12288" AB_VARIANTS="256m16,1024m16,0m16,vendor" timeout 900 python tools/ab_gemm_variants.py 5 0 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r06i_gemm10_plan_ab.txt 2>&1
cat gpurun_out/r06i_gemm10_plan_ab.txt
