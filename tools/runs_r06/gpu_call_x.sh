# Round 6, call X: gemm10 (one workgroup per CU) against gemm8 on cfg 3's K = 3072 shapes (M = 278528): does the plan's rule leave anything?
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( AB_SHAPES="278528x12288x3072,278528x9216x3072,278528x3072x3072,278528x3072x15360" AB_VARIANTS="256m16,1024m16" timeout 900 python tools/ab_gemm_variants.py 5 0 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r06x_gemm10_cfg3_shapes.txt 2>&1
( AB_SHAPES="278528x12288x3072" AB_VARIANTS="256m16,1024m16" timeout 900 python tools/ab_gemm_variants.py 5 1 2>&1 | grep -v amdgpu.ids ) >> gpurun_out/r06x_gemm10_cfg3_shapes.txt 2>&1
cat gpurun_out/r06x_gemm10_cfg3_shapes.txt
