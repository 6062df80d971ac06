cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( FK_GEMM_BN=258 timeout 600 python -m pytest tests/test_hip_kernels.py -x -q -k "gemm and not tile_choice" > gpurun_out/r02c_gemm8v1_tests.log 2>&1; echo "pytest(gemm8 var1) rc=$?" )
tail -2 gpurun_out/r02c_gemm8v1_tests.log
( AB_VARIANTS=256,257,258,259,260,vendor AB_SHAPES=2560x12288x3072,8704x9216x3072,8704x3072x12288,32768x3072x12288,32768x12288x3072 timeout 600 python tools/ab_gemm8.py 4 0 > gpurun_out/r02c_ab.log 2>&1; echo "ab rc=$?" )
tail -6 gpurun_out/r02c_ab.log
( AB_VARIANTS=256,257,258 AB_SHAPES=8704x12288x3072,32768x12288x3072 timeout 600 python tools/ab_gemm8.py 3 1 > gpurun_out/r02c_ab_gelu.log 2>&1; echo "ab gelu rc=$?" )
tail -3 gpurun_out/r02c_ab_gelu.log
SHAPE="32768 3072 12288" bash tools/pmc_gemm_compare.sh "g5:FK_GEMM_BN=256" "g8:FK_GEMM_BN=257" "g8s1:FK_GEMM_BN=258" > gpurun_out/r02c_pmc.txt 2>&1
cat gpurun_out/r02c_pmc.txt
