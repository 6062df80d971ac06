cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( FK_GEMM_BN=129 timeout 600 python -m pytest tests/test_hip_kernels.py -x -q -k "gemm and not tile_choice" > gpurun_out/r02f_gemm9_tests.log 2>&1; echo "pytest(gemm9) rc=$?" )
tail -2 gpurun_out/r02f_gemm9_tests.log
( AB_VARIANTS=128,129,256,vendor AB_SHAPES=2560x9216x3072,2560x3072x12288,2560x3072x15360,2560x3072x3072,2560x12288x3072,8704x3072x12288,8704x3072x3072,32768x3072x12288 timeout 600 python tools/ab_gemm_variants.py 4 0 > gpurun_out/r02f_ab.log 2>&1; echo "ab rc=$?" )
tail -9 gpurun_out/r02f_ab.log
( AB_VARIANTS=128,129 AB_SHAPES=2560x9216x3072,2560x3072x12288 timeout 600 python tools/ab_gemm_variants.py 3 3 > gpurun_out/r02f_ab_epi3.log 2>&1; echo "ab epi3 rc=$?" ); tail -3 gpurun_out/r02f_ab_epi3.log
SHAPE="32768 3072 12288" bash tools/pmc_gemm_compare.sh "g2:FK_GEMM_BN=128" "g9:FK_GEMM_BN=129" > gpurun_out/r02f_pmc.txt 2>&1
cat gpurun_out/r02f_pmc.txt
