cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( FK_GEMM_BN=257 timeout 600 python -m pytest tests/test_hip_kernels.py -x -q -k "gemm and not tile_choice" > gpurun_out/r02b_gemm8_tests.log 2>&1; echo "pytest(gemm8) rc=$?" )
tail -3 gpurun_out/r02b_gemm8_tests.log
( timeout 600 python tools/ab_gemm8.py 3 0 > gpurun_out/r02b_ab_gemm8.log 2>&1; echo "ab rc=$?" )
cat gpurun_out/r02b_ab_gemm8.log | tail -20
