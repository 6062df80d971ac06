# Round 6, call B: gemm10_kernel (4 waves, hand-placed K loop) -- parity against gemm8 (bit for bit) and the first A/B:
# 256 x 256 tiles of gemm8 (m16) vs gemm10 vs the launch plan's own form vs hipBLASLt, interleaved in one process.
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_hip_kernels.py -m gpu -q -s -k "gemm10" > gpurun_out/r06b_tests.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r06b_tests.log ); grep -E "^FAILED|^ERROR|passed|failed|Error|error" gpurun_out/r06b_tests.log | tail -12
( AB_VARIANTS="256m16,1024m16,0m16,vendor" timeout 900 python tools/ab_gemm_variants.py 5 > gpurun_out/r06b_gemm10_ab.txt 2>&1; echo "ab rc=$?" ); cat gpurun_out/r06b_gemm10_ab.txt | tail -40
