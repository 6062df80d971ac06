# call P: K-major GEMM operands (layout 1 / 2): parity against the transposed-copy path, then timing on the backward's shapes
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_hip_kernels.py -q -m gpu -k "gemm" -x > gpurun_out/r03p_tests.log 2>&1; echo "pytest rc=$?" ); tail -25 gpurun_out/r03p_tests.log | cut -c1-200
( timeout 300 python tools/ab_gemm_layouts.py > gpurun_out/r03p_gemm_layouts.txt 2>&1; echo "ab rc=$?" ); cat gpurun_out/r03p_gemm_layouts.txt | cut -c1-220
