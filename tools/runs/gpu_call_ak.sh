cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( timeout 280 python -m pytest tests/test_hip_mmdit.py tests/test_hip_pipeline.py tests/test_hip_kernels.py -x -q -m gpu > gpurun_out/r02ak_tests.log 2>&1; echo "pytest rc=$?" ); tail -2 gpurun_out/r02ak_tests.log
