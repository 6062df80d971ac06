cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( timeout 300 python -m pytest tests/test_hip_mmdit.py -x -q -m gpu > gpurun_out/r02ah_tests.log 2>&1; echo "pytest rc=$?" ); tail -3 gpurun_out/r02ah_tests.log
