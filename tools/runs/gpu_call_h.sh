cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_hip_backward.py -q -s > gpurun_out/r02h_bwd_tests.log 2>&1; echo "pytest rc=$?" )
grep -E "passed|failed|Error|error|assert" gpurun_out/r02h_bwd_tests.log | tail -25
