cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_hip_train_step.py -q -s -x > gpurun_out/r02i_train_tests.log 2>&1; echo "pytest rc=$?" )
grep -E "passed|failed|Error|error|assert|loss hip|worst|\[grad\]" gpurun_out/r02i_train_tests.log | tail -40
