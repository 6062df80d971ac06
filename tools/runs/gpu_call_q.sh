cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_hip_train_step.py tests/test_hip_backward.py -x -q -m gpu -s > gpurun_out/r02q_tests.log 2>&1; echo "pytest rc=$?" )
grep -v "^\[grad\]" gpurun_out/r02q_tests.log | tail -25
