cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( timeout 200 python -m pytest tests/test_hip_train_step.py -x -q -m gpu -s > gpurun_out/r02al_tests.log 2>&1; echo "pytest rc=$?" ); grep "prompt_embeds\|passed\|failed\|Error" gpurun_out/r02al_tests.log | tail -5
