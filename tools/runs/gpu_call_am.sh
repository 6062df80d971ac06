cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( timeout 150 python -m pytest tests/test_hip_cfg3.py tests/test_hip_vae.py -x -q -m gpu > gpurun_out/r02am_tests.log 2>&1; echo "pytest rc=$?" ); tail -2 gpurun_out/r02am_tests.log
