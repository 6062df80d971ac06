cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_hip_vae.py tests/test_hip_cfg3.py tests/test_hip_pixels.py -q -m gpu -x -k "vae or conv or decode or encode or pixel" > gpurun_out/r03g_tests.log 2>&1; echo "pytest rc=$?" ); tail -12 gpurun_out/r03g_tests.log | cut -c1-200
( timeout 300 python tools/ab_conv.py > gpurun_out/r03g_ab_conv.log 2>&1; echo "ab_conv rc=$?" ); grep conv3x3 gpurun_out/r03g_ab_conv.log; tail -3 gpurun_out/r03g_ab_conv.log | grep -v conv3x3
