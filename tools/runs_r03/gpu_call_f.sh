cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_hip_cfg3.py tests/test_hip_kernels.py -q -m gpu -x -k "attention" > gpurun_out/r03f_tests.log 2>&1; echo "pytest rc=$?" ); tail -3 gpurun_out/r03f_tests.log
( timeout 300 python tools/ab_attention_ring.py > gpurun_out/r03f_ab_ring.log 2>&1; echo "ring rc=$?" ); grep attention gpurun_out/r03f_ab_ring.log
( AB_ARMS="ring=3;ring=4" timeout 400 python tools/ab_edit_plans.py cfg2_single_512x512_28step 3 2 > gpurun_out/r03f_ab_edit_cfg2.log 2>&1; echo "ab_edit rc=$?" ); tail -5 gpurun_out/r03f_ab_edit_cfg2.log
( AB_ARMS="ring=3;ring=4" timeout 500 python tools/ab_edit_plans.py single_1024x1024_28step 2 1 > gpurun_out/r03f_ab_edit_1024.log 2>&1; echo "ab_edit1024 rc=$?" ); tail -5 gpurun_out/r03f_ab_edit_1024.log
