cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( timeout 900 python -m pytest tests -q -m gpu -x --durations=8 > gpurun_out/r03b_tests.log 2>&1; echo "pytest rc=$?" ); tail -14 gpurun_out/r03b_tests.log
( AB_SHAPES=2560x9216x3072,2560x12288x3072,2560x3072x12288,2560x3072x15360,2560x3072x3072,8704x9216x3072,8704x12288x3072,8704x3072x15360 timeout 300 python tools/ab_gemm_variants.py 3 0 > gpurun_out/r03b_ab_gemm.log 2>&1; echo "ab_gemm rc=$?" ); cat gpurun_out/r03b_ab_gemm.log | tail -12
( timeout 400 python tools/ab_edit_plans.py cfg2_single_512x512_28step 3 2 > gpurun_out/r03b_ab_edit_cfg2.log 2>&1; echo "ab_edit rc=$?" ); tail -8 gpurun_out/r03b_ab_edit_cfg2.log
( timeout 400 python tools/ab_edit_plans.py single_1024x1024_28step 2 1 > gpurun_out/r03b_ab_edit_1024.log 2>&1; echo "ab_edit1024 rc=$?" ); tail -8 gpurun_out/r03b_ab_edit_1024.log
