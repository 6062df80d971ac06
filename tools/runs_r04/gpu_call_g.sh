# Round 4, call G: stream-K ranges for the long-K GEMMs at M = 8704 -- parity, isolated A/B, A/B inside the 1024^2 edit and the train step.
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_hip_kernels.py -m gpu -x -q -s -k "hot_gemm or tile_choice or gemm_large or gemm_grouped" > gpurun_out/r04g_tests.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r04g_tests.log ); tail -3 gpurun_out/r04g_tests.log
grep -h "1024^2\] \|variant 640" gpurun_out/r04g_tests.log | cut -c1-200 | head
( AB_SHAPES="8704x3072x15360,8704x3072x12288,17408x3072x15360,2560x3072x15360" AB_VARIANTS="128,256,640,0,vendor" timeout 300 python tools/ab_gemm_variants.py 3 > gpurun_out/r04g_ab_gemm.txt 2>&1; echo "ab gemm rc=$?" ); tail -5 gpurun_out/r04g_ab_gemm.txt
( AB_SHAPES="8704x3072x15360,8704x3072x12288" AB_VARIANTS="256,640,0" timeout 300 python tools/ab_gemm_variants.py 3 3 >> gpurun_out/r04g_ab_gemm.txt 2>&1; echo "ab gemm epi3 rc=$?" ); tail -2 gpurun_out/r04g_ab_gemm.txt
( AB_ARMS="plan=1;plan=3" timeout 400 python tools/ab_edit_plans.py single_1024x1024_28step 2 1 > gpurun_out/r04g_ab_edit_1024.txt 2>&1; echo "ab edit rc=$?" ); tail -5 gpurun_out/r04g_ab_edit_1024.txt
for pl in 1 3; do ( FK_GEMM_PLAN=$pl TRAIN_STEPS=4 timeout 300 python tools/train_prof.py > gpurun_out/r04g_train_plan$pl.json 2> gpurun_out/r04g_train_plan$pl.err; echo "train plan=$pl rc=$?" ); python -c "
import json; d=json.load(open('gpurun_out/r04g_train_plan$pl.json')); print('cfg5 plan=$pl', d['ms_per_step'], d['host_enqueue_ms_per_step'], (d.get('T_step_e2e') or {}).get('ms_per_step'))"; done
