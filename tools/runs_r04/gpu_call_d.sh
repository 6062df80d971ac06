# Round 4, call D: hybrid stream-K (whole rounds + dealt-out tail) -- attention tests, isolated A/B, A/B inside the edits.
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_hip_cfg3.py tests/test_hip_pipeline.py tests/test_hip_kernels.py tests/test_hip_backward.py tests/test_hip_cfg5.py -m gpu -x -q -s -k "attention or batch32 or full_size or train_step_forward" > gpurun_out/r04d_tests.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r04d_tests.log ); tail -4 gpurun_out/r04d_tests.log
grep -h "stream-K\|outlier\|batch independence" gpurun_out/r04d_tests.log | cut -c1-260 | head -40
( timeout 300 python tools/ab_attention_split.py > gpurun_out/r04d_ab_attention_split.txt 2>&1; echo "ab rc=$?" ); tail -7 gpurun_out/r04d_ab_attention_split.txt
( AB_ARMS="split=0;split=1" timeout 400 python tools/ab_edit_plans.py single_1024x1024_28step 2 1 > gpurun_out/r04d_ab_edit_1024.txt 2>&1; echo "ab edit rc=$?" ); tail -5 gpurun_out/r04d_ab_edit_1024.txt
( AB_ARMS="split=0;split=1" timeout 400 python tools/ab_edit_plans.py cfg2cli_512x512_cond1mp_28step 2 1 > gpurun_out/r04d_ab_edit_cli.txt 2>&1; echo "ab edit cli rc=$?" ); tail -5 gpurun_out/r04d_ab_edit_cli.txt
for sp in 0 1; do ( FK_ATTN_SPLIT=$sp TRAIN_STEPS=4 timeout 300 python tools/train_prof.py > gpurun_out/r04d_train_split$sp.json 2> gpurun_out/r04d_train_split$sp.err; echo "train split=$sp rc=$?" ); python -c "
import json; d=json.load(open('gpurun_out/r04d_train_split$sp.json')); print('cfg5 split=$sp', d['ms_per_step'], d['peak_memory_gb'], d['host_enqueue_ms_per_step'])"; done
