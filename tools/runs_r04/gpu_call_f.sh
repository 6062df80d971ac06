# Round 4, call F: stream-K after the control-word fix and with tail ranges walked from their end -- tests, isolated A/B, in-edit A/B.
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_hip_cfg3.py tests/test_hip_backward.py -m gpu -x -q -s -k "stream_k or attention" > gpurun_out/r04f_tests.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r04f_tests.log ); tail -3 gpurun_out/r04f_tests.log
grep -h "stream-K" gpurun_out/r04f_tests.log | cut -c1-230 | head -30
( timeout 300 python tools/ab_attention_split.py > gpurun_out/r04f_ab_attention_split.txt 2>&1; echo "ab rc=$?" ); tail -7 gpurun_out/r04f_ab_attention_split.txt
( AB_ARMS="split=0;split=1" timeout 400 python tools/ab_edit_plans.py single_1024x1024_28step 2 1 > gpurun_out/r04f_ab_edit_1024.txt 2>&1; echo "ab edit rc=$?" ); tail -5 gpurun_out/r04f_ab_edit_1024.txt
for sp in 0 1; do ( FK_ATTN_SPLIT=$sp TRAIN_STEPS=4 timeout 300 python tools/train_prof.py > gpurun_out/r04f_train_split$sp.json 2> gpurun_out/r04f_train_split$sp.err; echo "train split=$sp rc=$?" ); python -c "
import json; d=json.load(open('gpurun_out/r04f_train_split$sp.json')); print('cfg5 split=$sp', d['ms_per_step'], d['host_enqueue_ms_per_step'], d.get('host_work_ms_per_step'), (d.get('T_step_e2e') or {}).get('ms_per_step'))"; done
