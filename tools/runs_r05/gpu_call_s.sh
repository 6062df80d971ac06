# Round 5, call S: the paired dK / dV pass of the attention backward on the stream-K grid: parity / determinism tests, ms per
# call against the previous build, the cfg 5 step.
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( timeout 900 python -m pytest -x -q -s tests/test_hip_backward.py -k attention tests/test_hip_cfg5.py tests/test_hip_train_step.py > gpurun_out/r05s_tests.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r05s_tests.log ); grep -E "stream-K d|passed|failed|Error" gpurun_out/r05s_tests.log | tail -14
O=gpurun_out/r05s_attention_bwd_ab.txt
: > $O
run() { env "$@" 2>&1 | grep -v amdgpu.ids >> $O; }
run AB_MODES=1 timeout 200 python tools/ab_attention_bwd.py dkv_streamk
run AB_MODES=1 FK_LIB_PATH=build_ab/base/gpt_image_edit_amd/libfk_gfx950.so timeout 200 python tools/ab_attention_bwd.py dkv_plain
run AB_MODES=1 timeout 200 python tools/ab_attention_bwd.py dkv_streamk
run AB_MODES=1 FK_LIB_PATH=build_ab/base/gpt_image_edit_amd/libfk_gfx950.so timeout 200 python tools/ab_attention_bwd.py dkv_plain
cat $O
( TRAIN_STEPS=4 timeout 400 python tools/train_prof.py 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('cfg5 step ms', round(d['ms_per_step'],1), 'e2e', (d.get('T_step_e2e') or {}).get('ms_per_step'), 'loss', d['loss'])" ) | tee gpurun_out/r05s_train.txt
