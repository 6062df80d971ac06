# Round 4, call K: the fp32-class VAE encoder: kernel / encoder parity tests, timing, and the cfg 5 step with it (T_step_e2e).
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_hip_vae.py tests/test_hip_kernels.py -m gpu -x -q -s > gpurun_out/r04k_tests.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r04k_tests.log ); grep "fp32\|passed\|failed\|Error" gpurun_out/r04k_tests.log | tail -30
( timeout 300 python tools/time_vae_encode.py > gpurun_out/r04k_vae_encode_time.txt 2>&1; echo "time rc=$?" ); cat gpurun_out/r04k_vae_encode_time.txt | tail -5
( TRAIN_STEPS=3 timeout 600 python tools/train_prof.py > gpurun_out/r04k_train.json 2> gpurun_out/r04k_train.err; echo "train rc=$?" ); tail -3 gpurun_out/r04k_train.err
python - <<'PY'
import json
try:
    t=json.loads(open('gpurun_out/r04k_train.json').read().strip().splitlines()[-1])
    print({k: v for k, v in t.items() if k in ('value','ms_per_step','error','peak_memory_gb')}); print(t.get('T_step_e2e'))
except Exception as e: print('ERR', e)
PY
