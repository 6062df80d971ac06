cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r02m_tests.log 2>&1; echo "pytest rc=$?" )
tail -2 gpurun_out/r02m_tests.log
( timeout 900 python bench.py --steps 5 --warmup 2 > gpurun_out/r02m_bench.json 2> gpurun_out/r02m_bench.err; echo "bench rc=$?" )
python - <<'PY'
import json
d=json.load(open('gpurun_out/r02m_bench.json'))
r=d['roofline']; e=d['extra']
print('cfg2', d['value'], 'gemm', r['achieved'], r['frac'], 'traffic', r['traffic'], 'attn', r['other_kernels']['attention']['tflops'])
x=e['single_1024x1024_28step']; print('1024', x['value'], 'gemm', x['roofline']['achieved'], 'attn', x['roofline']['other_kernels']['attention']['tflops'])
print('prompt', e['prompt_encode'].get('T_prompt_s'), e['prompt_encode'].get('error'))
print('cfg5', {k: v for k, v in e['cfg5_train_step_1024x1024_bs1'].items() if k in ('value','ms_per_step','error')})
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['t_step_s'])
PY
