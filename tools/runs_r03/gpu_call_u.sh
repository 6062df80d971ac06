# call U: the driver's round-end sequence on the final tree: smoke(), then the contract bench once more (another box)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r03u_smoke.log 2>&1; echo "smoke rc=$?" ); tail -4 gpurun_out/r03u_smoke.log | cut -c1-250
( timeout 800 python bench.py > gpurun_out/r03u_bench_default.json 2> gpurun_out/r03u_bench_default.err; echo "bench rc=$?" )
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03u_bench_default.json'))
r=d['roofline']; e=d.get('extra',{})
print('cfg2', d['value'], d['ms_per_step'], 'gemm', r['achieved'], r['frac'], 'attn', r['other_kernels']['attention']['tflops'])
x=e.get('single_1024x1024_28step'); print('1024', x and x['value'], x and x['roofline']['achieved'], x and x['roofline']['other_kernels']['attention']['tflops'])
print('cfg5', {k: v for k, v in e.get('cfg5_train_step_1024x1024_bs1',{}).items() if k in ('value','ms_per_step','error','peak_memory_gb')})
print('cpu', d.get('cpu_baseline',{}).get('value'), d.get('cpu_baseline',{}).get('t_step_s'))
PY
