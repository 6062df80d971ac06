# closing bench lines after the second-stream change: default (cfg 2 + extras) and the CLI-resize workload
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( timeout 700 python bench.py --steps 5 --warmup 2 > gpurun_out/r02ab_bench.json 2> gpurun_out/r02ab_bench.err; echo "bench rc=$?" )
( timeout 200 python bench.py --workload cfg2cli_512x512_cond1mp_28step --steps 3 --warmup 1 --no-extra --cpu-baseline none > gpurun_out/r02ab_cfg2cli.json 2>/dev/null; echo "cli rc=$?" )
python - <<'PY'
import json
d=json.load(open('gpurun_out/r02ab_bench.json'))
r=d['roofline']; e=d.get('extra',{})
print('cfg2', d['value'], 'gemm', r['achieved'], r['frac'], 'traffic', r['traffic'], 'attn', r['other_kernels']['attention']['tflops'])
x=e.get('single_1024x1024_28step'); print('1024', x and x['value'], x and x['roofline']['achieved'], x and x['roofline']['frac'], x and x['roofline']['other_kernels']['attention']['tflops'])
print('prompt', e.get('prompt_encode',{}).get('T_prompt_s'), e.get('prompt_encode',{}).get('T_e2e_s'))
print('cfg5', {k: v for k, v in e.get('cfg5_train_step_1024x1024_bs1',{}).items() if k in ('value','ms_per_step','error')})
print('cpu', d.get('cpu_baseline',{}).get('value'), d.get('cpu_baseline',{}).get('t_step_s'))
c=json.load(open('gpurun_out/r02ab_cfg2cli.json')); print('cli', c['value'], c['roofline']['achieved'], c['roofline']['other_kernels']['attention']['tflops'])
PY
