# Round 4, call H: closing measurements of the final tree: full GPU test suite with its [parity] lines, the contract bench
# (driver command), kernel-trace stats of the 512^2 / 1024^2 edits and of the cfg 5 step, attention traffic re-count, cli shape.
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -m gpu -x -q -s > gpurun_out/r04h_tests.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r04h_tests.log ); tail -3 gpurun_out/r04h_tests.log
( timeout 900 python bench.py > gpurun_out/r04h_bench_default.json 2> gpurun_out/r04h_bench_default.err; echo "bench rc=$?" )
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04h_bench_default.json'))
r=d['roofline']; e=d.get('extra',{})
a=r['other_kernels']['attention']
print('cfg2', d['value'], d['ms_per_step'], 'gemm', r['achieved'], r['frac'], 'attn', a['tflops'], a['hbm_gbps_algorithmic'], 'conv', r['other_kernels']['conv']['tflops'], r['other_kernels']['conv']['hbm_gbps_algorithmic'], 'host', d['host']['enqueue_ms_per_step'])
x=e.get('single_1024x1024_28step'); print('1024', x and x['value'], x and x['roofline']['achieved'], x and x['roofline']['other_kernels']['attention']['tflops'])
print('prompt', e.get('prompt_encode',{}).get('T_prompt_s'))
t=e.get('cfg5_train_step_1024x1024_bs1',{}); print('cfg5', {k: v for k, v in t.items() if k in ('value','ms_per_step','error','peak_memory_gb','host_enqueue_ms_per_step','host_work_ms_per_step')}, (t.get('T_step_e2e') or {}).get('ms_per_step'), (t.get('T_step_e2e') or {}).get('last_step_ms'))
print('cpu', d.get('cpu_baseline',{}).get('value'), d.get('cpu_baseline',{}).get('t_step_s'), d.get('cpu_baseline',{}).get('cores'))
PY
cd /tmp && export TMPDIR=/tmp
( timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_h -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-extra --cpu-baseline none > $GRAFT_REPO_ROOT/gpurun_out/r04h_prof_stdout.log 2>&1; echo "prof rc=$?" )
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py $(find /tmp/prof_h -name "*results.db" | head -1) gpurun_out/r04h_bench_kernel_stats.md "python bench.py --steps 1 --warmup 1 --no-extra --cpu-baseline none (cfg2; 3 edits: warm-up, timed, HIP-event pass; includes the model construction's init kernels)" > /dev/null 2>&1
head -14 gpurun_out/r04h_bench_kernel_stats.md
cd /tmp
( timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_h2 -o bench -- python $GRAFT_REPO_ROOT/bench.py --workload single_1024x1024_28step --steps 1 --warmup 1 --no-extra --cpu-baseline none > $GRAFT_REPO_ROOT/gpurun_out/r04h_prof1024_stdout.log 2>&1; echo "prof1024 rc=$?" )
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py $(find /tmp/prof_h2 -name "*results.db" | head -1) gpurun_out/r04h_bench_1024_kernel_stats.md "python bench.py --workload single_1024x1024_28step --steps 1 --warmup 1 --no-extra --cpu-baseline none (3 edits: warm-up, timed, HIP-event pass; one stream)" > /dev/null 2>&1
head -12 gpurun_out/r04h_bench_1024_kernel_stats.md
cd /tmp
( TRAIN_STEPS=3 timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_h3 -o train -- python $GRAFT_REPO_ROOT/tools/train_prof.py > $GRAFT_REPO_ROOT/gpurun_out/r04h_proftrain_stdout.log 2>&1; echo "proftrain rc=$?" )
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py $(find /tmp/prof_h3 -name "*results.db" | head -1) gpurun_out/r04h_train_kernel_stats.md "cfg 5 train step (1024^2, bs 1, full depth): python tools/train_prof.py = 2 warm-up + 3 timed core steps, then 4 end-to-end steps (VAE encodes + VLM forward + core step), incl. model / optimiser-state construction" > /dev/null 2>&1
head -16 gpurun_out/r04h_train_kernel_stats.md
( TRAFFIC_SKIP="ln_modulate,gemm,vae" PMC_PASSES="time fetch write hit" bash tools/pmc_traffic.sh > gpurun_out/r04h_traffic_passes.log 2>&1 ); cat gpurun_out/traffic/passes.txt
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/traffic_h && cp gpurun_out/traffic/*.txt gpurun_out/traffic/items.json gpurun_out/traffic_h/ 2>/dev/null
python tools/pmc_traffic_summary.py gpurun_out/traffic gpurun_out/r04h_traffic_attention > gpurun_out/r04h_traffic_summary.log 2>&1; grep "attention" gpurun_out/r04h_traffic_attention.md | head -4
( timeout 300 python bench.py --workload cfg2cli_512x512_cond1mp_28step --steps 3 --warmup 1 --no-extra --cpu-baseline none > gpurun_out/r04h_bench_cfg2cli.json 2> gpurun_out/r04h_cfg2cli.err; echo "cfg2cli rc=$?" )
python - <<'PY'
import json
try:
    d=json.load(open('gpurun_out/r04h_bench_cfg2cli.json')); r=d['roofline']
    print(d['config']['workload'], d['value'], d['ms_per_step'], 'gemm', r['achieved'], r['frac'], 'attn', r['other_kernels']['attention']['tflops'])
except Exception as e: print(e)
PY
