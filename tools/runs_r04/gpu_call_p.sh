# Round 4, call P: closing measurements of the FINAL tree (block API default, fp32-class encoder in T_step_e2e): full GPU suite
# with its [parity] lines, smoke, the contract bench (driver command), kernel-trace stats of the same command.
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -m gpu -x -q -s > gpurun_out/r04p_tests.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r04p_tests.log ); tail -3 gpurun_out/r04p_tests.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04p_smoke.log 2>&1; echo "smoke rc=$?" | tee -a gpurun_out/r04p_smoke.log ); tail -2 gpurun_out/r04p_smoke.log
( timeout 900 python bench.py > gpurun_out/r04p_bench_default.json 2> gpurun_out/r04p_bench_default.err; echo "bench rc=$?" )
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04p_bench_default.json'))
r=d['roofline']; e=d.get('extra',{})
a=r['other_kernels']['attention']
print('cfg2', d['value'], d['ms_per_step'], 'gemm', r['achieved'], r['frac'], r['launches_per_edit'], 'attn', a['tflops'], a['launches'], 'conv', r['other_kernels']['conv']['tflops'], 'host', d['host']['enqueue_ms_per_step'])
x=e.get('single_1024x1024_28step'); print('1024', x and x['value'], x and x['roofline']['achieved'], x and x['roofline']['other_kernels']['attention']['tflops'])
print('cfg3', e.get('cfg3_batch32_1024x1024_28step',{}).get('value'), e.get('cfg3_batch32_1024x1024_28step',{}).get('error'))
print('prompt', e.get('prompt_encode',{}).get('T_prompt_s'))
t=e.get('cfg5_train_step_1024x1024_bs1',{}); print('cfg5', {k: v for k, v in t.items() if k in ('value','ms_per_step','error','peak_memory_gb','host_enqueue_ms_per_step','host_work_ms_per_step')}, (t.get('T_step_e2e') or {}).get('ms_per_step'), (t.get('T_step_e2e') or {}).get('last_step_ms'), (t.get('T_step_e2e') or {}).get('vae_encode_x2_bf16_ms'))
print('cpu', d.get('cpu_baseline',{}).get('value'), d.get('cpu_baseline',{}).get('t_step_s'), d.get('cpu_baseline',{}).get('cores'))
PY
cd /tmp && export TMPDIR=/tmp
( timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_p -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-extra --cpu-baseline none > $GRAFT_REPO_ROOT/gpurun_out/r04p_prof_stdout.log 2>&1; echo "prof rc=$?" )
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py $(find /tmp/prof_p -name "*results.db" | head -1) gpurun_out/r04p_bench_kernel_stats.md "python bench.py --steps 1 --warmup 1 --no-extra --cpu-baseline none (cfg2; 3 edits: warm-up, timed, HIP-event pass; includes the model construction's init kernels)" > /dev/null 2>&1
head -14 gpurun_out/r04p_bench_kernel_stats.md
