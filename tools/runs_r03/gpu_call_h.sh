# Round-3 closing measurements: full GPU test suite, the contract bench (driver command), kernel-trace stats of the 512^2 and
# the 1024^2 edit (one stream), cfg 3 with two timed batches, cfg 1 as defined on the host cores, the cli-plumbing shape.
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03h_tests.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r03h_tests.log ); tail -3 gpurun_out/r03h_tests.log
( timeout 700 python bench.py > gpurun_out/r03h_bench_default.json 2> gpurun_out/r03h_bench_default.err; echo "bench rc=$?" )
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03h_bench_default.json'))
r=d['roofline']; e=d.get('extra',{})
print('cfg2', d['value'], d['ms_per_step'], 'gemm', r['achieved'], r['frac'], 'attn', r['other_kernels']['attention']['tflops'])
x=e.get('single_1024x1024_28step'); print('1024', x and x['value'], x and x['roofline']['achieved'], x and x['roofline']['other_kernels']['attention']['tflops'])
print('prompt', e.get('prompt_encode',{}).get('T_prompt_s'))
print('cfg5', {k: v for k, v in e.get('cfg5_train_step_1024x1024_bs1',{}).items() if k in ('value','ms_per_step','error','peak_memory_gb')})
print('cpu', d.get('cpu_baseline',{}).get('value'), d.get('cpu_baseline',{}).get('t_step_s'), d.get('cpu_baseline',{}).get('block_time_shared_vs_distinct_weights'))
PY
cd /tmp && export TMPDIR=/tmp
( timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_h -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-extra --cpu-baseline none > $GRAFT_REPO_ROOT/gpurun_out/r03h_prof_stdout.log 2>&1; echo "prof rc=$?" )
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py $(find /tmp/prof_h -name "*results.db" | head -1) gpurun_out/r03h_bench_kernel_stats.md "python bench.py --steps 1 --warmup 1 --no-extra --cpu-baseline none (cfg2; 3 edits: warm-up, timed, HIP-event pass; includes the model construction's init kernels)" > /dev/null 2>&1
head -14 gpurun_out/r03h_bench_kernel_stats.md
cd /tmp
( timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_h2 -o bench -- python $GRAFT_REPO_ROOT/bench.py --workload single_1024x1024_28step --steps 1 --warmup 1 --no-extra --cpu-baseline none > $GRAFT_REPO_ROOT/gpurun_out/r03h_prof1024_stdout.log 2>&1; echo "prof1024 rc=$?" )
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py $(find /tmp/prof_h2 -name "*results.db" | head -1) gpurun_out/r03h_bench_1024_kernel_stats.md "python bench.py --workload single_1024x1024_28step --steps 1 --warmup 1 --no-extra --cpu-baseline none (3 edits: warm-up, timed, HIP-event pass; ALL on one stream: FK_OVERLAP_MLP defaults to 0 since round 3)" > /dev/null 2>&1
head -12 gpurun_out/r03h_bench_1024_kernel_stats.md
( timeout 300 python bench.py --workload cfg2cli_512x512_cond1mp_28step --steps 3 --warmup 1 --no-extra --cpu-baseline none > gpurun_out/r03h_bench_cfg2cli.json 2> gpurun_out/r03h_cfg2cli.err; echo "cfg2cli rc=$?" )
( timeout 900 python bench.py --workload cfg3_batch32_1024x1024_28step --steps 2 --warmup 1 --no-extra --cpu-baseline none > gpurun_out/r03h_bench_cfg3.json 2> gpurun_out/r03h_cfg3.err; echo "cfg3 rc=$?" )
python - <<'PY'
import json
for f in ('gpurun_out/r03h_bench_cfg2cli.json','gpurun_out/r03h_bench_cfg3.json'):
    try:
        d=json.load(open(f)); r=d['roofline']
        print(d['config']['workload'], d['value'], d['ms_per_step'], 'gemm', r['achieved'], r['frac'], 'attn', r['other_kernels']['attention']['tflops'])
    except Exception as e: print(f, e)
PY
( timeout 900 python bench.py --cpu-baseline cfg1 --steps 1 --warmup 0 --no-extra --no-roofline > gpurun_out/r03h_cpu_cfg1.json 2> gpurun_out/r03h_cpu_cfg1.err; echo "cfg1 rc=$?" )
python -c "
import json; d=json.load(open('gpurun_out/r03h_cpu_cfg1.json'))['cpu_baseline']; print({k:d[k] for k in ('cfg1_4step_images_per_s','t_steps_s','t_vae_encode_s','t_vae_decode_s','cores','block_time_shared_vs_distinct_weights')})"
