# Round-3 closing measurements of the FINAL tree (call H measured commit 88f7b11; cfg 1 on the host cores is not repeated):
# full GPU test suite, the contract bench (driver command), kernel-trace stats of the 512^2 / 1024^2 edits and of the cfg 5
# step, the cli-plumbing shape, cfg 3 with two timed batches.
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03s_tests.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r03s_tests.log ); tail -3 gpurun_out/r03s_tests.log
( timeout 700 python bench.py > gpurun_out/r03s_bench_default.json 2> gpurun_out/r03s_bench_default.err; echo "bench rc=$?" )
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03s_bench_default.json'))
r=d['roofline']; e=d.get('extra',{})
print('cfg2', d['value'], d['ms_per_step'], 'gemm', r['achieved'], r['frac'], 'attn', r['other_kernels']['attention']['tflops'])
x=e.get('single_1024x1024_28step'); print('1024', x and x['value'], x and x['roofline']['achieved'], x and x['roofline']['other_kernels']['attention']['tflops'])
print('prompt', e.get('prompt_encode',{}).get('T_prompt_s'))
print('cfg5', {k: v for k, v in e.get('cfg5_train_step_1024x1024_bs1',{}).items() if k in ('value','ms_per_step','error','peak_memory_gb')})
print('cpu', d.get('cpu_baseline',{}).get('value'), d.get('cpu_baseline',{}).get('t_step_s'), d.get('cpu_baseline',{}).get('block_time_shared_vs_distinct_weights'))
PY
cd /tmp && export TMPDIR=/tmp
( timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_s -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-extra --cpu-baseline none > $GRAFT_REPO_ROOT/gpurun_out/r03s_prof_stdout.log 2>&1; echo "prof rc=$?" )
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py $(find /tmp/prof_s -name "*results.db" | head -1) gpurun_out/r03s_bench_kernel_stats.md "python bench.py --steps 1 --warmup 1 --no-extra --cpu-baseline none (cfg2; 3 edits: warm-up, timed, HIP-event pass; includes the model construction's init kernels)" > /dev/null 2>&1
head -14 gpurun_out/r03s_bench_kernel_stats.md
cd /tmp
( timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_s2 -o bench -- python $GRAFT_REPO_ROOT/bench.py --workload single_1024x1024_28step --steps 1 --warmup 1 --no-extra --cpu-baseline none > $GRAFT_REPO_ROOT/gpurun_out/r03s_prof1024_stdout.log 2>&1; echo "prof1024 rc=$?" )
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py $(find /tmp/prof_s2 -name "*results.db" | head -1) gpurun_out/r03s_bench_1024_kernel_stats.md "python bench.py --workload single_1024x1024_28step --steps 1 --warmup 1 --no-extra --cpu-baseline none (3 edits: warm-up, timed, HIP-event pass; ALL on one stream: FK_OVERLAP_MLP defaults to 0 since round 3)" > /dev/null 2>&1
head -12 gpurun_out/r03s_bench_1024_kernel_stats.md
( timeout 300 python bench.py --workload cfg2cli_512x512_cond1mp_28step --steps 3 --warmup 1 --no-extra --cpu-baseline none > gpurun_out/r03s_bench_cfg2cli.json 2> gpurun_out/r03s_cfg2cli.err; echo "cfg2cli rc=$?" )
( timeout 900 python bench.py --workload cfg3_batch32_1024x1024_28step --steps 2 --warmup 1 --no-extra --cpu-baseline none > gpurun_out/r03s_bench_cfg3.json 2> gpurun_out/r03s_cfg3.err; echo "cfg3 rc=$?" )
python - <<'PY'
import json
for f in ('gpurun_out/r03s_bench_cfg2cli.json','gpurun_out/r03s_bench_cfg3.json'):
    try:
        d=json.load(open(f)); r=d['roofline']
        print(d['config']['workload'], d['value'], d['ms_per_step'], 'gemm', r['achieved'], r['frac'], 'attn', r['other_kernels']['attention']['tflops'])
    except Exception as e: print(f, e)
PY
cd /tmp
( timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_s3 -o train -- python $GRAFT_REPO_ROOT/tools/train_prof.py > $GRAFT_REPO_ROOT/gpurun_out/r03s_proftrain_stdout.log 2>&1; echo "proftrain rc=$?" )
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py $(find /tmp/prof_s3 -name "*results.db" | head -1) gpurun_out/r03s_train_kernel_stats.md "cfg 5 train step (1024^2, bs 1, full depth): python tools/train_prof.py = 5 steps (2 warm-up + 3 timed) incl. model / optimiser-state construction" > /dev/null 2>&1
head -30 gpurun_out/r03s_train_kernel_stats.md
tail -1 gpurun_out/r03s_proftrain_stdout.log | cut -c1-400

