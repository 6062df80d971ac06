# Round-2 closing measurements: full GPU test suite, HBM-traffic counters of the CURRENT kernels, the contract bench
# (with the refreshed traffic side file), kernel-trace stats.
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02y_tests.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r02y_tests.log )
tail -3 gpurun_out/r02y_tests.log
rm -rf gpurun_out/traffic
bash tools/pmc_traffic.sh > gpurun_out/r02y_traffic.log 2>&1
tail -6 gpurun_out/r02y_traffic.log
mkdir -p gpurun_out/profiles_new
python tools/pmc_traffic_summary.py gpurun_out/traffic gpurun_out/profiles_new/r02_traffic > gpurun_out/r02y_traffic_summary.log 2>&1 && python tools/traffic_json.py gpurun_out/profiles_new/r02_traffic > /dev/null 2>&1 && cp gpurun_out/profiles_new/r02_traffic.json profiles/r02_traffic.json
echo "traffic summary rc=$?"
( timeout 700 python bench.py --steps 5 --warmup 2 > gpurun_out/r02y_bench.json 2> gpurun_out/r02y_bench.err; echo "bench rc=$?" )
python - <<'PY'
import json
d=json.load(open('gpurun_out/r02y_bench.json'))
r=d['roofline']; e=d.get('extra',{})
print('cfg2', d['value'], 'gemm', r['achieved'], r['frac'], 'traffic', r['traffic'], 'attn', r['other_kernels']['attention']['tflops'])
x=e.get('single_1024x1024_28step'); print('1024', x and x['value'], x and x['roofline']['achieved'], x and x['roofline']['other_kernels']['attention']['tflops'])
print('prompt', e.get('prompt_encode',{}).get('T_prompt_s'), e.get('prompt_encode',{}).get('error'))
print('cfg5', {k: v for k, v in e.get('cfg5_train_step_1024x1024_bs1',{}).items() if k in ('value','ms_per_step','error','trainable_params')})
print('cpu', d.get('cpu_baseline',{}).get('value'), d.get('cpu_baseline',{}).get('t_step_s'))
PY
cd /tmp && export TMPDIR=/tmp
( timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_y -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-extra --cpu-baseline none > $GRAFT_REPO_ROOT/gpurun_out/r02y_prof_stdout.log 2>&1; echo "prof rc=$?" )
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py $(find /tmp/prof_y -name "*results.db" | head -1) gpurun_out/r02y_bench_kernel_stats.md "python bench.py --steps 1 --warmup 1 --no-extra --cpu-baseline none (cfg2; 3 edits: warm-up, timed, HIP-event pass)" > /dev/null 2>&1
head -16 gpurun_out/r02y_bench_kernel_stats.md
