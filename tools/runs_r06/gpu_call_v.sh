# Round 6, call V: the closing sequence on the LAST tree of the round (after the K-major GEMM forms moved to 16x16x32): full GPU suite with its [parity] lines, smoke, the N = 2 self-launch of the
# bench (gloo on one shared GPU: code path + cfg 4 slice, no scaling claim), the contract bench with the driver's command line,
# kernel-trace stats of the cfg 2 edit and of the 1024^2 edit.
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( timeout 1800 python -m pytest tests -m gpu -q -s > gpurun_out/r06v_tests.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r06v_tests.log ); grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/r06v_tests.log | tail -8
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06v_smoke.log 2>&1; echo "smoke rc=$?" | tee -a gpurun_out/r06v_smoke.log ); tail -3 gpurun_out/r06v_smoke.log
( FK_BENCH_BACKEND=gloo timeout 1200 python bench.py --gpus 2 --steps 1 --warmup 1 --cpu-baseline none > gpurun_out/r06v_n2_weak.out 2> gpurun_out/r06v_n2_weak.err; echo "n2 rc=$?" ); grep -a '^{"metric"' gpurun_out/r06v_n2_weak.out > gpurun_out/r06v_n2_weak.json; python -c "
import json; d = json.load(open('gpurun_out/r06v_n2_weak.json')); print('n2', d['n_gpus'], d['dist'], round(d['value'], 4), {k: (round(v['value'], 4), v['n_gpus']) for k, v in d['extra'].items() if 'value' in v})"
( timeout 1700 python bench.py --steps 20 --warmup 5 > gpurun_out/r06v_bench_driver_cmd.json 2> gpurun_out/r06v_bench_driver_cmd.err; echo "bench rc=$?" ); tail -3 gpurun_out/r06v_bench_driver_cmd.err
python - <<'PY'
import json
d = json.load(open('gpurun_out/r06v_bench_driver_cmd.json'))
print('line bytes', len(json.dumps(d)))
print('cfg2', round(d['value'], 4), round(d['ms_per_step'], 1), d['ms_per_step_hip_events'])
print(json.dumps(d['roofline']['workloads'], indent=1))
print('cpu', {k: d['cpu_baseline'].get(k) for k in ('value', 'cfg1_4step_images_per_s', 'steps_executed', 'cores', 't_steps_s')})
PY
cd /tmp && export TMPDIR=/tmp
( timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_p -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-extra --cpu-baseline none > $GRAFT_REPO_ROOT/gpurun_out/r06v_prof_stdout.log 2>&1; echo "prof rc=$?" )
( timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_p2 -o bench -- python $GRAFT_REPO_ROOT/bench.py --workload single_1024x1024_28step --steps 1 --warmup 1 --no-extra --cpu-baseline none > $GRAFT_REPO_ROOT/gpurun_out/r06v_prof1024_stdout.log 2>&1; echo "prof1024 rc=$?" )
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py $(find /tmp/prof_p -name "*results.db" | head -1) gpurun_out/r06v_bench_kernel_stats.md "python bench.py --steps 1 --warmup 1 --no-extra --cpu-baseline none (cfg2; 3 edits: warm-up, timed, HIP-event pass; includes the model construction's init kernels)" > /dev/null 2>&1
python tools/rocpd_summary.py $(find /tmp/prof_p2 -name "*results.db" | head -1) gpurun_out/r06v_bench_1024_kernel_stats.md "python bench.py --workload single_1024x1024_28step --steps 1 --warmup 1 --no-extra --cpu-baseline none (3 edits)" > /dev/null 2>&1
head -16 gpurun_out/r06v_bench_kernel_stats.md; head -14 gpurun_out/r06v_bench_1024_kernel_stats.md | tail -6
