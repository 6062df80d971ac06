# Round 5, call P: closing measurements of the final tree (4-wave attention forward as the default): full GPU suite with its
# [parity] lines, smoke, the contract bench with the driver command line, kernel-trace stats of the cfg 2 edit and of the 1024^2 edit.
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -m gpu -q -s > gpurun_out/r05p_tests.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r05p_tests.log ); grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/r05p_tests.log | tail -8
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05p_smoke.log 2>&1; echo "smoke rc=$?" | tee -a gpurun_out/r05p_smoke.log ); tail -3 gpurun_out/r05p_smoke.log
( timeout 1500 python bench.py --steps 20 --warmup 5 > gpurun_out/r05p_bench_driver_cmd.json 2> gpurun_out/r05p_bench_driver_cmd.err; echo "bench rc=$?" ); tail -3 gpurun_out/r05p_bench_driver_cmd.err
python - <<'PY'
import json
d = json.load(open('gpurun_out/r05p_bench_driver_cmd.json'))
print('line bytes', len(json.dumps(d)))
print('cfg2', round(d['value'], 4), round(d['ms_per_step'], 1), d['ms_per_step_hip_events'])
print(json.dumps(d['roofline']['workloads'], indent=1))
print('cpu', {k: d['cpu_baseline'].get(k) for k in ('value', 'cfg1_4step_images_per_s', 'steps_executed', 'steps_not_executed', 'cores', 'torch_num_threads', 't_steps_s')})
print('prompt', {k: v for k, v in d['extra'].get('prompt_encode', {}).items() if k in ('T_prompt_s', 'T_qwen_s', 'T_t5_clip_s', 'error')})
PY
cd /tmp && export TMPDIR=/tmp
( timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_p -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-extra --cpu-baseline none > $GRAFT_REPO_ROOT/gpurun_out/r05p_prof_stdout.log 2>&1; echo "prof rc=$?" )
( timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_p2 -o bench -- python $GRAFT_REPO_ROOT/bench.py --workload single_1024x1024_28step --steps 1 --warmup 1 --no-extra --cpu-baseline none > $GRAFT_REPO_ROOT/gpurun_out/r05p_prof1024_stdout.log 2>&1; echo "prof1024 rc=$?" )
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py $(find /tmp/prof_p -name "*results.db" | head -1) gpurun_out/r05p_bench_kernel_stats.md "python bench.py --steps 1 --warmup 1 --no-extra --cpu-baseline none (cfg2; 3 edits: warm-up, timed, HIP-event pass; includes the model construction's init kernels)" > /dev/null 2>&1
python tools/rocpd_summary.py $(find /tmp/prof_p2 -name "*results.db" | head -1) gpurun_out/r05p_bench_1024_kernel_stats.md "python bench.py --workload single_1024x1024_28step --steps 1 --warmup 1 --no-extra --cpu-baseline none (3 edits)" > /dev/null 2>&1
head -16 gpurun_out/r05p_bench_kernel_stats.md; head -14 gpurun_out/r05p_bench_1024_kernel_stats.md | tail -6
