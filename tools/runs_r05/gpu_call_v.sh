# Round 5, call V: the closing sequence once more on the last tree (after call P: dK / dV kernel restructured with its stream-K
# form behind the test hook, s_nop 2 in the forward): full GPU suite, smoke, the driver's bench command; the CLI-shape edit.
# (Kernel traces: call P's -- no kernel of the edits changed structure since.)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -m gpu -q -s > gpurun_out/r05v_tests.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r05v_tests.log ); grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/r05v_tests.log | tail -8
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05v_smoke.log 2>&1; echo "smoke rc=$?" | tee -a gpurun_out/r05v_smoke.log ); tail -3 gpurun_out/r05v_smoke.log
( timeout 1500 python bench.py --steps 20 --warmup 5 > gpurun_out/r05v_bench_driver_cmd.json 2> gpurun_out/r05v_bench_driver_cmd.err; echo "bench rc=$?" ); tail -3 gpurun_out/r05v_bench_driver_cmd.err
python - <<'PY'
import json
d = json.load(open('gpurun_out/r05v_bench_driver_cmd.json'))
print('line bytes', len(json.dumps(d)))
print('cfg2', round(d['value'], 4), round(d['ms_per_step'], 1), d['ms_per_step_hip_events'])
print(json.dumps(d['roofline']['workloads'], indent=1))
print('roofline', {k: d['roofline'].get(k) for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic')})
print('cpu', {k: d['cpu_baseline'].get(k) for k in ('value', 'cfg1_4step_images_per_s', 'steps_executed', 'steps_not_executed', 'cores', 'torch_num_threads', 't_steps_s')})
print('prompt', {k: v for k, v in d['extra'].get('prompt_encode', {}).items() if k in ('T_prompt_s', 'T_qwen_s', 'T_t5_clip_s', 'error')})
PY
( timeout 300 python bench.py --workload cfg2cli_512x512_cond1mp_28step --steps 3 --warmup 1 --no-extra --cpu-baseline none > gpurun_out/r05v_bench_cfg2cli.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r05v_bench_cfg2cli.json')); print('cfg2cli', round(d['value'],4), round(d['ms_per_step'],1), d['roofline']['workloads'])" )
