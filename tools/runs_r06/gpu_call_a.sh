# Round 6, call A: the tree after the host-side items (bench self-launch, variant_used OUT field, gedit generator): full GPU
# suite, smoke, the N = 2 self-launch of bench.py (gloo on one shared GPU: the N > 1 code path without a launcher), and a short
# default bench (no cfg 3, no CPU baseline) as this round's baseline on this box.
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -m gpu -q -s > gpurun_out/r06a_tests.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r06a_tests.log ); grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/r06a_tests.log | tail -8
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06a_smoke.log 2>&1; echo "smoke rc=$?" | tee -a gpurun_out/r06a_smoke.log ); tail -3 gpurun_out/r06a_smoke.log
( FK_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 1 --warmup 1 --no-extra --cpu-baseline none > gpurun_out/r06a_n2_selflaunch.json 2> gpurun_out/r06a_n2_selflaunch.err; echo "n2 rc=$?" ); tail -c 600 gpurun_out/r06a_n2_selflaunch.json; tail -3 gpurun_out/r06a_n2_selflaunch.err
( FK_BENCH_CFG3=0 timeout 900 python bench.py --steps 5 --warmup 2 --cpu-baseline none > gpurun_out/r06a_bench_short.json 2> gpurun_out/r06a_bench_short.err; echo "bench rc=$?" ); tail -3 gpurun_out/r06a_bench_short.err
python - <<'PY'
import json
d = json.load(open('gpurun_out/r06a_bench_short.json'))
print('line bytes', len(json.dumps(d)))
print('cfg2', round(d['value'], 4), round(d['ms_per_step'], 1), d['ms_per_step_hip_events'])
print(json.dumps(d['roofline']['workloads'], indent=1))
PY
