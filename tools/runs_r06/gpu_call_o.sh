# Round 6, call O: dry run of the whole bench line (every extra, cfg 3 shortened to one batch, CPU baseline on 1 + 1 blocks) on the final library
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( FK_BENCH_CFG3_STEPS=1,0 timeout 1500 python bench.py --steps 3 --warmup 1 --cpu-baseline blocks > gpurun_out/r06o_bench_dry.json 2> gpurun_out/r06o_bench_dry.err; echo "bench rc=$?" ); tail -3 gpurun_out/r06o_bench_dry.err
python - <<'PY'
import json
d = json.load(open('gpurun_out/r06o_bench_dry.json'))
print('line bytes', len(json.dumps(d)))
print('cfg2', round(d['value'], 4), round(d['ms_per_step'], 1))
print(json.dumps(d['roofline']['workloads'], indent=1))
print(json.dumps(d['roofline']['other_kernels'], indent=1))
print({k: (v if not isinstance(v, dict) else '...') for k, v in d['extra'].get('cfg5_train_step_1024x1024_bs1', {}).items()})
PY
