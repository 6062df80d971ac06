# Round 4, call J: cfg 3 (batch 32 x 1024^2) on the round-4 tree: 1 warm-up batch + 1 timed batch.
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( timeout 900 python bench.py --workload cfg3_batch32_1024x1024_28step --steps 1 --warmup 1 --no-extra --cpu-baseline none > gpurun_out/r04j_bench_cfg3.json 2> gpurun_out/r04j_cfg3.err; echo "cfg3 rc=$?" )
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04j_bench_cfg3.json')); r=d['roofline']
print(d['config']['workload'], d['value'], d['ms_per_step'], 'gemm', r['achieved'], r['frac'], 'attn', r['other_kernels']['attention']['tflops'], 'host', d['host']['enqueue_ms_per_step'])
PY
