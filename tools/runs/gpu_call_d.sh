cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02d_tests.log 2>&1; echo "pytest rc=$?" )
tail -2 gpurun_out/r02d_tests.log
( timeout 600 python bench.py --steps 3 --warmup 1 --cpu-baseline none > gpurun_out/r02d_bench_cfg2.json 2> gpurun_out/r02d_bench_cfg2.err; echo "bench rc=$?" )
python - <<'PY'
import json
d=json.load(open('gpurun_out/r02d_bench_cfg2.json'))
r=d['roofline']; e=d['extra']['single_1024x1024_28step']
print('cfg2', d['value'], 'gemm', r['achieved'], 'attn', r['other_kernels']['attention']['tflops'])
print('1024', e['value'], 'gemm', e['roofline']['achieved'], 'attn', e['roofline']['other_kernels']['attention']['tflops'])
PY
