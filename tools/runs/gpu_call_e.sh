cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_hip_mmdit.py tests/test_hip_pipeline.py -m gpu -x -q > gpurun_out/r02e_tests.log 2>&1; echo "pytest rc=$?" )
tail -2 gpurun_out/r02e_tests.log
for sq in 0 1; do
( FK_SPLIT_QKV=$sq timeout 600 python bench.py --steps 3 --warmup 1 --cpu-baseline none --no-extra > gpurun_out/r02e_bench_split$sq.json 2> gpurun_out/r02e_bench_split$sq.err; echo "bench split=$sq rc=$?" )
python - <<PY
import json
d=json.load(open('gpurun_out/r02e_bench_split$sq.json'))
r=d['roofline']
print('split=$sq cfg2', d['value'], 'gemm', r['achieved'], r['ms_per_edit'], 'launches', r['launches_per_edit'], 'attn', r['other_kernels']['attention']['tflops'])
PY
done
