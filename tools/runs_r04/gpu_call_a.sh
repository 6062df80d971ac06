# Round 4, call A: the new parity tests (full depth, 28 steps, fp32-output hot kernels, graph capture, ADVICE fixes) with
# their [parity] lines, then the cfg 2 edit eager vs graph-captured.
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -m gpu -x -q -s > gpurun_out/r04a_tests.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r04a_tests.log ); tail -5 gpurun_out/r04a_tests.log
grep -h "\[parity\] full-depth\|\[parity\] floor growth\|d19s38\|28-step\|hot gemm\|halo f32" gpurun_out/r04a_tests.log | cut -c1-220 | head -80
( FK_GRAPH=0 timeout 400 python bench.py --steps 3 --warmup 1 --no-extra --cpu-baseline none > gpurun_out/r04a_bench_eager.json 2> gpurun_out/r04a_bench_eager.err; echo "bench eager rc=$?" )
( FK_GRAPH=1 timeout 400 python bench.py --steps 3 --warmup 1 --no-extra --cpu-baseline none > gpurun_out/r04a_bench_graph.json 2> gpurun_out/r04a_bench_graph.err; echo "bench graph rc=$?" )
python - <<'PY'
import json
for f in ('gpurun_out/r04a_bench_eager.json','gpurun_out/r04a_bench_graph.json'):
    try:
        d=json.load(open(f)); r=d['roofline']
        print(f, d['value'], d['ms_per_step'], 'host', d['host'], 'gemm', r['achieved'], r['frac'], 'attn', r['other_kernels']['attention']['tflops'])
    except Exception as e: print(f, 'ERR', e); print(open(f.replace('.json','.err')).read()[-1500:])
PY
