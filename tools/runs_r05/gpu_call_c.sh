# Round 5, call C: default MFMA shape = 16 x 16 x 32.  (1) interleaved A/B of both shapes on all 13 GEMM shapes (+ vendor);
# (2) whole-edit A/B FK_GEMM_MFMA=32 / 16, two runs each, same box; (3) the full GPU suite on the new default (no -x: every
# failure listed).
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( AB_VARIANTS=0m32,0m16,256m32,256m16,vendor timeout 600 python tools/ab_gemm_variants.py 3 > gpurun_out/r05c_gemm_mfma_ab.txt 2>&1; echo "ab rc=$?" ); grep -v amdgpu.ids gpurun_out/r05c_gemm_mfma_ab.txt | tail -16
for m in 32 16 32 16; do
  ( FK_GEMM_MFMA=$m timeout 300 python bench.py --steps 3 --warmup 1 --no-extra --cpu-baseline none > gpurun_out/r05c_bench_mfma${m}_$RANDOM.json 2> gpurun_out/r05c_bench_mfma${m}.err; echo "bench mfma$m rc=$?" )
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r05c_bench_mfma*.json')):
    try:
        d = json.load(open(f)); r = d['roofline']
        print(f, 'images/s', round(d['value'], 4), 'ms', round(d['ms_per_step'], 1), d['ms_per_step_hip_events'], 'gemm', round(r['achieved'], 1), 'attn', r['other_kernels']['attention']['tflops'], 'line bytes', len(json.dumps(d)))
    except Exception as e:
        print(f, 'unreadable', e)
PY
( timeout 1500 python -m pytest tests -m gpu -q -s > gpurun_out/r05c_tests.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r05c_tests.log )
grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/r05c_tests.log | tail -30
