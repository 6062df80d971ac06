# Round 5, call B: (1) single-wave MFMA cadence probes; (2) GEMM kernel tests incl. the 16 x 16 x 32 form, loss tests;
# (3) interleaved A/B of the MFMA shapes per GEMM shape (+ vendor); (4) whole-edit A/B, FK_GEMM_MFMA=32 vs 16 (same box).
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( timeout 120 tools/build/power_probe 1.2 > gpurun_out/r05b_power_probe.txt 2>&1; echo "probe rc=$?" ); grep "single-wave" gpurun_out/r05b_power_probe.txt
( timeout 900 python -m pytest -x -q -s tests/test_hip_kernels.py -k gemm tests/test_hip_training.py tests/test_hip_train_step.py::test_step_takes_the_stage2_loss_weights > gpurun_out/r05b_tests.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r05b_tests.log )
grep -E "passed|failed|Error|error" gpurun_out/r05b_tests.log | tail -8
( AB_VARIANTS=0,0m16,256,256m16,128,128m16,vendor timeout 600 python tools/ab_gemm_variants.py 3 > gpurun_out/r05b_gemm_mfma_ab.txt 2>&1; echo "ab rc=$?" ); cat gpurun_out/r05b_gemm_mfma_ab.txt | tail -16
for m in 32 16 32 16; do
  ( FK_GEMM_MFMA=$m timeout 300 python bench.py --steps 3 --warmup 1 --no-extra --cpu-baseline none > gpurun_out/r05b_bench_mfma${m}_$RANDOM.json 2> gpurun_out/r05b_bench_mfma${m}.err; echo "bench mfma$m rc=$?" )
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r05b_bench_mfma*.json')):
    try:
        d = json.load(open(f)); r = d['roofline']
        print(f, 'images/s', round(d['value'], 4), 'ms', round(d['ms_per_step'], 1), 'gemm', round(r['achieved'], 1), 'attn', r['other_kernels']['attention']['tflops'])
    except Exception as e:
        print(f, 'unreadable', e)
PY
