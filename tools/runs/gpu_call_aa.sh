# A/B: single-block MLP-up GEMM on a second stream (FK_OVERLAP_MLP=1) vs everything on one stream (0)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out; rm -f gpurun_out/r02aa_*
( timeout 600 python -m pytest tests/test_hip_mmdit.py tests/test_hip_pipeline.py tests/test_hip_cfg3.py -x -q -m gpu > gpurun_out/r02aa_tests.log 2>&1; echo "pytest rc=$?" ); tail -2 gpurun_out/r02aa_tests.log
run() { FK_OVERLAP_MLP=$1 timeout 200 python bench.py --workload $2 --steps $3 --warmup 1 --no-extra --cpu-baseline none > gpurun_out/r02aa_$2_ov$1_$4.json 2>/dev/null; }
for i in 1 2; do
  run 0 cfg2_single_512x512_28step 4 $i; run 1 cfg2_single_512x512_28step 4 $i
done
run 0 single_1024x1024_28step 2 1; run 1 single_1024x1024_28step 2 1
run 0 cfg2cli_512x512_cond1mp_28step 2 1; run 1 cfg2cli_512x512_cond1mp_28step 2 1
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r02aa_*.json")):
    try:
        d = json.load(open(f)); r = d["roofline"]
        print(f.split("r02aa_")[1], round(d["value"], 4), "gemm", round(r["achieved"]), "attn", round(r["other_kernels"]["attention"]["tflops"]))
    except Exception as e:
        print(f, "failed", e)
PY
