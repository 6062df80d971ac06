# A/B of the fused-QKV epilogue change (DPP row sums, packed rope table): previous commit under _base/ vs this tree
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_hip_kernels.py tests/test_hip_mmdit.py -x -q -m gpu > gpurun_out/r02p_tests.log 2>&1; echo "pytest rc=$?" )
tail -2 gpurun_out/r02p_tests.log
for i in 1 2; do
  ( cd _base && timeout 300 python bench.py --steps 4 --warmup 1 --no-extra --cpu-baseline none > ../gpurun_out/r02p_base_$i.json 2> ../gpurun_out/r02p_base_$i.err; echo "base rc=$?" )
  ( timeout 300 python bench.py --steps 4 --warmup 1 --no-extra --cpu-baseline none > gpurun_out/r02p_new_$i.json 2> gpurun_out/r02p_new_$i.err; echo "new rc=$?" )
done
python - <<'PY'
import json
for n in ("base_1","new_1","base_2","new_2"):
    try:
        d=json.load(open(f"gpurun_out/r02p_{n}.json")); r=d["roofline"]
        print(n, d["value"], d["ms_per_step"], "gemm", r["achieved"], "attn", r["other_kernels"]["attention"]["tflops"])
    except Exception as e:
        print(n, "failed", e)
PY
