# in-edit A/B of the attention main-loop instruction order (FK_ATTN_ILV=0 sequential phases, 1 interleaved)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out; rm -f gpurun_out/r02v_*
( timeout 120 python tools/ab_attention.py default 2>&1 | grep -v amdgpu.ids | head -3 )
run() { FK_ATTN_ILV=$1 timeout 150 python bench.py --workload $2 --steps $3 --warmup 1 --no-extra --cpu-baseline none > gpurun_out/r02v_$2_ilv$1_$4.json 2>/dev/null; }
for i in 1 2; do
  run 0 cfg2_single_512x512_28step 4 $i; run 1 cfg2_single_512x512_28step 4 $i
done
run 0 single_1024x1024_28step 2 1; run 1 single_1024x1024_28step 2 1
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r02v_*.json")):
    try:
        d = json.load(open(f)); r = d["roofline"]
        print(f.split("r02v_")[1], round(d["value"], 4), "gemm", round(r["achieved"]), "attn", round(r["other_kernels"]["attention"]["tflops"]))
    except Exception as e:
        print(f, "failed", e)
PY
