cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out; rm -f gpurun_out/r02x_*
for gm in 8 4 5 2; do
  echo "== FK_GROUP_M=$gm" | tee -a gpurun_out/r02x_groupm.txt
  FK_GROUP_M=$gm timeout 200 python tools/cold_weights_gemm.py 2>&1 | grep -v amdgpu.ids | sed 's/   vendor.*//' | tee -a gpurun_out/r02x_groupm.txt
done
run() { FK_GROUP_M=$1 timeout 150 python bench.py --steps 4 --warmup 1 --no-extra --cpu-baseline none > gpurun_out/r02x_cfg2_gm$1_$2.json 2>/dev/null; }
for i in 1 2; do run 8 $i; run 4 $i; done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r02x_cfg2*.json")):
    d = json.load(open(f)); r = d["roofline"]
    print(f.split("r02x_")[1], round(d["value"], 4), "gemm", round(r["achieved"]), "attn", round(r["other_kernels"]["attention"]["tflops"]))
PY
