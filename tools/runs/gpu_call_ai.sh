cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out; rm -f gpurun_out/r02ai_*
run() { FK_MLP_FIRST=$1 timeout 150 python bench.py --steps 5 --warmup 1 --no-extra --cpu-baseline none --no-roofline > gpurun_out/r02ai_first$1_$2.json 2>/dev/null; }
for i in 1 2 3; do run 0 $i; run 1 $i; done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r02ai_*.json")):
    d = json.load(open(f)); print(f.split("r02ai_")[1], round(d["value"], 4))
PY
