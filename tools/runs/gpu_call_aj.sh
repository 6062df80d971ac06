# A/B: 256 x 128 GEMM with the weight rings one K-tile deeper (FK_GEMM_BN=130) vs the shipped kernel
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out; rm -f gpurun_out/r02aj_*
( FK_GEMM_BN=130 timeout 300 python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "gemm or qkv" > gpurun_out/r02aj_tests.log 2>&1; echo "pytest(130) rc=$?" ); tail -2 gpurun_out/r02aj_tests.log
for v in 128 130; do echo "== FK_GEMM_BN=$v"; FK_GEMM_BN=$v timeout 120 python tools/cold_weights_gemm.py 2>&1 | grep -v amdgpu.ids | sed 's/   vendor.*//' | grep -v "12288x3072\|8704"; done | tee gpurun_out/r02aj_cold.txt
run() { env $1 timeout 150 python bench.py --steps 4 --warmup 1 --no-extra --cpu-baseline none --no-roofline > gpurun_out/r02aj_$2.json 2>/dev/null; }
run FK_X=1 base_1; run FK_GEMM_BN=130 deep_1; run FK_X=1 base_2; run FK_GEMM_BN=130 deep_2
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r02aj_*.json")):
    d = json.load(open(f)); print(f.split("r02aj_")[1], round(d["value"], 4))
PY
