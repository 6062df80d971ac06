# Round 6, call U: the K-major GEMM forms (layouts 1 / 2: data and weight gradients) on v_mfma_f32_16x16x32_bf16: parity on both
# shapes, the layout A/B per shape, and the cfg 5 step with FK_KMAJOR_MFMA=32 (rounds 3-5) against the new default, alternating.
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests/test_hip_kernels.py tests/test_hip_train_step.py tests/test_hip_backward.py tests/test_hip_cfg5.py tests/test_hip_train_seam.py tests/test_hip_training.py -m gpu -q -x 2>&1 | tail -4 )
for KM in 32 16 32 16; do
FK_KMAJOR_MFMA=$KM python - <<'PY'
import os, sys, json, torch
sys.path.insert(0, os.getcwd())
import bench
torch.cuda.set_device(0); torch.zeros(1, device="cuda")
r = bench.train_step_bench(torch.device("cuda", 0), steps=5, warmup=2, e2e=False)
print("FK_KMAJOR_MFMA=" + os.environ["FK_KMAJOR_MFMA"], json.dumps({k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.items() if k in ("ms_per_step", "loss", "peak_memory_gb")}), flush=True)
PY
done > gpurun_out/r06u_cfg5_kmajor_mfma.txt 2>&1
grep -v amdgpu.ids gpurun_out/r06u_cfg5_kmajor_mfma.txt
for KM in 32 16; do echo "== FK_KMAJOR_MFMA=$KM"; FK_KMAJOR_MFMA=$KM timeout 600 python tools/ab_gemm_layouts.py 2>&1 | grep -v amdgpu.ids | tail -14; done > gpurun_out/r06u_gemm_layouts_ab.txt 2>&1
cat gpurun_out/r06u_gemm_layouts_ab.txt
