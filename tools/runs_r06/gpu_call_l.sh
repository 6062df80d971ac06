# Round 6, call L: block-level backward entry points: bit-identity tests, then the cfg 5 step with and without them (host work, ms)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
for API in 1 0 1 0; do
FK_BWD_BLOCK_API=$API python - <<'PY'
import os, sys, json, torch
sys.path.insert(0, os.getcwd())
import bench
torch.cuda.set_device(0); torch.zeros(1, device="cuda")
r = bench.train_step_bench(torch.device("cuda", 0), steps=5, warmup=2, e2e=False)
print("FK_BWD_BLOCK_API=" + os.environ["FK_BWD_BLOCK_API"], json.dumps({k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.items() if k in ("ms_per_step", "host_enqueue_ms_per_step", "host_work_ms_per_step", "loss", "peak_memory_gb")}), flush=True)
PY
done > gpurun_out/r06l_cfg5_block_api.txt 2>&1
grep -v amdgpu.ids gpurun_out/r06l_cfg5_block_api.txt
