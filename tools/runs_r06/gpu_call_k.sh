# Round 6, call K: fused hd = 512 mid-block attention of the VAE: parity tests, decode / encode timing at 1024^2 fused vs three launches
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_hip_vae.py -m gpu -q -s 2>&1 | grep -E "parity|passed|failed|Error|error" | tail -40 ) > gpurun_out/r06k_vae_tests.log 2>&1; tail -25 gpurun_out/r06k_vae_tests.log
python - > gpurun_out/r06k_vae_attention_ab.txt 2>&1 <<'PY'
import os, sys, torch
sys.path.insert(0, os.getcwd())
from gpt_image_edit_amd import vae as hv
BF = torch.bfloat16
vae = hv.HipAutoencoderKL(device="cuda", init="synthetic", seed=1)
g = torch.Generator(device="cuda").manual_seed(0)
for side in (512, 1024):
    z = torch.randn(1, 16, side // 8, side // 8, generator=g, device="cuda").to(BF)
    img = (torch.rand(1, 3, side, side, generator=g, device="cuda") * 2 - 1)
    for fused in (True, False, True, False):
        hv.FUSED_MID_ATTENTION = fused
        for name, fn in (("decode", lambda: vae.decode(z, return_dict=False)), ("encode", lambda: vae.encode(img))):
            fn(); torch.cuda.synchronize()
            torch.cuda.reset_peak_memory_stats(); base = torch.cuda.memory_allocated()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): fn()
            e1.record(); e1.synchronize()
            print(f"{side}^2 {name} mid attention {'fused      ' if fused else 'three-launch'}: {e0.elapsed_time(e1) / 5:7.3f} ms   peak new memory {(torch.cuda.max_memory_allocated() - base) / 2**20:8.1f} MiB", flush=True)
PY
cat gpurun_out/r06k_vae_attention_ab.txt
