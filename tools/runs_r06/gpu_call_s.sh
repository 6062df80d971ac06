# Round 6, call S: fused VAE attention with the next tile prefetched into registers (4 waves per workgroup again): tests + A/B
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_hip_vae.py -m gpu -q 2>&1 | tail -3 )
python - > gpurun_out/r06s_vae_attention_ab.txt 2>&1 <<'PY'
import os, sys, torch
sys.path.insert(0, os.getcwd())
from gpt_image_edit_amd import vae as hv, ops
BF = torch.bfloat16
vae = hv.HipAutoencoderKL(device="cuda", init="synthetic", seed=1)
g = torch.Generator(device="cuda").manual_seed(0)
for S in (4096, 16384):
    qkv = torch.randn(1, S, 1536, generator=g, device="cuda").to(BF)
    o = ops.attention_hd512(qkv[:, :, :512], qkv[:, :, 512:1024], qkv[:, :, 1024:])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.attention_hd512(qkv[:, :, :512], qkv[:, :, 512:1024], qkv[:, :, 1024:], out=o)
    e1.record(); e1.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"attention_hd512 B1 S{S}: {ms * 1e3:8.1f} us  {4.0 * S * S * 512 / ms / 1e9:6.0f} TF/s", flush=True)
for side in (512, 1024):
    z = torch.randn(1, 16, side // 8, side // 8, generator=g, device="cuda").to(BF)
    img = (torch.rand(1, 3, side, side, generator=g, device="cuda") * 2 - 1)
    for fused in (True, False, True, False):
        hv.FUSED_MID_ATTENTION = fused
        for name, fn in (("decode", lambda: vae.decode(z, return_dict=False)), ("encode", lambda: vae.encode(img))):
            fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): fn()
            e1.record(); e1.synchronize()
            print(f"{side}^2 {name} mid attention {'fused      ' if fused else 'three-launch'}: {e0.elapsed_time(e1) / 5:7.3f} ms", flush=True)
PY
grep -v amdgpu.ids gpurun_out/r06s_vae_attention_ab.txt
