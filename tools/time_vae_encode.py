"""Times HipAutoencoderKL.encode at 1024^2 (and 512^2): the bf16 encoder, the fp32-class encoder with bf16 parameters (two-term
products) and with an fp32 checkpoint (three-term products); prints one line per case."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpt_image_edit_amd import flux_spec  # noqa: E402
from gpt_image_edit_amd.vae import HipAutoencoderKL  # noqa: E402

dev = torch.device("cuda", 0)
sd = flux_spec.synthetic_state(flux_spec.vae_param_shapes(), seed=2, device=dev, dtype=torch.float32)
v2 = HipAutoencoderKL(device=dev)
v2.load_state_dict({k: v.to(torch.bfloat16) for k, v in sd.items()})
v3 = HipAutoencoderKL(device=dev)
v3.load_fp32_state_dict(sd)


def timed(fn, n=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for side in (512, 1024):
    x = torch.rand(1, 3, side, side, device=dev) * 2 - 1
    t_bf = timed(lambda: v2.encode(x))
    t_2 = timed(lambda: v2.encode(x, fp32=True))
    t_3 = timed(lambda: v3.encode(x, fp32=True))
    a = v3.encode(x, fp32=True).latent_dist.mode()
    b = v3.encode(x).latent_dist.mode().float()
    print(f"encode {side}^2: bf16 {t_bf:.2f} ms | fp32-class, bf16 weights (2 terms) {t_2:.2f} ms | fp32-class, fp32 checkpoint "
          f"(3 terms) {t_3:.2f} ms | max |bf16 - fp32-class| {float((a - b).abs().max()):.4f} of {float(a.abs().max()):.3f}; "
          f"peak memory {torch.cuda.max_memory_allocated(dev) / 1e9:.2f} GB", flush=True)
