"""Which torch operators still launch kernels inside the cfg 5 optimisation step: one warm step, then one step under
torch.profiler; prints the operators by device time with their input shapes (the HIP kernels reached through ctypes do not
appear as operators: they are the rest).  `python tools/train_ops_prof.py [rows]`."""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpt_image_edit_amd import flux_spec  # noqa: E402
from gpt_image_edit_amd.projector import HipDenoiseProjector  # noqa: E402
from gpt_image_edit_amd.train_step import DenoiserTrainStep  # noqa: E402
from gpt_image_edit_amd.transformer import HipFluxTransformer2DModel  # noqa: E402

BF = torch.bfloat16
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 40
device = torch.device("cuda", 0)
torch.cuda.set_device(0)
model = HipFluxTransformer2DModel(dict(flux_spec.FLUX_KONTEXT_CONFIG), device=device, init="synthetic", seed=0)
projector = HipDenoiseProjector(device=device, init="synthetic", seed=1)
ts = DenoiserTrainStep(model, sharded=True, projector=projector, keep_grads=False)
g = torch.Generator(device=device).manual_seed(7)
B, h, w, L_vlm, L_t5 = 1, 128, 128, 256, 256
batch = dict(model_input=torch.randn(B, 16, h, w, generator=g, device=device),
             cond_latents=torch.randn(B, 16, h, w, generator=g, device=device),
             noise=torch.randn(B, 16, h, w, generator=g, device=device),
             sigmas=torch.rand(B, generator=g, device=device) * 0.8 + 0.1,
             vlm_hidden=torch.randn(B, L_vlm, 3584, generator=g, device=device).to(BF),
             prefix_prompt_embeds=torch.randn(B, L_t5, 4096, generator=g, device=device).to(BF),
             pooled=torch.randn(B, 768, generator=g, device=device).to(BF))
for _ in range(2):
    ts.step(**batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    ts.step(**batch)
    torch.cuda.synchronize()
ka = prof.key_averages(group_by_input_shape=True, group_by_stack_n=4)
evs = sorted(ka, key=lambda e: -getattr(e, "self_device_time_total", getattr(e, "self_cuda_time_total", 0)))
tot = 0.0
for e in evs[:rows]:
    t = getattr(e, "self_device_time_total", getattr(e, "self_cuda_time_total", 0))
    if t <= 0:
        continue
    tot += t
    stack = " <- ".join(s.split("/")[-1] for s in (e.stack or [])[:4])
    print(f"{t / 1e3:9.3f} ms  x{e.count:5d}  {e.key:40s} {str(e.input_shapes)[:90]:90s} {stack[:160]}")
print(f"listed: {tot / 1e3:.2f} ms")

# ---- where the HOST spends the step: one step under cProfile (no device synchronisation inside), then the wall time ----------
import cProfile  # noqa: E402
import pstats  # noqa: E402
import time  # noqa: E402

torch.cuda.synchronize()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
ts.step(**batch)
pr.disable()
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_wall = time.perf_counter() - t0
print(f"host enqueue {t_host * 1e3:.1f} ms (under cProfile), wall {t_wall * 1e3:.1f} ms")
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
t0 = time.perf_counter()
for _ in range(3):
    ts.step(**batch)
t_host = (time.perf_counter() - t0) / 3
torch.cuda.synchronize()
print(f"plain: host enqueue {t_host * 1e3:.1f} ms per step, wall {(time.perf_counter() - t0) / 3 * 1e3:.1f} ms per step")
