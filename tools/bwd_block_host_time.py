"""Host time to enqueue the backward of ONE MMDiT block from an idle queue (so that no launch waits for queue space): the per-launch
route of backward.py (one ctypes call per launch) against the block-level C entry points (fk_single_block_bwd / fk_double_block_bwd).
`host_work_ms_per_step` of the bench line is thread CPU time of the whole enqueue loop and includes the runtime's spinning on a full
queue (the step is GPU-bound: it does not move with the calling form); this is the work itself.
    python tools/bwd_block_host_time.py        (cfg 5 shape: 1024^2, bs 1, full width, 2 + 2 blocks)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpt_image_edit_amd import backward, flux_spec, training  # noqa: E402
from gpt_image_edit_amd.transformer import HipFluxTransformer2DModel  # noqa: E402

BF = torch.bfloat16
torch.cuda.set_device(0)
cfg = dict(flux_spec.FLUX_KONTEXT_CONFIG, num_layers=2, num_single_layers=2)
model = HipFluxTransformer2DModel(cfg, device="cuda", init="synthetic", seed=0)
names = list(model._pmap.keys())
bw = backward.FluxBackward(model, training.trainable_names(names), store_activations=True)
g = torch.Generator(device="cuda").manual_seed(1)
S_txt, S_img = 512, 8192
from gpt_image_edit_amd.helpers import _prepare_latent_image_ids as ids  # noqa: E402
img_ids = torch.cat([ids(1, 64, 64, "cuda", BF), ids(1, 64, 64, "cuda", BF)])
args = dict(hidden_states=torch.randn(1, S_img, 64, generator=g, device="cuda").to(BF),
            encoder_hidden_states=torch.randn(1, S_txt, 4096, generator=g, device="cuda").to(BF),
            pooled_projections=torch.randn(1, 768, generator=g, device="cuda").to(BF), timestep=torch.tensor([0.5], device="cuda"),
            img_ids=img_ids, txt_ids=torch.zeros(S_txt, 3, device="cuda", dtype=BF), guidance=torch.tensor([1.0], device="cuda"))


def one(kind, api, reps=6):
    out = []
    for _ in range(reps):
        bw.forward(**args)
        sv = bw._saved
        gbuf = bw._b("g", (1, sv.S, model.inner_dim), zero=True)
        gbuf.normal_()
        bw.__dict__["_mod_ready"] = None
        cb = bw._c_backward_ctx(sv, gbuf) if api else None
        assert (cb is not None) == bool(api)
        torch.cuda.synchronize()
        t0, c0 = time.perf_counter(), time.thread_time()
        bg = {}
        if kind == "single":
            (bw._single_backward_c(1, sv, cb, bg) if api else bw._single_backward(1, sv, gbuf, bg))
        else:
            (bw._double_backward_c(1, sv, cb, bg) if api else bw._double_backward(1, sv, gbuf, bg))
        t1, c1 = time.perf_counter(), time.thread_time()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        out.append(((t1 - t0) * 1e6, (c1 - c0) * 1e6, (t2 - t0) * 1e6))
    out.sort()
    return out[len(out) // 2]


for kind, n in (("single", 38), ("double", 19)):
    for api in (0, 1):
        wall, cpu, gpu = one(kind, api)
        print(f"{kind} block backward, {'C entry point ' if api else 'per-launch route'}: host enqueue {wall:7.0f} us (thread CPU {cpu:7.0f} us) for "
              f"{gpu:7.0f} us of GPU work; x {n} blocks = {wall * n / 1e3:6.1f} ms of host time per step", flush=True)
