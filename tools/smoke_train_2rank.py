"""Two-rank smoke of the sharded train step on ONE GPU (gloo, both ranks on cuda:0): the multi-rank code path of
`DenoiserTrainStep(sharded=True)` -- bucketed reduce-scatter of the gradients while the backward runs, global-norm
all-reduce, AdamW on this rank's chunks, in-place all-gather of the parameters -- with the HIP kernels doing the arithmetic.
gloo cannot exchange device tensors, so zero.ShardedAdamW stages the collectives through the host here (RCCL, which the
8-GPU runs use, refuses two ranks on one device).  Full-width model of reduced depth so that two replicas + optimiser
state fit one GPU.  NO scaling claim follows from this; it shows the path runs and the ranks end with the same weights.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 tools/smoke_train_2rank.py
"""
import json
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpt_image_edit_amd import flux_spec  # noqa: E402
from gpt_image_edit_amd.projector import HipDenoiseProjector  # noqa: E402
from gpt_image_edit_amd.train_step import DenoiserTrainStep  # noqa: E402
from gpt_image_edit_amd.transformer import HipFluxTransformer2DModel  # noqa: E402

BF = torch.bfloat16
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("gloo")
cfg = dict(flux_spec.FLUX_KONTEXT_CONFIG, num_layers=2, num_single_layers=4)
model = HipFluxTransformer2DModel(cfg, device=dev, init="synthetic", seed=0)          # same weights on both ranks
proj = HipDenoiseProjector(device=dev, init="synthetic", seed=1)
ts = DenoiserTrainStep(model, lr=1e-4, sharded=True, projector=proj, keep_grads=False, bucket_numel=150_000_000)
g = torch.Generator(device=dev).manual_seed(100 + rank)                                  # every rank its own sample
B, h, w = 1, 64, 64
batch = dict(model_input=torch.randn(B, 16, h, w, generator=g, device=dev), cond_latents=torch.randn(B, 16, h, w, generator=g, device=dev),
             noise=torch.randn(B, 16, h, w, generator=g, device=dev), sigmas=torch.rand(B, generator=g, device=dev) * 0.8 + 0.1,
             vlm_hidden=torch.randn(B, 128, 3584, generator=g, device=dev).to(BF),
             prefix_prompt_embeds=torch.randn(B, 128, 4096, generator=g, device=dev).to(BF),
             pooled=torch.randn(B, 768, generator=g, device=dev).to(BF))
losses, norms = [], []
torch.cuda.synchronize()
dist.barrier()
t0 = time.perf_counter()
for step in range(3):
    out = ts.step(**batch)
    losses.append(float(out["loss"].item()))
    norms.append(float(out["grad_sumsq"].sqrt().item()))
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
# every rank must hold the same parameters after the all-gathers: compare a checksum of the flat buffer
chk = ts.opt.flat_param.float().double().sum().cpu()
both = [torch.zeros_like(chk) for _ in range(world)]
dist.all_gather(both, chk)
same = all(bool(b == both[0]) for b in both)
nrm = torch.tensor(norms, dtype=torch.float64)
all_n = [torch.zeros_like(nrm) for _ in range(world)]
dist.all_gather(all_n, nrm)
if rank == 0:
    print(json.dumps({"what": "2-rank DenoiserTrainStep(sharded=True) smoke, both ranks on one GPU, gloo with host-staged collectives",
                      "world": world, "blocks": "2 double + 4 single (full width)", "seq_len": 256 + 2 * 1024,
                      "zero2_buckets": len(ts.opt.layout.buckets), "trainable_params": sum(int(v.numel()) for v in ts.opt.params.values()),
                      "ms_per_step": dt * 1e3, "loss_rank0": losses, "global_grad_norm": norms,
                      "grad_norm_equal_on_all_ranks": bool(all(torch.equal(a, all_n[0]) for a in all_n)),
                      "parameters_equal_on_all_ranks": same, "losses_decrease_or_move": losses[0] != losses[-1]}), flush=True)
assert same, "ranks ended with different parameters"
dist.barrier()
dist.destroy_process_group()
