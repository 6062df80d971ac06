"""Optimiser-state sharding for the denoiser's training step (cfg 5): ZeRO-2 over RCCL.

The reference trains under DeepSpeed ZeRO stage 2 (``scripts/accelerate_configs/zero2.json`` via accelerate,
``train_denoiser.py:707-760``): every rank holds the full bf16 weights, the fp32 master copy and the two Adam moments
of the trainable subset are partitioned over the ranks, gradients are reduce-scattered in fp32 and the updated
parameters all-gathered in bf16 -- 16.2 GB + 8.1 GB per step for the 4.04 B trainable parameters.

This module is that exchange written for one process per GPU over ``torch.distributed`` (``nccl`` = RCCL on the GPUs,
``gloo`` in the CPU tests), around the HIP kernels of ``csrc/train_kernels.hip``:

    flat fp32 gradients  --reduce_scatter_tensor(SUM)/world-->  this rank's slice
    fk_sumsq(slice) --all_reduce(SUM)--> global ||g||^2        (accelerator.clip_grad_norm_, train_denoiser.py:1171-1177)
    fk_adamw_step(master slice, moments, clip coefficient, bf16 slice)
    bf16 slice  --all_gather_into_tensor-->  flat bf16 parameters (the tensors the forward pass reads are views of it)

Two collectives per step, each ONE call on a contiguous buffer, so RCCL can drive every xGMI link; xGMI rings are
per-link bound, so the buffers are not cut into small buckets.  The arithmetic is injected (``kernels``): the default is
``gpt_image_edit_amd.ops`` (HIP, no fallback); the world-size-2 CPU test passes a torch stand-in of its own.
"""
import torch
import torch.distributed as dist

__all__ = ["FlatLayout", "ShardedAdamW"]

ALIGN = 64   # elements: every slice starts on a 256-byte (fp32) boundary


class FlatLayout:
    """Name-ordered packing of a set of tensors into one flat buffer padded to ``world`` equal, aligned slices."""

    def __init__(self, shapes, world):
        self.names = sorted(shapes)
        self.offsets, off = {}, 0
        for n in self.names:
            numel = 1
            for d in shapes[n]:
                numel *= int(d)
            self.offsets[n] = (off, numel, tuple(shapes[n]))
            off += numel
        self.used = off
        per = -(-off // world)                       # ceil
        self.slice_numel = -(-per // ALIGN) * ALIGN
        self.total = self.slice_numel * world
        self.world = world

    def views(self, flat):
        """dict name -> view of ``flat`` with the tensor's shape."""
        return {n: flat[o:o + k].view(shape) for n, (o, k, shape) in self.offsets.items()}

    def slice_of(self, flat, rank):
        return flat[rank * self.slice_numel:(rank + 1) * self.slice_numel]


class ShardedAdamW:
    def __init__(self, params, lr=1e-6, betas=(0.9, 0.99), eps=1e-8, weight_decay=0.0, max_grad_norm=1.0, kernels=None,
                 group=None):
        """``params``: dict name -> bf16 tensor (the trainable subset, e.g. ``training.trainable_names``).  After
        construction ``self.params`` holds views of ONE flat bf16 buffer that replace them in the model, and
        ``self.grads`` fp32 views of the flat gradient buffer the backward pass accumulates into."""
        self.group = group
        on = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if on else 1
        self.rank = dist.get_rank(group) if on else 0
        if kernels is None:
            from . import ops as kernels   # HIP; raises on CPU tensors
        self.k = kernels
        self.hp = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        self.max_grad_norm = max_grad_norm
        dev = next(iter(params.values())).device
        self.layout = FlatLayout({n: p.shape for n, p in params.items()}, self.world)
        L = self.layout
        self.flat_param = torch.zeros(L.total, dtype=torch.bfloat16, device=dev)
        self.flat_grad = torch.zeros(L.total, dtype=torch.float32, device=dev)
        self.params, self.grads = L.views(self.flat_param), L.views(self.flat_grad)
        for n, p in params.items():
            self.params[n].copy_(p)
        # this rank's slice of the optimiser state: fp32 master + both moments
        self.master = L.slice_of(self.flat_param, self.rank).float().contiguous()
        self.exp_avg = torch.zeros_like(self.master)
        self.exp_avg_sq = torch.zeros_like(self.master)
        self.grad_slice = torch.empty_like(self.master)
        self.step_count = 0
        self.last_grad_norm = None

    def state_bytes(self):
        """(replicated, sharded) bytes this rank holds for the optimiser: flat bf16 params + fp32 grads | 4 fp32 slices."""
        return self.layout.total * (2 + 4), self.layout.slice_numel * 4 * 4

    @torch.no_grad()
    def step(self):
        """Consume ``self.grads`` (this rank's local fp32 gradients), update, refresh ``self.params`` on every rank."""
        L = self.layout
        if self.world > 1:
            dist.reduce_scatter_tensor(self.grad_slice, self.flat_grad, op=dist.ReduceOp.SUM, group=self.group)
            self.grad_slice.mul_(1.0 / self.world)   # mean over the data-parallel ranks, like DDP / DeepSpeed
        else:
            self.grad_slice.copy_(L.slice_of(self.flat_grad, 0))
        sumsq = self.k.sumsq(self.grad_slice)            # fp64 [1]; padding elements are zero
        if self.world > 1:
            dist.all_reduce(sumsq, op=dist.ReduceOp.SUM, group=self.group)
        self.last_grad_norm = sumsq.sqrt()
        self.step_count += 1
        mine = L.slice_of(self.flat_param, self.rank)
        self.k.adamw_step(self.master, self.grad_slice, self.exp_avg, self.exp_avg_sq, self.step_count,
                          grad_sumsq=sumsq if self.max_grad_norm is not None else None,
                          max_grad_norm=self.max_grad_norm if self.max_grad_norm is not None else 0.0,
                          param_bf16=mine, **self.hp)
        if self.world > 1:
            dist.all_gather_into_tensor(self.flat_param, mine.clone(), group=self.group)
        self.flat_grad.zero_()
        return self.last_grad_norm
