"""Data-parallel batch generation: one process per GPU, strided shard, one final exchange.

Counterpart of the reference's torchrun eval generators (``univa/eval/gedit/step1_gen_samples.py:82-92``
init, ``:239`` ``inference_list[rank::world_size]``): every rank holds a full model replica and edits an
independent strided shard -- there is no collective inside the 28 denoise steps.  The single real
exchange of the path is an all-gather of the finished packed latents ([B,S_tgt,64] bf16 per rank:
0.5 MB per 1024^2 image), issued as ONE ``all_gather_into_tensor`` so RCCL can drive all xGMI links.
Backend-agnostic (``nccl`` = RCCL on the GPUs, ``gloo`` in the CPU tests).
"""
import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Mirror of the reference's init_gpu_env: reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*."""
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    return rank, local_rank, world


def shard(items, rank, world):
    """The reference's strided shard: items[rank::world]."""
    return list(items)[rank::world]


def shard_indices(n_items, rank, world):
    return list(range(rank, n_items, world))


def all_gather_latents(latents):
    """[B, S, C] per rank -> [world*B, S, C] on every rank, rank-major (one collective)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return latents
    world = dist.get_world_size()
    x = latents.contiguous()
    out = torch.empty((world * x.shape[0], *x.shape[1:]), device=x.device, dtype=x.dtype)
    if dist.get_backend() == "gloo":  # CPU / smoke-test path: gloo gathers a list of host tensors
        xc = x.cpu()
        parts = [torch.empty_like(xc) for _ in range(world)]
        dist.all_gather(parts, xc)
        return torch.cat(parts, dim=0).to(x.device)
    dist.all_gather_into_tensor(out, x)
    return out


def unshard_order(n_items, world):
    """Permutation that maps the rank-major gathered order back to the original item order
    (valid when n_items % world == 0, i.e. every rank holds the same number of items)."""
    per = n_items // world
    order = [0] * n_items
    for r in range(world):
        for j in range(per):
            order[r + j * world] = r * per + j
    return order


def generate_sharded(edit_fn, items, rank=None, world=None, gather=True):
    """Run ``edit_fn(item) -> packed latents [1,S,C]`` over this rank's strided shard and (optionally)
    gather all results in the original order on every rank."""
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
        world = dist.get_world_size() if dist.is_initialized() else 1
    mine = shard(items, rank, world)
    outs = [edit_fn(it) for it in mine]
    local = torch.cat(outs, dim=0) if outs else None
    if not gather or world == 1:
        return local
    if len(items) % world != 0:
        raise ValueError("gather needs len(items) divisible by the world size (pad the list)")
    full = all_gather_latents(local)
    return full[torch.tensor(unshard_order(len(items), world), device=full.device)]
