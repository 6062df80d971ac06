"""Optimiser-state sharding for the denoiser's training step (cfg 5): ZeRO-2 over RCCL.

The reference trains under DeepSpeed ZeRO stage 2 (``scripts/accelerate_configs/zero2.json`` via accelerate,
``train_denoiser.py:707-760``): every rank holds the full bf16 weights, the fp32 master copy and the two Adam moments
of the trainable subset are partitioned over the ranks, gradients are reduce-scattered in fp32 in buckets of 5.4e8
elements WHILE the backward pass runs (``overlap_comm``) and the updated parameters all-gathered in bf16 -- 16.2 GB +
8.1 GB per step for the 4.04 B trainable parameters.

This module is that exchange written for one process per GPU over ``torch.distributed`` (``nccl`` = RCCL on the GPUs,
``gloo`` in the CPU tests), around the HIP kernels of ``csrc/train_kernels.hip``:

  * The trainable tensors are laid out in the ORDER THE BACKWARD PASS PRODUCES THEIR GRADIENTS (last block first) and cut
    into buckets (default 5.4e8 elements like the reference; xGMI rings are per-link bound, so buckets stay large).
    A bucket is ``world`` equal chunks; rank r owns chunk r of every bucket (fp32 master + both moments).
  * ``accumulate(grads)`` -- called by the backward pass after every block -- casts the block's gradients into the
    bucket's fp32 staging buffer; the moment a bucket is complete its ``reduce_scatter_tensor(SUM)`` is issued
    asynchronously, so it runs under the backward of the earlier blocks.  Two staging buffers alternate; a rank keeps
    only ITS chunk of every reduced bucket (gradient memory: 2 buckets + total / world, not the whole set).
  * ``step()``: wait for the outstanding reductions; ``fk_sumsq`` of the local chunks + one scalar all-reduce = the global
    norm (``accelerator.clip_grad_norm_``, ``train_denoiser.py:1171-1177``); ``fk_adamw_step_scaled`` per bucket chunk
    -- the 1 / world of the gradient MEAN is folded into the clipping coefficient, no pass divides the sums -- writing
    the bf16 chunk straight into its place in the flat parameter buffer; ``all_gather_into_tensor`` per bucket, IN
    PLACE (the send buffer is the rank's chunk of the receive buffer; nothing is cloned).

The tensors the forward pass reads are views of the flat bf16 buffer.  The arithmetic is injected (``kernels``): the
default is ``gpt_image_edit_amd.ops`` (HIP, no fallback); the world-size-2 CPU tests pass a torch stand-in of their own.
Unmeasured on hardware so far: no multi-GPU node was available to the builder.
"""
import torch
import torch.distributed as dist

__all__ = ["FlatLayout", "ShardedAdamW", "backward_order"]

ALIGN = 64                      # elements: every chunk starts on a 256-byte (fp32) boundary
DEFAULT_BUCKET = 540_000_000    # zero2.json: reduce_bucket_size 5.4e8


def backward_order(names):
    """Names sorted by when ``backward.FluxBackward.backward`` finishes their gradients: single blocks 37 .. 0, double
    blocks 18 .. 0, then everything else (the ``denoise_projector``, fed by the gradient of ``prompt_embeds``)."""
    def inner(n):
        # q, k, v of one projection side by side in that order (the three biases first, then the three weights -- "bias"
        # sorts before "weight"; each trio is contiguous, which is all the aliasing needs): in the flat buffer they ARE
        # the fused [3D] / [3D, D] operands of the QKV GEMM, which the model then aliases instead of re-packing every step
        for trio, tag in ((("to_q", "to_k", "to_v"), "attn.0qkv"), (("add_q_proj", "add_k_proj", "add_v_proj"), "attn.0qkv_added")):
            for i, t in enumerate(trio):
                for wb in ("weight", "bias"):
                    if n.endswith(f"attn.{t}.{wb}"):
                        return (tag, wb, i)
        return (n, "", 0)

    def key(n):
        parts = n.split(".")
        if parts[0] == "single_transformer_blocks":
            return (0, -int(parts[1])) + inner(n)
        if parts[0] == "transformer_blocks":
            return (1, -int(parts[1])) + inner(n)
        return (2, 0) + inner(n)
    return sorted(names, key=key)


class FlatLayout:
    """Packing of a set of tensors into one flat buffer of buckets, each bucket ``world`` equal aligned chunks."""

    def __init__(self, shapes, world, order=None, bucket_numel=None):
        self.names = list(order) if order is not None else sorted(shapes)
        if sorted(self.names) != sorted(shapes):
            raise ValueError("order must be a permutation of the tensor names")
        self.world = world
        limit = bucket_numel if bucket_numel else float("inf")
        groups, cur, cur_n = [], [], 0
        for n in self.names:
            numel = 1
            for d in shapes[n]:
                numel *= int(d)
            if cur and cur_n + numel > limit:
                groups.append(cur)
                cur, cur_n = [], 0
            cur.append((n, numel, tuple(shapes[n])))
            cur_n += numel
        if cur:
            groups.append(cur)
        self.offsets, self.bucket_of, self.buckets = {}, {}, []
        off = state_off = used = 0
        for b, grp in enumerate(groups):
            raw = sum(k for _, k, _ in grp)
            chunk = -(-(-(-raw // world)) // ALIGN) * ALIGN              # ceil(raw / world) rounded up to ALIGN
            o = off
            for n, k, shape in grp:
                self.offsets[n] = (o, k, shape)
                self.bucket_of[n] = b
                o += k
            self.buckets.append(dict(offset=off, size=chunk * world, chunk=chunk, state_offset=state_off,
                                     names=[n for n, _, _ in grp], used=raw))
            off += chunk * world
            state_off += chunk
            used += raw
        self.used, self.total, self.slice_numel = used, off, state_off
        self.max_bucket = max(b["size"] for b in self.buckets)

    def views(self, flat):
        """dict name -> view of ``flat`` with the tensor's shape."""
        return {n: flat[o:o + k].view(shape) for n, (o, k, shape) in self.offsets.items()}

    def chunk_of(self, flat, b, rank):
        bk = self.buckets[b]
        return flat[bk["offset"] + rank * bk["chunk"]: bk["offset"] + (rank + 1) * bk["chunk"]]

    def bucket_view(self, flat, b):
        bk = self.buckets[b]
        return flat[bk["offset"]: bk["offset"] + bk["size"]]


class _Done:
    """Work handle of a collective that has already completed (host-staged exchange)."""

    def wait(self):
        return True


class ShardedAdamW:
    def __init__(self, params, lr=1e-6, betas=(0.9, 0.99), eps=1e-8, weight_decay=0.0, max_grad_norm=1.0, kernels=None,
                 group=None, order=None, bucket_numel=DEFAULT_BUCKET, stage_always=False, average_micro_batches=True):
        """``params``: dict name -> bf16 tensor (the trainable subset, e.g. ``training.trainable_names``).  After
        construction ``self.params`` holds views of ONE flat bf16 buffer that replace them in the model.  ``order``:
        the names in the order their gradients become available (``backward_order``); default: sorted.
        ``stage_always``: keep the staging buffers on one rank too (tests of the multi-rank intake on one process).
        ``average_micro_batches``: with gradient accumulation (``begin_micro_batch`` between backward passes, the
        reference's ``gradient_accumulation_steps``) the step uses the MEAN over the micro-batches, as
        ``accelerator.backward`` does by dividing every loss by the accumulation count; False: their sum."""
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
        self.layout = L = FlatLayout({n: p.shape for n, p in params.items()}, self.world, order=order, bucket_numel=bucket_numel)
        self.flat_param = torch.zeros(L.total, dtype=torch.bfloat16, device=dev)
        self.params = L.views(self.flat_param)
        for n, p in params.items():
            self.params[n].copy_(p)
        # this rank's chunk of every bucket: fp32 master + both moments + the reduced gradient
        self.master = torch.cat([L.chunk_of(self.flat_param, b, self.rank).float() for b in range(len(L.buckets))]).contiguous()
        self.exp_avg = torch.zeros_like(self.master)
        self.exp_avg_sq = torch.zeros_like(self.master)
        self.grad_slice = torch.zeros_like(self.master)
        # two alternating fp32 staging buffers: bucket b fills while bucket b - 1 is on the wire.  One rank: nothing goes on a
        # wire and a bucket's chunk is the bucket -- gradients are cast straight into `grad_slice`, no staging, no copy
        self.direct = self.world == 1 and not stage_always
        n_stage = 0 if self.direct else min(2, len(L.buckets))
        self.staging = [torch.zeros(L.max_bucket, dtype=torch.float32, device=dev) for _ in range(n_stage)]
        self.step_count = 0
        self.last_grad_norm = None
        self.average_micro_batches = average_micro_batches
        # gloo cannot exchange device tensors: with it and parameters on a GPU (two ranks SHARING one GPU -- the only way to
        # run the multi-rank code path on a one-GPU box; RCCL refuses duplicate devices) every collective is staged through
        # the host, synchronously.  nccl (= RCCL) and the CPU tests take the direct path.
        self._host_staged = self.world > 1 and dev.type == "cuda" and dist.get_backend(group) == "gloo"
        self._acc_tmp = []            # world > 1, micro-batch >= 2: reduce-scatter targets that are then ADDED to grad_slice
        self._begin()

    # ---- gradient intake ---------------------------------------------------------------------------------------------
    def _begin(self):
        self._micro = 0            # micro-batches whose gradients are already part of grad_slice
        self._begin_pass()

    def _begin_pass(self):
        self._seen = [set() for _ in self.layout.buckets]
        self._launched = [False] * len(self.layout.buckets)
        self._work = {}            # bucket -> (async work handle of its reduce-scatter, accumulation temp or None)
        self._stage_owner = [None] * len(self.staging)

    def begin_micro_batch(self):
        """Call before every backward pass.  The first one of a step is a no-op; a later one (gradient accumulation, the
        reference's ``gradient_accumulation_steps`` > 1) closes the previous pass -- buckets it left incomplete are
        reduced with zeros for the tensors that got no gradient -- and makes the next pass ADD to this rank's gradient
        chunks instead of overwriting them (DeepSpeed ZeRO-2 likewise reduces every micro-batch into the partition)."""
        if not any(self._seen) and not any(self._launched):
            return
        self._flush()
        self._micro += 1
        self._begin_pass()

    def _finish(self, b):
        """Wait for bucket b's reduce-scatter; an accumulating pass then adds its temporary into the gradient chunk."""
        w = self._work.pop(b, None)
        if w is None:
            return
        handle, tmp = w
        if handle is not None:
            handle.wait()
        if tmp is not None:
            bk = self.layout.buckets[b]
            self.grad_slice[bk["state_offset"]: bk["state_offset"] + bk["chunk"]].add_(tmp[: bk["chunk"]])

    def _flush(self):
        for b in range(len(self.layout.buckets)):        # buckets with tensors that received no gradient this pass (zeros)
            if not self._launched[b]:
                self._stage_for(b)
                self._reduce(b)
        for b in sorted(self._work):
            self._finish(b)

    def _stage_for(self, b):
        """Staging buffer of bucket b (waits for the reduction of the bucket that used it before)."""
        if self.direct:
            bk = self.layout.buckets[b]
            return self.grad_slice[bk["state_offset"]: bk["state_offset"] + bk["chunk"]]
        slot = b % len(self.staging)
        owner = self._stage_owner[slot]
        if owner != b:
            if owner is not None:
                if not self._launched[owner]:
                    # an older bucket that will never complete in this pass (a trainable tensor without a gradient, a
                    # custom arrival order): reduce it now, unseen tensors as zeros -- what step() does at the end anyway
                    self._reduce(owner)
                self._finish(owner)
            self._stage_owner[slot] = b
            self.staging[slot][: self.layout.buckets[b]["size"]].zero_()       # padding and absent tensors count as zero
        return self.staging[slot]

    def grad_view(self, name):
        """fp32 view (the tensor's shape) inside the staging buffer of the tensor's bucket."""
        L = self.layout
        b = L.bucket_of[name]
        o, k, shape = L.offsets[name]
        lo = o - L.buckets[b]["offset"]
        return self._stage_for(b)[lo: lo + k].view(shape)

    def accumulate(self, grads):
        """Take a block's gradients (dict name -> bf16 / fp32 tensor; a cast, no arithmetic) and start the reduction of
        every bucket they complete."""
        L = self.layout
        for n in sorted((n for n in grads if n in L.offsets), key=lambda n: L.offsets[n][0]):
            b = L.bucket_of[n]
            if self._launched[b]:
                raise RuntimeError(f"gradient of {n} arrived after its bucket was reduced: a second backward pass before "
                                   "step() must be announced with begin_micro_batch() (gradient accumulation), and within a "
                                   "pass a tensor's gradient may arrive only once per bucket flush")
            if self.direct and self._micro:                   # accumulating pass on one rank: straight into the chunk
                if n in self._seen[b]:
                    raise RuntimeError(f"gradient of {n} arrived twice in one accumulating pass")
                self.grad_view(n).add_(grads[n])
            else:
                self.grad_view(n).copy_(grads[n])
            self._seen[b].add(n)
            if len(self._seen[b]) == len(L.buckets[b]["names"]):
                self._reduce(b)

    def _reduce(self, b):
        bk = self.layout.buckets[b]
        stage = self._stage_for(b)[: bk["size"]]
        dst = self.grad_slice[bk["state_offset"]: bk["state_offset"] + bk["chunk"]]
        if self.world > 1:
            tmp = None
            if self._micro:                                   # accumulating pass: reduce into a temporary, add on completion
                while len(self._acc_tmp) < 2:
                    self._acc_tmp.append(torch.empty(max(k["chunk"] for k in self.layout.buckets), dtype=torch.float32,
                                                     device=self.grad_slice.device))
                other = [t for _, t in self._work.values() if t is not None]
                tmp = next((t for t in self._acc_tmp if all(t is not o for o in other)), None)
                if tmp is None:                               # both temporaries on the wire: finish the older one
                    self._finish(min(k for k, (_, t) in self._work.items() if t is not None))
                    other = [t for _, t in self._work.values() if t is not None]
                    tmp = next(t for t in self._acc_tmp if all(t is not o for o in other))
                dst = tmp[: bk["chunk"]]
            self._work[b] = (self._reduce_scatter(dst, stage), tmp)
        elif self.direct:   # `stage` IS `dst`; tensors that received no gradient this step count as zero (the padding never changes)
            if not self._micro:
                for n in bk["names"]:
                    if n not in self._seen[b]:
                        self.grad_view(n).zero_()
        elif self._micro:
            dst.add_(stage[: bk["chunk"]])
        else:
            dst.copy_(stage[: bk["chunk"]])
        self._launched[b] = True

    def _reduce_scatter(self, dst, src):
        if not self._host_staged:
            return dist.reduce_scatter_tensor(dst, src, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        h_src, h_dst = src.cpu(), torch.empty(dst.shape, dtype=dst.dtype)
        dist.reduce_scatter_tensor(h_dst, h_src, op=dist.ReduceOp.SUM, group=self.group)
        dst.copy_(h_dst)
        return _Done()

    def _all_gather(self, out, mine):
        if not self._host_staged:
            return dist.all_gather_into_tensor(out, mine, group=self.group, async_op=True)
        h_out = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(h_out, mine.cpu(), group=self.group)
        out.copy_(h_out)
        return _Done()

    # ---- compatibility: whole-gradient views (tests, small models) ------------------------------------------------------
    @property
    def grads(self):
        """dict name -> fp32 staging view; writing all of them and calling ``step()`` is the unbucketed use.  Only valid
        when everything fits the staging buffers (at most two buckets)."""
        if not self.direct and len(self.layout.buckets) > len(self.staging):
            raise RuntimeError("grads views need <= 2 buckets; feed gradients through accumulate()")
        for b, bk in enumerate(self.layout.buckets):     # the caller writes all of them
            self._seen[b].update(bk["names"])
        return {n: self.grad_view(n) for n in self.layout.names}

    # ---- skipped steps, save / resume ------------------------------------------------------------------------------------
    def zero_grad(self):
        """Discard every gradient taken in since the last ``step()`` (a backward pass whose step is deliberately skipped, e.g.
        a non-finite loss -- the reference's ``optimizer.zero_grad()``, train_denoiser.py:1180): reductions in flight are
        finished first, then the chunks and the micro-batch count are reset.  No NEW collective is started: buckets whose
        reduce-scatter has not been launched are simply dropped, so the call costs no traffic and a rank may make it without
        the others entering a collective for it.  What the ranks must still agree on is the DECISION to skip (every rank runs
        the same backward passes, so the launched buckets match): take it on an all-reduced loss / flag, as DeepSpeed's
        overflow check does -- a rank that alone skips ``step()`` leaves the others waiting in step()'s all-gather."""
        for b in sorted(self._work):                      # launched by every rank during the same backward pass: completes
            self._finish(b)
        self.grad_slice.zero_()
        self._begin()

    def layout_signature(self):
        """What a saved shard must agree with to be loadable: world size, tensor names in layout order with their shapes, and
        the bucket cuts (a different ``bucket_numel`` or ``order`` gives other chunks)."""
        L = self.layout
        return dict(world=self.world, names=list(L.names), shapes=[list(L.offsets[n][2]) for n in L.names],
                    chunks=[b["chunk"] for b in L.buckets])

    def state_dict(self):
        """THIS RANK's shard of the optimiser state (``accelerator.save_state`` under ZeRO-2 writes one file per rank too,
        train_denoiser.py:1229): fp32 master chunk, both Adam moments, the step count, hyper-parameters and the layout
        signature.  The bf16 parameters are not part of it -- they are bf16(master) and are rebuilt on load."""
        return dict(version=1, rank=self.rank, signature=self.layout_signature(), step=self.step_count,
                    master=self.master.detach().cpu().clone(), exp_avg=self.exp_avg.detach().cpu().clone(),
                    exp_avg_sq=self.exp_avg_sq.detach().cpu().clone(), hp=dict(self.hp), max_grad_norm=self.max_grad_norm)

    @torch.no_grad()
    def load_state_dict(self, sd):
        """Resume from ``state_dict()`` of the same rank of a run with the same world size and layout (collective: every
        rank calls it with its own shard).  Restores master / moments / step count, drops any gradient in flight, rewrites
        this rank's chunks of the flat bf16 parameters as bf16(master) -- what the last ``step()`` of the saved run left
        there -- and all-gathers them, so the next step continues bit for bit where the saved run stopped
        (``accelerator.load_state``, train_denoiser.py:349-367, 769)."""
        if sd.get("version") != 1:
            raise ValueError("not a ShardedAdamW state dict")
        if sd["signature"] != self.layout_signature():
            raise ValueError("optimiser shard does not fit this layout (world size, trainable set, order or bucket size "
                             f"changed): saved world {sd['signature']['world']} / {len(sd['signature']['names'])} tensors, "
                             f"now world {self.world} / {len(self.layout.names)} tensors; re-sharding is not supported")
        if sd["rank"] != self.rank:
            raise ValueError(f"shard of rank {sd['rank']} loaded on rank {self.rank}")
        self.zero_grad()
        for name in ("master", "exp_avg", "exp_avg_sq"):
            t = sd[name]
            if t.shape != getattr(self, name).shape or t.dtype != torch.float32:
                raise ValueError(f"{name}: saved {tuple(t.shape)} {t.dtype}, expected {tuple(getattr(self, name).shape)} float32")
            getattr(self, name).copy_(t)
        self.step_count = int(sd["step"])
        self.hp = dict(sd["hp"])
        self.max_grad_norm = sd["max_grad_norm"]
        L = self.layout
        works = []
        for b, bk in enumerate(L.buckets):
            mine = L.chunk_of(self.flat_param, b, self.rank)
            mine.copy_(self.master[bk["state_offset"]: bk["state_offset"] + bk["chunk"]])      # fp32 -> bf16, round to nearest even
            if self.world > 1:
                works.append(self._all_gather(L.bucket_view(self.flat_param, b), mine))
        for w in works:
            w.wait()

    @staticmethod
    def shard_file(directory, rank, world):
        import os
        return os.path.join(directory, f"zero2_optim_rank{rank:05d}_of{world:05d}.pt")

    def save(self, directory):
        """One file per rank under ``directory`` (created if missing); returns the path."""
        import os
        os.makedirs(directory, exist_ok=True)
        path = self.shard_file(directory, self.rank, self.world)
        torch.save(self.state_dict(), path)
        return path

    def load(self, directory):
        self.load_state_dict(torch.load(self.shard_file(directory, self.rank, self.world), map_location="cpu", weights_only=False))

    def state_bytes(self):
        """(replicated, sharded) bytes this rank holds for the optimiser: flat bf16 params + fp32 staging | 4 fp32 chunks."""
        return self.layout.total * 2 + sum(s.numel() for s in self.staging) * 4, self.layout.slice_numel * 4 * 4

    @torch.no_grad()
    def step(self):
        """Finish the gradient exchange, update this rank's chunks, refresh ``self.params`` on every rank."""
        L = self.layout
        self._flush()
        sumsq = self.k.sumsq(self.grad_slice)            # fp64 [1] over the SUMS; padding elements are zero
        if self.world > 1:
            if self._host_staged:
                h = sumsq.cpu()
                dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
                sumsq.copy_(h)
            else:
                dist.all_reduce(sumsq, op=dist.ReduceOp.SUM, group=self.group)
        scale = 1.0 / self.world                          # mean over the data-parallel ranks, like DDP / DeepSpeed
        if self.average_micro_batches:
            scale /= self._micro + 1                      # ... and over the accumulated micro-batches (accelerate)
        self.last_grad_norm = sumsq.sqrt() * scale
        self.step_count += 1
        works = []
        for b, bk in enumerate(L.buckets):
            so, ch = bk["state_offset"], bk["chunk"]
            mine = L.chunk_of(self.flat_param, b, self.rank)
            self.k.adamw_step(self.master[so:so + ch], self.grad_slice[so:so + ch], self.exp_avg[so:so + ch],
                              self.exp_avg_sq[so:so + ch], self.step_count,
                              grad_sumsq=sumsq if self.max_grad_norm is not None else None,
                              max_grad_norm=self.max_grad_norm if self.max_grad_norm is not None else 0.0,
                              param_bf16=mine, grad_scale=scale, **self.hp)
            if self.world > 1:   # in place: `mine` IS the rank's chunk of the bucket being gathered
                works.append(self._all_gather(L.bucket_view(self.flat_param, b), mine))
        for w in works:
            w.wait()
        self._begin()
        return self.last_grad_norm
