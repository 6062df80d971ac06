"""GPU parity of the MMDiT backward pass and of one whole optimisation step against the CPU oracle
(``oracle/train.py``: the reference's step with the MMDiT differentiated by torch autograd).

Full-width model (D = 3072, 24 heads), one double + one single block, small sequences.  Every trainable gradient of the
HIP path is compared with fp32 autograd on the same bf16-rounded weights; the yardstick is how far bf16 autograd (what
the reference runs) is from fp32 autograd on that tensor -- the HIP gradient must be within 2x that floor.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def _rel(a, b):
    return ((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-12)).item()


def _setup(B=2, S_txt=64, h=16, w=16, seed=0):
    from gpt_image_edit_amd import flux_spec, training
    cfg = dict(flux_spec.FLUX_KONTEXT_CONFIG, num_layers=1, num_single_layers=1)
    sd_bf = {k: v.to(BF) for k, v in flux_spec.synthetic_state(flux_spec.flux_param_shapes(cfg), seed=41).items()}
    g = torch.Generator().manual_seed(seed)
    batch = dict(model_input=torch.randn(B, 16, h, w, generator=g), cond_latents=torch.randn(B, 16, h, w, generator=g),
                 noise=torch.randn(B, 16, h, w, generator=g), sigmas=torch.tensor([0.25, 0.75][:B]),
                 prompt_embeds=torch.randn(B, S_txt, 4096, generator=g).to(BF), pooled=torch.randn(B, 768, generator=g).to(BF))
    trainable = training.trainable_names(list(sd_bf.keys()))
    return cfg, sd_bf, batch, trainable


def test_backward_matches_autograd_and_adamw_step():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from gpt_image_edit_amd.train_step import DenoiserTrainStep
    from gpt_image_edit_amd.transformer import HipFluxTransformer2DModel
    from oracle import train as otrain
    cfg, sd_bf, batch, trainable = _setup()
    assert len(trainable) == 12 + 10          # double: q k v out (w + b) + 2 norms + norm1.linear (w + b); single: q k v (w + b) + 2 + 2
    model = HipFluxTransformer2DModel(cfg, device="cuda")
    model.load_state_dict(sd_bf)
    ts = DenoiserTrainStep(model, lr=1e-3)     # large lr so that one step moves bf16 weights visibly
    dev_batch = {k: v.cuda() for k, v in batch.items()}
    loss, grads, d_enc = ts.forward_backward(**dev_batch)
    loss2, grads2, d_enc2 = ts.forward_backward(**dev_batch)
    torch.cuda.synchronize()
    assert set(grads) == set(trainable)
    assert torch.equal(loss, loss2) and torch.equal(d_enc, d_enc2) and all(torch.equal(grads[k], grads2[k]) for k in grads), \
        "backward is not deterministic"
    # oracle: fp32 autograd on the bf16-rounded weights, and bf16 autograd (the reference's own execution)
    sd32 = {k: v.float() for k, v in sd_bf.items()}
    b32 = dict(batch, prompt_embeds=batch["prompt_embeds"].float(), pooled=batch["pooled"].float())
    ref32 = otrain.train_step(sd32, trainable, b32, {}, flux_config=cfg, lr=1e-3)
    refbf = otrain.train_step(sd_bf, trainable, batch, {}, flux_config=cfg, lr=1e-3)
    print(f"loss hip {loss.item():.6f}  fp32-oracle {ref32['loss'].item():.6f}  bf16-oracle {refbf['loss'].item():.6f}")
    assert abs(loss.item() - ref32["loss"].item()) <= max(2 * abs(refbf["loss"].item() - ref32["loss"].item()), 2e-3 * ref32["loss"].item())
    worst = worst_bf = 0.0
    for k in trainable:
        e_hip, e_floor = _rel(grads[k].cpu(), ref32["grads"][k]), _rel(refbf["grads"][k], ref32["grads"][k])
        e_bf = _rel(grads[k].cpu(), refbf["grads"][k])
        worst, worst_bf = max(worst, e_hip / max(e_floor, 1e-3)), max(worst_bf, e_bf / max(e_floor, 1e-3))
        print(f"[grad] {k:50s} hip-vs-fp32 {e_hip:.3e}  bf16-autograd floor {e_floor:.3e}  hip-vs-bf16-autograd {e_bf:.3e}")
        assert e_hip <= max(2.0 * e_floor, 2e-2), f"{k}: HIP gradient {e_hip:.3e} from fp32 autograd, floor {e_floor:.3e}"
        # the forward's bf16 rounding points are shared with the bf16 oracle, so the HIP gradient sits much closer to
        # bf16 autograd than either sits to fp32 autograd
        assert e_bf <= max(0.6 * e_floor, 2e-2), f"{k}: HIP gradient {e_bf:.3e} from bf16 autograd (floor {e_floor:.3e})"
    print(f"worst hip / floor ratio {worst:.2f}; worst (hip vs bf16 autograd) / floor {worst_bf:.2f}")
    # the gradient w.r.t. prompt_embeds -- what the denoise_projector's backward continues from -- against autograd too

    def d_prompt(sd, b):
        pe = b["prompt_embeds"].detach().clone().requires_grad_(True)
        return torch.autograd.grad(otrain.denoiser_loss(sd, **dict(b, prompt_embeds=pe), flux_config=cfg), pe)[0]
    dp32, dpbf = d_prompt(sd32, b32), d_prompt(sd_bf, batch)
    e_hip, e_floor, e_bf = _rel(d_enc.cpu(), dp32), _rel(dpbf, dp32), _rel(d_enc.cpu(), dpbf)
    print(f"[grad] {'d(prompt_embeds)':50s} hip-vs-fp32 {e_hip:.3e}  bf16-autograd floor {e_floor:.3e}  hip-vs-bf16-autograd {e_bf:.3e}")
    assert e_hip <= max(2.0 * e_floor, 2e-2) and e_bf <= max(0.6 * e_floor, 2e-2)
    # the whole step: global-norm clipping + AdamW on fp32 masters, bf16 copies rewritten -- against the oracle's
    # optimiser arithmetic fed the SAME (HIP) gradients
    want_p, _, want_norm = otrain.adamw_step({k: sd32[k] for k in trainable}, {k: grads[k].float().cpu() for k in trainable},
                                             {}, lr=1e-3)
    norm = ts.optimizer_step(grads).sqrt().item()
    assert abs(norm - want_norm.item()) <= 1e-4 * want_norm.item()
    assert abs(norm - ref32["grad_norm"].item()) <= 5e-2 * ref32["grad_norm"].item()
    for k in trainable:
        new = model.p(k).detach().float().cpu()
        assert (new - want_p[k]).abs().max().item() <= 2.0 ** -8 * want_p[k].abs().max().item() + 1e-6, k   # bf16 copy of the fp32 master
        assert (ts.state[k][0].cpu() - want_p[k]).abs().max().item() <= 1e-5, k                              # the master itself
    # and the model keeps running after the update (fused weight copies and transposes are refreshed)
    loss3, _, _ = ts.forward_backward(**dev_batch)
    assert torch.isfinite(loss3).all() and loss3.item() != loss.item()


def test_sharded_optimizer_layout_gives_the_same_step():
    """DenoiserTrainStep(sharded=True) -- parameters as views of one flat bf16 buffer, fp32 flat gradients, ZeRO-2 state
    (zero.ShardedAdamW; one process here) -- must update the model exactly like the per-tensor path."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from gpt_image_edit_amd.train_step import DenoiserTrainStep
    from gpt_image_edit_amd.transformer import HipFluxTransformer2DModel
    cfg, sd_bf, batch, trainable = _setup(B=1, S_txt=40, h=8, w=12, seed=3)
    dev_batch = {k: v.cuda() for k, v in batch.items()}
    out = []
    # also: activations stored (no recomputation) vs one checkpoint per block + recomputation -- same values either way
    for sharded, store in ((False, False), (True, True)):
        model = HipFluxTransformer2DModel(cfg, device="cuda")
        model.load_state_dict(sd_bf)
        model.enable_gradient_checkpointing()          # accepted (train_denoiser.py:486)
        ts = DenoiserTrainStep(model, lr=1e-3, sharded=sharded, store_activations=store)
        r1 = ts.step(**dev_batch)
        r2 = ts.step(**dev_batch)                      # second step runs on the refreshed fused / transposed weights
        out.append((r1["loss"].item(), r2["loss"].item(), {k: model.p(k).detach().clone() for k in trainable}))
    assert out[0][0] == out[1][0] and out[0][1] == out[1][1]
    assert out[0][1] != out[0][0]
    for k in trainable:
        assert torch.equal(out[0][2][k], out[1][2][k]), k


def test_projector_trains_along():
    """denoise_projector (in the reference's trainable set, train_denoiser.py:71-119): forward_train == forward bit for
    bit, its backward against fp32 autograd of the same Sequential on the host, and the train step carrying the gradient
    of prompt_embeds into it (per-tensor and sharded optimiser layouts update it identically)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import torch.nn.functional as F
    from gpt_image_edit_amd.projector import HipDenoiseProjector
    from gpt_image_edit_amd.train_step import DenoiserTrainStep
    from gpt_image_edit_amd.transformer import HipFluxTransformer2DModel
    B, L = 2, 37
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, L, 3584, generator=g).to(BF)
    dy = (torch.randn(B, L, 4096, generator=g) * 0.1).to(BF)
    proj = HipDenoiseProjector(device="cuda", init="synthetic", seed=7)
    sd = {k: v.detach().cpu() for k, v in proj.state_dict().items()}
    y = proj(x.cuda())
    y_train = proj.forward_train(x.cuda())
    assert torch.equal(y, y_train)
    grads = proj.backward(dy.cuda())
    with pytest.raises(RuntimeError):
        proj.backward(dy.cuda())                      # the saved activations are consumed
    w = {k: v.float().requires_grad_(True) for k, v in sd.items()}
    h1 = F.linear(x.float(), w["0.weight"], w["0.bias"])
    ref = F.linear(F.silu(h1), w["2.weight"], w["2.bias"])
    ref.backward(dy.float())
    assert _rel(y.cpu(), ref.detach()) < 1e-2
    for k in sd:
        e = _rel(grads[k].cpu(), w[k].grad)
        print(f"[projector grad] {k:10s} rel {e:.3e}")
        assert e < 1.5e-2, k                          # bf16 activations / bf16 gradient storage vs fp32 autograd
    # inside the step
    cfg, sd_bf, batch, trainable = _setup(B=B, S_txt=24, h=8, w=8, seed=9)
    dev_batch = {k: v.cuda() for k, v in batch.items() if k != "prompt_embeds"}
    dev_batch.update(vlm_hidden=x.cuda(), prefix_prompt_embeds=batch["prompt_embeds"].cuda())
    out = []
    for sharded in (False, True):
        model = HipFluxTransformer2DModel(cfg, device="cuda")
        model.load_state_dict(sd_bf)
        pj = HipDenoiseProjector(device="cuda", init="synthetic", seed=7)
        ts = DenoiserTrainStep(model, lr=1e-3, sharded=sharded, projector=pj)
        assert ts.trainable_names() == set(trainable) | {"denoise_projector." + k for k in sd}
        r = ts.step(**dev_batch)
        assert set(r["grads"]) == ts.trainable_names()
        assert r["d_prompt_embeds"].shape == (B, L + 24, 4096)
        # the projector's share of the step is exactly its backward on the leading L rows of d(prompt_embeds)
        pj2 = HipDenoiseProjector(device="cuda", init="synthetic", seed=7)
        pj2.forward_train(x.cuda())
        g2 = pj2.backward(r["d_prompt_embeds"][:, :L])
        for k in sd:
            assert torch.equal(g2[k], r["grads"]["denoise_projector." + k]), k
            assert r["grads"]["denoise_projector." + k].float().abs().max().item() > 0
        r2 = ts.step(**dev_batch)
        out.append((r["loss"].item(), r2["loss"].item(), {k: pj.p(k).detach().clone() for k in sd}))
        assert any(not torch.equal(out[-1][2][k].cpu(), sd[k]) for k in sd), "the projector did not move"
    assert out[0][:2] == out[1][:2]
    for k in sd:
        assert torch.equal(out[0][2][k], out[1][2][k]), k


def test_k_major_operands_give_the_same_gradients(monkeypatch):
    """backward.K_MAJOR 0 / 1 / 2: transposed copies, token-major weight-gradient operands (fk_gemm_args.layout 2), stored
    weights in the data gradients as well (layout 1) -- the same products summed in the same order: every gradient, the loss
    and d(prompt_embeds) bit for bit (B = 1: whole-tile shapes take the K-major path; B = 2 slices fall back per call)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from gpt_image_edit_amd import backward, ops
    from gpt_image_edit_amd.train_step import DenoiserTrainStep
    from gpt_image_edit_amd.transformer import HipFluxTransformer2DModel
    ops.gemm_set_plan(1)      # no split-K pairs: the row-major and K-major launches must add in the same order
    try:
        for shape, B in ((32, 1), (32, 2), (16, 1), (16, 2)):   # ... on the same MFMA shape, both of them (K-major forms follow fk_gemm_args.mfma)
            ops.gemm_set_mfma(shape)
            cfg, sd_bf, batch, trainable = _setup(B=B, S_txt=64, h=32, w=32)
            res = {}
            for level in (0, 1, 2):
                monkeypatch.setattr(backward, "K_MAJOR", level)
                model = HipFluxTransformer2DModel(cfg, device="cuda")
                model.load_state_dict(sd_bf)
                ts = DenoiserTrainStep(model, lr=1e-3)
                loss, grads, d_enc = ts.forward_backward(**{k: v.cuda() for k, v in batch.items()})
                torch.cuda.synchronize()
                res[level] = (loss.clone(), {k: v.clone() for k, v in grads.items()}, d_enc.clone())
            for level in (1, 2):
                assert torch.equal(res[0][0], res[level][0]) and torch.equal(res[0][2], res[level][2])
                for k in trainable:
                    assert torch.equal(res[0][1][k], res[level][1][k]), f"mfma {shape}, K_MAJOR {level}, B {B}: {k}"
    finally:
        ops.gemm_set_plan(3)
        ops.gemm_set_mfma(0)


@pytest.mark.parametrize("which", ["image_branch", "everything", "sharded"])
def test_block_level_backward_entry_points_give_the_same_bits(monkeypatch, which):
    """fk_single_block_bwd / fk_double_block_bwd (csrc/blocks_bwd.hip: one C call per block) against the per-launch route of
    backward.py: loss, d(prompt_embeds) and EVERY gradient bit for bit -- with the reference's trainable set, with every
    parameter of the blocks trainable (text branch and MLPs: only_tune_image_branch false), and on the ZeRO-2 layout, where
    the fused QKV operands are views of the flat parameter buffer.  The entry points must really be taken (counted)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from gpt_image_edit_amd import backward, libfk, training
    from gpt_image_edit_amd.train_step import DenoiserTrainStep
    from gpt_image_edit_amd.transformer import HipFluxTransformer2DModel
    cfg, sd_bf, batch, trainable = _setup(B=1, S_txt=64, h=32, w=32)
    if which == "everything":
        trainable = training.trainable_names(list(sd_bf.keys()), only_img_branch=False)
    calls = {"single": 0, "double": 0}
    lib = libfk.load()
    res = {}
    for api in (0, 1):
        monkeypatch.setattr(backward, "BLOCK_API", api)
        model = HipFluxTransformer2DModel(cfg, device="cuda")
        model.load_state_dict(sd_bf)
        ts = DenoiserTrainStep(model, lr=1e-3, trainable=trainable, sharded=which == "sharded", store_activations=True)
        if api:
            for kind in ("single", "double"):
                orig = getattr(backward.FluxBackward, f"_{kind}_backward_c")

                def counted(self, *a, _orig=orig, _kind=kind, **k):
                    calls[_kind] += 1
                    return _orig(self, *a, **k)
                monkeypatch.setattr(backward.FluxBackward, f"_{kind}_backward_c", counted)
        loss, grads, d_enc = ts.forward_backward(**{k: v.cuda() for k, v in batch.items()})
        torch.cuda.synchronize()
        res[api] = (loss.clone(), {k: v.clone() for k, v in grads.items()}, d_enc.clone())
    assert calls == {"single": 1, "double": 1}, calls
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][2], res[1][2])
    assert set(res[0][1]) == set(res[1][1]) and set(trainable) <= set(res[1][1])
    for k in sorted(res[0][1]):
        a, b = res[0][1][k], res[1][1][k]
        assert a.shape == b.shape and a.dtype == b.dtype and torch.equal(a, b), f"{which}: {k}"
    assert all(torch.isfinite(v.float()).all() and v.float().abs().max() > 0 for v in res[1][1].values())
    # shapes outside the entry points' scope fall back to the per-launch route: batch 2, recomputation
    monkeypatch.setattr(backward, "BLOCK_API", 1)
    cfg2, sd2, batch2, tr2 = _setup(B=2, S_txt=64, h=16, w=16)
    model = HipFluxTransformer2DModel(cfg2, device="cuda")
    model.load_state_dict(sd2)
    before = dict(calls)
    DenoiserTrainStep(model, lr=1e-3).forward_backward(**{k: v.cuda() for k, v in batch2.items()})
    model = HipFluxTransformer2DModel(cfg, device="cuda")
    model.load_state_dict(sd_bf)
    DenoiserTrainStep(model, lr=1e-3, store_activations=False).forward_backward(**{k: v.cuda() for k, v in batch.items()})
    assert calls == before


def test_sharded_gradient_accumulation_and_modified_gradient_error():
    """with sharded=True a second forward_backward before optimizer_step is a further micro-batch
    (the reference's gradient_accumulation_steps) -- its gradients are ADDED to the optimiser's chunks and the step uses
    their mean, which equals the per-tensor path stepped on the summed-and-halved gradients; gradients that were
    modified after forward_backward sank them are refused instead of silently ignored."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from gpt_image_edit_amd.train_step import DenoiserTrainStep
    from gpt_image_edit_amd.transformer import HipFluxTransformer2DModel
    cfg, sd_bf, batch, trainable = _setup(B=1, S_txt=40, h=8, w=12, seed=5)
    b1 = {k: v.cuda() for k, v in batch.items()}
    b2 = dict(b1, noise=torch.randn(1, 16, 8, 12, generator=torch.Generator().manual_seed(9)).cuda(),
              sigmas=torch.tensor([0.5]).cuda())
    res = []
    for sharded in (False, True):
        model = HipFluxTransformer2DModel(cfg, device="cuda")
        model.load_state_dict(sd_bf)
        ts = DenoiserTrainStep(model, lr=1e-3, sharded=sharded)
        l1, g1, _ = ts.forward_backward(**b1)
        g1 = {k: v.clone() for k, v in g1.items()}
        l2, g2, _ = ts.forward_backward(**b2)
        if sharded:
            ts.optimizer_step(g2)                       # both passes already sit in the optimiser: mean of the two
        else:
            ts.optimizer_step({k: ((g1[k].float() + g2[k].float()) / 2).contiguous() for k in g2})
        res.append((l1.item(), l2.item(), {k: model.p(k).detach().float().cpu() for k in trainable}))
    assert res[0][0] == res[1][0] and res[0][1] == res[1][1] and res[0][0] != res[0][1]
    for k in trainable:   # one path sums fp32 casts of bf16 gradients in the chunk, the other on the caller's side: same values
        assert (res[0][2][k] - res[1][2][k]).abs().max().item() <= 2.0 ** -8 * res[0][2][k].abs().max().item() + 1e-6, k
    # a gradient the caller scaled after forward_backward had handed it to the buckets
    model = HipFluxTransformer2DModel(cfg, device="cuda")
    model.load_state_dict(sd_bf)
    ts = DenoiserTrainStep(model, lr=1e-3, sharded=True)
    _, g, _ = ts.forward_backward(**b1)
    next(iter(g.values())).mul_(0.5)
    with pytest.raises(RuntimeError, match="not the one forward_backward already handed"):
        ts.optimizer_step(g)


def test_prepared_conditioning_notices_an_optimiser_step():
    """the prepared all-steps modulation is keyed on the tensors it was prepared from AND on the
    version stamps of the embedder / modulation weights, so a caller that keeps the prepared timesteps across an optimiser
    step (torch's, or fk_adamw_step writing through raw pointers) gets fresh conditioning, not the stale rows."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from gpt_image_edit_amd import flux_spec, ops
    from gpt_image_edit_amd.transformer import HipFluxTransformer2DModel
    from test_hip_mmdit import _inputs
    cfg = dict(flux_spec.FLUX_KONTEXT_CONFIG, num_layers=1, num_single_layers=1)
    m = HipFluxTransformer2DModel(cfg, device="cuda", init="synthetic", seed=5)
    hs, enc, pooled, _, gd, img_ids, txt_ids = (x.cuda() for x in _inputs(1, 40, 6, 8, cfg, seed=3))
    steps = torch.tensor([[1.0], [0.5]]).to(BF).cuda()
    kw = dict(hidden_states=hs, encoder_hidden_states=enc, pooled_projections=pooled, guidance=gd, txt_ids=txt_ids,
              img_ids=img_ids, return_dict=False)
    m.prepare_conditioning(steps, gd, pooled)
    before = m(timestep=steps[1], **kw)[0].clone()
    assert m._cond is not None
    w = m.p("single_transformer_blocks.0.norm.linear.weight")
    with torch.no_grad():
        w.mul_(1.5)                                     # what a torch optimiser does: an in-place write, version bumped
    after = m(timestep=steps[1], **kw)[0].clone()
    assert m._cond is None and not torch.equal(before, after)
    fresh = m(timestep=steps[1].clone(), **kw)[0]        # on-the-fly conditioning with the new weight
    assert torch.equal(after, fresh)
    # the same through fk_adamw_step's raw-pointer write of the bf16 copy
    m.prepare_conditioning(steps, gd, pooled)
    assert torch.equal(m(timestep=steps[1], **kw)[0], after) and m._cond is not None
    b = m.p("single_transformer_blocks.0.norm.linear.bias")
    master = b.detach().float().contiguous()
    ops.adamw_step(master, torch.ones_like(master), torch.zeros_like(master), torch.zeros_like(master), 1, 0.5,
                   param_bf16=b.data, grad_sumsq=None)
    moved = m(timestep=steps[1], **kw)[0]
    assert m._cond is None and not torch.equal(moved, after)
    # ids built afresh every call (the reference's training loop, modeling_univa_denoise_tower.py:73-75) give the same bits
    assert torch.equal(m(timestep=steps[1].clone(), **dict(kw, txt_ids=txt_ids.clone(), img_ids=img_ids.clone()))[0], moved)


def test_step_takes_the_stage2_loss_weights():
    """VERDICT r4 missing #3: ``forward_backward`` carries the loss of the shipped stage-2 config (``mask_weight_type: 'log'``,
    train_denoiser.py:1123-1165).  Per-pixel maps given at IMAGE resolution are nearest-resized to the latent size as the
    reference does; the loss equals ``oracle.train.flow_matching_loss`` on the HIP model's own prediction with the same
    maps, all-ones maps reproduce the unweighted step bit for bit, and a masked-out region contributes no gradient."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import torch.nn.functional as F
    from gpt_image_edit_amd import ops
    from gpt_image_edit_amd.train_step import DenoiserTrainStep
    from gpt_image_edit_amd.transformer import HipFluxTransformer2DModel
    from oracle import helpers, train as otrain
    cfg, sd_bf, batch, trainable = _setup()
    model = HipFluxTransformer2DModel(cfg, device="cuda")
    model.load_state_dict(sd_bf)
    ts = DenoiserTrainStep(model)
    dev = {k: v.cuda() for k, v in batch.items()}
    B, C, h, w = batch["model_input"].shape
    g = torch.Generator().manual_seed(3)
    area = torch.log1p(torch.rand(B, 1, 8 * h, 8 * w, generator=g) * 30.0) + 0.1      # at pixel resolution
    mask = torch.ones(B, 1, h, w)
    mask[1, :, h // 2:, :] = 0.0                                                        # sample 1: lower half is padding
    weighting = torch.tensor([1.5, 0.5])
    loss0, grads0, _ = ts.forward_backward(**dev)
    loss1, grads1, _ = ts.forward_backward(**dev, weighting=torch.ones(B), area_mask_weights=torch.ones(B, 1, h, w))
    assert torch.equal(loss0, loss1) and all(torch.equal(grads0[k], grads1[k]) for k in grads0)
    loss2, grads2, _ = ts.forward_backward(**dev, weighting=weighting, area_mask_weights=area, weight_mask=mask)
    # the oracle's loss on the HIP forward's own prediction (the forward itself is held to the oracle elsewhere)
    inp, S_tgt = ts.prepare_inputs(dev["model_input"], dev["cond_latents"], dev["noise"], dev["sigmas"], dev["prompt_embeds"], dev["pooled"])
    pred = ts.bw.forward(inp["hidden_states"], inp["encoder_hidden_states"], inp["pooled_projections"], inp["timestep"],
                         inp["img_ids"], inp["txt_ids"], inp["guidance"])[:, :S_tgt].cpu()
    area_l = F.interpolate(area, size=(h, w), mode="nearest")
    wfull = weighting.view(B, 1, 1, 1).float() * area_l.float() * mask.float()
    ref = otrain.flow_matching_loss(helpers.unpack_latents(pred, h * 8, w * 8), batch["model_input"], batch["noise"], wfull, weight_mask=mask)
    assert float(loss2) == pytest.approx(float(ref), rel=1e-5)
    assert abs(float(loss2) - float(loss0)) > 1e-3 * float(loss0)
    assert any(not torch.equal(grads2[k], grads0[k]) for k in grads0)
    # the loss kernel's gradient is zero on the padding
    _, dpred = ops.flow_loss(pred.cuda(), dev["model_input"].contiguous(), dev["noise"].contiguous(), weight=weighting.cuda(),
                             area_mask_weights=area_l.cuda().contiguous(), weight_mask=mask.cuda())
    pad = helpers.pack_latents((1.0 - mask).expand(B, C, h, w).contiguous()) > 0
    assert float(dpred.cpu().float()[pad].abs().max()) == 0 and float(dpred.cpu().float().abs().max()) > 0


@pytest.mark.parametrize("sharded", [True, False])
def test_optimizer_state_save_resume_and_discard(sharded):
    """``DenoiserTrainStep.state_dict`` / ``load_state_dict`` (accelerator.save_state / load_state, train_denoiser.py:1229,
    769): a run resumed from the saved optimiser state takes the same next step, bit for bit, as the uninterrupted one --
    masters, moments, step count and the bf16 weights the forward reads.  ``discard()`` drops a backward pass whose step
    is skipped instead of folding it into the next one as a micro-batch."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from gpt_image_edit_amd.train_step import DenoiserTrainStep
    from gpt_image_edit_amd.transformer import HipFluxTransformer2DModel
    cfg, sd_bf, batch, trainable = _setup()
    dev = {k: v.cuda() for k, v in batch.items()}
    dev2 = dict(dev, sigmas=torch.tensor([0.6, 0.1]).cuda())

    def fresh(state):
        m = HipFluxTransformer2DModel(cfg, device="cuda")
        m.load_state_dict(state)
        return m, DenoiserTrainStep(m, lr=1e-3, sharded=sharded)
    model, ts = fresh(sd_bf)
    ts.step(**dev)
    ts.step(**dev2)
    saved = ts.state_dict()
    weights = {k: v.detach().clone() for k, v in model.state_dict().items()}          # the bf16 checkpoint written beside it
    ts.forward_backward(**dev2)                                                       # a backward pass whose step is skipped ...
    ts.discard()                                                                      # ... must leave no trace
    out_a = ts.step(**dev)
    want = {k: model.p(k).detach().clone() for k in trainable}
    model_b, ts_b = fresh(weights)
    ts_b.load_state_dict(saved)
    assert ts_b.step_count == 2
    out_b = ts_b.step(**dev)
    torch.cuda.synchronize()
    assert torch.equal(out_a["loss"], out_b["loss"]) and torch.equal(torch.as_tensor(out_a["grad_sumsq"]).cpu(), torch.as_tensor(out_b["grad_sumsq"]).cpu())
    assert all(torch.equal(want[k], model_b.p(k)) for k in trainable), "the resumed run's next step differs"
    with pytest.raises(ValueError):
        fresh(weights)[1].__class__(fresh(weights)[0], lr=1e-3, sharded=not sharded).load_state_dict(saved)
