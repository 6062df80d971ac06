"""The training seam of the reference: ``train_denoiser.py`` selects parameters through
``named_modules()`` (:534-548), calls the denoiser inside ``lvlm_model(...)`` (:1073-1093), computes the loss in torch
(:1095-1167), ``accelerator.backward(loss)`` (:1172) and ``optimizer.step()`` (:1179).  These tests drive the HIP model
the same way -- requires_grad flags, a torch loss, ``loss.backward()``, a stock ``torch.optim.AdamW`` -- and hold the
result against the step API (``DenoiserTrainStep``) and against the oracle's autograd.
"""
import pytest
import torch
from torch import nn

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def _rel(a, b):
    return ((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-12)).item()


def _setup(B=2, S_txt=64, h=16, w=16, seed=0, only_img_branch=True):
    from gpt_image_edit_amd import flux_spec, training
    cfg = dict(flux_spec.FLUX_KONTEXT_CONFIG, num_layers=1, num_single_layers=1)
    sd_bf = {k: v.to(BF) for k, v in flux_spec.synthetic_state(flux_spec.flux_param_shapes(cfg), seed=41).items()}
    g = torch.Generator().manual_seed(seed)
    batch = dict(model_input=torch.randn(B, 16, h, w, generator=g), cond_latents=torch.randn(B, 16, h, w, generator=g),
                 noise=torch.randn(B, 16, h, w, generator=g), sigmas=torch.tensor([0.25, 0.75][:B]),
                 prompt_embeds=torch.randn(B, S_txt, 4096, generator=g).to(BF), pooled=torch.randn(B, 768, generator=g).to(BF))
    trainable = training.trainable_names(list(sd_bf.keys()), layers_to_train=(0, 19), only_img_branch=only_img_branch)
    return cfg, sd_bf, batch, trainable


def _reference_selection(lvlm, only_img_branch=True):
    """train_denoiser.py:478-548 on a 1 + 1 block denoiser (layer 0 = the double block, layer 19 = the single block)."""
    from gpt_image_edit_amd import training
    lvlm.requires_grad_(False)
    comps = training.get_trainable_params(layers_to_train=[0, 19], only_img_branch=only_img_branch)
    for name, module in lvlm.named_modules():
        if training.check_param_is_in_components(name, comps):
            module.requires_grad_(True)
    return [p for p in lvlm.parameters() if p.requires_grad]


def test_loss_backward_fills_grads_like_the_step_api_and_a_stock_optimizer_steps():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from gpt_image_edit_amd import helpers
    from gpt_image_edit_amd.backward import FluxBackward
    from gpt_image_edit_amd.train_step import DenoiserTrainStep
    from gpt_image_edit_amd.transformer import HipFluxTransformer2DModel
    cfg, sd_bf, batch, trainable = _setup()
    model = HipFluxTransformer2DModel(cfg, device="cuda")
    model.load_state_dict(sd_bf)
    dev_batch = {k: v.cuda() for k, v in batch.items()}
    ts = DenoiserTrainStep(model, lr=1e-3)
    loss_api, grads_api, d_enc_api = ts.forward_backward(**dev_batch)
    inp, S_tgt = ts.prepare_inputs(**dev_batch)

    # ---- the reference's way --------------------------------------------------------------------------------
    lvlm = nn.Module()
    lvlm.denoise_tower = nn.Module()
    lvlm.denoise_tower.denoiser = model
    params = _reference_selection(lvlm)
    assert sorted(model.grad_parameter_names()) == sorted(trainable)
    model.enable_gradient_checkpointing()                                 # :486 -> recompute policy
    pe = inp["encoder_hidden_states"].detach().clone().requires_grad_(True)
    B, C, h, w = batch["model_input"].shape
    model_pred = model(**dict(inp, encoder_hidden_states=pe), joint_attention_kwargs={}, return_dict=False)[0]
    assert model_pred.requires_grad and model_pred.grad_fn is not None
    model_pred = helpers._unpack_latents(model_pred[:, :S_tgt], h * 8, w * 8, 8)     # :1095-1104
    target = dev_batch["noise"] - dev_batch["model_input"]
    weighting = torch.ones(B, 1, 1, 1, device="cuda")
    loss = (weighting.float() * (model_pred.float() - target.float()) ** 2).reshape(B, -1).mean()   # :1157-1167
    loss.backward()                                                          # accelerator.backward(loss), :1172
    torch.cuda.synchronize()
    assert abs(loss.item() - loss_api.item()) <= 1e-5 * abs(loss_api.item())
    worst = 0.0
    for k in trainable:
        g = model.p(k).grad
        assert g is not None and g.dtype == BF and g.shape == model.p(k).shape, k
        e = _rel(g, grads_api[k])
        worst = max(worst, e)
        # same kernels; the only difference is where d loss / d sample is rounded to bf16 (torch's cast vs fk_flow_loss)
        assert e <= 5e-3, f"{k}: loss.backward() vs step API {e:.3e}"
    assert _rel(pe.grad, d_enc_api) <= 5e-3
    frozen = [k for k in sd_bf if k not in trainable]
    assert all(model.p(k).grad is None for k in frozen)
    print(f"[seam] loss.backward() vs DenoiserTrainStep: worst relative difference {worst:.3e} over {len(trainable)} tensors")

    # ---- bit for bit: the autograd node IS FluxBackward ---------------------------------------------------------
    dsample = (torch.randn(inp["hidden_states"].shape, device="cuda", generator=torch.Generator("cuda").manual_seed(1)) * 1e-3).to(BF)
    for prm in params:
        prm.grad = None
    pe2 = pe.detach().clone().requires_grad_(True)
    out = model(**dict(inp, encoder_hidden_states=pe2), return_dict=False)[0]
    out.backward(dsample)
    bw = FluxBackward(model, trainable=trainable, store_activations=False)
    out_d = bw.forward(inp["hidden_states"], pe2.detach(), inp["pooled_projections"], inp["timestep"], inp["img_ids"],
                       inp["txt_ids"], inp["guidance"])
    g_d, d_enc_d = bw.backward(dsample)
    assert torch.equal(out.detach(), out_d)
    assert torch.equal(pe2.grad, d_enc_d)
    for k in trainable:
        assert torch.equal(model.p(k).grad, g_d[k].to(BF)), k
    # stored activations (no enable_gradient_checkpointing) give the same bits
    model.disable_gradient_checkpointing()
    for prm in params:
        prm.grad = None
    out = model(**dict(inp, encoder_hidden_states=pe2.detach()), return_dict=False)[0]
    out.backward(dsample)
    for k in trainable:
        assert torch.equal(model.p(k).grad, g_d[k].to(BF)), k

    # ---- optimizer.step() of a stock optimiser (:1179), then the model must run on the NEW weights -------------
    with torch.no_grad():
        before = model(**inp, return_dict=False)[0].clone()
    opt = torch.optim.AdamW(params, lr=1e-2, betas=(0.9, 0.99), weight_decay=0.0)
    opt.step()
    opt.zero_grad()
    with torch.no_grad():
        after = model(**inp, return_dict=False)[0].clone()
    fresh = HipFluxTransformer2DModel(cfg, device="cuda")
    fresh.load_state_dict(model.state_dict())
    with torch.no_grad():
        want = fresh(**inp, return_dict=False)[0]
    assert not torch.equal(before, after), "the optimiser step did not reach the fused weight copies"
    assert torch.equal(after, want), "stale fused QKV / modulation copies after optimizer.step()"
    # and the backward's transposed weights follow as well
    out = model(**dict(inp, encoder_hidden_states=pe2.detach().clone().requires_grad_(True)), return_dict=False)[0]
    out.backward(dsample)
    bw2 = FluxBackward(fresh, trainable=trainable, store_activations=False)
    bw2.forward(inp["hidden_states"], pe2.detach(), inp["pooled_projections"], inp["timestep"], inp["img_ids"], inp["txt_ids"],
                inp["guidance"])
    g2, _ = bw2.backward(dsample)
    for k in trainable:
        assert torch.equal(model.p(k).grad, g2[k].to(BF)), k


def test_text_branch_and_mlp_gradients_match_autograd():
    """``only_tune_image_branch: false`` (train_denoiser.py:96-109) adds norm1_context.linear, the added q/k norms, both
    MLPs of the double blocks and proj_mlp / proj_out of the single blocks: every one of them against the oracle."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from gpt_image_edit_amd.train_step import DenoiserTrainStep
    from gpt_image_edit_amd.transformer import HipFluxTransformer2DModel
    from oracle import train as otrain
    cfg, sd_bf, batch, trainable = _setup(B=1, S_txt=48, h=8, w=8, seed=2, only_img_branch=False)
    extra = [k for k in trainable if any(t in k for t in ("ff.net", "ff_context.net", "norm1_context", "norm_added", "proj_mlp", "proj_out"))]
    assert len(extra) == 2 + 2 + 8 + 4
    # plus what the selection does not name but the backward can produce
    more = [f"transformer_blocks.0.attn.{n}.{wb}" for n in ("add_q_proj", "add_k_proj", "add_v_proj", "to_add_out") for wb in ("weight", "bias")]
    names = sorted(set(trainable) | set(more))
    model = HipFluxTransformer2DModel(cfg, device="cuda")
    model.load_state_dict(sd_bf)
    ts = DenoiserTrainStep(model, lr=1e-3, trainable=names)
    loss, grads, _ = ts.forward_backward(**{k: v.cuda() for k, v in batch.items()})
    assert set(grads) == set(names)
    sd32 = {k: v.float() for k, v in sd_bf.items()}
    b32 = dict(batch, prompt_embeds=batch["prompt_embeds"].float(), pooled=batch["pooled"].float())
    ref32 = otrain.train_step(sd32, names, b32, {}, flux_config=cfg, lr=1e-3)
    refbf = otrain.train_step(sd_bf, names, batch, {}, flux_config=cfg, lr=1e-3)
    for k in names:
        e_hip, e_floor = _rel(grads[k].cpu(), ref32["grads"][k]), _rel(refbf["grads"][k], ref32["grads"][k])
        e_bf = _rel(grads[k].cpu(), refbf["grads"][k])
        print(f"[grad] {k:55s} hip-vs-fp32 {e_hip:.3e}  bf16-autograd floor {e_floor:.3e}  hip-vs-bf16-autograd {e_bf:.3e}")
        assert e_hip <= max(2.0 * e_floor, 2e-2), k
        # 80 tokens: a few bias gradients are small sums of rounding-dominated terms (measured: to_k.bias 2.2e-2 from bf16
        # autograd with bf16 autograd itself 2.3e-2 from fp32) -- as close to bf16 autograd as bf16 autograd is to fp32
        assert e_bf <= max(1.0 * e_floor, 3e-2), k
    with pytest.raises(NotImplementedError):
        DenoiserTrainStep(model, trainable=["proj_out.weight"])


def test_weight_gradient_buffers_are_clean_between_token_counts():
    """wgrad's transposed operands live in buffers keyed by the token count padded to 64; a call with
    fewer tokens than its predecessor (VLM length 310, then 300) must not see the predecessor's tail."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from gpt_image_edit_amd.backward import wgrad
    bufs = {}

    def buf(name, shape, dtype=BF, zero=False):
        t = bufs.get(name)
        if t is None or tuple(t.shape) != tuple(shape):
            t = (torch.zeros if zero else torch.empty)(shape, device="cuda", dtype=dtype)
            bufs[name] = t
        return t
    g = torch.Generator().manual_seed(3)
    N, K = 256, 192
    for R in (310, 300, 257, 319):          # all pad to 320
        dy = torch.randn(1, R, N, generator=g).to(BF).cuda()
        x = torch.randn(1, R, K, generator=g).to(BF).cuda()
        got = wgrad(buf, dy, x)
        fresh = wgrad(lambda name, shape, dtype=BF, zero=False: torch.zeros(shape, device="cuda", dtype=dtype), dy, x)
        want = dy[0].float().t() @ x[0].float()
        assert torch.equal(got, fresh), f"R = {R}: stale pad columns entered the weight gradient"
        assert _rel(got, want) < 1e-2
    assert len(bufs) == 2                    # one pair of buffers served all four calls


def test_projector_autograd_node():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from gpt_image_edit_amd.projector import HipDenoiseProjector
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 37, 3584, generator=g).to(BF).cuda()
    dy = (torch.randn(2, 37, 4096, generator=g) * 0.1).to(BF).cuda()
    pj = HipDenoiseProjector(device="cuda", init="synthetic", seed=7)
    with torch.no_grad():
        y0 = pj(x)
    for name, param in pj.named_parameters():          # train_denoiser.py:545-548
        param.requires_grad_(True)
    y = pj(x)
    assert torch.equal(y.detach(), y0)
    y.backward(dy)
    pj.forward_train(x)
    want = pj.backward(dy)
    for k in ("0.weight", "0.bias", "2.weight", "2.bias"):
        assert torch.equal(pj.p(k).grad, want[k].to(BF)), k
