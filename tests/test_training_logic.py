"""Training-step host logic (gpt_image_edit_amd/training.py) and its oracle (oracle/train.py) -- CPU only.

Pinned to the reference's own functions through tests/golden/train.npz (oracle/make_golden.py::g_train lifts them out
of train_denoiser.py); the optimiser arithmetic of the oracle is pinned to torch's own AdamW / clip_grad_norm_; the
autograd gradient of the oracle loss is checked against a finite difference."""
import os

import numpy as np
import pytest
import torch

from gpt_image_edit_amd import flux_spec, training
from oracle import train as otrain


@pytest.fixture(scope="module")
def golden(golden_dir):
    return np.load(os.path.join(golden_dir, "train.npz"))


@pytest.mark.parametrize("impl", [training, otrain])
def test_trainable_parameter_selection_matches_reference(golden, impl):
    keys = sorted(flux_spec.flux_param_shapes(flux_spec.FLUX_KONTEXT_CONFIG))
    assert len(keys) == int(golden["n_keys"])
    for tag, kw in (("img", dict(only_img_branch=True)), ("all", dict(only_img_branch=False)),
                    ("img_layers_0_1_12_19_30", dict(layers_to_train=[0, 1, 12, 19, 30], only_img_branch=True))):
        comps = impl.get_trainable_params(**kw)
        assert len(comps) == int(golden[f"n_components_{tag}"])
        mask = np.array([impl.check_param_is_in_components("denoise_tower.denoiser." + k, comps) for k in keys])
        assert np.array_equal(mask, golden[f"trainable_{tag}"]), tag
    # the stage-2 configuration: attention projections, q/k norms and the modulation linears of every block, image side
    names = training.trainable_names(keys)
    assert len(names) == int(golden["trainable_img"].sum()) == 608
    assert "transformer_blocks.0.attn.to_q.weight" in names and "transformer_blocks.0.attn.add_q_proj.weight" not in names
    assert "single_transformer_blocks.37.norm.linear.bias" in names and "single_transformer_blocks.0.proj_mlp.weight" not in names
    n_params = sum(int(np.prod(flux_spec.flux_param_shapes(flux_spec.FLUX_KONTEXT_CONFIG)[k])) for k in names)
    assert 3.9e9 < n_params < 4.2e9   # ~4.04 B trainable with the projector


@pytest.mark.parametrize("impl", [training, otrain])
def test_sigma_shift_and_lookup_match_reference(golden, impl):
    sig = torch.from_numpy(golden["shift_in"])
    for (h, w), ref in zip(golden["shift_hw"], golden["shift_out"]):
        got = impl.apply_flux_schedule_shift(sig.clone(), int(h), int(w))
        np.testing.assert_allclose(got.numpy(), ref, rtol=1e-6, atol=1e-7)
    st, ss = torch.from_numpy(golden["sched_timesteps"]), torch.from_numpy(golden["sched_sigmas"])
    got = impl.get_sigmas(torch.from_numpy(golden["get_sigmas_t"]), st, ss, n_dim=4, dtype=torch.float32)
    assert got.shape == (4, 1, 1, 1)
    np.testing.assert_array_equal(got.numpy(), golden["get_sigmas_out"])


def test_sampling_and_weighting():
    g1, g2 = torch.Generator().manual_seed(5), torch.Generator().manual_seed(5)
    s_a, t_a = training.sample_sigmas(8, 128, 128, generator=g1)
    s_b, t_b = otrain.sample_sigmas_continuous(8, 128, 128, generator=g2)
    assert torch.equal(s_a, s_b) and torch.equal(t_a, t_b) and torch.equal(t_a, s_a * 1000.0)
    assert float(s_a.min()) > 0 and float(s_a.max()) < 1
    # the shift pushes sigmas towards 1 more strongly at higher resolution (mu = 1.15 at 4096 tokens, 0.5 at 256)
    assert float(training.apply_flux_schedule_shift(torch.tensor([0.5]), 128, 128)) == pytest.approx(
        np.exp(1.15) / (np.exp(1.15) + 1), rel=1e-6)
    sig = torch.tensor([0.1, 0.5, 0.9])
    for scheme in ("logit_normal", "mode", "none", "sigma_sqrt", "cosmap"):
        assert torch.equal(training.loss_weighting(scheme, sig), otrain.compute_loss_weighting_for_sd3(scheme, sig))
    assert torch.equal(training.loss_weighting("logit_normal", sig), torch.ones(3))
    assert training.loss_weighting("logit_normal", sig, sigmas_as_weight=True) is sig
    u = otrain.compute_density_for_timestep_sampling("mode", 1000, generator=torch.Generator().manual_seed(1))
    assert 0.0 <= float(u.min()) and float(u.max()) <= 1.0


def test_oracle_optimizer_matches_torch():
    torch.manual_seed(3)
    params = {f"p{i}": torch.randn(7, 5 + i) for i in range(3)}
    ref = {k: torch.nn.Parameter(v.clone()) for k, v in params.items()}
    opt = torch.optim.AdamW(ref.values(), lr=1e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=1e-2)
    state = {}
    for step in range(3):
        grads = {k: torch.randn_like(v) * (5.0 if step == 0 else 0.05) for k, v in params.items()}
        for k in ref:
            ref[k].grad = grads[k].clone()
        norm_ref = torch.nn.utils.clip_grad_norm_(ref.values(), 1.0)
        opt.step()
        params, state, norm = otrain.adamw_step(params, grads, state, lr=1e-3, betas=(0.9, 0.99), eps=1e-8,
                                                weight_decay=1e-2, max_grad_norm=1.0)
        assert float(norm) == pytest.approx(float(norm_ref), rel=1e-6)
        for k in params:
            torch.testing.assert_close(params[k], ref[k].detach(), rtol=2e-6, atol=2e-7)


def test_oracle_train_step_gradient_is_the_loss_gradient():
    cfg = dict(flux_spec.FLUX_KONTEXT_CONFIG, num_layers=1, num_single_layers=1, num_attention_heads=2,
               joint_attention_dim=64, pooled_projection_dim=32)
    sd = {k: v.double() for k, v in flux_spec.synthetic_state(flux_spec.flux_param_shapes(cfg), seed=3).items()}
    g = torch.Generator().manual_seed(2)
    B, h, w = 2, 4, 6
    batch = dict(model_input=torch.randn(B, 16, h, w, generator=g).double(), cond_latents=torch.randn(B, 16, h, w, generator=g).double(),
                 noise=torch.randn(B, 16, h, w, generator=g).double(), sigmas=torch.tensor([0.3, 0.8]),
                 prompt_embeds=torch.randn(B, 5, 64, generator=g).double(), pooled=torch.randn(B, 32, generator=g).double())
    names = training.trainable_names(sorted(sd))
    assert names and all(("attn." in n or "norm" in n) for n in names)
    out = otrain.train_step(sd, names, batch, {}, flux_config=cfg, lr=1e-3)
    assert torch.isfinite(out["loss"]) and float(out["grad_norm"]) > 0
    # finite difference along one weight entry and along a whole-tensor direction
    key = "transformer_blocks.0.attn.to_q.weight"
    # (the oracle keeps the reference's fp32 islands -- RMSNorm statistics, RoPE, the loss -- so the difference quotient
    # is only good to ~1e-7 / eps; step along the gradient itself for the largest signal)
    d = out["grads"][key] / out["grads"][key].norm() * sd[key].norm()
    eps = 3e-3
    lp = otrain.denoiser_loss({**sd, key: sd[key] + eps * d}, flux_config=cfg, **batch)
    lm = otrain.denoiser_loss({**sd, key: sd[key] - eps * d}, flux_config=cfg, **batch)
    fd = float((lp - lm) / (2 * eps))
    an = float((out["grads"][key] * d).sum())
    print(f"directional derivative: autograd {an:.6e}, finite difference {fd:.6e}")
    assert an > 0 and an == pytest.approx(fd, rel=5e-2)
    # frozen tensors are untouched, trainable ones moved against their gradient
    assert set(out["params"]) == set(names)
    assert float(((out["params"][key] - sd[key]) * out["grads"][key]).sum()) < 0


def test_oracle_train_step_selfpin(golden):
    # same tiny case as oracle/make_golden.py::_tiny_train_case, fp32
    cfg = dict(flux_spec.FLUX_KONTEXT_CONFIG, num_layers=1, num_single_layers=1, num_attention_heads=2,
               joint_attention_dim=64, pooled_projection_dim=32)
    sd = flux_spec.synthetic_state(flux_spec.flux_param_shapes(cfg), seed=3)
    g = torch.Generator().manual_seed(2)
    B, h, w = 2, 4, 6
    batch = dict(model_input=torch.randn(B, 16, h, w, generator=g), cond_latents=torch.randn(B, 16, h, w, generator=g),
                 noise=torch.randn(B, 16, h, w, generator=g), sigmas=torch.tensor([0.3, 0.8]),
                 prompt_embeds=torch.randn(B, 5, 64, generator=g), pooled=torch.randn(B, 32, generator=g))
    res = otrain.train_step(sd, training.trainable_names(sorted(sd)), batch, {}, flux_config=cfg, lr=1e-3)
    np.testing.assert_allclose(res["loss"].numpy(), golden["step_loss"], rtol=1e-5)
    np.testing.assert_allclose(res["grad_norm"].numpy(), golden["step_grad_norm"], rtol=1e-4)
    np.testing.assert_allclose(res["grads"]["transformer_blocks.0.attn.to_q.weight"][:4, :8].numpy(), golden["step_grad_to_q"],
                               rtol=1e-3, atol=1e-8)
    np.testing.assert_allclose(res["params"]["single_transformer_blocks.0.attn.norm_q.weight"].numpy(),
                               golden["step_new_norm_q"], rtol=1e-6)


def test_reference_named_modules_selection_drives_the_hip_model(golden):
    """The reference un-freezes by MODULE name (train_denoiser.py:534-548): ``for name, module in
    lvlm_model.named_modules(): if check_param_is_in_components(name, trainable_components):
    module.requires_grad_(True)``, after ``lvlm_model.requires_grad_(False)`` (:478-479), then the projector by
    parameter name (:545-548).  Run exactly that loop on the HIP model nested the way the reference nests it
    (``lvlm_model.denoise_tower.denoiser`` / ``.denoise_projector``; meta device: names and shapes only) and
    hold the resulting trainable set against the mask the reference's own functions produced (train.npz)."""
    from torch import nn
    from gpt_image_edit_amd.projector import HipDenoiseProjector
    from gpt_image_edit_amd.transformer import HipFluxTransformer2DModel

    lvlm = nn.Module()
    lvlm.denoise_tower = nn.Module()
    lvlm.denoise_tower.denoiser = HipFluxTransformer2DModel(device="meta", init="empty")
    lvlm.denoise_tower.denoise_projector = HipDenoiseProjector(device="meta", init="empty")
    lvlm.requires_grad_(False)
    lvlm.denoise_tower.denoiser.enable_gradient_checkpointing()           # :486
    comps = training.get_trainable_params(layers_to_train=list(range(57)), only_img_branch=True)   # pinned above
    for name, module in lvlm.named_modules():                              # :538-543
        if training.check_param_is_in_components(name, comps):
            module.requires_grad_(True)
    den = lvlm.denoise_tower.denoiser
    keys = sorted(flux_spec.flux_param_shapes(flux_spec.FLUX_KONTEXT_CONFIG))
    assert sorted(k for k, _ in den.named_parameters()) == keys           # no name mangling in state_dict / parameters
    got = np.array([den.p(k).requires_grad for k in keys])
    assert np.array_equal(got, golden["trainable_img"])
    assert sorted(den.grad_parameter_names()) == sorted(training.trainable_names(keys))
    for name, param in lvlm.named_parameters():                            # :545-548 (with_tune_mlp2)
        if "denoise_tower.denoise_projector" in name:
            param.requires_grad_(True)
    n_train = sum(p.numel() for p in lvlm.parameters() if p.requires_grad)
    assert 4.03e9 < n_train < 4.05e9                                       # 4.04 B
    # the text-branch configuration (only_tune_image_branch: false) is accepted by the backward as well
    from gpt_image_edit_amd.backward import FluxBackward
    lvlm.requires_grad_(False)
    comps_all = training.get_trainable_params(only_img_branch=False)
    for name, module in lvlm.named_modules():
        if training.check_param_is_in_components(name, comps_all):
            module.requires_grad_(True)
    assert np.array_equal(np.array([den.p(k).requires_grad for k in keys]), golden["trainable_all"])
    FluxBackward(den, trainable=den.grad_parameter_names())               # every name has a weight gradient
    with pytest.raises(NotImplementedError, match="x_embedder"):
        FluxBackward(den, trainable=["x_embedder.weight"])


def test_save_pretrained_round_trip(tmp_path):
    """``save_pretrained`` (train_denoiser.py:493) writes the diffusers transformer layout; ``from_pretrained`` and
    ``checkpoint.read_flux_transformer`` read it back bit for bit."""
    import json
    from gpt_image_edit_amd.transformer import HipFluxTransformer2DModel
    cfg = dict(num_layers=1, num_single_layers=1, num_attention_heads=2, joint_attention_dim=64, pooled_projection_dim=32)
    m = HipFluxTransformer2DModel(config=cfg, device="cpu", init="synthetic", seed=3)
    d = m.save_pretrained(str(tmp_path / "transformer"), max_shard_size=200_000)
    raw = json.load(open(os.path.join(d, "config.json")))
    assert raw["_class_name"] == "FluxTransformer2DModel" and raw["num_layers"] == 1 and raw["axes_dims_rope"] == [16, 56, 56]
    assert any(f.endswith(".safetensors.index.json") for f in os.listdir(d))     # sharded like the 24 GB original
    m2 = HipFluxTransformer2DModel.from_pretrained(d, device="cpu")
    sd, sd2 = m.state_dict(), m2.state_dict()
    assert list(sd) == list(sd2) and all(torch.equal(sd[k], sd2[k]) for k in sd)
    assert vars(m2.config) == vars(m.config)


def test_training_packs_alias_the_flat_zero_buffer():
    """With ZeRO's flat buffer in ``backward_order`` the q | k | v thirds of every fused QKV operand lie side by side: the
    model's training packs are VIEWS of the parameters (an optimiser step through raw pointers needs no re-pack), the frozen
    text-branch operands are copies made once, and the modulation weights are not fused at all."""
    from gpt_image_edit_amd import flux_spec, training
    from gpt_image_edit_amd.transformer import HipFluxTransformer2DModel
    from gpt_image_edit_amd.zero import FlatLayout, backward_order
    cfg = dict(flux_spec.FLUX_KONTEXT_CONFIG, num_layers=2, num_single_layers=2, num_attention_heads=2, joint_attention_dim=64,
               pooled_projection_dim=32)
    m = HipFluxTransformer2DModel(cfg, device="cpu", init="synthetic", seed=3)
    D = m.inner_dim
    fused_before = m.pack_weights()
    assert fused_before.mod_w is not None and not fused_before.aliased
    ref_w = fused_before.single[1].wqkv.clone()
    names = sorted(training.trainable_names(list(m._pmap.keys())))
    order = backward_order(names)
    blk = [n for n in order if n.startswith("single_transformer_blocks.1.attn.to_")]
    assert blk == [f"single_transformer_blocks.1.attn.{t}.{wb}" for wb in ("bias", "weight") for t in ("to_q", "to_k", "to_v")]
    L = FlatLayout({n: m.p(n).shape for n in names}, 1, order=order, bucket_numel=3 * D * D)     # several buckets
    flat = torch.zeros(L.total, dtype=torch.bfloat16)
    views = L.views(flat)
    for n in names:
        views[n].copy_(m.p(n).data)
        m.p(n).data = views[n]
    m._train_packs, m._packed = True, None
    pk = m.packed()
    assert pk.mod_w is None and [n for n, _, _ in pk.mod_parts][-1] == "norm_out.linear" and pk.mod_parts[-1][1] + 2 * D == pk.mod_total
    assert torch.equal(pk.single[1].wqkv, ref_w)
    # image-branch and single-block operands alias the flat buffer unless a bucket boundary separates the thirds
    aliased = set(pk.aliased)
    assert "single_transformer_blocks.1.attn.to_q.bias" in aliased and len(aliased) >= 0.5 * 6 * 4
    assert not any("add_q_proj" in n for n in aliased) and any("add_q_proj" in n for n in pk.sources)     # frozen: copied once
    # an update behind torch's back (fk_adamw_step writes through raw pointers) shows through the aliased operand at once ...
    views["single_transformer_blocks.1.attn.to_k.weight"].view(torch.int16).add_(1)
    assert m.packed() is pk and not torch.equal(pk.single[1].wqkv, ref_w)
    assert torch.equal(pk.single[1].wqkv[D:2 * D], m.p("single_transformer_blocks.1.attn.to_k.weight").data)
    # ... and repack() refreshes, in place, the copies built from rewritten tensors (none of the trainable ones here, unless a
    # bucket boundary split a triple)
    ptrs = [t.data_ptr() for t, _ in pk.copies]
    for t, ns in pk.copies:
        if any(n in set(names) for n in ns):
            m.p(ns[0]).data.view(torch.int16).add_(1)
    m.repack(set(names))
    assert m.packed() is pk and ptrs == [t.data_ptr() for t, _ in pk.copies]
    for t, ns in pk.copies:
        assert torch.equal(t, torch.cat([m.p(n).data for n in ns]))
    # moving an aliased parameter rebuilds the packs
    m.p("single_transformer_blocks.1.attn.to_q.weight").data = m.p("single_transformer_blocks.1.attn.to_q.weight").data.clone()
    pk2 = m.packed()
    assert pk2 is not pk and "single_transformer_blocks.1.attn.to_q.weight" not in pk2.aliased
