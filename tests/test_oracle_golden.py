"""CPU tests: the oracle and the product's host logic against the committed golden vectors.

helpers.npz / anyres.npz were produced by EXECUTING THE REFERENCE's own functions (oracle/make_golden.py);
torch_ops.npz pins torch's CPU definitions of the building-block ops; oracle_selfpin.npz guards the
(upstream-unpinned) MMDiT / VAE / scheduler restatement against accidental edits.
"""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from gpt_image_edit_amd import anyres_util, flux_spec, helpers as phelpers
from gpt_image_edit_amd.scheduler import FlowMatchEulerDiscreteScheduler
from oracle import helpers as ohelpers
from oracle import mmdit, scheduler as osched, vae as ovae


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_pack_unpack_ids_shift_match_reference(golden_dir):
    g = _load(golden_dir, "helpers.npz")
    x = torch.from_numpy(g["pack_in"])
    for impl_pack, impl_unpack in ((ohelpers.pack_latents, ohelpers.unpack_latents),
                                   (lambda t: phelpers._pack_latents(t, *t.shape), phelpers._unpack_latents)):
        packed = impl_pack(x)
        assert np.array_equal(packed.numpy(), g["pack_out"])
        assert np.array_equal(impl_unpack(packed, 64, 96, 8).numpy(), g["unpack_out"])
        assert np.array_equal(impl_unpack(torch.from_numpy(g["unpack2_in"]), 32, 48, 8).numpy(), g["unpack2_out"])
    assert np.array_equal(ohelpers.prepare_latent_image_ids(4, 6).numpy(), g["ids_4x6"])
    assert np.array_equal(phelpers._prepare_latent_image_ids(1, 4, 6, "cpu", torch.float32).numpy(), g["ids_4x6"])
    ids_bf = phelpers._prepare_latent_image_ids(3, 64, 64, "cpu", torch.bfloat16).float().numpy()
    assert np.array_equal(ids_bf, g["ids_64x64_bf16"])
    assert np.array_equal(ohelpers.prepare_latent_image_ids(64, 64, torch.bfloat16).float().numpy(), g["ids_64x64_bf16"])
    for fn in (ohelpers.calculate_shift, phelpers.calculate_shift):
        mu = np.array([fn(int(s)) for s in g["shift_seq"]])
        assert np.array_equal(mu, g["shift_mu"])
        mu2 = np.array([fn(int(s), 256, 4096, 0.5, 1.16) for s in g["shift_seq"]])
        assert np.array_equal(mu2, g["shift_mu_custom"])
    assert abs(ohelpers.calculate_shift(4096) - 1.15) < 1e-12 and abs(ohelpers.calculate_shift(1024) - 0.63) < 5e-3


def test_pack_unpack_roundtrip_edge_cases():
    for shape in [(1, 16, 2, 2), (3, 16, 128, 128), (2, 4, 6, 10)]:
        x = torch.randn(*shape)
        p = ohelpers.pack_latents(x)
        assert p.shape == (shape[0], shape[2] * shape[3] // 4, shape[1] * 4)
        assert torch.equal(ohelpers.unpack_latents(p, shape[2] * 8, shape[3] * 8), x)
        assert torch.equal(phelpers._unpack_latents(phelpers._pack_latents(x, *shape), shape[2] * 8, shape[3] * 8, 8), x)


def test_anyres_matches_reference(golden_dir):
    g = _load(golden_dir, "anyres.npz")
    modes = [str(m) for m in g["modes"]]
    for row in g["table"]:
        h, w, mi, anchor, rw, rh, nh, nw, ch, cw, mh, mw = (int(v) for v in row)
        for mod in (ohelpers, anyres_util):
            assert mod.pick_ratio(h, w, modes[mi]) == (rw, rh)
            assert mod.dynamic_resize(h, w, modes[mi], anchor_pixels=anchor) == (nh, nw)
            assert mod.compute_size(rw, rh, 32, anchor_pixels=anchor) == (ch, cw)
            assert mod.compute_size(rw, rh, 32, min_pixels=256 * 256, max_pixels=768 * 768) == (mh, mw)
    # the cli default: 1024x1024 anchor, any_11ratio (cli.py:82-97)
    assert anyres_util.dynamic_resize(768, 1024, "any_11ratio", anchor_pixels=1024 * 1024) == ohelpers.dynamic_resize(
        768, 1024, "any_11ratio", anchor_pixels=1024 * 1024)


def test_resolution_quirks_F6_F7():
    # F6: any requested size is rescaled to max_area (default 1024^2) and floored to multiples of 16
    assert ohelpers.kontext_target_size(512, 512) == (1024, 1024)
    assert phelpers.fit_to_max_area(512, 512, 1024 ** 2, 16) == (1024, 1024)
    assert phelpers.fit_to_max_area(512, 512, 512 ** 2, 16) == (512, 512)
    assert phelpers.fit_to_max_area(720, 1280, 1024 ** 2, 16) == ohelpers.kontext_target_size(720, 1280)
    # F7: a pixel condition image snaps to the nearest preferred ~1 MP resolution
    assert ohelpers.preferred_condition_size(512, 512) == (1024, 1024)
    assert phelpers.preferred_condition_size(512, 512, 16) == (1024, 1024)
    assert phelpers.preferred_condition_size(512, 512, 16, auto_resize=False) == (512, 512)
    for h, w in [(480, 854), (1080, 1920), (1000, 333), (37, 41)]:
        assert phelpers.preferred_condition_size(h, w, 16) == ohelpers.preferred_condition_size(h, w)


def test_torch_building_blocks_are_pinned(golden_dir):
    g = _load(golden_dir, "torch_ops.npz")
    x = torch.from_numpy(g["x"])
    tol = dict(rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(F.gelu(x, approximate="tanh"), torch.from_numpy(g["gelu_tanh"]), **tol)
    torch.testing.assert_close(F.silu(x), torch.from_numpy(g["silu"]), **tol)
    torch.testing.assert_close(mmdit.layer_norm(x), torch.from_numpy(g["layer_norm"]), **tol)
    img = torch.from_numpy(g["gn_in"])
    gn = F.group_norm(img, 32, torch.from_numpy(g["gn_w"]), torch.from_numpy(g["gn_b"]), 1e-6)
    torch.testing.assert_close(gn, torch.from_numpy(g["group_norm"]), rtol=1e-4, atol=1e-5)
    q, k, v = (torch.from_numpy(g[n]) for n in "qkv")
    torch.testing.assert_close(F.scaled_dot_product_attention(q, k, v), torch.from_numpy(g["sdpa"]), rtol=1e-4, atol=1e-5)
    ref = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(16), -1) @ v
    torch.testing.assert_close(ref, torch.from_numpy(g["sdpa"]), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(F.interpolate(img[:, :4], scale_factor=2.0, mode="nearest"), torch.from_numpy(g["nearest2x"]))
    w, b = torch.from_numpy(g["conv_w"]), torch.from_numpy(g["conv_b"])
    torch.testing.assert_close(F.conv2d(img, w, b, padding=1), torch.from_numpy(g["conv3x3"]), rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(F.conv2d(F.pad(img, (0, 1, 0, 1)), w, b, stride=2), torch.from_numpy(g["conv3x3_s2"]),
                               rtol=1e-4, atol=1e-4)


def test_oracle_selfpin(golden_dir):
    g = _load(golden_dir, "oracle_selfpin.npz")
    cfg = dict(num_layers=2, num_single_layers=2, attention_head_dim=16, num_attention_heads=4,
               joint_attention_dim=32, pooled_projection_dim=24, in_channels=16, out_channels=16,
               axes_dims_rope=(4, 6, 6))
    sd = flux_spec.synthetic_state(flux_spec.flux_param_shapes(cfg), seed=3)
    gen = torch.Generator().manual_seed(5)
    hs = torch.randn(2, 12, 16, generator=gen)
    enc = torch.randn(2, 5, 32, generator=gen)
    pooled = torch.randn(2, 24, generator=gen)
    img_ids = torch.cat([ohelpers.prepare_latent_image_ids(2, 3), ohelpers.prepare_latent_image_ids(2, 3, first=1.0)])
    txt_ids = torch.zeros(5, 3)
    y = mmdit.flux_forward(sd, hs, enc, pooled, torch.tensor([0.75, 0.31]), img_ids, txt_ids, torch.tensor([3.5, 1.0]),
                           config=cfg)
    torch.testing.assert_close(y, torch.from_numpy(g["mmdit_tiny_out"]), rtol=1e-4, atol=1e-5)
    cos, sin = mmdit.rope_tables(torch.cat([txt_ids, img_ids]), (4, 6, 6))
    torch.testing.assert_close(cos, torch.from_numpy(g["rope_cos"]))
    torch.testing.assert_close(sin, torch.from_numpy(g["rope_sin"]))
    torch.testing.assert_close(mmdit.sinusoid_256(torch.tensor([0.0, 1.0, 750.0, 3504.0])), torch.from_numpy(g["sinusoid"]),
                               rtol=1e-5, atol=1e-6)
    vcfg = dict(block_out_channels=(32, 32, 64, 64), latent_channels=4)
    vsd = flux_spec.synthetic_state(flux_spec.vae_param_shapes(vcfg), seed=4)
    z = torch.randn(1, 4, 4, 6, generator=gen)
    torch.testing.assert_close(ovae.decode(vsd, z), torch.from_numpy(g["vae_dec_tiny"]), rtol=1e-4, atol=1e-5)
    im = torch.rand(1, 3, 32, 48, generator=gen) * 2 - 1
    torch.testing.assert_close(ovae.encode_moments(vsd, im), torch.from_numpy(g["vae_enc_tiny"]), rtol=1e-4, atol=1e-5)


def test_scheduler_host_logic(golden_dir):
    g = _load(golden_dir, "oracle_selfpin.npz")
    ts, sg = osched.shifted_sigmas(28, 1.15)
    assert np.array_equal(ts.numpy(), g["sched_timesteps_mu1.15"]) and np.array_equal(sg.numpy(), g["sched_sigmas_mu1.15"])
    s = FlowMatchEulerDiscreteScheduler()
    s.set_timesteps(sigmas=np.linspace(1.0, 1 / 28, 28), mu=1.15, device="cpu")
    assert np.array_equal(s.timesteps.numpy(), ts.numpy()) and np.array_equal(s.sigmas.numpy(), sg.numpy())
    assert s.order == 1 and s.config.get("base_image_seq_len", 256) == 256 and s.config.get("max_shift") == 1.15
    # first sigma is exactly 1 (t = 1000), last appended sigma is 0, strictly decreasing
    assert float(sg[0]) == 1.0 and float(sg[-1]) == 0.0 and bool((sg[1:] < sg[:-1]).all())
    for i in range(28):
        assert s.dsigma(i) == float(sg[i + 1] - sg[i])
    with pytest.raises(ValueError):
        FlowMatchEulerDiscreteScheduler().set_timesteps(sigmas=[1.0, 0.5], device="cpu")  # mu missing
    # closed form of the dynamic shift
    mu = 0.63
    ts4, sg4 = osched.shifted_sigmas(4, mu)
    lin = np.linspace(1.0, 0.25, 4)
    np.testing.assert_allclose(sg4[:-1].numpy(), math.exp(mu) / (math.exp(mu) + (1 / lin - 1)), rtol=1e-6)


def test_euler_step_reference_rounding():
    # 0-dim fp32 sigma times a bf16 tensor is formed in bf16 (torch promotion), then added in fp32
    v = torch.tensor([1.5, -2.25, 0.3333], dtype=torch.bfloat16)
    x = torch.tensor([0.1, 0.2, 0.3], dtype=torch.bfloat16)
    out = osched.euler_step(v, torch.tensor(0.9), torch.tensor(0.8765), x)
    assert out.dtype == torch.bfloat16
    ds = (torch.tensor(0.8765) - torch.tensor(0.9)).to(torch.bfloat16).float()
    manual = (x.float() + (ds * v.float()).to(torch.bfloat16).float()).to(torch.bfloat16)
    assert torch.equal(out, manual)


def test_param_layout_matches_checkpoint_names():
    shapes = flux_spec.flux_param_shapes()
    assert shapes["transformer_blocks.18.norm1.linear.weight"] == (18432, 3072)
    assert shapes["single_transformer_blocks.37.proj_out.weight"] == (3072, 15360)
    assert shapes["time_text_embed.guidance_embedder.linear_1.weight"] == (3072, 256)
    assert shapes["proj_out.weight"] == (64, 3072) and shapes["context_embedder.weight"] == (3072, 4096)
    n = sum(math.prod(s) for s in shapes.values())
    assert 11.8e9 < n < 12.0e9  # ~11.9 B parameters
    v = flux_spec.vae_param_shapes()
    assert v["decoder.conv_in.weight"] == (512, 16, 3, 3) and v["encoder.conv_out.weight"] == (32, 512, 3, 3)
    assert "decoder.up_blocks.2.resnets.0.conv_shortcut.weight" in v and "decoder.up_blocks.0.resnets.0.conv_shortcut.weight" not in v
    pj = flux_spec.projector_param_shapes()
    assert pj["denoise_projector.0.weight"] == (12288, 3584) and pj["denoise_projector.2.weight"] == (4096, 12288)
    a = flux_spec.synthetic_state({"x.weight": (4, 4), "x.bias": (4,)}, seed=1)
    b = flux_spec.synthetic_state({"x.bias": (4,), "x.weight": (4, 4)}, seed=1)
    assert torch.equal(a["x.weight"], b["x.weight"]) and torch.equal(a["x.bias"], b["x.bias"])


# ---- G7: the reference cli's own pixel / size helpers (univa/serve/cli.py:82-116), executed by make_golden.py ------
def _write_pngs(tmp_path, arrays, prefix):
    from PIL import Image
    paths = []
    for i, a in enumerate(arrays):
        fn = str(tmp_path / f"{prefix}{i}.png")
        Image.fromarray(a).save(fn)
        paths.append(fn)
    return paths


def test_cli_prepare_condition_images_matches_reference(golden_dir, tmp_path):
    from gpt_image_edit_amd.serve import cli
    from oracle import vae as ovae
    g = _load(golden_dir, "cli.npz")
    u8, want = g["cond_u8"], torch.from_numpy(g["cond_f32"])
    got = cli.prepare_condition_images(_write_pngs(tmp_path, u8, "c"), "cpu")
    assert got.dtype == torch.float32 and torch.equal(got, want)
    px = cli.prepare_condition_pixels(_write_pngs(tmp_path, u8, "d"))
    assert px.dtype == torch.uint8 and np.array_equal(px.numpy(), u8)
    # the oracle's restatement of the whole float route, at native size: same numbers, rounded to bf16
    assert torch.equal(ovae.preprocess_uint8(torch.from_numpy(u8), 12, 20), want.to(torch.bfloat16))


def test_cli_update_size_matches_reference(golden_dir, tmp_path):
    from gpt_image_edit_amd.serve import cli
    g = _load(golden_dir, "cli.npz")
    files = _write_pngs(tmp_path, [np.zeros((h, w, 3), dtype=np.uint8) for w, h in g["update_size_wh"]], "s")
    for n, anchor, nh, nw in g["update_size_rows"]:
        args = [None, None] if n == 0 else ([files[0], None] if n == 1 else files)
        assert cli.update_size(args[0], args[1], "any_11ratio", int(anchor)) == (nh, nw)


def test_cli_flags_match_reference_defaults():
    """univa/serve/cli.py:271-283"""
    from gpt_image_edit_amd.serve import cli
    a = cli.build_parser().parse_args(["--model_path", "m", "--flux_path", "f"])
    assert (a.height, a.width, a.num_inference_steps, a.guidance_scale) == (1024, 1024, 28, 3.5)
    assert not (a.no_auto_hw or a.ocr_enhancer or a.no_joint_with_t5)


def test_scheduler_shift_pinned_to_the_references_own_restatement(golden_dir):
    """The one piece of the diffusers scheduler the reference restates IN-TREE: ``apply_flux_schedule_shift``
    (``train_denoiser.py:972-986``: sigma * e^mu / (1 + (e^mu - 1) sigma), mu = calculate_shift(h w / 4) with the
    scheduler config's base / max shift) -- algebraically the scheduler's dynamic shift e^mu / (e^mu + 1/sigma - 1).
    ``tests/golden/train.npz`` holds that function's outputs (executed from the reference's source by
    ``oracle/make_golden.py::g_train``); both ``oracle.scheduler.shifted_sigmas`` and the product's
    ``FlowMatchEulerDiscreteScheduler.set_timesteps`` must reproduce them from the same sigmas and resolutions."""
    from gpt_image_edit_amd import helpers
    g = _load(golden_dir, "train.npz")
    sig = g["shift_in"].astype(np.float32)
    cfg = FlowMatchEulerDiscreteScheduler().config
    for (h, w), ref in zip(g["shift_hw"], g["shift_out"]):
        mu = helpers.calculate_shift((int(h) * int(w)) // 4, cfg["base_image_seq_len"], cfg["max_image_seq_len"],
                                     cfg["base_shift"], cfg["max_shift"])          # flux_pipeline.py:994-999
        ts, sg = osched.shifted_sigmas(len(sig), mu, sigmas=sig)
        np.testing.assert_allclose(sg[:-1].numpy(), ref, rtol=3e-6, atol=1e-7)
        np.testing.assert_allclose(ts.numpy(), ref * 1000.0, rtol=3e-6, atol=1e-4)
        s = FlowMatchEulerDiscreteScheduler()
        s.set_timesteps(sigmas=sig, mu=mu, device="cpu")
        np.testing.assert_allclose(s.sigmas[:-1].numpy(), ref, rtol=3e-6, atol=1e-7)
        assert np.array_equal(s.sigmas.numpy(), sg.numpy()) and np.array_equal(s.timesteps.numpy(), ts.numpy())
        assert float(s.sigmas[-1]) == 0.0
    # mu at the two BASELINE resolutions, as the reference's calculate_shift gives them (helpers.npz pins the function)
    assert helpers.calculate_shift(4096) == pytest.approx(1.15) and helpers.calculate_shift(1024) == pytest.approx(0.63, abs=5e-3)


# ---- the diffusers pin (oracle/make_golden_diffusers.py; the fixtures can only be made where diffusers is installed) ----------
def _diffusers_fixture(golden_dir, name):
    path = os.path.join(golden_dir, f"diffusers_{name}.npz")
    if not os.path.exists(path):
        pytest.skip(f"parity unpinned (fixture absent): tests/golden/diffusers_{name}.npz is written by "
                    "`python oracle/make_golden_diffusers.py` where diffusers==0.32.2 is installed")
    return np.load(path)


def _oracle_mmdit_case(case, dt):
    from oracle import make_golden_diffusers as mg
    dtype = getattr(torch, dt)
    cfg = case["cfg"]
    sd = {k: v.to(dtype) for k, v in flux_spec.synthetic_state(flux_spec.flux_param_shapes(cfg), seed=case["weight_seed"]).items()}
    inp = mg.mmdit_inputs(case, dtype)
    return mmdit.flux_forward(sd, inp["hidden_states"], inp["encoder_hidden_states"], inp["pooled_projections"], inp["timestep"],
                              inp["img_ids"], inp["txt_ids"], inp["guidance"], config=cfg, return_intermediates=True)


@pytest.mark.parametrize("name", ["tiny", "width"])
def test_diffusers_mmdit_cases_run_on_the_oracle(name):
    """The cases of the pin script execute on the oracle (inputs regenerate from their seeds, every tapped tensor exists):
    what a maintainer's fixture will be compared with.  Not a parity claim."""
    from oracle import make_golden_diffusers as mg
    case = mg.MMDIT_CASES[name]
    y, inter = _oracle_mmdit_case(case, "float32")
    s_img = 2 * case["grid"][0] * case["grid"][1]
    assert y.shape == (case["batch"], s_img, case["cfg"].get("out_channels") or case["cfg"]["in_channels"]) and torch.isfinite(y).all()
    for i in range(case["cfg"]["num_layers"]):
        assert inter[f"double{i}.h"].shape[1] == s_img and inter[f"double{i}.c"].shape[1] == case["s_txt"]
    for i in range(case["cfg"]["num_single_layers"]):
        assert inter[f"single{i}.s"].shape[1] == s_img + case["s_txt"]


def test_oracle_matches_diffusers_fixture_mmdit(golden_dir):
    from oracle import make_golden_diffusers as mg
    g = _diffusers_fixture(golden_dir, "mmdit")
    for name, case in mg.MMDIT_CASES.items():
        for dt in case["dtypes"]:
            y, inter = _oracle_mmdit_case(case, dt)
            # fp32: two fp32 executions of the same graph (different kernels: 1e-4 relative); bf16: same rounding points, the
            # matmul accumulation order may differ by a bf16 ulp here and there
            tol = dict(rtol=1e-4, atol=1e-5) if dt == "float32" else dict(rtol=2e-2, atol=2e-2)
            torch.testing.assert_close(y.float(), torch.from_numpy(g[f"{name}.{dt}.out"]), **tol)
            for key, val in inter.items():
                fk = f"{name}.{dt}.{key}"
                if fk in g.files:
                    torch.testing.assert_close(val.float(), torch.from_numpy(g[fk]), **tol)
            n_taps = sum(1 for f in g.files if f.startswith(f"{name}.{dt}.double") or f.startswith(f"{name}.{dt}.single"))
            assert n_taps == 2 * case["cfg"]["num_layers"] + case["cfg"]["num_single_layers"]


def test_oracle_matches_diffusers_fixture_vae(golden_dir):
    from oracle import make_golden_diffusers as mg
    g = _diffusers_fixture(golden_dir, "vae")
    sd = flux_spec.synthetic_state(flux_spec.vae_param_shapes(mg.VAE_CASE["cfg"]), seed=mg.VAE_CASE["weight_seed"])
    z, im = mg.vae_inputs()
    torch.testing.assert_close(ovae.encode_moments(sd, im), torch.from_numpy(g["moments"]), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(ovae.encode_mode(sd, im), torch.from_numpy(g["mode"]), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(ovae.decode(sd, z), torch.from_numpy(g["decode"]), rtol=1e-4, atol=1e-5)


def test_oracle_matches_diffusers_fixture_scheduler(golden_dir):
    from oracle import make_golden_diffusers as mg
    g = _diffusers_fixture(golden_dir, "sched")
    for n, mu in mg.SCHED_CASES:
        ts, sg = osched.shifted_sigmas(n, mu)
        assert np.array_equal(ts.numpy(), g[f"timesteps.{n}.{mu}"]) and np.array_equal(sg.numpy(), g[f"sigmas.{n}.{mu}"])
        s = FlowMatchEulerDiscreteScheduler()                       # the product's host scheduler, same arguments
        s.set_timesteps(sigmas=np.linspace(1.0, 1 / n, n), mu=mu, device="cpu")
        assert np.array_equal(s.timesteps.numpy(), g[f"timesteps.{n}.{mu}"]) and np.array_equal(s.sigmas.numpy(), g[f"sigmas.{n}.{mu}"])
    n, mu = mg.SCHED_CASES[0]
    _, sg = osched.shifted_sigmas(n, mu)
    for dt in ("float32", "bfloat16"):
        v, x = (t.to(getattr(torch, dt)) for t in mg.sched_step_inputs())
        for i in range(3):
            x = osched.euler_step(v, sg[i], sg[i + 1], x)
            assert np.array_equal(x.float().numpy(), g[f"step{i}.{dt}"])    # one fma per element: bit-exact


def test_diffusers_vae_and_scheduler_cases_run_on_the_oracle():
    from oracle import make_golden_diffusers as mg
    sd = flux_spec.synthetic_state(flux_spec.vae_param_shapes(mg.VAE_CASE["cfg"]), seed=mg.VAE_CASE["weight_seed"])
    z, im = mg.vae_inputs()
    assert ovae.encode_moments(sd, im).shape == (1, 8, 4, 6) and ovae.decode(sd, z).shape == (1, 3, 32, 48)
    v, x = mg.sched_step_inputs()
    _, sg = osched.shifted_sigmas(*mg.SCHED_CASES[0])
    assert osched.euler_step(v.bfloat16(), sg[0], sg[1], x.bfloat16()).dtype == torch.bfloat16
    try:
        import diffusers  # noqa: F401
    except ImportError:
        with pytest.raises(SystemExit, match="diffusers is not installed"):      # the script says why it cannot run here
            mg.main()
