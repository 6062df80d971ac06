"""GPU parity of the whole HIP MMDiT forward (HipFluxTransformer2DModel) against the CPU oracle.

The model is built at FULL width (D = 3072, 24 heads, FF 12288, joint dim 4096) but reduced depth so
the fp32 CPU oracle finishes in seconds; weights are the seeded synthetic set with the real
checkpoint key names.  Two comparisons:
  * vs the oracle run in bf16 (= the reference's rounding points): tight, element-wise;
  * vs the oracle run in fp32 on the same bf16-rounded weights/inputs: bounded by bf16 round-off
    accumulated over the blocks (reported; asserted relative to the output scale).
"""
import pytest
import torch

from conftest import report

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def _inputs(B, S_txt, h, w, cfg, seed=0):
    from oracle.helpers import prepare_latent_image_ids
    g = torch.Generator().manual_seed(seed)
    S_tgt = h * w
    hs = torch.randn(B, 2 * S_tgt, cfg["in_channels"], generator=g).to(BF)
    enc = torch.randn(B, S_txt, cfg["joint_attention_dim"], generator=g).to(BF)
    pooled = torch.randn(B, cfg["pooled_projection_dim"], generator=g).to(BF)
    # t*1000 and g*1000 are chosen exactly representable in bf16 (500, 250, 4000): the model's
    # `.to(bf16) * 1000` then rounds identically in the fp32 and bf16 oracles, so their difference is
    # pure arithmetic round-off (3.5 -> 3504 vs 3500 would instead change the embedding itself)
    t = torch.tensor([0.5, 0.25][:B]).to(BF)
    gd = torch.full((B,), 4.0)
    img_ids = torch.cat([prepare_latent_image_ids(h, w), prepare_latent_image_ids(h, w, first=1.0)])
    txt_ids = torch.zeros(S_txt, 3)
    return hs, enc, pooled, t, gd, img_ids, txt_ids


@pytest.mark.parametrize("n_double,n_single,B,S_txt,h,w", [(1, 0, 1, 64, 8, 8), (0, 1, 1, 64, 8, 8),
                                                            (2, 2, 2, 77, 10, 12)])
def test_mmdit_forward_matches_oracle(n_double, n_single, B, S_txt, h, w):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from gpt_image_edit_amd import flux_spec
    from gpt_image_edit_amd.transformer import HipFluxTransformer2DModel
    from oracle import mmdit

    cfg = dict(flux_spec.FLUX_KONTEXT_CONFIG, num_layers=n_double, num_single_layers=n_single)
    sd32 = flux_spec.synthetic_state(flux_spec.flux_param_shapes(cfg), seed=1)
    sd_bf = {k: v.to(BF) for k, v in sd32.items()}
    model = HipFluxTransformer2DModel(cfg, device="cuda")
    model.load_state_dict(sd_bf)
    hs, enc, pooled, t, gd, img_ids, txt_ids = _inputs(B, S_txt, h, w, cfg)

    out = model(hidden_states=hs.cuda(), timestep=t.cuda(), guidance=gd.cuda(), pooled_projections=pooled.cuda(),
                encoder_hidden_states=enc.cuda(), txt_ids=txt_ids.cuda(), img_ids=img_ids.cuda(),
                joint_attention_kwargs={}, return_dict=False)[0]
    torch.cuda.synchronize()
    out = out.cpu()
    assert out.shape == (B, 2 * h * w, 64) and not torch.isnan(out.float()).any()

    ref_bf = mmdit.flux_forward(sd_bf, hs, enc, pooled, t, img_ids, txt_ids, gd, config=cfg)
    sd_r = {k: v.float() for k, v in sd_bf.items()}  # fp32 math on the bf16-rounded weights
    ref32 = mmdit.flux_forward(sd_r, hs.float(), enc.float(), pooled.float(), t, img_ids, txt_ids, gd, config=cfg)
    tag = f"mmdit d{n_double}s{n_single}"
    d_bf = report(tag + " vs bf16-oracle", out, ref_bf)
    d_32 = report(tag + " vs fp32-oracle", out, ref32)
    d_ref = report(tag + " bf16-oracle vs fp32-oracle (round-off floor)", ref_bf, ref32)
    scale = ref32.abs().max().item()
    # the HIP path must be as close to the exact result as the reference's own bf16 execution is -- relative to that
    # floor only, no absolute arm (round 3 logs: mean within 1 % of the floor's, max 0.9-1.13 x the floor's)
    assert d_32.max().item() <= 1.25 * d_ref.max().item()
    assert d_32.mean().item() <= 1.1 * d_ref.mean().item()
    assert d_bf.max().item() <= 1.25 * d_ref.max().item()       # and no further from the bf16 oracle than that is from fp32


class _StreamedState:
    """The oracle's state dict for a model too deep to copy: every access reads ONE tensor back from the HIP model's own
    parameters (``model.state_dict()`` holds references to the device tensors) and converts it, so the host never holds
    more than the tensors of the block being evaluated (fp32 full depth would be 47.6 GB)."""

    def __init__(self, device_state, dtype):
        self.sd, self.dtype = device_state, dtype

    def __getitem__(self, k):
        return self.sd[k].detach().to("cpu").to(self.dtype)

    def get(self, k, default=None):
        return self[k] if k in self.sd else default

    def __contains__(self, k):
        return k in self.sd


def test_full_depth_mmdit_matches_block_streamed_oracle():
    """the real 19 + 38-block HipFluxTransformer2DModel (full width, every block its own weights) at
    S = 640 (128 text + 16 x 16 target + 16 x 16 condition tokens) against ``oracle.mmdit.flux_forward`` run block-streamed
    on the host in fp32 (exact arithmetic on the bf16-rounded weights) and in bf16 (the reference's rounding points).
    Error growth over 57 residual blocks is held to the bf16 oracle's own distance from fp32 -- no absolute arm.
    Matches flux_pipeline.py:1067-1077 at full depth."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import time
    from gpt_image_edit_amd import flux_spec
    from gpt_image_edit_amd.transformer import HipFluxTransformer2DModel
    from oracle import mmdit
    cfg = dict(flux_spec.FLUX_KONTEXT_CONFIG)
    assert cfg["num_layers"] == 19 and cfg["num_single_layers"] == 38
    model = HipFluxTransformer2DModel(cfg, device="cuda", init="synthetic", seed=17)
    hs, enc, pooled, t, gd, img_ids, txt_ids = _inputs(1, 128, 16, 16, cfg, seed=6)
    out = model(hidden_states=hs.cuda(), timestep=t.cuda(), guidance=gd.cuda(), pooled_projections=pooled.cuda(),
                encoder_hidden_states=enc.cuda(), txt_ids=txt_ids.cuda(), img_ids=img_ids.cuda(),
                joint_attention_kwargs={}, return_dict=False)[0]
    out2 = model(hidden_states=hs.cuda(), timestep=t.cuda(), guidance=gd.cuda(), pooled_projections=pooled.cuda(),
                 encoder_hidden_states=enc.cuda(), txt_ids=txt_ids.cuda(), img_ids=img_ids.cuda(), return_dict=False)[0]
    torch.cuda.synchronize()
    assert torch.equal(out, out2)
    out = out.cpu()
    assert out.shape == (1, 512, 64) and torch.isfinite(out.float()).all()
    dev_sd = model.state_dict()
    t0 = time.time()
    ref32, inter32 = mmdit.flux_forward(_StreamedState(dev_sd, torch.float32), hs.float(), enc.float(), pooled.float(), t,
                                        img_ids, txt_ids, gd, config=cfg, return_intermediates=True)
    t1 = time.time()
    ref_bf, inter_bf = mmdit.flux_forward(_StreamedState(dev_sd, BF), hs, enc, pooled, t, img_ids, txt_ids, gd, config=cfg,
                                          return_intermediates=True)
    print(f"[parity] full-depth oracle on the host: fp32 {t1 - t0:.1f} s, bf16 {time.time() - t1:.1f} s", flush=True)
    # how the bf16 execution drifts from the exact one over the 57 residual blocks (the floor the HIP path is held to)
    for name in ("double0.h", "double9.h", "double18.h", "single0.s", "single18.s", "single37.s"):
        a, b = inter_bf[name].float(), inter32[name]
        print(f"[parity] floor growth {name:10s}: bf16-oracle vs fp32-oracle max {(a - b).abs().max().item():.3e} "
              f"mean {(a - b).abs().mean().item():.3e} at scale {b.abs().max().item():.2f}", flush=True)
    d_bf = report("mmdit d19s38 (full depth) vs bf16-oracle", out, ref_bf)
    d_32 = report("mmdit d19s38 (full depth) vs fp32-oracle", out, ref32)
    d_ref = report("mmdit d19s38 (full depth) bf16-oracle vs fp32-oracle (round-off floor)", ref_bf, ref32)
    assert d_32.max().item() <= 1.25 * d_ref.max().item()
    assert d_32.mean().item() <= 1.1 * d_ref.mean().item()
    assert d_bf.max().item() <= 1.25 * d_ref.max().item() and d_bf.mean().item() <= 1.1 * d_ref.mean().item()
    _direct_full_depth_bound("mmdit d19s38 S=640", d_bf, ref_bf)


# Direct HIP-vs-bf16-oracle bound after 57 residual blocks, as fractions of the output's largest magnitude -- independent of
# the bf16-vs-fp32 floor (VERDICT r4 next #1a).  <= 2 x the r04 log: max 7.8e-2, mean 1.56e-2 at scale 4.09 (1.9 % / 0.38 %).
FULL_DEPTH_MAX, FULL_DEPTH_MEAN = 2.0 ** -5, 0.0075


def _direct_full_depth_bound(name, d_bf, ref_bf):
    scale = ref_bf.float().abs().max().item()
    mx, mn = d_bf.max().item(), d_bf.mean().item()
    print(f"[parity] {name}: direct bound vs bf16-oracle: max {mx:.3e} <= {FULL_DEPTH_MAX * scale:.3e}, mean {mn:.3e} <= "
          f"{FULL_DEPTH_MEAN * scale:.3e} (scale {scale:.2f})", flush=True)
    assert mx <= FULL_DEPTH_MAX * scale and mn <= FULL_DEPTH_MEAN * scale, f"{name}: HIP path disagrees with the bf16 oracle"


@pytest.mark.timeout(1200, method="thread")
def test_full_depth_mmdit_at_cfg2_size_matches_bf16_oracle():
    """VERDICT r4 next #1b: the real 19 + 38-block model AT BASELINE configs[1]'s own shape -- S = 2560 = 512 text + 32 x 32
    target + 32 x 32 condition tokens -- against the block-streamed bf16 oracle (the fp32 pass is left to the S = 640 case:
    it is what makes that one take a minute).  Every launch plan the 512^2 edit uses is on this path: mixed 256 x 256 /
    256 x 128 grid for the fused QKV projection, split-K pairs for the K = 12288 / 15360 projections, the one-round
    attention grid.  Matches flux_pipeline.py:1054-1120 / SURVEY Appendix A.1 at the cfg 2 shape."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import time
    from gpt_image_edit_amd import flux_spec
    from gpt_image_edit_amd.transformer import HipFluxTransformer2DModel
    from oracle import mmdit
    cfg = dict(flux_spec.FLUX_KONTEXT_CONFIG)
    model = HipFluxTransformer2DModel(cfg, device="cuda", init="synthetic", seed=23)
    hs, enc, pooled, t, gd, img_ids, txt_ids = _inputs(1, 512, 32, 32, cfg, seed=9)
    assert enc.shape[1] + hs.shape[1] == 2560
    out = model(hidden_states=hs.cuda(), timestep=t.cuda(), guidance=gd.cuda(), pooled_projections=pooled.cuda(),
                encoder_hidden_states=enc.cuda(), txt_ids=txt_ids.cuda(), img_ids=img_ids.cuda(), return_dict=False)[0]
    torch.cuda.synchronize()
    out = out.cpu()
    assert torch.isfinite(out.float()).all()
    t0 = time.time()
    ref_bf = mmdit.flux_forward(_StreamedState(model.state_dict(), BF), hs, enc, pooled, t, img_ids, txt_ids, gd, config=cfg)
    print(f"[parity] full-depth bf16 oracle at S = 2560 on the host: {time.time() - t0:.1f} s", flush=True)
    d_bf = report("mmdit d19s38 (full depth, S = 2560) vs bf16-oracle", out, ref_bf)
    _direct_full_depth_bound("mmdit d19s38 S=2560", d_bf, ref_bf)


def test_forward_outputs_do_not_alias():
    """Two sequential calls (the reference pipeline's true-CFG branch keeps noise_pred while computing
    neg_noise_pred, flux_pipeline.py:1067-1095) must return distinct tensors."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from gpt_image_edit_amd import flux_spec
    from gpt_image_edit_amd.transformer import HipFluxTransformer2DModel
    cfg = dict(flux_spec.FLUX_KONTEXT_CONFIG, num_layers=1, num_single_layers=0)
    m = HipFluxTransformer2DModel(cfg, device="cuda", init="synthetic", seed=8)
    hs, enc, pooled, t, gd, img_ids, txt_ids = (x.cuda() for x in _inputs(1, 40, 6, 8, cfg, seed=4))
    kw = dict(hidden_states=hs, pooled_projections=pooled, timestep=t[:1], guidance=gd[:1], txt_ids=txt_ids,
              img_ids=img_ids, return_dict=False)
    pos = m(encoder_hidden_states=enc, **kw)[0]
    keep = pos.clone()
    neg = m(encoder_hidden_states=-enc, **kw)[0]
    assert pos.data_ptr() != neg.data_ptr()
    assert torch.equal(pos, keep) and not torch.equal(pos, neg)


def test_second_stream_gives_the_same_bits():
    """The single blocks' MLP-up GEMM on the second stream (transformer.OVERLAP_MLP; chosen automatically at S = 5632 /
    8704, B = 1) is the same launches in another order: forced on and forced off must agree bit for bit, call after
    call (the event pair per block is what orders the two streams)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from gpt_image_edit_amd import flux_spec, transformer
    cfg = dict(flux_spec.FLUX_KONTEXT_CONFIG, num_layers=1, num_single_layers=3)
    model = transformer.HipFluxTransformer2DModel(cfg, device="cuda", init="synthetic", seed=3)
    hs, enc, pooled, t, gd, img_ids, txt_ids = _inputs(1, 200, 24, 24, cfg, seed=5)       # S = 1352
    kw = dict(hidden_states=hs.cuda(), timestep=t.cuda(), guidance=gd.cuda(), pooled_projections=pooled.cuda(),
              encoder_hidden_states=enc.cuda(), txt_ids=txt_ids.cuda(), img_ids=img_ids.cuda(), return_dict=False)
    saved = transformer.OVERLAP_MLP
    try:
        outs = []
        for mode in (False, True, True, False, True):
            transformer.OVERLAP_MLP = mode
            outs.append(model(**kw)[0].clone())
        torch.cuda.synchronize()
    finally:
        transformer.OVERLAP_MLP = saved
    assert torch.isfinite(outs[0].float()).all()
    for o in outs[1:]:
        assert torch.equal(o, outs[0])


def test_state_dict_roundtrip_and_seam():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from gpt_image_edit_amd import flux_spec
    from gpt_image_edit_amd.transformer import HipFluxTransformer2DModel
    cfg = dict(flux_spec.FLUX_KONTEXT_CONFIG, num_layers=1, num_single_layers=1)
    m = HipFluxTransformer2DModel(cfg, device="cuda", init="synthetic", seed=2)
    sd = m.state_dict()
    assert set(sd) == set(flux_spec.flux_param_shapes(cfg))
    assert sd["transformer_blocks.0.attn.to_q.weight"].shape == (3072, 3072)
    assert m.config.in_channels == 64 and m.config.guidance_embeds and m.dtype == BF
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(hidden_states=torch.zeros(1, 4, 64), encoder_hidden_states=torch.zeros(1, 4, 4096))


def test_prepare_conditioning_is_bit_identical():
    """All-steps modulation precompute (one weight-streaming pass) must reproduce the per-step path exactly."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from gpt_image_edit_amd import flux_spec
    from gpt_image_edit_amd.transformer import HipFluxTransformer2DModel
    cfg = dict(flux_spec.FLUX_KONTEXT_CONFIG, num_layers=1, num_single_layers=1)
    m = HipFluxTransformer2DModel(cfg, device="cuda", init="synthetic", seed=5)
    hs, enc, pooled, _, gd, img_ids, txt_ids = _inputs(2, 40, 6, 8, cfg, seed=3)
    hs, enc, pooled, gd, img_ids, txt_ids = (x.cuda() for x in (hs, enc, pooled, gd, img_ids, txt_ids))
    steps = (torch.tensor([[1.0, 1.0], [0.8516, 0.8516], [0.4375, 0.4375]]).to(BF)).cuda()
    kw = dict(hidden_states=hs, encoder_hidden_states=enc, pooled_projections=pooled, guidance=gd, txt_ids=txt_ids,
              img_ids=img_ids, return_dict=False)
    plain = [m(timestep=steps[i].clone(), **kw)[0].clone() for i in range(3)]
    m.prepare_conditioning(steps, gd, pooled)
    cached = [m(timestep=steps[i], **kw)[0].clone() for i in range(3)]
    for a, b in zip(plain, cached):
        assert torch.equal(a, b)
    # a timestep tensor that is not part of the prepared schedule falls back to on-the-fly conditioning
    other = m(timestep=torch.tensor([0.25, 0.25]).to(BF).cuda(), **kw)[0]
    assert not torch.equal(other, cached[0])


def test_denoise_projector_matches_oracle():
    """a12: Linear(3584,12288) -> SiLU -> Linear(12288,4096) (modeling_univa_denoise_tower.py:31-47)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from conftest import bf16_ulp_diff
    from gpt_image_edit_amd import flux_spec
    from gpt_image_edit_amd.projector import HipDenoiseProjector
    from oracle import mmdit as omm
    shapes = flux_spec.projector_param_shapes()
    sd = {k: v.to(torch.bfloat16) for k, v in flux_spec.synthetic_state(shapes, seed=5).items()}
    proj = HipDenoiseProjector(device="cuda")
    proj.load_state_dict({k[len("denoise_projector."):]: v for k, v in sd.items()})
    g = torch.Generator().manual_seed(2)
    x = torch.randn(1, 300, 3584, generator=g).to(torch.bfloat16)
    got = proj(x.cuda()).cpu()
    ref = omm.denoise_projector(sd, x)
    ref32 = omm.denoise_projector({k: v.float() for k, v in sd.items()}, x.float())
    assert got.shape == (1, 300, 4096)
    ulp = bf16_ulp_diff(got, ref)
    frac1, worst = (ulp <= 1).float().mean().item(), ulp.max().item()
    print(f"projector: {frac1 * 100:.3f}% within 1 bf16 ulp of the bf16 oracle, max {worst} ulp")
    # two chained GEMMs (K = 3584, 12288): a 1-ulp difference in the hidden activation moves a few outputs by >1 ulp
    assert frac1 > 0.99
    assert (got.float() - ref32).abs().max().item() <= 2.0 * (ref.float() - ref32).abs().max().item() + 1e-3
    assert torch.equal(proj(x[0].cuda()).cpu(), got[0])       # [L, 3584] input form


def test_checkpoint_roundtrip_into_hip_models(tmp_path):
    """checkpoint.save_* -> load_* fills the HIP modules; forward is bit-identical to load_state_dict."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from gpt_image_edit_amd import checkpoint, flux_spec
    from gpt_image_edit_amd.transformer import HipFluxTransformer2DModel
    cfg = dict(flux_spec.FLUX_KONTEXT_CONFIG, num_layers=1, num_single_layers=1)
    a = HipFluxTransformer2DModel(cfg, device="cuda", init="synthetic", seed=9)
    d = str(tmp_path / "uw")
    checkpoint.save_uniworld_directory(d, a.state_dict(), max_shard_bytes=1 << 28)
    b = HipFluxTransformer2DModel(cfg, device="cuda")
    checkpoint.load_flux_transformer(b, d)
    g = torch.Generator(device="cuda").manual_seed(4)
    BFl = torch.bfloat16
    hs = torch.randn(1, 128, 64, generator=g, device="cuda").to(BFl)
    enc = torch.randn(1, 64, 4096, generator=g, device="cuda").to(BFl)
    pooled = torch.randn(1, 768, generator=g, device="cuda").to(BFl)
    from gpt_image_edit_amd.helpers import _prepare_latent_image_ids as ids
    kw = dict(hidden_states=hs, encoder_hidden_states=enc, pooled_projections=pooled,
              timestep=torch.tensor([0.5], device="cuda").to(BFl), guidance=torch.full((1,), 3.5, device="cuda"),
              txt_ids=torch.zeros(64, 3, device="cuda", dtype=BFl), img_ids=ids(1, 8, 16, "cuda", BFl), return_dict=False)
    assert torch.equal(a(**kw)[0], b(**kw)[0])


def test_block_entry_points_give_the_same_bits():
    """SURVEY 8b / VERDICT r3 missing #6: the block-level C entry points (fk_double_block_fwd, fk_single_block_fwd,
    fk_mmdit_blocks_fwd: one call per block / per forward) enqueue the launches of the per-kernel path with the same
    arguments: the forward must agree bit for bit in all three forms, at a ragged shape with batch 2 and at a shape whose
    K-long GEMMs run as split-K pairs, and call after call (the argument structs are cached)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from gpt_image_edit_amd import flux_spec, transformer
    cfg = dict(flux_spec.FLUX_KONTEXT_CONFIG, num_layers=2, num_single_layers=3)
    model = transformer.HipFluxTransformer2DModel(cfg, device="cuda", init="synthetic", seed=23)
    saved = transformer.BLOCK_API
    try:
        for (B, S_txt, h, w) in ((2, 77, 10, 12), (1, 200, 24, 24)):
            hs, enc, pooled, t, gd, img_ids, txt_ids = _inputs(B, S_txt, h, w, cfg, seed=9)
            kw = dict(hidden_states=hs.cuda(), timestep=t.cuda(), guidance=gd.cuda(), pooled_projections=pooled.cuda(),
                      encoder_hidden_states=enc.cuda(), txt_ids=txt_ids.cuda(), img_ids=img_ids.cuda(), return_dict=False)
            outs = []
            for api in (0, 1, 2, 2, 1, 0):
                transformer.BLOCK_API = api
                outs.append(model(**kw)[0].clone())
            torch.cuda.synchronize()
            assert torch.isfinite(outs[0].float()).all()
            for o in outs[1:]:
                assert torch.equal(o, outs[0])
        # the cached structs follow the weights: after a parameter is REPLACED (new storage) the C path must see the new one
        transformer.BLOCK_API = 2
        before = model(**kw)[0].clone()
        prm = model.p("single_transformer_blocks.1.proj_out.weight")
        prm.data = (prm.data.float() * 0.5).to(BF)
        after_c = model(**kw)[0].clone()
        transformer.BLOCK_API = 0
        after_py = model(**kw)[0].clone()
        assert not torch.equal(before, after_c) and torch.equal(after_c, after_py)
    finally:
        transformer.BLOCK_API = saved
