"""Host logic of the fp32-class VAE encoder (no GPU): how fp32 operands are cut into bf16 parts and how the packed
weights line up with them.  The K-concatenation [a_hi | a_lo | a_hi] . [w_hi | w_hi | w_lo] is evaluated here with plain
torch fp32 matmuls over the SAME packed tensors the HIP kernels read (vae._pack_conv_parts, HipAutoencoderKL._packed_f32)
and compared with the fp32 convolution / projection of the oracle -- a wrong part order or channel offset shows up here,
before any kernel runs.  Reference: train_denoiser.py:458 (VAE loaded in fp32), :887-918 (encodes inside the step)."""
import torch
import torch.nn.functional as F

from gpt_image_edit_amd import checkpoint, flux_spec
from gpt_image_edit_amd.vae import HipAutoencoderKL, _pack_conv_parts

BF = torch.bfloat16
SMALL_V = dict(block_out_channels=(32, 32, 64, 64), layers_per_block=1)


def _act_parts(x, P):
    """activation-side parts along the channel axis of an NHWC tensor: (hi, lo) or (hi, lo, hi)"""
    hi = x.to(BF).float()
    lo = (x - hi).to(BF).float()
    return torch.cat([hi, lo, hi][:P] if P == 3 else [hi, lo], dim=-1)


def _conv_from_packed(xp_nhwc, w_packed, cin_total, cout, ks, **kw):
    w = w_packed[:cout, : ks * ks * cin_total].float().view(cout, ks, ks, cin_total).permute(0, 3, 1, 2)
    return F.conv2d(xp_nhwc.permute(0, 3, 1, 2), w, **kw)


def test_packed_weight_parts_reproduce_the_fp32_convolution():
    g = torch.Generator().manual_seed(1)
    for cin, cout, cin_pad in [(64, 96, 64), (3, 128, 32)]:
        x = torch.randn(2, 7, 9, cin, generator=g)
        w = torch.randn(cout, cin, 3, 3, generator=g) * 0.05
        ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), padding=1).float()
        xpad = torch.zeros(2, 7, 9, cin_pad)
        xpad[..., :cin] = x
        hi = w.to(BF)
        lo = (w - hi.float()).to(BF)
        errs = {}
        for P, parts in ((3, [hi, hi, lo]), (2, [hi, hi])):
            xp = torch.cat([t for t in _act_parts(xpad, P).split(cin_pad, dim=-1)], dim=-1)     # [.., P * cin_pad]
            wp = _pack_conv_parts(parts, cin_pad, cout)
            assert wp.dtype == BF and wp.shape == (cout, (9 * P * cin_pad + 63) // 64 * 64)
            got = _conv_from_packed(xp, wp, P * cin_pad, cout, 3, padding=1)
            errs[P] = (got - ref).abs().max().item() / ref.abs().max().item()
        # three parts: everything but a_lo . w_lo (2^-16 class); two parts additionally drop a_hi . w_lo (2^-9 class)
        assert errs[3] < 3e-5, errs
        assert 1e-4 < errs[2] < 1e-2, errs
        # with weights that ARE bf16 numbers the two-part product is already of the 2^-16 class
        ref_bf = F.conv2d(x.permute(0, 3, 1, 2).double(), hi.double(), padding=1).float()
        got = _conv_from_packed(_act_parts(xpad, 2), _pack_conv_parts([hi, hi], cin_pad, cout), 2 * cin_pad, cout, 3, padding=1)
        assert (got - ref_bf).abs().max().item() / ref_bf.abs().max().item() < 3e-5


def _small_vae(fp32):
    cfg = dict(flux_spec.FLUX_VAE_CONFIG)
    cfg.update(SMALL_V)
    shapes = flux_spec.vae_param_shapes(cfg)
    sd32 = flux_spec.synthetic_state(shapes, seed=9, dtype=torch.float32)
    vae = HipAutoencoderKL(config=SMALL_V, device="cpu")
    if fp32:
        vae.load_fp32_state_dict(sd32)
    else:
        vae.load_state_dict({k: v.to(BF) for k, v in sd32.items()})
    return vae, sd32


def test_packed_f32_tables_follow_the_checkpoint():
    for fp32 in (True, False):
        vae, sd32 = _small_vae(fp32)
        pk = vae._packed_f32()
        assert pk["parts"] == (3 if fp32 else 2) and vae._packed_f32() is pk            # cached
        P = pk["parts"]
        want = (lambda k: sd32[k]) if fp32 else (lambda k: sd32[k].to(BF).float())
        # every encoder convolution: the packed parts reproduce the fp32 convolution of the checkpoint's weight
        g = torch.Generator().manual_seed(2)
        n_conv = 0
        for name, prm in vae.state_dict().items():
            if not (name.startswith("encoder.") and name.endswith(".weight") and prm.dim() == 4):
                continue
            base = name[: -len(".weight")]
            wp, b, cout_pad = pk[base]
            co, ci, ks, _ = prm.shape
            cin_pad = 32 if ci < 32 else ci
            assert b.dtype == torch.float32 and torch.equal(b[:co], want(base + ".bias")) and cout_pad % 8 == 0
            x = torch.zeros(1, 5, 6, cin_pad)
            x[..., :ci] = torch.randn(1, 5, 6, ci, generator=g)
            got = _conv_from_packed(_act_parts(x, P), wp, P * cin_pad, co, ks, padding=ks // 2)
            ref = F.conv2d(x[..., :ci].permute(0, 3, 1, 2).double(), want(name).double(), padding=ks // 2).float()
            assert (got - ref).abs().max().item() <= 3e-5 * ref.abs().max().item(), base
            n_conv += 1
        assert n_conv >= 12
        # norm vectors in fp32, projections as [N, P * K] over the activation parts
        gam, bet = pk["encoder.conv_norm_out"]
        assert gam.dtype == torch.float32 and torch.equal(gam, want("encoder.conv_norm_out.weight")) and torch.equal(bet, want("encoder.conv_norm_out.bias"))
        a = "encoder.mid_block.attentions.0."
        C = vae.p(a + "to_q.weight").shape[0]
        wqkv, bqkv = pk[a + "qkv"]
        assert wqkv.shape == (3 * C, P * C) and bqkv.shape == (3 * C,) and bqkv.dtype == torch.float32
        n = torch.randn(11, C, generator=g)
        got = _act_parts(n, P) @ wqkv.float().T + bqkv
        ref = torch.cat([n.double() @ want(a + f"{t}.weight").double().T + want(a + f"{t}.bias").double() for t in ("to_q", "to_k", "to_v")], dim=1).float()
        assert (got - ref).abs().max().item() <= 3e-5 * ref.abs().max().item()
        wo, bo = pk[a + "to_out.0"]
        assert wo.shape == (C, P * C) and torch.equal(bo, want(a + "to_out.0.bias"))
        # loading new weights drops the tables
        vae.load_state_dict({k: v.to(BF) for k, v in sd32.items()})
        assert vae._pk32 is None and vae._f32_state is None and vae._packed_f32()["parts"] == 2


def test_load_vae_fp32_reads_the_checkpoint_in_fp32(tmp_path):
    cfg = dict(flux_spec.FLUX_VAE_CONFIG)
    cfg.update(SMALL_V)
    tcfg = dict(flux_spec.FLUX_KONTEXT_CONFIG)
    tcfg.update(dict(num_layers=1, num_single_layers=1, num_attention_heads=2, joint_attention_dim=64, pooled_projection_dim=32))
    v32 = flux_spec.synthetic_state(flux_spec.vae_param_shapes(cfg), seed=3, dtype=torch.float32)
    tstate = flux_spec.synthetic_state(flux_spec.flux_param_shapes(tcfg), seed=4, dtype=BF)
    d = str(tmp_path / "flux")
    checkpoint.save_flux_directory(d, tstate, v32, tcfg, cfg)
    vae = HipAutoencoderKL(config=SMALL_V, device="cpu")
    checkpoint.load_vae(vae, d, fp32=True)
    assert vae._f32_state is not None and all(k.startswith("encoder.") or k.startswith("quant_conv") for k in vae._f32_state)
    k = "encoder.conv_in.weight"
    assert torch.equal(vae._f32_state[k], v32[k]) and torch.equal(vae.p(k), v32[k].to(BF)) and vae.p(k).dtype == BF
    assert vae._packed_f32()["parts"] == 3
    checkpoint.load_vae(vae, d)                                   # the bf16 load forgets the fp32 values
    assert vae._f32_state is None and vae._packed_f32()["parts"] == 2


def test_fp32_state_follows_the_module():
    vae, sd32 = _small_vae(True)
    vae.to("cpu")                                    # a device move (here a no-op) keeps the fp32 values and drops the packed tables
    assert vae._pk32 is None and vae._f32_state is not None
    k = "encoder.conv_in.weight"
    assert vae._f32_state[k].dtype == torch.float32 and torch.equal(vae._f32_state[k], sd32[k])
    assert vae._packed_f32()["parts"] == 3


def test_fp32_encode_refuses_cpu_tensors_and_nhwc_input():
    import pytest
    vae, _ = _small_vae(True)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        vae.encode(torch.zeros(1, 3, 16, 16), fp32=True)
