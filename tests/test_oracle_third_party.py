"""The arithmetic oracle against INDEPENDENT third-party implementations of the same layers.

``oracle/vae.py`` and ``oracle/mmdit.py`` restate diffusers 0.32.2, which is not installed here (and cannot be: no
network), so they cannot be pinned to diffusers itself.  What IS installed (here and on the GPU box) is transformers,
which carries its own implementations of the same published building blocks, written by other people from the same
lineage (the taming-transformers VQGAN encoder / decoder that diffusers' AutoencoderKL descends from; the DiT /
AdaLN-Zero blocks of Qwen2.5-Omni's token2wav model; the RMSNorm of the Qwen2 family).  These tests hold the oracle to
them on shared seeded weights, in fp32:

* the WHOLE FLUX VAE encoder and decoder topology (128/256/512/512 channels, 2 (+1) ResNet blocks per level, mid block
  with single-head attention, (0,1,0,1)-padded stride-2 downsampling, nearest-2x upsampling, GroupNorm(32, eps 1e-6)
  + SiLU heads) = ``JanusVQVAEEncoder`` / ``JanusVQVAEDecoder`` configured to those sizes with the per-level attention
  lists of the lowest resolution emptied (AutoencoderKL has attention in the mid block only);
* AdaLayerNorm-Zero (chunk order shift, scale, gate, shift_mlp, scale_mlp, gate_mlp; LN * (1 + scale) + shift) and
  the final AdaLayerNorm (chunk order scale, shift = diffusers' AdaLayerNormContinuous);
* the GELU(tanh) feed-forward; the interleaved-pair rotary convention; RMSNorm with the cast before the weight.

It is a pin to second opinions, not to diffusers: DESIGN.md section 5 keeps the "parity unpinned" label for the wiring
of the MMDiT blocks themselves.
"""
import pytest
import torch
import torch.nn.functional as F

from oracle import mmdit, vae

janus = pytest.importorskip("transformers.models.janus.modeling_janus")
omni = pytest.importorskip("transformers.models.qwen2_5_omni.modeling_qwen2_5_omni")


def _seed_module(mod, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in sorted(mod.named_parameters()):
            if name.endswith("norm1.weight") or name.endswith("norm2.weight") or name.endswith("norm.weight") \
                    or name.endswith("norm_out.weight"):
                p.copy_(1 + 0.1 * torch.randn(p.shape, generator=g))
            elif p.dim() > 1:
                p.copy_(torch.randn(p.shape, generator=g) * (0.7 / (p[0].numel() ** 0.5)))
            else:
                p.copy_(0.05 * torch.randn(p.shape, generator=g))
    return mod.eval()


def _resnet_sd(block, prefix, sd):
    for n in ("norm1", "conv1", "norm2", "conv2"):
        sd[prefix + n + ".weight"], sd[prefix + n + ".bias"] = getattr(block, n).weight.data, getattr(block, n).bias.data
    if block.in_channels != block.out_channels:
        sd[prefix + "conv_shortcut.weight"] = block.nin_shortcut.weight.data
        sd[prefix + "conv_shortcut.bias"] = block.nin_shortcut.bias.data


def _mid_sd(mid, prefix, sd):
    _resnet_sd(mid.block_1, prefix + "resnets.0.", sd)
    _resnet_sd(mid.block_2, prefix + "resnets.1.", sd)
    a, p = mid.attn_1, prefix + "attentions.0."
    sd[p + "group_norm.weight"], sd[p + "group_norm.bias"] = a.norm.weight.data, a.norm.bias.data
    for src, dst in (("q", "to_q"), ("k", "to_k"), ("v", "to_v"), ("proj_out", "to_out.0")):
        conv = getattr(a, src)
        sd[p + dst + ".weight"], sd[p + dst + ".bias"] = conv.weight.data[:, :, 0, 0], conv.bias.data   # 1x1 conv == Linear


def _janus_cfg():
    from transformers.models.janus.configuration_janus import JanusVQVAEConfig
    return JanusVQVAEConfig(base_channels=128, channel_multiplier=(1, 2, 4, 4), num_res_blocks=2, latent_channels=16,
                            double_latent=True, in_channels=3, out_channels=3, dropout=0.0)


def test_vae_decoder_matches_the_taming_lineage_decoder():
    dec = janus.JanusVQVAEDecoder(_janus_cfg())
    dec.up[0].attn = torch.nn.ModuleList()          # AutoencoderKL: attention in the mid block only
    _seed_module(dec, 11)
    sd = {"decoder.conv_in.weight": dec.conv_in.weight.data, "decoder.conv_in.bias": dec.conv_in.bias.data,
          "decoder.conv_norm_out.weight": dec.norm_out.weight.data, "decoder.conv_norm_out.bias": dec.norm_out.bias.data,
          "decoder.conv_out.weight": dec.conv_out.weight.data, "decoder.conv_out.bias": dec.conv_out.bias.data}
    _mid_sd(dec.mid, "decoder.mid_block.", sd)
    for i, up in enumerate(dec.up):                 # Janus lists the levels lowest resolution first, like diffusers' up_blocks
        for j, blk in enumerate(up.block):
            _resnet_sd(blk, f"decoder.up_blocks.{i}.resnets.{j}.", sd)
        if hasattr(up, "upsample"):
            sd[f"decoder.up_blocks.{i}.upsamplers.0.conv.weight"] = up.upsample.conv.weight.data
            sd[f"decoder.up_blocks.{i}.upsamplers.0.conv.bias"] = up.upsample.conv.bias.data
    # the oracle's decoder walks exactly the parameter set of the FLUX VAE decoder
    from gpt_image_edit_amd import flux_spec
    want = {k for k in flux_spec.vae_param_shapes() if k.startswith("decoder.")}
    assert set(sd) == want, sorted(set(sd) ^ want)[:6]
    z = torch.randn(2, 16, 6, 5, generator=torch.Generator().manual_seed(3))
    with torch.no_grad():
        ref = dec(z.clone())
    got = vae.decode(sd, z)
    assert got.shape == ref.shape == (2, 3, 48, 40)
    torch.testing.assert_close(got, ref, rtol=1e-4, atol=1e-4)


def test_vae_encoder_matches_the_taming_lineage_encoder():
    enc = janus.JanusVQVAEEncoder(_janus_cfg())
    enc.down[-1].attn = torch.nn.ModuleList()
    _seed_module(enc, 12)
    sd = {"encoder.conv_in.weight": enc.conv_in.weight.data, "encoder.conv_in.bias": enc.conv_in.bias.data,
          "encoder.conv_norm_out.weight": enc.norm_out.weight.data, "encoder.conv_norm_out.bias": enc.norm_out.bias.data,
          "encoder.conv_out.weight": enc.conv_out.weight.data, "encoder.conv_out.bias": enc.conv_out.bias.data}
    _mid_sd(enc.mid, "encoder.mid_block.", sd)
    for i, down in enumerate(enc.down):
        for j, blk in enumerate(down.block):
            _resnet_sd(blk, f"encoder.down_blocks.{i}.resnets.{j}.", sd)
        if hasattr(down, "downsample"):
            sd[f"encoder.down_blocks.{i}.downsamplers.0.conv.weight"] = down.downsample.conv.weight.data
            sd[f"encoder.down_blocks.{i}.downsamplers.0.conv.bias"] = down.downsample.conv.bias.data
    from gpt_image_edit_amd import flux_spec
    want = {k for k in flux_spec.vae_param_shapes() if k.startswith("encoder.")}
    assert set(sd) == want, sorted(set(sd) ^ want)[:6]
    x = torch.randn(2, 3, 40, 56, generator=torch.Generator().manual_seed(4))
    with torch.no_grad():
        ref = enc(x.clone())
    got = vae.encode_moments(sd, x)
    assert got.shape == ref.shape == (2, 32, 5, 7)
    torch.testing.assert_close(got, ref, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(vae.encode_mode(sd, x), ref[:, :16], rtol=1e-4, atol=1e-4)


def test_adaln_zero_and_final_match_the_dit_modules():
    D, B, S = 96, 2, 7
    g = torch.Generator().manual_seed(5)
    x, temb = torch.randn(B, S, D, generator=g), torch.randn(B, D, generator=g)
    zero = _seed_module(omni.Qwen2_5_OmniAdaLayerNormZero(D), 21)
    sd = {"n.linear.weight": zero.linear.weight.data, "n.linear.bias": zero.linear.bias.data}
    with torch.no_grad():
        ref = zero(x, temb)
    # the oracle's double block, first lines: chunk order and modulation
    sh, sc, gt, sh2, sc2, gt2 = mmdit.linear(sd, "n.linear", F.silu(temb)).chunk(6, dim=1)
    got = (mmdit.layer_norm(x) * (1 + sc[:, None]) + sh[:, None], gt, sh2, sc2, gt2)
    for a, b in zip(got, ref):
        torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-5)
    final = _seed_module(omni.Qwen2_5_OmniAdaLayerNormZero_Final(D), 22)
    with torch.no_grad():
        ref = final(x, temb)
    sd = {"norm_out.linear.weight": final.linear.weight.data, "norm_out.linear.bias": final.linear.bias.data}
    scale, shift = mmdit.linear(sd, "norm_out.linear", F.silu(temb)).chunk(2, dim=1)      # AdaLayerNormContinuous order
    torch.testing.assert_close(mmdit.layer_norm(x) * (1 + scale[:, None]) + shift[:, None], ref, rtol=1e-5, atol=1e-5)


def test_feed_forward_rope_and_rmsnorm_conventions():
    g = torch.Generator().manual_seed(6)
    D = 64
    ff = _seed_module(omni.DiTMLP(D, mult=4), 31)
    x = torch.randn(2, 5, D, generator=g)
    sd = {"ff.net.0.proj.weight": ff.ff[0].weight.data, "ff.net.0.proj.bias": ff.ff[0].bias.data,
          "ff.net.2.weight": ff.ff[3].weight.data, "ff.net.2.bias": ff.ff[3].bias.data}
    with torch.no_grad():
        torch.testing.assert_close(mmdit.feed_forward(sd, "ff", x), ff(x), rtol=1e-5, atol=1e-5)
    # interleaved-pair rotation == de-interleave + half-split rotation (the convention transformers documents for
    # checkpoints trained with interleaved RoPE)
    ids = torch.zeros(9, 3)
    ids[:, 1], ids[:, 2] = torch.arange(9) // 3, torch.arange(9) % 3
    cos, sin = mmdit.rope_tables(ids, axes_dim=(4, 6, 6))
    q = torch.randn(1, 2, 9, 16, generator=g)
    got = mmdit.apply_rope(q, cos, sin)
    cos_h, sin_h = torch.cat([cos[:, 0::2], cos[:, 0::2]], -1), torch.cat([sin[:, 0::2], sin[:, 0::2]], -1)
    ref, _ = omni.apply_rotary_pos_emb(omni.deinterleave_head_dim(q), omni.deinterleave_head_dim(q), cos_h[None], sin_h[None])
    torch.testing.assert_close(omni.deinterleave_head_dim(got), ref, rtol=1e-6, atol=1e-6)
    # RMSNorm: fp32 statistics, cast to the storage dtype BEFORE the weight multiply
    from transformers.models.qwen2.modeling_qwen2 import Qwen2RMSNorm
    norm = Qwen2RMSNorm(128, eps=1e-6)
    with torch.no_grad():
        norm.weight.copy_(1 + 0.1 * torch.randn(128, generator=g))
    for dt in (torch.float32, torch.bfloat16):
        xh = torch.randn(2, 3, 11, 128, generator=g).to(dt)
        with torch.no_grad():
            ref = norm.to(dt)(xh)
        got = mmdit.rms_norm(xh, norm.weight.data.to(dt))
        assert got.dtype == ref.dtype
        torch.testing.assert_close(got.float(), ref.float(), rtol=0, atol=0)
