"""Parameter layout of the FLUX-Kontext transformer and the FLUX AutoencoderKL.

Key names / shapes follow the diffusers checkpoints the reference loads with
``FluxKontextPipeline.from_pretrained(path, transformer=model.denoise_tower.denoiser, ...)``
(reference ``univa/serve/cli.py:64-68``); names are corroborated in-tree by
``train_denoiser.py:74-119``.  Real safetensors checkpoints therefore drop
in unchanged; offline we fill the same layout with seeded synthetic values (``synthetic_state``).
"""
import zlib
from collections import OrderedDict

import torch

FLUX_KONTEXT_CONFIG = dict(
    patch_size=1, in_channels=64, out_channels=64, num_layers=19, num_single_layers=38,
    attention_head_dim=128, num_attention_heads=24, joint_attention_dim=4096,
    pooled_projection_dim=768, guidance_embeds=True, axes_dims_rope=(16, 56, 56),
)

FLUX_VAE_CONFIG = dict(
    in_channels=3, out_channels=3, latent_channels=16, block_out_channels=(128, 256, 512, 512),
    layers_per_block=2, norm_num_groups=32, scaling_factor=0.3611, shift_factor=0.1159,
)

SCHEDULER_CONFIG = dict(
    num_train_timesteps=1000, shift=3.0, use_dynamic_shifting=True, base_shift=0.5, max_shift=1.15,
    base_image_seq_len=256, max_image_seq_len=4096,
)


def _lin(shapes, name, n_out, n_in):
    shapes[name + ".weight"] = (n_out, n_in)
    shapes[name + ".bias"] = (n_out,)


def flux_param_shapes(cfg=None):
    c = dict(FLUX_KONTEXT_CONFIG)
    c.update(cfg or {})
    hd, nh = c["attention_head_dim"], c["num_attention_heads"]
    D = hd * nh
    FF = 4 * D
    s = OrderedDict()
    _lin(s, "x_embedder", D, c["in_channels"])
    _lin(s, "context_embedder", D, c["joint_attention_dim"])
    for emb in ("timestep_embedder", "guidance_embedder"):
        _lin(s, f"time_text_embed.{emb}.linear_1", D, 256)
        _lin(s, f"time_text_embed.{emb}.linear_2", D, D)
    _lin(s, "time_text_embed.text_embedder.linear_1", D, c["pooled_projection_dim"])
    _lin(s, "time_text_embed.text_embedder.linear_2", D, D)
    for i in range(c["num_layers"]):
        p = f"transformer_blocks.{i}."
        _lin(s, p + "norm1.linear", 6 * D, D)
        _lin(s, p + "norm1_context.linear", 6 * D, D)
        for n in ("to_q", "to_k", "to_v", "add_q_proj", "add_k_proj", "add_v_proj"):
            _lin(s, p + "attn." + n, D, D)
        for n in ("norm_q", "norm_k", "norm_added_q", "norm_added_k"):
            s[p + f"attn.{n}.weight"] = (hd,)
        _lin(s, p + "attn.to_out.0", D, D)
        _lin(s, p + "attn.to_add_out", D, D)
        for ff in ("ff", "ff_context"):
            _lin(s, p + ff + ".net.0.proj", FF, D)
            _lin(s, p + ff + ".net.2", D, FF)
    for i in range(c["num_single_layers"]):
        p = f"single_transformer_blocks.{i}."
        _lin(s, p + "norm.linear", 3 * D, D)
        _lin(s, p + "proj_mlp", FF, D)
        _lin(s, p + "proj_out", D, D + FF)
        for n in ("to_q", "to_k", "to_v"):
            _lin(s, p + "attn." + n, D, D)
        for n in ("norm_q", "norm_k"):
            s[p + f"attn.{n}.weight"] = (hd,)
    _lin(s, "norm_out.linear", 2 * D, D)
    _lin(s, "proj_out", c["out_channels"] or c["in_channels"], D)
    return s


def _conv(shapes, name, c_out, c_in, k):
    shapes[name + ".weight"] = (c_out, c_in, k, k)
    shapes[name + ".bias"] = (c_out,)


def _norm(shapes, name, c):
    shapes[name + ".weight"] = (c,)
    shapes[name + ".bias"] = (c,)


def _resnet(shapes, p, c_in, c_out):
    _norm(shapes, p + "norm1", c_in)
    _conv(shapes, p + "conv1", c_out, c_in, 3)
    _norm(shapes, p + "norm2", c_out)
    _conv(shapes, p + "conv2", c_out, c_out, 3)
    if c_in != c_out:
        _conv(shapes, p + "conv_shortcut", c_out, c_in, 1)


def _mid(shapes, p, c):
    _resnet(shapes, p + "resnets.0.", c, c)
    a = p + "attentions.0."
    _norm(shapes, a + "group_norm", c)
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        _lin(shapes, a + n, c, c)
    _resnet(shapes, p + "resnets.1.", c, c)


def vae_param_shapes(cfg=None):
    c = dict(FLUX_VAE_CONFIG)
    c.update(cfg or {})
    boc = tuple(c["block_out_channels"])
    lat = c["latent_channels"]
    s = OrderedDict()
    # encoder
    _conv(s, "encoder.conv_in", boc[0], c["in_channels"], 3)
    ch = boc[0]
    for i, co in enumerate(boc):
        for j in range(c["layers_per_block"]):
            _resnet(s, f"encoder.down_blocks.{i}.resnets.{j}.", ch, co)
            ch = co
        if i < len(boc) - 1:
            _conv(s, f"encoder.down_blocks.{i}.downsamplers.0.conv", ch, ch, 3)
    _mid(s, "encoder.mid_block.", ch)
    _norm(s, "encoder.conv_norm_out", ch)
    _conv(s, "encoder.conv_out", 2 * lat, ch, 3)
    # decoder
    rev = boc[::-1]
    _conv(s, "decoder.conv_in", rev[0], lat, 3)
    _mid(s, "decoder.mid_block.", rev[0])
    ch = rev[0]
    for i, co in enumerate(rev):
        for j in range(c["layers_per_block"] + 1):
            _resnet(s, f"decoder.up_blocks.{i}.resnets.{j}.", ch, co)
            ch = co
        if i < len(rev) - 1:
            _conv(s, f"decoder.up_blocks.{i}.upsamplers.0.conv", ch, ch, 3)
    _norm(s, "decoder.conv_norm_out", ch)
    _conv(s, "decoder.conv_out", c["out_channels"], ch, 3)
    return s


def projector_param_shapes(input_hidden=3584, output_hidden=4096):
    """denoise_projector of UnivaDenoiseTower (reference modeling_univa_denoise_tower.py:34-44)."""
    s = OrderedDict()
    _lin(s, "denoise_projector.0", output_hidden * 3, input_hidden)
    _lin(s, "denoise_projector.2", output_hidden, output_hidden * 3)
    return s


def synthetic_state(shapes, seed=0, device="cpu", dtype=torch.float32, gen_device=None):
    """Seeded synthetic parameters: weights N(0, 0.02^2), biases N(0, 0.01^2),
    norm gains 1 + N(0, 0.1^2).  Every key is seeded independently (crc32 of its name) so any subset
    can be regenerated bit-identically.  ``gen_device='cuda'`` draws directly on the GPU (different
    stream of numbers than the CPU generator, used only for bench-scale models)."""
    gdev = gen_device or "cpu"
    out = OrderedDict()
    for name, shape in shapes.items():
        g = torch.Generator(device=gdev)
        g.manual_seed((zlib.crc32(name.encode()) + 7919 * seed) & 0x7FFFFFFF)
        x = torch.randn(shape, generator=g, device=gdev, dtype=torch.float32)
        is_gain = len(shape) == 1 and name.endswith(".weight")
        if is_gain:
            x = 1.0 + 0.1 * x
        elif name.endswith(".bias"):
            x = 0.01 * x
        else:
            x = 0.02 * x
        out[name] = x.to(device=device, dtype=dtype)
    return out
