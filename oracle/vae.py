"""Oracle: FLUX ``AutoencoderKL`` encode/decode (diffusers 0.32.2).  TEST INFRASTRUCTURE.

PARITY UNPINNED against the real third-party source (diffusers is not installable here); restated from SURVEY.md
Appendix A.3.  Second opinion: the whole encoder and decoder are held to an INDEPENDENT implementation of the same
taming-lineage topology that IS installed (transformers' ``JanusVQVAEEncoder`` / ``JanusVQVAEDecoder`` configured to the
FLUX VAE's sizes) on shared seeded weights, ``tests/test_oracle_third_party.py``.
Reference call sites: ``univa/utils/flux_pipeline.py:604-611`` (encode + ``mode()``),
``:1127-1129`` (decode); ``train_denoiser.py:428-432,887,895,1505``.

Functional over a flat state dict with the diffusers key names (``encoder.*`` / ``decoder.*``).
"""
import torch
import torch.nn.functional as F

VAE_CONFIG = dict(
    in_channels=3, out_channels=3, latent_channels=16, block_out_channels=(128, 256, 512, 512),
    layers_per_block=2, norm_num_groups=32, scaling_factor=0.3611, shift_factor=0.1159,
)


def conv(sd, name, x, stride=1, padding=1):
    return F.conv2d(x, sd[name + ".weight"], sd[name + ".bias"], stride=stride, padding=padding)


def group_norm(sd, name, x, groups=32, eps=1e-6):
    return F.group_norm(x, groups, sd[name + ".weight"], sd[name + ".bias"], eps)


def resnet(sd, p, x):
    """ResnetBlock2D (no time embedding, output scale 1)."""
    y = conv(sd, p + "conv1", F.silu(group_norm(sd, p + "norm1", x)))
    y = conv(sd, p + "conv2", F.silu(group_norm(sd, p + "norm2", y)))
    if (p + "conv_shortcut.weight") in sd:
        x = conv(sd, p + "conv_shortcut", x, padding=0)
    return x + y


def mid_attention(sd, p, x):
    """Single-head (hd = C) spatial self-attention of UNetMidBlock2D with residual."""
    b, c, hh, ww = x.shape
    res = x
    t = x.view(b, c, hh * ww)
    t = F.group_norm(t, 32, sd[p + "group_norm.weight"], sd[p + "group_norm.bias"], 1e-6).transpose(1, 2)
    q = F.linear(t, sd[p + "to_q.weight"], sd[p + "to_q.bias"])[:, None]
    k = F.linear(t, sd[p + "to_k.weight"], sd[p + "to_k.bias"])[:, None]
    v = F.linear(t, sd[p + "to_v.weight"], sd[p + "to_v.bias"])[:, None]
    o = F.scaled_dot_product_attention(q, k, v)[:, 0]
    o = F.linear(o, sd[p + "to_out.0.weight"], sd[p + "to_out.0.bias"])
    return o.transpose(1, 2).reshape(b, c, hh, ww) + res


def mid_block(sd, p, x):
    x = resnet(sd, p + "resnets.0.", x)
    x = mid_attention(sd, p + "attentions.0.", x)
    return resnet(sd, p + "resnets.1.", x)


def decode(sd, z):
    """Decoder: z [B,16,h,w] -> image [B,3,8h,8w] (no latent rescale; the pipeline does that)."""
    x = conv(sd, "decoder.conv_in", z)
    x = mid_block(sd, "decoder.mid_block.", x)
    for i in range(4):
        for j in range(3):
            x = resnet(sd, f"decoder.up_blocks.{i}.resnets.{j}.", x)
        if i < 3:
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
            x = conv(sd, f"decoder.up_blocks.{i}.upsamplers.0.conv", x)
    x = F.silu(group_norm(sd, "decoder.conv_norm_out", x))
    return conv(sd, "decoder.conv_out", x)


def encode_moments(sd, x):
    """Encoder: image [B,3,H,W] -> moments [B,32,H/8,W/8] (mean | logvar)."""
    x = conv(sd, "encoder.conv_in", x)
    for i in range(4):
        for j in range(2):
            x = resnet(sd, f"encoder.down_blocks.{i}.resnets.{j}.", x)
        if i < 3:
            x = F.pad(x, (0, 1, 0, 1), mode="constant", value=0)
            x = conv(sd, f"encoder.down_blocks.{i}.downsamplers.0.conv", x, stride=2, padding=0)
    x = mid_block(sd, "encoder.mid_block.", x)
    x = F.silu(group_norm(sd, "encoder.conv_norm_out", x))
    return conv(sd, "encoder.conv_out", x)


def encode_mode(sd, x):
    """``vae.encode(x).latent_dist.mode()`` = mean half of the moments."""
    return encode_moments(sd, x).chunk(2, dim=1)[0]


def _scalar_op(x, op, c):
    """``x (op) python_scalar`` with the rounding the reference's CUDA torch build applies to a
    bf16/fp16 tensor: the scalar stays fp32 (opmath), only the result is rounded to x.dtype.
    (CPU torch instead rounds the scalar itself to bf16 first -- 0.1159 -> 0.11572 -- which is a
    CPU-only artefact the GPU reference never sees, so the oracle does not reproduce it.)"""
    y = x.float()
    y = y / c if op == "div" else (y * c if op == "mul" else y + c)
    return y.to(x.dtype)


def encode_for_pipeline(sd, image, cfg=VAE_CONFIG):
    """flux_pipeline.py:600-613: (mode(z) - shift) * scale."""
    z = encode_mode(sd, image)
    return _scalar_op(_scalar_op(z, "add", -cfg["shift_factor"]), "mul", cfg["scaling_factor"])


def unscale_latents(latents, cfg=VAE_CONFIG):
    """flux_pipeline.py:1128: latents / scaling_factor + shift_factor."""
    return _scalar_op(_scalar_op(latents, "div", cfg["scaling_factor"]), "add", cfg["shift_factor"])


def decode_for_pipeline(sd, latents, cfg=VAE_CONFIG):
    """flux_pipeline.py:1128-1129: decode(z / scale + shift)."""
    return decode(sd, unscale_latents(latents, cfg))


def postprocess_uint8(image):
    """VaeImageProcessor.postprocess(..., 'pil') up to the uint8 array: NHWC uint8."""
    x = (image / 2 + 0.5).clamp(0, 1)
    x = x.cpu().permute(0, 2, 3, 1).float().numpy()
    return (x * 255).round().astype("uint8")


def preprocess_uint8(u8_nhwc, out_h, out_w):
    """uint8 [N,H,W,3] pixels -> the bf16 NCHW tensor the reference hands to vae.encode:
    ``cli.prepare_condition_images`` (univa/serve/cli.py:99-116: ``/255``, ``(x-0.5)/0.5`` in fp32), then the pipeline's
    ``image_processor.resize`` (tensor => ``F.interpolate`` nearest) and ``preprocess`` (``2x-1`` iff nothing is negative;
    univa/utils/flux_pipeline.py:960-972), then ``image.to(dtype)`` in prepare_latents."""
    x = u8_nhwc.to(torch.float32) / 255.0
    x = x.permute(0, 3, 1, 2)
    x = (x - 0.5) / 0.5
    if tuple(x.shape[2:]) != (out_h, out_w):
        x = F.interpolate(x, size=(out_h, out_w))
    if x.min() >= 0:
        x = 2.0 * x - 1.0
    return x.to(torch.bfloat16)
