"""Oracle: FLUX MMDiT forward (``FluxTransformer2DModel`` of diffusers 0.32.2).  TEST INFRASTRUCTURE.

PARITY UNPINNED against the real third-party source (not vendored in the reference, not
installed, no network): this restates the published algorithm as summarised in SURVEY.md
Appendix A.1.  Anchors in the reference: the transformer is called at
``univa/utils/flux_pipeline.py:1067-1077``, ``univa/models/modeling_univa_denoise_tower.py:103-110``
and ``univa/models/qwen2p5vl/modeling_univa_qwen2p5vl.py:355``; parameter names are corroborated
by ``train_denoiser.py:74-119``.  Second opinions on the building blocks (AdaLayerNorm-Zero / -Continuous chunk order
and modulation, GELU(tanh) feed-forward, interleaved-pair RoPE, RMSNorm cast order) from independent implementations
installed with transformers: ``tests/test_oracle_third_party.py``; the wiring of the blocks stays unpinned.

All functions are functional over a flat state dict ``sd`` with the diffusers key names
 and run in whatever dtype the tensors carry: fp32 tensors give the
"exact" oracle, bf16 tensors reproduce the rounding points of the reference's bf16 execution
(torch rounds every op output to bf16 exactly like the reference's torch build does).
"""
import math

import torch
import torch.nn.functional as F

DEFAULT_CONFIG = dict(
    in_channels=64, num_layers=19, num_single_layers=38, attention_head_dim=128,
    num_attention_heads=24, joint_attention_dim=4096, pooled_projection_dim=768,
    guidance_embeds=True, axes_dims_rope=(16, 56, 56),
)


def linear(sd, name, x):
    return F.linear(x, sd[name + ".weight"], sd.get(name + ".bias"))


# ---- A.1.1 time / guidance / pooled-text embedding ----------------------------------------------
def sinusoid_256(v):
    """``get_timestep_embedding(v, 256, flip_sin_to_cos=True, downscale_freq_shift=0)`` -> fp32 [B,256]."""
    half = 128
    freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
    ang = v.float()[:, None] * freqs[None, :]
    return torch.cat([torch.cos(ang), torch.sin(ang)], dim=-1)  # cos first


def time_text_embed(sd, timestep, guidance, pooled, prefix="time_text_embed."):
    """CombinedTimestepGuidanceTextProjEmbeddings: temb = (T(t) + G(g)) + P(pooled)."""
    def mlp(name, x):
        return linear(sd, prefix + name + ".linear_2", F.silu(linear(sd, prefix + name + ".linear_1", x)))

    t_emb = mlp("timestep_embedder", sinusoid_256(timestep).to(pooled.dtype))
    g_emb = mlp("guidance_embedder", sinusoid_256(guidance).to(pooled.dtype))
    return (t_emb + g_emb) + mlp("text_embedder", pooled)


# ---- A.1.2 rotary tables ---------------------------------------------------------------------------
def rope_tables(ids, axes_dim=(16, 56, 56), theta=10000.0):
    """FluxPosEmbed: ids [S,3] -> (cos, sin) each [S, sum(axes_dim)] fp32, interleave-repeated."""
    pos = ids.float()
    cos_parts, sin_parts = [], []
    for i, d in enumerate(axes_dim):
        freqs = 1.0 / (theta ** (torch.arange(0, d, 2, dtype=torch.float64) / d))
        ang = torch.outer(pos[:, i], freqs)  # fp32 x fp64 -> fp64
        cos_parts.append(ang.cos().repeat_interleave(2, dim=1).float())
        sin_parts.append(ang.sin().repeat_interleave(2, dim=1).float())
    return torch.cat(cos_parts, dim=-1), torch.cat(sin_parts, dim=-1)


def apply_rope(x, cos, sin):
    """x [B,H,S,hd]; interleaved-pair rotation in fp32, result cast back to x.dtype."""
    xr = x.reshape(*x.shape[:-1], -1, 2)
    rot = torch.stack([-xr[..., 1], xr[..., 0]], dim=-1).flatten(3)
    return (x.float() * cos[None, None] + rot.float() * sin[None, None]).to(x.dtype)


def rms_norm(x, weight, eps=1e-6):
    """diffusers RMSNorm: fp32 variance, cast to the weight dtype BEFORE the weight multiply."""
    var = x.float().pow(2).mean(-1, keepdim=True)
    y = x * torch.rsqrt(var + eps)  # promotes to fp32
    if weight.dtype in (torch.float16, torch.bfloat16):
        y = y.to(weight.dtype)
    return y * weight


def layer_norm(x, eps=1e-6):
    return F.layer_norm(x, (x.shape[-1],), None, None, eps)


def heads(x, n_heads):
    b, s, d = x.shape
    return x.view(b, s, n_heads, d // n_heads).transpose(1, 2)


def sdpa(q, k, v):
    o = F.scaled_dot_product_attention(q, k, v, attn_mask=None, dropout_p=0.0, is_causal=False)
    b, h, s, hd = o.shape
    return o.transpose(1, 2).reshape(b, s, h * hd).to(q.dtype)


def feed_forward(sd, prefix, x):
    """FeedForward(gelu-approximate): Linear -> GELU(tanh) -> Linear."""
    return linear(sd, prefix + ".net.2", F.gelu(linear(sd, prefix + ".net.0.proj", x), approximate="tanh"))


# ---- A.1.3 double-stream block ---------------------------------------------------------------------
def double_block(sd, p, h, c, temb, rope, n_heads=24):
    """FluxTransformerBlock.  h: image stream [B,S_img,D]; c: text stream [B,S_txt,D]."""
    cos, sin = rope
    act = F.silu(temb)
    sh, sc, gt, sh2, sc2, gt2 = linear(sd, p + "norm1.linear", act).chunk(6, dim=1)
    nh = layer_norm(h) * (1 + sc[:, None]) + sh[:, None]
    csh, csc, cgt, csh2, csc2, cgt2 = linear(sd, p + "norm1_context.linear", act).chunk(6, dim=1)
    nc = layer_norm(c) * (1 + csc[:, None]) + csh[:, None]

    q_i = rms_norm(heads(linear(sd, p + "attn.to_q", nh), n_heads), sd[p + "attn.norm_q.weight"])
    k_i = rms_norm(heads(linear(sd, p + "attn.to_k", nh), n_heads), sd[p + "attn.norm_k.weight"])
    v_i = heads(linear(sd, p + "attn.to_v", nh), n_heads)
    q_t = rms_norm(heads(linear(sd, p + "attn.add_q_proj", nc), n_heads), sd[p + "attn.norm_added_q.weight"])
    k_t = rms_norm(heads(linear(sd, p + "attn.add_k_proj", nc), n_heads), sd[p + "attn.norm_added_k.weight"])
    v_t = heads(linear(sd, p + "attn.add_v_proj", nc), n_heads)

    q = apply_rope(torch.cat([q_t, q_i], dim=2), cos, sin)  # text tokens first
    k = apply_rope(torch.cat([k_t, k_i], dim=2), cos, sin)
    v = torch.cat([v_t, v_i], dim=2)
    o = sdpa(q, k, v)
    s_txt = c.shape[1]
    o_t, o_i = o[:, :s_txt], o[:, s_txt:]

    h = h + gt[:, None] * linear(sd, p + "attn.to_out.0", o_i)
    nh2 = layer_norm(h) * (1 + sc2[:, None]) + sh2[:, None]
    h = h + gt2[:, None] * feed_forward(sd, p + "ff", nh2)

    c = c + cgt[:, None] * linear(sd, p + "attn.to_add_out", o_t)
    nc2 = layer_norm(c) * (1 + csc2[:, None]) + csh2[:, None]
    c = c + cgt2[:, None] * feed_forward(sd, p + "ff_context", nc2)
    if c.dtype == torch.float16:
        c = c.clip(-65504, 65504)
    return c, h


# ---- A.1.4 single-stream block ---------------------------------------------------------------------
def single_block(sd, p, s, temb, rope, n_heads=24):
    """FluxSingleTransformerBlock on the concatenated [txt, img] sequence."""
    cos, sin = rope
    sh, sc, gt = linear(sd, p + "norm.linear", F.silu(temb)).chunk(3, dim=1)
    ns = layer_norm(s) * (1 + sc[:, None]) + sh[:, None]
    m = F.gelu(linear(sd, p + "proj_mlp", ns), approximate="tanh")
    q = apply_rope(rms_norm(heads(linear(sd, p + "attn.to_q", ns), n_heads), sd[p + "attn.norm_q.weight"]), cos, sin)
    k = apply_rope(rms_norm(heads(linear(sd, p + "attn.to_k", ns), n_heads), sd[p + "attn.norm_k.weight"]), cos, sin)
    v = heads(linear(sd, p + "attn.to_v", ns), n_heads)
    a = sdpa(q, k, v)
    out = gt[:, None] * linear(sd, p + "proj_out", torch.cat([a, m], dim=2))  # attention first
    return s + out


# ---- A.1 whole model -------------------------------------------------------------------------------
def flux_forward(sd, hidden_states, encoder_hidden_states, pooled_projections, timestep, img_ids,
                 txt_ids, guidance, config=None, return_intermediates=False):
    """FluxTransformer2DModel.forward -> [B, S_img, 64].

    ``timestep`` is t/1000 (in [0,1]) exactly as the pipeline passes it (flux_pipeline.py:1069).
    """
    cfg = dict(DEFAULT_CONFIG)
    cfg.update(config or {})
    n_heads = cfg["num_attention_heads"]
    h = linear(sd, "x_embedder", hidden_states)
    t = timestep.to(h.dtype) * 1000
    g = guidance.to(h.dtype) * 1000
    temb = time_text_embed(sd, t, g, pooled_projections)
    c = linear(sd, "context_embedder", encoder_hidden_states)
    rope = rope_tables(torch.cat([txt_ids, img_ids], dim=0), cfg["axes_dims_rope"])
    inter = {"temb": temb, "h0": h, "c0": c}
    for i in range(cfg["num_layers"]):
        c, h = double_block(sd, f"transformer_blocks.{i}.", h, c, temb, rope, n_heads)
        if return_intermediates:
            inter[f"double{i}.h"], inter[f"double{i}.c"] = h, c
    s = torch.cat([c, h], dim=1)
    for i in range(cfg["num_single_layers"]):
        s = single_block(sd, f"single_transformer_blocks.{i}.", s, temb, rope, n_heads)
        if return_intermediates:
            inter[f"single{i}.s"] = s
    h = s[:, c.shape[1]:]
    # A.1.5 AdaLayerNormContinuous: scale first, then shift
    e = linear(sd, "norm_out.linear", F.silu(temb).to(h.dtype))
    scale, shift = e.chunk(2, dim=1)
    h = layer_norm(h) * (1 + scale)[:, None, :] + shift[:, None, :]
    out = linear(sd, "proj_out", h)
    return (out, inter) if return_intermediates else out


# ---- A.4 in-tree glue ------------------------------------------------------------------------------
def denoise_projector(sd, x, prefix="denoise_projector."):
    """Linear(3584,12288) -> SiLU -> Linear(12288,4096).

    reference: univa/models/modeling_univa_denoise_tower.py:31-47 (applied at
    univa/models/qwen2p5vl/modeling_univa_qwen2p5vl.py:521-523).
    """
    return linear(sd, prefix + "2", F.silu(linear(sd, prefix + "0", x)))


def flops_forward(S, D=3072, n_double=19, n_single=38):
    """Algorithmic FLOPs of one sample-forward (BASELINE.md section 2 formula, embedders omitted)."""
    n_blocks = n_double + n_single
    return n_blocks * 24 * D * D * S + n_blocks * 4 * S * S * D + 2 * (n_double * 12 + n_single * 3 + 2) * D * D
