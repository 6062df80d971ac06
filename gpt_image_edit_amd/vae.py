"""HipAutoencoderKL -- the FLUX 16-channel VAE on the HIP kernels, behind the diffusers interface the
reference pipeline uses: ``vae.encode(x).latent_dist.mode()`` (reference
``univa/utils/flux_pipeline.py:604-609``), ``vae.decode(z, return_dict=False)[0]`` (``:1129``),
``vae.config.{block_out_channels, latent_channels, scaling_factor, shift_factor}`` (``:255-258,611,1128``)
and ``vae.dtype`` (``:601``).  Parameters carry the diffusers key names (flux_spec.vae_param_shapes).

Internally activations are NHWC bf16; every conv is the implicit-GEMM MFMA kernel (taps gathered in
the loader, nearest-2x upsample and the stride-2 (0,1,0,1) padding fused into the addressing, residual
add fused into the epilogue); GroupNorm is a two-pass stats + fused apply/SiLU; the single-head
mid-block attention is QK^T (fp32 scores) -> row softmax -> PV on the same GEMM kernel.
No torch math runs on the data path; there is no CPU fallback.

``encode(x, fp32=True)`` is the fp32-class encoder of the reference's optimisation step (``train_denoiser.py:458`` loads
the VAE in fp32, ``:887-918`` encodes target and condition with it): fp32 NHWC activations, every product on the bf16
MFMA as the K-concatenation [a_hi | a_lo | a_hi] . [w_hi | w_hi | w_lo] with fp32 accumulation (include/fk.h
"fp32-class encoder").  With the module's bf16 parameters the weight side has no low part (two terms);
``load_fp32_state_dict`` keeps the low parts of an fp32 checkpoint next to the bf16 parameters (three terms).
"""
import os
from types import SimpleNamespace

import torch
from torch import nn

from . import flux_spec, ops
from .param_tree import ParamTreeMixin, build_param_tree

HALO = os.environ.get("FK_VAE_HALO", "1") != "0"
# the mid-block attention (1 head x 512) as one fused launch (csrc/vae_attention.hip); 0 = the three launches of rounds 1-5 (A/B)
FUSED_MID_ATTENTION = os.environ.get("FK_VAE_FUSED_ATTN", "1") != "0"

BF16 = torch.bfloat16


def _pack_conv(w, cin_pad=None, cout_pad=None):
    """OIHW -> [Cout_pad, Kpad] with k = (kh*KW + kw)*Cin_pad + ci, zero padded (one-time weight prep)."""
    co, ci, kh, kw = w.shape
    cin_pad = cin_pad or ci
    cout_pad = cout_pad or co
    t = torch.zeros((cout_pad, kh, kw, cin_pad), device=w.device, dtype=w.dtype)
    t[:co, :, :, :ci] = w.permute(0, 2, 3, 1)
    k = kh * kw * cin_pad
    kpad = (k + 63) // 64 * 64
    out = torch.zeros((cout_pad, kpad), device=w.device, dtype=w.dtype)
    out[:, :k] = t.reshape(cout_pad, k)
    return out.contiguous()


def _pack_conv_parts(ws, cin_pad, cout_pad):
    """OIHW weight parts -> [Cout_pad, Kpad] with k = (kh*KW + kw)*(P*Cin_pad) + p*Cin_pad + ci."""
    co, ci, kh, kw = ws[0].shape
    P = len(ws)
    t = torch.zeros((cout_pad, kh, kw, P * cin_pad), device=ws[0].device, dtype=BF16)
    for p_, w in enumerate(ws):
        t[:co, :, :, p_ * cin_pad: p_ * cin_pad + ci] = w.permute(0, 2, 3, 1)
    k = kh * kw * P * cin_pad
    kpad = (k + 63) // 64 * 64
    out = torch.zeros((cout_pad, kpad), device=ws[0].device, dtype=BF16)
    out[:, :k] = t.reshape(cout_pad, k)
    return out.contiguous()


def _pad_vec(b, n):
    out = torch.zeros(n, device=b.device, dtype=b.dtype)
    out[: b.shape[0]] = b
    return out


class _LatentDist:
    """DiagonalGaussianDistribution surface used by the reference (mode / sample / mean / logvar)."""

    def __init__(self, moments):
        self.mean, self.logvar = moments.chunk(2, dim=1)

    def mode(self):
        return self.mean.contiguous()

    def sample(self, generator=None):
        std = torch.exp(0.5 * self.logvar.float().clamp(-30.0, 20.0))
        eps = torch.randn(self.mean.shape, generator=generator, device=self.mean.device, dtype=torch.float32)
        return (self.mean.float() + std * eps).to(self.mean.dtype)


class HipAutoencoderKL(ParamTreeMixin, nn.Module):
    def __init__(self, config=None, device="cuda", dtype=BF16, init="empty", seed=0):
        super().__init__()
        if dtype != BF16:
            raise ValueError("HipAutoencoderKL computes in bf16")
        cfg = dict(flux_spec.FLUX_VAE_CONFIG)
        cfg.update(config or {})
        self.config = SimpleNamespace(**cfg)
        shapes = flux_spec.vae_param_shapes(cfg)
        if init == "synthetic":
            state = flux_spec.synthetic_state(shapes, seed=seed, device=device, dtype=dtype)
        else:
            state = {k: torch.empty(s, device=device, dtype=dtype) for k, s in shapes.items()}
        self.__dict__["_pmap"] = build_param_tree(self, state, requires_grad=False)   # diffusers module / key names
        self._pk = None
        self._pk32 = None
        self._f32_state = None      # fp32 checkpoint values (load_fp32_state_dict), else None

    def load_state_dict(self, state_dict, strict=True, **kwargs):
        self._pk = self._pk32 = self._f32_state = None
        return super().load_state_dict(state_dict, strict=strict, **kwargs)

    def load_fp32_state_dict(self, state_dict, strict=True):
        """Load an fp32 checkpoint: the module parameters become bf16(w) (what decode / encode compute with) and the fp32
        values stay beside them for ``encode(fp32=True)`` (weights as bf16 hi + lo parts, norm / bias vectors in fp32)."""
        r = self.load_state_dict({k: v.to(BF16) for k, v in state_dict.items()}, strict=strict)
        dev = self.device
        self._f32_state = {k: v.detach().to(device=dev, dtype=torch.float32) for k, v in state_dict.items()
                           if k.startswith("encoder.") or k.startswith("quant_conv")}
        return r

    def _apply(self, fn, *a, **k):
        self._pk = self._pk32 = None
        r = super()._apply(fn, *a, **k)
        if self._f32_state is not None:      # follows the module's device; stays fp32 whatever dtype the move asked for
            dev = self.device
            self._f32_state = {n: v.to(dev) for n, v in self._f32_state.items()}
        return r

    @property
    def dtype(self):
        return BF16

    @property
    def device(self):
        return self.p("decoder.conv_in.weight").device

    # slicing / tiling passthroughs exist on the reference pipeline (flux_pipeline.py:616-646); the HIP
    # path decodes whole batches, so they are accepted and ignored.
    def enable_slicing(self): pass
    def disable_slicing(self): pass
    def enable_tiling(self): pass
    def disable_tiling(self): pass

    # ---- weight prep -----------------------------------------------------------------------------------
    def _packed(self):
        if self._pk is not None:
            return self._pk
        pk = {}
        for name, prm in self.state_dict().items():
            if name.endswith(".weight") and prm.dim() == 4:
                base = name[: -len(".weight")]
                ci, co = prm.shape[1], prm.shape[0]
                cin_pad = max(32, (ci + 7) // 8 * 8) if ci < 32 else ci
                cout_pad = (co + 7) // 8 * 8
                pk[base] = (_pack_conv(prm, cin_pad, cout_pad), _pad_vec(self.p(base + ".bias"), cout_pad), cout_pad)
        for side in ("encoder", "decoder"):
            a = f"{side}.mid_block.attentions.0."
            pk[a + "qkv"] = (torch.cat([self.p(a + f"{n}.weight") for n in ("to_q", "to_k", "to_v")]).contiguous(),
                             torch.cat([self.p(a + f"{n}.bias") for n in ("to_q", "to_k", "to_v")]).contiguous())
        self._pk = pk
        return pk

    # ---- building blocks (NHWC) ------------------------------------------------------------------------------
    def _conv(self, name, x, stride=1, pad=1, upsample=False, res=None):
        w, b, cout = self._packed()[name]
        ks = 3 if self.p(name + ".weight").shape[-1] == 3 else 1
        if pad == 1 and self._halo_ok(name, x, stride):
            return ops.conv3x3_halo(x, w, b, cout, upsample2x=upsample, res=res)
        return ops.conv2d_nhwc(x, w, b, cout, ksize=ks, stride=stride, pad=pad if ks == 3 else 0,
                               upsample2x=upsample, res=res)

    def _gn(self, name, x, silu):
        return ops.group_norm_nhwc(x, self.p(name + ".weight"), self.p(name + ".bias"), silu)

    def _halo_ok(self, name, x, stride=1):
        """3 x 3 / stride 1 convolutions with Cin % 64 == 0 and >= 64 output channels run on the LDS halo-tiled kernel
        (csrc/conv_halo.hip); FK_VAE_HALO=0 keeps everything on the implicit-GEMM kernel (A/B, same rounding points)."""
        w = self.p(name + ".weight")
        return HALO and stride == 1 and w.shape[-1] == 3 and x.shape[-1] % 64 == 0 and w.shape[0] >= 64

    def _gn_conv(self, norm, conv, x, res=None):
        """conv(silu(group_norm(x))) (+ res): GroupNorm-apply + SiLU as the convolution's operand prologue."""
        if not self._halo_ok(conv, x):
            return self._conv(conv, self._gn(norm, x, True), res=res)
        w, b, cout = self._packed()[conv]
        stats = ops.group_norm_stats(x)
        return ops.conv3x3_halo(x, w, b, cout, res=res, gn=(stats, self.p(norm + ".weight"), self.p(norm + ".bias"), True))

    def _resnet(self, p, x):
        t = self._gn_conv(p + "norm1", p + "conv1", x)
        xs = self._conv(p + "conv_shortcut", x) if self.has(p + "conv_shortcut.weight") else x
        return self._gn_conv(p + "norm2", p + "conv2", t, res=xs)

    def _mid_attention(self, p, x):
        B, H, W, C = x.shape
        S = H * W
        n = self._gn(p + "group_norm", x, False).view(B, S, C)
        wqkv, bqkv = self._packed()[p + "qkv"]
        qkv = ops.gemm(n, wqkv, bqkv)                               # [B, S, 3C]
        if C == 512 and FUSED_MID_ATTENTION:
            # one fused launch for the whole batch (csrc/vae_attention.hip): no S x S scores, no probabilities, no V^T copy
            o = ops.attention_hd512(qkv[:, :, :C], qkv[:, :, C:2 * C], qkv[:, :, 2 * C:], scale=C ** -0.5)
        else:
            o = self._mid_attention_3launch(qkv, B, S, C)
        xr = x.view(B, S, C)
        out = ops.gemm(o, self.p(p + "to_out.0.weight"), self.p(p + "to_out.0.bias"), epilogue=ops.FK_EPI_RES, res=xr)
        return out.view(B, H, W, C)

    def _mid_attention_3launch(self, qkv, B, S, C):
        """Rounds 1-5 (kept for head dimensions other than 512 -- the narrow test VAEs -- and as FK_VAE_FUSED_ATTN=0): per image
        fp32 scores [S, S] on the GEMM kernel, a row softmax, P V on the GEMM kernel against a transposed copy of V."""
        o = torch.empty((B, S, C), device=qkv.device, dtype=BF16)
        S_pad = (S + 63) // 64 * 64                                 # GEMM K granularity for P @ V
        scores = torch.empty((S, S_pad), device=qkv.device, dtype=torch.float32)
        probs = torch.zeros((S, S_pad), device=qkv.device, dtype=BF16)
        vt = torch.zeros((1, C, S_pad), device=qkv.device, dtype=BF16)
        for b in range(B):
            q, k, v = qkv[b, :, :C], qkv[b, :, C:2 * C], qkv[b, :, 2 * C:]
            ops.gemm(q, k, None, out=scores[:, :S], epilogue=ops.FK_EPI_SCALE, alpha=C ** -0.5, out_fp32=True)
            ops.softmax_rows(scores[:, :S], out=probs[:, :S])
            ops.transpose(v.unsqueeze(0), vt[:, :, :S])
            ops.gemm(probs, vt[0], None, out=o[b])
        return o

    def _mid(self, p, x):
        x = self._resnet(p + "resnets.0.", x)
        x = self._mid_attention(p + "attentions.0.", x)
        return self._resnet(p + "resnets.1.", x)

    # ---- fp32-class encoder --------------------------------------------------------------------------------------
    def _packed_f32(self):
        """Operand parts of the encoder's weights: {conv name: (w_parts [Cout_pad, Kpad] bf16, bias fp32, Cout_pad)},
        {norm name: (gamma fp32, beta fp32)}, the attention projections as [N, P*K] bf16 + fp32 bias; P = 3 with an fp32
        checkpoint (weight parts hi, hi, lo), 2 with the bf16 parameters."""
        if self._pk32 is not None:
            return self._pk32
        f32 = self._f32_state
        P = 3 if f32 is not None else 2

        def val(name):              # fp32 value of a parameter
            return f32[name] if f32 is not None and name in f32 else self.p(name).float()

        def wparts(name):           # bf16 weight parts in weight order (hi, hi[, lo])
            w = val(name)
            hi = w.to(BF16)
            return [hi, hi] + ([(w - hi.float()).to(BF16)] if P == 3 else [])

        pk = {"parts": P}
        for name, prm in self.state_dict().items():
            if not name.startswith("encoder."):
                continue
            base = name.rsplit(".", 1)[0]
            if name.endswith(".weight") and prm.dim() == 4:
                co, ci = prm.shape[0], prm.shape[1]
                cin_pad = 32 if ci < 32 else ci
                cout_pad = (co + 7) // 8 * 8
                pk[base] = (_pack_conv_parts(wparts(name), cin_pad, cout_pad), _pad_vec(val(base + ".bias"), cout_pad), cout_pad)
            elif name.endswith(".weight") and prm.dim() == 1:
                pk[base] = (val(name).contiguous(), val(base + ".bias").contiguous())
        a = "encoder.mid_block.attentions.0."
        pk[a + "qkv"] = (torch.cat([torch.cat(wparts(a + f"{n}.weight"), dim=1) for n in ("to_q", "to_k", "to_v")]).contiguous(),
                         torch.cat([val(a + f"{n}.bias") for n in ("to_q", "to_k", "to_v")]).contiguous())
        pk[a + "to_out.0"] = (torch.cat(wparts(a + "to_out.0.weight"), dim=1).contiguous(), val(a + "to_out.0.bias").contiguous())
        self._pk32 = pk
        return pk

    def _split32(self, x):
        P = self._packed_f32()["parts"]
        C = x.shape[-1]
        out = torch.empty((*x.shape[:-1], P * C), device=x.device, dtype=BF16)
        ops.split_f32_rows(x.view(-1, C), out.view(-1, P * C), parts=P)
        return out

    def _conv32(self, name, x_parts, stride=1, pad=1, res=None):
        w, b, cout = self._packed_f32()[name]
        ks = self.p(name + ".weight").shape[-1]
        halo = HALO and ks == 3 and stride == 1 and pad == 1 and x_parts.shape[-1] % 64 == 0 and cout >= 64
        return ops.conv2d_nhwc_f32out(x_parts, w, b, cout, ksize=ks, stride=stride, pad=pad if ks == 3 else 0, res=res, halo=halo)

    def _gn32(self, name, x, silu):
        pk = self._packed_f32()
        g, b = pk[name]
        return ops.group_norm_f32_parts(x, g, b, silu, pk["parts"])

    def _resnet32(self, p, x):
        t = self._conv32(p + "conv1", self._gn32(p + "norm1", x, True))
        xs = self._conv32(p + "conv_shortcut", self._split32(x)) if self.has(p + "conv_shortcut.weight") else x
        return self._conv32(p + "conv2", self._gn32(p + "norm2", t, True), res=xs)

    def _mid_attention32(self, p, x):
        pk = self._packed_f32()
        P = pk["parts"]
        B, H, W, C = x.shape
        S = H * W
        dev = x.device
        n = self._gn32(p + "group_norm", x, False).view(B * S, P * C)
        wqkv, bqkv = pk[p + "qkv"]
        qkv = ops.gemm(n, wqkv, bqkv, out_fp32=True).view(B, S, 3 * C)              # fp32
        if S % 4 != 0:
            raise ValueError(f"fp32-class mid-block attention: {H} x {W} = {S} positions, the part kernels need a multiple of 4")
        S_pad = (S + 63) // 64 * 64
        # scores (fp32 S x S) and the probability parts are 1 GiB + 1.5 GiB at 1024^2: kept per shape, and the padding columns
        # of probs / vt are zeroed ONCE (the kernels write columns < S only) instead of a 1.5 GiB memset per call
        ws = self.__dict__.setdefault("_mid32_ws", {})
        key = (S, C, str(dev))
        if key not in ws:
            ws.clear()                                                                  # one shape at a time: bounded memory
            ws[key] = (torch.empty((S, 3 * C), device=dev, dtype=BF16), torch.empty((S, 3 * C), device=dev, dtype=BF16),
                       torch.empty((S, 2 * C), device=dev, dtype=BF16), torch.empty((S, S_pad), device=dev, dtype=torch.float32),
                       torch.zeros((S, 3 * S_pad), device=dev, dtype=BF16), torch.zeros((1, C, 3 * S_pad), device=dev, dtype=BF16))
        qp, kp, vp, scores, probs, vt = ws[key]
        o = torch.empty((B, S, C), device=dev, dtype=torch.float32)
        for b in range(B):
            ops.split_f32_rows(qkv[b, :, :C], qp, parts=3)                           # (hi, lo, hi)
            ops.split_f32_rows(qkv[b, :, C:2 * C], kp, parts=3, weight_order=True)   # (hi, hi, lo)
            ops.split_f32_rows(qkv[b, :, 2 * C:], vp, parts=2)                       # (hi, lo)
            ops.gemm(qp, kp, None, out=scores[:, :S], epilogue=ops.FK_EPI_SCALE, alpha=C ** -0.5, out_fp32=True)
            ops.softmax_rows_parts(scores[:, :S], probs)
            v_hi, v_lo = vp[:, :C].unsqueeze(0), vp[:, C:].unsqueeze(0)
            ops.transpose(v_hi, vt[:, :, :S])
            ops.transpose(v_hi, vt[:, :, S_pad:S_pad + S])
            ops.transpose(v_lo, vt[:, :, 2 * S_pad:2 * S_pad + S])
            ops.gemm(probs, vt[0], None, out=o[b], out_fp32=True)
        op = torch.empty((B * S, P * C), device=dev, dtype=BF16)
        ops.split_f32_rows(o.view(B * S, C), op, parts=P)
        wo, bo = pk[p + "to_out.0"]
        out = ops.gemm(op, wo, bo, res=x.view(B * S, C), out_fp32=True)
        return out.view(B, H, W, C)

    def _encode32(self, x, post_add, post_mul):
        P = self._packed_f32()["parts"]
        t = ops.nchw_f32_to_nhwc_parts(x.float().contiguous(), 32, P)
        t = self._conv32("encoder.conv_in", t)
        n_down = len(self.config.block_out_channels)
        for i in range(n_down):
            for j in range(self.config.layers_per_block):
                t = self._resnet32(f"encoder.down_blocks.{i}.resnets.{j}.", t)
            if i < n_down - 1:
                t = self._conv32(f"encoder.down_blocks.{i}.downsamplers.0.conv", self._split32(t), stride=2, pad=0)
        t = self._resnet32("encoder.mid_block.resnets.0.", t)
        t = self._mid_attention32("encoder.mid_block.attentions.0.", t)
        t = self._resnet32("encoder.mid_block.resnets.1.", t)
        t = self._conv32("encoder.conv_out", self._gn32("encoder.conv_norm_out", t, True))
        return ops.nhwc_f32_to_nchw(t, 2 * self.config.latent_channels, post_add, post_mul)

    # ---- public interface -----------------------------------------------------------------------------------
    @torch.no_grad()
    def decode(self, z, return_dict=True, generator=None, pre_div=1.0, pre_add=0.0):
        """z [B,16,h,w] -> image [B,3,8h,8w] bf16.  ``pre_div/pre_add`` fuse the pipeline's
        ``z / scaling_factor + shift_factor`` (flux_pipeline.py:1128) into the layout change."""
        if not z.is_cuda:
            raise RuntimeError("HipAutoencoderKL needs GPU tensors: there is no CPU fallback")
        x = ops.nchw_to_nhwc(z.contiguous(), 32, pre_div, pre_add)
        x = self._conv("decoder.conv_in", x)
        x = self._mid("decoder.mid_block.", x)
        n_up = len(self.config.block_out_channels)
        for i in range(n_up):
            for j in range(self.config.layers_per_block + 1):
                x = self._resnet(f"decoder.up_blocks.{i}.resnets.{j}.", x)
            if i < n_up - 1:
                x = self._conv(f"decoder.up_blocks.{i}.upsamplers.0.conv", x, upsample=True)
        x = self._gn("decoder.conv_norm_out", x, True)
        x = self._conv("decoder.conv_out", x)
        img = ops.nhwc_to_nchw(x, self.config.out_channels)
        if not return_dict:
            return (img,)
        return SimpleNamespace(sample=img)

    @torch.no_grad()
    def encode(self, x, return_dict=True, post_add=0.0, post_mul=1.0, nhwc=False, fp32=False):
        """image [B,3,H,W] (fp32 or bf16, in [-1,1]) -> latent distribution with [B,16,H/8,W/8] moments.
        ``nhwc=True``: x is already the internal layout, NHWC bf16 [B,H,W,32] (``image_processor.pixels_to_latent_input``).
        ``fp32=True``: the fp32-class encoder (module docstring); moments are fp32."""
        if not x.is_cuda:
            raise RuntimeError("HipAutoencoderKL needs GPU tensors: there is no CPU fallback")
        if fp32:
            if nhwc:
                raise ValueError("the fp32-class encoder takes NCHW input")
            dist = _LatentDist(self._encode32(x, post_add, post_mul))
            return (dist,) if not return_dict else SimpleNamespace(latent_dist=dist)
        if nhwc:
            if x.dtype != BF16 or x.dim() != 4 or x.shape[3] != 32 or not x.is_contiguous():
                raise ValueError("nhwc input must be contiguous bf16 [B, H, W, 32]")
            t = x
        else:
            t = ops.nchw_to_nhwc(x.contiguous(), 32)
        t = self._conv("encoder.conv_in", t)
        n_down = len(self.config.block_out_channels)
        for i in range(n_down):
            for j in range(self.config.layers_per_block):
                t = self._resnet(f"encoder.down_blocks.{i}.resnets.{j}.", t)
            if i < n_down - 1:
                t = self._conv(f"encoder.down_blocks.{i}.downsamplers.0.conv", t, stride=2, pad=0)
        t = self._mid("encoder.mid_block.", t)
        t = self._gn("encoder.conv_norm_out", t, True)
        t = self._conv("encoder.conv_out", t)
        moments = ops.nhwc_to_nchw(t, 2 * self.config.latent_channels, post_add, post_mul)
        dist = _LatentDist(moments)
        if not return_dict:
            return (dist,)
        return SimpleNamespace(latent_dist=dist)
