"""``denoise_projector`` of ``UnivaDenoiseTower`` on HIP.

Reference: ``univa/models/modeling_univa_denoise_tower.py:31-47`` builds
``nn.Sequential(Linear(3584, 12288), SiLU(), Linear(12288, 4096))`` and
``univa/models/qwen2p5vl/modeling_univa_qwen2p5vl.py:521-523`` applies it to the VLM's last hidden states to make
the first part of ``prompt_embeds``.  Here: two ``fk_gemm_bf16`` calls, the SiLU fused into the first one's epilogue.
The projector is one of the modules the reference's stage-2 run trains (``train_denoiser.py:71-119``), so it also has
the training pair ``forward_train`` / ``backward`` (same numbers as ``forward``; gradients of both Linears from the
gradient of ``prompt_embeds`` that ``backward.FluxBackward.backward`` returns; the frozen VLM needs no input gradient).
Parameter names are the Sequential's (``0.weight, 0.bias, 2.weight, 2.bias``), i.e. what
``checkpoint.read_projector`` returns.
"""
import torch
from torch import nn

from . import flux_spec, ops
from .param_tree import ParamTreeMixin, build_param_tree

BF16 = torch.bfloat16


class HipDenoiseProjector(ParamTreeMixin, nn.Module):
    def __init__(self, input_hidden_size=3584, output_hidden_size=4096, device="cuda", init="empty", seed=0):
        super().__init__()
        shapes = {k[len("denoise_projector."):]: v
                  for k, v in flux_spec.projector_param_shapes(input_hidden_size, output_hidden_size).items()}
        if init == "synthetic":
            state = flux_spec.synthetic_state(shapes, seed=seed, device=device, dtype=BF16)
        else:
            state = {k: torch.empty(s, device=device, dtype=BF16) for k, s in shapes.items()}
        self.__dict__["_pmap"] = build_param_tree(self, state, requires_grad=False)   # children "0" and "2", as nn.Sequential

    def load_state_dict(self, state_dict, strict=True, **kwargs):
        return super().load_state_dict({k: v.to(BF16) for k, v in state_dict.items()}, strict=strict, **kwargs)

    def forward(self, hidden_states):
        """[B, L, 3584] (or [L, 3584]) bf16 -> [B, L, 4096] bf16.  Under ``torch.enable_grad()`` with trainable
        parameters (the reference's ``with_tune_mlp2`` / ``only_tune_mlp2``, ``train_denoiser.py:527-548``) the call
        records one autograd node: forward_train / backward below."""
        if not hidden_states.is_cuda:
            raise RuntimeError("HipDenoiseProjector needs GPU tensors: there is no CPU fallback")
        if torch.is_grad_enabled() and any(prm.requires_grad for prm in self._pmap.values()):
            names = ("0.weight", "0.bias", "2.weight", "2.bias")
            return _ProjectorTrainFunction.apply(self, names, hidden_states, *[self.p(n) for n in names])
        return self._forward_infer(hidden_states)

    @torch.no_grad()
    def _forward_infer(self, hidden_states):
        x = hidden_states.to(BF16).contiguous()
        squeeze = x.dim() == 2
        if squeeze:
            x = x.unsqueeze(0)
        h = ops.gemm(x, self.p("0.weight"), self.p("0.bias"), epilogue=ops.FK_EPI_SILU)
        y = ops.gemm(h, self.p("2.weight"), self.p("2.bias"))
        return y[0] if squeeze else y

    # ---- training ---------------------------------------------------------------------------------------------------
    def _b(self, name, shape, dtype=BF16, zero=False):
        bufs = self.__dict__.setdefault("_train_bufs", {})
        t = bufs.get(name)
        if t is None or tuple(t.shape) != tuple(shape) or t.dtype != dtype:
            t = (torch.zeros if zero else torch.empty)(shape, device=self.p("0.weight").device, dtype=dtype)
            bufs[name] = t
        return t

    @torch.no_grad()
    def forward_train(self, hidden_states):
        """``forward`` with the pre-activation kept for :meth:`backward` (the SiLU runs as its own pass over the first
        Linear's bf16 output -- the same rounding points as the fused epilogue, bit-identical result)."""
        if not hidden_states.is_cuda:
            raise RuntimeError("HipDenoiseProjector needs GPU tensors: there is no CPU fallback")
        x = hidden_states.to(BF16).contiguous()
        if x.dim() == 2:
            x = x.unsqueeze(0)
        h1 = ops.gemm(x, self.p("0.weight"), self.p("0.bias"))
        a = ops.silu(h1)
        self.__dict__["_saved"] = (x, h1, a)
        return ops.gemm(a, self.p("2.weight"), self.p("2.bias"))

    @torch.no_grad()
    def backward(self, dy):
        """dy: gradient of ``forward_train``'s output [B, L, out] -> {Sequential name: gradient} (weights bf16 from the
        MFMA GEMM's fp32 accumulators, biases fp32 column sums)."""
        from .backward import wgrad
        saved = self.__dict__.pop("_saved", None)
        if saved is None:
            raise RuntimeError("HipDenoiseProjector.backward needs a preceding forward_train")
        x, h1, a = saved
        dy = dy.to(BF16)
        if dy.dim() == 2:
            dy = dy.unsqueeze(0)
        if dy.shape[:2] != x.shape[:2] or dy.shape[2] != self.p("2.weight").shape[0]:
            raise ValueError(f"gradient shape {tuple(dy.shape)} does not match the projector output")
        dy = dy.contiguous()
        grads = {"2.weight": wgrad(self._b, dy, a), "2.bias": ops.colsum(dy)}
        w2 = self.p("2.weight")                                           # [out, inner] -> dgrad operand [inner, out]
        w2T = self._b("w2T", (w2.shape[1], w2.shape[0]))
        ops.transpose(w2.unsqueeze(0), w2T.unsqueeze(0))
        da = ops.gemm(dy, w2T)
        dh1 = ops.silu_bwd(h1, da, out=da)
        grads["0.weight"] = wgrad(self._b, dh1, x)
        grads["0.bias"] = ops.colsum(dh1)
        return grads


class _ProjectorTrainFunction(torch.autograd.Function):
    """Autograd node of the projector: the gradient of ``prompt_embeds`` (from the MMDiT's node) becomes the four
    parameter gradients.  The VLM upstream is frozen in every configuration the reference ships, so no input gradient."""

    @staticmethod
    def forward(ctx, proj, names, hidden_states, *params):
        ctx.proj, ctx.names, ctx.squeeze = proj, names, hidden_states.dim() == 2
        y = proj.forward_train(hidden_states)
        return y[0] if ctx.squeeze else y

    @staticmethod
    def backward(ctx, dy):
        grads = ctx.proj.backward(dy.unsqueeze(0) if ctx.squeeze else dy)
        return (None, None, None, *[grads[n].to(BF16) for n in ctx.names])
