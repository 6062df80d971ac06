"""MMDiT training forward + backward on the HIP kernels.

Counterpart of what ``accelerator.backward(loss)`` (reference ``train_denoiser.py:1172``) does to the FLUX-Kontext
denoiser.  Two activation policies: (a) the reference's ``enable_gradient_checkpointing()`` (``:486``, forced by 80 GB
GPUs): the forward keeps ONE tensor per block -- the residual stream entering it -- and the backward walks the blocks in
reverse, re-running each block's forward with its pre-gate / pre-activation tensors stored (un-fused epilogues, same
rounding points, hence the same values as the inference forward) and then its adjoint; (b) with 288 GB of HBM the
forward simply keeps every block's intermediates (37 GB at 1024^2, batch 1) and nothing is recomputed -- the default
whenever it fits.  The attention output and its row statistics are kept as well, so the recomputation
skips the attention kernel.  Gradients are produced for the parameters the reference un-freezes
(``train_denoiser.py:70-119``, ``training.trainable_names``): in every double block the image-stream ``attn.to_q / to_k
/ to_v / to_out.0``, ``attn.norm_q / norm_k`` and ``norm1.linear``; in every single block ``attn.to_q / to_k / to_v``,
``attn.norm_q / norm_k`` and ``norm.linear`` -- plus the gradient w.r.t. ``encoder_hidden_states`` (for the
``denoise_projector``).  Frozen weights (MLPs, text-stream projections, embedders) only carry the data gradient.

Everything is a libfk call:
  data gradients   dX = dY W          fk_gemm_bf16 on W^T (transposed once: frozen weights at first use, trainable ones
                                      after every optimiser step -- ``refresh()``)
  weight gradients dW = dY^T X        fk_gemm_bf16 on the two token-major operands transposed by fk_transpose_bf16
                                      (tokens zero-padded to a multiple of 64 = the GEMM's K granule)
  adjoints of the fused kernels       fk_attention_bwd_bf16, fk_ln_modulate_bwd_bf16, fk_gate_res_bwd_bf16,
                                      fk_gelu_bwd_bf16, fk_qkv_post_bwd_bf16, fk_colsum_bf16, fk_rowdot_bf16
Token reductions are fixed-order two-stage sums and the attention backward uses no atomics: gradients are bit-identical
from run to run.  Weight gradients are bf16 (what autograd produces for bf16 parameters); bias, norm-weight and
modulation reductions stay fp32 (``fk_adamw_step`` takes either).
"""
import ctypes
import os
from types import SimpleNamespace

import torch

from .param_tree import raw_write_epoch
from . import ops

BF16 = torch.bfloat16
# K-major GEMM operands (fk_gemm_args.layout 1 / 2): gradients read weights and activations as they lie in memory instead of
# physically transposed copies (same results bit for bit).  FK_BWD_K_MAJOR: 0 = copies everywhere, 1 = weight gradients read
# dY and X token-major (layout 2), 2 (default since round 5) = data gradients read the stored weight as well (layout 1: no W^T
# copies, 17 GB less).  Until round 5 the K-major kernels carried a compiler-inserted wait for ALL LDS-DMA requests in front
# of every group's first transpose read and ran 12-40 % below the row-major form; with the requests hidden from hipcc
# (fk_common.h, buffer_lds_opaque) they match it, and level 2 is the fastest step: 492 (level 1, before) -> 481 (level 1) ->
# 470 ms (level 2) on one box (profiles/r05_train_step_kmajor.txt, profiles/r05_gemm_kmajor_opaque_dma_ab.txt).
K_MAJOR = int(os.environ.get("FK_BWD_K_MAJOR", "2"))
# Block-level C entry points of the backward (fk_single_block_bwd / fk_double_block_bwd, csrc/blocks_bwd.hip): one ctypes call
# per block instead of one per launch (~80 per double block), where the stage-2 shape allows -- stored activations, batch 1,
# token counts that are multiples of 64, FK_BWD_K_MAJOR = 2; bit-identical to the per-launch route (0 selects it everywhere).
BLOCK_API = int(os.environ.get("FK_BWD_BLOCK_API", "1"))


def _pad64(n):
    return (n + 63) // 64 * 64


def wgrad(buf, dy, x, out=None):
    """Weight gradient dW[N, K] = dY^T X over the rows of two [B, R, *] views: one fk_gemm_bf16 on the views themselves
    (layout 2) when the shapes allow (whole 256 x 256 tiles, a multiple of 64 tokens, uniformly strided rows); else both
    operands transposed into K-major buffers whose row count is padded to the GEMM's 64 (pad columns zeroed on every call).
    ``buf(name, shape, zero=...)`` hands out the caller's cached workspace tensors."""
    B, R, N = dy.shape
    K = x.shape[-1]
    uniform = B == 1 or (dy.stride(0) == R * dy.stride(1) and x.stride(0) == R * x.stride(1))
    if K_MAJOR >= 1 and N % 256 == 0 and K % 256 == 0 and (B * R) % 64 == 0 and uniform:
        return ops.gemm(dy, x, out=out, layout=2)      # both operands token-major, as the forward left them
    Mp = _pad64(B * R)
    dyT = buf(f"dyT{N}x{Mp}", (N, Mp), zero=True)
    xT = buf(f"xT{K}x{Mp}", (K, Mp), zero=True)
    if B * R < Mp:
        # the buffers are shared by every call whose token count pads to the same multiple of 64 (VLM length 310, then
        # 300: both 320) and fk_transpose_bf16 writes only the B*R real columns: clear the <= 63 pad columns each time,
        # or the previous call's tail would be multiplied into this gradient
        dyT[:, B * R:].zero_()
        xT[:, B * R:].zero_()
    ops.transpose(dy, torch.as_strided(dyT, (B, N, R), (R, Mp, 1)))
    ops.transpose(x, torch.as_strided(xT, (B, K, R), (R, Mp, 1)))
    return ops.gemm(dyT, xT, out=out)


class FluxBackward:
    """Training forward / backward around a ``HipFluxTransformer2DModel`` (which keeps owning the parameters)."""

    def __init__(self, model, trainable=None, store_activations="auto", activation_budget_bytes=96 << 30):
        """store_activations: True = keep every block's intermediates from the forward (no recomputation: 37 GB at
        1024^2, bs 1 -- sized for 288 GB of HBM, where the reference's 80 GB GPUs force checkpointing); False = one
        checkpoint per block + recomputation in the backward; "auto" = store when it fits ``activation_budget_bytes``."""
        from . import training
        self.store_activations, self.activation_budget = store_activations, activation_budget_bytes
        self.m = model
        names = list(model._pmap.keys())
        self.trainable = set(trainable if trainable is not None else training.trainable_names(names))
        # every parameter of the 57 blocks has a weight gradient here; the embedders, norm_out and the output
        # projection do not (the reference never un-freezes them: train_denoiser.py:74-76, :93-95 are commented out)
        unsupported = sorted(k for k in self.trainable if not self.producible(k))
        if unsupported:
            raise NotImplementedError(
                "FluxBackward has no weight gradient for " + ", ".join(unsupported[:6]) + (" ..." if len(unsupported) > 6 else "")
                + ": trainable parameters must belong to transformer_blocks.* / single_transformer_blocks.*")
        if not model._train_packs:     # from now on the model fuses nothing an optimiser rewrites (transformer.pack_weights)
            model._train_packs, model._packed = True, None
        self._wT = {}          # name -> (W^T (bf16 [in, out]) for the data gradients, stamp of its source)
        self._buf = {}
        self._saved = None

    # ---- transposed weights ------------------------------------------------------------------------------------------
    @staticmethod
    def producible(name):
        return name.startswith("transformer_blocks.") or name.startswith("single_transformer_blocks.")

    def _transposed(self, key, w, stamp=None):
        """W^T for the data gradient, re-made when its source was rewritten (an optimiser step bumps the parameter's
        version; a re-pack of the fused QKV weights gets a new serial)."""
        if stamp is None:
            # + the count of raw-pointer parameter writes (fk_adamw_step bumps no torch version): a cached transpose of a
            # trainable weight is re-made after an optimiser step even when the caller skipped refresh()
            stamp = (w.data_ptr(), w._version, raw_write_epoch())
        hit = self._wT.get(key)
        if hit is None or hit[1] != stamp:
            t = hit[0] if hit is not None else torch.empty((w.shape[1], w.shape[0]), device=w.device, dtype=BF16)
            ops.transpose(w.unsqueeze(0), t.unsqueeze(0))
            hit = (t, stamp)
            self._wT[key] = hit
        return hit[0]

    def wT(self, name):
        return self._transposed(name, self.m.p(name))

    def _packedT(self, key, w, pk):
        return self._transposed("packed:" + key, w, stamp=("pack", pk.serial))

    # ---- operands of the data gradient dX = dY W ------------------------------------------------------------------------
    @staticmethod
    def _k_major_ok(w):
        """fk_gemm_args.layout 1 takes the weight [out, in] as stored: in % 256 == 0, out % 64 == 0 (every MMDiT linear)."""
        return K_MAJOR >= 2 and w.shape[1] % 256 == 0 and w.shape[0] % 64 == 0 and w.stride(1) == 1

    def dW(self, name, cols=None):
        """``w=, layout=`` of the GEMM that carries a gradient back through the linear ``name`` (``cols``: only these input
        columns of it): the stored weight itself when the K-major path takes it, else its cached transpose."""
        w = self.m.p(name).data
        w = w if cols is None else w[:, cols]
        if self._k_major_ok(w):
            return dict(w=w, layout=1)
        wt = self.wT(name)
        return dict(w=wt if cols is None else wt[cols], layout=0)

    def dWp(self, key, w, pk):
        """The same for a fused operand of the model's packs (QKV)."""
        return dict(w=w, layout=1) if self._k_major_ok(w) else dict(w=self._packedT(key, w, pk), layout=0)

    def refresh(self):
        """After an optimiser step through libfk (``fk_adamw_step`` rewrites the bf16 parameters through raw pointers,
        which torch's version counters do not see): invalidate the transposes of the trainable weights and the model's
        fused copies.  Updates made by torch itself (a stock optimiser, an all-gather) are caught by the stamps."""
        for k, (t, _) in list(self._wT.items()):
            if k in self.trainable or k.startswith("packed:"):
                self._wT[k] = (t, None)
        self.m.repack(self.trainable)

    # ---- buffers ---------------------------------------------------------------------------------------------------
    def _b(self, name, shape, dtype=BF16, zero=False):
        t = self._buf.get(name)
        if t is None or tuple(t.shape) != tuple(shape) or t.dtype != dtype:
            t = (torch.zeros if zero else torch.empty)(shape, device=self.m.device, dtype=dtype)
            self._buf[name] = t
        return t

    def _block_bufs(self, sv, idx, double):
        """Where block ``idx`` keeps what its backward reads: per-block tensors when activations are stored, the shared
        workspace (overwritten by every block, refilled by the recomputation) otherwise."""
        ws, D, H, B, S = sv.ws, self.m.inner_dim, self.m.num_heads, sv.B, sv.S
        if not sv.store:
            t = lambda name, *shape: self._b(name, shape)  # noqa: E731
            nb = SimpleNamespace(n=ws.n, qkv=ws.qkv, q=ws.q, k=ws.k, y1=t("y1", B, S, D), h1=t("h1", B, S, 4 * D))
            if double:
                nb.x1, nb.n2, nb.y2 = t("x1", B, S, D), t("n2", B, S, D), t("y2", B, S, D)
            return nb
        key = f"blk{idx}"
        nb = self._buf.get(key)
        if nb is None or nb.shape != (B, S):
            e = lambda *shape: torch.empty(shape, device=self.m.device, dtype=BF16)  # noqa: E731
            nb = SimpleNamespace(shape=(B, S), n=e(B, S, D), qkv=e(B, S, 3 * D), q=e(B, H, S, 128), k=e(B, H, S, 128),
                                 y1=e(B, S, D), h1=e(B, S, 4 * D))
            if double:
                nb.x1, nb.n2, nb.y2 = e(B, S, D), e(B, S, D), e(B, S, D)
            self._buf[key] = nb
        return nb

    def _should_store(self, B, S, nd, ns):
        if self.store_activations != "auto":
            return bool(self.store_activations)
        unit = B * S * self.m.inner_dim * 2
        return (nd * 14 + ns * 11) * unit <= self.activation_budget

    def _wgrad(self, dy, x, out=None):
        return wgrad(self._b, dy, x, out)

    # ---- forward (training): the inference kernels, one checkpoint per block -------------------------------------------
    @torch.no_grad()
    def forward(self, hidden_states, encoder_hidden_states, pooled_projections, timestep, img_ids, txt_ids, guidance):
        m, P = self.m, self.m.p
        pk = m.packed()
        D = m.inner_dim
        B, S_img, _ = hidden_states.shape
        S_txt = encoder_hidden_states.shape[1]
        ws = m._workspace(B, S_txt, S_img)
        S = ws.S
        cos, sin = m._rope(txt_ids, img_ids)
        hs = hidden_states.to(BF16).contiguous()
        enc = encoder_hidden_states.to(BF16).contiguous()
        s = ws.s
        ops.gemm(hs, P("x_embedder.weight"), P("x_embedder.bias"), out=s[:, S_txt:])
        ops.gemm(enc, P("context_embedder.weight"), P("context_embedder.bias"), out=s[:, :S_txt])
        m._conditioning(timestep, guidance, pooled_projections.to(BF16).contiguous(), ws.tproj, ws.e1, ws.t_emb, ws.g_emb,
                        ws.p_emb, ws.temb, ws.act, ws.mod)
        nblk = len(pk.double) + len(pk.single)
        ckpt = self._b("ckpt", (nblk + 1, B, S, D))
        # besides the block inputs, the attention output and its row statistics are kept (3 GB at 1024^2, bs 1): the
        # recomputation then skips the one kernel that costs as much as the block's four linears together
        o_ckpt = self._b("o_ckpt", (nblk, B, S, D))
        lse_ckpt = self._b("lse_ckpt", (nblk, B, m.num_heads, S), torch.float32)
        sv = SimpleNamespace(B=B, S=S, S_txt=S_txt, S_img=S_img, cos=cos, sin=sin, ckpt=ckpt, o_ckpt=o_ckpt, lse_ckpt=lse_ckpt,
                             enc=enc, hs=hs, ws=ws, pk=pk, nd=len(pk.double),
                             store=self._should_store(B, S, len(pk.double), len(pk.single)))
        for i in range(len(pk.double)):
            ckpt[i].copy_(s)
            self._double_forward(i, sv, s, save=sv.store, first=True)
        for j in range(len(pk.single)):
            ckpt[len(pk.double) + j].copy_(s)
            self._single_forward(j, sv, s, save=sv.store, first=True)
        ckpt[nblk].copy_(s)
        mod = ws.mod
        n_img = ws.n[:, S_txt:]
        ops.ln_modulate(s[:, S_txt:], mod[:, pk.mod_out + D: pk.mod_out + 2 * D], mod[:, pk.mod_out: pk.mod_out + D], out=n_img)
        self._saved = sv
        return ops.gemm(n_img, P("proj_out.weight"), P("proj_out.bias"))

    def _chunk(self, sv, off, j):
        D = self.m.inner_dim
        return sv.ws.mod[:, off + j * D: off + (j + 1) * D]

    def _double_forward(self, i, sv, s, save, first=False):
        """One FluxTransformerBlock on the joint buffer ``s`` (in place), un-fused epilogues.  save=True leaves n1, raw
        qkv, q, k, y1 (pre-gate), x1, n2, h1 (pre-GELU), y2 in the block's buffers for the backward; ``first`` = the
        training forward (computes and keeps the attention output; the recomputation reads it back)."""
        m, P, ws, pk = self.m, self.m.p, sv.ws, sv.pk
        D, H, S_txt = m.inner_dim, m.num_heads, sv.S_txt
        blk, p = pk.double[i], f"transformer_blocks.{i}."
        mi, mt = blk.mod_img, blk.mod_txt
        ch = lambda off, j: self._chunk(sv, off, j)  # noqa: E731
        bb = self._block_bufs(sv, i, True)
        n, qkv = bb.n, bb.qkv
        img, txt = slice(S_txt, None), slice(0, S_txt)
        ops.ln_modulate2(s, ch(mt, 0), ch(mt, 1), ch(mi, 0), ch(mi, 1), S_txt, out=n)
        ops.gemm_grouped([dict(a=n[:, img], w=blk.wqkv_img, bias=blk.bqkv_img, out=qkv[:, img]),
                          dict(a=n[:, txt], w=blk.wqkv_txt, bias=blk.bqkv_txt, out=qkv[:, txt])])
        ops.qkv_post(qkv, bb.q, bb.k, P(p + "attn.norm_q.weight"), P(p + "attn.norm_k.weight"),
                     P(p + "attn.norm_added_q.weight"), P(p + "attn.norm_added_k.weight"), sv.cos, sv.sin, S_txt)
        o = sv.o_ckpt[i]
        if first:
            ops.attention_lse(bb.q, bb.k, qkv[:, :, 2 * D:], o, sv.lse_ckpt[i])
        y1 = bb.y1
        ops.gemm_grouped([dict(a=o[:, img], w=P(p + "attn.to_out.0.weight"), bias=P(p + "attn.to_out.0.bias"), out=y1[:, img]),
                          dict(a=o[:, txt], w=P(p + "attn.to_add_out.weight"), bias=P(p + "attn.to_add_out.bias"), out=y1[:, txt])])
        x1 = bb.x1 if save else s
        ops.gate_res_fwd(s[:, img], y1[:, img], ch(mi, 2), x1[:, img])
        ops.gate_res_fwd(s[:, txt], y1[:, txt], ch(mt, 2), x1[:, txt])
        n2 = bb.n2
        ops.ln_modulate2(x1, ch(mt, 3), ch(mt, 4), ch(mi, 3), ch(mi, 4), S_txt, out=n2)
        h1 = bb.h1
        ops.gemm_grouped([dict(a=n2[:, img], w=P(p + "ff.net.0.proj.weight"), bias=P(p + "ff.net.0.proj.bias"), out=h1[:, img]),
                          dict(a=n2[:, txt], w=P(p + "ff_context.net.0.proj.weight"), bias=P(p + "ff_context.net.0.proj.bias"),
                               out=h1[:, txt])])
        ops.gelu_tanh(h1, ws.ff)
        y2 = bb.y2
        ops.gemm_grouped([dict(a=ws.ff[:, img], w=P(p + "ff.net.2.weight"), bias=P(p + "ff.net.2.bias"), out=y2[:, img]),
                          dict(a=ws.ff[:, txt], w=P(p + "ff_context.net.2.weight"), bias=P(p + "ff_context.net.2.bias"),
                               out=y2[:, txt])])
        ops.gate_res_fwd(x1[:, img], y2[:, img], ch(mi, 5), s[:, img])
        ops.gate_res_fwd(x1[:, txt], y2[:, txt], ch(mt, 5), s[:, txt])

    def _single_forward(self, j, sv, s, save, first=False):
        m, P, ws, pk = self.m, self.m.p, sv.ws, sv.pk
        D, H = m.inner_dim, m.num_heads
        blk, p = pk.single[j], f"single_transformer_blocks.{j}."
        ch = lambda k: self._chunk(sv, blk.mod, k)  # noqa: E731
        bb = self._block_bufs(sv, sv.nd + j, False)
        n, qkv = bb.n, bb.qkv
        ops.ln_modulate(s, ch(0), ch(1), out=n)
        ops.gemm(n, blk.wqkv, blk.bqkv, out=qkv)
        ops.qkv_post(qkv, bb.q, bb.k, P(p + "attn.norm_q.weight"), P(p + "attn.norm_k.weight"), None, None, sv.cos, sv.sin, 0)
        h1 = bb.h1
        ops.gemm(n, P(p + "proj_mlp.weight"), P(p + "proj_mlp.bias"), out=h1)
        ops.gelu_tanh(h1, ws.cat[:, :, D:])
        o = sv.o_ckpt[sv.nd + j]
        if first:
            ops.attention_lse(bb.q, bb.k, qkv[:, :, 2 * D:], o, sv.lse_ckpt[sv.nd + j])
        ws.cat[:, :, :D].copy_(o)          # proj_out reads [attn | mlp] as one K = 5D operand
        y1 = bb.y1
        ops.gemm(ws.cat, P(p + "proj_out.weight"), P(p + "proj_out.bias"), out=y1)
        ops.gate_res_fwd(s, y1, ch(2), s)

    # ---- backward ----------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def backward(self, dsample, sink=None, keep=True):
        """dsample: gradient of the loss w.r.t. the forward's output [B, S_img, 64] (bf16).
        Returns (grads: dict name -> tensor, d_encoder_hidden_states [B, S_txt, joint_dim] bf16).
        ``sink``: called with every block's trainable gradients as soon as that block's backward has been enqueued
        (``zero.ShardedAdamW.accumulate``: the bucket's reduce-scatter then overlaps the earlier blocks' backward);
        ``keep=False`` drops them from the returned dict afterwards (they live on in the sink)."""
        sv = self._saved
        if sv is None:
            raise RuntimeError("FluxBackward.backward() needs the forward() of the same step first")
        self.__dict__["_mod_ready"] = None
        m, P, ws, pk = self.m, self.m.p, sv.ws, sv.pk
        D, S_txt, B, S = m.inner_dim, sv.S_txt, sv.B, sv.S
        nd, ns = len(pk.double), len(pk.single)
        grads = {}
        mod = ws.mod
        g = self._b("g", (B, S, D), zero=True)           # gradient of the residual stream
        g[:, :S_txt].zero_()
        dmod_out = self._b("dmod_out", (B, 2 * D), torch.float32)
        # head: sample = proj_out(LN(h) (1 + scale) + shift); AdaLayerNormContinuous and proj_out are frozen
        dn = self._b("dn", (B, S, D))
        ops.gemm(dsample.to(BF16).contiguous(), out=dn[:, S_txt:], **self.dW("proj_out.weight"))
        ops.ln_modulate_bwd(sv.ckpt[nd + ns][:, S_txt:], dn[:, S_txt:], mod[:, pk.mod_out: pk.mod_out + D], g[:, S_txt:], dmod_out)
        s = ws.s

        def emit(block_grads):
            bg = {k: v for k, v in block_grads.items() if k in self.trainable}
            if sink is not None:
                sink(bg)
            if keep or sink is None:
                grads.update(bg)
        cb = self._c_backward_ctx(sv, g) if BLOCK_API else None     # None: this pass takes the per-launch route
        for j in reversed(range(ns)):
            if not sv.store:
                s.copy_(sv.ckpt[nd + j])
                self._single_forward(j, sv, s, save=True)
            bg = {}
            if cb is not None:
                self._single_backward_c(j, sv, cb, bg)
            else:
                self._single_backward(j, sv, g, bg)
            emit(bg)
        for i in reversed(range(nd)):
            if not sv.store:
                s.copy_(sv.ckpt[i])
                self._double_forward(i, sv, s, save=True)
            bg = {}
            if cb is not None:
                self._double_backward_c(i, sv, cb, bg)
            else:
                self._double_backward(i, sv, g, bg)
            emit(bg)
        d_enc = ops.gemm(g[:, :S_txt], **self.dW("context_embedder.weight"))
        self._saved = None
        return grads, d_enc

    def _double_backward(self, i, sv, g, grads):
        m, P, ws, pk = self.m, self.m.p, sv.ws, sv.pk
        D, H, S_txt, B, S = m.inner_dim, m.num_heads, sv.S_txt, sv.B, sv.S
        blk, p = pk.double[i], f"transformer_blocks.{i}."
        mi, mt = blk.mod_img, blk.mod_txt
        ch = lambda off, j: self._chunk(sv, off, j)  # noqa: E731
        img, txt = slice(S_txt, None), slice(0, S_txt)
        bb = self._block_bufs(sv, i, True)
        x0, x1, y1, y2, n1, n2, h1 = sv.ckpt[i], bb.x1, bb.y1, bb.y2, bb.n, bb.n2, bb.h1
        dmod = self._b("dmod_d", (B, 12 * D), torch.float32, zero=True)     # [image 6D | text 6D], chunk order of the block
        dm_i, dm_t = dmod[:, :6 * D], dmod[:, 6 * D:]
        dy = self._b("dy", (B, S, D))
        dff = self._b("dff", (B, S, 4 * D))
        dn = self._b("dn", (B, S, D))
        # -- MLP: x2 = x1 + gate_mlp * y2
        ops.gate_res_bwd(g[:, img], y2[:, img], ch(mi, 5), dy[:, img], dm_i[:, 5 * D:6 * D])
        ops.gate_res_bwd(g[:, txt], y2[:, txt], ch(mt, 5), dy[:, txt], dm_t[:, 5 * D:6 * D])
        T = self.trainable
        ffs = (("ff", img), ("ff_context", txt))      # only_tune_image_branch: false (train_denoiser.py:96-101) un-freezes the MLPs
        if any(p + f"{ff}.net.2.weight" in T for ff, _ in ffs):
            ops.gelu_tanh(h1, ws.ff)                  # ws.ff is shared scratch: this block's GELU output again
            for ff, sl in ffs:
                if p + f"{ff}.net.2.weight" in T:
                    grads[p + f"{ff}.net.2.weight"] = self._wgrad(dy[:, sl], ws.ff[:, sl])
                    grads[p + f"{ff}.net.2.bias"] = ops.colsum(dy[:, sl])
        ops.gemm_grouped([dict(a=dy[:, img], out=dff[:, img], **self.dW(p + "ff.net.2.weight")),
                          dict(a=dy[:, txt], out=dff[:, txt], **self.dW(p + "ff_context.net.2.weight"))])
        ops.gelu_bwd(h1, dff, out=dff)
        for ff, sl in ffs:
            if p + f"{ff}.net.0.proj.weight" in T:
                grads[p + f"{ff}.net.0.proj.weight"] = self._wgrad(dff[:, sl], n2[:, sl])
                grads[p + f"{ff}.net.0.proj.bias"] = ops.colsum(dff[:, sl])
        ops.gemm_grouped([dict(a=dff[:, img], out=dn[:, img], **self.dW(p + "ff.net.0.proj.weight")),
                          dict(a=dff[:, txt], out=dn[:, txt], **self.dW(p + "ff_context.net.0.proj.weight"))])
        ops.ln_modulate_bwd(x1[:, img], dn[:, img], ch(mi, 4), g[:, img], dm_i[:, 3 * D:5 * D], dx_in=g[:, img])
        ops.ln_modulate_bwd(x1[:, txt], dn[:, txt], ch(mt, 4), g[:, txt], dm_t[:, 3 * D:5 * D], dx_in=g[:, txt])
        # -- attention output projection: x1 = x0 + gate_msa * y1
        ops.gate_res_bwd(g[:, img], y1[:, img], ch(mi, 2), dy[:, img], dm_i[:, 2 * D:3 * D])
        ops.gate_res_bwd(g[:, txt], y1[:, txt], ch(mt, 2), dy[:, txt], dm_t[:, 2 * D:3 * D])
        do = self._b("do", (B, S, D))
        ops.gemm_grouped([dict(a=dy[:, img], out=do[:, img], **self.dW(p + "attn.to_out.0.weight")),
                          dict(a=dy[:, txt], out=do[:, txt], **self.dW(p + "attn.to_add_out.weight"))])
        o, lse = sv.o_ckpt[i], sv.lse_ckpt[i]
        if p + "attn.to_out.0.weight" in self.trainable:
            grads[p + "attn.to_out.0.weight"] = self._wgrad(dy[:, img], o[:, img])     # ops.gemm hands out a fresh tensor
            grads[p + "attn.to_out.0.bias"] = ops.colsum(dy[:, img])
        if p + "attn.to_add_out.weight" in self.trainable:
            grads[p + "attn.to_add_out.weight"] = self._wgrad(dy[:, txt], o[:, txt])
            grads[p + "attn.to_add_out.bias"] = ops.colsum(dy[:, txt])
        # -- joint attention
        dqkv = self._b("dqkv", (B, S, 3 * D))
        dq, dk = self._b("dq", (B, H, S, 128)), self._b("dk", (B, H, S, 128))
        dsum = ops.rowdot(do, o, H, out=self._b("dsum", (B, H, S), torch.float32))
        ops.attention_bwd(bb.q, bb.k, bb.qkv[:, :, 2 * D:], do, lse, dsum, dq, dk, dqkv[:, :, 2 * D:])
        dw = ops.qkv_post_bwd(dq, dk, bb.qkv, dqkv, P(p + "attn.norm_q.weight"), P(p + "attn.norm_k.weight"),
                              P(p + "attn.norm_added_q.weight"), P(p + "attn.norm_added_k.weight"), sv.cos, sv.sin, S_txt)
        # views of this call's own fresh [2, 2, 128] result: nothing to clone
        grads[p + "attn.norm_q.weight"], grads[p + "attn.norm_k.weight"] = dw[0, 0], dw[1, 0]
        grads[p + "attn.norm_added_q.weight"], grads[p + "attn.norm_added_k.weight"] = dw[0, 1], dw[1, 1]
        ops.gemm_grouped([dict(a=dqkv[:, img], out=dn[:, img], **self.dWp(p + "qkv_img", blk.wqkv_img, pk)),
                          dict(a=dqkv[:, txt], out=dn[:, txt], **self.dWp(p + "qkv_txt", blk.wqkv_txt, pk))])
        if p + "attn.to_q.weight" in self.trainable:
            dwqkv = self._wgrad(dqkv[:, img], n1[:, img])
            dbqkv = ops.colsum(dqkv[:, img])
            for k, nm in enumerate(("to_q", "to_k", "to_v")):
                grads[p + f"attn.{nm}.weight"] = dwqkv[k * D:(k + 1) * D]     # row blocks of this block's own [3D, D] gradient
                grads[p + f"attn.{nm}.bias"] = dbqkv[k * D:(k + 1) * D]
        if p + "attn.add_q_proj.weight" in self.trainable:
            dwqkv = self._wgrad(dqkv[:, txt], n1[:, txt])
            dbqkv = ops.colsum(dqkv[:, txt])
            for k, nm in enumerate(("add_q_proj", "add_k_proj", "add_v_proj")):
                grads[p + f"attn.{nm}.weight"] = dwqkv[k * D:(k + 1) * D]
                grads[p + f"attn.{nm}.bias"] = dbqkv[k * D:(k + 1) * D]
        ops.ln_modulate_bwd(x0[:, img], dn[:, img], ch(mi, 1), g[:, img], dm_i[:, 0:2 * D], dx_in=g[:, img])
        ops.ln_modulate_bwd(x0[:, txt], dn[:, txt], ch(mt, 1), g[:, txt], dm_t[:, 0:2 * D], dx_in=g[:, txt])
        if p + "norm1.linear.weight" in self.trainable:
            grads[p + "norm1.linear.weight"], grads[p + "norm1.linear.bias"] = self._mod_grads(dm_i, ws.act)
        if p + "norm1_context.linear.weight" in self.trainable:
            grads[p + "norm1_context.linear.weight"], grads[p + "norm1_context.linear.bias"] = self._mod_grads(dm_t, ws.act)

    def _mod_grads(self, dmod, act):
        """AdaLN linear mod = W silu(temb) + b:  dW[n, k] = sum_b dmod[b, n] act[b, k],  db[n] = sum_b dmod[b, n] -- two
        GEMMs whose K dimension is the batch (zero-padded to 64), the second against a matrix of ones."""
        B, n = dmod.shape
        D = act.shape[1]
        dT = self._b(f"dmodT{n}", (n, 64))
        aT, ones = self._mod_operands(B, act)
        ops.f32_to_bf16_transposed(dmod, dT)
        return ops.gemm(dT, aT), ops.gemm(dT, ones)[:, 0]      # fresh tensors; the bias gradient is a strided column view

    def _mod_operands(self, B, act):
        """silu(temb)^T [D, 64] and the ones matrix [64, 64] (first B columns) of the AdaLN weight-gradient GEMMs: the same for
        all 57 blocks of a step, made once per backward()."""
        D = act.shape[1]
        aT = self._b("actT", (D, 64), zero=True)
        ones = self._b("onesT", (64, 64), zero=True)
        if self.__dict__.get("_mod_ready") != (B, id(self._saved)):
            ones.zero_()
            ones[:, :B] = 1.0
            aT.zero_()
            ops.transpose(act.unsqueeze(0), torch.as_strided(aT, (1, D, B), (0, 64, 1)))
            self.__dict__["_mod_ready"] = (B, id(self._saved))
        return aT, ones

    # ---- the same two block backwards through the C entry points (csrc/blocks_bwd.hip): one call per block -------------------
    def _c_backward_ctx(self, sv, g):
        """fk_bwd_ws of this pass, or None when the pass does not fit the entry points' scope (then: the per-launch route)."""
        m, ws, pk = self.m, sv.ws, sv.pk
        D, H, B, S = m.inner_dim, m.num_heads, sv.B, sv.S
        if not (K_MAJOR >= 2 and sv.store and B == 1 and sv.S_txt % 64 == 0 and sv.S_img % 64 == 0 and D % 256 == 0):
            return None
        st = m._block_weight_structs(pk)
        for blk in list(pk.double) + list(pk.single):      # stored weights as the K-major GEMM forms take them: contiguous rows
            for w in (blk.wqkv_img, blk.wqkv_txt) if hasattr(blk, "wqkv_img") else (blk.wqkv,):
                if not (w.stride(1) == 1 and w.stride(0) == w.shape[1]):
                    return None
        from . import libfk
        b, f32 = self._b, torch.float32
        c = libfk.BwdWs()
        c.B, c.S_txt, c.S_img, c.H, c.eps = B, sv.S_txt, sv.S_img, H, 1e-6
        bufs = dict(g=g, dy=b("dy", (B, S, D)), dff=b("dff", (B, S, 4 * D)), dn=b("dn", (B, S, D)), d_o=b("do", (B, S, D)),
                    dqkv=b("dqkv", (B, S, 3 * D)), dq=b("dq", (B, H, S, 128)), dk=b("dk", (B, H, S, 128)),
                    dsum=b("dsum", (B, H, S), f32), dmod=b("dmod_d", (B, 12 * D), f32, zero=True), ff=ws.ff, cat=ws.cat,
                    cos=sv.cos, sin=sv.sin, red_ws=ops.bwd_workspace(g.device), attn_ws=ops.attention_workspace(g.device),
                    mod=ws.mod, dmodT=b("dmodT_c", (6 * D, 64)))
        bufs["actT"], bufs["onesT"] = self._mod_operands(B, ws.act)
        for k, t in bufs.items():
            setattr(c, k, t.data_ptr())
        c.attn_ws_bytes, c.mod_batch_stride = bufs["attn_ws"].numel(), ws.mod.stride(0)
        L = ops.LAUNCH
        c.gemm_variant, c.gemm_plan, c.gemm_group_m, c.gemm_mfma = L.gemm_variant, L.gemm_plan, L.gemm_group_m, L.gemm_mfma
        c.attn_grid, c.attn_passes = L.attn_grid, L.attn_bwd_passes
        c.gemm_variant_used = ctypes.pointer(ops._variant_slot())
        return SimpleNamespace(c=c, st=st, bufs=bufs, lib=libfk.load(), stream=ctypes.c_void_p(torch.cuda.current_stream().cuda_stream),
                               saved=libfk.BlockSaved, dev=g.device)

    def _saved_struct(self, cb, sv, idx, bb, double):
        sb = cb.saved()
        sb.x0, sb.n1, sb.qkv, sb.q, sb.k = sv.ckpt[idx].data_ptr(), bb.n.data_ptr(), bb.qkv.data_ptr(), bb.q.data_ptr(), bb.k.data_ptr()
        sb.y1, sb.h1, sb.o, sb.lse = bb.y1.data_ptr(), bb.h1.data_ptr(), sv.o_ckpt[idx].data_ptr(), sv.lse_ckpt[idx].data_ptr()
        if double:
            sb.x1, sb.n2, sb.y2 = bb.x1.data_ptr(), bb.n2.data_ptr(), bb.y2.data_ptr()
        return sb

    def _single_backward_c(self, j, sv, cb, grads):
        from . import libfk
        D, T, dev = self.m.inner_dim, self.trainable, cb.dev
        p = f"single_transformer_blocks.{j}."
        idx = len(sv.pk.double) + j
        bb = self._block_bufs(sv, idx, False)
        e = lambda *shape, dtype=BF16: torch.empty(shape, device=dev, dtype=dtype)  # noqa: E731
        out = {"dnorm": e(2, 2, 128, dtype=torch.float32)}          # fresh tensors, as the per-launch route hands out
        if p + "attn.to_q.weight" in T:
            out["dwqkv"], out["dbqkv"] = e(3 * D, D), e(3 * D, dtype=torch.float32)
        if p + "proj_mlp.weight" in T:
            out["dw_mlp"], out["db_mlp"] = e(4 * D, D), e(4 * D, dtype=torch.float32)
        if p + "proj_out.weight" in T:
            out["dw_out"], out["db_out"] = e(D, 5 * D), e(D, dtype=torch.float32)
        if p + "norm.linear.weight" in T:
            out["dw_mod"], out["db_mod"] = e(3 * D, D), e(3 * D, 64)
        gs = libfk.SingleBlockGrads()
        for k, t in out.items():
            setattr(gs, k, t.data_ptr())
        sb = self._saved_struct(cb, sv, idx, bb, False)
        libfk.check(cb.lib.fk_single_block_bwd(ctypes.byref(cb.c), ctypes.byref(sb), ctypes.byref(cb.st.sgl[j]), ctypes.byref(gs), cb.stream),
                    "fk_single_block_bwd")
        dw = out["dnorm"]
        grads[p + "attn.norm_q.weight"], grads[p + "attn.norm_k.weight"] = dw[0, 0], dw[1, 0]
        if "dwqkv" in out:
            for k, nm in enumerate(("to_q", "to_k", "to_v")):
                grads[p + f"attn.{nm}.weight"] = out["dwqkv"][k * D:(k + 1) * D]
                grads[p + f"attn.{nm}.bias"] = out["dbqkv"][k * D:(k + 1) * D]
        if "dw_mlp" in out:
            grads[p + "proj_mlp.weight"], grads[p + "proj_mlp.bias"] = out["dw_mlp"], out["db_mlp"]
        if "dw_out" in out:
            grads[p + "proj_out.weight"], grads[p + "proj_out.bias"] = out["dw_out"], out["db_out"]
        if "dw_mod" in out:
            grads[p + "norm.linear.weight"], grads[p + "norm.linear.bias"] = out["dw_mod"], out["db_mod"][:, 0]

    def _double_backward_c(self, i, sv, cb, grads):
        from . import libfk
        D, T, dev = self.m.inner_dim, self.trainable, cb.dev
        p = f"transformer_blocks.{i}."
        bb = self._block_bufs(sv, i, True)
        f32 = torch.float32
        e = lambda *shape, dtype=BF16: torch.empty(shape, device=dev, dtype=dtype)  # noqa: E731
        out = {"dnorm": e(2, 2, 128, dtype=f32)}
        pairs = {   # struct field stem -> (parameter stem, weight shape)
            "qkv_img": ("attn.to_q", (3 * D, D)), "qkv_txt": ("attn.add_q_proj", (3 * D, D)),
            "_out": ("attn.to_out.0", (D, D)), "_add_out": ("attn.to_add_out", (D, D)),
            "_ff1": ("ff.net.0.proj", (4 * D, D)), "_ff1_ctx": ("ff_context.net.0.proj", (4 * D, D)),
            "_ff2": ("ff.net.2", (D, 4 * D)), "_ff2_ctx": ("ff_context.net.2", (D, 4 * D)),
        }
        for stem, (name, shape) in pairs.items():
            if p + name + ".weight" in T:
                out["dw" + stem], out["db" + stem] = e(*shape), e(shape[0], dtype=f32)
        if p + "norm1.linear.weight" in T:
            out["dw_mod_img"], out["db_mod_img"] = e(6 * D, D), e(6 * D, 64)
        if p + "norm1_context.linear.weight" in T:
            out["dw_mod_txt"], out["db_mod_txt"] = e(6 * D, D), e(6 * D, 64)
        gs = libfk.DoubleBlockGrads()
        for k, t in out.items():
            setattr(gs, k, t.data_ptr())
        sb = self._saved_struct(cb, sv, i, bb, True)
        libfk.check(cb.lib.fk_double_block_bwd(ctypes.byref(cb.c), ctypes.byref(sb), ctypes.byref(cb.st.dbl[i]), ctypes.byref(gs), cb.stream),
                    "fk_double_block_bwd")
        dw = out["dnorm"]
        grads[p + "attn.norm_q.weight"], grads[p + "attn.norm_k.weight"] = dw[0, 0], dw[1, 0]
        grads[p + "attn.norm_added_q.weight"], grads[p + "attn.norm_added_k.weight"] = dw[0, 1], dw[1, 1]
        for stem, names in (("qkv_img", ("to_q", "to_k", "to_v")), ("qkv_txt", ("add_q_proj", "add_k_proj", "add_v_proj"))):
            if "dw" + stem in out:
                for k, nm in enumerate(names):
                    grads[p + f"attn.{nm}.weight"] = out["dw" + stem][k * D:(k + 1) * D]
                    grads[p + f"attn.{nm}.bias"] = out["db" + stem][k * D:(k + 1) * D]
        for stem, (name, _) in pairs.items():
            if stem.startswith("_") and "dw" + stem in out:
                grads[p + name + ".weight"], grads[p + name + ".bias"] = out["dw" + stem], out["db" + stem]
        if "dw_mod_img" in out:
            grads[p + "norm1.linear.weight"], grads[p + "norm1.linear.bias"] = out["dw_mod_img"], out["db_mod_img"][:, 0]
        if "dw_mod_txt" in out:
            grads[p + "norm1_context.linear.weight"], grads[p + "norm1_context.linear.bias"] = out["dw_mod_txt"], out["db_mod_txt"][:, 0]

    def _single_backward(self, j, sv, g, grads):
        m, P, ws, pk = self.m, self.m.p, sv.ws, sv.pk
        D, H, B, S = m.inner_dim, m.num_heads, sv.B, sv.S
        blk, p = pk.single[j], f"single_transformer_blocks.{j}."
        ch = lambda k: self._chunk(sv, blk.mod, k)  # noqa: E731
        bb = self._block_bufs(sv, len(pk.double) + j, False)
        x0, y1, n1, h1 = sv.ckpt[len(pk.double) + j], bb.y1, bb.n, bb.h1
        dmod = self._b("dmod_s", (B, 3 * D), torch.float32, zero=True)
        dy = self._b("dy", (B, S, D))
        do = self._b("do", (B, S, D))
        dff = self._b("dff", (B, S, 4 * D))
        dn = self._b("dn", (B, S, D))
        # x' = x + gate * y, y = proj_out([attn | gelu(mlp)])
        ops.gate_res_bwd(g, y1, ch(2), dy, dmod[:, 2 * D:3 * D])
        # proj_out [D, 5D]: input columns [0, D) -> attention, [D, 5D) -> MLP
        ops.gemm(dy, out=do, **self.dW(p + "proj_out.weight", slice(0, D)))
        ops.gemm(dy, out=dff, **self.dW(p + "proj_out.weight", slice(D, 5 * D)))
        ops.gelu_bwd(h1, dff, out=dff)
        dqkv = self._b("dqkv", (B, S, 3 * D))
        dq, dk = self._b("dq", (B, H, S, 128)), self._b("dk", (B, H, S, 128))
        o, lse = sv.o_ckpt[len(pk.double) + j], sv.lse_ckpt[len(pk.double) + j]
        dsum = ops.rowdot(do, o, H, out=self._b("dsum", (B, H, S), torch.float32))
        ops.attention_bwd(bb.q, bb.k, bb.qkv[:, :, 2 * D:], do, lse, dsum, dq, dk, dqkv[:, :, 2 * D:])
        dw = ops.qkv_post_bwd(dq, dk, bb.qkv, dqkv, P(p + "attn.norm_q.weight"), P(p + "attn.norm_k.weight"), None, None,
                              sv.cos, sv.sin, 0)
        grads[p + "attn.norm_q.weight"], grads[p + "attn.norm_k.weight"] = dw[0, 0], dw[1, 0]
        ops.gemm(dqkv, out=dn, **self.dWp(p + "qkv", blk.wqkv, pk))
        ops.gemm(dff, out=dn, epilogue=ops.FK_EPI_RES, res=dn, **self.dW(p + "proj_mlp.weight"))
        if p + "attn.to_q.weight" in self.trainable:
            dwqkv = self._wgrad(dqkv, n1)
            dbqkv = ops.colsum(dqkv)
            for k, nm in enumerate(("to_q", "to_k", "to_v")):
                grads[p + f"attn.{nm}.weight"] = dwqkv[k * D:(k + 1) * D]     # row blocks of this block's own [3D, D] gradient
                grads[p + f"attn.{nm}.bias"] = dbqkv[k * D:(k + 1) * D]
        if p + "proj_mlp.weight" in self.trainable:
            grads[p + "proj_mlp.weight"] = self._wgrad(dff, n1)
            grads[p + "proj_mlp.bias"] = ops.colsum(dff)
        if p + "proj_out.weight" in self.trainable:
            if sv.store:   # cat is shared scratch: rebuild [attn | gelu(mlp)] of this block
                ws.cat[:, :, :D].copy_(o)
                ops.gelu_tanh(h1, ws.cat[:, :, D:])
            grads[p + "proj_out.weight"] = self._wgrad(dy, ws.cat)
            grads[p + "proj_out.bias"] = ops.colsum(dy)
        ops.ln_modulate_bwd(x0, dn, ch(1), g, dmod[:, 0:2 * D], dx_in=g)
        if p + "norm.linear.weight" in self.trainable:
            grads[p + "norm.linear.weight"], grads[p + "norm.linear.bias"] = self._mod_grads(dmod, ws.act)


class FluxTrainFunction(torch.autograd.Function):
    """ONE autograd node for the whole MMDiT: forward = ``FluxBackward.forward``, backward = ``FluxBackward.backward``.

    This is what lets the reference's loop drive the HIP model unchanged: ``model_pred = denoiser(...)`` builds the
    node, ``accelerator.backward(loss)`` (``train_denoiser.py:1172``) reaches it with d loss / d sample, and the
    gradients come back as the node's outputs for the parameters passed in (the ones the reference's
    ``named_modules()`` loop un-froze, ``:538-543``) and for ``encoder_hidden_states`` (-> the ``denoise_projector``
    and, if it were trainable, the VLM).  Weight gradients are bf16 like the parameters; bias / norm / modulation
    gradients are reduced in fp32 and rounded once.  No gradient flows to ``hidden_states`` (the noisy latents)."""

    @staticmethod
    def forward(ctx, bw, names, hidden_states, encoder_hidden_states, pooled_projections, timestep, img_ids, txt_ids,
                guidance, *params):
        ctx.bw, ctx.names = bw, names
        ctx.enc_dtype = encoder_hidden_states.dtype
        ctx.need_enc = encoder_hidden_states.requires_grad
        return bw.forward(hidden_states, encoder_hidden_states, pooled_projections, timestep, img_ids, txt_ids, guidance)

    @staticmethod
    def backward(ctx, dsample):
        grads, d_enc = ctx.bw.backward(dsample.contiguous())
        out = []
        for n in ctx.names:
            g = grads.get(n)
            if g is None:
                raise RuntimeError(f"FluxBackward produced no gradient for {n}")
            out.append(g if g.dtype == BF16 else g.to(BF16))
        d_enc = d_enc.to(ctx.enc_dtype) if ctx.need_enc else None
        return (None, None, None, d_enc, None, None, None, None, None, *out)
