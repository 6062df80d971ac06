"""Tensor-level wrappers over the C ABI (include/fk.h): tensors in, tensors out.

Only plumbing lives here: argument checking, stride extraction, raw pointers and the current HIP
stream.  All arithmetic happens in libfk.so; nothing falls back to torch ops.
"""
import ctypes
import threading
from types import SimpleNamespace
import os

import torch

from . import libfk, param_tree
from .libfk import (FK_EPI_GATE_RES, FK_EPI_GELU_TANH, FK_EPI_NONE, FK_EPI_QKV, FK_EPI_RES, FK_EPI_SCALE,  # noqa: F401
                    FK_EPI_SILU, GemmArgs, Rows)

BF16 = torch.bfloat16


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("gpt_image_edit_amd ops need GPU tensors: the HIP path has no CPU fallback")


def pack_rope(cos, sin, check=True):
    """FluxPosEmbed's cos / sin [S, 128] (each value repeated over the two columns of its rotary pair) -> the
    [S, 64, 2] (cos, sin)-per-pair table the fused QKV epilogue reads.  Step-invariant: callers on the hot path
    pack once (transformer._rope) and pass ``cs``; passing cos / sin packs per call and checks the repetition."""
    if cos.shape != sin.shape or cos.dim() != 2 or cos.shape[1] != 128:
        raise ValueError("cos / sin must be [S, 128]")
    # check=False: tables that come straight from transformer.rope_tables (torch.equal on device tensors is a host sync)
    if check and not (torch.equal(cos[:, 0::2], cos[:, 1::2]) and torch.equal(sin[:, 0::2], sin[:, 1::2])):
        raise ValueError("cos / sin do not repeat over the columns of a rotary pair: not FluxPosEmbed tables")
    return torch.stack([cos[:, 0::2], sin[:, 0::2]], dim=-1).to(torch.float32).contiguous()


def rows_of(t):
    """(M, Rows) for a [M, K] or [B, R, K] tensor (or view) whose last dimension is contiguous."""
    if t.stride(-1) != 1:
        raise ValueError("last dimension must be contiguous")
    if t.dim() == 2:
        return t.shape[0], Rows(t.stride(0), 0, 0)
    if t.dim() == 3:
        b, r, _ = t.shape
        return b * r, Rows(t.stride(1), r, t.stride(0))
    raise ValueError(f"expected a 2-D or 3-D tensor, got {t.dim()}-D")


# Split-K workspace (fk.h: fk_gemm_args.splitk_ws): one per (device, stream) -- launches that share one must be ordered.
# 128 slots cover every grid the planner splits (two half-K workgroups per tile on at most all 256 CUs); 32 MiB each.
SPLITK_SLOTS = 256      # one per cut of a stream-K launch (= CUs) / per tile of a split-K pair launch
_SPLITK_WS = {}


def splitk_workspace(device):
    """(uint8 buffer, slots) for the current stream of ``device``; control words zeroed once (monotonic tickets after)."""
    key = (device.index if device.index is not None else torch.cuda.current_device(), torch.cuda.current_stream(device).cuda_stream)
    ws = _SPLITK_WS.get(key)
    if ws is None:
        ws = torch.zeros(SPLITK_SLOTS * libfk.FK_SPLITK_SLOT_BYTES, device=device, dtype=torch.uint8)
        _SPLITK_WS[key] = ws
    return ws, SPLITK_SLOTS


_VARIANT_USED = threading.local()


def _variant_slot():
    """This thread's int32 that every GEMM call of this module hands to the library as fk_gemm_args.variant_used (OUT)."""
    slot = getattr(_VARIANT_USED, "slot", None)
    if slot is None:
        slot = _VARIANT_USED.slot = ctypes.c_int32(0)
    return slot


def gemm_last_variant():
    """Launch form of this thread's last ``gemm`` / ``gemm_grouped`` / block-level call: 128 = 256 x 128 tiles, 256 = 256 x 256,
    384 = mixed grid, 512 = split-K pairs, 640 = stream-K ranges, 0 = the 128 x 128 kernel (tests, profiling).  The value is
    the call's own OUT field (fk_gemm_args.variant_used): the library keeps no record."""
    return int(_variant_slot().value)


def _gemm_args(a, w, bias, out, epilogue, res, gate, out_fp32, alpha, qkv=None, layout=0):
    _need_cuda(a, w, bias, out, res, gate)
    if a.dtype != BF16 or w.dtype != BF16:
        raise TypeError("fk_gemm_bf16 takes bf16 operands")
    if layout == 0:
        M, ra = rows_of(a)
        N, K = w.shape
        if a.shape[-1] != K:
            raise ValueError(f"K mismatch: a {tuple(a.shape)} vs w {tuple(w.shape)}")
        out_shape = (*a.shape[:-1], N)
    elif layout == 1:     # w is [K, N]: out = a @ w
        M, ra = rows_of(a)
        K, N = w.shape
        if a.shape[-1] != K or w.stride(1) != 1:
            raise ValueError(f"layout 1: a {tuple(a.shape)} @ w {tuple(w.shape)} (w rows contiguous)")
        out_shape = (*a.shape[:-1], N)
    elif layout == 2:     # a is [K, M] (or [B, R, M] rows = K), w is [K, N]: out = a^T @ w
        a = a[0] if a.dim() == 3 and a.shape[0] == 1 else a      # one batch: any batch stride
        w = w[0] if w.dim() == 3 and w.shape[0] == 1 else w
        K, ra = rows_of(a)
        M, N = a.shape[-1], w.shape[-1]
        Kw, rw = rows_of(w)
        if Kw != K or w.stride(-1) != 1 or (rw.rows_per_batch > 0 and rw.batch_stride != rw.rows_per_batch * rw.ld):
            raise ValueError(f"layout 2: a {tuple(a.shape)} ^T @ w {tuple(w.shape)} (uniformly strided w rows)")
        out_shape = (M, N)
    else:
        raise ValueError(f"layout {layout}")
    if out is None:
        out = torch.empty(out_shape, device=a.device, dtype=torch.float32 if out_fp32 else BF16)
    Mo, rc = rows_of(out)
    if Mo != M or out.shape[-1] != N:
        raise ValueError(f"output shape {tuple(out.shape)} does not match M={M}, N={N}")
    args = GemmArgs()
    args.layout = layout
    args.A, args.a = a.data_ptr(), ra
    args.W, args.ldw = w.data_ptr(), (w.stride(0) if layout != 2 else rw.ld)
    args.bias = bias.data_ptr() if bias is not None else None
    args.C, args.c = out.data_ptr(), rc
    if res is not None:
        Mr, rr = rows_of(res)
        if Mr != M:
            raise ValueError("residual rows mismatch")
        args.res, args.r = res.data_ptr(), rr
    if gate is not None:
        if a.dim() != 3 or gate.dim() != 2 or gate.shape[0] != a.shape[0] or gate.stride(1) != 1:
            raise ValueError("gate must be a [B, N] view matching a 3-D activation")
        args.gate = gate.data_ptr()
        args.gate_batch_stride = gate.stride(0)
        args.gate_rows_per_batch = a.shape[1]
    args.M, args.N, args.K = M, N, K
    _apply_gemm_launch(args)
    args.variant_used = ctypes.pointer(_variant_slot())
    args.epilogue, args.out_fp32, args.alpha = epilogue, int(out_fp32), float(alpha)
    if int(out_fp32) == 1:       # fp32-class VAE encoder: fp32 bias / fp32 residual added to the fp32 output
        args.f32_flags = ((1 if bias is not None and bias.dtype == torch.float32 else 0) |
                          (2 if res is not None and res.dtype == torch.float32 and epilogue == FK_EPI_NONE else 0))
    if K >= 6144 and N % 256 == 0 and M <= 256 * SPLITK_SLOTS and int(out_fp32) != 1 and layout == 0:   # the only shapes the planner may split
        ws, slots = splitk_workspace(a.device)
        args.splitk_ws, args.splitk_slots = ws.data_ptr(), slots
    if epilogue == FK_EPI_QKV:
        cs = qkv["cs"] if "cs" in qkv else pack_rope(qkv["cos"], qkv["sin"])
        _need_cuda(qkv["q_out"], qkv["k_out"], qkv["wq"], qkv["wk"], cs)
        if cs.dtype != torch.float32 or not cs.is_contiguous() or tuple(cs.shape) != (qkv["q_out"].shape[2], 64, 2):
            raise ValueError("cs must be a contiguous fp32 [S_total, 64, 2] table (see pack_rope)")
        args.q_out, args.k_out = qkv["q_out"].data_ptr(), qkv["k_out"].data_ptr()
        args.wq, args.wk = qkv["wq"].data_ptr(), qkv["wk"].data_ptr()
        args.rope_cs = cs.data_ptr()
        args.qkv_s_offset, args.qkv_s_total, args.qkv_heads = qkv["s_offset"], qkv["q_out"].shape[2], qkv["q_out"].shape[1]
    return args, out


def gemm(a, w, bias=None, out=None, epilogue=FK_EPI_NONE, res=None, gate=None, out_fp32=False, alpha=1.0, qkv=None, layout=0):
    """out = epilogue(a @ w.T + bias).  a: [M,K] / [B,R,K] view; w: [N,K]; out likewise (may alias res).

    out_fp32: True / 1 = fp32(acc + bias) by the 128 x 128 register-staged kernel; 2 = the same from the LARGE-TILE kernels'
    own main loops (256 x 256 / 256 x 128 / mixed / split-K, whichever the launch plan or ``gemm_set_variant`` selects) --
    the parity build that holds the kernels carrying the FLOPs to rtol 1e-3 / atol 1e-4.

    layout (fk_gemm_args.layout): 1 = ``w`` is [K, N] and out = a @ w (the data gradient reads the weight as stored);
    2 = ``a`` is [K, M] / [B, R, M] and ``w`` [K, N] / [B, R, N], out [M, N] = a^T @ w (the weight gradient reads both
    operands token-major).  N % 256 == 0 (2: M % 256 == 0), K % 64 == 0, no epilogue (1: FK_EPI_RES allowed).

    gate: [B, N] view (row stride = batch stride); used with FK_EPI_GATE_RES and a 3-D ``a``.
    qkv (FK_EPI_QKV): dict(q_out, k_out [B,H,S_total,128], wq, wk [128], cs fp32 [S_total,64,2] (pack_rope; or cos, sin
    fp32 [S_total,128], packed per call), s_offset)
    -- the fused QKV projection: q / k thirds get RMSNorm + RoPE + head-major layout, the v third lands in out.
    """
    args, out = _gemm_args(a, w, bias, out, epilogue, res, gate, out_fp32, alpha, qkv, layout)
    libfk.check(libfk.load().fk_gemm_bf16(ctypes.byref(args), _stream()), "fk_gemm_bf16")
    return out


# ---- launch controls --------------------------------------------------------------------------------------------------------
# The C library keeps NO mutable launch state (round 5): every call carries its launch form (fk_gemm_args.variant / plan /
# group_m / mfma, the `grid` / `passes` arguments of the attention entry points, fk_block_ws.gemm_* / attn_grid).  What the
# HOST wants as its defaults lives here, in the application layer: the setters below change this module's defaults (and only
# this process's Python callers see them); the environment variables FK_GEMM_PLAN, FK_GEMM_BN, FK_GEMM_GROUP_M, FK_GEMM_MFMA,
# FK_ATTN_SPLIT, FK_ATTN_BWD of rounds 2-4 are read once, here.
FK_GEMM_PLAN_EXPLICIT = 8


def _env_int(name, default):
    v = os.environ.get(name)
    return default if v is None or v == "" else int(v)


LAUNCH = SimpleNamespace(
    gemm_variant=_env_int("FK_GEMM_BN", 0),         # 0 = launch plan; 128 / 256 / 384 / 512 / 640 force a form where it applies
    gemm_plan=(FK_GEMM_PLAN_EXPLICIT | (_env_int("FK_GEMM_PLAN", 3) & 7)) if os.environ.get("FK_GEMM_PLAN") else 0,
    gemm_group_m=_env_int("FK_GEMM_GROUP_M", 0),
    gemm_mfma=_env_int("FK_GEMM_MFMA", 0),          # 0 = built default (16); 16 / 32
    attn_grid={0: -1, 1: 0}.get(_env_int("FK_ATTN_SPLIT", 1), _env_int("FK_ATTN_SPLIT", 1)),   # 0 default, -1 plain grid, >= 2 forced
    attn_bwd_passes=0 if _env_int("FK_ATTN_BWD", 1) else 3)
_LAUNCH_CFG_EPOCH = [0]


def launch_config_epoch():
    """Bumped by every launch-control setter of this module: anything that froze host-side launch decisions (a captured
    hipGraph of the denoise loop) keys on it."""
    return _LAUNCH_CFG_EPOCH[0]


def _set_launch(**kw):
    for k, v in kw.items():
        setattr(LAUNCH, k, v)
    _LAUNCH_CFG_EPOCH[0] += 1


def gemm_set_variant(variant):
    """Force the launch form of the large-tile GEMMs where it applies: 128 = 256 x 128 tiles, 256 = 256 x 256 tiles, 384 =
    mixed grid, 512 = split-K pairs, 640 = stream-K ranges, 1024 = the 4-wave hand-placed 256 x 256 kernel (gemm10_kernel);
    0 = the launch plan chooses per problem.  (fk_gemm_args.variant)"""
    if int(variant) not in (0, 128, 256, 384, 512, 640, 1024):
        raise ValueError(f"gemm_set_variant: {variant} is not one of 0, 128, 256, 384, 512, 640, 1024")
    _set_launch(gemm_variant=int(variant))


def gemm_set_plan(allow):
    """Which launch forms the plan may use (fk_gemm_args.plan): bit 0 = mixed grids (bit-identical results), bit 1 = split-K
    pairs (last-bit differences against the unsplit sum, so a sample's result then depends on how full the grid is), bit 2 =
    stream-K ranges.  ``gemm_set_plan(1)`` = batch-invariant; ``gemm_set_plan(3)`` = the default."""
    if not 0 <= int(allow) <= 7:
        raise ValueError(f"gemm_set_plan: {allow} is not in 0..7")
    _set_launch(gemm_plan=FK_GEMM_PLAN_EXPLICIT | int(allow))


def gemm_set_group_m(depth):
    """Depth (in 256-row tiles) of the grouped tile order (fk_gemm_args.group_m; results do not depend on it); 0 = default."""
    if not 0 <= int(depth) <= 4096:
        raise ValueError(f"gemm_set_group_m: {depth} is not 0 (default) or 1..4096 row tiles")
    _set_launch(gemm_group_m=int(depth))


def gemm_set_mfma(shape):
    """MFMA shape of the layout-0 large-tile GEMM kernels (fk_gemm_args.mfma): 32 (v_mfma_f32_32x32x16_bf16), 16
    (v_mfma_f32_16x16x32_bf16) or 0 = the built default (16).  The two differ in the last bits; see include/fk.h."""
    if int(shape) not in (0, 16, 32):
        raise ValueError(f"gemm_set_mfma: {shape} is not 0, 16 or 32")
    _set_launch(gemm_mfma=int(shape))


def _apply_gemm_launch(args):
    args.variant, args.plan, args.group_m, args.mfma = LAUNCH.gemm_variant, LAUNCH.gemm_plan, LAUNCH.gemm_group_m, LAUNCH.gemm_mfma


def gemm_grouped(problems, epilogue=FK_EPI_NONE):
    """Several independent GEMMs sharing (N, K, epilogue) in ONE launch.  ``problems`` is a list of dicts
    with the keyword arguments of :func:`gemm` (a, w, bias, out, res, gate); returns the outputs."""
    n = len(problems)
    arr = (GemmArgs * n)()
    outs = []
    for i, pr in enumerate(problems):
        args, out = _gemm_args(pr["a"], pr["w"], pr.get("bias"), pr.get("out"), epilogue, pr.get("res"),
                               pr.get("gate"), False, 1.0, pr.get("qkv"), pr.get("layout", 0))
        arr[i] = args
        outs.append(out)
    libfk.check(libfk.load().fk_gemm_bf16_grouped(arr, n, _stream()), "fk_gemm_bf16_grouped")
    return outs


def ln_modulate(x, shift, scale, out=None, eps=1e-6):
    """out = LN(x) * (1 + scale[b]) + shift[b]; x/out: [B,R,D] views, shift/scale: [B,D] views."""
    _need_cuda(x, shift, scale, out)
    if x.dim() != 3:
        raise ValueError("x must be [B, R, D]")
    B, R, D = x.shape
    if out is None:
        out = torch.empty((B, R, D), device=x.device, dtype=BF16)
    M, rx = rows_of(x)
    _, ro = rows_of(out)
    if shift.stride(0) != scale.stride(0) or shift.stride(1) != 1 or scale.stride(1) != 1:
        raise ValueError("shift/scale must be [B, D] views with a common batch stride")
    lib = libfk.load()
    libfk.check(lib.fk_ln_modulate_bf16(_ptr(x), rx, _ptr(out), ro, _ptr(shift), _ptr(scale),
                                        shift.stride(0), R, M, D, eps, _stream()), "fk_ln_modulate_bf16")
    return out


def ln_modulate2(x, shift_a, scale_a, shift_b, scale_b, split, out=None, eps=1e-6):
    """Joint-sequence LN+modulate: rows [0, split) of every batch use (shift_a, scale_a), the rest (b)."""
    _need_cuda(x, shift_a, scale_a, shift_b, scale_b, out)
    B, R, D = x.shape
    if out is None:
        out = torch.empty((B, R, D), device=x.device, dtype=BF16)
    M, rx = rows_of(x)
    _, ro = rows_of(out)
    st = shift_a.stride(0)
    if any(t.stride(0) != st or t.stride(1) != 1 for t in (scale_a, shift_b, scale_b)):
        raise ValueError("modulation views must share one batch stride")
    libfk.check(libfk.load().fk_ln_modulate2_bf16(_ptr(x), rx, _ptr(out), ro, _ptr(shift_a), _ptr(scale_a),
                                                  _ptr(shift_b), _ptr(scale_b), split, st, R, M, D, eps, _stream()),
                "fk_ln_modulate2_bf16")
    return out


def qkv_post(qkv, q_out, k_out, wq_img, wk_img, wq_txt, wk_txt, cos, sin, s_txt, eps=1e-6):
    """RMSNorm + RoPE + re-layout of q, k: qkv [B,S,3*H*128] -> q/k [B,H,S,128] (V stays in qkv)."""
    _need_cuda(qkv, q_out, k_out, cos, sin)
    B, S, D3 = qkv.shape
    H = D3 // 384
    if not qkv.is_contiguous():
        raise ValueError("qkv must be contiguous")
    lib = libfk.load()
    libfk.check(lib.fk_qkv_post_bf16(_ptr(qkv), _ptr(q_out), _ptr(k_out), _ptr(wq_img), _ptr(wk_img), _ptr(wq_txt),
                                     _ptr(wk_txt), _ptr(cos), _ptr(sin), B, S, s_txt, H, eps, _stream()),
                "fk_qkv_post_bf16")


_ATTN_WS = {}


def attention_workspace(device):
    """Stream-K workspace of the attention forward for the current stream of ``device`` (fk_attention_ws_bytes(), zeroed
    once: monotonic tickets afterwards)."""
    key = (device.index if device.index is not None else torch.cuda.current_device(), torch.cuda.current_stream(device).cuda_stream)
    ws = _ATTN_WS.get(key)
    if ws is None:
        ws = torch.zeros(libfk.load().fk_attention_ws_bytes(), device=device, dtype=torch.uint8)
        _ATTN_WS[key] = ws
    return ws


def attention_set_split(mode):
    """Grid of the attention forward and of the backward's dQ pass (the `grid` argument of fk_attention_fwd_ws_bf16 /
    fk_attention_bwd_ws_bf16): 1 = stream-K grids where the plain grid wastes a round (default), 0 = never (batch-invariant),
    >= 2 = a persistent grid of that many workgroups wherever every block is cut at most once (test hook)."""
    mode = int(mode)
    if mode < 0:
        raise ValueError(f"attention_set_split: {mode} is not 0, 1 or a workgroup count >= 2")
    _set_launch(attn_grid={0: -1, 1: 0}.get(mode, mode))


def attention(q, k, v, out, scale=None, lse=None):
    """out[b, s, h*128:(h+1)*128] = softmax(q k^T * scale) v.

    q, k: [B,H,S,128] contiguous; v: [B,S,H*128] view with contiguous last dim (e.g. qkv[:, :, 2D:]);
    out: [B,S,>=H*128] view; lse: optional fp32 [B,H,S] (log2 domain, for :func:`attention_bwd`)."""
    _need_cuda(q, k, v, out, lse)
    B, H, S, hd = q.shape
    if hd != 128:
        raise ValueError("head_dim must be 128")
    if v.dim() != 3 or v.shape[-1] != H * 128 or v.stride(2) != 1:
        raise ValueError("v must be a [B, S, H*128] view with a contiguous last dimension")
    if lse is not None and (lse.dtype != torch.float32 or lse.shape != (B, H, S) or not lse.is_contiguous()):
        raise ValueError("lse must be a contiguous fp32 [B,H,S] tensor")
    if scale is None:
        scale = hd ** -0.5
    ws = attention_workspace(q.device)
    libfk.check(libfk.load().fk_attention_fwd_ws_bf16(_ptr(q), _ptr(k), _ptr(v), _ptr(out), _ptr(lse), B, H, S, v.stride(1),
                                                     v.stride(0), out.stride(1), out.stride(0), scale, _ptr(ws), ws.numel(),
                                                     LAUNCH.attn_grid, _stream()), "fk_attention_fwd_bf16")
    return out


def attention_f32_debug(q, k, v, scale=None):
    """Parity build of :func:`attention` (fk_attention_fwd_f32_debug): fp32 [B, S, H*128] output, P as hi + lo bf16."""
    _need_cuda(q, k, v)
    B, H, S, hd = q.shape
    if hd != 128 or v.dim() != 3 or v.shape[-1] != H * 128 or v.stride(2) != 1:
        raise ValueError("q, k: [B,H,S,128]; v: [B,S,H*128] view with a contiguous last dimension")
    out = torch.empty((B, S, H * 128), device=q.device, dtype=torch.float32)
    libfk.check(libfk.load().fk_attention_fwd_f32_debug(_ptr(q), _ptr(k), _ptr(v), _ptr(out), B, H, S, v.stride(1),
                                                       v.stride(0), out.stride(1), out.stride(0),
                                                       hd ** -0.5 if scale is None else scale, _stream()),
                "fk_attention_fwd_f32_debug")
    return out


def silu(x, out=None):
    _need_cuda(x)
    if out is None:
        out = torch.empty_like(x)
    libfk.check(libfk.load().fk_silu_bf16(_ptr(x), _ptr(out), x.numel(), _stream()), "fk_silu_bf16")
    return out


def add3(a, b, c, out=None):
    _need_cuda(a, b, c)
    if out is None:
        out = torch.empty_like(a)
    libfk.check(libfk.load().fk_add3_bf16(_ptr(a), _ptr(b), _ptr(c), _ptr(out), a.numel(), _stream()),
                "fk_add3_bf16")
    return out


def true_cfg(pos, neg, scale, out=None):
    """out = neg + scale * (pos - neg) with the reference's bf16 rounding points (flux_pipeline.py:1095)."""
    _need_cuda(pos, neg, out)
    if pos.shape != neg.shape or pos.dtype != BF16 or neg.dtype != BF16 or not (pos.is_contiguous() and neg.is_contiguous()):
        raise ValueError("true_cfg takes two contiguous bf16 tensors of one shape")
    if out is None:
        out = torch.empty_like(pos)
    libfk.check(libfk.load().fk_true_cfg_bf16(_ptr(pos), _ptr(neg), _ptr(out), float(scale), pos.numel(), _stream()),
                "fk_true_cfg_bf16")
    return out


def timestep_proj(v, freqs, out=None):
    """[B] (bf16 or fp32) -> [B,256] bf16 sinusoid of bf16(v)*1000 (cos first)."""
    _need_cuda(v, freqs)
    B = v.shape[0]
    if v.dtype not in (BF16, torch.float32):
        raise TypeError("timestep must be bf16 or fp32")
    if out is None:
        out = torch.empty((B, 256), device=v.device, dtype=BF16)
    libfk.check(libfk.load().fk_timestep_proj(_ptr(v), int(v.dtype == torch.float32), _ptr(freqs), _ptr(out),
                                              B, _stream()), "fk_timestep_proj")
    return out


def euler_step(x, v, s_tgt, dsigma):
    """In place: x[:, :s_tgt] += bf16(bf16(dsigma) * v[:, :s_tgt]) with the reference's rounding."""
    _need_cuda(x, v)
    B, _, C = x.shape
    libfk.check(libfk.load().fk_euler_step_bf16(_ptr(x), x.stride(0), _ptr(v), v.stride(0), B, s_tgt, C,
                                                float(dsigma), _stream()), "fk_euler_step_bf16")
    return x


def transpose(src, dst):
    """dst[b, c, r] = src[b, r, c] for 3-D views with contiguous last dims."""
    _need_cuda(src, dst)
    Bn, R, C = src.shape
    libfk.check(libfk.load().fk_transpose_bf16(_ptr(src), src.stride(1), src.stride(0), _ptr(dst),
                                               dst.stride(1), dst.stride(0), R, C, Bn, _stream()),
                "fk_transpose_bf16")
    return dst


def attention_hd512(q, k, v, out=None, scale=None):
    """Fused single-head attention with head dimension 512 (the VAE mid block): q, k, v bf16 [B, S, 512] views with a common
    row / batch stride (column blocks of one fused projection), out [B, S, 512]; nothing of size S x S is allocated."""
    _need_cuda(q, k, v, out)
    B, S, C = q.shape
    if C != 512 or k.shape != q.shape or v.shape != q.shape or q.dtype != BF16:
        raise ValueError("attention_hd512 takes bf16 [B, S, 512] q, k, v")
    if not (q.stride() == k.stride() == v.stride()) or q.stride(2) != 1:
        raise ValueError("attention_hd512: q, k, v must share their strides (views of one projection buffer)")
    if out is None:
        out = torch.empty((B, S, C), device=q.device, dtype=BF16)
    if out.stride(2) != 1:
        raise ValueError("attention_hd512: out rows must be contiguous")
    libfk.check(libfk.load().fk_attention_hd512_bf16(_ptr(q), _ptr(k), _ptr(v), q.stride(1), q.stride(0), _ptr(out), out.stride(1),
                                                     out.stride(0), B, S, float(scale if scale is not None else C ** -0.5), _stream()),
                "fk_attention_hd512_bf16")
    return out


def softmax_rows(x, out=None):
    """fp32 [rows, n] -> bf16 softmax rows."""
    _need_cuda(x)
    rows, n = x.shape
    if out is None:
        out = torch.empty((rows, n), device=x.device, dtype=BF16)
    libfk.check(libfk.load().fk_softmax_rows(_ptr(x), x.stride(0), _ptr(out), out.stride(0), rows, n,
                                             _stream()), "fk_softmax_rows")
    return out



# ---- fp32-class VAE encoder (include/fk.h "fp32-class encoder"; reference train_denoiser.py:458,887-918) ---------------
F32 = torch.float32


def split_f32_rows(x, out, parts=3, weight_order=False):
    """fp32 [rows, n] view -> bf16 parts in ``out`` [rows, parts * n]-like view: part p of x[r, c] at out[r, p*ps + c] with
    ps = out.shape[1] // parts.  Order (hi, lo, hi), or (hi, hi, lo) for the weight side of a product."""
    _need_cuda(x, out)
    rows, n = x.shape
    ps = out.shape[1] // parts
    if x.dtype != F32 or out.dtype != BF16 or out.shape[0] != rows or ps < n or x.stride(1) != 1 or out.stride(1) != 1:
        raise ValueError("split_f32_rows: fp32 [rows, n] -> bf16 [rows, parts * ps], ps >= n")
    libfk.check(libfk.load().fk_split_f32_rows(_ptr(x), x.stride(0), _ptr(out), out.stride(0), ps, rows, n, parts,
                                               int(weight_order), _stream()), "fk_split_f32_rows")
    return out


def group_norm_f32_parts(x, gamma, beta, silu, parts, eps=1e-6):
    """GroupNorm(32) (+ SiLU) of fp32 NHWC x (fp32 gamma / beta) -> bf16 parts [B, ..., parts * C]."""
    _need_cuda(x, gamma, beta)
    B, C = x.shape[0], x.shape[-1]
    HW = x.numel() // (B * C)
    if x.dtype != F32 or gamma.dtype != F32 or beta.dtype != F32 or not x.is_contiguous():
        raise TypeError("group_norm_f32_parts takes contiguous fp32 tensors")
    lib = libfk.load()
    ws, stats = _gn_workspace(lib, B, HW, C, x.device)
    out = torch.empty((*x.shape[:-1], parts * C), device=x.device, dtype=BF16)
    libfk.check(lib.fk_groupnorm_f32_nhwc(_ptr(x), _ptr(out), _ptr(stats), _ptr(ws), _ptr(gamma), _ptr(beta), B, HW, C, 32,
                                          eps, int(silu), parts, _stream()), "fk_groupnorm_f32_nhwc")
    return out


def conv2d_nhwc_f32out(x_parts, w_packed, bias, cout, ksize=3, stride=1, pad=1, res=None, halo=False):
    """x_parts: [B,H,W,parts*C] bf16 operand parts; w_packed [cout, Kpad] bf16 over the same channel layout; bias / res /
    result fp32.  ``halo=True`` (3 x 3, stride 1, pad 1, parts*C % 64 == 0): the LDS halo-tiled kernel instead of the
    implicit GEMM."""
    _need_cuda(x_parts, w_packed, bias, res)
    B, Hin, Win, Cin = x_parts.shape
    if stride == 1:
        Hout, Wout = Hin, Win
    else:
        Hout, Wout = (Hin + 1 - 3) // 2 + 1, (Win + 1 - 3) // 2 + 1
    if x_parts.dtype != BF16 or bias.dtype != F32 or (res is not None and (res.dtype != F32 or not res.is_contiguous())) \
            or not x_parts.is_contiguous():
        raise TypeError("conv2d_nhwc_f32out: bf16 parts in, fp32 bias / residual")
    out = torch.empty((B, Hout, Wout, cout), device=x_parts.device, dtype=F32)
    a = libfk.ConvArgs()
    a.x, a.w, a.bias, a.y = x_parts.data_ptr(), w_packed.data_ptr(), bias.data_ptr(), out.data_ptr()
    a.res = res.data_ptr() if res is not None else None
    a.B, a.Hin, a.Win, a.Cin, a.Cout = B, Hin, Win, Cin, cout
    a.ksize, a.stride, a.pad, a.upsample2x = ksize, stride, pad, 0
    a.Hout, a.Wout = Hout, Wout
    if halo:
        libfk.check(libfk.load().fk_conv3x3_halo_f32out(ctypes.byref(a), _stream()), "fk_conv3x3_halo_f32out")
    else:
        libfk.check(libfk.load().fk_conv2d_nhwc_f32out(ctypes.byref(a), _stream()), "fk_conv2d_nhwc_f32out")
    return out


def nchw_f32_to_nhwc_parts(x, cpad, parts):
    _need_cuda(x)
    B, C, H, W = x.shape
    if x.dtype != F32 or not x.is_contiguous():
        raise TypeError("nchw_f32_to_nhwc_parts takes contiguous fp32 NCHW")
    out = torch.empty((B, H, W, parts * cpad), device=x.device, dtype=BF16)
    libfk.check(libfk.load().fk_nchw_f32_to_nhwc_parts(_ptr(x), _ptr(out), B, C, cpad, H, W, parts, _stream()),
                "fk_nchw_f32_to_nhwc_parts")
    return out


def nhwc_f32_to_nchw(x, c, add=0.0, mul=1.0):
    _need_cuda(x)
    B, H, W, cpad = x.shape
    if x.dtype != F32 or not x.is_contiguous():
        raise TypeError("nhwc_f32_to_nchw takes contiguous fp32 NHWC")
    out = torch.empty((B, c, H, W), device=x.device, dtype=F32)
    libfk.check(libfk.load().fk_nhwc_f32_to_nchw(_ptr(x), _ptr(out), B, c, cpad, H, W, float(add), float(mul), _stream()),
                "fk_nhwc_f32_to_nchw")
    return out


def softmax_rows_parts(x, out):
    """fp32 [rows, n] -> the bf16 parts (hi, lo, hi) of the fp32 softmax in ``out`` [rows, 3 * ps]."""
    _need_cuda(x, out)
    rows, n = x.shape
    ps = out.shape[1] // 3
    libfk.check(libfk.load().fk_softmax_rows_parts(_ptr(x), x.stride(0), _ptr(out), out.stride(0), ps, rows, n, _stream()),
                "fk_softmax_rows_parts")
    return out


# ---- backward pass of the MMDiT (include/fk.h "backward pass"; reference: autograd, train_denoiser.py:1172) -----------
_BWD_WS = {}


def bwd_workspace(device):
    """fp32 scratch of the two-stage reductions (one per device; kernels on one stream are ordered)."""
    key = str(device)
    ws = _BWD_WS.get(key)
    if ws is None:
        ws = torch.empty(libfk.load().fk_bwd_ws_floats(), device=device, dtype=torch.float32)
        _BWD_WS[key] = ws
    return ws


def attn_view(t, head_major):
    """fk_attn_view of a bf16 tensor: head-major [B,H,S,128] (contiguous rows) or token-major [B,S,H*128] view."""
    v = libfk.AttnView()
    v.p = t.data_ptr()
    if head_major:
        if t.dim() != 4 or t.shape[3] != 128 or t.stride(3) != 1:
            raise ValueError("head-major view must be [B,H,S,128] with a contiguous last dimension")
        v.ld, v.head_stride, v.batch_stride = t.stride(2), t.stride(1), t.stride(0)
    else:
        if t.dim() != 3 or t.stride(2) != 1:
            raise ValueError("token-major view must be [B,S,H*128] with a contiguous last dimension")
        v.ld, v.head_stride, v.batch_stride = t.stride(1), 128, t.stride(0)
    return v


def attention_lse(q, k, v, out, lse, scale=None):
    """:func:`attention` that also writes lse [B,H,S] fp32 (log2 domain) for :func:`attention_bwd`."""
    if lse is None:
        raise ValueError("attention_lse needs an lse tensor")
    return attention(q, k, v, out, scale=scale, lse=lse)


def rowdot(a, c, heads, out=None):
    """out[b,h,s] = sum_d a[b,s,h*128+d] * c[b,s,h*128+d]; a, c: [B,S,heads*128] views."""
    _need_cuda(a, c)
    B, S, _ = a.shape
    if out is None:
        out = torch.empty((B, heads, S), device=a.device, dtype=torch.float32)
    libfk.check(libfk.load().fk_rowdot_bf16(_ptr(a), a.stride(1), a.stride(0), _ptr(c), c.stride(1), c.stride(0), _ptr(out),
                                            B, S, heads, _stream()), "fk_rowdot_bf16")
    return out


def attention_bwd_set_mode(mode):
    """`passes` of fk_attention_bwd_ws_bf16: 1 = dQ pass + paired dK / dV pass (default), 0 = three passes."""
    if int(mode) not in (0, 1):
        raise ValueError(f"attention_bwd_set_mode: {mode} is not 0 (three passes) or 1 (dQ pass + paired dK / dV pass)")
    _set_launch(attn_bwd_passes=0 if int(mode) else 3)


def attention_bwd(q, k, v, dout, lse, dsum, dq, dk, dv, scale=None):
    """dq, dk [B,H,S,128] and dv ([B,S,H*128] view) of softmax(q k^T scale) v given dout ([B,S,H*128] view), lse, dsum."""
    _need_cuda(q, k, v, dout, lse, dsum, dq, dk, dv)
    B, H, S, hd = q.shape
    views = [attn_view(q, True), attn_view(k, True), attn_view(v, False), attn_view(dout, False),
             attn_view(dq, True), attn_view(dk, True), attn_view(dv, False)]
    r = [ctypes.byref(x) for x in views]
    ws = attention_workspace(q.device)      # the forward's stream-K workspace: the dQ pass uses the same scheme
    libfk.check(libfk.load().fk_attention_bwd_ws_bf16(r[0], r[1], r[2], r[3], _ptr(lse), _ptr(dsum), r[4], r[5], r[6], B, H, S,
                                                     hd ** -0.5 if scale is None else scale, _ptr(ws), ws.numel(), LAUNCH.attn_grid,
                                                     LAUNCH.attn_bwd_passes, _stream()),
                "fk_attention_bwd_bf16")


def ln_modulate_bwd(x, dn, scale, dx_out, dmod_shift, dx_in=None, eps=1e-6):
    """Adjoint of :func:`ln_modulate` for one stream view x / dn / dx [B,R,D]; scale: [B,D] view of the modulation
    vector; dmod_shift: fp32 [B, >=2D] view whose columns [0,D) receive dshift and [D,2D) dscale."""
    _need_cuda(x, dn, scale, dx_out, dmod_shift, dx_in)
    B, R, D = x.shape
    M, rx = rows_of(x)
    _, rg = rows_of(dn)
    _, ro = rows_of(dx_out)
    ri = rows_of(dx_in)[1] if dx_in is not None else Rows(0, 0, 0)
    if dmod_shift.dtype != torch.float32 or dmod_shift.stride(1) != 1 or dmod_shift.shape[1] < 2 * D:
        raise ValueError("dmod_shift must be an fp32 [B, >= 2D] view")
    base = dmod_shift.data_ptr()
    libfk.check(libfk.load().fk_ln_modulate_bwd_bf16(_ptr(x), rx, _ptr(dn), rg, _ptr(scale), scale.stride(0), R, _ptr(dx_in), ri,
                                                    _ptr(dx_out), ro, ctypes.c_void_p(base), ctypes.c_void_p(base + 4 * D),
                                                    dmod_shift.stride(0), _ptr(bwd_workspace(x.device)), B, D, eps, _stream()),
                "fk_ln_modulate_bwd_bf16")
    return dx_out


def gate_res_bwd(dout, y, gate, dy, dgate):
    """dy = dout * gate[b]; dgate[b] = sum_s dout * y.  dout / y / dy: [B,R,N] views; gate: [B,N] view; dgate fp32 [B,N] view."""
    _need_cuda(dout, y, gate, dy, dgate)
    B, R, N = dout.shape
    _, r0 = rows_of(dout)
    _, r1 = rows_of(y)
    _, r2 = rows_of(dy)
    libfk.check(libfk.load().fk_gate_res_bwd_bf16(_ptr(dout), r0, _ptr(y), r1, _ptr(gate), gate.stride(0), R, _ptr(dy), r2,
                                                 _ptr(dgate), dgate.stride(0), _ptr(bwd_workspace(dout.device)), B, N, _stream()),
                "fk_gate_res_bwd_bf16")
    return dy


def gelu_bwd(h, df, out=None):
    """out = df * gelu_tanh'(h) (contiguous tensors of one shape)."""
    _need_cuda(h, df)
    if not (h.is_contiguous() and df.is_contiguous()):
        raise ValueError("gelu_bwd takes contiguous tensors")
    if out is None:
        out = torch.empty_like(df)
    libfk.check(libfk.load().fk_gelu_bwd_bf16(_ptr(h), _ptr(df), _ptr(out), h.numel(), _stream()), "fk_gelu_bwd_bf16")
    return out


def silu_bwd(h, df, out=None):
    """out = df * silu'(h) (contiguous tensors of one shape)."""
    _need_cuda(h, df)
    if not (h.is_contiguous() and df.is_contiguous()):
        raise ValueError("silu_bwd takes contiguous tensors")
    if out is None:
        out = torch.empty_like(df)
    libfk.check(libfk.load().fk_silu_bwd_bf16(_ptr(h), _ptr(df), _ptr(out), h.numel(), _stream()), "fk_silu_bwd_bf16")
    return out


def qkv_post_bwd(dq, dk, qkv, dqkv, wq_img, wk_img, wq_txt, wk_txt, cos, sin, s_txt, eps=1e-6):
    """Adjoint of :func:`qkv_post`; returns dw fp32 [2 (q,k)][2 (image,text)][128]."""
    _need_cuda(dq, dk, qkv, dqkv, cos, sin)
    B, S, D3 = qkv.shape
    H = D3 // 384
    if not (qkv.is_contiguous() and dqkv.is_contiguous() and dq.is_contiguous() and dk.is_contiguous()):
        raise ValueError("qkv_post_bwd takes contiguous tensors")
    dw = torch.empty((2, 2, 128), device=qkv.device, dtype=torch.float32)
    libfk.check(libfk.load().fk_qkv_post_bwd_bf16(_ptr(dq), _ptr(dk), _ptr(qkv), _ptr(dqkv), _ptr(wq_img), _ptr(wk_img),
                                                 _ptr(wq_txt), _ptr(wk_txt), _ptr(cos), _ptr(sin), _ptr(dw),
                                                 _ptr(bwd_workspace(qkv.device)), B, S, s_txt, H, eps, _stream()),
                "fk_qkv_post_bwd_bf16")
    return dw


def gate_res_fwd(res, y, gate, out):
    """out = res + gate[b] * y with the fused epilogue's rounding; res / y / out: [B,R,N] views, gate: [B,N] view."""
    _need_cuda(res, y, gate, out)
    B, R, N = y.shape
    M, r1 = rows_of(y)
    _, r0 = rows_of(res)
    _, r2 = rows_of(out)
    libfk.check(libfk.load().fk_gate_res_fwd_bf16(_ptr(res), r0, _ptr(y), r1, _ptr(gate), gate.stride(0), R, _ptr(out), r2, M, N,
                                                 _stream()), "fk_gate_res_fwd_bf16")
    return out


def gelu_tanh(x, out):
    """out = gelu_tanh(x) (bf16, the GEMM epilogue's function); x / out: [M,N] or [B,R,N] views."""
    _need_cuda(x, out)
    M, r0 = rows_of(x)
    _, r1 = rows_of(out)
    libfk.check(libfk.load().fk_gelu_tanh_bf16(_ptr(x), r0, _ptr(out), r1, M, x.shape[-1], _stream()), "fk_gelu_tanh_bf16")
    return out


def f32_to_bf16_transposed(src, dst):
    """dst[c, r] = bf16(src[r, c]) for an fp32 [R,C] view and a bf16 [C, ld >= R] buffer (extra columns zeroed)."""
    _need_cuda(src, dst)
    R, C = src.shape
    if src.dtype != torch.float32 or src.stride(1) != 1 or dst.dtype != BF16 or not dst.is_contiguous() or dst.shape[0] != C:
        raise ValueError("f32_to_bf16_transposed: fp32 [R,C] view -> contiguous bf16 [C, ld]")
    libfk.check(libfk.load().fk_f32_to_bf16_transposed(_ptr(src), src.stride(0), _ptr(dst), dst.shape[1], R, C, _stream()),
                "fk_f32_to_bf16_transposed")
    return dst


def colsum(x, out=None):
    """fp32 [N] = sum over the rows of a [M,N] / [B,R,N] view (bias gradients)."""
    _need_cuda(x)
    M, rx = rows_of(x)
    N = x.shape[-1]
    if out is None:
        out = torch.empty(N, device=x.device, dtype=torch.float32)
    libfk.check(libfk.load().fk_colsum_bf16(_ptr(x), rx, M, N, _ptr(out), _ptr(bwd_workspace(x.device)), _stream()),
                "fk_colsum_bf16")
    return out


# ---- FLUX AutoencoderKL pieces (NHWC bf16 activations) -------------------------------------------------
def conv2d_nhwc(x, w_packed, bias, cout, ksize=3, stride=1, pad=1, upsample2x=False, res=None, out=None):
    """x: [B,H,W,Cin] bf16; w_packed: [cout, Kpad] (k = (kh*ks+kw)*Cin + ci, zero padded to 64)."""
    _need_cuda(x, w_packed, bias, res)
    B, Hin, Win, Cin = x.shape
    He, We = (Hin * 2, Win * 2) if upsample2x else (Hin, Win)
    if stride == 1:
        Hout, Wout = He, We
    else:  # F.pad(x, (0,1,0,1)) + 3x3 stride-2 valid conv
        Hout, Wout = (He + 1 - 3) // 2 + 1, (We + 1 - 3) // 2 + 1
    if out is None:
        out = torch.empty((B, Hout, Wout, cout), device=x.device, dtype=BF16)
    a = libfk.ConvArgs()
    a.x, a.w, a.bias, a.y = x.data_ptr(), w_packed.data_ptr(), bias.data_ptr(), out.data_ptr()
    a.res = res.data_ptr() if res is not None else None
    a.B, a.Hin, a.Win, a.Cin, a.Cout = B, Hin, Win, Cin, cout
    a.ksize, a.stride, a.pad, a.upsample2x = ksize, stride, pad, int(upsample2x)
    a.Hout, a.Wout = Hout, Wout
    if not x.is_contiguous() or (res is not None and not res.is_contiguous()):
        raise ValueError("conv2d_nhwc needs contiguous NHWC tensors")
    libfk.check(libfk.load().fk_conv2d_nhwc_bf16(ctypes.byref(a), _stream()), "fk_conv2d_nhwc_bf16")
    return out


_GN_WS = {}


def _gn_workspace(lib, B, HW, C, device):
    """Partial-sum workspace + statistics of one GroupNorm shape, allocated once per (device, shape): calls on one
    stream are ordered, so consecutive GroupNorms may share it."""
    key = (str(device), B, HW, C)
    hit = _GN_WS.get(key)
    if hit is None:
        if len(_GN_WS) > 64:
            _GN_WS.clear()
        hit = (torch.empty(lib.fk_groupnorm_ws_floats(B, HW, C), device=device, dtype=torch.float32),
               torch.empty((B, 32, 2), device=device, dtype=torch.float32))
        _GN_WS[key] = hit
    return hit


def group_norm_stats(x, eps=1e-6):
    """(mean, rstd) fp32 [B, 32, 2] of GroupNorm(32) over [B, ..., C] NHWC bf16.  The buffer is shared by every call of
    one shape on the device: consume it (``group_norm_nhwc`` / ``conv3x3_halo(gn=...)``) before the next call."""
    _need_cuda(x)
    B, C = x.shape[0], x.shape[-1]
    HW = x.numel() // (B * C)
    lib = libfk.load()
    ws, stats = _gn_workspace(lib, B, HW, C, x.device)
    libfk.check(lib.fk_groupnorm_stats_nhwc_bf16(_ptr(x), _ptr(stats), _ptr(ws), B, HW, C, 32, eps, _stream()),
                "fk_groupnorm_stats_nhwc_bf16")
    return stats


def conv3x3_halo(x, w_packed, bias, cout, upsample2x=False, res=None, gn=None, out=None, out_fp32=False):
    """3 x 3 / stride 1 / pad 1 convolution over NHWC bf16 by the LDS halo-tiled kernel (csrc/conv_halo.hip), optionally
    with GroupNorm(32) (+ SiLU) of the INPUT applied while the halo tile is staged: ``gn = (stats, gamma, beta, silu)``
    with ``stats`` from :func:`group_norm_stats`.  Cin % 64 == 0."""
    _need_cuda(x, w_packed, bias, res)
    B, Hin, Win, Cin = x.shape
    Hout, Wout = (Hin * 2, Win * 2) if upsample2x else (Hin, Win)
    if out is None:
        out = torch.empty((B, Hout, Wout, cout), device=x.device, dtype=torch.float32 if out_fp32 else BF16)
    if out.dtype != (torch.float32 if out_fp32 else BF16) or not out.is_contiguous():
        raise TypeError("conv3x3_halo: out must be a contiguous bf16 tensor (fp32 with out_fp32: the parity build)")
    if not x.is_contiguous() or (res is not None and not res.is_contiguous()):
        raise ValueError("conv3x3_halo needs contiguous NHWC tensors")
    a = libfk.ConvArgs()
    a.x, a.w, a.bias, a.y = x.data_ptr(), w_packed.data_ptr(), bias.data_ptr(), out.data_ptr()
    a.res = res.data_ptr() if res is not None else None
    a.B, a.Hin, a.Win, a.Cin, a.Cout = B, Hin, Win, Cin, cout
    a.ksize, a.stride, a.pad, a.upsample2x = 3, 1, 1, int(upsample2x)
    a.Hout, a.Wout = Hout, Wout
    stats, gamma, beta, silu = gn if gn is not None else (None, None, None, False)
    _need_cuda(stats, gamma, beta)
    fn = libfk.load().fk_conv3x3_halo_f32_debug if out_fp32 else libfk.load().fk_conv3x3_halo_bf16
    libfk.check(fn(ctypes.byref(a), _ptr(stats), _ptr(gamma), _ptr(beta), 32, int(bool(silu)), _stream()), "fk_conv3x3_halo_bf16")
    return out


def group_norm_nhwc(x, gamma, beta, silu, eps=1e-6, out=None):
    """GroupNorm(32) (+SiLU) over [B, ..., C] NHWC bf16."""
    _need_cuda(x, gamma, beta)
    B, C = x.shape[0], x.shape[-1]
    HW = x.numel() // (B * C)
    lib = libfk.load()
    stats = group_norm_stats(x, eps)
    if out is None:
        out = torch.empty_like(x)
    libfk.check(lib.fk_groupnorm_apply_nhwc_bf16(_ptr(x), _ptr(out), _ptr(stats), _ptr(gamma), _ptr(beta), B, HW,
                                                 C, 32, int(silu), _stream()), "fk_groupnorm_apply_nhwc_bf16")
    return out


def nchw_to_nhwc(x, cpad, div=1.0, add=0.0):
    _need_cuda(x)
    B, C, H, W = x.shape
    if x.dtype not in (BF16, torch.float32) or not x.is_contiguous():
        raise TypeError("nchw_to_nhwc takes contiguous fp32/bf16 NCHW")
    out = torch.empty((B, H, W, cpad), device=x.device, dtype=BF16)
    libfk.check(libfk.load().fk_nchw_to_nhwc_bf16(_ptr(x), int(x.dtype == torch.float32), _ptr(out), B, C, cpad,
                                                  H, W, float(div), float(add), _stream()), "fk_nchw_to_nhwc_bf16")
    return out


def nhwc_to_nchw(x, c, add=0.0, mul=1.0, dtype=BF16):
    _need_cuda(x)
    B, H, W, cpad = x.shape
    out = torch.empty((B, c, H, W), device=x.device, dtype=dtype)
    libfk.check(libfk.load().fk_nhwc_to_nchw(_ptr(x), _ptr(out), int(dtype == torch.float32), B, c, cpad, H, W,
                                             float(add), float(mul), _stream()), "fk_nhwc_to_nchw")
    return out


def pixels_to_nhwc(u8, out_h, out_w, cpad=32, renorm=False):
    """uint8 NHWC pixels [B,Hin,Win,3] -> NHWC bf16 [B,out_h,out_w,cpad] in [-1,1]: cli.py:106-109 normalisation,
    VaeImageProcessor tensor resize (nearest) and the bf16 cast, one gather kernel."""
    _need_cuda(u8)
    if u8.dtype != torch.uint8 or u8.dim() != 4 or u8.shape[3] != 3 or not u8.is_contiguous():
        raise TypeError("pixels_to_nhwc takes a contiguous uint8 [B, H, W, 3] tensor")
    B, Hin, Win, _ = u8.shape
    out = torch.empty((B, out_h, out_w, cpad), device=u8.device, dtype=BF16)
    libfk.check(libfk.load().fk_pixels_u8_to_nhwc_bf16(_ptr(u8), _ptr(out), B, Hin, Win, out_h, out_w, cpad,
                                                       int(bool(renorm)), _stream()), "fk_pixels_u8_to_nhwc_bf16")
    return out


def image_to_u8(img):
    """decoder output [B,C,H,W] (bf16/fp32, [-1,1]) -> uint8 [B,H,W,C] (VaeImageProcessor.postprocess, 'pil' array)."""
    _need_cuda(img)
    if img.dtype not in (BF16, torch.float32) or img.dim() != 4 or not img.is_contiguous():
        raise TypeError("image_to_u8 takes a contiguous fp32/bf16 NCHW tensor")
    B, C, H, W = img.shape
    out = torch.empty((B, H, W, C), device=img.device, dtype=torch.uint8)
    libfk.check(libfk.load().fk_image_to_u8_nhwc(_ptr(img), int(img.dtype == torch.float32), _ptr(out), B, C, H, W,
                                                 _stream()), "fk_image_to_u8_nhwc")
    return out


# ---- optimisation step of the denoiser (include/fk.h; reference train_denoiser.py:935-1181) ------------------------
def _reduce_ws(device):
    return torch.empty(libfk.load().fk_reduce_ws_doubles(), dtype=torch.float64, device=device)


def _f32c(t, what):
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise TypeError(f"{what} must be a contiguous fp32 tensor")


def flow_noisy_tokens(x, noise, sigma, out=None):
    """Packed bf16 tokens of (1 - sigma_b) * x + sigma_b * noise; x, noise fp32 [B,C,h,w], sigma fp32 [B].

    ``out``: a [B, (h/2)(w/2), 4C] bf16 view with contiguous samples (e.g. the target half of the token buffer)."""
    _need_cuda(x, noise, sigma, out)
    _f32c(x, "x"); _f32c(noise, "noise"); _f32c(sigma, "sigma")
    B, C, h, w = x.shape
    if out is None:
        out = torch.empty(B, (h // 2) * (w // 2), 4 * C, device=x.device, dtype=BF16)
    if out.shape != (B, (h // 2) * (w // 2), 4 * C) or out.dtype != BF16 or out.stride(2) != 1 or out.stride(1) != 4 * C:
        raise ValueError("out must be a [B, S, 4C] bf16 view with contiguous samples")
    libfk.check(libfk.load().fk_flow_noisy_tokens_bf16(_ptr(x), _ptr(noise), _ptr(sigma), _ptr(out), out.stride(0), B, C,
                                                       h, w, _stream()), "fk_flow_noisy_tokens_bf16")
    return out


def flow_loss(pred, x, noise, weight=None, want_grad=True, area_mask_weights=None, weight_mask=None):
    """(loss fp64 [1], grad bf16 like pred or None) of the reference's flow-matching loss (train_denoiser.py:1106-1166);
    pred: packed bf16 [B, S, 4C] view with contiguous samples.  weighting = weight[b] * area_mask_weights[b, 0, y, x] *
    weight_mask[b, 0, y, x] (fp32 [B] / [B, 1, h, w], each optional); the sum of weighting * (unpack(pred) - (noise - x))^2 is
    divided by weight_mask.sum() * C when weight_mask is given (:1163-1165), by the element count otherwise (loss.mean())."""
    _need_cuda(pred, x, noise, weight, area_mask_weights, weight_mask)
    _f32c(x, "x"); _f32c(noise, "noise")
    B, C, h, w = x.shape
    if pred.shape != (B, (h // 2) * (w // 2), 4 * C) or pred.dtype != BF16 or pred.stride(2) != 1 or pred.stride(1) != 4 * C:
        raise ValueError("pred must be a [B, S, 4C] bf16 view with contiguous samples")
    if weight is not None:
        _f32c(weight, "weight")
        if weight.numel() != B:
            raise ValueError("weight: one fp32 value per sample")
    for name, m in (("area_mask_weights", area_mask_weights), ("weight_mask", weight_mask)):
        if m is not None:
            _f32c(m, name)
            if tuple(m.shape) != (B, 1, h, w):
                raise ValueError(f"{name} must be [B, 1, h, w] at the latent size (nearest-resize it as train_denoiser.py:1131-1148 does)")
    mask_sum = weight_mask.sum().reshape(1) if weight_mask is not None else None     # device scalar: no host round trip
    grad = torch.empty(pred.shape, device=pred.device, dtype=BF16) if want_grad else None
    loss = torch.empty(1, dtype=torch.float64, device=pred.device)
    libfk.check(libfk.load().fk_flow_loss_weighted_bf16(
        _ptr(pred), pred.stride(0), _ptr(x), _ptr(noise), _ptr(weight), _ptr(area_mask_weights), _ptr(weight_mask), _ptr(mask_sum),
        _ptr(grad), grad.stride(0) if want_grad else 0, _ptr(loss), _ptr(_reduce_ws(pred.device)), B, C, h, w, _stream()),
        "fk_flow_loss_weighted_bf16")
    return loss, grad


def sumsq(tensors, out=None):
    """fp64 [1]: sum of squares over a list of contiguous fp32 / bf16 tensors (the squared global gradient norm)."""
    tensors = [tensors] if isinstance(tensors, torch.Tensor) else list(tensors)
    _need_cuda(*tensors)
    dev = tensors[0].device
    if out is None:
        out = torch.zeros(1, dtype=torch.float64, device=dev)
    ws = _reduce_ws(dev)
    lib = libfk.load()
    for i, t in enumerate(tensors):
        if t.dtype not in (torch.float32, BF16) or not t.is_contiguous():
            raise TypeError("sumsq takes contiguous fp32 / bf16 tensors")
        libfk.check(lib.fk_sumsq(_ptr(t), int(t.dtype == BF16), t.numel(), int(i > 0), _ptr(out), _ptr(ws), _stream()),
                    "fk_sumsq")
    return out


def adamw_step(master, grad, exp_avg, exp_avg_sq, step, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2,
               grad_sumsq=None, max_grad_norm=1.0, param_bf16=None, grad_scale=1.0):
    """In-place AdamW update of the fp32 ``master`` (and the bf16 copy); ``grad_sumsq`` (fp64 [1]) enables clipping.
    ``grad_scale``: ``grad`` holds sums that still have to be multiplied by it (1 / world after a reduce-scatter)."""
    _need_cuda(master, grad, exp_avg, exp_avg_sq, grad_sumsq, param_bf16)
    for t, what in ((master, "master"), (exp_avg, "exp_avg"), (exp_avg_sq, "exp_avg_sq")):
        _f32c(t, what)
    if grad.dtype not in (torch.float32, BF16) or not grad.is_contiguous() or grad.numel() != master.numel():
        raise TypeError("grad must be a contiguous fp32 / bf16 tensor of the parameter's size")
    libfk.check(libfk.load().fk_adamw_step_scaled(_ptr(master), _ptr(param_bf16), _ptr(grad), int(grad.dtype == BF16),
                                                  _ptr(exp_avg), _ptr(exp_avg_sq), _ptr(grad_sumsq), float(max_grad_norm),
                                                  float(grad_scale), float(lr), float(betas[0]), float(betas[1]), float(eps),
                                                  float(weight_decay), int(step), master.numel(), _stream()), "fk_adamw_step")
    if param_bf16 is not None:
        param_tree.note_raw_write()      # a parameter changed behind torch's version counters
    return master
