"""ctypes binding of libfk.so (C ABI declared in include/fk.h).

There is deliberately NO fallback: if the HIP library is missing or does not export a symbol the
import fails loudly, and every op raises ``RuntimeError`` carrying ``fk_last_error()`` on a non-zero
return code.  PyTorch is used only for device memory and streams (tensor.data_ptr(), current stream).
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# not "libfk.so": python would import it as module `libfk`.  FK_LIB_PATH: A/B-test another build of the library.
LIB_PATH = os.environ.get("FK_LIB_PATH") or os.path.join(_HERE, "libfk_gfx950.so")

c_i32, c_i64, c_f32, c_vp = ctypes.c_int32, ctypes.c_int64, ctypes.c_float, ctypes.c_void_p

FK_SPLITK_SLOT_BYTES = 256 * 256 * 4 + 8   # include/fk.h: one fp32 partial tile + its (ticket, flag) words
FK_EPI_NONE, FK_EPI_GELU_TANH, FK_EPI_SILU, FK_EPI_GATE_RES, FK_EPI_RES, FK_EPI_SCALE, FK_EPI_QKV = range(7)


class Rows(ctypes.Structure):
    _fields_ = [("ld", c_i64), ("rows_per_batch", c_i64), ("batch_stride", c_i64)]


class GemmArgs(ctypes.Structure):
    _fields_ = [
        ("A", c_vp), ("a", Rows),
        ("W", c_vp), ("ldw", c_i64),
        ("bias", c_vp),
        ("C", c_vp), ("c", Rows),
        ("res", c_vp), ("r", Rows),
        ("gate", c_vp), ("gate_batch_stride", c_i64), ("gate_rows_per_batch", c_i64),
        ("M", c_i32), ("N", c_i32), ("K", c_i32),
        ("epilogue", c_i32), ("out_fp32", c_i32), ("alpha", c_f32),
        ("q_out", c_vp), ("k_out", c_vp), ("wq", c_vp), ("wk", c_vp), ("rope_cs", c_vp), ("splitk_ws", c_vp),
        ("qkv_s_offset", c_i32), ("qkv_s_total", c_i32), ("qkv_heads", c_i32), ("splitk_slots", c_i32), ("layout", c_i32), ("f32_flags", c_i32),
        ("variant", c_i32), ("plan", c_i32), ("group_m", c_i32), ("mfma", c_i32),       # per-call launch controls (include/fk.h)
        ("variant_used", ctypes.POINTER(c_i32)),                                         # OUT, optional: the launch form used
    ]


class AttnView(ctypes.Structure):
    _fields_ = [("p", c_vp), ("ld", c_i64), ("head_stride", c_i64), ("batch_stride", c_i64)]


class ConvArgs(ctypes.Structure):
    _fields_ = [
        ("x", c_vp), ("w", c_vp), ("bias", c_vp), ("y", c_vp), ("res", c_vp),
        ("B", c_i32), ("Hin", c_i32), ("Win", c_i32), ("Cin", c_i32), ("Cout", c_i32),
        ("ksize", c_i32), ("stride", c_i32), ("pad", c_i32), ("upsample2x", c_i32),
        ("Hout", c_i32), ("Wout", c_i32),
    ]


class BlockWs(ctypes.Structure):          # fk_block_ws
    _fields_ = [(n, c_vp) for n in ("s", "n", "qkv", "q", "k", "o", "ff", "cat", "rope_cs", "splitk_ws", "attn_ws")] + [
        ("attn_ws_bytes", c_i64), ("splitk_slots", c_i32), ("B", c_i32), ("S_txt", c_i32), ("S_img", c_i32), ("H", c_i32),
        ("eps", c_f32), ("gemm_variant", c_i32), ("gemm_plan", c_i32), ("gemm_group_m", c_i32), ("gemm_mfma", c_i32), ("attn_grid", c_i32),
        ("gemm_variant_used", ctypes.POINTER(c_i32))]


class BwdWs(ctypes.Structure):            # fk_bwd_ws
    _fields_ = [("B", c_i32), ("S_txt", c_i32), ("S_img", c_i32), ("H", c_i32), ("eps", c_f32), ("splitk_slots", c_i32)] + [
        (n, c_vp) for n in ("g", "dy", "dff", "dn", "d_o", "dqkv", "dq", "dk", "dsum", "dmod", "ff", "cat", "cos", "sin", "red_ws", "attn_ws")] + [
        ("attn_ws_bytes", c_i64), ("splitk_ws", c_vp), ("mod", c_vp), ("mod_batch_stride", c_i64), ("actT", c_vp), ("onesT", c_vp),
        ("dmodT", c_vp), ("gemm_variant", c_i32), ("gemm_plan", c_i32), ("gemm_group_m", c_i32), ("gemm_mfma", c_i32), ("attn_grid", c_i32),
        ("attn_passes", c_i32), ("gemm_variant_used", ctypes.POINTER(c_i32))]


class BlockSaved(ctypes.Structure):       # fk_block_saved
    _fields_ = [(n, c_vp) for n in ("x0", "n1", "qkv", "q", "k", "y1", "h1", "o", "lse", "x1", "n2", "y2")]


SINGLE_GRAD_FIELDS = ("dwqkv", "dbqkv", "dw_mlp", "db_mlp", "dw_out", "db_out", "dnorm", "dw_mod", "db_mod")
DOUBLE_GRAD_FIELDS = ("dwqkv_img", "dbqkv_img", "dwqkv_txt", "dbqkv_txt", "dw_out", "db_out", "dw_add_out", "db_add_out", "dw_ff1", "db_ff1",
                      "dw_ff1_ctx", "db_ff1_ctx", "dw_ff2", "db_ff2", "dw_ff2_ctx", "db_ff2_ctx", "dnorm", "dw_mod_img", "db_mod_img",
                      "dw_mod_txt", "db_mod_txt")


class SingleBlockGrads(ctypes.Structure):  # fk_single_block_grads
    _fields_ = [(n, c_vp) for n in SINGLE_GRAD_FIELDS]


class DoubleBlockGrads(ctypes.Structure):  # fk_double_block_grads
    _fields_ = [(n, c_vp) for n in DOUBLE_GRAD_FIELDS]


DOUBLE_BLOCK_FIELDS = ("wqkv_img", "bqkv_img", "wqkv_txt", "bqkv_txt", "norm_q", "norm_k", "norm_added_q", "norm_added_k",
                       "w_out", "b_out", "w_add_out", "b_add_out", "w_ff1", "b_ff1", "w_ff1_ctx", "b_ff1_ctx",
                       "w_ff2", "b_ff2", "w_ff2_ctx", "b_ff2_ctx")
SINGLE_BLOCK_FIELDS = ("wqkv", "bqkv", "norm_q", "norm_k", "w_mlp", "b_mlp", "w_out", "b_out")


class DoubleBlockWeights(ctypes.Structure):   # fk_double_block_weights
    _fields_ = [(n, c_vp) for n in DOUBLE_BLOCK_FIELDS] + [("mod_off_img", c_i64), ("mod_off_txt", c_i64)]


class SingleBlockWeights(ctypes.Structure):   # fk_single_block_weights
    _fields_ = [(n, c_vp) for n in SINGLE_BLOCK_FIELDS] + [("mod_off", c_i64)]


# symbol -> (restype, argtypes); must list every entry point of include/fk.h
SIGNATURES = {
    "fk_gemm_bf16": (c_i32, [ctypes.POINTER(GemmArgs), c_vp]),
    "fk_gemm_bf16_grouped": (c_i32, [ctypes.POINTER(GemmArgs), c_i32, c_vp]),
    "fk_ln_modulate_bf16": (c_i32, [c_vp, Rows, c_vp, Rows, c_vp, c_vp, c_i64, c_i64, c_i64, c_i32, c_f32, c_vp]),
    "fk_ln_modulate2_bf16": (c_i32, [c_vp, Rows, c_vp, Rows, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_i32, c_f32, c_vp]),
    "fk_qkv_post_bf16": (c_i32, [c_vp] * 9 + [c_i32] * 4 + [c_f32, c_vp]),
    "fk_attention_fwd_bf16": (c_i32, [c_vp] * 4 + [c_i32] * 3 + [c_i64] * 4 + [c_f32, c_vp]),
    "fk_attention_fwd_ws_bf16": (c_i32, [c_vp] * 5 + [c_i32] * 3 + [c_i64] * 4 + [c_f32, c_vp, c_i64, c_i32, c_vp]),
    "fk_attention_ws_bytes": (c_i64, []),
    "fk_attention_fwd_f32_debug": (c_i32, [c_vp] * 4 + [c_i32] * 3 + [c_i64] * 4 + [c_f32, c_vp]),
    "fk_attention_fwd_lse_bf16": (c_i32, [c_vp] * 5 + [c_i32] * 3 + [c_i64] * 4 + [c_f32, c_vp]),
    "fk_attention_bwd_bf16": (c_i32, [ctypes.POINTER(AttnView)] * 4 + [c_vp, c_vp] + [ctypes.POINTER(AttnView)] * 3 + [c_i32] * 3 + [c_f32, c_vp]),
    "fk_attention_bwd_ws_bf16": (c_i32, [ctypes.POINTER(AttnView)] * 4 + [c_vp, c_vp] + [ctypes.POINTER(AttnView)] * 3 + [c_i32] * 3 + [c_f32, c_vp, c_i64, c_i32, c_i32, c_vp]),
    "fk_bwd_ws_floats": (c_i64, []),
    "fk_ln_modulate_bwd_bf16": (c_i32, [c_vp, Rows, c_vp, Rows, c_vp, c_i64, c_i64, c_vp, Rows, c_vp, Rows, c_vp, c_vp, c_i64, c_vp, c_i32, c_i32, c_f32, c_vp]),
    "fk_gate_res_bwd_bf16": (c_i32, [c_vp, Rows, c_vp, Rows, c_vp, c_i64, c_i64, c_vp, Rows, c_vp, c_i64, c_vp, c_i32, c_i32, c_vp]),
    "fk_gelu_bwd_bf16": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_vp]),
    "fk_silu_bwd_bf16": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_vp]),
    "fk_qkv_post_bwd_bf16": (c_i32, [c_vp] * 12 + [c_i32] * 4 + [c_f32, c_vp]),
    "fk_gate_res_fwd_bf16": (c_i32, [c_vp, Rows, c_vp, Rows, c_vp, c_i64, c_i64, c_vp, Rows, c_i64, c_i32, c_vp]),
    "fk_gelu_tanh_bf16": (c_i32, [c_vp, Rows, c_vp, Rows, c_i64, c_i32, c_vp]),
    "fk_f32_to_bf16_transposed": (c_i32, [c_vp, c_i64, c_vp, c_i32, c_i32, c_i32, c_vp]),
    "fk_colsum_bf16": (c_i32, [c_vp, Rows, c_i64, c_i32, c_vp, c_vp, c_vp]),
    "fk_rowdot_bf16": (c_i32, [c_vp, c_i64, c_i64, c_vp, c_i64, c_i64, c_vp, c_i32, c_i32, c_i32, c_vp]),
    "fk_silu_bf16": (c_i32, [c_vp, c_vp, c_i64, c_vp]),
    "fk_timestep_proj": (c_i32, [c_vp, c_i32, c_vp, c_vp, c_i32, c_vp]),
    "fk_add3_bf16": (c_i32, [c_vp] * 4 + [c_i64, c_vp]),
    "fk_true_cfg_bf16": (c_i32, [c_vp, c_vp, c_vp, c_f32, c_i64, c_vp]),
    "fk_reduce_ws_doubles": (c_i64, []),
    "fk_flow_noisy_tokens_bf16": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i64] + [c_i32] * 4 + [c_vp]),
    "fk_flow_loss_bf16": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp] + [c_i32] * 4 + [c_vp]),
    "fk_flow_loss_weighted_bf16": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp] + [c_i32] * 4 + [c_vp]),
    "fk_sumsq": (c_i32, [c_vp, c_i32, c_i64, c_i32, c_vp, c_vp, c_vp]),
    "fk_adamw_step": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_vp, c_vp, c_vp] + [c_f32] * 6 + [c_i32, c_i64, c_vp]),
    "fk_adamw_step_scaled": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_vp, c_vp, c_vp] + [c_f32] * 7 + [c_i32, c_i64, c_vp]),
    "fk_euler_step_bf16": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_i32, c_i32, c_i32, c_f32, c_vp]),
    "fk_transpose_bf16": (c_i32, [c_vp, c_i64, c_i64, c_vp, c_i64, c_i64, c_i32, c_i32, c_i32, c_vp]),
    "fk_attention_hd512_bf16": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_i64, c_vp, c_i64, c_i64, c_i32, c_i32, c_f32, c_vp]),
    "fk_softmax_rows": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_i64, c_i32, c_vp]),
    "fk_softmax_rows_parts": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_i64, c_i64, c_i32, c_vp]),
    "fk_split_f32_rows": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_i64, c_i64, c_i32, c_i32, c_i32, c_vp]),
    "fk_groupnorm_f32_nhwc": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i64, c_i32, c_i32, c_f32, c_i32, c_i32, c_vp]),
    "fk_conv2d_nhwc_f32out": (c_i32, [c_vp, c_vp]),
    "fk_conv3x3_halo_f32out": (c_i32, [c_vp, c_vp]),
    "fk_nchw_f32_to_nhwc_parts": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "fk_nhwc_f32_to_nchw": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_f32, c_f32, c_vp]),
    "fk_conv2d_nhwc_bf16": (c_i32, [ctypes.POINTER(ConvArgs), c_vp]),
    "fk_conv3x3_halo_bf16": (c_i32, [ctypes.POINTER(ConvArgs), c_vp, c_vp, c_vp, c_i32, c_i32, c_vp]),
    "fk_conv3x3_halo_f32_debug": (c_i32, [ctypes.POINTER(ConvArgs), c_vp, c_vp, c_vp, c_i32, c_i32, c_vp]),
    "fk_groupnorm_ws_floats": (c_i64, [c_i32, c_i64, c_i32]),
    "fk_groupnorm_stats_nhwc_bf16": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i64, c_i32, c_i32, c_f32, c_vp]),
    "fk_groupnorm_apply_nhwc_bf16": (c_i32, [c_vp] * 5 + [c_i32, c_i64, c_i32, c_i32, c_i32, c_vp]),
    "fk_nchw_to_nhwc_bf16": (c_i32, [c_vp, c_i32, c_vp] + [c_i32] * 5 + [c_f32, c_f32, c_vp]),
    "fk_nhwc_to_nchw": (c_i32, [c_vp, c_vp] + [c_i32] * 6 + [c_f32, c_f32, c_vp]),
    "fk_pixels_u8_to_nhwc_bf16": (c_i32, [c_vp, c_vp] + [c_i32] * 7 + [c_vp]),
    "fk_image_to_u8_nhwc": (c_i32, [c_vp, c_i32, c_vp] + [c_i32] * 4 + [c_vp]),
    "fk_double_block_fwd": (c_i32, [ctypes.POINTER(BlockWs), ctypes.POINTER(DoubleBlockWeights), c_vp, c_i64, c_vp]),
    "fk_single_block_fwd": (c_i32, [ctypes.POINTER(BlockWs), ctypes.POINTER(SingleBlockWeights), c_vp, c_i64, c_vp]),
    "fk_single_block_bwd": (c_i32, [ctypes.POINTER(BwdWs), ctypes.POINTER(BlockSaved), ctypes.POINTER(SingleBlockWeights),
                                    ctypes.POINTER(SingleBlockGrads), c_vp]),
    "fk_double_block_bwd": (c_i32, [ctypes.POINTER(BwdWs), ctypes.POINTER(BlockSaved), ctypes.POINTER(DoubleBlockWeights),
                                    ctypes.POINTER(DoubleBlockGrads), c_vp]),
    "fk_mmdit_blocks_fwd": (c_i32, [ctypes.POINTER(BlockWs), ctypes.POINTER(DoubleBlockWeights), c_i32,
                                    ctypes.POINTER(SingleBlockWeights), c_i32, c_vp, c_i64, c_vp]),
    "fk_last_error": (ctypes.c_char_p, []),
    "fk_version": (ctypes.c_char_p, []),
}

_lib = None


def load():
    """Load libfk.so once; raises (never falls back) when it is absent or incomplete."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run `python -c 'import "
            f"__graft_entry__ as g; g.build()'` (or `make -C gpt_image_edit_amd/csrc`). There is no CPU "
            f"fallback for the FLUX-Kontext hot path.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(code, what):
    if code != 0:
        msg = load().fk_last_error().decode(errors="replace")
        raise RuntimeError(f"{what} failed (code {code}): {msg}")
