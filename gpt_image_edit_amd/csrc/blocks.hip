// Block-level entry points of the C ABI (SURVEY.md section 8b lists `fk_double_block_fwd`, `fk_single_block_fwd`,
// `fk_mmdit_fwd` beside the per-kernel calls): ONE call enqueues every launch of a FluxTransformerBlock /
// FluxSingleTransformerBlock (diffusers 0.32.2, reached from reference flux_pipeline.py:1067-1077), or of all the blocks of a
// forward, on the caller's stream.  Host code only: it fills the argument structs of the per-kernel entry points exactly as
// the Python adaptor does (gpt_image_edit_amd/ops.py: `rows_of` addressing of the text / image slices of the joint buffers,
// split-K and stream-K workspaces) and calls them in the adaptor's order, so the launches -- and therefore the results --
// are the same bit for bit (tests/test_hip_mmdit.py::test_block_entry_points_give_the_same_bits).  What it removes is host
// work: ~5 400 ctypes calls per edit become 57 x 28 (per-block form) or 28 (whole-stack form).
#include "fk_common.h"

namespace {

constexpr int HD = 128;

struct Dims {
  int B, S_txt, S_img, S, H, D;
};

// a [B, R, cols] slice (rows [r0, r0 + R) of every batch, columns from c0) of a joint [B, S, ld] buffer
struct View {
  const void* p;
  fk_rows r;
};
View view(const void* base, const Dims& d, int64_t ld, int64_t r0, int64_t R, int64_t c0) {
  return View{(const char*)base + (r0 * ld + c0) * 2, fk_rows{ld, R, (int64_t)d.S * ld}};
}

void set_ws(fk_gemm_args& g, const fk_block_ws& ws) {   // ops._gemm_args: only the shapes the planner may split get the workspace
  if (g.K >= 6144 && g.N % 256 == 0 && (int64_t)g.M <= 256ll * ws.splitk_slots && ws.splitk_ws) {
    g.splitk_ws = ws.splitk_ws;
    g.splitk_slots = ws.splitk_slots;
  }
}

// launch controls of the block's GEMMs (fk_block_ws.gemm_*): a grouped launch reads them from its first problem
void ctl(fk_gemm_args& g, const fk_block_ws& ws) {
  g.variant = ws.gemm_variant; g.plan = ws.gemm_plan; g.group_m = ws.gemm_group_m; g.mfma = ws.gemm_mfma;
  g.variant_used = ws.gemm_variant_used;
}

fk_gemm_args gemm(const View& a, const void* w, const void* bias, const View& c, int M, int N, int K, int epi) {
  fk_gemm_args g = {};
  g.A = a.p; g.a = a.r;
  g.W = w; g.ldw = K;
  g.bias = bias;
  g.C = (void*)c.p; g.c = c.r;
  g.M = M; g.N = N; g.K = K;
  g.epilogue = epi;
  g.alpha = 1.0f;
  return g;
}
void gate_res(fk_gemm_args& g, const View& res, const void* gate, int64_t gate_bs, int64_t rows_per_batch) {
  g.res = res.p; g.r = res.r;
  g.gate = gate; g.gate_batch_stride = gate_bs; g.gate_rows_per_batch = rows_per_batch;
}
void qkv_epi(fk_gemm_args& g, const fk_block_ws& ws, const Dims& d, const void* wq, const void* wk, int s_offset) {
  g.q_out = ws.q; g.k_out = ws.k; g.wq = wq; g.wk = wk; g.rope_cs = ws.rope_cs;
  g.qkv_s_offset = s_offset; g.qkv_s_total = d.S; g.qkv_heads = d.H;
}

int check_ws(const fk_block_ws* ws, Dims& d, const char* who) {
  FK_CHECK_ARG(ws != nullptr, "%s: null workspace", who);
  FK_CHECK_ARG(ws->B > 0 && ws->S_txt >= 0 && ws->S_img > 0 && ws->H > 0, "%s: bad B / S_txt / S_img / H %d %d %d %d", who, ws->B,
               ws->S_txt, ws->S_img, ws->H);
  FK_CHECK_ARG(ws->s && ws->n && ws->qkv && ws->q && ws->k && ws->rope_cs, "%s: null activation buffer", who);
  d = Dims{ws->B, ws->S_txt, ws->S_img, ws->S_txt + ws->S_img, ws->H, ws->H * HD};
  FK_CHECK_ARG((int64_t)d.B * d.S < (1ll << 31), "%s: B * S too large", who);
  return FK_OK;
}

#define FK_TRY(expr)          \
  do {                        \
    const int rc_ = (expr);   \
    if (rc_ != FK_OK) return rc_; \
  } while (0)

int double_block(const fk_block_ws& ws, const Dims& d, const fk_double_block_weights& w, const void* mod, int64_t mod_bs,
                 fk_stream_t st) {
  FK_CHECK_ARG(ws.o && ws.ff && d.S_txt > 0, "fk_double_block_fwd: needs the o / ff buffers and a text stream");
  const int D = d.D, B = d.B, Mi = B * d.S_img, Mt = B * d.S_txt;
  const char* mi = (const char*)mod + w.mod_off_img * 2;     // shift, scale, gate, shift_mlp, scale_mlp, gate_mlp: D each
  const char* mt = (const char*)mod + w.mod_off_txt * 2;
  auto chunk = [&](const char* m, int j) { return (const void*)(m + (int64_t)j * D * 2); };
  const View s_all = view(ws.s, d, D, 0, d.S, 0), n_all = view(ws.n, d, D, 0, d.S, 0);
  const View h = view(ws.s, d, D, d.S_txt, d.S_img, 0), cx = view(ws.s, d, D, 0, d.S_txt, 0);
  const View n_img = view(ws.n, d, D, d.S_txt, d.S_img, 0), n_txt = view(ws.n, d, D, 0, d.S_txt, 0);
  const int64_t M = (int64_t)B * d.S;
  // text + image streams share every launch: joint LN + modulate, grouped GEMMs (one grid, two weights)
  FK_TRY(fk_ln_modulate2_bf16(s_all.p, s_all.r, (void*)n_all.p, n_all.r, chunk(mt, 0), chunk(mt, 1), chunk(mi, 0), chunk(mi, 1),
                              d.S_txt, mod_bs, d.S, M, D, ws.eps, st));
  {
    fk_gemm_args g[2];
    g[0] = gemm(n_img, w.wqkv_img, w.bqkv_img, view(ws.qkv, d, 3 * D, d.S_txt, d.S_img, 0), Mi, 3 * D, D, FK_EPI_QKV);
    qkv_epi(g[0], ws, d, w.norm_q, w.norm_k, d.S_txt);
    g[1] = gemm(n_txt, w.wqkv_txt, w.bqkv_txt, view(ws.qkv, d, 3 * D, 0, d.S_txt, 0), Mt, 3 * D, D, FK_EPI_QKV);
    qkv_epi(g[1], ws, d, w.norm_added_q, w.norm_added_k, 0);
    ctl(g[0], ws); FK_TRY(fk_gemm_bf16_grouped(g, 2, st));
  }
  FK_TRY(fk_attention_fwd_ws_bf16(ws.q, ws.k, (const char*)ws.qkv + (int64_t)2 * D * 2, ws.o, nullptr, B, d.H, d.S, 3 * D,
                                  (int64_t)d.S * 3 * D, D, (int64_t)d.S * D, 0.08838834764831845f, ws.attn_ws, ws.attn_ws_bytes, ws.attn_grid, st));
  {
    fk_gemm_args g[2];
    g[0] = gemm(view(ws.o, d, D, d.S_txt, d.S_img, 0), w.w_out, w.b_out, h, Mi, D, D, FK_EPI_GATE_RES);
    gate_res(g[0], h, chunk(mi, 2), mod_bs, d.S_img);
    g[1] = gemm(view(ws.o, d, D, 0, d.S_txt, 0), w.w_add_out, w.b_add_out, cx, Mt, D, D, FK_EPI_GATE_RES);
    gate_res(g[1], cx, chunk(mt, 2), mod_bs, d.S_txt);
    set_ws(g[0], ws); set_ws(g[1], ws);
    ctl(g[0], ws); FK_TRY(fk_gemm_bf16_grouped(g, 2, st));
  }
  FK_TRY(fk_ln_modulate2_bf16(s_all.p, s_all.r, (void*)n_all.p, n_all.r, chunk(mt, 3), chunk(mt, 4), chunk(mi, 3), chunk(mi, 4),
                              d.S_txt, mod_bs, d.S, M, D, ws.eps, st));
  {
    fk_gemm_args g[2];
    g[0] = gemm(n_img, w.w_ff1, w.b_ff1, view(ws.ff, d, 4 * D, d.S_txt, d.S_img, 0), Mi, 4 * D, D, FK_EPI_GELU_TANH);
    g[1] = gemm(n_txt, w.w_ff1_ctx, w.b_ff1_ctx, view(ws.ff, d, 4 * D, 0, d.S_txt, 0), Mt, 4 * D, D, FK_EPI_GELU_TANH);
    ctl(g[0], ws); FK_TRY(fk_gemm_bf16_grouped(g, 2, st));
  }
  {
    fk_gemm_args g[2];
    g[0] = gemm(view(ws.ff, d, 4 * D, d.S_txt, d.S_img, 0), w.w_ff2, w.b_ff2, h, Mi, D, 4 * D, FK_EPI_GATE_RES);
    gate_res(g[0], h, chunk(mi, 5), mod_bs, d.S_img);
    g[1] = gemm(view(ws.ff, d, 4 * D, 0, d.S_txt, 0), w.w_ff2_ctx, w.b_ff2_ctx, cx, Mt, D, 4 * D, FK_EPI_GATE_RES);
    gate_res(g[1], cx, chunk(mt, 5), mod_bs, d.S_txt);
    set_ws(g[0], ws); set_ws(g[1], ws);
    ctl(g[0], ws); FK_TRY(fk_gemm_bf16_grouped(g, 2, st));
  }
  return FK_OK;
}

int single_block(const fk_block_ws& ws, const Dims& d, const fk_single_block_weights& w, const void* mod, int64_t mod_bs,
                 fk_stream_t st) {
  FK_CHECK_ARG(ws.cat != nullptr, "fk_single_block_fwd: needs the [attn | mlp] buffer");
  const int D = d.D, B = d.B;
  const int Ms = B * d.S;
  const char* m0 = (const char*)mod + w.mod_off * 2;     // shift, scale, gate
  auto chunk = [&](int j) { return (const void*)(m0 + (int64_t)j * D * 2); };
  const View s_all = view(ws.s, d, D, 0, d.S, 0), n_all = view(ws.n, d, D, 0, d.S, 0);
  FK_TRY(fk_ln_modulate_bf16(s_all.p, s_all.r, (void*)n_all.p, n_all.r, chunk(0), chunk(1), mod_bs, d.S, (int64_t)Ms, D, ws.eps, st));
  {
    fk_gemm_args g = gemm(n_all, w.wqkv, w.bqkv, view(ws.qkv, d, 3 * D, 0, d.S, 0), Ms, 3 * D, D, FK_EPI_QKV);
    qkv_epi(g, ws, d, w.norm_q, w.norm_k, 0);
    ctl(g, ws); FK_TRY(fk_gemm_bf16(&g, st));
  }
  // attention writes columns [0, D) of the [B, S, 5D] buffer, the MLP-up GEMM columns [D, 5D): proj_out reads one operand
  FK_TRY(fk_attention_fwd_ws_bf16(ws.q, ws.k, (const char*)ws.qkv + (int64_t)2 * D * 2, ws.cat, nullptr, B, d.H, d.S, 3 * D,
                                  (int64_t)d.S * 3 * D, 5 * D, (int64_t)d.S * 5 * D, 0.08838834764831845f, ws.attn_ws,
                                  ws.attn_ws_bytes, ws.attn_grid, st));
  {
    fk_gemm_args g = gemm(n_all, w.w_mlp, w.b_mlp, view(ws.cat, d, 5 * D, 0, d.S, D), Ms, 4 * D, D, FK_EPI_GELU_TANH);
    ctl(g, ws); FK_TRY(fk_gemm_bf16(&g, st));
  }
  {
    fk_gemm_args g = gemm(view(ws.cat, d, 5 * D, 0, d.S, 0), w.w_out, w.b_out, s_all, Ms, D, 5 * D, FK_EPI_GATE_RES);
    gate_res(g, s_all, chunk(2), mod_bs, d.S);
    set_ws(g, ws);
    ctl(g, ws); FK_TRY(fk_gemm_bf16(&g, st));
  }
  return FK_OK;
}

}  // namespace

extern "C" int fk_double_block_fwd(const fk_block_ws* ws, const fk_double_block_weights* w, const void* mod,
                                   int64_t mod_batch_stride, fk_stream_t stream) {
  Dims d;
  FK_TRY(check_ws(ws, d, "fk_double_block_fwd"));
  FK_CHECK_ARG(w && mod, "fk_double_block_fwd: null weights / modulation");
  return double_block(*ws, d, *w, mod, mod_batch_stride, stream);
}

extern "C" int fk_single_block_fwd(const fk_block_ws* ws, const fk_single_block_weights* w, const void* mod,
                                   int64_t mod_batch_stride, fk_stream_t stream) {
  Dims d;
  FK_TRY(check_ws(ws, d, "fk_single_block_fwd"));
  FK_CHECK_ARG(w && mod, "fk_single_block_fwd: null weights / modulation");
  return single_block(*ws, d, *w, mod, mod_batch_stride, stream);
}

extern "C" int fk_mmdit_blocks_fwd(const fk_block_ws* ws, const fk_double_block_weights* dbl, int32_t n_double,
                                   const fk_single_block_weights* sgl, int32_t n_single, const void* mod,
                                   int64_t mod_batch_stride, fk_stream_t stream) {
  Dims d;
  FK_TRY(check_ws(ws, d, "fk_mmdit_blocks_fwd"));
  FK_CHECK_ARG(mod && n_double >= 0 && n_single >= 0 && (n_double == 0 || dbl) && (n_single == 0 || sgl),
               "fk_mmdit_blocks_fwd: null weights / modulation");
  for (int i = 0; i < n_double; ++i) FK_TRY(double_block(*ws, d, dbl[i], mod, mod_batch_stride, stream));
  for (int i = 0; i < n_single; ++i) FK_TRY(single_block(*ws, d, sgl[i], mod, mod_batch_stride, stream));
  return FK_OK;
}
