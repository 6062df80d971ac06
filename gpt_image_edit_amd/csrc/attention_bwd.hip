// Backward of the joint attention (adjoint of attention_fwd.hip; reference: autograd through
// F.scaled_dot_product_attention in FluxAttnProcessor2_0, reached from train_denoiser.py:1172).
//
//   P = softmax(Q K^T c),  O = P V             D_q = sum_d dO O  (fk_rowdot_bf16),  lse_q from the forward (log2 domain)
//   dV = P^T dO          dP = dO V^T          dS = P o (dP - D) c          dQ = dS K          dK = dS^T Q
//
// Three passes of ONE kernel skeleton, each the shape of the forward kernel -- a "row" operand held in registers (32
// rows per wave, 256 per workgroup), the "column" operand streamed through LDS in tiles of 64 by LDS-DMA, two MFMA
// products that land transposed so that every lane owns one row, an elementwise stage, and one accumulating product
// through the LDS transpose read:
//   MODE_DQ  rows = queries (Q, dO in registers), columns = keys:     s = K Q^T, dp = V dO^T, w = p (dp - D) c, dQ^T += K^T w
//   MODE_DV  rows = keys (K in registers),        columns = queries:  s = Q K^T,              w = p,            dV^T += dO^T w
//   MODE_DK  rows = keys (K, V in registers),     columns = queries:  s = Q K^T, dp = dO V^T, w = p (dp - D) c, dK^T += Q^T w
// with p = exp2(s c' - lse).  No atomics, no cross-workgroup reduction: gradients are deterministic; the price is that
// S and dP are recomputed per pass (8 tile products instead of the 5 of a fused backward).  Every streamed tensor that
// feeds both a row-fragment product and a transposed product is staged twice (two LDS images, two swizzles), as the
// forward does for K (row fragments) and V (transpose read).
#include <type_traits>

#include "fk_common.h"

namespace {

constexpr int HD = 128;
constexpr int CBLK = 64;                        // columns per tile
constexpr int IMG = CBLK * HD * 2;              // one LDS image of a tile: 16 KiB
constexpr int STAGE_BYTES = 3 * IMG + 512;      // A | Bt | C | lse[64] dsum[64]
constexpr int STAGES = 3, PF = STAGES - 1;
enum { MODE_DQ = 0, MODE_DV = 1, MODE_DK = 2 };

struct TView {           // element (b, h, s, d) at p + b*bs + h*hs + s*ld + d
  const bf16_t* p;
  int64_t ld, hs, bs;
};
struct BwdParams {
  TView q, k, v, dout;
  const float* lse;      // [B, H, S] log2-domain log-sum-exp of the forward
  const float* dsum;     // [B, H, S] sum_d dO * O
  bf16_t* out;           // gradient written by this pass
  int64_t o_ld, o_hs, o_bs;
  int B, H, S;
  float scale, scale_log2;
};

typedef __attribute__((address_space(3))) void lds_void;
typedef short s16x4_t __attribute__((ext_vector_type(4)));

template <int BYTES>
FK_DEV void buffer_lds(__amdgpu_buffer_rsrc_t rsrc, char* lds_dst, int voffset, int soffset) {
#if defined(__HIP_DEVICE_COMPILE__)
  // the size operand must be a literal, not a template-dependent expression
  if constexpr (BYTES == 16) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)lds_dst, 16, voffset, soffset, 0, 0);
  else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)lds_dst, 4, voffset, soffset, 0, 0);
#else
  (void)rsrc; (void)lds_dst; (void)voffset; (void)soffset;
#endif
}
FK_DEV s16x4_t lds_tr16(const char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p);
}
template <int N>
FK_DEV void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int MODE>
__global__ __launch_bounds__(512, 2) void attention_bwd_kernel(const BwdParams p) {
  constexpr bool HAS_C = MODE != MODE_DV;          // second product (dp) and its streamed image
  constexpr bool COL_STATS = MODE != MODE_DQ;      // lse / D vary along the streamed dimension
  constexpr int LOADS = 4 + (HAS_C ? 2 : 0) + (COL_STATS ? (MODE == MODE_DK ? 2 : 1) : 0);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ql = lane & 31, hh = lane >> 5;

  const int nrb = (p.S + 255) / 256;
  int t0;
  {
    const int nwg = gridDim.x;
    const int q8 = nwg >> 3, r8 = nwg & 7;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    t0 = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
  }
  const int rb = t0 % nrb;
  const int bh = t0 / nrb;
  const int b = bh / p.H, h = bh - b * p.H;

  // which tensors play which role
  const TView& X1 = MODE == MODE_DQ ? p.q : p.k;
  const TView& X2 = MODE == MODE_DQ ? p.dout : p.v;
  const TView& TA = MODE == MODE_DQ ? p.k : p.q;                            // row-fragment image A
  const TView& TB = MODE == MODE_DQ ? p.k : (MODE == MODE_DV ? p.dout : p.q);   // transposed image Bt
  const TView& TC = MODE == MODE_DQ ? p.v : p.dout;                         // row-fragment image C

  // ---- stationary row operands (B operand of the swapped products): lane holds X[row][16 kk + 8 hh .. +8] ----------
  const int row = rb * 256 + wave * 32 + ql;
  const int rowc = min(row, p.S - 1);
  bf16x8_t x1f[8], x2f[8];
  {
    const bf16_t* xp = X1.p + (int64_t)b * X1.bs + (int64_t)h * X1.hs + (int64_t)rowc * X1.ld + 8 * hh;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) x1f[kk] = *(const bf16x8_t*)(xp + 16 * kk);
    if constexpr (HAS_C) {
      const bf16_t* yp = X2.p + (int64_t)b * X2.bs + (int64_t)h * X2.hs + (int64_t)rowc * X2.ld + 8 * hh;
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) x2f[kk] = *(const bf16x8_t*)(yp + 16 * kk);
    }
  }
  float lse_l = 0.f, d_l = 0.f;
  if constexpr (MODE == MODE_DQ) {
    lse_l = p.lse[(int64_t)bh * p.S + rowc];
    d_l = p.dsum[(int64_t)bh * p.S + rowc];
  }

  // ---- LDS-DMA of the streamed tiles: piece = 4 rows x 256 B, lane -> (row = lane / 16, 16-byte slot = lane % 16) ---
  const int prow = lane >> 4, pslot = lane & 15;
  auto rsrc_of = [&](const TView& t) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)(t.p + (int64_t)b * t.bs + (int64_t)h * t.hs), 0,
                                             (int)(((int64_t)(p.S - 1) * t.ld + HD) * 2), 0x00020000);
  };
  const __amdgpu_buffer_rsrc_t rs_a = rsrc_of(TA), rs_b = rsrc_of(TB), rs_c = rsrc_of(TC);
  const __amdgpu_buffer_rsrc_t rs_l = __builtin_amdgcn_make_buffer_rsrc((void*)(p.lse + (int64_t)bh * p.S), 0, p.S * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_d = __builtin_amdgcn_make_buffer_rsrc((void*)(p.dsum + (int64_t)bh * p.S), 0, p.S * 4, 0x00020000);
  const int nt = (p.S + CBLK - 1) / CBLK;
  const bool ragged = p.S % CBLK != 0;
  // byte offset of tile row r (clamped to the last valid row of the tensor: nothing beyond S is ever fetched)
  auto voff_rowfrag = [&](const TView& t, int r) { return (int)((r * t.ld + ((pslot ^ (r & 15)) << 3)) * 2); };
  auto voff_transp = [&](const TView& t, int r, int rs) {
    const int vcol = ((((pslot >> 2) ^ (rs & 3)) << 5) + ((pslot & 3) << 3));
    return (int)((r * t.ld + vcol) * 2);
  };
  int va[2], vb[2], vc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = (wave * 2 + i) * 4 + prow;
    va[i] = voff_rowfrag(TA, r);
    vb[i] = voff_transp(TB, r, r);
    vc[i] = voff_rowfrag(TC, r);
  }
  auto issue_tile = [&](int t, int stage) {
    char* sb = smem + stage * STAGE_BYTES;
    const int base_row = t * CBLK;
    int a0 = va[0], a1 = va[1], b0 = vb[0], b1 = vb[1], c0 = vc[0], c1 = vc[1], sl = lane * 4;
    if (ragged && t == nt - 1) {   // rows beyond S: re-read the last valid row (its products are masked to zero)
      const int last = p.S - 1 - base_row;
      const int r0 = (wave * 2) * 4 + prow, r1 = r0 + 4;
      const int q0 = min(r0, last), q1 = min(r1, last);
      // the swizzle follows the LDS row (r), the source row is clamped (q)
      a0 = (int)((q0 * TA.ld + ((pslot ^ (r0 & 15)) << 3)) * 2);
      a1 = (int)((q1 * TA.ld + ((pslot ^ (r1 & 15)) << 3)) * 2);
      b0 = voff_transp(TB, q0, r0);
      b1 = voff_transp(TB, q1, r1);
      c0 = (int)((q0 * TC.ld + ((pslot ^ (r0 & 15)) << 3)) * 2);
      c1 = (int)((q1 * TC.ld + ((pslot ^ (r1 & 15)) << 3)) * 2);
      sl = min(lane, last) * 4;
    }
    buffer_lds<16>(rs_a, sb + (wave * 2) * 1024, a0, (int)(base_row * TA.ld * 2));
    buffer_lds<16>(rs_a, sb + (wave * 2 + 1) * 1024, a1, (int)(base_row * TA.ld * 2));
    buffer_lds<16>(rs_b, sb + IMG + (wave * 2) * 1024, b0, (int)(base_row * TB.ld * 2));
    buffer_lds<16>(rs_b, sb + IMG + (wave * 2 + 1) * 1024, b1, (int)(base_row * TB.ld * 2));
    if constexpr (HAS_C) {
      buffer_lds<16>(rs_c, sb + 2 * IMG + (wave * 2) * 1024, c0, (int)(base_row * TC.ld * 2));
      buffer_lds<16>(rs_c, sb + 2 * IMG + (wave * 2 + 1) * 1024, c1, (int)(base_row * TC.ld * 2));
    }
    if constexpr (COL_STATS) {   // every wave writes the same 64 floats (identical values): uniform request counts
      buffer_lds<4>(rs_l, sb + 3 * IMG, sl, base_row * 4);
      if constexpr (MODE == MODE_DK) buffer_lds<4>(rs_d, sb + 3 * IMG + 256, sl, base_row * 4);
    }
  };

  // ---- operand reads -------------------------------------------------------------------------------------------------
  const int k_rd = ql * 256, k_sw = ql & 15;
  const int tj = (lane & 15) >> 2, tq = lane & 3, tdh = (lane >> 4) & 1;
  const int v_rd = IMG + (4 * hh + tj) * 256 + tdh * 32 + tq * 8;
  auto rowfrag = [&](const char* sb, int img, int kb, int kk) {
    return *(const bf16x8_t*)(sb + img * IMG + k_rd + kb * 8192 + (((2 * kk + hh) ^ k_sw) << 4));
  };
  auto trfrag = [&](const char* sb, int st, int df) {   // columns 16 st + {0, 8} + 4 hh + 0..3, d block df
    const char* vp = sb + v_rd + st * 4096 + ((df ^ tj) << 6);
    const s16x4_t lo = lds_tr16(vp);
    const s16x4_t hi = lds_tr16(vp + 2048);
    bf16x8_t f;
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
    f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    return f;
  };

  f32x16_t acc[4];
#pragma unroll
  for (int df = 0; df < 4; ++df)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[df][r] = 0.f;

  int st_cur = 0, st_pf = PF;
#pragma unroll
  for (int s = 0; s < PF; ++s)
    if (s < nt) issue_tile(s, s);
  // waves w and w + 4 share a SIMD; the later-dispatched half loses the VALU arbitration at the head of every segment:
  // one static priority raise for it (MI355X_MICROARCH.md, "Two waves per SIMD", item 4); the condition is wave-uniform
  if (wave >= 4) __builtin_amdgcn_s_setprio(1);

  for (int t = 0; t < nt; ++t) {
    if (t + PF - 1 < nt) wait_vmcnt<(PF - 1) * LOADS>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (t + PF < nt) issue_tile(t + PF, st_pf);
    const char* sb = smem + st_cur * STAGE_BYTES;
    const bool mask = ragged && t == nt - 1;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      // Operand fragments are read two MFMAs ahead of their use and the order is pinned (one DS read, then one MFMA), as
      // in attention_fwd.hip: left alone, hipcc issues every ds_read right in front of its MFMA and the wave sits out an
      // LDS round trip 48 times per tile.  Same products in the same order: results are unchanged bit for bit.
      f32x16_t s, dp;
      auto product = [&](f32x16_t& out, int img, const bf16x8_t (&xf)[8]) {
        bf16x8_t kf[3];
        kf[0] = rowfrag(sb, img, kb, 0);
        kf[1] = rowfrag(sb, img, kb, 1);
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
          if (kk + 2 < 8) kf[(kk + 2) % 3] = rowfrag(sb, img, kb, kk + 2);
          out = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kk % 3], xf[kk], kk == 0 ? f32x16_t{} : out, 0, 0, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        }
      };
      product(s, 0, x1f);
      if constexpr (HAS_C) product(dp, 2, x2f);
      // ---- elementwise: w = p (DV) or p (dp - D) scale (DQ, DK), p = exp2(s c' - lse) <= 1 -------------------------
      float w[16];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4_t lv = {lse_l, lse_l, lse_l, lse_l}, dv = {d_l, d_l, d_l, d_l};
        if constexpr (COL_STATS) {
          lv = *(const f32x4_t*)(sb + 3 * IMG + (32 * kb + 8 * g + 4 * hh) * 4);
          if constexpr (MODE == MODE_DK) dv = *(const f32x4_t*)(sb + 3 * IMG + 256 + (32 * kb + 8 * g + 4 * hh) * 4);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int r = 4 * g + j;
          const float pr = __builtin_amdgcn_exp2f(fminf(fmaf(s[r], p.scale_log2, -lv[j]), 0.f));
          float wv = pr;
          if constexpr (HAS_C) wv = pr * (dp[r] - dv[j]) * p.scale;
          if (mask && t * CBLK + 32 * kb + 8 * g + 4 * hh + j >= p.S) wv = 0.f;
          w[r] = wv;
        }
      }
      // ---- acc^T[d][row] += Bt^T w for the two 16-column steps of this half ------------------------------------------
      bf16x8_t pfs[2];
#pragma unroll
      for (int step = 0; step < 2; ++step) {
        u32x4_t pw;
#pragma unroll
        for (int e = 0; e < 4; ++e) pw[e] = pack_bf2(w[8 * step + 2 * e], w[8 * step + 2 * e + 1]);
        pfs[step] = __builtin_bit_cast(bf16x8_t, pw);
      }
      {
        bf16x8_t tf[3];
        tf[0] = trfrag(sb, 2 * kb, 0);
        tf[1] = trfrag(sb, 2 * kb, 1);
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);   // two transpose reads per fragment
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int df = i & 3;
          if (i + 2 < 8) tf[(i + 2) % 3] = trfrag(sb, 2 * kb + ((i + 2) >> 2), (i + 2) & 3);
          acc[df] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tf[i % 3], pfs[i >> 2], acc[df], 0, 0, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        }
      }
    }
    st_cur = (st_cur == STAGES - 1) ? 0 : st_cur + 1;
    st_pf = (st_pf == STAGES - 1) ? 0 : st_pf + 1;
  }

  // ---- store: lane (row = ql) holds d = 32 df + 8 g + 4 hh + (0..3) ------------------------------------------------
  if (row < p.S) {
    bf16_t* op = p.out + (int64_t)b * p.o_bs + (int64_t)h * p.o_hs + (int64_t)row * p.o_ld + 4 * hh;
#pragma unroll
    for (int df = 0; df < 4; ++df)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        u32x2_t pk;
        pk[0] = pack_bf2(acc[df][4 * g + 0], acc[df][4 * g + 1]);
        pk[1] = pack_bf2(acc[df][4 * g + 2], acc[df][4 * g + 3]);
        *(u32x2_t*)(op + 32 * df + 8 * g) = pk;
      }
  }
}

template <int MODE>
int launch_bwd(const BwdParams& p, hipStream_t stream) {
  constexpr int SMEM = STAGES * STAGE_BYTES;
  auto kern = attention_bwd_kernel<MODE>;
  FK_ENSURE_MAX_LDS(kern, SMEM, "fk_attention_bwd_bf16");
  const int nrb = (p.S + 255) / 256;
  hipLaunchKernelGGL(kern, dim3(nrb * p.H * p.B), dim3(512), SMEM, stream, p);
  FK_CHECK_LAUNCH("fk_attention_bwd_bf16");
  return FK_OK;
}

bool view_ok(const fk_attn_view& v, int64_t S) {
  return v.p && ((uintptr_t)v.p % 16 == 0) && v.ld > 0 && v.ld % 8 == 0 && v.head_stride % 8 == 0 && v.batch_stride % 8 == 0 &&
         ((S - 1) * v.ld + HD) * 2 < (1ll << 31) && (S + CBLK) * v.ld * 2 < (1ll << 31);
}
TView tv(const fk_attn_view& v) { return TView{(const bf16_t*)v.p, v.ld, v.head_stride, v.batch_stride}; }

}  // namespace

extern "C" int fk_attention_bwd_bf16(const fk_attn_view* q, const fk_attn_view* k, const fk_attn_view* v,
                                     const fk_attn_view* dout, const float* lse, const float* dsum, const fk_attn_view* dq,
                                     const fk_attn_view* dk, const fk_attn_view* dv, int32_t B, int32_t H, int32_t S,
                                     float scale, fk_stream_t stream_) {
  FK_CHECK_ARG(q && k && v && dout && lse && dsum && dq && dk && dv, "fk_attention_bwd_bf16: null pointer");
  FK_CHECK_ARG(B > 0 && H > 0 && S > 0, "fk_attention_bwd_bf16: bad B/H/S %d %d %d", B, H, S);
  const fk_attn_view* all[7] = {q, k, v, dout, dq, dk, dv};
  for (const fk_attn_view* a : all)
    FK_CHECK_ARG(view_ok(*a, S), "fk_attention_bwd_bf16: every view needs a 16-byte aligned pointer, strides that are multiples "
                                 "of 8 elements and a (batch, head) extent below 2 GiB");
  BwdParams p;
  p.q = tv(*q); p.k = tv(*k); p.v = tv(*v); p.dout = tv(*dout);
  p.lse = lse; p.dsum = dsum;
  p.B = B; p.H = H; p.S = S;
  p.scale = scale; p.scale_log2 = scale * 1.4426950408889634f;
  hipStream_t stream = (hipStream_t)stream_;
  auto set_out = [&](const fk_attn_view& o) {
    p.out = (bf16_t*)o.p; p.o_ld = o.ld; p.o_hs = o.head_stride; p.o_bs = o.batch_stride;
  };
  set_out(*dq);
  int rc = launch_bwd<MODE_DQ>(p, stream);
  if (rc != FK_OK) return rc;
  set_out(*dv);
  rc = launch_bwd<MODE_DV>(p, stream);
  if (rc != FK_OK) return rc;
  set_out(*dk);
  return launch_bwd<MODE_DK>(p, stream);
}
