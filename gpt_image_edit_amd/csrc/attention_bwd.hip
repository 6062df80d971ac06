// Backward of the joint attention (adjoint of attention_fwd.hip; reference: autograd through
// F.scaled_dot_product_attention in FluxAttnProcessor2_0, reached from train_denoiser.py:1172).
//
//   P = softmax(Q K^T c),  O = P V             D_q = sum_d dO O  (fk_rowdot_bf16),  lse_q from the forward (log2 domain)
//   dV = P^T dO          dP = dO V^T          dS = P o (dP - D) c          dQ = dS K          dK = dS^T Q
//
// The passes share ONE kernel skeleton, the shape of the forward kernel -- a "row" operand held in registers (32 rows
// per wave), the "column" operand streamed through LDS in tiles of 64 by LDS-DMA, MFMA products that land transposed so
// that every lane owns one row, an elementwise stage, and an accumulating product through the LDS transpose read:
//   MODE_DQ  rows = queries (Q, dO in registers), columns = keys:     s = K Q^T, dp = V dO^T, w = p (dp - D) c, dQ^T += K^T w
//   MODE_DV  rows = keys (K in registers),        columns = queries:  s = Q K^T,              w = p,            dV^T += dO^T w
//   MODE_DK  rows = keys (K, V in registers),     columns = queries:  s = Q K^T, dp = dO V^T, w = p (dp - D) c, dK^T += Q^T w
// with p = exp2(s c' - lse).  No atomics, no cross-workgroup reduction: gradients are deterministic.
// The elementwise stage is one fused multiply-add, one exponential and one multiply per score: the dp product's accumulator
// STARTS at D and its stationary operand (dO resp. V) is held negated, so the MFMAs deliver D - dP; the sign and the softmax
// scale c multiply the accumulator once at the store; the ragged last tile (compares / selects per score) is its own
// instantiation outside the main loop.  These kernels are VALU-issue-bound, not MFMA-bound: each of those steps paid.
//
// Default since round 3: TWO passes, 7 tile products and 2 exponentials per score instead of 8 and 3.  dK and dV come out
// of one launch (attention_bwd_dkv_kernel) in which the two waves that share a SIMD split the work on the same 32 keys:
//   producer (waves 0-3)  s = Q K^T, p = exp2(s c' - lse), hands bf16(p) over through LDS,          dV^T += dO^T p
//   consumer (waves 4-7)  dp = dO V^T, w = bf16(p) (dp - D) c,                                      dK^T += Q^T w
// Both accumulators and both stationary operands do not fit 256 registers of one wave; split like this every wave needs
// ~150, every product is computed once, and the hand-over is lane-to-same-lane (the producer's packed B-operand fragments
// ARE the consumer's), one tile behind, ordered by the per-tile workgroup barrier -- no flags, no polling.
// fk_attention_bwd_set_mode(0) / FK_ATTN_BWD=0 selects the three-pass form (dK then differs in the last bf16 bit: there
// w is formed from the fp32 p).
//
// One LDS image per streamed tensor: the 16-byte slot c of tile row r sits at slot c ^ swz(r), swz(r) = the two bit pairs
// of (r & 15) exchanged.  Row-fragment reads (ds_read_b128, 16 rows x one logical slot per lane group) see 16 distinct
// slots because swz is a bijection of r & 15; transpose reads (ds_read_b64_tr_b16, 4 rows x the 4 slots of one 64-byte
// block per 32-lane group) see 4 distinct blocks because the high pair of swz(r) is r & 3.  So K in the dQ pass and Q in
// the dK pass are staged once, not twice (two DMA pieces per wave and tile less).
#include <type_traits>

#include "fk_common.h"

namespace {

constexpr int HD = 128;
constexpr int CBLK = 64;                        // columns per tile
constexpr int IMG = CBLK * HD * 2;              // one LDS image of a tile: 16 KiB
constexpr int STAGE_BYTES = 2 * IMG + 512;      // image 0 | image 1 | lse[64] dsum[64]
constexpr int STAGES = 3, PF = STAGES - 1;
#ifndef FK_BWD_AHEAD
#define FK_BWD_AHEAD 2
#endif
#ifndef FK_BWD_PRIO
#define FK_BWD_PRIO 1
#endif
constexpr int AH = FK_BWD_AHEAD;                // operand fragments are read AH MFMAs ahead of their use (AH <= 4: d blocks of one step)
constexpr int PBUF_BYTES = 4 * 2 * 4096;        // dK+dV kernel: per wave pair, two tiles of packed p (2 halves x 2 steps x 64 lanes x 16 B)
enum { MODE_DQ = 0, MODE_DV = 1, MODE_DK = 2 };

struct TView {           // element (b, h, s, d) at p + b*bs + h*hs + s*ld + d
  const bf16_t* p;
  int64_t ld, hs, bs;
};
struct BwdParams {
  TView q, k, v, dout;
  const float* lse;      // [B, H, S] log2-domain log-sum-exp of the forward
  const float* dsum;     // [B, H, S] sum_d dO * O
  bf16_t* out;           // gradient written by this pass (dK of the dK+dV pass)
  int64_t o_ld, o_hs, o_bs;
  bf16_t* out2;          // dV of the dK+dV pass
  int64_t o2_ld, o2_hs, o2_bs;
  int B, H, S;
  float scale, scale_log2;
  // stream-K launch of the dQ pass (attention_fwd.hip has the scheme): sk_rounds whole rounds of one 256-row item per
  // workgroup, then the remaining items' column tiles dealt out as G equal contiguous ranges; the two parts of a cut item
  // add their fp32 accumulators through workspace slot j (ticket / flag pair sk_ctl[2 j], [2 j + 1])
  int n_items, sk_rounds, min_part;
  float* sk_partials;
  unsigned* sk_ctl;
};
constexpr int PART_FLOATS = 256 * 128 + 2 * 512;   // the attention forward's slot layout (the (l, m) pairs are unused here)

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
template <int BYTES>
FK_DEV void buffer_lds(const BufDesc& d, char* lds_dst, int voffset, int soffset) {
  buffer_lds_opaque<BYTES>(d, lds_addr_of(lds_dst), voffset, soffset);
}
FK_DEV s16x4_t lds_tr16(const char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p);
}
template <int N>
FK_DEV void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
FK_DEV int swz(int r) { return ((r & 3) << 2) | ((r >> 2) & 3); }   // slot XOR of LDS row r (header)

template <int MODE, bool STREAMK = false>
__global__ __launch_bounds__(512, 2) void attention_bwd_kernel(const BwdParams p) {
  constexpr bool HAS_C = MODE != MODE_DV;          // second product (dp) and its streamed image
  constexpr bool COL_STATS = MODE != MODE_DQ;      // lse / D vary along the streamed dimension
  constexpr int LOADS = 4 + (COL_STATS ? (MODE == MODE_DK ? 2 : 1) : 0);
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
  const int nt = (p.S + CBLK - 1) / CBLK;
  const bool ragged = p.S % CBLK != 0;
  // ---- work list (attention_fwd.hip, "the workgroup's work list"): plain launch = the one item t0 ---------------------------
  int u = 0, u_end = 0, round = 0;
  if constexpr (STREAMK) {
    const unsigned G = gridDim.x;
    const unsigned U = (unsigned)(p.n_items - p.sk_rounds * (int)G) * (unsigned)nt;
    const unsigned qU = U / G, rU = U - qU * G;
    auto cut = [&](unsigned j) __attribute__((always_inline)) {
      unsigned c = qU * j + (rU * j) / G;
      const unsigned r = c % (unsigned)nt;
      if (r != 0 && r < (unsigned)p.min_part) c -= r;
      else if (r != 0 && (unsigned)nt - r < (unsigned)p.min_part) c += (unsigned)nt - r;
      return (int)c;
    };
    u = cut(t0);
    u_end = cut(t0 + 1);
  }
  // operand read addresses (lane constants)
  const int k_rd = ql * 256, k_sw = swz(ql);
  const int tj = (lane & 15) >> 2, tq = lane & 3, tdh = (lane >> 4) & 1;
  // transpose read of tile rows 16 st + 4 hh + tj (lo) and + 8 (hi): logical slot 4 df + 2 tdh + tq / 2, physical slot
  // ^ swz(row) = 4 (df ^ tj) + ((2 tdh + tq / 2) ^ (hh + 2 hi))
  const int t_lo = (4 * hh + tj) * 256 + (((2 * tdh + (tq >> 1)) ^ hh) << 4) + (tq & 1) * 8;
  const int t_hi = (4 * hh + tj + 8) * 256 + (((2 * tdh + (tq >> 1)) ^ (hh + 2)) << 4) + (tq & 1) * 8;
  const int prow = lane >> 4, pslot = lane & 15;
  // waves w and w + 4 share a SIMD; the later-dispatched half loses the VALU arbitration at the head of every segment:
  // one static priority raise for it (MI355X_MICROARCH.md, "Two waves per SIMD", item 4); the condition is wave-uniform
  if (wave >= 4) __builtin_amdgcn_s_setprio(1);

  for (;;) {   // one pass per (item, column-tile range); a plain launch makes exactly one
  int item = t0, tb = 0, te = nt;                       // this pass: column tiles [tb, te) of the item
  if constexpr (STREAMK) {
    if (round < p.sk_rounds) {
      item = round * (int)gridDim.x + t0;
      ++round;
    } else {
      if (u >= u_end) break;      // walked from the END of the range: every pass but the last starts at tile 0 (attention_fwd.hip)
      const int ti = (unsigned)(u_end - 1) / (unsigned)nt;
      item = p.sk_rounds * (int)gridDim.x + ti;
      te = u_end - ti * nt;
      tb = max(u - ti * nt, 0);
      u_end -= te - tb;
    }
  }
  const int rb = item % nrb;
  const int bh = item / nrb;
  const int b = bh / p.H, h = bh - b * p.H;

  // which tensors play which role: image 0 feeds the s product (row fragments), image 1 the dp product (row fragments) or,
  // in the dV pass, the accumulating product (transpose reads); the dQ / dK passes transpose-read image 0
  const TView& X1 = MODE == MODE_DQ ? p.q : p.k;
  const TView& X2 = MODE == MODE_DQ ? p.dout : p.v;
  const TView& T0 = MODE == MODE_DQ ? p.k : p.q;
  const TView& T1 = MODE == MODE_DQ ? p.v : p.dout;
  constexpr int TR_IMG = MODE == MODE_DV ? 1 : 0;

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
      for (int kk = 0; kk < 8; ++kk)   // NEGATED: the dp product accumulates D - dO V^T (dQ: D - V dO^T) on top of D, see below
        x2f[kk] = __builtin_bit_cast(bf16x8_t, __builtin_bit_cast(u32x4_t, *(const bf16x8_t*)(yp + 16 * kk)) ^ 0x80008000u);
    }
  }
  float lse_l = 0.f, d_l = 0.f;
  if constexpr (MODE == MODE_DQ) {
    lse_l = p.lse[(int64_t)bh * p.S + rowc];
    d_l = p.dsum[(int64_t)bh * p.S + rowc];
  }
  f32x16_t dini;   // dQ pass: the dp accumulator's start, D of this lane's row in all 16 elements
#pragma unroll
  for (int r = 0; r < 16; ++r) dini[r] = d_l;

  // ---- LDS-DMA of the streamed tiles: piece = 4 rows x 256 B, lane -> (row = lane / 16, 16-byte slot = lane % 16) ---
  auto rsrc_of = [&](const TView& t) __attribute__((always_inline)) {
    return make_dma_desc(t.p + (int64_t)b * t.bs + (int64_t)h * t.hs, ((int64_t)(p.S - 1) * t.ld + HD) * 2);
  };
  const DmaDesc rs_0 = rsrc_of(T0), rs_1 = rsrc_of(T1);
  const DmaDesc rs_l = make_dma_desc(p.lse + (int64_t)bh * p.S, (int64_t)p.S * 4);
  const DmaDesc rs_d = make_dma_desc(p.dsum + (int64_t)bh * p.S, (int64_t)p.S * 4);
  // byte offset of source row q for LDS row r: the swizzle follows the LDS row
  auto voff = [&](const TView& t, int q, int r) __attribute__((always_inline)) { return (int)((q * t.ld + ((pslot ^ swz(r)) << 3)) * 2); };
  int v0[2], v1[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = (wave * 2 + i) * 4 + prow;
    v0[i] = voff(T0, r, r);
    v1[i] = voff(T1, r, r);
  }
  auto issue_tile = [&](int t, int stage) __attribute__((always_inline)) {
    char* sb = smem + stage * STAGE_BYTES;
    const int base_row = t * CBLK;
    int a0 = v0[0], a1 = v0[1], c0 = v1[0], c1 = v1[1], sl = lane * 4;
    if (ragged && t == nt - 1) {   // rows beyond S: re-read the last valid row (its products are masked to zero)
      const int last = p.S - 1 - base_row;
      const int r0 = (wave * 2) * 4 + prow, r1 = r0 + 4;
      const int q0 = min(r0, last), q1 = min(r1, last);
      a0 = voff(T0, q0, r0);
      a1 = voff(T0, q1, r1);
      c0 = voff(T1, q0, r0);
      c1 = voff(T1, q1, r1);
      sl = min(lane, last) * 4;
    }
    buffer_lds<16>(rs_0, sb + (wave * 2) * 1024, a0, (int)(base_row * T0.ld * 2));
    buffer_lds<16>(rs_0, sb + (wave * 2 + 1) * 1024, a1, (int)(base_row * T0.ld * 2));
    buffer_lds<16>(rs_1, sb + IMG + (wave * 2) * 1024, c0, (int)(base_row * T1.ld * 2));
    buffer_lds<16>(rs_1, sb + IMG + (wave * 2 + 1) * 1024, c1, (int)(base_row * T1.ld * 2));
    if constexpr (COL_STATS) {   // every wave writes the same 64 floats (identical values): uniform request counts
      buffer_lds<4>(rs_l, sb + 2 * IMG, sl, base_row * 4);
      if constexpr (MODE == MODE_DK) buffer_lds<4>(rs_d, sb + 2 * IMG + 256, sl, base_row * 4);
    }
  };

  // ---- operand reads -------------------------------------------------------------------------------------------------
  auto rowfrag = [&](const char* sb, int img, int kb, int kk) __attribute__((always_inline)) {
    return *(const bf16x8_t*)(sb + img * IMG + k_rd + kb * 8192 + (((2 * kk + hh) ^ k_sw) << 4));
  };
  auto trfrag = [&](const char* sb, int st, int df) __attribute__((always_inline)) {   // columns 16 st + {0, 8} + 4 hh + 0..3, d block df
    const char* vp = sb + TR_IMG * IMG + st * 4096 + ((df ^ tj) << 6);
    const s16x4_t lo = lds_tr16(vp + t_lo);
    const s16x4_t hi = lds_tr16(vp + t_hi);
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
    if (tb + s < te) issue_tile(tb + s, s);

  // one streamed tile; with the mask tag the ragged last one (w of the columns beyond S zeroed) -- two instantiations,
  // so the 135 full tiles of a 136-tile row do not carry the compare / select pairs
  auto tile_body = [&](const char* sb, int t, auto mask_tag) __attribute__((always_inline)) {
    constexpr bool MASK = decltype(mask_tag)::value;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      // Operand fragments are read two MFMAs ahead of their use and the order is pinned (one DS read, then one MFMA), as
      // in attention_fwd.hip: left alone, hipcc issues every ds_read right in front of its MFMA and the wave sits out an
      // LDS round trip 48 times per tile.  Same products in the same order: results are unchanged bit for bit.
      f32x16_t s, dp;
      auto product = [&](f32x16_t& out, int img, const bf16x8_t (&xf)[8], const f32x16_t& init) {
        bf16x8_t kf[AH + 1];
#pragma unroll
        for (int a = 0; a < AH; ++a) kf[a] = rowfrag(sb, img, kb, a);
        __builtin_amdgcn_sched_group_barrier(0x100, AH, 0);
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
          if (kk + AH < 8) kf[(kk + AH) % (AH + 1)] = rowfrag(sb, img, kb, kk + AH);
          out = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kk % (AH + 1)], xf[kk], kk == 0 ? init : out, 0, 0, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        }
      };
      product(s, 0, x1f, f32x16_t{});
      // dp: the accumulator STARTS at D (of the row: dQ pass; of the streamed columns: dK pass) and the stationary operand is
      // negated, so the product delivers D - dP and the elementwise stage is one multiply; the sign joins the scale at the store
      if constexpr (MODE == MODE_DQ) product(dp, 1, x2f, dini);
      if constexpr (MODE == MODE_DK) {
        f32x16_t dcol;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4_t dv = *(const f32x4_t*)(sb + 2 * IMG + 256 + (32 * kb + 8 * g + 4 * hh) * 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) dcol[4 * g + j] = dv[j];
        }
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
        product(dp, 1, x2f, dcol);
      }
      // ---- elementwise: w = p (DV) or p (dp - D) scale (DQ, DK), p = exp2(s c' - lse) <= 1 -------------------------
      float w[16];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4_t lv = {lse_l, lse_l, lse_l, lse_l};
        if constexpr (COL_STATS) lv = *(const f32x4_t*)(sb + 2 * IMG + (32 * kb + 8 * g + 4 * hh) * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int r = 4 * g + j;
          const float pr = __builtin_amdgcn_exp2f(fmaf(s[r], p.scale_log2, -lv[j]));
          float wv = pr;
          if constexpr (HAS_C) wv = pr * dp[r];   // = -p (dP - D); sign and softmax scale multiply the accumulator at the store
          if constexpr (MASK)
            if (t * CBLK + 32 * kb + 8 * g + 4 * hh + j >= p.S) wv = 0.f;
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
        bf16x8_t tf[AH + 1];
#pragma unroll
        for (int a = 0; a < AH; ++a) tf[a] = trfrag(sb, 2 * kb, a);
        __builtin_amdgcn_sched_group_barrier(0x100, 2 * AH, 0);   // two transpose reads per fragment
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int df = i & 3;
          if (i + AH < 8) tf[(i + AH) % (AH + 1)] = trfrag(sb, 2 * kb + ((i + AH) >> 2), (i + AH) & 3);
          acc[df] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tf[i % (AH + 1)], pfs[i >> 2], acc[df], 0, 0, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        }
      }
    }
  };

  auto acquire = [&](int t) __attribute__((always_inline)) {   // tile t has landed everywhere; the stage of tile t - 1 is free for tile t + PF
    if (t + PF - 1 < te) wait_vmcnt<(PF - 1) * LOADS>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (t + PF < te) issue_tile(t + PF, st_pf);
    const char* sb = smem + st_cur * STAGE_BYTES;
    st_cur = (st_cur == STAGES - 1) ? 0 : st_cur + 1;
    st_pf = (st_pf == STAGES - 1) ? 0 : st_pf + 1;
    return sb;
  };
  const bool masked = ragged && te == nt;               // the pass ends with the item's ragged tile
  const int n_plain = masked ? te - 1 : te;
  for (int t = tb; t < n_plain; ++t) tile_body(acquire(t), t, std::false_type{});
  if (masked) tile_body(acquire(nt - 1), nt - 1, std::true_type{});   // after the loop: one accumulator live range each

  // ---- stream-K seam (attention_fwd.hip): the two parts of a cut item ADD their accumulators; whoever arrives second does it
  if constexpr (STREAMK) {
    if (tb > 0 || te < nt) {                        // workgroup-uniform
      typedef __attribute__((address_space(1))) unsigned gu32;
      const int slot = te < nt ? t0 + 1 : t0;       // the cut's index, 1 .. G - 1
      gu32* const ctl = (gu32*)(p.sk_ctl + 2 * (size_t)slot);
      const __amdgpu_buffer_rsrc_t rs_p = __builtin_amdgcn_make_buffer_rsrc(
          (void*)(p.sk_partials + (size_t)slot * PART_FLOATS), 0, PART_FLOATS * 4, 0x00020000);
      __syncthreads();   // every wave is done with the ring: its first word now carries the ticket
      if (tid == 0) *(volatile unsigned*)smem = __hip_atomic_fetch_add(ctl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __syncthreads();
      const unsigned ticket = __builtin_amdgcn_readfirstlane(*(volatile unsigned*)smem);
      if ((ticket & 1u) == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const f32x16_t& a = acc[r >> 2];
          const int q4 = r & 3;
          const u32x4_t v = {__float_as_uint(a[4 * q4]), __float_as_uint(a[4 * q4 + 1]), __float_as_uint(a[4 * q4 + 2]),
                             __float_as_uint(a[4 * q4 + 3])};
          __builtin_amdgcn_raw_buffer_store_b128(v, rs_p, tid * 16, r * (512 * 16), /*sc1: write through*/ 16);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_store(ctl + 1, ticket + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        continue;
      }
      if (tid == 0) {
        int spins = 0;    // bounded (attention_fwd.hip): a corrupted workspace ends in NaN rows, not in a hung device
        while (__hip_atomic_load(ctl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != ticket && spins < (1 << 22)) {
          __builtin_amdgcn_s_sleep(8);
          ++spins;
        }
        *(volatile unsigned*)smem = spins >= (1 << 22);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      }
      __syncthreads();
      const bool gave_up = *(volatile unsigned*)smem != 0;
#pragma unroll
      for (int r0 = 0; r0 < 16; r0 += 4) {
        u32x4_t v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = __builtin_amdgcn_raw_buffer_load_b128(rs_p, tid * 16, (r0 + e) * (512 * 16), /*sc1*/ 16);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          f32x16_t& a = acc[(r0 + e) >> 2];
          const int q4 = (r0 + e) & 3;
#pragma unroll
          for (int j = 0; j < 4; ++j) a[4 * q4 + j] += __uint_as_float(v[e][j]);   // fp32 addition commutes: symmetric
        }
      }
      if (gave_up)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][r] = __builtin_nanf("");
    }
  }

  // ---- store: lane (row = ql) holds d = 32 df + 8 g + 4 hh + (0..3) ------------------------------------------------
  if (row < p.S) {
    const float osc = HAS_C ? -p.scale : 1.f;   // dQ, dK: the sign and the softmax scale left out of w
    bf16_t* op = p.out + (int64_t)b * p.o_bs + (int64_t)h * p.o_hs + (int64_t)row * p.o_ld + 4 * hh;
#pragma unroll
    for (int df = 0; df < 4; ++df)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        u32x2_t pk;
        pk[0] = pack_bf2(acc[df][4 * g + 0] * osc, acc[df][4 * g + 1] * osc);
        pk[1] = pack_bf2(acc[df][4 * g + 2] * osc, acc[df][4 * g + 3] * osc);
        *(u32x2_t*)(op + 32 * df + 8 * g) = pk;
      }
  }
  if constexpr (!STREAMK) break;
  else __syncthreads();   // every wave is done with the ring and the ticket word before the next pass
  }   // passes
}

// ---- dK and dV in one launch: producer / consumer wave pairs (header) ---------------------------------------------------
// Workgroup = 128 keys = 4 pairs of waves (w, w + 4), which the dispatcher places on the same SIMD.  Iteration i: the
// producers work on query tile i, the consumers on tile i - 1 (whose p the producers stored in iteration i - 1), the DMA
// fetches tile i + 1 into the stage the consumers left in iteration i - 1; one workgroup barrier per iteration orders all
// three.  LDS: 3 stages x (Q | dO | lse, D) + 32 KiB of packed p = 132 608 B.
// STREAMK (round 5): the dQ pass's persistent grid and seam for this pass too.  An item = 128 keys of one (b, h), its units the
// nt query tiles it walks; 1 632 items at S = 8704 are 6.4 rounds of 256 workgroups (run as 7: 9 % of the CU time idle).  The
// two parts of a cut item ADD their fp32 accumulators (the producers' dV^T, the consumers' dK^T) through the cut's workspace
// slot: commutative, so still deterministic.  Built, parity-green, and no faster than the plain grid (the launcher has the
// numbers): not the default.
template <bool STREAMK>
__global__ __launch_bounds__(512, 2) void attention_bwd_dkv_kernel(const BwdParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ql = lane & 31, hh = lane >> 5;
  const int pair = wave & 3;
  const bool producer = wave < 4;

  const int nrb = (p.S + 127) / 128;
  int t0;
  {
    const int nwg = gridDim.x;
    const int q8 = nwg >> 3, r8 = nwg & 7;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    t0 = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
  }
  const int nt = (p.S + CBLK - 1) / CBLK;
  const bool ragged = p.S % CBLK != 0;
  int u = 0, u_end = 0, round = 0;      // the work list of attention_bwd_kernel<.., STREAMK>
  if constexpr (STREAMK) {
    const unsigned G = gridDim.x;
    const unsigned U = (unsigned)(p.n_items - p.sk_rounds * (int)G) * (unsigned)nt;
    const unsigned qU = U / G, rU = U - qU * G;
    auto cut = [&](unsigned j) __attribute__((always_inline)) {
      unsigned c = qU * j + (rU * j) / G;
      const unsigned r = c % (unsigned)nt;
      if (r != 0 && r < (unsigned)p.min_part) c -= r;
      else if (r != 0 && (unsigned)nt - r < (unsigned)p.min_part) c += (unsigned)nt - r;
      return (int)c;
    };
    u = cut(t0);
    u_end = cut(t0 + 1);
  }
#if FK_BWD_PRIO == 1
  if (wave >= 4) __builtin_amdgcn_s_setprio(1);   // as in attention_bwd_kernel: the consumers
#elif FK_BWD_PRIO == 2
  if (wave < 4) __builtin_amdgcn_s_setprio(1);    // the producers
#endif

  for (;;) {   // one pass per (item, query-tile range); a plain launch makes exactly one
  int item = t0, tb = 0, te = nt;
  if constexpr (STREAMK) {
    if (round < p.sk_rounds) {
      item = round * (int)gridDim.x + t0;
      ++round;
    } else {
      if (u >= u_end) break;
      const int ti = (unsigned)(u_end - 1) / (unsigned)nt;
      item = p.sk_rounds * (int)gridDim.x + ti;
      te = u_end - ti * nt;
      tb = max(u - ti * nt, 0);
      u_end -= te - tb;
    }
  }
  const int rb = item % nrb;
  const int bh = item / nrb;
  const int b = bh / p.H, h = bh - b * p.H;

  // ---- stationary row operand: K rows for the producer, V rows for the consumer ----------------------------------------
  const int row = rb * 128 + pair * 32 + ql;
  const int rowc = min(row, p.S - 1);
  bf16x8_t xf[8];
  {
    const bf16_t* xb = producer ? p.k.p + (int64_t)b * p.k.bs + (int64_t)h * p.k.hs + (int64_t)rowc * p.k.ld
                                : p.v.p + (int64_t)b * p.v.bs + (int64_t)h * p.v.hs + (int64_t)rowc * p.v.ld;
    const uint32_t sgn = producer ? 0u : 0x80008000u;   // the consumer's V is NEGATED: its product delivers D - dP (below)
#pragma unroll
    for (int kk = 0; kk < 8; ++kk)
      xf[kk] = __builtin_bit_cast(bf16x8_t, __builtin_bit_cast(u32x4_t, *(const bf16x8_t*)(xb + 8 * hh + 16 * kk)) ^ sgn);
  }

  // ---- LDS-DMA of the query tiles: image 0 = Q, image 1 = dO, then lse and D of the 64 queries ------------------------
  const int prow = lane >> 4, pslot = lane & 15;
  auto rsrc_of = [&](const TView& t) {
    return make_dma_desc(t.p + (int64_t)b * t.bs + (int64_t)h * t.hs, ((int64_t)(p.S - 1) * t.ld + HD) * 2);
  };
  const DmaDesc rs_0 = rsrc_of(p.q), rs_1 = rsrc_of(p.dout);
  const DmaDesc rs_l = make_dma_desc(p.lse + (int64_t)bh * p.S, (int64_t)p.S * 4);
  const DmaDesc rs_d = make_dma_desc(p.dsum + (int64_t)bh * p.S, (int64_t)p.S * 4);
  auto voff = [&](const TView& t, int q, int r) { return (int)((q * t.ld + ((pslot ^ swz(r)) << 3)) * 2); };
  int v0[2], v1[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = (wave * 2 + i) * 4 + prow;
    v0[i] = voff(p.q, r, r);
    v1[i] = voff(p.dout, r, r);
  }
  auto issue_tile = [&](int t, int stage) {
    char* sb = smem + stage * STAGE_BYTES;
    const int base_row = t * CBLK;
    int a0 = v0[0], a1 = v0[1], c0 = v1[0], c1 = v1[1], sl = lane * 4;
    if (ragged && t == nt - 1) {   // queries beyond S: re-read the last valid one (the producer zeroes their p)
      const int last = p.S - 1 - base_row;
      const int r0 = (wave * 2) * 4 + prow, r1 = r0 + 4;
      const int q0 = min(r0, last), q1 = min(r1, last);
      a0 = voff(p.q, q0, r0);
      a1 = voff(p.q, q1, r1);
      c0 = voff(p.dout, q0, r0);
      c1 = voff(p.dout, q1, r1);
      sl = min(lane, last) * 4;
    }
    buffer_lds<16>(rs_0, sb + (wave * 2) * 1024, a0, (int)(base_row * p.q.ld * 2));
    buffer_lds<16>(rs_0, sb + (wave * 2 + 1) * 1024, a1, (int)(base_row * p.q.ld * 2));
    buffer_lds<16>(rs_1, sb + IMG + (wave * 2) * 1024, c0, (int)(base_row * p.dout.ld * 2));
    buffer_lds<16>(rs_1, sb + IMG + (wave * 2 + 1) * 1024, c1, (int)(base_row * p.dout.ld * 2));
    buffer_lds<4>(rs_l, sb + 2 * IMG, sl, base_row * 4);   // every wave writes the same 2 x 64 floats: uniform request counts
    buffer_lds<4>(rs_d, sb + 2 * IMG + 256, sl, base_row * 4);
  };

  // ---- operand reads (as in attention_bwd_kernel) ------------------------------------------------------------------------
  const int k_rd = ql * 256, k_sw = swz(ql);
  const int tj = (lane & 15) >> 2, tq = lane & 3, tdh = (lane >> 4) & 1;
  const int t_lo = (4 * hh + tj) * 256 + (((2 * tdh + (tq >> 1)) ^ hh) << 4) + (tq & 1) * 8;
  const int t_hi = (4 * hh + tj + 8) * 256 + (((2 * tdh + (tq >> 1)) ^ (hh + 2)) << 4) + (tq & 1) * 8;
  auto rowfrag = [&](const char* sb, int img, int kb, int kk) {
    return *(const bf16x8_t*)(sb + img * IMG + k_rd + kb * 8192 + (((2 * kk + hh) ^ k_sw) << 4));
  };
  auto trfrag = [&](const char* sb, int img, int st, int df) {
    const char* vp = sb + img * IMG + st * 4096 + ((df ^ tj) << 6);
    const s16x4_t lo = lds_tr16(vp + t_lo);
    const s16x4_t hi = lds_tr16(vp + t_hi);
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

  // one 32 x 32 tile product against the stationary rows, operand reads pinned two MFMAs ahead (attention_bwd_kernel)
  auto product = [&](f32x16_t& out, const char* sb, int img, int kb, const f32x16_t& init) {
    bf16x8_t kf[AH + 1];
#pragma unroll
    for (int a = 0; a < AH; ++a) kf[a] = rowfrag(sb, img, kb, a);
    __builtin_amdgcn_sched_group_barrier(0x100, AH, 0);
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      if (kk + AH < 8) kf[(kk + AH) % (AH + 1)] = rowfrag(sb, img, kb, kk + AH);
      out = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kk % (AH + 1)], xf[kk], kk == 0 ? init : out, 0, 0, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    }
  };
  // acc^T[d][row] += image^T w for the two 16-query steps of half kb
  auto accumulate = [&](const char* sb, int img, int kb, const bf16x8_t (&pf)[2]) {
    bf16x8_t tf[AH + 1];
#pragma unroll
    for (int a = 0; a < AH; ++a) tf[a] = trfrag(sb, img, 2 * kb, a);
    __builtin_amdgcn_sched_group_barrier(0x100, 2 * AH, 0);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int df = i & 3;
      if (i + AH < 8) tf[(i + AH) % (AH + 1)] = trfrag(sb, img, 2 * kb + ((i + AH) >> 2), (i + AH) & 3);
      acc[df] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tf[i % (AH + 1)], pf[i >> 2], acc[df], 0, 0, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    }
  };

  // ---- the producer's tile: S, exponentials, hand-over, dV, half by half --------------------------------------------------
  auto pack_store = [&](const f32x16_t& w, bf16x8_t (&pf)[2], char* dst) {
#pragma unroll
    for (int step = 0; step < 2; ++step) {
      u32x4_t pw;
#pragma unroll
      for (int e = 0; e < 4; ++e) pw[e] = pack_bf2(w[8 * step + 2 * e], w[8 * step + 2 * e + 1]);
      pf[step] = __builtin_bit_cast(bf16x8_t, pw);
      *(bf16x8_t*)(dst + step * 1024) = pf[step];
    }
  };
  // the plain order (half by half: S, exponentials, dV); with the mask tag the ragged last tile, p of the queries beyond S zeroed
  auto producer_plain = [&](const char* sb, char* pb, int t, auto mask_tag) {
    constexpr bool MASK = decltype(mask_tag)::value;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      f32x16_t s;
      product(s, sb, 0, kb, f32x16_t{});
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4_t lv = *(const f32x4_t*)(sb + 2 * IMG + (32 * kb + 8 * g + 4 * hh) * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int r = 4 * g + j;
          float pr = __builtin_amdgcn_exp2f(fmaf(s[r], p.scale_log2, -lv[j]));
          if constexpr (MASK)
            if (t * CBLK + 32 * kb + 8 * g + 4 * hh + j >= p.S) pr = 0.f;
          s[r] = pr;
        }
      }
      bf16x8_t pf[2];
      pack_store(s, pf, pb + kb * 2048);
      accumulate(sb, 1, kb, pf);
    }
  };
  // ---- the consumer's tile: dP, w from the producer's p, dK, half by half ----------------------------------------------------
  auto consumer_plain = [&](const char* sb, const char* pb) {
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      f32x16_t dp;   // starts at D of the 32 queries, accumulates -dO V^T on top: D - dP
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4_t dv = *(const f32x4_t*)(sb + 2 * IMG + 256 + (32 * kb + 8 * g + 4 * hh) * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) dp[4 * g + j] = dv[j];
      }
      __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
      product(dp, sb, 1, kb, dp);
      bf16x8_t wfs[2];
#pragma unroll
      for (int step = 0; step < 2; ++step) {
        const u32x4_t pp = __builtin_bit_cast(u32x4_t, *(const bf16x8_t*)(pb + (2 * kb + step) * 1024));
        u32x4_t pw;
#pragma unroll
        for (int e = 0; e < 4; ++e) {   // = -p (dP - D); sign and softmax scale multiply dK at the store
          const int r = 8 * step + 2 * e;
          pw[e] = pack_bf2(bf_lo(pp[e]) * dp[r], bf_hi(pp[e]) * dp[r + 1]);
        }
        wfs[step] = __builtin_bit_cast(bf16x8_t, pw);
      }
      accumulate(sb, 0, kb, wfs);
    }
  };

  char* const pbuf = smem + STAGES * STAGE_BYTES + pair * 8192 + lane * 16;
  issue_tile(tb, 0);

  // Iteration i = tb .. te of BOTH roles: wait for the own DMA pieces of tile i and the own p stores of tile i - 1, meet,
  // start the DMA of tile i + 1 into the stage the consumers left in iteration i - 1.  The roles run separate loops (one
  // accumulator live range each, no copies where they would join); the barrier counts arrivals, not program counters.
  int st_pro = 0, st_con = STAGES - 1, st_dma = 1;   // stages of tiles i, i - 1, i + 1
  auto meet = [&](int i) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (i + 1 < te) issue_tile(i + 1, st_dma);
  };
  auto rotate = [&]() {
    st_con = st_pro;
    st_pro = st_dma;
    st_dma = (st_dma == STAGES - 1) ? 0 : st_dma + 1;
  };
  const bool masked_last = ragged && te == nt;       // the ragged tile is the item's last one
  if (producer) {
    const int n_plain = masked_last ? te - 1 : te;
    for (int i = tb; i < n_plain; ++i) {
      meet(i);
      producer_plain(smem + st_pro * STAGE_BYTES, pbuf + ((i - tb) & 1) * 4096, i, std::false_type{});
      rotate();
    }
    if (masked_last) {
      meet(nt - 1);
      producer_plain(smem + st_pro * STAGE_BYTES, pbuf + ((nt - 1 - tb) & 1) * 4096, nt - 1, std::true_type{});
      rotate();
    }
    meet(te);
  } else {
    meet(tb);
    rotate();
    for (int i = tb + 1; i <= te; ++i) {
      meet(i);
      consumer_plain(smem + st_con * STAGE_BYTES, pbuf + ((i - 1 - tb) & 1) * 4096);
      rotate();
    }
  }

  // ---- the seam of a cut item (attention_bwd_kernel has the protocol): the part that finishes first leaves its accumulators
  // in the cut's slot, the second adds them to its own and stores the item
  if constexpr (STREAMK) {
    if (tb > 0 || te < nt) {                        // workgroup-uniform
      typedef __attribute__((address_space(1))) unsigned gu32;
      const int slot = te < nt ? t0 + 1 : t0;
      gu32* const ctl = (gu32*)(p.sk_ctl + 2 * (size_t)slot);
      const __amdgpu_buffer_rsrc_t rs_p = __builtin_amdgcn_make_buffer_rsrc(
          (void*)(p.sk_partials + (size_t)slot * PART_FLOATS), 0, PART_FLOATS * 4, 0x00020000);
      __syncthreads();   // every wave is done with the ring and the hand-over buffer: the first word now carries the ticket
      if (tid == 0) *(volatile unsigned*)smem = __hip_atomic_fetch_add(ctl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __syncthreads();
      const unsigned ticket = __builtin_amdgcn_readfirstlane(*(volatile unsigned*)smem);
      if ((ticket & 1u) == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const f32x16_t& a = acc[r >> 2];
          const int q4 = r & 3;
          const u32x4_t v = {__float_as_uint(a[4 * q4]), __float_as_uint(a[4 * q4 + 1]), __float_as_uint(a[4 * q4 + 2]),
                             __float_as_uint(a[4 * q4 + 3])};
          __builtin_amdgcn_raw_buffer_store_b128(v, rs_p, tid * 16, r * (512 * 16), /*sc1: write through*/ 16);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_store(ctl + 1, ticket + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        continue;
      }
      if (tid == 0) {
        int spins = 0;    // bounded: a corrupted workspace ends in NaN rows, not in a hung device
        while (__hip_atomic_load(ctl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != ticket && spins < (1 << 22)) {
          __builtin_amdgcn_s_sleep(8);
          ++spins;
        }
        *(volatile unsigned*)smem = spins >= (1 << 22);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      }
      __syncthreads();
      const bool gave_up = *(volatile unsigned*)smem != 0;
#pragma unroll
      for (int r0 = 0; r0 < 16; r0 += 4) {
        u32x4_t v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = __builtin_amdgcn_raw_buffer_load_b128(rs_p, tid * 16, (r0 + e) * (512 * 16), /*sc1*/ 16);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          f32x16_t& a = acc[(r0 + e) >> 2];
          const int q4 = (r0 + e) & 3;
#pragma unroll
          for (int j = 0; j < 4; ++j) a[4 * q4 + j] += __uint_as_float(v[e][j]);   // fp32 addition commutes: symmetric
        }
      }
      if (gave_up)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][r] = __builtin_nanf("");
    }
  }

  // ---- store: lane (row = ql) holds d = 32 df + 8 g + 4 hh + (0..3); producers hold dV, consumers dK -----------------
  if (row < p.S) {
    bf16_t* op = producer ? p.out2 + (int64_t)b * p.o2_bs + (int64_t)h * p.o2_hs + (int64_t)row * p.o2_ld
                          : p.out + (int64_t)b * p.o_bs + (int64_t)h * p.o_hs + (int64_t)row * p.o_ld;
    op += 4 * hh;
    const float osc = producer ? 1.f : -p.scale;   // dK: the sign and the softmax scale left out of w
#pragma unroll
    for (int df = 0; df < 4; ++df)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        u32x2_t pk;
        pk[0] = pack_bf2(acc[df][4 * g + 0] * osc, acc[df][4 * g + 1] * osc);
        pk[1] = pack_bf2(acc[df][4 * g + 2] * osc, acc[df][4 * g + 3] * osc);
        *(u32x2_t*)(op + 32 * df + 8 * g) = pk;
      }
  }
  if constexpr (!STREAMK) break;
  else __syncthreads();   // every wave is done with the ring, the hand-over buffer and the ticket word
  }   // passes
}

template <int MODE, bool STREAMK = false>
int launch_bwd(const BwdParams& p, hipStream_t stream, int grid = 0) {
  constexpr int SMEM = STAGES * STAGE_BYTES;
  auto kern = attention_bwd_kernel<MODE, STREAMK>;
  FK_ENSURE_MAX_LDS(kern, SMEM, "fk_attention_bwd_bf16");
  const int nrb = (p.S + 255) / 256;
  hipLaunchKernelGGL(kern, dim3(STREAMK ? grid : nrb * p.H * p.B), dim3(512), SMEM, stream, p);
  FK_CHECK_LAUNCH("fk_attention_bwd_bf16");
  return FK_OK;
}

int bwd_cu_count() {
  static int cus = 0;
  if (!cus) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    else return 256;
  }
  return cus;
}
constexpr int BWD_MIN_PART = 8;
constexpr int BWD_CTL_BYTES = 16384;   // = ATTN_CTL_BYTES of attention_fwd.hip

template <bool STREAMK>
int launch_dkv(const BwdParams& p, hipStream_t stream, int grid = 0) {
  constexpr int SMEM = STAGES * STAGE_BYTES + PBUF_BYTES;
  auto kern = attention_bwd_dkv_kernel<STREAMK>;
  FK_ENSURE_MAX_LDS(kern, SMEM, "fk_attention_bwd_bf16");
  const int nrb = (p.S + 127) / 128;
  hipLaunchKernelGGL(kern, dim3(STREAMK ? grid : nrb * p.H * p.B), dim3(512), SMEM, stream, p);
  FK_CHECK_LAUNCH("fk_attention_bwd_bf16");
  return FK_OK;
}

bool view_ok(const fk_attn_view& v, int64_t S) {
  return v.p && ((uintptr_t)v.p % 16 == 0) && v.ld > 0 && v.ld % 8 == 0 && v.head_stride % 8 == 0 && v.batch_stride % 8 == 0 &&
         ((S - 1) * v.ld + HD) * 2 < (1ll << 31) && (S + CBLK) * v.ld * 2 < (1ll << 31);
}
TView tv(const fk_attn_view& v) { return TView{(const bf16_t*)v.p, v.ld, v.head_stride, v.batch_stride}; }

}  // namespace

static int attention_bwd_entry(const fk_attn_view* q, const fk_attn_view* k, const fk_attn_view* v,
                               const fk_attn_view* dout, const float* lse, const float* dsum, const fk_attn_view* dq,
                               const fk_attn_view* dk, const fk_attn_view* dv, int32_t B, int32_t H, int32_t S,
                               float scale, void* ws, int64_t ws_bytes, int grid, int passes, fk_stream_t stream_) {
  FK_CHECK_ARG(q && k && v && dout && lse && dsum && dq && dk && dv, "fk_attention_bwd_bf16: null pointer");
  FK_CHECK_ARG(grid >= -1 && grid != 1, "fk_attention_bwd_ws_bf16: grid %d is not 0 (default), -1 (plain grid) or a workgroup count >= 2", grid);
  FK_CHECK_ARG(passes == 0 || passes == 2 || passes == 3, "fk_attention_bwd_ws_bf16: passes %d is not 0 / 2 (dQ pass + paired dK / dV "
               "pass) or 3 (three passes)", passes);
  FK_CHECK_ARG(B > 0 && H > 0 && S > 0, "fk_attention_bwd_bf16: bad B/H/S %d %d %d", B, H, S);
  const fk_attn_view* all[7] = {q, k, v, dout, dq, dk, dv};
  for (const fk_attn_view* a : all)
    FK_CHECK_ARG(view_ok(*a, S), "fk_attention_bwd_bf16: every view needs a 16-byte aligned pointer, strides that are multiples "
                                 "of 8 elements and a (batch, head) extent below 2 GiB");
  BwdParams p{};
  p.q = tv(*q); p.k = tv(*k); p.v = tv(*v); p.dout = tv(*dout);
  p.lse = lse; p.dsum = dsum;
  p.B = B; p.H = H; p.S = S;
  p.scale = scale; p.scale_log2 = scale * 1.4426950408889634f;
  hipStream_t stream = (hipStream_t)stream_;
  auto set_out = [&](const fk_attn_view& o) {
    p.out = (bf16_t*)o.p; p.o_ld = o.ld; p.o_hs = o.head_stride; p.o_bs = o.batch_stride;
  };
  set_out(*dq);
  // Stream-K grid where one workgroup per item would waste >= 4 % of its rounds of CUs (attention_fwd.hip); items of `rows` rows
  const int mode = grid == 0 ? 1 : (grid < 0 ? 0 : grid);
  const int G = mode >= 2 ? (mode < bwd_cu_count() ? mode : bwd_cu_count()) : bwd_cu_count();
  const int64_t nt = (S + CBLK - 1) / CBLK;
  auto stream_k = [&](int rows) {      // fills p.n_items / sk_rounds / ... and says whether the persistent grid is to be used
    const int64_t n_items = (int64_t)((S + rows - 1) / rows) * H * B;
    const int64_t rounds = (n_items + G - 1) / G;
    const bool wasteful = mode >= 2 || (n_items > G && (rounds * G - n_items) * 25 >= rounds * G);
    int sk_rounds = (int)(n_items / G) - 1;
    while (sk_rounds >= 0 && (n_items - (int64_t)sk_rounds * G) * nt < (int64_t)G * (nt + 2 * BWD_MIN_PART)) --sk_rounds;
    const int64_t need = BWD_CTL_BYTES + (int64_t)G * PART_FLOATS * 4;
    if (!(mode && wasteful && sk_rounds >= 0 && ws && ws_bytes >= need && G <= BWD_CTL_BYTES / 8 && n_items * nt < (1ll << 31)))
      return false;
    p.n_items = (int)n_items; p.sk_rounds = sk_rounds; p.min_part = BWD_MIN_PART;
    p.sk_ctl = (unsigned*)ws;                                   // the forward's layout: control words first (16 KiB)
    p.sk_partials = (float*)((char*)ws + BWD_CTL_BYTES);
    return true;
  };
  FK_CHECK_ARG(!ws || (uintptr_t)ws % 16 == 0, "fk_attention_bwd_ws_bf16: workspace must be 16-byte aligned");
  int rc = stream_k(256) ? launch_bwd<MODE_DQ, true>(p, stream, G) : launch_bwd<MODE_DQ>(p, stream);
  if (rc != FK_OK) return rc;
  if (passes != 3) {
    set_out(*dk);
    p.out2 = (bf16_t*)dv->p; p.o2_ld = dv->ld; p.o2_hs = dv->head_stride; p.o2_bs = dv->batch_stride;
    // The paired pass keeps the plain grid by default: on the persistent grid (same workspace -- the two launches are ordered
    // on the stream, tickets are monotonic per cut) the seams' 128 KiB partials cost what the 9 % of idle CU time would return
    // (S = 8704: 2.91 against 2.89-2.91 ms per call; S = 2560, 1.9 rounds: 0.276 against 0.264;
    // profiles/r05_attention_bwd_dkv_streamk_ab.txt).  A forced grid (>= 2 workgroups, the test hook) takes it.
    return mode >= 2 && stream_k(128) ? launch_dkv<true>(p, stream, G) : launch_dkv<false>(p, stream);
  }
  set_out(*dv);
  rc = launch_bwd<MODE_DV>(p, stream);
  if (rc != FK_OK) return rc;
  set_out(*dk);
  return launch_bwd<MODE_DK>(p, stream);
}

extern "C" int fk_attention_bwd_bf16(const fk_attn_view* q, const fk_attn_view* k, const fk_attn_view* v,
                                     const fk_attn_view* dout, const float* lse, const float* dsum, const fk_attn_view* dq,
                                     const fk_attn_view* dk, const fk_attn_view* dv, int32_t B, int32_t H, int32_t S,
                                     float scale, fk_stream_t stream_) {
  return attention_bwd_entry(q, k, v, dout, lse, dsum, dq, dk, dv, B, H, S, scale, nullptr, 0, 0, 0, stream_);
}

extern "C" int fk_attention_bwd_ws_bf16(const fk_attn_view* q, const fk_attn_view* k, const fk_attn_view* v,
                                        const fk_attn_view* dout, const float* lse, const float* dsum, const fk_attn_view* dq,
                                        const fk_attn_view* dk, const fk_attn_view* dv, int32_t B, int32_t H, int32_t S,
                                        float scale, void* ws, int64_t ws_bytes, int32_t grid, int32_t passes, fk_stream_t stream_) {
  return attention_bwd_entry(q, k, v, dout, lse, dsum, dq, dk, dv, B, H, S, scale, ws, ws_bytes, grid, passes, stream_);
}
