// Large-tile bf16 MFMA GEMMs for the MMDiT linears (same contract as gemm_bf16.hip, used when M >= 192).
// v_mfma_f32_32x32x16_bf16, operands swapped like gemm_bf16.hip (W rows -> MFMA A operand) so a lane owns 4
// consecutive output columns of one row.  Two tile shapes, chosen per problem by (tile-quantisation efficiency) x
// (measured steady-state rate); both accumulate over K in the same order, so they agree bit for bit:
//
//   gemm8_kernel   256 x 256 x 64, 8 waves (2 x 4, 128 x 64 per wave), 8 half-tile LDS slots = 128 KiB
//   gemm9_kernel   256 x 128 x 64, 8 waves (4 x 2,  64 x 64 per wave), 3 stages of 48 KiB
//
// Both run their 8 waves as TWO GROUPS (one wave of each per SIMD) that alternate between an MFMA phase (8 MFMAs =
// 256 matrix-pipe cycles) and a load phase (LDS fragment reads for the next phase + LDS-DMA requests for tiles
// 3-6 phases ahead), two raw s_barriers per phase, the groups one barrier apart ("ping-pong").
//
// What the measurements say (rounds 1 and 2; DESIGN.md section 4, profiles/r02_gemm_variants.txt):
//  * The chip is POWER-bound under MFMA load: 1.27-1.45 GHz instead of 2.4.  What a kernel can win is (matrix pipe
//    busy) x (clock its energy per flop leaves).  gemm8_kernel keeps the pipe 89 % busy (the vendor library's
//    hand-written kernel: 89 %) but clocks 1.27 GHz against 1.42: 0.75 LDS fragment reads per MFMA against 0.5 for
//    128 x 128 wave tiles.  gemm9_kernel (1.0 reads per MFMA): 74 % busy at 1.30 GHz.
//  * An LDS-DMA request costs its wave >= 60 ISSUE cycles, longer than the 28-cycle shadow of an MFMA: with one wave
//    per SIMD (round 1's 4-wave 256 x 256 kernels: 76 % busy) the pipe drains behind every piece; with all 8 waves in
//    lockstep (round 1's 256 x 128 kernel) 69 % busy; with two groups in opposite phases the requests are free.
//  * `global_load_lds` is a FLAT-class instruction: while one is pending hipcc turns every `lgkmcnt(N)` wait into
//    `lgkmcnt(0)`.  The MUBUF form (`buffer_load_dwordx4 ... lds`) keeps the counts exact.
//  * One barrier per phase instead of two (-1.5 %) and dropping s_setprio around the MFMA cluster (+-0) were measured
//    on gemm8_kernel and not kept.
//
// The LDS image of a tile row is 128 B (64 k); the bank-conflict swizzle (16-byte chunk ^ ((row >> 1) & 7)) is
// applied where the tile is written (DMA source address) and again on the ds_read_b128 side.
// Up to FK_MAX_GROUP problems with identical (N, K, epilogue) share one launch ("grouped GEMM"): the
// text- and image-stream linears of a double block become one grid.
#include <stdlib.h>

#include <type_traits>

#include "fk_common.h"

namespace {

constexpr int BM = 256;
#ifndef FK_GROUP_M
#define FK_GROUP_M 8
#endif
constexpr int GROUP_M = FK_GROUP_M;
// steady-state rate of the 256 x 256 kernel relative to the 256 x 128 one (measured: 1.23-1.27 on full grids)
#ifndef FK_RATE_256
#define FK_RATE_256 1.22
#endif


struct GroupArgs {
  fk_gemm_args p[FK_MAX_GROUP];
  int tiles_before[FK_MAX_GROUP + 1];  // prefix sums of tile counts (mixed launch: of the 256 x 256 tiles)
  int n;
  // mixed launch (gemm_mix_kernel): column tiles [0, big_cols) of width 256 are 256 x 256 tiles, the columns from
  // big_cols * 256 on are 256 x 128 tiles; per XCD (blockIdx % 8) the chunk of each class it works off
  int big_cols;
  int small_before[FK_MAX_GROUP + 1];
  int xcd_big_start[8], xcd_big_cnt[8], xcd_small_start[8];
  // split-K launch (gemm8_kernel<.., SPLITK>): two workgroups per 256 x 256 tile, each over half of K; fp32 partial
  // tiles (256 KiB apiece) and one (ticket, flag) word pair per tile in a caller-owned workspace
  float* sk_partials;
  unsigned* sk_ctl;
  int sk_min_part;     // stream-K launch (gemm8_streamk_kernel): the shortest part (in K-tiles) a cut may leave
  int group_m;         // depth (in row tiles) of the grouped tile order; >= the row-tile count: every XCD owns a column range
};

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;

typedef short s16x4_t __attribute__((ext_vector_type(4)));
FK_DEV s16x4_t lds_tr16(const char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p);
}

// buffer form of the LDS-DMA load.  The builtin exists for the device target only; seen by the host pass it silently
// suppresses the kernel's host stub.
FK_DEV void buffer_lds16(__amdgpu_buffer_rsrc_t rsrc, char* lds_dst, int voffset, int soffset) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)lds_dst, 16, voffset, soffset, 0, 0);
#else
  (void)rsrc; (void)lds_dst; (void)voffset; (void)soffset;
#endif
}

// K-major operand paths only: the request as an opaque statement (fk_common.h).  Their transpose reads are builtins on LDS
// pointers hipcc has no alias information for, so behind the builtin request it waited for ALL outstanding requests in front of
// every group's first transpose read (s_waitcnt vmcnt(0) beside the kernel's own counted wait, three times per K-tile pair);
// the row-major default forms never had that wait and keep the builtin.
FK_DEV void buffer_lds16(const BufDesc& d, char* lds_dst, int voffset, int soffset) {
  buffer_lds_opaque<16>(d, lds_addr_of(lds_dst), voffset, soffset);
}

template <int N>
FK_DEV void wait_vmcnt() {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
  else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else static_assert(N == 0, "add the vmcnt literal");
}

// fk_rows addressing of the (at most BM) rows of one tile without per-row 64-bit divisions: one division per tile
// for the tile's first row, then a compare per row (a 32-bit division when a batch has fewer rows than a tile).
// The launcher guarantees that every offset relative to the tile's first row fits 31 bits (in bytes).
struct TileRows {
  int ld, wrap, rpb, b0, r0;
  float inv;    // 1 / rpb when a batch is shorter than a tile
  bool big;     // rpb >= BM (or no batching): a tile crosses at most one batch boundary
  FK_DEV TileRows(const fk_rows& r, int m0) {
    ld = (int)r.ld;
    if (r.rows_per_batch <= 0) { rpb = 0x7fffffff; wrap = 0; b0 = 0; r0 = 0; big = true; inv = 0.f; }
    else {
      rpb = (int)(r.rows_per_batch > 0x7fffffff ? 0x7fffffff : r.rows_per_batch);
      b0 = (unsigned)m0 / (unsigned)rpb;
      r0 = m0 - b0 * rpb;
      wrap = (int)(r.batch_stride - (int64_t)rpb * r.ld);
      big = rpb >= BM;
      inv = 1.0f / (float)rpb;
    }
  }
  // batches crossed between the tile's first row and its row ml (branch-free: r0 + ml < rpb + BM, so for short
  // batches the quotient is a small integer that the float product resolves exactly)
  FK_DEV int crossed(int ml) const {
    const int r = r0 + ml;
    const int one = r >= rpb ? 1 : 0;
    const int many = (int)(((float)r + 0.5f) * inv);
    return big ? one : many;
  }
  // element offset of row ml of the tile relative to its first row
  FK_DEV int off(int ml) const { return ml * ld + crossed(ml) * wrap; }
};

// sum over the 16 lanes of a DPP row, every lane receiving the total: four row-rotate adds on the VALU (no LDS
// round trips).  Bit-identical to the xor butterfly 8, 4, 2, 1: after the step of distance d the partial sums have
// period d within the row, so "rotate by d" and "xor d" name the same partner value.
template <int N>
FK_DEV float row_ror(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x120 + N, 0xf, 0xf, false));
}
FK_DEV float row16_sum(float v) {
  v += row_ror<8>(v);
  v += row_ror<4>(v);
  v += row_ror<2>(v);
  v += row_ror<1>(v);
  return v;
}

// tile selection: XCD chunking over the whole grid (workgroup b runs on XCD b % 8: consecutive tiles of the order
// below -- which share A rows and W columns -- stay on one XCD's L2), then problem, then grouped (GROUP_M deep) order
FK_DEV int xcd_chunk_index() {
  const int nwg = gridDim.x;
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}
// tile t of a class of tiles that covers the column tiles [col0 / BN, col0 / BN + nbn) of every problem
template <int BN>
FK_DEV void tile_of(const GroupArgs& ga, const int (&before)[FK_MAX_GROUP + 1], int t, int nbn, int col0, int& pi, int& m0,
                    int& n0) {
  pi = 0;
#pragma unroll
  for (int i = 1; i < FK_MAX_GROUP; ++i)
    if (i < ga.n && t >= before[i]) pi = i;
  const fk_gemm_args& p = ga.p[pi];
  t -= before[pi];
  const int nbm = (p.M + BM - 1) / BM;
  const int per_group = ga.group_m * nbn;
  const int g = t / per_group;
  const int first_m = g * ga.group_m;
  const int gm = min(nbm - first_m, ga.group_m);
  const int rem = t - g * per_group;
  m0 = (first_m + rem % gm) * BM;
  n0 = col0 + (rem / gm) * BN;
}
template <int BN>
FK_DEV void select_tile(const GroupArgs& ga, int t, int& pi, int& m0, int& n0) {
  tile_of<BN>(ga, ga.tiles_before, t, (ga.p[0].N + BN - 1) / BN, 0, pi, m0, n0);
}

// ---- MFMA shape ------------------------------------------------------------------------------------------------------
// M16 = false: v_mfma_f32_32x32x16_bf16 (8 passes, 16 k per instruction); M16 = true: v_mfma_f32_16x16x32_bf16 (4 passes, 32 k
// per instruction) on the SAME LDS image, the same accumulator registers and the same number of ds_read_b128 per K-tile.
// Why both exist (tools/power_probe.hip, profiles/r05_power_probe.txt): under the chip's power limit a pure MFMA stream of
// the 16 x 16 x 32 form sustains 2 017 TF/s at 1.97 GHz against 1 800 TF/s at 1.76 GHz for the 32 x 32 x 16 form -- half the
// accumulator register traffic per flop -- and the GEMM main loops are power-bound.  A 32 x 32 accumulator block (nf, mf) of
// the epilogue holds, as quad q = 2 * n16 + m16 (4 registers = 4 consecutive output columns of one row), the 16 x 16 block
// (n16, m16) in the M16 form and columns 8 q + 4 (lane >> 5) of row (lane & 31) in the other.
template <bool M16>
struct FragMap {
  static FK_DEV int row(int lane, int q) { return M16 ? (q & 1) * 16 + (lane & 15) : (lane & 31); }
  static FK_DEV int col(int lane, int q) { return M16 ? (q >> 1) * 16 + (lane >> 4) * 4 : 8 * q + 4 * (lane >> 5); }
};
FK_DEV f32x4_t quad_get(const f32x16_t& v, int q) { return f32x4_t{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]}; }
FK_DEV void quad_set(f32x16_t& v, int q, const f32x4_t& c) {
  v[4 * q] = c[0]; v[4 * q + 1] = c[1]; v[4 * q + 2] = c[2]; v[4 * q + 3] = c[3];
}
// one 32 (n) x 32 (m) x 32 (k) step on a 32 x 32 accumulator block as four 16 x 16 x 32 MFMAs: w16[n16], a16[m16]
FK_DEV void mma16_block(f32x16_t& blk, const bf16x8_t& w0, const bf16x8_t& w1, const bf16x8_t& a0, const bf16x8_t& a1) {
  quad_set(blk, 0, __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, a0, quad_get(blk, 0), 0, 0, 0));
  quad_set(blk, 2, __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, a0, quad_get(blk, 2), 0, 0, 0));
  quad_set(blk, 1, __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, a1, quad_get(blk, 1), 0, 0, 0));
  quad_set(blk, 3, __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, a1, quad_get(blk, 3), 0, 0, 0));
}

// epilogue (as gemm_bf16.hip): bias/activation -> bf16 -> LDS tile -> coalesced 16-byte rows.
// acc[nf][mf] is the 32x32 block (n-block nf, m-block mf) of wave (wm, wn) in MFMA-output layout with the
// swapped operands: lane l holds row (l & 31) of the m-block, columns 8*q + 4*(l >> 5) + j of the n-block.
//
// With one workgroup per CU nothing overlaps the epilogue, so it must not be a chain of load -> wait -> use steps
// (measured: a bias load in front of every quad and a residual / gate / rope-table load in front of every stored
// chunk made the epilogue ~15 % of a K = 3072 tile).  All global reads are therefore issued unconditionally with
// clamped indices, a batch of EPI_BATCH chunks at a time, ahead of the arithmetic of the batch; only the final
// store is predicated.
template <int EPI, int BN, class C>
FK_DEV void store_tile(const f32x16_t (&acc)[C::NF][C::MF], const fk_gemm_args& p, char* smem, int m0, int n0,
                       int wm, int wn) {
  int tid = threadIdx.x;
  // gemm10 (the only 256-thread caller) may run this inside a per-CU tile loop around an asm statement that leaves 36 free
  // VGPRs: an opaque copy keeps hipcc from hoisting the 32 per-lane row indices below out of that loop and spilling them
  if constexpr (C::NTHREADS == 256) asm volatile("" : "+v"(tid));
  const int lane = tid & 63;
  using FM = FragMap<C::M16>;
  // bias of this lane's 4-column quads (all loads in flight before the barrier below)
  u32x2_t bw[C::NF][4];
#pragma unroll
  for (int nf = 0; nf < C::NF; ++nf)
#pragma unroll
    for (int q = 0; q < 4; ++q) bw[nf][q] = u32x2_t{0u, 0u};
  if constexpr (EPI != FK_EPI_SCALE) {
    if (p.bias) {
#pragma unroll
      for (int nf = 0; nf < C::NF; ++nf)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = n0 + C::tile_col(wn, nf) + FM::col(lane, q);
          bw[nf][q] = *(const u32x2_t*)((const bf16_t*)p.bias + min(n, p.N - 4));   // columns >= N are never stored
        }
    }
  }
  if constexpr (EPI == FK_EPI_F32DBG) {
    // parity build: fp32(acc + bias) straight from the accumulator registers (a lane owns 4 consecutive columns of a row)
#pragma unroll
    for (int nf = 0; nf < C::NF; ++nf)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = n0 + C::tile_col(wn, nf) + FM::col(lane, q);
        const float b[4] = {bf_lo(bw[nf][q][0]), bf_hi(bw[nf][q][0]), bf_lo(bw[nf][q][1]), bf_hi(bw[nf][q][1])};
#pragma unroll
        for (int mf = 0; mf < C::MF; ++mf) {
          const int m = m0 + C::tile_row(wm, mf) + FM::row(lane, q);
          if (m < p.M && n < p.N)
            *(f32x4_t*)((float*)p.C + fk_row_offset(p.c, m) + n) =
                f32x4_t{acc[nf][mf][4 * q + 0] + b[0], acc[nf][mf][4 * q + 1] + b[1], acc[nf][mf][4 * q + 2] + b[2],
                        acc[nf][mf][4 * q + 3] + b[3]};
        }
      }
    return;
  }
  __syncthreads();  // every wave is done reading the last stage before the C tile aliases it
  bf16_t* ct = (bf16_t*)smem;
#pragma unroll
  for (int nf = 0; nf < C::NF; ++nf) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int nl = C::tile_col(wn, nf) + FM::col(lane, q);
      const float b[4] = {bf_lo(bw[nf][q][0]), bf_hi(bw[nf][q][0]), bf_lo(bw[nf][q][1]), bf_hi(bw[nf][q][1])};
#pragma unroll
      for (int mf = 0; mf < C::MF; ++mf) {
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float x = acc[nf][mf][q * 4 + j];
          if constexpr (EPI == FK_EPI_SCALE) v[j] = x * p.alpha;
          else v[j] = x + b[j];
        }
        if constexpr (EPI == FK_EPI_GELU_TANH || EPI == FK_EPI_SILU) {
          // the reference graph rounds the Linear's output to bf16 before the activation
          round_bf_pair(v[0], v[1]);
          round_bf_pair(v[2], v[3]);
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = EPI == FK_EPI_GELU_TANH ? gelu_tanh_f(v[j]) : silu_f(v[j]);
        }
        u32x2_t pk;
        pk[0] = pack_bf2(v[0], v[1]);
        pk[1] = pack_bf2(v[2], v[3]);
        const int ml = C::tile_row(wm, mf) + FM::row(lane, q);
        *(u32x2_t*)(ct + ml * C::CT_LD + nl) = pk;
      }
    }
  }
  __syncthreads();
  constexpr int CPR = BN / 8;  // 16-byte chunks per tile row
  constexpr int ITERS = BM * CPR / C::NTHREADS;
  // chunks per batch; the fused QKV epilogue keeps 16 table registers per chunk, so the 4-wave kernel (32 chunks per
  // thread, VGPRs full) batches 4 and the 8-wave kernel takes all 8 of a thread's chunks at once
  constexpr int U = (EPI == FK_EPI_QKV && ITERS > 8) ? 4 : 8;
  static_assert(ITERS % U == 0 && C::NTHREADS % CPR == 0, "epilogue batching");
  // per-tile (scalar) row addressing of the output, the residual and the gate
  const TileRows crow(p.c, m0);
  bf16_t* const cbase = (bf16_t*)p.C + fk_row_offset(p.c, m0);
  const TileRows rrow = (EPI == FK_EPI_GATE_RES || EPI == FK_EPI_RES) ? TileRows(p.r, m0) : crow;
  const bf16_t* const rbase =
      (EPI == FK_EPI_GATE_RES || EPI == FK_EPI_RES) ? (const bf16_t*)p.res + fk_row_offset(p.r, m0) : nullptr;
  fk_rows gr = {0, 0, 0};
  if constexpr (EPI == FK_EPI_GATE_RES) gr.rows_per_batch = p.gate_rows_per_batch;
  const TileRows grow(gr, m0);   // b0 + crossed(ml) = the gate row of tile row ml
  const int D = EPI == FK_EPI_QKV ? p.qkv_heads * 128 : 1;
  const int which = EPI == FK_EPI_QKV ? n0 / D : 0;   // 0 = q, 1 = k, 2 = v: a tile never straddles q | k | v
  // a thread keeps its chunk column over the iterations (NTHREADS % CPR == 0): only the row advances
  const int cc = tid % CPR, ml0 = tid / CPR;
  constexpr int ML_STEP = C::NTHREADS / CPR;
  const int n = n0 + cc * 8;
  const int nc = min(n, p.N - 8);          // clamped column for the unconditional loads
  const int mlast = p.M - 1 - m0;          // last valid tile row

  if (EPI == FK_EPI_QKV && which < 2) {
    // 16 consecutive lanes hold one 128-wide head row of the tile: RMSNorm (weight) + interleaved-pair RoPE, then
    // the head-major [B, H, S_total, 128] layout
    const int hn = n - which * D;              // column inside q or k
    const int head = hn >> 7, dch = hn & 127;  // dch = 8 * chunk-in-head
    const u32x4_t ww = *(const u32x4_t*)((const bf16_t*)(which == 0 ? p.wq : p.wk) + dch);
    bf16_t* const dst = (bf16_t*)(which == 0 ? p.q_out : p.k_out);
#pragma unroll
    for (int j0 = 0; j0 < ITERS; j0 += U) {
      u32x4_t y[U];
      f32x4_t t0[U], t1[U];   // (cos, sin) of the chunk's four rotary pairs
      int srow[U], bidx[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int ml = ml0 + ML_STEP * (j0 + u);
        y[u] = *(const u32x4_t*)(ct + ml * C::CT_LD + cc * 8);
        // token of row m: batch = m / rows_per_batch (one batch when c.rows_per_batch <= 0), s = s_offset + m % rows_per_batch
        const int mlc = min(ml, mlast);
        const int nb = crow.crossed(mlc);
        bidx[u] = crow.b0 + nb;
        srow[u] = p.qkv_s_offset + crow.r0 + mlc - nb * crow.rpb;
        const float* tp = p.rope_cs + (int64_t)srow[u] * 128 + dch;   // pair dch/2 + e at floats 2e, 2e + 1
        t0[u] = *(const f32x4_t*)tp;
        t1[u] = *(const f32x4_t*)(tp + 4);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int ml = ml0 + ML_STEP * (j0 + u);
        float xv[8];
        float ss = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          xv[2 * e] = bf_lo(y[u][e]);
          xv[2 * e + 1] = bf_hi(y[u][e]);
          ss += xv[2 * e] * xv[2 * e] + xv[2 * e + 1] * xv[2 * e + 1];
        }
        ss = row16_sum(ss);
        const float rs = __builtin_amdgcn_rsqf(ss * (1.0f / 128) + 1e-6f);   // argument >= 1e-6: no denormal scaling
        const float cs[4] = {t0[u][0], t0[u][2], t1[u][0], t1[u][2]};
        const float sn[4] = {t0[u][1], t0[u][3], t1[u][1], t1[u][3]};
        u32x4_t ow;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float re = xv[2 * e] * rs, im = xv[2 * e + 1] * rs;
          round_bf_pair(re, im);
          re *= bf_lo(ww[e]);
          im *= bf_hi(ww[e]);
          round_bf_pair(re, im);
          const float o0 = __fadd_rn(__fmul_rn(re, cs[e]), __fmul_rn(-im, sn[e]));
          const float o1 = __fadd_rn(__fmul_rn(im, cs[e]), __fmul_rn(re, sn[e]));
          ow[e] = pack_bf2(o0, o1);
        }
        // head-major row index in 32 bits (the launcher checks batches * heads * s_total < 2^31)
        const int hrow = (bidx[u] * p.qkv_heads + head) * p.qkv_s_total + srow[u];
        if (ml <= mlast && n < p.N) *(u32x4_t*)(dst + (int64_t)hrow * 128 + dch) = ow;
      }
    }
    return;
  }

#pragma unroll
  for (int j0 = 0; j0 < ITERS; j0 += U) {
    u32x4_t y[U], rv[U], gv[U];
    int coff[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int ml = ml0 + ML_STEP * (j0 + u);
      const int mlc = min(ml, mlast);
      y[u] = *(const u32x4_t*)(ct + ml * C::CT_LD + cc * 8);
      coff[u] = crow.off(mlc) + n;
      if constexpr (EPI == FK_EPI_GATE_RES || EPI == FK_EPI_RES) rv[u] = *(const u32x4_t*)(rbase + rrow.off(mlc) + nc);
      if constexpr (EPI == FK_EPI_GATE_RES) {
        const int64_t b = grow.b0 + grow.crossed(mlc);
        gv[u] = *(const u32x4_t*)((const bf16_t*)p.gate + b * p.gate_batch_stride + nc);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int ml = ml0 + ML_STEP * (j0 + u);
      u32x4_t o = y[u];
      if constexpr (EPI == FK_EPI_GATE_RES || EPI == FK_EPI_RES) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float y0 = bf_lo(o[e]), y1 = bf_hi(o[e]);
          if constexpr (EPI == FK_EPI_GATE_RES) {
            y0 *= bf_lo(gv[u][e]);
            y1 *= bf_hi(gv[u][e]);
            round_bf_pair(y0, y1);
          }
          o[e] = pack_bf2(bf_lo(rv[u][e]) + y0, bf_hi(rv[u][e]) + y1);
        }
      }
      if (ml <= mlast && n < p.N) *(u32x4_t*)(cbase + coff[u]) = o;
    }
  }
}


// ---- 8 waves in two groups that alternate between "multiply" and "load" ("ping-pong") ----------------------------
// 256 x 256 x 64 tile, waves 2 (M) x 4 (N); a wave's output is FOUR 64 x 32 quadrants: rows {128 i + 64 wm + [0,64)} x
// columns {128 j + 32 wn + [0,32)}, i, j in {0,1}.  The A and W tiles are kept in LDS as two 128-row half-tiles each
// (A0 | A1 | W0 | W1, 16 KiB apiece, two K-tiles = 128 KiB), and a K-tile is worked off in four phases, one quadrant
// each (8 MFMAs = 256 matrix-pipe cycles):
//     phase   reads into registers        DMA request (one half-tile = 2 pieces per wave)    multiplies
//     P1      A0(t)          (8 b128)     A1(t+1)                                            quadrant (0,0)
//     P2      W1(t)          (4)          W0(t+2)                                            (0,1)
//     P3      A1(t)          (8)          A0(t+2)                                            (1,0)
//     P4      W0(t+1)        (4)          W1(t+2)                                            (1,1)
//   phase := { reads, DMA requests, s_waitcnt vmcnt(10) } s_barrier { lgkmcnt(0), MFMAs } s_barrier
// Waves 0-3 (wm = 0) and 4-7 (wm = 1) -- one of each per SIMD -- run this stream ONE barrier apart, so while one
// group multiplies the other issues its LDS reads and DMA requests: an LDS-DMA request costs its wave >= 60 issue
// cycles (measured), which stalls the matrix pipe when the same wave also owns the MFMA stream (the
// one-wave-per-SIMD kernels) and costs nothing here.  Every half-tile is requested exactly 6 phases before it is read
// and the wait that retires it (vmcnt(10): five younger half-tiles may stay in flight) sits one phase before the
// read, with a barrier in between; a slot is re-requested no earlier than two phases after its last read.  All eight
// LDS slots are therefore always in use: 5 phases (~1.3 k cycles) of latency cover per request out of 128 KiB.
template <int BN, bool M16_ = false>
struct Cfg8 {
  static_assert(BN == 256, "the ping-pong kernel is instantiated for the 256 x 256 tile only");
  static constexpr bool M16 = M16_;
  static constexpr int NTHREADS = 512;
  static constexpr int BK = 64, ROW_BYTES = 128;
  static constexpr int WAVES_M = 2, WAVES_N = 4;
  static constexpr int MF = 4, NF = 2;
  static constexpr int HALF_BYTES = 128 * ROW_BYTES;      // one half-tile: 128 rows x 64 k
  static constexpr int BUF_BYTES = 4 * HALF_BYTES;        // A0 | A1 | W0 | W1 of one K-tile
  static constexpr int RING_BYTES = 2 * BUF_BYTES;
  static constexpr int CT_LD = BN + 8;
  static constexpr int CT_BYTES = BM * CT_LD * 2;
  static constexpr int SMEM_BYTES = RING_BYTES > CT_BYTES ? RING_BYTES : CT_BYTES;
  static FK_DEV int swz(int row) { return (row >> 1) & 7; }
  static FK_DEV int tile_row(int wm, int mf) { return (mf >> 1) * 128 + wm * 64 + (mf & 1) * 32; }
  static FK_DEV int tile_col(int wn, int nf) { return nf * 128 + wn * 32; }
};

// This workgroup multiplies the K-tiles [kt_first, kt_first + nk) of the tile.  sk_slot < 0: that is the whole K range
// (or its result stands alone).  sk_slot >= 0 (split-K pairs, stream-K ranges): the range is one of the two parts of the
// tile's K range and meets the other part through workspace slot sk_slot (below).
//
// LAY (fk_gemm_args.layout) -- the backward pass's operands as they lie in memory, no transposed copies:
//   0  A [M, K], W [N, K]                 (K contiguous in both: the forward)
//   1  A [M, K], W [K, N]                 data gradient dX = dY W: the weight as stored
//   2  A [K, M], W [K, N]                 weight gradient dW = dY^T X: both operands token-major
// A K-major operand's half-tile is staged as 64 k-rows x 128 columns (256-byte rows, the 64-byte block b of row r at
// block b ^ (r & 3)) and its MFMA fragments come through ds_read_b64_tr_b16 -- two reads of rows 8 hh + tj and + 4, which
// deliver k = 8 hh + 0..7 in the slot order of the ds_read_b128 path, so a K-major operand multiplies a row-major one and
// the sums are those of the LAY 0 kernel on transposed copies bit for bit (attention_fwd.hip's V^T operand is the recipe).
template <int EPI, int BN, int LAY = 0, bool M16 = false>
FK_DEV void gemm8_body(const GroupArgs& ga, char* smem, int pi, int m0, int n0, int kt_first, int nk, int sk_slot) {
  using C = Cfg8<BN, M16>;
  constexpr bool AT = LAY == 2, WT = LAY >= 1;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;   // group = wm; waves w and w + 4 share a SIMD
  const fk_gemm_args& p = ga.p[pi];
  const int kbase = kt_first * (C::BK * 2);   // byte offset of this workgroup's first K-tile in a (K-contiguous) row

  // ---- LDS-DMA sources: piece = 8 rows x 128 B, lane -> (row, slot), source chunk = slot ^ swz(row).
  // Wave w requests pieces 2w and 2w + 1 of every half-tile.
  const int lrow = lane >> 3, slot = lane & 7;
  const int prow = lane >> 4, pslot = lane & 15;   // K-major operands: piece = 4 k-rows x 256 B, lane -> (row, 16-byte chunk)
  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(
      (void*)((const bf16_t*)p.A + (AT ? (int64_t)m0 : fk_row_offset(p.a, m0))), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(
      (void*)((const bf16_t*)p.W + (WT ? (int64_t)n0 : (int64_t)n0 * p.ldw)), 0, 0x7fffffff, 0x00020000);
  const BufDesc od_a = make_buf_desc((const bf16_t*)p.A + (AT ? (int64_t)m0 : fk_row_offset(p.a, m0)), 0x7fffffffu);   // K-major forms
  const BufDesc od_w = make_buf_desc((const bf16_t*)p.W + (WT ? (int64_t)n0 : (int64_t)n0 * p.ldw), 0x7fffffffu);
  int a_voff[2][2], w_voff[2][2];   // [half][piece]
  {
    const TileRows arow(p.a, m0);
    const int ldw2 = (int)p.ldw * 2, lda2 = (int)p.a.ld * 2;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int rl = h * 128 + (wave * 2 + j) * 8 + lrow;   // row inside the 256-row tile
        const int sw = (slot ^ C::swz(rl)) << 4;
        // K-major: k-row kl of the K-tile, logical 16-byte chunk of the half's 128 columns behind physical chunk pslot
        const int kl = (wave * 2 + j) * 4 + prow;
        const int ct = (h * 128 + (((((pslot >> 2) ^ prow) << 2) | (pslot & 3)) << 3)) * 2;
        a_voff[h][j] = AT ? kl * lda2 + ct : arow.off(min(rl, p.M - 1 - m0)) * 2 + sw;
        w_voff[h][j] = WT ? kl * ldw2 + ct : min(rl, p.N - 1 - n0) * ldw2 + sw;
      }
  }
  // which: 0 = A0, 1 = A1, 2 = W0, 3 = W1; kt is clamped (surplus requests are never read)
  const int kstep_a = AT ? C::BK * (int)p.a.ld * 2 : C::BK * 2, kstep_w = WT ? C::BK * (int)p.ldw * 2 : C::BK * 2;
  auto dma_half = [&](int which, int buf, int kt) {
    const int ktc = min(kt, nk - 1);
    char* dst = smem + buf * C::BUF_BYTES + which * C::HALF_BYTES + wave * 2048;
    if (which < 2) {
      const int koff = (AT ? 0 : kbase) + ktc * kstep_a;
      if constexpr (AT || WT) {
        buffer_lds16(od_a, dst, a_voff[which][0], koff);
        buffer_lds16(od_a, dst + 1024, a_voff[which][1], koff);
      } else {
        buffer_lds16(rs_a, dst, a_voff[which][0], koff);
        buffer_lds16(rs_a, dst + 1024, a_voff[which][1], koff);
      }
    } else {
      const int koff = (WT ? 0 : kbase) + ktc * kstep_w;
      if constexpr (AT || WT) {
        buffer_lds16(od_w, dst, w_voff[which - 2][0], koff);
        buffer_lds16(od_w, dst + 1024, w_voff[which - 2][1], koff);
      } else {
        buffer_lds16(rs_w, dst, w_voff[which - 2][0], koff);
        buffer_lds16(rs_w, dst + 1024, w_voff[which - 2][1], koff);
      }
    }
  };

  // ---- MFMA operand addressing -----------------------------------------------------------------------------------
  // 32 x 32 x 16: lane -> row (lane & 31), k-octet (lane >> 5) of each of the K-tile's 4 k-steps; 16 x 16 x 32: row (lane & 15),
  // k-octet (lane >> 4) of each of its 2 k-steps.  Either way one ds_read_b128 per fragment, 8 + 4 per phase, the same
  // swizzle (the XOR only depends on (row >> 1) & 7, which a 16-row sub-block offset does not change).
  constexpr int NKS = M16 ? 2 : 4;          // k-steps per K-tile
  constexpr int ASUB = M16 ? 4 : 2;         // fragments per k-step over the wave's 64 A rows
  constexpr int WSUB = M16 ? 2 : 1;         // ... over its 32 W rows
  constexpr int SUB_BYTES = (M16 ? 16 : 32) * C::ROW_BYTES;
  const int frow = M16 ? (lane & 15) : (lane & 31), fhalf = M16 ? (lane >> 4) : (lane >> 5), fsw = C::swz(frow);
  int koffs[NKS];
#pragma unroll
  for (int kk = 0; kk < NKS; ++kk) koffs[kk] = ((kk * (M16 ? 4 : 2) + fhalf) ^ fsw) << 4;
  const int a_rd = (wm * 64 + frow) * C::ROW_BYTES;   // + sub * SUB_BYTES inside half-tile A_i
  const int w_rd = (wn * 32 + frow) * C::ROW_BYTES;   // inside half-tile W_j

  f32x16_t acc[C::NF][C::MF];
#pragma unroll
  for (int i = 0; i < C::NF; ++i)
#pragma unroll
    for (int j = 0; j < C::MF; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  bf16x8_t af[ASUB][NKS], wf[2][WSUB][NKS];   // A half in use [sub][kk]; W halves [j][sub][kk]
  // K-major half-tile: fragment (k-step kk, column block df) through two transpose reads.  32 x 32 x 16: df = 32-column block, the
  // two 16-lane groups of a k-octet (lane >> 5) take its two 16-column halves; 16 x 16 x 32 (round 6): df = 16-column block, every
  // 16-lane group g = lane >> 4 is the k-octet g of the 32-deep step -- rows 8 g + tj and + 4 deliver k = 8 g + 0..7 in the slot order
  // of the ds_read_b128 path either way, so both shapes multiply a K-major operand with a row-major one bit for bit like layout 0
  const int tj = (lane & 15) >> 2;
  const int t_lo = M16 ? (8 * (lane >> 4) + tj) * 256 + (lane & 3) * 8
                       : (8 * (lane >> 5) + tj) * 256 + ((lane >> 4) & 1) * 32 + (lane & 3) * 8;
  auto trfrag = [&](const char* half, int kk, int df) {
    const char* vp = M16 ? half + kk * 8192 + (((df >> 1) ^ tj) << 6) + (df & 1) * 32 + t_lo
                         : half + kk * 4096 + ((df ^ tj) << 6) + t_lo;
    const s16x4_t lo = lds_tr16(vp);
    const s16x4_t hi = lds_tr16(vp + 4 * 256);
    bf16x8_t f;
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
    f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    return f;
  };
  auto read_a = [&](int buf, int h) {
    const char* b = smem + buf * C::BUF_BYTES + h * C::HALF_BYTES;
#pragma unroll
    for (int sub = 0; sub < ASUB; ++sub)
#pragma unroll
      for (int kk = 0; kk < NKS; ++kk) {
        if constexpr (AT) af[sub][kk] = trfrag(b, kk, wm * ASUB + sub);
        else af[sub][kk] = *(const bf16x8_t*)(b + a_rd + sub * SUB_BYTES + koffs[kk]);
      }
  };
  auto read_w = [&](int buf, int j) {
    const char* b = smem + buf * C::BUF_BYTES + (2 + j) * C::HALF_BYTES;
#pragma unroll
    for (int sub = 0; sub < WSUB; ++sub)
#pragma unroll
      for (int kk = 0; kk < NKS; ++kk) {
        if constexpr (WT) wf[j][sub][kk] = trfrag(b, kk, wn * WSUB + sub);
        else wf[j][sub][kk] = *(const bf16x8_t*)(b + w_rd + sub * SUB_BYTES + koffs[kk]);
      }
  };
  // quadrant (i, j): 2 m-blocks x 1 n-block x the K-tile, accumulators alternating (8 MFMAs of 32 x 32 x 16 or 16 of 16 x 16 x 32)
  auto mma = [&](int i, int j) {
#pragma unroll
    for (int kk = 0; kk < NKS; ++kk)
#pragma unroll
      for (int sub = 0; sub < 2; ++sub) {
        if constexpr (M16) mma16_block(acc[j][2 * i + sub], wf[j][0][kk], wf[j][1][kk], af[2 * sub][kk], af[2 * sub + 1][kk]);
        else acc[j][2 * i + sub] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[j][0][kk], af[sub][kk], acc[j][2 * i + sub], 0, 0, 0);
      }
  };
  auto phase_sync_mma = [&](int i, int j) {
    wait_vmcnt<10>();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    mma(i, j);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

  // ---- prologue: the seven half-tiles read in phases 0 .. 6, in reading order ------------------------------------
  dma_half(2, 0, 0);   // W0(0)   read in "phase 0" (below)
  dma_half(0, 0, 0);   // A0(0)   P1 of tile 0
  dma_half(3, 0, 0);   // W1(0)   P2
  dma_half(1, 0, 0);   // A1(0)   P3
  dma_half(2, 1, 1);   // W0(1)   P4
  dma_half(0, 1, 1);   // A0(1)   P1 of tile 1
  dma_half(3, 1, 1);   // W1(1)   P2 of tile 1
  wait_vmcnt<10>();    // W0(0), A0(0) (own pieces) landed
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  read_w(0, 0);
  if (wm == 1) __builtin_amdgcn_s_barrier();   // the stagger: group 1 runs one barrier behind group 0
  __builtin_amdgcn_sched_barrier(0);

  auto tile_body = [&](auto bc, int kt) {
    constexpr int b = decltype(bc)::value;
    read_a(b, 0);            dma_half(1, b ^ 1, kt + 1);  phase_sync_mma(0, 0);
    read_w(b, 1);            dma_half(2, b, kt + 2);      phase_sync_mma(0, 1);
    read_a(b, 1);            dma_half(0, b, kt + 2);      phase_sync_mma(1, 0);
    read_w(b ^ 1, 0);        dma_half(3, b, kt + 2);      phase_sync_mma(1, 1);
  };
  for (int kt = 0; kt < nk; kt += 2) {
    tile_body(std::integral_constant<int, 0>{}, kt);
    if (kt + 1 < nk) tile_body(std::integral_constant<int, 1>{}, kt + 1);
  }
  if (wm == 0) __builtin_amdgcn_s_barrier();   // balance the stagger
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // surplus (clamped) requests must not land in the C tile

  if (sk_slot >= 0) {
    // ---- split-K rendezvous ---------------------------------------------------------------------------------------
    // The two workgroups of a tile each hold the fp32 partial sums of half the K range.  Whoever finishes FIRST
    // (ticket = an agent-scope fetch-add on the tile's counter word: even -> first) writes its accumulators to the
    // tile's workspace slot with write-through stores, drains them, publishes flag = ticket + 1 and exits; the
    // SECOND (odd ticket) waits for flag == its ticket, acquires, adds the stored partials to its own and runs the
    // epilogue.  fp32 addition commutes, so own + other is the same bits whichever half arrives first: the result
    // is deterministic.  The wait cannot deadlock under any dispatch order: it is only ever for a workgroup that has
    // already drawn its ticket, i.e. is resident and a few microseconds from publishing.  Counter and flag are
    // monotonic (every launch adds exactly two tickets per slot it uses), so nothing is reset between launches.
    // Recipe: cdna_hip_programming.md Guideline 16 R1 (sc1 payload -> per-wave vmcnt(0) -> barrier -> one-lane
    // relaxed agent flag store; consumer: one-lane relaxed poll -> ONE agent acquire -> barrier -> loads).
    typedef __attribute__((address_space(1))) unsigned gu32;
    gu32* const ctl = (gu32*)(ga.sk_ctl + 2 * (size_t)sk_slot);
    const __amdgpu_buffer_rsrc_t rs_p = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(ga.sk_partials + (size_t)sk_slot * (BM * BN)), 0, BM * BN * 4, 0x00020000);
    __syncthreads();   // every wave is done with the operand ring: its first word now carries the ticket
    if (tid == 0) *(volatile unsigned*)smem = __hip_atomic_fetch_add(ctl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const unsigned ticket = __builtin_amdgcn_readfirstlane(*(volatile unsigned*)smem);
    constexpr int NV = C::NF * C::MF * 4;   // 16-byte pieces of the accumulators per thread: piece r of thread t at (r * 512 + t) * 16
    if ((ticket & 1u) == 0) {
#pragma unroll
      for (int r = 0; r < NV; ++r) {
        const f32x16_t& a = acc[r / (C::MF * 4)][(r / 4) % C::MF];
        const int q = r & 3;
        const u32x4_t v = {__float_as_uint(a[4 * q]), __float_as_uint(a[4 * q + 1]), __float_as_uint(a[4 * q + 2]),
                           __float_as_uint(a[4 * q + 3])};
        __builtin_amdgcn_raw_buffer_store_b128(v, rs_p, tid * 16, r * (C::NTHREADS * 16), /*sc1: write through*/ 16);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains its own stores
      __syncthreads();
      if (tid == 0) __hip_atomic_store(ctl + 1, ticket + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return;
    }
    if (tid == 0) {
      // bounded (~seconds): the partner has drawn its ticket and is microseconds from publishing; a corrupted workspace must
      // end in a wrong tile (the parity tests see it), not in a hung device
      int spins = 0;
      while (__hip_atomic_load(ctl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != ticket && spins < (1 << 22)) {
        __builtin_amdgcn_s_sleep(8);
        ++spins;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
#pragma unroll
    for (int r0 = 0; r0 < NV; r0 += 4) {
      u32x4_t v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u)
        v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs_p, tid * 16, (r0 + u) * (C::NTHREADS * 16), /*sc1*/ 16);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int r = r0 + u;
        f32x16_t& a = acc[r / (C::MF * 4)][(r / 4) % C::MF];
        const int q = r & 3;
#pragma unroll
        for (int j = 0; j < 4; ++j) a[4 * q + j] += __uint_as_float(v[u][j]);
      }
      __builtin_amdgcn_sched_barrier(0);   // keep the batches apart: all 32 loads in flight at once would need 128 more VGPRs
      // (hipcc still moves one or two accumulator tiles through scratch around this block: <= 8 stores + 8 loads per
      //  workgroup, outside the K loop; tests/test_kernel_resources.py budgets exactly that for this instantiation)
    }
  }
  store_tile<EPI, BN, C>(acc, p, smem, m0, n0, wm, wn);
}

template <int EPI, int BN, bool SPLITK, int LAY = 0, bool M16 = false>
__global__ __launch_bounds__(512, 2) void gemm8_kernel(const GroupArgs ga) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int pi, m0, n0;
  const int t = xcd_chunk_index();
  const int nk_all = ga.p[0].K / Cfg8<BN>::BK;
  if constexpr (SPLITK) {
    select_tile<BN>(ga, t >> 1, pi, m0, n0);
    gemm8_body<EPI, BN, LAY, M16>(ga, smem, pi, m0, n0, (t & 1) * (nk_all >> 1), nk_all >> 1, t >> 1);
  } else {
    select_tile<BN>(ga, t, pi, m0, n0);
    gemm8_body<EPI, BN, LAY, M16>(ga, smem, pi, m0, n0, 0, nk_all, -1);
  }
}

// ---- gemm10: four waves, one per SIMD, 128 x 128 per wave, every instruction of the K loop placed by hand (round 6) ---------
// 256 x 256 x 64 tile on v_mfma_f32_16x16x32_bf16 with all 256 accumulator registers pinned in the AGPR half: 0.25 ds_read_b128
// per MFMA (gemm8_kernel: 0.75), half the waves, ONE barrier per K-tile (gemm8_kernel: 8), operands staged through registers
// (buffer_load_dwordx4 one K-tile ahead + ds_write_b128) because an LDS-DMA request costs a wave that is alone on its SIMD four
// MFMA slots.  The loop is ONE asm statement (gemm10_loop.inc, written by gemm10_gen.py: schedule, register map and the reasons
// are in that file's header); its operands are pinned to fixed registers, so the text names registers directly.  Same LDS image,
// same fragment -> k mapping and same K order as gemm8_kernel<.., M16 = true>: the sums agree bit for bit.
struct Cfg10 {
  static constexpr bool M16 = true;
  static constexpr int NTHREADS = 256;
  static constexpr int BK = 64, ROW_BYTES = 128;
  static constexpr int MF = 4, NF = 4;
  static constexpr int HALF_BYTES = 128 * ROW_BYTES;      // one half-tile: 128 rows x 64 k
  static constexpr int BUF_BYTES = 4 * HALF_BYTES;        // A0 | A1 | W0 | W1 of one K-tile
  static constexpr int RING_BYTES = 2 * BUF_BYTES;
  static constexpr int CT_LD = 256 + 8;
  static constexpr int CT_BYTES = BM * CT_LD * 2;
  static constexpr int SMEM_BYTES = RING_BYTES > CT_BYTES ? RING_BYTES : CT_BYTES;
  static FK_DEV int tile_row(int wm, int mf) { return wm * 128 + mf * 32; }
  static FK_DEV int tile_col(int wn, int nf) { return wn * 128 + nf * 32; }
};
typedef int i32x8_t __attribute__((ext_vector_type(8)));
typedef int i32x4_t __attribute__((ext_vector_type(4)));

// v64 .. v255 as clobber strings (staging and fragment registers of gemm10_loop.inc)
#define G10_V4(a) "v" #a "0", "v" #a "1", "v" #a "2", "v" #a "3"
#define G10_V10(a) "v" #a "0", "v" #a "1", "v" #a "2", "v" #a "3", "v" #a "4", "v" #a "5", "v" #a "6", "v" #a "7", "v" #a "8", "v" #a "9"
#define G10_CLOBBER_V64_255 \
  "v64", "v65", "v66", "v67", "v68", "v69", G10_V10(7), G10_V10(8), G10_V10(9), G10_V10(10), G10_V10(11), G10_V10(12), G10_V10(13), \
  G10_V10(14), G10_V10(15), G10_V10(16), G10_V10(17), G10_V10(18), G10_V10(19), G10_V10(20), G10_V10(21), G10_V10(22), G10_V10(23), \
  G10_V10(24), "v250", "v251", "v252", "v253", "v254", "v255"
#define G10_OPERANDS \
      : "={a[0:15]}"(acc[0][0]), "={a[16:31]}"(acc[0][1]), "={a[32:47]}"(acc[0][2]), "={a[48:63]}"(acc[0][3]), \
        "={a[64:79]}"(acc[1][0]), "={a[80:95]}"(acc[1][1]), "={a[96:111]}"(acc[1][2]), "={a[112:127]}"(acc[1][3]), \
        "={a[128:143]}"(acc[2][0]), "={a[144:159]}"(acc[2][1]), "={a[160:175]}"(acc[2][2]), "={a[176:191]}"(acc[2][3]), \
        "={a[192:207]}"(acc[3][0]), "={a[208:223]}"(acc[3][1]), "={a[224:239]}"(acc[3][2]), "={a[240:255]}"(acc[3][3]) \
      : "{v[16:23]}"(a_voff), "{v[24:31]}"(w_voff), "{v[32:39]}"(rd), "{v[40:43]}"(wr), "{s[40:43]}"(od_a.w), "{s[44:47]}"(od_w.w), \
        "{s48}"(k0), "{s49}"(nks), "{s50}"(kmax) \
      : "memory", "scc", "s52", "s53", "s54", G10_CLOBBER_V64_255

#define G10_OPERANDS_X \
      : "={a[0:15]}"(acc[0][0]), "={a[16:31]}"(acc[0][1]), "={a[32:47]}"(acc[0][2]), "={a[48:63]}"(acc[0][3]), \
        "={a[64:79]}"(acc[1][0]), "={a[80:95]}"(acc[1][1]), "={a[96:111]}"(acc[1][2]), "={a[112:127]}"(acc[1][3]), \
        "={a[128:143]}"(acc[2][0]), "={a[144:159]}"(acc[2][1]), "={a[160:175]}"(acc[2][2]), "={a[176:191]}"(acc[2][3]), \
        "={a[192:207]}"(acc[3][0]), "={a[208:223]}"(acc[3][1]), "={a[224:239]}"(acc[3][2]), "={a[240:255]}"(acc[3][3]), "={s[56:57]}"(t0), "={s[58:59]}"(t1), "={s[60:61]}"(ta) \
      : "{v[16:23]}"(a_voff), "{v[24:31]}"(w_voff), "{v[32:39]}"(rd), "{v[40:43]}"(wr), "{s[40:43]}"(od_a.w), "{s[44:47]}"(od_w.w), \
        "{s48}"(k0), "{s49}"(nks), "{s50}"(kmax) \
      : "memory", "scc", "s52", "s53", "s54", G10_CLOBBER_V64_255

template <int EPI, int V = 0>
FK_DEV void gemm10_body(const GroupArgs& ga, char* smem, int pi, int m0, int n0, int kt_first, int nk) {
  using C = Cfg10;
#ifdef FK_G10_EXPERIMENTS
  const unsigned long long g10_real0 = __builtin_amdgcn_s_memrealtime(), g10_cyc0 = __builtin_amdgcn_s_memtime();
#endif
  // the lane's addresses are derived from an opaque copy of threadIdx.x: inside the persistent grid's tile loop hipcc would
  // otherwise hoist all of them out of the loop, where they live across the asm statement -- which leaves it 36 free VGPRs --
  // and spill (228 bytes of scratch, twice the epilogue's instructions: measured 35-50 k cycles per epilogue instead of 9.5 k)
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const fk_gemm_args& p = ga.p[pi];
  const BufDesc od_a = make_buf_desc((const bf16_t*)p.A + fk_row_offset(p.a, m0), 0x7fffffffu);
  const BufDesc od_w = make_buf_desc((const bf16_t*)p.W + (int64_t)n0 * p.ldw, 0x7fffffffu);
  // global side: wave w stages rows [64 w, 64 w + 64) of the A tile and of the W tile as 8 pieces of 8 rows x 128 B each;
  // lane -> (row lane >> 3, 16-byte chunk lane & 7); rows beyond the problem are clamped (their products are never stored)
  const int lrow = lane >> 3, chunk = lane & 7;
  i32x8_t a_voff, w_voff;
  {
    const TileRows arow(p.a, m0);
    const int ldw2 = (int)p.ldw * 2;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int rl = wave * 64 + i * 8 + lrow;
      a_voff[i] = arow.off(min(rl, p.M - 1 - m0)) * 2 + chunk * 16;
      w_voff[i] = min(rl, p.N - 1 - n0) * ldw2 + chunk * 16;
    }
  }
  // LDS side.  Write: piece i of the wave lands at rows 8 ((w & 1) 8 + i) + lrow of half-tile w >> 1 (A) / of W's at + 32 KiB;
  // logical chunk c of row r at physical chunk c ^ ((r >> 1) & 7), and (r >> 1) & 7 = (4 (i & 1) + (lrow >> 1)) & 7.
  // Read: gemm8_kernel's M16 fragments -- row lane & 15 of 16-row block m / n (the immediate), k-octet lane >> 4 of k-step kk.
  const int base = (int)lds_addr_of(smem);
  i32x8_t rd;
  i32x4_t wr;
  {
    const int frow = lane & 15, fhalf = lane >> 4, fsw = (frow >> 1) & 7;
#pragma unroll
    for (int buf = 0; buf < 2; ++buf)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const int ko = ((kk * 4 + fhalf) ^ fsw) << 4;
        rd[2 * buf + kk] = base + buf * C::BUF_BYTES + wm * C::HALF_BYTES + frow * C::ROW_BYTES + ko;
        rd[4 + 2 * buf + kk] = base + buf * C::BUF_BYTES + (2 + wn) * C::HALF_BYTES + frow * C::ROW_BYTES + ko;
      }
#pragma unroll
    for (int buf = 0; buf < 2; ++buf)
#pragma unroll
      for (int par = 0; par < 2; ++par)
        wr[2 * buf + par] = base + buf * C::BUF_BYTES + (wave >> 1) * C::HALF_BYTES + (wave & 1) * 8192 + lrow * C::ROW_BYTES +
                            ((chunk ^ ((par * 4 + (lrow >> 1)) & 7)) << 4);
  }
  const int k0 = __builtin_amdgcn_readfirstlane(kt_first * (C::BK * 2));
  const int nks = __builtin_amdgcn_readfirstlane(nk);
  const int kmax = k0 + (nks - 1) * (C::BK * 2);

  f32x16_t acc[C::NF][C::MF];
  if constexpr (V == 0) {
    asm volatile(
#include "gemm10_loop.inc"
        G10_OPERANDS);
  }
#ifdef FK_G10_EXPERIMENTS
  unsigned long long t0 = 0, t1 = 0, ta = 0;
  (void)t0; (void)t1; (void)ta;
  if constexpr (V == 1) { asm volatile(
#include "gemm10_loop_x1.inc"
        G10_OPERANDS_X); }
  else if constexpr (V == 2) { asm volatile(
#include "gemm10_loop_x2.inc"
        G10_OPERANDS_X); }
  else if constexpr (V == 3) { asm volatile(
#include "gemm10_loop_x3.inc"
        G10_OPERANDS_X); }
  else if constexpr (V == 4) { asm volatile(
#include "gemm10_loop_x4.inc"
        G10_OPERANDS_X); }
  else if constexpr (V == 5) { asm volatile(
#include "gemm10_loop_x5.inc"
        G10_OPERANDS_X); }
  else if constexpr (V == 6) { asm volatile(
#include "gemm10_loop_x6.inc"
        G10_OPERANDS_X); }
  else if constexpr (V == 7) { asm volatile(
#include "gemm10_loop_x7.inc"
        G10_OPERANDS_X); }
  else if constexpr (V == 8) { asm volatile(
#include "gemm10_loop_x8.inc"
        G10_OPERANDS_X); }
  else if constexpr (V == 9) { asm volatile(
#include "gemm10_loop_x9.inc"
        G10_OPERANDS_X); }
  else if constexpr (V == 10) { asm volatile(
#include "gemm10_loop_x10.inc"
        G10_OPERANDS_X); }
  else if constexpr (V == 11) { asm volatile(
#include "gemm10_loop_x11.inc"
        G10_OPERANDS_X); }
  else if constexpr (V == 12) { asm volatile(
#include "gemm10_loop_x12.inc"
        G10_OPERANDS_X); }
  else if constexpr (V == 13) { asm volatile(
#include "gemm10_loop_x13.inc"
        G10_OPERANDS_X); }
  else if constexpr (V == 14) { asm volatile(
#include "gemm10_loop_x14.inc"
        G10_OPERANDS_X); }
  else if constexpr (V == 15) { asm volatile(
#include "gemm10_loop_x15.inc"
        G10_OPERANDS_X); }
  else if constexpr (V == 16) { asm volatile(
#include "gemm10_loop_x16.inc"
        G10_OPERANDS_X); }
  else if constexpr (V == 17) { asm volatile(
#include "gemm10_loop_x17.inc"
        G10_OPERANDS_X); }
  else if constexpr (V == 18) { asm volatile(
#include "gemm10_loop_x18.inc"
        G10_OPERANDS_X); }
  else if constexpr (V == 19) { asm volatile(
#include "gemm10_loop_x19.inc"
        G10_OPERANDS_X); }
  else if constexpr (V == 20) { asm volatile(
#include "gemm10_loop_x20.inc"
        G10_OPERANDS_X); }
#endif
  store_tile<EPI, 256, C>(acc, p, smem, m0, n0, wm, wn);
#ifdef FK_G10_EXPERIMENTS
  if constexpr (V > 0) {   // per workgroup, behind C: where it ran, when (wall clock), and wave 0's shader-cycle stamps of kernel entry,
    if (tid == 0) {        // asm statement entry, loop start, loop end (s_memtime inside the statement) and its own last store issued
      unsigned long long* rec = (unsigned long long*)((bf16_t*)p.C + (int64_t)p.M * p.c.ld) + (size_t)blockIdx.x * 8;   // behind C's last row
      rec[0] = (unsigned long long)__builtin_amdgcn_s_getreg(63492) | ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32);  // HW_ID, XCC_ID
      rec[1] = g10_real0; rec[2] = __builtin_amdgcn_s_memrealtime();
      rec[3] = g10_cyc0; rec[4] = t0; rec[5] = t1; rec[6] = __builtin_amdgcn_s_memtime();
      rec[7] = ta;
    }
  }
#endif
}

template <int EPI, int V = 0, bool PERSIST = false>
__global__ __launch_bounds__(256, 1) void gemm10_kernel(const GroupArgs ga) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int nk = ga.p[0].K / Cfg10::BK;
  if constexpr (PERSIST) {
    // one workgroup per CU walks the tile list with stride gridDim: the 32 workgroups of an XCD work on 32 consecutive tiles
    // of the grouped order in every round (what the dispatcher does with a plain grid), without the relaunch gap and with the
    // kernel arguments warm in the scalar cache from the second tile on
    const int total = ga.tiles_before[FK_MAX_GROUP];
    for (int t = xcd_chunk_index(); t < total; t += gridDim.x) {
      int pi, m0, n0;
      select_tile<256>(ga, t, pi, m0, n0);
      gemm10_body<EPI, V>(ga, smem, pi, m0, n0, 0, nk);
      // every wave is done READING the C tile before the next tile's prologue refills the ring: an LDS-only barrier -- a
      // __syncthreads() would also wait for this tile's global stores to drain (all CUs at once: measured 35-50 k cycles)
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
  } else {
    int pi, m0, n0;
    select_tile<256>(ga, xcd_chunk_index(), pi, m0, n0);
    gemm10_body<EPI, V>(ga, smem, pi, m0, n0, 0, nk);
  }
}

// ---- stream-K ranges (round 4) --------------------------------------------------------------------------------------------
// A long-K GEMM whose 256 x 256 tiling needs a poorly filled last round (M = 8704, N = 3072, K = 12288 / 15360: 408 tiles =
// 1.59 rounds of 256 CUs, run as 2) as a PERSISTENT grid of one workgroup per CU: the tiles' K-tiles are dealt out as G
// equal contiguous ranges (cut j at floor(U j / G), moved onto a tile boundary when it would leave a part shorter than
// sk_min_part K-tiles); the launcher guarantees U / G >= nk + 2 sk_min_part, so a tile is cut at most once and its two parts
// meet through the split-K pairs' rendezvous (slot = the cut's index).  Round 1's stream-K (every CU a share of EVERY last-round
// tile, device-scope fences, partials re-read in K order) was 2x slower; this one moves one 256 KiB partial per CU and launch.
// The range is walked from its end: the head part of its last tile first (the part that publishes), the tail part of its
// first tile last (the part that merges) -- whoever merges finds the partial already there.
template <int EPI, int BN, bool M16 = false>
__global__ __launch_bounds__(512, 2) void gemm8_streamk_kernel(const GroupArgs ga) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int pos = xcd_chunk_index();
  const unsigned G = gridDim.x, nk = ga.p[0].K / Cfg8<BN>::BK;
  const unsigned U = (unsigned)ga.tiles_before[FK_MAX_GROUP] * nk, qU = U / G, rU = U - qU * G;
  auto cut = [&](unsigned j) __attribute__((always_inline)) {
    unsigned c = qU * j + (rU * j) / G;                      // floor(U j / G) without a 64-bit product
    const unsigned r = c % nk;
    if (r != 0 && r < (unsigned)ga.sk_min_part) c -= r;
    else if (r != 0 && nk - r < (unsigned)ga.sk_min_part) c += nk - r;
    return (int)c;
  };
  const int u = cut(pos);
  int u_end = cut(pos + 1);
  while (u < u_end) {
    const int t = (unsigned)(u_end - 1) / nk;
    const int k1 = u_end - t * (int)nk, k0 = max(u - t * (int)nk, 0);
    u_end -= k1 - k0;
    int pi, m0, n0;
    select_tile<BN>(ga, t, pi, m0, n0);
    const bool part = k0 > 0 || k1 < (int)nk;
    gemm8_body<EPI, BN, 0, M16>(ga, smem, pi, m0, n0, k0, k1 - k0, part ? (k1 < (int)nk ? pos + 1 : pos) : -1);
    __syncthreads();   // every wave is done with the C tile / the ticket word before the next pass refills the ring
  }
}

template <int BN>
int count_tiles(GroupArgs& ga, const fk_gemm_args* probs, int n) {
  int total = 0;
  for (int i = 0; i < FK_MAX_GROUP; ++i) {
    ga.tiles_before[i] = total;
    if (i < n) total += ((probs[i].M + BM - 1) / BM) * ((probs[i].N + BN - 1) / BN);
  }
  ga.tiles_before[FK_MAX_GROUP] = total;
  return total;
}

template <int EPI, int BN, bool SPLITK, int LAY = 0, bool M16 = false>
int launch8(GroupArgs& ga, const fk_gemm_args* probs, int n, hipStream_t stream) {
  const int total = count_tiles<BN>(ga, probs, n);
  auto kern = gemm8_kernel<EPI, BN, SPLITK, LAY, M16>;
  FK_ENSURE_MAX_LDS(kern, Cfg8<BN>::SMEM_BYTES, "fk_gemm_bf16 (256 x 256 tile, 8 waves ping-pong)");
  hipLaunchKernelGGL(kern, dim3(SPLITK ? 2 * total : total), dim3(512), Cfg8<BN>::SMEM_BYTES, stream, ga);
  FK_CHECK_LAUNCH("fk_gemm_bf16 (256 x 256 tile, 8 waves ping-pong)");
  return FK_OK;
}

template <int EPI, int BN, bool M16 = false>
int launch8_streamk(GroupArgs& ga, const fk_gemm_args* probs, int n, int grid, hipStream_t stream) {
  count_tiles<BN>(ga, probs, n);
  auto kern = gemm8_streamk_kernel<EPI, BN, M16>;
  FK_ENSURE_MAX_LDS(kern, Cfg8<BN>::SMEM_BYTES, "fk_gemm_bf16 (256 x 256 tile, stream-K ranges)");
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), Cfg8<BN>::SMEM_BYTES, stream, ga);
  FK_CHECK_LAUNCH("fk_gemm_bf16 (256 x 256 tile, stream-K ranges)");
  return FK_OK;
}


int cu_count();
template <int EPI>
int launch10(GroupArgs& ga, const fk_gemm_args* probs, int n, hipStream_t stream) {
  const int total = count_tiles<256>(ga, probs, n);
#ifdef FK_G10_EXPERIMENTS
  if constexpr (EPI == FK_EPI_NONE) {   // measurement forms of the loop (gemm10_gen.py EXPERIMENTS): FK_G10_X=<n>
    static const int x = getenv("FK_G10_X") ? atoi(getenv("FK_G10_X")) : 0;
    static const int persist = getenv("FK_G10_PERSIST") ? atoi(getenv("FK_G10_PERSIST")) : 0;
    void (*kx)(const GroupArgs) = nullptr;
    if (persist) kx = x == 1 ? gemm10_kernel<EPI, 1, true> : gemm10_kernel<EPI, 0, true>;
    else switch (x) {
      case 1: kx = gemm10_kernel<EPI, 1>; break;
      case 2: kx = gemm10_kernel<EPI, 2>; break;
      case 3: kx = gemm10_kernel<EPI, 3>; break;
      case 4: kx = gemm10_kernel<EPI, 4>; break;
      case 5: kx = gemm10_kernel<EPI, 5>; break;
      case 6: kx = gemm10_kernel<EPI, 6>; break;
      case 7: kx = gemm10_kernel<EPI, 7>; break;
      case 8: kx = gemm10_kernel<EPI, 8>; break;
      case 9: kx = gemm10_kernel<EPI, 9>; break;
      case 10: kx = gemm10_kernel<EPI, 10>; break;
      case 11: kx = gemm10_kernel<EPI, 11>; break;
      case 12: kx = gemm10_kernel<EPI, 12>; break;
      case 13: kx = gemm10_kernel<EPI, 13>; break;
      case 14: kx = gemm10_kernel<EPI, 14>; break;
      case 15: kx = gemm10_kernel<EPI, 15>; break;
      case 16: kx = gemm10_kernel<EPI, 16>; break;
      case 17: kx = gemm10_kernel<EPI, 17>; break;
      case 18: kx = gemm10_kernel<EPI, 18>; break;
      case 19: kx = gemm10_kernel<EPI, 19>; break;
      case 20: kx = gemm10_kernel<EPI, 20>; break;
      default: break;
    }
    if (kx) {
      if (hipFuncSetAttribute((const void*)kx, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg10::SMEM_BYTES) != hipSuccess) return FK_EINVAL;
      hipLaunchKernelGGL(kx, dim3(persist ? (total < cu_count() ? total : cu_count()) : total), dim3(256), Cfg10::SMEM_BYTES, stream, ga);
      FK_CHECK_LAUNCH("fk_gemm_bf16 (gemm10 experiment)");
      return FK_OK;
    }
  }
#endif
  // a deep grid (>= 4 tiles per CU) runs as one workgroup per CU walking the tile list: no relaunch gap, kernel arguments warm
  // from the second tile on (+2.5 % at 32768 x 3072 x 12288, profiles/r06_gemm10_persistent_ab_callH.txt); shallow grids keep
  // the plain one, which lets the dispatcher balance the last round (-4 % at 480 tiles otherwise)
  const int G = cu_count();
  if (total >= 4 * G) {
    auto kern = gemm10_kernel<EPI, 0, true>;
    FK_ENSURE_MAX_LDS(kern, Cfg10::SMEM_BYTES, "fk_gemm_bf16 (256 x 256 tile, 4 waves, hand-placed loop, one workgroup per CU)");
    hipLaunchKernelGGL(kern, dim3(G), dim3(256), Cfg10::SMEM_BYTES, stream, ga);
    FK_CHECK_LAUNCH("fk_gemm_bf16 (256 x 256 tile, 4 waves, hand-placed loop, one workgroup per CU)");
    return FK_OK;
  }
  auto kern = gemm10_kernel<EPI, 0, false>;
  FK_ENSURE_MAX_LDS(kern, Cfg10::SMEM_BYTES, "fk_gemm_bf16 (256 x 256 tile, 4 waves, hand-placed loop)");
  hipLaunchKernelGGL(kern, dim3(total), dim3(256), Cfg10::SMEM_BYTES, stream, ga);
  FK_CHECK_LAUNCH("fk_gemm_bf16 (256 x 256 tile, 4 waves, hand-placed loop)");
  return FK_OK;
}

// ---- the same two-group alternation on the 256 x 128 tile ---------------------------------------------------------
// For grids whose 256 x 256 tiling leaves the last round of 256 CUs poorly filled (M = 2560: N = 3072, 9216).
// +1..4 % over the lockstep 256 x 128 kernel of round 1 on every shape of the path (69 -> 74 % MFMA-busy).
// Waves 4 (M) x 2 (N), 64 x 64 per wave; group = the N half (waves w, w + 4 share a SIMD).  A K-tile is two phases of
// 8 MFMAs: (A rows of the wave) x (W block j), j = 0, 1.  LDS: three stages of [A_lo | A_hi | W_0 | W_1] = 48 KiB,
// where W_j holds the j-th 32-row block of both N halves (so that a phase's W operand is one slot):
//     phase   reads into registers            DMA requests per wave                         multiplies
//     P1(u)   A(u)               (8 b128)     A_lo(u+2), A_hi(u+2)     (4)   vmcnt(10)      j = 0
//     P2(u)   W_1(u), W_0(u+1)   (4 + 4)      W_0(u+3), W_1(u+2)       (2)   vmcnt(8)       j = 1
// Every request is retired three phases after its issue and read one phase later; a slot is re-requested no earlier
// than two phases after its last read (same rules as gemm8_kernel).
template <int BN, bool M16_ = false>
struct Cfg9 {
  static_assert(BN == 128, "instantiated for the 256 x 128 tile only");
  static constexpr bool M16 = M16_;
  static constexpr int NTHREADS = 512;
  static constexpr int BK = 64, ROW_BYTES = 128;
  static constexpr int WAVES_M = 4, WAVES_N = 2;
  static constexpr int WTM = 64, WTN = 64, MF = 2, NF = 2;
  static constexpr int A_HALF = 128 * ROW_BYTES, W_PART = 64 * ROW_BYTES;
  static constexpr int STAGE_BYTES = 2 * A_HALF + 2 * W_PART;   // 48 KiB
  static constexpr int STAGES = 3;
  static constexpr int CT_LD = BN + 8;
  static constexpr int CT_BYTES = BM * CT_LD * 2;
  static constexpr int SMEM_BYTES = STAGES * STAGE_BYTES > CT_BYTES ? STAGES * STAGE_BYTES : CT_BYTES;
  static FK_DEV int swz(int row) { return (row >> 1) & 7; }
  static FK_DEV int tile_row(int wm, int mf) { return wm * WTM + mf * 32; }
  static FK_DEV int tile_col(int wn, int nf) { return wn * WTN + nf * 32; }
};

template <int EPI, int BN, bool M16 = false>
FK_DEV void gemm9_body(const GroupArgs& ga, char* smem, int pi, int m0, int n0) {
  using C = Cfg9<BN, M16>;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 3, wn = wave >> 2;   // group = wn
  const fk_gemm_args& p = ga.p[pi];
  const int nk = p.K / C::BK;

  const int lrow = lane >> 3, slot = lane & 7;
  const __amdgpu_buffer_rsrc_t rs_a =
      __builtin_amdgcn_make_buffer_rsrc((void*)((const bf16_t*)p.A + fk_row_offset(p.a, m0)), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w =
      __builtin_amdgcn_make_buffer_rsrc((void*)((const bf16_t*)p.W + (int64_t)n0 * p.ldw), 0, 0x7fffffff, 0x00020000);
  int a_voff[2][2], w_voff[2];   // A: [half][piece]; W: [part j], one piece per wave
  {
    const TileRows arow(p.a, m0);
    const int ldw2 = (int)p.ldw * 2;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int rl = h * 128 + (wave * 2 + j) * 8 + lrow;
        a_voff[h][j] = arow.off(min(rl, p.M - 1 - m0)) * 2 + ((slot ^ C::swz(rl)) << 4);
      }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int sr = wave * 8 + lrow;                      // row inside the 64-row part W_j
      const int rl = (sr >> 5) * 64 + j * 32 + (sr & 31);  // row inside the 128-row W tile
      w_voff[j] = min(rl, p.N - 1 - n0) * ldw2 + ((slot ^ C::swz(sr)) << 4);
    }
  }
  auto dma_a = [&](int stage, int kt) {   // both halves of the A tile: 4 requests
    const int koff = min(kt, nk - 1) * (C::BK * 2);
    char* dst = smem + stage * C::STAGE_BYTES + wave * 2048;
    buffer_lds16(rs_a, dst, a_voff[0][0], koff);
    buffer_lds16(rs_a, dst + 1024, a_voff[0][1], koff);
    buffer_lds16(rs_a, dst + C::A_HALF, a_voff[1][0], koff);
    buffer_lds16(rs_a, dst + C::A_HALF + 1024, a_voff[1][1], koff);
  };
  auto dma_w = [&](int j, int stage, int kt) {   // part W_j: 1 request
    const int koff = min(kt, nk - 1) * (C::BK * 2);
    buffer_lds16(rs_w, smem + stage * C::STAGE_BYTES + 2 * C::A_HALF + j * C::W_PART + wave * 1024, w_voff[j], koff);
  };

  constexpr int NKS = M16 ? 2 : 4, ASUB = M16 ? 4 : 2, WSUB = M16 ? 2 : 1;   // as gemm8_body
  constexpr int SUB_BYTES = (M16 ? 16 : 32) * C::ROW_BYTES;
  const int frow = M16 ? (lane & 15) : (lane & 31), fhalf = M16 ? (lane >> 4) : (lane >> 5), fsw = C::swz(frow);
  int koffs[NKS];
#pragma unroll
  for (int kk = 0; kk < NKS; ++kk) koffs[kk] = ((kk * (M16 ? 4 : 2) + fhalf) ^ fsw) << 4;
  const int a_rd = (wm * 64 + frow) * C::ROW_BYTES;                   // rows 64 wm .. of the 256-row A image (+ sub * SUB_BYTES)
  const int w_rd = 2 * C::A_HALF + (wn * 32 + frow) * C::ROW_BYTES;   // inside part W_j (+ j * W_PART)

  f32x16_t acc[C::NF][C::MF];
#pragma unroll
  for (int i = 0; i < C::NF; ++i)
#pragma unroll
    for (int j = 0; j < C::MF; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  bf16x8_t af[ASUB][NKS], wf[2][WSUB][NKS];   // A [sub][kk]; W [j][sub][kk]
  auto read_a = [&](int stage) {
    const char* b = smem + stage * C::STAGE_BYTES + a_rd;
#pragma unroll
    for (int sub = 0; sub < ASUB; ++sub)
#pragma unroll
      for (int kk = 0; kk < NKS; ++kk) af[sub][kk] = *(const bf16x8_t*)(b + sub * SUB_BYTES + koffs[kk]);
  };
  auto read_w = [&](int stage, int j) {
    const char* b = smem + stage * C::STAGE_BYTES + w_rd + j * C::W_PART;
#pragma unroll
    for (int sub = 0; sub < WSUB; ++sub)
#pragma unroll
      for (int kk = 0; kk < NKS; ++kk) wf[j][sub][kk] = *(const bf16x8_t*)(b + sub * SUB_BYTES + koffs[kk]);
  };
  auto mma = [&](int j) {
#pragma unroll
    for (int kk = 0; kk < NKS; ++kk)
#pragma unroll
      for (int mf = 0; mf < 2; ++mf) {
        if constexpr (M16) mma16_block(acc[j][mf], wf[j][0][kk], wf[j][1][kk], af[2 * mf][kk], af[2 * mf + 1][kk]);
        else acc[j][mf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[j][0][kk], af[mf][kk], acc[j][mf], 0, 0, 0);
      }
  };
  auto phase_sync_mma = [&](auto vm, int j) {
    wait_vmcnt<decltype(vm)::value>();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    mma(j);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  using V10 = std::integral_constant<int, 10>;
  using V8 = std::integral_constant<int, 8>;

  // ---- prologue: everything read in phases 0 .. 4, in reading order (13 requests) --------------------------------
  dma_w(0, 0, 0);   // W_0(0)   "phase 0"
  dma_a(0, 0);      // A(0)     P1(0)
  dma_w(1, 0, 0);   // W_1(0)   P2(0)
  dma_w(0, 1, 1);   // W_0(1)   P2(0)
  dma_a(1, 1);      // A(1)     P1(1)
  dma_w(1, 1, 1);   // W_1(1)   P2(1)
  dma_w(0, 2, 2);   // W_0(2)   P2(1)
  wait_vmcnt<8>();  // W_0(0), A(0) landed (own pieces)
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  read_w(0, 0);
  if (wn == 1) __builtin_amdgcn_s_barrier();   // the stagger
  __builtin_amdgcn_sched_barrier(0);

  auto tile_body = [&](auto sc, int kt) {
    constexpr int st = decltype(sc)::value, st1 = (st + 1) % 3, st2 = (st + 2) % 3;
    read_a(st);                        dma_a(st2, kt + 2);                             phase_sync_mma(V10{}, 0);
    read_w(st, 1); read_w(st1, 0);     dma_w(0, st, kt + 3); dma_w(1, st2, kt + 2);    phase_sync_mma(V8{}, 1);
  };
  for (int kt = 0; kt < nk; kt += 3) {
    tile_body(std::integral_constant<int, 0>{}, kt);
    if (kt + 1 < nk) tile_body(std::integral_constant<int, 1>{}, kt + 1);
    if (kt + 2 < nk) tile_body(std::integral_constant<int, 2>{}, kt + 2);
  }
  if (wn == 0) __builtin_amdgcn_s_barrier();   // balance the stagger
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  store_tile<EPI, BN, C>(acc, p, smem, m0, n0, wm, wn);
}

template <int EPI, int BN, bool M16 = false>
__global__ __launch_bounds__(512, 2) void gemm9_kernel(const GroupArgs ga) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int pi, m0, n0;
  select_tile<BN>(ga, xcd_chunk_index(), pi, m0, n0);
  gemm9_body<EPI, BN, M16>(ga, smem, pi, m0, n0);
}

template <int EPI, int BN, bool M16 = false>
int launch9(GroupArgs& ga, const fk_gemm_args* probs, int n, hipStream_t stream) {
  const int total = count_tiles<BN>(ga, probs, n);
  auto kern = gemm9_kernel<EPI, BN, M16>;
  FK_ENSURE_MAX_LDS(kern, Cfg9<BN>::SMEM_BYTES, "fk_gemm_bf16 (256 x 128 tile, 8 waves ping-pong)");
  hipLaunchKernelGGL(kern, dim3(total), dim3(512), Cfg9<BN>::SMEM_BYTES, stream, ga);
  FK_CHECK_LAUNCH("fk_gemm_bf16 (256 x 128 tile, 8 waves ping-pong)");
  return FK_OK;
}

// ---- mixed launch: 256 x 256 tiles for the first big_cols column tiles, 256 x 128 tiles for the rest -----------------
// One workgroup per CU means a grid runs in rounds of #CUs tiles.  At M = 2560 the fused QKV projection (N = 9216) is
// 360 tiles of 256 x 256 (1.41 rounds -> 2) or 720 of 256 x 128 (2.81 -> 3 rounds of 0.61): ONE round of 256 x 256
// tiles followed by the remaining columns as 256 x 128 tiles takes 1 + 0.61 -- 12 % less than the better pure grid.
// Both bodies accumulate over K in the same order, so WHICH tile shape computes an output element does not change its
// bits: the split of the columns is free to follow the grid.  Workgroups are dispatched in blockIdx order, XCD = b % 8:
// each XCD's list is its chunk of the big tiles first, then its chunk of the small ones.
template <int EPI, bool M16 = false>
__global__ __launch_bounds__(512, 2) void gemm_mix_kernel(const GroupArgs ga) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int nbig = ga.xcd_big_cnt[xcd];
  int pi, m0, n0;
  if (idx < nbig) {
    tile_of<256>(ga, ga.tiles_before, ga.xcd_big_start[xcd] + idx, ga.big_cols, 0, pi, m0, n0);
    gemm8_body<EPI, 256, 0, M16>(ga, smem, pi, m0, n0, 0, ga.p[0].K / 64, -1);
  } else {
    tile_of<128>(ga, ga.small_before, ga.xcd_small_start[xcd] + idx - nbig, (ga.p[0].N - ga.big_cols * 256 + 127) / 128,
                 ga.big_cols * 256, pi, m0, n0);
    gemm9_body<EPI, 128, M16>(ga, smem, pi, m0, n0);
  }
}

constexpr int MIX_SMEM = Cfg8<256>::SMEM_BYTES > Cfg9<128>::SMEM_BYTES ? Cfg8<256>::SMEM_BYTES : Cfg9<128>::SMEM_BYTES;

template <int EPI, bool M16 = false>
int launch_mix(GroupArgs& ga, const fk_gemm_args* probs, int n, int big_cols, hipStream_t stream) {
  const int ncols128 = (probs[0].N - big_cols * 256 + 127) / 128;
  int tb = 0, ts = 0;
  for (int i = 0; i < FK_MAX_GROUP; ++i) {
    ga.tiles_before[i] = tb;
    ga.small_before[i] = ts;
    if (i < n) {
      const int nbm = (probs[i].M + BM - 1) / BM;
      tb += nbm * big_cols;
      ts += nbm * ncols128;
    }
  }
  ga.tiles_before[FK_MAX_GROUP] = tb;
  ga.small_before[FK_MAX_GROUP] = ts;
  ga.big_cols = big_cols;
  const int W = tb + ts;
  int bs = 0, ss = 0;
  for (int x = 0; x < 8; ++x) {
    const int wx = W / 8 + (x < W % 8 ? 1 : 0), bx = tb / 8 + (x < tb % 8 ? 1 : 0);
    if (wx < bx) return FK_E2BIG_STRIDES;   // cannot happen for the grids the planner proposes (ts >= 8); caller falls back
    ga.xcd_big_start[x] = bs;
    ga.xcd_big_cnt[x] = bx;
    ga.xcd_small_start[x] = ss;
    bs += bx;
    ss += wx - bx;
  }
  auto kern = gemm_mix_kernel<EPI, M16>;
  FK_ENSURE_MAX_LDS(kern, MIX_SMEM, "fk_gemm_bf16 (mixed 256 x 256 / 256 x 128 tiles)");
  hipLaunchKernelGGL(kern, dim3(W), dim3(512), MIX_SMEM, stream, ga);
  FK_CHECK_LAUNCH("fk_gemm_bf16 (mixed 256 x 256 / 256 x 128 tiles)");
  return FK_OK;
}

// variant: 128 -> gemm9_kernel (256 x 128), 256 -> gemm8_kernel (256 x 256), 384 -> mixed, 512 -> split-K pairs of 256 x 256,
// 640 -> stream-K ranges over 256 x 256 tiles (big_cols carries the grid size)
template <int EPI, bool M16>
int launch_variant_m(GroupArgs& ga, const fk_gemm_args* probs, int n, int variant, int big_cols, hipStream_t stream) {
  switch (variant) {
    case 256: return launch8<EPI, 256, false, 0, M16>(ga, probs, n, stream);
    case 384: return launch_mix<EPI, M16>(ga, probs, n, big_cols, stream);
    case 512: return launch8<EPI, 256, true, 0, M16>(ga, probs, n, stream);
    case 640: return launch8_streamk<EPI, 256, M16>(ga, probs, n, big_cols, stream);
    case 1024: return launch10<EPI>(ga, probs, n, stream);   // 16 x 16 x 32 only: agrees with the M16 forms bit for bit
    default: return launch9<EPI, 128, M16>(ga, probs, n, stream);
  }
}
template <int EPI>
int launch_variant(GroupArgs& ga, const fk_gemm_args* probs, int n, int variant, int big_cols, bool m16, hipStream_t stream) {
  return m16 ? launch_variant_m<EPI, true>(ga, probs, n, variant, big_cols, stream)
             : launch_variant_m<EPI, false>(ga, probs, n, variant, big_cols, stream);
}

int cu_count() {
  static int cus = 0;
  if (!cus) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    else return 256;   // no device visible (build / ABI checks on a CPU box): MI355X
  }
  return cus;
}

// ---- launch plan ----------------------------------------------------------------------------------------------------
// Time of a grid in units of one 256 x 256 tile, workgroups handed to G CUs in launch order (list scheduling):
// `nb` tiles of cost 1 first, then `ns` tiles of cost cs.
double makespan(long nb, long ns, double cs, int G) {
  const long full = nb / G, rem = nb % G;
  // group A: G - rem CUs free at `full`; group B: rem CUs free at full + 1 (absent when rem == 0)
  double ta = (double)full, tb = (double)full + 1.0, end = rem ? tb : ta;
  const long na = G - rem;
  if (nb == 0) end = 0.0;
  while (ns > 0) {
    if (rem == 0 || ta <= tb) {
      const long k = ns < na ? ns : na;
      ta += cs; ns -= k;
      if (ta > end) end = ta;
    } else {
      const long k = ns < rem ? ns : rem;
      tb += cs; ns -= k;
      if (tb > end) end = tb;
    }
  }
  return end;
}

struct Plan { int variant, big_cols; };
constexpr int SK_MIN_PART = 16;   // stream-K: the shortest part of a tile's K range a cut may leave, in K-tiles of 64

// nbm: row tiles summed over the problems of the launch; N, K shared.  allow: bit 0 mixed, bit 1 split-K.
Plan plan_launch(long nbm, int N, int K, int G, int allow, bool have_ws, int ws_slots) {
  const double c128 = 0.5 * FK_RATE_256;   // a 256 x 128 tile in units of a 256 x 256 tile (measured rate ratio)
  const long t128 = nbm * ((N + 127) / 128), t256 = nbm * ((N + 255) / 256);
  Plan best = {128, 0};
  double tbest = makespan(0, t128, c128, G);
  if (N % 256 != 0) return best;
  const double t_256 = makespan(t256, 0, c128, G);
  if (t_256 < tbest) { tbest = t_256; best = {256, 0}; }
  if (allow & 1) {
    const int nb256 = N / 256;
    for (int cb = 1; cb < nb256; ++cb) {
      const long ts = nbm * 2 * (nb256 - cb);
      if (ts < 8) continue;
      const double t = makespan(nbm * cb, ts, c128, G);
      if (t < tbest * 0.97) { tbest = t; best = {384, cb}; }   // a mixed grid must pay for its larger kernel image
    }
  }
  if ((allow & 2) && have_ws && K >= 6144 && (K / 64) % 4 == 0 && 2 * t256 <= G && t256 <= ws_slots) {
    // two half-K workgroups per tile in ONE round; the exchange costs ~17 us (256 KiB written through, read back, two
    // barriers and a fence) against K x 25.8 ns for a whole tile
    const double t = 0.5 + 17.0e-6 / (K * 25.8e-9);
    if (t < tbest * 0.97) { tbest = t; best = {512, 0}; }
  }
  // MEASURED AND OFF BY DEFAULT (allow bit 2, FK_GEMM_PLAN=7): inside the 1024^2 edit the stream-K form of the two long-K launch
  // classes (408 tiles = 1.59 rounds) LOSES 1.8 % of the GEMM time against the plain two-round grid (2643 vs 2596 ms per edit,
  // same box, interleaved; train step 531.6 vs 525.9 ms): the plain grid's half-empty second round already runs 6 % below the
  // full-grid rate only (idle CUs hand their power budget to the busy ones), and one 256 KiB exchange per CU plus the
  // persistent loop's per-pass prologues cost more than that (profiles/r04_gemm_streamk_ab.txt)
  if ((allow & 4) && have_ws && K >= 6144 && t256 > G && G <= ws_slots &&
      t256 * (K / 64) >= (long)G * (K / 64 + 2 * SK_MIN_PART) && t256 * (long)(K / 64) < (1l << 31)) {
    // stream-K ranges: every CU the same share of the K-tiles, one exchange per CU (a tile is cut at most once)
    const double t = (double)t256 / G + 17.0e-6 / (K * 25.8e-9);
    if (t < tbest * 0.97) { tbest = t; best = {640, G}; }
  }
  return best;
}
}  // namespace

// fk_gemm_args.variant_used (OUT, optional): 128 / 256: the 256 x 128 / 256 x 256 kernel, 384: mixed grid, 512: split-K pairs,
// 640: stream-K ranges.  The library keeps no record of its own.
static inline void report_variant(const fk_gemm_args* probs, int v) {
  if (probs[0].variant_used) *probs[0].variant_used = v;
}

// MFMA shape of the layout-0 large-tile kernels: 32 = v_mfma_f32_32x32x16_bf16, 16 = v_mfma_f32_16x16x32_bf16 (FragMap above).
// The two differ in the last bits (16 against 32 products per hardware sum); every launch form of ONE shape agrees bit
// for bit with the others.  Per call: fk_gemm_args.mfma (0 = default); the K-major layouts (1, 2) follow it since round 6.
// Default 16 (round 5): +3.1 .. +4.1 % on every launch form of the M = 2560 QKV / MLP-up shapes, interleaved in one process
// (profiles/r05_gemm_mfma_ab.txt); one wave per SIMD issues the 4-pass instruction every 17.3 cycles instead of 16, which is
// why the ping-pong kernels keep 2/3 of the pure-MFMA stream's 12 %.
#ifndef FK_GEMM_MFMA_DEFAULT
#define FK_GEMM_MFMA_DEFAULT 16
#endif
constexpr int GEMM_MFMA_DEFAULT = FK_GEMM_MFMA_DEFAULT;

// Used by fk_gemm_bf16 / fk_gemm_bf16_grouped after argument validation.
// variant_hint: 128 / 256 / 384 / 512 force a launch form where it is applicable; 0 = choose per problem.  The 256 x 256
// kernel has the higher steady-state rate (measured 1.1-1.2x for large grids), but one workgroup per CU means the grid
// runs in rounds of #CUs tiles: plan_launch picks the form with the shortest list-scheduling makespan.  (A stream-K form
// of the 256 x 256 kernel that shares the last round's K-iterations among all CUs was built in round 1 and measured 2x
// slower: DESIGN.md section 4b; git history.)
int fk_gemm2_launch(const fk_gemm_args* probs, int n, int variant_hint, hipStream_t stream) {
  // launch controls come with the call (fk_gemm_args.variant / plan / group_m / mfma of the first problem): no process state
  const fk_gemm_args& ctl = probs[0];
  if (variant_hint == 0) variant_hint = ctl.variant;
  if (!(variant_hint == 0 || variant_hint == 128 || variant_hint == 256 || variant_hint == 384 || variant_hint == 512 || variant_hint == 640 ||
        variant_hint == 1024)) {
    fk_set_error("fk_gemm_bf16: variant %d is not one of 0 (launch plan), 128, 256, 384 (mixed), 512 (split-K pairs), 640 (stream-K ranges), "
                 "1024 (4-wave hand-placed 256 x 256 kernel)", variant_hint);
    return FK_EINVAL;
  }
  if (ctl.plan != 0 && (ctl.plan & ~15) != 0 || (ctl.plan != 0 && !(ctl.plan & FK_GEMM_PLAN_EXPLICIT))) {
    fk_set_error("fk_gemm_bf16: plan %d is not 0 (default) or FK_GEMM_PLAN_EXPLICIT | allow bits 0..2", ctl.plan);
    return FK_EINVAL;
  }
  if (ctl.group_m < 0 || ctl.group_m > 4096 || !(ctl.mfma == 0 || ctl.mfma == 16 || ctl.mfma == 32)) {
    fk_set_error("fk_gemm_bf16: group_m %d must be 0 (default) or 1..4096, mfma %d one of 0 / 16 / 32", ctl.group_m, ctl.mfma);
    return FK_EINVAL;
  }
  const int plan_allow = ctl.plan ? (ctl.plan & 7) : 3;
  GroupArgs ga;
  ga.n = n;
  ga.group_m = ctl.group_m ? ctl.group_m : GROUP_M;
  ga.big_cols = 0;
  ga.sk_partials = nullptr;
  ga.sk_ctl = nullptr;
  long nbm_total = 0;
  bool ok256 = probs[0].N % 256 == 0, ok32 = true;   // ok32: a tile's rows are addressable with 32-bit byte offsets
  for (int i = 0; i < FK_MAX_GROUP; ++i) {
    ga.p[i] = probs[i < n ? i : 0];
    if (i < n) {
      nbm_total += (probs[i].M + BM - 1) / BM;
      // both kernels address a tile's rows with 32-bit byte offsets from the tile's first row
      auto span = [](const fk_rows& r) {
        const long long ld = r.ld < 0 ? -r.ld : r.ld, bs = r.batch_stride < 0 ? -r.batch_stride : r.batch_stride;
        return (BM * ld + (r.rows_per_batch > 0 ? bs : 0)) * 2;
      };
      if ((long long)BM * probs[i].ldw * 2 >= (1ll << 31) || probs[i].ldw < 0) ok32 = false;
      const bool has_res = probs[i].epilogue == FK_EPI_GATE_RES || probs[i].epilogue == FK_EPI_RES;
      const fk_rows* ops[3] = {&probs[i].a, &probs[i].c, has_res ? &probs[i].r : nullptr};
      for (const fk_rows* r : ops) {
        if (!r) continue;
        if (span(*r) >= (1ll << 31) || r->ld < 0) ok32 = false;
        // rows of the next batch must lie after this batch's (offsets relative to a tile's first row stay >= 0)
        if (r->rows_per_batch > 0 && r->batch_stride < r->rows_per_batch * r->ld) ok32 = false;
      }
    }
  }
  if (!ok32) return FK_E2BIG_STRIDES;   // caller falls back to the 128 x 128 kernel (64-bit addressing)

  // K-major operands (layout 1: W [K, N]; layout 2: A [K, M] too): the 256 x 256 kernel only, whole tiles only
  const int lay = probs[0].layout;
  if (lay != 0) {
    for (int i = 0; i < n; ++i) {
      const fk_gemm_args& q = probs[i];
      const bool uniform_a = q.a.rows_per_batch <= 0 || q.a.batch_stride == q.a.rows_per_batch * q.a.ld;
      if (q.layout != lay || q.N % 256 != 0 || q.K % 64 != 0 || q.ldw % 8 != 0 || (int64_t)q.K * q.ldw * 2 >= (1ll << 31) ||
          (lay == 2 && (q.M % 256 != 0 || !uniform_a || q.a.ld % 8 != 0 || (int64_t)q.K * q.a.ld * 2 >= (1ll << 31))) ||
          (q.epilogue != FK_EPI_NONE && !(lay == 1 && q.epilogue == FK_EPI_RES))) {
        fk_set_error("fk_gemm_bf16: layout %d needs N %% 256 == 0, K %% 64 == 0%s, epilogue none%s and operands below 2 GiB "
                     "(M %d N %d K %d epilogue %d)", lay, lay == 2 ? ", M % 256 == 0, uniformly strided A rows" : "",
                     lay == 1 ? " / residual" : "", q.M, q.N, q.K, q.epilogue);
        return FK_EUNSUPPORTED;
      }
    }
    report_variant(probs, 256);
    // round 6: the K-major forms follow fk_gemm_args.mfma like layout 0 (FK_KMAJOR_MFMA=32, read once: the forms of rounds 3-5 for the A/B)
    static const bool km32 = getenv("FK_KMAJOR_MFMA") && atoi(getenv("FK_KMAJOR_MFMA")) == 32;
    if (!km32 && (ctl.mfma ? ctl.mfma : GEMM_MFMA_DEFAULT) == 16) {
      if (lay == 1 && probs[0].epilogue == FK_EPI_RES) return launch8<FK_EPI_RES, 256, false, 1, true>(ga, probs, n, stream);
      if (lay == 1) return launch8<FK_EPI_NONE, 256, false, 1, true>(ga, probs, n, stream);
      if (lay == 2) return launch8<FK_EPI_NONE, 256, false, 2, true>(ga, probs, n, stream);
    }
    if (lay == 1 && probs[0].epilogue == FK_EPI_RES) return launch8<FK_EPI_RES, 256, false, 1>(ga, probs, n, stream);
    if (lay == 1) return launch8<FK_EPI_NONE, 256, false, 1>(ga, probs, n, stream);
    if (lay == 2) return launch8<FK_EPI_NONE, 256, false, 2>(ga, probs, n, stream);
    fk_set_error("fk_gemm_bf16: unknown layout %d", lay);
    return FK_EUNSUPPORTED;
  }

  // split-K workspace (optional, caller-owned): slots x 256 KiB of fp32 partial tiles, then slots x 2 control words
  const int ws_slots = probs[0].splitk_ws ? probs[0].splitk_slots : 0;
  const int N = probs[0].N, K = probs[0].K;
  const int G = cu_count();
  const long t256 = nbm_total * ((N + 255) / 256);
  const bool sk_ok = ws_slots > 0 && ok256 && (K / 64) % 4 == 0 && t256 <= ws_slots;   // split-K pairs
  Plan plan;
  switch (variant_hint) {
    case 128: plan = {128, 0}; break;
    case 256: plan = {ok256 ? 256 : 128, 0}; break;
    case 384: {
      plan = plan_launch(nbm_total, N, K, G, 1, false, 0);
      if (plan.variant != 384) {   // forced: the best split of the columns even where it does not pay
        plan = {128, 0};
        if (ok256 && N >= 512) {
          double tb = 1e30;
          for (int cb = 1; cb < N / 256; ++cb) {
            if (nbm_total * 2 * (N / 256 - cb) < 8) continue;
            const double t = makespan(nbm_total * cb, nbm_total * 2 * (N / 256 - cb), 0.5 * FK_RATE_256, G);
            if (t < tb) { tb = t; plan = {384, cb}; }
          }
        }
      }
      break;
    }
    case 512: plan = {sk_ok ? 512 : (ok256 ? 256 : 128), 0}; break;
    case 1024: plan = {ok256 ? 1024 : 128, 0}; break;   // gemm10_kernel: plain grid of 256 x 256 tiles
    case 640: {   // forced: stream-K ranges wherever every tile would be cut at most once (test hook: also on small grids)
      const int Gs = G < ws_slots ? G : ws_slots;
      const bool ok = ws_slots > 0 && ok256 && t256 * (K / 64) >= (long)Gs * (K / 64 + 2 * SK_MIN_PART) && t256 * (long)(K / 64) < (1l << 31);
      plan = ok ? Plan{640, Gs} : Plan{ok256 ? 256 : 128, 0};
      break;
    }
    default: plan = plan_launch(nbm_total, N, K, G, plan_allow, ws_slots > 0 && ok256, ws_slots); break;
  }
  ga.sk_min_part = SK_MIN_PART;
  if (plan.variant == 512 || plan.variant == 640) {
    ga.sk_partials = (float*)probs[0].splitk_ws;
    ga.sk_ctl = (unsigned*)((char*)probs[0].splitk_ws + (size_t)ws_slots * (BM * 256 * 4));
  }
  const bool m16 = (ctl.mfma ? ctl.mfma : GEMM_MFMA_DEFAULT) == 16;
  // the launch plan's pure 256 x 256 grid as gemm10_kernel where that kernel measures ahead of gemm8_kernel: long K (the
  // per-tile prologue / epilogue of a one-wave-per-SIMD kernel is ~20 k cycles: 4 % of a K = 12288 tile, 15 % of a K = 3072
  // one) and a grid deep enough for its one-workgroup-per-CU form; same bits either way (FK_GEMM10=0: never)
  static const bool g10_auto = !(getenv("FK_GEMM10") && atoi(getenv("FK_GEMM10")) == 0);
  if (plan.variant == 256 && variant_hint == 0 && m16 && g10_auto && K >= 6144 && t256 >= 4L * G) plan.variant = 1024;
  report_variant(probs, plan.variant);
  int rc;
  switch (probs[0].out_fp32 == 2 ? FK_EPI_F32DBG : probs[0].epilogue) {
    case FK_EPI_F32DBG: rc = launch_variant<FK_EPI_F32DBG>(ga, probs, n, plan.variant, plan.big_cols, m16, stream); break;
    case FK_EPI_NONE: rc = launch_variant<FK_EPI_NONE>(ga, probs, n, plan.variant, plan.big_cols, m16, stream); break;
    case FK_EPI_GELU_TANH: rc = launch_variant<FK_EPI_GELU_TANH>(ga, probs, n, plan.variant, plan.big_cols, m16, stream); break;
    case FK_EPI_SILU: rc = launch_variant<FK_EPI_SILU>(ga, probs, n, plan.variant, plan.big_cols, m16, stream); break;
    case FK_EPI_GATE_RES: rc = launch_variant<FK_EPI_GATE_RES>(ga, probs, n, plan.variant, plan.big_cols, m16, stream); break;
    case FK_EPI_RES: rc = launch_variant<FK_EPI_RES>(ga, probs, n, plan.variant, plan.big_cols, m16, stream); break;
    case FK_EPI_SCALE: rc = launch_variant<FK_EPI_SCALE>(ga, probs, n, plan.variant, plan.big_cols, m16, stream); break;
    case FK_EPI_QKV: rc = launch_variant<FK_EPI_QKV>(ga, probs, n, plan.variant, plan.big_cols, m16, stream); break;
    default: fk_set_error("fk_gemm_bf16: unknown epilogue %d", probs[0].epilogue); return FK_EUNSUPPORTED;
  }
  if (rc == FK_E2BIG_STRIDES && plan.variant == 384) {   // degenerate XCD split of a mixed grid: plain 256 x 128 grid
    report_variant(probs, 128);
    return fk_gemm2_launch(probs, n, 128, stream);
  }
  return rc;
}
