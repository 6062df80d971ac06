// Large-tile bf16 MFMA GEMM for the MMDiT linears (second-generation kernel; same contract as
// gemm_bf16.hip, used when M is large enough to fill 256-row tiles).
//
//   tile 256(M) x BN(N) x 64(K), 512 threads = 8 waves, v_mfma_f32_32x32x16_bf16, operands swapped like
//   gemm_bf16.hip (W rows -> MFMA A operand) so a lane owns 4 consecutive output columns of one row.
//     BN = 128: waves 4(M) x 2(N), 64x64 per wave (2x2 accumulators), 3-stage LDS ring (3 x 48 KiB)
//     BN = 256: waves 2(M) x 4(N), 128x64 per wave (4x2 accumulators), 2-stage LDS ring (2 x 64 KiB)
//
// Staging is LDS-DMA (`global_load_lds_dwordx4`: no VGPR round trip, no ds_write pass).  Tile kt+PF
// (PF = stages-1) is issued while tile kt is multiplied -- the DMA instructions are spread over the four
// k-steps of the tile, between the MFMA groups, instead of in one burst after the barrier -- and each wave
// waits with a COUNTED `s_waitcnt vmcnt(N)` so that with 3 stages a whole tile stays in flight across the
// raw `s_barrier`.  One barrier per K-tile.  The LDS image of a DMA instruction is lane-linear (8 rows x
// 128 B), so the bank-conflict swizzle (16-byte chunk ^ ((row >> 1) & 7)) is applied to the per-lane SOURCE
// address and again on the ds_read_b128 side (cdna guide rule 21).  Fragments of k-step kk+1 are read
// while k-step kk multiplies.
//
// Measured on MI355X (tools/): MFMA-only ceiling of this loop (LDS-DMA disabled) is 1.40 PF/s
// (BN=128) / 1.56 PF/s (BN=256); with the DMA it reaches 1.05 / 1.20.  Fetching the same bytes into registers
// instead costs only ~7 %, and the L2 -> LDS fill path alone sustains 23 TB/s (tests/probes/probe_fill.hip), so
// the loss is LDS-port contention between DMA writes and fragment reads, not the fetch.  Dead ends tried and
// removed (git history): phase-staggered waves ("8-phase", BK=32 ring, setprio): equal at BN=256, -15 % at
// BN=128; A operand loaded straight to registers in fragment layout: 32-byte row pieces run the TA at half
// speed (450 TF/s).
//
// Up to FK_MAX_GROUP problems with identical (N, K, epilogue) share one launch ("grouped GEMM"): the
// text- and image-stream linears of a double block become one grid, which fills the 256 CUs at batch 1.
#include "fk_common.h"

namespace {

constexpr int BM = 256, BK = 64;
constexpr int NT = 512;
#ifndef FK_GROUP_M
#define FK_GROUP_M 8
#endif
constexpr int GROUP_M = FK_GROUP_M;
#ifndef FK_GEMM_SETPRIO
#define FK_GEMM_SETPRIO 0
#endif
constexpr bool SETPRIO = FK_GEMM_SETPRIO;

struct GroupArgs {
  fk_gemm_args p[FK_MAX_GROUP];
  int tiles_before[FK_MAX_GROUP + 1];  // prefix sums of tile counts
  int n;
};

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;

FK_DEV void glds16(const bf16_t* src, char* lds_dst) {
  __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)lds_dst, 16, 0, 0);
}

template <int BN>
struct Cfg {
  static constexpr int WAVES_M = (BN == 128) ? 4 : 2;
  static constexpr int WAVES_N = 8 / WAVES_M;
  static constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;  // wave tile
  static constexpr int MF = WTM / 32, NF = WTN / 32;
  static constexpr int A_BYTES = BM * BK * 2, W_BYTES = BN * BK * 2;
  static constexpr int STAGE_BYTES = A_BYTES + W_BYTES;
  static constexpr int STAGES = (BN == 128) ? 3 : 2;
  static constexpr int PF = STAGES - 1;
  static constexpr int A_LOADS = BM / 64, W_LOADS = BN / 64;  // DMA instructions per wave per tile
  static constexpr int LOADS = A_LOADS + W_LOADS;
  static constexpr int CT_LD = BN + 8;
  static constexpr int CT_BYTES = BM * CT_LD * 2;
  static constexpr int SMEM_BYTES = (STAGES * STAGE_BYTES > CT_BYTES) ? STAGES * STAGE_BYTES : CT_BYTES;
};

template <int N>
FK_DEV void wait_vmcnt() {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else static_assert(N == 0, "add the vmcnt literal");
}

template <int EPI, int BN>
__global__ __launch_bounds__(NT, 2) void gemm2_kernel(const GroupArgs ga) {
  using C = Cfg<BN>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave % C::WAVES_M, wn = wave / C::WAVES_M;

  // ---- tile selection: XCD chunking over the whole grid, then problem, then grouped order ------------
  int t;
  {
    const int nwg = gridDim.x;
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  int pi = 0;
#pragma unroll
  for (int i = 1; i < FK_MAX_GROUP; ++i)
    if (i < ga.n && t >= ga.tiles_before[i]) pi = i;
  const fk_gemm_args& p = ga.p[pi];
  t -= ga.tiles_before[pi];
  const int nbm = (p.M + BM - 1) / BM, nbn = (p.N + BN - 1) / BN;
  int tm, tn;
  {
    const int per_group = GROUP_M * nbn;
    const int g = t / per_group;
    const int first_m = g * GROUP_M;
    const int gm = min(nbm - first_m, GROUP_M);
    const int rem = t - g * per_group;
    tm = first_m + rem % gm;
    tn = rem / gm;
  }
  const int m0 = tm * BM, n0 = tn * BN;

  // ---- LDS-DMA sources: lane -> (row = base + lane/8, slot = lane%8), source chunk = slot ^ f(row) -----
  const int lrow = lane >> 3, slot = lane & 7;
  const bf16_t* a_src[C::A_LOADS];
  const bf16_t* w_src[C::W_LOADS];
#pragma unroll
  for (int j = 0; j < C::A_LOADS; ++j) {
    const int rl = (wave * C::A_LOADS + j) * 8 + lrow;  // row inside the A tile
    const int m = min(m0 + rl, p.M - 1);
    a_src[j] = (const bf16_t*)p.A + fk_row_offset(p.a, m) + ((slot ^ ((rl >> 1) & 7)) << 3);
  }
#pragma unroll
  for (int j = 0; j < C::W_LOADS; ++j) {
    const int rl = (wave * C::W_LOADS + j) * 8 + lrow;
    const int n = min(n0 + rl, p.N - 1);
    w_src[j] = (const bf16_t*)p.W + (int64_t)n * p.ldw + ((slot ^ ((rl >> 1) & 7)) << 3);
  }
  // piece i of the tile's DMA list (A pieces first); `koff` = element offset of the K-tile
  auto issue_piece = [&](int i, int64_t koff, char* sb) {
    if (i < C::A_LOADS) glds16(a_src[i] + koff, sb + (wave * C::A_LOADS + i) * 1024);
    else glds16(w_src[i - C::A_LOADS] + koff, sb + C::A_BYTES + (wave * C::W_LOADS + (i - C::A_LOADS)) * 1024);
  };

  // ---- MFMA operand addressing (same swizzle on the read side) -----------------------------------------
  const int frow = lane & 31, fhalf = lane >> 5, fsw = (frow >> 1) & 7;
  const int a_rd = (wm * C::WTM + frow) * 128;               // + mf*4096
  const int w_rd = C::A_BYTES + (wn * C::WTN + frow) * 128;  // + nf*4096

  f32x16_t acc[C::NF][C::MF];
#pragma unroll
  for (int i = 0; i < C::NF; ++i)
#pragma unroll
    for (int j = 0; j < C::MF; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = p.K / BK;
#pragma unroll
  for (int s = 0; s < C::PF; ++s)
    if (s < nk) {
#pragma unroll
      for (int i = 0; i < C::LOADS; ++i) issue_piece(i, (int64_t)s * BK, smem + s * C::STAGE_BYTES);
    }

  int st_cur = 0, st_pf = C::PF;  // stage of tile kt, stage receiving tile kt+PF
  // 3 stages: pieces spread over k-steps 0..2; 2 stages: all pieces in k-steps 0..1 so the last one
  // still has two k-steps of MFMA work to land under before the next tile's vmcnt(0)
  constexpr int PER_KK = (C::STAGES == 2) ? (C::LOADS + 1) / 2 : (C::LOADS + 2) / 3;
  for (int kt = 0; kt < nk; ++kt) {
    // tile kt has landed once at most (PF-1) newer tiles of this wave remain outstanding
    if (kt + C::PF - 1 < nk) wait_vmcnt<(C::PF - 1) * C::LOADS>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();

    const char* sb = smem + st_cur * C::STAGE_BYTES;
    char* sb_pf = smem + st_pf * C::STAGE_BYTES;
    const bool do_pf = kt + C::PF < nk;
    const int64_t koff_pf = (int64_t)(kt + C::PF) * BK;

    bf16x8_t af[2][C::MF], wf[2][C::NF];
    {
      const int coff = ((fhalf ^ fsw) << 4);
#pragma unroll
      for (int mf = 0; mf < C::MF; ++mf) af[0][mf] = *(const bf16x8_t*)(sb + a_rd + mf * 4096 + coff);
#pragma unroll
      for (int nf = 0; nf < C::NF; ++nf) wf[0][nf] = *(const bf16x8_t*)(sb + w_rd + nf * 4096 + coff);
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int cb = kk & 1, nb = cb ^ 1;
      if (kk < 3) {
        const int coff = ((((kk + 1) * 2 + fhalf) ^ fsw) << 4);
#pragma unroll
        for (int mf = 0; mf < C::MF; ++mf) af[nb][mf] = *(const bf16x8_t*)(sb + a_rd + mf * 4096 + coff);
#pragma unroll
        for (int nf = 0; nf < C::NF; ++nf) wf[nb][nf] = *(const bf16x8_t*)(sb + w_rd + nf * 4096 + coff);
      }
      if (do_pf) {
#pragma unroll
        for (int i = kk * PER_KK; i < (kk + 1) * PER_KK && i < C::LOADS; ++i) issue_piece(i, koff_pf, sb_pf);
      }
      if constexpr (SETPRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int nf = 0; nf < C::NF; ++nf)
#pragma unroll
        for (int mf = 0; mf < C::MF; ++mf)
          acc[nf][mf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[cb][nf], af[cb][mf], acc[nf][mf], 0, 0, 0);
      if constexpr (SETPRIO) __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
    }
    st_cur = (st_cur == C::STAGES - 1) ? 0 : st_cur + 1;
    st_pf = (st_pf == C::STAGES - 1) ? 0 : st_pf + 1;
  }

  // ---- epilogue (as gemm_bf16.hip): bias/activation -> bf16 -> LDS tile -> coalesced 16-byte rows -------
  __syncthreads();  // every wave is done reading the last stage before the C tile aliases it
  bf16_t* ct = (bf16_t*)smem;
#pragma unroll
  for (int nf = 0; nf < C::NF; ++nf)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int nl = wn * C::WTN + nf * 32 + 8 * q + 4 * fhalf;
      const int n = n0 + nl;
      float b[4] = {0.f, 0.f, 0.f, 0.f};
      if constexpr (EPI != FK_EPI_SCALE) {
        if (p.bias && n < p.N) {
          const u32x2_t bw = *(const u32x2_t*)((const bf16_t*)p.bias + n);
          b[0] = bf_lo(bw[0]); b[1] = bf_hi(bw[0]); b[2] = bf_lo(bw[1]); b[3] = bf_hi(bw[1]);
        }
      }
#pragma unroll
      for (int mf = 0; mf < C::MF; ++mf) {
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float x = acc[nf][mf][q * 4 + j];
          if constexpr (EPI == FK_EPI_SCALE) x = x * p.alpha;
          else x = x + b[j];
          if constexpr (EPI == FK_EPI_GELU_TANH) x = gelu_tanh_f(round_bf(x));
          if constexpr (EPI == FK_EPI_SILU) x = silu_f(round_bf(x));
          v[j] = x;
        }
        u32x2_t pk;
        pk[0] = pack_bf2(v[0], v[1]);
        pk[1] = pack_bf2(v[2], v[3]);
        const int ml = wm * C::WTM + mf * 32 + frow;
        *(u32x2_t*)(ct + ml * C::CT_LD + nl) = pk;
      }
    }
  __syncthreads();
  constexpr int CPR = BN / 8;  // 16-byte chunks per tile row
#pragma unroll
  for (int j = 0; j < BM * CPR / NT; ++j) {
    const int id = tid + NT * j;
    const int ml = id / CPR, cc = id % CPR;
    const int m = m0 + ml, n = n0 + cc * 8;
    if (m >= p.M || n >= p.N) continue;
    u32x4_t y = *(const u32x4_t*)(ct + ml * C::CT_LD + cc * 8);
    if constexpr (EPI == FK_EPI_QKV) {
      // 16 consecutive lanes hold one 128-wide head row of the tile (tiles never straddle q | k | v)
      const int D = p.qkv_heads * 128;
      const int which = n / D;  // 0 = q, 1 = k, 2 = v
      if (which < 2) {
        float xv[8];
        float ss = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          xv[2 * e] = bf_lo(y[e]);
          xv[2 * e + 1] = bf_hi(y[e]);
          ss += xv[2 * e] * xv[2 * e] + xv[2 * e + 1] * xv[2 * e + 1];
        }
#pragma unroll
        for (int off = 8; off >= 1; off >>= 1) ss += __shfl_xor(ss, off);
        const float rs = rsqrtf(ss * (1.0f / 128) + 1e-6f);
        const int rpb = p.c.rows_per_batch > 0 ? (int)p.c.rows_per_batch : p.M;
        const int bidx = m / rpb;
        const int srow = p.qkv_s_offset + (m - bidx * rpb);
        const int hn = n - which * D;            // column inside q or k
        const int head = hn >> 7, dch = hn & 127;  // dch = 8 * chunk-in-head
        const u32x4_t ww = *(const u32x4_t*)((const bf16_t*)(which == 0 ? p.wq : p.wk) + dch);
        const float* cp = p.rope_cos + (int64_t)srow * 128 + dch;
        const float* sp = p.rope_sin + (int64_t)srow * 128 + dch;
        const f32x4_t c0 = *(const f32x4_t*)cp, c1 = *(const f32x4_t*)(cp + 4);
        const f32x4_t s0 = *(const f32x4_t*)sp, s1 = *(const f32x4_t*)(sp + 4);
        const float cs[8] = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
        const float sn[8] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
        u32x4_t ow;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float re = round_bf(round_bf(xv[2 * e] * rs) * bf_lo(ww[e]));
          const float im = round_bf(round_bf(xv[2 * e + 1] * rs) * bf_hi(ww[e]));
          const float o0 = __fadd_rn(__fmul_rn(re, cs[2 * e]), __fmul_rn(-im, sn[2 * e]));
          const float o1 = __fadd_rn(__fmul_rn(im, cs[2 * e + 1]), __fmul_rn(re, sn[2 * e + 1]));
          ow[e] = pack_bf2(o0, o1);
        }
        bf16_t* dst = (bf16_t*)(which == 0 ? p.q_out : p.k_out);
        *(u32x4_t*)(dst + (((int64_t)bidx * p.qkv_heads + head) * p.qkv_s_total + srow) * 128 + dch) = ow;
        continue;
      }
    }
    if constexpr (EPI == FK_EPI_GATE_RES || EPI == FK_EPI_RES) {
      const u32x4_t rv = *(const u32x4_t*)((const bf16_t*)p.res + fk_row_offset(p.r, m) + n);
      u32x4_t gv;
      if constexpr (EPI == FK_EPI_GATE_RES) {
        const int64_t b = m / p.gate_rows_per_batch;
        gv = *(const u32x4_t*)((const bf16_t*)p.gate + b * p.gate_batch_stride + n);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float y0 = bf_lo(y[e]), y1 = bf_hi(y[e]);
        if constexpr (EPI == FK_EPI_GATE_RES) {
          y0 = round_bf(bf_lo(gv[e]) * y0);
          y1 = round_bf(bf_hi(gv[e]) * y1);
        }
        y[e] = pack_bf2(bf_lo(rv[e]) + y0, bf_hi(rv[e]) + y1);
      }
    }
    *(u32x4_t*)((bf16_t*)p.C + fk_row_offset(p.c, m) + n) = y;
  }
}

template <int EPI, int BN>
int launch(GroupArgs& ga, const fk_gemm_args* probs, int n, hipStream_t stream) {
  int total = 0;
  for (int i = 0; i < FK_MAX_GROUP; ++i) {
    ga.tiles_before[i] = total;
    if (i < n) total += ((probs[i].M + BM - 1) / BM) * ((probs[i].N + BN - 1) / BN);
  }
  ga.tiles_before[FK_MAX_GROUP] = total;
  auto kern = gemm2_kernel<EPI, BN>;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg<BN>::SMEM_BYTES);
    attr_done = true;
  }
  hipLaunchKernelGGL(kern, dim3(total), dim3(NT), Cfg<BN>::SMEM_BYTES, stream, ga);
  FK_CHECK_LAUNCH("fk_gemm_bf16 (256-row tile)");
  return FK_OK;
}

template <int EPI>
int launch_bn(GroupArgs& ga, const fk_gemm_args* probs, int n, int bn, hipStream_t stream) {
  return bn == 256 ? launch<EPI, 256>(ga, probs, n, stream) : launch<EPI, 128>(ga, probs, n, stream);
}

}  // namespace

// Used by fk_gemm_bf16 / fk_gemm_bf16_grouped after argument validation.
// bn_hint: 128 / 256 force the N tile; 0 = choose per problem.  The 256x256 tile has 1.5x the flop/byte of
// 256x128 (measured ~1.18x the steady-state rate: the L2 -> LDS fill path is what limits these kernels), but
// one workgroup per CU means the grid runs in rounds of 256 tiles: pick the tile with the better
// (quantisation efficiency) x (steady-state rate).
int fk_gemm2_launch(const fk_gemm_args* probs, int n, int bn_hint, hipStream_t stream) {
  GroupArgs ga;
  ga.n = n;
  long t128 = 0, t256 = 0;
  for (int i = 0; i < FK_MAX_GROUP; ++i) {
    ga.p[i] = probs[i < n ? i : 0];
    if (i < n) {
      const long nbm = (probs[i].M + BM - 1) / BM;
      t128 += nbm * ((probs[i].N + 127) / 128);
      t256 += nbm * ((probs[i].N + 255) / 256);
    }
  }
  int bn = bn_hint;
  if (bn != 128 && bn != 256) {
    auto eff = [](long tiles) { return (double)tiles / (double)(((tiles + 255) / 256) * 256); };
    const bool ok256 = probs[0].N % 256 == 0;
    bn = (ok256 && 1.18 * eff(t256) > eff(t128)) ? 256 : 128;
  }
  switch (probs[0].epilogue) {
    case FK_EPI_NONE: return launch_bn<FK_EPI_NONE>(ga, probs, n, bn, stream);
    case FK_EPI_GELU_TANH: return launch_bn<FK_EPI_GELU_TANH>(ga, probs, n, bn, stream);
    case FK_EPI_SILU: return launch_bn<FK_EPI_SILU>(ga, probs, n, bn, stream);
    case FK_EPI_GATE_RES: return launch_bn<FK_EPI_GATE_RES>(ga, probs, n, bn, stream);
    case FK_EPI_RES: return launch_bn<FK_EPI_RES>(ga, probs, n, bn, stream);
    case FK_EPI_SCALE: return launch_bn<FK_EPI_SCALE>(ga, probs, n, bn, stream);
    case FK_EPI_QKV: return launch_bn<FK_EPI_QKV>(ga, probs, n, bn, stream);
    default: fk_set_error("fk_gemm_bf16: unknown epilogue %d", probs[0].epilogue); return FK_EUNSUPPORTED;
  }
}
