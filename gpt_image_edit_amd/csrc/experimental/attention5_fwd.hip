// Joint attention forward, third schedule: four waves (one per SIMD), 32 query rows per wave, pipelined over KV tiles.
//
// EXPERIMENTAL (FK_ATTN_VARIANT=45).  Same contract, operand layouts and numerics as attention4_fwd.hip (fixed per-row
// exponent reference instead of a running rescale, exact K-only restart when a row outgrows it, padded LDS rows with
// immediate-offset addressing, buffer_load -> VGPR -> ds_write_b128 staging), but a quarter of its register demand:
// one query block per wave, and the overlap of matrix and vector work comes from running the softmax of tile u+1
// under the MFMAs of tile u:
//
//   iteration u, slot 1   MFMA  S(u+1) = K(u+1) Q^T      |  VALU  softmax(u),   second half (exp, sum, pack -> P(u))
//                slot 2   MFMA  O     += V(u)^T P(u)     |  VALU  softmax(u+1), first half  (read out S(u+1), row max, exps)
//
// K ring: 2 stages (K(u+1) read, K(u+2) written), V ring: 2 stages (V(u) read, V(u+1) written).
#include <stdlib.h>

#include <type_traits>

#include "fk_common.h"

namespace {

constexpr int HD = 128;
constexpr int KVBLK = 64;
constexpr int KP = HD * 2 + 16, VP = HD * 2 + 64;  // padded LDS row pitches (see attention4_fwd.hip)
constexpr int K_TILE = KVBLK * KP, V_TILE = KVBLK * VP;
constexpr int V_RING = 2 * K_TILE;
constexpr int SMEM_BYTES = 2 * K_TILE + 2 * V_TILE;   // + one flag word (allocated by the launcher)
constexpr int NTHREADS = 256, QBLK = 128;             // 4 waves x 32 rows
constexpr int PIECES = 4;                             // 1 KiB pieces of K (and of V) per wave per tile
constexpr float TAU = 40.0f;

struct Attn5Params {
  const bf16_t* q;
  const bf16_t* k;
  const bf16_t* v;
  bf16_t* o;
  int B, H, S;
  int64_t v_ld, v_bs;
  int64_t o_ld, o_bs;
  float scale_log2;
};

typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned g_t;

FK_DEV s16x4_t lds_tr16(const char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p);
}

template <int I>
using ic = std::integral_constant<int, I>;
using T = std::true_type;
using F = std::false_type;

template <int N, class Fn, int I = 0>
FK_DEV void static_for(Fn&& f) {
  if constexpr (I < N) {
    f(ic<I>{});
    static_for<N, Fn, I + 1>(static_cast<Fn&&>(f));
  }
}

#define FK_INL __attribute__((always_inline))

__global__ __launch_bounds__(NTHREADS) __attribute__((amdgpu_waves_per_eu(1, 1))) void attention5_kernel(const Attn5Params p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ql = lane & 31, hh = lane >> 5;

  const int nqb = (p.S + QBLK - 1) / QBLK;
  int t0;
  {
    const int nwg = gridDim.x;
    const int q8 = nwg >> 3, r8 = nwg & 7;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    t0 = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
  }
  const int qb = t0 % nqb;
  const int bh = t0 / nqb;
  const int b = bh / p.H, h = bh - b * p.H;
  const bf16_t* Kg = p.k + (int64_t)bh * p.S * HD;
  const bf16_t* Vg = p.v + (int64_t)b * p.v_bs + h * HD;

  const int q_row = qb * QBLK + wave * 32 + ql;
  bf16x8_t qf[8];
  {
    const bf16_t* qp = p.q + ((int64_t)bh * p.S + min(q_row, p.S - 1)) * HD + 8 * hh;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) qf[kk] = *(const bf16x8_t*)(qp + 16 * kk);
  }

  // ---- staging: piece i of a tile = 4 key rows x 256 B; lane -> (row = lane/16, 16-byte chunk = lane%16) -----
  const int prow = lane >> 4, pchunk = lane & 15;
  const __amdgpu_buffer_rsrc_t rs_k = __builtin_amdgcn_make_buffer_rsrc((void*)Kg, 0, p.S * HD * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_v =
      __builtin_amdgcn_make_buffer_rsrc((void*)Vg, 0, (int)(((int64_t)(p.S - 1) * p.v_ld + HD) * 2), 0x00020000);
  int k_voff[PIECES], v_voff[PIECES], k_dst[PIECES], v_dst[PIECES];
#pragma unroll
  for (int i = 0; i < PIECES; ++i) {
    const int r = (wave * PIECES + i) * 4 + prow;
    k_voff[i] = r * (HD * 2) + pchunk * 16;
    v_voff[i] = (int)(r * p.v_ld * 2) + pchunk * 16;
    k_dst[i] = r * KP + pchunk * 16;
    v_dst[i] = V_RING + r * VP + pchunk * 16;
  }
  const int k_tile_stride = KVBLK * HD * 2;
  const int v_tile_stride = (int)(KVBLK * p.v_ld * 2);
  g_t gk[PIECES], gv[PIECES];
  auto load_k = [&](int i, int kt) FK_INL { gk[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_k, k_voff[i], kt * k_tile_stride, 0); };
  auto load_v = [&](int i, int kt) FK_INL { gv[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_v, v_voff[i], kt * v_tile_stride, 0); };

  const int k_rd = ql * KP + hh * 16;
  const int tj = (lane & 15) >> 2, tq = lane & 3, tdh = (lane >> 4) & 1;
  const int v_rd = V_RING + (4 * hh + tj) * VP + tdh * 32 + tq * 8;

  f32x16_t o[4];
  f32x16_t sacc[2];     // S^T accumulators of the tile whose QK slot ran last
  float sv[2][32];      // [parity of the tile]: scores -> probabilities -> P fragments packed in place
  float mxp[4], m_ref, nm, l_run, psum;
  bool overflow = false;
  const int nkt = (p.S + KVBLK - 1) / KVBLK;
  const bool ragged = p.S % KVBLK != 0;

  bf16x8_t kf_cur, kf_nx, vf_cur, vf_nx;
  auto k_frag = [&](int i, int kad) FK_INL {   // K operand of QK MFMA i: kk = i / 2, kb = i % 2
    return *(const bf16x8_t*)(smem + kad + (i & 1) * 32 * KP + (i >> 1) * 32);
  };
  auto v_frag = [&](int i, int vad) FK_INL {   // V^T operand of PV MFMA i: st = i / 4, df = i % 4
    const char* vp = smem + vad + (i >> 2) * 16 * VP + (i & 3) * 64;
    const s16x4_t lo = lds_tr16(vp);
    const s16x4_t hi = lds_tr16(vp + 8 * VP);
    bf16x8_t vf;
    vf[0] = lo[0]; vf[1] = lo[1]; vf[2] = lo[2]; vf[3] = lo[3];
    vf[4] = hi[0]; vf[5] = hi[1]; vf[6] = hi[2]; vf[7] = hi[3];
    return vf;
  };
  auto qk_mfma = [&](int i) FK_INL {
    const int kk = i >> 1, kb = i & 1;
    sacc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf_cur, qf[kk], kk == 0 ? f32x16_t{} : sacc[kb], 0, 0, 0);
  };
  auto pv_mfma = [&](int P, int i) FK_INL {   // P fragment of key step st = dwords [8 st, 8 st + 4) of sv[P]
    const int st = i >> 2, df = i & 3, e0 = 8 * st;
    const f32x4_t pw = {sv[P][e0], sv[P][e0 + 1], sv[P][e0 + 2], sv[P][e0 + 3]};
    o[df] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf_cur, __builtin_bit_cast(bf16x8_t, pw), o[df], 0, 0, 0);
  };
  // softmax of the tile with parity P, key tile kt, in 32 chunks (0..15 first half, 16..31 second half)
  auto sm_chunk = [&](int P, int c, auto mask_tag, int kt) FK_INL {
    constexpr bool MASK = decltype(mask_tag)::value;
    if (c < 4) {                       // read the scores out of the accumulators, partial row max
      const int kb = c >> 1, r0 = (c & 1) * 8;
      float m = -3.0e38f;
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        float v = sacc[kb][r0 + r];
        if constexpr (MASK) {
          const int key = kt * KVBLK + 4 * hh + 32 * kb + ((r0 + r) & 3) + 8 * ((r0 + r) >> 2);
          if (key >= p.S) v = -1.0e30f;
        }
        sv[P][16 * kb + r0 + r] = v;
        m = fmaxf(m, v);
      }
      mxp[c] = m;
    } else if (c == 4) {
      float m = fmaxf(fmaxf(mxp[0], mxp[1]), fmaxf(mxp[2], mxp[3]));
      m = fmaxf(m, __shfl_xor(m, 32)) * p.scale_log2;
      overflow = overflow || (__builtin_amdgcn_ballot_w64(m > m_ref + TAU) != 0);
      psum = 0.f;
    } else if (c < 21) {               // c = 5..20: two probabilities each
      const int e = (c - 5) * 2;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const float pv = __builtin_amdgcn_exp2f(fmaf(sv[P][e + u], p.scale_log2, nm));
        sv[P][e + u] = pv;
        psum += pv;
      }
    } else if (c < 29) {               // c = 21..28: pack in place: dwords e0 + j0 + {0,1} <- pairs of e0 + 2 j0 + {0..3}
      const int hfrag = c - 21;
      const int e0 = 8 * (hfrag >> 1), j0 = 2 * (hfrag & 1);
      const float a0 = sv[P][e0 + 2 * j0], a1 = sv[P][e0 + 2 * j0 + 1];
      const float b0 = sv[P][e0 + 2 * j0 + 2], b1 = sv[P][e0 + 2 * j0 + 3];
      sv[P][e0 + j0] = __builtin_bit_cast(float, pack_bf2(a0, a1));
      sv[P][e0 + j0 + 1] = __builtin_bit_cast(float, pack_bf2(b0, b1));
    } else if (c == 29) {
      l_run += psum;
    }
  };
  // staging step j (0..7) in iteration u: K registers hold tile u+2 -> LDS, then request u+3; V: tile u+1, then u+2
  auto stage_step = [&](int j, int u) FK_INL {
    if (j < PIECES) {
      *(g_t*)(smem + ((u + 2) & 1) * K_TILE + k_dst[j]) = gk[j];
      load_k(j, u + 3);
    } else {
      *(g_t*)(smem + ((u + 1) & 1) * V_TILE + v_dst[j - PIECES]) = gv[j - PIECES];
      load_v(j - PIECES, u + 2);
    }
  };
  // pipeline (re)fill: K(0), K(1), V(0) -> LDS; registers: K(2), V(1)
  auto fill = [&]() FK_INL {
#pragma unroll
    for (int i = 0; i < PIECES; ++i) { load_k(i, 0); load_v(i, 0); }
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
      *(g_t*)(smem + k_dst[i]) = gk[i];
      *(g_t*)(smem + v_dst[i]) = gv[i];
    }
#pragma unroll
    for (int i = 0; i < PIECES; ++i) load_k(i, 1);
#pragma unroll
    for (int i = 0; i < PIECES; ++i) *(g_t*)(smem + K_TILE + k_dst[i]) = gk[i];
#pragma unroll
    for (int i = 0; i < PIECES; ++i) { load_k(i, 2); load_v(i, 1); }
    __syncthreads();
  };
  // S(kt) = K(kt) Q^T, unpipelined (prologue and the restart pre-pass)
  auto qk_plain = [&](int kt) FK_INL {
    const int kad = k_rd + (kt & 1) * K_TILE;
    static_for<16>([&](auto I) FK_INL {
      constexpr int i = decltype(I)::value;
      kf_cur = k_frag(i, kad);
      qk_mfma(i);
    });
  };
  auto rowmax_of_sacc = [&](int kt) FK_INL {
    float m = -3.0e38f;
    const bool ragged_tile = ragged && kt == nkt - 1;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kt * KVBLK + 4 * hh + 32 * kb + (r & 3) + 8 * (r >> 2);
        m = fmaxf(m, (ragged_tile && key >= p.S) ? -1.0e30f : sacc[kb][r]);
      }
    return fmaxf(m, __shfl_xor(m, 32)) * p.scale_log2;
  };

  // iteration u with compile-time parity PAR = u % 2; HAS_NEXT: tile u+1 exists; mask tags for tiles u and u+1
  auto iterate = [&](auto par_tag, auto has_next_tag, auto mask_u, auto mask_n, int u) FK_INL {
    constexpr int PAR = decltype(par_tag)::value;
    constexpr bool HAS_NEXT = decltype(has_next_tag)::value;
    const int kad = k_rd + ((u + 1) & 1) * K_TILE;
    const int vad = v_rd + (u & 1) * V_TILE;
    if constexpr (HAS_NEXT) kf_nx = k_frag(0, kad);
    // slot 1: S(u+1) | softmax(u) second half
    static_for<16>([&](auto I) FK_INL {
      constexpr int i = decltype(I)::value;
      if constexpr (HAS_NEXT) {
        kf_cur = kf_nx;
        if (i < 15) kf_nx = k_frag(i + 1, kad);
      }
      if (i == 15) vf_nx = v_frag(0, vad);
      if constexpr (HAS_NEXT) qk_mfma(i);
      sm_chunk(PAR, 16 + i, mask_u, u);
      if (i % 4 == 1) stage_step(i / 4, u);
      __builtin_amdgcn_sched_barrier(0);
    });
    // slot 2: O += V(u)^T P(u) | softmax(u+1) first half
    static_for<16>([&](auto I) FK_INL {
      constexpr int i = decltype(I)::value;
      vf_cur = vf_nx;
      if (i < 15) vf_nx = v_frag(i + 1, vad);
      pv_mfma(PAR, i);
      if constexpr (HAS_NEXT) sm_chunk(1 - PAR, i, mask_n, u + 1);
      if (i % 4 == 1) stage_step(PIECES + i / 4, u);
      __builtin_amdgcn_sched_barrier(0);
    });
    // K(u+2), V(u+1) complete in LDS; every wave done reading K(u+1), V(u)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  };

  int* const wg_flag = (int*)(smem + SMEM_BYTES);
  for (int attempt = 0; attempt < 2; ++attempt) {
    fill();
    if (attempt == 0) {
      qk_plain(0);
      m_ref = rowmax_of_sacc(0);                   // reference = row maximum over the first tile
    } else {
      // exact row maxima: K-only pre-pass (V rides along in the staging steps); tiles u+2 / u+1 are staged per step
      m_ref = -3.0e38f;
      for (int u = 0; u < nkt; ++u) {
        qk_plain(u);
        m_ref = fmaxf(m_ref, rowmax_of_sacc(u));
        // keep the rings rolling exactly like the main loop does in the iteration that READS K(u): iteration u-1
        if (u >= 1) {
#pragma unroll
          for (int j = 0; j < 2 * PIECES; ++j) stage_step(j, u - 1);
        }
        __syncthreads();
      }
      fill();
      qk_plain(0);
    }
    nm = -m_ref;
    l_run = 0.f;
    overflow = false;
#pragma unroll
    for (int df = 0; df < 4; ++df)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[df][r] = 0.f;
    // every wave is done reading K(0) (iteration 0 stages K(2) into its place)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // softmax(0) first half (S(0) is in sacc)
    if (ragged && nkt == 1) static_for<16>([&](auto I) FK_INL { sm_chunk(0, decltype(I)::value, T{}, 0); });
    else static_for<16>([&](auto I) FK_INL { sm_chunk(0, decltype(I)::value, F{}, 0); });

    // main loop: iterations u = 0 .. nkt-1; the last one has no tile u+1; masks apply to the ragged last tile
    for (int u = 0; u + 2 < nkt; ++u) {
      if (u & 1) iterate(ic<1>{}, T{}, F{}, F{}, u);
      else iterate(ic<0>{}, T{}, F{}, F{}, u);
    }
    if (nkt >= 2) {
      const int u = nkt - 2;     // tile u+1 is the last tile
      if (ragged) { if (u & 1) iterate(ic<1>{}, T{}, F{}, T{}, u); else iterate(ic<0>{}, T{}, F{}, T{}, u); }
      else { if (u & 1) iterate(ic<1>{}, T{}, F{}, F{}, u); else iterate(ic<0>{}, T{}, F{}, F{}, u); }
    }
    {
      const int u = nkt - 1;
      if (ragged) { if (u & 1) iterate(ic<1>{}, F{}, T{}, F{}, u); else iterate(ic<0>{}, F{}, T{}, F{}, u); }
      else { if (u & 1) iterate(ic<1>{}, F{}, F{}, F{}, u); else iterate(ic<0>{}, F{}, F{}, F{}, u); }
    }
    if (attempt == 0) {
      if (tid == 0) *wg_flag = 0;
      __syncthreads();
      if (overflow && lane == 0) atomicOr(wg_flag, 1);
      __syncthreads();
      if (*wg_flag == 0) break;
      __syncthreads();
    }
  }

  // ---- finalize: O = O^T / l ; lane (q = ql) holds d = 32 df + 8 (r>>2) + 4 hh + (r&3) --------------------
  const float l_tot = l_run + __shfl_xor(l_run, 32);
  const float inv = 1.0f / l_tot;
  if (q_row < p.S) {
    bf16_t* op = p.o + (int64_t)b * p.o_bs + (int64_t)q_row * p.o_ld + h * HD + 4 * hh;
#pragma unroll
    for (int df = 0; df < 4; ++df)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        u32x2_t pk;
        pk[0] = pack_bf2(o[df][4 * g + 0] * inv, o[df][4 * g + 1] * inv);
        pk[1] = pack_bf2(o[df][4 * g + 2] * inv, o[df][4 * g + 3] * inv);
        *(u32x2_t*)(op + 32 * df + 8 * g) = pk;
      }
  }
}

}  // namespace

int fk_attention5_launch(const void* q, const void* k, const void* v, void* o, int B, int H, int S, int64_t v_ld,
                         int64_t v_bs, int64_t o_ld, int64_t o_bs, float scale, hipStream_t stream) {
  Attn5Params p;
  p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.v = (const bf16_t*)v; p.o = (bf16_t*)o;
  p.B = B; p.H = H; p.S = S; p.v_ld = v_ld; p.v_bs = v_bs; p.o_ld = o_ld; p.o_bs = o_bs;
  p.scale_log2 = scale * 1.4426950408889634f;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)attention5_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES + 16);
    attr_done = true;
  }
  const int nqb = (S + QBLK - 1) / QBLK;
  hipLaunchKernelGGL(attention5_kernel, dim3(nqb * H * B), dim3(NTHREADS), SMEM_BYTES + 16, stream, p);
  FK_CHECK_LAUNCH("fk_attention_fwd_bf16 (4 waves x 32 rows)");
  return FK_OK;
}
