// Flash-style joint attention forward for the FLUX MMDiT blocks (K4 in SURVEY.md section 2.2):
//   O = softmax(Q K^T / sqrt(128)) V,   non-causal, no mask, head_dim = 128, bf16 in / bf16 out,
//   fp32 scores, fp32 online softmax, fp32 accumulation (= F.scaled_dot_product_attention as the
//   reference reaches it through diffusers' FluxAttnProcessor2_0).
//
// One workgroup = NW waves x 32 query rows of one (batch, head); the keys are walked in tiles of 64.
// Both products are formed transposed so that the softmax axis is lane-local ("swapped QK^T"):
//   S^T[key][q] = mfma_32x32x16(A = K rows, B = Q rows)      -> lane (q = lane&31) holds 16 keys / block
//   O^T[d][q]  += mfma_32x32x16(A = V^T,    B = P^T)         -> lane (q = lane&31) holds 64 d's
// so row max / row sum need ONE cross-lane exchange (lane ^ 32) and the O rescale is a per-lane scalar.
// P is consumed in exactly the register order the first MFMA produced it: MFMA k-slot (h = lane>>5, j)
// is bound to key 16*step + 4h + (j&3) + 8*(j>>2) for BOTH operands, so P needs no lane permutation.
//
// K and V tiles ([64 keys][128 d], row-major, V read in place from the fused QKV projection output)
// arrive by LDS-DMA (`global_load_lds_dwordx4`, no staging registers) into a ring of STAGES stages with
// counted `vmcnt` waits and one raw `s_barrier` per tile.  The V^T operand is produced by the gfx950 LDS
// transpose read `ds_read_b64_tr_b16` (semantics measured in profiles/r01_probe_lds_semantics.txt:
// within 16 lanes, lane i receives element (i&3) of the 8-byte pieces addressed by lanes 4j + (i>>2)),
// so V is never transposed in memory.  LDS swizzles (applied on the DMA source address and the read):
//   K: 16-byte chunk ^= key & 15        (ds_read_b128 over 256-byte rows: conflict-free)
//   V: 64-byte block ^= key & 3         (the four key rows of a transpose read hit four distinct blocks)
#include <stdlib.h>

#include <type_traits>

#include "fk_common.h"

namespace {

constexpr int HD = 128;
constexpr int KVBLK = 64;                       // keys per tile
constexpr int K_TILE_BYTES = KVBLK * HD * 2;    // 16 KiB
constexpr int STAGE_BYTES = 2 * K_TILE_BYTES;   // K + V

struct AttnParams {
  const bf16_t* q;
  const bf16_t* k;
  const bf16_t* v;
  bf16_t* o;
  int B, H, S;
  int64_t v_ld, v_bs;  // V row (token) stride / batch stride in elements; head h at column h*128
  int64_t o_ld, o_bs;
  float scale_log2;    // scale * log2(e)
  float* lse;          // optional [B, H, S]: log2-domain log-sum-exp of every row (saved for the backward pass)
  // The (b, h, 256-row block) list: blocks [0, n_full) are one workgroup each (8 waves x 32 rows); every later block is
  // cut into `light_subs` (4 / 2) "light" workgroups in which all 8 waves keep loading K / V tiles but only the first
  // 8 / light_subs waves own query rows (see attention_entry).  light_subs = 0: no light workgroups.
  int n_full, light_subs;
};

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;
typedef short s16x4_t __attribute__((ext_vector_type(4)));

// buffer form of the LDS-DMA load (descriptor in SGPRs, one 32-bit offset VGPR, SGPR tile offset).  The builtin
// exists for the device target only; seen by the host pass it silently suppresses the kernel's host stub.
FK_DEV void buffer_lds16(__amdgpu_buffer_rsrc_t rsrc, char* lds_dst, int voffset, int soffset) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)lds_dst, 16, voffset, soffset, 0, 0);
#else
  (void)rsrc; (void)lds_dst; (void)voffset; (void)soffset;
#endif
}
FK_DEV s16x4_t lds_tr16(const char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p);
}
template <int N>
FK_DEV void wait_vmcnt() {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  else static_assert(N == 0, "add the vmcnt literal");
}

#ifndef FK_ATTN_PRIO
#define FK_ATTN_PRIO 1   // static s_setprio(1) for the younger half of the workgroup (waves NW/2 .. NW-1)
#endif

// F32OUT (parity / debug build of the same kernel, fk_attention_fwd_f32_debug): the output is written as fp32 and
// the probabilities enter the PV product as TWO bf16 terms (p = hi + lo, 16 mantissa bits instead of 8), so that
// the result can be held against an fp32 reference at rtol 1e-3 / atol 1e-4 -- with one bf16 term the rounding of
// P alone (2^-9 per term) sits above that tolerance whatever the kernel does.
template <int NW, int STAGES, bool F32OUT, bool ILV>
__global__ __launch_bounds__(NW * 64, NW >= 8 ? 2 : 1) void attention_fwd_kernel(const AttnParams p) {
  constexpr int QBLK = NW * 32;
  constexpr int LOADS = 32 / NW;      // DMA instructions per wave per tile (16 K pieces + 16 V pieces / NW)
  constexpr int KL = LOADS / 2;       // K pieces per wave (same number of V pieces)
  constexpr int PF = STAGES - 1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ql = lane & 31;   // query row inside the wave / operand row
  const int hh = lane >> 5;   // half

  // XCD-aware block order: workgroups of one (b, h) -- which share K / V -- stay on one XCD's L2
  constexpr int QSPAN = 256;                // query rows per entry of the block list (QBLK * p.subs)
  const int nqb = (p.S + QSPAN - 1) / QSPAN;
  int t;
  {
    const int nwg = gridDim.x;
    const int q8 = nwg >> 3, r8 = nwg & 7;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    t = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
  }
  int tb = t, sub = 0, nact = NW;           // nact: waves of this workgroup that own query rows
  if (t >= p.n_full) {
    const int l = t - p.n_full;
    tb = p.n_full + l / p.light_subs;
    sub = l - (l / p.light_subs) * p.light_subs;
    nact = NW / p.light_subs;
  }
  const int qb = tb % nqb;
  const int bh = tb / nqb;
  const int b = bh / p.H, h = bh - b * p.H;
  const int q_row0 = qb * QSPAN + sub * nact * 32;
  if (q_row0 >= p.S) return;                // ragged last block: nothing for this light workgroup
  const bool active = wave < nact;          // wave-uniform; the other waves only feed the K / V ring and keep the barriers

  const bf16_t* Kg = p.k + (int64_t)bh * p.S * HD;
  const bf16_t* Vg = p.v + (int64_t)b * p.v_bs + h * HD;

  // ---- Q operand fragments (B operand of S^T = K Q^T): lane holds Q[q][16kk + 8hh .. +8] ----------
  const int q_row = q_row0 + wave * 32 + ql;
  bf16x8_t qf[8];
  {
    const bf16_t* qp = p.q + ((int64_t)bh * p.S + min(q_row, p.S - 1)) * HD + 8 * hh;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) qf[kk] = *(const bf16x8_t*)(qp + 16 * kk);
  }

  // ---- LDS-DMA pieces: one instruction = 4 key rows x 256 B; lane -> (row = lane/16, 16-byte slot = lane%16)
  const int prow = lane >> 4, pslot = lane & 15;
  // Buffer form of the LDS-DMA load: descriptor in SGPRs, ONE 32-bit offset VGPR per piece (constant over the
  // tiles), the tile offset in an SGPR.  Rows >= S lie beyond num_records and are fetched as zeros (they are masked).
  const __amdgpu_buffer_rsrc_t rs_k = __builtin_amdgcn_make_buffer_rsrc((void*)Kg, 0, p.S * HD * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_v =
      __builtin_amdgcn_make_buffer_rsrc((void*)Vg, 0, (int)(((int64_t)(p.S - 1) * p.v_ld + HD) * 2), 0x00020000);
  int k_voff[KL], v_voff[KL];
#pragma unroll
  for (int i = 0; i < KL; ++i) {
    const int r = (wave * KL + i) * 4 + prow;                       // key row inside the tile
    k_voff[i] = (r * HD + ((pslot ^ (r & 15)) << 3)) * 2;
    const int vcol = ((((pslot >> 2) ^ (r & 3)) << 5) + ((pslot & 3) << 3));
    v_voff[i] = (int)((r * p.v_ld + vcol) * 2);
  }
  const int k_tile_bytes = KVBLK * HD * 2, v_tile_bytes = (int)(KVBLK * p.v_ld * 2);
  auto issue_tile = [&](int kt, int stage) {
    char* sb = smem + stage * STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < KL; ++i) {
      buffer_lds16(rs_k, sb + (wave * KL + i) * 1024, k_voff[i], kt * k_tile_bytes);
      buffer_lds16(rs_v, sb + K_TILE_BYTES + (wave * KL + i) * 1024, v_voff[i], kt * v_tile_bytes);
    }
  };

  // ---- operand read addresses ------------------------------------------------------------------------
  // K operand (A of S^T): row key = 32 kb + ql, chunk 2kk + hh, swizzled by key & 15 (= ql & 15)
  const int k_rd = ql * 256;           // + kb*8192 + (((2kk + hh) ^ (ql & 15)) << 4)
  const int k_sw = ql & 15;
  // V^T operand via transpose read: lane supplies the 8-byte piece V[key0 + j][32 df + 16 dhalf + 4 q4 ..]
  const int tj = (lane & 15) >> 2, tq = lane & 3, tdh = (lane >> 4) & 1;
  const int v_rd = K_TILE_BYTES + (4 * hh + tj) * 256 + tdh * 32 + tq * 8;  // + (16 st + 8 part)*256 + ((df ^ tj) << 6)

  f32x16_t o[4];
  // Exponent reference.  The textbook online softmax rescales O by exp(m_old - m_new) on every tile (64 multiplies
  // per wave and tile, on the critical VALU path: the counters show this kernel's MFMA and VALU phases added, not
  // overlapped).  Here every row keeps a FIXED reference m_ref = its row maximum over the FIRST KV tile, so
  // p = exp2(s - m_ref) may exceed 1 when later tiles hold larger scores -- harmless for the fp32 sums and for the
  // bf16 P fragments (bf16 has fp32's exponent range and the same relative precision at every magnitude) as long as
  // nothing overflows.  If some row ever outgrows its reference by more than the fp32 exponent range (adversarial
  // inputs; the tests build one) its sums turn inf / NaN; that is detected once, after the pass, and the workgroup
  // repeats the pass with the exact row maxima from a K-only pre-pass: the result is then the plain two-pass softmax.
  // The reference sits REF_BIAS above the first block's maximum: later scores may then grow by ~127 + 24 log2 units
  // (~100 nats) before a restart is needed, and the first block's own probabilities (>= 2^-24 at its maximum) are
  // still far from fp32 / bf16 underflow.
  constexpr float REF_BIAS = 24.0f;
  float m_ref = 0.f, l_run = 0.f;
  bool overflow = false;         // wave-uniform

  const int nkt = (p.S + KVBLK - 1) / KVBLK;
  // STAGES == 3: one barrier per tile, tile kt + 2 requested when tile kt is published.
  // STAGES == 4 ("pairs"): ONE barrier per TWO tiles -- at every even tile the wave waits for the pair (kt, kt + 1),
  // passes the barrier and requests the pair (kt + 2, kt + 3) into the two stages the previous pair has just left.
  // Measured (profiles/r03_attention_variants.txt): 1-3 % SLOWER at every shape, isolated and inside the edits -- the
  // per-tile barrier is not the cost the parked-wave counter suggests; it keeps the eight waves' K / V reads together.
  // Kept as a switch (fk_attention_set_ring), not the default.
  constexpr bool PAIRS = STAGES == 4;
  int st_cur = 0, st_pf = PF;
  auto fill = [&]() {
#pragma unroll
    for (int s = 0; s < (PAIRS ? 2 : PF); ++s)
      if (s < nkt) issue_tile(s, s);
    st_cur = 0;
    st_pf = PAIRS ? 2 : PF;
  };
  // ring bookkeeping shared by the main loop and the pre-pass: wait for tile kt, publish it, request the tile(s) ahead
  auto acquire_tile = [&](int kt) {
    if constexpr (PAIRS) {
      if ((kt & 1) == 0) {
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (kt + 2 < nkt) issue_tile(kt + 2, st_pf);
        if (kt + 3 < nkt) issue_tile(kt + 3, st_pf == STAGES - 1 ? 0 : st_pf + 1);
      }
    } else {
      if (kt + PF - 1 < nkt) wait_vmcnt<(PF - 1) * LOADS>();
      else wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      if (kt + PF < nkt) issue_tile(kt + PF, st_pf);
    }
    return smem + st_cur * STAGE_BYTES;
  };
  auto release_tile = [&]() {
    st_cur = (st_cur == STAGES - 1) ? 0 : st_cur + 1;
    st_pf = (st_pf == STAGES - 1) ? 0 : st_pf + 1;
  };
  // Operand fragments.  The reads are issued PF_DEPTH MFMAs ahead of their use and the order is pinned
  // (sched_group_barrier: one DS read, then one MFMA): left alone, hipcc issues every ds_read right in front of
  // its MFMA and the wave waits for an LDS round trip 32 times per tile.
  auto k_frag = [&](const char* sb, int kb, int kk) {
    return *(const bf16x8_t*)(sb + k_rd + kb * 8192 + (((2 * kk + hh) ^ k_sw) << 4));
  };
  auto v_frag = [&](const char* sb, int st, int df) {   // keys 16 st + {0, 8} + 4 hh + 0..3, d block df
    const char* vp = sb + v_rd + st * 4096 + ((df ^ tj) << 6);
    const s16x4_t lo = lds_tr16(vp);
    const s16x4_t hi = lds_tr16(vp + 2048);
    bf16x8_t vf;
    vf[0] = lo[0]; vf[1] = lo[1]; vf[2] = lo[2]; vf[3] = lo[3];
    vf[4] = hi[0]; vf[5] = hi[1]; vf[6] = hi[2]; vf[7] = hi[3];
    return vf;
  };
  // S^T block kb (32 keys x 32 queries) of the tile at sb, masked beyond S in the ragged last tile
  auto scores = [&](const char* sb, int kb, int kt, auto mask_tag) {
    constexpr bool MASK = decltype(mask_tag)::value;
    f32x16_t s;
    bf16x8_t kf[3];
    kf[0] = k_frag(sb, kb, 0);
    kf[1] = k_frag(sb, kb, 1);
    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);     // the two leading reads
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      if (kk + 2 < 8) kf[(kk + 2) % 3] = k_frag(sb, kb, kk + 2);
      // first k-step takes a literal zero C operand (inline constant): no per-tile re-zeroing of the accumulator
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kk % 3], qf[kk], kk == 0 ? f32x16_t{} : s, 0, 0, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // the DS read of this slot first ...
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // ... then its MFMA
    }
    if constexpr (MASK) {
      const int kbase = kt * KVBLK + 32 * kb + 4 * hh;
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (kbase + (r & 3) + 8 * (r >> 2) >= p.S) s[r] = -1.0e30f;
    }
    return s;
  };
  auto block_max = [&](const f32x16_t& s) {   // row maximum over one 32-key block, in log2 units
    float mx = s[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
    return fmaxf(mx, __shfl_xor(mx, 32)) * p.scale_log2;
  };

  // One KV tile = two 32-key halves, each: 8 QK MFMAs -> exp2 (fixed reference: no dependence on the tile's
  // maximum) -> pack -> 8 PV MFMAs.  Only one 32 x 32 score block is live at a time.
  // MASK = the tile holds keys >= S (only ever the last tile), FIRST = tile 0 of the first attempt (sets the
  // exponent reference): compiled as separate copies so the steady-state loop carries no selects.
  // ILV: the same arithmetic in another order.  Per wave: QK(0); then QK(1) with the exponentials of block 0 issued in
  // the shadow of its MFMAs; then PV(0) with the exponentials of block 1 in the shadow; then PV(1).  The interleave is
  // pinned (sched_group_barrier: one MFMA, then its share of VALU / transcendental work); hipcc by itself emits the
  // phases back to back.  Bit-identical results; which order is faster depends on the clock the kernel runs at
  // (profiles/r02_attention_variants.txt), so the launcher chooses.
  // MASK = the tile holds keys >= S (only ever the last tile), FIRST = tile 0 of the first attempt (sets the
  // exponent reference): compiled as separate copies so the steady-state loop carries no selects.
  auto do_tile = [&](int kt, auto mask_tag, auto first_tag) {
    if constexpr (!ILV) {
      constexpr bool FIRST = decltype(first_tag)::value;
      const char* sb = acquire_tile(kt);
      if (!active) { release_tile(); return; }
      float psum = 0.f;
  #pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        f32x16_t s = scores(sb, kb, kt, mask_tag);
        if constexpr (FIRST) {
          if (kb == 0) m_ref = block_max(s) + REF_BIAS;   // reference = row maximum over the first 32 keys + bias
        }
        const float nm = -m_ref;
        // ---- softmax numerators (log2 domain; raw v_exp_f32, denormal results may flush) ----------------------
  #pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float pv = __builtin_amdgcn_exp2f(fmaf(s[r], p.scale_log2, nm));
          s[r] = pv;
          psum += pv;
        }
        // ---- O^T += V^T P^T for the two 16-key steps of this half ------------------------------------------------
        bf16x8_t vf[3];
        vf[0] = v_frag(sb, 2 * kb, 0);
        vf[1] = v_frag(sb, 2 * kb, 1);
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);   // the leading reads (two transpose reads per fragment)
        bf16x8_t pf, pf_lo;
  #pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int st = 2 * kb + (i >> 2), df = i & 3;
          if (df == 0) {
            const int r0 = 8 * (i >> 2);
            u32x4_t pw;
            pw[0] = pack_bf2(s[r0 + 0], s[r0 + 1]);
            pw[1] = pack_bf2(s[r0 + 2], s[r0 + 3]);
            pw[2] = pack_bf2(s[r0 + 4], s[r0 + 5]);
            pw[3] = pack_bf2(s[r0 + 6], s[r0 + 7]);
            pf = __builtin_bit_cast(bf16x8_t, pw);
            if constexpr (F32OUT) {
              u32x4_t pl;
  #pragma unroll
              for (int e = 0; e < 4; ++e)
                pl[e] = pack_bf2(s[r0 + 2 * e] - bf_lo(pw[e]), s[r0 + 2 * e + 1] - bf_hi(pw[e]));
              pf_lo = __builtin_bit_cast(bf16x8_t, pl);
            }
          }
          if (i + 2 < 8) vf[(i + 2) % 3] = v_frag(sb, 2 * kb + ((i + 2) >> 2), (i + 2) & 3);
          o[df] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[i % 3], pf, o[df], 0, 0, 0);
          if constexpr (F32OUT) o[df] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[i % 3], pf_lo, o[df], 0, 0, 0);
          else {
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   // the two transpose reads of this slot first ...
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // ... then its MFMA
          }
        }
      }
      l_run += psum;
      release_tile();
      } else {
      constexpr bool FIRST = decltype(first_tag)::value;
      constexpr bool MASK = decltype(mask_tag)::value;
      const char* sb = acquire_tile(kt);
      if (!active) { release_tile(); return; }
      float psum = 0.f;
      f32x16_t s0 = scores(sb, 0, kt, mask_tag);
      if constexpr (FIRST) m_ref = block_max(s0) + REF_BIAS;   // reference = row maximum over the first 32 keys + bias
      const float nm = -m_ref;
      // softmax numerators of two scores (log2 domain; raw v_exp_f32, denormal results may flush)
      auto expo2 = [&](f32x16_t& s, int r) {
  #pragma unroll
        for (int e = 0; e < 2; ++e) {
          const float pv = __builtin_amdgcn_exp2f(fmaf(s[r + e], p.scale_log2, nm));
          s[r + e] = pv;
          psum += pv;
        }
      };
      auto pack8 = [&](const f32x16_t& s, int r0, bf16x8_t& pf, bf16x8_t& pf_lo) {
        u32x4_t pw;
        pw[0] = pack_bf2(s[r0 + 0], s[r0 + 1]);
        pw[1] = pack_bf2(s[r0 + 2], s[r0 + 3]);
        pw[2] = pack_bf2(s[r0 + 4], s[r0 + 5]);
        pw[3] = pack_bf2(s[r0 + 6], s[r0 + 7]);
        pf = __builtin_bit_cast(bf16x8_t, pw);
        if constexpr (F32OUT) {
          u32x4_t pl;
  #pragma unroll
          for (int e = 0; e < 4; ++e)
            pl[e] = pack_bf2(s[r0 + 2 * e] - bf_lo(pw[e]), s[r0 + 2 * e + 1] - bf_hi(pw[e]));
          pf_lo = __builtin_bit_cast(bf16x8_t, pl);
        }
      };
      // ---- S^T of block 1 under which block 0's exponentials run ---------------------------------------------------
      f32x16_t s1;
      {
        bf16x8_t kf[3];
        kf[0] = k_frag(sb, 1, 0);
        kf[1] = k_frag(sb, 1, 1);
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
  #pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
          if (kk + 2 < 8) kf[(kk + 2) % 3] = k_frag(sb, 1, kk + 2);
          s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kk % 3], qf[kk], kk == 0 ? f32x16_t{} : s1, 0, 0, 0);
          expo2(s0, 2 * kk);
          if constexpr (!F32OUT) {
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // the DS read of this slot
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // its MFMA
            __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);   // two scale-and-shift FMAs
            __builtin_amdgcn_sched_group_barrier(0x400, 2, 0);   // two exponentials
            __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);   // two row-sum adds
          }
        }
        if constexpr (MASK) {
          const int kbase = kt * KVBLK + 32 + 4 * hh;
  #pragma unroll
          for (int r = 0; r < 16; ++r)
            if (kbase + (r & 3) + 8 * (r >> 2) >= p.S) s1[r] = -1.0e30f;
        }
      }
      // ---- O^T += V^T P^T: block 0 (block 1's exponentials in the shadow), then block 1 ------------------------------
  #pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        const f32x16_t& sp = kb == 0 ? s0 : s1;
        bf16x8_t vf[3];
        vf[0] = v_frag(sb, 2 * kb, 0);
        vf[1] = v_frag(sb, 2 * kb, 1);
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);   // the leading reads (two transpose reads per fragment)
        bf16x8_t pf, pf_lo;
  #pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int st = 2 * kb + (i >> 2), df = i & 3;
          if (df == 0) pack8(sp, 8 * (i >> 2), pf, pf_lo);
          if (i + 2 < 8) vf[(i + 2) % 3] = v_frag(sb, 2 * kb + ((i + 2) >> 2), (i + 2) & 3);
          o[df] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[i % 3], pf, o[df], 0, 0, 0);
          if constexpr (F32OUT) o[df] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[i % 3], pf_lo, o[df], 0, 0, 0);
          if (kb == 0) expo2(s1, 2 * i);
          if constexpr (!F32OUT) {
            if (df == 0) __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);   // the four packs this MFMA group consumes
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   // the two transpose reads of this slot
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // its MFMA
            if (kb == 0) {
              __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
              __builtin_amdgcn_sched_group_barrier(0x400, 2, 0);
              __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
            }
          }
        }
      }
      l_run += psum;
      release_tile();
      }
  };

  using TT = std::true_type;
  using FF = std::false_type;
#if FK_ATTN_PRIO
  // Two waves share every SIMD (waves w and w + NW/2); the later-dispatched one loses the VALU arbitration (priority,
  // then age) at the head of every segment.  ONE static priority raise for that half, no per-segment flips.  The
  // condition must be provably wave-uniform: s_setprio ignores EXEC.
  if (wave >= NW / 2) __builtin_amdgcn_s_setprio(1);
#endif
  const bool ragged = p.S % KVBLK != 0;
  int* const wg_flag = (int*)(smem + STAGES * STAGE_BYTES);   // one word past the ring (allocated by the launcher)
  for (int attempt = 0; attempt < 2; ++attempt) {
    if (attempt == 1) {
      // exact row maxima: K-only pre-pass over all tiles (V rides along in the ring)
      fill();
      m_ref = -3.0e38f;
      for (int kt = 0; kt < nkt; ++kt) {
        const char* sb = acquire_tile(kt);
        if (active) {
#pragma unroll
          for (int kb = 0; kb < 2; ++kb) {
            if (ragged && kt == nkt - 1) m_ref = fmaxf(m_ref, block_max(scores(sb, kb, kt, TT{})));
            else m_ref = fmaxf(m_ref, block_max(scores(sb, kb, kt, FF{})));
          }
        }
        release_tile();
      }
      __syncthreads();   // every wave is done with the ring before it is refilled
    }
    fill();
    l_run = 0.f;
    overflow = false;
#pragma unroll
    for (int df = 0; df < 4; ++df)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[df][r] = 0.f;
    if (attempt == 0) {
      if (nkt == 1) {
        if (ragged) do_tile(0, TT{}, TT{});
        else do_tile(0, FF{}, TT{});
      } else {
        do_tile(0, FF{}, TT{});
      }
    } else {
      if (nkt == 1 && ragged) do_tile(0, TT{}, FF{});
      else do_tile(0, FF{}, FF{});
    }
    for (int kt = 1; kt < nkt - 1; ++kt) do_tile(kt, FF{}, FF{});
    if (nkt > 1) {
      if (ragged) do_tile(nkt - 1, TT{}, FF{});
      else do_tile(nkt - 1, FF{}, FF{});
    }
    // the waves share the K/V ring and its barriers: they repeat the pass together or not at all
    if (attempt == 0) {
      // Did some row outgrow the exponent range?  No per-tile maximum is kept for this (that was ~1 VALU instruction
      // per score, a fifth of the softmax work): a score more than ~127 log2 units above the reference makes its
      // numerator +inf, which poisons the row's sum and its O (inf, or NaN from inf * 0) -- visible here, once.
      float mag = fabsf(l_run);
#pragma unroll
      for (int df = 0; df < 4; ++df)
#pragma unroll
        for (int r = 0; r < 16; ++r) mag += fabsf(o[df][r]);
      overflow = __builtin_amdgcn_ballot_w64(!(mag <= 3.0e38f)) != 0;
      __syncthreads();
      if (tid == 0) *wg_flag = 0;
      __syncthreads();
      if (overflow && lane == 0) atomicOr(wg_flag, 1);
      __syncthreads();
      if (*wg_flag == 0) break;
    }
  }

  if (!active) return;
  // ---- finalize: O = O^T / l ; lane (q = ql) holds d = 32 df + 8 g + 4 hh + (0..3), g = r >> 2 -------------
  const float l_tot = l_run + __shfl_xor(l_run, 32);
  const float inv = 1.0f / l_tot;
  if (p.lse && hh == 0 && q_row < p.S) p.lse[(int64_t)bh * p.S + q_row] = m_ref + __builtin_amdgcn_logf(l_tot);
  if constexpr (F32OUT) {
    if (q_row < p.S) {
      float* op = (float*)p.o + (int64_t)b * p.o_bs + (int64_t)q_row * p.o_ld + h * HD + 4 * hh;
#pragma unroll
      for (int df = 0; df < 4; ++df)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *(f32x4_t*)(op + 32 * df + 8 * g) = f32x4_t{o[df][4 * g + 0] * inv, o[df][4 * g + 1] * inv,
                                                       o[df][4 * g + 2] * inv, o[df][4 * g + 3] * inv};
    }
  } else {
    // The two half-waves hold neighbouring 8-byte pieces of one output row.  One v_permlane32_swap per dword pairs
    // them up: afterwards lanes 0..31 own the 16 bytes of columns 8g .. 8g+7 and lanes 32..63 those of 8(g+1) ..
    // 8(g+1)+7 -- 8 dwordx4 stores per lane instead of 16 dwordx2 (the store tail is issue-bound, not bandwidth-bound).
    bf16_t* const orow = p.o + (int64_t)b * p.o_bs + (int64_t)min(q_row, p.S - 1) * p.o_ld + h * HD;
    const bool wide = ((p.o_ld | p.o_bs) & 7) == 0 && ((uintptr_t)p.o & 15) == 0;   // wave-uniform
#pragma unroll
    for (int df = 0; df < 4; ++df)
#pragma unroll
      for (int g = 0; g < 4; g += 2) {
        u32x2_t a, c;
        a[0] = pack_bf2(o[df][4 * g + 0] * inv, o[df][4 * g + 1] * inv);
        a[1] = pack_bf2(o[df][4 * g + 2] * inv, o[df][4 * g + 3] * inv);
        c[0] = pack_bf2(o[df][4 * g + 4] * inv, o[df][4 * g + 5] * inv);
        c[1] = pack_bf2(o[df][4 * g + 6] * inv, o[df][4 * g + 7] * inv);
        if (wide) {
#if defined(__HIP_DEVICE_COMPILE__)
          const auto r0 = __builtin_amdgcn_permlane32_swap(a[0], c[0], false, false);
          const auto r1 = __builtin_amdgcn_permlane32_swap(a[1], c[1], false, false);
          const u32x4_t w = {r0[0], r1[0], r0[1], r1[1]};
          if (q_row < p.S) *(u32x4_t*)(orow + 32 * df + 8 * g + 8 * hh) = w;
#endif
        } else if (q_row < p.S) {
          *(u32x2_t*)(orow + 32 * df + 8 * g + 4 * hh) = a;
          *(u32x2_t*)(orow + 32 * df + 8 * (g + 1) + 4 * hh) = c;
        }
      }
  }
}

// One-wave-per-SIMD schedules (round 1, git history: two query blocks per wave half a tile apart; one block with the
// softmax of tile u+1 under the MFMAs of tile u; both with a fixed exponent reference, padded LDS rows and register
// staging) are correct but slower (630-830 and 540-660 TF/s against 790-1060 here): with a single wave on a SIMD every
// instruction of the stream -- S read-out, LDS reads, staging, softmax -- takes an issue slot of its own, 7-8 per MFMA
// against the 5 that fit an MFMA shadow; with two waves per SIMD the hardware overlaps one wave's VALU with the
// other's MFMAs at no issue cost.
//
// Round 2: the two-group ("ping-pong") structure that took the GEMM to 89 % MFMA-busy -- groups of 4 waves alternating
// between a 16-MFMA phase from registers (S^T of block b+1 and O^T += V^T P^T of block b) and a softmax + LDS-read
// phase, two barriers per phase, K / V rings of 4 x 16 KiB -- is correct (all attention tests) and 7-14 % SLOWER than
// this kernel at every shape (profiles/r02_attention_variants.txt): the softmax phase is VALU work that the partner's
// MFMA phase starves of issue slots (with s_setprio on the MFMA phase another -3 %), and every phase boundary pays an
// LDS round trip; free-running waves overlap better than barrier-enforced alternation here.  Removed (git history).
//
// Round-1 dead ends (git history): (1) issuing QK^T of tile t+1 before the softmax of tile t inside one wave WITHOUT
// pinning the order (spills, the compiler does not interleave: 760 TF vs 844) -- the ILV order above is its working
// form, within one tile and pinned by sched_group_barrier; (2) rotating waves 4..7 by one phase so that softmax of one
// wave meets MFMA of its SIMD partner (4-stage ring): 725 TF vs 844.
// Round-2 probes of the main loop (profiles/r02_attention_variants.txt): v_pk_fma/v_pk_add_f32 for the scale-shift and
// the row sums (25 fewer VALU per tile) 3-5 % slower; operand fragments read 3-4 MFMAs ahead: no gain; no per-tile
// barrier (timing probe): +1..6 %.  Counters at B = 4, S = 8704: matrix pipe 52 % busy at 1.86 GHz, waves 37 % parked,
// 32 % issue-stalled, LDS array ~26 % busy, no bank conflicts.

template <int NW, int STAGES, bool F32OUT, bool ILV>
int launch(AttnParams p, hipStream_t stream, int n_full, int n_light_blocks, int light_subs) {
  constexpr int SMEM = STAGES * STAGE_BYTES + 16;   // ring + the restart flag word
  auto kern = attention_fwd_kernel<NW, STAGES, F32OUT, ILV>;
  FK_ENSURE_MAX_LDS(kern, SMEM, "fk_attention_fwd_bf16");
  p.n_full = n_full;
  p.light_subs = light_subs;
  hipLaunchKernelGGL(kern, dim3(n_full + n_light_blocks * light_subs), dim3(NW * 64), SMEM, stream, p);
  FK_CHECK_LAUNCH("fk_attention_fwd_bf16");
  return FK_OK;
}

// The last, partly filled round of a grid.  One workgroup (8 waves x 32 query rows) per CU means the grid runs in
// rounds of #CUs blocks; at batch 1 and S = 8704 that is 816 blocks = 3 full rounds + 48 blocks that hold 48 CUs for a
// whole fourth round while 208 idle.  Those tail blocks can be run as FOUR (or two) "light" workgroups each, at the end
// of the same launch: all 8 waves of a light workgroup keep issuing their share of the K / V LDS-DMA requests (the
// request issue, ~60 cycles apiece, is what a wave cannot afford to do alone: a 2-wave workgroup that loads for itself
// measured SLOWER in the edit), but only 2 (4) waves own query rows -- one computing wave per SIMD instead of two, so
// each runs faster and the tail spreads over 192 CUs.  Every query row's arithmetic is the same whichever workgroup
// shape carries it: bit-identical output (tests/test_hip_cfg3.py).  MEASURED, BOTH FORMS, AND OFF BY DEFAULT
// (profiles/r03_attention_variants.txt): isolated the light workgroups gain 1.5-3 % at S = 8704 / 5632 as a second launch of
// 2-wave workgroups and lose 2 % inside the launch; inside the 1024^2 edit both forms LOSE (1492 -> 1524 ms and
// 1634 -> 1695 ms of attention per edit): a lone wave per SIMD does not run enough faster than two sharing one to pay
// for the extra workgroups' prologues, and the idle CUs of the plain grid's last round are not wasted -- they hand
// their power budget to the busy ones.  fk_attention_set_tail(1) / FK_ATTN_TAIL=1 select it for measurements.
static int g_attn_tail = -2;
static int g_attn_ring = -1;
static int attn_tail_mode() {
  if (g_attn_tail == -2) {
    const char* e = getenv("FK_ATTN_TAIL");
    g_attn_tail = e ? atoi(e) : 0;
  }
  return g_attn_tail;
}
static int attn_cu_count() {
  static int cus = 0;
  if (!cus) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    else return 256;
  }
  return cus;
}


// FK_ATTN_ILV=0|1 forces the instruction order of the main loop (A/B measurement); default: by grid size.
static bool use_interleaved(const AttnParams& p) {
  static int ov = -2;
  if (ov == -2) {
    const char* e = getenv("FK_ATTN_ILV");
    ov = e ? atoi(e) : -1;
  }
  if (ov >= 0) return ov != 0;
  // Measured (profiles/r02_attention_variants.txt): isolated, the interleaved order is +3-4 % on grids of many rounds
  // (B = 4, S = 8704: 1157 vs 1114 TF/s) and -2 % on 1-3 round grids; inside an edit, where the kernel runs at the
  // clock the neighbouring GEMMs leave it, it is +1.7 % at S = 8704, B = 1 and even at S = 2560 (240 workgroups).
  const int64_t nwg = (int64_t)((p.S + 255) / 256) * p.H * p.B;
  return nwg >= 512;
}

int attention_entry(const void* q, const void* k, const void* v, void* o, int32_t B, int32_t H, int32_t S, int64_t v_ld,
                    int64_t v_batch_stride, int64_t o_ld, int64_t o_batch_stride, float scale, bool f32out,
                    hipStream_t stream, float* lse = nullptr) {
  FK_CHECK_ARG(q && k && v && o, "fk_attention_fwd_bf16: null pointer");
  FK_CHECK_ARG(B > 0 && H > 0 && S > 0, "fk_attention_fwd_bf16: bad B/H/S %d %d %d", B, H, S);
  FK_CHECK_ARG(o_ld % 4 == 0 && o_batch_stride % 4 == 0 && ((uintptr_t)o % (f32out ? 16 : 8) == 0),
               "fk_attention_fwd_bf16: output must be 8-byte aligned (o_ld %% 4 == 0)");
  FK_CHECK_ARG(v_ld % 8 == 0 && v_batch_stride % 8 == 0, "fk_attention_fwd_bf16: V strides must be multiples of 8");
  FK_CHECK_ARG(((uintptr_t)q % 16 == 0) && ((uintptr_t)k % 16 == 0) && ((uintptr_t)v % 16 == 0),
               "fk_attention_fwd_bf16: q/k/v must be 16-byte aligned");
  // one (batch, head)'s K and V are addressed through 32-bit buffer descriptors / offsets
  FK_CHECK_ARG(v_ld > 0 && (int64_t)S * HD * 2 < (1ll << 31) && ((int64_t)(S - 1) * v_ld + HD) * 2 < (1ll << 31) &&
                   (int64_t)(S + KVBLK) * v_ld * 2 < (1ll << 31),
               "fk_attention_fwd_bf16: S = %d with v_ld = %lld exceeds the 2 GiB a (batch, head)'s K / V may span", S,
               (long long)v_ld);
  AttnParams p;
  p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.v = (const bf16_t*)v; p.o = (bf16_t*)o;
  p.B = B; p.H = H; p.S = S; p.v_ld = v_ld; p.v_bs = v_batch_stride; p.o_ld = o_ld; p.o_bs = o_batch_stride;
  p.scale_log2 = scale * 1.4426950408889634f;
  p.lse = lse;
  const int nblk = ((S + 255) / 256) * H * B;
  if (f32out) return launch<8, 3, true, false>(p, stream, nblk, 0, 0);
  const int G = attn_cu_count();
  int tail = 0, subs = 0;                          // blocks handed to light workgroups, light workgroups per block
  if (attn_tail_mode() && nblk > G && nblk % G != 0) {
    const int rem = nblk % G;
    if (4 * rem <= G) { tail = rem; subs = 4; }
    else if (2 * rem <= G) { tail = rem; subs = 2; }
  }
  if (g_attn_ring < 0) {                            // FK_ATTN_RING=4: 4-stage ring, one barrier per two KV tiles
    const char* e = getenv("FK_ATTN_RING");
    g_attn_ring = e ? atoi(e) : 3;
  }
  if (g_attn_ring == 4)
    return use_interleaved(p) ? launch<8, 4, false, true>(p, stream, nblk - tail, tail, subs)
                              : launch<8, 4, false, false>(p, stream, nblk - tail, tail, subs);
  return use_interleaved(p) ? launch<8, 3, false, true>(p, stream, nblk - tail, tail, subs)
                            : launch<8, 3, false, false>(p, stream, nblk - tail, tail, subs);
}

}  // namespace

extern "C" int fk_attention_set_ring(int32_t stages) {
  FK_CHECK_ARG(stages == 3 || stages == 4, "fk_attention_set_ring: %d is not 3 (a barrier per KV tile) or 4 (a barrier per two)", stages);
  g_attn_ring = stages;
  return FK_OK;
}

extern "C" int fk_attention_set_tail(int32_t mode) {
  FK_CHECK_ARG(mode == 0 || mode == 1, "fk_attention_set_tail: %d is not 0 (plain grid) or 1 (light workgroups for the last round)", mode);
  g_attn_tail = mode;
  return FK_OK;
}

extern "C" int fk_attention_fwd_bf16(const void* q, const void* k, const void* v, void* o, int32_t B,
                                     int32_t H, int32_t S, int64_t v_ld, int64_t v_batch_stride, int64_t o_ld,
                                     int64_t o_batch_stride, float scale, fk_stream_t stream_) {
  return attention_entry(q, k, v, o, B, H, S, v_ld, v_batch_stride, o_ld, o_batch_stride, scale, false,
                         (hipStream_t)stream_);
}

extern "C" int fk_attention_fwd_lse_bf16(const void* q, const void* k, const void* v, void* o, float* lse, int32_t B,
                                         int32_t H, int32_t S, int64_t v_ld, int64_t v_batch_stride, int64_t o_ld,
                                         int64_t o_batch_stride, float scale, fk_stream_t stream_) {
  FK_CHECK_ARG(lse != nullptr, "fk_attention_fwd_lse_bf16: null lse");
  return attention_entry(q, k, v, o, B, H, S, v_ld, v_batch_stride, o_ld, o_batch_stride, scale, false,
                         (hipStream_t)stream_, lse);
}

extern "C" int fk_attention_fwd_f32_debug(const void* q, const void* k, const void* v, float* o, int32_t B,
                                          int32_t H, int32_t S, int64_t v_ld, int64_t v_batch_stride, int64_t o_ld,
                                          int64_t o_batch_stride, float scale, fk_stream_t stream_) {
  return attention_entry(q, k, v, o, B, H, S, v_ld, v_batch_stride, o_ld, o_batch_stride, scale, true,
                         (hipStream_t)stream_);
}
