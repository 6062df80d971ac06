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
#include "attention_common.h"

namespace {

// F32OUT (parity / debug build of the same kernel, fk_attention_fwd_f32_debug): the output is written as fp32 and
// the probabilities enter the PV product as TWO bf16 terms (p = hi + lo, 16 mantissa bits instead of 8), so that
// the result can be held against an fp32 reference at rtol 1e-3 / atol 1e-4 -- with one bf16 term the rounding of
// P alone (2^-9 per term) sits above that tolerance whatever the kernel does.
template <int NW, bool F32OUT, bool ILV, bool STREAMK>
__global__ __launch_bounds__(NW * 64, NW >= 8 ? 2 : 1) void attention_fwd_kernel(const AttnParams p) {
  constexpr int STAGES = 3;           // K / V ring: one barrier per tile, tile kt + 2 requested when tile kt is published
  constexpr int QBLK = NW * 32;
  static_assert(QBLK == 256, "the item list and the stream-K partial layout assume 256 query rows per workgroup");
  constexpr int LOADS = 32 / NW;      // DMA instructions per wave per tile (16 K pieces + 16 V pieces / NW)
  constexpr int KL = LOADS / 2;       // K pieces per wave (same number of V pieces)
  constexpr int PF = STAGES - 1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ql = lane & 31;   // query row inside the wave / operand row
  const int hh = lane >> 5;   // half

  // XCD-aware order: consecutive positions -- items of one (b, h), which share K / V -- stay on one XCD's L2
  const int nqb = (p.S + QBLK - 1) / QBLK;
  const int nkt = (p.S + KVBLK - 1) / KVBLK;
  int pos;
  {
    const int nwg = gridDim.x;
    const int q8 = nwg >> 3, r8 = nwg & 7;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    pos = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
  }
  // ---- the workgroup's work list ------------------------------------------------------------------------------------------
  // Plain launch: the one item `pos`.  Stream-K: sk_rounds whole rounds first (item r * G + pos in round r: the 32
  // workgroups of an XCD then walk 32 consecutive items -- one head's K / V -- in step, which is what keeps K / V in the XCD's
  // L2; dealing out ALL tiles contiguously measured 5-10 % below the plain grid's rate at equal occupancy), then the tail:
  // with U = (n_items - sk_rounds * G) * nkt units left, cut j lies at floor(U * j / G), moved onto the item boundary
  // when it would leave a part shorter than min_part tiles (a seam costs about two tiles' time); the launcher guarantees
  // U / G >= nkt + 2 * min_part, so an item is cut at most once.  The two parts of a cut item meet through workspace slot j
  // (below, after the pass).
  int u = 0, u_end = 0, round = 0;          // fit 32 bits: the launcher checks n_items * nkt < 2^31
  if constexpr (STREAMK) {
    const unsigned G = gridDim.x;
    const unsigned U = (unsigned)(p.n_items - p.sk_rounds * (int)G) * (unsigned)nkt;
    const unsigned qU = U / G, rU = U - qU * G;
    auto cut = [&](unsigned j) __attribute__((always_inline)) {
      unsigned c = qU * j + (rU * j) / G;                      // floor(U * j / G) without a 64-bit product
      const unsigned r = c % (unsigned)nkt;
      if (r != 0 && r < (unsigned)p.min_part) c -= r;
      else if (r != 0 && (unsigned)nkt - r < (unsigned)p.min_part) c += (unsigned)nkt - r;
      return (int)c;
    };
    u = cut(pos);
    u_end = cut(pos + 1);
  }

  const int prow = lane >> 4, pslot = lane & 15;   // LDS-DMA pieces: one instruction = 4 key rows x 256 B
  // V^T operand via transpose read: lane supplies the 8-byte piece V[key0 + j][32 df + 16 dhalf + 4 q4 ..]
  const int tj = (lane & 15) >> 2, tq = lane & 3, tdh = (lane >> 4) & 1;
  const int v_rd = K_TILE_BYTES + (4 * hh + tj) * 256 + tdh * 32 + tq * 8;  // + (16 st + 8 part)*256 + ((df ^ tj) << 6)
  // K operand (A of S^T): row key = 32 kb + ql, chunk 2kk + hh, swizzled by key & 15 (= ql & 15)
  const int k_rd = ql * 256;           // + kb*8192 + (((2kk + hh) ^ (ql & 15)) << 4)
  const int k_sw = ql & 15;
#if FK_ATTN_PRIO
  // Two waves share every SIMD (waves w and w + NW/2); the later-dispatched one loses the VALU arbitration (priority,
  // then age) at the head of every segment.  ONE static priority raise for that half, no per-segment flips.  The
  // condition must be provably wave-uniform: s_setprio ignores EXEC.
  if (wave >= NW / 2) __builtin_amdgcn_s_setprio(1);
#endif

  for (;;) {   // one pass per (item, KV-tile range); a plain launch makes exactly one
  int item = pos, kt0 = 0, kt1 = nkt;                   // this pass: KV tiles [kt0, kt1) of the item
  if constexpr (STREAMK) {
    if (round < p.sk_rounds) {
      item = round * (int)gridDim.x + pos;
      ++round;
    } else {
      // The range is walked from its END: the head part of the last item first (it starts at tile 0, like every whole
      // item), the tail part of the first item last.  All workgroups of an XCD then stream K / V tiles 0, 1, 2, ... of
      // (mostly) one head in step again; walked from the front, their first tiles were spread over the whole sequence
      // and the XCD's L2 served 32 unrelated streams (counted: 6.5 x the algorithmic bytes at S = 8704 against 1.6 x
      // for the plain grid, L2 hit 65 % against 92 %: profiles/r04_traffic.md).  The seam merge is symmetric: any order.
      if (u >= u_end) break;
      const int ti = (unsigned)(u_end - 1) / (unsigned)nkt;
      item = p.sk_rounds * (int)gridDim.x + ti;
      kt1 = u_end - ti * nkt;
      kt0 = max(u - ti * nkt, 0);
      u_end -= kt1 - kt0;
    }
  }
  const int qb = item % nqb;
  const int bh = item / nqb;
  const int b = bh / p.H, h = bh - b * p.H;
  const int q_row0 = qb * QBLK;

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
  // Buffer form of the LDS-DMA load: descriptor in SGPRs, ONE 32-bit offset VGPR per piece (constant over the
  // tiles), the tile offset in an SGPR.  Rows >= S lie beyond num_records and are fetched as zeros (they are masked).
  const DmaDesc rs_k = make_dma_desc(Kg, (int64_t)p.S * HD * 2);
  const DmaDesc rs_v = make_dma_desc(Vg, ((int64_t)(p.S - 1) * p.v_ld + HD) * 2);
  int k_voff[KL], v_voff[KL];
#pragma unroll
  for (int i = 0; i < KL; ++i) {
    const int r = (wave * KL + i) * 4 + prow;                       // key row inside the tile
    k_voff[i] = (r * HD + ((pslot ^ (r & 15)) << 3)) * 2;
    const int vcol = ((((pslot >> 2) ^ (r & 3)) << 5) + ((pslot & 3) << 3));
    v_voff[i] = (int)((r * p.v_ld + vcol) * 2);
  }
  const int k_tile_bytes = KVBLK * HD * 2, v_tile_bytes = (int)(KVBLK * p.v_ld * 2);
  // (every lambda of the kernel is force-inlined: a tile body left as an out-of-line call takes its captures by address and
  //  the accumulator arrays land in private memory -- 25x slower; tests/test_kernel_resources.py watches for it)
  auto issue_tile = [&](int kt, int stage) __attribute__((always_inline)) {
    char* sb = smem + stage * STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < KL; ++i) {
      buffer_lds16(rs_k, sb + (wave * KL + i) * 1024, k_voff[i], kt * k_tile_bytes);
      buffer_lds16(rs_v, sb + K_TILE_BYTES + (wave * KL + i) * 1024, v_voff[i], kt * v_tile_bytes);
    }
  };

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

  int st_cur = 0, st_pf = PF;
  auto fill = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int s = 0; s < PF; ++s)
      if (kt0 + s < kt1) issue_tile(kt0 + s, s);
    st_cur = 0;
    st_pf = PF;
  };
  // ring bookkeeping shared by the main loop and the pre-pass: wait for tile kt, publish it, request the tile ahead
  // (a four-stage ring with one barrier per two tiles was measured 1-3 % slower at every shape in round 3 and is gone)
  auto acquire_tile = [&](int kt) __attribute__((always_inline)) {
    if (kt + PF - 1 < kt1) wait_vmcnt<(PF - 1) * LOADS>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (kt + PF < kt1) issue_tile(kt + PF, st_pf);
    return smem + st_cur * STAGE_BYTES;
  };
  auto release_tile = [&]() __attribute__((always_inline)) {
    st_cur = (st_cur == STAGES - 1) ? 0 : st_cur + 1;
    st_pf = (st_pf == STAGES - 1) ? 0 : st_pf + 1;
  };
  // Operand fragments.  The reads are issued PF_DEPTH MFMAs ahead of their use and the order is pinned
  // (sched_group_barrier: one DS read, then one MFMA): left alone, hipcc issues every ds_read right in front of
  // its MFMA and the wave waits for an LDS round trip 32 times per tile.
  auto k_frag = [&](const char* sb, int kb, int kk) __attribute__((always_inline)) {
    return *(const bf16x8_t*)(sb + k_rd + kb * 8192 + (((2 * kk + hh) ^ k_sw) << 4));
  };
  auto v_frag = [&](const char* sb, int st, int df) __attribute__((always_inline)) {   // keys 16 st + {0, 8} + 4 hh + 0..3, d block df
    const char* vp = sb + v_rd + st * 4096 + ((df ^ tj) << 6);
    const s16x4_t lo = lds_tr16(vp);
    const s16x4_t hi = lds_tr16(vp + 2048);
    bf16x8_t vf;
    vf[0] = lo[0]; vf[1] = lo[1]; vf[2] = lo[2]; vf[3] = lo[3];
    vf[4] = hi[0]; vf[5] = hi[1]; vf[6] = hi[2]; vf[7] = hi[3];
    return vf;
  };
  // S^T block kb (32 keys x 32 queries) of the tile at sb, masked beyond S in the ragged last tile
  auto scores = [&](const char* sb, int kb, int kt, auto mask_tag) __attribute__((always_inline)) {
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
  auto block_max = [&](const f32x16_t& s) __attribute__((always_inline)) {   // row maximum over one 32-key block, in log2 units
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
  auto do_tile = [&](int kt, auto mask_tag, auto first_tag) __attribute__((always_inline)) {
    if constexpr (!ILV) {
      constexpr bool FIRST = decltype(first_tag)::value;
      const char* sb = acquire_tile(kt);
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
      float psum = 0.f;
      f32x16_t s0 = scores(sb, 0, kt, mask_tag);
      if constexpr (FIRST) m_ref = block_max(s0) + REF_BIAS;   // reference = row maximum over the first 32 keys + bias
      const float nm = -m_ref;
      // softmax numerators of two scores (log2 domain; raw v_exp_f32, denormal results may flush)
      auto expo2 = [&](f32x16_t& s, int r) __attribute__((always_inline)) {
  #pragma unroll
        for (int e = 0; e < 2; ++e) {
          const float pv = __builtin_amdgcn_exp2f(fmaf(s[r + e], p.scale_log2, nm));
          s[r + e] = pv;
          psum += pv;
        }
      };
      auto pack8 = [&](const f32x16_t& s, int r0, bf16x8_t& pf, bf16x8_t& pf_lo) __attribute__((always_inline)) {
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
  const bool ragged = p.S % KVBLK != 0;
  const bool last_masked = ragged && kt1 == nkt;     // the pass ends with the item's ragged tile
  int* const wg_flag = (int*)(smem + STAGES * STAGE_BYTES);   // one word past the ring (allocated by the launcher)
  // tiles [kt0, kt1): the first one sets the exponent reference (attempt 0), only the item's last one can need the mask
  auto run_tiles = [&](auto first_tag) __attribute__((always_inline)) {
    const int last = kt1 - 1;
    if (kt0 == last) {
      if (last_masked) do_tile(kt0, TT{}, first_tag);
      else do_tile(kt0, FF{}, first_tag);
      return;
    }
    do_tile(kt0, FF{}, first_tag);
    for (int kt = kt0 + 1; kt < last; ++kt) do_tile(kt, FF{}, FF{});
    if (last_masked) do_tile(last, TT{}, FF{});
    else do_tile(last, FF{}, FF{});
  };
  for (int attempt = 0; attempt < 2; ++attempt) {
    if (attempt == 1) {
      // exact row maxima: K-only pre-pass over the pass's tiles (V rides along in the ring)
      fill();
      m_ref = -3.0e38f;
      for (int kt = kt0; kt < kt1; ++kt) {
        const char* sb = acquire_tile(kt);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
          if (last_masked && kt == kt1 - 1) m_ref = fmaxf(m_ref, block_max(scores(sb, kb, kt, TT{})));
          else m_ref = fmaxf(m_ref, block_max(scores(sb, kb, kt, FF{})));
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
    if (attempt == 0) run_tiles(TT{});
    else run_tiles(FF{});
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

  // ---- stream-K seam: the two parts of a cut item ------------------------------------------------------------------------
  // Each part holds, per query row, unnormalised sums against its OWN exponent reference: (O^T, l, m_ref).  Whoever
  // finishes FIRST (ticket = an agent-scope fetch-add on the cut's counter word: even -> first) writes its triple to the
  // cut's workspace slot with write-through stores, drains them, publishes flag = ticket + 1 and goes on to its next item;
  // the SECOND (odd ticket) waits for flag == its ticket, acquires, rescales both triples to the larger reference,
  //   O = O_self * 2^(m_self - m) + O_other * 2^(m_other - m)        (two rounded products, one rounded sum: symmetric)
  // and finalises.  The result does not depend on which part arrives second.  The wait is only ever for a workgroup that
  // has already drawn its ticket -- resident and microseconds from publishing -- so no dispatch order is assumed; counter
  // and flag are monotonic (every launch adds exactly two tickets per cut it uses): nothing is reset between launches
  // or graph replays.  Recipe: cdna_hip_programming.md Guideline 16 R1, as the split-K GEMM pairs use it.
  if constexpr (STREAMK) {
    if (kt0 > 0 || kt1 < nkt) {                     // workgroup-uniform
      typedef __attribute__((address_space(1))) unsigned gu32;
      const int slot = kt1 < nkt ? pos + 1 : pos;   // the cut's index, 1 .. G - 1
      gu32* const ctl = (gu32*)(p.sk_ctl + 2 * (size_t)slot);
      const __amdgpu_buffer_rsrc_t rs_p = __builtin_amdgcn_make_buffer_rsrc(
          (void*)(p.sk_partials + (size_t)slot * PART_FLOATS), 0, PART_FLOATS * 4, 0x00020000);
      __syncthreads();   // every wave is done with the ring: its first word now carries the ticket
      if (tid == 0) *(volatile unsigned*)smem = __hip_atomic_fetch_add(ctl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __syncthreads();
      const unsigned ticket = __builtin_amdgcn_readfirstlane(*(volatile unsigned*)smem);
      constexpr int LM_OFF = 16 * 512 * 16;         // (l, m_ref) pairs behind the 16 x 512 accumulator pieces
      if ((ticket & 1u) == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const f32x16_t& a = o[r >> 2];
          const int q4 = r & 3;
          const u32x4_t v = {__float_as_uint(a[4 * q4]), __float_as_uint(a[4 * q4 + 1]), __float_as_uint(a[4 * q4 + 2]),
                             __float_as_uint(a[4 * q4 + 3])};
          __builtin_amdgcn_raw_buffer_store_b128(v, rs_p, tid * 16, r * (512 * 16), /*sc1: write through*/ 16);
        }
        __builtin_amdgcn_raw_buffer_store_b64(u32x2_t{__float_as_uint(l_run), __float_as_uint(m_ref)}, rs_p, LM_OFF + tid * 8, 0, 16);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains its own stores
        __syncthreads();
        if (tid == 0) __hip_atomic_store(ctl + 1, ticket + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();                                   // the ring is refilled by the next pass
        continue;
      }
      if (tid == 0) {
        // bounded spin (~seconds): the partner has drawn its ticket, so it is resident and microseconds from publishing; a
        // corrupted workspace must end in NaN rows (below), not in a hung device
        int spins = 0;
        while (__hip_atomic_load(ctl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != ticket && spins < (1 << 22)) {
          __builtin_amdgcn_s_sleep(8);
          ++spins;
        }
        *(volatile unsigned*)smem = spins >= (1 << 22);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      }
      __syncthreads();
      const bool gave_up = *(volatile unsigned*)smem != 0;
      const u32x2_t lm = __builtin_amdgcn_raw_buffer_load_b64(rs_p, LM_OFF + tid * 8, 0, /*sc1*/ 16);
      const float l_o = __uint_as_float(lm[0]), m_o = __uint_as_float(lm[1]);
      const float m_new = fmaxf(m_ref, m_o);
      const float w_s = m_ref == m_new ? 1.0f : __builtin_amdgcn_exp2f(m_ref - m_new);
      const float w_o = m_o == m_new ? 1.0f : __builtin_amdgcn_exp2f(m_o - m_new);
#pragma unroll
      for (int r0 = 0; r0 < 16; r0 += 4) {
        u32x4_t v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = __builtin_amdgcn_raw_buffer_load_b128(rs_p, tid * 16, (r0 + e) * (512 * 16), /*sc1*/ 16);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          f32x16_t& a = o[(r0 + e) >> 2];
          const int q4 = (r0 + e) & 3;
#pragma unroll
          for (int j = 0; j < 4; ++j) a[4 * q4 + j] = merge2(a[4 * q4 + j], w_s, __uint_as_float(v[e][j]), w_o);
        }
      }
      l_run = merge2(l_run, w_s, l_o, w_o);
      if (gave_up) l_run = __builtin_nanf("");
      m_ref = m_new;
    }
  }

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
  if constexpr (!STREAMK) break;
  else __syncthreads();   // every wave is done with the ring and the ticket word before the next pass
  }   // passes
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

template <int NW, bool F32OUT, bool ILV, bool STREAMK>
int launch(const AttnParams& p, int grid, hipStream_t stream) {
  constexpr int SMEM = 3 * STAGE_BYTES + 16;   // ring + the restart flag word
  auto kern = attention_fwd_kernel<NW, F32OUT, ILV, STREAMK>;
  FK_ENSURE_MAX_LDS(kern, SMEM, "fk_attention_fwd_bf16");
  hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), SMEM, stream, p);
  FK_CHECK_LAUNCH("fk_attention_fwd_bf16");
  return FK_OK;
}

// FK_ATTN_KERNEL=4|8 (read once; A/B and tests): the two kernels give the same bits, so the choice is a launch decision like
// the instruction order (FK_ATTN_ILV).  Default: the 4-wave x 64-row kernel of attention_fwd4.hip, 5 to 14 % faster than the
// 8-wave one at every shape of the path (profiles/r05_attention_ab_callJ.txt); the 8-wave kernel stays as the second
// implementation the parity tests compare it with (and serves the fp32-output debug form).
static int attn_kernel_choice() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("FK_ATTN_KERNEL");
    v = e && atoi(e) == 8 ? 8 : 4;
  }
  return v;
}

// ---- the grid ----------------------------------------------------------------------------------------------------------
// One workgroup (8 waves x 32 query rows) per CU means a plain grid runs in rounds of #CUs items: at batch 1 and S = 8704
// that is 816 items = 3 full rounds + 48 items that hold 48 CUs for a whole fourth round while 208 idle (S = 5632: 528
// items, 2.06 rounds -> 3).  Round 3 measured two ways of thinning that last round (light workgroups, in a second launch
// and inside the launch) and both lost inside the edit; they are gone (git history, profiles/r03_attention_variants.txt).
// Round 4: STREAM-K.  A persistent grid of one workgroup per CU; the n_items * nkt KV-tile units are dealt out as G equal
// contiguous ranges, so every CU works the same time and an item that straddles two CUs is finished by whichever of the
// two arrives second (kernel, "stream-K seam").  Used when the plain grid would waste >= 4 % of its rounds and the items
// are long enough that each is cut at most once.  An item's bits then depend on WHERE it is cut, i.e. on the grid:
// `grid` = -1 of fk_attention_fwd_ws_bf16 ("batch-invariant", like fk_gemm_args.plan = FK_GEMM_PLAN_BATCH_INVARIANT) keeps
// the plain grid.  The choice comes with the CALL: the library keeps no launch state.
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
constexpr int ATTN_MIN_PART = 8;           // KV tiles: the shortest part a cut may leave
constexpr int ATTN_CTL_BYTES = 16384;      // head of the workspace: (ticket, flag) pairs of up to 2048 cuts

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
  return p.n_items >= 512;
}

int attention_entry(const void* q, const void* k, const void* v, void* o, int32_t B, int32_t H, int32_t S, int64_t v_ld,
                    int64_t v_batch_stride, int64_t o_ld, int64_t o_batch_stride, float scale, bool f32out,
                    hipStream_t stream, float* lse = nullptr, void* ws = nullptr, int64_t ws_bytes = 0, int grid = 0) {
  FK_CHECK_ARG(grid >= -1 && grid != 1, "fk_attention_fwd_ws_bf16: grid %d is not 0 (stream-K grid where the plain one wastes a round), "
               "-1 (always one workgroup per 256-row block) or a workgroup count >= 2 (test hook)", grid);
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
  const int64_t n_items = (int64_t)((S + 255) / 256) * H * B, nkt = (S + KVBLK - 1) / KVBLK;
  FK_CHECK_ARG(n_items * nkt < (1ll << 31), "fk_attention_fwd_bf16: B * H * S^2 too large for 32-bit tile counters");
  AttnParams p;
  p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.v = (const bf16_t*)v; p.o = (bf16_t*)o;
  p.B = B; p.H = H; p.S = S; p.v_ld = v_ld; p.v_bs = v_batch_stride; p.o_ld = o_ld; p.o_bs = o_batch_stride;
  p.scale_log2 = scale * 1.4426950408889634f;
  p.lse = lse;
  p.n_items = (int)n_items;
  p.min_part = ATTN_MIN_PART;
  p.sk_rounds = 0;
  p.sk_partials = nullptr;
  p.sk_ctl = nullptr;
  if (f32out) return launch<8, true, false, false>(p, (int)n_items, stream);
  const int mode = grid == 0 ? 1 : (grid < 0 ? 0 : grid);   // 0: never; 1: where it pays; >= 2 (test hook): a persistent grid of `mode` workgroups
  const int G = mode >= 2 ? (mode < attn_cu_count() ? mode : attn_cu_count()) : attn_cu_count();
  const int64_t rounds = (n_items + G - 1) / G;
  const bool wasteful = mode >= 2 || (n_items > G && (rounds * G - n_items) * 25 >= rounds * G);   // >= 4 % of the rounds' CU time idle
  // whole rounds in front of the dealt-out tail: as many as leave every workgroup's share of the tail longer than an item
  // plus two minimum parts (an item is then cut at most once)
  int sk_rounds = (int)(n_items / G) - 1;
  while (sk_rounds >= 0 && (n_items - (int64_t)sk_rounds * G) * nkt < (int64_t)G * (nkt + 2 * ATTN_MIN_PART)) --sk_rounds;
  const bool cut_once = sk_rounds >= 0;
  const int64_t need = ATTN_CTL_BYTES + (int64_t)G * PART_FLOATS * 4;
  if (mode && wasteful && cut_once && ws && ws_bytes >= need && G <= ATTN_CTL_BYTES / 8) {
    p.sk_rounds = sk_rounds;
    FK_CHECK_ARG((uintptr_t)ws % 16 == 0, "fk_attention_fwd_ws_bf16: workspace must be 16-byte aligned");
    // control words FIRST, at a place that does not depend on the grid (a test-hook grid of 7 workgroups would otherwise
    // read its tickets from what a 256-workgroup launch used as partial storage), partials behind them
    p.sk_ctl = (unsigned*)ws;
    p.sk_partials = (float*)((char*)ws + ATTN_CTL_BYTES);
    if (attn_kernel_choice() == 4) return fk_attention_fwd4_launch(p, G, true, stream);
    return use_interleaved(p) ? launch<8, false, true, true>(p, G, stream) : launch<8, false, false, true>(p, G, stream);
  }
  if (attn_kernel_choice() == 4) return fk_attention_fwd4_launch(p, (int)n_items, false, stream);
  return use_interleaved(p) ? launch<8, false, true, false>(p, (int)n_items, stream)
                            : launch<8, false, false, false>(p, (int)n_items, stream);
}

}  // namespace

// one slot per cut of the persistent grid (= per CU of the current device): the fp32 partial + its (ticket, flag) pair
extern "C" int64_t fk_attention_ws_bytes(void) { return ATTN_CTL_BYTES + (int64_t)attn_cu_count() * PART_FLOATS * 4; }

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

extern "C" int fk_attention_fwd_ws_bf16(const void* q, const void* k, const void* v, void* o, float* lse, int32_t B, int32_t H,
                                        int32_t S, int64_t v_ld, int64_t v_batch_stride, int64_t o_ld, int64_t o_batch_stride,
                                        float scale, void* ws, int64_t ws_bytes, int32_t grid, fk_stream_t stream_) {
  return attention_entry(q, k, v, o, B, H, S, v_ld, v_batch_stride, o_ld, o_batch_stride, scale, false,
                         (hipStream_t)stream_, lse, ws, ws_bytes, grid);
}

extern "C" int fk_attention_fwd_f32_debug(const void* q, const void* k, const void* v, float* o, int32_t B,
                                          int32_t H, int32_t S, int64_t v_ld, int64_t v_batch_stride, int64_t o_ld,
                                          int64_t o_batch_stride, float scale, fk_stream_t stream_) {
  return attention_entry(q, k, v, o, B, H, S, v_ld, v_batch_stride, o_ld, o_batch_stride, scale, true,
                         (hipStream_t)stream_);
}
