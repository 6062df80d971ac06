// Attention forward, 4 waves x 64 query rows (one wave per SIMD): its own translation unit so that it can be built with
// -fno-slp-vectorize (v_pk_* forms beside MFMAs cost more than the scalar pairs they replace: +10 % on this kernel) without
// touching the code hipcc generates for the 8-wave kernel of attention_fwd.hip.
#include "attention_common.h"

namespace {

// Build-time switches of the experiments behind DESIGN.md section 4.0's attention table (defaults = what measured best):
#ifndef FK_A4_DMA
#define FK_A4_DMA 2      // a tile's 8 LDS-DMA requests: 0 in front of the first K-fragment reads, 1 behind them (under their
#endif                   // latency), 2 one per MFMA slot of the tile's first group, 3 four in each block's third group, 4 two in the first and third group of each block
#ifndef FK_A4_EARLY
#define FK_A4_EARLY 1    // 1: V^T fragments read one group earlier (behind the MFMA that frees the register), K likewise
#endif
#ifndef FK_A4_EWAIT
#define FK_A4_EWAIT 2    // idle states in front of a block's first softmax step (s_nop operand; the hazard needs 12 states in all)
#endif
#define FK_STR2(x) #x
#define FK_STR(x) FK_STR2(x)
constexpr int A4_STAGES = 3, A4_DMA = FK_A4_DMA;
constexpr bool A4_EARLY = FK_A4_EARLY != 0;

// ---- 4 waves, one per SIMD, 64 query rows per wave (round 5; the default forward) --------------------------------------------
// The 8-wave kernel of attention_fwd.hip is issue-bound: per KV tile and wave 32 MFMAs beside ~6.5 other instructions each, and
// the counters say matrix time and vector time ADD on a SIMD that two waves share.  What moves that
// bound is fewer non-matrix instructions per MFMA, and the one large item is the operand reads: here a wave owns TWO 32-row
// query blocks (A, B), so every K fragment and every V^T fragment it reads from LDS feeds two MFMAs -- 0.75 LDS reads per
// MFMA instead of 1.5 -- and the wave has the SIMD's whole register file (O^T of both blocks, 128 accumulator registers, lives
// in the AGPR half).  With one wave per SIMD nothing else hides the softmax arithmetic, so the wave overlaps it with its OWN
// matrix work: the two query blocks run as two streams half a step apart.  Per 32-key block k, four groups of 8 MFMA slots
// (the table in front of groups123 has the exact step-to-slot map):
//     matrix pipe                      issued in the shadow of those MFMAs, one share per slot
//     S_A(k)   = K(k) Q_A^T            softmax steps 7..14 of S_B(k-1);   first block of a tile: the next tile's 8 LDS-DMA requests
//     O_B     += V(k-1) P_B(k-1)       last steps of S_B(k-1), steps 0..6 of S_A(k);   this block's 8 V^T fragments (16 tr reads)
//     S_B(k)   = K(k) Q_B^T            steps 7..14 of S_A(k);             first block of a tile: the second block's 8 K fragments
//     O_A     += V(k) P_A(k)           last steps of S_A(k), steps 0..6 of S_B(k)
// Per tile (64 MFMAs): 224 vector instructions, 48 LDS reads, 8 requests = 4.4 per MFMA, ~445 instructions with the waits,
// address adds and scalar work hipcc adds (the guide's budget for a single wave: 5 besides the MFMA).  Measured:
// 62 % matrix-pipe busy at 1.76 GHz against the 8-wave kernel's 52 % at 1.84 GHz, 1.13-1.31 PF/s = 1.05-1.14 x.
// Only registers cross a tile boundary (S_B, its packed numerators, the V fragments), so the K / V ring and its barrier per
// tile are the 8-wave kernel's.  Every row's sums are formed in the same order as there (tile sums, then the running sum): the
// two kernels agree bit for bit -- outputs and log-sum-exps, plain and stream-K grids -- which is the parity test
// (tests/test_hip_kernels.py::test_attention_two_kernels_agree_bit_for_bit); restart path, ragged last tile and stream-K seam
// likewise (the seam's partial layout is private to this kernel).
template <bool STREAMK>
__global__ __launch_bounds__(256, 1) void attention_fwd4_kernel(const AttnParams p) {
  constexpr int NW = 4, STAGES = A4_STAGES, QBLK = 256, LOADS = 32 / NW, KL = LOADS / 2, PF = STAGES - 1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ql = lane & 31, hh = lane >> 5;
  const int nqb = (p.S + QBLK - 1) / QBLK;
  const int nkt = (p.S + KVBLK - 1) / KVBLK;
  int pos;
  {
    const int nwg = gridDim.x;
    const int q8 = nwg >> 3, r8 = nwg & 7;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    pos = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
  }
  int u = 0, u_end = 0, round = 0;
  if constexpr (STREAMK) {   // the 8-wave kernel's work list (whole rounds, then the tail dealt out from its end)
    const unsigned G = gridDim.x;
    const unsigned U = (unsigned)(p.n_items - p.sk_rounds * (int)G) * (unsigned)nkt;
    const unsigned qU = U / G, rU = U - qU * G;
    auto cut = [&](unsigned j) __attribute__((always_inline)) {
      unsigned c = qU * j + (rU * j) / G;
      const unsigned r = c % (unsigned)nkt;
      if (r != 0 && r < (unsigned)p.min_part) c -= r;
      else if (r != 0 && (unsigned)nkt - r < (unsigned)p.min_part) c += (unsigned)nkt - r;
      return (int)c;
    };
    u = cut(pos);
    u_end = cut(pos + 1);
  }
  const int prow = lane >> 4, pslot = lane & 15;
  const int tj = (lane & 15) >> 2, tq = lane & 3, tdh = (lane >> 4) & 1;
  const int v_rd = K_TILE_BYTES + (4 * hh + tj) * 256 + tdh * 32 + tq * 8;
  const int k_rd = ql * 256;
  const int k_sw = ql & 15;

  for (;;) {
  int item = pos, kt0 = 0, kt1 = nkt;
  if constexpr (STREAMK) {
    if (round < p.sk_rounds) {
      item = round * (int)gridDim.x + pos;
      ++round;
    } else {
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
  const int q_row0 = qb * QBLK + wave * 64;              // rows q_row0 + 32 X + ql, X = 0 (block A), 1 (block B)
  const bf16_t* Kg = p.k + (int64_t)bh * p.S * HD;
  const bf16_t* Vg = p.v + (int64_t)b * p.v_bs + h * HD;

  bf16x8_t qfA[8], qfB[8];
  {
    const bf16_t* qa = p.q + ((int64_t)bh * p.S + min(q_row0 + ql, p.S - 1)) * HD + 8 * hh;
    const bf16_t* qbp = p.q + ((int64_t)bh * p.S + min(q_row0 + 32 + ql, p.S - 1)) * HD + 8 * hh;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      qfA[kk] = *(const bf16x8_t*)(qa + 16 * kk);
      qfB[kk] = *(const bf16x8_t*)(qbp + 16 * kk);
    }
  }
  const DmaDesc rs_k = make_dma_desc(Kg, (int64_t)p.S * HD * 2);
  const DmaDesc rs_v = make_dma_desc(Vg, ((int64_t)(p.S - 1) * p.v_ld + HD) * 2);
  // LDS-DMA requests: a piece = 4 rows x 256 B (lane -> row prow, 16-byte slot pslot); wave w moves pieces w, w + 4, w + 8,
  // w + 12 of a tile's 16.  Interleaved like that the swizzle term of the K source address (r & 15 = 4 w + prow) and the V
  // one (r & 3 = prow) are the same for all four pieces: ONE lane offset each, the piece selected through the scalar offset
  // (two lane constants live across the tile loop instead of eight -- the loop has no register to spare).
  const int k_voff0 = ((wave * 4 + prow) * HD + ((pslot ^ (wave * 4 + prow)) << 3)) * 2;
  const int v_voff0 = (int)(((wave * 4 + prow) * p.v_ld + ((((pslot >> 2) ^ prow) << 5) + ((pslot & 3) << 3))) * 2);
  const int k_tile_bytes = KVBLK * HD * 2, v_tile_bytes = (int)(KVBLK * p.v_ld * 2);
  const int v_piece_step = (int)(16 * p.v_ld * 2);       // pieces w + 4 i: 16 rows apart
  auto issue_tile = [&](int kt, int stage) __attribute__((always_inline)) {
    char* sb = smem + stage * STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < KL; ++i) {
      buffer_lds16(rs_k, sb + (i * NW + wave) * 1024, k_voff0, kt * k_tile_bytes + i * 4096);
      buffer_lds16(rs_v, sb + K_TILE_BYTES + (i * NW + wave) * 1024, v_voff0, kt * v_tile_bytes + i * v_piece_step);
    }
  };

  f32x16_t oA[4], oB[4];
  constexpr float REF_BIAS = 24.0f;
  float mA = 0.f, mB = 0.f, lA = 0.f, lB = 0.f, tA = 0.f, tB = 0.f;   // exponent references, running and tile row sums
  int st_cur = 0, st_pf = PF;
  // The ring.  Forms 0 / 1 request a tile only if the pass has it (counted waits with a tail case); form 2 requests EVERY
  // tile slot -- past the pass's end the last tile again, which nobody reads -- so that exactly 8 requests per tile are in
  // flight and one counted wait serves every tile.
  auto fill = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int s = 0; s < PF; ++s)
      if (A4_DMA >= 2 || kt0 + s < kt1) issue_tile(min(kt0 + s, kt1 - 1), s);
    st_cur = 0;
    st_pf = PF;
  };
  auto acquire_tile = [&](int kt) __attribute__((always_inline)) {      // the restart's pre-pass (fill_plain in front of it)
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
  auto k_frag = [&](const char* sb, int kb, int kk) __attribute__((always_inline)) {
    return *(const bf16x8_t*)(sb + k_rd + kb * 8192 + (((2 * kk + hh) ^ k_sw) << 4));
  };
  auto v_frag = [&](const char* sb, int st, int df) __attribute__((always_inline)) {
    const char* vp = sb + v_rd + st * 4096 + ((df ^ tj) << 6);
    const s16x4_t lo = lds_tr16(vp);
    const s16x4_t hi = lds_tr16(vp + 2048);
    bf16x8_t vf;
    vf[0] = lo[0]; vf[1] = lo[1]; vf[2] = lo[2]; vf[3] = lo[3];
    vf[4] = hi[0]; vf[5] = hi[1]; vf[6] = hi[2]; vf[7] = hi[3];
    return vf;
  };
  auto mask_block = [&](f32x16_t& s, int kt, int kb) __attribute__((always_inline)) {
    const int kbase = kt * KVBLK + 32 * kb + 4 * hh;
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if (kbase + (r & 3) + 8 * (r >> 2) >= p.S) s[r] = -1.0e30f;
  };
  auto block_max = [&](const f32x16_t& s) __attribute__((always_inline)) {
    float mx = s[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
    return fmaxf(mx, __shfl_xor(mx, 32)) * p.scale_log2;
  };
  // pipeline registers
  bf16x8_t kf[8], vfr[8];
  f32x16_t sA, sB;
  u32x4_t pA[2], pB[2];      // packed numerators of the block in flight: keys 0..15 / 16..31 of the block

  // ---- hand-placed groups ------------------------------------------------------------------------------------------------
  // hipcc's own schedule of this loop (pins by sched_group_barrier) clusters the MFMAs and parks the S^T blocks in AGPRs, which
  // the vector unit cannot read (64 v_accvgpr_read per tile): 0.92-0.94 x the 8-wave kernel.  Here every group is written as
  // its eight MFMA slots in program order -- one MFMA, then that slot's share of the other block's softmax (one scale-and-shift
  // FMA, one exponential, one row-sum add, a pack every second slot) and of the fragment reads -- with a scheduling fence per
  // slot, so the source order IS the issue order; and the S^T chains are inline-asm MFMAs on VGPR accumulators (the builtin's
  // result class is the compiler's choice).  hipcc sees neither the matrix instruction nor its hazards inside an asm statement:
  // an XDL result needs 12 wait states before a vector instruction may read it (hipcc puts s_nop 11 between the two when they
  // are adjacent).  The softmax of a block therefore starts in the SECOND slot of the group behind its chain -- the first slot
  // holds the last step of the other stream's half instead -- which leaves 6 to 8 instructions between the chain's last MFMA
  // and the first read (hipcc moves a slot's MFMA about inside its fences); e_wait tops that up.  (As first written the softmax started in the first slot behind 13 idle states: four
  // times a tile the matrix pipe stood still for them.)
#define FK_SLOT_FENCE() __builtin_amdgcn_sched_barrier(0)
  auto s_mfma = [&](f32x16_t& acc, const bf16x8_t& a, const bf16x8_t& b, bool first) __attribute__((always_inline)) {
    if (first) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
  };
  auto s_ready = [&](f32x16_t& acc) __attribute__((always_inline)) {   // the full distance (mask / first-block maximum)
    asm volatile("s_nop 7\n\ts_nop 4" : "+v"(acc));
  };
  auto e_wait = [&](f32x16_t& acc) __attribute__((always_inline)) {    // FK_A4_EWAIT + 1 states + the >= 10 instructions in between
    asm volatile("s_nop " FK_STR(FK_A4_EWAIT) : "+v"(acc));
  };
  // Step t (0..15) of a block's softmax: scale + exponential of element t, row sum of element t - 1, pack of the pair that
  // element t - 1 completed; step 16 is what is left after the last exponential.  The row sums are formed exactly as the 8-wave
  // kernel forms them -- a tile's 32 numerators summed in element order from zero (first block kb = 0, then kb = 1), the tile
  // sum then added to the running sum -- whatever slots the steps land in: bit-identical l, hence lse and O.
  auto e_step = [&](f32x16_t& s, int t, int kb, float nm, float& tsum, float& lrun, u32x4_t (&pk)[2]) __attribute__((always_inline)) {
    if (t < 16) s[t] = __builtin_amdgcn_exp2f(fmaf(s[t], p.scale_log2, nm));
    if (t == 1 && kb == 0) tsum = s[0];
    else if (t > 0) tsum += s[t - 1];
    if (t == 16 && kb == 1) lrun += tsum;
    if (t >= 2 && (t & 1) == 0) pk[(t - 2) >> 3][((t - 2) >> 1) & 3] = pack_bf2(s[t - 2], s[t - 1]);
  };
  auto pfrag = [&](const u32x4_t& w) __attribute__((always_inline)) { return __builtin_bit_cast(bf16x8_t, w); };
  // request i (0..7) of the tile that goes into stage st_pf: K piece i / 2 (even i) or V piece i / 2 (odd i) of this wave
  auto issue_piece = [&](int kt, int i) __attribute__((always_inline)) {
    char* sb = smem + st_pf * STAGE_BYTES;
    const int pc = i >> 1;
    if ((i & 1) == 0) buffer_lds16(rs_k, sb + (pc * NW + wave) * 1024, k_voff0, kt * k_tile_bytes + pc * 4096);
    else buffer_lds16(rs_v, sb + K_TILE_BYTES + (pc * NW + wave) * 1024, v_voff0, kt * v_tile_bytes + pc * v_piece_step);
  };
  // Slot j of a group runs: j = 0 the pending step of the stream that was in the previous group, j >= 1 step base + j - 1 of
  // this group's stream.
  //   group 1  S_A(k)     | B(k-1): step 7 ; steps 8..14                    (+ form 2: the next tile's requests, one per slot)
  //   group 2  PV_B(k-1)  | B(k-1): steps 15, 16 ; A(k): steps 0..6
  //   group 3  S_B(k)     | A(k): step 7 ; steps 8..14                      + this block's V^T fragments, one per slot
  //   group 4  PV_A(k)    | A(k): steps 15, 16 ; B(k): steps 0..6           + the next block's K fragments, one per slot
  // (FK_A4_EARLY moves the V^T reads to group 2 and the second block's K reads to group 3, each behind the MFMA that was the
  //  register's last reader.)
  auto groups123 = [&](const char* sb, int kt, int kb, int kt_req, auto mask_tag, auto first_tag) __attribute__((always_inline)) {
    constexpr bool MASK = decltype(mask_tag)::value, FIRST = decltype(first_tag)::value;
    {
      const float nmB = -mB;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        s_mfma(sA, kf[i], qfA[i], i == 0);
        e_step(sB, 7 + i, kb ^ 1, nmB, tB, lB, pB);
        if (A4_DMA == 2 && kb == 0) issue_piece(kt_req, i);
        if (A4_DMA == 4 && (i & 3) == 2) issue_piece(kt_req, 4 * kb + (i >> 2));        // form 4: two requests in groups 1 and 3 of each block
        FK_SLOT_FENCE();
      }
    }
    if constexpr (MASK || FIRST) s_ready(sA);
    if constexpr (MASK) mask_block(sA, kt, kb);
    if constexpr (FIRST) {
      if (kb == 0) mA = block_max(sA) + REF_BIAS;
    }
    {
      const float nmA = -mA, nmB = -mB;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        oB[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfr[i], pfrag(pB[i >> 2]), oB[i & 3], 0, 0, 0);
        if constexpr (A4_EARLY) vfr[i] = v_frag(sb, 2 * kb + (i >> 2), i & 3);
        if (i == 0) {
          e_step(sB, 15, kb ^ 1, nmB, tB, lB, pB);
          e_step(sB, 16, kb ^ 1, nmB, tB, lB, pB);
        } else {
          if (i == 1 && !(MASK || FIRST)) e_wait(sA);
          e_step(sA, i - 1, kb, nmA, tA, lA, pA);
        }
        FK_SLOT_FENCE();
      }
    }
    {
      const float nmA = -mA;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        s_mfma(sB, kf[i], qfB[i], i == 0);
        if constexpr (!A4_EARLY) vfr[i] = v_frag(sb, 2 * kb + (i >> 2), i & 3);
        else if (kb == 0) kf[i] = k_frag(sb, 1, i);
        if (A4_DMA == 3 && (i & 1)) issue_piece(kt_req, (i >> 1) + 4 * kb);      // form 3: four requests in each block's group 3
        if (A4_DMA == 4 && (i & 3) == 2) issue_piece(kt_req, 4 * kb + 2 + (i >> 2));
        e_step(sA, 7 + i, kb, nmA, tA, lA, pA);
        FK_SLOT_FENCE();
      }
    }
    if constexpr (MASK || FIRST) s_ready(sB);
    if constexpr (MASK) mask_block(sB, kt, kb);
    if constexpr (FIRST) {
      if (kb == 0) mB = block_max(sB) + REF_BIAS;
    }
  };
  // group 4; the S_B chain is >= 18 instructions away from its first reader here (fragment reads and their waits in between):
  // no e_wait (tools/a4_census.py and tests/test_kernel_resources.py check both distances in the generated code)
  auto group4 = [&](const char* nsb, int nkb, int kb, auto next_tag) __attribute__((always_inline)) {
    constexpr bool NEXT = decltype(next_tag)::value;
    const float nmA = -mA, nmB = -mB;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      oA[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfr[i], pfrag(pA[i >> 2]), oA[i & 3], 0, 0, 0);
      if constexpr (NEXT && !A4_EARLY) kf[i] = k_frag(nsb, nkb, i);
      if (i == 0) {
        e_step(sA, 15, kb, nmA, tA, lA, pA);
        e_step(sA, 16, kb, nmA, tA, lA, pA);
      } else {
        if (i == 1 && A4_EARLY) e_wait(sB);
        e_step(sB, i - 1, kb, nmB, tB, lB, pB);
      }
      FK_SLOT_FENCE();
    }
  };
#undef FK_SLOT_FENCE
  // A tile = its two blocks.  The first block's K fragments are read right behind the tile's barrier -- the one LDS round trip
  // per tile that nothing hides; the tile's LDS-DMA requests are issued under it (form 1) or spread over the first group
  // (form 2).  Moving the barrier to the middle of the tile, so that these fragments too arrive under the previous group, was
  // tried: the extra addressing (two stage bases per tile) costs more issue slots than the round trip (0.90-0.95 x, call I).
  auto do_tile = [&](int kt, auto mask_tag, auto first_tag) __attribute__((always_inline)) {
    if (A4_DMA >= 2 || kt + PF - 1 < kt1) wait_vmcnt<(PF - 1) * LOADS>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    const char* sb = smem + st_cur * STAGE_BYTES;
    if (A4_DMA == 0 && kt + PF < kt1) issue_tile(kt + PF, st_pf);
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) kf[kk] = k_frag(sb, 0, kk);
    if (A4_DMA == 1 && kt + PF < kt1) issue_tile(kt + PF, st_pf);
    groups123(sb, kt, 0, min(kt + PF, kt1 - 1), mask_tag, first_tag);
    group4(sb, 1, 0, std::true_type{});
    groups123(sb, kt, 1, min(kt + PF, kt1 - 1), mask_tag, std::false_type{});
    group4(sb, 0, 1, std::false_type{});
    release_tile();
  };
  // start of a pass: an empty "previous block" of stream B, stopped where group 1 picks a block up -- steps 0..6 done
  // (numerators zero), elements 7..15 still scores (-3e38: their exponentials are zero); V fragments zero: O_B += 0
  auto prime = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int r = 0; r < 16; ++r) sB[r] = r < 7 ? 0.f : -3.0e38f;
#pragma unroll
    for (int i = 0; i < 8; ++i) vfr[i] = bf16x8_t{0, 0, 0, 0, 0, 0, 0, 0};
    pB[0] = u32x4_t{0, 0, 0, 0};
    pB[1] = u32x4_t{0, 0, 0, 0};
    tB = 0.f;      // the empty block counts as the second block of a tile: its zero tile sum joins l_B in group 2
  };
  // end of a pass: what stream B still owes (steps 7..16 of its last block, that block's PV)
  auto drain = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int t = 7; t <= 16; ++t) e_step(sB, t, 1, -mB, tB, lB, pB);
#pragma unroll
    for (int i = 0; i < 8; ++i)
      oB[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfr[i], pfrag(pB[i >> 2]), oB[i & 3], 0, 0, 0);
  };

  using TT = std::true_type;
  using FF = std::false_type;
  const bool ragged = p.S % KVBLK != 0;
  const bool last_masked = ragged && kt1 == nkt;
  int* const wg_flag = (int*)(smem + STAGES * STAGE_BYTES);
  auto run_tiles = [&](auto first_tag) __attribute__((always_inline)) {
    const int last = kt1 - 1;
    prime();
    if (kt0 == last) {
      if (last_masked) do_tile(kt0, TT{}, first_tag);
      else do_tile(kt0, FF{}, first_tag);
    } else {
      do_tile(kt0, FF{}, first_tag);
      for (int kt = kt0 + 1; kt < last; ++kt) do_tile(kt, FF{}, FF{});
      if (last_masked) do_tile(last, TT{}, FF{});
      else do_tile(last, FF{}, FF{});
    }
    drain();
    if (A4_DMA >= 2) wait_vmcnt<0>();     // the two repeat requests behind the last tile: nothing may land after the pass
  };
  // plain S^T block (the restart's K-only pre-pass): query block X of the wave against key block kb
  auto scores_plain = [&](const char* sb, int kb, const bf16x8_t (&qf)[8]) __attribute__((always_inline)) {
    f32x16_t s;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk)
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k_frag(sb, kb, kk), qf[kk], kk == 0 ? f32x16_t{} : s, 0, 0, 0);
    return s;
  };
  bool overflow = false;
  for (int attempt = 0; attempt < 2; ++attempt) {
    if (attempt == 1) {
      fill();
      mA = -3.0e38f;
      mB = -3.0e38f;
      for (int kt = kt0; kt < kt1; ++kt) {
        const char* sb = acquire_tile(kt);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
          f32x16_t a = scores_plain(sb, kb, qfA), c = scores_plain(sb, kb, qfB);
          if (last_masked && kt == kt1 - 1) { mask_block(a, kt, kb); mask_block(c, kt, kb); }
          mA = fmaxf(mA, block_max(a));
          mB = fmaxf(mB, block_max(c));
        }
        release_tile();
      }
      __syncthreads();
    }
    fill();
    lA = 0.f;
    lB = 0.f;
    overflow = false;
#pragma unroll
    for (int df = 0; df < 4; ++df)
#pragma unroll
      for (int r = 0; r < 16; ++r) { oA[df][r] = 0.f; oB[df][r] = 0.f; }
    if (attempt == 0) run_tiles(TT{});
    else run_tiles(FF{});
    if (attempt == 0) {
      float mag = fabsf(lA) + fabsf(lB);
#pragma unroll
      for (int df = 0; df < 4; ++df)
#pragma unroll
        for (int r = 0; r < 16; ++r) mag += fabsf(oA[df][r]) + fabsf(oB[df][r]);
      overflow = __builtin_amdgcn_ballot_w64(!(mag <= 3.0e38f)) != 0;
      __syncthreads();
      if (tid == 0) *wg_flag = 0;
      __syncthreads();
      if (overflow && lane == 0) atomicOr(wg_flag, 1);
      __syncthreads();
      if (*wg_flag == 0) break;
    }
  }

  if constexpr (STREAMK) {
    if (kt0 > 0 || kt1 < nkt) {
      typedef __attribute__((address_space(1))) unsigned gu32;
      const int slot = kt1 < nkt ? pos + 1 : pos;
      gu32* const ctl = (gu32*)(p.sk_ctl + 2 * (size_t)slot);
      const __amdgpu_buffer_rsrc_t rs_p = __builtin_amdgcn_make_buffer_rsrc(
          (void*)(p.sk_partials + (size_t)slot * PART_FLOATS), 0, PART_FLOATS * 4, 0x00020000);
      __syncthreads();
      if (tid == 0) *(volatile unsigned*)smem = __hip_atomic_fetch_add(ctl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __syncthreads();
      const unsigned ticket = __builtin_amdgcn_readfirstlane(*(volatile unsigned*)smem);
      constexpr int LM_OFF = 32 * 256 * 16;      // (l, m_ref) pairs behind the 2 x 16 pieces of 256 threads
      if ((ticket & 1u) == 0) {
#pragma unroll
        for (int r = 0; r < 32; ++r) {
          const f32x16_t& a = r < 16 ? oA[(r & 15) >> 2] : oB[(r & 15) >> 2];
          const int q4 = r & 3;
          const u32x4_t v = {__float_as_uint(a[4 * q4]), __float_as_uint(a[4 * q4 + 1]), __float_as_uint(a[4 * q4 + 2]),
                             __float_as_uint(a[4 * q4 + 3])};
          __builtin_amdgcn_raw_buffer_store_b128(v, rs_p, tid * 16, r * (256 * 16), 16);
        }
        __builtin_amdgcn_raw_buffer_store_b64(u32x2_t{__float_as_uint(lA), __float_as_uint(mA)}, rs_p, LM_OFF + tid * 8, 0, 16);
        __builtin_amdgcn_raw_buffer_store_b64(u32x2_t{__float_as_uint(lB), __float_as_uint(mB)}, rs_p, LM_OFF + (256 + tid) * 8, 0, 16);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_store(ctl + 1, ticket + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        continue;
      }
      if (tid == 0) {
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
      auto merge = [&](f32x16_t (&o)[4], float& l_run, float& m_ref, int X) __attribute__((always_inline)) {
        const u32x2_t lm = __builtin_amdgcn_raw_buffer_load_b64(rs_p, LM_OFF + (X * 256 + tid) * 8, 0, 16);
        const float l_o = __uint_as_float(lm[0]), m_o = __uint_as_float(lm[1]);
        const float m_new = fmaxf(m_ref, m_o);
        const float w_s = m_ref == m_new ? 1.0f : __builtin_amdgcn_exp2f(m_ref - m_new);
        const float w_o = m_o == m_new ? 1.0f : __builtin_amdgcn_exp2f(m_o - m_new);
#pragma unroll
        for (int r0 = 0; r0 < 16; r0 += 4) {
          u32x4_t v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = __builtin_amdgcn_raw_buffer_load_b128(rs_p, tid * 16, (X * 16 + r0 + e) * (256 * 16), 16);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            f32x16_t& a = o[(r0 + e) >> 2];
            const int q4 = (r0 + e) & 3;
#pragma unroll
            for (int j = 0; j < 4; ++j)
              a[4 * q4 + j] = merge2(a[4 * q4 + j], w_s, __uint_as_float(v[e][j]), w_o);
          }
        }
        l_run = merge2(l_run, w_s, l_o, w_o);
        if (gave_up) l_run = __builtin_nanf("");
        m_ref = m_new;
      };
      merge(oA, lA, mA, 0);
      merge(oB, lB, mB, 1);
    }
  }

  // ---- finalize (per query block, as the 8-wave kernel does for its one) ----------------------------------------------------
  auto finalize = [&](const f32x16_t (&o)[4], float l_run, float m_ref, int q_row) __attribute__((always_inline)) {
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = 1.0f / l_tot;
    if (p.lse && hh == 0 && q_row < p.S) p.lse[(int64_t)bh * p.S + q_row] = m_ref + __builtin_amdgcn_logf(l_tot);
    bf16_t* const orow = p.o + (int64_t)b * p.o_bs + (int64_t)min(q_row, p.S - 1) * p.o_ld + h * HD;
    const bool wide = ((p.o_ld | p.o_bs) & 7) == 0 && ((uintptr_t)p.o & 15) == 0;
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
  };
  finalize(oA, lA, mA, q_row0 + ql);
  finalize(oB, lB, mB, q_row0 + 32 + ql);
  if constexpr (!STREAMK) break;
  else __syncthreads();
  }   // passes
}

template <bool STREAMK>
int launch4(const AttnParams& p, int grid, hipStream_t stream) {
  constexpr int SMEM = A4_STAGES * STAGE_BYTES + 16;
  auto kern = attention_fwd4_kernel<STREAMK>;
  FK_ENSURE_MAX_LDS(kern, SMEM, "fk_attention_fwd_bf16 (4 waves x 64 rows)");
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), SMEM, stream, p);
  FK_CHECK_LAUNCH("fk_attention_fwd_bf16 (4 waves x 64 rows)");
  return FK_OK;
}
}  // namespace

int fk_attention_fwd4_launch(const AttnParams& p, int grid, bool streamk, hipStream_t stream) {
  return streamk ? launch4<true>(p, grid, stream) : launch4<false>(p, grid, stream);
}
