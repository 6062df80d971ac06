// NOT BUILT -- record of a round-1 experiment (README.md in this directory).
// `gemm6_kernel`: gemm5_kernel's shape (256 x 256, 4 waves, one per SIMD, accumulators in AGPRs, slot-pinned k-loop)
// with gemm2_kernel's buffer-form LDS-DMA staging instead of registers + ds_write: K-tiles of 32 in a 4-stage ring.
// To build it, paste both blocks into ../gemm2_bf16.hip in front of `launch` / `launch_bn` and dispatch `bn == 258`
// to `launch6`.  It passed every GEMM parity test on the first run and measured 3-8 % below gemm5_kernel:
//   8192 x 12288 x 3072: 1129 vs 1163 TF/s, 32768 x 3072 x 12288: 1209 vs 1307, 2560 x 12288 x 3072: 1149 vs 1177.

// ---- 4 waves, LDS-DMA staged (experimental: FK_GEMM6=1) ---------------------------------------------------------
// The vendor library's best bf16 kernel for these shapes (MT256x256x64, 4 waves, one per SIMD) stages BOTH operands
// with `buffer_load_dwordx4 ... lds` and still keeps the matrix pipe ~90 % busy, so the 54-68 cycles of issue time
// measured for `global_load_lds` (the FLAT form) cannot be the cost of the MUBUF form.  This kernel is gemm5_kernel's
// shape and slot discipline with gemm2_kernel's buffer-form DMA: no staging registers, no ds_write, K-tiles of 32 in
// a 4-stage ring (3 tiles = 3 x 1024 MFMA cycles of lookahead; 4 x 32 KB), one barrier per K-tile.
template <int BN>
struct Cfg6 {
  static_assert(BN == 256, "the LDS-DMA 4-wave kernel is instantiated for the 256 x 256 tile only");
  static constexpr int NTHREADS = 256;
  static constexpr int BK = 32, STAGES = 4, KS = 2, CH = 4, ROW_BYTES = 64, RPI = 16;
  static constexpr int WAVES_M = 2, WAVES_N = 2;
  static constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
  static constexpr int MF = WTM / 32, NF = WTN / 32;
  static constexpr int A_BYTES = BM * ROW_BYTES, W_BYTES = BN * ROW_BYTES;
  static constexpr int STAGE_BYTES = A_BYTES + W_BYTES;
  static constexpr int PF = STAGES - 1;
  static constexpr int A_LOADS = A_BYTES / 1024 / 4, W_LOADS = W_BYTES / 1024 / 4;  // DMA pieces per wave per K-tile
  static constexpr int LOADS = A_LOADS + W_LOADS;
  static constexpr int PPK = LOADS / KS;                // pieces per k-step
  static constexpr int FRAG_STRIDE = 32 * ROW_BYTES;
  static constexpr int CT_LD = BN + 8;
  static constexpr int CT_BYTES = BM * CT_LD * 2;
  static constexpr int SMEM_BYTES = (STAGES * STAGE_BYTES > CT_BYTES) ? STAGES * STAGE_BYTES : CT_BYTES;
  static constexpr int BSLOT = 3;
  static_assert(LOADS % KS == 0 && NF * MF >= BSLOT + 1 + NF + MF, "slot schedule");
  static FK_DEV int swz(int row) { return (row >> 2) & 3; }
};

template <int EPI, int BN>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm6_kernel(const GroupArgs ga) {
  using C = Cfg6<BN>;
  constexpr int BK = C::BK;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave % C::WAVES_M, wn = wave / C::WAVES_M;
  int pi, m0, n0;
  select_tile<BN>(ga, pi, m0, n0);
  const fk_gemm_args& p = ga.p[pi];
  const int nk = p.K / BK;

  // DMA sources, as in gemm2_kernel: lane -> (row = base + lane / CH, slot = lane % CH), source chunk = slot ^ swz(row)
  const int lrow = lane / C::CH, slot = lane % C::CH;
  const __amdgpu_buffer_rsrc_t rs_a =
      __builtin_amdgcn_make_buffer_rsrc((void*)((const bf16_t*)p.A + fk_row_offset(p.a, m0)), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w =
      __builtin_amdgcn_make_buffer_rsrc((void*)((const bf16_t*)p.W + (int64_t)n0 * p.ldw), 0, 0x7fffffff, 0x00020000);
  int voff[C::LOADS];
  const TileRows arow(p.a, m0);
  const int ldw2 = (int)p.ldw * 2;
#pragma unroll
  for (int j = 0; j < C::LOADS; ++j) {
    const bool isA = j < C::A_LOADS;
    const int rl = (wave * (isA ? C::A_LOADS : C::W_LOADS) + (isA ? j : j - C::A_LOADS)) * C::RPI + lrow;
    const int sw = (slot ^ C::swz(rl)) << 4;
    voff[j] = (isA ? arow.off(min(rl, p.M - 1 - m0)) * 2 : min(rl, p.N - 1 - n0) * ldw2) + sw;
  }
  auto issue_piece = [&](int i, int koff, char* sb) {
    if (i < C::A_LOADS) buffer_lds16(rs_a, sb + (wave * C::A_LOADS + i) * 1024, voff[i], koff);
    else buffer_lds16(rs_w, sb + C::A_BYTES + (wave * C::W_LOADS + (i - C::A_LOADS)) * 1024, voff[i], koff);
  };

  const int frow = lane & 31, fhalf = lane >> 5, fsw = C::swz(frow);
  const int a_rd = (wm * C::WTM + frow) * C::ROW_BYTES;
  const int w_rd = C::A_BYTES + (wn * C::WTN + frow) * C::ROW_BYTES;

  f32x16_t acc[C::NF][C::MF];
#pragma unroll
  for (int i = 0; i < C::NF; ++i)
#pragma unroll
    for (int j = 0; j < C::MF; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  bf16x8_t af[2][C::MF], wf[2][C::NF];
  auto read_frag = [&](int buf, const char* sb, int kk, int f) {
    const int coff = (((kk * 2 + fhalf) ^ fsw) << 4);
    if (f < C::MF) af[buf][f] = *(const bf16x8_t*)(sb + a_rd + f * C::FRAG_STRIDE + coff);
    else wf[buf][f - C::MF] = *(const bf16x8_t*)(sb + w_rd + (f - C::MF) * C::FRAG_STRIDE + coff);
  };

  // fill: tiles 0 .. PF-1 requested (indices clamped: surplus requests are never multiplied), tile 0 landed
#pragma unroll
  for (int s = 0; s < C::PF; ++s) {
    const int koff = min(s, nk - 1) * (BK * 2);
#pragma unroll
    for (int i = 0; i < C::LOADS; ++i) issue_piece(i, koff, smem + s * C::STAGE_BYTES);
  }
  wait_vmcnt<(C::PF - 1) * C::LOADS>();
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int f = 0; f < C::MF + C::NF; ++f) read_frag(0, smem, 0, f);

  // One k-step = NM MFMA slots pinned in source order: slots R0+1 .. R0+NFR read one fragment of the next k-step each,
  // the last PPK odd slots request one DMA piece of tile j+PF each.  The last k-step of a K-tile carries the tile's
  // barrier in front of slot BSLOT: tile j+1 has landed (own pieces: only the PF-1 newer tiles' requests issued so far
  // may still be outstanding) and every wave's reads of tile j are complete.
  constexpr int NM = C::NF * C::MF, NFR = C::MF + C::NF, D0 = NM - 2 * C::PPK + 1;
  auto kstep = [&](int cb, const char* sb_rd, int kk_rd, int p0, char* sb_pf, int koff, int r0, bool barrier) {
    const int nb = cb ^ 1;
#pragma unroll
    for (int i = 0; i < NM; ++i) {
      const int nf = i / C::MF, mf = i % C::MF;
      if (barrier && i == r0) {
        // newer than tile j+1: all of tile j+2 and the part of tile j+PF requested in this K-tile's earlier k-steps
        wait_vmcnt<(C::PF - 2) * C::LOADS + C::PPK * (C::KS - 1)>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
      }
      acc[nf][mf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[cb][nf], af[cb][mf], acc[nf][mf], 0, 0, 0);
      if (i >= r0 + 1 && i - r0 - 1 < NFR) read_frag(nb, sb_rd, kk_rd, i - r0 - 1);
      if (i >= D0 && ((i - D0) & 1) == 0) issue_piece(p0 + (i - D0) / 2, koff, sb_pf);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  for (int j = 0; j < nk; ++j) {
    const char* sb = smem + (j & 3) * C::STAGE_BYTES;
    const char* sb_nx = smem + ((j + 1) & 3) * C::STAGE_BYTES;
    char* sb_pf = smem + ((j + C::PF) & 3) * C::STAGE_BYTES;      // = the stage of tile j-1: its reads ended at the last barrier
    const int koff_pf = min(j + C::PF, nk - 1) * (BK * 2);
    kstep(0, sb, 1, 0, sb_pf, koff_pf, 0, false);
    __builtin_amdgcn_sched_barrier(0);
    kstep(1, sb_nx, 0, C::PPK, sb_pf, koff_pf, C::BSLOT, true);
    __builtin_amdgcn_sched_barrier(0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // surplus (clamped) requests must not land in the C tile
  store_tile<EPI, BN, C>(acc, p, smem, m0, n0, wm, wn);
}



template <int EPI, int BN>
int launch6(GroupArgs& ga, const fk_gemm_args* probs, int n, hipStream_t stream) {
  int total = 0;
  for (int i = 0; i < FK_MAX_GROUP; ++i) {
    ga.tiles_before[i] = total;
    if (i < n) total += ((probs[i].M + BM - 1) / BM) * ((probs[i].N + BN - 1) / BN);
  }
  ga.tiles_before[FK_MAX_GROUP] = total;
  auto kern = gemm6_kernel<EPI, BN>;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg6<BN>::SMEM_BYTES);
    attr_done = true;
  }
  hipLaunchKernelGGL(kern, dim3(total), dim3(256), Cfg6<BN>::SMEM_BYTES, stream, ga);
  FK_CHECK_LAUNCH("fk_gemm_bf16 (256 x 256 tile, 4 waves, LDS-DMA staged)");
  return FK_OK;
}

