// 3 x 3 convolution of the FLUX AutoencoderKL as an LDS halo-tiled MFMA kernel with the preceding GroupNorm(32) + SiLU
// applied in its operand prologue (K9a / K9b of SURVEY.md section 2.2: "NHWC implicit-GEMM MFMA conv, LDS halo tiles,
// fused GN-apply + SiLU prologue, fused residual add"; reference call sites flux_pipeline.py:609, 1127-1129 ->
// diffusers ResnetBlock2D / Upsample2D).
//
// The implicit-GEMM conv of gemm_bf16.hip gathers every input element nine times (once per filter tap) from global
// memory; a normalisation there would be evaluated nine times per element.  Here a workgroup owns a 16 x 16 tile of
// output pixels x 128 output channels and walks the input channels in chunks of 64:
//   * the 18 x 18 x 64 HALO tile of the chunk is read ONCE (16 B per lane), normalised --
//     y = bf16(silu(bf16((x - mean) * rstd * gamma + beta))), the arithmetic and rounding points of gn_apply_kernel,
//     zeros outside the image (the padding belongs to the normalised activation) -- and written to LDS, one 128-byte row
//     per pixel, 16-byte chunks swizzled by the pixel index;
//   * for each of the nine taps the [128 cout] x [64 ci] weight slice arrives by LDS-DMA (three-slot ring, requested
//     two taps ahead, counted vmcnt) and 16 MFMAs (v_mfma_f32_32x32x16_bf16, weight rows as the A operand like the GEMM
//     kernels, so a lane owns 4 consecutive output channels of one pixel) read the activation fragments from the halo
//     tile at the tap's pixel shift -- the shift is address arithmetic only;
//   * the next chunk's halo is fetched into registers at tap 0 and written to the other halo buffer at tap 1, under
//     the MFMAs of the remaining taps.
// Epilogue: bias (+ residual) -> bf16 through an LDS C tile -> 16-byte NHWC row stores.  Optional nearest-2x upsample of
// the input (Upsample2D) in the halo addressing.  Accumulation order: input-channel chunk outer, tap inner, fp32.
#include "fk_common.h"

namespace {

constexpr int TH = 16, TW = 16;                 // output pixels per workgroup
constexpr int BN = 128;                         // output channels per workgroup
constexpr int CK = 64;                          // input channels per chunk
constexpr int HW_ = TW + 2, HH_ = TH + 2;       // halo extent
constexpr int HALO_PIX = HW_ * HH_;             // 324
constexpr int HALO_BYTES = HALO_PIX * 128;      // 41 472
constexpr int W_SLOT = BN * 128;                // 16 KiB: 128 cout rows x 64 ci
constexpr int NSLOT = 3;
constexpr int CT_LD = BN + 8;
constexpr int SMEM_BYTES = 2 * HALO_BYTES + NSLOT * W_SLOT;   // 132 096 B (the C tile of the epilogue aliases the halo buffers)
static_assert(TH * TW * CT_LD * 2 <= 2 * HALO_BYTES, "C tile must fit the halo buffers");
constexpr int PIECES = HALO_PIX * 8;            // 16-byte pieces of one halo tile
constexpr int HITER = (PIECES + 511) / 512;     // 6

struct HaloArgs {
  const bf16_t* x;
  const bf16_t* w;
  const bf16_t* bias;
  const bf16_t* res;
  bf16_t* y;
  const float* stats;      // [B, 32, 2] (mean, rstd) of the INPUT, or null: no GroupNorm prologue
  const bf16_t* gamma;
  const bf16_t* beta;
  int B, H, W, Cin, Cout;  // H, W: output size; the input is (H >> up) x (W >> up)
  int up, silu, groups;
  int tiles_x, tiles_y, tiles_n;
  int ldw;                 // weight row stride in elements (>= 9 * Cin)
  int f32io;               // F32OUT only: bias and res are fp32 (fk_conv3x3_halo_f32out, the fp32-class encoder)
};

typedef __attribute__((address_space(3))) void lds_void;

FK_DEV void buffer_lds16(__amdgpu_buffer_rsrc_t rsrc, char* lds_dst, int voffset, int soffset) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)lds_dst, 16, voffset, soffset, 0, 0);
#else
  (void)rsrc; (void)lds_dst; (void)voffset; (void)soffset;
#endif
}
template <int N>
FK_DEV void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// workgroup barrier that leaves LDS-DMA requests in flight (a `__syncthreads()` would drain them: its fence waits vmcnt(0)
// while an LDS-DMA is pending); the counted vmcnt in front of it retires what the next step reads
FK_DEV void barrier_keep_dma() {
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
}

// F32OUT (parity build of the same kernel, fk_conv3x3_halo_f32_debug): y = fp32(acc + bias), written straight from the
// accumulator registers, so the main loop -- halo staging, GroupNorm prologue, tap order -- can be held to an fp32
// reference at rtol 1e-3 / atol 1e-4 (one bf16 rounding of the output alone is 2^-9 relative).
template <bool GN, bool RES, bool F32OUT = false>
__global__ __launch_bounds__(512, 2) void conv3x3_halo_kernel(const HaloArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 3, wn = wave >> 2;        // 4 (pixels) x 2 (channels) waves, 64 x 64 per wave

  // ---- tile: XCD-chunked order over (batch, tile row, tile column, channel block) ------------------------------------
  int t;
  {
    const int nwg = gridDim.x;
    const int q8 = nwg >> 3, r8 = nwg & 7;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    t = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
  }
  const int tn = t % p.tiles_n;
  t /= p.tiles_n;
  const int tx = t % p.tiles_x;
  t /= p.tiles_x;
  const int ty = t % p.tiles_y;
  const int b = t / p.tiles_y;
  const int y0 = ty * TH, x0 = tx * TW, n0 = tn * BN;
  const int Hin = p.H >> p.up, Win = p.W >> p.up;
  const int nchunks = p.Cin / CK;
  const int nsteps = 9 * nchunks;

  // ---- halo staging: piece i = tid + 512 it -> (halo pixel, 16-byte chunk) ---------------------------------------------
  const bf16_t* const xb = p.x + (int64_t)b * Hin * Win * p.Cin;
  int h_src[HITER];    // element offset of the piece inside the image at chunk 0, or -1: outside the image / beyond the tile
  int h_lds[HITER];    // byte offset inside a halo buffer
#pragma unroll
  for (int it = 0; it < HITER; ++it) {
    const int i = tid + 512 * it;
    const int hp = i >> 3, ch = i & 7;
    const int hy = hp / HW_, hx = hp - hy * HW_;
    const int oy = y0 + hy - 1, ox = x0 + hx - 1;
    const bool inb = i < PIECES && oy >= 0 && oy < p.H && ox >= 0 && ox < p.W;
    h_src[it] = inb ? ((oy >> p.up) * Win + (ox >> p.up)) * p.Cin + ch * 8 : -1;
    h_lds[it] = i < PIECES ? hp * 128 + ((ch ^ (hp & 7)) << 4) : -1;
  }
  u32x4_t hreg[HITER];
  u32x4_t g_gw = {0u, 0u, 0u, 0u}, g_bw = {0u, 0u, 0u, 0u};   // gamma / beta of this thread's 8 channels (the chunk index
  fk_f32x2_t g_st[4];                                          // tid & 7 is the same for all of its pieces) and (mean, rstd)
  const float* const st = p.stats + (int64_t)b * p.groups * 2;
  const int cpg = p.Cin / p.groups;
  // The fetch is issued in front of the step's weight request (sched_barrier below): whatever the exact number of loads,
  // `vmcnt(2)` at the end of a step then leaves at most the newest weight slice in flight.
  auto halo_fetch = [&](int c) {
#pragma unroll
    for (int it = 0; it < HITER; ++it) hreg[it] = *(const u32x4_t*)(xb + max(h_src[it], 0) + c * CK);
    if constexpr (GN) {
      const int c0 = c * CK + (tid & 7) * 8;
      g_gw = *(const u32x4_t*)(p.gamma + c0);
      g_bw = *(const u32x4_t*)(p.beta + c0);
#pragma unroll
      for (int e = 0; e < 4; ++e) g_st[e] = *(const fk_f32x2_t*)(st + 2 * ((c0 + 2 * e) / cpg));   // both halves of a dword share a group
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  auto halo_write = [&](char* buf) {
#pragma unroll
    for (int it = 0; it < HITER; ++it) {
      if (h_lds[it] < 0) continue;
      u32x4_t ow = h_src[it] >= 0 ? hreg[it] : u32x4_t{0u, 0u, 0u, 0u};
      if constexpr (GN) {
        if (h_src[it] >= 0) {   // (pixels outside the image stay zero: the padding of the NORMALISED activation)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float mean = g_st[e][0], rstd = g_st[e][1];
            float v0 = (bf_lo(ow[e]) - mean) * rstd * bf_lo(g_gw[e]) + bf_lo(g_bw[e]);
            float v1 = (bf_hi(ow[e]) - mean) * rstd * bf_hi(g_gw[e]) + bf_hi(g_bw[e]);
            if (p.silu) {
              v0 = silu_f(round_bf(v0));
              v1 = silu_f(round_bf(v1));
            }
            ow[e] = pack_bf2(v0, v1);
          }
        }
      }
      *(u32x4_t*)(buf + h_lds[it]) = ow;
    }
  };

  // ---- weight slices by LDS-DMA: piece = 8 rows x 128 B; wave w requests pieces 2w and 2w + 1 of the 128-row slice ---
  const int lrow = lane >> 3, slot8 = lane & 7;
  const __amdgpu_buffer_rsrc_t rs_w =
      __builtin_amdgcn_make_buffer_rsrc((void*)(p.w + (int64_t)n0 * p.ldw), 0, 0x7fffffff, 0x00020000);
  int w_voff[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int rl = (wave * 2 + j) * 8 + lrow;
    w_voff[j] = min(rl, p.Cout - 1 - n0) * p.ldw * 2 + ((slot8 ^ ((rl >> 1) & 7)) << 4);
  }
  char* const wbase = smem + 2 * HALO_BYTES;
  auto dma_w = [&](int s) {   // weight slice of step s = (chunk, tap): columns tap * Cin + chunk * 64
    const int c = s / 9, tap = s - 9 * c;
    char* dst = wbase + (s % NSLOT) * W_SLOT + wave * 2048;
    const int koff = (tap * p.Cin + c * CK) * 2;
    buffer_lds16(rs_w, dst, w_voff[0], koff);
    buffer_lds16(rs_w, dst + 1024, w_voff[1], koff);
  };

  // ---- MFMA operand addressing --------------------------------------------------------------------------------------------
  const int frow = lane & 31, fhalf = lane >> 5;
  int a_hp[2];           // halo pixel of this lane's row at tap (0, 0) for the wave's two 32-pixel blocks
#pragma unroll
  for (int mf = 0; mf < 2; ++mf) {
    const int pix = wm * 64 + mf * 32 + frow;
    a_hp[mf] = (pix >> 4) * HW_ + (pix & 15);
  }
  int w_rd[2];
#pragma unroll
  for (int nf = 0; nf < 2; ++nf) w_rd[nf] = (wn * 64 + nf * 32 + frow) * 128;
  const int w_sw = (frow >> 1) & 7;   // rows 32 nf + frow: (row >> 1) & 7 depends on frow only

  f32x16_t acc[2][2];    // [nf][mf]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // ---- prologue ---------------------------------------------------------------------------------------------------------
  halo_fetch(0);
  dma_w(0);
  if (nsteps > 1) dma_w(1);
  halo_write(smem);
  if (nsteps > 1) wait_vmcnt<2>();
  else wait_vmcnt<0>();
  barrier_keep_dma();

  for (int s = 0; s < nsteps; ++s) {
    const int c = s / 9, tap = s - 9 * c;
    const int dy = tap / 3, dx = tap - 3 * dy;
    const bool more_w = s + 2 < nsteps;
    const bool stage = tap == 0 && c + 1 < nchunks;
    if (stage) halo_fetch(c + 1);
    if (more_w) dma_w(s + 2);          // its slot was last read in step s - 1: every wave is past that step's barrier
    // ---- 16 MFMAs: (2 weight blocks) x (2 pixel blocks) x (4 k-steps of 16) --------------------------------------------
    const char* hb = smem + (c & 1) * HALO_BYTES;
    const char* wb = wbase + (s % NSLOT) * W_SLOT;
    bf16x8_t af[2][4], wf[2][4];
#pragma unroll
    for (int mf = 0; mf < 2; ++mf) {
      const int hp = a_hp[mf] + dy * HW_ + dx;
      const int sw = hp & 7;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) af[mf][kk] = *(const bf16x8_t*)(hb + hp * 128 + (((2 * kk + fhalf) ^ sw) << 4));
    }
#pragma unroll
    for (int nf = 0; nf < 2; ++nf)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) wf[nf][kk] = *(const bf16x8_t*)(wb + w_rd[nf] + (((2 * kk + fhalf) ^ w_sw) << 4));
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int nf = 0; nf < 2; ++nf)
#pragma unroll
        for (int mf = 0; mf < 2; ++mf)
          acc[nf][mf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[nf][kk], af[mf][kk], acc[nf][mf], 0, 0, 0);
    // the next chunk's halo goes to the OTHER buffer (last read during chunk c - 1) under the remaining taps
    if (tap == 1 && c + 1 < nchunks) halo_write(smem + ((c + 1) & 1) * HALO_BYTES);
    // ---- publish step s + 1's weight slice: of everything requested, at most W(s + 2) (two requests) may stay in flight -----
    if (more_w) wait_vmcnt<2>();
    else wait_vmcnt<0>();
    barrier_keep_dma();
  }

  if constexpr (F32OUT) {
#pragma unroll
    for (int nf = 0; nf < 2; ++nf)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = n0 + wn * 64 + nf * 32 + 8 * q + 4 * fhalf;
        const int nb = min(n, p.Cout - 4);
        f32x4_t bv;
        if (p.f32io) {
          bv = *(const f32x4_t*)((const float*)p.bias + nb);
        } else {
          const u32x2_t bw = *(const u32x2_t*)(p.bias + nb);
          bv = f32x4_t{bf_lo(bw[0]), bf_hi(bw[0]), bf_lo(bw[1]), bf_hi(bw[1])};
        }
#pragma unroll
        for (int mf = 0; mf < 2; ++mf) {
          const int pix = wm * 64 + mf * 32 + frow;
          const int oy = y0 + (pix >> 4), ox = x0 + (pix & 15);
          if (oy < p.H && ox < p.W && n < p.Cout) {
            const int64_t off = (((int64_t)b * p.H + oy) * p.W + ox) * p.Cout + n;
            f32x4_t o = f32x4_t{acc[nf][mf][4 * q + 0], acc[nf][mf][4 * q + 1], acc[nf][mf][4 * q + 2], acc[nf][mf][4 * q + 3]} + bv;
            if (p.f32io && p.res) o += *(const f32x4_t*)((const float*)p.res + off);
            *(f32x4_t*)((float*)p.y + off) = o;
          }
        }
      }
    return;
  }
  // ---- epilogue: bias -> bf16 -> LDS C tile -> (+ residual) -> 16-byte NHWC rows ---------------------------------------------
  bf16_t* ct = (bf16_t*)smem;   // aliases the halo buffers: every wave is past the last step's barrier
#pragma unroll
  for (int nf = 0; nf < 2; ++nf)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int nl = wn * 64 + nf * 32 + 8 * q + 4 * fhalf;
      const u32x2_t bw = *(const u32x2_t*)(p.bias + min(n0 + nl, p.Cout - 4));
      const float bv[4] = {bf_lo(bw[0]), bf_hi(bw[0]), bf_lo(bw[1]), bf_hi(bw[1])};
#pragma unroll
      for (int mf = 0; mf < 2; ++mf) {
        u32x2_t pk;
        pk[0] = pack_bf2(acc[nf][mf][4 * q + 0] + bv[0], acc[nf][mf][4 * q + 1] + bv[1]);
        pk[1] = pack_bf2(acc[nf][mf][4 * q + 2] + bv[2], acc[nf][mf][4 * q + 3] + bv[3]);
        const int pix = wm * 64 + mf * 32 + frow;
        *(u32x2_t*)(ct + pix * CT_LD + nl) = pk;
      }
    }
  __syncthreads();
  constexpr int CPR = BN / 8;                     // 16-byte chunks per pixel row of the tile
  constexpr int ITERS = TH * TW * CPR / 512;      // 8
  const int cc = tid % CPR, pix0 = tid / CPR;     // 32 pixels per iteration
#pragma unroll
  for (int it = 0; it < ITERS; ++it) {
    const int pix = pix0 + 32 * it;
    const int oy = y0 + (pix >> 4), ox = x0 + (pix & 15);
    const int n = n0 + cc * 8;
    if (oy < p.H && ox < p.W && n < p.Cout) {
      u32x4_t o = *(const u32x4_t*)(ct + pix * CT_LD + cc * 8);
      const int64_t off = (((int64_t)b * p.H + oy) * p.W + ox) * p.Cout + n;
      if constexpr (RES) {
        const u32x4_t rv = *(const u32x4_t*)(p.res + off);
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = pack_bf2(bf_lo(rv[e]) + bf_lo(o[e]), bf_hi(rv[e]) + bf_hi(o[e]));
      }
      *(u32x4_t*)(p.y + off) = o;
    }
  }
}

template <bool GN, bool RES, bool F32OUT = false>
int launch_halo(const HaloArgs& p, hipStream_t stream) {
  auto kern = conv3x3_halo_kernel<GN, RES, F32OUT>;
  FK_ENSURE_MAX_LDS(kern, SMEM_BYTES, "fk_conv3x3_halo_bf16");
  const int grid = p.B * p.tiles_y * p.tiles_x * p.tiles_n;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), SMEM_BYTES, stream, p);
  FK_CHECK_LAUNCH("fk_conv3x3_halo_bf16");
  return FK_OK;
}

}  // namespace

static int conv3x3_halo_entry(const fk_conv_args* args, const float* gn_stats, const void* gn_gamma, const void* gn_beta,
                              int32_t gn_groups, int32_t gn_silu, bool f32out, fk_stream_t stream_, bool f32io = false) {
  FK_CHECK_ARG(args != nullptr, "fk_conv3x3_halo_bf16: null args");
  const fk_conv_args& a = *args;
  FK_CHECK_ARG(a.x && a.w && a.y && a.bias, "fk_conv3x3_halo_bf16: null x / w / bias / y");
  FK_CHECK_ARG(a.ksize == 3 && a.stride == 1 && a.pad == 1, "fk_conv3x3_halo_bf16: 3 x 3, stride 1, padding 1 only");
  FK_CHECK_ARG(a.Cin % 64 == 0 && a.Cout % 8 == 0 && a.B > 0 && a.Hin > 0 && a.Win > 0,
               "fk_conv3x3_halo_bf16: Cin must be a multiple of 64, Cout of 8");
  const int up = a.upsample2x ? 1 : 0;
  FK_CHECK_ARG(a.Hout == (a.Hin << up) && a.Wout == (a.Win << up), "fk_conv3x3_halo_bf16: output size does not match the input");
  FK_CHECK_ARG((int64_t)a.Hin * a.Win * a.Cin < (1ll << 31), "fk_conv3x3_halo_bf16: one image must stay below 2^31 elements");
  FK_CHECK_ARG(((uintptr_t)a.x % 16 == 0) && ((uintptr_t)a.w % 16 == 0) && ((uintptr_t)a.y % 16 == 0) &&
                   (!a.res || (uintptr_t)a.res % 16 == 0) && ((uintptr_t)a.bias % (f32io ? 16 : 8) == 0),
               "fk_conv3x3_halo_bf16: alignment");
  const bool gn = gn_stats != nullptr;
  if (gn) {
    FK_CHECK_ARG(gn_gamma && gn_beta && gn_groups > 0 && a.Cin % gn_groups == 0 && (a.Cin / gn_groups) % 2 == 0 &&
                     ((uintptr_t)gn_gamma % 16 == 0) && ((uintptr_t)gn_beta % 16 == 0),
                 "fk_conv3x3_halo_bf16: GroupNorm prologue needs gamma / beta and an even number of channels per group");
  }
  HaloArgs p;
  p.x = (const bf16_t*)a.x; p.w = (const bf16_t*)a.w; p.bias = (const bf16_t*)a.bias; p.res = (const bf16_t*)a.res;
  p.y = (bf16_t*)a.y;
  p.stats = gn_stats; p.gamma = (const bf16_t*)gn_gamma; p.beta = (const bf16_t*)gn_beta;
  p.B = a.B; p.H = a.Hout; p.W = a.Wout; p.Cin = a.Cin; p.Cout = a.Cout;
  p.up = up; p.silu = gn_silu; p.groups = gn ? gn_groups : 1;
  p.tiles_x = (a.Wout + TW - 1) / TW; p.tiles_y = (a.Hout + TH - 1) / TH; p.tiles_n = (a.Cout + BN - 1) / BN;
  p.ldw = (9 * a.Cin + 63) / 64 * 64;
  p.f32io = f32io ? 1 : 0;
  FK_CHECK_ARG((int64_t)BN * p.ldw * 2 < (1ll << 31), "fk_conv3x3_halo_bf16: weight slice too large");
  hipStream_t stream = (hipStream_t)stream_;
  if (f32out) {
    FK_CHECK_ARG(f32io || !a.res, "fk_conv3x3_halo_f32_debug: no residual in the parity build");
    FK_CHECK_ARG(!f32io || (!gn && !up && a.Cout % 4 == 0), "fk_conv3x3_halo_f32out: no GroupNorm prologue, no upsample");
    return gn ? launch_halo<true, false, true>(p, stream) : launch_halo<false, false, true>(p, stream);
  }
  if (gn) return a.res ? launch_halo<true, true>(p, stream) : launch_halo<true, false>(p, stream);
  return a.res ? launch_halo<false, true>(p, stream) : launch_halo<false, false>(p, stream);
}

extern "C" int fk_conv3x3_halo_bf16(const fk_conv_args* args, const float* gn_stats, const void* gn_gamma,
                                    const void* gn_beta, int32_t gn_groups, int32_t gn_silu, fk_stream_t stream_) {
  return conv3x3_halo_entry(args, gn_stats, gn_gamma, gn_beta, gn_groups, gn_silu, false, stream_);
}

extern "C" int fk_conv3x3_halo_f32_debug(const fk_conv_args* args, const float* gn_stats, const void* gn_gamma,
                                         const void* gn_beta, int32_t gn_groups, int32_t gn_silu, fk_stream_t stream_) {
  return conv3x3_halo_entry(args, gn_stats, gn_gamma, gn_beta, gn_groups, gn_silu, true, stream_);
}

// The halo kernel over the operand parts of the fp32-class encoder (vae_kernels.hip): args->x / w hold the bf16 parts side by
// side along the channel axis (Cin = parts * C, a multiple of 64), args->bias / res / y are fp32.
extern "C" int fk_conv3x3_halo_f32out(const fk_conv_args* args, fk_stream_t stream_) {
  return conv3x3_halo_entry(args, nullptr, nullptr, nullptr, 0, 0, true, stream_, true);
}
