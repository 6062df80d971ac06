// HBM-bound kernels of the FLUX AutoencoderKL (K9b and the layout changes at its boundary).
// Activations are NHWC bf16 ([B, HW, C], C % 32 == 0).  GroupNorm(32 groups, eps 1e-6) is two passes:
//   stats : per-block partial (sum, sum of squares) per group -> deterministic fixed-order merge
//   apply : y = act(bf16((x - mean) * rstd * gamma + beta)), vectorised 16 bytes per lane
// (the convolutions themselves are the implicit-GEMM path in gemm_bf16.hip).
#include "fk_common.h"

namespace {

constexpr int GN_THREADS = 256;
constexpr int GN_MAX_BLOCKS = 512;  // partial-sum blocks per image

inline int gn_blocks(int64_t HW, int C) {
  const int64_t pix_per_iter = GN_THREADS / (C / 8);
  int64_t nb = (HW + pix_per_iter * 8 - 1) / (pix_per_iter * 8);  // >= 8 iterations per block
  if (nb < 1) nb = 1;
  if (nb > GN_MAX_BLOCKS) nb = GN_MAX_BLOCKS;
  return (int)nb;
}

// grid (nblk, B).  Thread t owns channel chunk (t % (C/8)) of pixels (t / (C/8)) + k * ppi.
// T = bf16_t, or float for the fp32-class encoder (fk_groupnorm_f32_nhwc): same sums, same fixed merge order.
template <typename T>
__global__ __launch_bounds__(GN_THREADS) void gn_partial_kernel(const T* x, float* ws, int64_t HW, int C,
                                                                int groups) {
  __shared__ float hs[2 * GN_THREADS], hq[2 * GN_THREADS];
  const int tid = threadIdx.x;
  const int cpr = C / 8;                 // 16-byte chunks per pixel
  const int ppi = GN_THREADS / cpr;      // pixels per iteration
  const int chunk = tid % cpr, prow = tid / cpr;
  const int b = blockIdx.y, nblk = gridDim.x;
  const int64_t per_blk = (HW + nblk - 1) / nblk;
  const int64_t p0 = (int64_t)blockIdx.x * per_blk;
  const int64_t p1 = (p0 + per_blk < HW) ? p0 + per_blk : HW;
  float s_lo = 0.f, q_lo = 0.f, s_hi = 0.f, q_hi = 0.f;  // channels [0,4) and [4,8) of the chunk
  if (prow < ppi) {
    const T* xb = x + (int64_t)b * HW * C + chunk * 8;
    for (int64_t p = p0 + prow; p < p1; p += ppi) {
      float v0, v1, v2, v3, v4, v5, v6, v7;
      if constexpr (sizeof(T) == 4) {
        const f32x4_t a = *(const f32x4_t*)(xb + p * C), c = *(const f32x4_t*)(xb + p * C + 4);
        v0 = a[0]; v1 = a[1]; v2 = a[2]; v3 = a[3]; v4 = c[0]; v5 = c[1]; v6 = c[2]; v7 = c[3];
      } else {
        const u32x4_t w = *(const u32x4_t*)(xb + p * C);
        v0 = bf_lo(w[0]); v1 = bf_hi(w[0]); v2 = bf_lo(w[1]); v3 = bf_hi(w[1]);
        v4 = bf_lo(w[2]); v5 = bf_hi(w[2]); v6 = bf_lo(w[3]); v7 = bf_hi(w[3]);
      }
      s_lo += (v0 + v1) + (v2 + v3);
      q_lo += (v0 * v0 + v1 * v1) + (v2 * v2 + v3 * v3);
      s_hi += (v4 + v5) + (v6 + v7);
      q_hi += (v4 * v4 + v5 * v5) + (v6 * v6 + v7 * v7);
    }
  }
  // In-block merge in a FIXED order (no float atomics: their order, hence the rounding, would vary run to run):
  // every thread publishes its two half-chunk sums; thread g then adds the 16 entries of group g -- pixel rows
  // outer, half-chunks inner -- one after the other.
  hs[tid * 2] = s_lo;
  hq[tid * 2] = q_lo;
  hs[tid * 2 + 1] = s_hi;
  hq[tid * 2 + 1] = q_hi;
  __syncthreads();
  if (tid < groups) {
    const int cpg = C / groups;   // channels per group: 4, 8 or 16 (C = 128, 256, 512)
    const int hpg = cpg / 4;      // half-chunks (4 channels) per group
    float s = 0.f, q = 0.f;
    for (int pr = 0; pr < ppi; ++pr)
      for (int h = 0; h < hpg; ++h) {
        const int hc = tid * hpg + h;                    // half-chunk index inside a pixel
        const int e = (pr * cpr + (hc >> 1)) * 2 + (hc & 1);
        s += hs[e];
        q += hq[e];
      }
    float* o = ws + (((int64_t)b * nblk + blockIdx.x) * groups + tid) * 2;
    o[0] = s;
    o[1] = q;
  }
}

// grid (groups, B), 64 threads: merge the per-block partials of one (batch, group) in a fixed order
// (lane i sums blocks i, i+64, ...; then a fixed-shape LDS tree) in fp64 to avoid cancellation.
__global__ void gn_finalize_kernel(const float* ws, float* stats, int nblk, int groups, double count, float eps) {
  __shared__ double sh[2][64];
  const int g = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
  double s = 0.0, q = 0.0;
  for (int i = lane; i < nblk; i += 64) {
    const float* o = ws + (((int64_t)b * nblk + i) * groups + g) * 2;
    s += (double)o[0];
    q += (double)o[1];
  }
  sh[0][lane] = s;
  sh[1][lane] = q;
  __syncthreads();
  for (int off = 32; off >= 1; off >>= 1) {
    if (lane < off) {
      sh[0][lane] += sh[0][lane + off];
      sh[1][lane] += sh[1][lane + off];
    }
    __syncthreads();
  }
  if (lane == 0) {
    const double mean = sh[0][0] / count;
    double var = sh[1][0] / count - mean * mean;
    if (var < 0.0) var = 0.0;
    stats[((int64_t)b * groups + g) * 2 + 0] = (float)mean;
    stats[((int64_t)b * groups + g) * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
  }
}

__global__ __launch_bounds__(256) void gn_apply_kernel(const bf16_t* x, bf16_t* y, const float* stats,
                                                       const bf16_t* gamma, const bf16_t* beta, int64_t HW,
                                                       int C, int groups, int silu) {
  const int b = blockIdx.y;
  const int cpr = C / 8, cpg = C / groups;
  const int64_t nvec = HW * cpr;
  const bf16_t* xb = x + (int64_t)b * HW * C;
  bf16_t* yb = y + (int64_t)b * HW * C;
  const float* st = stats + (int64_t)b * groups * 2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
    const int c0 = (int)(i % cpr) * 8;
    const u32x4_t w = *(const u32x4_t*)(xb + i * 8);
    const u32x4_t gw = *(const u32x4_t*)(gamma + c0);
    const u32x4_t bw = *(const u32x4_t*)(beta + c0);
    u32x4_t ow;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int g = (c0 + 2 * e) / cpg;  // both halves of a dword share a group (cpg >= 2)
      const float mean = st[2 * g], rstd = st[2 * g + 1];
      float v0 = (bf_lo(w[e]) - mean) * rstd * bf_lo(gw[e]) + bf_lo(bw[e]);
      float v1 = (bf_hi(w[e]) - mean) * rstd * bf_hi(gw[e]) + bf_hi(bw[e]);
      if (silu) {
        v0 = silu_f(round_bf(v0));
        v1 = silu_f(round_bf(v1));
      }
      ow[e] = pack_bf2(v0, v1);
    }
    *(u32x4_t*)(yb + i * 8) = ow;
  }
}

// NCHW (fp32 / bf16) -> NHWC bf16 with channel padding; tiled over (32 pixels x all channels).
template <typename T>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const T* src, bf16_t* dst, int C, int Cpad,
                                                           int64_t HW, float div, float add) {
  const int b = blockIdx.y;
  const int64_t total = HW * Cpad;
  const T* sb = src + (int64_t)b * C * HW;
  bf16_t* db = dst + (int64_t)b * HW * Cpad;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % Cpad);
    const int64_t p = i / Cpad;
    float v = 0.f;
    if (c < C) {
      if constexpr (sizeof(T) == 4) v = sb[(int64_t)c * HW + p];
      else v = bf2f(sb[(int64_t)c * HW + p]);
      v = round_bf(round_bf(v) / div) + add;  // bf16 tensor / scalar, then + scalar (torch rounding points)
    }
    db[i] = f2bf(v);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const bf16_t* src, T* dst, int C, int Cpad,
                                                           int64_t HW, float add, float mul) {
  const int b = blockIdx.y;
  const int64_t total = HW * C;
  const bf16_t* sb = src + (int64_t)b * HW * Cpad;
  T* db = dst + (int64_t)b * C * HW;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i / HW);
    const int64_t p = i - (int64_t)c * HW;
    const float v = round_bf(round_bf(bf2f(sb[p * Cpad + c]) + add) * mul);
    if constexpr (sizeof(T) == 4) db[i] = v;
    else db[i] = f2bf(v);
  }
}

// uint8 NHWC pixels (PIL layout) -> NHWC bf16, C zero padded: nearest resize + [-1, 1] normalisation in one gather.
//   v = ((u / 255) - 0.5) / 0.5 in fp32 (reference cli.py:106-109), optionally 2v - 1 once more
//   (VaeImageProcessor.preprocess normalises whenever the tensor has no negative value), then bf16
//   (`image.to(dtype)` in prepare_latents).  Source row/col = min(floor(dst * (float)in / out), in - 1):
//   torch's `F.interpolate(mode="nearest")`, which is what VaeImageProcessor.resize runs on tensors.
__global__ __launch_bounds__(256) void pixels_u8_to_nhwc_kernel(const uint8_t* src, bf16_t* dst, int Hin, int Win,
                                                                int Hout, int Wout, int Cpad, int renorm) {
  const int b = blockIdx.y;
  const int64_t total = (int64_t)Hout * Wout * Cpad;
  const uint8_t* sb = src + (int64_t)b * Hin * Win * 3;
  bf16_t* db = dst + (int64_t)b * total;
  const float sy = (float)Hin / (float)Hout, sx = (float)Win / (float)Wout;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % Cpad);
    const int64_t pix = i / Cpad;
    float v = 0.f;
    if (c < 3) {
      const int y = (int)(pix / Wout), x = (int)(pix - (int64_t)y * Wout);
      const int yi = min((int)floorf(__fmul_rn((float)y, sy)), Hin - 1);
      const int xi = min((int)floorf(__fmul_rn((float)x, sx)), Win - 1);
      const float u = (float)sb[((int64_t)yi * Win + xi) * 3 + c];
      v = __fdiv_rn(__fsub_rn(__fdiv_rn(u, 255.0f), 0.5f), 0.5f);
      if (renorm) v = __fsub_rn(__fmul_rn(2.0f, v), 1.0f);
    }
    db[i] = f2bf(v);
  }
}

// decoder output NCHW (bf16 / fp32, in [-1, 1]) -> uint8 NHWC: VaeImageProcessor.postprocess up to the PIL array,
//   t = clamp(bf16(bf16(x / 2) + 0.5), 0, 1) in the tensor's dtype, then rint(float(t) * 255) (numpy round = half even)
template <typename T>
__global__ __launch_bounds__(256) void image_to_u8_kernel(const T* src, uint8_t* dst, int C, int64_t HW) {
  const int b = blockIdx.y;
  const int64_t total = HW * C;
  const T* sb = src + (int64_t)b * total;
  uint8_t* db = dst + (int64_t)b * total;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    const int64_t p = i / C;
    float t;
    if constexpr (sizeof(T) == 4) t = __fadd_rn(__fdiv_rn(sb[(int64_t)c * HW + p], 2.0f), 0.5f);
    else t = round_bf(round_bf(bf2f(sb[(int64_t)c * HW + p]) * 0.5f) + 0.5f);
    t = fminf(fmaxf(t, 0.f), 1.f);
    db[i] = (uint8_t)rintf(__fmul_rn(t, 255.0f));
  }
}

// ---- fp32-class encoder (reference: train_denoiser.py:458,887-918 keeps the VAE in fp32) --------------------------------
// Activations are fp32 NHWC.  A matrix product a . w is computed on the bf16 MFMA as the K-concatenation
//   [a_hi | a_lo | a_hi] . [w_hi | w_hi | w_lo]   (a_hi = bf16(a), a_lo = bf16(a - a_hi); fp32 accumulation),
// i.e. every fp32 operand is handed to the GEMM / implicit-GEMM kernels as 2 (bf16 weights: w_lo = 0) or 3 bf16 "parts"
// laid side by side along K; the dropped term a_lo . w_lo is ~2^-16 of the product.
FK_DEV void split_f32(float v, float& hi, float& lo) {
  hi = round_bf(v);
  lo = v - hi;   // exact in fp32; rounded to bf16 when packed
}

// part slots of (hi, lo): activation order (hi, lo, hi), weight order (hi, hi, lo); parts = 2: (hi, lo)
FK_DEV void store_parts(bf16_t* y, int64_t ps, int parts, int order, const float* hi, const float* lo) {
  u32x2_t h, l;
  h[0] = pack_bf2(hi[0], hi[1]); h[1] = pack_bf2(hi[2], hi[3]);
  l[0] = pack_bf2(lo[0], lo[1]); l[1] = pack_bf2(lo[2], lo[3]);
  *(u32x2_t*)y = h;
  if (parts == 2) {
    *(u32x2_t*)(y + ps) = l;
  } else {
    *(u32x2_t*)(y + ps) = order ? h : l;
    *(u32x2_t*)(y + 2 * ps) = order ? l : h;
  }
}

__global__ __launch_bounds__(256) void split_rows_kernel(const float* x, int64_t ldx, bf16_t* y, int64_t ldy, int64_t ps,
                                                         int64_t rows, int n, int parts, int order) {
  const int nv = n / 4;
  const int64_t total = rows * nv;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / nv;
    const int c = (int)(i - r * nv) * 4;
    const f32x4_t v = *(const f32x4_t*)(x + r * ldx + c);
    float hi[4], lo[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) split_f32(v[e], hi[e], lo[e]);
    store_parts(y + r * ldy + c, ps, parts, order, hi, lo);
  }
}

__global__ __launch_bounds__(256) void gn_apply_f32_kernel(const float* x, bf16_t* y, const float* stats,
                                                           const float* gamma, const float* beta, int64_t HW, int C,
                                                           int groups, int silu, int parts) {
  const int b = blockIdx.y;
  const int cpr = C / 4, cpg = C / groups;   // cpg >= 4: the four channels of a vector share a group
  const int64_t nvec = HW * cpr;
  const float* xb = x + (int64_t)b * HW * C;
  bf16_t* yb = y + (int64_t)b * HW * C * parts;
  const float* st = stats + (int64_t)b * groups * 2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
    const int64_t pix = i / cpr;
    const int c0 = (int)(i - pix * cpr) * 4;
    const f32x4_t v = *(const f32x4_t*)(xb + i * 4);
    const f32x4_t gw = *(const f32x4_t*)(gamma + c0), bw = *(const f32x4_t*)(beta + c0);
    const int g = c0 / cpg;
    const float mean = st[2 * g], rstd = st[2 * g + 1];
    float hi[4], lo[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float t = (v[e] - mean) * rstd * gw[e] + bw[e];
      if (silu) t = t / (1.0f + expf(-t));
      split_f32(t, hi[e], lo[e]);
    }
    store_parts(yb + pix * (int64_t)C * parts + c0, C, parts, 0, hi, lo);
  }
}

// NCHW fp32 -> NHWC bf16 parts, every part zero padded to Cpad channels: [B, HW, parts * Cpad]
__global__ __launch_bounds__(256) void nchw_f32_to_parts_kernel(const float* src, bf16_t* dst, int C, int Cpad, int64_t HW,
                                                                int parts) {
  const int b = blockIdx.y;
  const int64_t total = HW * Cpad;
  const float* sb = src + (int64_t)b * C * HW;
  bf16_t* db = dst + (int64_t)b * HW * Cpad * parts;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % Cpad);
    const int64_t p = i / Cpad;
    float hi = 0.f, lo = 0.f;
    if (c < C) split_f32(sb[(int64_t)c * HW + p], hi, lo);
    bf16_t* o = db + p * Cpad * parts + c;
    o[0] = f2bf(hi);
    o[Cpad] = f2bf(lo);
    if (parts == 3) o[2 * Cpad] = f2bf(hi);
  }
}

__global__ __launch_bounds__(256) void nhwc_f32_to_nchw_kernel(const float* src, float* dst, int C, int Cpad, int64_t HW,
                                                               float add, float mul) {
  const int b = blockIdx.y;
  const int64_t total = HW * C;
  const float* sb = src + (int64_t)b * HW * Cpad;
  float* db = dst + (int64_t)b * C * HW;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i / HW);
    const int64_t p = i - (int64_t)c * HW;
    db[i] = __fmul_rn(__fadd_rn(sb[p * Cpad + c], add), mul);
  }
}

inline int ew_grid(int64_t n) {
  int64_t g = (n + 255) / 256;
  return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

}  // namespace

extern "C" int64_t fk_groupnorm_ws_floats(int32_t B, int64_t HW, int32_t C) {
  if (B <= 0 || HW <= 0 || C <= 0 || C % 32 != 0) return -1;
  return (int64_t)B * gn_blocks(HW, C) * 32 * 2;
}

extern "C" int fk_groupnorm_stats_nhwc_bf16(const void* x, float* stats, float* ws, int32_t B, int64_t HW,
                                            int32_t C, int32_t groups, float eps, fk_stream_t stream_) {
  FK_CHECK_ARG(x && stats && ws && B > 0 && HW > 0, "fk_groupnorm_stats: bad arguments");
  FK_CHECK_ARG(groups == 32 && C % 128 == 0 && C <= 1024 && (C / 8) <= GN_THREADS,
               "fk_groupnorm_stats: needs 32 groups and C in {128, 256, 512} (got C=%d groups=%d)", C, groups);
  FK_CHECK_ARG((uintptr_t)x % 16 == 0, "fk_groupnorm_stats: alignment");
  hipStream_t stream = (hipStream_t)stream_;
  const int nblk = gn_blocks(HW, C);
  hipLaunchKernelGGL(gn_partial_kernel<bf16_t>, dim3(nblk, B), dim3(GN_THREADS), 0, stream, (const bf16_t*)x, ws, HW, C, groups);
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(groups, B), dim3(64), 0, stream, (const float*)ws, stats, nblk, groups,
                     (double)HW * (C / groups), eps);
  FK_CHECK_LAUNCH("fk_groupnorm_stats_nhwc_bf16");
  return FK_OK;
}

extern "C" int fk_groupnorm_apply_nhwc_bf16(const void* x, void* y, const float* stats, const void* gamma,
                                            const void* beta, int32_t B, int64_t HW, int32_t C, int32_t groups,
                                            int32_t silu, fk_stream_t stream_) {
  FK_CHECK_ARG(x && y && stats && gamma && beta && B > 0 && HW > 0, "fk_groupnorm_apply: bad arguments");
  FK_CHECK_ARG(groups == 32 && C % 64 == 0, "fk_groupnorm_apply: needs 32 groups, C %% 64 == 0");
  FK_CHECK_ARG(((uintptr_t)x % 16 == 0) && ((uintptr_t)y % 16 == 0) && ((uintptr_t)gamma % 16 == 0) &&
                   ((uintptr_t)beta % 16 == 0), "fk_groupnorm_apply: alignment");
  hipLaunchKernelGGL(gn_apply_kernel, dim3(ew_grid(HW * (C / 8)), B), dim3(256), 0, (hipStream_t)stream_,
                     (const bf16_t*)x, (bf16_t*)y, stats, (const bf16_t*)gamma, (const bf16_t*)beta, HW, C, groups,
                     silu);
  FK_CHECK_LAUNCH("fk_groupnorm_apply_nhwc_bf16");
  return FK_OK;
}

extern "C" int fk_nchw_to_nhwc_bf16(const void* src, int32_t src_is_fp32, void* dst, int32_t B, int32_t C,
                                    int32_t Cpad, int32_t H, int32_t W, float div, float add,
                                    fk_stream_t stream_) {
  FK_CHECK_ARG(src && dst && B > 0 && C > 0 && Cpad >= C && H > 0 && W > 0, "fk_nchw_to_nhwc_bf16: bad arguments");
  const int64_t HW = (int64_t)H * W;
  const dim3 grid(ew_grid(HW * Cpad), B), block(256);
  hipStream_t s = (hipStream_t)stream_;
  if (src_is_fp32) hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, grid, block, 0, s, (const float*)src, (bf16_t*)dst, C, Cpad, HW, div, add);
  else hipLaunchKernelGGL(nchw_to_nhwc_kernel<bf16_t>, grid, block, 0, s, (const bf16_t*)src, (bf16_t*)dst, C, Cpad, HW, div, add);
  FK_CHECK_LAUNCH("fk_nchw_to_nhwc_bf16");
  return FK_OK;
}

extern "C" int fk_nhwc_to_nchw(const void* src, void* dst, int32_t dst_is_fp32, int32_t B, int32_t C,
                               int32_t Cpad, int32_t H, int32_t W, float add, float mul, fk_stream_t stream_) {
  FK_CHECK_ARG(src && dst && B > 0 && C > 0 && Cpad >= C && H > 0 && W > 0, "fk_nhwc_to_nchw: bad arguments");
  const int64_t HW = (int64_t)H * W;
  const dim3 grid(ew_grid(HW * C), B), block(256);
  hipStream_t s = (hipStream_t)stream_;
  if (dst_is_fp32) hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, grid, block, 0, s, (const bf16_t*)src, (float*)dst, C, Cpad, HW, add, mul);
  else hipLaunchKernelGGL(nhwc_to_nchw_kernel<bf16_t>, grid, block, 0, s, (const bf16_t*)src, (bf16_t*)dst, C, Cpad, HW, add, mul);
  FK_CHECK_LAUNCH("fk_nhwc_to_nchw");
  return FK_OK;
}

extern "C" int fk_pixels_u8_to_nhwc_bf16(const void* src, void* dst, int32_t B, int32_t Hin, int32_t Win, int32_t Hout,
                                         int32_t Wout, int32_t Cpad, int32_t renorm, fk_stream_t stream_) {
  FK_CHECK_ARG(src && dst && B > 0 && Hin > 0 && Win > 0 && Hout > 0 && Wout > 0 && Cpad >= 3,
               "fk_pixels_u8_to_nhwc_bf16: bad arguments");
  const dim3 grid(ew_grid((int64_t)Hout * Wout * Cpad), B), block(256);
  hipLaunchKernelGGL(pixels_u8_to_nhwc_kernel, grid, block, 0, (hipStream_t)stream_, (const uint8_t*)src, (bf16_t*)dst,
                     Hin, Win, Hout, Wout, Cpad, renorm);
  FK_CHECK_LAUNCH("fk_pixels_u8_to_nhwc_bf16");
  return FK_OK;
}

extern "C" int fk_image_to_u8_nhwc(const void* src, int32_t src_is_fp32, void* dst, int32_t B, int32_t C, int32_t H,
                                   int32_t W, fk_stream_t stream_) {
  FK_CHECK_ARG(src && dst && B > 0 && C > 0 && H > 0 && W > 0, "fk_image_to_u8_nhwc: bad arguments");
  const int64_t HW = (int64_t)H * W;
  const dim3 grid(ew_grid(HW * C), B), block(256);
  hipStream_t s = (hipStream_t)stream_;
  if (src_is_fp32) hipLaunchKernelGGL(image_to_u8_kernel<float>, grid, block, 0, s, (const float*)src, (uint8_t*)dst, C, HW);
  else hipLaunchKernelGGL(image_to_u8_kernel<bf16_t>, grid, block, 0, s, (const bf16_t*)src, (uint8_t*)dst, C, HW);
  FK_CHECK_LAUNCH("fk_image_to_u8_nhwc");
  return FK_OK;
}

// ---- fp32-class encoder entry points ----------------------------------------------------------------------------------
extern "C" int fk_split_f32_rows(const float* x, int64_t ldx, void* y, int64_t ldy, int64_t part_stride, int64_t rows,
                                 int32_t n, int32_t parts, int32_t weight_order, fk_stream_t stream_) {
  FK_CHECK_ARG(x && y && rows > 0 && n > 0 && n % 4 == 0 && (parts == 2 || parts == 3), "fk_split_f32_rows: bad arguments");
  FK_CHECK_ARG(ldx % 4 == 0 && ldy % 4 == 0 && part_stride % 4 == 0 && ((uintptr_t)x % 16 == 0) && ((uintptr_t)y % 8 == 0),
               "fk_split_f32_rows: alignment");
  hipLaunchKernelGGL(split_rows_kernel, dim3(ew_grid(rows * (n / 4))), dim3(256), 0, (hipStream_t)stream_, x, ldx,
                     (bf16_t*)y, ldy, part_stride, rows, n, parts, weight_order ? 1 : 0);
  FK_CHECK_LAUNCH("fk_split_f32_rows");
  return FK_OK;
}

extern "C" int fk_groupnorm_f32_nhwc(const float* x, void* y_parts, float* stats, float* ws, const float* gamma,
                                     const float* beta, int32_t B, int64_t HW, int32_t C, int32_t groups, float eps,
                                     int32_t silu, int32_t parts, fk_stream_t stream_) {
  FK_CHECK_ARG(x && y_parts && stats && ws && gamma && beta && B > 0 && HW > 0 && (parts == 2 || parts == 3),
               "fk_groupnorm_f32_nhwc: bad arguments");
  FK_CHECK_ARG(groups == 32 && C % 128 == 0 && C <= 1024, "fk_groupnorm_f32_nhwc: needs 32 groups and C in {128, 256, 512}");
  FK_CHECK_ARG(((uintptr_t)x % 16 == 0) && ((uintptr_t)y_parts % 16 == 0) && ((uintptr_t)gamma % 16 == 0) &&
                   ((uintptr_t)beta % 16 == 0), "fk_groupnorm_f32_nhwc: alignment");
  hipStream_t stream = (hipStream_t)stream_;
  const int nblk = gn_blocks(HW, C);
  hipLaunchKernelGGL(gn_partial_kernel<float>, dim3(nblk, B), dim3(GN_THREADS), 0, stream, x, ws, HW, C, groups);
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(groups, B), dim3(64), 0, stream, (const float*)ws, stats, nblk, groups,
                     (double)HW * (C / groups), eps);
  hipLaunchKernelGGL(gn_apply_f32_kernel, dim3(ew_grid(HW * (C / 4)), B), dim3(256), 0, stream, x, (bf16_t*)y_parts,
                     (const float*)stats, gamma, beta, HW, C, groups, silu, parts);
  FK_CHECK_LAUNCH("fk_groupnorm_f32_nhwc");
  return FK_OK;
}

extern "C" int fk_nchw_f32_to_nhwc_parts(const float* src, void* dst, int32_t B, int32_t C, int32_t Cpad, int32_t H,
                                         int32_t W, int32_t parts, fk_stream_t stream_) {
  FK_CHECK_ARG(src && dst && B > 0 && C > 0 && Cpad >= C && H > 0 && W > 0 && (parts == 2 || parts == 3),
               "fk_nchw_f32_to_nhwc_parts: bad arguments");
  const int64_t HW = (int64_t)H * W;
  hipLaunchKernelGGL(nchw_f32_to_parts_kernel, dim3(ew_grid(HW * Cpad), B), dim3(256), 0, (hipStream_t)stream_, src,
                     (bf16_t*)dst, C, Cpad, HW, parts);
  FK_CHECK_LAUNCH("fk_nchw_f32_to_nhwc_parts");
  return FK_OK;
}

extern "C" int fk_nhwc_f32_to_nchw(const float* src, float* dst, int32_t B, int32_t C, int32_t Cpad, int32_t H, int32_t W,
                                   float add, float mul, fk_stream_t stream_) {
  FK_CHECK_ARG(src && dst && B > 0 && C > 0 && Cpad >= C && H > 0 && W > 0, "fk_nhwc_f32_to_nchw: bad arguments");
  const int64_t HW = (int64_t)H * W;
  hipLaunchKernelGGL(nhwc_f32_to_nchw_kernel, dim3(ew_grid(HW * C), B), dim3(256), 0, (hipStream_t)stream_, src, dst, C,
                     Cpad, HW, add, mul);
  FK_CHECK_LAUNCH("fk_nhwc_f32_to_nchw");
  return FK_OK;
}
