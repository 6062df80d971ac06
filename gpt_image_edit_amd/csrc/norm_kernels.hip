// HBM-bound fused normalisation kernels of the MMDiT blocks (K2, K3 in SURVEY.md section 2.2).
//   fk_ln_modulate_bf16 : LayerNorm(no affine) + AdaLN modulate, one read + one write per element.
//   fk_qkv_post_bf16    : per-head RMSNorm + RoPE of q, k and [B,S,3D] -> [B,H,S,128] re-layout (V stays in place).
// Every intermediate is rounded to bf16 where the reference's bf16 torch graph rounds it (see fk.h).
#include "fk_common.h"

namespace {

// ------------------------------------------------------------------------------------------------------
// LN + modulate: one wave per row of D = 512 * NV elements, row held in registers (NV x 16 bytes / lane).
template <int NV>
__global__ __launch_bounds__(256) void ln_modulate_kernel(const bf16_t* x, fk_rows xr, bf16_t* out,
                                                          fk_rows outr, const bf16_t* shift,
                                                          const bf16_t* scale, const bf16_t* shift_b,
                                                          const bf16_t* scale_b, int64_t split,
                                                          int64_t mod_bs, int64_t mod_rpb, int64_t M,
                                                          float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  constexpr int D = NV * 512;
  const bf16_t* xp = x + fk_row_offset(xr, row) + lane * 8;
  // every global read of the row's work is issued up front (the modulation vectors do not depend on the statistics):
  // one memory latency per row instead of two
  const int64_t b = row / mod_rpb;
  const bool second = (row - b * mod_rpb) >= split;  // rows >= split of each batch use the second vector set
  const bf16_t* sc = (second ? scale_b : scale) + b * mod_bs + lane * 8;
  const bf16_t* sh = (second ? shift_b : shift) + b * mod_bs + lane * 8;
  u32x4_t xw[NV], scw[NV], shw[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) xw[i] = *(const u32x4_t*)(xp + i * 512);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    scw[i] = *(const u32x4_t*)(sc + i * 512);
    shw[i] = *(const u32x4_t*)(sh + i * 512);
  }
  float v[NV][8];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[i][2 * e] = bf_lo(xw[i][e]);
      v[i][2 * e + 1] = bf_hi(xw[i][e]);
      sum += v[i][2 * e] + v[i][2 * e + 1];
    }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off);
  const float mean = sum * (1.0f / D);
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float d = v[i][e] - mean;
      sq += d * d;
    }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) sq += __shfl_xor(sq, off);
  const float rstd = rsqrtf(sq * (1.0f / D) + eps);

  bf16_t* op = out + fk_row_offset(outr, row) + lane * 8;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    u32x4_t ow;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      // rounding points of the reference graph: LN -> bf16, (1 + scale) -> bf16, product -> bf16, sum -> bf16
      float y0 = (v[i][2 * e] - mean) * rstd, y1 = (v[i][2 * e + 1] - mean) * rstd;
      round_bf_pair(y0, y1);
      float g0 = 1.0f + bf_lo(scw[i][e]), g1 = 1.0f + bf_hi(scw[i][e]);
      round_bf_pair(g0, g1);
      y0 *= g0;
      y1 *= g1;
      round_bf_pair(y0, y1);
      ow[e] = pack_bf2(y0 + bf_lo(shw[i][e]), y1 + bf_hi(shw[i][e]));
    }
    *(u32x4_t*)(op + i * 512) = ow;
  }
}

// ------------------------------------------------------------------------------------------------------
// QKV post-processing.  grid = (S_pad/64, H, B), 256 threads; one block = 64 tokens of one head.
constexpr int HD = 128;

__global__ __launch_bounds__(256) void qkv_post_kernel(const bf16_t* qkv, bf16_t* q_out, bf16_t* k_out,
                                                       const bf16_t* wq_img,
                                                       const bf16_t* wk_img, const bf16_t* wq_txt,
                                                       const bf16_t* wk_txt, const float* cosT,
                                                       const float* sinT, int B, int S, int S_txt, int H,
                                                       float eps) {
  const int tid = threadIdx.x;
  const int s0 = blockIdx.x * 64, h = blockIdx.y, b = blockIdx.z;
  const int D3 = 3 * H * HD;
  const int chunk = tid & 15;       // 8 consecutive head-dim elements
  const int r0 = tid >> 4;          // token row inside the tile (+16 i)
  const bf16_t* base = qkv + (int64_t)b * S * D3 + h * HD + chunk * 8;

  // --- q and k: RMSNorm over the 128-wide head (16 lanes x 8 elements), weight, RoPE ----------------
#pragma unroll
  for (int which = 0; which < 2; ++which) {
    bf16_t* dst = which == 0 ? q_out : k_out;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int s = s0 + r0 + 16 * i;
      const bool valid = s < S;  // uniform across the 16 lanes of a row
      float xv[8];
      u32x4_t w = {0u, 0u, 0u, 0u};
      if (valid) w = *(const u32x4_t*)(base + (int64_t)s * D3 + which * H * HD);
      float ss = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        xv[2 * e] = bf_lo(w[e]);
        xv[2 * e + 1] = bf_hi(w[e]);
        ss += xv[2 * e] * xv[2 * e] + xv[2 * e + 1] * xv[2 * e + 1];
      }
#pragma unroll
      for (int off = 8; off >= 1; off >>= 1) ss += __shfl_xor(ss, off);
      if (!valid) continue;
      const float rs = rsqrtf(ss * (1.0f / HD) + eps);
      const bf16_t* wsel = (s < S_txt) ? (which == 0 ? wq_txt : wk_txt) : (which == 0 ? wq_img : wk_img);
      const u32x4_t ww = *(const u32x4_t*)(wsel + chunk * 8);
      const f32x4_t c0 = *(const f32x4_t*)(cosT + (int64_t)s * HD + chunk * 8);
      const f32x4_t c1 = *(const f32x4_t*)(cosT + (int64_t)s * HD + chunk * 8 + 4);
      const f32x4_t n0 = *(const f32x4_t*)(sinT + (int64_t)s * HD + chunk * 8);
      const f32x4_t n1 = *(const f32x4_t*)(sinT + (int64_t)s * HD + chunk * 8 + 4);
      const float cs[8] = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
      const float sn[8] = {n0[0], n0[1], n0[2], n0[3], n1[0], n1[1], n1[2], n1[3]};
      float y[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        // x * rsqrt(var + eps) in fp32 -> bf16 -> * weight (bf16 multiply)
        y[2 * e] = round_bf(round_bf(xv[2 * e] * rs) * bf_lo(ww[e]));
        y[2 * e + 1] = round_bf(round_bf(xv[2 * e + 1] * rs) * bf_hi(ww[e]));
      }
      u32x4_t ow;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        // out = x*cos + rot(x)*sin with rot = (-x_imag, x_real); fp32, no contraction
        const float re = y[2 * e], im = y[2 * e + 1];
        const float o0 = __fadd_rn(__fmul_rn(re, cs[2 * e]), __fmul_rn(-im, sn[2 * e]));
        const float o1 = __fadd_rn(__fmul_rn(im, cs[2 * e + 1]), __fmul_rn(re, sn[2 * e + 1]));
        ow[e] = pack_bf2(o0, o1);
      }
      *(u32x4_t*)(dst + (((int64_t)b * H + h) * S + s) * HD + chunk * 8) = ow;
    }
  }

}

}  // namespace

extern "C" int fk_ln_modulate2_bf16(const void* x, fk_rows xr, void* out, fk_rows outr, const void* shift,
                                    const void* scale, const void* shift_b, const void* scale_b, int64_t split,
                                    int64_t mod_batch_stride, int64_t mod_rows_per_batch, int64_t M, int32_t D,
                                    float eps, fk_stream_t stream_) {
  FK_CHECK_ARG(x && out && shift && scale && shift_b && scale_b, "fk_ln_modulate_bf16: null pointer");
  FK_CHECK_ARG(((uintptr_t)shift_b % 16 == 0) && ((uintptr_t)scale_b % 16 == 0), "fk_ln_modulate_bf16: alignment");
  FK_CHECK_ARG(M > 0 && mod_rows_per_batch > 0, "fk_ln_modulate_bf16: bad M / rows per batch");
  FK_CHECK_ARG(xr.ld % 8 == 0 && outr.ld % 8 == 0 && mod_batch_stride % 8 == 0 &&
                   (xr.rows_per_batch <= 0 || xr.batch_stride % 8 == 0) &&
                   (outr.rows_per_batch <= 0 || outr.batch_stride % 8 == 0),
               "fk_ln_modulate_bf16: strides must be multiples of 8 elements");
  FK_CHECK_ARG(((uintptr_t)x % 16 == 0) && ((uintptr_t)out % 16 == 0) && ((uintptr_t)shift % 16 == 0) &&
                   ((uintptr_t)scale % 16 == 0),
               "fk_ln_modulate_bf16: pointers must be 16-byte aligned");
  hipStream_t stream = (hipStream_t)stream_;
  const dim3 grid((unsigned)((M + 3) / 4)), block(256);
#define FK_LN_CASE(NV)                                                                              \
  case NV * 512:                                                                                    \
    hipLaunchKernelGGL(ln_modulate_kernel<NV>, grid, block, 0, stream, (const bf16_t*)x, xr,        \
                       (bf16_t*)out, outr, (const bf16_t*)shift, (const bf16_t*)scale,              \
                       (const bf16_t*)shift_b, (const bf16_t*)scale_b, split, mod_batch_stride,     \
                       mod_rows_per_batch, M, eps);                                                 \
    break;
  switch (D) {
    FK_LN_CASE(1)
    FK_LN_CASE(2)
    FK_LN_CASE(6)
    default:
      fk_set_error("fk_ln_modulate_bf16: D=%d unsupported (512, 1024, 3072)", D);
      return FK_EUNSUPPORTED;
  }
#undef FK_LN_CASE
  FK_CHECK_LAUNCH("fk_ln_modulate_bf16");
  return FK_OK;
}

extern "C" int fk_ln_modulate_bf16(const void* x, fk_rows xr, void* out, fk_rows outr, const void* shift,
                                   const void* scale, int64_t mod_batch_stride, int64_t mod_rows_per_batch,
                                   int64_t M, int32_t D, float eps, fk_stream_t stream) {
  return fk_ln_modulate2_bf16(x, xr, out, outr, shift, scale, shift, scale, mod_rows_per_batch, mod_batch_stride,
                              mod_rows_per_batch, M, D, eps, stream);
}

extern "C" int fk_qkv_post_bf16(const void* qkv, void* q_out, void* k_out, const void* wq_img,
                                const void* wk_img, const void* wq_txt, const void* wk_txt,
                                const float* cos, const float* sin, int32_t B, int32_t S, int32_t S_txt,
                                int32_t H, float eps, fk_stream_t stream_) {
  FK_CHECK_ARG(qkv && q_out && k_out && wq_img && wk_img && cos && sin, "fk_qkv_post_bf16: null pointer");
  FK_CHECK_ARG(S_txt == 0 || (wq_txt && wk_txt), "fk_qkv_post_bf16: text-stream norm weights missing");
  FK_CHECK_ARG(B > 0 && S > 0 && H > 0 && S_txt >= 0 && S_txt <= S, "fk_qkv_post_bf16: bad sizes");
  FK_CHECK_ARG(((uintptr_t)qkv % 16 == 0) && ((uintptr_t)q_out % 16 == 0) && ((uintptr_t)k_out % 16 == 0) &&
                   ((uintptr_t)cos % 16 == 0) && ((uintptr_t)sin % 16 == 0),
               "fk_qkv_post_bf16: pointers must be 16-byte aligned");
  if (!wq_txt) { wq_txt = wq_img; wk_txt = wk_img; }
  hipLaunchKernelGGL(qkv_post_kernel, dim3((S + 63) / 64, H, B), dim3(256), 0, (hipStream_t)stream_,
                     (const bf16_t*)qkv, (bf16_t*)q_out, (bf16_t*)k_out,
                     (const bf16_t*)wq_img, (const bf16_t*)wk_img, (const bf16_t*)wq_txt,
                     (const bf16_t*)wk_txt, cos, sin, B, S, S_txt, H, eps);
  FK_CHECK_LAUNCH("fk_qkv_post_bf16");
  return FK_OK;
}
