// Small HBM-/latency-bound kernels around the MMDiT forward and the Euler update (K6, K8).
#include "fk_common.h"

namespace {

__global__ __launch_bounds__(256) void silu_kernel(const bf16_t* x, bf16_t* y, int64_t nvec) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
    u32x4_t w = *(const u32x4_t*)(x + i * 8);
#pragma unroll
    for (int e = 0; e < 4; ++e) w[e] = pack_bf2(silu_f(bf_lo(w[e])), silu_f(bf_hi(w[e])));
    *(u32x4_t*)(y + i * 8) = w;
  }
}

__global__ __launch_bounds__(256) void add3_kernel(const bf16_t* a, const bf16_t* b, const bf16_t* c,
                                                   bf16_t* out, int64_t nvec) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
    const u32x4_t aw = *(const u32x4_t*)(a + i * 8), bw = *(const u32x4_t*)(b + i * 8),
                  cw = *(const u32x4_t*)(c + i * 8);
    u32x4_t ow;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float s0 = round_bf(bf_lo(aw[e]) + bf_lo(bw[e]));
      const float s1 = round_bf(bf_hi(aw[e]) + bf_hi(bw[e]));
      ow[e] = pack_bf2(s0 + bf_lo(cw[e]), s1 + bf_hi(cw[e]));
    }
    *(u32x4_t*)(out + i * 8) = ow;
  }
}

// true CFG: out = neg + scale * (pos - neg), each bf16 tensor op rounded like the reference graph
__global__ __launch_bounds__(256) void true_cfg_kernel(const bf16_t* pos, const bf16_t* neg, bf16_t* out, float scale,
                                                       int64_t nvec) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
    const u32x4_t pw = *(const u32x4_t*)(pos + i * 8), nw = *(const u32x4_t*)(neg + i * 8);
    u32x4_t ow;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float n0 = bf_lo(nw[e]), n1 = bf_hi(nw[e]);
      const float d0 = round_bf(bf_lo(pw[e]) - n0), d1 = round_bf(bf_hi(pw[e]) - n1);
      ow[e] = pack_bf2(n0 + round_bf(__fmul_rn(scale, d0)), n1 + round_bf(__fmul_rn(scale, d1)));
    }
    *(u32x4_t*)(out + i * 8) = ow;
  }
}

// Timesteps(256, flip_sin_to_cos=True, shift 0) on bf16(v)*1000 (bf16 multiply): out[b] = [cos | sin]
__global__ void timestep_proj_kernel(const void* v, int v_is_fp32, const float* freqs, bf16_t* out, int B) {
  const int b = blockIdx.x;
  const int k = threadIdx.x;  // 0..127
  float t = v_is_fp32 ? ((const float*)v)[b] : bf2f(((const bf16_t*)v)[b]);
  t = round_bf(round_bf(t) * 1000.0f);
  const float ang = __fmul_rn(t, freqs[k]);
  out[(int64_t)b * 256 + k] = f2bf(cosf(ang));
  out[(int64_t)b * 256 + 128 + k] = f2bf(sinf(ang));
}

__global__ __launch_bounds__(256) void euler_kernel(bf16_t* x, int64_t x_bs, const bf16_t* v, int64_t v_bs,
                                                    int S_tgt, int C, float dsig_bf) {
  const int b = blockIdx.y;
  const int64_t nvec = (int64_t)S_tgt * C / 8;
  bf16_t* xb = x + (int64_t)b * x_bs;
  const bf16_t* vb = v + (int64_t)b * v_bs;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
    const u32x4_t xw = *(const u32x4_t*)(xb + i * 8), vw = *(const u32x4_t*)(vb + i * 8);
    u32x4_t ow;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float p0 = round_bf(dsig_bf * bf_lo(vw[e]));
      const float p1 = round_bf(dsig_bf * bf_hi(vw[e]));
      ow[e] = pack_bf2(bf_lo(xw[e]) + p0, bf_hi(xw[e]) + p1);
    }
    *(u32x4_t*)(xb + i * 8) = ow;
  }
}

// [R, C] -> [C, R] through a 64x64 LDS tile (padded), 256 threads.
__global__ __launch_bounds__(256) void transpose_kernel(const bf16_t* src, int64_t lds_, int64_t sbs,
                                                        bf16_t* dst, int64_t ldd, int64_t dbs, int R, int C) {
  __shared__ bf16_t tile[64][66];
  src += (int64_t)blockIdx.z * sbs;
  dst += (int64_t)blockIdx.z * dbs;
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4) {
    const int r = r0 + i, c = c0 + tx;
    tile[i][tx] = (r < R && c < C) ? src[(int64_t)r * lds_ + c] : (bf16_t)0;
  }
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {
    const int c = c0 + i, r = r0 + tx;
    if (c < C && r < R) dst[(int64_t)c * ldd + r] = tile[tx][i];
  }
}

// Row softmax: fp32 scores in, bf16 probabilities out; one block (256 threads) per row, n <= 32768.
template <int NV>
// parts = 3 (fp32-class VAE encoder): the fp32 probability is written as the bf16 parts (hi, lo, hi), ps columns apart.
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* x, bf16_t* y, int64_t ldx,
                                                           int64_t ldy, int n, int64_t ps, int parts) {
  __shared__ float red[8];
  const int tid = threadIdx.x;
  const float* xr = x + (int64_t)blockIdx.x * ldx;
  bf16_t* yr = y + (int64_t)blockIdx.x * ldy;
  float v[NV][4];
  float mx = -3.0e38f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int idx = (i * 256 + tid) * 4;
    if (idx < n) {
      const f32x4_t w = *(const f32x4_t*)(xr + idx);
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[i][e] = w[e]; mx = fmaxf(mx, w[e]); }
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[i][e] = -3.0e38f;
    }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
  if ((tid & 63) == 0) red[tid >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[i][e] = expf(v[i][e] - mx);
      sum += v[i][e];
    }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off);
  if ((tid & 63) == 0) red[4 + (tid >> 6)] = sum;
  __syncthreads();
  const float inv = 1.0f / ((red[4] + red[5]) + (red[6] + red[7]));
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int idx = (i * 256 + tid) * 4;
    if (idx < n) {
      u32x2_t pk;
      pk[0] = pack_bf2(v[i][0] * inv, v[i][1] * inv);
      pk[1] = pack_bf2(v[i][2] * inv, v[i][3] * inv);
      *(u32x2_t*)(yr + idx) = pk;
      if (parts == 3) {
        u32x2_t lo;
        lo[0] = pack_bf2(v[i][0] * inv - bf_lo(pk[0]), v[i][1] * inv - bf_hi(pk[0]));
        lo[1] = pack_bf2(v[i][2] * inv - bf_lo(pk[1]), v[i][3] * inv - bf_hi(pk[1]));
        *(u32x2_t*)(yr + ps + idx) = lo;
        *(u32x2_t*)(yr + 2 * ps + idx) = pk;
      }
    }
  }
}

inline int ew_grid(int64_t nvec) {
  int64_t g = (nvec + 255) / 256;
  return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

}  // namespace

extern "C" int fk_silu_bf16(const void* x, void* y, int64_t n, fk_stream_t stream) {
  FK_CHECK_ARG(x && y && n > 0 && n % 8 == 0, "fk_silu_bf16: n=%lld must be a positive multiple of 8", (long long)n);
  FK_CHECK_ARG(((uintptr_t)x % 16 == 0) && ((uintptr_t)y % 16 == 0), "fk_silu_bf16: alignment");
  hipLaunchKernelGGL(silu_kernel, dim3(ew_grid(n / 8)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)x, (bf16_t*)y, n / 8);
  FK_CHECK_LAUNCH("fk_silu_bf16");
  return FK_OK;
}

extern "C" int fk_add3_bf16(const void* a, const void* b, const void* c, void* out, int64_t n,
                            fk_stream_t stream) {
  FK_CHECK_ARG(a && b && c && out && n > 0 && n % 8 == 0, "fk_add3_bf16: n must be a positive multiple of 8");
  FK_CHECK_ARG(((uintptr_t)a % 16 == 0) && ((uintptr_t)b % 16 == 0) && ((uintptr_t)c % 16 == 0) &&
                   ((uintptr_t)out % 16 == 0), "fk_add3_bf16: alignment");
  hipLaunchKernelGGL(add3_kernel, dim3(ew_grid(n / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)a,
                     (const bf16_t*)b, (const bf16_t*)c, (bf16_t*)out, n / 8);
  FK_CHECK_LAUNCH("fk_add3_bf16");
  return FK_OK;
}

extern "C" int fk_true_cfg_bf16(const void* pos, const void* neg, void* out, float scale, int64_t n,
                                fk_stream_t stream) {
  FK_CHECK_ARG(pos && neg && out && n > 0 && n % 8 == 0, "fk_true_cfg_bf16: n must be a positive multiple of 8");
  FK_CHECK_ARG(((uintptr_t)pos % 16 == 0) && ((uintptr_t)neg % 16 == 0) && ((uintptr_t)out % 16 == 0),
               "fk_true_cfg_bf16: alignment");
  hipLaunchKernelGGL(true_cfg_kernel, dim3(ew_grid(n / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)pos,
                     (const bf16_t*)neg, (bf16_t*)out, scale, n / 8);
  FK_CHECK_LAUNCH("fk_true_cfg_bf16");
  return FK_OK;
}

extern "C" int fk_timestep_proj(const void* v, int32_t v_is_fp32, const float* freqs, void* out, int32_t B,
                                fk_stream_t stream) {
  FK_CHECK_ARG(v && freqs && out && B > 0, "fk_timestep_proj: bad arguments");
  hipLaunchKernelGGL(timestep_proj_kernel, dim3(B), dim3(128), 0, (hipStream_t)stream, v, v_is_fp32, freqs,
                     (bf16_t*)out, B);
  FK_CHECK_LAUNCH("fk_timestep_proj");
  return FK_OK;
}

extern "C" int fk_euler_step_bf16(void* x, int64_t x_batch_stride, const void* v, int64_t v_batch_stride,
                                  int32_t B, int32_t S_tgt, int32_t C, float dsigma, fk_stream_t stream) {
  FK_CHECK_ARG(x && v && B > 0 && S_tgt > 0 && C > 0 && C % 8 == 0, "fk_euler_step_bf16: bad sizes");
  FK_CHECK_ARG(x_batch_stride % 8 == 0 && v_batch_stride % 8 == 0 && ((uintptr_t)x % 16 == 0) &&
                   ((uintptr_t)v % 16 == 0), "fk_euler_step_bf16: alignment");
  // the reference multiplies a 0-dim fp32 tensor into a bf16 tensor: the scalar is first cast to bf16
  uint32_t u;
  __builtin_memcpy(&u, &dsigma, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  u &= 0xffff0000u;
  float dsig_bf;
  __builtin_memcpy(&dsig_bf, &u, 4);
  const int64_t nvec = (int64_t)S_tgt * C / 8;
  hipLaunchKernelGGL(euler_kernel, dim3(ew_grid(nvec), B), dim3(256), 0, (hipStream_t)stream, (bf16_t*)x,
                     x_batch_stride, (const bf16_t*)v, v_batch_stride, S_tgt, C, dsig_bf);
  FK_CHECK_LAUNCH("fk_euler_step_bf16");
  return FK_OK;
}

extern "C" int fk_transpose_bf16(const void* src, int64_t lds, int64_t src_batch_stride, void* dst,
                                 int64_t ldd, int64_t dst_batch_stride, int32_t R, int32_t C, int32_t batch,
                                 fk_stream_t stream) {
  FK_CHECK_ARG(src && dst && R > 0 && C > 0 && batch > 0, "fk_transpose_bf16: bad arguments");
  hipLaunchKernelGGL(transpose_kernel, dim3((C + 63) / 64, (R + 63) / 64, batch), dim3(256), 0,
                     (hipStream_t)stream, (const bf16_t*)src, lds, src_batch_stride, (bf16_t*)dst, ldd,
                     dst_batch_stride, R, C);
  FK_CHECK_LAUNCH("fk_transpose_bf16");
  return FK_OK;
}

static int softmax_rows_entry(const float* x, int64_t ldx, void* y, int64_t ldy, int64_t ps, int parts, int64_t rows,
                              int32_t n, fk_stream_t stream) {
  FK_CHECK_ARG(x && y && rows > 0 && n > 0 && n % 4 == 0 && n <= 32768, "fk_softmax_rows: n must be a multiple of 4, <= 32768");
  FK_CHECK_ARG(ldx % 4 == 0 && ldy % 4 == 0 && ps % 4 == 0 && ((uintptr_t)x % 16 == 0) && ((uintptr_t)y % 8 == 0),
               "fk_softmax_rows: alignment");
  const dim3 grid((unsigned)rows), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (n <= 1024) hipLaunchKernelGGL(softmax_rows_kernel<1>, grid, block, 0, s, x, (bf16_t*)y, ldx, ldy, n, ps, parts);
  else if (n <= 4096) hipLaunchKernelGGL(softmax_rows_kernel<4>, grid, block, 0, s, x, (bf16_t*)y, ldx, ldy, n, ps, parts);
  else if (n <= 16384) hipLaunchKernelGGL(softmax_rows_kernel<16>, grid, block, 0, s, x, (bf16_t*)y, ldx, ldy, n, ps, parts);
  else hipLaunchKernelGGL(softmax_rows_kernel<32>, grid, block, 0, s, x, (bf16_t*)y, ldx, ldy, n, ps, parts);
  FK_CHECK_LAUNCH("fk_softmax_rows");
  return FK_OK;
}

extern "C" int fk_softmax_rows(const float* x, int64_t ldx, void* y, int64_t ldy, int64_t rows, int32_t n,
                               fk_stream_t stream) {
  return softmax_rows_entry(x, ldx, y, ldy, 0, 1, rows, n, stream);
}

extern "C" int fk_softmax_rows_parts(const float* x, int64_t ldx, void* y, int64_t ldy, int64_t part_stride, int64_t rows,
                                     int32_t n, fk_stream_t stream) {
  FK_CHECK_ARG(part_stride >= n, "fk_softmax_rows_parts: part_stride < n");
  return softmax_rows_entry(x, ldx, y, ldy, part_stride, 3, rows, n, stream);
}
