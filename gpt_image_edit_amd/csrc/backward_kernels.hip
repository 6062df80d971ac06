// HBM-bound kernels of the MMDiT BACKWARD pass (reference: the autograd graph
// that `accelerator.backward(loss)` walks, train_denoiser.py:1172, through diffusers' FluxTransformerBlock /
// FluxSingleTransformerBlock).  Each kernel is the adjoint of one fused forward kernel of this library:
//
//   fk_ln_modulate_bwd_bf16   adjoint of fk_ln_modulate_bf16:  n = LN(x) (1 + scale_b) + shift_b
//                             -> dx (+= into the residual-stream gradient), dshift_b = sum_s dn, dscale_b = sum_s dn LN(x)
//   fk_gate_res_bwd_bf16      adjoint of the FK_EPI_GATE_RES epilogue:  out = res + gate_b * y
//                             -> dy = dout * gate_b, dgate_b = sum_s dout * y          (dres = dout, no kernel needed)
//   fk_gelu_bwd_bf16          adjoint of FK_EPI_GELU_TANH:  dh = df * gelu_tanh'(h)
//   fk_silu_bwd_bf16          adjoint of FK_EPI_SILU (denoise_projector):  dh = df * silu'(h)
//   fk_qkv_post_bwd_bf16      adjoint of fk_qkv_post_bf16 (RoPE, per-head RMSNorm with weight, head-major layout)
//                             -> d(raw q | k) into the [B, S, 3D] gradient buffer, d(norm weights)
//   fk_colsum_bf16            bias gradients: out[n] = sum_m x[m, n]
//   fk_rowdot_bf16            D[b, h, s] = sum_d dO * O, the softmax-backward row term of attention
//
// Arithmetic is fp32 on bf16 inputs, rounded to bf16 once per output tensor (torch's bf16 autograd rounds after every
// op; the parity tests bound the difference by the bf16-autograd round-off floor).  Reductions over tokens are two-stage
// with a FIXED order (per-workgroup partial rows in a caller-provided fp32 workspace, then one finalising kernel), so
// gradients are bit-identical from run to run.
#include "fk_common.h"

namespace {

constexpr int HD = 128;

// out[b * out_bs + c] (+)= sum over `nparts` partial rows of part[(b * nparts + p) * n + c], in a fixed order:
// block = 64 columns x 4 part-groups (group g sums parts g, g + 4, ...), then the four group sums in order.
__global__ __launch_bounds__(256) void finalize_partials_kernel(const float* part, float* out, int64_t out_bs, int nparts,
                                                                int n, int accumulate) {
  __shared__ float red[4][64];
  const int cl = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const int b = blockIdx.y;
  float s = 0.f;
  if (c < n) {
    // eight independent running sums per thread (parts grp + 4 (8 j + u)): the loads of one trip are in flight together
    // instead of one memory latency per part (measured 53 us per launch as a serial chain, 2.4 % of a training step);
    // combined in a fixed order, so the result stays bit-identical from run to run
    const float* p = part + ((int64_t)b * nparts) * n + c;
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int i = grp; i < nparts; i += 32) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int j = i + 4 * u;
        const float v = p[(int64_t)min(j, nparts - 1) * n];
        a[u] += j < nparts ? v : 0.f;
      }
    }
    s = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  }
  red[grp][cl] = s;
  __syncthreads();
  if (grp == 0 && c < n) {
    const float t = ((red[0][cl] + red[1][cl]) + red[2][cl]) + red[3][cl];
    float* o = out + (int64_t)b * out_bs + c;
    *o = accumulate ? *o + t : t;
  }
}

// ---- LN + modulate backward ----------------------------------------------------------------------------------------
// grid = (chunks, B); block = 256 threads = 4 waves, one wave per row at a time (row of D = 512 NV in registers).
template <int NV>
__global__ __launch_bounds__(256) void ln_modulate_bwd_kernel(const bf16_t* x, fk_rows xr, const bf16_t* dn, fk_rows dnr,
                                                              const bf16_t* scale, int64_t mod_bs, const bf16_t* dx_in,
                                                              fk_rows dxi, bf16_t* dx_out, fk_rows dxo, float* part,
                                                              int64_t rpb, float eps) {
  constexpr int D = NV * 512;
  __shared__ float red[2 * D];   // the workgroup's column partials (dshift | dscale), waves added one after the other
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int chunk = blockIdx.x, nchunks = gridDim.x, b = blockIdx.y;
  float gsc[NV][8];
  {
    const bf16_t* sc = scale + (int64_t)b * mod_bs + lane * 8;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const u32x4_t w = *(const u32x4_t*)(sc + i * 512);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        gsc[i][2 * e] = 1.0f + bf_lo(w[e]);
        gsc[i][2 * e + 1] = 1.0f + bf_hi(w[e]);
      }
    }
  }
  float a_sh[NV][8], a_sc[NV][8];
#pragma unroll
  for (int i = 0; i < NV; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) a_sh[i][e] = a_sc[i][e] = 0.f;

  for (int64_t r = (int64_t)chunk * 4 + wave; r < rpb; r += (int64_t)nchunks * 4) {
    const int64_t row = (int64_t)b * rpb + r;
    const bf16_t* xp = x + fk_row_offset(xr, row) + lane * 8;
    const bf16_t* gp = dn + fk_row_offset(dnr, row) + lane * 8;
    u32x4_t xw[NV], gw[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      xw[i] = *(const u32x4_t*)(xp + i * 512);
      gw[i] = *(const u32x4_t*)(gp + i * 512);
    }
    float v[NV][8], g[NV][8];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[i][2 * e] = bf_lo(xw[i][e]);
        v[i][2 * e + 1] = bf_hi(xw[i][e]);
        g[i][2 * e] = bf_lo(gw[i][e]);
        g[i][2 * e + 1] = bf_hi(gw[i][e]);
        sum += v[i][2 * e] + v[i][2 * e + 1];
      }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off);
    const float mean = sum * (1.0f / D);
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        v[i][e] -= mean;
        sq += v[i][e] * v[i][e];
      }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) sq += __shfl_xor(sq, off);
    const float rstd = rsqrtf(sq * (1.0f / D) + eps);
    // ln = v * rstd; dshift += dn; dscale += dn * ln; gl = dn * (1 + scale) is the gradient wrt ln
    float m1 = 0.f, m2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float ln = v[i][e] * rstd;
        a_sh[i][e] += g[i][e];
        a_sc[i][e] += g[i][e] * ln;
        const float gl = g[i][e] * gsc[i][e];
        g[i][e] = gl;
        v[i][e] = ln;
        m1 += gl;
        m2 += gl * ln;
      }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      m1 += __shfl_xor(m1, off);
      m2 += __shfl_xor(m2, off);
    }
    m1 *= (1.0f / D);
    m2 *= (1.0f / D);
    bf16_t* op = dx_out + fk_row_offset(dxo, row) + lane * 8;
    const bf16_t* ip = dx_in ? dx_in + fk_row_offset(dxi, row) + lane * 8 : nullptr;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      u32x4_t acc = {0u, 0u, 0u, 0u};
      if (ip) acc = *(const u32x4_t*)(ip + i * 512);
      u32x4_t ow;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float d0 = rstd * (g[i][2 * e] - m1 - v[i][2 * e] * m2) + bf_lo(acc[e]);
        const float d1 = rstd * (g[i][2 * e + 1] - m1 - v[i][2 * e + 1] * m2) + bf_hi(acc[e]);
        ow[e] = pack_bf2(d0, d1);
      }
      *(u32x4_t*)(op + i * 512) = ow;
    }
  }
  // column partials of the workgroup: the four waves accumulate into LDS in wave order (fixed order -> deterministic),
  // every lane touching only its own columns; one row [2][D] per workgroup goes to the workspace
  for (int wv = 0; wv < 4; ++wv) {
    if (wave == wv) {
#pragma unroll
      for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int c = i * 512 + lane * 8 + e;
          red[c] = wv == 0 ? a_sh[i][e] : red[c] + a_sh[i][e];
          red[D + c] = wv == 0 ? a_sc[i][e] : red[D + c] + a_sc[i][e];
        }
    }
    __syncthreads();
  }
  float* po = part + ((int64_t)b * nchunks + chunk) * (2 * D);
  for (int c = threadIdx.x * 4; c < 2 * D; c += 256 * 4) *(f32x4_t*)(po + c) = *(const f32x4_t*)(red + c);
}

// ---- gate * y + residual backward ----------------------------------------------------------------------------------
// dy = dout * gate_b (bf16), dgate_b[c] = sum_s dout * y.  grid = (chunks, B, N / 2048), 256 threads, 8 columns each.
__global__ __launch_bounds__(256) void gate_res_bwd_kernel(const bf16_t* dout, fk_rows dor, const bf16_t* y, fk_rows yr,
                                                           const bf16_t* gate, int64_t gate_bs, bf16_t* dy, fk_rows dyr,
                                                           float* part, int64_t rpb, int N) {
  const int col = blockIdx.z * 2048 + threadIdx.x * 8;
  const int chunk = blockIdx.x, nchunks = gridDim.x, b = blockIdx.y;
  if (col >= N) return;
  float gt[8], acc[8];
  {
    const u32x4_t w = *(const u32x4_t*)(gate + (int64_t)b * gate_bs + col);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      gt[2 * e] = bf_lo(w[e]);
      gt[2 * e + 1] = bf_hi(w[e]);
      acc[2 * e] = acc[2 * e + 1] = 0.f;
    }
  }
  for (int64_t r = chunk; r < rpb; r += nchunks) {
    const int64_t row = (int64_t)b * rpb + r;
    const u32x4_t dw = *(const u32x4_t*)(dout + fk_row_offset(dor, row) + col);
    const u32x4_t yw = *(const u32x4_t*)(y + fk_row_offset(yr, row) + col);
    u32x4_t ow;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float d0 = bf_lo(dw[e]), d1 = bf_hi(dw[e]);
      acc[2 * e] += d0 * bf_lo(yw[e]);
      acc[2 * e + 1] += d1 * bf_hi(yw[e]);
      ow[e] = pack_bf2(d0 * gt[2 * e], d1 * gt[2 * e + 1]);
    }
    *(u32x4_t*)(dy + fk_row_offset(dyr, row) + col) = ow;
  }
  float* po = part + ((int64_t)b * nchunks + chunk) * N + col;
#pragma unroll
  for (int e = 0; e < 8; ++e) po[e] = acc[e];
}

// ---- GELU(tanh) backward, elementwise ------------------------------------------------------------------------------
FK_DEV float gelu_tanh_grad(float x) {
  // y = x * sigmoid(2u), u = sqrt(2/pi) (x + 0.044715 x^3):  y' = sg + x sg (1 - sg) 2 u'
  const float k0 = 0.7978845608028654f, k1 = 0.044715f;
  const float x2 = x * x;
  const float u = k0 * x * fmaf(k1, x2, 1.0f);
  const float sg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.0f * 1.4426950408889634f * u));
  const float du = k0 * fmaf(3.0f * k1, x2, 1.0f);
  return sg + x * sg * (1.0f - sg) * 2.0f * du;
}
// y = x * sigmoid(x):  y' = sg (1 + x (1 - sg))
FK_DEV float silu_grad(float x) {
  const float sg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
  return sg * fmaf(x, 1.0f - sg, 1.0f);
}
template <bool SILU>
__global__ __launch_bounds__(256) void act_bwd_kernel(const bf16_t* h, const bf16_t* df, bf16_t* out, int64_t n8) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
    const u32x4_t hw = *(const u32x4_t*)(h + i * 8);
    const u32x4_t dw = *(const u32x4_t*)(df + i * 8);
    u32x4_t ow;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      ow[e] = SILU ? pack_bf2(bf_lo(dw[e]) * silu_grad(bf_lo(hw[e])), bf_hi(dw[e]) * silu_grad(bf_hi(hw[e])))
                   : pack_bf2(bf_lo(dw[e]) * gelu_tanh_grad(bf_lo(hw[e])), bf_hi(dw[e]) * gelu_tanh_grad(bf_hi(hw[e])));
    *(u32x4_t*)(out + i * 8) = ow;
  }
}

// ---- q / k post-processing backward --------------------------------------------------------------------------------
// grid = (ceil(S / 64), H, B), 256 threads: thread = (token row r0 + 16 i, 8-element chunk of the head).
// dq / dk: [B, H, S, 128] (gradients of the RoPE'd, normalised heads); qkv: the raw projection [B, S, 3 H 128];
// dqkv: gradient of the raw projection, q and k thirds written here (the v third comes from the attention backward).
// part: [blocks][which 2][stream 2][128] partial sums of d(norm weight).
__global__ __launch_bounds__(256) void qkv_post_bwd_kernel(const bf16_t* dq, const bf16_t* dk, const bf16_t* qkv, bf16_t* dqkv,
                                                           const bf16_t* wq_img, const bf16_t* wk_img, const bf16_t* wq_txt,
                                                           const bf16_t* wk_txt, const float* cosT, const float* sinT,
                                                           float* part, int B, int S, int S_txt, int H, float eps) {
  __shared__ float red[16][2][2][HD];   // [token row group][which][stream][d]
  const int tid = threadIdx.x;
  const int s0 = blockIdx.x * 64, h = blockIdx.y, b = blockIdx.z;
  const int D3 = 3 * H * HD;
  const int chunk = tid & 15, r0 = tid >> 4;
  float dw[2][2][8];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int e = 0; e < 8; ++e) dw[a][c][e] = 0.f;
#pragma unroll
  for (int which = 0; which < 2; ++which) {
    const bf16_t* dsrc = which == 0 ? dq : dk;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int s = s0 + r0 + 16 * i;
      const bool valid = s < S;   // uniform across the 16 lanes of a row
      const int sc = valid ? s : S - 1;
      const u32x4_t xw = *(const u32x4_t*)(qkv + ((int64_t)b * S + sc) * D3 + which * H * HD + h * HD + chunk * 8);
      const u32x4_t gw = *(const u32x4_t*)(dsrc + (((int64_t)b * H + h) * S + sc) * HD + chunk * 8);
      const bool txt = sc < S_txt;
      const bf16_t* wsel = txt ? (which == 0 ? wq_txt : wk_txt) : (which == 0 ? wq_img : wk_img);
      const u32x4_t ww = *(const u32x4_t*)(wsel + chunk * 8);
      const f32x4_t c0 = *(const f32x4_t*)(cosT + (int64_t)sc * HD + chunk * 8);
      const f32x4_t c1 = *(const f32x4_t*)(cosT + (int64_t)sc * HD + chunk * 8 + 4);
      const f32x4_t n0 = *(const f32x4_t*)(sinT + (int64_t)sc * HD + chunk * 8);
      const f32x4_t n1 = *(const f32x4_t*)(sinT + (int64_t)sc * HD + chunk * 8 + 4);
      const float cs[8] = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
      const float sn[8] = {n0[0], n0[1], n0[2], n0[3], n1[0], n1[1], n1[2], n1[3]};
      float xv[8], gy[8], ss = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        xv[2 * e] = bf_lo(xw[e]);
        xv[2 * e + 1] = bf_hi(xw[e]);
        ss += xv[2 * e] * xv[2 * e] + xv[2 * e + 1] * xv[2 * e + 1];
        // RoPE adjoint: forward o0 = re c0 - im s0, o1 = im c1 + re s1
        const float g0 = bf_lo(gw[e]), g1 = bf_hi(gw[e]);
        gy[2 * e] = g0 * cs[2 * e] + g1 * sn[2 * e + 1];
        gy[2 * e + 1] = g1 * cs[2 * e + 1] - g0 * sn[2 * e];
      }
#pragma unroll
      for (int off = 8; off >= 1; off >>= 1) ss += __shfl_xor(ss, off);
      const float rs = rsqrtf(ss * (1.0f / HD) + eps);
      // y = (x rs) w:  dw += gy * (x rs);  g = gy * w;  dx = rs (g - xhat mean(g xhat))
      float dot = 0.f, gg[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float xh0 = xv[2 * e] * rs, xh1 = xv[2 * e + 1] * rs;
        if (valid) {
          dw[which][txt ? 1 : 0][2 * e] += gy[2 * e] * xh0;
          dw[which][txt ? 1 : 0][2 * e + 1] += gy[2 * e + 1] * xh1;
        }
        gg[2 * e] = gy[2 * e] * bf_lo(ww[e]);
        gg[2 * e + 1] = gy[2 * e + 1] * bf_hi(ww[e]);
        dot += gg[2 * e] * xh0 + gg[2 * e + 1] * xh1;
        xv[2 * e] = xh0;
        xv[2 * e + 1] = xh1;
      }
#pragma unroll
      for (int off = 8; off >= 1; off >>= 1) dot += __shfl_xor(dot, off);
      dot *= (1.0f / HD);
      if (valid) {
        u32x4_t ow;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          ow[e] = pack_bf2(rs * (gg[2 * e] - xv[2 * e] * dot), rs * (gg[2 * e + 1] - xv[2 * e + 1] * dot));
        *(u32x4_t*)(dqkv + ((int64_t)b * S + s) * D3 + which * H * HD + h * HD + chunk * 8) = ow;
      }
    }
  }
  // d(norm weight) partials of this block: 16 token-row groups summed in fixed order
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int e = 0; e < 8; ++e) red[r0][a][c][chunk * 8 + e] = dw[a][c][e];
  __syncthreads();
  const int blk = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
  for (int c = tid; c < 4 * HD; c += 256) {
    const int a = c / (2 * HD), st = (c / HD) & 1, d = c % HD;
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) sum += red[r][a][st][d];
    part[(int64_t)blk * (4 * HD) + c] = sum;
  }
}

// ---- column sums (bias gradients) ----------------------------------------------------------------------------------
// grid = (chunks, 1, ceil(N / 2048)); part[chunk][N]
__global__ __launch_bounds__(256) void colsum_kernel(const bf16_t* x, fk_rows xr, float* part, int64_t M, int N) {
  const int col = blockIdx.z * 2048 + threadIdx.x * 8;
  if (col >= N) return;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int64_t r = blockIdx.x; r < M; r += gridDim.x) {
    const u32x4_t w = *(const u32x4_t*)(x + fk_row_offset(xr, r) + col);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      acc[2 * e] += bf_lo(w[e]);
      acc[2 * e + 1] += bf_hi(w[e]);
    }
  }
  float* po = part + (int64_t)blockIdx.x * N + col;
#pragma unroll
  for (int e = 0; e < 8; ++e) po[e] = acc[e];
}

// ---- D[b, h, s] = sum_d a[b, s, h, d] * c[b, s, h, d] --------------------------------------------------------------
// a, c: [B, S, H * 128] views (row stride ld, batch stride bs); 16 lanes per (row, head).
__global__ __launch_bounds__(256) void rowdot_kernel(const bf16_t* a, int64_t a_ld, int64_t a_bs, const bf16_t* c, int64_t c_ld,
                                                     int64_t c_bs, float* out, int B, int S, int H) {
  const int64_t unit = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);   // (b, s, h) flattened, h fastest
  const int chunk = threadIdx.x & 15;
  const int64_t total = (int64_t)B * S * H;
  const int64_t u = unit < total ? unit : total - 1;
  const int h = (int)(u % H);
  const int64_t bs_ = u / H;
  const int s = (int)(bs_ % S), b = (int)(bs_ / S);
  const u32x4_t aw = *(const u32x4_t*)(a + (int64_t)b * a_bs + (int64_t)s * a_ld + h * HD + chunk * 8);
  const u32x4_t cw = *(const u32x4_t*)(c + (int64_t)b * c_bs + (int64_t)s * c_ld + h * HD + chunk * 8);
  float d = 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e) d += bf_lo(aw[e]) * bf_lo(cw[e]) + bf_hi(aw[e]) * bf_hi(cw[e]);
#pragma unroll
  for (int off = 8; off >= 1; off >>= 1) d += __shfl_xor(d, off);
  if (chunk == 0 && unit < total) out[((int64_t)b * H + h) * S + s] = d;
}

// ---- forward helpers of the recomputed (un-fused) block forward ----------------------------------------------------
// out = bf16(res + bf16(gate_b * y)): the FK_EPI_GATE_RES epilogue applied to an already stored y (same rounding points)
__global__ __launch_bounds__(256) void gate_res_fwd_kernel(const bf16_t* res, fk_rows rr, const bf16_t* y, fk_rows yr,
                                                           const bf16_t* gate, int64_t gate_bs, int64_t rpb, bf16_t* out,
                                                           fk_rows orr, int64_t M, int N) {
  const int cpr = N / 8;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < M * cpr; i += (int64_t)gridDim.x * 256) {
    const int64_t m = i / cpr;
    const int col = (int)(i - m * cpr) * 8;
    const int64_t b = m / rpb;
    const u32x4_t rw = *(const u32x4_t*)(res + fk_row_offset(rr, m) + col);
    const u32x4_t yw = *(const u32x4_t*)(y + fk_row_offset(yr, m) + col);
    const u32x4_t gw = *(const u32x4_t*)(gate + b * gate_bs + col);
    u32x4_t ow;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float y0 = bf_lo(yw[e]) * bf_lo(gw[e]), y1 = bf_hi(yw[e]) * bf_hi(gw[e]);
      round_bf_pair(y0, y1);
      ow[e] = pack_bf2(bf_lo(rw[e]) + y0, bf_hi(rw[e]) + y1);
    }
    *(u32x4_t*)(out + fk_row_offset(orr, m) + col) = ow;
  }
}
// y = bf16(gelu_tanh(x)), same function as the FK_EPI_GELU_TANH epilogue
__global__ __launch_bounds__(256) void gelu_fwd_kernel(const bf16_t* x, fk_rows xr, bf16_t* y, fk_rows yr, int64_t M, int N) {
  const int cpr = N / 8;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < M * cpr; i += (int64_t)gridDim.x * 256) {
    const int64_t m = i / cpr;
    const int col = (int)(i - m * cpr) * 8;
    const u32x4_t xw = *(const u32x4_t*)(x + fk_row_offset(xr, m) + col);
    u32x4_t ow;
#pragma unroll
    for (int e = 0; e < 4; ++e) ow[e] = pack_bf2(gelu_tanh_f(bf_lo(xw[e])), gelu_tanh_f(bf_hi(xw[e])));
    *(u32x4_t*)(y + fk_row_offset(yr, m) + col) = ow;
  }
}
// dst[c, ld-padded] (bf16) = src[r, c] (fp32), r < R (<= ld): the [B, n] fp32 modulation gradients as the K-padded
// operand of the weight-gradient GEMM; columns r >= R are zeroed.
__global__ __launch_bounds__(256) void f32_to_bf16_t_kernel(const float* src, int64_t src_ld, bf16_t* dst, int ld, int R, int C) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)C * ld) return;
  const int c = (int)(i / ld), r = (int)(i - (int64_t)c * ld);
  dst[i] = r < R ? f2bf(src[(int64_t)r * src_ld + c]) : (bf16_t)0;
}

int pick_chunks(int64_t rows, int per_iter, int max_chunks = 64) {
  int64_t c = (rows + per_iter - 1) / per_iter;
  if (c > max_chunks) c = max_chunks;
  if (c < 1) c = 1;
  return (int)c;
}

}  // namespace

#define FK_ALIGNED16(p) (((uintptr_t)(p) % 16) == 0)

extern "C" int64_t fk_bwd_ws_floats(void) { return (int64_t)1 << 22; }   // 16 MiB of fp32 partials covers every kernel here

extern "C" int fk_ln_modulate_bwd_bf16(const void* x, fk_rows xr, const void* dn, fk_rows dnr, const void* scale,
                                       int64_t mod_batch_stride, int64_t rows_per_batch, const void* dx_in, fk_rows dxi,
                                       void* dx_out, fk_rows dxo, float* dshift, float* dscale, int64_t dmod_batch_stride,
                                       float* ws, int32_t B, int32_t D, float eps, fk_stream_t stream_) {
  FK_CHECK_ARG(x && dn && scale && dx_out && dshift && dscale && ws, "fk_ln_modulate_bwd_bf16: null pointer");
  FK_CHECK_ARG(B > 0 && rows_per_batch > 0, "fk_ln_modulate_bwd_bf16: bad B / rows per batch");
  FK_CHECK_ARG(FK_ALIGNED16(x) && FK_ALIGNED16(dn) && FK_ALIGNED16(scale) && FK_ALIGNED16(dx_out) && FK_ALIGNED16(dx_in),
               "fk_ln_modulate_bwd_bf16: pointers must be 16-byte aligned");
  FK_CHECK_ARG(xr.ld % 8 == 0 && dnr.ld % 8 == 0 && dxo.ld % 8 == 0 && mod_batch_stride % 8 == 0 && (!dx_in || dxi.ld % 8 == 0),
               "fk_ln_modulate_bwd_bf16: strides must be multiples of 8 elements");
  hipStream_t stream = (hipStream_t)stream_;
  int chunks = pick_chunks(rows_per_batch, 4 * 4, 512);
  while (chunks > 1 && (int64_t)B * chunks * 2 * D > fk_bwd_ws_floats()) chunks /= 2;
  FK_CHECK_ARG((int64_t)B * chunks * 2 * D <= fk_bwd_ws_floats(), "fk_ln_modulate_bwd_bf16: workspace too small");
  const dim3 grid(chunks, B), block(256);
#define FK_LNB_CASE(NV)                                                                                                   \
  case NV * 512:                                                                                                          \
    hipLaunchKernelGGL(ln_modulate_bwd_kernel<NV>, grid, block, 0, stream, (const bf16_t*)x, xr, (const bf16_t*)dn, dnr,   \
                       (const bf16_t*)scale, mod_batch_stride, (const bf16_t*)dx_in, dxi, (bf16_t*)dx_out, dxo, ws,        \
                       rows_per_batch, eps);                                                                               \
    break;
  switch (D) {
    FK_LNB_CASE(1)
    FK_LNB_CASE(6)
    default:
      fk_set_error("fk_ln_modulate_bwd_bf16: D=%d unsupported (512, 3072)", D);
      return FK_EUNSUPPORTED;
  }
#undef FK_LNB_CASE
  FK_CHECK_LAUNCH("fk_ln_modulate_bwd_bf16");
  // partial rows are [b][chunk][2][D]: dshift = first D of each, dscale = second D
  const dim3 fgrid((2 * D + 63) / 64, B);
  // one pass writes both: treat the 2D-wide row as one vector when dscale follows dshift at +D ...
  if (dscale == dshift + D) {
    hipLaunchKernelGGL(finalize_partials_kernel, fgrid, block, 0, stream, ws, dshift, dmod_batch_stride, chunks, 2 * D, 0);
  } else {
    fk_set_error("fk_ln_modulate_bwd_bf16: dscale must be dshift + D (the (shift, scale) chunk pair of the modulation vector)");
    return FK_EINVAL;
  }
  FK_CHECK_LAUNCH("fk_ln_modulate_bwd_bf16 (finalize)");
  return FK_OK;
}

extern "C" int fk_gate_res_bwd_bf16(const void* dout, fk_rows dor, const void* y, fk_rows yr, const void* gate,
                                    int64_t gate_batch_stride, int64_t rows_per_batch, void* dy, fk_rows dyr, float* dgate,
                                    int64_t dgate_batch_stride, float* ws, int32_t B, int32_t N, fk_stream_t stream_) {
  FK_CHECK_ARG(dout && y && gate && dy && dgate && ws, "fk_gate_res_bwd_bf16: null pointer");
  FK_CHECK_ARG(B > 0 && rows_per_batch > 0 && N > 0 && N % 8 == 0, "fk_gate_res_bwd_bf16: bad sizes");
  FK_CHECK_ARG(FK_ALIGNED16(dout) && FK_ALIGNED16(y) && FK_ALIGNED16(gate) && FK_ALIGNED16(dy) && dor.ld % 8 == 0 &&
                   yr.ld % 8 == 0 && dyr.ld % 8 == 0 && gate_batch_stride % 8 == 0,
               "fk_gate_res_bwd_bf16: 16-byte alignment");
  hipStream_t stream = (hipStream_t)stream_;
  int chunks = pick_chunks(rows_per_batch, 16, 256);
  while (chunks > 1 && (int64_t)B * chunks * N > fk_bwd_ws_floats()) chunks /= 2;
  FK_CHECK_ARG((int64_t)B * chunks * N <= fk_bwd_ws_floats(), "fk_gate_res_bwd_bf16: workspace too small");
  hipLaunchKernelGGL(gate_res_bwd_kernel, dim3(chunks, B, (N + 2047) / 2048), dim3(256), 0, stream, (const bf16_t*)dout, dor,
                     (const bf16_t*)y, yr, (const bf16_t*)gate, gate_batch_stride, (bf16_t*)dy, dyr, ws, rows_per_batch, N);
  FK_CHECK_LAUNCH("fk_gate_res_bwd_bf16");
  hipLaunchKernelGGL(finalize_partials_kernel, dim3((N + 63) / 64, B), dim3(256), 0, stream, ws, dgate, dgate_batch_stride,
                     chunks, N, 0);
  FK_CHECK_LAUNCH("fk_gate_res_bwd_bf16 (finalize)");
  return FK_OK;
}

template <bool SILU>
static int act_bwd(const char* name, const void* h, const void* df, void* out, int64_t n, fk_stream_t stream_) {
  FK_CHECK_ARG(h && df && out && n > 0 && n % 8 == 0, "%s: bad arguments", name);
  FK_CHECK_ARG(FK_ALIGNED16(h) && FK_ALIGNED16(df) && FK_ALIGNED16(out), "%s: 16-byte alignment", name);
  const int64_t n8 = n / 8;
  const int blocks = (int)((n8 + 255) / 256 > 65536 ? 65536 : (n8 + 255) / 256);
  hipLaunchKernelGGL(act_bwd_kernel<SILU>, dim3(blocks), dim3(256), 0, (hipStream_t)stream_, (const bf16_t*)h,
                     (const bf16_t*)df, (bf16_t*)out, n8);
  FK_CHECK_LAUNCH(name);
  return FK_OK;
}
extern "C" int fk_gelu_bwd_bf16(const void* h, const void* df, void* out, int64_t n, fk_stream_t stream_) {
  return act_bwd<false>("fk_gelu_bwd_bf16", h, df, out, n, stream_);
}
extern "C" int fk_silu_bwd_bf16(const void* h, const void* df, void* out, int64_t n, fk_stream_t stream_) {
  return act_bwd<true>("fk_silu_bwd_bf16", h, df, out, n, stream_);
}

extern "C" int fk_qkv_post_bwd_bf16(const void* dq, const void* dk, const void* qkv, void* dqkv, const void* wq_img,
                                    const void* wk_img, const void* wq_txt, const void* wk_txt, const float* cos,
                                    const float* sin, float* dw, float* ws, int32_t B, int32_t S, int32_t S_txt, int32_t H,
                                    float eps, fk_stream_t stream_) {
  FK_CHECK_ARG(dq && dk && qkv && dqkv && wq_img && wk_img && cos && sin && dw && ws, "fk_qkv_post_bwd_bf16: null pointer");
  FK_CHECK_ARG(S_txt == 0 || (wq_txt && wk_txt), "fk_qkv_post_bwd_bf16: text-stream norm weights missing");
  FK_CHECK_ARG(B > 0 && S > 0 && H > 0 && S_txt >= 0 && S_txt <= S, "fk_qkv_post_bwd_bf16: bad sizes");
  if (!wq_txt) { wq_txt = wq_img; wk_txt = wk_img; }
  const dim3 grid((S + 63) / 64, H, B);
  const int nblk = grid.x * grid.y * grid.z;
  FK_CHECK_ARG((int64_t)nblk * 4 * HD <= fk_bwd_ws_floats(), "fk_qkv_post_bwd_bf16: workspace too small");
  hipStream_t stream = (hipStream_t)stream_;
  hipLaunchKernelGGL(qkv_post_bwd_kernel, grid, dim3(256), 0, stream, (const bf16_t*)dq, (const bf16_t*)dk, (const bf16_t*)qkv,
                     (bf16_t*)dqkv, (const bf16_t*)wq_img, (const bf16_t*)wk_img, (const bf16_t*)wq_txt, (const bf16_t*)wk_txt,
                     cos, sin, ws, B, S, S_txt, H, eps);
  FK_CHECK_LAUNCH("fk_qkv_post_bwd_bf16");
  // dw: [which 2][stream 2 (0 = image, 1 = text)][128]
  hipLaunchKernelGGL(finalize_partials_kernel, dim3(4 * HD / 64, 1), dim3(256), 0, stream, ws, dw, 0, nblk, 4 * HD, 0);
  FK_CHECK_LAUNCH("fk_qkv_post_bwd_bf16 (finalize)");
  return FK_OK;
}

extern "C" int fk_gate_res_fwd_bf16(const void* res, fk_rows rr, const void* y, fk_rows yr, const void* gate,
                                    int64_t gate_batch_stride, int64_t rows_per_batch, void* out, fk_rows orr, int64_t M,
                                    int32_t N, fk_stream_t stream_) {
  FK_CHECK_ARG(res && y && gate && out && M > 0 && N > 0 && N % 8 == 0 && rows_per_batch > 0, "fk_gate_res_fwd_bf16: bad arguments");
  FK_CHECK_ARG(FK_ALIGNED16(res) && FK_ALIGNED16(y) && FK_ALIGNED16(gate) && FK_ALIGNED16(out) && rr.ld % 8 == 0 &&
                   yr.ld % 8 == 0 && orr.ld % 8 == 0 && gate_batch_stride % 8 == 0,
               "fk_gate_res_fwd_bf16: 16-byte alignment");
  const int64_t n = M * (N / 8);
  const int blocks = (int)((n + 255) / 256 > 65536 ? 65536 : (n + 255) / 256);
  hipLaunchKernelGGL(gate_res_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream_, (const bf16_t*)res, rr,
                     (const bf16_t*)y, yr, (const bf16_t*)gate, gate_batch_stride, rows_per_batch, (bf16_t*)out, orr, M, N);
  FK_CHECK_LAUNCH("fk_gate_res_fwd_bf16");
  return FK_OK;
}

extern "C" int fk_gelu_tanh_bf16(const void* x, fk_rows xr, void* y, fk_rows yr, int64_t M, int32_t N, fk_stream_t stream_) {
  FK_CHECK_ARG(x && y && M > 0 && N > 0 && N % 8 == 0, "fk_gelu_tanh_bf16: bad arguments");
  FK_CHECK_ARG(FK_ALIGNED16(x) && FK_ALIGNED16(y) && xr.ld % 8 == 0 && yr.ld % 8 == 0, "fk_gelu_tanh_bf16: 16-byte alignment");
  const int64_t n = M * (N / 8);
  const int blocks = (int)((n + 255) / 256 > 65536 ? 65536 : (n + 255) / 256);
  hipLaunchKernelGGL(gelu_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream_, (const bf16_t*)x, xr, (bf16_t*)y, yr, M, N);
  FK_CHECK_LAUNCH("fk_gelu_tanh_bf16");
  return FK_OK;
}

extern "C" int fk_f32_to_bf16_transposed(const float* src, int64_t src_ld, void* dst, int32_t dst_ld, int32_t R, int32_t C,
                                         fk_stream_t stream_) {
  FK_CHECK_ARG(src && dst && R > 0 && C > 0 && dst_ld >= R, "fk_f32_to_bf16_transposed: bad arguments");
  const int64_t n = (int64_t)C * dst_ld;
  hipLaunchKernelGGL(f32_to_bf16_t_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, src, src_ld,
                     (bf16_t*)dst, dst_ld, R, C);
  FK_CHECK_LAUNCH("fk_f32_to_bf16_transposed");
  return FK_OK;
}

extern "C" int fk_colsum_bf16(const void* x, fk_rows xr, int64_t M, int32_t N, float* out, float* ws, fk_stream_t stream_) {
  FK_CHECK_ARG(x && out && ws && M > 0 && N > 0 && N % 8 == 0, "fk_colsum_bf16: bad arguments");
  FK_CHECK_ARG(FK_ALIGNED16(x) && xr.ld % 8 == 0, "fk_colsum_bf16: 16-byte alignment");
  int chunks = pick_chunks(M, 32, 256);
  while (chunks > 1 && (int64_t)chunks * N > fk_bwd_ws_floats()) chunks /= 2;
  FK_CHECK_ARG((int64_t)chunks * N <= fk_bwd_ws_floats(), "fk_colsum_bf16: workspace too small");
  hipStream_t stream = (hipStream_t)stream_;
  hipLaunchKernelGGL(colsum_kernel, dim3(chunks, 1, (N + 2047) / 2048), dim3(256), 0, stream, (const bf16_t*)x, xr, ws, M, N);
  FK_CHECK_LAUNCH("fk_colsum_bf16");
  hipLaunchKernelGGL(finalize_partials_kernel, dim3((N + 63) / 64, 1), dim3(256), 0, stream, ws, out, 0, chunks, N, 0);
  FK_CHECK_LAUNCH("fk_colsum_bf16 (finalize)");
  return FK_OK;
}

extern "C" int fk_rowdot_bf16(const void* a, int64_t a_ld, int64_t a_batch_stride, const void* c, int64_t c_ld,
                              int64_t c_batch_stride, float* out, int32_t B, int32_t S, int32_t H, fk_stream_t stream_) {
  FK_CHECK_ARG(a && c && out && B > 0 && S > 0 && H > 0, "fk_rowdot_bf16: bad arguments");
  FK_CHECK_ARG(FK_ALIGNED16(a) && FK_ALIGNED16(c) && a_ld % 8 == 0 && c_ld % 8 == 0 && a_batch_stride % 8 == 0 &&
                   c_batch_stride % 8 == 0,
               "fk_rowdot_bf16: 16-byte alignment");
  const int64_t total = (int64_t)B * S * H;
  hipLaunchKernelGGL(rowdot_kernel, dim3((unsigned)((total + 15) / 16)), dim3(256), 0, (hipStream_t)stream_, (const bf16_t*)a,
                     a_ld, a_batch_stride, (const bf16_t*)c, c_ld, c_batch_stride, out, B, S, H);
  FK_CHECK_LAUNCH("fk_rowdot_bf16");
  return FK_OK;
}
