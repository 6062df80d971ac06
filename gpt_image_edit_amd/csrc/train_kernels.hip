// HBM-bound kernels of the denoiser's optimisation step (reference train_denoiser.py:935-1181).
// The MMDiT backward is NOT here yet: these are the pieces around it -- the noisy-input mix fused with the 2x2 token
// packing, the flow-matching loss fused with its gradient, the gradient norm, and the AdamW update -- each one pass
// over its operands.  Reductions are two-stage with a fixed order (per-block partials, then one block), so results
// are bit-identical from run to run.
#include "fk_common.h"

namespace {

constexpr int RED_THREADS = 256;

// wave (shuffle) then block (LDS, fixed order) sum; valid in thread 0
FK_DEV double block_sum(double v, double* sh) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  double t = 0.0;
  if (threadIdx.x == 0)
    for (int i = 0; i < RED_THREADS / 64; ++i) t += sh[i];
  __syncthreads();
  return t;
}

// token (b, s, j) of the packed [B, (h/2)(w/2), 4C] layout <-> latent element (b, c, 2r + ph, 2q + pw):
// s = r * (w/2) + q, j = c * 4 + ph * 2 + pw  (FluxKontextPipeline._pack_latents, flux_pipeline.py:576-581)
FK_DEV int64_t latent_index(int64_t tok, int C, int h, int w) {
  const int j = (int)(tok % (4 * C));
  const int64_t bs = tok / (4 * C);
  const int hw2 = (h / 2) * (w / 2);
  const int s = (int)(bs % hw2);
  const int64_t b = bs / hw2;
  const int c = j >> 2, ph = (j >> 1) & 1, pw = j & 1;
  const int r = s / (w / 2), q = s - r * (w / 2);
  return ((b * C + c) * h + 2 * r + ph) * (int64_t)w + 2 * q + pw;
}

// noisy = (1 - sigma_b) * x + sigma_b * noise in fp32 (train_denoiser.py:994), cast to bf16 and packed
// (prepare_latents(latents=noisy, dtype=bf16) + _pack_latents, :1009-1027)
__global__ __launch_bounds__(256) void flow_mix_pack_kernel(const float* x, const float* noise, const float* sigma,
                                                            bf16_t* tokens, int64_t tok_bs, int C, int h, int w,
                                                            int64_t n_per_batch, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t b = i / n_per_batch, r = i - b * n_per_batch;
    const int64_t src = latent_index(i, C, h, w);
    const float sg = sigma[b];
    const float v = __fadd_rn(__fmul_rn(1.0f - sg, x[src]), __fmul_rn(sg, noise[src]));
    tokens[b * tok_bs + r] = f2bf(v);
  }
}

// d = float(pred) - (noise - x); weighting = weight_b * area[b, y, x] * mask[b, y, x] (every factor optional, multiplied in the
// reference's order, train_denoiser.py:1106-1149); partial[block] = sum of weighting * d^2;
// grad = bf16(2 * weighting * d / denominator), denominator = mask_sum[0] * C (:1163-1165) or the element count (loss.mean())
__global__ __launch_bounds__(RED_THREADS) void flow_loss_kernel(const bf16_t* pred, int64_t pred_bs, const float* x,
                                                                const float* noise, const float* weight, const float* area,
                                                                const float* mask, const float* mask_sum, bf16_t* grad,
                                                                int64_t grad_bs, double* partial, int C, int h, int w,
                                                                int64_t n_per_batch, int64_t total, float inv_count) {
  __shared__ double sh[RED_THREADS / 64];
  double acc = 0.0;
  const int hw = h * w;
  const float inv = mask_sum ? 1.0f / (mask_sum[0] * (float)C) : inv_count;
  for (int64_t i = (int64_t)blockIdx.x * RED_THREADS + threadIdx.x; i < total; i += (int64_t)gridDim.x * RED_THREADS) {
    const int64_t b = i / n_per_batch, r = i - b * n_per_batch;
    const int64_t src = latent_index(i, C, h, w);
    float wt = weight ? weight[b] : 1.0f;
    if (area || mask) {
      const int64_t pix = b * hw + src % hw;
      if (area) wt = __fmul_rn(wt, area[pix]);
      if (mask) wt = __fmul_rn(wt, mask[pix]);
    }
    const float d = bf2f(pred[b * pred_bs + r]) - (noise[src] - x[src]);
    acc += (double)__fmul_rn(wt, __fmul_rn(d, d));
    if (grad) grad[b * grad_bs + r] = f2bf(2.0f * wt * d * inv);
  }
  const double t = block_sum(acc, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

__global__ __launch_bounds__(RED_THREADS) void sumsq_kernel(const void* g, int g_is_bf16, int64_t n, double* partial) {
  __shared__ double sh[RED_THREADS / 64];
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * RED_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * RED_THREADS) {
    const float v = g_is_bf16 ? bf2f(((const bf16_t*)g)[i]) : ((const float*)g)[i];
    acc += (double)v * (double)v;
  }
  const double t = block_sum(acc, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

// out[0] (+)= scale * sum(partial[0..n))  -- one block, fixed order.  denom (device scalar, optional): scale = 1 / (denom[0] * denom_mul)
__global__ __launch_bounds__(RED_THREADS) void finish_sum_kernel(const double* partial, int n, double scale, int accumulate,
                                                                 double* out, const float* denom = nullptr, double denom_mul = 1.0) {
  __shared__ double sh[RED_THREADS / 64];
  double acc = 0.0;
  for (int i = threadIdx.x; i < n; i += RED_THREADS) acc += partial[i];
  const double t = block_sum(acc, sh);
  if (denom) scale = 1.0 / ((double)denom[0] * denom_mul);
  if (threadIdx.x == 0) out[0] = (accumulate ? out[0] : 0.0) + scale * t;
}

// torch.optim.AdamW (single-tensor form) on fp32 masters with the clipping coefficient folded into the gradient read:
//   g *= min(1, max_norm / (sqrt(sumsq) + 1e-6));  p *= decay (= 1 - lr*wd);  m = lerp(m, g, 1-b1);
//   v = b2*v + (1-b2)*g*g;  p -= step_size (= lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps);  bf16 copy of p
__global__ __launch_bounds__(256) void adamw_kernel(float* master, bf16_t* param_bf16, const void* grad, int g_is_bf16,
                                                    float* m, float* v, const double* grad_sumsq, float max_norm,
                                                    float grad_scale, float decay, float b1, float b2, float eps,
                                                    float step_size, float bc2_sqrt, int64_t n) {
  // grad_scale: the stored gradients are grad_scale^-1 times the gradient to apply (sums over the data-parallel ranks:
  // grad_scale = 1 / world) -- folded into the clipping coefficient, so no separate pass divides them
  float coef = grad_scale;
  if (grad_sumsq) {
    const float total = __fmul_rn((float)sqrt(grad_sumsq[0]), grad_scale);
    coef = __fmul_rn(fminf(max_norm / (total + 1e-6f), 1.0f), grad_scale);
  }
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    float g = g_is_bf16 ? bf2f(((const bf16_t*)grad)[i]) : ((const float*)grad)[i];
    g = __fmul_rn(g, coef);
    float p = __fmul_rn(master[i], decay);
    const float mi = m[i], vi = v[i];
    const float mn = __fadd_rn(mi, __fmul_rn(1.0f - b1, __fsub_rn(g, mi)));            // lerp_
    const float vn = __fadd_rn(__fmul_rn(vi, b2), __fmul_rn(__fmul_rn(1.0f - b2, g), g));   // mul_ + addcmul_
    const float denom = __fadd_rn(__fdiv_rn(__fsqrt_rn(vn), bc2_sqrt), eps);
    p = __fadd_rn(p, __fmul_rn(-step_size, __fdiv_rn(mn, denom)));                        // addcdiv_
    master[i] = p; m[i] = mn; v[i] = vn;
    if (param_bf16) param_bf16[i] = f2bf(p);
  }
}

int grid_for(int64_t n, int threads) {
  const int64_t want = (n + threads - 1) / threads;
  return (int)(want < 1 ? 1 : (want > 2048 ? 2048 : want));
}

}  // namespace

extern "C" int64_t fk_reduce_ws_doubles(void) { return 2048 + 8; }

extern "C" int fk_flow_noisy_tokens_bf16(const float* x, const float* noise, const float* sigma, void* tokens,
                                         int64_t tokens_batch_stride, int32_t B, int32_t C, int32_t h, int32_t w,
                                         fk_stream_t stream) {
  FK_CHECK_ARG(x && noise && sigma && tokens, "fk_flow_noisy_tokens_bf16: null pointer");
  FK_CHECK_ARG(B > 0 && C > 0 && h > 0 && w > 0 && h % 2 == 0 && w % 2 == 0, "fk_flow_noisy_tokens_bf16: bad sizes");
  const int64_t npb = (int64_t)C * h * w, total = npb * B;
  FK_CHECK_ARG(tokens_batch_stride >= npb, "fk_flow_noisy_tokens_bf16: batch stride smaller than one sample's tokens");
  hipLaunchKernelGGL(flow_mix_pack_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, x, noise, sigma,
                     (bf16_t*)tokens, tokens_batch_stride, C, h, w, npb, total);
  FK_CHECK_LAUNCH("fk_flow_noisy_tokens_bf16");
  return FK_OK;
}

extern "C" int fk_flow_loss_weighted_bf16(const void* pred, int64_t pred_batch_stride, const float* x, const float* noise,
                                          const float* weight, const float* area_weights, const float* weight_mask,
                                          const float* mask_sum, void* grad, int64_t grad_batch_stride, double* loss, double* ws,
                                          int32_t B, int32_t C, int32_t h, int32_t w, fk_stream_t stream) {
  FK_CHECK_ARG(pred && x && noise && loss && ws, "fk_flow_loss_bf16: null pointer");
  FK_CHECK_ARG(B > 0 && C > 0 && h > 0 && w > 0 && h % 2 == 0 && w % 2 == 0, "fk_flow_loss_bf16: bad sizes");
  FK_CHECK_ARG(!mask_sum || weight_mask, "fk_flow_loss_weighted_bf16: mask_sum (the weight_mask.sum() normaliser) without weight_mask");
  const int64_t npb = (int64_t)C * h * w, total = npb * B;
  FK_CHECK_ARG(pred_batch_stride >= npb && (!grad || grad_batch_stride >= npb), "fk_flow_loss_bf16: bad batch stride");
  const int blocks = grid_for(total, RED_THREADS);
  hipLaunchKernelGGL(flow_loss_kernel, dim3(blocks), dim3(RED_THREADS), 0, (hipStream_t)stream, (const bf16_t*)pred,
                     pred_batch_stride, x, noise, weight, area_weights, weight_mask, mask_sum, (bf16_t*)grad, grad_batch_stride, ws,
                     C, h, w, npb, total, 1.0f / (float)total);
  hipLaunchKernelGGL(finish_sum_kernel, dim3(1), dim3(RED_THREADS), 0, (hipStream_t)stream, (const double*)ws, blocks,
                     1.0 / (double)total, 0, loss, mask_sum, (double)C);
  FK_CHECK_LAUNCH("fk_flow_loss_bf16");
  return FK_OK;
}

extern "C" int fk_flow_loss_bf16(const void* pred, int64_t pred_batch_stride, const float* x, const float* noise,
                                 const float* weight, void* grad, int64_t grad_batch_stride, double* loss, double* ws,
                                 int32_t B, int32_t C, int32_t h, int32_t w, fk_stream_t stream) {
  return fk_flow_loss_weighted_bf16(pred, pred_batch_stride, x, noise, weight, nullptr, nullptr, nullptr, grad, grad_batch_stride,
                                    loss, ws, B, C, h, w, stream);
}

extern "C" int fk_sumsq(const void* g, int32_t g_is_bf16, int64_t n, int32_t accumulate, double* out, double* ws,
                        fk_stream_t stream) {
  FK_CHECK_ARG(g && out && ws && n > 0, "fk_sumsq: bad arguments");
  const int blocks = grid_for(n, RED_THREADS);
  hipLaunchKernelGGL(sumsq_kernel, dim3(blocks), dim3(RED_THREADS), 0, (hipStream_t)stream, g, g_is_bf16, n, ws);
  hipLaunchKernelGGL(finish_sum_kernel, dim3(1), dim3(RED_THREADS), 0, (hipStream_t)stream, (const double*)ws, blocks, 1.0,
                     accumulate, out);
  FK_CHECK_LAUNCH("fk_sumsq");
  return FK_OK;
}

extern "C" int fk_adamw_step_scaled(float* master, void* param_bf16, const void* grad, int32_t grad_is_bf16, float* exp_avg,
                                    float* exp_avg_sq, const double* grad_sumsq, float max_grad_norm, float grad_scale,
                                    float lr, float beta1, float beta2, float eps, float weight_decay, int32_t step,
                                    int64_t n, fk_stream_t stream) {
  FK_CHECK_ARG(master && grad && exp_avg && exp_avg_sq && n > 0 && step >= 1 && grad_scale > 0.f,
               "fk_adamw_step: bad arguments");
  // the scalar factors in double on the host, like torch's python-side arithmetic, then fp32 into the kernel
  const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  const float step_size = (float)((double)lr / bc1), bc2_sqrt = (float)sqrt(bc2);
  const float decay = (float)(1.0 - (double)lr * (double)weight_decay);
  hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, master, (bf16_t*)param_bf16,
                     grad, grad_is_bf16, exp_avg, exp_avg_sq, grad_sumsq, max_grad_norm, grad_scale, decay, beta1, beta2,
                     eps, step_size, bc2_sqrt, n);
  FK_CHECK_LAUNCH("fk_adamw_step");
  return FK_OK;
}

extern "C" int fk_adamw_step(float* master, void* param_bf16, const void* grad, int32_t grad_is_bf16, float* exp_avg,
                             float* exp_avg_sq, const double* grad_sumsq, float max_grad_norm, float lr, float beta1,
                             float beta2, float eps, float weight_decay, int32_t step, int64_t n, fk_stream_t stream) {
  return fk_adamw_step_scaled(master, param_bf16, grad, grad_is_bf16, exp_avg, exp_avg_sq, grad_sumsq, max_grad_norm, 1.0f,
                              lr, beta1, beta2, eps, weight_decay, step, n, stream);
}
