// Flash-style joint attention forward for the FLUX MMDiT blocks (K4 in SURVEY.md section 2.2):
//   O = softmax(Q K^T / sqrt(128)) V,   non-causal, no mask, head_dim = 128, bf16 in / bf16 out,
//   fp32 scores, fp32 online softmax, fp32 accumulation (= F.scaled_dot_product_attention as the
//   reference reaches it through diffusers' FluxAttnProcessor2_0).
//
// One workgroup = 8 waves = 256 query rows of one (batch, head); each wave owns 32 query rows and
// walks the keys in tiles of 64.  Both products are formed transposed so that the softmax axis is
// lane-local (cdna guide: "swapped QK^T"):
//   S^T[key][q] = mfma_32x32x16(A = K rows, B = Q rows)      -> lane (q = lane&31) holds 16 keys / block
//   O^T[d][q]  += mfma_32x32x16(A = V^T rows, B = P^T)       -> lane (q = lane&31) holds 64 d's
// so row max / row sum need ONE cross-lane exchange (lane ^ 32) and the O rescale is a per-lane scalar.
// The P operand is consumed in exactly the register order the first MFMA produced it: MFMA k-slot
// (h = lane>>5, j) is bound to key 16*step + 4h + (j&3) + 8*(j>>2) for BOTH operands, so no lane
// permutation of P is needed; V^T is fetched from LDS as two 8-byte pieces per operand instead.
//
// K tile [64 keys][128 d] and V^T tile [128 d][64 keys] are staged global -> VGPR -> LDS one tile
// ahead (two LDS stages, one barrier per tile).  LDS swizzles:
//   K  : 16-byte chunk ^= key & 15            (ds_read_b128, 256-byte rows: conflict-free)
//   V^T: 8-byte unit   ^= (d >> 1) & 15       (ds_read_b64, 128-byte rows: conflict-free)
// V arrives pre-transposed ([B,H,128,S_pad], zero padded) from fk_qkv_post_bf16.
#include "fk_common.h"

namespace {

constexpr int HD = 128;
constexpr int QBLK = 256;   // query rows per workgroup
constexpr int KVBLK = 64;   // keys per tile
constexpr int NTHREADS = 512;
constexpr int K_TILE_BYTES = KVBLK * HD * 2;   // 16 KiB
constexpr int STAGE_BYTES = 2 * K_TILE_BYTES;  // K + V^T
constexpr int SMEM_BYTES = 2 * STAGE_BYTES;    // 64 KiB

struct AttnParams {
  const bf16_t* q;
  const bf16_t* k;
  const bf16_t* vt;
  bf16_t* o;
  int B, H, S, S_pad;
  int64_t o_ld, o_bs;
  float scale_log2;  // scale * log2(e)
};

__global__ __launch_bounds__(NTHREADS, 2) void attention_fwd_kernel(const AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int ql = lane & 31;   // query row inside the wave / operand row
  const int hh = lane >> 5;   // half

  // XCD-aware block order: workgroups of one (b, h) -- which share K / V -- stay on one XCD's L2
  const int nqb = (p.S + QBLK - 1) / QBLK;
  const int nwg = gridDim.x;
  int t;
  {
    const int q8 = nwg >> 3, r8 = nwg & 7;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    t = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
  }
  const int qb = t % nqb;
  const int bh = t / nqb;
  const int b = bh / p.H, h = bh - b * p.H;

  const bf16_t* Kg = p.k + (int64_t)bh * p.S * HD;
  const bf16_t* Vg = p.vt + (int64_t)bh * HD * p.S_pad;

  // ---- Q operand fragments (B operand of S^T = K Q^T): lane holds Q[q][16kk + 8hh .. +8] ----------
  const int q_row = qb * QBLK + wave * 32 + ql;
  bf16x8_t qf[8];
  {
    const bf16_t* qp = p.q + ((int64_t)bh * p.S + min(q_row, p.S - 1)) * HD + 8 * hh;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) qf[kk] = *(const bf16x8_t*)(qp + 16 * kk);
  }

  // ---- staging addresses ---------------------------------------------------------------------------
  // K tile: 1024 16-byte chunks; thread -> chunk (row = tid/16 + 32 i, c = tid%16)
  const int k_row = tid >> 4, k_c = tid & 15;
  const int k_st = k_row * 256 + ((k_c ^ (k_row & 15)) << 4);  // + i*8192
  // V^T tile: 128 rows (d) x 8 chunks of 16 B (8 keys); thread -> (d = tid/8 + 64 i, c = tid%8)
  const int v_d = tid >> 3, v_c = tid & 7;
  const int v_s = (v_d >> 1) & 15;                              // (d + 64 i) >> 1 & 15 is the same for i = 0, 1
  const int v_st = K_TILE_BYTES + v_d * 128 + ((v_c ^ (v_s >> 1)) << 4);  // + i*8192
  const bool v_swap = v_s & 1;

  u32x4_t kreg[2], vreg[2];
  auto load_tile = [&](int kt) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int key = min(kt * KVBLK + k_row + 32 * i, p.S - 1);
      kreg[i] = *(const u32x4_t*)(Kg + (int64_t)key * HD + k_c * 8);
      vreg[i] = *(const u32x4_t*)(Vg + (int64_t)(v_d + 64 * i) * p.S_pad + kt * KVBLK + v_c * 8);
    }
  };
  auto store_tile = [&](int stage) {
    char* base = smem + stage * STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      *(u32x4_t*)(base + k_st + i * 8192) = kreg[i];
      u32x4_t v = vreg[i];
      if (v_swap) { u32x4_t w; w[0] = v[2]; w[1] = v[3]; w[2] = v[0]; w[3] = v[1]; v = w; }
      *(u32x4_t*)(base + v_st + i * 8192) = v;
    }
  };

  // ---- operand read addresses ------------------------------------------------------------------------
  // K operand (A of S^T): row key = 32 kb + ql, chunk 2kk + hh, swizzled by key & 15 (= ql & 15)
  const int k_rd = ql * 256;           // + kb*8192 + (((2kk + hh) ^ (ql & 15)) << 4)
  const int k_sw = ql & 15;
  // V^T operand (A of O^T): row d = 32 df + ql, units (4 st + hh) and (4 st + 2 + hh), swizzled by (d>>1)&15
  const int v_rd = K_TILE_BYTES + ql * 128;  // + df*4096 + ((u ^ v_sw) << 3)
  const int v_sw = (ql >> 1) & 15;

  f32x16_t o[4];
#pragma unroll
  for (int df = 0; df < 4; ++df)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[df][r] = 0.f;
  float m_run = -1.0e30f, l_run = 0.f;

  const int nkt = p.S_pad / KVBLK;
  load_tile(0);
  store_tile(0);
  __syncthreads();

  for (int kt = 0; kt < nkt; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nkt) load_tile(kt + 1);
    const char* sb = smem + cur * STAGE_BYTES;

    // ---- S^T = K Q^T for the two 32-key blocks -------------------------------------------------------
    f32x16_t s[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        const bf16x8_t kf = *(const bf16x8_t*)(sb + k_rd + kb * 8192 + (((2 * kk + hh) ^ k_sw) << 4));
        s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[kk], s[kb], 0, 0, 0);
      }
    }
    // ---- mask the key tail (last tile only) ---------------------------------------------------------
    if (kt == nkt - 1 && p.S_pad != p.S) {
      const int kbase = kt * KVBLK + 4 * hh;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kbase + 32 * kb + (r & 3) + 8 * (r >> 2);
          if (key >= p.S) s[kb][r] = -1.0e30f;
        }
    }
    // ---- online softmax (log2 domain) -----------------------------------------------------------------
    float mx = s[0][0];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kb][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(m_run, mx * p.scale_log2);
    const float alpha = exp2f(m_run - m_new);
    m_run = m_new;
    float psum = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pv = exp2f(fmaf(s[kb][r], p.scale_log2, -m_new));
        s[kb][r] = pv;
        psum += pv;
      }
    l_run = l_run * alpha + psum;
#pragma unroll
    for (int df = 0; df < 4; ++df)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[df][r] *= alpha;

    // ---- O^T += V^T P^T --------------------------------------------------------------------------------
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      const int kb = st >> 1, r0 = 8 * (st & 1);
      u32x4_t pw;
      pw[0] = pack_bf2(s[kb][r0 + 0], s[kb][r0 + 1]);
      pw[1] = pack_bf2(s[kb][r0 + 2], s[kb][r0 + 3]);
      pw[2] = pack_bf2(s[kb][r0 + 4], s[kb][r0 + 5]);
      pw[3] = pack_bf2(s[kb][r0 + 6], s[kb][r0 + 7]);
      const bf16x8_t pf = __builtin_bit_cast(bf16x8_t, pw);
      const int u0 = ((4 * st + hh) ^ v_sw) << 3;
      const int u1 = ((4 * st + 2 + hh) ^ v_sw) << 3;
#pragma unroll
      for (int df = 0; df < 4; ++df) {
        u32x4_t vw;
        const u32x2_t lo = *(const u32x2_t*)(sb + v_rd + df * 4096 + u0);
        const u32x2_t hi = *(const u32x2_t*)(sb + v_rd + df * 4096 + u1);
        vw[0] = lo[0]; vw[1] = lo[1]; vw[2] = hi[0]; vw[3] = hi[1];
        const bf16x8_t vf = __builtin_bit_cast(bf16x8_t, vw);
        o[df] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o[df], 0, 0, 0);
      }
    }

    if (kt + 1 < nkt) store_tile(cur ^ 1);
    __syncthreads();
  }

  // ---- finalize: O = O^T / l ; lane (q = ql) holds d = 32 df + 8 (r>>2) + 4 hh + (r&3) ---------------
  const float l_tot = l_run + __shfl_xor(l_run, 32);
  const float inv = 1.0f / l_tot;
  if (q_row < p.S) {
    bf16_t* op = p.o + (int64_t)b * p.o_bs + (int64_t)q_row * p.o_ld + h * HD + 4 * hh;
#pragma unroll
    for (int df = 0; df < 4; ++df)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        u32x2_t pk;
        pk[0] = pack_bf2(o[df][4 * g + 0] * inv, o[df][4 * g + 1] * inv);
        pk[1] = pack_bf2(o[df][4 * g + 2] * inv, o[df][4 * g + 3] * inv);
        *(u32x2_t*)(op + 32 * df + 8 * g) = pk;
      }
  }
}

}  // namespace

extern "C" int fk_attention_fwd_bf16(const void* q, const void* k, const void* vt, void* o, int32_t B,
                                     int32_t H, int32_t S, int32_t S_pad, int64_t o_ld,
                                     int64_t o_batch_stride, float scale, fk_stream_t stream_) {
  FK_CHECK_ARG(q && k && vt && o, "fk_attention_fwd_bf16: null pointer");
  FK_CHECK_ARG(B > 0 && H > 0 && S > 0, "fk_attention_fwd_bf16: bad B/H/S %d %d %d", B, H, S);
  FK_CHECK_ARG(S_pad % KVBLK == 0 && S_pad >= S && S_pad - S < KVBLK,
               "fk_attention_fwd_bf16: S_pad=%d must be S=%d rounded up to %d", S_pad, S, KVBLK);
  FK_CHECK_ARG(o_ld % 4 == 0 && o_batch_stride % 4 == 0 && ((uintptr_t)o % 8 == 0),
               "fk_attention_fwd_bf16: output must be 8-byte aligned (o_ld %% 4 == 0)");
  FK_CHECK_ARG(((uintptr_t)q % 16 == 0) && ((uintptr_t)k % 16 == 0) && ((uintptr_t)vt % 16 == 0),
               "fk_attention_fwd_bf16: q/k/vt must be 16-byte aligned");
  AttnParams p;
  p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.vt = (const bf16_t*)vt; p.o = (bf16_t*)o;
  p.B = B; p.H = H; p.S = S; p.S_pad = S_pad; p.o_ld = o_ld; p.o_bs = o_batch_stride;
  p.scale_log2 = scale * 1.4426950408889634f;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)attention_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                              SMEM_BYTES);
    attr_done = true;
  }
  const int nqb = (S + QBLK - 1) / QBLK;
  hipLaunchKernelGGL(attention_fwd_kernel, dim3(nqb * H * B), dim3(NTHREADS), SMEM_BYTES,
                     (hipStream_t)stream_, p);
  FK_CHECK_LAUNCH("fk_attention_fwd_bf16");
  return FK_OK;
}
