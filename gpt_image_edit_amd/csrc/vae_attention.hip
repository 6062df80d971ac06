// Fused single-head attention with head dimension 512: the mid-block attention of the FLUX AutoencoderKL
// (diffusers `Attention` inside `UNetMidBlock2D`, heads = 1, dim_head = 512; reached from the reference at
// univa/utils/flux_pipeline.py:604-611 (encode) and :1127-1129 (decode); SURVEY.md K9d "K4 variant").
//
// Until round 5 this was three launches per image -- fp32 scores [S, S] (1 GiB at 1024^2: S = 16384 latent pixels), a row
// softmax, P V on the GEMM kernel -- i.e. the only GB-scale per-call allocation of the inference path.  Here: flash-style,
// nothing of size S^2 exists.
//
//   workgroup = 4 waves x 16 query rows; key tiles of 32 (the next one prefetched into registers under this one's products); K and V tiles [32 keys][512] in LDS (rows padded to 1056 B)
//   S^T = K Q^T   16 x 16 x 32 MFMAs, A = K fragment (ds_read_b128), B = the wave's Q rows, held in registers for the whole pass
//   softmax       lane = query row (lane & 15); its keys sit in 4 registers x 2 key blocks x 4 lane groups: two xor-shuffles
//   O^T += V^T P^T  A = V^T fragment by two ds_read_b64_tr_b16 straight from the [keys][d] tile (no transposed copy of V),
//                 B = P in the S^T output registers as they are: the k-slot <-> key binding {4 g + r, 16 + 4 g + r} is shared
//   O^T (512 x 16 per wave) = 128 accumulator registers + 64 of prefetch: one workgroup per SIMD set (290 registers).
// Numerics: scores and the running (max, sum) in fp32, p = exp2 in fp32 rounded to bf16 for the product, O in fp32, one bf16
// rounding of O / l at the end -- the rounding points of a fused SDPA.  0.1 % of an edit: built for footprint, not for rate.
#include "fk_common.h"

namespace {

constexpr int HD = 512;
constexpr int ROW = HD * 2 + 32;          // LDS row pitch in bytes: keys 8 banks apart -> b128 fragment reads and tr reads conflict-free
constexpr int KT = 32;                    // keys per tile
constexpr int TILE_BYTES = KT * ROW;
constexpr int QROWS = 64;                 // query rows per workgroup: 4 waves x 16.  Fewer waves per workgroup (more workgroups for the
                                          // small latents) measured SLOWER: every workgroup stages all of K and V (call R: 512^2 decode 5.8
                                          // against 4.8 ms) -- the kernel is bound by its staging, so the next tile is prefetched instead

typedef short s16x4_t __attribute__((ext_vector_type(4)));
FK_DEV s16x4_t lds_tr16(const char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p);
}

struct Hd512Params {
  const bf16_t* q; const bf16_t* k; const bf16_t* v; bf16_t* o;
  int64_t ld_qkv, bs_qkv, ld_o, bs_o;     // row / batch strides in elements
  int S;
  float scale_log2e;
};

__global__ __launch_bounds__(256, 1) void attention_hd512_kernel(const Hd512Params p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 tiles = 66 KiB: above the static limit
  char* const ks = smem;
  char* const vs = smem + TILE_BYTES;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.y;
  const int q0 = blockIdx.x * QROWS + wave * 16;
  const int qi = lane & 15, g = lane >> 4;
  const bf16_t* const qb = p.q + (int64_t)b * p.bs_qkv;
  const bf16_t* const kb = p.k + (int64_t)b * p.bs_qkv;
  const bf16_t* const vb = p.v + (int64_t)b * p.bs_qkv;

  // the wave's 16 query rows as B-operand fragments: lane (row qi, octet g) holds d = 32 step + 8 g .. + 7 of its row
  bf16x8_t qf[HD / 32];
  {
    const bf16_t* qrow = qb + (int64_t)min(q0 + qi, p.S - 1) * p.ld_qkv + 8 * g;
#pragma unroll
    for (int s = 0; s < HD / 32; ++s) qf[s] = *(const bf16x8_t*)(qrow + 32 * s);
  }
  f32x4_t acc[HD / 16];
#pragma unroll
  for (int i = 0; i < HD / 16; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;

  // staging: a wave moves one 1 KiB tile row per sweep (thread -> 16-byte chunk tid & 63 of row tid >> 6 of the sweep), 8 sweeps per
  // tile and tensor.  The NEXT tile's 16 loads are issued before this tile's products and land in registers under them; they go to
  // LDS once every wave is done with this tile -- single LDS buffer, global latency hidden (one workgroup per CU at S = 16384)
  const int srow = tid >> 6, schunk = tid & 63;
  const int nt = (p.S + KT - 1) / KT;
  u32x4_t kreg[KT / 4], vreg[KT / 4];
  auto prefetch = [&](int t) {
    const int key0 = t * KT;
#pragma unroll
    for (int i = 0; i < KT / 4; ++i) {
      const int64_t off = (int64_t)min(key0 + i * 4 + srow, p.S - 1) * p.ld_qkv + schunk * 8;   // keys beyond S: clamped rows, masked below
      kreg[i] = *(const u32x4_t*)(kb + off);
      vreg[i] = *(const u32x4_t*)(vb + off);
    }
  };
  prefetch(0);
  for (int t = 0; t < nt; ++t) {
    const int key0 = t * KT;
    __syncthreads();                       // every wave is done with the previous tile
#pragma unroll
    for (int i = 0; i < KT / 4; ++i) {
      const int r = i * 4 + srow;
      *(u32x4_t*)(ks + r * ROW + schunk * 16) = kreg[i];
      *(u32x4_t*)(vs + r * ROW + schunk * 16) = vreg[i];
    }
    __syncthreads();
    if (t + 1 < nt) prefetch(t + 1);
    // ---- S^T = K Q^T for the tile's two 16-key blocks -------------------------------------------------------------
    f32x4_t s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
    const char* kf = ks + qi * ROW + g * 16;
#pragma unroll
    for (int s = 0; s < HD / 32; ++s) {
      const bf16x8_t a0 = *(const bf16x8_t*)(kf + s * 64);
      const bf16x8_t a1 = *(const bf16x8_t*)(kf + 16 * ROW + s * 64);
      s0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, qf[s], s0, 0, 0, 0);
      s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, qf[s], s1, 0, 0, 0);
    }
    // lane (query qi, group g): s0[r] = key key0 + 4 g + r, s1[r] = key key0 + 16 + 4 g + r
    float x[8];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      x[r] = key0 + 4 * g + r < p.S ? s0[r] * p.scale_log2e : -INFINITY;
      x[4 + r] = key0 + 16 + 4 * g + r < p.S ? s1[r] * p.scale_log2e : -INFINITY;
    }
    float mx = fmaxf(fmaxf(fmaxf(x[0], x[1]), fmaxf(x[2], x[3])), fmaxf(fmaxf(x[4], x[5]), fmaxf(x[6], x[7])));
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(m_run, mx);          // finite: key0 < S, so every row sees at least one valid key per tile
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    float e[8], ls = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      e[r] = __builtin_amdgcn_exp2f(x[r] - m_new);
      ls += e[r];
    }
    ls += __shfl_xor(ls, 16);
    ls += __shfl_xor(ls, 32);
    l_run = l_run * alpha + ls;
    if (__builtin_amdgcn_ballot_w64(m_new != m_run) != 0) {     // some row's maximum moved: rescale the wave's O^T
#pragma unroll
      for (int i = 0; i < HD / 16; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][r] *= alpha;
    }
    m_run = m_new;
    u32x4_t pw = {pack_bf2(e[0], e[1]), pack_bf2(e[2], e[3]), pack_bf2(e[4], e[5]), pack_bf2(e[6], e[7])};
    const bf16x8_t pf = __builtin_bit_cast(bf16x8_t, pw);
    // ---- O^T += V^T P^T: V^T fragment of 16 d-columns x {keys 4 g + r, 16 + 4 g + r} by two transpose reads.  Within a
    // 16-lane group, lane j points at 4 consecutive d of key row (j >> 2) and receives column j: out[r] = V[key r][d0 + j]
    const char* vf = vs + (4 * g + (qi >> 2)) * ROW + (qi & 3) * 8;
#pragma unroll
    for (int db = 0; db < HD / 16; ++db) {
      const s16x4_t lo = lds_tr16(vf + db * 32);
      const s16x4_t hi = lds_tr16(vf + 16 * ROW + db * 32);
      const bf16x8_t a = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      acc[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, pf, acc[db], 0, 0, 0);
    }
  }
  // ---- O = O^T / l: lane (query qi, group g) holds d = 16 db + 4 g + r
  const int qrow = q0 + qi;
  if (qrow < p.S) {
    const float inv = 1.0f / l_run;
    bf16_t* orow = p.o + (int64_t)b * p.bs_o + (int64_t)qrow * p.ld_o + 4 * g;
#pragma unroll
    for (int db = 0; db < HD / 16; ++db) {
      const u32x2_t w = {pack_bf2(acc[db][0] * inv, acc[db][1] * inv), pack_bf2(acc[db][2] * inv, acc[db][3] * inv)};
      *(u32x2_t*)(orow + 16 * db) = w;
    }
  }
}

}  // namespace

extern "C" int fk_attention_hd512_bf16(const void* q, const void* k, const void* v, int64_t ld_qkv, int64_t batch_stride_qkv,
                                       void* o, int64_t ld_o, int64_t batch_stride_o, int32_t B, int32_t S, float scale,
                                       fk_stream_t stream_) {
  FK_CHECK_ARG(q && k && v && o && B >= 1 && S >= 1, "fk_attention_hd512_bf16: null pointer or empty shape (B %d, S %d)", B, S);
  FK_CHECK_ARG(ld_qkv % 8 == 0 && ld_o % 4 == 0 && ld_qkv >= HD && ld_o >= HD,
               "fk_attention_hd512_bf16: row strides %lld / %lld (q, k, v rows 16-byte aligned, o rows 8-byte aligned)",
               (long long)ld_qkv, (long long)ld_o);
  FK_CHECK_ARG(((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) % 16 == 0 && (uintptr_t)o % 8 == 0, "fk_attention_hd512_bf16: unaligned pointer");
  FK_CHECK_ARG(B <= 65535, "fk_attention_hd512_bf16: batch %d", B);
  Hd512Params p;
  p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.v = (const bf16_t*)v; p.o = (bf16_t*)o;
  p.ld_qkv = ld_qkv; p.bs_qkv = batch_stride_qkv; p.ld_o = ld_o; p.bs_o = batch_stride_o;
  p.S = S;
  p.scale_log2e = scale * 1.4426950408889634f;
  FK_ENSURE_MAX_LDS(attention_hd512_kernel, 2 * TILE_BYTES, "fk_attention_hd512_bf16");
  hipLaunchKernelGGL(attention_hd512_kernel, dim3((S + QROWS - 1) / QROWS, B), dim3(256), 2 * TILE_BYTES, (hipStream_t)stream_, p);
  FK_CHECK_LAUNCH("fk_attention_hd512_bf16");
  return FK_OK;
}
