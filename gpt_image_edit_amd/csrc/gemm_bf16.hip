// bf16 MFMA GEMM with fused epilogues for the FLUX MMDiT linears (K1 in SURVEY.md section 2.2).
//
//   C[M,N] = epi(A[M,K] . W[N,K]^T + bias)        A, W: bf16, K-contiguous;  fp32 accumulate.
//
// Tiling (gfx950): 128x128 block tile, BK = 64, 256 threads = 4 waves in a 2(m) x 2(n) grid, each
// wave owns a 64x64 sub-tile as 2x2 v_mfma_f32_32x32x16_bf16 accumulators.  The product is formed
// "swapped" (W rows feed the MFMA A operand, activation rows the B operand) so that every lane ends
// up with 4 consecutive output columns of ONE output row in consecutive accumulator registers --
// the epilogue packs them to bf16 and writes 8-byte pieces into an LDS staging tile, from which the
// block streams fully coalesced 16-byte rows (with the residual / gate reads equally coalesced).
//
// Staging: global -> VGPR (dwordx4, issued one K-tile ahead) -> LDS (ds_write_b128), two LDS stages,
// one barrier per K-tile.  LDS rows are 128 B (64 bf16); the 16-byte chunk index is XORed with
// ((row >> 1) & 7) so that every ds_read_b128 lane group (16 lanes, rows r..r+15) hits 16 distinct
// 16-byte slots of the 256-byte bank row (conflict-free, see MI355X_MICROARCH LDS table).
//
// Tile order: XCD-aware (block b runs on XCD b % 8, so each XCD is given a contiguous chunk of the
// tile list) and grouped 8 m-tiles deep so the ~64 tiles resident on one XCD form an ~8x8 patch that
// shares A / W K-slices through that XCD's L2.
#include <stdlib.h>

#include "fk_common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int NTHREADS = 256;
constexpr int STAGE_BYTES = (BM + BN) * BK * 2;  // 32 KiB
constexpr int CT_LD = BN + 8;                    // staging-tile row stride in elements (272 B)
constexpr int SMEM_BYTES = 2 * STAGE_BYTES;      // 64 KiB (C staging tile aliases it)
constexpr int GROUP_M = 8;

struct TileCoord {
  int tm, tn;
};

// Implicit-GEMM view of a 2-D convolution over NHWC activations (VAE, K9 in SURVEY.md 2.2):
// row m = output pixel (b, oh, ow); K index k = (kh*KW + kw)*Cin + ci; W repacked to [Cout, Kpad].
struct ConvGeom {
  int Hin, Win, Cin, Hout, Wout;
  int ksize, stride, pad, upsample;
};

FK_DEV TileCoord map_tile(int bid, int nwg, int nbm, int nbn) {
  // bijective XCD chunking (cdna guide T1): XCD x gets a contiguous run of the tile list
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  // grouped ordering: GROUP_M m-tiles deep, m fastest inside a group
  const int per_group = GROUP_M * nbn;
  const int g = t / per_group;
  const int first_m = g * GROUP_M;
  const int gm = min(nbm - first_m, GROUP_M);
  const int rem = t - g * per_group;
  TileCoord c;
  c.tm = first_m + rem % gm;
  c.tn = rem / gm;
  return c;
}

template <int EPI, bool OUT_F32, bool CONV>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_bf16_kernel(const fk_gemm_args p, const ConvGeom g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1;

  const int nbm = (p.M + BM - 1) / BM, nbn = (p.N + BN - 1) / BN;
  const TileCoord tc = map_tile(blockIdx.x, gridDim.x, nbm, nbn);
  const int m0 = tc.tm * BM, n0 = tc.tn * BN;

  // ---- global -> register staging: thread owns chunk kc of rows (tid/8 + 32 i), i = 0..3 ----------
  const int ld_row = tid >> 3;  // 0..31
  const int kc = tid & 7;       // 16-byte chunk inside the 128-byte K slice
  const bf16_t* a_ptr[4];
  const bf16_t* w_ptr[4];
  int c_pix[4], c_ih[4], c_iw[4];  // CONV: image base pixel, top-left input coordinate of the window
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int m = min(m0 + ld_row + 32 * i, p.M - 1);
    int n = min(n0 + ld_row + 32 * i, p.N - 1);
    if constexpr (CONV) {
      const int hw = g.Hout * g.Wout;
      const int b = m / hw, rem = m - b * hw;
      const int oh = rem / g.Wout, ow = rem - oh * g.Wout;
      c_pix[i] = b * g.Hin * g.Win;
      c_ih[i] = oh * g.stride - g.pad;
      c_iw[i] = ow * g.stride - g.pad;
      a_ptr[i] = (const bf16_t*)p.A;
    } else {
      a_ptr[i] = (const bf16_t*)p.A + fk_row_offset(p.a, m) + kc * 8;
    }
    w_ptr[i] = (const bf16_t*)p.W + (int64_t)n * p.ldw + kc * 8;
  }
  const int st_off = ld_row * 128 + ((kc ^ ((ld_row >> 1) & 7)) << 4);  // + i*4096 (+ 16 KiB for W)

  u32x4_t a_reg[4], w_reg[4];
  auto load_tile = [&](int kt) {
    if constexpr (CONV) {
      const int k = kt * BK + kc * 8;
      const int tap = k / g.Cin, ci = k - tap * g.Cin;
      const int kh = tap / g.ksize, kw = tap - kh * g.ksize;
      const int hlim = g.Hin << g.upsample, wlim = g.Win << g.upsample;
      const bool tap_ok = tap < g.ksize * g.ksize;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int ih = c_ih[i] + kh, iw = c_iw[i] + kw;
        const bool ok = tap_ok && ih >= 0 && ih < hlim && iw >= 0 && iw < wlim;
        ih >>= g.upsample;
        iw >>= g.upsample;
        const u32x4_t z = {0u, 0u, 0u, 0u};
        a_reg[i] = ok ? *(const u32x4_t*)(a_ptr[i] + ((int64_t)(c_pix[i] + ih * g.Win + iw) * g.Cin + ci)) : z;
        w_reg[i] = *(const u32x4_t*)(w_ptr[i] + (int64_t)kt * BK);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        a_reg[i] = *(const u32x4_t*)(a_ptr[i] + (int64_t)kt * BK);
        w_reg[i] = *(const u32x4_t*)(w_ptr[i] + (int64_t)kt * BK);
      }
    }
  };
  auto store_tile = [&](int stage) {
    char* base = smem + stage * STAGE_BYTES + st_off;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *(u32x4_t*)(base + i * 4096) = a_reg[i];
      *(u32x4_t*)(base + BM * BK * 2 + i * 4096) = w_reg[i];
    }
  };

  // ---- MFMA operand addressing ----------------------------------------------------------------------
  // operand row = base + (lane & 31); chunk = 2*kk + (lane >> 5), swizzled by (row >> 1) & 7.
  const int frow = lane & 31;
  const int fsw = (frow >> 1) & 7;   // bases are multiples of 32 -> do not disturb the swizzle bits
  const int fhalf = lane >> 5;
  const int a_rd = (wm * 64 + frow) * 128;                  // + mf*32*128
  const int w_rd = BM * BK * 2 + (wn * 64 + frow) * 128;    // + nf*32*128

  f32x16_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = p.K / BK;
  load_tile(0);
  store_tile(0);
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) load_tile(kt + 1);
    const char* sb = smem + cur * STAGE_BYTES;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int coff = (((kk * 2 + fhalf) ^ fsw) << 4);
      bf16x8_t af[2], wf[2];
#pragma unroll
      for (int mf = 0; mf < 2; ++mf) af[mf] = *(const bf16x8_t*)(sb + a_rd + mf * 4096 + coff);
#pragma unroll
      for (int nf = 0; nf < 2; ++nf) wf[nf] = *(const bf16x8_t*)(sb + w_rd + nf * 4096 + coff);
#pragma unroll
      for (int nf = 0; nf < 2; ++nf)
#pragma unroll
        for (int mf = 0; mf < 2; ++mf)
          acc[nf][mf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[nf], af[mf], acc[nf][mf], 0, 0, 0);
    }
    if (kt + 1 < nk) store_tile(cur ^ 1);
    __syncthreads();
  }

  // ---- epilogue ---------------------------------------------------------------------------------------
  // lane holds, for (nf, mf, r): row m = wm*64 + mf*32 + (lane & 31),
  //                              col n = wn*64 + nf*32 + 8*(r>>2) + 4*(lane>>5) + (r&3)
  if constexpr (OUT_F32) {
    float* C = (float*)p.C;
#pragma unroll
    for (int mf = 0; mf < 2; ++mf) {
      const int m = m0 + wm * 64 + mf * 32 + frow;
      if (m >= p.M) continue;
      const int64_t roff = fk_row_offset(p.c, m);
      // f32_flags (fp32-class VAE encoder): bit 0 = the bias is fp32, bit 1 = add the fp32 tensor `res` (rows r)
      const float* resf = (p.f32_flags & 2) ? (const float*)p.res + fk_row_offset(p.r, m) : nullptr;
#pragma unroll
      for (int nf = 0; nf < 2; ++nf)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int n = n0 + wn * 64 + nf * 32 + 8 * (r >> 2) + 4 * fhalf + (r & 3);
          if (n < p.N) {
            float v = acc[nf][mf][r];
            if constexpr (EPI == FK_EPI_SCALE) v *= p.alpha;
            else if (p.bias) v += (p.f32_flags & 1) ? ((const float*)p.bias)[n] : bf2f(((const bf16_t*)p.bias)[n]);
            if (resf) v += resf[n];
            C[roff + n] = v;
          }
        }
    }
    return;
  } else {
    // phase 1: bias (+activation), round to bf16, park in the LDS staging tile Ct[128][CT_LD]
    bf16_t* ct = (bf16_t*)smem;  // safe: the loop's final barrier ordered all operand reads
#pragma unroll
    for (int nf = 0; nf < 2; ++nf)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int nl = wn * 64 + nf * 32 + 8 * q + 4 * fhalf;  // local column of the 4-group
        const int n = n0 + nl;
        float b[4] = {0.f, 0.f, 0.f, 0.f};
        if constexpr (EPI != FK_EPI_SCALE) {
          if (p.bias && n < p.N) {
            const u32x2_t bw = *(const u32x2_t*)((const bf16_t*)p.bias + n);
            b[0] = bf_lo(bw[0]); b[1] = bf_hi(bw[0]); b[2] = bf_lo(bw[1]); b[3] = bf_hi(bw[1]);
          }
        }
#pragma unroll
        for (int mf = 0; mf < 2; ++mf) {
          float v[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float t = acc[nf][mf][q * 4 + j];
            if constexpr (EPI == FK_EPI_SCALE) t = t * p.alpha;
            else t = t + b[j];
            if constexpr (EPI == FK_EPI_GELU_TANH) t = gelu_tanh_f(round_bf(t));
            if constexpr (EPI == FK_EPI_SILU) t = silu_f(round_bf(t));
            v[j] = t;
          }
          u32x2_t pk;
          pk[0] = pack_bf2(v[0], v[1]);
          pk[1] = pack_bf2(v[2], v[3]);
          const int ml = wm * 64 + mf * 32 + frow;
          *(u32x2_t*)(ct + ml * CT_LD + nl) = pk;
        }
      }
    __syncthreads();
    // phase 2: coalesced 16-byte rows; residual / gate applied on 8-wide vectors
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int id = tid + NTHREADS * j;
      const int ml = id >> 4, cc = id & 15;
      const int m = m0 + ml, n = n0 + cc * 8;
      if (m >= p.M || n >= p.N) continue;
      u32x4_t y = *(const u32x4_t*)(ct + ml * CT_LD + cc * 8);
      if constexpr (EPI == FK_EPI_GATE_RES || EPI == FK_EPI_RES) {
        const u32x4_t rv = *(const u32x4_t*)((const bf16_t*)p.res + fk_row_offset(p.r, m) + n);
        u32x4_t gv;
        if constexpr (EPI == FK_EPI_GATE_RES) {
          const int64_t b = m / p.gate_rows_per_batch;
          gv = *(const u32x4_t*)((const bf16_t*)p.gate + b * p.gate_batch_stride + n);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float y0 = bf_lo(y[e]), y1 = bf_hi(y[e]);
          if constexpr (EPI == FK_EPI_GATE_RES) {
            y0 = round_bf(bf_lo(gv[e]) * y0);
            y1 = round_bf(bf_hi(gv[e]) * y1);
          }
          y[e] = pack_bf2(bf_lo(rv[e]) + y0, bf_hi(rv[e]) + y1);
        }
      }
      *(u32x4_t*)((bf16_t*)p.C + fk_row_offset(p.c, m) + n) = y;
    }
  }
}

template <int EPI, bool OUT_F32, bool CONV = false>
int launch(const fk_gemm_args& p, hipStream_t stream, const ConvGeom& g = ConvGeom()) {
  const int nbm = (p.M + BM - 1) / BM, nbn = (p.N + BN - 1) / BN;
  auto kern = gemm_bf16_kernel<EPI, OUT_F32, CONV>;
  FK_ENSURE_MAX_LDS(kern, SMEM_BYTES, CONV ? "fk_conv2d_nhwc_bf16" : "fk_gemm_bf16");
  hipLaunchKernelGGL(kern, dim3(nbm * nbn), dim3(NTHREADS), SMEM_BYTES, stream, p, g);
  FK_CHECK_LAUNCH(CONV ? "fk_conv2d_nhwc_bf16" : "fk_gemm_bf16");
  return FK_OK;
}

}  // namespace

static int validate_gemm(const fk_gemm_args& p) {
  FK_CHECK_ARG(p.A && p.W && p.C, "fk_gemm_bf16: null A/W/C");
  FK_CHECK_ARG(p.M > 0 && p.N > 0 && p.K > 0, "fk_gemm_bf16: bad M/N/K %d %d %d", p.M, p.N, p.K);
  FK_CHECK_ARG(p.K % BK == 0, "fk_gemm_bf16: K=%d must be a multiple of %d", p.K, BK);
  FK_CHECK_ARG(p.a.ld % 8 == 0 && p.ldw % 8 == 0, "fk_gemm_bf16: lda/ldw must be multiples of 8");
  FK_CHECK_ARG(((uintptr_t)p.A % 16 == 0) && ((uintptr_t)p.W % 16 == 0) && ((uintptr_t)p.C % 16 == 0),
               "fk_gemm_bf16: A/W/C must be 16-byte aligned");
  FK_CHECK_ARG(p.a.rows_per_batch <= 0 || p.a.batch_stride % 8 == 0, "fk_gemm_bf16: A batch stride % 8");
  if (p.out_fp32 == 2) {   // parity build of the large-tile kernels: fp32(acc + bias), 16-byte row pieces
    FK_CHECK_ARG(p.epilogue == FK_EPI_NONE && p.layout == 0 && p.N % 8 == 0 && p.c.ld % 4 == 0 &&
                     (p.c.rows_per_batch <= 0 || p.c.batch_stride % 4 == 0) && (!p.bias || ((uintptr_t)p.bias % 8 == 0)),
                 "fk_gemm_bf16: out_fp32 = 2 needs epilogue none, layout 0, N %% 8 == 0 and ldc %% 4 == 0");
  }
  if (!p.out_fp32) {
    FK_CHECK_ARG(p.N % 8 == 0, "fk_gemm_bf16: N=%d must be a multiple of 8 for bf16 output", p.N);
    FK_CHECK_ARG(p.c.ld % 8 == 0 && (p.c.rows_per_batch <= 0 || p.c.batch_stride % 8 == 0),
                 "fk_gemm_bf16: ldc / C batch stride must be multiples of 8");
    FK_CHECK_ARG(!p.bias || ((uintptr_t)p.bias % 8 == 0), "fk_gemm_bf16: bias must be 8-byte aligned");
  }
  if (p.epilogue == FK_EPI_GATE_RES || p.epilogue == FK_EPI_RES) {
    FK_CHECK_ARG(p.res != nullptr && ((uintptr_t)p.res % 16 == 0) && p.r.ld % 8 == 0 &&
                     (p.r.rows_per_batch <= 0 || p.r.batch_stride % 8 == 0),
                 "fk_gemm_bf16: residual pointer/stride invalid");
    FK_CHECK_ARG(!p.out_fp32, "fk_gemm_bf16: residual epilogues have no fp32 output");
  }
  if (p.epilogue == FK_EPI_GATE_RES) {
    FK_CHECK_ARG(p.gate != nullptr && ((uintptr_t)p.gate % 16 == 0) && p.gate_batch_stride % 8 == 0 &&
                     p.gate_rows_per_batch > 0,
                 "fk_gemm_bf16: gate pointer/stride invalid");
  }
  FK_CHECK_ARG(p.epilogue >= FK_EPI_NONE && p.epilogue <= FK_EPI_QKV, "fk_gemm_bf16: unknown epilogue %d", p.epilogue);
  FK_CHECK_ARG(p.f32_flags == 0 || (p.out_fp32 == 1 && (p.f32_flags & ~3) == 0 && (!(p.f32_flags & 2) || p.res)),
               "fk_gemm_bf16: f32_flags (fp32 bias / fp32 residual) belong to out_fp32 = 1");
  if (p.epilogue == FK_EPI_QKV) {
    FK_CHECK_ARG(!p.out_fp32 && p.q_out && p.k_out && p.wq && p.wk && p.rope_cs,
                 "fk_gemm_bf16: FK_EPI_QKV needs q_out/k_out/wq/wk/rope_cs");
    // N = 3*H*128: q | k | v columns; N = 2*H*128: the q | k columns only (the v columns as a plain GEMM of their own)
    FK_CHECK_ARG(p.qkv_heads > 0 && (p.N == 3 * p.qkv_heads * 128 || p.N == 2 * p.qkv_heads * 128) && p.qkv_s_total > 0 &&
                     p.qkv_s_offset >= 0,
                 "fk_gemm_bf16: FK_EPI_QKV needs N = 3*H*128 (q | k | v) or 2*H*128 (q | k)");
    {
      const int64_t nb = p.c.rows_per_batch > 0 ? (p.M + p.c.rows_per_batch - 1) / p.c.rows_per_batch : 1;
      FK_CHECK_ARG(nb * p.qkv_heads * p.qkv_s_total < (int64_t)1 << 31,
                   "fk_gemm_bf16: FK_EPI_QKV head-major outputs are limited to 2^31 rows");
    }
    FK_CHECK_ARG(((uintptr_t)p.q_out % 16 == 0) && ((uintptr_t)p.k_out % 16 == 0) && ((uintptr_t)p.wq % 16 == 0) &&
                     ((uintptr_t)p.wk % 16 == 0) && ((uintptr_t)p.rope_cs % 16 == 0),
                 "fk_gemm_bf16: FK_EPI_QKV pointers must be 16-byte aligned");
  }
  return FK_OK;
}

// FK_GEMM_IMPL=small forces the 128x128 kernel, =large the 256x128 LDS-DMA kernel (A/B benchmarking).
static int gemm_impl_override() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("FK_GEMM_IMPL");
    v = !e ? 0 : (e[0] == 's' ? 1 : (e[0] == 'l' ? 2 : 0));
  }
  return v;
}

extern "C" int fk_gemm_bf16(const fk_gemm_args* args, fk_stream_t stream_) {
  FK_CHECK_ARG(args != nullptr, "fk_gemm_bf16: null args");
  const fk_gemm_args& p = *args;
  hipStream_t stream = (hipStream_t)stream_;
  const int rc = validate_gemm(p);
  if (rc != FK_OK) return rc;
  if (p.variant_used) *p.variant_used = 0;   // the large-tile launcher overwrites it with the form it chose
  if (p.layout != 0) {   // K-major operands: the 256 x 256 ping-pong kernel only (it reports what it cannot take)
    FK_CHECK_ARG(!p.out_fp32, "fk_gemm_bf16: layout %d has bf16 output only", p.layout);
    const int rc2 = fk_gemm2_launch(&p, 1, 0, stream);
    if (rc2 == FK_E2BIG_STRIDES) {
      fk_set_error("fk_gemm_bf16: layout %d needs row strides that keep a 256-row tile within 2 GiB", p.layout);
      return FK_EUNSUPPORTED;
    }
    return rc2;
  }
  if (p.out_fp32 == 2) {   // the large-tile kernels' own main loops with an fp32 epilogue (launch form: plan / set_variant)
    const int rc2 = fk_gemm2_launch(&p, 1, 0, stream);
    if (rc2 == FK_E2BIG_STRIDES) {
      fk_set_error("fk_gemm_bf16: out_fp32 = 2 needs row strides that keep a 256-row tile within 2 GiB");
      return FK_EUNSUPPORTED;
    }
    return rc2;
  }
  if (p.out_fp32) {
    switch (p.epilogue) {
      case FK_EPI_NONE: return launch<FK_EPI_NONE, true>(p, stream);
      case FK_EPI_SCALE: return launch<FK_EPI_SCALE, true>(p, stream);
      default: fk_set_error("fk_gemm_bf16: fp32 output supports FK_EPI_NONE / FK_EPI_SCALE only"); return FK_EUNSUPPORTED;
    }
  }
  const int ov = gemm_impl_override();
  if (ov == 2 || p.epilogue == FK_EPI_QKV || (ov == 0 && p.M >= 192)) {
    const int rc2 = fk_gemm2_launch(&p, 1, 0, stream);
    if (rc2 != FK_E2BIG_STRIDES) return rc2;   // row strides beyond 32-bit tile offsets: 64-bit-addressing kernel below
    if (p.epilogue == FK_EPI_QKV) {
      fk_set_error("fk_gemm_bf16: FK_EPI_QKV needs row strides that keep a 256-row tile within 2 GiB");
      return FK_EUNSUPPORTED;
    }
  }
  switch (p.epilogue) {
    case FK_EPI_NONE: return launch<FK_EPI_NONE, false>(p, stream);
    case FK_EPI_GELU_TANH: return launch<FK_EPI_GELU_TANH, false>(p, stream);
    case FK_EPI_SILU: return launch<FK_EPI_SILU, false>(p, stream);
    case FK_EPI_GATE_RES: return launch<FK_EPI_GATE_RES, false>(p, stream);
    case FK_EPI_RES: return launch<FK_EPI_RES, false>(p, stream);
    case FK_EPI_SCALE: return launch<FK_EPI_SCALE, false>(p, stream);
    default: fk_set_error("fk_gemm_bf16: unknown epilogue %d", p.epilogue); return FK_EUNSUPPORTED;
  }
}

extern "C" int fk_gemm_bf16_grouped(const fk_gemm_args* args, int32_t n, fk_stream_t stream_) {
  FK_CHECK_ARG(args != nullptr && n >= 1 && n <= FK_MAX_GROUP, "fk_gemm_bf16_grouped: 1 <= n <= %d", FK_MAX_GROUP);
  for (int i = 0; i < n; ++i) {
    const int rc = validate_gemm(args[i]);
    if (rc != FK_OK) return rc;
    FK_CHECK_ARG(!args[i].out_fp32, "fk_gemm_bf16_grouped: bf16 output only");
    FK_CHECK_ARG(args[i].N == args[0].N && args[i].K == args[0].K && args[i].epilogue == args[0].epilogue,
                 "fk_gemm_bf16_grouped: all problems must share N, K and the epilogue");
  }
  hipStream_t stream = (hipStream_t)stream_;
  int rc2 = FK_E2BIG_STRIDES;
  if (args[0].variant_used) *args[0].variant_used = 0;
  if (args[0].layout != 0) {
    rc2 = fk_gemm2_launch(args, n, 0, stream);
    if (rc2 == FK_E2BIG_STRIDES) {
      fk_set_error("fk_gemm_bf16_grouped: layout %d needs row strides that keep a 256-row tile within 2 GiB", args[0].layout);
      return FK_EUNSUPPORTED;
    }
    return rc2;
  }
  if (gemm_impl_override() != 1 || args[0].epilogue == FK_EPI_QKV) {
    rc2 = fk_gemm2_launch(args, n, 0, stream);
    if (rc2 != FK_E2BIG_STRIDES) return rc2;
    if (args[0].epilogue == FK_EPI_QKV) {
      fk_set_error("fk_gemm_bf16_grouped: FK_EPI_QKV needs row strides that keep a 256-row tile within 2 GiB");
      return FK_EUNSUPPORTED;
    }
  }
  {  // one 128x128 launch per problem (A/B override, or strides beyond 32-bit tile offsets)
    for (int i = 0; i < n; ++i) {
      fk_gemm_args one = args[i];
      int rc;
      switch (one.epilogue) {
        case FK_EPI_NONE: rc = launch<FK_EPI_NONE, false>(one, stream); break;
        case FK_EPI_GELU_TANH: rc = launch<FK_EPI_GELU_TANH, false>(one, stream); break;
        case FK_EPI_SILU: rc = launch<FK_EPI_SILU, false>(one, stream); break;
        case FK_EPI_GATE_RES: rc = launch<FK_EPI_GATE_RES, false>(one, stream); break;
        case FK_EPI_RES: rc = launch<FK_EPI_RES, false>(one, stream); break;
        default: rc = launch<FK_EPI_SCALE, false>(one, stream); break;
      }
      if (rc != FK_OK) return rc;
    }
    return FK_OK;
  }
}

// Conv2d over NHWC bf16 as an implicit GEMM on the same MFMA main loop (A rows gathered per filter tap,
// zero padding / stride-2 asymmetric padding / fused nearest-2x upsample resolved in the loader).
extern "C" int fk_conv2d_nhwc_bf16(const fk_conv_args* args, fk_stream_t stream_) {
  FK_CHECK_ARG(args != nullptr, "fk_conv2d_nhwc_bf16: null args");
  const fk_conv_args& a = *args;
  FK_CHECK_ARG(a.x && a.w && a.y, "fk_conv2d_nhwc_bf16: null x/w/y");
  FK_CHECK_ARG(a.ksize == 1 || a.ksize == 3, "fk_conv2d_nhwc_bf16: ksize must be 1 or 3");
  FK_CHECK_ARG(a.stride == 1 || a.stride == 2, "fk_conv2d_nhwc_bf16: stride must be 1 or 2");
  FK_CHECK_ARG(a.Cin % 8 == 0 && a.Cout % 8 == 0, "fk_conv2d_nhwc_bf16: Cin/Cout must be multiples of 8 (pad channels)");
  FK_CHECK_ARG(a.B > 0 && a.Hin > 0 && a.Win > 0 && a.Hout > 0 && a.Wout > 0, "fk_conv2d_nhwc_bf16: bad sizes");
  FK_CHECK_ARG(!(a.upsample2x && a.stride != 1), "fk_conv2d_nhwc_bf16: upsample needs stride 1");
  const int64_t Mll = (int64_t)a.B * a.Hout * a.Wout;
  FK_CHECK_ARG(Mll < (1ll << 31) && (int64_t)a.B * a.Hin * a.Win < (1ll << 31), "fk_conv2d_nhwc_bf16: too many pixels");
  const int Kreal = a.ksize * a.ksize * a.Cin;
  const int Kpad = (Kreal + BK - 1) / BK * BK;  // weight rows are zero padded to Kpad by the caller
  fk_gemm_args p = {};
  p.A = a.x; p.a = {0, 0, 0};
  p.W = a.w; p.ldw = Kpad;
  p.bias = a.bias;
  p.C = a.y; p.c = {a.Cout, 0, 0};
  p.res = a.res; p.r = {a.Cout, 0, 0};
  p.M = (int)Mll; p.N = a.Cout; p.K = Kpad;
  p.epilogue = a.res ? FK_EPI_RES : FK_EPI_NONE;
  ConvGeom g = {a.Hin, a.Win, a.Cin, a.Hout, a.Wout, a.ksize, a.stride, a.pad, a.upsample2x ? 1 : 0};
  FK_CHECK_ARG(((uintptr_t)a.x % 16 == 0) && ((uintptr_t)a.w % 16 == 0) && ((uintptr_t)a.y % 16 == 0) &&
                   (!a.res || (uintptr_t)a.res % 16 == 0) && (!a.bias || (uintptr_t)a.bias % 8 == 0),
               "fk_conv2d_nhwc_bf16: alignment");
  hipStream_t stream = (hipStream_t)stream_;
  if (a.res) return launch<FK_EPI_RES, false, true>(p, stream, g);
  return launch<FK_EPI_NONE, false, true>(p, stream, g);
}

// The same implicit GEMM with fp32 output for the fp32-class encoder: x / w hold the bf16 parts of fp32 operands side by
// side along the channel axis ([a_hi | a_lo | a_hi] . [w_hi | w_hi | w_lo], see vae_kernels.hip), bias / res / y are fp32.
extern "C" int fk_conv2d_nhwc_f32out(const fk_conv_args* args, fk_stream_t stream_) {
  FK_CHECK_ARG(args != nullptr, "fk_conv2d_nhwc_f32out: null args");
  const fk_conv_args& a = *args;
  FK_CHECK_ARG(a.x && a.w && a.y, "fk_conv2d_nhwc_f32out: null x/w/y");
  FK_CHECK_ARG((a.ksize == 1 || a.ksize == 3) && (a.stride == 1 || a.stride == 2) && !a.upsample2x,
               "fk_conv2d_nhwc_f32out: ksize 1 / 3, stride 1 / 2, no upsample");
  FK_CHECK_ARG(a.Cin % 8 == 0 && a.Cout > 0, "fk_conv2d_nhwc_f32out: Cin must be a multiple of 8 (pad channels)");
  FK_CHECK_ARG(a.B > 0 && a.Hin > 0 && a.Win > 0 && a.Hout > 0 && a.Wout > 0, "fk_conv2d_nhwc_f32out: bad sizes");
  const int64_t Mll = (int64_t)a.B * a.Hout * a.Wout;
  FK_CHECK_ARG(Mll < (1ll << 31) && (int64_t)a.B * a.Hin * a.Win < (1ll << 31), "fk_conv2d_nhwc_f32out: too many pixels");
  FK_CHECK_ARG(((uintptr_t)a.x % 16 == 0) && ((uintptr_t)a.w % 16 == 0) && ((uintptr_t)a.y % 16 == 0),
               "fk_conv2d_nhwc_f32out: alignment");
  const int Kreal = a.ksize * a.ksize * a.Cin;
  const int Kpad = (Kreal + BK - 1) / BK * BK;
  fk_gemm_args p = {};
  p.A = a.x; p.a = {0, 0, 0};
  p.W = a.w; p.ldw = Kpad;
  p.bias = a.bias;
  p.C = a.y; p.c = {a.Cout, 0, 0};
  p.res = a.res; p.r = {a.Cout, 0, 0};
  p.M = (int)Mll; p.N = a.Cout; p.K = Kpad;
  p.epilogue = FK_EPI_NONE;
  p.out_fp32 = 1;
  p.f32_flags = 1 | (a.res ? 2 : 0);
  ConvGeom g = {a.Hin, a.Win, a.Cin, a.Hout, a.Wout, a.ksize, a.stride, a.pad, 0};
  return launch<FK_EPI_NONE, true, true>(p, (hipStream_t)stream_, g);
}
