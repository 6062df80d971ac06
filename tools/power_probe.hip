// Sustained-rate probes of the matrix pipe under the chip's power limit (gfx950): what a GEMM main loop's ingredients cost
// in CLOCK.  DESIGN.md section 7 argues that the GEMM family is power-bound (1.27 GHz at 89 % matrix-pipe occupancy) and that
// the vendor kernel's advantage is joules per flop; this tool measures the ingredients one at a time so that the next kernel
// is designed from numbers:
//   * v_mfma_f32_32x32x16_bf16 against v_mfma_f32_16x16x32_bf16 (same flops per cycle, different register traffic),
//   * 8 waves per CU (two per SIMD, 128 accumulator registers) against 4 (one per SIMD, 256),
//   * ds_read_b128 operand fragments at 0 / 0.5 / 0.75 / 1.0 reads per 32-cycle MFMA slot,
//   * LDS-DMA refill (buffer_load ... lds from an L2-resident source) at the GEMM's rate of 64 KiB per 256 x 256 x 64 tile.
// Every probe runs ~1.5 s of back-to-back launches on all CUs with random bf16 operands (zero operands clock ~20 % higher:
// cdna_hip_programming.md rule 25) and reports the steady-state TFLOP/s and the shader clock (s_memtime over s_memrealtime).
//   build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/power_probe.hip -o tools/build/power_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((address_space(3))) void lds_void;

#define CHECK(x)                                                                              \
  do {                                                                                        \
    hipError_t e_ = (x);                                                                      \
    if (e_ != hipSuccess) {                                                                   \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));       \
      exit(1);                                                                                \
    }                                                                                         \
  } while (0)

struct Clocks { unsigned long long shader, real; };

__device__ __forceinline__ void stamp(Clocks* c, int which) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const unsigned long long s = __builtin_amdgcn_s_memtime(), r = __builtin_amdgcn_s_memrealtime();
    if (which == 0) { c->shader = s; c->real = r; }
    else { c->shader = s - c->shader; c->real = r - c->real; }
  }
}

// ---- matrix pipe only -------------------------------------------------------------------------------------------------
// NACC accumulator blocks per wave; every MFMA takes a different (a, b) pair of the 8 + 8 operand fragments the wave holds.
template <int NT, int NACC, int WPE = 1>   // WPE = 2 keeps the kernel in the 256-register (VGPR-form MFMA) budget: no AGPR copies in the loop
__global__ __launch_bounds__(NT, WPE) void mfma32_only(const bf16x8_t* __restrict__ src, float* out, int iters, Clocks* clk) {
  bf16x8_t a[8], b[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    a[i] = src[(threadIdx.x + i * 1024) & 8191];
    b[i] = src[(threadIdx.x + i * 1024 + 512) & 8191];
  }
  f32x16_t acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  stamp(clk, 0);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 64 / NACC; ++rep)
#pragma unroll
      for (int i = 0; i < NACC; ++i)
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(i + rep) & 7], b[(i * 3 + rep) & 7], acc[i], 0, 0, 0);
  }
  stamp(clk, 1);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * NT + threadIdx.x] = s;
}

template <int NT, int NACC, int WPE = 1>   // NACC 16 x 16 blocks (4 registers each); 128 MFMAs of 16 cycles per iteration = the same 2048 cycles
__global__ __launch_bounds__(NT, WPE) void mfma16_only(const bf16x8_t* __restrict__ src, float* out, int iters, Clocks* clk) {
  bf16x8_t a[8], b[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    a[i] = src[(threadIdx.x + i * 1024) & 8191];
    b[i] = src[(threadIdx.x + i * 1024 + 512) & 8191];
  }
  f32x4_t acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
  stamp(clk, 0);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 128 / NACC; ++rep)
#pragma unroll
      for (int i = 0; i < NACC; ++i)
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(i + rep) & 7], b[(i * 3 + rep) & 7], acc[i], 0, 0, 0);
  }
  stamp(clk, 1);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) s += acc[i][r];
  out[blockIdx.x * NT + threadIdx.x] = s;
}

// ---- matrix pipe + LDS operand reads (+ LDS-DMA refill) ----------------------------------------------------------------
// 8 waves (512 threads), 8 accumulator blocks per wave = gemm8_kernel's register picture.  One "phase" = 8 MFMAs; READS
// ds_read_b128 fragments per phase replace operand registers round-robin (so that every read is consumed), from a 128 KiB LDS
// image of random data addressed like the GEMM's swizzled rows (conflict-free).  DMA: pieces of 1 KiB per wave and phase
// (gemm8_kernel: 2) from a 2 MiB source that stays in the XCD's L2.  No barriers: the probe measures power, not a schedule.
constexpr int LDS_BYTES = 128 * 1024;
template <int READS, int DMA, bool M16 = false, int AUX = 0>   // AUX: cache-policy bits of the LDS-DMA loads (2 = nt, 1 = sc0, 16 = sc1)
__global__ __launch_bounds__(512, 2) void mfma32_lds(const bf16x8_t* __restrict__ src, float* out, int iters, Clocks* clk) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < LDS_BYTES / 16; i += 512) ((bf16x8_t*)smem)[i] = src[(i + blockIdx.x * 64) & 131071];
  __syncthreads();
  bf16x8_t f[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) f[i] = src[(tid + i * 512) & 8191];
  f32x16_t acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  // fragment address: row (lane & 31) of a 128-byte-row image, 16-byte chunk (2 kk + (lane >> 5)) ^ ((row >> 1) & 7)
  const int frow = lane & 31, fhalf = lane >> 5, fsw = (frow >> 1) & 7;
  int koffs[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) koffs[kk] = ((kk * 2 + fhalf) ^ fsw) << 4;
  const int rd0 = frow * 128;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 2 << 20, 0x00020000);
  const int voff = lane * 16;
  stamp(clk, 0);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int ph = 0; ph < 8; ++ph) {
      // 32-row block (it, ph, wave) of the image: 4 KiB apart, all 32 blocks visited
      const char* blk = smem + (((it * 8 + ph) * 5 + wave * 3) & 31) * 4096 + rd0;
#pragma unroll
      for (int r = 0; r < READS; ++r) f[(ph * READS + r) & 7] = *(const bf16x8_t*)(blk + koffs[r & 3] + (r >> 2) * 2048);
      if constexpr (DMA > 0) {
#pragma unroll
        for (int d = 0; d < DMA; ++d) {
          const int piece = ((it * 8 + ph) * DMA + d) * 8 + wave;   // 1 KiB pieces of the 2 MiB source, 128 KiB ring in LDS
#if defined(__HIP_DEVICE_COMPILE__)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(smem + LDS_BYTES + 4096 + (piece & 15) * 1024), 16, voff,
                                                   (piece & 2047) * 1024, 0, AUX);
#endif
        }
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if constexpr (M16) {   // the same 32 x 32 x 32 of work per accumulator block as four 16 x 16 x 32 instructions (gemm8's M16 form)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            f32x4_t c = {acc[i][4 * q], acc[i][4 * q + 1], acc[i][4 * q + 2], acc[i][4 * q + 3]};
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[(i + q) & 7], f[(i + 3 + ph + q) & 7], c, 0, 0, 0);
            acc[i][4 * q] = c[0]; acc[i][4 * q + 1] = c[1]; acc[i][4 * q + 2] = c[2]; acc[i][4 * q + 3] = c[3];
          }
        } else {
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[i], f[(i + 3 + ph) & 7], acc[i], 0, 0, 0);
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  stamp(clk, 1);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 512 + tid] = s;
}

// 4 waves (one per SIMD), 16 accumulator blocks = a 128 x 128 wave tile: READS per 16 MFMAs (8 = 0.5 per MFMA)
template <int READS>
__global__ __launch_bounds__(256) void mfma32_lds_4w(const bf16x8_t* __restrict__ src, float* out, int iters, Clocks* clk) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < LDS_BYTES / 16; i += 256) ((bf16x8_t*)smem)[i] = src[(i + blockIdx.x * 64) & 131071];
  __syncthreads();
  bf16x8_t f[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) f[i] = src[(tid + i * 512) & 8191];
  f32x16_t acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  const int frow = lane & 31, fhalf = lane >> 5, fsw = (frow >> 1) & 7;
  int koffs[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) koffs[kk] = ((kk * 2 + fhalf) ^ fsw) << 4;
  const int rd0 = frow * 128;
  stamp(clk, 0);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int ph = 0; ph < 4; ++ph) {
      const char* blk = smem + (((it * 4 + ph) * 5 + wave * 3) & 31) * 4096 + rd0;
#pragma unroll
      for (int r = 0; r < READS; ++r) f[(ph * READS + r) & 7] = *(const bf16x8_t*)(blk + koffs[r & 3] + (r >> 2) * 2048);
#pragma unroll
      for (int i = 0; i < 16; ++i)
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[i & 7], f[(i + 3 + ph) & 7], acc[i], 0, 0, 0);
    }
  }
  stamp(clk, 1);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + tid] = s;
}

// A 4-wave GEMM main loop WITHOUT its barriers (upper bound of a hipcc-built one-wave-per-SIMD 256 x 256 kernel): 128 x 128
// wave tile on 32 x 32 x 16 MFMAs (the 16 x 16 x 32 form does not survive hipcc's AGPR allocation at 256 accumulator registers:
// ~3 v_accvgpr moves per MFMA in the loop), 8 fragment reads (0.5 per MFMA) and, with DMA, the 16 LDS-DMA pieces per K-tile that
// a wave of such a kernel issues itself -- in its own instruction stream, next to its MFMAs.
template <int DMA>
__global__ __launch_bounds__(256) void skeleton_4w(const bf16x8_t* __restrict__ src, float* out, int iters, Clocks* clk) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < LDS_BYTES / 16; i += 256) ((bf16x8_t*)smem)[i] = src[(i + blockIdx.x * 64) & 131071];
  __syncthreads();
  f32x16_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int frow = lane & 31, fq = lane >> 5;
  const int rd0 = frow * 128 + ((fq ^ ((frow >> 1) & 7)) << 4);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 2 << 20, 0x00020000);
  const int voff = lane * 16;
  bf16x8_t af[2][4], wf[2][4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { af[0][i] = *(const bf16x8_t*)(smem + rd0 + i * 4096); wf[0][i] = *(const bf16x8_t*)(smem + 65536 + rd0 + i * 4096); }
  stamp(clk, 0);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {       // one K-tile of 64 = four k-steps of 16 MFMAs
      const int cb = ks & 1, nb = cb ^ 1;
      const char* blk = smem + (((it * 4 + ks) * 5 + wave * 3) & 3) * 32;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        af[nb][i] = *(const bf16x8_t*)(blk + rd0 + i * 4096);
        wf[nb][i] = *(const bf16x8_t*)(blk + 65536 + rd0 + i * 4096);
        if constexpr (DMA > 0) {           // 4 pieces per k-step = 16 per K-tile and wave (64 KiB per K-tile and workgroup)
          const int piece = ((it * 4 + ks) * 4 + i) * 4 + wave;
#if defined(__HIP_DEVICE_COMPILE__)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(smem + LDS_BYTES + 4096 + (piece & 15) * 1024), 16, voff,
                                                   (piece & 2047) * 1024, 0, 0);
#endif
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[cb][i], af[cb][j], acc[i][j], 0, 0, 0);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  stamp(clk, 1);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  out[blockIdx.x * 256 + tid] = s;
}

struct Result { double tflops, ghz; };

template <class Launch>
Result sustained(const char* name, double flops_per_launch, Launch launch, Clocks* d_clk, double seconds) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  launch();   // warm-up (also raises the LDS limit where needed)
  CHECK(hipDeviceSynchronize());
  std::vector<double> tf, ghz;
  double elapsed = 0;
  while (elapsed < seconds) {
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < 4; ++i) launch();
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    Clocks c;
    CHECK(hipMemcpy(&c, d_clk, sizeof(c), hipMemcpyDeviceToHost));
    tf.push_back(4 * flops_per_launch / (ms * 1e-3) / 1e12);
    ghz.push_back(c.real ? (double)c.shader / (double)c.real * 0.1 : 0.0);   // s_memrealtime ticks at 100 MHz
    elapsed += ms * 1e-3;
  }
  // steady state = the second half of the run
  std::vector<double> t2(tf.begin() + tf.size() / 2, tf.end()), g2(ghz.begin() + ghz.size() / 2, ghz.end());
  std::sort(t2.begin(), t2.end());
  std::sort(g2.begin(), g2.end());
  Result r = {t2[t2.size() / 2], g2[g2.size() / 2]};
  printf("%-66s first %7.1f  steady %7.1f TF/s  (%.3f of 2500)  clock %.3f GHz  [%zu samples]\n", name, tf[0], r.tflops,
         r.tflops / 2500.0, r.ghz, tf.size());
  fflush(stdout);
  CHECK(hipEventDestroy(e0));
  CHECK(hipEventDestroy(e1));
  return r;
}

int main(int argc, char** argv) {
  const double seconds = argc > 1 ? atof(argv[1]) : 1.5;
  int dev = 0, cus = 0;
  CHECK(hipGetDevice(&dev));
  CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  printf("# power_probe: %d CUs, %.1f s per probe, random bf16 operands in [-1, 1)\n", cus, seconds);
  const size_t nsrc = 131072 + 8192;   // fragments of 16 bytes: 2 MiB + slack
  std::vector<uint16_t> h(nsrc * 8);
  uint32_t s = 12345u;
  for (auto& v : h) {
    s = s * 1664525u + 1013904223u;
    const float x = ((s >> 8) & 0xffff) / 32768.0f - 1.0f;
    uint32_t u;
    memcpy(&u, &x, 4);
    v = (uint16_t)(u >> 16);
  }
  bf16x8_t* d_src;
  float* d_out;
  Clocks* d_clk;
  CHECK(hipMalloc(&d_src, nsrc * 16));
  CHECK(hipMemcpy(d_src, h.data(), nsrc * 16, hipMemcpyHostToDevice));
  CHECK(hipMalloc(&d_out, (size_t)cus * 512 * 4));
  CHECK(hipMalloc(&d_clk, sizeof(Clocks)));
  const int iters = 4000;                                 // 4000 x 2048 matrix-pipe cycles ~ 6 ms at 1.3 GHz
  // flops per launch: every SIMD issues `iters` x 64 MFMA slots of 32 cycles (32 x 32 x 16 x 2 flops each) per wave sharing it
  const double f_slot = 2.0 * 32 * 32 * 16;
  const double f8 = (double)cus * 8 * iters * 64 * f_slot, f4 = (double)cus * 4 * iters * 64 * f_slot;

  sustained("mfma 32x32x16, 8 waves/CU, no LDS", f8, [&] { hipLaunchKernelGGL((mfma32_only<512, 8>), dim3(cus), dim3(512), 0, 0, d_src, d_out, iters, d_clk); }, d_clk, seconds);
  sustained("mfma 16x16x32, 8 waves/CU, no LDS", f8, [&] { hipLaunchKernelGGL((mfma16_only<512, 32>), dim3(cus), dim3(512), 0, 0, d_src, d_out, iters, d_clk); }, d_clk, seconds);
  sustained("mfma 32x32x16, 4 waves/CU, no LDS", f4, [&] { hipLaunchKernelGGL((mfma32_only<256, 16>), dim3(cus), dim3(256), 0, 0, d_src, d_out, iters, d_clk); }, d_clk, seconds);
  sustained("mfma 16x16x32, 4 waves/CU, no LDS", f4, [&] { hipLaunchKernelGGL((mfma16_only<256, 64>), dim3(cus), dim3(256), 0, 0, d_src, d_out, iters, d_clk); }, d_clk, seconds);

  // ONE wave per SIMD issuing MFMAs back to back from a small register set (the picture of a ping-pong group's MFMA phase:
  // its partner wave issues no matrix work meanwhile): the single-wave issue cadence of the two shapes
  sustained("mfma 32x32x16, 4 waves/CU, 8 acc blocks (single-wave cadence)", f4, [&] { hipLaunchKernelGGL((mfma32_only<256, 8, 2>), dim3(cus), dim3(256), 0, 0, d_src, d_out, iters, d_clk); }, d_clk, seconds);
  sustained("mfma 16x16x32, 4 waves/CU, 32 acc blocks (single-wave cadence)", f4, [&] { hipLaunchKernelGGL((mfma16_only<256, 32, 2>), dim3(cus), dim3(256), 0, 0, d_src, d_out, iters, d_clk); }, d_clk, seconds);

  auto lds8 = [&](auto kern, int extra) {
    CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES + 4096 + extra));
    return [=] { hipLaunchKernelGGL(kern, dim3(cus), dim3(512), LDS_BYTES + 4096 + extra, 0, d_src, d_out, iters, d_clk); };
  };
  sustained("8 waves + 0.50 ds_read_b128 per MFMA", f8, lds8(mfma32_lds<4, 0>, 0), d_clk, seconds);
  sustained("8 waves + 0.75 ds_read_b128 per MFMA", f8, lds8(mfma32_lds<6, 0>, 0), d_clk, seconds);
  sustained("8 waves + 1.00 ds_read_b128 per MFMA", f8, lds8(mfma32_lds<8, 0>, 0), d_clk, seconds);
  sustained("8 waves + 0.75 reads + LDS-DMA at GEMM rate", f8, lds8(mfma32_lds<6, 2>, 16384), d_clk, seconds);
  sustained("8 waves + 0.50 reads + LDS-DMA at GEMM rate", f8, lds8(mfma32_lds<4, 2>, 16384), d_clk, seconds);
  sustained("8 waves, 16x16x32 + 0.75 reads (gemm8, round 5 default)", f8, lds8(mfma32_lds<6, 0, true>, 0), d_clk, seconds);
  sustained("8 waves, 16x16x32 + 0.75 reads + LDS-DMA at GEMM rate", f8, lds8(mfma32_lds<6, 2, true>, 16384), d_clk, seconds);
  sustained("8 waves, 16x16x32 + 0.75 reads + LDS-DMA, nt loads (aux 2)", f8, lds8(mfma32_lds<6, 2, true, 2>, 16384), d_clk, seconds);
  sustained("8 waves, 16x16x32 + 0.75 reads + LDS-DMA, sc1 loads (aux 16)", f8, lds8(mfma32_lds<6, 2, true, 16>, 16384), d_clk, seconds);
  {
    auto k0 = skeleton_4w<0>;
    auto k1 = skeleton_4w<1>;
    CHECK(hipFuncSetAttribute((const void*)k0, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES + 4096 + 16384));
    CHECK(hipFuncSetAttribute((const void*)k1, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES + 4096 + 16384));
    sustained("4-wave 128x128 main loop, no barriers, 0.5 reads, no DMA", f4, [=] { hipLaunchKernelGGL(k0, dim3(cus), dim3(256), LDS_BYTES + 4096 + 16384, 0, d_src, d_out, iters, d_clk); }, d_clk, seconds);
    sustained("4-wave 128x128 main loop, no barriers, 0.5 reads + own LDS-DMA", f4, [=] { hipLaunchKernelGGL(k1, dim3(cus), dim3(256), LDS_BYTES + 4096 + 16384, 0, d_src, d_out, iters, d_clk); }, d_clk, seconds);
  }
  {
    auto kern = mfma32_lds_4w<8>;
    CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES + 4096));
    sustained("4 waves (128x128 wave tile) + 0.50 reads", f4, [=] { hipLaunchKernelGGL(kern, dim3(cus), dim3(256), LDS_BYTES + 4096, 0, d_src, d_out, iters, d_clk); }, d_clk, seconds);
  }
  CHECK(hipDeviceSynchronize());
  return 0;
}
