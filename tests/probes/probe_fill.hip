// Probe (not product code): sustained L2 -> LDS fill rate of global_load_lds_dwordx4 per CU, and the
// same with plain global_load_dwordx4 to registers, from an L2-resident region.  Informs GEMM tile sizing.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int SWZ>
__global__ __launch_bounds__(512, 2) void fill_lds(const char* src, size_t region, int iters, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // each wave streams rows of 128 B at a 6 KiB row stride (like a [M, 3072] bf16 operand), 8 rows per instr
  const size_t row_stride = 6144;
  size_t row = (size_t)(blockIdx.x * 8 + wave) * 64;
  const int lrow = lane >> 3, slot = lane & 7;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const size_t r = (row + j * 8 + lrow);
      const int chunk = SWZ ? (slot ^ ((r >> 1) & 7)) : slot;
      const char* g = src + ((r * row_stride + (size_t)it * 128) % region) + chunk * 16;
      __builtin_amdgcn_global_load_lds((gbl_void*)g, (lds_void*)(smem + ((it % 3) * 6 + j) * 8192 + wave * 1024), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) sink[blockIdx.x] = ((unsigned*)smem)[5];
}

// 16 rows x 64 B per instruction (BK = 32 staging): two consecutive instructions fetch the two halves of a line
__global__ __launch_bounds__(512, 2) void fill_lds_half(const char* src, size_t region, int iters, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const size_t row_stride = 6144;
  size_t row = (size_t)(blockIdx.x * 8 + wave) * 64;
  const int lrow = lane >> 2, slot = lane & 3;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const size_t r = (row + (j >> 1) * 16 + lrow);
      const char* g = src + ((r * row_stride + (size_t)it * 128) % region) + (j & 1) * 64 + slot * 16;
      __builtin_amdgcn_global_load_lds((gbl_void*)g, (lds_void*)(smem + ((it % 3) * 6 + j) * 8192 + wave * 1024), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) sink[blockIdx.x] = ((unsigned*)smem)[5];
}

__global__ __launch_bounds__(512, 2) void fill_reg(const char* src, size_t region, int iters, unsigned* sink) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const size_t row_stride = 6144;
  size_t row = (size_t)(blockIdx.x * 8 + wave) * 64;
  const int lrow = lane >> 3, slot = lane & 7;
  u32x4 acc = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const size_t r = (row + j * 8 + lrow);
      const char* g = src + ((r * row_stride + (size_t)it * 128) % region) + slot * 16;
      const u32x4 v = *(const u32x4*)g;
      acc ^= v;
    }
  }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[blockIdx.x] = 1;
}

int main() {
  const size_t region = 48ull << 20;  // 48 MiB: L2 (32 MiB aggregate) + MALL resident
  char* src; unsigned* sink;
  hipMalloc(&src, region + (1 << 20)); hipMalloc(&sink, 4096);
  hipMemset(src, 1, region + (1 << 20));
  hipFuncSetAttribute((const void*)fill_lds<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 147456);
  hipFuncSetAttribute((const void*)fill_lds<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 147456);
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipFuncSetAttribute((const void*)fill_lds_half, hipFuncAttributeMaxDynamicSharedMemorySize, 147456);
  for (int variant = 0; variant < 4; ++variant) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (variant == 0) hipLaunchKernelGGL(fill_lds<0>, dim3(256), dim3(512), 147456, 0, src, region, iters, sink);
      else if (variant == 1) hipLaunchKernelGGL(fill_lds<1>, dim3(256), dim3(512), 147456, 0, src, region, iters, sink);
      else if (variant == 2) hipLaunchKernelGGL(fill_reg, dim3(256), dim3(512), 0, 0, src, region, iters, sink);
      else hipLaunchKernelGGL(fill_lds_half, dim3(256), dim3(512), 147456, 0, src, region, iters, sink);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double bytes = 256.0 * 8 * 6 * 1024 * iters;
      if (rep == 1)
        printf("%s: %.2f TB/s chip, %.1f GB/s per CU, %.1f B/clk/CU @2.0GHz\n",
               variant == 0 ? "glds linear" : (variant == 1 ? "glds swizzled-src" : (variant == 2 ? "global_load->reg" : "glds 16 rows x 64 B")),
               bytes / ms / 1e9, bytes / ms / 1e6 / 256, bytes / ms / 1e6 / 256 / 2.0);
    }
  }
  printf("status: %s\n", hipGetErrorString(hipDeviceSynchronize()));
  return 0;
}
