// Hardware-semantics probe (not product code): prints what ds_read_b64_tr_b16 and
// global_load_lds_dwordx4 actually deliver on gfx950, so later kernel rounds can rely on measured
// layouts instead of documentation.  Built and run by tests/test_probes.py on the GPU box.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

__global__ void probe_tr(uint16_t* out) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  const int l = threadIdx.x;
  // per-lane address: documented as "lane l, elem j reads lds[(l&15) + j*16 + (l>>4)*64]" when every
  // lane passes base + (l>>4)*128 bytes?  We pass row-structured addresses and print what comes back.
  // variant A: every lane passes its own 8-byte aligned address = 8 * l
  uint32_t addrA = (uint32_t)(uintptr_t)lds + 8u * l;
  uint64_t ra;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(ra) : "v"(addrA) : "memory");
  // variant B: lanes of a 16-lane group pass rows of a [4 x 16]-element block: addr = ((l>>4)*64 + (l&3)*16 + ((l&15)>>2)*4)*2
  uint32_t addrB = (uint32_t)(uintptr_t)lds + 2u * ((l >> 4) * 64 + (l & 3) * 16 + ((l & 15) >> 2) * 4);
  uint64_t rb;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(rb) : "v"(addrB) : "memory");
  for (int j = 0; j < 4; ++j) {
    out[l * 8 + j] = (uint16_t)(ra >> (16 * j));
    out[l * 8 + 4 + j] = (uint16_t)(rb >> (16 * j));
  }
}

__global__ void probe_glds(const uint32_t* src, uint32_t* out) {
  __shared__ __attribute__((aligned(16))) uint32_t lds[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = 0xdeadbeefu;
  __syncthreads();
  const int l = threadIdx.x;
  // each lane supplies its own global address (reversed order) ; LDS base is wave-uniform
  const uint32_t* g = src + 4 * (63 - l);
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)(lds + 64), 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 1024; i += 64) out[i] = lds[i];
}

int main() {
  uint16_t* d;
  hipMalloc(&d, 64 * 8 * 2);
  hipLaunchKernelGGL(probe_tr, dim3(1), dim3(64), 0, 0, d);
  uint16_t h[512];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("ds_read_b64_tr_b16 variant A (addr = base + 8*lane): lane -> 4 element indices\n");
  for (int l = 0; l < 64; ++l) printf("A lane %2d: %4d %4d %4d %4d\n", l, h[l * 8], h[l * 8 + 1], h[l * 8 + 2], h[l * 8 + 3]);
  printf("variant B\n");
  for (int l = 0; l < 64; ++l) printf("B lane %2d: %4d %4d %4d %4d\n", l, h[l * 8 + 4], h[l * 8 + 5], h[l * 8 + 6], h[l * 8 + 7]);

  uint32_t hs[256], *ds, *dout, ho[1024];
  for (int i = 0; i < 256; ++i) hs[i] = i;
  hipMalloc(&ds, sizeof(hs));
  hipMalloc(&dout, sizeof(ho));
  hipMemcpy(ds, hs, sizeof(hs), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe_glds, dim3(1), dim3(64), 0, 0, ds, dout);
  hipMemcpy(ho, dout, sizeof(ho), hipMemcpyDeviceToHost);
  printf("global_load_lds dwordx4: lds dword index -> value (src dword index), non-sentinel only\n");
  for (int i = 0; i < 1024; ++i) if (ho[i] != 0xdeadbeefu) printf("glds lds[%4d] = %u\n", i, ho[i]);
  hipError_t e = hipDeviceSynchronize();
  printf("status: %s\n", hipGetErrorString(e));
  return 0;
}
