// Shared device/host helpers for libfk (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/fk.h"

typedef uint16_t bf16_t;  // raw bf16 storage

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;   // 8 bf16 = 4 VGPRs (MFMA A/B operand)
typedef __attribute__((ext_vector_type(4))) short bf16x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;  // 32x32 MFMA accumulator
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;

#define FK_DEV __device__ __forceinline__

FK_DEV float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// round-to-nearest-even, like torch's float -> bfloat16 conversion (NaN handling not needed here)
FK_DEV bf16_t f2bf(float f) {
  uint32_t u = __float_as_uint(f);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}
FK_DEV float round_bf(float f) { return bf2f(f2bf(f)); }

// two floats -> packed bf16 pair in ONE v_cvt_pk_bf16_f32 (round-to-nearest-even, like torch)
typedef __bf16 fk_bf16x2_t __attribute__((ext_vector_type(2)));
typedef float fk_f32x2_t __attribute__((ext_vector_type(2)));
FK_DEV uint32_t pack_bf2(float lo, float hi) {
  const fk_f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, fk_bf16x2_t));
}

FK_DEV float bf_lo(uint32_t w) { return __uint_as_float(w << 16); }
FK_DEV float bf_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
// round two floats to bf16 precision in place: one v_cvt_pk_bf16_f32 + two unpacks instead of two integer sequences
FK_DEV void round_bf_pair(float& a, float& b) {
  const uint32_t w = pack_bf2(a, b);
  a = bf_lo(w);
  b = bf_hi(w);
}

FK_DEV float gelu_tanh_f(float x) {
  // torch: 0.5 * x * (1 + tanh(u)), u = sqrt(2/pi) * (x + 0.044715 x^3).  Since 0.5 * (1 + tanh(u)) = sigmoid(2u):
  //   y = x / (1 + exp(-2u)) = x * rcp(1 + exp2(-x * (c + c*k1*x^2))),  c = 2 * sqrt(2/pi) * log2(e)
  // on the hardware exp2 / rcp (relative error ~1e-6, far below the bf16 rounding that follows): 7 VALU
  // instructions per element instead of 12 -- the GELU epilogue of the MLP-up GEMM is 256 of these per thread and
  // tile.  Saturates correctly: exp2 -> inf gives x * 0, exp2 -> 0 gives x.
  const float c = 2.302208198f, ck1 = 0.1029432396f;
  const float w = x * fmaf(ck1, x * x, c);
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-w));
}
// x * sigmoid(x); hardware exp2 / rcp, same accuracy argument as above
FK_DEV float silu_f(float x) {
  const float e = __builtin_amdgcn_exp2f(-x * 1.4426950408889634f);
  return x * __builtin_amdgcn_rcpf(1.0f + e);
}

// ---- LDS-DMA request as one opaque statement ---------------------------------------------------------------------------------
// hipcc knows that __builtin_amdgcn_raw_ptr_buffer_load_lds writes LDS and, having no alias information for it, puts
// s_waitcnt vmcnt(0) in front of the next LDS read it cannot prove disjoint -- inside a tile loop that is a wait for the
// prefetch just issued (the attention kernels' steady loops all carried one, 6 to 16 MFMAs behind their requests).  A kernel
// that orders its ring itself (counted vmcnt + barrier before a stage is read) issues the request through this form: the
// compiler sees neither a load nor an LDS write, its own vmcnt counts stay conservative (requests retire in order), and the
// descriptor is spelled out because the opaque resource type is no asm operand.  FK_OPAQUE_DMA=0 restores the builtin (A/B).
#ifndef FK_OPAQUE_DMA
#define FK_OPAQUE_DMA 1
#endif
struct BufDesc { u32x4_t w; };
FK_DEV BufDesc make_buf_desc(const void* base, unsigned bytes) {
  const uint64_t a = (uint64_t)base;
  return BufDesc{u32x4_t{(uint32_t)a, (uint32_t)(a >> 32) & 0xffffu, bytes, 0x00020000u}};
}
template <int BYTES>
FK_DEV void buffer_lds_opaque(const BufDesc& d, unsigned lds_addr, int voffset, int soffset) {
  static_assert(BYTES == 16 || BYTES == 4, "LDS-DMA piece: 16 or 4 bytes per lane");
  // M0: hipcc sets it next to each of its own uses; the s_nop is the SALU-write-M0 -> LDS-DMA wait state it also emits.
  // M0 is NOT in the clobber list: clang treats it as a reserved register ("may not be preserved ... undefined behaviour"
  // warning on every instantiation).  The rule instead: a kernel issues ALL of its LDS-DMA requests through this form or ALL
  // through the builtin (the GEMM / attention files pick the form per kernel instantiation with `if constexpr`), and uses no
  // other M0 consumer (s_sendmsg, GWS, ds_*_gs) -- then no compiler-managed M0 value is ever live across the statement.
  if constexpr (BYTES == 16)
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :: "s"(lds_addr), "v"(voffset), "s"(d.w), "s"(soffset) : "memory");
  else
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen lds"
                 :: "s"(lds_addr), "v"(voffset), "s"(d.w), "s"(soffset) : "memory");
}
FK_DEV unsigned lds_addr_of(const char* lds_ptr) {
  return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)lds_ptr;
}
#if FK_OPAQUE_DMA
typedef BufDesc DmaDesc;
FK_DEV DmaDesc make_dma_desc(const void* base, int64_t bytes) { return make_buf_desc(base, (unsigned)bytes); }
#else
typedef __amdgpu_buffer_rsrc_t DmaDesc;
FK_DEV DmaDesc make_dma_desc(const void* base, int64_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
}
#endif

FK_DEV int64_t fk_row_offset(const fk_rows& r, int64_t m) {
  if (r.rows_per_batch <= 0) return m * r.ld;
  int64_t b = m / r.rows_per_batch;
  return b * r.batch_stride + (m - b * r.rows_per_batch) * r.ld;
}

// internal epilogue of the large-tile GEMM kernels: fk_gemm_args.out_fp32 == 2 (parity build of the SAME main loops:
// C = fp32(acc + bias), written straight from the accumulator registers; tests hold it to rtol 1e-3 / atol 1e-4)
constexpr int FK_EPI_F32DBG = 64;

// internal: returned by fk_gemm2_launch when a 256-row tile's rows are not addressable with 32-bit byte offsets
constexpr int FK_E2BIG_STRIDES = -100;
int fk_gemm2_launch(const fk_gemm_args* probs, int n, int bn_hint, hipStream_t stream);  // gemm_pingpong_bf16.hip

// host side ---------------------------------------------------------------------------------------
void fk_set_error(const char* fmt, ...);

// hipFuncAttributeMaxDynamicSharedMemorySize applies to the CURRENT device only: remember per (kernel, device)
// whether it has been raised.  `done` is the calling launcher's static table (one per kernel instantiation);
// the race on it is benign (the call is idempotent).  Returns hipSuccess or the runtime's error.
constexpr int FK_MAX_DEVICES = 64;
struct fk_lds_attr_table { bool done[FK_MAX_DEVICES]; };
inline hipError_t fk_ensure_max_lds(fk_lds_attr_table& t, const void* kern, int bytes) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (dev >= 0 && dev < FK_MAX_DEVICES && t.done[dev]) return hipSuccess;
  e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess && dev >= 0 && dev < FK_MAX_DEVICES) t.done[dev] = true;
  return e;
}
#define FK_ENSURE_MAX_LDS(kern, bytes, name)                                                       \
  do {                                                                                             \
    static fk_lds_attr_table tbl_ = {};                                                            \
    hipError_t ea_ = fk_ensure_max_lds(tbl_, (const void*)(kern), (bytes));                        \
    if (ea_ != hipSuccess) {                                                                       \
      fk_set_error("%s: cannot raise the dynamic LDS limit to %d bytes: %s", name, (int)(bytes),   \
                   hipGetErrorString(ea_));                                                        \
      return FK_ELAUNCH;                                                                           \
    }                                                                                              \
  } while (0)
#define FK_CHECK_ARG(cond, ...)      \
  do {                               \
    if (!(cond)) {                   \
      fk_set_error(__VA_ARGS__);     \
      return FK_EINVAL;              \
    }                                \
  } while (0)
#define FK_CHECK_LAUNCH(name)                                               \
  do {                                                                      \
    hipError_t e_ = hipGetLastError();                                      \
    if (e_ != hipSuccess) {                                                 \
      fk_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));   \
      return FK_ELAUNCH;                                                    \
    }                                                                       \
  } while (0)
