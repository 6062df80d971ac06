// Shared by attention_fwd.hip (8 waves x 32 query rows) and attention_fwd4.hip (4 waves x 64 query rows): parameters, tile
// constants, LDS-DMA / transpose-read helpers.  Everything has internal linkage (one copy per translation unit).
#pragma once
#include <stdlib.h>

#include <type_traits>

#include "fk_common.h"

// external linkage: the parameter block crosses from attention_fwd.hip (which fills it) to attention_fwd4.hip
struct AttnParams {
  const bf16_t* q;
  const bf16_t* k;
  const bf16_t* v;
  bf16_t* o;
  int B, H, S;
  int64_t v_ld, v_bs;  // V row (token) stride / batch stride in elements; head h at column h*128
  int64_t o_ld, o_bs;
  float scale_log2;    // scale * log2(e)
  float* lse;          // optional [B, H, S]: log2-domain log-sum-exp of every row (saved for the backward pass)
  // Work list: n_items = B * H * ceil(S / 256) (b, h, 256-row block) items of nkt = ceil(S / 64) KV tiles each.
  // Plain launch: one workgroup per item.  Stream-K launch (attention_fwd_kernel<.., STREAMK = true>): a persistent grid
  // of G workgroups; workgroup `pos` first takes the items pos, G + pos, ... of sk_rounds whole rounds, then works off
  // the contiguous range [cut(pos), cut(pos + 1)) of the remaining items' KV-tile units, i.e. the tail of one item, maybe
  // a whole item, the head of another (see the kernel).
  int n_items, min_part;
  int sk_rounds;       // stream-K launch: whole rounds of one item per workgroup in front of the dealt-out tail
  float* sk_partials;  // stream-K workspace: per cut (G slots) the fp32 partial of one item part ...
  unsigned* sk_ctl;    // ... and its (ticket, flag) word pair
};

namespace {

constexpr int HD = 128;
constexpr int KVBLK = 64;                       // keys per tile
constexpr int K_TILE_BYTES = KVBLK * HD * 2;    // 16 KiB
constexpr int STAGE_BYTES = 2 * K_TILE_BYTES;   // K + V

constexpr int PART_FLOATS = 256 * 128 + 2 * 512;   // O^T accumulators of 8 waves x 32 rows + (l, m_ref) per lane

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;
typedef short s16x4_t __attribute__((ext_vector_type(4)));

// buffer form of the LDS-DMA load (descriptor in SGPRs, one 32-bit offset VGPR, SGPR tile offset).  The builtin
// exists for the device target only; seen by the host pass it silently suppresses the kernel's host stub.
FK_DEV void buffer_lds16(__amdgpu_buffer_rsrc_t rsrc, char* lds_dst, int voffset, int soffset) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)lds_dst, 16, voffset, soffset, 0, 0);
#else
  (void)rsrc; (void)lds_dst; (void)voffset; (void)soffset;
#endif
}
FK_DEV void buffer_lds16(const BufDesc& d, char* lds_dst, int voffset, int soffset) {
  buffer_lds_opaque<16>(d, lds_addr_of(lds_dst), voffset, soffset);
}
FK_DEV s16x4_t lds_tr16(const char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p);
}
template <int N>
FK_DEV void wait_vmcnt() {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  else static_assert(N == 0, "add the vmcnt literal");
}

// a w_a + b w_b of the stream-K seam, both products ROUNDED before the sum: the two parts of a cut item must merge to the same
// bits whichever of them arrives second.  __fmul_rn / __fadd_rn are plain operators to hipcc, and under its default
// -ffp-contract=fast-honor-pragmas the expression became v_mul + v_fmac -- one product exact, the other rounded, i.e. a
// result that depends on the arrival order in about one merged row in five (found when the two forward kernels, which
// happen to meet in opposite orders, disagreed in 18 of 17 M output elements at S = 5632).
FK_DEV float merge2(float a, float wa, float b, float wb) {
#pragma clang fp contract(off)
  const float pa = a * wa;
  const float pb = b * wb;
  return pa + pb;
}

#ifndef FK_ATTN_PRIO
#define FK_ATTN_PRIO 1   // static s_setprio(1) for the younger half of the workgroup (waves NW/2 .. NW-1)
#endif

}  // namespace

// attention_fwd4.hip: the 4-wave kernel's launcher (grid = work items or, stream-K, one workgroup per CU)
int fk_attention_fwd4_launch(const AttnParams& p, int grid, bool streamk, hipStream_t stream);
