/*
 * fk.h -- C ABI of libfk.so: the MI355X (gfx950) native FLUX-Kontext denoiser hot path.
 *
 * Drop-in boundary.  The reference (wyhlovecpp/GPT-Image-Edit) is 100 %
 * Python and reaches this arithmetic through torch/diffusers module calls; it has no FFI of its
 * own.  Each entry point below therefore names the reference call site / third-party module whose
 * device work it replaces.  A Python maintainer binds these with ctypes (INTEGRATION.md shows the
 * stub); nothing here mentions torch types: plain device pointers (from tensor.data_ptr()),
 * integer sizes/strides in ELEMENTS, scalars, and the hipStream_t to enqueue on.
 *
 * Conventions
 *   - every function returns 0 on success or a negative FK_E* code; fk_last_error() returns a
 *     thread-local message for the last failure.  Nothing throws across the ABI.
 *   - all work is stream-ordered on `stream`; no function synchronises the device or allocates
 *     device memory: the caller (PyTorch caching allocator) owns every buffer incl. workspaces.
 *   - bf16 buffers are raw uint16 storage (torch.bfloat16), row-major, inner dimension contiguous
 *     and 16-byte aligned unless a stride parameter says otherwise.
 *   - re-entrant; no global mutable state.
 */
#ifndef FK_H_
#define FK_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* fk_stream_t; /* hipStream_t */

enum {
  FK_OK = 0,
  FK_EINVAL = -1,  /* bad shape / alignment / null pointer */
  FK_EUNSUPPORTED = -2,
  FK_ELAUNCH = -3, /* hipLaunch / runtime error */
};

/* Epilogues of fk_gemm_bf16 (applied in this order; every stage rounds to bf16 exactly where the
 * reference's bf16 torch graph does):
 *   y = bf16(acc + bias)                                   (nn.Linear output)
 *   FK_EPI_GELU_TANH : y = bf16(gelu_tanh(y))              (FeedForward "gelu-approximate")
 *   FK_EPI_SILU      : y = bf16(silu(y))                   (TimestepEmbedding / PixArtAlphaTextProjection)
 *   FK_EPI_GATE_RES  : y = bf16(res + bf16(gate[b] * y))   (gate_msa.unsqueeze(1) * attn_out; h + ...)
 *   FK_EPI_RES       : y = bf16(res + y)                   (ResnetBlock2D / VAE attention residual)
 *   FK_EPI_SCALE     : y = bf16(alpha * acc)  (no bias)    (attention scores for the VAE mid block)
 *   FK_EPI_QKV       : fused QKV projection of FluxAttnProcessor2_0 (N = 3*H*128 = q | k | v, or 2*H*128 = q | k): the q and k
 *                      thirds get per-head RMSNorm(eps 1e-6, weight) + interleaved RoPE and are written
 *                      head-major to q_out / k_out [B, H, S_total, 128]; the v third is stored like
 *                      FK_EPI_NONE into C (= the qkv buffer the attention kernel reads V from).
 */
enum {
  FK_EPI_NONE = 0,
  FK_EPI_GELU_TANH = 1,
  FK_EPI_SILU = 2,
  FK_EPI_GATE_RES = 3,
  FK_EPI_RES = 4,
  FK_EPI_SCALE = 5,
  FK_EPI_QKV = 6,
};

/* Row addressing used for A, C and the residual: logical row m lives at
 *   base + (m / rows_per_batch) * batch_stride + (m % rows_per_batch) * ld        (elements)
 * so that the text / image streams of a double block can read from and write into slices of one
 * joint [B, S, *] buffer without copies.  rows_per_batch <= 0 means "one batch" (offset = m*ld). */
typedef struct fk_rows {
  int64_t ld;
  int64_t rows_per_batch;
  int64_t batch_stride;
} fk_rows;

/* C[M,N] = epilogue(A[M,K] . W[N,K]^T + bias[N]);  A, W bf16 K-contiguous; fp32 accumulation on MFMA.
 * Replaces every nn.Linear of FluxTransformer2DModel / FeedForward / AutoencoderKL.Attention
 * (diffusers 0.32.2; reached from reference univa/utils/flux_pipeline.py:1067-1077).
 * K % 64 == 0, ldw % 8 == 0, A/W/C row starts 16-byte aligned.  N % 8 == 0 unless out_fp32. */
typedef struct fk_gemm_args {
  const void* A; fk_rows a;
  const void* W; int64_t ldw;
  const void* bias;              /* bf16 [N] or NULL */
  void* C; fk_rows c;            /* bf16, or fp32 when out_fp32 (debug/parity: acc + bias only; out_fp32 = 1: the 128 x 128
                                  * register-staged kernel, 2: the large-tile kernels' own main loops -- 256 x 256, 256 x 128,
                                  * mixed grid, split-K pairs, as the launch plan / fk_gemm_set_variant selects) */
  const void* res; fk_rows r;    /* FK_EPI_GATE_RES / FK_EPI_RES */
  const void* gate;              /* bf16, gate row b at gate + b*gate_batch_stride; b = m / gate_rows_per_batch */
  int64_t gate_batch_stride;
  int64_t gate_rows_per_batch;
  int32_t M, N, K;
  int32_t epilogue;
  int32_t out_fp32;
  float alpha;                   /* FK_EPI_SCALE */
  /* FK_EPI_QKV only (row m of this problem is token s = qkv_s_offset + m % c.rows_per_batch of batch
   * m / c.rows_per_batch; c.rows_per_batch <= 0: one batch): */
  void* q_out; void* k_out;      /* bf16 [B, H, S_total, 128] */
  const void* wq; const void* wk;/* bf16 [128] RMSNorm weights of this stream (norm_q/norm_k or norm_added_*) */
  const float* rope_cs;          /* fp32 [S_total, 64, 2]: (cos, sin) of every rotary pair (FluxPosEmbed repeats each
                                  * value over the two columns of its pair, so the [S,128] cos / sin tables hold
                                  * every number twice; the epilogue reads 512 B per token row instead of 1 KiB) */
  /* Optional split-K workspace (caller-owned, one per stream that issues GEMMs): splitk_slots x 256 KiB of fp32 partial
   * tiles followed by splitk_slots x 2 uint32 control words, the control words zeroed ONCE at allocation (they are
   * monotonic tickets afterwards).  With it, a K >= 6144 GEMM whose 256 x 256 tiling fills at most half the CUs runs
   * as two half-K workgroups per tile (deterministic: the two fp32 partials are added once, and fp32 addition
   * commutes).  NULL: never split.  Launches that share a workspace must be ordered (same stream). */
  void* splitk_ws;
  int32_t qkv_s_offset, qkv_s_total, qkv_heads;
  int32_t splitk_slots;
  /* Operand layout (the backward pass's operands as they lie in memory; reference: autograd through nn.Linear,
   * train_denoiser.py:1172):  0 = A [M, K], W [N, K] (above);  1 = W is [K, N] (ldw = its row stride): the data gradient
   * dX = dY W reads the weight as stored;  2 = A is [K, M] as well (a = its K rows, uniformly strided): the weight
   * gradient dW = dY^T X reads both operands token-major.  1 / 2: N % 256 == 0, K % 64 == 0, epilogue none (1: also
   * FK_EPI_RES), 2: M % 256 == 0; same sums, bit for bit, as layout 0 on transposed copies. */
  int32_t layout;
  int32_t f32_flags;         /* out_fp32 = 1 only: bit 0 = bias is fp32 [N]; bit 1 = `res` is an fp32 tensor (rows r) added to C */
  /* Launch controls of the large-tile kernels -- PER CALL: the library keeps no mutable launch state (round 5; rounds 2-4 had
   * process-wide fk_gemm_set_* hooks).  All zero = the defaults.  A grouped launch takes them from its first problem.
   *   variant : 0 = the launch plan chooses per problem; 128 = 256 x 128 tiles, 256 = 256 x 256 tiles, 384 = mixed grid,
   *             512 = split-K pairs, 640 = stream-K ranges -- forced where the form applies (tests, measurement).
   *   plan    : 0 = default (mixed grids and split-K pairs allowed); otherwise FK_GEMM_PLAN_EXPLICIT | allow-bits: bit 0 mixed
   *             grids (bit-identical results), bit 1 split-K pairs (results differ in the last bits from the unsplit sum: with
   *             bit 1 clear a sample's result does not depend on the grid it runs in -- "batch-invariant"), bit 2 stream-K
   *             ranges for long-K launches with a poorly filled last round (measured slower inside the edits: off by default).
   *   group_m : 0 = default (8): depth in row tiles of the grouped tile order; >= the row-tile count: every XCD owns a column
   *             range.  Results do not depend on it.
   *   mfma    : 0 = default (16); 16 = v_mfma_f32_16x16x32_bf16, 32 = v_mfma_f32_32x32x16_bf16 (the two differ in the last
   *             bits; every launch form of ONE shape agrees bit for bit with the others).  Layouts 1 / 2 follow it too (round 6; rounds 3-5: always 32). */
  int32_t variant, plan, group_m, mfma;
  /* OUT, optional (NULL: not wanted): which launch form this call used -- 128 = 256 x 128 tiles, 256 = 256 x 256, 384 = mixed
   * grid, 512 = split-K pairs of 256 x 256 tiles, 640 = stream-K ranges, 0 = none of the large-tile kernels (the 128 x 128
   * register-staged kernel).  Written by the host code of the call before it returns (tests, profiling); a grouped launch
   * writes through its first problem's pointer.  Replaces the thread-local fk_gemm_last_variant() of rounds 2-5. */
  int32_t* variant_used;
} fk_gemm_args;
#define FK_GEMM_PLAN_EXPLICIT 8
#define FK_GEMM_PLAN_BATCH_INVARIANT (FK_GEMM_PLAN_EXPLICIT | 1)   /* mixed grids only: no K split of any kind */
#define FK_SPLITK_SLOT_BYTES (256 * 256 * 4 + 8)

int fk_gemm_bf16(const fk_gemm_args* args, fk_stream_t stream);

/* n (<= FK_MAX_GROUP) independent problems that share N, K and the epilogue in ONE launch: the text- and
 * image-stream linears of a FluxTransformerBlock (different weights, different row counts) fill the GPU
 * together.  args is an array of n fk_gemm_args; bf16 output only. */
#define FK_MAX_GROUP 4
int fk_gemm_bf16_grouped(const fk_gemm_args* args, int32_t n, fk_stream_t stream);

/* out = LN(x; eps, no affine) * (1 + scale[b]) + shift[b], rows of width D (=3072), bf16 in/out.
 * Rounds like the reference graph: LN -> bf16, (1+scale) -> bf16, product -> bf16, sum -> bf16.
 * Replaces AdaLayerNormZero / AdaLayerNormZeroSingle / AdaLayerNormContinuous / norm2 (+modulate)
 * of diffusers FluxTransformerBlock.
 * Row m of x / out uses fk_rows addressing; scale/shift row b = m / mod_rows_per_batch at
 * base + b * mod_batch_stride. */
int fk_ln_modulate_bf16(const void* x, fk_rows xr, void* out, fk_rows outr, const void* shift,
                        const void* scale, int64_t mod_batch_stride, int64_t mod_rows_per_batch,
                        int64_t M, int32_t D, float eps, fk_stream_t stream);

/* Same, over a joint [text | image] sequence: rows with (m % rows_per_batch) < split use (shift, scale),
 * the others (shift_b, scale_b) -- both AdaLayerNormZero streams of a FluxTransformerBlock in one launch. */
int fk_ln_modulate2_bf16(const void* x, fk_rows xr, void* out, fk_rows outr, const void* shift,
                         const void* scale, const void* shift_b, const void* scale_b, int64_t split,
                         int64_t mod_batch_stride, int64_t mod_rows_per_batch, int64_t M, int32_t D, float eps,
                         fk_stream_t stream);

/* QKV post-processing of FluxAttnProcessor2_0: per-head RMSNorm(eps, weight) on q and k
 * (text rows s < s_txt use the *_added weights), interleaved-pair RoPE in fp32, and re-layout:
 *   qkv [B, S, 3*H*128] (q | k | v, head-major inside each)  ->  q_out, k_out [B, H, S, 128] bf16.
 * V is NOT copied: fk_attention_fwd_bf16 reads it in place from the qkv buffer.
 * cos/sin: fp32 [S, 128].  head_dim is fixed at 128. */
int fk_qkv_post_bf16(const void* qkv, void* q_out, void* k_out, const void* wq_img, const void* wk_img,
                     const void* wq_txt, const void* wk_txt, const float* cos, const float* sin, int32_t B,
                     int32_t S, int32_t S_txt, int32_t H, float eps, fk_stream_t stream);

/* O = softmax(Q K^T * scale) V, non-causal, no mask (F.scaled_dot_product_attention as called by
 * FluxAttnProcessor2_0).  q, k: [B, H, S, 128] contiguous; v: token s of batch b, head h at
 * v + b*v_batch_stride + s*v_ld + h*128 (e.g. the V third of the qkv buffer: v_ld = 3*H*128);
 * o: rows (b, s) at o + b*o_batch_stride + s*o_ld, head h at column h*128 -> the [B, S, H*128] layout the
 * next Linear consumes (o_ld lets the single block write straight into its [attn | mlp] buffer). */
int fk_attention_fwd_bf16(const void* q, const void* k, const void* v, void* o, int32_t B, int32_t H,
                          int32_t S, int64_t v_ld, int64_t v_batch_stride, int64_t o_ld,
                          int64_t o_batch_stride, float scale, fk_stream_t stream);
/* The same with a caller-owned stream-K workspace (fk_attention_ws_bytes() bytes, 16-byte aligned, zeroed ONCE at
 * allocation: its control words are monotonic tickets afterwards; launches that share it must be ordered, i.e. one
 * workspace per stream; layout: 16 KiB of (ticket, flag) pairs, then one fp32 partial slot per CU) and an optional lse output ([B, H, S] fp32, log2 domain; NULL: none).  With the workspace, a grid
 * that would leave >= 4 % of its rounds of one-workgroup-per-CU idle (B = 1: S = 8704 is 816 blocks = 3.19 rounds,
 * S = 5632 2.06) runs as a PERSISTENT grid: the KV tiles of all (b, h, 256-row block) items are dealt out as equal
 * contiguous ranges, one per CU, and a block whose keys straddle two CUs is finished by whichever arrives second
 * (fp32 partials through the workspace, agent-scope ticket + flag; deterministic: the merge is symmetric).  Where a block's
 * keys are cut depends on the grid, so the choice comes with the call -- `grid`: 0 = as described, -1 = always one workgroup
 * per block ("batch-invariant"; also what ws = NULL gives), >= 2 = a persistent grid of that many workgroups wherever every
 * block is cut at most once (test hook).  The library keeps no launch state. */
int fk_attention_fwd_ws_bf16(const void* q, const void* k, const void* v, void* o, float* lse, int32_t B, int32_t H,
                             int32_t S, int64_t v_ld, int64_t v_batch_stride, int64_t o_ld, int64_t o_batch_stride,
                             float scale, void* ws, int64_t ws_bytes, int32_t grid, fk_stream_t stream);
int64_t fk_attention_ws_bytes(void);

/* Parity / debug build of the SAME kernel (same tiling, LDS layouts, softmax, key <-> MFMA k-slot binding): the output
 * is fp32 (o_ld / o_batch_stride in fp32 elements, 16-byte aligned) and every probability enters the PV product as
 * two bf16 terms (hi + lo), so the result can be compared with an fp32 reference at the tolerance BASELINE.json
 * states (rtol 1e-3 / atol 1e-4); ~1.5x slower, never used by the product path. */
int fk_attention_fwd_f32_debug(const void* q, const void* k, const void* v, float* o, int32_t B, int32_t H,
                               int32_t S, int64_t v_ld, int64_t v_batch_stride, int64_t o_ld,
                               int64_t o_batch_stride, float scale, fk_stream_t stream);

/* ---- block-level entry points ------------------------------------------------------------------------
 * ONE call enqueues every launch of a FluxTransformerBlock / FluxSingleTransformerBlock (diffusers 0.32.2 as the reference
 * reaches it at flux_pipeline.py:1067-1077), or of all blocks of a forward, on the caller's stream -- the same launches, in
 * the same order, with the same arguments as the per-kernel calls above, hence the same bits; what they remove is host work
 * (~5 400 calls per 28-step edit become 28).  All buffers are caller-owned bf16 (PyTorch's allocator), D = H * 128:
 *   s   [B, S, D]   residual stream, text rows [0, S_txt) first, image rows after (S = S_txt + S_img); updated in place
 *   n   [B, S, D]   LN + modulate output                 qkv [B, S, 3D]   fused QKV projection (V is read from it in place)
 *   q,k [B, H, S, 128]                                    o   [B, S, D]    attention output (double blocks)
 *   ff  [B, S, 4D]  MLP hidden (double blocks)            cat [B, S, 5D]   [attention | MLP hidden] (single blocks)
 * rope_cs: fp32 [S, 64, 2] (cos, sin) per rotary pair.  splitk_ws / attn_ws: the optional workspaces of fk_gemm_args /
 * fk_attention_fwd_ws_bf16 (NULL: never split).  mod: bf16 [B, mod_total] with row stride mod_batch_stride (elements), the
 * modulation vectors of every block (Linear(SiLU(temb)) of norm1 / norm1_context / norm); a block's weights struct carries its
 * element offset(s) into a row: double block (shift, scale, gate, shift_mlp, scale_mlp, gate_mlp) x D for the image stream at
 * mod_off_img and for the text stream at mod_off_txt; single block (shift, scale, gate) x D at mod_off. */
typedef struct fk_block_ws {
  void *s, *n, *qkv, *q, *k, *o, *ff, *cat;
  const float* rope_cs;
  void* splitk_ws;
  void* attn_ws;
  int64_t attn_ws_bytes;
  int32_t splitk_slots;
  int32_t B, S_txt, S_img, H;
  float eps;                       /* LayerNorm eps (1e-6) */
  /* launch controls handed to every GEMM / attention launch of the block (all zero = defaults): fk_gemm_args.variant / plan /
   * group_m / mfma and fk_attention_fwd_ws_bf16's `grid` */
  int32_t gemm_variant, gemm_plan, gemm_group_m, gemm_mfma, attn_grid;
  int32_t* gemm_variant_used;      /* OUT, optional: fk_gemm_args.variant_used of every GEMM of the call (the last one stays) */
} fk_block_ws;
typedef struct fk_double_block_weights {   /* bf16; Linear weights [N, K] K-contiguous, fused q|k|v as [3D, D] / [3D] */
  const void *wqkv_img, *bqkv_img, *wqkv_txt, *bqkv_txt;          /* attn.to_{q,k,v} / attn.add_{q,k,v}_proj */
  const void *norm_q, *norm_k, *norm_added_q, *norm_added_k;      /* RMSNorm weights [128] */
  const void *w_out, *b_out, *w_add_out, *b_add_out;              /* attn.to_out.0 / attn.to_add_out */
  const void *w_ff1, *b_ff1, *w_ff1_ctx, *b_ff1_ctx;              /* ff.net.0.proj / ff_context.net.0.proj  [4D, D] */
  const void *w_ff2, *b_ff2, *w_ff2_ctx, *b_ff2_ctx;              /* ff.net.2 / ff_context.net.2            [D, 4D] */
  int64_t mod_off_img, mod_off_txt;
} fk_double_block_weights;
typedef struct fk_single_block_weights {
  const void *wqkv, *bqkv, *norm_q, *norm_k;                      /* attn.to_{q,k,v} fused, attn.norm_{q,k} */
  const void *w_mlp, *b_mlp, *w_out, *b_out;                      /* proj_mlp [4D, D], proj_out [D, 5D] */
  int64_t mod_off;
} fk_single_block_weights;
int fk_double_block_fwd(const fk_block_ws* ws, const fk_double_block_weights* w, const void* mod, int64_t mod_batch_stride,
                        fk_stream_t stream);
int fk_single_block_fwd(const fk_block_ws* ws, const fk_single_block_weights* w, const void* mod, int64_t mod_batch_stride,
                        fk_stream_t stream);
/* All blocks of FluxTransformer2DModel.forward between the embedders and the output head: n_double double blocks, then
 * n_single single blocks (the embedders, the conditioning GEMMs and norm_out / proj_out stay per-kernel calls: 8 launches). */
int fk_mmdit_blocks_fwd(const fk_block_ws* ws, const fk_double_block_weights* dbl, int32_t n_double,
                        const fk_single_block_weights* sgl, int32_t n_single, const void* mod, int64_t mod_batch_stride,
                        fk_stream_t stream);

/* ---- backward pass of the MMDiT (train_denoiser.py:1172 `accelerator.backward(loss)` through diffusers' blocks) ---- */
/* Forward attention that also saves lse[b, h, s] = log2(sum_j exp(q.k_j * scale)) (fp32) for the backward pass. */
int fk_attention_fwd_lse_bf16(const void* q, const void* k, const void* v, void* o, float* lse, int32_t B, int32_t H,
                              int32_t S, int64_t v_ld, int64_t v_batch_stride, int64_t o_ld, int64_t o_batch_stride,
                              float scale, fk_stream_t stream);
/* A [B, H, S, 128] bf16 tensor in any of the layouts the path uses: element (b, h, s, d) at
 * p + b*batch_stride + h*head_stride + s*ld + d   (head-major q / k: ld = 128, head_stride = S*128;
 * token-major slices of a [B, S, n*H*128] buffer: ld = n*H*128, head_stride = 128). */
typedef struct fk_attn_view {
  const void* p;
  int64_t ld, head_stride, batch_stride;
} fk_attn_view;
/* dQ, dK, dV of O = softmax(Q K^T scale) V given dO, the forward's lse and dsum[b, h, s] = sum_d dO * O (fk_rowdot_bf16).
 * Three deterministic passes (no atomics); dq / dk / dv are written, not accumulated. */
int fk_attention_bwd_bf16(const fk_attn_view* q, const fk_attn_view* k, const fk_attn_view* v, const fk_attn_view* dout,
                          const float* lse, const float* dsum, const fk_attn_view* dq, const fk_attn_view* dk,
                          const fk_attn_view* dv, int32_t B, int32_t H, int32_t S, float scale, fk_stream_t stream);
/* The same with the attention forward's stream-K workspace (fk_attention_ws_bytes(); shared with the forward on one stream):
 * the dQ pass then runs as a persistent grid where one workgroup per 256-row block would waste part of a round of CUs
 * (the parts of a cut block ADD their fp32 accumulators through the workspace: symmetric, deterministic). */
int fk_attention_bwd_ws_bf16(const fk_attn_view* q, const fk_attn_view* k, const fk_attn_view* v, const fk_attn_view* dout,
                             const float* lse, const float* dsum, const fk_attn_view* dq, const fk_attn_view* dk,
                             const fk_attn_view* dv, int32_t B, int32_t H, int32_t S, float scale, void* ws, int64_t ws_bytes,
                             int32_t grid, int32_t passes, fk_stream_t stream);
/* grid: as fk_attention_fwd_ws_bf16 (0 default, -1 plain grid, >= 2 forced persistent grid).  passes (measurement / parity):
 * 0 or 2 = two launches, the dQ pass and one pass in which wave pairs produce dK and dV together (7 tile products, default),
 * 3 = three launches (dQ, dV, dK: 8 tile products).  dQ and dV are the same bit for bit in both; dK differs in the last bf16
 * bit (the paired pass forms p (dP - D) from the bf16 p that also enters dV, the three-pass form from the fp32 p). */

/* fp32 elements of workspace `ws` the reductions below need (per-workgroup partial sums, fixed-order finalisation). */
int64_t fk_bwd_ws_floats(void);
/* Adjoint of fk_ln_modulate_bf16 for one stream (rows [b*rows_per_batch, (b+1)*rows_per_batch) use scale row b):
 *   dx_out = (dx_in ? dx_in : 0) + dLN/dx,   dshift[b] = sum_s dn,   dscale[b] = sum_s dn * LN(x)     (fp32 outputs;
 * dscale must be dshift + D: the (shift, scale) chunk pair of the block's modulation vector, batch stride given). */
int fk_ln_modulate_bwd_bf16(const void* x, fk_rows xr, const void* dn, fk_rows dnr, const void* scale,
                            int64_t mod_batch_stride, int64_t rows_per_batch, const void* dx_in, fk_rows dxi, void* dx_out,
                            fk_rows dxo, float* dshift, float* dscale, int64_t dmod_batch_stride, float* ws, int32_t B,
                            int32_t D, float eps, fk_stream_t stream);
/* Adjoint of the FK_EPI_GATE_RES epilogue out = res + gate[b] * y:  dy = dout * gate[b] (bf16),
 * dgate[b, :] = sum_s dout * y (fp32);  dres = dout. */
int fk_gate_res_bwd_bf16(const void* dout, fk_rows dor, const void* y, fk_rows yr, const void* gate,
                         int64_t gate_batch_stride, int64_t rows_per_batch, void* dy, fk_rows dyr, float* dgate,
                         int64_t dgate_batch_stride, float* ws, int32_t B, int32_t N, fk_stream_t stream);
/* out = df * gelu_tanh'(h) over n elements (h = the Linear's bf16 output before the activation); out may alias df. */
int fk_gelu_bwd_bf16(const void* h, const void* df, void* out, int64_t n, fk_stream_t stream);
/* The same for FK_EPI_SILU (the denoise_projector's activation, modeling_univa_denoise_tower.py:36-41; the projector is
 * among the parameters train_denoiser.py:71-119 trains). */
int fk_silu_bwd_bf16(const void* h, const void* df, void* out, int64_t n, fk_stream_t stream);
/* Adjoint of fk_qkv_post_bf16: dq, dk [B, H, S, 128] -> the q and k thirds of dqkv [B, S, 3*H*128] (gradient of the raw
 * projection `qkv`), and dw [2 (q, k)][2 (image, text)][128] fp32 = gradients of the RMSNorm weights. */
int fk_qkv_post_bwd_bf16(const void* dq, const void* dk, const void* qkv, void* dqkv, const void* wq_img, const void* wk_img,
                         const void* wq_txt, const void* wk_txt, const float* cos, const float* sin, float* dw, float* ws,
                         int32_t B, int32_t S, int32_t S_txt, int32_t H, float eps, fk_stream_t stream);
/* Un-fused pieces of the block forward that the recomputation (gradient checkpointing, train_denoiser.py:486) runs so
 * that the pre-gate / pre-activation tensors exist: out = bf16(res + bf16(gate[b] * y)) (the FK_EPI_GATE_RES epilogue on a
 * stored y) and y = bf16(gelu_tanh(x)) (the FK_EPI_GELU_TANH epilogue on a stored x); same rounding points. */
int fk_gate_res_fwd_bf16(const void* res, fk_rows rr, const void* y, fk_rows yr, const void* gate, int64_t gate_batch_stride,
                         int64_t rows_per_batch, void* out, fk_rows orr, int64_t M, int32_t N, fk_stream_t stream);
int fk_gelu_tanh_bf16(const void* x, fk_rows xr, void* y, fk_rows yr, int64_t M, int32_t N, fk_stream_t stream);
/* dst[c, r] (bf16, row length dst_ld >= R, columns r >= R zeroed) = src[r, c] (fp32, row stride src_ld): the fp32
 * modulation-vector gradients [B, n] as the K-padded operand of their weight-gradient GEMM. */
int fk_f32_to_bf16_transposed(const float* src, int64_t src_ld, void* dst, int32_t dst_ld, int32_t R, int32_t C,
                              fk_stream_t stream);
/* out[n] = sum_m x[m, n] (fp32): bias gradients. */
int fk_colsum_bf16(const void* x, fk_rows xr, int64_t M, int32_t N, float* out, float* ws, fk_stream_t stream);
/* out[b, h, s] = sum_d a[b, s, h*128 + d] * c[b, s, h*128 + d] (fp32): the softmax-backward row term. */
int fk_rowdot_bf16(const void* a, int64_t a_ld, int64_t a_batch_stride, const void* c, int64_t c_ld,
                   int64_t c_batch_stride, float* out, int32_t B, int32_t S, int32_t H, fk_stream_t stream);

/* Block-level entry points of the backward pass (the forward's: fk_double_block_fwd / fk_single_block_fwd above): ONE call
 * enqueues the backward of a FluxTransformerBlock / FluxSingleTransformerBlock (diffusers 0.32.2), i.e. what autograd runs for
 * the reference at `accelerator.backward(loss)` (train_denoiser.py:1172), from the activations the training forward stored.
 * Host code only: the same per-kernel calls with the same arguments, in the same order, as the Python adaptor
 * (gpt_image_edit_amd/backward.py) -- bit-identical results.  Data gradients read the stored weights (fk_gemm_args.layout 1),
 * weight gradients both operands token-major (layout 2): B * S_txt and B * S_img must be multiples of 64.
 * fk_bwd_ws = shapes, scratch shared by every block of a pass, launch controls; caller-owned, nothing is allocated. */
typedef struct fk_bwd_ws {
  int32_t B, S_txt, S_img, H;
  float eps;                               /* LayerNorm / RMSNorm eps (1e-6) */
  int32_t splitk_slots;
  void* g;                                 /* bf16 [B, S, D]: gradient of the residual stream, IN (block output) / OUT (block input) */
  void *dy, *dff, *dn, *d_o, *dqkv;        /* bf16 scratch [B,S,D], [B,S,4D], [B,S,D], [B,S,D], [B,S,3D] */
  void *dq, *dk;                           /* bf16 scratch [B, H, S, 128] */
  float* dsum;                             /* fp32 [B, H, S] */
  float* dmod;                             /* fp32 [B, 12 D] (double block: image 6D | text 6D) / [B, 3 D] (single) */
  void *ff, *cat;                          /* bf16 [B,S,4D] / [B,S,5D]: rebuilt GELU outputs (only when ff.net.2 / proj_out train) */
  const float *cos, *sin;                  /* fp32 [S, 128] rotary tables */
  float* red_ws;                           /* fk_bwd_ws_floats() */
  void* attn_ws; int64_t attn_ws_bytes;    /* fk_attention_ws_bytes(), as the forward's */
  void* splitk_ws;
  const void* mod; int64_t mod_batch_stride;   /* bf16 modulation vectors of the step (all blocks), row stride in elements */
  const void *actT, *onesT;                /* bf16 [D, 64] = silu(temb)^T zero-padded, [64, 64] ones in the first B columns */
  void* dmodT;                             /* bf16 [6 D, 64] scratch */
  int32_t gemm_variant, gemm_plan, gemm_group_m, gemm_mfma, attn_grid, attn_passes;   /* as fk_gemm_args / fk_attention_bwd_ws_bf16 */
  int32_t* gemm_variant_used;              /* OUT, optional */
} fk_bwd_ws;
typedef struct fk_block_saved {            /* what the training forward kept of ONE block (bf16 unless said) */
  const void *x0;                          /* [B,S,D] block input */
  const void *n1, *qkv, *q, *k;            /* LN+modulate output, raw fused projection [B,S,3D], q / k after RMSNorm + RoPE [B,H,S,128] */
  const void *y1, *h1, *o;                 /* pre-gate projection output [B,S,D], pre-GELU MLP hidden [B,S,4D], attention output */
  const float* lse;                        /* fp32 [B, H, S] */
  const void *x1, *n2, *y2;                /* double blocks only: stream after attention, second LN+modulate, pre-gate MLP output */
} fk_block_saved;
/* Gradient outputs: bf16 [N, K] weights as stored, fp32 [N] biases, fp32 [2 (q, k)][2 (image, text)][128] RMSNorm weights (always
 * written), AdaLN linear as bf16 [n, D] + bf16 [n, 64] whose COLUMN 0 is the bias gradient.  NULL weight pointer = frozen. */
typedef struct fk_single_block_grads {
  void* dwqkv; float* dbqkv;               /* attn.to_{q,k,v} fused [3D, D] / [3D] */
  void* dw_mlp; float* db_mlp;             /* proj_mlp [4D, D] */
  void* dw_out; float* db_out;             /* proj_out [D, 5D] */
  float* dnorm;
  void *dw_mod, *db_mod;                   /* norm.linear [3D, D] / [3D, 64] */
} fk_single_block_grads;
typedef struct fk_double_block_grads {
  void* dwqkv_img; float* dbqkv_img; void* dwqkv_txt; float* dbqkv_txt;
  void* dw_out; float* db_out; void* dw_add_out; float* db_add_out;
  void* dw_ff1; float* db_ff1; void* dw_ff1_ctx; float* db_ff1_ctx;       /* ff.net.0.proj / ff_context.net.0.proj [4D, D] */
  void* dw_ff2; float* db_ff2; void* dw_ff2_ctx; float* db_ff2_ctx;       /* ff.net.2 / ff_context.net.2 [D, 4D] */
  float* dnorm;
  void *dw_mod_img, *db_mod_img, *dw_mod_txt, *db_mod_txt;                /* norm1.linear / norm1_context.linear [6D, D] / [6D, 64] */
} fk_double_block_grads;
int fk_single_block_bwd(const fk_bwd_ws* ws, const fk_block_saved* saved, const fk_single_block_weights* w,
                        const fk_single_block_grads* grads, fk_stream_t stream);
int fk_double_block_bwd(const fk_bwd_ws* ws, const fk_block_saved* saved, const fk_double_block_weights* w,
                        const fk_double_block_grads* grads, fk_stream_t stream);

/* Elementwise / tiny kernels ------------------------------------------------------------------ */
/* y = bf16(silu(x)) over n elements (n % 8 == 0). */
int fk_silu_bf16(const void* x, void* y, int64_t n, fk_stream_t stream);
/* out[b, :256] = bf16(cat(cos, sin)(bf16(bf16(v[b]) * 1000) * freqs[k])), k < 128:
 * Timesteps(256, flip_sin_to_cos=True) after the model's `.to(dtype) * 1000`; v bf16 or fp32 [B];
 * freqs = exp(-ln(1e4) * k / 128) as a device fp32[128] table (computed once by the host). */
int fk_timestep_proj(const void* v, int32_t v_is_fp32, const float* freqs, void* out, int32_t B,
                     fk_stream_t stream);
/* out = bf16(bf16(a + b) + c) over n elements: temb = (T + G) + P. */
int fk_add3_bf16(const void* a, const void* b, const void* c, void* out, int64_t n, fk_stream_t stream);
/* True classifier-free guidance, `noise_pred = neg + true_cfg_scale * (noise_pred - neg)`
 * (univa/utils/flux_pipeline.py:1095): out = bf16(neg + bf16(scale * bf16(pos - neg))) over n elements
 * (the python-float scale stays fp32, as on the GPU the reference runs on); out may alias pos or neg. */
int fk_true_cfg_bf16(const void* pos, const void* neg, void* out, float scale, int64_t n, fk_stream_t stream);
/* ---- optimisation step of the denoiser (reference train_denoiser.py:935-1181): the HBM-bound pieces around the MMDiT forward /
 * backward ("backward pass" section above) ---- */
/* Doubles of workspace the two reductions below need. */
int64_t fk_reduce_ws_doubles(void);
/* noisy = (1 - sigma[b]) * x + sigma[b] * noise in fp32 (train_denoiser.py:994), rounded to bf16 and written as the
 * 2x2-packed tokens FluxKontextPipeline.prepare_latents / _pack_latents produce (:1009-1027):
 * x, noise fp32 [B, C, h, w]; tokens bf16, sample b at tokens + b*tokens_batch_stride, [ (h/2)(w/2), 4C ] row-major --
 * the stride lets it land in the target half of the [target | condition] token buffer. */
int fk_flow_noisy_tokens_bf16(const float* x, const float* noise, const float* sigma, void* tokens,
                              int64_t tokens_batch_stride, int32_t B, int32_t C, int32_t h, int32_t w, fk_stream_t stream);
/* Flow-matching loss and its gradient in one pass (train_denoiser.py:1096-1166, unpadded batch):
 *   d = float(pred) - (noise - x)  on the packed bf16 prediction (the `_unpack_latents` index map is applied here),
 *   loss[0] = sum(weight[b] * d^2) / (B*C*h*w)   (weight NULL = ones: the shipped logit_normal scheme),
 *   grad    = bf16(2 * weight[b] * d / (B*C*h*w)) in pred's packed layout (NULL: loss only).
 * ws: fk_reduce_ws_doubles() doubles; fixed-order two-stage sum, bit-identical from run to run. */
int fk_flow_loss_bf16(const void* pred, int64_t pred_batch_stride, const float* x, const float* noise,
                      const float* weight, void* grad, int64_t grad_batch_stride, double* loss, double* ws,
                      int32_t B, int32_t C, int32_t h, int32_t w, fk_stream_t stream);
/* The stage-2 loss AS CONFIGURED (scripts/denoiser/flux_qwen2p5vl_7b_vlm_stage2_1024.yaml:27 `mask_weight_type: 'log'`;
 * train_denoiser.py:1123-1165): per-pixel weights on top of the per-sample one,
 *   weighting[b, y, x] = weight[b] * area_weights[b, 0, y, x] * weight_mask[b, 0, y, x]   (fp32, that order; NULL factor = 1;
 *                        the maps are [B, 1, h, w] contiguous, already nearest-resized to the latent size as :1131-1148 does),
 *   loss[0] = sum(weighting * d^2) / denominator,  grad = bf16(2 * weighting * d / denominator),
 *   denominator = mask_sum[0] * C when mask_sum (a DEVICE scalar holding weight_mask.sum(), :1163-1165) is given, else B*C*h*w
 *   (loss.mean()).  All map arguments NULL: fk_flow_loss_bf16 bit for bit. */
int fk_flow_loss_weighted_bf16(const void* pred, int64_t pred_batch_stride, const float* x, const float* noise,
                               const float* weight, const float* area_weights, const float* weight_mask, const float* mask_sum,
                               void* grad, int64_t grad_batch_stride, double* loss, double* ws,
                               int32_t B, int32_t C, int32_t h, int32_t w, fk_stream_t stream);
/* out[0] = (accumulate ? out[0] : 0) + sum(g^2) over n fp32 (or bf16) elements: the global gradient norm of
 * accelerator.clip_grad_norm_ (train_denoiser.py:1171-1177) accumulated tensor by tensor, in double. */
int fk_sumsq(const void* g, int32_t g_is_bf16, int64_t n, int32_t accumulate, double* out, double* ws, fk_stream_t stream);
/* torch.optim.AdamW's update (single-tensor form) on fp32 master weights, with the clipping coefficient
 * min(1, max_grad_norm / (sqrt(grad_sumsq[0]) + 1e-6)) folded into the gradient read (grad_sumsq NULL: no clipping)
 * and the bf16 copy the forward pass reads written in the same pass (param_bf16 NULL: none).  step counts from 1. */
int fk_adamw_step(float* master, void* param_bf16, const void* grad, int32_t grad_is_bf16, float* exp_avg,
                  float* exp_avg_sq, const double* grad_sumsq, float max_grad_norm, float lr, float beta1, float beta2,
                  float eps, float weight_decay, int32_t step, int64_t n, fk_stream_t stream);
/* The same with gradients that are stored UNSCALED sums: the gradient applied is grad_scale * grad (grad_scale = 1 / world
 * for the sums a ZeRO-2 reduce-scatter leaves), grad_sumsq is the squared norm of the stored sums, and the clipping
 * coefficient becomes grad_scale * min(1, max_grad_norm / (grad_scale * sqrt(grad_sumsq[0]) + 1e-6)) -- one multiply per
 * element either way, no pass of its own for the mean.  grad_scale = 1 is fk_adamw_step bit for bit. */
int fk_adamw_step_scaled(float* master, void* param_bf16, const void* grad, int32_t grad_is_bf16, float* exp_avg,
                         float* exp_avg_sq, const double* grad_sumsq, float max_grad_norm, float grad_scale, float lr,
                         float beta1, float beta2, float eps, float weight_decay, int32_t step, int64_t n,
                         fk_stream_t stream);

/* FlowMatchEulerDiscreteScheduler.step fused with the pipeline's `noise_pred[:, :S_tgt]` slice:
 *   x[b, s, :] = bf16(float(x) + float(bf16(bf16(dsigma) * v[b, s, :])))   for s < S_tgt
 * x rows at x + b*x_batch_stride + s*C, v rows at v + b*v_batch_stride + s*C.
 * (reference univa/utils/flux_pipeline.py:1078,1099) */
int fk_euler_step_bf16(void* x, int64_t x_batch_stride, const void* v, int64_t v_batch_stride,
                       int32_t B, int32_t S_tgt, int32_t C, float dsigma, fk_stream_t stream);
/* dst[r, c] = src[c, r] for a [R, C] bf16 matrix with leading dimensions lds/ldd (elements);
 * batched over `batch` with the given batch strides. */
int fk_transpose_bf16(const void* src, int64_t lds, int64_t src_batch_stride, void* dst, int64_t ldd,
                      int64_t dst_batch_stride, int32_t R, int32_t C, int32_t batch, fk_stream_t stream);
/* Single-head attention with head dimension 512, fused (flash-style: nothing of size S x S is stored): the mid-block attention
 * of the FLUX AutoencoderKL (diffusers `Attention` in UNetMidBlock2D, heads = 1, dim_head = 512; reference call sites
 * univa/utils/flux_pipeline.py:604-611 encode, :1127-1129 decode).  q, k, v: bf16 [B, S, >= 512] views with a common row
 * stride ld_qkv (e.g. the three column blocks of one fused [B, S, 1536] projection), o: bf16 [B, S, 512] rows ld_o apart;
 * strides in elements, ld_qkv % 8 == 0, ld_o % 4 == 0; any S >= 1.  o = softmax(scale q k^T) v with fp32 scores and sums. */
int fk_attention_hd512_bf16(const void* q, const void* k, const void* v, int64_t ld_qkv, int64_t batch_stride_qkv, void* o,
                            int64_t ld_o, int64_t batch_stride_o, int32_t B, int32_t S, float scale, fk_stream_t stream);
/* y[r, :n] = bf16(softmax(x[r, :n])) with fp32 scores x (row strides ldx / ldy in elements), n % 4 == 0,
 * n <= 32768: the softmax of the VAE mid-block attention (scores kept in fp32 like a fused SDPA). */
int fk_softmax_rows(const float* x, int64_t ldx, void* y, int64_t ldy, int64_t rows, int32_t n,
                    fk_stream_t stream);

/* FLUX AutoencoderKL pieces (diffusers AutoencoderKL; reference flux_pipeline.py:604-611,1127-1129).
 * Activations are NHWC bf16 with C % 32 == 0 (3- and 16-channel inputs are zero padded to 32). */
typedef struct fk_conv_args {
  const void* x;       /* [B, Hin, Win, Cin] bf16 */
  const void* w;       /* [Cout, KH, KW, Cin] bf16 (repacked from OIHW, Cin zero padded) */
  const void* bias;    /* bf16 [Cout] */
  void* y;             /* [B, Hout, Wout, Cout] bf16 */
  const void* res;     /* optional residual, same layout as y: y = bf16(res + y) */
  int32_t B, Hin, Win, Cin, Cout;
  int32_t ksize;       /* 1 or 3 */
  int32_t stride;      /* 1 or 2 */
  int32_t pad;         /* 3x3: 1 = symmetric pad 1; 0 with stride 2 = F.pad(x,(0,1,0,1)) then valid conv */
  int32_t upsample2x;  /* 1: input is nearest-upsampled 2x on the fly (Upsample2D + conv) */
  int32_t Hout, Wout;
} fk_conv_args;
int fk_conv2d_nhwc_bf16(const fk_conv_args* args, fk_stream_t stream);
/* The 3 x 3 / stride 1 / pad 1 convolutions of ResnetBlock2D / Upsample2D (Cin % 64 == 0) as an LDS halo-tiled MFMA kernel:
 * a workgroup stages the 18 x 18 x 64-channel halo of its 16 x 16 output pixels ONCE per channel chunk and reads the nine
 * taps from LDS.  With gn_stats != NULL (fk_groupnorm_stats_nhwc_bf16 of x) the GroupNorm(gn_groups) + optional SiLU that
 * precedes the convolution in the reference graph (conv(act(norm(x)))) is applied while the halo is staged -- same
 * arithmetic and rounding points as fk_groupnorm_apply_nhwc_bf16, zero padding outside the image AFTER the normalisation --
 * so the normalised activation is never written to memory.  args->res: y = bf16(res + y).  args->upsample2x as above. */
int fk_conv3x3_halo_bf16(const fk_conv_args* args, const float* gn_stats, const void* gn_gamma, const void* gn_beta,
                         int32_t gn_groups, int32_t gn_silu, fk_stream_t stream);
/* Parity build of the same kernel: args->y is fp32 [B, Hout, Wout, Cout] = acc + bias (no residual, no output rounding),
 * so that the halo staging / GroupNorm prologue / tap loop can be held to an fp32 reference at rtol 1e-3 / atol 1e-4. */
int fk_conv3x3_halo_f32_debug(const fk_conv_args* args, const float* gn_stats, const void* gn_gamma, const void* gn_beta,
                              int32_t gn_groups, int32_t gn_silu, fk_stream_t stream);

/* GroupNorm(32 groups, eps) statistics: stats[b, g] = (mean, rstd) fp32; ws: fp32 workspace of
 * fk_groupnorm_ws_floats(B, HW, C) floats. */
int64_t fk_groupnorm_ws_floats(int32_t B, int64_t HW, int32_t C);
int fk_groupnorm_stats_nhwc_bf16(const void* x, float* stats, float* ws, int32_t B, int64_t HW,
                                 int32_t C, int32_t groups, float eps, fk_stream_t stream);
/* y = act(bf16((x - mean) * rstd * gamma + beta)); act = SiLU when silu != 0. */
int fk_groupnorm_apply_nhwc_bf16(const void* x, void* y, const float* stats, const void* gamma,
                                 const void* beta, int32_t B, int64_t HW, int32_t C, int32_t groups,
                                 int32_t silu, fk_stream_t stream);
/* Layout changes at the VAE boundary. src NCHW (fp32 or bf16) -> dst NHWC bf16 with C zero padded to Cpad,
 * applying y = bf16(bf16(bf16(x) / div) + add) (latent un-scaling `z / scaling + shift` of
 * flux_pipeline.py:1128; div = 1, add = 0 is the plain `image.to(vae.dtype)` cast). */
int fk_nchw_to_nhwc_bf16(const void* src, int32_t src_is_fp32, void* dst, int32_t B, int32_t C,
                         int32_t Cpad, int32_t H, int32_t W, float div, float add, fk_stream_t stream);
/* src NHWC bf16 (channel stride Cpad) -> dst NCHW (fp32 or bf16), first C channels,
 * y = bf16(bf16(x + add) * mul)  (`(z - shift) * scaling` of flux_pipeline.py:611). */
int fk_nhwc_to_nchw(const void* src, void* dst, int32_t dst_is_fp32, int32_t B, int32_t C, int32_t Cpad,
                    int32_t H, int32_t W, float add, float mul, fk_stream_t stream);

/* ---- fp32-class encoder (reference: train_denoiser.py:458 loads the VAE in fp32 and :887-918 encodes the target and the
 * condition image with it inside every optimisation step).  Activations are fp32 NHWC; a product a . w runs on the bf16
 * MFMA as the K-concatenation [a_hi | a_lo | a_hi] . [w_hi | w_hi | w_lo] with a_hi = bf16(a), a_lo = bf16(a - a_hi) and
 * fp32 accumulation (parts = 3; parts = 2 drops the third term: exact when the weights ARE bf16 numbers).  The term left
 * out, a_lo . w_lo, is ~2^-16 of the product; tests hold the encoder to the fp32 oracle at rtol 1e-3 / atol 1e-4. ---- */
/* y[r, p * part_stride + c] = part p of x[r, c] (fp32, n % 4 == 0); activation order (hi, lo, hi), or weight order
 * (hi, hi, lo) when weight_order != 0; parts = 2: (hi, lo). */
int fk_split_f32_rows(const float* x, int64_t ldx, void* y, int64_t ldy, int64_t part_stride, int64_t rows, int32_t n,
                      int32_t parts, int32_t weight_order, fk_stream_t stream);
/* GroupNorm(32) (+ SiLU, fp32 expf) of fp32 NHWC x with fp32 gamma / beta, written as the bf16 parts of the fp32 result:
 * y_parts [B, HW, parts * C] = [hi | lo | hi] per pixel -- the operand layout of fk_conv2d_nhwc_f32out / fk_gemm_bf16.
 * stats: [B, 32, 2] fp32 scratch, ws: fk_groupnorm_ws_floats(B, HW, C) floats. */
int fk_groupnorm_f32_nhwc(const float* x, void* y_parts, float* stats, float* ws, const float* gamma, const float* beta,
                          int32_t B, int64_t HW, int32_t C, int32_t groups, float eps, int32_t silu, int32_t parts,
                          fk_stream_t stream);
/* Implicit-GEMM convolution (fk_conv2d_nhwc_bf16's kernel) over operand parts: args->x [B, Hin, Win, Cin] and args->w hold
 * the parts side by side along the channel axis (Cin = parts * C), args->bias / res / y are fp32. */
int fk_conv2d_nhwc_f32out(const fk_conv_args* args, fk_stream_t stream);
/* The same for the 3 x 3 / stride 1 / pad 1 convolutions with parts * C % 64 == 0 on the LDS halo-tiled kernel
 * (fk_conv3x3_halo_bf16's main loop, fp32 epilogue): the operand parts are read once per channel chunk instead of nine times. */
int fk_conv3x3_halo_f32out(const fk_conv_args* args, fk_stream_t stream);
/* src NCHW fp32 -> NHWC bf16 parts [B, HW, parts * Cpad], every part zero padded to Cpad channels. */
int fk_nchw_f32_to_nhwc_parts(const float* src, void* dst, int32_t B, int32_t C, int32_t Cpad, int32_t H, int32_t W,
                              int32_t parts, fk_stream_t stream);
/* src NHWC fp32 (channel stride Cpad) -> dst NCHW fp32, first C channels, y = (x + add) * mul in fp32. */
int fk_nhwc_f32_to_nchw(const float* src, float* dst, int32_t B, int32_t C, int32_t Cpad, int32_t H, int32_t W, float add,
                        float mul, fk_stream_t stream);
/* fk_softmax_rows with the fp32 probability written as the three bf16 parts (hi, lo, hi), part_stride columns apart. */
int fk_softmax_rows_parts(const float* x, int64_t ldx, void* y, int64_t ldy, int64_t part_stride, int64_t rows, int32_t n,
                          fk_stream_t stream);

/* Pixels in: uint8 NHWC [B, Hin, Win, 3] (PIL / numpy layout) -> NHWC bf16 [B, Hout, Wout, Cpad] (channels >= 3 zero),
 * fusing the three host steps the reference runs in front of vae.encode:
 *   `(img / 255 - 0.5) / 0.5` in fp32                    (univa/serve/cli.py:106-109),
 *   VaeImageProcessor.resize on a tensor = F.interpolate(mode="nearest"): src = min(floor(dst * in/out), in - 1)
 *   and VaeImageProcessor.preprocess, which normalises (2x - 1) once more iff the tensor has no negative value
 *   (renorm != 0: the caller checks min(u8) >= 128)      (univa/utils/flux_pipeline.py:960-972),
 *   `image.to(dtype)` = bf16                             (flux_pipeline.py prepare_latents). */
int fk_pixels_u8_to_nhwc_bf16(const void* src, void* dst, int32_t B, int32_t Hin, int32_t Win, int32_t Hout,
                              int32_t Wout, int32_t Cpad, int32_t renorm, fk_stream_t stream);
/* Pixels out: decoder output NCHW (bf16 or fp32) -> uint8 NHWC, VaeImageProcessor.postprocess up to the PIL array
 * (flux_pipeline.py:1130): clamp(x / 2 + 0.5, 0, 1) in the tensor dtype, then rint(float * 255). */
int fk_image_to_u8_nhwc(const void* src, int32_t src_is_fp32, void* dst, int32_t B, int32_t C, int32_t H, int32_t W,
                        fk_stream_t stream);

const char* fk_last_error(void);
/* Build identification: "fk <version> gfx950". */
const char* fk_version(void);

#ifdef __cplusplus
}
#endif
#endif /* FK_H_ */
