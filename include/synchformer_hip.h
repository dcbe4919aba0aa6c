/*
 * synchformer_hip.h - C ABI of libsynchformer_hip.so: the MI355X (gfx950 / CDNA4) kernels behind
 * Synchformer.forward().
 *
 * This is the drop-in boundary of the build.  The reference has no native
 * layer of its own - every op below is reached in the reference through a torch.nn module call; each entry
 * point cites the reference call sites it replaces (paths relative to v-iashin/Synchformer).
 *
 * Conventions
 *   - plain pointers to DEVICE memory + sizes; no torch / HIP types in signatures (`stream` is a hipStream_t
 *     passed as void*; NULL = the null stream).  The caller owns every buffer; nothing here allocates.
 *   - bf16 operands are raw uint16_t bit patterns; accumulation and all statistics are fp32.
 *   - return 0 on success; non-zero = hipError_t of a failed launch, or -1 for a rejected argument.
 *     `sf_last_error()` returns a thread-local human-readable message for the last non-zero return.
 *   - every launcher is asynchronous on `stream` and re-entrant (no global mutable state but the error string).
 *   - "row map": `const int64_t map[6] = {n12, n2, sA, s1, s2, off}` sends logical row r to physical row
 *         (r / n12) * sA + ((r % n12) / n2) * s1 + (r % n2) * s2 + off          (NULL = identity)
 *     which expresses every reshape / CLS-drop / concat / transpose the reference does with view/cat/einops.
 */
#ifndef SYNCHFORMER_HIP_H
#define SYNCHFORMER_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { SF_F32 = 0, SF_BF16 = 1, SF_F16 = 2, SF_U8 = 3 };   /* element types */
enum { SF_EPI_NONE = 0, SF_EPI_GELU = 1 };                  /* GEMM epilogue activation */

#define SF_ABI_VERSION 10
int sf_abi_version(void);
const char* sf_last_error(void);
/* "gfx950" + build flags; lets the host assert it loaded the library it built */
const char* sf_build_info(void);
/* C[cmap(m), n] = epi(sum_k A[m,k] * W[n,k] + bias[n]) (+ R[rmap(m), n]);  A: M x K bf16 (row stride lda),
 * W: N x K bf16 (an nn.Linear / flattened conv weight, row stride ldw), bias fp32 or NULL, C bf16|fp32,
 * R fp32 or NULL (may alias C for an in-place residual).  K % 64 == 0.  Exact-erf GELU when SF_EPI_GELU.
 * Replaces nn.Linear at vit_helper.py:103,155,392-396; modeling_ast.py:142-146,199,263,274;
 * modules/transformer.py:59-61,74,86-91; nn.MultiheadAttention in/out proj + linear1/2 at
 * motionformer.py:329; Conv3d vit_helper.py:436-443 and Conv2d modeling_ast.py:113-117 (after sf_im2col_*);
 * vproj/aproj sync_model.py:55-56; off_head/sync_head sync_model.py:172,189. */
int sf_gemm_bf16(const uint16_t* A, int64_t lda, const uint16_t* W, int64_t ldw, const float* bias, void* C,
                 int c_dtype, int64_t ldc, const int64_t* c_map, const float* R, int64_t ldr, const int64_t* r_map,
                 int epilogue, int64_t M, int64_t N, int64_t K, void* stream);

/* Strided-batched small GEMM (train step: attention backward).  For b0 < batch_outer, b1 < batch_inner:
 * C[b0,b1] (M x N) = A[b0,b1] (M x K) * W[b0,b1]^T (N x K) + bias, X[b0,b1] = X + b0*sX0 + b1*sX1 (element strides).
 * K % 32 == 0 (zero-pad the contraction dimension); any M, N; C bf16|fp32.  Backward of the matmuls at
 * modules/transformer.py:67-70. */
int sf_gemm_bf16_batched(const uint16_t* A, int64_t lda, int64_t sA0, int64_t sA1, const uint16_t* W, int64_t ldw, int64_t sW0,
                         int64_t sW1, const float* bias, void* C, int c_dtype, int64_t ldc, int64_t sC0, int64_t sC1, int64_t M,
                         int64_t N, int64_t K, int batch_outer, int batch_inner, void* stream);

/* Split-K weight-gradient product of the train steps (autograd of every nn.Linear on the path: dW = dY^T X, e.g. the qkv / proj / fc1 / fc2
 * layers of vit_helper.py:87-141 and modeling_ast.py:199-330): part[s] (N x K, fp32) = sum over token rows m in chunk s of dY[m,:]^T X[m,:],
 * chunk s = rows [s*kc, min((s+1)*kc, M)).  dY (M x N) and X (M x K) are the row-major bf16 activations as they are - no transposed copies.
 * N % 128 == 0, K % 128 == 0, kc % 64 == 0, split*kc >= M.  The caller sums the `split` partials (sf_seqsum).
 * bias_part (split x N, fp32) or NULL: the same chunks' column sums of dY - the bias gradient of the layer - from four extra MFMAs against an
 * all-ones fragment in the workgroups that already hold those dY columns; summed by sf_seqsum(bias_part, N, split, 1, N, ...). */
int sf_gemm_tn_splitk(const uint16_t* dY, int64_t ldy, const uint16_t* X, int64_t ldx, float* part, float* bias_part, int64_t M, int64_t N,
                      int64_t K, int split, int64_t kc, void* stream);
/* The same product on the quadrant-phased 256 x 256 x 64 schedule (sf_gemm_tn_pp.hip) for the big weight gradients: N % 256 == 0, K % 256 == 0, kc % 128 == 0,
 * no empty chunk ((split - 1) * kc < M); rows beyond M contribute zero (buffer range check of the LDS-DMA).  bias_part: (split, N) fp32 or NULL - per-chunk
 * column sums of dY (the bias gradient), summed by sf_seqsum(bias_part, N, split, 1, N, ...). */
int sf_gemm_tn_pp(const uint16_t* dY, int64_t ldy, const uint16_t* X, int64_t ldx, float* part, float* bias_part, int64_t M, int64_t N, int64_t K, int split,
                  int64_t kc, void* stream);
/* dW = sum over the `split` chunk planes of `part` (n_w = N * K floats each) and, in the same launch, db (=|+=) the sum of the `split` rows of bias_part (n_b = N
 * floats each; bias_part may be NULL): the reduction behind sf_gemm_tn_splitk / sf_gemm_tn_pp.  n_w % 4 == 0, n_b % 4 == 0, 16-byte aligned. */
int sf_wgrad_sum(const float* part, int64_t n_w, int split, float* dw, const float* bias_part, int64_t n_b, float* db, int accumulate_bias, void* stream);

/* Full-row projection fused with the residual add and the NEXT LayerNorm (N = 768 fixed):
 *   X[m,:] = A[m,:] W^T + bias + R[m,:]  (fp32; X may alias R),   Y[m,:] = LayerNorm(X[m,:]) * gamma + beta  (bf16; Y may alias A).
 * A: M x K bf16, W: 768 x K bf16 (row stride ldw), or - ldw == 32 - the same weight re-laid out k-step-major as [K/32][768][32] so that the 48 KiB
 * slice of every 32-deep k-step is contiguous (full cache lines); K % 32 == 0; R / X / Y below 4 GiB.  One workgroup owns 128 complete rows, so the fp32 residual
 * stream is read once and written once per sub-layer and the separate sf_layernorm768 launch disappears.  Replaces, inside
 * DividedSpaceTimeBlock.forward (vit_helper.py:364-376), `x + temporal_fc/proj(...)` -> norm1, `x + proj(attn(...))` -> norm2 and
 * `x + mlp.fc2(...)` -> the next block's norm3. */
int sf_gemm_res_ln768(const uint16_t* A, int64_t lda, const uint16_t* W, int64_t ldw, const float* bias, const float* R, int64_t ldr,
                      float* X, int64_t ldx, const float* gamma, const float* beta, float eps, uint16_t* Y, int64_t ldy, int64_t M,
                      int64_t K, void* stream);

/* ---- MX-FP8 path of the frozen feature extractors in the synchronizability fine-tune (BASELINE configs[4]; configs/ft_synchability.yaml:7,19:
 * is_trainable False towers; the reference itself has no fp8 - fp16 autocast only, scripts/train_sync.py:178).  OCP Microscaling MXFP8: e4m3
 * elements, one E8M0 scale byte per 32 consecutive k of a row. ---- */
/* q (rows x K bytes, row stride ldq) and scales <- bf16 x (rows x K); K % 128 == 0.  Scales are STAGE-major for the GEMM's loads: plane k/128 (lds
 * BYTES apart, lds >= rows * 4) holds 4 bytes per row = the E8M0 scales of that row's four 32-blocks in the 128-deep stage. */
int sf_quantize_mxfp8(const uint16_t* x, int64_t ldx, uint8_t* q, int64_t ldq, uint8_t* scales, int64_t lds, int64_t rows, int64_t K, void* stream);
/* sf_gemm_bf16's contract (bias, exact-erf GELU, fp32 residual, bf16|fp32 output, identity row maps) on MXFP8 operands: A (M x K) / W (N x K)
 * e4m3 bytes with their stage-major scale planes sA / sW (K/128 planes, ldsa / ldsw bytes apart, 4 bytes per row, the rows of a plane PADDED to whole
 * 256-row tiles: ldsa >= ceil(M/256) * 1024, ldsw >= ceil(N/256) * 1024, 16-byte aligned - a tile's scales of one stage are fetched as one contiguous KiB); K % 128 == 0, N % 64 == 0.  v_mfma_scale_f32_32x32x64_f8f6f4, fp32 accumulate.
 * Replaces the nn.Linear calls at vit_helper.py:103,155,392-396 (qkv / proj / fc1 / fc2 of the DividedSpaceTimeBlocks) when the engine is built
 * with fp8 towers. */
int sf_gemm_mxfp8(const uint8_t* A, int64_t lda, const uint8_t* sA, int64_t ldsa, const uint8_t* W, int64_t ldw, const uint8_t* sW, int64_t ldsw,
                  const float* bias, void* C, int c_dtype, int64_t ldc, uint8_t* sC, int64_t ldsc, const float* R, int64_t ldr, int epilogue, int64_t M,
                  int64_t N, int64_t K, void* stream);
/* c_dtype SF_U8: the output itself leaves as MXFP8 (C = e4m3 bytes, sC / ldsc = its stage-major scale planes) - the fc1 + GELU hidden activations feeding
 * fc2 - quantised from the bf16-rounded value exactly as sf_quantize_mxfp8 would; N % 128 == 0, no residual. */
/* LayerNorm(768) whose output leaves as MXFP8 (the A operand of the qkv / fc1 MX GEMMs): sf_layernorm768 followed by sf_quantize_mxfp8, in one pass. */
int sf_layernorm768_mxfp8(const float* x, int64_t ldx, const float* gamma, const float* beta, uint8_t* q, int64_t ldq, uint8_t* scales, int64_t lds,
                          int64_t rows, float eps, void* stream);
/* sf_gemm_res_ln768 on MXFP8 operands with an MXFP8 output: X = dq(A) dq(W)^T + bias + R (fp32, 768 columns, X may alias R), (Y, sY) = the MXFP8
 * quantisation of bf16(LayerNorm(X) * gamma + beta) - sf_gemm_mxfp8 with the residual epilogue followed by sf_layernorm768_mxfp8, in one launch
 * (`x = x + proj(...)` / `x = x + fc2(...)` and the LayerNorm that opens the next sub-layer, vit_helper.py:364-376, in the fp8 towers).  A (M x K) / W (768 x K)
 * e4m3 bytes, K % 128 == 0; scale planes as for sf_gemm_mxfp8 with ldsa >= ceil(M/128) * 512, ldsw >= 3072, ldsy >= 4 M (6 planes).  Y / sY may alias
 * A / sA when K == 768. */
int sf_gemm_mx_res_ln768(const uint8_t* A, int64_t lda, const uint8_t* sA, int64_t ldsa, const uint8_t* W, int64_t ldw, const uint8_t* sW, int64_t ldsw,
                         const float* bias, const float* R, int64_t ldr, float* X, int64_t ldx, const float* gamma, const float* beta, float eps,
                         uint8_t* Y, int64_t ldy, uint8_t* sY, int64_t ldsy, int64_t M, int64_t K, void* stream);
/* sf_qkv_time_attention on MXFP8 operands (the temporal qkv projection of the fp8 towers with the 8-frame time attention in its epilogue): X (rows, 768) / W (2304, 768)
 * e4m3 bytes with stage-major scale planes (6 planes, one dword per row, ldsx / ldsw bytes apart); qkv_cls / out bf16 and cls_partial fp32 as in
 * sf_qkv_time_attention.  Replaces sf_gemm_mxfp8 (bf16 output) + sf_attention (time groups) + sf_attention_cls on the MX path. */
int sf_qkv_time_attention_mx(const uint8_t* X, int64_t ldx, const uint8_t* sX, int64_t ldsx, const uint8_t* W, int64_t ldw, const uint8_t* sW, int64_t ldsw,
                             const float* bias, const uint16_t* qkv_cls, int64_t ldc, uint16_t* out, int64_t ldo, float* cls_partial, int64_t n_seq, int n_groups,
                             float scale, void* stream);
/* ... with the attention output (patch rows) written as MXFP8 instead of bf16: out_q e4m3 bytes (rows, 768), row stride ldq bytes; out_s the scale planes [6][rows][4],
 * splane bytes apart - byte for byte sf_quantize_mxfp8 of sf_qkv_time_attention_mx's bf16 output.  out_q / out_s must not alias X / sX.  The CLS rows come from
 * sf_attention_cls_combine_mx on cls_partial. */
int sf_qkv_time_attention_mx_q(const uint8_t* X, int64_t ldx, const uint8_t* sX, int64_t ldsx, const uint8_t* W, int64_t ldw, const uint8_t* sW, int64_t ldsw,
                               const float* bias, const uint16_t* qkv_cls, int64_t ldc, uint8_t* out_q, int64_t ldq, uint8_t* out_s, int64_t splane,
                               float* cls_partial, int64_t n_seq, int n_groups, float scale, void* stream);

/* Tuning / test hook (state of the CALLING THREAD only; the launchers stay re-entrant): force the GEMM tile configuration of this thread's subsequent
 * sf_gemm_bf16 calls.  -1 = automatic choice by shape (default); 0 = 128x128x64, 4 waves, two workgroups per CU; 7 = persistent 256x256x64,
 * 8 waves, v_mfma_f32_32x32x16_bf16 (round 2's schedule: one 64-KiB stage of prefetch); 11 = the same tile with the quadrant-phased schedule of round 3
 * (half-tile LDS-DMA stream 1.5 stages ahead, counted waits, staggered wave groups: sf_gemm_pp.hip; K %% 128 == 0) - the automatic choice for the big
 * token GEMMs; 10 = 4 waves of 128x128; 1-6, 8, 9 = the other tilings measured in profiles/r01_gemm_configs.md (tools/bench_gemm.py). */
void sf_gemm_force_config(int cfg);
/* The configuration the automatic choice takes for a row-major, identity-mapped GEMM of this shape (0, 7 or 11): bench.py files its live launch timings
 * under the kernel symbol rocprofv3 reports for that configuration. */
int sf_gemm_bf16_auto_config(int64_t M, int64_t N, int64_t K, int has_residual);
/* fc1 of a trained MLP in one launch (Stage-1 towers; vit_helper.py Mlp / modeling_ast.py ASTIntermediate): pre = A W^T + bias and act = gelu(pre), both bf16 with
 * row stride ldc, both kept for the backward.  Returns SF_NOT_APPLICABLE (-2: no hipError_t, no argument error) without launching when the shape is outside
 * config 11's range (K % 128 == 0, K >= 256, N % 64 == 0, M, N >= 256, aligned): the caller then runs sf_gemm_bf16 + sf_gelu_fwd. */
#define SF_NOT_APPLICABLE (-2)
int sf_gemm_bf16_gelu_dual(const uint16_t* A, int64_t lda, const uint16_t* W, int64_t ldw, const float* bias, uint16_t* pre, uint16_t* act, int64_t ldc,
                           int64_t M, int64_t N, int64_t K, void* stream);
/* The same kind of hook for sf_gemm_res_ln768's main-loop schedule: -1 default (quadrant-phased, round 3), 0 = round 2's loop (one stage of prefetch),
 * 1 = quadrant-phased.  Both sum every accumulator in the same order: bit-identical outputs (the tests compare them).
 * 2 = round 4's 192-row tiles in two 384-column passes on the fused spatial kernel's main loop (row-major W, K % 128 == 0, lda and ldw multiples of
 * 64; other operands fall back to schedule 1): X differs from schedules 0/1 only by fp32 summation order, Y by at most one bf16 rounding.  The
 * environment variable SF_RL_SCHED=2 selects it when nothing is forced. */
void sf_gemm_res_ln_force_schedule(int sched);
void sf_qkv_time_force_schedule(int sched);        /* the same for sf_qkv_time_attention */
void sf_gemm_mx_force_schedule(int sched);         /* ... and for sf_gemm_mxfp8 (quadrant-phased needs K % 256 == 0) */

/* y[omap(r), :] (=|+=) LayerNorm(x[imap(r), :]) * gamma + beta over 768 columns; x fp32, y bf16|fp32.
 * Replaces nn.LayerNorm at vit_helper.py:366-375, motionformer.py:232, modeling_ast.py:301,315,535,
 * sync_model.py:157,169, modules/transformer.py:94-95 and norm1/norm2 inside motionformer.py:329. */
int sf_layernorm768(const float* x, int64_t ldx, const int64_t* in_map, const float* gamma, const float* beta, void* y,
                    int y_dtype, int64_t ldy, const int64_t* out_map, int accumulate, int64_t rows, float eps,
                    void* stream);

/* dst[(s*dst_seq_rows + l), :] = table[l, :], l < L, s < n_seq (fp32, 768 cols): lays positional tables and
 * CLS/DISTILL/OFF/MOD token rows under every sequence (video_model_builder.py:221-254, modeling_ast.py:84-90,
 * sync_model.py:153-165, motionformer.py:306-307). */
int sf_broadcast_rows768(float* dst, int64_t ld, int64_t dst_seq_rows, const float* table, int64_t L, int64_t n_seq,
                         void* stream);

/* y[r, :] = cast(x[imap(r), :]) (768 cols, x fp32, y bf16|fp32): the x[:, 0] / x[:, 1:] style slicing at
 * motionformer.py:231,332, ast.py:232-236, sync_model.py:172. */
int sf_gather_rows768(const float* x, int64_t ldx, const int64_t* in_map, void* y, int y_dtype, int64_t ldy,
                      int64_t rows, void* stream);

/* Patch gather for Conv3d(3->768, k=s=(2,16,16)) (vit_helper.py:436-444) with extract_vfeats' permute
 * (sync_model.py:74) folded in: vid (n_seg,16,3,224,224) of `dtype` -> out bf16 (n_seg*1568, 1536).
 * dtype SF_U8 additionally applies RGBToHalfToZeroOne + RGBNormalize(0.5,0.5) (dataset/transforms.py:647-669)
 * with the reference's fp16 roundings. */
int sf_im2col_video(const void* vid, int dtype, uint16_t* out, int64_t n_seg, void* stream);

/* Patch gather for Conv2d(1->768, k16, stride 10) (modeling_ast.py:113-117): spec fp32 (n_seg, F, Ta) ->
 * out bf16 (n_seg*nf*nt, 256), nf=(F-16)/10+1, nt=(Ta-16)/10+1, row = (n*nf + fi)*nt + ti. */
int sf_im2col_spec(const float* spec, uint16_t* out, int64_t n_seg, int F, int Ta, void* stream);

/* Grouped multi-head attention, out = softmax(scale * q k^T) v.  q/k/v/out are column slices of packed bf16
 * matrices (head h at columns h*head_dim..); a sequence is `seq_rows` rows; group g of a sequence owns tokens
 * row0 + g*group_stride + i*tok_stride (i < n_tok), plus row `cls_row` as an extra first key when >= 0.
 * n_tok + (cls_row>=0) <= 208; head_dim 64 or 96.  Replaces qkv_attn on the regrouped patches in
 * DividedAttention.forward (vit_helper.py:129-147), ASTSelfAttention (modeling_ast.py:156-176) and
 * SelfAttention (modules/transformer.py:67-70). */
int sf_attention(const uint16_t* q, const uint16_t* k, const uint16_t* v, int64_t ld, uint16_t* out, int64_t ldo,
                 int64_t n_seq, int64_t seq_rows, int n_groups, int row0, int group_stride, int tok_stride, int n_tok,
                 int cls_row, int heads, int head_dim, float scale, void* stream);

/* sf_attention that ALSO emits, per (sequence, head, group), the partial softmax state (m, l, o[64]; base-2 domain, fp32, 66
 * floats) of the CLS QUERY (row cls_row) over that group's keys - the CLS key itself counted in group 0 only - so the global
 * CLS row of DividedAttention (vit_helper.py:126) costs no second pass over K/V.  Needs cls_row >= 0, head_dim 64 and a free
 * query slot (n_tok <= 8 or n_tok % 16 != 0).  cls_partial: [n_seq][heads][n_groups][66]. */
int sf_attention_cls_partial(const uint16_t* q, const uint16_t* k, const uint16_t* v, int64_t ld, uint16_t* out, int64_t ldo,
                             int64_t n_seq, int64_t seq_rows, int n_groups, int row0, int group_stride, int tok_stride, int n_tok,
                             int cls_row, int heads, int head_dim, float scale, float* cls_partial, void* stream);
/* Merge those partials: out[seq*out_seq_rows + out_row, head*64 + d] = sum_p o_p[d] 2^(m_p-M) / sum_p l_p 2^(m_p-M). */
int sf_attention_cls_combine(const float* partials, int n_part, uint16_t* out, int64_t ldo, int64_t out_seq_rows, int out_row,
                             int64_t n_seq, int heads, void* stream);
/* The two above writing MXFP8 - e4m3 bytes out_q (rows, heads * 64), row stride ldq bytes, and E8M0 bytes out_s in the scale planes [heads * 64 / 128][rows][4]
 * (plane stride splane bytes): the A-operand layout of sf_gemm_mxfp8 / sf_gemm_mx_res_ln768.  Bytes and scales equal sf_quantize_mxfp8 applied to the bf16 output
 * of sf_attention_cls_partial / sf_attention_cls_combine (the fine-tuning forward's attention output is only ever that operand).  head_dim 64, an even number of
 * heads; sf_attention_cls_partial_mx: 192 <= n_tok <= 207 (the 13-key-tile space attention of DividedSpaceTimeBlock, vit_helper.py:368). */
int sf_attention_cls_partial_mx(const uint16_t* q, const uint16_t* k, const uint16_t* v, int64_t ld, uint8_t* out_q, int64_t ldq, uint8_t* out_s, int64_t splane,
                                int64_t n_seq, int64_t seq_rows, int n_groups, int row0, int group_stride, int tok_stride, int n_tok, int cls_row, int heads,
                                float scale, float* cls_partial, void* stream);
int sf_attention_cls_combine_mx(const float* partials, int n_part, uint8_t* out_q, int64_t ldq, uint8_t* out_s, int64_t splane, int64_t out_seq_rows, int out_row,
                                int64_t n_seq, int heads, void* stream);

/* The spatial half of DividedSpaceTimeBlock in ONE launch (round 4; vit_helper.py:370 `self.attn(self.norm1(x), ..., 'b (f n) d', '(b f) n d')`, DividedAttention.forward
 * vit_helper.py:97-150): qkv projection (vit_helper.py:107) of every PATCH token + the per-frame attention over [CLS key; the frame's 196 patches] in the GEMM's epilogue -
 * the 2304-wide projection never reaches HBM.  Work item = (frame, head pair): the frame's first 192 token rows x q | k | v of two heads on the matrix cores, the four
 * left-over tokens of each frame and the CLS row joined from `side`.
 * X (n_seq * 1569, 768) bf16 = norm1(x), rows [CLS; frame-major patches] per sequence; W (2304, 768) bf16 = qkv.weight, bias 2304 fp32 or NULL;
 * side (n_seq * 33, 2304) bf16 = the same projection (sf_gemm_bf16 on gathered rows) of [the CLS row; for frame f = 0..7 its tokens 192..195]: row seq * 33 and rows
 * seq * 33 + 1 + 4 f + i; out (rows as X, 768) bf16: patch rows only, must not alias X; cls_partial [n_seq][12][8][66] fp32: the CLS query's softmax partial per
 * frame, as sf_attention_cls_partial writes them (merge with sf_attention_cls_combine, n_part = 8).  n_tok must be 196.  Replaces sf_gemm_bf16 (spatial qkv) +
 * sf_attention_cls_partial (space groups). */
int sf_qkv_space_attention(const uint16_t* X, int64_t ldx, const uint16_t* W, int64_t ldw, const float* bias, const uint16_t* side, int64_t lds_,
                           uint16_t* out, int64_t ldo, float* cls_partial, int64_t n_seq, int n_tok, float scale, void* stream);
/* ... with TOKEN MASKS (round 5; Synchformer.forward(vis_mask=...), model/sync_model.py:72-80 -> the -inf key masks of vit_helper.py:107-141,
 * video_model_builder.py:185-225): key_keep holds one byte per row of X (what sf_token_mask_video writes); a row with flag 0 is a masked KEY for every query of its
 * frame's group and for the CLS query - its own output row is still computed, as in the reference.  An all-ones mask is bit-identical to sf_qkv_space_attention.
 * Replaces sf_gemm_bf16 + sf_attention_cls_partial_masked on masked forwards. */
int sf_qkv_space_attention_masked(const uint16_t* X, int64_t ldx, const uint16_t* W, int64_t ldw, const float* bias, const uint16_t* side, int64_t lds_,
                                  uint16_t* out, int64_t ldo, float* cls_partial, int64_t n_seq, int n_tok, float scale, const uint8_t* key_keep, void* stream);

/* The rows the two fused attention launches do not project themselves, gathered for their small GEMM in ONE launch (round 5; was two sf_copy_rows_bf16 launches, and three
 * torch index ops on the MXFP8 path): out row seq * 33 <- X row seq * (1 + 8 n_tok) (the CLS row), out row seq * 33 + 1 + 4 f + i <- token n_tok - 4 + i of frame f.
 * row_bytes per row (1536 = a bf16 row of 768; 768 = the e4m3 bytes of an MXFP8 operand), strides in BYTES; with sX / sOut the rows' scale dwords of n_planes stage-major
 * planes as well (plane k at sX + k ldsx, one dword per row -> sOut + k ldso).  Reference: the CLS row and the tokens of vit_helper.py:100-158 that 196 = 6 x 32 + 4 = 8 x 24
 * + 4 leaves over. */
int sf_side_rows(const void* X, int64_t ldx_bytes, void* out, int64_t ldo_bytes, int row_bytes, const uint8_t* sX, int64_t ldsx, uint8_t* sOut, int64_t ldso,
                 int n_planes, int64_t n_seq, int n_tok, void* stream);

/* sf_qkv_space_attention on MXFP8 operands (fp8 towers): X (rows, 768) / W (2304, 768) e4m3 bytes with stage-major scale planes (6 planes, one dword per row, ldsx / ldsw
 * bytes apart: what sf_gemm_mx_res_ln768 / sf_quantize_mxfp8 write); side (n_seq * 33, 2304) bf16 from sf_gemm_mxfp8 on gathered copies of the side rows and their scale
 * dwords.  Output EITHER out (bf16) OR out_q / out_s (e4m3 bytes (rows, 768), row stride ldq, + E8M0 bytes in the scale planes [6][rows][4], splane bytes apart: byte for
 * byte sf_quantize_mxfp8 of the bf16 output; buffers of their own, not X / sX); the other pointer NULL.  cls_partial as in sf_qkv_space_attention.  Replaces sf_gemm_mxfp8
 * (spatial qkv, bf16 output) + sf_attention_cls_partial(_mx) on the MX path. */
int sf_qkv_space_attention_mx(const uint8_t* X, int64_t ldx, const uint8_t* sX, int64_t ldsx, const uint8_t* W, int64_t ldw, const uint8_t* sW, int64_t ldsw,
                              const float* bias, const uint16_t* side, int64_t lds_, uint16_t* out, int64_t ldo, uint8_t* out_q, int64_t ldq, uint8_t* out_s,
                              int64_t splane, float* cls_partial, int64_t n_seq, int n_tok, float scale, void* stream);

/* sf_qkv_time_attention on the 192 x 384 main loop of sf_qkv_space_attention (round 4; same reference lines as sf_qkv_time_attention below).  Work item = (sequence,
 * block of 24 patches, head pair): the block's 24 patches x 8 frames x q | k | v of two heads on the matrix cores, then per patch the attention over [CLS key; its 8 frames];
 * the 4 left-over patches of a sequence (196 = 8 x 24 + 4) and the CLS row come from `side`, the (n_seq * 33, 2304) buffer sf_qkv_space_attention takes (row seq * 33 = the
 * CLS row's q | k | v, rows seq * 33 + 1 + 4 f + i = token 192 + i of frame f).  out (rows as X, 768) bf16: patch rows only, must not alias X; cls_partial
 * [n_seq][12][33][66] fp32: the CLS query's softmax partials, four per block (records 4 tb .. 4 tb + 3) + record 32 (the left-over patches and the CLS key itself) -
 * merge with sf_attention_cls_combine, n_part = 33.  n_tok must be 196; ldx and ldw multiples of 64 elements.  Replaces sf_gemm_bf16 (CLS rows) + sf_qkv_time_attention. */
int sf_qkv_time_attention2(const uint16_t* X, int64_t ldx, const uint16_t* W, int64_t ldw, const float* bias, const uint16_t* side, int64_t lds_,
                           uint16_t* out, int64_t ldo, float* cls_partial, int64_t n_seq, int n_tok, float scale, void* stream);
/* ... with TOKEN MASKS (round 5; same reference lines as sf_qkv_space_attention_masked): key_keep[row] == 0 masks that KEY for the queries of its patch's 8-frame group
 * and for the CLS query; an all-ones mask is bit-identical to sf_qkv_time_attention2.  Replaces sf_qkv_time_attention_masked on masked forwards. */
int sf_qkv_time_attention2_masked(const uint16_t* X, int64_t ldx, const uint16_t* W, int64_t ldw, const float* bias, const uint16_t* side, int64_t lds_,
                                  uint16_t* out, int64_t ldo, float* cls_partial, int64_t n_seq, int n_tok, float scale, const uint8_t* key_keep, void* stream);

/* sf_qkv_time_attention2 on MXFP8 operands (round 5: the temporal half of the fp8 towers on the round-4 schedule; arguments as sf_qkv_space_attention_mx, cls_partial
 * [n_seq][12][33][66]).  Output EITHER out (bf16) OR out_q / out_s (byte for byte sf_quantize_mxfp8 of the bf16 output).  Replaces sf_gemm_mxfp8 (CLS rows) +
 * sf_qkv_time_attention_mx(_q). */
int sf_qkv_time_attention2_mx(const uint8_t* X, int64_t ldx, const uint8_t* sX, int64_t ldsx, const uint8_t* W, int64_t ldw, const uint8_t* sW, int64_t ldsw,
                              const float* bias, const uint16_t* side, int64_t lds_, uint16_t* out, int64_t ldo, uint8_t* out_q, int64_t ldq, uint8_t* out_s,
                              int64_t splane, float* cls_partial, int64_t n_seq, int n_tok, float scale, void* stream);

/* The temporal half of DividedSpaceTimeBlock in ONE launch (vit_helper.py:366 `self.timeattn(self.norm3(x), ..., 'b (f n) d', '(b n) f d')`,
 * DividedAttention.forward vit_helper.py:97-150): qkv projection (vit_helper.py:107) of every PATCH token + the 8-frame attention over
 * [CLS key; the patch's 8 frames] in the GEMM's epilogue - the 2304-wide projection never reaches HBM.
 * X (n_seq * (1 + 8 * n_groups), 768) bf16 = norm3(x), rows [CLS; frame-major patches] per sequence; W (2304, 768) bf16 = qkv.weight, bias 2304 fp32
 * or NULL; qkv_cls (n_seq, 2304) bf16 = the same projection of the CLS rows (sf_gemm_bf16 on the strided CLS rows, lda = seq_rows * ldx);
 * out (rows as X, 768) bf16: patch rows only; cls_partial [n_seq][12][n_groups / 4][66] fp32, records as in sf_attention_cls_partial (one per 4
 * patches) - the CLS row of `out` (vit_helper.py:126) is then written by sf_attention_cls_combine(cls_partial, n_groups / 4, out, ...).
 * 12 heads x 64; n_groups % 4 == 0; X and W below 4 GiB. */
int sf_qkv_time_attention(const uint16_t* X, int64_t ldx, const uint16_t* W, int64_t ldw, const float* bias, const uint16_t* qkv_cls, int64_t ldc,
                          uint16_t* out, int64_t ldo, float* cls_partial, int64_t n_seq, int n_groups, float scale, void* stream);

/* One query row per sequence against n_keys consecutive rows (head_dim 64): the Motionformer CLS query
 * (vit_helper.py:126) and the only output row the aggregator layers ever read (motionformer.py:329-332). */
int sf_attention_cls(const uint16_t* q, int64_t q_seq_rows, int q_row, const uint16_t* k, const uint16_t* v, int64_t ld,
                     int64_t kv_seq_rows, int kv_row0, int n_keys, uint16_t* out, int64_t ldo, int64_t out_seq_rows,
                     int out_row, int64_t n_seq, int heads, int head_dim, float scale, void* stream);

/* Audio front-end (dataset/transforms.py:815-889; configs/sync.yaml:183-202): wave fp32 (n_seg, n_samples) ->
 * out fp32 (n_seg, n_mels, pad_to) = ((log(mel(|STFT|^2) + 1e-6), right-padded with 0.0) - mean) / (2 std), the
 * (S, 1, F, Ta) layout Synchformer.forward receives.  STFT: n_fft 1024, periodic Hann(400) centred in the frame, hop 160,
 * center/reflect.  tw_cos/tw_sin: (400, 513) twiddles with the window folded in; fb: (513, n_mels) HTK mel filterbank;
 * fb_lo/fb_hi: per-filter non-zero bin range; power_ws: fp32 workspace (n_seg * min(n_samples/hop+1, pad_to) * 513). */
int sf_mel_frontend(const float* wave, int64_t n_seg, int n_samples, int hop, const float* tw_cos, const float* tw_sin,
                    const float* fb, const int* fb_lo, const int* fb_hi, int n_mels, float* power_ws, float* out, int pad_to,
                    float mean, float std, void* stream);

/* ---- Stage-2 train step (scripts/train_sync.py:153-237, train_utils.py:373-386): backward of vproj/aproj + the sync
 * transformer and the optimizer.  GEMM-shaped work reuses sf_gemm_bf16 / sf_gemm_bf16_batched on transposed copies. ---- */

/* out[b][c][r] = in[b][r][c] (bf16), columns r in [R, R_pad) zero-filled; two-level batch strides in elements. */
int sf_transpose_bf16(const uint16_t* in, int64_t ld_in, int64_t sI0, int64_t sI1, uint16_t* out, int64_t ld_out, int64_t sO0,
                      int64_t sO1, int R, int C, int R_pad, int batch_outer, int batch_inner, void* stream);
/* n_tensors transposes of sf_transpose_bf16's wide kind (C % 8 == 0, R_pad % 8 == 0, 16-byte aligned rows) in one launch: table[t] = { in, out, ld_in, ld_out,
 * R, C, R_pad, ceil(R_pad / 64) } as int64 and tile_prefix[t] = 64 x 64 tiles before tensor t (n_tensors + 1 ints), both in DEVICE memory.  The bf16 W^T
 * operand copies of all trainable weights after an optimizer step. */
int sf_transpose_bf16_multi(const int64_t* table, const int* tile_prefix, int n_tensors, int total_tiles, void* stream);
/* y = bf16(scale * x) for a (rows, cols) fp32 matrix, cols % 4 == 0. */
int sf_cast_bf16(const float* x, int64_t ldx, uint16_t* y, int64_t ldy, int64_t rows, int cols, float scale, void* stream);
/* P[r,:L] = softmax(scale * S[r,:L]) (bf16), P[r,L:L_pad] = 0; L <= 256 (attention probabilities, modules/transformer.py:67-69). */
int sf_softmax_rows(const float* S, int64_t lds, uint16_t* P, int64_t ldp, int64_t rows, int L, int L_pad, float scale, void* stream);
/* dS = scale * P * (dP - rowsum(P * dP)) (bf16, zero-padded): backward of the softmax + 1/sqrt(d) scaling. */
int sf_softmax_bwd_rows(const uint16_t* P, int64_t ldp, const float* dP, int64_t lddp, uint16_t* dS, int64_t ldds, int64_t rows, int L,
                        int L_pad, float scale, void* stream);
/* LayerNorm(768) backward: dx[(dx_map)] (=|+=) ..., dgamma / dbeta (=|+=) column sums; statistics recomputed from x.
 * workspace: fp32, 2 * 768 * ceil(rows / 4) elements. */
int sf_layernorm768_bwd(const float* x, int64_t ldx, const int64_t* x_map, const float* gamma, const float* dy, int64_t lddy,
                        const int64_t* dy_map, float* dx, int64_t lddx, const int64_t* dx_map, int accumulate_dx, float* dgamma,
                        float* dbeta, int accumulate_dparams, float* workspace, int64_t rows, float eps, void* stream);
/* sf_layernorm768_bwd with the incoming gradient dy in bf16 (lddy % 4 == 0, 8-byte aligned rows). */
int sf_layernorm768_bwd_bf16(const float* x, int64_t ldx, const int64_t* x_map, const float* gamma, const uint16_t* dy, int64_t lddy,
                        const int64_t* dy_map, float* dx, int64_t lddx, const int64_t* dx_map, int accumulate_dx, float* dgamma,
                        float* dbeta, int accumulate_dparams, float* workspace, int64_t rows, float eps, void* stream);
/* sf_layernorm768_bwd (identity row maps, dy fp32 or bf16 by dy_dtype) that also writes the head of the NEXT residual branch's backward from the updated dx - what
 * sf_branch_grad would compute in another pass over it: y_next = bf16(seq_scale[row / seq_rows] * dx) (seq_scale NULL = 1: no stochastic depth on that branch) and
 * dbias_next[c] = sum_r seq_scale * dx[r, c] (fp32, assigned).  workspace: fp32, 3 * 768 * ceil(rows / 4) elements. */
int sf_layernorm768_bwd_branch(const float* x, int64_t ldx, const float* gamma, const void* dy, int dy_dtype, int64_t lddy, float* dx, int64_t lddx,
                               int accumulate_dx, float* dgamma, float* dbeta, int accumulate_dparams, uint16_t* y_next, int64_t ldyn, const float* seq_scale,
                               int64_t seq_rows, float* dbias_next, float* workspace, int64_t rows, float eps, void* stream);
/* out[c] (=|+=) sum_r x[r, c] (bias gradients); x fp32|bf16; workspace fp32 cols * ceil(rows / 64). */
int sf_colsum(const void* x, int x_dtype, int64_t ldx, int64_t rows, int cols, float* out, int accumulate, float* workspace, void* stream);
/* out[l, c] (=|+=) sum_b x[b*L + l, c]: gradient of the broadcast positional / token table. */
int sf_seqsum(const float* x, int64_t ldx, int n_seq, int L, int cols, float* out, int accumulate, void* stream);
/* act = gelu_erf(pre) / dpre = dact * gelu_erf'(pre) on bf16 pre-activations (n elements, n % 4 == 0). */
int sf_gelu_fwd(const uint16_t* pre, uint16_t* act, int64_t n, void* stream);
int sf_gelu_bwd(const uint16_t* pre, const float* dact, uint16_t* dpre, int64_t n, void* stream);
/* sf_gelu_bwd with the incoming gradient in bf16 (n % 8 == 0, 16-byte aligned pointers). */
int sf_gelu_bwd_bf16(const uint16_t* pre, const uint16_t* dact, uint16_t* dpre, int64_t n, void* stream);
/* loss = mean cross-entropy of (B, C) logits vs int64 targets (sync_model.py:95-96); dlogits (optional) scaled by grad_scale / B. */
int sf_cross_entropy(const float* logits, int64_t ld, const int64_t* targets, int B, int C, float* loss, float* dlogits, int64_t ldd,
                     float grad_scale, void* stream);
/* y = dropout_p(x) (+ residual): keep(i) = hash(seed, i) >= p * 2^32 over the linear element index i, kept values scaled by
 * 1/(1-p); x, y fp32|bf16 (same dtype), residual fp32 or NULL (fp32 x only).  The same (seed, shape) regenerates the mask in
 * the backward.  Dropout sites: sync_model.py:166, modules/transformer.py:70,73,90. */
int sf_dropout(const void* x, int dtype, int64_t ldx, const float* residual, int64_t ldr, void* y, int64_t ldy, int64_t rows, int cols,
               float p, uint32_t seed, void* stream);
/* Whole-token dropout of the sync transformer's inputs - GlobalTransformer.tok_drop_vis / tok_drop_aud, torch.nn.Dropout1d on (B, S, D) = whole tokens dropped
 * (model/sync_model.py:131-134, 160-161, train mode with tok_pdrop > 0): y[y_map(r), :] (=|+=) row_scale[r] * x[x_map(r), :], fp32, cols % 4 == 0; row_scale[r] is 0 or
 * 1 / (1 - p) per token (sf_dropout over a vector of ones); x_map / y_map = 6-integer row maps or NULL (identity).  Forward: the input LayerNorm's rows into the token
 * matrix on top of the positional table (accumulate = 1); backward: the token matrix's gradient rows, scaled, as the LayerNorm backward's dY (accumulate = 0). */
int sf_scale_rows_map(const float* x, int64_t ldx, const int64_t* x_map, const float* row_scale, float* y, int64_t ldy, const int64_t* y_map, int64_t rows, int cols,
                      int accumulate, void* stream);
/* Head of a residual branch's backward in the Stage-1 towers (the `x = x + drop_path(branch(x))` sites of vit_helper.py:364-376 / modeling_ast.py ASTLayer):
 * y = bf16(s[r / seq_rows] * dx[r, :]) (the dY operand of the branch's output Linear) and dbias (=|+=) the fp32 column sums of the scaled gradient, in one
 * pass over dx (seq_scale may be NULL).  Replaces sf_scale_seq_add -> sf_cast_bf16 -> sf_colsum.  cols % 32 == 0; workspace >= cols * ceil(rows / 64) floats. */
int sf_branch_grad(const float* dx, int64_t ldx, const float* seq_scale, int64_t seq_rows, uint16_t* y, int64_t ldy, int64_t rows, int cols, float* dbias,
                   int accumulate, float* workspace, void* stream);
/* x = residual + seq_scale[r / seq_rows] * branch (fp32; seq_scale may be NULL = 1) and y = bf16(LayerNorm(x) * gamma + beta) in one pass: the stochastic-depth
 * residual add and the next sub-layer's norm of the Stage-1 forward (vit_helper.py:364-376); sf_scale_seq_add followed by sf_layernorm768. */
int sf_add_scale_ln768(const float* branch, int64_t ldb, const float* seq_scale, int64_t seq_rows, const float* residual, int64_t ldr, float* x, int64_t ldx,
                       const float* gamma, const float* beta, uint16_t* y, int64_t ldy, int64_t rows, float eps, void* stream);
/* Stochastic depth of the Stage-1 visual tower (DropPath at vit_helper.py:356,372,375, rates video_model_builder.py:86-87 = linspace(0, 0.2, 12)):
 * y[r,:] = (residual ? residual[r,:] : 0) + seq_scale[r / seq_rows] * x[r,:], fp32, cols % 4 == 0; seq_scale[i] is 0 or 1/keep_prob per
 * segment.  The forward applies it to a residual branch, the backward to the incoming gradient with the same scales. */
int sf_scale_seq_add(const float* x, int64_t ldx, const float* seq_scale, int64_t seq_rows, const float* residual, int64_t ldr, float* y,
                     int64_t ldy, int64_t rows, int cols, void* stream);
/* norm_out[0] = ||g||_2 of a flat fp32 buffer (deterministic two-stage; workspace fp32 1024). */
int sf_grad_norm(const float* g, int64_t n, float* norm_out, float* workspace, void* stream);
/* clip_grad_norm_(max_norm, device-resident norm) + Adam(betas, eps, no weight decay) on flat fp32 buffers; also writes the
 * bf16 copy of the updated parameters when p_bf16 != NULL.  step >= 1 is the Adam step count (bias correction). */
int sf_adam_clip_step(float* p, const float* g, float* m, float* v, uint16_t* p_bf16, int64_t n, const float* norm, float max_norm,
                      float lr, float beta1, float beta2, float eps, int step, void* stream);

/* ---- Stage-1 segment-level contrastive head (train_clip_src/open_clip/model.py:449-585) ------------------------------ */
/* y[r, :768] = mean_j x[r*t + j, :768], then (normalize != 0) divided by max(||.||_2, 1e-12):  AveragePooling 'BS t D -> BS D'
 * (motionformer.py:395-409, ast.py:87-88) and F.normalize (open_clip/model.py:530-531); fp32, row strides % 4 == 0. */
int sf_meanpool_l2norm768(const float* x, int64_t ldx, int t, float* y, int64_t ldy, int normalize, int64_t n, void* stream);
/* out[i, j] = scale * <a_i, b_j> for fp32 a (n, d), b (m, d), d % 16 == 0: sim = feat @ feat_all^T / logit_scale
 * (open_clip/model.py:508-509); kept in fp32 because scale can be as large as 1 / 0.001. */
int sf_similarity_f32(const float* a, int64_t lda, const float* b, int64_t ldb, float* out, int64_t ldo, int n, int m, int d, float scale,
                      void* stream);

/* ---- feature-extractor backward helpers (Stage-1 training; the matmul-shaped parts reuse the GEMM entry points) ------------ */
/* dst[dst_map(r), :cols] = src[src_map(r), :cols] for r < rows (bf16, cols % 8 == 0): builds / scatters the per-group
 * sequences [CLS; group tokens] of divided space-time attention (vit_helper.py:100-158, 341-344). */
int sf_copy_rows_bf16(const uint16_t* src, int64_t ld_src, const int64_t* src_map, uint16_t* dst, int64_t ld_dst, const int64_t* dst_map,
                      int64_t rows, int cols, void* stream);
/* out[s*out_seq_stride + c] (=|+=) sum_{g<G} in[s*in_seq_stride + g*in_group_stride + c]: the CLS row is a key/value of every
 * group, so its dk|dv is the sum over groups (strides in elements). */
int sf_reduce_groups_bf16(const uint16_t* in, int64_t in_seq_stride, int64_t in_group_stride, int G, uint16_t* out, int64_t out_seq_stride,
                          int cols, int64_t n_seq, int accumulate, void* stream);
/* Backward of sf_attention_cls (same addressing): dO row (seq*do_seq_rows + do_row) -> dq at the query row (=), dk / dv at the
 * n_keys key rows (=|+=); head_dim 64, n_keys <= 2048. */
int sf_attention_cls_bwd(const uint16_t* q, int64_t q_seq_rows, int q_row, const uint16_t* k, const uint16_t* v, int64_t ld,
                         int64_t kv_seq_rows, int kv_row0, int n_keys, const uint16_t* dO, int64_t lddo, int64_t do_seq_rows, int do_row,
                         uint16_t* dq, uint16_t* dk, uint16_t* dv, int64_t ldg, int64_t n_seq, int heads, int head_dim, float scale,
                         int accumulate_kv, void* stream);
/* Backward of sf_meanpool_l2norm768: dx[r*t + j, :] = d(mean -> normalize)/dx applied to dy[r, :] (x = the forward input). */
int sf_meanpool_l2norm768_bwd(const float* x, int64_t ldx, int t, const float* dy, int64_t lddy, float* dx, int64_t lddx, int normalize,
                              int64_t n, void* stream);

/* ---- device-side segmenting (dataset/transforms.py:402-500 GenerateMultipleSegments; SURVEY §8f "next" rank 1) ------------------
 * The segments of a clip overlap by 50 % (step_size_seg 0.5): instead of materialising (S, 16, C, H, W) / (S, n_samples) copies on
 * the host and shipping 1.8x the bytes over PCIe, the two front-end gathers read the segment windows straight from the clip:
 * segment (clip, s) = frames [frame0 + s*seg_stride, +16) of vid (n_clips, clip_frames, 3, 224, 224), and samples
 * [sample0 + s*seg_stride, +n_samples) of wave (n_clips, clip_samples).  Reflect padding of the STFT stays inside a segment. */
int sf_im2col_video_clips(const void* vid, int dtype, int64_t n_clips, int64_t clip_frames, int frame0, int seg_stride, int n_seg,
                          uint16_t* out, void* stream);
/* ... into the TOKEN layout: out bf16 (n_clips * n_seg * 1569, 1536), segment n's 1568 patch rows at n * 1569 + 1 .., its row 0 (the CLS slot) zeroed - the
 * patch-embedding GEMM (Conv3d of PatchEmbed3D, motionformer.py) then runs with identity row maps straight onto the (n * 1569, 768) token matrix. */
int sf_im2col_video_tokens(const void* vid, int dtype, int64_t n_clips, int64_t clip_frames, int frame0, int seg_stride, int n_seg, uint16_t* out, void* stream);
int sf_mel_frontend_clips(const float* wave, int64_t n_clips, int64_t clip_samples, int64_t sample0, int64_t seg_stride, int n_seg,
                          int n_samples, int hop, const float* tw_cos, const float* tw_sin, const float* fb, const int* fb_lo,
                          const int* fb_hi, int n_mels, float* power_ws, float* out, int pad_to, float mean, float std, void* stream);

/* ---- token masks (Synchformer.forward(vis_mask=, aud_mask=), sync_model.py:38-89) -------------------------------
 * content_keep: bool bytes shaped like the input ((n,16,3,224,224) / (n,F,Ta)), 1 = kept.  tok_keep[n*L + t] = 0 iff the masked
 * elements of token t's patch meet filter-0 weights of both signs or a zero weight (the reference's inf -> NaN trick,
 * video_model_builder.py:185-201, modeling_ast.py:515-530); w0_sign = sign of filter 0 per patch element (int8), CLS/DISTILL kept. */
int sf_token_mask_video(const uint8_t* content_keep, int64_t n_seg, const int8_t* w0_sign, uint8_t* tok_keep, void* stream);
int sf_token_mask_spec(const uint8_t* content_keep, int64_t n_seg, int F, int Ta, const int8_t* w0_sign, uint8_t* tok_keep, void* stream);
/* sf_attention / sf_attention_cls with a key mask: key_keep[row] == 0 gives K/V row `row` (same row indexing as k) a score of -inf
 * (qkv_attn tok_mask, vit_helper.py:34-42; ASTSelfAttention, modeling_ast.py:160-163; aggregators' src_mask, motionformer.py:308-329). */
int sf_attention_masked(const uint16_t* q, const uint16_t* k, const uint16_t* v, int64_t ld, uint16_t* out, int64_t ldo, int64_t n_seq,
                        int64_t seq_rows, int n_groups, int row0, int group_stride, int tok_stride, int n_tok, int cls_row, int heads,
                        int head_dim, float scale, const uint8_t* key_keep, void* stream);
int sf_attention_cls_masked(const uint16_t* q, int64_t q_seq_rows, int q_row, const uint16_t* k, const uint16_t* v, int64_t ld,
                            int64_t kv_seq_rows, int kv_row0, int n_keys, uint16_t* out, int64_t ldo, int64_t out_seq_rows, int out_row,
                            int64_t n_seq, int heads, int head_dim, float scale, const uint8_t* key_keep, void* stream);
/* The masked forward on the fused schedules (round 3): sf_attention_cls_partial and sf_qkv_time_attention with the same key flags (one byte per K/V /
 * token row, 0 = masked: -inf before the softmax, for the patch queries and for the CLS query's partial records; a record whose keys are all masked is
 * (m = -inf, l = 0, o = 0) and drops out in sf_attention_cls_combine).  Reference: vit_helper.py:34-42, 107-141, sync_model.py:72-89. */
int sf_attention_cls_partial_masked(const uint16_t* q, const uint16_t* k, const uint16_t* v, int64_t ld, uint16_t* out, int64_t ldo, int64_t n_seq,
                                    int64_t seq_rows, int n_groups, int row0, int group_stride, int tok_stride, int n_tok, int cls_row, int heads,
                                    int head_dim, float scale, float* cls_partial, const uint8_t* key_keep, void* stream);
int sf_qkv_time_attention_masked(const uint16_t* X, int64_t ldx, const uint16_t* W, int64_t ldw, const float* bias, const uint16_t* qkv_cls, int64_t ldc,
                                 uint16_t* out, int64_t ldo, float* cls_partial, int64_t n_seq, int n_groups, float scale, const uint8_t* key_keep,
                                 void* stream);

/* Stage-1 zero-shot sync check (shift_and_get_preds, train_clip_src/training/train.py:549-579): G = audio-to-video segment similarity
 * (n_clips*S x n_clips*S, fp32, clip-major); per clip, window sims are W-long diagonal sums of its S x S block;
 * preds_a[b, j] = argmax_i, preds_v[b, i] = argmax_j over the S - W + 1 (<= 32) shifts. */
int sf_shift_window_preds(const float* G, int64_t ldg, int n_clips, int S, int W, int64_t* preds_a, int64_t* preds_v, void* stream);

/* Backward of sf_attention for tiny groups (n_tok <= 8, head_dim 64: Motionformer time attention, vit_helper.py:343-344): same
 * addressing as the forward; dq | dk | dv rows of the group's tokens are written (=), the CLS key's dk | dv of every (seq, group) goes to
 * cls_part (n_seq * n_groups, 2 * heads * 64) bf16 for sf_reduce_groups_bf16. */
int sf_attention_tiny_bwd(const uint16_t* q, const uint16_t* k, const uint16_t* v, int64_t ld, const uint16_t* dO, int64_t lddo, uint16_t* dq,
                          uint16_t* dk, uint16_t* dv, int64_t ldg, uint16_t* cls_part, int64_t n_seq, int64_t seq_rows, int n_groups, int row0,
                          int group_stride, int tok_stride, int n_tok, int cls_row, int heads, int head_dim, float scale, void* stream);

/* out[r] (=|+=) sum_c x[r, c] for a bf16 (rows, cols) matrix (cols % 8 == 0): bias gradients from the transposed gradient copy. */
int sf_rowsum_bf16(const uint16_t* x, int64_t ldx, int rows, int64_t cols, float* out, int accumulate, void* stream);

/* Fused backward of sf_attention for groups of up to 208 keys, head_dim 64 (Motionformer space attention, AST): same addressing and
 * outputs as sf_attention_tiny_bwd; scores / probabilities are recomputed, nothing but dq | dk | dv (and cls_part) touches HBM. */
int sf_attention_group_bwd(const uint16_t* q, const uint16_t* k, const uint16_t* v, int64_t ld, const uint16_t* dO, int64_t lddo, uint16_t* dq,
                           uint16_t* dk, uint16_t* dv, int64_t ldg, uint16_t* cls_part, int64_t n_seq, int64_t seq_rows, int n_groups, int row0,
                           int group_stride, int tok_stride, int n_tok, int cls_row, int heads, int head_dim, float scale, void* stream);

/* sf_attention_group_bwd with the CLS QUERY's backward (sf_attention_cls_bwd's job: vit_helper.py:126, the CLS query attends every key of the sequence) in the same
 * launch: the CLS query rides as one more query row of every group, normalised with the forward's statistics cls_stats [n_seq][heads][2] = (m in the base-2 domain
 * incl. the scale, l) from sf_attention_cls_combine_stats, delta = <dO, o> from the forward's output row o[(seq * seq_rows + cls_row) * ldo + ...].  dk / dv (and
 * cls_part) then hold both contributions - no read-modify-write pass over them - and the CLS query's dq comes out as one partial row per group, dq_cls_part
 * (n_seq * n_groups, heads * 64) bf16: sf_reduce_groups_bf16 sums them into row cls_row of dq.  n_tok % 16 != 0 (a free query slot). */
int sf_attention_group_bwd_clsq(const uint16_t* q, const uint16_t* k, const uint16_t* v, int64_t ld, const uint16_t* dO, int64_t lddo, uint16_t* dq,
                                uint16_t* dk, uint16_t* dv, int64_t ldg, uint16_t* cls_part, const float* cls_stats, const uint16_t* o, int64_t ldo,
                                uint16_t* dq_cls_part, int64_t n_seq, int64_t seq_rows, int n_groups, int row0, int group_stride, int tok_stride, int n_tok,
                                int cls_row, int heads, int head_dim, float scale, void* stream);
/* The same for the tiny time groups (sf_attention_tiny_bwd; statistics from sf_attention_cls_stats = sf_attention_cls + stats[(seq * heads + head) * 2 + {0, 1}]). */
int sf_attention_tiny_bwd_clsq(const uint16_t* q, const uint16_t* k, const uint16_t* v, int64_t ld, const uint16_t* dO, int64_t lddo, uint16_t* dq, uint16_t* dk,
                               uint16_t* dv, int64_t ldg, uint16_t* cls_part, const float* cls_stats, const uint16_t* o, int64_t ldo, uint16_t* dq_cls_part,
                               int64_t n_seq, int64_t seq_rows, int n_groups, int row0, int group_stride, int tok_stride, int n_tok, int cls_row, int heads,
                               int head_dim, float scale, void* stream);
int sf_attention_cls_stats(const uint16_t* q, int64_t q_seq_rows, int q_row, const uint16_t* k, const uint16_t* v, int64_t ld, int64_t kv_seq_rows, int kv_row0,
                           int n_keys, uint16_t* out, int64_t ldo, int64_t out_seq_rows, int out_row, int64_t n_seq, int heads, int head_dim, float scale,
                           float* stats, void* stream);
/* sf_attention_cls_combine that also writes the merged softmax statistics stats[(seq * heads + head) * 2 + {0, 1}] = (M, L) of the CLS query. */
int sf_attention_cls_combine_stats(const float* partials, int n_part, uint16_t* out, int64_t ldo, int64_t out_seq_rows, int out_row, int64_t n_seq, int heads,
                                   float* stats, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SYNCHFORMER_HIP_H */
