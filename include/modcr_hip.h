/* libmodcr_hip -- C ABI of the MI355X-native ModCR hot path (gfx950 / CDNA4).
 *
 * The reference (YunxinLi/Multimodal-Context-Reasoning) has no FFI: its boundary for this path is
 * the Python class API of modeling/ (*.py).  The drop-in classes in
 * multimodal-context-reasoning_amd/modeling/ keep that API and route every piece of arithmetic
 * through the entry points below (ctypes; see INTEGRATION.md for the binding a maintainer adds).
 * Each entry names the reference code it replaces (paths relative to the reference root;
 * a_bert = a_transformers.zip!a_transformers/modeling_bert.py, v10 =
 * modeling/modeling_vcr_chunkalign_v10.py).
 *
 * Conventions
 *  - plain pointers + sizes, no torch types; all pointers are DEVICE pointers unless noted.
 *  - the caller owns every buffer (inputs, outputs, workspaces); the library allocates nothing and
 *    enqueues all work on `stream` (a hipStream_t) without synchronising.  Re-entrant from several
 *    host threads: the only process-wide state is a set of write-once flags / constants (one
 *    hipFuncSetAttribute per kernel that needs > 64 KB of LDS, the device's CU count), idempotent
 *    and benign when raced; the last-error string is thread-local.
 *  - the product library (libmodcr_hip.so) reads NO environment variable.  A/B and timing-only
 *    ablation knobs exist only in libmodcr_hip_tuning.so (same sources, -DMODCR_TUNING; csrc/common.h).
 *  - return 0 (MODCR_OK) or a negative error code; modcr_last_error() gives a thread-local message.
 *  - `dtype` selects the storage type of activations AND weight matrices: MODCR_BF16 (bf16
 *    storage, fp32 accumulate, MFMA) or MODCR_F32 (exact-fp32 parity path).  Biases, LayerNorm
 *    affine parameters, embedding tables, masks, probabilities and losses are always fp32.
 *  - weight matrices are torch.nn.Linear layout: W[out_features][in_features], row-major.
 *  - masks are the reference's 0/1 float masks (1 = attend); the kernels apply the reference's
 *    additive -10000 (modeling_transfomres.py:641) themselves.
 */
#ifndef MODCR_HIP_H
#define MODCR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MODCR_VERSION 100

enum { MODCR_OK = 0, MODCR_ERR_INVALID = -1, MODCR_ERR_LAUNCH = -2, MODCR_ERR_UNSUPPORTED = -3 };
enum { MODCR_BF16 = 0, MODCR_F32 = 1, MODCR_F16 = 2 /* IEEE half: output of modcr_linear_fwd / input of the LayerNorm passes only */ };
enum { MODCR_ACT_NONE = 0, MODCR_ACT_GELU = 1, MODCR_ACT_TANH = 2 };

typedef void* modcr_stream_t; /* hipStream_t */

int modcr_version(void);
const char* modcr_last_error(void);

/* ---------------------------------------------------------------------------------------------
 * Fused QKV projection + masked scaled-dot-product attention (one launch per encoder layer).
 * Replaces CaptionBertSelfAttention.forward: modeling/modeling_bert.py:34-75 (global_enc) and
 * v10:55-107 (seq_enc incl. do_chunk_cross, v10:66-78).
 *
 *   x        [N,S,H]    hidden states
 *   hist     [N,P,H]    history_state / prefix rows prepended for K and V only (P may be 0 -> NULL)
 *   wqkv     [3H,H]     rows 0..H-1 = query.weight, H..2H-1 = key.weight, 2H..3H-1 = value.weight
 *   bqkv     [3H] fp32  the three biases, same order
 *   key_mask [N,P+S] fp32 0/1   broadcast padding mask (attention_mask of BertImgModel.forward)
 *   dense_mask_bits [N,S,LW] u32, LW = ceil((P+S)/32): bit j of row i = query i may see key j;
 *            NULL = broadcast key mask only.  When given, key_mask is NOT applied on top (the
 *            reference's phase masks already contain it, v10:179-206).
 *   chunk_id [N,T] int32: chunk index of text token t (row t of x), -1 = leave the query row
 *            alone; NULL = no chunk-mean query.  Rows with the same id get the mean of their
 *            query rows (v10:66-78).
 *   ctx      [N,S,H]    merged-head context (output)
 *   probs    [N,A,S,P+S] fp32 softmax probabilities, or NULL (output_attentions, modeling_bert.py:74)
 *   align_map [N,T,R] fp32, or NULL: += sum over heads of probs[:, :, :T, P+T:P+T+R] (atomic; the
 *            caller zeroes it; consumed at v10:982).  T = align_t, R = S - T.
 *   H = A*64 (head size 64: BERT-base/large, Oscar-base/large).  P+S <= 256.
 */
int modcr_qkv_attn_fwd(const void* x, const void* hist, const void* wqkv, const float* bqkv,
                       const float* key_mask, const uint32_t* dense_mask_bits,
                       const int32_t* chunk_id, int32_t chunk_t, void* ctx, float* probs,
                       float* align_map, int32_t align_t, int32_t N, int32_t S, int32_t P,
                       int32_t H, int32_t A, void* workspace, int64_t workspace_bytes, int32_t dtype,
                       modcr_stream_t stream);
/* The same with nn.Dropout on the attention probabilities (modeling_bert.py:69 / v10:101, training mode; p comes from the
 * checkpoint's config.attention_probs_dropout_prob): the context rows use the masked, 1/(1-p)-scaled probabilities, the
 * align map the unmasked ones.  Counter-based mask from (seed, offset), one hash per four consecutive keys of a query
 * row (counter row length: the tile kernels' token tile, 128 or 192, for 64 < S <= 192 with P = 0, an even head count and
 * H % 128 == 0; 256 otherwise).  bf16 path, probs = NULL; MODCR_ERR_UNSUPPORTED otherwise.  attn_p = 0: modcr_qkv_attn_fwd. */
int modcr_qkv_attn_dropout_fwd(const void* x, const void* hist, const void* wqkv, const float* bqkv,
                               const float* key_mask, const uint32_t* dense_mask_bits, const int32_t* chunk_id,
                               int32_t chunk_t, void* ctx, float* probs, float* align_map, int32_t align_t, int32_t N,
                               int32_t S, int32_t P, int32_t H, int32_t A, float attn_p, uint64_t seed, uint64_t offset,
                               void* workspace, int64_t workspace_bytes, int32_t dtype, modcr_stream_t stream);
/* The same with one more output for a layer that will be differentiated (the trainable-encoder variants, the prefix
 * RoBERTa body): lse [N, A, S] fp32 = log2 of every query row's sum of exp2(log2e x (q.k / 8 + mask)), the row statistics
 * modcr_qkv_attn_lse_bwd rebuilds the attention probabilities of modeling_bert.py:57-66 from instead of recomputing row maxima
 * and sums.  Written by the tile kernels (bf16, 64 < P + S <= 256 on their shapes); MODCR_ERR_UNSUPPORTED on any other route.
 * lse = NULL: modcr_qkv_attn_dropout_fwd.
 * qkv_dump (or NULL; with lse, P = 0, 64 < S <= 192): receives the Q | K | V images the kernel held in LDS, as plain rows
 * [N][A][3][LP][64] bf16 (LP = 128 for S <= 128, else 192; modcr_qkv_attn_dump_bytes) -- Q scaled by log2e / 8 and
 * chunk-averaged exactly as the softmax saw it -- so that modcr_qkv_attn_lse_bwd need not recompute the projections. */
int64_t modcr_qkv_attn_dump_bytes(int32_t N, int32_t S, int32_t A);
int modcr_qkv_attn_lse_fwd(const void* x, const void* hist, const void* wqkv, const float* bqkv,
                           const float* key_mask, const uint32_t* dense_mask_bits, const int32_t* chunk_id,
                           int32_t chunk_t, void* ctx, float* probs, float* align_map, int32_t align_t, float* lse, void* qkv_dump, int32_t N,
                           int32_t S, int32_t P, int32_t H, int32_t A, float attn_p, uint64_t seed, uint64_t offset,
                           void* workspace, int64_t workspace_bytes, int32_t dtype, modcr_stream_t stream);
/* The most general forward entry: modcr_qkv_attn_lse_fwd + flags.
 *   MODCR_ATTN_SIDE_POST_DROPOUT  the side outputs (probs, align_map) leave AFTER the attention-probability dropout,
 *       P o m / (1 - p): what the reference's modules return in training mode (modeling_bert.py:69-74 and
 *       modeling_vcr_chunkalign_v10.py:94-106 apply self.dropout before `outputs = (context_layer, attention_probs)`;
 *       ChunkAlign_CLS_enc4_align sums them into its align loss, v10:1067-1075).  Without the flag the side outputs are the
 *       un-dropped probabilities (same expectation) and a probabilities output under dropout is refused.  Tile kernels only
 *       (bf16, 64 < P + S <= 256 on their shapes); irrelevant when attn_p = 0. */
#define MODCR_ATTN_SIDE_POST_DROPOUT 1
int modcr_qkv_attn_opt_fwd(const void* x, const void* hist, const void* wqkv, const float* bqkv,
                           const float* key_mask, const uint32_t* dense_mask_bits, const int32_t* chunk_id,
                           int32_t chunk_t, void* ctx, float* probs, float* align_map, int32_t align_t, float* lse, void* qkv_dump, int32_t N,
                           int32_t S, int32_t P, int32_t H, int32_t A, float attn_p, uint64_t seed, uint64_t offset, int32_t flags,
                           void* workspace, int64_t workspace_bytes, int32_t dtype, modcr_stream_t stream);
/* Measurement hook: the NEXT bf16 modcr_qkv_attn_fwd / _dropout_fwd launch issued by the calling thread stamps the two
 * hipEvent_t (caller-created, timing enabled) at the start and the end of its kernel (hipExtLaunchKernel) -- the kernel's
 * own duration, as rocprofv3 --kernel-trace reports it, without the two extra barrier packets of a hipEventRecord pair.
 * Consumed by that one launch; (NULL, NULL) clears it.  bench.py's `roofline` uses it. */
int modcr_time_next_attn(void* start_event, void* stop_event);
/* bytes of `workspace` modcr_qkv_attn_fwd needs (0 for the fused bf16 path) */
int64_t modcr_qkv_attn_workspace(int32_t N, int32_t S, int32_t P, int32_t H, int32_t dtype);

/* Chunk-mean query on its own (v10:66-78): q [N,S,H] in place, chunk_id [N,T] as above. */
int modcr_chunk_mean_q_fwd(void* q, int64_t row_stride, int64_t seq_stride, const int32_t* chunk_id,
                           int32_t N, int32_t T, int32_t H, int32_t dtype, modcr_stream_t stream);

/* Build the seq_enc phase-1 / phase-3 masks of CaptionBertEncoder.forward (v10:179-206) as bits:
 *   input_mask [N,T+R] fp32 0/1, chunk_mask [N,T,T] fp32 0/1 -> bits [N,S,ceil(S/32)] u32.
 *   phase 1: text->text chunk mask, text->image pad, image->text blocked, image->image pad.
 *   phase 3: text->text chunk mask, text->image pad, image row i sees only itself.            */
int modcr_build_phase_mask(const float* input_mask, const float* chunk_mask, uint32_t* bits,
                           int32_t N, int32_t T, int32_t R, int32_t phase, modcr_stream_t stream);
/* Short sequences packed k to a row block (the image-only pass of global_enc: S = 1 + R rows per sequence, modeling_ensemble.py:466-471):
 * x [N, S, H] viewed as [N / k, k S, H] is ONE attention problem per block under the block-diagonal mask built here --
 *   key_mask [N, S] fp32 0/1 -> bits [N / k, k S, ceil(k S / 32)] u32: key j visible to query i iff same sequence and key_mask != 0 --
 * so that the token tiles of modcr_qkv_attn_fwd are filled (k S <= 192: two heads per workgroup; <= 256: one).  Every other op of the
 * layer is row-wise and does not see the packing. */
int modcr_build_packed_mask(const float* key_mask, uint32_t* bits, int32_t N, int32_t S, int32_t k, modcr_stream_t stream);
/* Generic 0/1 float mask [rows, L] -> bits [rows, ceil(L/32)]. */
int modcr_pack_mask_bits(const float* mask, uint32_t* bits, int64_t rows, int32_t L,
                         modcr_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * C[M,N] = act(A[M,K] . W[N,K]^T + bias) (+ residual).  nn.Linear + activation epilogue.
 * Replaces BertIntermediate (a_bert:425-437, act = GELU-erf), BertPooler (a_bert:634-646, tanh),
 * img_embedding (modeling_transfomres.py:676), the mapping networks (modeling_ensemble.py:439-457),
 * cls_ensemble_1 (v10:912) and the q/k/v/out projections of cross_attention_lyx (v10:710-729,796).
 * lda/ldw/ldr/ldc are row strides in ELEMENTS.  out_dtype may differ from dtype (fp32 out of a
 * bf16 GEMM feeds LayerNorm).  bf16 path: K a multiple of 64 (zero-pad with modcr_cast_pad), lda, ldw multiples of 8, 16-byte aligned bases.
 */
int modcr_linear_fwd(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias,
                     const void* residual, int64_t ldr, int32_t res_dtype, void* C, int64_t ldc,
                     int32_t M, int32_t N, int32_t K, int32_t act, int32_t dtype, int32_t out_dtype,
                     modcr_stream_t stream);
/* The same product for the few-row GEMMs of the trainable heads (M = 256: one row of output tiles would fill 3-20 of the
 * 256 CUs): split-K work items into fp32 partials in the caller's workspace, then one pass that sums them and applies bias /
 * activation.  modcr_linear_splitk_workspace returns the bytes needed, 0 when the shape has no plan (use modcr_linear_fwd).
 * bf16 operands, no residual. */
int64_t modcr_linear_splitk_workspace(int32_t M, int32_t N, int32_t K);
int modcr_linear_splitk_fwd(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, void* C, int64_t ldc,
                            int32_t M, int32_t N, int32_t K, int32_t act, int32_t out_dtype, void* workspace,
                            int64_t workspace_bytes, modcr_stream_t stream);
int modcr_ffn_up_gelu_fwd(const void* x, const void* w1, const float* b1, void* out, int32_t M,
                          int32_t H, int32_t I, int32_t dtype, modcr_stream_t stream);

/* out = LayerNorm(A.W^T + bias + residual) * gamma + beta.
 * Replaces BertSelfOutput (a_bert:362-373) and BertOutput (a_bert:440-451).  `workspace` holds the
 * fp32 pre-LayerNorm rows: M*N*4 bytes.  eps = config.layer_norm_eps. */
int modcr_linear_residual_ln_fwd(const void* A, int64_t lda, const void* W, const float* bias,
                                 const void* residual, const float* gamma, const float* beta,
                                 float eps, void* out, int32_t M, int32_t N, int32_t K,
                                 void* workspace, int64_t workspace_bytes, int32_t dtype,
                                 modcr_stream_t stream);
/* out = LayerNorm(dropout(A.W^T + bias) + residual): BertSelfOutput / BertOutput with the training-mode hidden dropout that
 * stays live inside the frozen encoders (a_bert:369-373, :446-451; p = 0: eval arithmetic) as ONE call: the GEMM's own rows go
 * through `workspace` (IEEE half on the bf16 path: 2 bytes, 11 significant bits; fp32 on the parity path), the row pass applies
 * the mask of modcr_dropout_residual_ln_fwd (counter = offset + row * N + column), adds the residual and normalises.
 * `workspace`: modcr_linear_dropout_residual_ln_workspace bytes; A dense rows (lda >= K), residual / out [M,N].
 * `pre_out` (may be NULL): [M,N] in `pre_dtype` (MODCR_F32, or MODCR_F16 = IEEE half, saturating: half the bytes of the rows the
 * backward reads back, 11 significant bits against the 8 of the bf16 activations saved beside them), receives the pre-LayerNorm rows
 * dropout(A.W^T + bias) + residual -- what the backward of a TRAINABLE layer needs (modcr_linear_residual_ln_dropout_bwd; pass
 * the same `pre_dtype` there), written by the same row pass. */
int64_t modcr_linear_dropout_residual_ln_workspace(int32_t M, int32_t N, int32_t K, int32_t dtype);
int modcr_linear_dropout_residual_ln_fwd(const void* A, int64_t lda, const void* W, const float* bias,
                                         const void* residual, const float* gamma, const float* beta, float eps,
                                         void* out, void* pre_out, int32_t pre_dtype, int32_t M, int32_t N, int32_t K, float p,
                                         uint64_t seed, uint64_t offset, void* workspace, int64_t workspace_bytes,
                                         int32_t dtype, modcr_stream_t stream);
int modcr_proj_residual_ln_fwd(const void* ctx, const void* wo, const float* bo, const void* x,
                               const float* gamma, const float* beta, float eps, void* out,
                               int32_t M, int32_t H, void* workspace, int64_t workspace_bytes,
                               int32_t dtype, modcr_stream_t stream);
int modcr_ffn_down_residual_ln_fwd(const void* inter, const void* w2, const float* b2, const void* a,
                                   const float* gamma, const float* beta, float eps, void* out,
                                   int32_t M, int32_t H, int32_t I, void* workspace,
                                   int64_t workspace_bytes, int32_t dtype, modcr_stream_t stream);

/* Row LayerNorm: y = LN(x (+ residual)).  x is in_dtype, y is out_dtype (MODCR_BF16 / MODCR_F32).
 * Output row m is written at out + ((m / rows_per_group) * group_stride + m % rows_per_group)
 * * H elements (rows_per_group = 0 -> dense), which lets region rows land behind the text rows of
 * each sequence (torch.cat at modeling_transfomres.py:684). */
int modcr_layernorm_fwd(const void* x, int32_t in_dtype, const void* residual, int32_t res_dtype,
                        const float* gamma, const float* beta, float eps, void* y, int32_t out_dtype,
                        int64_t M, int32_t H, int32_t rows_per_group, int64_t group_stride,
                        modcr_stream_t stream);

/* BertEmbeddings.forward (a_bert:184-211): out[n, t] = LN(word[ids] + pos[position] + type[tt]).
 * Tables are fp32.  position_ids NULL = arange(T).  Output rows go to out + (n*seq_stride + t)*H. */
int modcr_embed_ln_fwd(const int64_t* input_ids, const int64_t* token_type_ids,
                       const int64_t* position_ids, const float* word, const float* pos,
                       const float* type, const float* gamma, const float* beta, float eps,
                       void* out, int32_t N, int32_t T, int32_t H, int64_t seq_stride,
                       int32_t vocab, int32_t max_pos, int32_t type_vocab, int32_t out_dtype,
                       modcr_stream_t stream);

/* The same followed by BertEmbeddings.dropout (a_bert:209-210; p = 0: exactly the call above).  Decision of element (n, t, c): counter
 * offset + (n*seq_stride + t)*H + c -- the flat index inside the caller's [N, seq_stride, H] buffer, i.e. the mask modcr_dropout over
 * that whole buffer with the same (seed, offset) applies to these rows. */
int modcr_embed_ln_dropout_fwd(const int64_t* input_ids, const int64_t* token_type_ids,
                               const int64_t* position_ids, const float* word, const float* pos,
                               const float* type, const float* gamma, const float* beta, float eps,
                               void* out, int32_t N, int32_t T, int32_t H, int64_t seq_stride,
                               int32_t vocab, int32_t max_pos, int32_t type_vocab, int32_t out_dtype,
                               float p, uint64_t seed, uint64_t offset, modcr_stream_t stream);
/* Rows src [M,H] -> dst rows row0 + (m % rows_per_group) of sequence m / rows_per_group of a [*, group_stride, H] buffer, under
 * nn.Dropout (p = 0: a strided copy): the LayerNorm-ed region rows behind the text rows of each sequence with the img dropout in one
 * pass (modeling_transfomres.py:676-684, modeling_vcr_chunkalign_v10.py:338-345).  Counter of (dst row, c) = offset + dst_row*H + c,
 * the destination buffer's flat index.  dtype = MODCR_BF16 | MODCR_F32 (src and dst alike), H % 4 == 0. */
int modcr_rows_scatter_dropout(const void* src, void* dst, int64_t M, int32_t H, int32_t rows_per_group, int64_t group_stride,
                               int32_t row0, int32_t dtype, float p, uint64_t seed, uint64_t offset, modcr_stream_t stream);

/* fp32 [M,K] -> dtype [M,Kp] (Kp >= K, zero padded).  Region features arrive as fp32 with
 * K = 2054 (Data/VCRChunkAlign.py:713); the bf16 GEMM wants 16-byte aligned rows. */
int modcr_cast_pad(const float* src, int64_t lds_, void* dst, int64_t ldd, int64_t M, int32_t K,
                   int32_t Kp, int32_t dtype, modcr_stream_t stream);
/* fp32 [M,K] -> bf16 [M,3K] split for fp32-accurate products on the bf16 MFMA path (the trainable
 * CLS-path linears: cls_ensemble_1 v10:912, q/out_proj v10:710,796, ClsLayer_lyx FFN v10:868-869):
 * x = hi + lo; mode 0 (activations) writes [hi|lo|hi], mode 1 (weights) [hi|hi|lo], so a GEMM over
 * the tripled K sums a_hi*w_hi + a_lo*w_hi + a_hi*w_lo. */
int modcr_split3_bf16(const float* src, int64_t lds_, void* dst, int64_t ldd, int64_t M, int32_t K,
                      int32_t mode, modcr_stream_t stream);
/* generic dtype conversion of a contiguous buffer (weight packing) */
int modcr_convert(const void* src, int32_t src_dtype, void* dst, int32_t dst_dtype, int64_t n,
                  modcr_stream_t stream);
/* `count` element-wise conversions (fp32 <-> bf16, or copies) in one launch per eight segments: src[k] -> dst[k], n[k] elements; the
 * pointer tables are HOST arrays.  The trainable layers re-cast their weights with it after every optimizer step (q | k | v land side
 * by side in one [3H,H] matrix: consecutive destinations). */
int modcr_convert_segments(const void* const* src, void* const* dst, const int64_t* n, int32_t count, int32_t src_dtype,
                           int32_t dst_dtype, modcr_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Multi-view alignment attention core of cross_attention_lyx (v10:741-795) for one query token:
 *   q [N,E] fp32 (projected; `scale` = head_dim^-0.5 of v10:710 is applied inside),
 *   k,v [N,L,E] in `dtype` (projected encoder states) -> out [N,E] fp32 (heads merged, before
 *   out_proj), probs [N,heads,L] fp32 or NULL (the UNMASKED softmax).  No mask (v10:857 passes none).
 *   (p, seed, offset): F.dropout on the attention weights in training mode (v10:780, dropout=0.1 at v10:846);
 *   counter = offset + (n * heads + head) * L + key; p = 0 in eval mode.
 *   key_bias [N,L] fp32 or NULL: additive key mask, the same for every head (ClsLayer2's word_mask, v10:816-829, which is
 *   this kernel with heads = 1, scale = 1 and k == v). */
int modcr_align_attn_fwd(const float* q, const void* k, const void* v, int64_t ldkv, float* out,
                         float* probs, int32_t N, int32_t L, int32_t E, int32_t heads, float scale,
                         float p, uint64_t seed, uint64_t offset, const float* key_bias, int32_t dtype,
                         modcr_stream_t stream);
/* backward of the same: dout [N,E] fp32 -> dq [N,E] fp32, dk, dv [N,L,E] in `dtype` */
int modcr_align_attn_bwd(const float* dout, const float* q, const void* k, const void* v, int64_t ldkv,
                         const float* probs, float* dq, void* dk, void* dv, int64_t lddkv, int32_t N,
                         int32_t L, int32_t E, int32_t heads, float scale, float p, uint64_t seed,
                         uint64_t offset, int32_t dtype, modcr_stream_t stream);

/* The same attention (v10:741-795 as ClsLayer_lyx calls it at :857: one [CLS] query per sequence, 8 heads, no mask) over FROZEN
 * bf16 states, reassociated so that keys and values are never projected:
 *   score[h][j] = q_h . (Wk_h x_j + bk_h) = (Wk_h^T q_h) . x_j + const_h         (const_h cancels in the softmax over j)
 *   out_h       = Wv_h (sum_j p'[h][j] x_j) + bv_h sum_j p'[h][j]                (p' = dropout(p), v10:780)
 * qt [N,heads,E] fp32 = Wk_h^T q_h with the reference's q scaling folded in (the caller's few-row GEMM); the key rows are up to
 * three row blocks addressed in place (block i: base x_blocks[i], rows[i] rows of E bf16 at row stride ldx, sequence stride
 * seq_strides[i] elements -- v10:913's [global | chunk-align | chunk-hidden] text rows without the concatenated copy; the three
 * arrays are HOST arrays read at launch).  Out: ctx [N,heads,E] fp32 = sum_j p'[h][j] x_j, ssum [N,heads] = sum_j p'[h][j]
 * (1 when p = 0), probs [N,heads,L] fp32 (unmasked, for the backward), L = sum(rows).  Dropout counters as modcr_align_attn_fwd.
 * heads = 8, E <= 1024 and E % 4 == 0 only: MODCR_ERR_UNSUPPORTED otherwise (use the projected form). */
int modcr_cls_xattn_fwd(const float* qt, const void* const* x_blocks, const int64_t* seq_strides, const int32_t* rows,
                        int32_t nblocks, int64_t ldx, float* ctx, float* ssum, float* probs, int32_t N, int32_t E,
                        int32_t heads, float p, uint64_t seed, uint64_t offset, modcr_stream_t stream);

/* backward of the same with respect to qt (the states are frozen): dctx [N,heads,E], dssum [N,heads] -> dqt [N,heads,E];
 * ctx / ssum / probs as the forward wrote them, (p, seed, offset) as the forward took them. */
int modcr_cls_xattn_bwd(const float* dctx, const float* dssum, const float* ctx, const float* ssum, const float* probs,
                        const void* const* x_blocks, const int64_t* seq_strides, const int32_t* rows, int32_t nblocks,
                        int64_t ldx, float* dqt, int32_t N, int32_t E, int32_t heads, float p, uint64_t seed,
                        uint64_t offset, modcr_stream_t stream);

/* 4-way multiple-choice soft-label cross entropy, forward + backward in one launch
 * (modeling_ensemble.py:528-537): loss = mean_b(-sum_c label*log_softmax(logits)),
 * dlogits = grad_scale * (softmax*sum_c(label) - label)/B (one-hot labels: (softmax - label)/B).
 * logits, label, dlogits [B,C] fp32; loss: 1 fp32 (written, not accumulated); grad_scale: DEVICE
 * pointer to the upstream gradient of the loss (NULL = 1.0).  loss or dlogits may be NULL. */
int modcr_mc_ce_fwd_bwd(const float* logits, const float* label, float* loss, float* dlogits,
                        const float* grad_scale, int32_t B, int32_t C, modcr_stream_t stream);

/* ---- backward pieces for the trainable heads (cls_layer_lyx, mappers, scorer) ---------------
 * dX[M,K] = dY[M,N] . W[N,K]                      (modcr_linear_bwd_input)
 * dW[N,K] (+)= dY[M,N]^T . X[M,K], db[N] (+)= sum_m dY   (modcr_linear_bwd_weight; fp32 grads)
 * dY is dy_dtype, W / X are `dtype`; dW/db are fp32.  accumulate != 0 adds into dW/db.
 * With a workspace (bytes from the *_workspace queries) both run on the MFMA path: operands are
 * transposed to bf16 with the contraction dimension contiguous, dW uses split-K fp32 partials.
 * Without one (NULL) an exact-fp32 VALU kernel is used (parity path, tiny shapes).
 * Size the workspace with the query and nothing else: modcr_linear_bwd_weight_workspace(M, N, K) is the maximum over the forms the
 * call may take (plain, half-TN, the swapped dW^T form with its transposing reduction, the per-row-block bias partials), and the
 * fast forms are taken only when workspace_bytes >= that figure -- a buffer sized by an older rule (X^T + splits * N * K * 4)
 * is still CORRECT (the call falls back to the generic split-K route) but slower, with no error. */
int64_t modcr_linear_bwd_input_workspace(int32_t M, int32_t N, int32_t K);
int modcr_linear_bwd_input(const void* dY, int64_t lddy, int32_t dy_dtype, const void* W, int64_t ldw,
                           void* dX, int64_t lddx, int32_t M, int32_t N, int32_t K, int32_t dtype,
                           int32_t out_dtype, void* workspace, int64_t workspace_bytes,
                           modcr_stream_t stream);
int64_t modcr_linear_bwd_weight_workspace(int32_t M, int32_t N, int32_t K);
int modcr_linear_bwd_weight(const void* dY, int64_t lddy, int32_t dy_dtype, const void* X, int64_t ldx,
                            float* dW, float* db, int32_t M, int32_t N, int32_t K, int32_t accumulate,
                            int32_t dtype, void* workspace, int64_t workspace_bytes,
                            modcr_stream_t stream);
/* y = LN(x + residual): dX (= d residual) from dY; dgamma/dbeta fp32, ACCUMULATED (atomics; the
 * caller zeroes them).  residual may be NULL.  All fp32. */
int modcr_layernorm_bwd(const float* dY, const float* x, const float* residual, const float* gamma,
                        float eps, float* dX, float* dgamma, float* dbeta, int64_t M, int32_t H,
                        modcr_stream_t stream);
/* dpre = dact * act'(pre)  for GELU-erf / tanh; all fp32 [n] */
int modcr_act_bwd(const float* dact, const float* pre, float* dpre, int64_t n, int32_t act,
                  modcr_stream_t stream);

/* ---- backward of modcr_qkv_attn_fwd without prefix rows (autograd of CaptionBertSelfAttention, modeling_bert.py:34-75 /
 * v10:55-107, for the trainable-encoder variants: SURVEY 8f-1 / 8f-4 and BASELINE config 3).  From dctx [N,S,H]:
 * dx [N,S,H] (storage dtype), dwqkv [3H,H] and dbqkv [3H] (fp32; accumulate != 0 adds into them).  Nothing is saved
 * by the forward: q|k|v rows are recomputed into the workspace (fp32); the attention core runs on the matrix pipe
 * (attn_bwd_mfma_kernel: bf16 path, S <= 192; row statistics recomputed, masks as the forward) or as an exact-fp32
 * VALU kernel (fp32 parity path, S > 192); the chunk-mean of the queries is applied to dq as its own adjoint, then
 * dX = dqkv.Wqkv and dWqkv = dqkv^T.X on the GEMM path. */
int64_t modcr_qkv_attn_bwd_workspace(int32_t N, int32_t S, int32_t H, int32_t dtype);
int modcr_qkv_attn_bwd(const void* dctx, const void* x, const void* wqkv, const float* bqkv,
                       const float* key_mask, const uint32_t* dense_mask_bits, const int32_t* chunk_id,
                       int32_t chunk_t, void* dx, float* dwqkv, float* dbqkv, int32_t accumulate, int32_t N,
                       int32_t S, int32_t H, int32_t A, void* workspace, int64_t workspace_bytes, int32_t dtype,
                       modcr_stream_t stream);
/* The same for a forward that ran modcr_qkv_attn_dropout_fwd with (attn_p, seed, offset): the mask is regenerated, dV takes
 * the masked probabilities, the softmax backward the masked dP (bf16 path, 64 < S <= 256: the forward's tile kernels; the MFMA
 * cores up to S = 192, the exact core on the 256-token tile's counter layout above that).
 * d_align [N, align_t, S - align_t] fp32 or NULL: gradient of the align map the forward accumulated (v10:1067-1073, the
 * align loss of ChunkAlign_CLS_enc4_align): added to dP of every head on the text-query x region-key block.
 * dx_residual [N,S,H] fp32 or NULL: added to dx in the epilogue of its GEMM (the residual-stream gradient of the layer: the
 * sum the layer backward otherwise forms in a pass of its own). */
int modcr_qkv_attn_dropout_bwd(const void* dctx, const void* x, const void* wqkv, const float* bqkv,
                               const float* key_mask, const uint32_t* dense_mask_bits, const int32_t* chunk_id,
                               int32_t chunk_t, const float* dx_residual, void* dx, float* dwqkv, float* dbqkv, int32_t accumulate, int32_t N,
                               int32_t S, int32_t H, int32_t A, float attn_p, uint64_t seed, uint64_t offset,
                               const float* d_align, int32_t align_t,
                               void* workspace, int64_t workspace_bytes, int32_t dtype, modcr_stream_t stream);
/* The same with what modcr_qkv_attn_lse_fwd left behind: ctx [N,S,H] (its context rows, which the layer keeps for the output
 * projection's weight gradient anyway) and lse [N,A,S] (both, or both NULL = modcr_qkv_attn_dropout_bwd).  bf16 path,
 * S <= 192, no d_align: the attention core is attn_bwd5_kernel (csrc/attn_bwd.hip) -- P = exp2(score - lse),
 * delta = rowsum(dO o ctx), five MFMA products per (query block, key block) with dK / dV resident in the accumulators of
 * the wave that owns the keys and only dS crossing LDS (autograd of modeling_bert.py:46-72).  Any other call takes the
 * older cores.  qkv_dump (or NULL; with ctx + lse, 64 < S <= 192): the Q | K | V images modcr_qkv_attn_lse_fwd dumped -- the
 * projections are then not recomputed and the core is attn_bwd6_kernel (one 16-key tile per compute wave, loader waves). */
int modcr_qkv_attn_lse_bwd(const void* dctx, const void* x, const void* wqkv, const float* bqkv,
                           const float* key_mask, const uint32_t* dense_mask_bits, const int32_t* chunk_id,
                           int32_t chunk_t, const float* dx_residual, void* dx, float* dwqkv, float* dbqkv, int32_t accumulate, int32_t N,
                           int32_t S, int32_t H, int32_t A, float attn_p, uint64_t seed, uint64_t offset,
                           const float* d_align, int32_t align_t, const void* ctx, const float* lse, const void* qkv_dump,
                           void* workspace, int64_t workspace_bytes, int32_t dtype, modcr_stream_t stream);
/* modcr_qkv_attn_lse_bwd + flags: with MODCR_ATTN_SIDE_POST_DROPOUT d_align is the gradient of an align map written by
 * modcr_qkv_attn_opt_fwd under the same flag, and enters dP under the forward's mask: m / (1 - p) o (dO V^T + d_align). */
int modcr_qkv_attn_opt_bwd(const void* dctx, const void* x, const void* wqkv, const float* bqkv,
                           const float* key_mask, const uint32_t* dense_mask_bits, const int32_t* chunk_id,
                           int32_t chunk_t, const float* dx_residual, void* dx, float* dwqkv, float* dbqkv, int32_t accumulate, int32_t N,
                           int32_t S, int32_t H, int32_t A, float attn_p, uint64_t seed, uint64_t offset,
                           const float* d_align, int32_t align_t, const void* ctx, const float* lse, const void* qkv_dump, int32_t flags,
                           void* workspace, int64_t workspace_bytes, int32_t dtype, modcr_stream_t stream);

/* ---- backward of the encoder layer's GEMM blocks (autograd of BertSelfOutput / BertIntermediate / BertOutput,
 * a_bert:362-373, :425-437, :440-451).  Gradients of parameters are fp32; dgamma / dbeta are ACCUMULATED (caller
 * zeroes), everything else is written.  bf16 path: products on the MFMA route (workspace from the query);
 * fp32 path: exact VALU kernels (workspace may be NULL).
 *   linear_residual_ln_bwd: out = LN(A.W^T + bias + residual).  `pre` = the fp32 pre-LN rows [M,N] the forward left
 *     in its workspace.  d_pre [M,N] fp32 = gradient of the GEMM output = gradient of the residual input;
 *     dA [M,K] in `dtype`.  proj_ / ffn_down_ are the BertSelfOutput / BertOutput shapes of it.
 *   ffn_up_gelu_bwd: inter = gelu(x.W1^T + b1); dinter [M,I] (fp32 or bf16) -> dx [M,H] fp32 (+ dx_residual [M,H] fp32 when
 *     not NULL: the gradient arriving at x through the residual branch, added in the GEMM's epilogue), dW1, db1; the GELU
 *     input is recomputed into the workspace.
 *   chunk_mean_q_bwd: adjoint of modcr_chunk_mean_q_fwd (the same segment mean, applied to the gradient rows). */
int64_t modcr_linear_residual_ln_bwd_workspace(int32_t M, int32_t N, int32_t K);
int modcr_linear_residual_ln_bwd(const float* dY, const float* pre, const void* A, int64_t lda, const void* W,
                                 const float* gamma, float eps, float* d_pre, void* dA, float* dW, float* dbias,
                                 float* dgamma, float* dbeta, int32_t M, int32_t N, int32_t K, void* workspace,
                                 int64_t workspace_bytes, int32_t dtype, modcr_stream_t stream);
/* the same with the forward's hidden dropout (out = LN(dropout(A.W^T + bias) + residual), modcr_dropout_residual_ln_fwd):
 * (p, seed, offset) as the forward consumed them; dY fp32 or bf16.  On the bf16 route one LayerNorm-backward pass
 * (modcr_layernorm_dropout_bwd) writes d_pre and the masked bf16 operand of the two GEMMs. */
int modcr_linear_residual_ln_dropout_bwd(const void* dY, int32_t dy_dtype, const void* pre, int32_t pre_dtype, const void* A, int64_t lda,
                                         const void* W, const float* gamma, float eps, float* d_pre, void* dA, float* dW,
                                         float* dbias, float* dgamma, float* dbeta, int32_t M, int32_t N, int32_t K,
                                         float p, uint64_t seed, uint64_t offset, void* workspace, int64_t workspace_bytes,
                                         int32_t dtype, modcr_stream_t stream);
int modcr_layernorm_dropout_bwd(const void* dY, int32_t dy_dtype, const void* pre, int32_t pre_dtype, const float* gamma, float eps,
                                float* d_pre, void* d_sub_bf16, float* dgamma, float* dbeta, int64_t M, int32_t H,
                                float p, uint64_t seed, uint64_t offset, modcr_stream_t stream);
int modcr_proj_residual_ln_bwd(const float* dY, const float* pre, const void* ctx, const void* wo, const float* gamma,
                               float eps, float* d_pre, void* dctx, float* dwo, float* dbo, float* dgamma,
                               float* dbeta, int32_t M, int32_t H, void* workspace, int64_t workspace_bytes,
                               int32_t dtype, modcr_stream_t stream);
int modcr_ffn_down_residual_ln_bwd(const float* dY, const float* pre, const void* inter, const void* w2,
                                   const float* gamma, float eps, float* d_pre, void* dinter, float* dw2, float* db2,
                                   float* dgamma, float* dbeta, int32_t M, int32_t H, int32_t I, void* workspace,
                                   int64_t workspace_bytes, int32_t dtype, modcr_stream_t stream);
int64_t modcr_ffn_up_gelu_bwd_workspace(int32_t M, int32_t H, int32_t I);
int modcr_ffn_up_gelu_bwd(const void* dinter, int32_t dinter_dtype, const void* x, const void* w1, const float* b1,
                          const float* dx_residual, float* dx, float* dw1, float* db1, int32_t M, int32_t H, int32_t I, void* workspace,
                          int64_t workspace_bytes, int32_t dtype, modcr_stream_t stream);
/* The trainable FFN with the GELU input kept (reference: BertIntermediate.forward / BertOutput.forward under autograd,
 * modeling_transfomres.py:421-452 -- autograd saves the GELU input too).  bf16 route, M >= 256, M % 8 == 0, H in
 * {256, 512, 768, 1024}, I % 256 == 0 (modcr_ffn_keep_supported returns 1); other shapes use modcr_ffn_up_gelu_fwd /
 * modcr_ffn_up_gelu_bwd, which recompute the GELU input.
 *   ffn_up_gelu_keep_fwd:            out = gelu(x.W1^T + b1) and pre_act = x.W1^T + b1, both bf16 [M,I], one GEMM
 *   ffn_down_residual_ln_gelu_bwd:   modcr_linear_residual_ln_dropout_bwd of BertOutput whose dX product leaves
 *                                    d_u = (d_sub.W2) * gelu'(pre_act) (bf16 [M,I]) instead of d_inter
 *                                    db_u (may be NULL): fp32 [I] = colsum(d_u), the bias gradient of BertIntermediate, summed in
 *                                    the same epilogue (before the bf16 rounding of d_u)
 *   ffn_up_du_bwd:                   dW1 = d_u^T.x, db1 = colsum(d_u) (db1 may be NULL when db_u above was taken: the product is
 *                                    then formed transposed with d_u token-major, no transpose of d_u), dx = d_u.W1 (+ dx_residual: the
 *                                    sum is formed in fp32 in the epilogue) in the storage dtype `dtype` [M,H] -- bf16 on the bf16 route,
 *                                    like every gradient that crosses a layer boundary; it is the dY of BertSelfOutput's LayerNorm backward */
int modcr_ffn_keep_supported(int32_t M, int32_t H, int32_t I, int32_t dtype);
int modcr_ffn_up_gelu_keep_fwd(const void* x, const void* w1, const float* b1, void* out, void* pre_act, int32_t M,
                               int32_t H, int32_t I, int32_t dtype, modcr_stream_t stream);
int64_t modcr_ffn_down_gelu_bwd_workspace(int32_t M, int32_t H, int32_t I);
int modcr_ffn_down_residual_ln_gelu_bwd(const void* dY, int32_t dy_dtype, const void* pre, int32_t pre_dtype, const void* inter, const void* w2,
                                        const float* gamma, float eps, const void* pre_act, float* d_pre, void* d_u, float* db_u,
                                        float* dw2, float* db2, float* dgamma, float* dbeta, int32_t M, int32_t H, int32_t I,
                                        float p, uint64_t seed, uint64_t offset, void* workspace, int64_t workspace_bytes,
                                        int32_t dtype, modcr_stream_t stream);
int64_t modcr_ffn_up_du_bwd_workspace(int32_t M, int32_t H, int32_t I);
int modcr_ffn_up_du_bwd(const void* du, const void* x, const void* w1, const float* dx_residual, void* dx, float* dw1,
                        float* db1, int32_t M, int32_t H, int32_t I, void* workspace, int64_t workspace_bytes,
                        int32_t dtype, modcr_stream_t stream);
int modcr_chunk_mean_q_bwd(void* dq, int64_t row_stride, int64_t seq_stride, const int32_t* chunk_id, int32_t N,
                           int32_t T, int32_t H, int32_t dtype, modcr_stream_t stream);

/* ---- dropout, training mode (nn.Dropout of BertEmbeddings / BertSelfOutput / BertOutput / the mapping networks /
 * ClsLayer_lyx stays active inside the frozen encoders under model.train(): run_PMR_ModCR.py:171).
 * Counter-based: element i keeps its value (scaled by 1/(1-p)) iff the 15-bit uniform of counter offset + i is >= round(p 2^15)
 * (four uniforms per hash of the counter's group (offset + i) / 4: pass offsets that are multiples of 4 and the row kernels
 * hash once per 16-byte piece), so a backward pass regenerates the mask from (seed, offset) -- call modcr_dropout on the
 * gradient with the same pair.
 *   modcr_dropout: out = dropout(x) over n contiguous elements of `dtype` (in place allowed).
 *   modcr_dropout_residual_ln_fwd: out = LN(dropout(x) + residual); x [M,H] = the GEMM's output without residual, fp32 or
 *     (bf16 path) MODCR_F16; element index = row * H + column; pre_out (may be NULL): [M,H] copy of dropout(x) + residual in
 *     pre_dtype (MODCR_F32 or MODCR_F16). */
int modcr_dropout(const void* x, void* out, int64_t n, int32_t dtype, float p, uint64_t seed, uint64_t offset,
                  modcr_stream_t stream);
int modcr_dropout_residual_ln_fwd(const void* x, int32_t x_dtype, const void* residual, int32_t res_dtype, const float* gamma,
                                  const float* beta, float eps, void* out, int32_t out_dtype, void* pre_out, int32_t pre_dtype, int64_t M,
                                  int32_t H, float p, uint64_t seed, uint64_t offset, modcr_stream_t stream);

/* out[n] = a[n] + b[n]: a fp32, b / out fp32 or bf16 (the residual-gradient sums of the layer backward) */
int modcr_add(const float* a, const void* b, int32_t b_dtype, void* out, int32_t out_dtype, int64_t n,
              modcr_stream_t stream);

/* Embedding-table backward (autograd of BertEmbeddings' lookups, a_transformers.../modeling_bert.py:184-211, and of
 * RobertaEmbeddings' in the prefix body): dw[id, :] += sum over the rows r of dy [M,H] fp32 with ids[r] == id, for every id but
 * padding_idx (-1 = none).  sorted_ids = the flat ids sorted ascending (stable), order = the permutation that sorts them; one
 * workgroup owns one table row and adds in sorted order: deterministic, no atomics.  dw fp32 [V,H] is ADDED into; ids outside
 * [0, V) are skipped (they cannot come from a forward that ran; a caller's bug must not write outside dw).
 * modcr_embedding_bwd is the same without the table height (ids are not bounds-checked): the entry's original signature, kept so
 * that a caller built against an older header does not silently pass V where padding_idx is read. */
int modcr_embedding_bwd_v(const int64_t* sorted_ids, const int64_t* order, const float* dy, float* dw, int32_t M, int32_t H,
                          int64_t V, int64_t padding_idx, modcr_stream_t stream);
int modcr_embedding_bwd(const int64_t* sorted_ids, const int64_t* order, const float* dy, float* dw, int32_t M, int32_t H,
                        int64_t padding_idx, modcr_stream_t stream);

/* ---- optimizer step over the flat gradient buffer (run_PMR_ModCR.py:216,224-227: clip_grad_norm_(all, max_norm),
 * AdamW step).  Everything stays on the device: modcr_sumsq_f32 ADDS sum(x^2) to *out (one fp32 the
 * caller zeroes first; several calls accumulate the global norm over several buffers); the step kernels read that
 * scalar, form clip = min(1, max_norm / (sqrt(sumsq) + 1e-6)) as torch.nn.utils.clip_grad_norm_ does (max_norm <= 0
 * or sumsq == NULL: no clipping) and update p, m, v [n] in place, g' = clip * g, m = b1 m + (1 - b1) g',
 * v = b2 v + (1 - b2) g'^2, bc1 = 1 - b1^t, bc2 = 1 - b2^t given by the caller.  The gradients are left unscaled.
 *   modcr_adamw_hf_step -- transformers.AdamW, the optimizer the reference trains with (run_PMR_ModCR.py:24,137;
 *     transformers 4.x optimization.py::AdamW.step with correct_bias=True):
 *       p -= lr * sqrt(bc2) / bc1 * m / (sqrt(v) + eps);   then p -= lr * weight_decay * p.
 *   modcr_adamw_step    -- torch.optim.AdamW (kept for A/B; eps enters after the bias correction of v):
 *       p *= 1 - lr * weight_decay;   p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps). */
int modcr_sumsq_f32(const float* x, int64_t n, float* out, modcr_stream_t stream);
/* The same sum in a FIXED order (what FlatAdamW uses): per-workgroup partials into the caller's `partials`
 * (max_partials floats, modcr_sumsq_partials() of them are used at most), folded in index order by a second launch and
 * ADDED to *out.  modcr_sumsq_f32 adds its workgroups' partials with a float atomic, i.e. in scheduling order: two
 * data-parallel replicas holding the same reduced gradient could get clip coefficients that differ in the last bit
 * and drift apart; with this entry the clip (run_PMR_ModCR.py:216) is a pure function of the gradient buffer. */
int modcr_sumsq_partials(void);
int modcr_sumsq_f32_ordered(const float* x, int64_t n, float* out, float* partials, int32_t max_partials, modcr_stream_t stream);
int modcr_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, const float* sumsq,
                     float max_norm, float lr, float beta1, float beta2, float eps, float weight_decay,
                     float bc1, float bc2, modcr_stream_t stream);
int modcr_adamw_hf_step(float* p, const float* g, float* m, float* v, int64_t n, const float* sumsq,
                        float max_norm, float lr, float beta1, float beta2, float eps, float weight_decay,
                        float bc1, float bc2, modcr_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MODCR_HIP_H */
