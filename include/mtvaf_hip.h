/* mtvaf_hip.h -- C ABI of libmtvaf_hip.so: the MI355X (gfx950) kernels behind MTVAF's prefix-fused
 * BERT/RoBERTa forward/backward path.
 *
 * The reference (MKMaS-GUET/MTVAF) has no FFI layer: the path sits behind Python nn.Modules
 * (models/bert_model.py::TVNetSAModel2, models/modeling_bert.py::BertModel).  Each entry point below
 * replaces the eager torch ops of the cited reference lines; the Python side (mtvaf_amd/hip.py) binds
 * them with ctypes and wraps them in torch.autograd.Functions inside modules that keep the reference's
 * class names, constructor/forward signatures and parameter names.
 *
 * Conventions
 *   - all tensors are dense row-major fp32 device buffers owned by the caller (torch); the library never
 *     allocates; scratch is passed in (`*_workspace_bytes` queries give the size);
 *   - every call only enqueues work on `stream` (a hipStream_t) and never synchronises;
 *   - return 0 on success, <0 = MTVAF_ERR_* (bad shape / alignment / argument / workspace), >0 = hipError_t;
 *   - float4 vector access: row strides are multiples of 4 floats and bases 16-byte aligned unless a
 *     function says otherwise (the GEMM falls back to scalar loads when they are not);
 *   - dropout masks are a pure function of (seed, offset, element index): backward calls take the same
 *     (p_drop, seed, offset) as their forward and regenerate the mask;
 *   - `accumulate` != 0 adds parameter gradients into the destination instead of overwriting it.
 */
#ifndef MTVAF_HIP_H
#define MTVAF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* mtvaf_stream_t; /* hipStream_t */

#define MTVAF_OK 0
#define MTVAF_ERR_SHAPE (-1)
#define MTVAF_ERR_ALIGN (-2)
#define MTVAF_ERR_ARG (-3)
#define MTVAF_ERR_WORKSPACE (-4)

/* operand layouts of mtvaf_gemm_f32: KC = reduction index contiguous, KM = reduction index is the row */
#define MTVAF_KC 0
#define MTVAF_KM 1
/* epilogues */
#define MTVAF_EPI_NONE 0
#define MTVAF_EPI_GELU 1  /* C = gelu_erf(acc + bias), pre-activation -> aux  (modeling_bert.py:420-421) */
#define MTVAF_EPI_TANH 2  /* C = tanh(acc + bias)                            (bert_model.py:446-454)   */
#define MTVAF_EPI_DGELU 3 /* C = acc * gelu_erf'(aux)                                                   */
#define MTVAF_EPI_DTANH 4 /* C = acc * (1 - aux^2)                                                      */

/* Device-side dropout epoch for captured launches (HIP graph replay freezes kernel arguments, so host-fed (seed, offset)
 * pairs would repeat the same masks): while a device uint64 word is registered, every dropout-drawing kernel of the
 * library folds its value into its counter stream; mtvaf_rng_epoch_advance (*word += 1, captured as the first node of a
 * training-step graph) makes every replay draw fresh masks, identically in its forward and backward kernels.  NULL
 * restores host-fed masks.  Process-global state (like the launch profiler). */
int mtvaf_rng_set_epoch_ptr(const uint64_t* dev_word);
int mtvaf_rng_epoch_advance(uint64_t* dev_word, mtvaf_stream_t stream);
int mtvaf_version(void);
int mtvaf_device_cus(void);

/* ---- dense projections (fp32 MFMA): nn.Linear forward / dX / dW ------------------------------------
 * replaces query/key/value (modeling_bert.py:266,283-284), BertSelfOutput.dense (:353),
 * BertIntermediate.dense + GELU (:420-421), BertOutput.dense (:433), encoder_conv / projectors /
 * img_classifier (bert_model.py:446-459, 541-542, 552, 568), fc (bert_model.py:510) and their autograd
 * backward.  C[M,N] = opA[M,K] . opB[K,N] (+bias[N]) with epilogue `epi`; cfg/splits < 0 = heuristic.
 * allow_split enables a deterministic split-K (ordered slab reduction) through `workspace`. */
size_t mtvaf_gemm_f32_workspace_bytes(int M, int N, int K, int allow_split);
/* launch profiler (bench.py roofline): HIP events around the main GEMM kernel of every call, on its stream */
int mtvaf_prof_start(int capacity);
int mtvaf_prof_stop(int* n_out, int* keys /* [n][8] */, float* ms /* [n] */, int max_records);
int mtvaf_gemm_f32_plan(int layout_a, int layout_b, int M, int N, int K, int epi, int allow_split, int* cfg,
                        int* splits);
int mtvaf_gemm_f32(int layout_a, int layout_b, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                   int M, int N, int K, const float* bias, int epi, float* aux, int ldaux, int accumulate,
                   int allow_split, void* workspace, size_t workspace_bytes, int cfg, int splits,
                   mtvaf_stream_t stream);

/* bf16-compute variant for the mixed-precision configurations: identical contract (fp32 buffers), operands are
 * rounded to bf16 while staged and multiplied on the bf16 MFMA with fp32 accumulation. */
int mtvaf_gemm_bf16(int layout_a, int layout_b, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                    int M, int N, int K, const float* bias, int epi, float* aux, int ldaux, int accumulate,
                    int allow_split, void* workspace, size_t workspace_bytes, int cfg, int splits,
                    mtvaf_stream_t stream);

/* ---- fused prefix self-attention ---------------------------------------------------------------------
 * replaces BertSelfAttention.forward's prefix concat + scores + mask + softmax + dropout + PV + head merge
 * (modeling_bert.py:282-286, 303, 320-337; modeling_roberta.py:218-222) and its backward.
 * qkv [B*S,3H] token-major (Q|K|V); pk/pv [B,P*H] raw reshape(bsz,12,-1,64) slabs (bert_model.py:585);
 * addmask [B,P+S] = (1-mask)*-10000; ctx [B*S,H]; lse [B,NH,S].  head_dim must be 64. */
int mtvaf_prefix_attn_fwd(const float* qkv, const float* pk, const float* pv, const float* addmask, float* ctx,
                          float* lse, int B, int S, int P, int NH, int head_dim, float p_drop, uint64_t seed,
                          uint64_t offset, mtvaf_stream_t stream);
int mtvaf_prefix_attn_bwd(const float* dctx, const float* qkv, const float* pk, const float* pv,
                          const float* addmask, const float* ctx, const float* lse, float* delta, float* dqkv,
                          float* dpk, float* dpv, int B, int S, int P, int NH, int head_dim, float p_drop,
                          uint64_t seed, uint64_t offset, mtvaf_stream_t stream);
/* mtvaf_prefix_attn_bwd for callers that vouch (zero_tail != 0) that dctx is EXACTLY zero for the queries behind a sentence's
 * last unmasked text position -- trailing padding nothing downstream reads: the contract under which the weight gradients
 * walk k-tile lists.  Those queries have dQ = 0 and add nothing to dK / dV, so the query loops stop there; under the
 * contract the results equal mtvaf_prefix_attn_bwd's bit for bit. */
int mtvaf_prefix_attn_bwd_tail(const float* dctx, const float* qkv, const float* pk, const float* pv, const float* addmask,
                               const float* ctx, const float* lse, float* delta, float* dqkv, float* dpk, float* dpv, int B,
                               int S, int P, int NH, int head_dim, float p_drop, uint64_t seed, uint64_t offset, int zero_tail,
                               mtvaf_stream_t stream);

/* The same attention over PACKED token rows (padding-free execution): cu [B+1] int32 -- sentence b owns rows
 * cu[b] .. cu[b+1]-1 of qkv / ctx / dctx / dqkv (its unmasked tokens in order, at most S).  Every kept key is unmasked,
 * so no additive mask is read (a masked key contributes exp(-10000) = 0 in the padded form: dropping it is exact);
 * lse / delta stay [B,NH,S].  pad_rows: the rows behind the last sentence that pad the packed image to whole tiles; an
 * extra slice of the launch zero-fills them in ctx / dqkv (an unwritten row could hold a NaN: 0 x NaN would poison dW). */
int mtvaf_prefix_attn_varlen_fwd(const float* qkv, const float* pk, const float* pv, const int* cu, int pad_rows, float* ctx,
                                 float* lse, int B, int S, int P, int NH, int head_dim, float p_drop, uint64_t seed, uint64_t offset,
                                 mtvaf_stream_t stream);
int mtvaf_prefix_attn_varlen_bwd(const float* dctx, const float* qkv, const float* pk, const float* pv, const int* cu, int pad_rows,
                                 const float* ctx, const float* lse, float* delta, float* dqkv, float* dpk, float* dpv, int B, int S,
                                 int P, int NH, int head_dim, float p_drop, uint64_t seed, uint64_t offset, mtvaf_stream_t stream);

/* Round 5, pre-split operands: the packed-row attention that ALSO writes the tile-blocked plane image of its GEMM-operand result --
 * the context as [H / 32][3][rows][32] (read by the Wo product, modeling_bert.py:353, and its weight gradient) / dQ | dK | dV as
 * [3H / 32][3][rows][32] (read by the QKV dX product and its weight gradient); rows = the packed image's row count.  Bit for bit what
 * mtvaf_f32_split_planes writes over the fp32 result, which is written as before. */
int mtvaf_prefix_attn_varlen_fwd_planes(const float* qkv, const float* pk, const float* pv, const int* cu, int pad_rows, float* ctx,
                                        float* lse, int B, int S, int P, int NH, int head_dim, float p_drop, uint64_t seed,
                                        uint64_t offset, void* ctx_planes, int rows, mtvaf_stream_t st);
int mtvaf_prefix_attn_varlen_bwd_planes(const float* dctx, const float* qkv, const float* pk, const float* pv, const int* cu, int pad_rows,
                                        const float* ctx, const float* lse, float* delta, float* dqkv, float* dpk, float* dpv, int B,
                                        int S, int P, int NH, int head_dim, float p_drop, uint64_t seed, uint64_t offset,
                                        void* dqkv_planes, int rows, mtvaf_stream_t st);
/* dst[r][:] = map[r] >= 0 ? src[map[r]][:] : 0  (rows_dst rows of H floats): packs / unpacks token rows. */
int mtvaf_gather_rows(const float* src, const int* map, float* dst, int rows_dst, int H, mtvaf_stream_t stream);
/* Token packing of a batch from its additive mask [B, T = P + S] (text keys at columns P..; kept: > -5000): cu [B+1] row
 * offsets of the sentences, inv [B*S] flat token -> packed row (-1 masked), rowmap [B*S] packed row -> flat token (-1 beyond
 * the kept rows), mv_out[0] = number of kept rows.  All int32, device. */
int mtvaf_build_packing(const float* addmask, int B, int T, int P, int S, int* cu, int* inv, int* rowmap, int* mv_out,
                        mtvaf_stream_t stream);
/* Round 6: the same with cu [2 B + 1]: behind the offsets, cu[B + 1 + z] = the sentence that the varlen attention launches
 * (mtvaf_prefix_attn_varlen_* / _bf16_varlen_*) run in slot z of their grid -- longest first -- and cu[0] = -1 marks the list as
 * present (a cu with cu[0] = 0 means sentence z in slot z).  Placement only: a launch lasts as long as its busiest CU, and in sorted
 * order the blocks a CU receives come from the long, the middle and the short third of the batch; results are bit-identical. */
int mtvaf_build_packing_ordered(const float* addmask, int B, int T, int P, int S, int* cu, int* inv, int* rowmap, int* mv_out,
                                mtvaf_stream_t stream);
/* k-tile list for mtvaf_gemm_f32_ktiles: the bk-row tiles of the [B*S] token axis that hold at least one unmasked token
 * (additive mask [B, T = P + S], kept: > -5000), in order; kcnt[0] = how many.  (B*S) % bk == 0.  Device int32 arrays. */
int mtvaf_build_ktiles(const float* addmask, int B, int T, int P, int S, int bk, int* klist, int* kcnt,
                       mtvaf_stream_t stream);
/* mtvaf_gemm_f32 for a weight-gradient product (layouts KM x KM, the reduction index is the token row) whose operand A is
 * EXACTLY ZERO outside the listed 32-row k-tiles: the reduction runs over klist[0 .. *kcnt) only (no host sync).  Gradients
 * of token rows that nothing downstream reads -- padded positions -- are exact zeros, so skipping their k-tiles changes
 * nothing but time; plans that cannot use the list reduce over the whole range (same result). */
int mtvaf_gemm_f32_ktiles(int layout_a, int layout_b, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                          int M, int N, int K, const float* bias, int epi, float* aux, int ldaux, int accumulate,
                          int allow_split, void* workspace, size_t workspace_bytes, int cfg, int splits, const int* klist,
                          const int* kcnt, mtvaf_stream_t stream);
/* p[0..n) = 0 (a kernel, not hipMemsetAsync: usable inside captured graphs, see rowops.hip). */
int mtvaf_zero_f32(float* p, long n, mtvaf_stream_t stream);

/* ---- embeddings + LayerNorm + dropout -----------------------------------------------------------------
 * replaces BertEmbeddings.forward (modeling_bert.py:212-222) / RobertaEmbeddings.forward
 * (modeling_roberta.py:102-140) and create_position_ids_from_input_ids (:1706-1719).  pos_ids == NULL
 * means arange(S) (BERT, past_key_values_length hard-coded 0 at modeling_bert.py:1050). */
int mtvaf_roberta_position_ids(const int64_t* ids, int32_t* pos_ids, int B, int S, int pad_idx, mtvaf_stream_t stream);
int mtvaf_embed_ln_fwd(const int64_t* ids, const int64_t* type_ids, const int32_t* pos_ids, const float* word,
                       const float* pos, const float* type, const float* gamma, const float* beta, float* out,
                       float* mean, float* rstd, int B, int S, int H, float eps, float p_drop, uint64_t seed,
                       uint64_t offset, void* out_bf16 /* nullable: bf16 copy of out (mixed-precision GEMM operand) */,
                       mtvaf_stream_t stream);
size_t mtvaf_embed_ln_bwd_workspace_bytes(int M, int H, int vocab, int max_pos);
int mtvaf_embed_ln_bwd(const float* dout, const int64_t* ids, const int64_t* type_ids, const int32_t* pos_ids,
                       const float* word, const float* pos, const float* type, const float* gamma, const float* mean,
                       const float* rstd, float* dword, float* dpos, float* dtype, float* dgamma, float* dbeta,
                       int accumulate, int B, int S, int H, int vocab, int max_pos, int type_vocab, int word_pad,
                       int pos_pad, float p_drop, uint64_t seed, uint64_t offset, float* dz_ws, void* workspace,
                       size_t workspace_bytes, mtvaf_stream_t stream);
/* How mtvaf_embed_ln_bwd adds token-row gradients into the word (and RoBERTa position) table: 0 (default) one owner wave per
 * table row sums its tokens in row order -- bit-reproducible; 1 float atomics (MTVAF_EMBED_ATOMIC=1).  mode 0 / 1 sets, any
 * other value queries; returns the mode in force.  (nn.Embedding's backward, modeling_bert.py:170-172.) */
int mtvaf_embed_scatter_mode(int mode);

/* ---- dropout + residual + LayerNorm --------------------------------------------------------------------
 * replaces BertSelfOutput / BertOutput `LayerNorm(dropout(dense_out) + input)` (modeling_bert.py:354-355,
 * 434-435); the dense bias is added by the GEMM epilogue.  The backward also emits the column sums of dx
 * (dbias_x, nullable) = the gradient of that dense bias, saving a separate reduction pass. */
size_t mtvaf_ln_bwd_workspace_bytes(int M, int H);
int mtvaf_dropout_res_ln_fwd(const float* x, const float* res, const float* gamma, const float* beta, float* out,
                             float* mean, float* rstd, int M, int H, float eps, float p_drop, uint64_t seed,
                             uint64_t offset, void* out_bf16 /* nullable */, mtvaf_stream_t stream);
int mtvaf_dropout_res_ln_bwd(const float* dout, const float* x, const float* res, const float* gamma,
                             const float* mean, const float* rstd, float* dx, float* dres, int dres_accumulate,
                             float* dgamma, float* dbeta, float* dbias_x, int accumulate, int M, int H, float p_drop,
                             uint64_t seed, uint64_t offset, void* workspace, size_t workspace_bytes,
                             void* dx_bf16 /* nullable: dx rounded to bf16; dx may then be NULL */, mtvaf_stream_t stream);
/* The two halves of mtvaf_dropout_res_ln_bwd as separate calls: _rows writes dx / dres and per-block partial sums into `part`
 * (mtvaf_ln_bwd_workspace_bytes(M, H) bytes, the caller's until _finish has run); _finish reduces them, in fixed order, into
 * dgamma / dbeta / dbias_x.  The layer executor runs _finish on its weight-gradient stream: the sums feed parameter
 * gradients only (autograd backward of modeling_bert.py:354-355, 434-435). */
int mtvaf_dropout_res_ln_bwd_rows(const float* dout, const float* x, const float* res, const float* gamma, const float* mean,
                                  const float* rstd, float* dx, float* dres, int dres_accumulate, int M, int H, float p_drop,
                                  uint64_t seed, uint64_t offset, float* part, void* dx_bf16 /* nullable */,
                                  mtvaf_stream_t stream);
int mtvaf_dropout_res_ln_bwd_finish(const float* part, int M, int H, float* dgamma, float* dbeta, float* dbias_x /* nullable */,
                                    int accumulate, mtvaf_stream_t stream);

/* ---- small reductions / elementwise ---------------------------------------------------------------------
 * bias gradients (column sums of dY) and nn.Dropout on the sequence output (bert_model.py:506). */
size_t mtvaf_colsum_workspace_bytes(int rows, int cols);
int mtvaf_colsum(const float* x, int rows, int cols, int ld, float* out, int accumulate, void* workspace,
                 size_t workspace_bytes, mtvaf_stream_t stream);
int mtvaf_dropout(const float* x, float* y, long n, float p_drop, uint64_t seed, uint64_t offset,
                  mtvaf_stream_t stream);

/* ---- linear-chain CRF head --------------------------------------------------------------------------------
 * replaces torchcrf.CRF(num_tags, batch_first=True): forward(..., reduction='mean') negated at
 * bert_model.py:521 and decode at :511.  emissions [B,S,C], tags int64 [B,S], mask uint8 [B,S] with
 * mask[:,0] == 1, C <= 16.  loss[0] = -mean_b(score(gold) - logZ).  The workspace written by the forward
 * is read by the backward.  tags_out int32 [B,S] (-1 padded), lens_out int32 [B]. */
size_t mtvaf_crf_workspace_bytes(int B, int S, int C);
int mtvaf_crf_nll_fwd(const float* emissions, const int64_t* tags, const uint8_t* mask, const float* start,
                      const float* end, const float* trans, float* loss, int B, int S, int C, void* workspace,
                      size_t workspace_bytes, mtvaf_stream_t stream);
int mtvaf_crf_nll_bwd(const float* grad_out, const float* emissions, const int64_t* tags, const uint8_t* mask,
                      const float* start, const float* end, const float* trans, float* demissions, float* dstart,
                      float* dend, float* dtrans, int accumulate, int B, int S, int C, void* workspace,
                      size_t workspace_bytes, mtvaf_stream_t stream);
int mtvaf_crf_viterbi(const float* emissions, const uint8_t* mask, const float* start, const float* end,
                      const float* trans, int32_t* tags_out, int32_t* lens_out, int B, int S, int C,
                      mtvaf_stream_t stream);

/* ---- visual prompt generator + VAO loss ----------------------------------------------------------------------
 * replaces TVNetSAModel2.get_visual_prompt's split-mean / gates / gated sums / cat / reshape
 * (bert_model.py:544-545, 566-585) and the KLDiv(batchmean) ANP loss (:549-563).
 * enc [NI,B,L,4W] (encoder_conv output, NI = 1+n_aux), sm [NI*B,L*W], gate/logits [NI*B,NL*4],
 * pkv [NL,2,B,(NI*L)*(W/2)] -- the slabs mtvaf_prefix_attn_* read. */
int mtvaf_split_mean(const float* enc, float* sm, long rows, int W, mtvaf_stream_t stream);
int mtvaf_gate_fwd(const float* logits, float* gate, long ngroups, mtvaf_stream_t stream);
int mtvaf_prompt_mix_fwd(const float* enc, const float* gate, float* pkv, int NI, int B, int L, int W, int NL,
                         mtvaf_stream_t stream);
int mtvaf_prompt_mix_bwd_gate(const float* enc, const float* dpkv, const float* logits, const float* gate,
                              float* dgate_part, float* dlogits, int NI, int B, int L, int W, int NL,
                              mtvaf_stream_t stream);
int mtvaf_prompt_mix_bwd_enc(const float* gate, const float* dpkv, const float* dsm, float* denc, int NI, int B,
                             int L, int W, int NL, mtvaf_stream_t stream);
int mtvaf_kl_logsoftmax_fwd(const float* logits, const float* target, float* loss, float* row_ws, int B, int N,
                            mtvaf_stream_t stream);
int mtvaf_kl_logsoftmax_bwd(const float* grad_out, float gscale, const float* logits, const float* target,
                            float* dlogits, int B, int N, mtvaf_stream_t stream);
int mtvaf_mean_l_fwd(const float* enc, float* out, long n, int L, int W4, mtvaf_stream_t stream);
int mtvaf_mean_l_bwd(const float* dmean, float* denc, long n, int L, int W4, mtvaf_stream_t stream);

/* ---- span model heads: TVNetSAModel.classification + the loss of TVNetSAModel.forward --------------------------
 * Replaces models/bert_model.py:140-179 (flatten_emb_by_sentence, get_span_representation,
 * get_self_att_representation), :181-190 (distant_cross_entropy), :288-303 (CrossEntropyLoss on the span logits).
 * mask [B,S] u8, span_starts/ends [B,M] int64 (token offsets inside each sentence), seq [B*S,H] fp32 (H % 4 == 0,
 * H <= 1024).  `index` is a device int32 block of mtvaf_span_index_ints(B,S,M) entries that mtvaf_span_index fills
 * (valid-token compaction, per-span offsets/widths, text length and JR = widest span, clamped to S) -- the sizes
 * the reference reads back to the host stay on the device.  pooled [B*M,H], stats [B*M,2].
 * Contract: 0 <= span_starts, span widths <= S (wider spans are clamped; the reference would index past S only
 * for spans that leave the sentence). */
size_t mtvaf_span_index_ints(int B, int S, int M);
int mtvaf_span_index(const uint8_t* mask, const int64_t* span_starts, const int64_t* span_ends, int* index, int B,
                     int S, int M, mtvaf_stream_t stream);
int mtvaf_span_pool_fwd(const float* seq, const float* w_unary, const float* b_unary, const int* index,
                        float* pooled, float* stats, int B, int S, int M, int H, mtvaf_stream_t stream);
size_t mtvaf_span_pool_bwd_workspace_bytes(int B, int S, int M, int H);
/* dseq [B*S,H] overwritten (deterministic, no atomics); *dw_part_out [B*M,H] / *db_part_out [B*M] point into the
 * workspace: their column sums (mtvaf_colsum) are the unary_affine weight / bias gradients. */
int mtvaf_span_pool_bwd(const float* dpooled, const float* pooled, const float* stats, const float* seq,
                        const float* w_unary, const float* b_unary, const int* index, float* dseq,
                        float** dw_part_out, float** db_part_out, int B, int S, int M, int H, void* ws,
                        size_t ws_bytes, mtvaf_stream_t stream);
/* loss (+)= scale * -mean_b( sum_s pos log_softmax(logits)_s / sum_s pos );  logits element (b,s) at
 * logits[(b*S+s)*ld] (the start / end logits are the two columns of the binary_affine output, ld = 2);
 * positions [B,S] fp32 multi-hot; row_ws [B,3] is kept for the backward. */
int mtvaf_distant_ce_fwd(const float* logits, int ld, const float* positions, float* loss, float* row_ws, int B,
                         int S, float scale, int accumulate, mtvaf_stream_t stream);
int mtvaf_distant_ce_bwd(const float* grad_out, float scale, const float* logits, int ld, const float* positions,
                         const float* row_ws, float* dlogits, int ldd, int B, int S, mtvaf_stream_t stream);
/* mean cross entropy over rows with label != -100 (C <= 64); ws2 [2] is kept for the backward. */
int mtvaf_ce_fwd(const float* logits, const int64_t* labels, float* loss, float* ws2, int N, int C,
                 mtvaf_stream_t stream);
int mtvaf_ce_bwd(const float* grad_out, const float* logits, const int64_t* labels, const float* ws2,
                 float* dlogits, int N, int C, mtvaf_stream_t stream);

/* Cutoff augmentation on the embedding output (modules/augument.py:99-159): out = x * row_keep[b,s] * col_keep[b,:]
 * (either mask may be NULL); x/out [B,S,H] fp32, row_keep [B*S], col_keep [B,H].  Self-adjoint: the backward is the
 * same call on the gradient. */
int mtvaf_mask_mul(const float* x, const float* row_keep, const float* col_keep, float* out, int B, int S, int H,
                   mtvaf_stream_t stream);

/* bf16-OPERAND GEMM of the mixed-precision mode (new functionality: the reference has no mixed precision, SURVEY.md
 * fact 8): C[M,N] = opA[M,K] . opB[K,N] with A, B stored as bf16 and fp32 accumulation.  layout 0 (KC): reduction index
 * contiguous (A[m][k], B[n][k]); layout 1 (KM): reduction index is the row (A[k][m], B[k][n]) -- so the forward product
 * (0,0), dX = dY.W (0,1) and dW = dY^T.X (1,1) all read the same row-major bf16 tensors (transposing LDS reads; no
 * transposed copies).  Results: C32 fp32 and / or C16 bf16 (either may be NULL); accumulate adds into C32.  epi 0 none,
 * 1 bias + erf-GELU (pre-activation saved to aux16 as bf16), 3 multiply by GELU'(aux16).  colpart [M/128][N] (optional):
 * per-tile column sums of the result, finished by mtvaf_colsum_small (the bias gradient of the layer that produced the
 * operand).  Deterministic split-K as mtvaf_gemm_f32 (fp32 result only).  Aligned shapes only (M % 128, N % 128 or N % 96, K % 64, ld % 8, 16-byte aligned pointers): MTVAF_ERR_SHAPE / _ALIGN otherwise.
 * tile: 0 auto, 1 128x96, 2 128x128, 3 256x128 (8 waves; layout_a 0, M % 256 == 0, no colpart), 4 256x192 (8 waves;
 * layout_a 0, M % 256 == 0, N % 192 == 0); stages: 0 auto, 2 (two blocks per CU), 3..5 (one block, stages - 1 k-tiles
 * in flight; the 256x192 tile always runs its 2-stage ring).
 * mtvaf_cast_bf16 makes the bf16 copies of the fp32 master weights. */
int mtvaf_gemm_bf16x(int layout_a, int layout_b, const void* A, int lda, const void* B, int ldb, float* C32, int ldc32,
                     void* C16, int ldc16, int M, int N, int K, const float* bias, int epi, void* aux16, int ldaux,
                     int accumulate, float* colpart, int allow_split, void* workspace, size_t workspace_bytes, int tile,
                     int splits, int stages, mtvaf_stream_t stream);
/* ... for a weight-gradient product (KM x KM) whose operand A is exactly zero outside the listed 64-row k-tiles of the token
 * axis (mtvaf_build_ktiles with bk = 64): see mtvaf_gemm_f32_ktiles. */
int mtvaf_gemm_bf16x_ktiles(int layout_a, int layout_b, const void* A, int lda, const void* B, int ldb, float* C32, int ldc32,
                            void* C16, int ldc16, int M, int N, int K, const float* bias, int epi, void* aux16, int ldaux,
                            int accumulate, float* colpart, int allow_split, void* workspace, size_t workspace_bytes, int tile,
                            int splits, int stages, const int* klist, const int* kcnt, mtvaf_stream_t stream);
int mtvaf_colsum_small(const float* part, int rows, int cols, float* out, int accumulate, mtvaf_stream_t stream);
/* mtvaf_gemm_f32 computed on the bf16 matrix pipe: each fp32 operand value is split into three bf16 planes while its tile
 * is staged into the LDS and a product is the six significant bf16 MFMA products, accumulated in fp32 (the dropped terms are
 * below 2^-26 |a||b|: accuracy of the fp32 pipe; operands and results stay fp32 in memory).  Same arguments as
 * mtvaf_gemm_f32 (modeling_bert.py:266, 283-284, 353, 420-421, 433 and their autograd backward); shapes it does not cover
 * (not whole 64x64 tiles, K % 32 != 0, unaligned operands) run the fp32 pipe.  mtvaf_f32_split(1 / 0): mtvaf_gemm_f32 and
 * mtvaf_gemm_f32_ktiles use it everywhere they can / never (default: 1; MTVAF_F32_SPLIT=0 in the environment: 0); -1 queries. */
int mtvaf_gemm_f32x3(int layout_a, int layout_b, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                     int M, int N, int K, const float* bias, int epi, float* aux, int ldaux, int accumulate,
                     int allow_split, void* workspace, size_t workspace_bytes, int cfg, int splits,
                     mtvaf_stream_t stream);
int mtvaf_f32_split(int on);
/* profiling hook (tools/x3_trace.py): block 0 of every following wave-specialised split launch stores per wave and k-tile the
 * shader clock around the tile barrier into buf ([8 waves][64 k-tiles][4] + 17 int64 on the device); NULL switches it off */
/* Round 5: a dense product in front of a LayerNorm may leave its split-K slabs unreduced -- the LayerNorm adds them itself, in
 * the order the reduction launch would (modeling_bert.py:353-355, 433-435 and their autograd backward).  mtvaf_gemm_f32_slabs =
 * mtvaf_gemm_f32 with a plain epilogue: *splits_out == 1: C holds the result (+ bias, accumulate); s > 1: `workspace` holds s slabs
 * [M][N] (slab stride M * N floats), neither bias nor accumulate applied, C untouched.  mtvaf_dropout_res_ln_fwd_slabs: x = slab 0
 * + ... + bias is stored to x_out (the backward pass reads it), then as mtvaf_dropout_res_ln_fwd.  mtvaf_dropout_res_ln_bwd_rows_slabs:
 * dout = (slab 0 + ...) + dout_base, then as mtvaf_dropout_res_ln_bwd_rows. */
int mtvaf_gemm_f32_slabs(int layout_a, int layout_b, const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N,
                         int K, const float* bias, int accumulate, void* workspace, size_t workspace_bytes, int* splits_out,
                         mtvaf_stream_t stream);
int mtvaf_dropout_res_ln_fwd_slabs(const float* slabs, int nslab, const float* bias, float* x_out, const float* res, const float* gamma,
                                   const float* beta, float* out, float* mean, float* rstd, int M, int H, float eps, float p_drop,
                                   uint64_t seed, uint64_t offset, void* out_bf16, mtvaf_stream_t st);
int mtvaf_dropout_res_ln_bwd_rows_slabs(const float* dout_base, const float* slabs, int nslab, const float* x, const float* res,
                                        const float* gamma, const float* mean, const float* rstd, float* dx, float* dres,
                                        int dres_accumulate, int M, int H, float p_drop, uint64_t seed, uint64_t offset, float* part,
                                        void* dx_bf16, mtvaf_stream_t st);
/* Round 5, pre-split operands: the two LayerNorm kernels that ALSO write the tile-blocked plane image [H / 32][3][M][32] of their
 * output (H % 32 == 0), bit for bit what mtvaf_f32_split_planes would write behind them.  _fwd_planes: nslab == 0 -> x is the dense
 * output (as mtvaf_dropout_res_ln_fwd), nslab >= 1 -> x = the product's unreduced slabs (as mtvaf_dropout_res_ln_fwd_slabs).
 * _bwd_rows_planes: nslab == 0 -> dout = dout_base (as mtvaf_dropout_res_ln_bwd_rows), nslab >= 1 -> dout = slabs + dout_base; dx may
 * be NULL (a gradient only GEMMs read). */
int mtvaf_dropout_res_ln_fwd_planes(const float* x, int nslab, const float* bias, float* x_out, const float* res, const float* gamma,
                                    const float* beta, float* out, float* mean, float* rstd, int M, int H, float eps, float p_drop,
                                    uint64_t seed, uint64_t offset, void* out_planes, mtvaf_stream_t st);
int mtvaf_dropout_res_ln_bwd_rows_planes(const float* dout_base, const float* slabs, int nslab, const float* x, const float* res,
                                         const float* gamma, const float* mean, const float* rstd, float* dx, float* dres,
                                         int dres_accumulate, int M, int H, float p_drop, uint64_t seed, uint64_t offset, float* part,
                                         void* dx_planes, mtvaf_stream_t st);
int mtvaf_f32x3_trace(void* buf);

/* Pre-split operands (csrc/gemm_f32p.hip, round 5; the same nn.Linear products, modeling_bert.py:266, 283-284, 353, 420-421,
 * 433).  mtvaf_f32_split_planes writes the three bf16 planes x = x1 + x2 + x3 of an fp32 matrix [rows][cols] (the split of
 * csrc/gemm_f32x3.hip, bit for bit) as a plane image addressed by three BYTE strides: plane -> plane, row -> row, 32-column
 * k-tile -> k-tile (natural form: cols * 2 * rows / ld * 2 / 64; tile-blocked form [k-tile][plane][rows][32]: rows * 64 / 64 /
 * 3 * rows * 64).  mtvaf_gemm_f32p: C[M,N] = A[M,K] . op(B) (+ bias, epilogue, accumulate) from the plane images of both
 * operands on v_mfma_f32_16x16x32_bf16, nothing split inside the k-loop; layout_b 0: B [N][K] (forward), 1: B [K][N] (the dX
 * products: natural plane image, b_row = N * 2, b_kt = 32 * N * 2, b_col = 256); layout_a 1 (with layout_b 1): A [K][M] too, the
 * weight-gradient products; M, N % 128 == 0, K % 32 == 0; splits > 1 = deterministic
 * split-K through `workspace`; ablate = 0 (research switches: 1 no MFMAs, 2 no requests, 4 no fragment reads: timing only).
 * mtvaf_f32p_trace: shader-clock stamps of block 0 ([8 waves][64 k-tiles][2] + 17 int64), NULL = off. */
int mtvaf_f32_split_planes(const float* src, void* dst, int rows, int cols, int ld, long s_plane, long s_row, long s_kt,
                           mtvaf_stream_t stream);
int mtvaf_gemm_f32p(int layout_a, const void* Aplanes, long a_plane, long a_row, long a_kt, long a_col, int layout_b, const void* Bplanes,
                    long b_plane, long b_row, long b_kt, long b_col, float* C, int ldc, int M, int N, int K, const float* bias, int epi,
                    float* aux, int ldaux, int accumulate, int splits, void* workspace, size_t workspace_bytes, int ablate,
                    mtvaf_stream_t stream);
int mtvaf_f32p_trace(void* buf);
/* The tile of mtvaf_gemm_f32p* (round 6): which products may run on the 128 x 256 tile (csrc/gemm_f32pw.hip; N % 256 == 0) instead of
 * the 128 x 128 tile -- a mask: 1 = forward products (layout_b 0; these may also take a 128 x 192 tile when N % 192 == 0 and no
 * plane-image result is asked for), 2 = dX products (layout_b 1), 4 = weight gradients (layout_a 1; the grouped launch when every
 * N_i % 256 == 0).  Among the admitted tiles a launch takes the cheapest by an estimate of rounds x tile time (QKV forward at 2432
 * rows: 192 columns, one round of 228 tiles; FFN-1: 256; N = 768 and 4096-row launches: mostly 128 x 128).  8 = never the 128 x 128
 * tile where another can serve (tests), 16 = never the 128 x 192 tile.  mask >= 0 sets, -1 queries; returns the mask in force
 * (default: MTVAF_P16_WIDE, 7 if unset).  Process-global; placement only -- the kernels issue the same MFMA products in the same
 * order for every output element and agree bit for bit. */
int mtvaf_f32p_wide(int mask);
/* research entry: up to four weight-gradient products C_i [M_i][N_i] = A_i^T . B_i from plane images (A_i [K][M_i], B_i [K][N_i], the
 * token rows K shared) in ONE unsplit launch over the 128 x 128 tiles of all of them -- what the grouped launch of
 * mtvaf_gemm_f32_dw_group becomes with pre-split operands (no bias sums, no k-tile list).  strides: eight byte strides per product
 * (a_plane, a_row, a_kt, a_col, b_plane, b_row, b_kt, b_col; natural image of [K][C]: C*K*2, C*2, 64*C, 256; tile-blocked image
 * [C/32][3][K][32]: K*64, 64, 2048, 12*K*64).  M_i, N_i % 128 == 0, K % 32 == 0. */
int mtvaf_gemm_f32p_dw_group(int n, const void* const* Aplanes, const void* const* Bplanes, const long* strides, float* const* C, const int* ldc,
                             const int* M, const int* N, int K, mtvaf_stream_t stream);
/* ... with up to eight column-sum jobs as extra blocks of the same launch: job j sums the fp32 matrix cs_src[j] [cs_rows[j]][cs_cols[j]]
 * (leading dimension cs_ld[j]) over its rows into cs_dst[j]; fixed summation order.  The small reductions of a layer's backward pass
 * (QKV / FFN-1 bias gradients, the two LayerNorm-backward finishes) ride in the CUs the last round of tiles leaves idle. */
int mtvaf_gemm_f32p_dw_group_colsum(int n, const void* const* Aplanes, const void* const* Bplanes, const long* strides, float* const* C,
                                    const int* ldc, const int* M, const int* N, int K, int njobs, const float* const* cs_src,
                                    const int* cs_rows, const int* cs_cols, const int* cs_ld, float* const* cs_dst, mtvaf_stream_t stream);
/* mtvaf_gemm_f32p with a plain epilogue that leaves a split-K plan's slabs unreduced (as mtvaf_gemm_f32_slabs: *splits_out = 1 -> C
 * holds the result; s > 1 -> `workspace` holds s slabs [M][N], bias / accumulate not applied): the LayerNorm kernels behind the Wo /
 * FFN-2 / FFN-1 dX products add them (modeling_bert.py:353-355, 433-435 and their backward). */
int mtvaf_gemm_f32p_slabs(int layout_a, const void* Aplanes, long a_plane, long a_row, long a_kt, long a_col, int layout_b, const void* Bplanes,
                          long b_plane, long b_row, long b_kt, long b_col, float* C, int ldc, int M, int N, int K, const float* bias,
                          int accumulate, int splits, void* workspace, size_t workspace_bytes, int* splits_out, mtvaf_stream_t stream);
/* mtvaf_gemm_f32p, unsplit, whose result is written as a tile-blocked plane image c_planes [N / 32][3][M][32] -- beside the fp32
 * result (C != NULL) or instead of it (C == NULL: a tensor that only GEMMs read: the GELU output between the two FFN products, the
 * GELU' output between their dX products) -- and, optionally, its per-tile column sums colpart [M / 128][N] (finished by
 * mtvaf_colsum_small: the FFN-1 bias gradient). */
int mtvaf_gemm_f32p_ep(int layout_a, const void* Aplanes, long a_plane, long a_row, long a_kt, long a_col, int layout_b, const void* Bplanes,
                       long b_plane, long b_row, long b_kt, long b_col, float* C, int ldc, void* c_planes, float* colpart, int M, int N, int K,
                       const float* bias, int epi, float* aux, int ldaux, int accumulate, mtvaf_stream_t stream);

/* The (up to four) weight-gradient products of one encoder layer in fp32, dW_i[M_i,N_i] = A_i^T . B_i with A_i [K,M_i], B_i [K,N_i]
 * row-major, as ONE launch of the 128x96 LDS-DMA kernel (autograd backward of modeling_bert.py:266, 283-284, 353, 420-421, 433);
 * klist / kcnt as mtvaf_gemm_f32_ktiles (NULL: the whole reduction); deterministic split-K through per-product slabs in
 * `workspace` (splits <= 0: planned).  M_i % 128 == 0, N_i % 96 == 0, K % 32 == 0.  Under the split arithmetic (mtvaf_f32_split),
 * for K > 1024 and M_i, N_i % 128 == 0, the launch is the GROUP form of the wave-specialised split kernel on 128x128 tiles
 * instead (unsplit once the tiles cover three quarters of the CUs: 432 tiles at BERT-base).  mtvaf_dw_group_rows: the executor
 * groups layers of at most that many token rows on the fp32 pipe's ring (default 1024, MTVAF_DW_GROUP_ROWS; rows >= 0 sets,
 * -1 queries); mtvaf_dw_group_wanted(rows, H, I): the executor's whole rule (1: this layer's weight gradients go out grouped). */
int mtvaf_gemm_f32_dw_group(int n, const float* const* A, const int* lda, const float* const* B, const int* ldb, float* const* C,
                            const int* ldc, const int* M, const int* N, int K, const int* klist, const int* kcnt,
                            void* workspace, size_t workspace_bytes, int splits, mtvaf_stream_t stream);
/* bytes of split-K slabs that call needs for these products (0: no split planned; too little workspace is MTVAF_ERR_WORKSPACE,
 * never a silently different plan) */
size_t mtvaf_gemm_f32_dw_group_workspace_bytes(int n, const int* M, const int* N, int K, int splits);
/* The same launch + the bias gradients that go with the weight gradients: dbias (nullable array of n nullable pointers),
 * dbias[i][0 .. M_i) <- column sums of A_i over its K rows (A_i is the dY of the dense layer: autograd backward of the bias of
 * modeling_bert.py:266, 283-284, 420-421), 16-byte aligned, overwritten.  The unsplit grouped launch of the split kernel
 * takes them from the A tiles it stages anyway; any other plan runs mtvaf_colsum behind the products (workspace_bytes must
 * then also cover mtvaf_colsum_workspace_bytes(K, M_i)). */
int mtvaf_gemm_f32_dw_group_bias(int n, const float* const* A, const int* lda, const float* const* B, const int* ldb,
                                 float* const* C, const int* ldc, const int* M, const int* N, int K, const int* klist,
                                 const int* kcnt, float* const* dbias, void* workspace, size_t workspace_bytes, int splits,
                                 mtvaf_stream_t stream);
int mtvaf_dw_group_rows(int rows);
int mtvaf_dw_group_wanted(int rows, int H, int I);

/* Stream-K form of the 256x256 eight-phase bf16 kernel (csrc/gemm_bf16p.hip; tile 5 = tile-per-block, tile 6 = stream-K forced,
 * tile 0 = the planner decides): one block per CU walks an equal run of the k-tile steps of all output tiles; a tile whose
 * reduction is shared by several blocks is combined inside the launch (fp32 contributions in block order = k order, published
 * with an agent-scope release and consumed behind an agent-scope acquire: deterministic for a given scratch size).  The
 * contributions live in a caller-owned scratch attached per stream: at least mtvaf_streamk_scratch_bytes(256) bytes, 16-byte
 * aligned, zero-initialised by the caller, alive until replaced or detached (scratch = NULL).  Replaces nothing in the
 * reference (new functionality of the mixed-precision mode); the products are those of modeling_bert.py:266, 283-284, 353,
 * 420-421, 433 and their autograd backward. */
int mtvaf_streamk_attach(void* scratch, size_t bytes, mtvaf_stream_t stream);
size_t mtvaf_streamk_scratch_bytes(int blocks);
int mtvaf_streamk_attached(mtvaf_stream_t stream); /* -> blocks the attached scratch serves, 0: none */
/* The (up to four) weight-gradient products of one encoder layer, dW_i[M_i,N_i] = A_i^T . B_i with A_i [K,M_i], B_i [K,N_i] bf16
 * row-major and fp32 results, in ONE stream-K launch (autograd backward of modeling_bert.py:266, 283-284, 353, 420-421, 433).
 * Needs an attached scratch (MTVAF_ERR_WORKSPACE otherwise); M_i % 256 == 0, N_i % 256 == 0, K % 64 == 0. */
int mtvaf_gemm_bf16x_dw_group(int n, const void* const* A, const int* lda, const void* const* B, const int* ldb, float* const* C32,
                              const int* ldc32, const int* M, const int* N, int K, mtvaf_stream_t stream);

/* Prefix attention of the mixed-precision mode: the algorithm of mtvaf_prefix_attn_fwd / _bwd (same key order, mask,
 * dropout hash) on bf16 operands with fp32 softmax statistics and accumulation.  qkv16 [B*S,3H], pk16 / pv16 [B,P*H],
 * ctx16 [B*S,H], dctx16, dqkv16: bf16; lse [B,NH,S], dpk / dpv [B,P*H]: fp32.  The backward also emits per-block column
 * sums partq [B*ceil(S/64)][H] (dQ) and partkv [B*ceil((P+S)/64)][2H] (dK | dV over the text keys): summed over their rows
 * (mtvaf_colsum_small / mtvaf_colsum) they are the query / key / value bias gradients. */
int mtvaf_prefix_attn_bf16_fwd(const void* qkv16, const void* pk16, const void* pv16, const float* addmask, void* ctx16,
                               float* lse, int B, int S, int P, int NH, int head_dim, float p_drop, uint64_t seed,
                               uint64_t offset, mtvaf_stream_t stream);
int mtvaf_prefix_attn_bf16_bwd(const void* dctx16, const void* qkv16, const void* pk16, const void* pv16,
                               const float* addmask, const void* ctx16, const float* lse, void* dqkv16, float* dpk,
                               float* dpv, float* partq, float* partkv, int B, int S, int P, int NH, int head_dim,
                               float p_drop, uint64_t seed, uint64_t offset, mtvaf_stream_t stream);
/* mtvaf_prefix_attn_bf16_bwd under the zero-tail contract of mtvaf_prefix_attn_bwd_tail (the key side's query loop stops at the
 * last unmasked position; same bits) */
int mtvaf_prefix_attn_bf16_bwd_tail(const void* dctx16, const void* qkv16, const void* pk16, const void* pv16,
                                    const float* addmask, const void* ctx16, const float* lse, void* dqkv16, float* dpk,
                                    float* dpv, float* partq, float* partkv, int B, int S, int P, int NH, int head_dim,
                                    float p_drop, uint64_t seed, uint64_t offset, int zero_tail, mtvaf_stream_t stream);
/* ... over PACKED token rows (padding-free execution, see mtvaf_prefix_attn_varlen_fwd): cu [B+1] int32, no mask read;
 * partq / partkv keep their padded row counts (blocks beyond a sentence write zeros). */
int mtvaf_prefix_attn_bf16_varlen_fwd(const void* qkv16, const void* pk16, const void* pv16, const int* cu, int pad_rows, void* ctx16,
                                      float* lse, int B, int S, int P, int NH, int head_dim, float p_drop, uint64_t seed,
                                      uint64_t offset, mtvaf_stream_t stream);
int mtvaf_prefix_attn_bf16_varlen_bwd(const void* dctx16, const void* qkv16, const void* pk16, const void* pv16, const int* cu,
                                      int pad_rows, const void* ctx16, const float* lse, void* dqkv16, float* dpk, float* dpv, float* partq,
                                      float* partkv, int B, int S, int P, int NH, int head_dim, float p_drop, uint64_t seed,
                                      uint64_t offset, mtvaf_stream_t stream);
int mtvaf_cast_bf16(const float* x, int ldx, void* out, int ldo, void* outT, int ldt, int R, int C,
                    mtvaf_stream_t stream);

/* ---- native per-layer executor ----------------------------------------------------------
 * One call enqueues ALL kernels of a BertLayer forward / backward (models/modeling_bert.py:439-522 and its autograd
 * backward) -- the same kernels in the same order as the Python engine, composed inside the library -- so a training step
 * costs one host call per layer and direction instead of ~65.  `bf16` selects the kernel family: 0 = fp32 (qkv / cx / pre /
 * act are float*, weights w*, prefix slabs float*), 1 = mixed precision (those four are bf16, weights the bf16 images
 * w*_h, x_h / h1_h / h2_h bf16 copies, prefix slabs bf16).  Dropout sites: attention `offset`, attention output
 * `offset + 1`, FFN output `offset + 2`.  All buffers caller-owned; workspaces as the composed entry points need them. */
typedef struct {
  int B, S, P, NH, H, I;
  int bf16;
  float eps, p_hidden, p_attn;
  uint64_t seed, offset;
  const float *wqkv, *wo, *w1, *w2;
  const void *wqkv_h, *wo_h, *w1_h, *w2_h;
  const float *bqkv, *bo, *g1, *b1, *bi1, *bi2, *g2, *b2;
  const float* x;
  const void* x_h;
  const void *pk, *pv;
  const float* addmask;
  void *qkv, *cx;
  float *lse, *a, *h1;
  void* h1_h;
  float *mean1, *rstd1;
  void *pre, *act;
  float *f, *h2;
  void* h2_h;
  float *mean2, *rstd2;
  void* ws; size_t ws_bytes;
  /* padding-free execution: cu != NULL -> the token tensors (x, qkv ... h2, and the gradient buffers) hold Mp
   * PACKED rows -- the Mv unmasked tokens sentence by sentence (cu [B+1] int32 row offsets), then Mp - Mv zero rows that
   * pad the image to whole 128-row tiles; lse / delta keep [B,NH,S].  cu == NULL: the padded [B*S] layout. */
  const int* cu;
  int Mv, Mp;
  /* fp32 mode, optional (round 5; all of them, with the weights' plane images in w*_h, packed rows, H and I % 128 == 0): PRE-SPLIT
   * operands -- tile-blocked plane images ([cols / 32][3][Mp][32] bf16: mtvaf_f32_split_planes with strides Mp * 64 / 64 / 3 * Mp * 64) of
   * the layer input x_p (given), and of the attention context, the attention-block output, the GELU output and the layer output
   * (written here: cx_p, h1_p, act_p, h2_p = the next layer's x_p, or NULL).  The eight forward / dX products and the grouped weight
   * gradients then run on the pre-split kernels of csrc/gemm_f32p.hip (nothing split inside their k-loops). */
  const void* x_p;
  void *cx_p, *h1_p, *act_p, *h2_p;
} mtvaf_layer_t;

typedef struct {
  float* dh;                              /* in: d loss / d h2 [M,H]; out: d loss / d x */
  float* dh1;
  void *df, *dpre, *da, *dctx, *dqkv;
  float *part, *partq, *partkv;           /* bf16 mode: epilogue / attention column-sum partials */
  float* delta;                           /* fp32 mode: [B,NH,S] */
  float *dwqkv, *dbqkv, *dwo, *dbo, *dg1, *db1, *dw1, *dbi1, *dw2, *dbi2, *dg2, *db2;
  float *dpk, *dpv;
  void* ws_main; size_t ws_main_bytes;
  void* ws_side; size_t ws_side_bytes;
  /* optional: k-tile list of the token axis for the weight-gradient products (mtvaf_build_ktiles; 32-row tiles in fp32
   * mode, 64-row tiles in bf16 mode) */
  const int* klist;
  const int* kcnt;
  /* the caller's word: token rows behind a sentence's last unmasked position carry exactly-zero gradients (what a k-tile list
   * implies as well): the attention backward stops its query loops there (mtvaf_prefix_attn_bwd_tail) */
  int zero_tail;
  /* optional (both or neither): the layer's own LayerNorm-backward partials of the FFN / attention block,
   * mtvaf_ln_bwd_workspace_bytes(M, H) each -- their column sums then run on `side` instead of the main chain */
  float *lnpart2, *lnpart1;
  /* fp32 mode with pre-split operands (see mtvaf_layer_t::x_p): scratch plane images of the four upstream gradients that are GEMM
   * operands -- [Mp, H], [Mp, I], [Mp, H], [Mp, 3H] -- alive until the second stream has finished the layer's weight gradients */
  void *df_p, *dpre_p, *da_p, *dqkv_p;
} mtvaf_layer_grads_t;

int mtvaf_encoder_layer_fwd(const mtvaf_layer_t* layer, mtvaf_stream_t stream);
/* weight-gradient products on `side` behind events of `main` (side == main serialises); settle != 0: `side` also waits for
 * the layer's last main-stream kernel (an optimizer update hanging off the caller's hook must not overtake it). */
int mtvaf_encoder_layer_bwd(const mtvaf_layer_t* layer, const mtvaf_layer_grads_t* grads, mtvaf_stream_t main,
                            mtvaf_stream_t side, int settle);
size_t mtvaf_layer_struct_bytes(int which /* 0: mtvaf_layer_t, 1: mtvaf_layer_grads_t */);

/* ---- optimizer step ------------------------------------------------------------
 * AdamW exactly as torch.optim.AdamW, which the reference trainer builds (modules/train.py:887-926) and steps
 * (:621-625): decoupled weight decay, bias corrections passed in (bc1 = 1 - beta1^t, bc2_sqrt = sqrt(1 - beta2^t)).
 * mtvaf_adamw updates ONE flat fp32 tensor (an encoder layer's parameter buffer: all 16 parameters of a layer in one
 * launch, enqueued behind that layer's gradient all-reduce so the update overlaps the rest of the backward pass) and
 * optionally writes the updated parameters rounded to bf16 (the GEMM operand shadow of the bf16 compute mode);
 * max_blocks > 0 caps its grid (a background update that trickles under MFMA-bound kernels instead of taking the whole
 * HBM rate from them for its duration; 0 = full width).
 * mtvaf_adamw_multi updates `count` tensors of one parameter group per launch (host arrays of device pointers). */
int mtvaf_adamw(float* p, const float* g, float* m, float* v, long n, float lr, double beta1, double beta2, float eps,
                float weight_decay, float bc1, float bc2_sqrt, float grad_scale, void* p_bf16, int max_blocks,
                mtvaf_stream_t stream);
/* mtvaf_adamw over a flat buffer (n % 4 == 0, 16-byte aligned) that ALSO rewrites the plane images (the pre-split operand form of
 * csrc/gemm_f32p.hip, round 5) of nseg <= 4 row-major fp32 matrices inside it -- an encoder layer's wqkv / wo / w1 / w2: matrix s starts
 * at element seg_begin[s] (% 4 == 0), is [seg_rows[s]][seg_cols[s]] (cols % 32 == 0) and has its tile-blocked image
 * [cols / 32][3][rows][32] bf16 at seg_img[s].  The same update as mtvaf_adamw, bit for bit; the images match the updated weights. */
int mtvaf_adamw_planes(float* p, const float* g, float* m, float* v, long n, float lr, double beta1, double beta2, float eps,
                       float weight_decay, float bc1, float bc2_sqrt, float grad_scale, int nseg, const long* seg_begin,
                       const int* seg_rows, const int* seg_cols, void* const* seg_img, int max_blocks, mtvaf_stream_t stream);
int mtvaf_adamw_multi(int count, float* const* p, const float* const* g, float* const* m, float* const* v,
                      const long* n, float lr, double beta1, double beta2, float eps, float weight_decay, float bc1,
                      float bc2_sqrt, float grad_scale, mtvaf_stream_t stream);

/* ---- bf16 gradient wire format of the data-parallel exchange (mtvaf_amd/parallel.py; new functionality, the
 * reference never synchronises gradients: MTVAF_training.py:305-309).  pack: dst[i] = bf16(src[i]), zero padding up to
 * npad (npad % 8 == 0); reduce: out[c] = bf16(scale * sum_r recv[r*chunk + c]) with fp32 accumulation in rank order;
 * unpack: dst[i] = float(src[i]).  The collectives themselves (all_to_all, all_gather: RCCL over xGMI) are issued
 * through torch.distributed -- the section 8(b) proposal of mtvaf_rccl_* entry points is not built, see INTEGRATION.md. */
int mtvaf_grad_pack_bf16(const float* src, void* dst, long n, long npad, mtvaf_stream_t stream);
int mtvaf_grad_reduce_bf16(const void* recv, void* out, int world, long chunk, float scale, mtvaf_stream_t stream);
int mtvaf_grad_unpack_bf16(const void* src, float* dst, long n, mtvaf_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MTVAF_HIP_H */
