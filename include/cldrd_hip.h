/* cldrd_hip.h - C ABI of libcldrd_hip.so: the MI355X (gfx950) kernels behind the CL-DRD hot path.
 *
 * The reference (HansiZeng/CL-DRD) is pure Python and has no FFI: its hot path bottoms out in PyTorch /
 * HuggingFace / faiss calls.  Each entry point below replaces one of those call sites; the Python host code in
 * cl-drd_amd/ (a mirror of the reference's models / losses / retriever modules) binds them with ctypes, and
 * INTEGRATION.md shows the stub a maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HBM) unless stated; the caller owns all memory;
 *   - bf16 tensors are passed as void* (raw bfloat16 bits), fp32 as float*, token ids / masks as int64;
 *   - matrices are row-major; `ld*` are row strides in ELEMENTS;
 *   - `stream` is a hipStream_t (NULL = default stream); all calls are asynchronous and graph-capturable
 *     (no allocation, no host synchronisation inside);
 *   - return value 0 = launched; non-zero = rejected, cldrd_last_error() gives the reason (thread local).
 *     Nothing is computed on the CPU: without a gfx950 device the launches fail.
 *   - dropout: keep-mask = hash(seed, element index) (counter based); the backward entry points regenerate the
 *     mask from the same (p, seed), nothing is stored.  p = 0 disables it.
 */
#ifndef CLDRD_HIP_H
#define CLDRD_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

const char* cldrd_last_error(void);
int cldrd_version(void);
int cldrd_device_ok(void);            /* 1 if device 0 is a gfx950 */
/* The library reads NO environment variable.  The few kernel choices tests need to reach go through this call (process-wide):
 * key "gemm_splitk" (0 heuristic, 1 never split K, n > 1 n splits of the small-M Linear GEMM), "gemm_nt64" (1: the one-launch
 * 64 x 64 kernel for small-M GEMMs with K <= 1024 when gemm_splitk is 0, 0: never), "attn_fwd2" / "attn_bwd2" (1: persistent
 * attention kernels where they apply, 0: one item per workgroup).  Every choice computes the same function. */
int cldrd_set_tuning(const char* key, int value);

/* ---- encoder Linear layers -------------------------------------------------------------------------------
 * Replaces torch.nn.Linear inside the HF encoder (reference models/nway_dual_encoder.py:52,56,64 ->
 * transformers DistilBERT q_lin/k_lin/v_lin/out_lin/ffn.lin1/ffn.lin2, BERT query/key/value/dense).
 *   C[M,N] = epilogue(alpha * A[M,K] . B[N,K]^T)
 *   epilogue order: + bias[N] -> (store preact) -> act (bit 0 = erf-GELU) -> * gelu'(gelu_pre) -> dropout -> + residual
 *   act bit 1 = derivative form of the saved activation input: `preact` receives gelu'(pre-activation) instead of the
 *   pre-activation (forward, act = 3), and `gelu_pre` is taken to hold that derivative already (backward, act = 2: the epilogue
 *   multiplies by it and evaluates nothing) - what torch saves for GELU's backward is its input; saving the derivative moves the
 *   erf / exp out of the data-gradient GEMM's epilogue, which was VALU-bound.
 *   K % 64 == 0; A/B/C 16-byte aligned; out_f32 != 0 stores fp32 instead of bf16; res_f32 != 0: `residual` is fp32
 *   (the fp32 residual stream: out-projection / FFN2 add the fp32 LayerNorm output and store the fp32 pre-LN sum, as the
 *   reference's autocast does - trainer/multistep-curriculum/nway_listwise_1.py:334).
 *   fmt (enum cldrd_fmt16 below): the 16-bit format of A, B, a 16-bit C and the GELU tape.  The entry points are named "16", not "bf16": both
 *   formats run at the same MFMA rate and fp16 is the DEFAULT arithmetic mode of the package (the reference's own, CLDRD_AMP in README.md). */
enum cldrd_fmt16 {
    CLDRD_FMT_BF16 = 0,          /* A, B, 16-bit C, preact / gelu_pre: bf16 (CLDRD_AMP=bf16: every operand of every pass) */
    CLDRD_FMT_F16 = 1,           /* A, B and a 16-bit C: fp16 (evaluation passes of the fp16 mode; no preact / gelu_pre) */
    CLDRD_FMT_F16_C_BF16 = 3,    /* fp16 A and B, bf16 C: the QKV projection of an evaluation pass whose attention kernels are bf16 */
    CLDRD_FMT_F16_TAPE = 5       /* fp16 everywhere incl. preact / gelu_pre: a training pass of the fp16 mode */
};
int cldrd_gemm_nt16(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc,
                       const float* bias, const void* residual, int ldr, void* preact, const void* gelu_pre,
                       int act, float alpha, float dropout_p, unsigned long long seed, int out_f32, int res_f32, int fmt,
                       void* stream);

/* The same GEMM with the fp32 residual given as a LayerNorm still to be applied: `residual` holds the pre-LN sum s (fp32, res_f32 = 1) and
 * the epilogue adds (s - ln_mean[m]) * ln_rstd[m] * ln_gamma[n] + ln_beta[n] - exactly what cldrd_layernorm_fwd would have written
 * as its fp32 output, which it then need not write (HF: hidden = LayerNorm(...); out = dense(x) + hidden).  All four ln_* or none. */
int cldrd_gemm_nt16_ln(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc,
                          const float* bias, const void* residual, int ldr, void* preact, const void* gelu_pre,
                          int act, float alpha, float dropout_p, unsigned long long seed, int out_f32, int res_f32, int fmt,
                          const float* ln_mean, const float* ln_rstd, const float* ln_gamma, const float* ln_beta, void* stream);

/* The same with a workspace: problems of fewer than 1024 rows whose one-pass grid would leave most CUs idle (CLS-only last layer, query
 * tower) are split along K: fp32 partials in `workspace` (cldrd_gemm_nt_splitk_workspace() bytes; 0 = this shape is not split), summed in
 * a fixed order and finished with the same epilogue by a second launch.  workspace = NULL: cldrd_gemm_nt16_ln.
 * fmt != CLDRD_FMT_BF16 here also serves M >= 1024 (csrc/gemm_nt_ring16.hip).  c_copy_bf16: NULL (a bf16 copy of an fp16 C for the removed
 * "bf16 tape under an fp16 forward" mode of rounds 3-5; kept in the signature, rejected when non-NULL with any format but CLDRD_FMT_F16). */
size_t cldrd_gemm_nt_splitk_workspace(int M, int N, int K);
int cldrd_gemm_nt16_ws(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc,
                          const float* bias, const void* residual, int ldr, void* preact, const void* gelu_pre,
                          int act, float alpha, float dropout_p, unsigned long long seed, int out_f32, int res_f32, int fmt,
                          const float* ln_mean, const float* ln_rstd, const float* ln_gamma, const float* ln_beta,
                          void* c_copy_bf16, float* workspace, size_t workspace_bytes, void* stream);

/* Weight gradient dW[N1,N2] (+)= A[M,N1]^T . B[M,N2]   (A = dY, B = layer input; autograd's Linear backward), and,
 * when dbias != NULL, the bias gradient dbias[N1] (+)= column sums of A in the same pass.
 * N1, N2 multiples of 128 (or N1 % 256 == 0 and N2 % 192 == 0); rows >= M are never read (no padding contract).
 * cldrd_wgrad_group: n such problems in ONE launch (the pointer / shape arrays are HOST arrays).  The weight gradients are off the
 * critical path of the backward, so the trainer defers them: with a tower's 20+ problems in one launch every workgroup sweeps all
 * tokens of its output tile and writes dW once - no token split, no fp32 partial slabs, no reduction launches.
 * workspace (floats): cldrd_wgrad_group_workspace(...) for a group (0 when no token split is chosen),
 * cldrd_wgrad_splits(M,N1,N2) * (N1 * N2 + N1) for the single-problem form. */
int cldrd_wgrad_splits(int M, int N1, int N2);
int cldrd_wgrad16(const void* A, const void* B, float* dW, float* dbias, int M, int N1, int N2, int lda, int ldb,
                     float* workspace, size_t workspace_bytes, int accumulate, void* stream);
size_t cldrd_wgrad_group_workspace(const int* M, const int* N1, const int* N2, int n);
int cldrd_wgrad_group(const void* const* A, const void* const* B, float* const* dW, float* const* dbias, const int* M,
                      const int* N1, const int* N2, const int* lda, const int* ldb, int n, float* workspace,
                      size_t workspace_bytes, int accumulate, void* stream);

/* ---- attention (HF DistilBertSelfAttention / BertSelfAttention, head dim 64, L <= 256) --------------------
 * qkv: bf16 [nseq*L, 3*H*64] = Q | K | V;  mask: int64 [nseq, L], 0 = padded key, or NULL;
 * ctx: bf16 [nseq*L, H*64];  lse: fp32 [nseq, H, L] (NULL allowed in forward-only use). */
int cldrd_attention_fwd(const void* qkv, const long long* mask, void* ctx, float* lse, int nseq, int L, int H,
                        float dropout_p, unsigned long long seed, int fmt, void* stream);
int cldrd_attention_bwd(const void* qkv, const long long* mask, const void* ctx, const void* dctx, const float* lse,
                        void* dqkv, int nseq, int L, int H, float dropout_p, unsigned long long seed, void* stream);

/* Dropout keep bits.  For L <= 128 and at least two (sequence, head) items per CU the forward runs as a persistent loader / compute
 * kernel whose loader waves evaluate the dropout hash and leave the keep decisions behind as bits ([item][key block][query] dwords);
 * the backward for the same (nseq, L, H, dropout_p, seed) then reads them instead of hashing again.  cldrd_attention_bits_words()
 * says how many 32-bit words that is for a shape (0: this shape does not produce bits - pass null and the backward re-hashes). */
long long cldrd_attention_bits_words(int nseq, int L, int H, float dropout_p);
/* ctx_f16_copy (optional, bf16 pass only): the same context vectors in fp16 - the A operand of an fp16-operand out-projection GEMM (round 3:
 * the out-projection's operands carry most of the remaining logit drift of a 12-layer tower); ctx may then be NULL (no bf16 tape wanted). */
int cldrd_attention_fwd_bits(const void* qkv, const long long* mask, void* ctx, float* lse, int nseq, int L, int H,
                             float dropout_p, unsigned long long seed, int fmt, void* drop_bits_out, void* ctx_f16_copy, void* stream);
int cldrd_attention_bwd_bits(const void* qkv, const long long* mask, const void* ctx, const void* dctx, const float* lse,
                             void* dqkv, int nseq, int L, int H, float dropout_p, unsigned long long seed, const void* drop_bits,
                             void* stream);
/* the same with a format: fmt != CLDRD_FMT_BF16 - q / k / v, ctx, dctx and dqkv are fp16 (a training pass of the fp16 mode) */
int cldrd_attention_bwd_x(const void* qkv, const long long* mask, const void* ctx, const void* dctx, const float* lse,
                             void* dqkv, int nseq, int L, int H, float dropout_p, unsigned long long seed, const void* drop_bits, int fmt,
                             void* stream);

/* CLS-only attention of the LAST layer (the reference pools last_hidden_state[:, 0, :], models/nway_dual_encoder.py:52,56,64):
 * qc: bf16 [nseq, H*64] = queries of token 0; kv: bf16 [nseq*L, 2*H*64] = K | V of every token; ctx/dctx/dqc: bf16 [nseq, H*64];
 * probs: fp32 [nseq, H, L] (softmax row, saved for the backward); dkv: bf16 [nseq*L, 2*H*64] (every row written).
 * cldrd_add_rows_strided: dst[m * stride_rows] += src[m] for bf16 rows of d elements (puts the CLS-row gradients back). */
int cldrd_attention_cls_fwd(const void* qc, const void* kv, const long long* mask, void* ctx, float* probs, int nseq, int L,
                            int H, float dropout_p, unsigned long long seed, int fmt, void* ctx_f16_copy, void* stream);
int cldrd_attention_cls_bwd(const void* qc, const void* kv, const float* probs, const void* dctx, void* dqc, void* dkv,
                            int nseq, int L, int H, float dropout_p, unsigned long long seed, void* stream);
int cldrd_attention_cls_bwd_x(const void* qc, const void* kv, const float* probs, const void* dctx, void* dqc, void* dkv,
                            int nseq, int L, int H, float dropout_p, unsigned long long seed, int fmt, void* stream);
int cldrd_add_rows_strided(void* dst, const void* src, int M, int d, int stride_rows, int f32, void* stream);   /* f32: fp32 rows (fp32 gradient stream) */

/* The four attention calls on a PACKED batch (round 6; layout of "packed batches" below: sequence m owns rows cu_rows[m] .. cu_rows[m + 1] of every
 * [Tp, .] tensor, Tp = cu_rows[nseq]; cu_rows: int32 [nseq + 1] on the device, every length in 1 .. L).  The reference pads every sequence of a batch
 * to the longest (dataset/nway_dataset.py:105-106, dataset/sequence_dataset.py:50-51) and HF attention then computes on the padding; here keys
 * beyond a sequence's length are masked (right padding: what the tokenizer's attention_mask says; no mask tensor is read) and padding rows are
 * neither loaded nor stored.  qkv / ctx / dctx / dqkv / kv / dkv are [Tp, .]; lse, probs and the keep bits keep their [nseq, H, L(, ..)] shapes.
 * Results are bit for bit those of cldrd_unpack_rows16 -> the padded call -> cldrd_gather_rows. */
int cldrd_attention_fwd_varlen(const void* qkv_packed, const int* cu_rows, void* ctx_packed, float* lse, int nseq, int L, int H,
                               float dropout_p, unsigned long long seed, int fmt, void* drop_bits_out, void* ctx_f16_copy, void* stream);
int cldrd_attention_bwd_varlen(const void* qkv_packed, const int* cu_rows, const void* ctx_packed, const void* dctx_packed, const float* lse,
                               void* dqkv_packed, int nseq, int L, int H, float dropout_p, unsigned long long seed, const void* drop_bits,
                               int fmt, void* stream);
/* ... for a LIST of the batch's sequences (seq_list: device int32 [n_list], positions in 0 .. nseq - 1, each sequence at most Ltile <= L tokens
 * long): the launch runs the kernels of tile height Ltile.  A packed batch at L = 256 sends its sequences of at most 128 tokens (most of an
 * MS MARCO batch) through the persistent L <= 128 kernels and the others through a second call; LSE rows and dropout row keys keep the stride L
 * of the batch, so forward and backward of a sequence must be given the same L (the lists may differ). */
int cldrd_attention_fwd_varlen_list(const void* qkv_packed, const int* cu_rows, void* ctx_packed, float* lse, int nseq, int L, int H,
                                    float dropout_p, unsigned long long seed, int fmt, void* drop_bits_out, void* ctx_f16_copy,
                                    const int* seq_list, int n_list, int Ltile, void* stream);
int cldrd_attention_bwd_varlen_list(const void* qkv_packed, const int* cu_rows, const void* ctx_packed, const void* dctx_packed, const float* lse,
                                    void* dqkv_packed, int nseq, int L, int H, float dropout_p, unsigned long long seed, const void* drop_bits,
                                    int fmt, const int* seq_list, int n_list, int Ltile, void* stream);
int cldrd_attention_cls_fwd_varlen(const void* qc, const void* kv_packed, const int* cu_rows, void* ctx, float* probs, int nseq, int L, int H,
                                   float dropout_p, unsigned long long seed, int fmt, void* ctx_f16_copy, void* stream);
int cldrd_attention_cls_bwd_varlen(const void* qc, const void* kv_packed, const int* cu_rows, const float* probs, const void* dctx, void* dqc,
                                   void* dkv_packed, int nseq, int L, int H, float dropout_p, unsigned long long seed, int fmt, void* stream);

/* ---- embeddings + LayerNorm (HF Embeddings.forward, sa_layer_norm / output_layer_norm) --------------------
 * d <= 1024, d % 4 == 0.  `partial` scratch: cldrd_ln_partial_blocks(T) * 3 * d floats. */
int cldrd_ln_partial_blocks(int T);
int cldrd_embed_ln_fwd(const long long* ids, const float* word, const float* pos, const float* type0,
                       const float* gamma, const float* beta, void* out, float* mean, float* rstd, int T, int L,
                       int d, int vocab, float eps, float dropout_p, unsigned long long seed, float* out32, int out_f16,
                       const int* pos_idx, void* out_bf16_copy, void* stream);
int cldrd_embed_ln_bwd(const void* dy, const long long* ids, const float* word, const float* pos, const float* type0,
                       const float* gamma, const float* mean, const float* rstd, float* dword, float* dpos,
                       float* dtype0, float* dgamma, float* dbeta, float* partial, int T, int L, int d, int vocab,
                       float dropout_p, unsigned long long seed, int accumulate, const int* pos_idx, int dy_f32, const void* dy_branch,
                       void* stream);
/* out = LN(x)*gamma+beta (bf16); cls_out (fp32 [T/cls_stride, d], optional) receives rows r % cls_stride == 0:
 * the `[0][:, 0, :]` CLS pooling of models/nway_dual_encoder.py:52,56,64.
 * x_f32 != 0: x is fp32 (the pre-LN sum of the fp32 residual stream) and out32 (optional) receives the fp32 output next to
 * the bf16 copy the GEMMs read; cldrd_embed_ln_fwd has the same optional out32.
 * out_f16 != 0: `out` is fp16 (operand of an fp16 forward GEMM) and out_bf16_copy (optional) receives the same values in bf16: the
 * backward's MFMAs multiply the saved activations with bf16 gradients, so a training forward whose FFN GEMMs read fp16 keeps both. */
int cldrd_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* out, float* mean, float* rstd,
                        int T, int d, float eps, float* cls_out, int cls_stride, int x_f32, float* out32, int out_f16,
                        void* out_bf16_copy, void* stream);
/* dx = LN backward of dy; dx_dropped (optional) = dropout-masked dx for the branch that passed through dropout;
 * dgamma/dbeta/dbias (each optional) receive sum(dy*xhat), sum(dy), sum(dx_dropped or dx).
 * x_f32: bit 0 = x holds fp32 pre-LN sums; bit 1 = fp32 GRADIENT STREAM: dy is read and dx written as fp32 rows, dx_dropped (bf16, then
 * required and written without dropout too) is the MFMA operand of the next data-gradient GEMM, and dy_branch (bf16, optional) is added to dy
 * on load - the output of the data-gradient GEMM of the branch that joins the residual path here.  cldrd_embed_ln_bwd: dy_f32 / dy_branch
 * likewise.  bit 2 = dx_dropped / dy_branch are fp16, not bf16 (the all-fp16 training mode).  bit 3 = FP16 GRADIENT STREAM (round 5; implies
 * bit 2, needs bit 0, excludes bit 1): dy is read and dx written as fp16 rows carrying the loss scale; dx_dropped may be NULL when no dropout
 * separates the stream from the MFMA operand (dx then is that operand).  cldrd_embed_ln_bwd: dy_f32 bit 3 = dy is fp16.
 * cldrd_scatter_cls_grad(_idx) g_f32 / cldrd_add_rows_strided, cldrd_add_rows_idx f32: 0 = bf16 rows, 1 = fp32, 2 = fp16 (the add forms
 * then take an fp32 src: fp16 dst += fp32 src, one rounding). */
int cldrd_layernorm_bwd(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma,
                        void* dx, void* dx_dropped, float* dgamma, float* dbeta, float* dbias, float* partial, int T,
                        int d, float dropout_p, unsigned long long seed, int accumulate, int x_f32, const void* dy_branch, void* stream);
/* Deferred form: call cldrd_layernorm_bwd with dgamma = dbeta = dbias = NULL (its `partial` then keeps the per-block sums and must
 * stay untouched), and reduce the scratch buffers of n such calls in ONE launch afterwards.  T[i] = the T of call i; outputs as
 * above, bit-identical to the immediate form.  (The parameter gradients of a LayerNorm are not on the backward's critical path.) */
int cldrd_ln_reduce_group(const float* const* partial, const int* T, float* const* dgamma, float* const* dbeta,
                          float* const* dbias, int n, int d, int accumulate, void* stream);
/* out[N] (+)= column sums of bf16 x[T,N] (bias gradients).  partial: ceil(T/128) * N floats. */
int cldrd_colsum_bf16(const void* x, float* out, float* partial, int T, int N, int ld, int accumulate, void* stream);
/* g = zeros(bf16 [T,d]); g[r*stride] = dcls[r]  (gradient of the CLS pooling). */
int cldrd_scatter_cls_grad(const float* dcls, void* g, int R, int d, int stride, int T, int g_f32, void* stream);   /* g_f32: g holds fp32 rows */

/* ---- N-way scoring (models/nway_dual_encoder.py:30-47) ----------------------------------------------------
 * mode 0: logits[B,N]; 1: in-batch, all negatives [B,B*N]; 2: in-batch, next sample's N [B,2N]. fp32. */
int cldrd_score_fwd(const float* q, const float* p, float* logits, int B, int N, int d, int mode, void* stream);
int cldrd_score_bwd(const float* dlogits, const float* q, const float* p, float* dq, float* dp, int B, int N, int d,
                    int mode, void* stream);

/* ---- losses (losses/kl_div.py, margin_mse.py, ranknet.py, lambda_rank.py) ---------------------------------
 * kind 0 KLDiv(T) | 1 MarginMSE | 2 ranknet_loss | 3 lambda_mrr_loss (batch_weight != NULL: bweight_lambda_mrr_loss).
 * loss_out: float[2] = {loss, valid pair count}; grad [B,N] = dloss/dy_pred; workspace float[2*B]. */
int cldrd_loss_fwd_bwd(int kind, const float* y_pred, const float* y_true, const float* batch_weight, float* loss_out,
                       float* grad, float* workspace, int B, int N, float T, float pad_indicator, int mean_reduction,
                       void* stream);

/* Logit-norm regulariser `reg_loss = pred_logits.norm(2) * args.reg_lambda; loss += reg_loss`
 * (trainer/multistep-curriculum/nway_listwise_1.py:348-350): loss_out[0] += reg, grad[n] += d reg / d logits, *reg_out = reg
 * (reg_out may be null).  Same stream as, and after, cldrd_loss_fwd_bwd. */
int cldrd_logit_norm_reg(const float* logits, int n, float reg_lambda, float* loss_out, float* grad, float* reg_out, void* stream);

/* lambda_loss of losses/standard_lambda_rank.py:3-95 (allRank LambdaLoss framework): value + d loss / d y_pred in one call.
 * scheme: 0 None, 1 ndcgLoss1_scheme, 2 ndcgLoss2_scheme, 3 lambdaRank_scheme, 4 ndcgLoss2PP_scheme, 5 rankNet_scheme,
 * 6 rankNetWeightedByGTDiff_scheme, 7 rankNetWeightedByGTDiffPowed_scheme (:98-127); k <= 0: no truncation; gain_linear:
 * gain="linear" instead of "power"; log2_reduction: reduction_log="binary".  loss_out[2] = {loss, pairs}; workspace 2*B floats.
 * cldrd_loss_fwd_bwd kind 4 is weighted_pointwise_loss (losses/weighted_pointwise.py:3-14; y_true carries the weights). */
int cldrd_lambda_loss_fwd_bwd(const float* y_pred, const float* y_true, float* loss_out, float* grad, float* workspace, int B, int N,
                              int scheme, int k, float eps, float sigma, float mu, float pad_indicator, int mean_reduction,
                              int log2_reduction, int gain_linear, void* stream);

/* ---- optimizer step (trainer/multistep-curriculum/nway_listwise_1.py:353-367) ------------------------------
 * One flat fp32 buffer for all parameters.  clip out: float[3] = {grad L2 norm, clip coefficient, non-finite flag}.
 * decay_flags: one byte per 64 parameters: bit 0 = weight decay applies, bit 1 = leave the 16-bit shadows of this chunk unwritten
 * (embedding tables: read in fp32 by the embedding kernels). */
int cldrd_sqnorm_blocks(void);
int cldrd_grad_clip_coef(const float* g, size_t n, float max_norm, float* partial, float* out, void* stream);
/* The same norm taken in pieces (the trainer takes the part of it whose gradients are complete early on its second stream, under the
 * last weight-gradient launch): cldrd_sqnorm_partial writes exactly nblk partial sums of squares of g[0, n) to partial[0, nblk);
 * cldrd_clip_coef reduces nblk_total of them (fixed order, fp64) to out[3] as cldrd_grad_clip_coef does. */
int cldrd_sqnorm_partial(const float* g, size_t n, float* partial, int nblk, void* stream);
/* Clip-norm partial sums from the kernels that WRITE the gradients (round 5): while a sink is set (thread-local, like the loss scale),
 * cldrd_wgrad_group - when it reduces token-split slabs - and cldrd_ln_reduce_group also write one sum of squares per workgroup of the
 * values they wrote to slots[used ...]; cldrd_norm_sink_used() returns the number of slots written since cldrd_set_norm_sink, -1 when
 * a launch could not contribute (a weight-gradient group without slabs) or -2 when the sink was too small for a launch: the caller then
 * takes that range's norm with cldrd_sqnorm_partial as before.  cldrd_clip_coef reduces the slots like any others.  slots = NULL: off. */
void cldrd_set_norm_sink(float* slots, int capacity);
int cldrd_norm_sink_used(void);
int cldrd_clip_coef(const float* partial, int nblk_total, float max_norm, float* out, void* stream);
int cldrd_adamw_step(float* p, const float* g, float* m, float* v, const unsigned char* decay_flags, void* shadow,
                     size_t n, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                     const float* clip, void* stream);
/* The same step that also leaves an fp16 copy (RNE, as cldrd_cast_f16) of the updated parameters [h16_begin, h16_end) in shadow16
 * (shadow16[0] = parameter h16_begin; bounds multiples of 4): the query tower's high-precision forward reads fp16 weights. */
int cldrd_adamw_step_h16(float* p, const float* g, float* m, float* v, const unsigned char* decay_flags, void* shadow,
                         size_t n, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                         const float* clip, void* shadow16, size_t h16_begin, size_t h16_end, void* stream);
int cldrd_cast_bf16(const float* src, void* dst, size_t n, void* stream);
int cldrd_transpose_cast_batched(const float* src, void* dst, const long long* desc, const int* tile_prefix, int ndesc,
                                 int total_tiles, void* stream);
/* Same transposes from the bf16 shadow (rows and cols multiples of 64, offsets multiples of 8 elements; tile_prefix in 64x64 tiles). */
int cldrd_transpose_bf16_batched(const void* src, void* dst, const long long* desc, const int* tile_prefix, int ndesc, int total_tiles,
                                 void* stream);

/* ---- exact inner-product top-k over one index shard (retriever/retrieval_utils.py:131-153 -> faiss IndexFlatIP.search) ---
 * The shard lives in HBM twice: fp32 rows (exact scores) and a 16-bit shadow the scan streams (fp16 by default: 11-bit
 * significand, so |scan - exact| <= eps is 8x tighter than with bf16; `f16` selects the MFMA type of the scan entry points).
 *
 * cldrd_flatip_search is the whole search of nq device-resident queries in passes of qtile = 128 queries (the reference's
 * batching, retrieve_top_passages.py:88) or 256 (two reference batches share one pass over the index bytes), enqueued back to
 * back with no host round trip; per pass:
 *   scan:    16-bit MFMA scores Q[qtile,d] . P[rows,d]^T; (query, row) pairs with score >= thr[query] are appended to the query's
 *            candidate list (cand_rows / cand_scores: [qtile, cap]).  counts = qtile list lengths + 1 counter of hits the
 *            streaming kernel had to drop (its on-chip list overflowed within one tile), zeroed by the caller;
 *   select:  t^ = kk-th largest scan score of the list; rows with scan score >= t^ - 2 eps[query] are kept (rows2, n2) - no other
 *            row can be in the exact top-kk (proof in csrc/topk.hip); status[query] = 0 when that proof holds, else a bit mask
 *            (1 fewer than kk candidates, 2 list overflow, 4 dropped hits, 8 thr above t^ - 2 eps, 16 kept set > cap2);
 *   rescore: exact fp32 <q, P32[row]> of the kept rows (fixed summation order);
 *   sort:    (score desc, row asc), first k -> D[nq,k], I[nq,k] (row index, -1 / -inf = missing).
 * The caller reads `status` once per search and redoes unproven queries with thresholds of its choice (same entry point).
 * `exhaustive` is a bit mask: bit 0 (rows <= cap): no scan, every row is re-scored; bit 1: the scan runs through the tiled kernels
 * (cldrd_topk_scan_filter_tiled: no on-chip hit list, so status bit 4 cannot occur) - the retry form for passes that dropped hits.
 * Pieces, also callable one by one: prep (fp16 + bf16 copies and norms of the queries; *flag |= 1 if a value exceeds the fp16
 * range), kth (thr estimate: kth largest of sample scores), thresholds (eps[q] = bound on |scan - exact|, thr = est - 2 eps). */
int cldrd_cast_f16(const float* src, void* dst, size_t n, unsigned int* flag, void* stream);
int cldrd_topk_prep_queries(const float* q, void* q_f16, void* q_bf16, float* qnorm, int nq, int d, unsigned int* flag, void* stream);
int cldrd_topk_thresholds(const float* est, const float* qnorm, float pmax, int d, float* thr, float* eps, int nq, void* stream);
int cldrd_flatip_search(const float* q32, const void* q16, const float* thr, const float* eps, const void* P16, const float* P32,
                        long long rows, int d, int nq, int k, int qtile, int* counts, int* cand_rows, float* cand_scores, int cap,
                        int* rows2, float* scores2, int cap2, int* n2, int* status, float* khat, float* D, int* I,
                        int exhaustive, void* stream);
int cldrd_topk_scan_filter(const void* Q, const void* P, int nq, long long rows, int d, const float* thr, int* counts,
                           int* cand_rows, float* cand_scores, int cap, int f16, void* stream);
/* same contract through the tiled GEMM kernels (any d % 64 == 0; hits go straight to the global lists, nothing is dropped) */
int cldrd_topk_scan_filter_tiled(const void* Q, const void* P, int nq, long long rows, int d, const float* thr, int* counts,
                           int* cand_rows, float* cand_scores, int cap, int f16, void* stream);
int cldrd_topk_kth_largest(const float* scores, int ld, int nq, int S, int kth, float* thr, void* stream);
/* counts has nq + 1 entries (counts[nq] = dropped hits of the scan) */
int cldrd_topk_select(const int* counts, const int* cand_rows, const float* cand_scores, int nq, int cap, int kk, const float* thr,
                      const float* eps, int* rows2, int cap2, int* n2, int* status, float* khat, int exhaustive, void* stream);
int cldrd_topk_rescore(const float* q, const float* P, int d, const int* counts, const int* cand_rows, float* cand_scores,
                       int nq, int cap, void* stream);
int cldrd_topk_sort(const int* counts, const int* cand_rows, const float* cand_scores, int nq, int cap, int k, float* D,
                    int* I, void* stream);
int cldrd_row_sqnorm_max(const float* P, size_t rows, int d, unsigned int* out, void* stream);
int cldrd_gather_cast_rows(const float* src, void* dst, size_t n_out, size_t stride, int d, void* stream);
/* Attaching an index shard to a GPU (what faiss' index_cpu_to_gpu does behind retriever/retrieval_utils.py:155-162 - here: the scan shadow
 * and its error-bound statistics), three launches over the fp32 rows P[rows, d]:
 *   cldrd_index_col_mean    mu[d] = mean row (fp64 column sums, fixed order); workspace: cldrd_index_col_mean_workspace(rows, d) device bytes
 *   cldrd_index_center_cast P16[r] = fp16(P[r] - mu) (the scan's operand), sample[i] = bf16(P[i * s_stride] - mu) for i < s_rows (the threshold
 *                           sample; may be NULL), *cmax_bits = bit pattern of the fp32 value of max_r |P[r] - mu|^2 (fp64 row sums; zero it
 *                           first), *flag |= 1 when a centred value is outside the fp16 range
 *   cldrd_map_ids           out[i] = I[i] < 0 ? -1 : (ids ? ids[I[i]] : I[i] + id_offset): faiss IndexIDMap applied to a result list */
size_t cldrd_index_col_mean_workspace(size_t rows, int d);
int cldrd_index_col_mean(const float* P, size_t rows, int d, float* mu, void* workspace, size_t workspace_bytes, void* stream);
int cldrd_index_center_cast(const float* P, const float* mu, size_t rows, int d, void* P16, void* sample_bf16, size_t s_stride,
                            size_t s_rows, unsigned int* cmax_bits, unsigned int* flag, void* stream);
int cldrd_map_ids(const int* I, const long long* ids, long long id_offset, long long* out, size_t n, void* stream);

/* ---- variable-length packing (csrc/pack.hip) -----------------------------------------------------------------------------------
 * The reference pads every sequence of a batch to the longest one (dataset/sequence_dataset.py:50-51, nway_dataset.py:103-107) and the
 * encoder computes on the padding.  Packed layout: tokens of sequence m = rows cu[m] .. cu[m+1] of a [Tp, features] matrix (cu: device
 * int32 [nseq + 1]); Linear / LayerNorm / weight gradients run on the Tp real rows, attention on the same rows through the
 * cldrd_attention_*_varlen calls above (until round 6: on the padded [nseq * L, .] layout, rows moved by the two calls below).
 * cldrd_embed_ln_fwd / _bwd take pos_idx (device int32 [T], the position of every row inside its sequence; NULL: row % L).
 *   unpack_rows16: dst[m * L + j] = j < len[m] ? src[cu[m] + j] : 0  (16-bit rows of w elements);  gather_rows: dst[p] = src[idx[p]]
 *   (rows of row_bytes bytes: packing, CLS rows);  scatter_cls_grad_idx: g = 0, g[idx[r]] = bf16(dcls[r]);  add_rows_idx: dst[idx[m]] += src[m]. */
int cldrd_unpack_rows16(const void* src_packed, void* dst_padded, const int* cu, int nseq, int L, int w, void* stream);
int cldrd_gather_rows(const void* src, const int* idx, void* dst, int n, int row_bytes, void* stream);
int cldrd_gather_i64(const long long* src, const int* idx, long long* dst, int n, void* stream);      /* dst[p] = src[idx[p]]: token ids of the packed rows */
int cldrd_scatter_cls_grad_idx(const float* dcls, void* g, int R, int d, const int* idx, int T, int g_f32, void* stream);
int cldrd_add_rows_idx(void* dst, const void* src, int M, int d, const int* idx, int f32, void* stream);

/* ---- per-step state in device memory: what lets a whole training step be captured into a HIP graph and replayed -------------------
 * Kernel arguments are frozen at capture; a dropout seed and the optimizer's lr / bias-corrected step size change every step.  With a
 * seed base installed, every launch of the calling thread passes its `seed` argument as an OFFSET and the kernels add the 64-bit word at
 * `base` (device memory) at run time; with the optimizer hyper-parameters installed, cldrd_adamw_step* read {lr, step size} from the
 * device float[2].  cldrd_write_step_state writes both (seeds[0..1]: one word per tower) in stream order - the one launch the trainer
 * makes in front of each replay.  NULL uninstalls; nothing installed = the by-value arguments, as before (bit-identical results). */
void cldrd_set_seed_base(const unsigned long long* base);
void cldrd_set_optim_hyper(const float* hyper);
/* The all-fp16 training mode (round 4; the reference trains under fp16 autocast + torch.cuda.amp.GradScaler, trainer/multistep-curriculum/
 * nway_listwise_1.py:334-359): every 16-bit tensor of the backward is fp16 and carries a loss scale S, parameter gradients never do.
 * `scale` = device float[72] {S, 1 / S, finite steps since the headroom changed, skipped steps, headroom exponent h <= 0, 3 unused, 64 scratch},
 * or null (off).  While set, launches of THIS thread read it on the device at run time: cldrd_wgrad_group, cldrd_layernorm_bwd /
 * cldrd_embed_ln_bwd (their parameter-gradient sums and the embedding-table scatter) multiply by 1 / S; cldrd_clip_coef /
 * cldrd_grad_clip_coef run the safety net (non-finite gradient norm: AdamW skips the step and the next steps get 4x more headroom;
 * growth_interval finite steps in a row give a factor 2 back).  cldrd_loss_scale_adapt sets S = 2^(12 + h - ceil(log2 max(|a|, |b|)))
 * from the gradient that enters the towers (a, b = dL/dCLS of the two towers, fp32, from cldrd_score_bwd) and multiplies both by it in
 * place: a power of two recomputed from the data every step, where GradScaler searches by overflowing and backing off. */
void cldrd_set_loss_scale(const float* scale, int growth_interval);
int cldrd_loss_scale_adapt(float* a, size_t na, float* b, size_t nb, float* state, void* stream);
/* n <= 8 small device-to-device copies in one launch (host arrays of pointers / byte counts): the inputs of a captured step. */
int cldrd_copy_segments(const void* const* src, void* const* dst, const size_t* bytes, int n, void* stream);
/* n <= 8 device ranges (16-byte aligned, sizes multiples of 16) set to zero in one launch: what optimizer.zero_grad() is reduced to - only the
 * embedding tables' gradients are accumulated into (reference nway_listwise_1.py:365 zeroes every gradient). */
int cldrd_zero_segments(void* const* dst, const size_t* bytes, int n, void* stream);
int cldrd_write_step_state(unsigned long long* seeds, unsigned long long seed0, unsigned long long seed1, float* hyper, float lr,
                           float beta1, float beta2, int adam_step, const float* scale_state, void* stream);

/* ---- run file (retriever/retrieve_top_passages.py:98-105), HOST side: no GPU work -----------------------------------------
 * Writes `qid\tdocid\trank\tscore\n` for nq queries x k hits (rank 1..k) to `path`; the score text is Python's repr of the fp32
 * value widened to a double, which is what the reference's f-string prints.  All pointers are HOST pointers.  Formatted by nthreads
 * host threads (<= 0: one per hardware thread, at most 64).  Returns the number of lines or -1. */
long long cldrd_write_run_file(const char* path, const long long* qids, const long long* docids, const float* scores,
                               long long nq, int k, int nthreads);
int cldrd_py_float_repr(double x, char* out32);

/* ---- merge of per-shard top-k lists (the host-side merge of the 8-way sharded retrieve, BASELINE.json north_star; the reference meant
 * faiss' IndexShards for it: retriever/retrieval_utils.py:164-182, retrieve_top_passages.py:85-88) -------------------------------------
 * Shard r contributes, per query, scores fp32 [k_in] descending and global ids int64 [k_in] (-1 = missing, missing last).  Result: the
 * k_out best of the world * k_in candidates by (score desc; ties: shard asc, then list position asc = global row position asc, the tie
 * rule of the single-index search), padded with (-inf, -1).
 * cldrd_merge_topk: HOST pointers (shard_scores[r] / shard_ids[r] = that shard's [nq, k_in] block), nthreads host threads (<= 0: one
 *   per hardware thread, at most 64).  Unsorted lists are accepted (that query is partially sorted instead of merged).
 * cldrd_merge_topk_device: DEVICE pointers, scores / ids = [world, nq, k_in] (what a gather over RCCL leaves on rank 0), world * k_in
 *   <= 8192 and k_out <= world * k_in; one sort launch per call; `workspace` of cldrd_merge_topk_device_workspace() bytes. */
int cldrd_merge_topk(const float* const* shard_scores, const long long* const* shard_ids, int world, long long nq, int k_in, int k_out,
                     float* D, long long* I, int nthreads);
size_t cldrd_merge_topk_device_workspace(int world, long long nq, int k_in, int k_out);
int cldrd_merge_topk_device(const float* scores, const long long* ids, int world, long long nq, int k_in, int k_out, float* D,
                            long long* I, void* workspace, size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif
