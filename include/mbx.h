/* mbx.h -- C ABI of libmbx.so: the MI355X (gfx950) kernel set behind the DSTformer hot path.
 *
 * The reference (Walter0807/MotionBERT) has no FFI: its hot path is the Python class
 * lib/model/DSTformer.py:269-361 running on stock ATen ops.  This header is therefore the
 * boundary this project defines for that path; each entry names the reference arithmetic it
 * replaces (file:line in the reference checkout).  INTEGRATION.md shows the ctypes binding.
 *
 * Conventions
 *   - plain C types only; every pointer is DEVICE memory owned by the caller (PyTorch);
 *     the library never allocates, frees, caches or synchronises;
 *   - every call enqueues kernels on `stream` (a hipStream_t passed as void*) and returns;
 *   - return 0 on success, non-zero on argument/launch error; mbx_last_error() gives the
 *     text (thread-local).  No C++ exceptions cross the ABI;
 *   - re-entrant, no mutable global state (DataParallel replica threads, autograd thread);
 *   - `dtype` selects the operand type T of GEMM/attention tensors:
 *       MBX_BF16  bf16 operands, fp32 MFMA accumulation      (v_mfma_f32_32x32x16_bf16)
 *       MBX_F32   fp32 operands, exact fp32 MFMA              (v_mfma_f32_32x32x2_f32)
 *     residual stream, statistics and parameter gradients are always fp32;
 *   - token index m = (b*T + t)*J + j; tensors are dense row-major.
 */
#ifndef MBX_H
#define MBX_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum mbx_dtype {
    MBX_F32 = 0,
    MBX_BF16 = 1,
    MBX_BF16_LO = 2 /* mbx_prep_weights only: writes bf16(w - float(bf16(w))), the 'lo' plane of the bf16x3 split */
};

/* GEMM epilogues (fused into the store of the accumulator tile) */
enum mbx_epilogue {
    MBX_EPI_STORE = 0, /* out_t = acc + bias                          qkv (DSTformer.py:103,143) / dX GEMMs */
    MBX_EPI_GELU  = 1, /* out2_t = gelu_erf(acc + bias); out_t = acc + bias if non-NULL   fc1 + nn.GELU (:80-81) */
    MBX_EPI_RESID = 2, /* out_f = resid + acc + bias                  proj / fc2 + residual (:241-249)     */
    MBX_EPI_TANH  = 3, /* out_f = tanh(acc + bias)                    pre_logits fc + Tanh (:294-297,354)  */
    MBX_EPI_DGELU = 4, /* out_t = acc * gelu_erf'(aux_t)              backward of nn.GELU                  */
    MBX_EPI_LNBWD = 5, /* internal to mbx_gemm_nt_lnbwd (LayerNorm backward as a GEMM epilogue); not accepted by mbx_gemm_nt */
    /* 6, 7, 8: epilogues of rounds 3 / 4 that never had a caller in the default path; removed in round 5 */
    MBX_EPI_LNBWD_T = 9,  /* internal to mbx_gemm_nt_lnbwd_t (MBX_EPI_LNBWD with the gradient residual stream in bf16) */
    MBX_EPI_GELU_D = 10,  /* internal to mbx_gemm_nt_gelu_d: out2_t = gelu_erf(acc + bias), out_t = gelu_erf'(acc + bias) */
    MBX_EPI_MULAUX = 11   /* internal to mbx_gemm_nt_mul:    out_t = acc * aux_t */
};

enum mbx_attn_mode {
    MBX_ATTN_SPATIAL  = 0, /* softmax over the J joints of one frame   (DSTformer.py:178-186) */
    MBX_ATTN_TEMPORAL = 1  /* softmax over the T frames of one joint   (DSTformer.py:188-200) */
};

const char* mbx_last_error(void);
int mbx_version(void);

/* ---- weights ------------------------------------------------------------------------------
 * One launch converts every nn.Linear weight of the model: for descriptor i, src fp32 [N,K] ->
 * dst_n T [N,K] (may be NULL) and dst_t T [K,N] (transposed copy for the dX GEMMs, may be NULL).
 * `desc` is a device array of n_desc records {src, dst_n, dst_t, N, K} (5 x int64). */
int mbx_prep_weights(const int64_t* desc, int n_desc, int max_n, int max_k, int dtype, void* stream);

/* ---- embedding: joints_embed + pos_embed + temp_embed (DSTformer.py:330-337) --------------
 * x [B,T,J,Din] f32, w [C,Din], b [C], pos [J,C], temp [>=T,C]  ->  h [M,C] f32 */
int mbx_embed_fwd(const float* x, const float* w, const float* b, const float* pos, const float* temp,
                  float* h, int B, int T, int J, int Din, int C, void* stream);
/* dh [M,C] -> dw [C,Din], db [C], dpos [J,C], dtemp [T,C] (rows >= T untouched), dx [M,Din] or NULL.
 * ws: >= mbx_embed_bwd_ws(T,J,Din,C) bytes. */
size_t mbx_embed_bwd_ws(int T, int J, int Din, int C);
int mbx_embed_bwd(const float* dh, const float* x, const float* w, float* dw, float* db, float* dpos,
                  float* dtemp, float* dx, int B, int T, int J, int Din, int C, void* ws, void* stream);
/* mbx_embed_bwd with dh = dh_a + dh_b, both bf16 [B*T*J, C] (the input gradients of the two Blocks of level 0; see mbx_fuse_bwd_pair) */
int mbx_embed_bwd_pair(const void* dh_a, const void* dh_b, const float* x, const float* w, float* dw, float* db, float* dpos,
                       float* dtemp, float* dx, int B, int T, int J, int Din, int C, void* ws, void* stream);

/* ---- LayerNorm (nn.LayerNorm, biased variance, eps inside sqrt; DSTformer.py:221-222,230-231,292)
 * x [M,C] f32 -> y [M,C] T, mean [M], rstd [M].  gamma = beta = NULL: y = (x - mean) rstd (see "LayerNorm folded" below) */
int mbx_layernorm_fwd(const float* x, const float* gamma, const float* beta, float eps, void* y,
                      float* mean, float* rstd, int M, int C, int dtype, void* stream);
/* dx = LN'(dy) [+ dres] [+ extra]  (f32);  dx_t = T copy of dx (or NULL);  dgamma, dbeta [C].
 * ws: >= mbx_layernorm_bwd_ws(C) bytes. */
size_t mbx_layernorm_bwd_ws(int C);
int mbx_layernorm_bwd(const void* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                      const float* dres, const float* extra, float* dx, void* dx_t, float* dgamma, float* dbeta,
                      int M, int C, int dtype, void* ws, void* stream);
/* mbx_layernorm_bwd (T = f32) with the T copy of dx written as operand planes (dx_hi, dx_lo bf16 [M,C]): in the fp32-class mode the next
 * sub-layer's backward reads the gradient of the residual stream as fp32 (dx) and as a split-operand GEMM operand. */
int mbx_layernorm_bwd_planes(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma, const float* dres,
                             const float* extra, float* dx, void* dx_hi, void* dx_lo, float* dgamma, float* dbeta, int M, int C,
                             void* ws, void* stream);

/* ---- GEMMs on MFMA -------------------------------------------------------------------------
 * acc[M,N] = a[M,K] . w[N,K]^T  (nn.Linear, DSTformer.py:74,76,97,103,295), epilogue per mbx_epilogue.
 * K % 64 == 0 (bf16) / K % 32 == 0 (f32), N % 8 == 0. Unused pointers NULL. */
int mbx_gemm_nt(const void* a, const void* w, const float* bias, int epilogue, void* out_t, void* out2_t,
                float* out_f, const float* resid, const void* aux_t, int M, int N, int K, int dtype, void* stream);
/* dw[N,K] = dy[M,N]^T . a[M,K] (f32);  db[N] = column sums of dy (or NULL).  N % 32 == 0, K % 32 == 0.
 * ws: >= mbx_gemm_tn_ws(M,N,K) bytes. */
size_t mbx_gemm_tn_ws(int M, int N, int K);
int mbx_gemm_tn(const void* dy, const void* a, float* dw, float* db, int M, int N, int K, int dtype,
                void* ws, void* stream);

/* ---- LayerNorm folded into the Linear it feeds (bf16 path) ---------------------------------------------------------------
 * Every norm1 / norm2 of a Block feeds exactly one Linear (DSTformer.py:241-249 -> Attention.qkv :143 / MLP.fc1 :80):
 *     Linear(LayerNorm(x)) = xhat . (W diag(gamma))^T + (b + W beta) = xhat . W'^T + b',   xhat = (x - mean) rstd.
 * Forward: mbx_layernorm_fwd / mbx_fuse_ln_fwd with gamma = beta = NULL write xhat as the GEMM operand; the weights come from
 * mbx_fold_norm_weights.  Backward: with rsum[n] = sum_k W'[n,k] the two row means of nn.LayerNorm's backward are row dots of
 * dY with quantities the PRODUCER of dY holds,
 *     c1[m] = mean_k dxhat = (1/C) sum_n dY[m,n] rsum[n],    c2[m] = mean_k dxhat xhat = (1/C) sum_n dY[m,n] (Y[m,n] - b'[n]),
 * so mbx_attn_bwd_stats / mbx_gemm_nt_dgelu_stats emit them as partial sums, mbx_lnbwd_rowc finalises them and
 * mbx_gemm_nt_lnbwd applies  dx = dres [+ extra] + rstd (dY . W' - c1 - xhat c2)  as the epilogue of the dX GEMM: no stand-alone
 * LayerNorm-backward pass, no bf16 round trip of d(xn).  mbx_unfold_norm_grads turns the folded weight gradient into the
 * gradients of W, gamma and beta.  (The final `norm` -> pre_logits path, :350-354, keeps the plain kernels.)
 *
 * mbx_fold_norm_weights: `desc` = n_desc records of 10 x int64 {w f32 [N,K], bias f32 [N] or 0, gamma f32 [K], beta f32 [K],
 *   dst_n bf16 [N,K], dst_t bf16 [K,N] or 0, bias_f f32 [N], rsum f32 [N], N, K}; rsum sums the bf16-ROUNDED folded weights. */
int mbx_fold_norm_weights(const int64_t* desc, int n_desc, int max_n, int max_k, void* stream);
/* MBX_EPI_DGELU + part[N/64][M][2] f32 = per 64-column block and row { sum du rsum, sum du (aux_t - bias_f) } of the rounded
 * output du (rsum, bias_f enter rounded to bf16: packed-bf16 dot products).  bf16; N % 64 == 0, K % 64 == 0. */
int mbx_gemm_nt_dgelu_stats(const void* a, const void* w, void* out_t, const void* aux_t, const float* bias_f,
                            const float* rsum, float* part, int M, int N, int K, void* stream);
/* fc1 + nn.GELU with the DERIVATIVE saved for backward instead of the pre-activation (DSTformer.py:80-81; round 5):
 * out_g = gelu_erf(a . w^T + bias), out_d = gelu_erf'(a . w^T + bias) (from the fp32 accumulator), both bf16 [M,N]; the backward of the
 * activation is then mbx_gemm_nt_mul: out = (a . w^T) * aux.  bf16; N >= 256, N % 8 == 0, K % 64 == 0. */
int mbx_gemm_nt_gelu_d(const void* a, const void* w, const float* bias, void* out_d, void* out_g, int M, int N, int K, void* stream);
int mbx_gemm_nt_mul(const void* a, const void* w, const void* aux, void* out, int M, int N, int K, void* stream);
/* rowc[M][4] f32 = {rstd, rstd c1, rstd c2, 0} from part[nb][M][2] (block-major; nb = 2 x heads, or N/64 column blocks) */
int mbx_lnbwd_rowc(const float* part, int nb, const float* rstd, float* rowc, int M, int C, void* stream);
/* dx[M,N] f32 = dres [+ extra] + rowc.x (a . w^T) - rowc.y - xhat rowc.z;  dx_t = bf16 copy of dx or NULL.
 * a bf16 [M,K] (dY), w bf16 [N,K] (the transposed folded weight), xhat bf16 [M,N].  N % 8 == 0, K % 64 == 0. */
int mbx_gemm_nt_lnbwd(const void* a, const void* w, const void* xhat, const float* rowc, const float* dres,
                      const float* extra, float* dx, void* dx_t, int M, int N, int K, void* stream);
/* The same with the gradient of the residual stream carried as bf16 BETWEEN the four sub-layers of a Block (fp32 at the Block
 * boundaries, fp32 accumulation and arithmetic in the kernel): dres_t bf16 [M,N]; dx f32 or NULL; dx_t bf16 or NULL (at least one
 * of them).  Inside a Block dx_t alone is written -- it is the stream and the next dX / dW GEMMs' operand at once: 4 instead of 12
 * bytes per element and launch.  Numerics: tools/gradstream_numerics.py (every gate of the bf16 path unchanged). */
int mbx_gemm_nt_lnbwd_t(const void* a, const void* w, const void* xhat, const float* rowc, const void* dres_t,
                        const float* extra, float* dx, void* dx_t, int M, int N, int K, void* stream);
/* in place: dw[N,K] (= dY^T xhat, the folded weight gradient) <- gamma[k] dw[n,k] + db[n] beta[k];
 * dgamma[k] = sum_n w[n,k] dw'[n,k];  dbeta[k] = sum_n w[n,k] db[n].  ws: >= mbx_unfold_norm_grads_ws(N,K) bytes. */
size_t mbx_unfold_norm_grads_ws(int N, int K);
int mbx_unfold_norm_grads(float* dw, const float* db, const float* w, const float* gamma, const float* beta,
                          float* dgamma, float* dbeta, int N, int K, void* ws, void* stream);

/* ---- LayerNorm as a raw operand + the fused MLP forward (bf16, no-grad / inference path) -------------------------------------
 * One step beyond the folding above.  With W' = W diag(gamma), b' = b + W beta, rsum[n] = sum_k W'[n,k] (mbx_fold_norm_weights):
 *     Linear(LayerNorm(y)) = rstd (y . W'^T - mean rsum) + b',
 * so the CONSUMER of a residual-stream tensor y (the qkv / fc1 Linear behind norm1 / norm2, DSTformer.py:241-249) multiplies the
 * RAW rows and applies the row constants where its accumulators are: the stand-alone LayerNorm passes of a forward (read fp32 y,
 * write bf16 xhat) disappear.  The consumers read the fp32 rows of y themselves (mbx_rows_gemm_nk_ln, mbx_mlp_fused_fwd with
 * a = NULL, mbx_proj_mlp_fused_fwd) and round the row SHIFTED by its first element, so that the rounding error scales with the
 * spread of the row, not with its magnitude (LayerNorm does not see the shift).
 *
 * mbx_mlp_fused_fwd:   the whole MLP sub-layer of a Block (MLP.forward, DSTformer.py:79-85, inside :242 / :244 / :246 / :248)
 *     y = resid + fc2(gelu_erf(fc1(LN(.)))) in ONE kernel -- the [M, hidden] tensor never exists in HBM:
 *       a        bf16 [M,C]: raw_in = 0: the normalised operand xhat (mbx_layernorm_fwd / mbx_fuse_ln_fwd with gamma = NULL);
 *                            raw_in = 1: bf16(x) itself; mean / rstd of the row are taken in the kernel from these values;
 *                            a = NULL (raw_in = 1): the operand is bf16(resid), rounded in the kernel from the fp32 rows it loads for
 *                            the residual anyway (no second input stream), statistics of the fp32 rows
 *       packed   the fc1 (folded) and fc2 weights in MFMA-fragment order: mbx_mlp_pack_weights(w1 bf16 [hidden,C], w2 bf16 [C,hidden])
 *                -> mbx_mlp_pack_bytes(C, hidden) bytes
 *       b1 [hidden] (folded bias b'), b2 [C], rsum [hidden] (raw_in only), resid f32 [M,C] (= x; may alias y)
 *       y f32 [M,C];  y_t bf16 [M,C] = bf16(y) or NULL;  mean, rstd f32 [M] = LayerNorm statistics of the rows of y (eps) or NULL
 *     C in {256, 512}, hidden % 64 == 0, hidden <= 1536. */
size_t mbx_mlp_pack_bytes(int C, int hidden);
int mbx_mlp_pack_weights(const void* w1, const void* w2, void* packed, int C, int hidden, void* stream);
int mbx_mlp_fused_fwd(const void* a, int raw_in, const void* packed, const float* b1, const float* b2, const float* rsum,
                      const float* resid, float* y, void* y_t, float eps, float* mean, float* rstd, int M, int C, int hidden,
                      void* stream);
/* The attention's proj + residual in front of the MLP, same kernel (DSTformer.py:241-249: x + attn(..) then x + mlp(norm(..)); :103 proj):
 *     y1 = resid + o . Wp^T + bp        y = y1 + fc2(gelu_erf(fc1(LN(y1))))
 * o bf16 [M,C] = the attention output (mbx_attn_fwd); y1 exists only in the accumulator registers: its bf16 rounding is fc1's operand,
 * its LayerNorm statistics are taken from the same registers.  packed: mbx_proj_mlp_pack_weights(wp bf16 [C,C], w1 (folded), w2) ->
 * mbx_proj_mlp_pack_bytes(C, hidden) bytes;  bp [C], b1 [hidden] (folded), b2 [C], rsum [hidden];  y may alias resid. */
size_t mbx_proj_mlp_pack_bytes(int C, int hidden);
int mbx_proj_mlp_pack_weights(const void* wp, const void* w1, const void* w2, void* packed, int C, int hidden, void* stream);
int mbx_proj_mlp_fused_fwd(const void* o, const void* packed, const float* bp, const float* b1, const float* b2, const float* rsum,
                           const float* resid, float* y, float eps, int M, int C, int hidden, void* stream);

/* ---- "row owner" NT GEMMs (bf16; csrc/gemm_rows.hip) -----------------------------------------------
 * The same Linear layers (DSTformer.py:97 qkv, :69 fc1; the input gradients of :70 fc2 and :103 proj) on a kernel whose workgroup
 * owns 128 complete token rows: the token operand lives in registers, the weights stream as pre-packed MFMA fragments.
 * mbx_rows_pack_nk: w bf16 [N,K] row-major -> packed (mbx_rows_pack_bytes(N, K) bytes); K in {256, 512}, N % 64 == 0.
 * mbx_rows_gemm_nk: out bf16 [M,N] = a . w^T + bias (bias may be NULL);  with mean != NULL the raw-operand LayerNorm form of
*                   Linear: out = rstd[m] (a . w^T - mean[m] rsum[n]) + bias[n].
 * mbx_rows_gemm_nk_ln: the same from the fp32 rows x [M,K] of the residual stream themselves: operand bf16(x) rounded in the kernel,
 *                   (mean, rstd) of every row taken from the same loads (eps as in nn.LayerNorm): Linear(LayerNorm(x)) without a
 *                   LayerNorm pass and without a bf16 copy of x (norm1 + attn.qkv of a Block, DSTformer.py:241-249 / :139-143). */
size_t mbx_rows_pack_bytes(int N, int K);
int mbx_rows_pack_nk(const void* w, void* packed, int N, int K, void* stream);
int mbx_rows_gemm_nk(const void* a, const void* packed, const float* bias, const float* rsum, const float* mean, const float* rstd,
                     void* out, int M, int N, int K, void* stream);
int mbx_rows_gemm_nk_ln(const float* x, const void* packed, const float* bias, const float* rsum, float eps, void* out, int M, int N,
                        int K, void* stream);

/* g = gelu_erf(u), T-typed, n % 4 == 0: rebuilds the MLP's post-activation from the saved pre-activation in the engine's
 * low-memory (recompute) mode (nn.GELU, DSTformer.py:70,80-81). */
int mbx_gelu_fwd(const void* u, void* g, size_t n, int dtype, void* stream);

/* ---- fp32-class split-operand GEMMs (precision 'bf16x3') ------------------------------------------
 * The north-star gate (outputs within 1e-3 of the fp32 reference, BASELINE.json) cannot be met with bf16 operands and
 * gfx950 has no TF32: every fp32 operand x is split into two bf16 planes, hi = bf16(x), lo = bf16(x - hi), and
 * a.w^T = a_hi.w_hi^T + a_hi.w_lo^T + a_lo.w_hi^T runs as three bf16 MFMA passes into ONE fp32 accumulator (products
 * of bf16 are exact in fp32; the dropped lo.lo term is 2^-16 relative).  Same arithmetic sites as mbx_gemm_nt / mbx_gemm_tn;
 * every T-typed tensor (out_t, out2_t, aux_t) is fp32.  K % 64 == 0, N % 8 == 0. */
int mbx_split_bf16(const float* x, void* hi, void* lo, size_t n, void* stream);   /* n % 4 == 0 */
int mbx_gemm_nt_x3(const void* a_hi, const void* a_lo, const void* w_hi, const void* w_lo, const float* bias, int epilogue,
                   float* out_t, float* out2_t, float* out_f, const float* resid, const float* aux_t, int M, int N, int K,
                   void* stream);
/* The same with the LAST output of the epilogue written as the two bf16 planes of the operand split (hi = bf16(v), lo = bf16(v - hi),
 * the arithmetic of mbx_split_bf16) instead of fp32 -- for outputs whose only reader is another split-operand GEMM (the MLP's
 * activation, DSTformer.py:80-83, and its gradient): MBX_EPI_STORE (planes = out), MBX_EPI_GELU (out_t = pre-activation f32 or NULL,
 * planes = gelu), MBX_EPI_DGELU (planes = acc * gelu'(aux_t)). */
int mbx_gemm_nt_x3p(const void* a_hi, const void* a_lo, const void* w_hi, const void* w_lo, const float* bias, int epilogue,
                    float* out_t, void* pl_hi, void* pl_lo, const float* aux_t, int M, int N, int K, void* stream);
/* mbx_layernorm_fwd (T = f32) with y written as the operand planes (y_hi, y_lo bf16 [M,C]) of the Linear that follows. */
int mbx_layernorm_fwd_planes(const float* x, const float* gamma, const float* beta, float eps, void* y_hi, void* y_lo, float* mean,
                             float* rstd, int M, int C, void* stream);
size_t mbx_gemm_tn_x3_workspace(int M, int N, int K);
int mbx_gemm_tn_x3(const void* dy_hi, const void* dy_lo, const void* a_hi, const void* a_lo, float* dw, float* db, int M,
                   int N, int K, void* ws, void* stream);

/* ---- attention core (DSTformer.py:178-200), qkv [M,3C] T with channel order [3][H][hd] --------
 * o [M,C] T (heads concatenated), lse [M,H] f32 = log-sum-exp of the scaled scores.
 * hd = C/H must be 32 or 64; J <= 32; T <= 256. */
int mbx_attn_fwd(const void* qkv, void* o, float* lse, int B, int T, int J, int H, int hd, float scale,
                 int mode, int dtype, void* stream);
/* mbx_attn_fwd with nn.Dropout(p) on the probabilities (attn_drop, DSTformer.py:96,182,196): o = (mask P / (1 - p)) V with the
 * softmax statistics (and lse) of the undropped probabilities.  The mask is counter-based: element (query i, key j) of a
 * problem survives iff keep(flat index of that element in the reference's attn tensor -- [B T, H, J, J] spatial,
 * [B, H, J, T, T] temporal --, p, seed), motionbert_amd/dropmask.py.  0 <= p < 1; p = 0 is mbx_attn_fwd. */
int mbx_attn_fwd_drop(const void* qkv, void* o, float* lse, int B, int T, int J, int H, int hd, float scale,
                      int mode, int dtype, float p, uint64_t seed, void* stream);
/* dqkv [M,3C] T from do [M,C] T; probabilities are recomputed from q, k and lse. */
int mbx_attn_bwd(const void* qkv, const void* o, const void* d_o, const float* lse, void* dqkv, int B, int T,
                 int J, int H, int hd, float scale, int mode, int dtype, void* stream);
/* backward of mbx_attn_fwd_drop (same p and seed; o is the dropped forward output): dP = mask (dO V^T) / (1 - p), dV from the
 * dropped probabilities, dS = P (dP - rowsum(dO o)). */
int mbx_attn_bwd_drop(const void* qkv, const void* o, const void* d_o, const float* lse, void* dqkv, int B, int T,
                      int J, int H, int hd, float scale, int mode, int dtype, float p, uint64_t seed, void* stream);
/* mbx_attn_bwd / mbx_attn_bwd_drop of the fp32 kernels (precision 'bf16x3') with dq | dk | dv written as the two bf16 planes of the
 * operand split (dqkv_hi, dqkv_lo [M,3C]) instead of fp32: their only readers are the split-operand GEMMs of qkv's backward.  p = 0:
 * no dropout. */
int mbx_attn_bwd_planes(const void* qkv, const void* o, const void* d_o, const float* lse, void* dqkv_hi, void* dqkv_lo, int B, int T,
                        int J, int H, int hd, float scale, int mode, float p, uint64_t seed, void* stream);
/* mbx_attn_bwd (bf16) + part[2H][M][2] f32 = per (head, role, token) { sum dqkv rsum, sum dqkv (qkv - bias_f) } of the rounded
 * output over the head's q columns (role 0) and over its k and v columns (role 1) (LayerNorm folding, above; nb = 2H for
 * mbx_lnbwd_rowc; rsum, bias_f f32 [3C], entering rounded to bf16).  part must be 8-byte aligned. */
int mbx_attn_bwd_stats(const void* qkv, const void* o, const void* d_o, const float* lse, void* dqkv, const float* bias_f,
                       const float* rsum, float* part, int B, int T, int J, int H, int hd, float scale, int mode, void* stream);

/* ---- adaptive fusion of the two streams (DSTformer.py:343-349) -------------------------------
 * alpha = softmax(w . cat[x_st, x_ts] + b)  [M,2];  out = x_st*alpha0 + x_ts*alpha1.  w [2,2C], b [2]. */
int mbx_fuse_fwd(const float* x_st, const float* x_ts, const float* w, const float* b, float* out, float* alpha,
                 int M, int C, void* stream);
/* mbx_fuse_fwd plus the LayerNorm(s) that read its output (next level's Block.norm1_s / norm1_t, DSTformer.py:241,247, or the
 * final norm, :350): xn1 = LN(out; g1, b1), optionally xn2 = LN(out; g2, b2) (g2 = b2 = xn2 = NULL: one consumer), both T-typed,
 * sharing mean / rstd [M].  Same arithmetic as mbx_fuse_fwd followed by mbx_layernorm_fwd. */
int mbx_fuse_ln_fwd(const float* x_st, const float* x_ts, const float* w, const float* b, float* out, float* alpha,
                    const float* g1, const float* b1, void* xn1, const float* g2, const float* b2, void* xn2, float eps,
                    float* mean, float* rstd, int M, int C, int dtype, void* stream);
/* backward: d_st / d_ts f32 [M,C] and their T-typed copies d_st_t / d_ts_t; d_st = d_ts = NULL writes the T-typed copies only
 * (the gradient stream in the operand type, see mbx_gemm_nt_lnbwd_t) */
size_t mbx_fuse_bwd_ws(int C);
int mbx_fuse_bwd(const float* dh, const float* x_st, const float* x_ts, const float* alpha, const float* w,
                 float* d_st, float* d_ts, void* d_st_t, void* d_ts_t, float* dw, float* db, int M, int C,
                 int dtype, void* ws, void* stream);
/* mbx_fuse_bwd with dh = dh_a + dh_b, both bf16 [M,C]: the input gradients of the two Blocks of the level above, as their row-owner
 * LayerNorm-backward kernels (mbx_rows_lnbwd_t) leave them (round 5: the gradient of the residual stream in the operand type across
 * Block boundaries too).  bf16 outputs only (d_st_t, d_ts_t); ws as mbx_fuse_bwd. */
int mbx_fuse_bwd_pair(const void* dh_a, const void* dh_b, const float* x_st, const float* x_ts, const float* alpha, const float* w,
                      void* d_st_t, void* d_ts_t, float* dw, float* db, int M, int C, void* ws, void* stream);
/* att_fuse=False variant (DSTformer.py:351): out = (x_st + x_ts)/2 and its backward */
int mbx_average(const float* x_st, const float* x_ts, float* out, size_t n, void* stream);
int mbx_average_bwd(const float* dh, float* d_st, float* d_ts, void* d_st_t, void* d_ts_t, size_t n, int dtype,
                    void* stream);

/* ---- tail: head Linear(R -> Dout<=8) (DSTformer.py:300,357) and tanh' ---------------------------- */
int mbx_head_fwd(const float* rep, const float* w, const float* b, float* out, int M, int R, int Dout, void* stream);
size_t mbx_head_bwd_ws(int R, int Dout);
/* dpre_t = (dout . w) * (1 - rep^2) (T);  dw [Dout,R];  db [Dout] */
int mbx_head_bwd(const float* dout, const float* rep, const float* w, void* dpre_t, float* dw, float* db,
                 int M, int R, int Dout, int dtype, void* ws, void* stream);
int mbx_tanh_bwd(const float* drep, const float* rep, void* dpre_t, size_t n, int dtype, void* stream);

/* ---- the training step around the backbone (train.py:174-206,289) -----------------------------
 * Fused 3D pose loss of train.py:176-189 with the lambdas of configs/pose3d/MB_train_h36m.yaml:36-43 that are non-zero:
 *   total = loss_mpjpe + lambda_scale * n_mpjpe + lambda_velocity * loss_velocity        (lib/model/loss.py:56-62,81-91,133-142)
 * pred, gt [B,T,J,3] f32.  losses[4] = {mpjpe, n_mpjpe, velocity, total} stay on the device (the reference synchronises with
 * eight .item() calls per step).  dpred (or NULL) = grad_scale * d total / d pred, the cotangent to hand to backward.
 * ws: >= mbx_pose_loss_ws(B,T) bytes. */
size_t mbx_pose_loss_ws(int B, int T);
int mbx_pose_loss(const float* pred, const float* gt, float lambda_scale, float lambda_velocity, float* losses, float* dpred,
                  float grad_scale, int B, int T, int J, void* ws, void* stream);
/* 2D re-projection loss of the pre-training's 2D branch (lib/model/loss.py:72-77 loss_2d_weighted, train.py:200-203):
 * loss[1] = mean |(pred_xy - target_xy) * conf| and, if dpred != NULL, dpred [B,T,J,3] = grad_scale * d loss / d pred (z row 0).
 * pred [B,T,J,3] f32; target: x, y at target[tok * target_stride + {0,1}]; conf at conf[tok * conf_stride] -- with strides 3 / 3
 * the [B,T,J,3] 2D batch itself is target and (channel 2) confidence.  ws: >= mbx_loss_2d_weighted_ws(B,T) bytes. */
size_t mbx_loss_2d_weighted_ws(int B, int T);
int mbx_loss_2d_weighted(const float* pred, const float* target, int target_stride, const float* conf, int conf_stride, float* loss,
                         float* dpred, float grad_scale, int B, int T, int J, void* ws, void* stream);
/* AdamW (torch.optim.AdamW semantics, train.py:289) over ONE flat fp32 buffer -- or one contiguous RANGE of it -- of n
 * parameters, one launch.  p, g, m, v are parallel buffers: 4-byte aligned, same offset within a 16-byte line (a range that
 * skips frozen parameters, learning.py:69-77 / train.py:284-289, need not start on a 16-byte boundary).
 * state[2] on the device = {step count, learning rate}; tick != 0 advances the step count first (once per optimizer step). */
int mbx_adamw_step(float* p, const float* g, float* m, float* v, size_t n, float* state, float beta1, float beta2, float eps,
                   float weight_decay, int tick, void* stream);

/* ---- ActionNet pooling on the representation (lib/model/model_action.py:15-24,62-70) ---------------
 * rep [N*Mp*T*J, R] f32 (token order ((n Mp + m) T + t) J + j) -> pooled [N, J, R] = mean over persons and frames of the
 * element-wise dropped-out representation (p = dropout_ratio, 0 in evaluation; the keep mask is a counter-based hash of
 * (seed, element index), reproduced in backward, never stored). */
int mbx_pool_rep_fwd(const float* rep, float* pooled, int N, int Mp, int T, int J, int R, float p, uint64_t seed, void* stream);
/* dpre_t [tokens, R] T = broadcast(dpooled) / (Mp T) * keep / (1-p) * (1 - rep^2): the tail's tanh' (DSTformer.py:296) fused
 * with the backward of both means and of the dropout; the [N,Mp,T,J,R] cotangent is never materialised. */
int mbx_tanh_pool_bwd(const float* dpooled, const float* rep, void* dpre_t, int N, int Mp, int T, int J, int R, float p,
                      uint64_t seed, int dtype, void* stream);

/* ---- Dropout / DropPath with p > 0 in training (DSTformer.py:77,96,104,278; lib/model/drop.py:17-32) ------
 * Counter-based masks keep(seed, flat element index), recomputed in backward from the seed.
 * mbx_dropout: y = x * keep/(1-p) (in place allowed); T-typed by `dtype`; also the backward of itself. */
int mbx_dropout(const void* x, void* y, size_t n, float p, uint64_t seed, int dtype, void* stream);
/* y [rows,C] f32 holds x + branch (fused RESID epilogue): y <- x + branch * keep_e/(1-p) * keep_path/(1-p_path); DropPath draws
 * one value per `rows_per_sample` consecutive rows (drop.py:27: per leading index of the [B*T, J, C] tensor -> J). */
int mbx_residual_drop(float* y, const float* x, size_t rows, int C, int rows_per_sample, float p, uint64_t seed, float p_path,
                      uint64_t seed_path, void* stream);
/* dy_t [rows,C] T = dy * the same two masks: the gradient entering that branch. */
int mbx_grad_drop(const float* dy, void* dy_t, size_t rows, int C, int rows_per_sample, float p, uint64_t seed, float p_path,
                  uint64_t seed_path, int dtype, void* stream);

/* ---- the input stage on the device ---------------------------------------------------------------
 * Augmenter2D.add_noise (flags bit 0) and add_mask (bit 1) of lib/data/augmentation.py:29-81 in one kernel: x [B,T,J,Cin]
 * (Cin 2 or 3; only x, y are read when noise is on, as in the reference) -> y [B,T,J,3].  noise_mean / noise_std [J,2] and
 * noise_weight [J] are params/synthetic_noise.pth, d2c_* params/d2c_params.pkl, uniform_range 0.06, jitter_std 0.002 (:20,35),
 * mask ratios MB_pretrain.yaml:49-50.  Counter-based random numbers from `seed` (oracle/augment_oracle.py restates every draw). */
int mbx_augment2d(const float* x, float* y, int B, int T, int J, int Cin, const float* noise_mean, const float* noise_std,
                  const float* noise_weight, float uniform_range, float jitter_std, float d2c_a, float d2c_b, float d2c_m,
                  float d2c_s, float mask_ratio, float mask_T_ratio, int flags, uint64_t seed, void* stream);
/* flip test-time augmentation (train.py:67-72; lib/utils/utils_data.py:54-66): embedding of 2B samples from x [B,T,J,Din], sample
 * B+b = flip_data(x[b]) as an index remap (perm[j] = the joint whose values joint j takes; channel 0 negated); and the
 * flip-back + average of the [2B,T,J,D] output -> out [B,T,J,D]. */
int mbx_embed_fwd_tta(const float* x, const int* perm, const float* w, const float* b, const float* pos, const float* temp,
                      float* h, int B, int T, int J, int Din, int C, void* stream);
int mbx_flip_average(const float* out2, const int* perm, float* out, int B, int T, int J, int D, void* stream);

/* ---- "N-resident" row-owner GEMM with the LayerNorm backward as its epilogue (bf16; csrc/gemm_rows_n.hip; round 5) ---------------------
 * The input gradient of a folded (LayerNorm -> Linear) pair INSIDE a Block (DSTformer.py:241-249: norm1 -> attn.qkv :143, norm2 ->
 * mlp.fc1 :80; backward by autograd in the reference) in one launch, without row dots from the producers of dY:
 *     dxhat = dy . w^T   (dy bf16 [M,K], w bf16 [512,K] = mbx_fold_norm_weights' transposed folded weight, packed by mbx_rows_n_pack_many)
 *     dx_t  = bf16( dres_t + rstd (dxhat - mean_k dxhat - xhat mean_k(dxhat xhat)) )          all [M,512]
 * A workgroup owns 128 complete rows (256 accumulator registers per wave), so both row means come from the accumulators; xhat bf16
 * [M,512] and rstd f32 [M] are what mbx_layernorm_fwd (gamma = NULL) left, dres_t / dx_t the gradient of the residual stream in the
 * operand type (as mbx_gemm_nt_lnbwd_t with dx = NULL).  N = 512 (K >= 512) or, round 6, N = 256 (dim_feat of MotionBERT-Lite; 128
 * accumulator registers per wave, K >= 256), K % 256 == 0; M * N * 2 < 2^32 and M * K * 2 < 2^32 (32-bit lane offsets into dx_t / dy; checked).  dx_t must not alias
 * an input (checked for dres_t, xhat and dy).  Round 6: xhat is fetched by LDS-DMA during the last trip of the loop, in the 26 slots in
 * which the weight stream has nothing left to request (csrc/gemm_rows_n.hip). */
size_t mbx_rows_n_pack_bytes(int N, int K);
/* packs n_desc operands in ONE launch (a training step re-packs 60 weights): record r of desc = {w bf16 [N,K_r], packed_r
 * (mbx_rows_n_pack_bytes(N, K_r) bytes), K_r} (3 x int64, device memory); N = 512 or 256 for all of them; max_k = the largest K_r; every
 * K_r % 256 == 0 and >= 512 (N = 512) / >= 256 (N = 256) (the caller's responsibility: the records live on the device). */
int mbx_rows_n_pack_many(const int64_t* desc, int n_desc, int N, int max_k, void* stream);
int mbx_rows_lnbwd_t(const void* dy, const void* packed, const void* xhat, const float* rstd, const void* dres_t, void* dx_t, int M,
                     int N, int K, void* stream);
/* The FORWARD residual GEMM of a sub-layer that is followed by a LayerNorm, on the same row-owner shape (proj / fc2 + residual,
 * DSTformer.py:241-249, and the next norm1 / norm2): y = resid + a . w^T + bias (fp32 [M,512]); xhat = (y - mean(y)) rstd(y) (bf16 [M,512],
 * the plain normalisation: gamma and beta live in the folded weights of the Linear the LayerNorm feeds), mean / rstd fp32 [M] (two-pass
 * statistics of the fp32 rows, eps inside the square root).  packed = mbx_rows_n_pack_many's image of w [N,K].  bf16; N = 512 (K >= 512) or
 * 256 (K >= 256), K % 256 == 0, M * N * 4 < 2^32, M * K * 2 < 2^32 (checked).  y must not alias resid and xhat must not alias a (checked): rows past M are computed from row M - 1's
 * inputs and stored onto row M - 1 again, which is only harmless while those inputs are still the original ones. */
int mbx_rows_resid_ln(const void* a, const void* packed, const float* bias, const float* resid, float* y, void* xhat, float* mean,
                      float* rstd, float eps, int M, int N, int K, void* stream);

/* ---- measurement aid (bench.py `roofline.sustained_mfma_tflops`; not part of the model) --------------------------------------------
 * The bf16 MFMA rate the part sustains under its power cap with nothing but v_mfma_f32_32x32x16_bf16 in the loop (pseudo-random
 * operands; n_wg workgroups of 4 waves, `iters` x 16 MFMAs per wave).  ws: >= mbx_mfma_probe_ws(n_wg) bytes = a float sink
 * [n_wg * 256] followed by int64 [n_wg][2] = {shader cycles, 100 MHz real-time ticks} of every workgroup's loop (effective clock).
 * *flops = the matrix FLOPs of the launch; the caller times it with HIP events. */
size_t mbx_mfma_probe_ws(int n_wg);
int mbx_mfma_probe(void* ws, int n_wg, int iters, unsigned seed, double* flops, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MBX_H */
