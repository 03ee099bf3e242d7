/* C ABI of libctclip_hip.so -- hand-written gfx950 (MI355X) kernels for the CT-CLIP training step.
 *
 * The reference (injardav/CT-CLIP-UT) has no native/FFI layer: its hot path is a sequence of
 * PyTorch ops.  Each entry point below replaces one such op sequence; the reference lines are
 * cited per function (paths relative to the reference repo).  INTEGRATION.md shows the ctypes
 * binding a maintainer adds on the reference side.
 *
 * Conventions
 *   - every function returns a hipError_t as int (0 = success) and only enqueues work on `stream`
 *     (a hipStream_t passed as void*); nothing allocates or synchronises.  The only process state is
 *     idempotent: the first launch of a kernel that needs more than 64 KiB of LDS raises that kernel's
 *     dynamic-LDS limit once (hipFuncSetAttribute), and the CU count is read once.
 *   - environment: the library reads exactly two variables, both test hooks -- CTCLIP_GEMM_V2_ALL (lower
 *     the size gates of the pipelined GEMM kernels so that small test shapes reach every kernel) and
 *     CTCLIP_ATTN_SP_CHUNK (sequences per workgroup of the wave-per-sequence attention kernels, for
 *     ragged-chunk tests).  Results do not depend on either.  Development A/B switches exist only in
 *     builds compiled with -DCTCLIP_TUNING_KNOBS.
 *   - all pointers are DEVICE pointers owned by the caller; "bf16" buffers are raw uint16 storage.
 *   - matrices are row-major; `ld*` are row strides in ELEMENTS.  bf16 matrices need 16-byte aligned
 *     base pointers and strides that are multiples of 8.
 *   - gradient outputs named d<param> are ACCUMULATED into what the caller passes (the caller zeroes them).
 *   - reproducibility (the reference's attribution code asks for torch.use_deterministic_algorithms(True),
 *     src/utils/visualizations.py:29-39).  Every entry point with a `partials` argument reduces in two stages -- a row of
 *     partial sums per workgroup, stored with plain stores into `partials` (caller-provided scratch of at least
 *     CTCLIP_PARTIALS_FLOATS floats, private to the stream while the call runs), then added up in row order -- so its outputs
 *     are bit-identical from run to run: LayerNorm / head-norm / PEG parameter gradients, bias-gradient column sums, the
 *     squared gradient norm, d(temperature).  All forward kernels and all DATA gradients (dx, dq, dk, dv ...) are
 *     reproducible too.
 *     So are split-K products of ctclip_gemm_bf16 given a `splitk_ws` (the weight gradients, the 294 912 -> 512 visual
 *     projection).
 *     ORDER-DEPENDENT (f32 atomics / LDS locks, last-bit differences between runs): split-K products WITHOUT a workspace,
 *     d(bias) of ctclip_attn_bwd / ctclip_attn_hm_bwd (table and dense; ctclip_attn_dbias_ordered is the reproducible form,
 *     taken by the host layer under torch.use_deterministic_algorithms(True)), embed_sum of ctclip_vq_ema_accum
 *     (ctclip_vq_ema_accum_sorted is the reproducible form and the one the host layer uses).
 */
#ifndef CTCLIP_HIP_H
#define CTCLIP_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* floats of scratch behind every `partials` argument (8 MiB) */
#define CTCLIP_PARTIALS_FLOATS (1L << 21)
/* floats of scratch that hold the split-K partial products of every GEMM of the CT-CLIP step (144 MiB) */
#define CTCLIP_SPLITK_WS_FLOATS (36L << 20)

/* ---- GEMM (MFMA) -------------------------------------------------------------------------------
 * C[M,N] = alpha * opA(A) opB(B) (+bias[N]) (+resid[M,N]) ; act: 0 none, 1 erf-GELU (anything else:
 * hipErrorInvalidValue).
 * a_kmajor=1: A is [M][K]; 0: A is [K][M].  b_kmajor=1: B is [N][K] (nn.Linear weight); 0: [K][N].
 * c_fp32: output f32 instead of bf16.  accumulate=1: C (f32) += result (caller pre-initialises C; bias/resid added
 * once); split_k>1 splits K over workgroups and needs accumulate.  splitk_ws (optional scratch of splitk_ws_floats floats,
 * private to the stream while the call runs; CTCLIP_SPLITK_WS_FLOATS holds every product of the path): each split stores
 * its [M, N] partial product there with plain stores and the partials are added to C in split order -- reproducible, and
 * faster than float atomics (~1.3 TB/s chip-wide); the split count is cut to what the scratch holds.  Without it (or when
 * ldc != N) the splits add to C with f32 atomics.
 * Replaces every nn.Linear / einsum GEMM of the path: attention.py:47,50,118-119,124,142;
 * ctvit.py:50; ctclip.py:115-116,127; transformers BertSelfAttention/BertIntermediate/BertOutput
 * dense layers; and their autograd (dgrad: a_kmajor=1,b_kmajor=0; wgrad: 0,0). */
int ctclip_gemm_bf16(const void* A, const void* B, void* C, const float* bias, const float* resid,
                     int M, int N, int K, long lda, long ldb, long ldc, long ldr,
                     int a_kmajor, int b_kmajor, int c_fp32, int split_k, int accumulate, float alpha, int act,
                     float* splitk_ws, long splitk_ws_floats, void* stream);

/* The same k-major x k-major product on the one-wave-per-SIMD kernel (csrc/gemm5.hip: 4 waves x 128 x 128, accumulators in
 * AccVGPRs, paired full-line LDS-DMA) CALLED DIRECTLY: ctclip_gemm_bf16 / _geglu / _geglu_bwd send it the shapes it measured
 * faster on (FF1 + GEGLU, N >= 2048, plain products with K >= 1024); this entry takes any shape the kernel is eligible for -- K % 64 == 0, K >= 192, 16-byte
 * aligned outputs -- and returns hipErrorInvalidValue otherwise (so short rings, K = 192 .. 960, can be tested and measured
 * without a size gate).  act: 0 none, 1 erf-GELU, 2 = FF1 + GEGLU (C = h [M, N] in [value 32 | gate 32] blocks, G = g
 * [M, N / 2]), 3 = FF2 data gradient + GEGLU backward (G = h, overwritten by d(h); C unused).  Reference: every nn.Linear of
 * src/utils/attention.py:38-51,118-124. */
int ctclip_gemm5_bf16(const void* A, const void* B, void* C, const float* bias, const float* resid, int M, int N, int K, long lda,
                      long ldb, long ldc, long ldr, int c_fp32, float alpha, int act, void* G, long ldg, void* stream);

/* scores = A[M,K] B[N,K]^T without materialising them: per column n, the top-2 (value, row) of each 64-row
 * slab of M.  part_val/part_idx are [N][2*ceil(M/128)][2]; empty slots carry index 0x7fffffff.  VQ nearest-code
 * search, ctvit.py:118 (library vector_quantize_pytorch, cosine-sim codebook). */
int ctclip_gemm_argmax_partial(const void* A, const void* B, float* part_val, int* part_idx,
                               int M, int N, int K, long lda, long ldb, void* stream);
/* same search, one workgroup per 128 columns sweeping all rows with a running per-lane top-4: part_val/part_idx are
 * [N][16] (top-4 of each of 4 disjoint row subsets).  This is the variant the VQ module uses. */
int ctclip_vq_topk(const void* A, const void* B, float* part_val, int* part_idx, int M, int N, int K, long lda, long ldb,
                   void* stream);
/* the same search with the codebook split into `code_groups` groups of whole 256-code tiles, the workgroups an XCD holds at a
 * time sharing token tiles and code tiles through its L2 (csrc/gemm3.hip): part_val/part_idx are [N][16 * code_groups] (top-4 of
 * each of 4 row subsets of every group); code_groups divides 32 and M / 256, M % 256 == 0, K % 32 == 0.  ctclip_vq_select takes
 * the wider candidate list as it is.  Call site src/utils/ctvit.py:117-118. */
int ctclip_vq_topk_grouped(const void* A, const void* B, float* part_val, int* part_idx, int M, int N, int K, long lda, long ldb,
                           int code_groups, void* stream);

/* ---- LayerNorm: attention.py:27-34 (beta==NULL), attention.py:46, ctvit.py:51, BertLayerNorm; gamma == NULL (then beta is
 * ignored): the plain normalised rows ---- */
int ctclip_layernorm_fwd(const float* x, const float* gamma, const float* beta, void* y_bf16, float* y_f32,
                         float* mean, float* rstd, int rows, int dim, float eps, void* stream);
/* dx = dres + LN'(dy); optional bf16 copy of dx; dgamma/dbeta accumulated (dbeta may be NULL). */
int ctclip_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                         const float* dres, float* dx, void* dx_bf16, float* dgamma, float* dbeta, int rows, int dim,
                         float* partials, void* stream);
/* same with the LN-path gradient in bf16 (the output of a bf16 data-gradient GEMM) and an optional second, bf16,
 * residual-path term: dx = dres + dres2 + LN'(dy).  (the K/V projection of attention.py:138 reads the un-normalised x, so
 * its data gradient by-passes LN'.) */
/* The same LayerNorm with the token re-ordering between the CT-ViT's spatial and temporal transformers folded into its
 * row addressing (ctvit.py:96,99,101: `(b t)(h w) d -> (b h w) t d` and back; Transformer.norm_out, attention.py:311,336):
 * x rows are [B][A][C], y rows (and the backward's dy rows) are [B][C][A].  mean / rstd / dx stay in x order. */
int ctclip_layernorm_swap_fwd(const float* x, const float* gamma, const float* beta, float* y_f32, float* mean, float* rstd,
                              int rows, int dim, float eps, int A, int C, void* stream);
int ctclip_layernorm_swap_bwd(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                              float* dx, void* dx_bf16, float* dgamma, float* dbeta, int rows, int dim, int A, int C,
                              float* partials, void* stream);
int ctclip_layernorm_bwd_bf16(const void* dy_bf16, const float* x, const float* gamma, const float* mean, const float* rstd,
                              const float* dres, const void* dres2_bf16, float* dx, void* dx_bf16, float* dgamma,
                              float* dbeta, int rows, int dim, float* partials, void* stream);

/* LayerNorm backward from the saved NORMALISED rows xhat (bf16; ctclip_layernorm_fwd with gamma == NULL writes them) when the
 * affine part has been folded into the projection that follows (ctclip_patch_affine_fold / _bwd give d(gamma), d(beta) from
 * the weight-gradient product): dy is the gradient w.r.t. xhat, dx = dres + dres2 + rstd (dy - mean(dy) - xhat mean(dy xhat)).
 * Neither the f32 input row nor the mean is read (attention.py:27-34,140). */
int ctclip_layernorm_bwd_xhat(const void* dy_bf16, const void* xhat_bf16, const float* rstd, const float* dres,
                              const void* dres2_bf16, float* dx, void* dx_bf16, int rows, int dim, void* stream);

/* The LayerNorm backward of an attention block inside the data-gradient GEMM that feeds it (attention.py:140-142 backward): with
 * the LayerNorm's gamma folded into the q projection (Wqg), the block's input gradient is
 *   dx = rstd (dq Wqg - mean(.) - xhat mean(. xhat)) + dkv Wkv + dres
 *      = [rstd dq | dkv] [Wqg ; Wkv] - c1[row] - xhat[row][:] c2[row] + dres,      c1 = rstd/dim sum_k dq_k wbar_k,
 *                                                                                  c2 = rstd/dim sum_k dq_k q_k
 * (wbar_k = sum_c Wqg[k][c]; q = xhat Wqg^T is the raw projection the head-norm backward reads anyway): ONE product over
 * K = inner + 2 inner whose epilogue subtracts the two row terms -- no [tokens, dim] gradient leaves the chip twice and there is
 * no separate LayerNorm-backward pass.  A: [M, K] bf16 (row-scaled dq and c1 / c2 come from ctclip_headnorm_bwd_ln), B: [N, K]
 * bf16, xhat: [M, N] bf16 contiguous, dres: [M, N] f32 or NULL; dx [M, N] f32, dx_bf16 optional.  K % 32 == 0, N % 8 == 0. */
int ctclip_gemm_bf16_lnbwd(const void* A, const void* B, float* dx, void* dx_bf16, int M, int N, int K, long lda, long ldb,
                           const void* xhat, const float* c1, const float* c2, const float* dres, void* stream);

/* ---- per-head cosine normalisation: y = x/|x| * scale[d] * mult   (attention.py:151-153,155) ----
 * x_hm_n / y_hm_n > 0: that operand is in the HEAD-MAJOR layout [sequence][head][token][dhead] with x_hm_n tokens per
 * sequence (rows % x_hm_n == 0; its ld is ignored) -- the operand layout of ctclip_attn_hm_*; 0: row-major [rows, ld]. */
int ctclip_headnorm_fwd(const void* x, const float* scale, void* y, float* inv_norm, long rows, int heads, int dhead,
                        long ldx, long ldy, float mult, int x_hm_n, int y_hm_n, void* stream);
/* backward.  x_normed = 0: `x` is the raw projection the forward read.  x_normed = 1: `x` is the forward's OUTPUT y = u scale mult
 * (the projection normalised in its GEMM epilogue by ctclip_gemm_bf16_headnorm, nothing else of it kept): u = y / (scale mult),
 * raw row = u / inv_norm; a channel whose learned scale is exactly 0 gets no gradient through it. */
int ctclip_headnorm_bwd(const void* dy, const void* x, const float* inv_norm, const float* scale, void* dx,
                        float* dscale, long rows, int heads, int dhead, long lddy, long ldx, long lddx, float mult,
                        int x_hm_n, int x_normed, float* partials, void* stream);
/* the same for the q path of a block whose LayerNorm backward runs inside ctclip_gemm_bf16_lnbwd: also writes
 * dx_scaled[row][:] = rstd[row] dx[row][:] (bf16, row stride lddxs) and the two row constants of that epilogue,
 * c1[row] = rstd/ln_dim sum_k dx_k wbar_k and c2[row] = rstd/ln_dim sum_k dx_k x_k (dx as rounded to bf16).
 * heads * dhead == 256 and dhead == 32 (a row's heads are the 32 lanes of half a wave); x_hm_n / x_normed as above. */
int ctclip_headnorm_bwd_ln(const void* dy, const void* x, const float* inv_norm, const float* scale, void* dx,
                           float* dscale, long rows, int heads, int dhead, long lddy, long ldx, long lddx, float mult,
                           const float* rstd, const float* wbar, int ln_dim, void* dx_scaled, long lddxs, float* c1, float* c2,
                           int x_hm_n, int x_normed, float* partials, void* stream);

/* The q / k (or k | v) projection WITH the per-head cosine normalisation in the GEMM's register epilogue (attention.py:142,
 * 146-153): C = A[M,K] B[N,K]^T, and every head (32 columns) of the first norm_cols columns leaves as
 *   y = x / max(|x|, 1e-12) * scale[d] * mult        inv_norm[row][head] = 1 / max(|x|, 1e-12)   ([M, norm_cols / 32] f32)
 * computed from the f32 accumulators (a head's 32 columns sit in four lanes of the wave: two lane-row swaps); columns behind
 * norm_cols are stored as they are (the v half of a kv product).  The raw projection is never written: the backward works from y
 * (ctclip_headnorm_bwd with x_normed = 1).  n_tokens > 0: C in the head-major layout [part][sequence][head][token][32] of
 * ctclip_gemm_bf16_headmajor (ldc ignored); n_tokens = 0: row-major [M, ldc].  dhead is 32; K % 32 == 0, N % 64 == 0,
 * norm_cols % 64 == 0, C 16-byte aligned.  Replaces the head-norm pass over q and over k and the round trip of the raw q / k. */
int ctclip_gemm_bf16_headnorm(const void* A, const void* B, void* C, float* inv_norm, const float* scale, int M, int N, int K,
                              long lda, long ldb, long ldc, int n_tokens, int heads, int norm_cols, float mult, void* stream);

/* ---- fused attention (attention.py:155-180; BertSelfAttention) ----------------------------------
 * q,k,v,o: [nseq*n, ld] bf16, head h in columns h*dhead.. ; dhead in {32,64}.
 * bias: [heads,n,n] f32 or NULL; mask: additive [nseq,n] f32 or NULL; lse: [nseq,heads,n]. */
int ctclip_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse, const float* bias,
                    const float* mask, int nseq, int n, int heads, int dhead, long ldq, long ldk, long ldv, long ldo,
                    float scale, void* stream);
/* d(bias): dense [heads,n,n] atomics if dbias_dense != NULL; else reduced on chip into a [heads][table_size] table
 * indexed by relidx[n*n] (uint16) if relidx != NULL, or -- when grid_h*grid_w == n -- by the 2-D relative position
 * (yi-yj+h-1)*(2w-1) + (xi-xj+w-1) computed on the fly (table_size = (2h-1)(2w-1), the CT-ViT position bias,
 * attention.py:262-268); else skipped.  delta: scratch [nseq,heads,n]. */
int ctclip_attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* dO, const float* lse,
                    float* delta, void* dq, void* dk, void* dv, const float* bias, const float* mask,
                    float* dbias_dense, const uint16_t* relidx, float* dbias_table, int table_size, int grid_h, int grid_w,
                    int nseq, int n, int heads, int dhead, long ldq, long ldk, long ldv, long ldo, long lddo, long lddq,
                    long lddk, long lddv, float scale, void* stream);
/* The same pair with attention-probability dropout (transformers BertSelfAttention: `attention_probs = self.dropout(
 * attention_probs)`, modeling_bert.py): `keep` [nseq, heads, n, n] holds one byte per probability (non-zero = kept), drawn by
 * the caller; kept probabilities are multiplied by keep_scale = 1/(1-p) on their way into P.V, the softmax normaliser and
 * lse are those of the undropped row.  The backward applies the same flags, so the two passes agree by construction. */
int ctclip_attn_fwd_dropout(const void* q, const void* k, const void* v, void* o, float* lse, const float* bias,
                            const float* mask, const uint8_t* keep, float keep_scale, int nseq, int n, int heads, int dhead,
                            long ldq, long ldk, long ldv, long ldo, float scale, void* stream);
int ctclip_attn_bwd_dropout(const void* q, const void* k, const void* v, const void* o, const void* dO, const float* lse,
                            float* delta, void* dq, void* dk, void* dv, const float* bias, const float* mask,
                            const uint8_t* keep, float keep_scale, float* dbias_dense, const uint16_t* relidx,
                            float* dbias_table, int table_size, int grid_h, int grid_w, int nseq, int n, int heads, int dhead,
                            long ldq, long ldk, long ldv, long ldo, long lddo, long lddq, long lddk, long lddv, float scale,
                            void* stream);
/* ---- the same attention on HEAD-MAJOR operands (csrc/attention_hm.hip): the CT-ViT's spatial attention, attention.py:146-182
 * at n = 576, d_head = 32 with the relative-position bias shared by all sequences.
 *   q, k, v, dO: bf16 [nseq][heads][n][32] (written by ctclip_headnorm_fwd / ctclip_gemm_bf16_headmajor);
 *   o, dq, dk, dv: bf16 row-major [nseq * n, ld], head h in columns 32 h ..;  lse, delta: [nseq, heads, n] f32.
 * LOG2 DOMAIN: the caller has folded scale * log2(e) into q (head-norm's `mult`), so the natural logit is
 *   ln(2) * (q . k) + bias[h][i][j]   and   P = softmax of it;   dq, dk are the gradients w.r.t. the q, k GIVEN.
 * shift (forward, optional): [heads + 1] floats from ctclip_attn_shift -- per head a bound B_h on every |log2-logit| (a
 * function of q_scale, k_scale and the bias alone), and a flag.  While B_h <= 60 binades the softmax needs no maximum at all
 * (p = 2^s stays inside f32 / bf16 range, sums in f32); when the flag says the bound is wider, the online-softmax kernel of
 * the same launch pair runs instead (both are enqueued, one returns at once: no host synchronisation).
 * n % 32 == 0, n <= 768 (n <= 640 when a bias gradient is asked for: hipErrorInvalidValue beyond); d(bias) as ctclip_attn_bwd (dense, or the [heads][table_size] table). */
int ctclip_attn_shift(const float* q_scale, const float* k_scale, int dhead, float qk_mult, const float* bias, long bias_count,
                      long bias_head_stride, long bias_elem_stride, int heads, float* shift, void* stream);
int ctclip_attn_hm_fwd(const void* q, const void* k, const void* v, void* o, float* lse, const float* bias,
                       const float* shift, int nseq, int n, int heads, long ldo, void* stream);
int ctclip_attn_hm_bwd(const void* q, const void* k, const void* v, const void* o, const void* dO, const float* lse,
                       float* delta, void* dq, void* dk, void* dv, const float* bias, float* dbias_dense,
                       const uint16_t* relidx, float* dbias_table, int table_size, int grid_h, int grid_w, int nseq, int n,
                       int heads, long ldo, long lddq, long lddk, long lddv, void* stream);
/* C = A[M,K] B[N,K]^T (both k-major, K % 32 == 0, N % 64 == 0) written as bf16 in that head-major layout:
 * [part][sequence][head][token][32], token = row % n_tokens, head = (col / 32) % heads, part = col / (32 heads)
 * (the K/V projection attention.py:119,142 has two parts; the out-projection's data gradient one). */
int ctclip_gemm_bf16_headmajor(const void* A, const void* B, void* C, int M, int N, int K, long lda, long ldb, int n_tokens,
                               int heads, void* stream);
/* ---- reproducible d(bias) (csrc/attention_det.hip): dbias[heads, n, n] += sum over the sequences of dS = P (dP - delta), every
 * (head, 32 x 32 tile) summed by one workgroup in sequence order -- what ctclip_attn_bwd / _hm_bwd compute with LDS locks and
 * float atomics, bit-identical from run to run, at about the cost of a dQ pass.  Used when deterministic algorithms are
 * requested (the reference's attribution code does: src/utils/visualizations.py:29-39); the gradient passes are then called
 * without a bias gradient.  layout_hm = 1: q / k / v / dO head-major [nseq][heads][n][32] (ld* ignored); 0: row-major.
 * natural logit = scale * (q . k) + bias;  d_head 32 (narrower heads zero-padded), any n.  ctclip_attn_dbias_table gathers the dense gradient into
 * the [heads][(2 gh - 1)(2 gw - 1)] relative-position table of a gh x gw grid (one owner per entry, += ). */
int ctclip_attn_dbias_ordered(const void* q, const void* k, const void* v, const void* dO, const float* lse,
                              const float* delta, const float* bias, float* dbias, int nseq, int n, int heads, int layout_hm,
                              long ldq, long ldk, long ldv, long lddo, float scale, void* stream);
int ctclip_attn_dbias_table(const float* dbias_dense, float* dbias_table, int heads, int grid_h, int grid_w, void* stream);
/* probabilities [nseq,heads,n,n] f32, for callers that want Attention.forward's second output */
int ctclip_attn_probs(const void* q, const void* k, const float* lse, const float* bias, const float* mask,
                      float* probs, int nseq, int n, int heads, int dhead, long ldq, long ldk, float scale,
                      void* stream);

/* ---- elementwise ------------------------------------------------------------------------------- */
int ctclip_cast_f32_bf16(const float* x, void* y, long n, void* stream);
/* GEGLU attention.py:38-41: g = gelu(gate) * val.  h holds value and gate columns interleaved in blocks of `block`
 * columns ([val block | gate block | val block | ...]); block = inner is the reference's [val | gate] halves. */
int ctclip_geglu_fwd(const void* h, void* g, long rows, int inner, int block, long ldh, long ldg, void* stream);
int ctclip_geglu_bwd(const void* dg, const void* h, void* dh, long rows, int inner, int block, long lddg, long ldh,
                     void* stream);
/* Linear(dim, 2*inner, no bias) + GEGLU in one pass (attention.py:38-50): H[M, 2*inner] = A[M,K] Bw[2*inner,K]^T with the
 * rows of Bw interleaved in 32-row value / gate blocks, and G[M, inner] = gelu(gate) * value written by the same
 * epilogue (H is still needed by the backward).  inner % 64 == 0, K % 8 == 0. */
int ctclip_gemm_bf16_geglu(const void* A, const void* Bw, void* H, void* G, int M, int inner, int K, long lda, long ldb,
                           long ldh, long ldg, void* stream);
/* backward of the same pair: dg = dY[M,K] W2T[inner,K]^T never leaves the chip; the epilogue reads the value / gate
 * pre-activations from H (32-column interleaved blocks) and overwrites them with their gradients (H_dH in place).
 * dG_scratch [M, inner] bf16 is only used by the small-problem path (may be NULL for large ones). */
int ctclip_gemm_bf16_geglu_bwd(const void* dY, const void* W2T, void* H_dH, void* dG_scratch, int M, int inner, int K,
                               long lddy, long ldw, long ldh, long lddg, void* stream);
int ctclip_gelu_fwd(const void* h, void* m, long n, void* stream);
int ctclip_gelu_bwd(const void* dm, const void* h, void* dh, long n, void* stream);
/* out[b,c,a,:] = in[b,a,c,:] : the spatial<->temporal token re-orderings of ctvit.py:94-101 */
int ctclip_swap_middle_f32(const float* in, float* out, long B, int A, int C, int D, void* stream);
/* mean over the middle axis: ctclip.py:111 */
int ctclip_mean_mid_fwd(const float* x, void* y_bf16, float* y_f32, long B, int T, long F, void* stream);
int ctclip_mean_mid_bwd(const float* dy, float* dx, long B, int T, long F, void* stream);
int ctclip_add_f32(const float* a, const float* b, float* y, void* y_bf16, long n, void* stream);

/* ---- PEG depthwise causal conv + residual, channels-last, memory order (attention.py:55-83,325) ----
 * w27 is tap-major [27][d] (tap = (kt*3+kh)*3+kw), a transposed copy of dsconv.weight[d,1,3,3,3].
 * residual=1 fuses the `+ x` of attention.py:325 (forward) / the `+ dy` of its backward. */
int ctclip_peg_fwd(const float* x, const float* w27, const float* bias, float* y, void* y_bf16, long B, int T, int H,
                   int W, int d, int residual, void* stream);
int ctclip_peg_bwd_data(const float* dy, const float* w27, float* dx, void* dx_bf16, long B, int T, int H, int W, int d,
                        int residual, void* stream);
int ctclip_peg_bwd_weight(const float* dy, const float* x, float* dw27, float* dbias, long B, int T, int H, int W, int d,
                          float* partials, void* stream);

/* ctclip_peg_bwd_data and ctclip_peg_bwd_weight in ONE pass over dy (attention.py:55-83 backward): both gradients are sums over
 * the same 27 neighbours of dy -- dx[t,h,w] = [dy] + sum w27[tap] N(tap), dw27[tap] += x[t,h,w] N(tap), N(kt,kh,kw) = dy[t-kt+2,
 * h-kh+1, w-kw+1] -- so with the dy planes t, t+1, t+2 in LDS and x in registers each neighbour read feeds both.  dx / dx_bf16 as
 * ctclip_peg_bwd_data, dw27 / dbias accumulated as ctclip_peg_bwd_weight (two-stage, reproducible).  Takes the grids the plane
 * tiling takes (e.g. the CT-ViT's 24 x 24, d % 16 == 0); hipErrorInvalidValue otherwise. */
int ctclip_peg_bwd_fused(const float* dy, const float* x, const float* w27, float* dx, void* dx_bf16, float* dw27, float* dbias,
                         long B, int T, int H, int W, int d, int residual, float* partials, void* stream);

/* ---- tubelet gather + LayerNorm(c*pt*p*p) -> bf16 GEMM operand [tokens, ldA] (ctvit.py:44-49) ----
 * volume: [B,C,Dz,Hy,Wx] f32 or bf16; pad columns F..ldA-1 are written as zero.  gamma == NULL (then beta is ignored): the
 * plain normalised rows, for callers that fold the affine part into the projection (ctclip_patch_affine_fold). */
int ctclip_patch_ln_fwd(const void* volume, int volume_is_bf16, const float* gamma, const float* beta, void* A_bf16,
                        float* mean, float* rstd, int B, int C, int Dz, int Hy, int Wx, int pt, int p, long ldA,
                        float eps, void* stream);
/* The affine part of that LayerNorm folded into the tubelet projection (ctvit.py:49-50):
 *   z = (xhat gamma + beta) W^T + b = xhat (W gamma)^T + (b + W beta), so ctclip_patch_ln_fwd is called with gamma = 1, beta = 0
 *   and the GEMM with Wg_bf16[N][ldw] = bf16(W[n][f] gamma[f]) (pad columns zero) and bias_folded[n] = b[n] + sum_f W[n][f] beta[f].
 * Backward, from G[N,F] = dz^T xhat (f32, the one weight-gradient product) and db[N] = colsum(dz):
 *   dW[n][f] += G gamma[f] + db[n] beta[f];  dgamma[f] += sum_n W[n][f] G[n][f];  dbeta[f] += sum_n W[n][f] db[n]
 * -- exact, and no [tokens, F] gradient is ever formed.  W is [N][F] f32 contiguous (nn.Linear weight). */
/* Any Linear behind a LayerNorm can use the pair (the attention blocks' q projection does: attention.py:140,142): bias, beta,
 * bias_folded, db and dbeta may be NULL (a LayerNorm without beta, a Linear without bias). */
int ctclip_patch_affine_fold(const float* W, const float* bias, const float* gamma, const float* beta, void* Wg_bf16,
                             float* bias_folded, int N, int F, long ldw, void* stream);
/* G is [N][ldg] (ldg >= F + ncorr); ncorr > 0: the columns F .. F + ncorr - 1 of a row hold a correction that is subtracted from
 * every G[n][f] first (ctclip_patch_wgrad_fused); ncorr = 0, ldg = F: the plain product. */
int ctclip_patch_affine_bwd(const float* G, const float* db, const float* W, const float* gamma, const float* beta, float* dW,
                            float* dgamma, float* dbeta, int N, int F, long ldg, int ncorr, void* stream);
/* ---- the tubelet embedding as ONE pass over the volume (ctvit.py:44-50: Rearrange + LayerNorm(F) + Linear(F, N)) -------------
 * Z[tokens, N] (f32) = LayerNorm(F)(tubelet) (W gamma)^T + (b + W beta) with the MFMA operand built from the raw voxels inside
 * the GEMM (csrc/patch_gemm.hip): every voxel is read from HBM once, centred on a per-token constant c = bf16(mean of the
 * tubelet's first p2-run) -- exactly 0 for a constant tubelet, as the reference's xhat -- and rounded to bf16 in the load
 * block; mean and variance ride along in f32 and the epilogue applies  z = rstd (acc - (mean - c) wsum[n]) + bias_folded[n].
 * The [tokens, F] normalised operand of ctclip_patch_ln_fwd is never written.
 *   volume [B,C,Dz,Hy,Wx] bf16, 16-byte aligned;  Wfold_bf16 [N][ldw] and bias_folded [N] from ctclip_patch_affine_fold;
 *   wsum[n] = sum_f Wfold[n][f] (f32, of the bf16 values);  tstat [tokens][4] f32 output: (c, mean - c, rstd, mean) per token,
 *   what the backward needs (ctclip_patch_wgrad_fused; ctclip_patch_ln_bwd_dx takes columns 3 and 2).
 * Geometry it takes: p % 4 == 0, Wx % 4 == 0, F = C pt p p a multiple of 32 in [128, 4096], N == 512; hipErrorInvalidValue
 * otherwise (the host layer then keeps ctclip_patch_ln_fwd + ctclip_gemm_bf16). */
int ctclip_patch_embed_fused(const void* volume_bf16, const void* Wfold_bf16, long ldw, const float* wsum, const float* bias_folded,
                             float* Z, long ldz, float* tstat, int B, int C, int Dz, int Hy, int Wx, int pt, int p, int N,
                             float eps, void* stream);
/* The weight-gradient product of that projection, G[N][F] = dz^T xhat, with xhat RECOMPUTED from the volume (gemm4's tile and loop;
 * the feature operand's tile is written by the load block from registers as bf16((x - c) rstd) = xhat + (mean - c) rstd):
 *   G[n][0 .. F) (+)= sum_tokens dz[tok][n] bf16((x - c) rstd)[tok][f],    G[n][F], G[n][F + 1] (+)= sum_tokens dz[tok][n] v[tok]
 * where v = (mean - c) rstd split into a high and a low bf16 part: ctclip_patch_affine_bwd (ncorr = 2) subtracts G[n][F] + G[n][F + 1]
 * from every G[n][f] and gets d(W), d(gamma), d(beta) as before.  dz [tokens, lddz] bf16 (the LayerNorm(N) backward's bf16
 * copy), tstat from ctclip_patch_embed_fused, G [N][ldg] f32 with ldg == F + 2, accumulated into (the caller zeroes it); the
 * tokens are split over the workgroups, partial tiles go through splitk_ws and are summed in split order (reproducible).
 * Same geometry as the forward, tokens % 32 == 0, N % 8 == 0. */
int ctclip_patch_wgrad_fused(const void* volume_bf16, const void* dz_bf16, long lddz, const float* tstat, float* G, long ldg, int B,
                             int C, int Dz, int Hy, int Wx, int pt, int p, int N, float* splitk_ws, long splitk_ws_floats,
                             void* stream);
/* d(volume) [B,C,Dz,Hy,Wx] f32 of the gather + LayerNorm above (input attribution: integrated gradients,
 * src/utils/visualizations.py:851-910; training never needs it). */
int ctclip_patch_ln_bwd_dx(const void* volume, int volume_is_bf16, const void* dA_bf16, long ldd, const float* gamma,
                           const float* mean, const float* rstd, float* dvolume, int B, int C, int Dz, int Hy, int Wx, int pt,
                           int p, void* stream);

/* ---- row l2-normalisation (VQ input ctvit.py:118; latents ctclip.py:119-120) ---- */
int ctclip_rownorm_fwd(const float* x, void* y_bf16, float* y_f32, float* inv_norm, long rows, int dim, float eps,
                       void* stream);
int ctclip_rownorm_bwd(const float* dy, const float* x, const float* inv_norm, float* dx, long rows, int dim, void* stream);

/* ---- VQ: finalise arg-max + gather codebook rows; EMA codebook update (ctvit.py:117-118) ---- */
/* n_cand candidates per token from ctclip_gemm_argmax_partial; those within `margin` of the best bf16 score are
 * re-scored exactly in f32 (x*inv_norm . embed[c]); writes the arg-max and gathers its f32 codebook row. */
int ctclip_vq_select(const float* part_val, const int* part_idx, int n_cand, const float* x, const float* inv_norm,
                     const float* embed, long* idx_out, float* quant, long ntok, int dim, float margin, void* stream);
int ctclip_vq_ema_accum(const float* x, const float* inv_norm, const long* idx, float* bins, float* embed_sum, long ntok,
                        int dim, void* stream);
/* embed_sum of the same update WITHOUT atomics, from the tokens sorted by code (stable): order[p] = token, code_sorted[p] = its
 * code (nondecreasing).  Chunks of 256 sorted rows are summed in order; a code reaching across chunk borders is finished by
 * the chunk where it begins, in chunk order: bit-reproducible.  Scratch: edge [2 ceil(ntok/256)][dim] floats, edge_code
 * [3 ceil(ntok/256)] int64.  dim <= 1024.  (The counts are the segment lengths: the caller adds them to bins.) */
int ctclip_vq_ema_accum_sorted(const float* x, const float* inv_norm, const long* order, const long* code_sorted,
                               float* embed_sum, float* edge, long* edge_code, long ntok, int dim, void* stream);
int ctclip_vq_ema_update(float* embed, float* cluster, const float* bins, const float* embed_sum, int ncodes, int dim,
                         float decay, void* stream);

/* ---- f32 tail: generic f32 GEMM (same layout flags as ctclip_gemm_bf16; act 2 = leaky_relu(slope),
 * 3 = multiply by leaky_relu'(aux)); alpha_dev: optional device scalar multiplied in (exp'd if alpha_exp)
 * -- the `* self.temperature.exp()` of ctclip.py:127; position-bias MLP attention.py:272-277 ---- */
int ctclip_gemm_f32(const float* A, const float* B, float* C, const float* bias, const float* aux, int M, int N, int K,
                    long lda, long ldb, long ldc, long ldaux, int a_kmajor, int b_kmajor, float alpha,
                    const float* alpha_dev, int alpha_exp, int act, float slope, int accumulate, void* stream);
/* symmetric InfoNCE with targets=arange (CTClipTrainer.py:164-175): loss scalar + dloss/dsim; workspace 2*G floats */
int ctclip_infonce(const float* sim, float* loss, float* dsim, int G, float* workspace, void* stream);
/* bias[h][i][j] = table[relidx[i][j]][h]: expands the de-duplicated position-bias table (attention.py:277) */
int ctclip_bias_expand(const float* table, const uint16_t* relidx, float* bias, int heads, int n, void* stream);
int ctclip_scale_by_dev(const float* x, const float* s, float* y, long n, void* stream);
/* out[c] += sum_r x[r][c] (bias gradients); x is f32 or bf16 with row stride ld */
int ctclip_colsum_accum(const void* x, int x_is_bf16, long rows, int cols, long ld, float* out, float* partials, void* stream);
/* y = dy * (act > 0 ? 1 : slope): backward of leaky_relu given its OUTPUT (attention.py:18-19) */
int ctclip_leaky_bwd(const float* dy, const float* act, float* y, long n, float slope, void* stream);
int ctclip_dot_accum(const float* a, const float* b, float* out, long n, float* partials, void* stream);

/* ---- optimiser (CTClipTrainer.py:199-202, optimizer.py:42-54): out += sum(g^2); clip + Adam/AdamW over a flat
 * arena, clip coefficient min(1, max_norm/(sqrt(*gnorm_sq)+1e-6)) computed on device; optional bf16 shadow ---- */
int ctclip_sumsq_accum(const float* g, long n, float* out, float* partials, void* stream);
int ctclip_adam_step(float* p, const float* g, float* m, float* v, void* p_bf16, long n, float lr, float beta1,
                     float beta2, float eps, float weight_decay, int decoupled, float bias_corr1, float bias_corr2,
                     const float* gnorm_sq, float max_norm, void* stream);

/* ---- dropout of the text encoder in train mode (transformers modeling_bert.py: BertSelfAttention drops attention
 * probabilities, BertSelfOutput / BertOutput drop the dense output before the residual add; models/ctclip.py:107 calls the
 * encoder under model.train()).  keep(i) is a pure function of (seed, offset + i) -- a counter-based generator, one 32-bit
 * draw per element, keep <=> draw >= p 2^32 -- so the hidden-state dropouts store nothing: forward and backward evaluate
 * the same function.  The random stream is therefore this library's, seeded by the caller (text.py draws the seed from
 * torch's generator); it is not a replay of the reference's CUDA Philox stream. */
int ctclip_dropout_keep(uint8_t* keep, long n, float p, long seed, long offset, void* stream);      /* flags for ctclip_attn_*_dropout */
int ctclip_dropout_add(const float* x, const float* branch, float* out, long n, float p, long seed, long offset,
                       void* stream);                                                             /* out = x + keep branch / (1-p) */
int ctclip_dropout_bwd(const float* g, float* d, void* d_bf16, long n, float p, long seed, long offset,
                       void* stream);
/* FF1 weight gradient out of the blocked order: g_blocked [2 inner_padded, dim] f32 is the product d(h)^T n2 with rows in the
 * [value `block` | gate `block` | ...] order of the GEGLU weight shadow (ctclip_gemm_bf16_geglu); dw [2 inner, dim] (the
 * nn.Linear(dim, 2 inner) weight of attention.py:47: value rows, then gate rows) += its rows.  dim % 4 == 0. */
int ctclip_geglu_wgrad_unblock(const float* g_blocked, float* dw, int inner, int block, int dim, void* stream);                                                             /* d = keep g / (1-p), f32 and/or bf16 */

/* ---- BERT embeddings (transformers BertEmbeddings): word[ids] + pos[0..L) + type[token_type] ---- */
int ctclip_bert_embed_fwd(const long* ids, const long* token_type, const float* word, const float* pos, const float* type,
                          float* out, long rows, int L, int hidden, void* stream);
/* backward: no float atomics -- d(pos) one owner per entry, d(type) chunk partials + ordered sum, d(word) summed by the first row
 * of every id over its later occurrences in row order (LDS windows: any number of rows): bit-reproducible.  vocab = rows of the
 * word table (<= CTCLIP_PARTIALS_FLOATS: the scratch holds one int per id; ids outside [0, vocab) are ignored).  rows % L == 0,
 * type_vocab <= 4.  Everything is validated before the first launch. */
int ctclip_bert_embed_bwd(const long* ids, const long* token_type, const float* dy, float* dword, float* dpos,
                          float* dtype, long rows, int L, int hidden, int type_vocab, long vocab, float* partials, void* stream);

/* ---- volume ingest (src/utils/preprocess.py:84-152, model_type "ctclip"): raw scan [H,W,D] (f32 or i16) -> HU rescale
 * -> permute to [D,H,W] -> trilinear resample to (rD,rH,rW) (align_corners=False) -> clamp/1000 -> centre crop / pad with
 * pad_value to (oD,oH,oW), written as [oD,oH,oW] bf16 or f32 ---- */
int ctclip_ingest_volume(const void* raw, int raw_is_i16, int H, int W, int D, float slope, float intercept, int rD, int rH,
                         int rW, int oD, int oH, int oW, float pad_value, void* out, int out_bf16, void* stream);

/* ---- weight shadows (host side: ctclip_hip/ops.py ShadowPlan): the bf16 / transposed / zero-padded / block-interleaved /
 * gamma-scaled kernel-layout copies of the f32 master weights, all in ONE launch per optimiser step (the reference keeps one
 * fp16 autocast copy per nn.Linear call: torch.autocast in src/utils/CTClipTrainer.py:186-189).  table: ndesc descriptors of 8
 * 64-bit words {src, dst, rows, cols, src_ld, dst_ld, flags, scale}: flags bit 0 transposed destination, bit 1 bf16 source, bit 2
 * f32 destination, bit 3 row sums (dst[r] = sum_c bf16(src[r][c] scale[c])), bits 8.. blk (destination row r ->
 * (r / blk) 2 blk + r % blk; 0 = r); scale: per-column f32 factor or NULL.  tile_start[d]: first 32 x 64 tile of descriptor d,
 * total_tiles: their sum (row sums take one tile per 32 rows). */
int ctclip_shadow_multi(const void* table, const int* tile_start, int ndesc, int total_tiles, void* stream);

/* ---- diagnostic: register-resident MFMA 32x32x16 bf16 loop, blocks x 512 threads x iters x 16 MFMAs per wave; times
 * what the matrix pipes sustain at the clock the part holds under load (no reference counterpart) ---- */
int ctclip_probe_mfma(float* out, int blocks, int iters, void* stream);

/* ---- diagnostic: streaming copy of `bytes` (multiple of 16, 16-byte aligned pointers), eight independent 16-byte non-temporal
 * accesses in flight per lane: the HBM rate the part sustains, (read + write) bytes / time (no reference counterpart).
 * ctclip_probe_stream: mode 0 the same copy, 1 pure read of src (dst: >= 16 bytes of scratch), 2 pure write of dst ---- */
int ctclip_probe_copy(const void* src, void* dst, long bytes, void* stream);
int ctclip_probe_stream(const void* src, void* dst, long bytes, int mode, void* stream);

/* ---- diagnostic / test hook: fill the whole LDS (160 KiB) of every CU with a 32-bit pattern; `sink`: 4 bytes of device scratch.
 * LDS is not cleared between kernels, so a kernel that reads a word it never wrote sees the previous tenant's data; tests poison the
 * LDS with a NaN pattern in front of the kernels whose blocks are rounded up to whole waves (no reference counterpart) ---- */
int ctclip_probe_lds_fill(int pattern, void* sink, void* stream);

#ifdef __cplusplus
}
#endif
#endif
