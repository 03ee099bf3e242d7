/* C ABI of libctclip_hip.so -- hand-written gfx950 (MI355X) kernels for the CT-CLIP training step.
 *
 * The reference (injardav/CT-CLIP-UT) has no native/FFI layer: its hot path is a sequence of
 * PyTorch ops.  Each entry point below replaces one such op sequence; the reference lines are
 * cited per function (paths relative to the reference repo).  INTEGRATION.md shows the ctypes
 * binding a maintainer adds on the reference side.
 *
 * Conventions
 *   - every function returns a hipError_t as int (0 = success) and only enqueues work on `stream`
 *     (a hipStream_t passed as void*); nothing allocates, synchronises or keeps global state.
 *   - all pointers are DEVICE pointers owned by the caller; "bf16" buffers are raw uint16 storage.
 *   - matrices are row-major; `ld*` are row strides in ELEMENTS.  bf16 matrices need 16-byte aligned
 *     base pointers and strides that are multiples of 8.
 *   - gradient outputs named d<param> are ACCUMULATED with f32 atomics: the caller zeroes them.
 */
#ifndef CTCLIP_HIP_H
#define CTCLIP_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- GEMM (MFMA) -------------------------------------------------------------------------------
 * C[M,N] = alpha * opA(A) opB(B) (+bias[N]) (+resid[M,N]) ; act: 0 none, 1 erf-GELU.
 * a_kmajor=1: A is [M][K]; 0: A is [K][M].  b_kmajor=1: B is [N][K] (nn.Linear weight); 0: [K][N].
 * c_fp32: output f32 instead of bf16.  split_k>1: K is split over workgroups and C (f32) is
 * accumulated with atomics (caller pre-initialises C; bias/resid added once).
 * Replaces every nn.Linear / einsum GEMM of the path: attention.py:47,50,118-119,124,142;
 * ctvit.py:50; ctclip.py:115-116,127; transformers BertSelfAttention/BertIntermediate/BertOutput
 * dense layers; and their autograd (dgrad: a_kmajor=1,b_kmajor=0; wgrad: 0,0). */
int ctclip_gemm_bf16(const void* A, const void* B, void* C, const float* bias, const float* resid,
                     int M, int N, int K, long lda, long ldb, long ldc, long ldr,
                     int a_kmajor, int b_kmajor, int c_fp32, int split_k, float alpha, int act, void* stream);

/* scores = A[M,K] B[N,K]^T without materialising them: per column n, arg-max over each 64-row slab of
 * M.  part_val/part_idx are [N][2*ceil(M/128)].  VQ nearest-code search, ctvit.py:118 (library
 * vector_quantize_pytorch, cosine-sim codebook). */
int ctclip_gemm_argmax_partial(const void* A, const void* B, float* part_val, int* part_idx,
                               int M, int N, int K, long lda, long ldb, void* stream);

/* ---- LayerNorm: attention.py:27-34 (beta==NULL), attention.py:46, ctvit.py:51, BertLayerNorm ---- */
int ctclip_layernorm_fwd(const float* x, const float* gamma, const float* beta, void* y_bf16, float* y_f32,
                         float* mean, float* rstd, int rows, int dim, float eps, void* stream);
/* dx = dres + LN'(dy); optional bf16 copy of dx; dgamma/dbeta accumulated. */
int ctclip_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                         const float* dres, float* dx, void* dx_bf16, float* dgamma, float* dbeta, int rows, int dim,
                         void* stream);

/* ---- per-head cosine normalisation: y = x/|x| * scale[d] * mult   (attention.py:151-153,155) ---- */
int ctclip_headnorm_fwd(const void* x, const float* scale, void* y, float* inv_norm, long rows, int heads, int dhead,
                        long ldx, long ldy, float mult, void* stream);
int ctclip_headnorm_bwd(const void* dy, const void* x, const float* inv_norm, const float* scale, void* dx,
                        float* dscale, long rows, int heads, int dhead, long lddy, long ldx, long lddx, float mult,
                        void* stream);

/* ---- fused attention (attention.py:155-180; BertSelfAttention) ----------------------------------
 * q,k,v,o: [nseq*n, ld] bf16, head h in columns h*dhead.. ; dhead in {32,64}.
 * bias: [heads,n,n] f32 or NULL; mask: additive [nseq,n] f32 or NULL; lse: [nseq,heads,n]. */
int ctclip_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse, const float* bias,
                    const float* mask, int nseq, int n, int heads, int dhead, long ldq, long ldk, long ldv, long ldo,
                    float scale, void* stream);
/* d(bias): dense [heads,n,n] atomics if dbias_dense != NULL, else a [heads][table_size] table indexed by
 * relidx[n*n] (uint16) if relidx != NULL, else skipped.  delta: scratch [nseq,heads,n]. */
int ctclip_attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* dO, const float* lse,
                    float* delta, void* dq, void* dk, void* dv, const float* bias, const float* mask,
                    float* dbias_dense, const uint16_t* relidx, float* dbias_table, int table_size, int nseq, int n,
                    int heads, int dhead, long ldq, long ldk, long ldv, long ldo, long lddo, long lddq, long lddk,
                    long lddv, float scale, void* stream);
/* probabilities [nseq,heads,n,n] f32, for callers that want Attention.forward's second output */
int ctclip_attn_probs(const void* q, const void* k, const float* lse, const float* bias, const float* mask,
                      float* probs, int nseq, int n, int heads, int dhead, long ldq, long ldk, float scale,
                      void* stream);

/* ---- elementwise ------------------------------------------------------------------------------- */
int ctclip_cast_f32_bf16(const float* x, void* y, long n, void* stream);
/* GEGLU attention.py:38-41: h = [val | gate], g = gelu(gate) * val */
int ctclip_geglu_fwd(const void* h, void* g, long rows, int inner, long ldh, long ldg, void* stream);
int ctclip_geglu_bwd(const void* dg, const void* h, void* dh, long rows, int inner, long lddg, long ldh, void* stream);
int ctclip_gelu_fwd(const void* h, void* m, long n, void* stream);
int ctclip_gelu_bwd(const void* dm, const void* h, void* dh, long n, void* stream);
/* out[b,c,a,:] = in[b,a,c,:] : the spatial<->temporal token re-orderings of ctvit.py:94-101 */
int ctclip_swap_middle_f32(const float* in, float* out, long B, int A, int C, int D, void* stream);
/* mean over the middle axis: ctclip.py:111 */
int ctclip_mean_mid_fwd(const float* x, void* y_bf16, float* y_f32, long B, int T, long F, void* stream);
int ctclip_mean_mid_bwd(const float* dy, float* dx, long B, int T, long F, void* stream);
int ctclip_add_f32(const float* a, const float* b, float* y, void* y_bf16, long n, void* stream);

#ifdef __cplusplus
}
#endif
#endif
