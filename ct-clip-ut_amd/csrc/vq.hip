// Cosine-similarity vector quantiser pieces around the MFMA score/arg-max GEMM (ctclip_gemm_argmax_partial):
// row l2-normalisation (+backward), arg-max finalisation + codebook gather, EMA codebook update.
// reference call sites src/utils/ctvit.py:66,117-118; arithmetic = vector-quantize-pytorch's cosine-sim
// codebook (third party, parity unpinned -- see oracle/ctclip_oracle.py:vq_cosine).  The same row-norm
// kernels serve the latent normalisation of src/models/ctclip.py:119-120.
#include "common.h"

namespace {

// y = x / max(|x|, eps)   one wave per row, dim % 4 == 0
__global__ __launch_bounds__(256) void rownorm_fwd_kernel(const float* __restrict__ x, bf16_t* __restrict__ y16,
                                                          float* __restrict__ y32, float* __restrict__ inv_norm,
                                                          long rows, int dim, float eps) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float4* xr = (const float4*)(x + row * dim);
  const int nv = dim >> 2;
  float ss = 0.f;
  for (int c = lane; c < nv; c += 64) {
    const float4 v = xr[c];
    ss += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
  }
  const float inv = 1.0f / fmaxf(sqrtf(wave_sum(ss)), eps);
  if (lane == 0 && inv_norm) inv_norm[row] = inv;
  for (int c = lane; c < nv; c += 64) {
    float4 v = xr[c];
    v.x *= inv; v.y *= inv; v.z *= inv; v.w *= inv;
    if (y32) ((float4*)(y32 + row * dim))[c] = v;
    if (y16) {
      uint2 p;
      p.x = pack_bf16x2(v.x, v.y);
      p.y = pack_bf16x2(v.z, v.w);
      ((uint2*)(y16 + row * dim))[c] = p;
    }
  }
}

// dx = inv * (dy - u (u . dy)), u = x * inv
__global__ __launch_bounds__(256) void rownorm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                          const float* __restrict__ inv_norm, float* __restrict__ dx,
                                                          long rows, int dim) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float4* xr = (const float4*)(x + row * dim);
  const float4* dr = (const float4*)(dy + row * dim);
  const int nv = dim >> 2;
  const float inv = inv_norm[row];
  float dot = 0.f;
  for (int c = lane; c < nv; c += 64) {
    const float4 v = xr[c], d = dr[c];
    dot += (v.x * d.x + v.y * d.y) + (v.z * d.z + v.w * d.w);
  }
  dot = wave_sum(dot) * inv * inv;   // (u . dy) * inv  with u = x*inv  ->  x . dy * inv^2
  for (int c = lane; c < nv; c += 64) {
    const float4 v = xr[c], d = dr[c];
    float4 o;
    o.x = inv * (d.x - v.x * dot); o.y = inv * (d.y - v.y * dot);
    o.z = inv * (d.z - v.z * dot); o.w = inv * (d.w - v.w * dot);
    ((float4*)(dx + row * dim))[c] = o;
  }
}

// per token: finalise the nearest-code search.  The MFMA pass scored codes with bf16 operands and left the
// top-2 of every 64-code slab; every candidate within `margin` of the best bf16 score is re-scored exactly in
// f32 (x * inv_norm . embed[c]) and the exact arg-max wins (ties -> lowest code index, like torch.argmax).
// margin >= 2 * 2^-8 bounds the bf16 rounding error of two unit-vector dot products, so the result equals the
// f32 arg-max unless >2 near-tied codes share one slab.  One wave per token; then gathers the f32 codebook row.
__global__ __launch_bounds__(256) void vq_select_kernel(const float* __restrict__ part_val, const int* __restrict__ part_idx,
                                                        int n_cand, const float* __restrict__ x, const float* __restrict__ inv_norm,
                                                        const float* __restrict__ embed, long* __restrict__ idx_out,
                                                        float* __restrict__ quant, long ntok, int dim, float margin) {
  const int lane = threadIdx.x & 63;
  const long tok = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tok >= ntok) return;
  float best16 = -INFINITY;
  for (int p = lane; p < n_cand; p += 64)
    if (part_idx[tok * n_cand + p] != 0x7fffffff) best16 = fmaxf(best16, part_val[tok * n_cand + p]);
  best16 = wave_max(best16);
  const float thr = best16 - margin;
  const float inv = inv_norm[tok];
  const float* xr = x + tok * dim;
  float best = -INFINITY;
  int besti = 0x7fffffff;
  for (int p0 = 0; p0 < n_cand; p0 += 64) {
    const int p = p0 + lane;
    float v = -INFINITY;
    int ci = 0x7fffffff;
    if (p < n_cand) { v = part_val[tok * n_cand + p]; ci = part_idx[tok * n_cand + p]; }
    unsigned long long m = __ballot(ci != 0x7fffffff && v >= thr);
    while (m) {
      const int src = __ffsll((long long)m) - 1;
      m &= m - 1;
      const int c = __shfl(ci, src, 64);
      const float* e = embed + (long)c * dim;
      float d = 0.f;
      for (int k = lane; k < dim; k += 64) d += xr[k] * e[k];
      d = wave_sum(d) * inv;
      if (d > best || (d == best && c < besti)) { best = d; besti = c; }
    }
  }
  // a token whose scores are all NaN (or that has no candidate at all) has nothing to compare: give it code 0 rather
  // than let the empty-slot marker be used as a row number below and by the EMA accumulation
  if (besti == 0x7fffffff) besti = 0;
  if (lane == 0) idx_out[tok] = besti;
  if (quant) {
    const float4* e = (const float4*)(embed + (long)besti * dim);
    for (int c = lane; c < (dim >> 2); c += 64) ((float4*)(quant + tok * dim))[c] = e[c];
  }
}

// bins[c] += 1, embed_sum[c][:] += x[tok][:] * inv_norm[tok]    (one wave per token, f32 atomics)
__global__ __launch_bounds__(256) void vq_ema_accum_kernel(const float* __restrict__ x, const float* __restrict__ inv_norm,
                                                           const long* __restrict__ idx, float* __restrict__ bins,
                                                           float* __restrict__ embed_sum, long ntok, int dim) {
  const int lane = threadIdx.x & 63;
  const long tok = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tok >= ntok) return;
  const long c = idx[tok];
  const float inv = inv_norm[tok];
  if (lane == 0) atomicAdd(bins + c, 1.0f);
  for (int d = lane; d < dim; d += 64) atomicAdd(embed_sum + c * dim + d, x[tok * dim + d] * inv);
}

// The same sums WITHOUT atomics, from the tokens sorted by code (order[p] = token, code_sorted[p] = its code, nondecreasing; a
// stable sort, so equal codes keep token order).  Chunk k owns the sorted positions [k R, (k + 1) R): its 256 threads walk
// them in order, every thread a few columns of the running sum of the current code.  A code whose rows all lie inside the
// chunk is written to embed_sum by this chunk alone; the (at most two) codes that reach across a chunk border leave their
// partial sums in edge[k][0] (a code that began before the chunk) / edge[k][1] (one that goes on after it), and
// vq_ema_edge_kernel has the chunk where such a code BEGINS add the pieces up in chunk order.  One owner and one order per
// sum: bit-reproducible, and rows are read with plain loads instead of ~450 M float atomics at ~1.3 TB/s.
constexpr int EMA_R = 256;         // sorted rows per chunk
constexpr int EMA_DV = 4;          // columns per thread: dim <= 1024
__global__ __launch_bounds__(256) void vq_ema_chunk_kernel(const float* __restrict__ x, const float* __restrict__ inv_norm,
                                                           const long* __restrict__ order, const long* __restrict__ code_sorted,
                                                           long ntok, int dim, float* __restrict__ embed_sum,
                                                           float* __restrict__ edge, long* __restrict__ edge_code) {
  const long k = blockIdx.x, p0 = k * EMA_R, p1 = (p0 + EMA_R < ntok) ? p0 + EMA_R : ntok;
  const int tid = threadIdx.x;
  float acc[EMA_DV];
#pragma unroll
  for (int j = 0; j < EMA_DV; ++j) acc[j] = 0.f;
  long cur = code_sorted[p0];
  const bool first_began_before = p0 > 0 && code_sorted[p0 - 1] == cur;
  bool began_before = first_began_before;
  if (tid == 0) { edge_code[3 * k] = -1; edge_code[3 * k + 1] = -1; edge_code[3 * k + 2] = 0; }
  auto flush = [&](long c, bool ends_inside) {
    float* dst;
    if (!began_before && ends_inside) dst = embed_sum + c * dim;           // the whole code lies in this chunk
    else {
      const int slot = began_before ? 0 : 1;
      dst = edge + (2 * k + slot) * dim;
      if (tid == 0) {
        edge_code[3 * k + slot] = c;
        if (slot == 0) edge_code[3 * k + 2] = ends_inside ? 0 : 1;         // the code goes on after this chunk too
      }
    }
    const bool add = (!began_before && ends_inside);
#pragma unroll
    for (int j = 0; j < EMA_DV; ++j) {
      const int d = tid + 256 * j;
      if (d < dim) dst[d] = add ? dst[d] + acc[j] : acc[j];
      acc[j] = 0.f;
    }
  };
  for (long p = p0; p < p1; ++p) {
    const long c = code_sorted[p];
    if (c != cur) {
      flush(cur, true);
      cur = c;
      began_before = false;
    }
    const long tok = order[p];
    const float w = inv_norm[tok];
    const float* xr = x + tok * dim;
#pragma unroll
    for (int j = 0; j < EMA_DV; ++j) {
      const int d = tid + 256 * j;
      if (d < dim) acc[j] = fmaf(xr[d], w, acc[j]);
    }
  }
  flush(cur, p1 == ntok || code_sorted[p1] != cur);
}

__global__ __launch_bounds__(256) void vq_ema_edge_kernel(const float* __restrict__ edge, const long* __restrict__ edge_code,
                                                          long nchunks, int dim, float* __restrict__ embed_sum) {
  const long k = blockIdx.x;
  const long c = edge_code[3 * k + 1];
  if (c < 0) return;                                                       // no code begins here and goes on
  const int tid = threadIdx.x;
  float acc[EMA_DV];
#pragma unroll
  for (int j = 0; j < EMA_DV; ++j) {
    const int d = tid + 256 * j;
    acc[j] = d < dim ? edge[(2 * k + 1) * dim + d] : 0.f;
  }
  for (long kk = k + 1; kk < nchunks && edge_code[3 * kk] == c; ++kk) {
#pragma unroll
    for (int j = 0; j < EMA_DV; ++j) {
      const int d = tid + 256 * j;
      if (d < dim) acc[j] += edge[(2 * kk) * dim + d];
    }
    if (edge_code[3 * kk + 2] == 0) break;                                 // the code ended inside chunk kk
  }
#pragma unroll
  for (int j = 0; j < EMA_DV; ++j) {
    const int d = tid + 256 * j;
    if (d < dim) embed_sum[c * dim + d] += acc[j];
  }
}

// cluster = cluster*decay + bins*(1-decay); embed = embed*decay + unit(embed_sum/bins)*(1-decay) where bins>0
__global__ __launch_bounds__(256) void vq_ema_update_kernel(float* __restrict__ embed, float* __restrict__ cluster,
                                                            const float* __restrict__ bins, const float* __restrict__ embed_sum,
                                                            int ncodes, int dim, float decay) {
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= ncodes) return;
  const float n = bins[c];
  if (lane == 0) cluster[c] = cluster[c] * decay + n * (1.f - decay);
  const float denom = (n == 0.f) ? 1.f : n;
  float ss = 0.f;
  for (int d = lane; d < dim; d += 64) {
    const float m = embed_sum[(long)c * dim + d] / denom;
    ss += m * m;
  }
  const float inv = 1.0f / fmaxf(sqrtf(wave_sum(ss)), 1e-12f);
  for (int d = lane; d < dim; d += 64) {
    const float e = embed[(long)c * dim + d];
    const float tgt = (n == 0.f) ? e : (embed_sum[(long)c * dim + d] / denom) * inv;
    embed[(long)c * dim + d] = e * decay + tgt * (1.f - decay);
  }
}

}  // namespace

extern "C" {

int ctclip_rownorm_fwd(const float* x, void* y_bf16, float* y_f32, float* inv_norm, long rows, int dim, float eps,
                       void* stream) {
  if (rows <= 0) return 0;
  if (dim & 3) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(rownorm_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x,
                     (bf16_t*)y_bf16, y_f32, inv_norm, rows, dim, eps);
  CTCLIP_CHECK_LAUNCH();
}

int ctclip_rownorm_bwd(const float* dy, const float* x, const float* inv_norm, float* dx, long rows, int dim, void* stream) {
  if (rows <= 0) return 0;
  if (dim & 3) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(rownorm_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, dy, x,
                     inv_norm, dx, rows, dim);
  CTCLIP_CHECK_LAUNCH();
}

int ctclip_vq_select(const float* part_val, const int* part_idx, int n_cand, const float* x, const float* inv_norm,
                     const float* embed, long* idx_out, float* quant, long ntok, int dim, float margin, void* stream) {
  if (ntok <= 0) return 0;
  if (dim & 3) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(vq_select_kernel, dim3((unsigned)((ntok + 3) / 4)), dim3(256), 0, (hipStream_t)stream, part_val,
                     part_idx, n_cand, x, inv_norm, embed, idx_out, quant, ntok, dim, margin);
  CTCLIP_CHECK_LAUNCH();
}

int ctclip_vq_ema_accum(const float* x, const float* inv_norm, const long* idx, float* bins, float* embed_sum, long ntok,
                        int dim, void* stream) {
  if (ntok <= 0) return 0;
  hipLaunchKernelGGL(vq_ema_accum_kernel, dim3((unsigned)((ntok + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, inv_norm,
                     idx, bins, embed_sum, ntok, dim);
  CTCLIP_CHECK_LAUNCH();
}

int ctclip_vq_ema_accum_sorted(const float* x, const float* inv_norm, const long* order, const long* code_sorted,
                               float* embed_sum, float* edge, long* edge_code, long ntok, int dim, void* stream) {
  if (ntok <= 0) return 0;
  if (dim > 256 * EMA_DV || !edge || !edge_code) return (int)hipErrorInvalidValue;
  const long nchunks = (ntok + EMA_R - 1) / EMA_R;
  hipLaunchKernelGGL(vq_ema_chunk_kernel, dim3((unsigned)nchunks), dim3(256), 0, (hipStream_t)stream, x, inv_norm, order,
                     code_sorted, ntok, dim, embed_sum, edge, edge_code);
  hipLaunchKernelGGL(vq_ema_edge_kernel, dim3((unsigned)nchunks), dim3(256), 0, (hipStream_t)stream, edge, edge_code, nchunks,
                     dim, embed_sum);
  CTCLIP_CHECK_LAUNCH();
}

int ctclip_vq_ema_update(float* embed, float* cluster, const float* bins, const float* embed_sum, int ncodes, int dim,
                         float decay, void* stream) {
  if (ncodes <= 0) return 0;
  hipLaunchKernelGGL(vq_ema_update_kernel, dim3((unsigned)((ncodes + 3) / 4)), dim3(256), 0, (hipStream_t)stream, embed,
                     cluster, bins, embed_sum, ncodes, dim, decay);
  CTCLIP_CHECK_LAUNCH();
}

}  // extern "C"
