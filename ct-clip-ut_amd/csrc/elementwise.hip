// HBM-bound elementwise / data-movement kernels: casts, GEGLU, GELU, token-layout permutes, means.
// 16 bytes per lane everywhere; grid capped at 2048 blocks with a grid-stride loop.
#include "common.h"

namespace {

__device__ __forceinline__ float gelu_f(float x) { return gelu_erf_fast(x); }
__device__ __forceinline__ float gelu_grad_f(float x) { return gelu_erf_grad_fast(x); }
__device__ __forceinline__ void unpack8(const uint4& r, float (&f)[8]) {
  const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[2 * i] = __uint_as_float(w[i] << 16);
    f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
  }
}
__device__ __forceinline__ uint4 pack8(const float (&f)[8]) {
  uint4 o;
  o.x = pack_bf16x2(f[0], f[1]); o.y = pack_bf16x2(f[2], f[3]);
  o.z = pack_bf16x2(f[4], f[5]); o.w = pack_bf16x2(f[6], f[7]);
  return o;
}

__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, long n8) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
    const float4 a = ((const float4*)x)[2 * i], b = ((const float4*)x)[2 * i + 1];
    const float f[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    ((uint4*)y)[i] = pack8(f);
  }
}

// GEGLU, reference src/utils/attention.py:38-41: g = gelu(gate) * val.  The reference stores h as [val | gate] halves of
// width I; the fused feed-forward keeps them interleaved in blocks of `blk` columns ([val blk | gate blk | val blk | ...])
// so that a GEMM output tile holds matching value and gate columns.  blk = I is the reference layout.
// Output column j (chunks of 8, c = j/8): value column = (c / B8) * 2*B8 + c % B8 (in chunks), gate = value + B8.
__global__ __launch_bounds__(256) void geglu_fwd_kernel(const bf16_t* __restrict__ h, bf16_t* __restrict__ g, long rows,
                                                        int I8, int B8, long ldh, long ldg) {
  const long total = rows * I8;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long r = i / I8;
    const int c = (int)(i % I8);
    const int vc = (c / B8) * 2 * B8 + c % B8;
    float v[8], t[8], o[8];
    unpack8(*(const uint4*)(h + r * ldh + vc * 8), v);
    unpack8(*(const uint4*)(h + r * ldh + (long)(vc + B8) * 8), t);
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = gelu_f(t[k]) * v[k];
    *(uint4*)(g + r * ldg + c * 8) = pack8(o);
  }
}

__global__ __launch_bounds__(256) void geglu_bwd_kernel(const bf16_t* __restrict__ dg, const bf16_t* __restrict__ h,
                                                        bf16_t* __restrict__ dh, long rows, int I8, int B8, long lddg,
                                                        long ldh) {
  const long total = rows * I8;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long r = i / I8;
    const int c = (int)(i % I8);
    const int vc = (c / B8) * 2 * B8 + c % B8;
    float v[8], t[8], d[8], dv[8], dt[8];
    unpack8(*(const uint4*)(h + r * ldh + vc * 8), v);
    unpack8(*(const uint4*)(h + r * ldh + (long)(vc + B8) * 8), t);
    unpack8(*(const uint4*)(dg + r * lddg + c * 8), d);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      dv[k] = d[k] * gelu_f(t[k]);
      dt[k] = d[k] * v[k] * gelu_grad_f(t[k]);
    }
    *(uint4*)(dh + r * ldh + vc * 8) = pack8(dv);
    *(uint4*)(dh + r * ldh + (long)(vc + B8) * 8) = pack8(dt);
  }
}

// erf-GELU (transformers BertIntermediate)
__global__ __launch_bounds__(256) void gelu_fwd_kernel(const bf16_t* __restrict__ h, bf16_t* __restrict__ m, long n8) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
    float v[8], o[8];
    unpack8(((const uint4*)h)[i], v);
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = gelu_f(v[k]);
    ((uint4*)m)[i] = pack8(o);
  }
}
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const bf16_t* __restrict__ dm, const bf16_t* __restrict__ h,
                                                       bf16_t* __restrict__ dh, long n8) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
    float v[8], d[8], o[8];
    unpack8(((const uint4*)h)[i], v);
    unpack8(((const uint4*)dm)[i], d);
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = d[k] * gelu_grad_f(v[k]);
    ((uint4*)dh)[i] = pack8(o);
  }
}

// out[b, c, a, :] = in[b, a, c, :]   (f32 rows of D floats).  reference src/utils/ctvit.py:94-101 rearranges.
__global__ __launch_bounds__(256) void swap_middle_kernel(const float* __restrict__ in, float* __restrict__ out, long B,
                                                          int A, int C, int D4) {
  const long total = B * A * C * D4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int d = (int)(i % D4);
    long r = i / D4;  // output row index (b, c, a)
    const int a = (int)(r % A);
    r /= A;
    const int c = (int)(r % C);
    const long b = r / C;
    ((float4*)out)[i] = ((const float4*)in)[((b * A + a) * C + c) * D4 + d];
  }
}

// y[b, :] = mean_t x[b, t, :]  -> bf16 and/or f32.   reference src/models/ctclip.py:111
__global__ __launch_bounds__(256) void mean_mid_kernel(const float* __restrict__ x, bf16_t* __restrict__ y16,
                                                       float* __restrict__ y32, long B, int T, long F4) {
  const long total = B * F4;
  const float invT = 1.0f / (float)T;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long b = i / F4, f = i % F4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int t = 0; t < T; ++t) {
      const float4 v = ((const float4*)x)[(b * T + t) * F4 + f];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    s.x *= invT; s.y *= invT; s.z *= invT; s.w *= invT;
    if (y32) ((float4*)y32)[i] = s;
    if (y16) {
      uint2 p;
      p.x = pack_bf16x2(s.x, s.y);
      p.y = pack_bf16x2(s.z, s.w);
      ((uint2*)y16)[i] = p;
    }
  }
}

// dx[b, t, :] = dy[b, :] / T
__global__ __launch_bounds__(256) void mean_mid_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, long B,
                                                           int T, long F4) {
  const long total = B * T * F4;
  const float invT = 1.0f / (float)T;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long f = i % F4, b = i / (F4 * T);
    float4 v = ((const float4*)dy)[b * F4 + f];
    v.x *= invT; v.y *= invT; v.z *= invT; v.w *= invT;
    ((float4*)dx)[i] = v;
  }
}

// y = a + b (f32) with optional bf16 shadow of the sum
__global__ __launch_bounds__(256) void add_f32_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                      float* __restrict__ y, bf16_t* __restrict__ y16, long n4) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    float4 u = ((const float4*)a)[i];
    if (b) {
      const float4 v = ((const float4*)b)[i];
      u.x += v.x; u.y += v.y; u.z += v.z; u.w += v.w;
    }
    if (y) ((float4*)y)[i] = u;
    if (y16) {
      uint2 p;
      p.x = pack_bf16x2(u.x, u.y);
      p.y = pack_bf16x2(u.z, u.w);
      ((uint2*)y16)[i] = p;
    }
  }
}

inline unsigned grid_for(long work) {
  long b = (work + 255) / 256;
  if (b > 2048) b = 2048;
  if (b < 1) b = 1;
  return (unsigned)b;
}

}  // namespace

extern "C" {

int ctclip_cast_f32_bf16(const float* x, void* y, long n, void* stream) {
  if (n <= 0) return 0;
  if (n & 7) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(grid_for(n / 8)), dim3(256), 0, (hipStream_t)stream, x, (bf16_t*)y, n / 8);
  CTCLIP_CHECK_LAUNCH();
}

int ctclip_geglu_fwd(const void* h, void* g, long rows, int inner, int block, long ldh, long ldg, void* stream) {
  if (rows <= 0) return 0;
  if ((inner & 7) || (ldh & 7) || (ldg & 7) || block <= 0 || (block & 7) || (inner % block)) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(geglu_fwd_kernel, dim3(grid_for(rows * (inner / 8))), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)h, (bf16_t*)g, rows, inner / 8, block / 8, ldh, ldg);
  CTCLIP_CHECK_LAUNCH();
}

int ctclip_geglu_bwd(const void* dg, const void* h, void* dh, long rows, int inner, int block, long lddg, long ldh,
                     void* stream) {
  if (rows <= 0) return 0;
  if ((inner & 7) || (ldh & 7) || (lddg & 7) || block <= 0 || (block & 7) || (inner % block)) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(geglu_bwd_kernel, dim3(grid_for(rows * (inner / 8))), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)dg, (const bf16_t*)h, (bf16_t*)dh, rows, inner / 8, block / 8, lddg, ldh);
  CTCLIP_CHECK_LAUNCH();
}

int ctclip_gelu_fwd(const void* h, void* m, long n, void* stream) {
  if (n <= 0) return 0;
  if (n & 7) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(gelu_fwd_kernel, dim3(grid_for(n / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)h,
                     (bf16_t*)m, n / 8);
  CTCLIP_CHECK_LAUNCH();
}

int ctclip_gelu_bwd(const void* dm, const void* h, void* dh, long n, void* stream) {
  if (n <= 0) return 0;
  if (n & 7) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(gelu_bwd_kernel, dim3(grid_for(n / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dm,
                     (const bf16_t*)h, (bf16_t*)dh, n / 8);
  CTCLIP_CHECK_LAUNCH();
}

int ctclip_swap_middle_f32(const float* in, float* out, long B, int A, int C, int D, void* stream) {
  if (B * A * C <= 0) return 0;
  if (D & 3) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(swap_middle_kernel, dim3(grid_for(B * A * C * (D / 4))), dim3(256), 0, (hipStream_t)stream, in, out,
                     B, A, C, D / 4);
  CTCLIP_CHECK_LAUNCH();
}

int ctclip_mean_mid_fwd(const float* x, void* y_bf16, float* y_f32, long B, int T, long F, void* stream) {
  if (B * F <= 0) return 0;
  if (F & 3) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(mean_mid_kernel, dim3(grid_for(B * (F / 4))), dim3(256), 0, (hipStream_t)stream, x, (bf16_t*)y_bf16,
                     y_f32, B, T, F / 4);
  CTCLIP_CHECK_LAUNCH();
}

int ctclip_mean_mid_bwd(const float* dy, float* dx, long B, int T, long F, void* stream) {
  if (B * F <= 0) return 0;
  if (F & 3) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(mean_mid_bwd_kernel, dim3(grid_for(B * T * (F / 4))), dim3(256), 0, (hipStream_t)stream, dy, dx, B,
                     T, F / 4);
  CTCLIP_CHECK_LAUNCH();
}

int ctclip_add_f32(const float* a, const float* b, float* y, void* y_bf16, long n, void* stream) {
  if (n <= 0) return 0;
  if (n & 3) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(add_f32_kernel, dim3(grid_for(n / 4)), dim3(256), 0, (hipStream_t)stream, a, b, y, (bf16_t*)y_bf16,
                     n / 4);
  CTCLIP_CHECK_LAUNCH();
}

}  // extern "C"

// ---- BERT embeddings: out[r,:] = word[ids[r]] + pos[r % L] + type[tt[r]]  (transformers BertEmbeddings) ----
namespace {
__global__ __launch_bounds__(256) void bert_embed_fwd_kernel(const long* __restrict__ ids, const long* __restrict__ tt,
                                                             const float* __restrict__ word, const float* __restrict__ pos,
                                                             const float* __restrict__ type, float* __restrict__ out,
                                                             long rows, int L, int H4) {
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < rows * H4; e += (long)gridDim.x * 256) {
    const long r = e / H4;
    const int c = (int)(e % H4);
    const float4 a = ((const float4*)word)[ids[r] * H4 + c];
    const float4 b = ((const float4*)pos)[(r % L) * H4 + c];
    const float4 d = ((const float4*)type)[(tt ? tt[r] : 0) * H4 + c];
    ((float4*)out)[e] = make_float4(a.x + b.x + d.x, a.y + b.y + d.y, a.z + b.z + d.z, a.w + b.w + d.w);
  }
}
// Backward of the three embedding tables, without atomics (every sum has one owner and a fixed order, so the gradients are
// bit-reproducible; the f32-atomic scatter this replaces ran at the chip's ~1.3 TB/s atomic rate):
//   d(pos)[p]   += sum over the sequences b, in order, of dy[b L + p]                      one thread per (p, 4 columns)
//   d(type)[v]  += sum of dy[r] over the rows with token_type v                            chunk partials + ordered sum
//   d(word)[id] += sum of dy[r] over the rows with ids[r] == id, in row order              one workgroup per row: the FIRST
//       row of every id (first[id], an integer atomic min) owns the sum and walks the rows behind it (bert_embed_bwd_word_kernel)
__global__ __launch_bounds__(256) void bert_embed_bwd_pos_kernel(const float* __restrict__ dy, float* __restrict__ dpos,
                                                                 long rows, int L, int H4) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long)L * H4) return;
  const int p = (int)(e / H4), c = (int)(e % H4);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (long r = p; r < rows; r += L) {
    const float4 g = ((const float4*)dy)[r * H4 + c];
    acc.x += g.x; acc.y += g.y; acc.z += g.z; acc.w += g.w;
  }
  float4* o = (float4*)dpos + e;
  float4 v = *o;
  v.x += acc.x; v.y += acc.y; v.z += acc.z; v.w += acc.w;
  *o = v;
}

constexpr int EMB_NT = 4;          // token types handled per launch (BERT has 2)
__global__ __launch_bounds__(256) void bert_embed_bwd_type_kernel(const long* __restrict__ tt, const float* __restrict__ dy,
                                                                  float* __restrict__ partials, long rows, int H, int ntypes,
                                                                  int rows_per_chunk) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  const long r0 = (long)blockIdx.y * rows_per_chunk, r1 = (r0 + rows_per_chunk < rows) ? r0 + rows_per_chunk : rows;
  if (c >= H) return;
  float acc[EMB_NT];
#pragma unroll
  for (int v = 0; v < EMB_NT; ++v) acc[v] = 0.f;
  for (long r = r0; r < r1; ++r) {
    const int v = tt ? (int)tt[r] : 0;
    const float g = dy[r * H + c];
#pragma unroll
    for (int u = 0; u < EMB_NT; ++u) acc[u] += (u == v) ? g : 0.f;
  }
  for (int v = 0; v < ntypes; ++v) partials[((long)blockIdx.y * ntypes + v) * H + c] = acc[v];
}

// first[id] = the first row that holds id (integer atomic min: the result does not depend on the order of arrival)
__global__ __launch_bounds__(256) void bert_embed_first_fill_kernel(int* __restrict__ first, long vocab) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < vocab) first[i] = 0x7fffffff;
}
__global__ __launch_bounds__(256) void bert_embed_first_kernel(const long* __restrict__ ids, int* __restrict__ first, int rows,
                                                               long vocab) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= rows) return;
  const long id = ids[r];
  if (id >= 0 && id < vocab) atomicMin(first + id, r);
}
// One workgroup per row; the workgroup of the FIRST row of an id owns that id's sum.  It walks the rows behind it 256 at a time,
// collects the rows with its id in an LDS window (ascending) and, whenever the window is full, adds their dy rows to register
// accumulators in that order: any number of rows, any number of repeats, bit-reproducible.  Cost: (distinct ids) x (rows behind
// their first occurrence) id reads from L2 -- 45 056 rows of random ids (88 reports x 512 tokens) read ~4 GB, ~1 ms.
constexpr int EMB_WIN = 2048, EMB_NC = 8;            // window entries; column blocks of 256 per pass (hidden <= 2048 in one pass)
__global__ __launch_bounds__(256) void bert_embed_bwd_word_kernel(const long* __restrict__ ids, const int* __restrict__ first,
                                                                  const float* __restrict__ dy, float* __restrict__ dword, int rows,
                                                                  int H, long vocab) {
  __shared__ int win[EMB_WIN];
  __shared__ int wave_cnt[4], win_len;
  const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long id = ids[r];
  if (id < 0 || id >= vocab || first[id] != r) return;      // (uniform) an earlier row owns this id
  for (int c0 = 0; c0 < H; c0 += 256 * EMB_NC) {             // one pass for hidden <= 2048
    float acc[EMB_NC];
#pragma unroll
    for (int q = 0; q < EMB_NC; ++q) acc[q] = 0.f;
    auto flush = [&](int n) {
      for (int i = 0; i < n; ++i) {
        const float* row = dy + (long)win[i] * H + c0 + tid;
#pragma unroll
        for (int q = 0; q < EMB_NC; ++q)
          if (c0 + tid + 256 * q < H) acc[q] += row[256 * q];
      }
    };
    if (tid == 0) win_len = 0;
    __syncthreads();
    for (int base = r; base < rows; base += 256) {
      const int j = base + tid;
      const bool hit = j < rows && ids[j] == id;
      const unsigned long long m = __ballot(hit);
      if (lane == 0) wave_cnt[wave] = __popcll(m);
      __syncthreads();
      int off = win_len;
      for (int w = 0; w < wave; ++w) off += wave_cnt[w];
      if (hit) win[off + __popcll(m & ((1ull << lane) - 1ull))] = j;
      __syncthreads();
      const int n = win_len + wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
      __syncthreads();
      if (n + 256 > EMB_WIN) {                                // the next 256 rows might not fit: empty the window
        flush(n);
        __syncthreads();
        if (tid == 0) win_len = 0;
      } else if (tid == 0) {
        win_len = n;
      }
      __syncthreads();
    }
    flush(win_len);
#pragma unroll
    for (int q = 0; q < EMB_NC; ++q)
      if (c0 + tid + 256 * q < H) dword[id * H + c0 + tid + 256 * q] += acc[q];
    __syncthreads();
  }
}
}  // namespace

extern "C" {
int ctclip_bert_embed_fwd(const long* ids, const long* token_type, const float* word, const float* pos, const float* type,
                          float* out, long rows, int L, int hidden, void* stream) {
  if (rows <= 0) return 0;
  if (hidden & 3) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(bert_embed_fwd_kernel, dim3(grid_for(rows * (hidden / 4))), dim3(256), 0, (hipStream_t)stream, ids,
                     token_type, word, pos, type, out, rows, L, hidden / 4);
  CTCLIP_CHECK_LAUNCH();
}
int ctclip_bert_embed_bwd(const long* ids, const long* token_type, const float* dy, float* dword, float* dpos,
                          float* dtype, long rows, int L, int hidden, int type_vocab, long vocab, float* partials,
                          void* stream) {
  if (rows <= 0) return 0;
  // everything is validated before the first launch: nothing has been added to any gradient when an error is returned
  if ((hidden & 3) || L <= 0 || rows % L || type_vocab < 1 || type_vocab > EMB_NT || !partials || rows > (1L << 30) ||
      vocab < 1 || vocab > kPartialsFloats)
    return (int)hipErrorInvalidValue;
  long maxchunks = kPartialsFloats / ((long)type_vocab * hidden);
  if (maxchunks > 64) maxchunks = 64;
  if (maxchunks < 1) return (int)hipErrorInvalidValue;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(bert_embed_bwd_pos_kernel, dim3((unsigned)(((long)L * (hidden / 4) + 255) / 256)), dim3(256), 0, st, dy,
                     dpos, rows, L, hidden / 4);
  long rpc = (rows + maxchunks - 1) / maxchunks;
  if (rpc < 32) rpc = 32;
  const int nchunks = (int)((rows + rpc - 1) / rpc);
  hipLaunchKernelGGL(bert_embed_bwd_type_kernel, dim3((hidden + 255) / 256, nchunks), dim3(256), 0, st, token_type, dy, partials,
                     rows, hidden, type_vocab, (int)rpc);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return (int)e;
  if (int r = ctclip_reduce_partials(partials, nchunks, (long)type_vocab * hidden, type_vocab * hidden, dtype, st)) return r;
  // d(word): the scratch now holds first[vocab] (the token-type partials above have been consumed in stream order)
  int* first = (int*)partials;
  hipLaunchKernelGGL(bert_embed_first_fill_kernel, dim3((unsigned)((vocab + 255) / 256)), dim3(256), 0, st, first, vocab);
  hipLaunchKernelGGL(bert_embed_first_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, ids, first, (int)rows, vocab);
  hipLaunchKernelGGL(bert_embed_bwd_word_kernel, dim3((unsigned)rows), dim3(256), 0, st, ids, first, dy, dword, (int)rows, hidden,
                     vocab);
  CTCLIP_CHECK_LAUNCH();
}
}

// ---- dropout of the text encoder (transformers BertSelfAttention / BertSelfOutput / BertOutput in train mode) ----
// Keep flags are a pure function of (seed, element index): a counter-based generator (Widynski's "squares", four rounds of
// square-and-rotate on a 64-bit counter x key) gives every element its own 32-bit draw, keep <=> draw >= p 2^32.  Nothing
// is stored for the hidden-state dropouts: forward and backward evaluate the same function.
namespace {
__device__ __forceinline__ uint32_t draw32(uint64_t ctr, uint64_t key) {
  uint64_t x = ctr * key, y = x, z = y + key;
  x = x * x + y; x = (x >> 32) | (x << 32);
  x = x * x + z; x = (x >> 32) | (x << 32);
  x = x * x + y; x = (x >> 32) | (x << 32);
  return (uint32_t)((x * x + z) >> 32);
}
// splitmix64 of the caller's seed, forced odd with mixed upper bits: the multiplier key of the generator
inline uint64_t key_of(uint64_t seed) {
  uint64_t z = seed + 0x9e3779b97f4a7c15ull;
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  z ^= z >> 31;
  return z | 0x0100000000000001ull;
}
inline uint32_t threshold_of(float p) {
  const double t = (double)p * 4294967296.0;
  return t >= 4294967295.0 ? 0xffffffffu : (uint32_t)t;
}

__global__ __launch_bounds__(256) void dropout_keep_kernel(uint8_t* __restrict__ keep, long n, uint32_t thr, uint64_t key,
                                                           uint64_t off) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
    keep[i] = draw32(off + (uint64_t)i, key) >= thr ? 1 : 0;
}
// out = x + keep * branch / (1 - p)
__global__ __launch_bounds__(256) void dropout_add_kernel(const float* __restrict__ x, const float* __restrict__ br,
                                                          float* __restrict__ out, long n, uint32_t thr, float inv_keep,
                                                          uint64_t key, uint64_t off) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float b = draw32(off + (uint64_t)i, key) >= thr ? br[i] * inv_keep : 0.f;
    out[i] = x[i] + b;
  }
}
// d = g * keep / (1 - p), with an optional bf16 mirror
__global__ __launch_bounds__(256) void dropout_bwd_kernel(const float* __restrict__ g, float* __restrict__ d,
                                                          bf16_t* __restrict__ d16, long n, uint32_t thr, float inv_keep,
                                                          uint64_t key, uint64_t off) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float v = draw32(off + (uint64_t)i, key) >= thr ? g[i] * inv_keep : 0.f;
    if (d) d[i] = v;
    if (d16) d16[i] = f32_to_bf16(v);
  }
}
}  // namespace

extern "C" {
int ctclip_dropout_keep(uint8_t* keep, long n, float p, long seed, long offset, void* stream) {
  if (n <= 0) return 0;
  if (!(p >= 0.f && p < 1.f)) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(dropout_keep_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, keep, n, threshold_of(p),
                     key_of((uint64_t)seed), (uint64_t)offset);
  CTCLIP_CHECK_LAUNCH();
}
int ctclip_dropout_add(const float* x, const float* branch, float* out, long n, float p, long seed, long offset,
                       void* stream) {
  if (n <= 0) return 0;
  if (!(p >= 0.f && p < 1.f)) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(dropout_add_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, branch, out, n,
                     threshold_of(p), 1.0f / (1.0f - p), key_of((uint64_t)seed), (uint64_t)offset);
  CTCLIP_CHECK_LAUNCH();
}
int ctclip_dropout_bwd(const float* g, float* d, void* d_bf16, long n, float p, long seed, long offset, void* stream) {
  if (n <= 0) return 0;
  if (!(p >= 0.f && p < 1.f)) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(dropout_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, g, d, (bf16_t*)d_bf16, n,
                     threshold_of(p), 1.0f / (1.0f - p), key_of((uint64_t)seed), (uint64_t)offset);
  CTCLIP_CHECK_LAUNCH();
}
}

// ---- FF1 weight gradient: rows of the product come in the [value blk | gate blk | ...] block order of the GEGLU weight
// shadow; dw [2 inner, dim] += them in the reference's order (value rows 0 .. inner - 1, gate rows inner .. 2 inner - 1).
namespace {
__global__ __launch_bounds__(256) void geglu_wgrad_unblock_kernel(const float4* __restrict__ gp, float4* __restrict__ dw, int inner,
                                                                  int blk, int d4) {
  const long total = 2L * inner * d4;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int c = (int)(e % d4);
    const int r = (int)(e / d4);                       // reference row
    const int gate = r >= inner, q = gate ? r - inner : r;
    const long src = ((long)(q / blk) * 2 + gate) * blk + q % blk;
    const float4 a = gp[src * d4 + c];
    float4 b = dw[e];
    b.x += a.x; b.y += a.y; b.z += a.z; b.w += a.w;
    dw[e] = b;
  }
}
}  // namespace

extern "C" int ctclip_geglu_wgrad_unblock(const float* g_blocked, float* dw, int inner, int block, int dim, void* stream) {
  if (inner <= 0 || dim <= 0) return 0;
  if (block <= 0 || (dim & 3) || ((((uintptr_t)g_blocked) | ((uintptr_t)dw)) & 15)) return (int)hipErrorInvalidValue;
  const long total = 2L * inner * (dim / 4);
  hipLaunchKernelGGL(geglu_wgrad_unblock_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const float4*)g_blocked,
                     (float4*)dw, inner, block, dim / 4);
  CTCLIP_CHECK_LAUNCH();
}

// ---- weight shadows: every kernel-layout copy of the model's parameters in ONE launch per optimiser step ------------------------
// The GEMM kernels read bf16 (and transposed, zero-padded, block-interleaved, gamma-scaled) copies of the f32 master weights;
// rebuilding them with framework ops took ~180 small launches per step.  Here a table of descriptors -- written once, when
// the model is built -- drives one kernel: descriptor d copies a [rows, cols] source matrix (f32 or bf16) into its destination
//   dst[(map(r)) * dst_ld + c]        or, transposed,   dst[c * dst_ld + map(r)]
//   map(r) = r, or with blk > 0  (r / blk) * 2 blk + r % blk     (the [val blk | gate blk | ...] interleave of the GEGLU weight:
//                                                                  the gate half's dst pointer starts blk rows / columns in)
// multiplied by scale[c] when a column scale is given (a LayerNorm's gamma folded into the projection behind it), as bf16 or
// f32; ROWSUM instead writes dst[r] = sum_c bf16(src[r][c] scale[c]) (f32).  A workgroup takes one 32 x 64 tile; tile_start[d] is
// the first tile of descriptor d.
namespace {
struct ShadowDesc { const void* src; void* dst; long rows, cols, src_ld, dst_ld, flags; const float* scale; };
constexpr long SH_T = 1, SH_SRC16 = 2, SH_DST32 = 4, SH_ROWSUM = 8;

__global__ __launch_bounds__(256) void shadow_multi_kernel(const ShadowDesc* __restrict__ table, const int* __restrict__ tile_start,
                                                           int ndesc) {
  __shared__ float tile[32][65];
  __shared__ float rsum[32][9];
  int lo = 0, hi = ndesc - 1;                         // last descriptor whose first tile is <= this workgroup's
  const int me = blockIdx.x;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (tile_start[mid] <= me) lo = mid; else hi = mid - 1;
  }
  const ShadowDesc d = table[lo];
  const int t = me - tile_start[lo];
  const int blk = (int)(d.flags >> 8);
  const bool rowsum = d.flags & SH_ROWSUM;
  const long ctiles = rowsum ? 1 : (d.cols + 63) / 64;
  const long r0 = (t / ctiles) * 32, c0 = (t % ctiles) * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;       // 64 columns x 4 rows per pass
  auto ld = [&](long r, long c) -> float {
    float v = (d.flags & SH_SRC16) ? bf16_to_f32(((const bf16_t*)d.src)[r * d.src_ld + c]) : ((const float*)d.src)[r * d.src_ld + c];
    if (d.scale) v *= d.scale[c];
    return v;
  };
  auto rmap = [&](long r) -> long { return blk ? (r / blk) * 2 * blk + r % blk : r; };
  if (rowsum) {
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.f;
    for (long c = tx; c < d.cols; c += 64)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const long r = r0 + ty * 8 + i;
        if (r < d.rows) acc[i] += bf16_to_f32(f32_to_bf16(ld(r, c)));
      }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float s_ = wave_sum(acc[i]);
      const long r = r0 + ty * 8 + i;
      if (tx == 0 && r < d.rows) ((float*)d.dst)[r] = s_;
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const long r = r0 + ty + 4 * i, c = c0 + tx;
    tile[ty + 4 * i][tx] = (r < d.rows && c < d.cols) ? ld(r, c) : 0.f;
  }
  __syncthreads();
  if (!(d.flags & SH_T)) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const long r = r0 + ty + 4 * i, c = c0 + tx;
      if (r < d.rows && c < d.cols) {
        const long o = rmap(r) * d.dst_ld + c;
        if (d.flags & SH_DST32) ((float*)d.dst)[o] = tile[ty + 4 * i][tx];
        else ((bf16_t*)d.dst)[o] = f32_to_bf16(tile[ty + 4 * i][tx]);
      }
    }
  } else {
    const int rx = threadIdx.x & 31, cy = threadIdx.x >> 5;     // 32 source rows (contiguous in dst) x 8 source columns per pass
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const long r = r0 + rx, c = c0 + cy + 8 * i;
      if (r < d.rows && c < d.cols) {
        const long o = c * d.dst_ld + rmap(r);
        if (d.flags & SH_DST32) ((float*)d.dst)[o] = tile[rx][cy + 8 * i];
        else ((bf16_t*)d.dst)[o] = f32_to_bf16(tile[rx][cy + 8 * i]);
      }
    }
  }
}
}  // namespace

extern "C" int ctclip_shadow_multi(const void* table, const int* tile_start, int ndesc, int total_tiles, void* stream) {
  if (ndesc <= 0 || total_tiles <= 0) return 0;
  hipLaunchKernelGGL(shadow_multi_kernel, dim3((unsigned)total_tiles), dim3(256), 0, (hipStream_t)stream, (const ShadowDesc*)table,
                     tile_start, ndesc);
  CTCLIP_CHECK_LAUNCH();
}
