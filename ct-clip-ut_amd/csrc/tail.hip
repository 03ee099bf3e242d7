// Small f32 pieces of the step that must not lose precision: a generic f32 GEMM (position-bias MLP,
// latent projections, logits), the symmetric InfoNCE loss, relative-position bias expansion, and the
// fused grad-clip + Adam update.
#include "common.h"

namespace {

// ------------------------------------------------------------------------------------------------
// f32 GEMM, 64x64x16 tiles, 256 threads x (4x4) outputs.  Same operand-layout flags as the bf16 GEMM.
// act: 0 none, 2 leaky_relu(slope), 3 multiply by leaky_relu'(aux) (backward of act 2).
// ------------------------------------------------------------------------------------------------
struct SGemm {
  const float* A; const float* B; float* C; const float* bias; const float* aux; const float* alpha_dev;
  long lda, ldb, ldc, ldaux;
  int M, N, K, a_kmajor, b_kmajor, act, accumulate, alpha_exp;
  float alpha, slope;
};

__global__ __launch_bounds__(256) void sgemm_kernel(SGemm g) {
  __shared__ float As[16][65], Bs[16][65];
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int row0 = blockIdx.y * 64, col0 = blockIdx.x * 64;
  float acc[4][4] = {};
  for (int k0 = 0; k0 < g.K; k0 += 16) {
    for (int e = tid; e < 64 * 16; e += 256) {
      int r, k;
      if (g.a_kmajor) { r = e >> 4; k = e & 15; } else { k = e >> 6; r = e & 63; }
      float v = 0.f;
      if (row0 + r < g.M && k0 + k < g.K)
        v = g.a_kmajor ? g.A[(long)(row0 + r) * g.lda + k0 + k] : g.A[(long)(k0 + k) * g.lda + row0 + r];
      As[k][r] = v;
      if (g.b_kmajor) { r = e >> 4; k = e & 15; } else { k = e >> 6; r = e & 63; }
      v = 0.f;
      if (col0 + r < g.N && k0 + k < g.K)
        v = g.b_kmajor ? g.B[(long)(col0 + r) * g.ldb + k0 + k] : g.B[(long)(k0 + k) * g.ldb + col0 + r];
      Bs[k][r] = v;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      float a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { a[i] = As[k][ty * 4 + i]; b[i] = Bs[k][tx * 4 + i]; }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] += a[i] * b[j];
    }
    __syncthreads();
  }
  float alpha = g.alpha;
  if (g.alpha_dev) alpha *= g.alpha_exp ? __expf(*g.alpha_dev) : *g.alpha_dev;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = row0 + ty * 4 + i, c = col0 + tx * 4 + j;
      if (r >= g.M || c >= g.N) continue;
      float v = acc[i][j] * alpha + (g.bias ? g.bias[c] : 0.f);
      if (g.act == 2) v = v > 0.f ? v : v * g.slope;
      if (g.act == 3) v *= (g.aux[(long)r * g.ldaux + c] > 0.f) ? 1.f : g.slope;
      float* dst = g.C + (long)r * g.ldc + c;
      *dst = g.accumulate ? (*dst + v) : v;
    }
}

// ------------------------------------------------------------------------------------------------
// symmetric InfoNCE (CTClipTrainer.py:164-175 with targets = arange): one workgroup, G <= 4096
// loss = 0.5 * (mean_i (lse_row_i - s_ii) + mean_j (lse_col_j - s_jj)); dsim = dloss/dsim
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void infonce_kernel(const float* __restrict__ sim, float* __restrict__ loss,
                                                       float* __restrict__ dsim, int G, float* __restrict__ ws) {
  float* row_lse = ws;       // [G]
  float* col_lse = ws + G;   // [G]
  __shared__ float red[32];
  const int tid = threadIdx.x, nt = blockDim.x;
  for (int i = tid; i < G; i += nt) {
    float m = -INFINITY, mc = -INFINITY;
    for (int j = 0; j < G; ++j) { m = fmaxf(m, sim[(long)i * G + j]); mc = fmaxf(mc, sim[(long)j * G + i]); }
    float s = 0.f, sc = 0.f;
    for (int j = 0; j < G; ++j) { s += expf(sim[(long)i * G + j] - m); sc += expf(sim[(long)j * G + i] - mc); }
    row_lse[i] = m + logf(s);
    col_lse[i] = mc + logf(sc);
  }
  __syncthreads();
  float part = 0.f;
  for (int i = tid; i < G; i += nt) part += (row_lse[i] - sim[(long)i * G + i]) + (col_lse[i] - sim[(long)i * G + i]);
  part = wave_sum(part);
  if ((tid & 63) == 0) red[tid >> 6] = part;
  __syncthreads();
  if (tid == 0) {
    float t = 0.f;
    for (int w = 0; w < (nt + 63) / 64; ++w) t += red[w];
    *loss = 0.5f * t / (float)G;
  }
  const float k = 0.5f / (float)G;
  for (long e = tid; e < (long)G * G; e += nt) {
    const int i = (int)(e / G), j = (int)(e % G);
    const float s = sim[e];
    dsim[e] = k * (expf(s - row_lse[i]) + expf(s - col_lse[j]) - (i == j ? 2.f : 0.f));
  }
}

// bias[h][i][j] = table[relidx[i][j]][h]    (table is the [R, heads] output of the position MLP)
__global__ __launch_bounds__(256) void bias_expand_kernel(const float* __restrict__ table, const uint16_t* __restrict__ relidx,
                                                          float* __restrict__ bias, int heads, long nn) {
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < nn * heads; e += (long)gridDim.x * 256) {
    const int h = (int)(e / nn);
    const long ij = e % nn;
    bias[e] = table[(long)relidx[ij] * heads + h];
  }
}

__global__ __launch_bounds__(256) void scale_by_dev_kernel(const float* __restrict__ x, const float* __restrict__ s,
                                                           float* __restrict__ y, long n) {
  const float k = *s;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) y[e] = x[e] * k;
}

// one value per workgroup, summed in a fixed order: lanes by butterfly, the four waves left to right
__device__ __forceinline__ void block_partial(float s, float* __restrict__ partials) {
  __shared__ float wsum[4];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}

// partials[block] = sum(a * b) over the block's grid-stride share (d temperature = sum(dsim * sim), CTCLIP tail backward)
__global__ __launch_bounds__(256) void dot_accum_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                        float* __restrict__ partials, long n) {
  float s = 0.f;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) s += a[e] * b[e];
  block_partial(s, partials);
}

// ------------------------------------------------------------------------------------------------
// optimiser: global grad-norm + clip + Adam/AdamW in one pass over flat f32 arenas
// (CTClipTrainer.py:199-202, optimizer.py:42-54).  The clip coefficient is computed on device from the
// squared norm so the host never synchronises.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, long n4, long n, float* __restrict__ partials) {
  float s = 0.f;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n4; e += (long)gridDim.x * 256) {
    const float4 v = ((const float4*)g)[e];
    s += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
  }
  if (blockIdx.x == 0)
    for (long e = n4 * 4 + threadIdx.x; e < n; e += 256) s += g[e] * g[e];
  block_partial(s, partials);
}

struct AdamArgs {
  float* p; const float* g; float* m; float* v; bf16_t* p16;
  long n;
  float lr, b1, b2, eps, wd, bc1, bc2_sqrt, max_norm;
  int decoupled;
  const float* gnorm_sq;
};

__global__ __launch_bounds__(256) void adam_kernel(AdamArgs a) {
  float coef = 1.f;
  if (a.gnorm_sq && a.max_norm > 0.f) coef = fminf(1.f, a.max_norm / (sqrtf(*a.gnorm_sq) + 1e-6f));
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < a.n; e += (long)gridDim.x * 256) {
    float p = a.p[e], g = a.g[e] * coef;
    if (a.wd != 0.f) {
      if (a.decoupled) p *= (1.f - a.lr * a.wd);
      else g += a.wd * p;
    }
    const float m = a.b1 * a.m[e] + (1.f - a.b1) * g;
    const float v = a.b2 * a.v[e] + (1.f - a.b2) * g * g;
    const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
    p -= (a.lr / a.bc1) * (m / denom);
    a.p[e] = p; a.m[e] = m; a.v[e] = v;
    if (a.p16) a.p16[e] = f32_to_bf16(p);
  }
}

// partials[chunk][c] = sum of x[r][c] over the chunk's rows: a workgroup covers 64 columns x rows_per_block rows, 4 row-lanes
// per column combined through LDS in a fixed order; ctclip_reduce_partials adds the chunks up in chunk order.
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ x, long rows, int cols, long ld,
                                                     float* __restrict__ partials, int rows_per_block) {
  __shared__ float red[4][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const long r0 = (long)blockIdx.y * rows_per_block;
  const long r1 = (r0 + rows_per_block < rows) ? r0 + rows_per_block : rows;
  float s = 0.f;
  if (c < cols)
    for (long r = r0 + rl; r < r1; r += 4) {
      if (sizeof(T) == 2) s += bf16_to_f32(((const bf16_t*)x)[r * ld + c]);
      else s += ((const float*)x)[r * ld + c];
    }
  red[rl][cl] = s;
  __syncthreads();
  if (rl == 0 && c < cols) partials[(long)blockIdx.y * cols + c] = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
}

// ------------------------------------------------------------------------------------------------
// Second stage of every small reduction of the backward pass (column sums, LayerNorm / tubelet-norm d(gamma) d(beta),
// head-norm d(scale), PEG d(taps), squared gradient norm): out[c] += sum over parts of partials[part * ld + c], parts
// taken in index order by LANES interleaved row lanes whose totals are combined in a fixed tree -- the result does not
// depend on which workgroup finished first, so two runs of the same step give the same bits (the reference asks for
// torch.use_deterministic_algorithms(True) in its attribution code, src/utils/visualizations.py:29-39).
// ------------------------------------------------------------------------------------------------
template <int CPB>                                        // columns per workgroup: 64 (x 4 row lanes), 16 (x 16) or 1 (x 256)
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ partials, int nparts, long ld,
                                                              int width, float* __restrict__ out) {
  constexpr int LANES = 256 / CPB;
  __shared__ float red[LANES][CPB + 1];
  const int cl = threadIdx.x % CPB, rl = threadIdx.x / CPB;
  const int c = blockIdx.x * CPB + cl;
  // four independent chains per row lane (rows rl, rl + LANES, ... taken round-robin): the loads of a chain do not wait for
  // each other's adds -- the kernel is pure latency otherwise (a few microseconds for a sum of a few megabytes, ~190 of them
  // per training step); the association ((a0 + a1) + (a2 + a3)) is fixed, so the result stays reproducible
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (c < width) {
    const float* p = partials + c;
    int b = rl;
    for (; b + 3 * LANES < nparts; b += 4 * LANES) {
      a0 += p[(long)b * ld];
      a1 += p[(long)(b + LANES) * ld];
      a2 += p[(long)(b + 2 * LANES) * ld];
      a3 += p[(long)(b + 3 * LANES) * ld];
    }
    for (; b < nparts; b += LANES) a0 += p[(long)b * ld];
  }
  red[rl][cl] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  for (int half = LANES / 2; half > 0; half >>= 1) {      // fixed pairing: lane r += lane r + half
    if (rl < half) red[rl][cl] += red[rl + half][cl];
    __syncthreads();
  }
  if (rl == 0 && c < width) out[c] += red[0][cl];
}

__global__ __launch_bounds__(256) void leaky_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ act,
                                                        float* __restrict__ y, long n, float slope) {
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256)
    y[e] = dy[e] * (act[e] > 0.f ? 1.f : slope);
}

inline unsigned grid_for(long work, long cap = 2048) {
  long b = (work + 255) / 256;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (unsigned)b;
}

}  // namespace

int ctclip_reduce_partials(const float* partials, int nparts, long ld, int width, float* out, hipStream_t st) {
  if (nparts <= 0 || width <= 0) return 0;
  if (width >= 16384)                                      // split-K partial products: wide rows, 256-byte pieces per row lane
    hipLaunchKernelGGL(reduce_partials_kernel<64>, dim3((width + 63) / 64), dim3(256), 0, st, partials, nparts, ld, width, out);
  else if (width >= 16)
    hipLaunchKernelGGL(reduce_partials_kernel<16>, dim3((width + 15) / 16), dim3(256), 0, st, partials, nparts, ld, width, out);
  else
    hipLaunchKernelGGL(reduce_partials_kernel<1>, dim3(width), dim3(256), 0, st, partials, nparts, ld, width, out);
  return (int)hipGetLastError();
}

extern "C" {

int ctclip_gemm_f32(const float* A, const float* B, float* C, const float* bias, const float* aux, int M, int N, int K,
                    long lda, long ldb, long ldc, long ldaux, int a_kmajor, int b_kmajor, float alpha,
                    const float* alpha_dev, int alpha_exp, int act, float slope, int accumulate, void* stream) {
  if (M <= 0 || N <= 0) return 0;
  SGemm g{A, B, C, bias, aux, alpha_dev, lda, ldb, ldc, ldaux, M, N, K, a_kmajor, b_kmajor, act, accumulate, alpha_exp,
          alpha, slope};
  hipLaunchKernelGGL(sgemm_kernel, dim3((N + 63) / 64, (M + 63) / 64), dim3(256), 0, (hipStream_t)stream, g);
  CTCLIP_CHECK_LAUNCH();
}

int ctclip_infonce(const float* sim, float* loss, float* dsim, int G, float* workspace, void* stream) {
  if (G <= 0) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(infonce_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, sim, loss, dsim, G, workspace);
  CTCLIP_CHECK_LAUNCH();
}

int ctclip_bias_expand(const float* table, const uint16_t* relidx, float* bias, int heads, int n, void* stream) {
  const long nn = (long)n * n;
  hipLaunchKernelGGL(bias_expand_kernel, dim3(grid_for(nn * heads)), dim3(256), 0, (hipStream_t)stream, table, relidx,
                     bias, heads, nn);
  CTCLIP_CHECK_LAUNCH();
}

int ctclip_scale_by_dev(const float* x, const float* s, float* y, long n, void* stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(scale_by_dev_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, s, y, n);
  CTCLIP_CHECK_LAUNCH();
}

int ctclip_colsum_accum(const void* x, int x_is_bf16, long rows, int cols, long ld, float* out, float* partials, void* stream) {
  if (rows <= 0 || cols <= 0) return 0;
  if (!partials || cols > 8192) return (int)hipErrorInvalidValue;
  // at most 256 row chunks (and CTCLIP_PARTIALS_FLOATS / cols), at least 128 rows each
  long maxchunks = kPartialsFloats / cols;
  if (maxchunks > 256) maxchunks = 256;
  long rpb = (rows + maxchunks - 1) / maxchunks;
  if (rpb < 128) rpb = 128;
  rpb = (rpb + 3) / 4 * 4;
  const int nchunks = (int)((rows + rpb - 1) / rpb);
  dim3 grid((cols + 63) / 64, (unsigned)nchunks);
  if (x_is_bf16)
    hipLaunchKernelGGL(colsum_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, rows, cols, ld, partials, (int)rpb);
  else
    hipLaunchKernelGGL(colsum_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)x, rows, cols, ld, partials, (int)rpb);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return (int)e;
  return ctclip_reduce_partials(partials, nchunks, cols, cols, out, (hipStream_t)stream);
}

int ctclip_leaky_bwd(const float* dy, const float* act, float* y, long n, float slope, void* stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(leaky_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, dy, act, y, n, slope);
  CTCLIP_CHECK_LAUNCH();
}

int ctclip_dot_accum(const float* a, const float* b, float* out, long n, float* partials, void* stream) {
  if (n <= 0) return 0;
  if (!partials) return (int)hipErrorInvalidValue;
  const unsigned nb = grid_for(n, 256);
  hipLaunchKernelGGL(dot_accum_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, a, b, partials, n);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return (int)e;
  return ctclip_reduce_partials(partials, (int)nb, 1, 1, out, (hipStream_t)stream);
}

int ctclip_sumsq_accum(const float* g, long n, float* out, float* partials, void* stream) {
  if (n <= 0) return 0;
  if ((((uintptr_t)g) & 15) || !partials) return (int)hipErrorInvalidValue;
  const unsigned nb = grid_for(n / 4 + 1);
  hipLaunchKernelGGL(sumsq_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, g, n / 4, n, partials);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return (int)e;
  return ctclip_reduce_partials(partials, (int)nb, 1, 1, out, (hipStream_t)stream);
}

int ctclip_adam_step(float* p, const float* g, float* m, float* v, void* p_bf16, long n, float lr, float beta1,
                     float beta2, float eps, float weight_decay, int decoupled, float bias_corr1, float bias_corr2,
                     const float* gnorm_sq, float max_norm, void* stream) {
  if (n <= 0) return 0;
  AdamArgs a{p, g, m, v, (bf16_t*)p_bf16, n, lr, beta1, beta2, eps, weight_decay, bias_corr1, sqrtf(bias_corr2),
             max_norm, decoupled, gnorm_sq};
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n, 8192)), dim3(256), 0, (hipStream_t)stream, a);
  CTCLIP_CHECK_LAUNCH();
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------------
// diagnostic: register-resident MFMA loop (no memory traffic) -- what the matrix pipes sustain at the clock the part
// actually holds under this load; bench / DESIGN quote roofline fractions against the datasheet peak, this probe shows
// how much of the gap is power management rather than the kernels.
// ---------------------------------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(512) void mfma_probe_kernel(float* __restrict__ out, int iters) {
  bf16x8 a, b;
#pragma unroll
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.001f * (threadIdx.x + j)); b[j] = (__bf16)(0.002f * (threadIdx.x - j)); }
  f32x16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = mfma32(a, b, acc[i]);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 123.456f) out[0] = s;                   // keep the loop alive
}
}  // namespace

// diagnostic: the HBM rate this part sustains -- the yardstick next to the 8 TB/s datasheet figure for the HBM-bound kernels
// (tubelet gather, PEG, LayerNorm, Adam).  Eight independent 16-byte accesses per lane are in flight before the first is used
// (a one-load-in-flight grid-stride loop read 4.95 TB/s where this repo's own LayerNorm kernels move 5.6-5.7), non-temporal,
// each workgroup walking whole 32 KiB blocks.  MODE 0 copy (read + write bytes), 1 read only, 2 write only.
namespace {
typedef __attribute__((ext_vector_type(4))) float probe_f4;
template <int MODE>
__global__ __launch_bounds__(256) void stream_probe_kernel(const probe_f4* __restrict__ src, probe_f4* __restrict__ dst, long n16,
                                                           float* __restrict__ sink) {
  constexpr int U = 8;
  const long per = 256L * U;                                       // 16-byte elements per workgroup and iteration
  probe_f4 keep = {0.f, 0.f, 0.f, 0.f};
  for (long b0 = (long)blockIdx.x * per; b0 < n16; b0 += (long)gridDim.x * per) {
    probe_f4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long e = b0 + u * 256 + threadIdx.x;
      if (MODE != 2) v[u] = e < n16 ? __builtin_nontemporal_load(src + e) : keep;
      else v[u] = keep;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long e = b0 + u * 256 + threadIdx.x;
      if (MODE != 1) { if (e < n16) __builtin_nontemporal_store(v[u], dst + e); }
      else keep += v[u];
    }
  }
  if (MODE == 1 && keep.x + keep.y + keep.z + keep.w == 123.456f) sink[0] = keep.x;      // keep the loads alive
}
int stream_probe(int mode, const void* src, void* dst, long bytes, hipStream_t st) {
  if (bytes <= 0) return 0;
  if ((bytes & 15) || (((uintptr_t)src | (uintptr_t)dst) & 15)) return (int)hipErrorInvalidValue;
  const long n16 = bytes / 16;
  const long want = (n16 + 2047) / 2048;
  const unsigned grid = (unsigned)(want < 8192 ? want : 8192);
  if (mode == 0) hipLaunchKernelGGL(stream_probe_kernel<0>, dim3(grid), dim3(256), 0, st, (const probe_f4*)src, (probe_f4*)dst, n16, nullptr);
  else if (mode == 1) hipLaunchKernelGGL(stream_probe_kernel<1>, dim3(grid), dim3(256), 0, st, (const probe_f4*)src, nullptr, n16, (float*)dst);
  else hipLaunchKernelGGL(stream_probe_kernel<2>, dim3(grid), dim3(256), 0, st, nullptr, (probe_f4*)dst, n16, nullptr);
  CTCLIP_CHECK_LAUNCH();
}
}  // namespace

extern "C" int ctclip_probe_copy(const void* src, void* dst, long bytes, void* stream) {
  return stream_probe(0, src, dst, bytes, (hipStream_t)stream);
}
/* mode 1: reads `bytes` from src (dst: any 16-byte aligned scratch of >= 4 bytes, written only if a sum happens to match);
 * mode 2: writes `bytes` of zeros to dst (src unused, may alias dst) */
extern "C" int ctclip_probe_stream(const void* src, void* dst, long bytes, int mode, void* stream) {
  if (mode < 0 || mode > 2) return (int)hipErrorInvalidValue;
  return stream_probe(mode, mode == 2 ? dst : src, dst, bytes, (hipStream_t)stream);
}

extern "C" int ctclip_probe_mfma(float* out, int blocks, int iters, void* stream) {
  hipLaunchKernelGGL(mfma_probe_kernel, dim3(blocks), dim3(512), 0, (hipStream_t)stream, out, iters);
  CTCLIP_CHECK_LAUNCH();
}

// diagnostic / test hook: fill the whole LDS of every CU with a 32-bit pattern (LDS is not cleared between kernels: a kernel that
// reads a word it never wrote sees what the previous tenant left -- with two processes on one device, anything).  Tests poison the
// LDS with a NaN pattern in front of kernels that round their blocks up to whole waves (tests/test_hip_kernels.py).
namespace {
__global__ __launch_bounds__(1024) void lds_fill_kernel(uint32_t pattern, int words, uint32_t* sink) {
  extern __shared__ uint32_t lds_fill_smem[];
  for (int i = threadIdx.x; i < words; i += blockDim.x) lds_fill_smem[i] = pattern;
  __syncthreads();
  if (sink && lds_fill_smem[(threadIdx.x * 97) % words] != pattern) sink[0] = 1;    // keeps the stores alive
}
}  // namespace
extern "C" int ctclip_probe_lds_fill(int pattern, void* sink, void* stream) {
  constexpr int bytes = 160 * 1024;
  CTCLIP_LDS_LIMIT_ONCE(lds_fill_kernel, bytes);
  // one 160 KiB workgroup per CU at a time; several rounds so that every CU is visited whatever else is resident
  hipLaunchKernelGGL(lds_fill_kernel, dim3(ctclip_cu_count8() * 4), dim3(1024), bytes, (hipStream_t)stream, (uint32_t)pattern,
                     bytes / 4, (uint32_t*)sink);
  CTCLIP_CHECK_LAUNCH();
}

