// PEG: depthwise causal 3x3x3 convolution over a channels-last (b,t,h,w,d) token grid, fused with the
// residual add.  reference src/utils/attention.py:55-83 and the `peg(x) + x` of :325.
//
// HBM-bound (0.38 GFLOP per volume against 2 x 28 MB of f32 traffic): the kernels never permute to
// channels-first and never materialise the padded tensor -- taps are predicated.  The kernel works on the
// MEMORY order of the token tensor, which is exactly what the reference does (attention.py:69 reshapes
// flat memory to (b,t,h,w,d), also for the temporal transformer whose tokens are ordered (b h w) t).
//
// Weights are passed tap-major: w27[tap][d], tap = (kt*3 + kh)*3 + kw  (a transposed copy of
// dsconv.weight[d,1,3,3,3]) so that a wave reads one coalesced row per tap.
#include "common.h"

namespace {

struct Grid5 { long B; int T, H, W, d4; };

__device__ __forceinline__ void fma4(float4& a, const float4& w, const float4& x) {
  a.x += w.x * x.x; a.y += w.y * x.y; a.z += w.z * x.z; a.w += w.w * x.w;
}

// y = x + bias + conv(x)                 out[t] uses x[t + kt - 2], x[h + kh - 1], x[w + kw - 1]
__global__ __launch_bounds__(256) void peg_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w27,
                                                      const float* __restrict__ bias, float* __restrict__ y,
                                                      bf16_t* __restrict__ y16, Grid5 g, int residual) {
  const long total = g.B * g.T * g.H * g.W * g.d4;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int c = (int)(idx % g.d4);
    long pos = idx / g.d4;
    const int w_ = (int)(pos % g.W);
    const int h_ = (int)((pos / g.W) % g.H);
    const int t_ = (int)((pos / ((long)g.W * g.H)) % g.T);
    const long b = pos / ((long)g.W * g.H * g.T);
    float4 acc = residual ? ((const float4*)x)[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 bv = ((const float4*)bias)[c];
    acc.x += bv.x; acc.y += bv.y; acc.z += bv.z; acc.w += bv.w;
#pragma unroll
    for (int kt = 0; kt < 3; ++kt) {
      const int tt = t_ + kt - 2;
      if (tt < 0) continue;
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const int hh = h_ + kh - 1;
        if (hh < 0 || hh >= g.H) continue;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int ww = w_ + kw - 1;
          if (ww < 0 || ww >= g.W) continue;
          const long np = ((b * g.T + tt) * g.H + hh) * g.W + ww;
          fma4(acc, ((const float4*)w27)[((kt * 3 + kh) * 3 + kw) * g.d4 + c], ((const float4*)x)[np * g.d4 + c]);
        }
      }
    }
    if (y) ((float4*)y)[idx] = acc;
    if (y16) {
      uint2 p;
      p.x = pack_bf16x2(acc.x, acc.y);
      p.y = pack_bf16x2(acc.z, acc.w);
      ((uint2*)y16)[idx] = p;
    }
  }
}

// dx = dy + conv^T(dy):  x[pos] fed output (t - kt + 2, h - kh + 1, w - kw + 1) through tap (kt,kh,kw)
__global__ __launch_bounds__(256) void peg_bwd_data_kernel(const float* __restrict__ dy, const float* __restrict__ w27,
                                                           float* __restrict__ dx, bf16_t* __restrict__ dx16, Grid5 g,
                                                           int residual) {
  const long total = g.B * g.T * g.H * g.W * g.d4;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int c = (int)(idx % g.d4);
    long pos = idx / g.d4;
    const int w_ = (int)(pos % g.W);
    const int h_ = (int)((pos / g.W) % g.H);
    const int t_ = (int)((pos / ((long)g.W * g.H)) % g.T);
    const long b = pos / ((long)g.W * g.H * g.T);
    float4 acc = residual ? ((const float4*)dy)[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int kt = 0; kt < 3; ++kt) {
      const int tt = t_ - kt + 2;
      if (tt >= g.T) continue;
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const int hh = h_ - kh + 1;
        if (hh < 0 || hh >= g.H) continue;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int ww = w_ - kw + 1;
          if (ww < 0 || ww >= g.W) continue;
          const long np = ((b * g.T + tt) * g.H + hh) * g.W + ww;
          fma4(acc, ((const float4*)w27)[((kt * 3 + kh) * 3 + kw) * g.d4 + c], ((const float4*)dy)[np * g.d4 + c]);
        }
      }
    }
    if (dx) ((float4*)dx)[idx] = acc;
    if (dx16) {
      uint2 p;
      p.x = pack_bf16x2(acc.x, acc.y);
      p.y = pack_bf16x2(acc.z, acc.w);
      ((uint2*)dx16)[idx] = p;
    }
  }
}

// dw27[tap][c] += sum_pos dy[pos][c] * x[pos + off(tap)][c];  dbias[c] += sum_pos dy[pos][c]
// Each thread owns 4 channels and a strided subset of a position chunk; 28 float4 accumulators live in
// registers and are flushed with atomics once per workgroup chunk.
constexpr int PEG_CHUNK = 256;
__global__ __launch_bounds__(256) void peg_bwd_weight_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                             float* __restrict__ dw27, float* __restrict__ dbias, Grid5 g) {
  const int lanes_per_pos = g.d4;                    // threads covering one position
  const int npg = 256 / lanes_per_pos;               // positions processed concurrently
  const int c = threadIdx.x % lanes_per_pos, pg = threadIdx.x / lanes_per_pos;
  if (pg >= npg) return;
  const long npos = g.B * g.T * g.H * g.W;
  const long p0 = (long)blockIdx.x * PEG_CHUNK;
  float4 acc[28];
#pragma unroll
  for (int i = 0; i < 28; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  const float4 one = make_float4(1.f, 1.f, 1.f, 1.f);
  for (long pos = p0 + pg; pos < p0 + PEG_CHUNK && pos < npos; pos += npg) {
    const int w_ = (int)(pos % g.W);
    const int h_ = (int)((pos / g.W) % g.H);
    const int t_ = (int)((pos / ((long)g.W * g.H)) % g.T);
    const long b = pos / ((long)g.W * g.H * g.T);
    const float4 d = ((const float4*)dy)[pos * g.d4 + c];
    fma4(acc[27], d, one);
#pragma unroll
    for (int kt = 0; kt < 3; ++kt) {
      const int tt = t_ + kt - 2;
      if (tt < 0) continue;
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const int hh = h_ + kh - 1;
        if (hh < 0 || hh >= g.H) continue;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int ww = w_ + kw - 1;
          if (ww < 0 || ww >= g.W) continue;
          const long np = ((b * g.T + tt) * g.H + hh) * g.W + ww;
          fma4(acc[(kt * 3 + kh) * 3 + kw], d, ((const float4*)x)[np * g.d4 + c]);
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 28; ++i) {
    float* dst = (i < 27) ? dw27 + ((long)i * g.d4 + c) * 4 : dbias + c * 4;
    atomicAdd(dst + 0, acc[i].x); atomicAdd(dst + 1, acc[i].y); atomicAdd(dst + 2, acc[i].z); atomicAdd(dst + 3, acc[i].w);
  }
}

inline unsigned grid_for(long work) {
  long b = (work + 255) / 256;
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (unsigned)b;
}

}  // namespace

extern "C" {

int ctclip_peg_fwd(const float* x, const float* w27, const float* bias, float* y, void* y_bf16, long B, int T, int H,
                   int W, int d, int residual, void* stream) {
  if (B * T * H * W <= 0) return 0;
  if (d & 3) return (int)hipErrorInvalidValue;
  Grid5 g{B, T, H, W, d / 4};
  hipLaunchKernelGGL(peg_fwd_kernel, dim3(grid_for(B * T * H * W * g.d4)), dim3(256), 0, (hipStream_t)stream, x, w27,
                     bias, y, (bf16_t*)y_bf16, g, residual);
  CTCLIP_CHECK_LAUNCH();
}

int ctclip_peg_bwd_data(const float* dy, const float* w27, float* dx, void* dx_bf16, long B, int T, int H, int W, int d,
                        int residual, void* stream) {
  if (B * T * H * W <= 0) return 0;
  if (d & 3) return (int)hipErrorInvalidValue;
  Grid5 g{B, T, H, W, d / 4};
  hipLaunchKernelGGL(peg_bwd_data_kernel, dim3(grid_for(B * T * H * W * g.d4)), dim3(256), 0, (hipStream_t)stream, dy,
                     w27, dx, (bf16_t*)dx_bf16, g, residual);
  CTCLIP_CHECK_LAUNCH();
}

int ctclip_peg_bwd_weight(const float* dy, const float* x, float* dw27, float* dbias, long B, int T, int H, int W, int d,
                          void* stream) {
  const long npos = B * T * H * W;
  if (npos <= 0) return 0;
  if ((d & 3) || d / 4 > 256) return (int)hipErrorInvalidValue;
  Grid5 g{B, T, H, W, d / 4};
  hipLaunchKernelGGL(peg_bwd_weight_kernel, dim3((unsigned)((npos + PEG_CHUNK - 1) / PEG_CHUNK)), dim3(256), 0,
                     (hipStream_t)stream, dy, x, dw27, dbias, g);
  CTCLIP_CHECK_LAUNCH();
}

}  // extern "C"
