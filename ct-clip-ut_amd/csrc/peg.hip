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

constexpr int PEG_CW = 8;   // outputs per thread along w

// One thread owns 4 channels of PEG_CW consecutive w positions of one (b,t,h) row.  For each of the 9 (kt,kh) tap rows it
// loads the PEG_CW+2 inputs once and slides the 3 kw taps over them in registers: 11.25 loads per output instead of 27.
//   FWD : y[t,h,w]  = [x] + bias + sum w27[kt,kh,kw] * x[t+kt-2, h+kh-1, w+kw-1]                     (attention.py:73-76)
//   !FWD: dx[t,h,w] = [dy]      + sum w27[kt,kh,kw] * dy[t-kt+2, h-kh+1, w-kw+1]                     (its transpose)
template <bool FWD>
__global__ __launch_bounds__(256) void peg_conv_kernel(const float* __restrict__ x, const float* __restrict__ w27,
                                                       const float* __restrict__ bias, float* __restrict__ y,
                                                       bf16_t* __restrict__ y16, Grid5 g, int residual) {
  const int wchunks = (g.W + PEG_CW - 1) / PEG_CW;
  const long total = g.B * g.T * g.H * wchunks * g.d4;
  // one pass, no grid stride: block ids are remapped so each XCD walks a CONTIGUOUS range of (b,t,h) rows -- the
  // h+-1 / t-1 / t-2 neighbour rows are then re-read from that XCD's own L2 instead of from HBM / Infinity Cache.
  {
    const long idx = (long)xcd_remap(blockIdx.x, gridDim.x) * 256 + threadIdx.x;
    if (idx >= total) return;
    const int c = (int)(idx % g.d4);
    long r = idx / g.d4;
    const int wc = (int)(r % wchunks); r /= wchunks;
    const int h_ = (int)(r % g.H); r /= g.H;
    const int t_ = (int)(r % g.T);
    const long b = r / g.T;
    const int w0 = wc * PEG_CW;
    const float4* xv = (const float4*)x;
    float4 acc[PEG_CW];
    const long row_c = ((b * g.T + t_) * g.H + h_) * g.W;
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (FWD) bv = ((const float4*)bias)[c];
#pragma unroll
    for (int i = 0; i < PEG_CW; ++i) {
      acc[i] = bv;
      if (residual && w0 + i < g.W) {
        const float4 v = xv[(row_c + w0 + i) * g.d4 + c];
        acc[i].x += v.x; acc[i].y += v.y; acc[i].z += v.z; acc[i].w += v.w;
      }
    }
#pragma unroll
    for (int kt = 0; kt < 3; ++kt) {
      const int tt = FWD ? t_ + kt - 2 : t_ - kt + 2;
      if (tt < 0 || tt >= g.T) continue;
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const int hh = FWD ? h_ + kh - 1 : h_ - kh + 1;
        if (hh < 0 || hh >= g.H) continue;
        const long row = ((b * g.T + tt) * g.H + hh) * g.W;
        float4 xs[PEG_CW + 2];
#pragma unroll
        for (int i = 0; i < PEG_CW + 2; ++i) {
          const int ww = w0 + i - 1;
          xs[i] = (ww >= 0 && ww < g.W) ? xv[(row + ww) * g.d4 + c] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        const float4* wp = (const float4*)w27 + ((kt * 3 + kh) * 3) * g.d4 + c;
        const float4 k0 = wp[0], k1 = wp[g.d4], k2 = wp[2 * g.d4];
#pragma unroll
        for (int i = 0; i < PEG_CW; ++i) {
          if (FWD) { fma4(acc[i], k0, xs[i]); fma4(acc[i], k1, xs[i + 1]); fma4(acc[i], k2, xs[i + 2]); }
          else     { fma4(acc[i], k2, xs[i]); fma4(acc[i], k1, xs[i + 1]); fma4(acc[i], k0, xs[i + 2]); }
        }
      }
    }
#pragma unroll
    for (int i = 0; i < PEG_CW; ++i) {
      if (w0 + i >= g.W) continue;
      const long o = (row_c + w0 + i) * g.d4 + c;
      if (y) ((float4*)y)[o] = acc[i];
      if (y16) {
        uint2 p;
        p.x = pack_bf16x2(acc[i].x, acc[i].y);
        p.y = pack_bf16x2(acc[i].z, acc[i].w);
        ((uint2*)y16)[o] = p;
      }
    }
  }
}

// dw27[tap][c] += sum_pos dy[pos][c] * x[pos + off(tap)][c];  dbias[c] += sum_pos dy[pos][c]
// Same sliding window: a thread owns 4 channels, walks w-chunks of a range of (b,t,h) rows, keeps the 27+1 float4
// accumulators in registers and flushes them with atomics once.
constexpr int PEG_ROWS_PER_BLOCK = 16;
__global__ __launch_bounds__(256) void peg_bwd_weight_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                             float* __restrict__ dw27, float* __restrict__ dbias, Grid5 g) {
  const int c = threadIdx.x % g.d4, grp = threadIdx.x / g.d4, ngrp = 256 / g.d4;
  if (grp >= ngrp) return;
  const int wchunks = (g.W + PEG_CW - 1) / PEG_CW;
  const long nrows = g.B * g.T * g.H;
  const long work0 = (long)xcd_remap(blockIdx.x, gridDim.x) * PEG_ROWS_PER_BLOCK * wchunks;
  const long work1 = min(nrows * wchunks, work0 + (long)PEG_ROWS_PER_BLOCK * wchunks);
  const float4* xv = (const float4*)x;
  const float4* dv = (const float4*)dy;
  float4 acc[28];
#pragma unroll
  for (int i = 0; i < 28; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (long wk = work0 + grp; wk < work1; wk += ngrp) {
    const int wc = (int)(wk % wchunks);
    long r = wk / wchunks;
    const int h_ = (int)(r % g.H); r /= g.H;
    const int t_ = (int)(r % g.T);
    const long b = r / g.T;
    const int w0 = wc * PEG_CW;
    const long row_c = ((b * g.T + t_) * g.H + h_) * g.W;
    float4 d[PEG_CW];
#pragma unroll
    for (int i = 0; i < PEG_CW; ++i) {
      d[i] = (w0 + i < g.W) ? dv[(row_c + w0 + i) * g.d4 + c] : make_float4(0.f, 0.f, 0.f, 0.f);
      acc[27].x += d[i].x; acc[27].y += d[i].y; acc[27].z += d[i].z; acc[27].w += d[i].w;
    }
#pragma unroll
    for (int kt = 0; kt < 3; ++kt) {
      const int tt = t_ + kt - 2;
      if (tt < 0) continue;
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const int hh = h_ + kh - 1;
        if (hh < 0 || hh >= g.H) continue;
        const long row = ((b * g.T + tt) * g.H + hh) * g.W;
        float4 xs[PEG_CW + 2];
#pragma unroll
        for (int i = 0; i < PEG_CW + 2; ++i) {
          const int ww = w0 + i - 1;
          xs[i] = (ww >= 0 && ww < g.W) ? xv[(row + ww) * g.d4 + c] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < PEG_CW; ++i) {
          fma4(acc[(kt * 3 + kh) * 3 + 0], d[i], xs[i]);
          fma4(acc[(kt * 3 + kh) * 3 + 1], d[i], xs[i + 1]);
          fma4(acc[(kt * 3 + kh) * 3 + 2], d[i], xs[i + 2]);
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
  hipLaunchKernelGGL(peg_conv_kernel<true>, dim3(grid_for(B * T * H * ((W + PEG_CW - 1) / PEG_CW) * g.d4)), dim3(256), 0,
                     (hipStream_t)stream, x, w27, bias, y, (bf16_t*)y_bf16, g, residual);
  CTCLIP_CHECK_LAUNCH();
}

int ctclip_peg_bwd_data(const float* dy, const float* w27, float* dx, void* dx_bf16, long B, int T, int H, int W, int d,
                        int residual, void* stream) {
  if (B * T * H * W <= 0) return 0;
  if (d & 3) return (int)hipErrorInvalidValue;
  Grid5 g{B, T, H, W, d / 4};
  hipLaunchKernelGGL(peg_conv_kernel<false>, dim3(grid_for(B * T * H * ((W + PEG_CW - 1) / PEG_CW) * g.d4)), dim3(256), 0,
                     (hipStream_t)stream, dy, w27, (const float*)nullptr, dx, (bf16_t*)dx_bf16, g, residual);
  CTCLIP_CHECK_LAUNCH();
}

int ctclip_peg_bwd_weight(const float* dy, const float* x, float* dw27, float* dbias, long B, int T, int H, int W, int d,
                          void* stream) {
  const long npos = B * T * H * W;
  if (npos <= 0) return 0;
  if ((d & 3) || d / 4 > 256) return (int)hipErrorInvalidValue;
  Grid5 g{B, T, H, W, d / 4};
  const long nrows = B * T * H;
  hipLaunchKernelGGL(peg_bwd_weight_kernel, dim3((unsigned)((nrows + PEG_ROWS_PER_BLOCK - 1) / PEG_ROWS_PER_BLOCK)),
                     dim3(256), 0, (hipStream_t)stream, dy, x, dw27, dbias, g);
  CTCLIP_CHECK_LAUNCH();
}

}  // extern "C"
