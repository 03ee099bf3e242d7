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

// One thread owns 4 channels x PEG_CW consecutive w positions of one (b, h) row and SWEEPS t.  Input slab t' (the three
// h-neighbour rows, PEG_CW+2 positions each) is loaded once and scattered into three rotating accumulator sets, one per
// output time it feeds; when a slab has been consumed the oldest set is complete and is stored.  3.75 loads per output
// (the tap-gather form needs 27, the per-(kt,kh) sliding window 11.25) and the 27 x d weights sit in LDS.
//   FWD : y[t,h,w]  = [x] + bias + sum w27[kt,kh,kw] * x[t+kt-2, h+kh-1, w+kw-1]   -> slab t' feeds outputs t'..t'+2
//   !FWD: dx[t,h,w] = [dy]      + sum w27[kt,kh,kw] * dy[t-kt+2, h-kh+1, w-kw+1]   -> slab t' feeds outputs t'-2..t'
template <bool FWD>
__global__ __launch_bounds__(256) void peg_sweep_kernel(const float* __restrict__ x, const float* __restrict__ w27,
                                                        const float* __restrict__ bias, float* __restrict__ y,
                                                        bf16_t* __restrict__ y16, Grid5 g, int residual) {
  extern __shared__ __attribute__((aligned(16))) float4 wl[];          // [27][d4]
  for (int i = threadIdx.x; i < 27 * g.d4; i += 256) wl[i] = ((const float4*)w27)[i];
  __syncthreads();
  const int wchunks = (g.W + PEG_CW - 1) / PEG_CW;
  const long total = g.B * g.H * wchunks * g.d4;
  const long idx = (long)xcd_remap(blockIdx.x, gridDim.x) * 256 + threadIdx.x;
  if (idx >= total) return;
  const int c = (int)(idx % g.d4);
  long r = idx / g.d4;
  const int wc = (int)(r % wchunks); r /= wchunks;
  const int h_ = (int)(r % g.H);
  const long b = r / g.H;
  const int w0 = wc * PEG_CW;
  const float4* xv = (const float4*)x;
  const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 bv = zero;
  if (FWD) bv = ((const float4*)bias)[c];
  float4 acc[3][PEG_CW];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int i = 0; i < PEG_CW; ++i) acc[a][i] = zero;

  auto store_row = [&](int t, const float4 (&av)[PEG_CW]) {
    const long row = ((b * g.T + t) * g.H + h_) * g.W;
#pragma unroll
    for (int i = 0; i < PEG_CW; ++i) {
      if (w0 + i >= g.W) continue;
      const long o = (row + w0 + i) * g.d4 + c;
      float4 v = av[i];
      v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
      if (residual) { const float4 q = xv[o]; v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w; }
      if (y) ((float4*)y)[o] = v;
      if (y16) {
        uint2 p;
        p.x = pack_bf16x2(v.x, v.y);
        p.y = pack_bf16x2(v.z, v.w);
        ((uint2*)y16)[o] = p;
      }
    }
  };

  for (int tp = 0; tp < g.T; ++tp) {
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int hh = FWD ? h_ + kh - 1 : h_ - kh + 1;
      if (hh < 0 || hh >= g.H) continue;
      const long row = ((b * g.T + tp) * g.H + hh) * g.W;
      float4 xs[PEG_CW + 2];
#pragma unroll
      for (int i = 0; i < PEG_CW + 2; ++i) {
        const int ww = w0 + i - 1;
        xs[i] = (ww >= 0 && ww < g.W) ? xv[(row + ww) * g.d4 + c] : zero;
      }
#pragma unroll
      for (int kt = 0; kt < 3; ++kt) {
        const int a = FWD ? 2 - kt : kt;                                // accumulator set (output time offset)
        const float4 k0 = wl[((kt * 3 + kh) * 3 + 0) * g.d4 + c], k1 = wl[((kt * 3 + kh) * 3 + 1) * g.d4 + c],
                     k2 = wl[((kt * 3 + kh) * 3 + 2) * g.d4 + c];
#pragma unroll
        for (int i = 0; i < PEG_CW; ++i) {
          if (FWD) { fma4(acc[a][i], k0, xs[i]); fma4(acc[a][i], k1, xs[i + 1]); fma4(acc[a][i], k2, xs[i + 2]); }
          else     { fma4(acc[a][i], k2, xs[i]); fma4(acc[a][i], k1, xs[i + 1]); fma4(acc[a][i], k0, xs[i + 2]); }
        }
      }
    }
    // FWD: output tp is complete (fed by slabs tp-2, tp-1, tp).  !FWD: output tp-2 is complete (slabs tp-2, tp-1, tp).
    const int tout = FWD ? tp : tp - 2;
    if (tout >= 0) store_row(tout, acc[0]);
#pragma unroll
    for (int i = 0; i < PEG_CW; ++i) { acc[0][i] = acc[1][i]; acc[1][i] = acc[2][i]; acc[2][i] = zero; }
  }
  if (!FWD) {                                                            // drain: outputs T-2 and T-1
    if (g.T >= 2) store_row(g.T - 2, acc[0]);
    store_row(g.T - 1, acc[1]);
  }
}

// dw27[kt,kh,kw][c] += sum dy[t,h,w][c] * x[t+kt-2, h+kh-1, w+kw-1][c];  dbias[c] += sum dy.
// Same sweep over t: a thread keeps the dy rows of three consecutive output times in registers, loads each x row once and
// feeds all 9 (kt,kw) taps of that kh from it; the 27+1 float4 accumulators are flushed with atomics at the end.
constexpr int PEG_CWW = 4;
__global__ __launch_bounds__(256) void peg_bwd_weight_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                             float* __restrict__ dw27, float* __restrict__ dbias, Grid5 g) {
  const int wchunks = (g.W + PEG_CWW - 1) / PEG_CWW;
  const long total = g.B * g.H * wchunks * g.d4;
  const long idx = (long)xcd_remap(blockIdx.x, gridDim.x) * 256 + threadIdx.x;
  if (idx >= total) return;
  const int c = (int)(idx % g.d4);
  long r = idx / g.d4;
  const int wc = (int)(r % wchunks); r /= wchunks;
  const int h_ = (int)(r % g.H);
  const long b = r / g.H;
  const int w0 = wc * PEG_CWW;
  const float4* xv = (const float4*)x;
  const float4* dv = (const float4*)dy;
  const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 acc[28];
#pragma unroll
  for (int i = 0; i < 28; ++i) acc[i] = zero;
  float4 d[3][PEG_CWW];                                                  // dy rows of output times tp, tp+1, tp+2
  auto load_dy = [&](int t, float4 (&dst)[PEG_CWW]) {
    const long row = ((b * g.T + t) * g.H + h_) * g.W;
#pragma unroll
    for (int i = 0; i < PEG_CWW; ++i) {
      dst[i] = (t < g.T && w0 + i < g.W) ? dv[(row + w0 + i) * g.d4 + c] : zero;
      acc[27].x += dst[i].x; acc[27].y += dst[i].y; acc[27].z += dst[i].z; acc[27].w += dst[i].w;
    }
  };
  load_dy(0, d[0]);
  load_dy(1, d[1]);
  for (int tp = 0; tp < g.T; ++tp) {
    load_dy(tp + 2, d[2]);
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int hh = h_ + kh - 1;
      if (hh < 0 || hh >= g.H) continue;
      const long row = ((b * g.T + tp) * g.H + hh) * g.W;
      float4 xs[PEG_CWW + 2];
#pragma unroll
      for (int i = 0; i < PEG_CWW + 2; ++i) {
        const int ww = w0 + i - 1;
        xs[i] = (ww >= 0 && ww < g.W) ? xv[(row + ww) * g.d4 + c] : zero;
      }
#pragma unroll
      for (int kt = 0; kt < 3; ++kt) {                                   // x slab tp is tap kt of output time tp - kt + 2
#pragma unroll
        for (int i = 0; i < PEG_CWW; ++i) {
          fma4(acc[(kt * 3 + kh) * 3 + 0], d[2 - kt][i], xs[i]);
          fma4(acc[(kt * 3 + kh) * 3 + 1], d[2 - kt][i], xs[i + 1]);
          fma4(acc[(kt * 3 + kh) * 3 + 2], d[2 - kt][i], xs[i + 2]);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < PEG_CWW; ++i) { d[0][i] = d[1][i]; d[1][i] = d[2][i]; }
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
  const size_t lds = (size_t)27 * d * sizeof(float);
  if (lds > 160 * 1024) return (int)hipErrorInvalidValue;
  if (lds > 65536) hipFuncSetAttribute((const void*)peg_sweep_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(peg_sweep_kernel<true>, dim3(grid_for(B * H * ((W + PEG_CW - 1) / PEG_CW) * g.d4)), dim3(256), lds,
                     (hipStream_t)stream, x, w27, bias, y, (bf16_t*)y_bf16, g, residual);
  CTCLIP_CHECK_LAUNCH();
}

int ctclip_peg_bwd_data(const float* dy, const float* w27, float* dx, void* dx_bf16, long B, int T, int H, int W, int d,
                        int residual, void* stream) {
  if (B * T * H * W <= 0) return 0;
  if (d & 3) return (int)hipErrorInvalidValue;
  Grid5 g{B, T, H, W, d / 4};
  const size_t lds = (size_t)27 * d * sizeof(float);
  if (lds > 160 * 1024) return (int)hipErrorInvalidValue;
  if (lds > 65536) hipFuncSetAttribute((const void*)peg_sweep_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(peg_sweep_kernel<false>, dim3(grid_for(B * H * ((W + PEG_CW - 1) / PEG_CW) * g.d4)), dim3(256), lds,
                     (hipStream_t)stream, dy, w27, (const float*)nullptr, dx, (bf16_t*)dx_bf16, g, residual);
  CTCLIP_CHECK_LAUNCH();
}

int ctclip_peg_bwd_weight(const float* dy, const float* x, float* dw27, float* dbias, long B, int T, int H, int W, int d,
                          void* stream) {
  const long npos = B * T * H * W;
  if (npos <= 0) return 0;
  if ((d & 3) || d / 4 > 256) return (int)hipErrorInvalidValue;
  Grid5 g{B, T, H, W, d / 4};
  hipLaunchKernelGGL(peg_bwd_weight_kernel, dim3(grid_for(B * H * ((W + PEG_CWW - 1) / PEG_CWW) * g.d4)), dim3(256), 0,
                     (hipStream_t)stream, dy, x, dw27, dbias, g);
  CTCLIP_CHECK_LAUNCH();
}

}  // extern "C"
