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
#include <stdlib.h>
#ifndef PEG_STAMPS
#define PEG_STAMPS 0                // diagnostic build: per-wave cycle totals of the plane sweep's phases (tools/bench_peg.py STAMPS=1)
#endif

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
// feeds all 9 (kt,kw) taps of that kh from it; the 27+1 float4 accumulators of a (b, h, w chunk) group are stored as one row
// [28][d] of `partials` (groups group0 .. group0 + ngroups - 1 per launch; ctclip_reduce_partials sums the rows in order).
constexpr int PEG_CWW = 4;
__global__ __launch_bounds__(256) void peg_bwd_weight_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                             float* __restrict__ partials, Grid5 g, long group0, int ngroups) {
  const int wchunks = (g.W + PEG_CWW - 1) / PEG_CWW;
  const long total = (long)ngroups * g.d4;
  const long idx = (long)xcd_remap(blockIdx.x, gridDim.x) * 256 + threadIdx.x;
  if (idx >= total) return;
  const int c = (int)(idx % g.d4);
  const long part = idx / g.d4;
  long r = group0 + part;
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
  float4* prow = (float4*)(partials + part * 28 * g.d4 * 4);
#pragma unroll
  for (int i = 0; i < 28; ++i) prow[(long)i * g.d4 + c] = acc[i];
}


// ------------------------------------------------------------------------------------------------------------------
// Plane-tiled sweep (the production path for h*w*4/PL_P*... <= 640 threads, i.e. the 24x24 CT-ViT grid).
// A workgroup owns (batch item, 16-channel slice) and sweeps t.  Each (t) plane of that slice is loaded ONCE from global
// memory into a zero-bordered LDS image [(H+2)][(Wp+2)][4 float4] (double-buffered, the next plane is in flight while
// this one is consumed); the 3x3 (h,w) neighbourhood then comes from LDS, so global traffic is exactly one read of x and
// one write of y (+ y16) -- peg_sweep_kernel above pulls every element 3.75 times through L2 and re-reads x for the
// residual.  One thread = 4 channels x PL_P consecutive w positions, three rotating accumulator sets as above.
constexpr int PL_P = 6;            // outputs per thread along w (24 = 4 strips)
// LDS plane images: rows of `pitch` 64-byte position blocks (16 channels).  A position block's banks are (column mod 4), and the
// four blocks one LDS pass serves (16 lanes x 16 bytes, or 32 lanes x 8 bytes) belong to the four strips of one row: with the
// strips side by side (PL_P = 6 columns apart) they sat on banks 0, 2, 0, 2 -- every access took two passes.  So the image
// leaves one unused column between strips: padded column c (0 = left border, c = w + 1) lives at  c + strip(c) + 1  (the left
// border at 0), a strip starts every PL_P + 1 = 7 columns (banks 0, 3, 2, 1), and a thread's PL_P + 2 taps sit at the
// compile-time offsets plane_tap(i) from ITS strip's first column.
__host__ __device__ inline int plane_pitch(int strips) { return strips * (PL_P + 1) + 3; }
__host__ __device__ constexpr int plane_tap(int i) { return i == 0 ? 0 : (i <= PL_P ? i + 1 : PL_P + 3); }
constexpr int PL_CG = 4;            // float4 channel groups per workgroup (16 channels)
constexpr int PL_MAXT = 512;        // 8 waves -> 2 per SIMD -> 256 registers per thread

__device__ __forceinline__ void lds_only_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <bool FWD>
__global__ __launch_bounds__(PL_MAXT) void peg_plane_kernel(const float* __restrict__ x, const float* __restrict__ w27,
                                                            const float* __restrict__ bias, float* __restrict__ y,
                                                            bf16_t* __restrict__ y16, Grid5 g, int residual, int strips) {
  extern __shared__ __attribute__((aligned(16))) f32x4 pl_smem[];
  const int tid = threadIdx.x, nthreads = blockDim.x;
  const int nslices = g.d4 / PL_CG;
  const int blk = xcd_remap(blockIdx.x, gridDim.x);
  const int slice = blk % nslices;
  const long b = blk / nslices;
  const int cg = tid & (PL_CG - 1), sid = tid >> 2;
  const int h_ = sid / strips, strip = sid % strips, w0 = strip * PL_P;
  const bool active = h_ < g.H;
  const int pitch = plane_pitch(strips);                     // row length of the image (position blocks)
  const int plane_f4 = (g.H + 2) * pitch * PL_CG;
  f32x4* wl = pl_smem;                                      // [27][PL_CG]
  f32x4* plane = pl_smem + 27 * PL_CG;                      // [2][(H+2)][pitch][PL_CG]
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  for (int i = tid; i < 27 * PL_CG; i += nthreads) wl[i] = ((const f32x4*)w27)[(i / PL_CG) * g.d4 + slice * PL_CG + (i % PL_CG)];
  for (int i = tid; i < 2 * plane_f4; i += nthreads) plane[i] = zero;
  const int c = slice * PL_CG + cg;
  const f32x4* xv = (const f32x4*)x;
  f32x4 bv = zero;
  if (FWD && active) bv = ((const f32x4*)bias)[c];
  f32x4 acc[3][PL_P];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int i = 0; i < PL_P; ++i) acc[a][i] = zero;

  // two planes in flight per thread (the next one and the one after): with one, 37 KiB of loads per CU are outstanding at
  // a time and the sweep waits on HBM latency
  f32x4 nx[PL_P], nx2[PL_P];
  auto load_plane = [&](int t, f32x4 (&dst)[PL_P]) {
    const long row = ((b * g.T + t) * g.H + h_) * g.W;
#pragma unroll
    for (int i = 0; i < PL_P; ++i) dst[i] = (active && w0 + i < g.W) ? xv[(row + w0 + i) * g.d4 + c] : zero;
  };
  auto stage_plane = [&](int buf) {
    if (!active) return;
    f32x4* dst = plane + buf * plane_f4 + ((h_ + 1) * pitch + w0 + strip + 2) * PL_CG + cg;
#pragma unroll
    for (int i = 0; i < PL_P; ++i) dst[i * PL_CG] = nx[i];
  };
  auto store_row = [&](int t, const f32x4 (&av)[PL_P]) {
    if (!active) return;
    const long row = ((b * g.T + t) * g.H + h_) * g.W;
#pragma unroll
    for (int i = 0; i < PL_P; ++i) {
      if (w0 + i >= g.W) continue;
      const long o = (row + w0 + i) * g.d4 + c;
      const f32x4 v = av[i] + bv;
      if (y) ((f32x4*)y)[o] = v;
      if (y16) {
        uint2 pk;
        pk.x = pack_bf16x2(v[0], v[1]);
        pk.y = pack_bf16x2(v[2], v[3]);
        ((uint2*)y16)[o] = pk;
      }
    }
  };

  load_plane(0, nx);
  __syncthreads();                                           // zero fill + weights visible
  // the bias row came from a global load: the barrier above has waited for it, but inside the loop the compiler's wait-count
  // bookkeeping no longer knows how old that load is and put `s_waitcnt vmcnt(0)` in front of every use -- i.e. in front of
  // each of the six store groups of a step, every one then waiting for the store before it.  A value redefined here has no
  // load behind it.
  asm volatile("" : "+v"(bv));
  stage_plane(0);
  if (g.T > 1) load_plane(1, nx);
  __syncthreads();
#if PEG_STAMPS
  long long cs[6] = {0, 0, 0, 0, 0, 0};
  const long long c_begin = __builtin_readcyclecounter();
  const long long r_begin = __builtin_amdgcn_s_memrealtime();
#define PEG_ST(i) { const long long c_ = __builtin_readcyclecounter(); cs[i] += c_ - c_last; c_last = c_; }
#else
#define PEG_ST(i)
#endif
  for (int tp = 0; tp < g.T; ++tp) {
#if PEG_STAMPS
    long long c_last = __builtin_readcyclecounter();
#endif
    const int buf = tp & 1;
    if (tp + 2 < g.T) load_plane(tp + 2, nx2);               // in flight while this plane and the next are consumed
    PEG_ST(0)
    const f32x4* pb = plane + buf * plane_f4;
    int wofs = 0;
    asm volatile("" : "+s"(wofs));                           // opaque zero: keeps the 27 weight reads inside the loop
    const f32x4* wk = wl + wofs;                            // (hoisted they would pin 108 registers)
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int hp = FWD ? h_ + kh : h_ + 2 - kh;           // padded row of input row h_ + kh - 1 (FWD) / h_ - kh + 1
      const f32x4* prow = pb + (hp * pitch + w0 + strip) * PL_CG + cg;
      f32x4 xs[PL_P + 2];
#pragma unroll
      for (int i = 0; i < PL_P + 2; ++i) xs[i] = prow[plane_tap(i) * PL_CG];
      if (residual && kh == 1) {                             // the residual term of output time tp is this plane's centre
#pragma unroll
        for (int i = 0; i < PL_P; ++i) {
          acc[FWD ? 0 : 2][i] += xs[i + 1];
        }
      }
#pragma unroll
      for (int kt = 0; kt < 3; ++kt) {
        const int a = FWD ? 2 - kt : kt;
        const f32x4 k0 = wk[((kt * 3 + kh) * 3 + 0) * PL_CG + cg], k1 = wk[((kt * 3 + kh) * 3 + 1) * PL_CG + cg],
                     k2 = wk[((kt * 3 + kh) * 3 + 2) * PL_CG + cg];
#pragma unroll
        for (int i = 0; i < PL_P; ++i) {
          if (FWD) acc[a][i] += k0 * xs[i] + k1 * xs[i + 1] + k2 * xs[i + 2];
          else     acc[a][i] += k2 * xs[i] + k1 * xs[i + 1] + k0 * xs[i + 2];
        }
        // pin this (kt, kh) step: its FMAs finish here and the next LDS reads start after it -- left alone the compiler
        // issues every LDS read of the plane up front and keeps ~300 registers live
#pragma unroll
        for (int i = 0; i < PL_P; ++i) asm volatile("" : "+v"(acc[a][i]) : : "memory");
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    PEG_ST(1)
    const int tout = FWD ? tp : tp - 2;
    if (tout >= 0) store_row(tout, acc[0]);
    PEG_ST(2)
#pragma unroll
    for (int i = 0; i < PL_P; ++i) { acc[0][i] = acc[1][i]; acc[1][i] = acc[2][i]; acc[2][i] = zero; }
    if (tp + 1 < g.T) stage_plane(buf ^ 1);
#pragma unroll
    for (int i = 0; i < PL_P; ++i) nx[i] = nx2[i];
    PEG_ST(3)
    lds_only_barrier();
    PEG_ST(4)
  }
  if (!FWD) {
    if (g.T >= 2) store_row(g.T - 2, acc[0]);
    store_row(g.T - 1, acc[1]);
  }
#if PEG_STAMPS
  if ((tid & 63) == 0 && y) {                                 // beyond the real output: [block][wave][8] floats (the bench allocates it)
    float* st = y + (long)g.B * g.T * g.H * g.W * g.d4 * 4 + ((long)blockIdx.x * (PL_MAXT / 64) + (tid >> 6)) * 8;
    for (int i = 0; i < 5; ++i) st[i] = (float)cs[i];
    st[5] = (float)(__builtin_readcyclecounter() - c_begin);
    st[6] = (float)(__builtin_amdgcn_s_memrealtime() - r_begin);
    st[7] = (float)(r_begin & 0xffffff);
  }
#endif
}

// plane path geometry: threads (multiple of 64) or 0 when the plane does not fit a workgroup / LDS
inline int plane_threads(int H, int W, int d, int* strips, size_t* lds) {
  if (CTCLIP_KNOB("CTCLIP_PEG_SWEEP")) return 0;
  if ((d / 4) % PL_CG) return 0;
  const int st = (W + PL_P - 1) / PL_P;
  const int items = H * st * PL_CG;
  const int threads = (items + 63) / 64 * 64;
  const size_t bytes = ((size_t)27 * PL_CG + (size_t)2 * (H + 2) * plane_pitch(st) * PL_CG) * sizeof(float4);
  if (threads > PL_MAXT || bytes > 160 * 1024) return 0;
  *strips = st;
  *lds = bytes;
  return threads;
}

// Weight / bias gradient on the same plane tiling.  A workgroup owns a 16-channel slice and a CHUNK of batch items; per
// item it sweeps t keeping the x planes t-2, t-1, t in an LDS ring (each x plane is read from global memory once) and its
// own dy values in registers; the 27 + 1 per-thread partial sums live in registers for the whole chunk, are reduced across
// the workgroup (wave shuffles, then LDS) and leave as 448 global atomics per workgroup -- peg_bwd_weight_kernel above
// issues 112 atomics per THREAD.
//   dw27[kt,kh,kw][c] += sum dy[t,h,w][c] * x[t+kt-2, h+kh-1, w+kw-1][c];  dbias[c] += sum dy
constexpr int WG_MAXT = 768;        // 12 waves -> 3 per SIMD -> 168 registers per thread
typedef __attribute__((ext_vector_type(2))) float f32x2;

// WG_C2 = float2 channel pairs per slice: 8 (16-channel slices, 768 threads, one workgroup per CU).  Measured with 4 (8-channel
// slices, 384 threads, half the LDS, two workgroups per CU): 2677 vs 1824 us -- 32-byte pieces of every line per workgroup
template <int WG_C2>
__global__ __launch_bounds__(WG_MAXT * WG_C2 / 8, 3) void peg_wgrad_plane_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                 float* __restrict__ partials, Grid5 g,
                                                                 int strips, int bchunk) {
  // one thread = 2 channels x PL_P consecutive w positions (float2 keeps the 28 accumulators at 56 registers)
  extern __shared__ __attribute__((aligned(16))) f32x2 wg_smem[];
  const int tid = threadIdx.x, nthreads = blockDim.x, lane = tid & 63, wave = tid >> 6;
  const int d2 = g.d4 * 2;
  const int nslices = d2 / WG_C2;
  const int blk = xcd_remap(blockIdx.x, gridDim.x);
  const int slice = blk % nslices;
  const long b0 = (long)(blk / nslices) * bchunk, b1 = (b0 + bchunk < g.B) ? b0 + bchunk : g.B;
  const int c2 = tid & (WG_C2 - 1), sid = tid / WG_C2;
  const int h_raw = sid / strips, strip = sid % strips, w0 = strip * PL_P;
  const bool active = h_raw < g.H;
  const int h_ = active ? h_raw : 0;                         // idle threads read row 0's neighbours (see peg_bwd_fused_kernel)
  const int pitch = plane_pitch(strips);
  const int plane_f2 = (g.H + 2) * pitch * WG_C2;
  f32x2* ring = wg_smem;                                     // [3][(H+2)][pitch][WG_C2]
  const f32x2 zero = {0.f, 0.f};
  for (int i = tid; i < 3 * plane_f2; i += nthreads) ring[i] = zero;
  const int c = slice * WG_C2 + c2;                          // float2 column of this thread
  const f32x2* xv = (const f32x2*)x;
  const f32x2* dv = (const f32x2*)dy;
  f32x2 acc[28];
#pragma unroll
  for (int i = 0; i < 28; ++i) acc[i] = zero;
  const int own = ((h_ + 1) * pitch + w0 + strip + 2) * WG_C2 + c2;  // this thread's first interior position inside a plane

  f32x2 nx[PL_P], dcur[PL_P], dnext[PL_P];
  // x and dy share the layout: uniform plane base + one 32-bit per-thread offset (the positions follow at stride d2)
  const uint32_t poff = (uint32_t)((h_ * g.W + w0) * d2 + c);
  const long plane_elems = (long)g.H * g.W * d2;
  auto load_x = [&](long b, int t) {
    const f32x2* base = xv + (b * g.T + t) * plane_elems;
#pragma unroll
    for (int i = 0; i < PL_P; ++i) nx[i] = (active && w0 + i < g.W) ? base[poff + (uint32_t)(i * d2)] : zero;
  };
  auto load_dy = [&](long b, int t, f32x2 (&dst)[PL_P]) {
    const f32x2* base = dv + (b * g.T + t) * plane_elems;
#pragma unroll
    for (int i = 0; i < PL_P; ++i) dst[i] = (active && w0 + i < g.W) ? base[poff + (uint32_t)(i * d2)] : zero;
  };
  auto put = [&](int slot, const f32x2 (&v)[PL_P]) {
    if (!active) return;
    f32x2* dst = ring + slot * plane_f2 + own;
#pragma unroll
    for (int i = 0; i < PL_P; ++i) dst[i * WG_C2] = v[i];
  };
  __syncthreads();                                           // zero fill visible (borders stay zero for good)

  for (long b = b0; b < b1; ++b) {
    load_x(b, 0);
    load_dy(b, 0, dcur);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // see the end of the t loop: no load is pending on entry either
#pragma unroll
    for (int i = 0; i < PL_P; ++i) asm volatile("" : "+v"(nx[i]), "+v"(dcur[i]));
    lds_only_barrier();                                      // the previous item's planes are no longer read
    {
      f32x2 zs[PL_P];
#pragma unroll
      for (int i = 0; i < PL_P; ++i) zs[i] = zero;
      put(1, zs);                                            // x[-2] and x[-1]: the causal padding
      put(2, zs);
    }
    put(0, nx);
    lds_only_barrier();
    for (int t = 0; t < g.T; ++t) {
      if (t + 1 < g.T) { load_x(b, t + 1); load_dy(b, t + 1, dnext); }
#pragma unroll
      for (int i = 0; i < PL_P; ++i) acc[27] += dcur[i];
#pragma unroll
      for (int kt = 0; kt < 3; ++kt) {
        const f32x2* pb = ring + ((t + kt + 1) % 3) * plane_f2;      // plane t + kt - 2
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
          const f32x2* prow = pb + ((h_ + kh) * pitch + w0 + strip) * WG_C2 + c2;
          f32x2 xs[PL_P + 2];
#pragma unroll
          for (int i = 0; i < PL_P + 2; ++i) xs[i] = prow[plane_tap(i) * WG_C2];
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            f32x2 a = acc[(kt * 3 + kh) * 3 + kw];
#pragma unroll
            for (int i = 0; i < PL_P; ++i) a += dcur[i] * xs[i + kw];
            acc[(kt * 3 + kh) * 3 + kw] = a;
          }
          // pin the step (see peg_plane_kernel)
          asm volatile("" : "+v"(acc[(kt * 3 + kh) * 3 + 0]), "+v"(acc[(kt * 3 + kh) * 3 + 1]), "+v"(acc[(kt * 3 + kh) * 3 + 2]) : : "memory");
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      // The loads of step t + 1 (issued at the top of this step) are waited for HERE, explicitly, and the registers pass
      // through an empty asm: left to the compiler's wait-count bookkeeping, the loop-carried dy registers counted as "load
      // of unknown age pending" at the top of the next step and `s_waitcnt vmcnt(0)` sat right behind the issue of that
      // step's loads -- nothing was ever in flight during the arithmetic.
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < PL_P; ++i) asm volatile("" : "+v"(nx[i]), "+v"(dnext[i]));
      lds_only_barrier();                                    // everybody is done with plane t-2
      if (t + 1 < g.T) put((t + 1) % 3, nx);
      lds_only_barrier();
#pragma unroll
      for (int i = 0; i < PL_P; ++i) dcur[i] = dnext[i];
    }
  }

  // reduce the 28 partial sums over the threads that share a channel pair: lanes l ^ {WG_C2 .. 32}, then the waves
#pragma unroll
  for (int i = 0; i < 28; ++i) {
#pragma unroll
    for (int o = WG_C2; o < 64; o <<= 1) {
      f32x2 v = acc[i];
      v[0] = __shfl_xor(v[0], o, 64); v[1] = __shfl_xor(v[1], o, 64);
      acc[i] += v;
    }
  }
  __syncthreads();                                           // ring is free
  f32x2* red = wg_smem;                                      // [waves][WG_C2][28]
  if (lane < WG_C2) {
#pragma unroll
    for (int i = 0; i < 28; ++i) red[(wave * WG_C2 + lane) * 28 + i] = acc[i];
  }
  __syncthreads();
  const int nwaves = nthreads >> 6;
  for (int idx = tid; idx < WG_C2 * 28; idx += nthreads) {
    const int rc = idx / 28, i = idx % 28;
    f32x2 sum = zero;
    for (int wv = 0; wv < nwaves; ++wv) sum += red[(wv * WG_C2 + rc) * 28 + i];
    // row (batch chunk) of `partials`: [28][d]; the chunks are added up in chunk order by ctclip_reduce_partials
    float* dst = partials + ((long)(blk / nslices) * 28 + i) * (d2 * 2) + (slice * WG_C2 + rc) * 2;
    dst[0] = sum[0]; dst[1] = sum[1];
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Data AND weight gradient in one pass over dy (round 5).  Both are sums over the SAME 27 neighbours of dy:
//     dx[t,h,w]       = [dy[t,h,w]] + sum_taps w27[kt,kh,kw] N(kt,kh,kw),        N = dy[t - kt + 2, h - kh + 1, w - kw + 1]
//     dw27[kt,kh,kw] += x[t,h,w] N(kt,kh,kw)                 (the weight gradient re-indexed onto the position of x),  dbias += dy
// so with the ring of the weight-gradient kernel holding the dy planes t, t+1, t+2 (instead of x planes t-2 .. t) and the thread's
// own x values in registers, every LDS read of a neighbour feeds two multiply-adds: one into dx, one into the tap's sum.  The
// separate data-gradient sweep -- a second read of dy and 6.9 GB of the 12.4 GB the two kernels moved per call at 96 pairs -- is
// gone; the 27 weights of the slice sit in LDS behind the ring (read as broadcasts).  Same decomposition as
// peg_wgrad_plane_kernel: a workgroup owns a 16-channel slice and a chunk of batch items, one thread = 2 channels x PL_P
// positions, partial sums of the chunk leave as one row of `partials`.
// Shape: one thread = 2 channels x FP = 12 consecutive w positions (the 24-wide CT-ViT grid = 2 strips), 384 threads for 24 rows x 2
// strips x 8 channel pairs -- at most two waves per SIMD, i.e. 256 registers: the 28 + 12 running sums, the thread's own x and the
// two planes in flight need ~190 (with 6 positions per thread and 768 threads the same kernel spilled 164 registers at the 168 it
// may use).  A wider thread also reads less: 14 neighbour reads + 3 weights per 12 outputs and (kt, kh) instead of 8 + 3 per 6.
// Image rows as in plane_pitch(): one unused column between strips, and the pitch padded to 2 (mod 4) position blocks so that the
// (row, strip) pairs of one LDS pass sit on different bank groups.
constexpr int FP = 12;
__host__ __device__ inline int fused_pitch(int strips) { int p = strips * (FP + 1) + 3; while ((p & 3) != 2) ++p; return p; }
__host__ __device__ constexpr int fused_tap(int i) { return i == 0 ? 0 : (i <= FP ? i + 1 : FP + 3); }
constexpr int FUSED_MAXT = 512;

template <int WG_C2>
__global__ __launch_bounds__(FUSED_MAXT, 2) void peg_bwd_fused_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                               const float* __restrict__ w27, float* __restrict__ dx,
                                                               bf16_t* __restrict__ dx16, float* __restrict__ partials, Grid5 g,
                                                               int strips, int bchunk, int residual) {
  extern __shared__ __attribute__((aligned(16))) f32x2 wg_smem[];
  const int tid = threadIdx.x, nthreads = blockDim.x, lane = tid & 63, wave = tid >> 6;
  const int d2 = g.d4 * 2;
  const int nslices = d2 / WG_C2;
  const int blk = xcd_remap(blockIdx.x, gridDim.x);
  const int slice = blk % nslices;
  const long b0 = (long)(blk / nslices) * bchunk, b1 = (b0 + bchunk < g.B) ? b0 + bchunk : g.B;
  const int c2 = tid & (WG_C2 - 1), sid = tid / WG_C2;
  const int h_raw = sid / strips, strip = sid % strips, w0 = strip * FP;
  const bool active = h_raw < g.H;
  // threads past the last row (the block is rounded up to whole waves) still run the neighbour reads: on row 0, inside the
  // initialised ring -- their own x is zero, and 0 x (whatever a stray LDS word holds, possibly a NaN pattern) must not reach the
  // weight-gradient sums they share with the active lanes
  const int h_ = active ? h_raw : 0;
  const int pitch = fused_pitch(strips);
  const int plane_f2 = (g.H + 2) * pitch * WG_C2;
  f32x2* ring = wg_smem;                                     // [3][(H+2)][pitch][WG_C2]: dy planes t, t+1, t+2
  f32x2* wl = wg_smem + 3 * plane_f2;                        // [27][WG_C2]
  const f32x2 zero = {0.f, 0.f};
  for (int i = tid; i < 3 * plane_f2; i += nthreads) ring[i] = zero;
  const int c = slice * WG_C2 + c2;                          // float2 column of this thread
  for (int i = tid; i < 27 * WG_C2; i += nthreads) wl[i] = ((const f32x2*)w27)[(i / WG_C2) * d2 + slice * WG_C2 + (i % WG_C2)];
  const f32x2* xv = (const f32x2*)x;
  const f32x2* dv = (const f32x2*)dy;
  f32x2 acc[28];
#pragma unroll
  for (int i = 0; i < 28; ++i) acc[i] = zero;
  const int own = ((h_ + 1) * pitch + w0 + strip + 2) * WG_C2 + c2;  // this thread's first interior position inside a plane

  f32x2 xcur[FP], xnext[FP], dnext[FP];
  const uint32_t poff = (uint32_t)((h_ * g.W + w0) * d2 + c);
  const long plane_elems = (long)g.H * g.W * d2;
  auto load_plane = [&](const f32x2* src, long b, int t, f32x2 (&dst)[FP]) {      // t >= T: zeros (the planes past the end)
    const f32x2* base = src + (b * g.T + (t < g.T ? t : 0)) * plane_elems;
#pragma unroll
    for (int i = 0; i < FP; ++i) dst[i] = (active && t < g.T && w0 + i < g.W) ? base[poff + (uint32_t)(i * d2)] : zero;
  };
  auto put = [&](int slot, const f32x2 (&v)[FP]) {         // into the ring; the bias gradient sums every dy element once, here
    if (!active) return;
    f32x2* dst = ring + slot * plane_f2 + own;
#pragma unroll
    for (int i = 0; i < FP; ++i) { dst[i * WG_C2] = v[i]; acc[27] += v[i]; }
  };
  __syncthreads();                                           // zero fill + weights visible (borders stay zero for good)

  for (long b = b0; b < b1; ++b) {
    lds_only_barrier();                                      // the previous item's planes are no longer read
#pragma unroll
    for (int s = 0; s < 3; ++s) {                            // dy planes 0, 1, 2
      load_plane(dv, b, s, dnext);
      put(s, dnext);
    }
    load_plane(xv, b, 0, xcur);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < FP; ++i) asm volatile("" : "+v"(xcur[i]));
    lds_only_barrier();
    for (int t = 0; t < g.T; ++t) {
      load_plane(dv, b, t + 3, dnext);                       // in flight while this plane is consumed
      load_plane(xv, b, t + 1, xnext);
      f32x2 dxa[FP];
#pragma unroll
      for (int i = 0; i < FP; ++i) dxa[i] = zero;
      int wofs = 0;
      asm volatile("" : "+s"(wofs));                         // opaque zero: keeps the 27 weight reads inside the loop
      const f32x2* wk = wl + wofs;
#pragma unroll
      for (int kt = 0; kt < 3; ++kt) {
        const f32x2* pb = ring + ((t + 2 - kt) % 3) * plane_f2;      // dy plane t - kt + 2
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
          const f32x2* prow = pb + ((h_ + 2 - kh) * pitch + w0 + strip) * WG_C2 + c2;       // row h - kh + 1
          f32x2 xs[FP + 2];
#pragma unroll
          for (int i = 0; i < FP + 2; ++i) xs[i] = prow[fused_tap(i) * WG_C2];
          if (residual && kt == 2 && kh == 1) {              // N(2,1,1) is dy[t,h,w] itself: the residual path
#pragma unroll
            for (int i = 0; i < FP; ++i) dxa[i] += xs[i + 1];
          }
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            const f32x2 wt = wk[((kt * 3 + kh) * 3 + kw) * WG_C2 + c2];
            f32x2 a = acc[(kt * 3 + kh) * 3 + kw];
#pragma unroll
            for (int i = 0; i < FP; ++i) {                 // neighbour w - kw + 1 of output i sits at padded column i + 2 - kw
              const f32x2 nb = xs[i + 2 - kw];
              dxa[i] += wt * nb;
              a += xcur[i] * nb;
            }
            acc[(kt * 3 + kh) * 3 + kw] = a;
          }
          // pin the step (see peg_plane_kernel): both sets of sums, or the compiler issues every LDS read of the plane up front
          asm volatile("" : "+v"(acc[(kt * 3 + kh) * 3 + 0]), "+v"(acc[(kt * 3 + kh) * 3 + 1]), "+v"(acc[(kt * 3 + kh) * 3 + 2]) : : "memory");
#pragma unroll
          for (int i = 0; i < FP; ++i) asm volatile("" : "+v"(dxa[i]) : : "memory");
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      // the loads of this step are waited for HERE (see peg_wgrad_plane_kernel) -- before this step's stores are issued, so that
      // the wait does not drain stores: they go out last and retire under the next step's arithmetic
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < FP; ++i) asm volatile("" : "+v"(xnext[i]), "+v"(dnext[i]));
      lds_only_barrier();                                    // everybody is done with dy plane t
      put(t % 3, dnext);                                     // dy plane t + 3 takes its slot
      lds_only_barrier();
#pragma unroll
      for (int i = 0; i < FP; ++i) xcur[i] = xnext[i];
      if (active) {                                          // dx[t] (f32 and its bf16 mirror)
        const long o0 = (b * g.T + t) * plane_elems + poff;
#pragma unroll
        for (int i = 0; i < FP; ++i) {
          if (w0 + i >= g.W) continue;
          if (dx) ((f32x2*)dx)[o0 + (uint32_t)(i * d2)] = dxa[i];
          if (dx16) ((uint32_t*)dx16)[o0 + (uint32_t)(i * d2)] = pack_bf16x2(dxa[i][0], dxa[i][1]);
        }
      }
    }
  }

  // reduce the 28 partial sums over the threads that share a channel pair: lanes l ^ {WG_C2 .. 32}, then the waves
#pragma unroll
  for (int i = 0; i < 28; ++i) {
#pragma unroll
    for (int o = WG_C2; o < 64; o <<= 1) {
      f32x2 v = acc[i];
      v[0] = __shfl_xor(v[0], o, 64); v[1] = __shfl_xor(v[1], o, 64);
      acc[i] += v;
    }
  }
  __syncthreads();                                           // ring is free
  f32x2* red = wg_smem;                                      // [waves][WG_C2][28]
  if (lane < WG_C2) {
#pragma unroll
    for (int i = 0; i < 28; ++i) red[(wave * WG_C2 + lane) * 28 + i] = acc[i];
  }
  __syncthreads();
  const int nwaves = nthreads >> 6;
  for (int idx = tid; idx < WG_C2 * 28; idx += nthreads) {
    const int rc = idx / 28, i = idx % 28;
    f32x2 sum = zero;
    for (int wv = 0; wv < nwaves; ++wv) sum += red[(wv * WG_C2 + rc) * 28 + i];
    float* dst = partials + ((long)(blk / nslices) * 28 + i) * (d2 * 2) + (slice * WG_C2 + rc) * 2;
    dst[0] = sum[0]; dst[1] = sum[1];
  }
}

inline int wgrad_plane_threads(int H, int W, int d, int WG_C2, int* strips, size_t* lds) {
  if (CTCLIP_KNOB("CTCLIP_PEG_SWEEP")) return 0;
  if ((d / 4) % PL_CG) return 0;
  const int st = (W + PL_P - 1) / PL_P;
  const int threads = (H * st * WG_C2 + 63) / 64 * 64;
  size_t bytes = (size_t)3 * (H + 2) * plane_pitch(st) * WG_C2 * sizeof(float2);
  const size_t red = (size_t)(threads / 64) * WG_C2 * 28 * sizeof(float2);
  if (bytes < red) bytes = red;
  if (threads > WG_MAXT * WG_C2 / 8 || bytes > (size_t)160 * 1024 * WG_C2 / 8) return 0;
  *strips = st;
  *lds = bytes;
  return threads;
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
  {
    int strips = 0;
    size_t plds = 0;
    const int threads = plane_threads(H, W, d, &strips, &plds);
    if (threads > 0 && B * (g.d4 / PL_CG) < (1L << 30)) {
      if (plds > 65536) hipFuncSetAttribute((const void*)peg_plane_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)plds);
      hipLaunchKernelGGL(peg_plane_kernel<true>, dim3((unsigned)(B * (g.d4 / PL_CG))), dim3(threads), plds, (hipStream_t)stream,
                         x, w27, bias, y, (bf16_t*)y_bf16, g, residual, strips);
      CTCLIP_CHECK_LAUNCH();
    }
  }
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
  {
    int strips = 0;
    size_t plds = 0;
    const int threads = plane_threads(H, W, d, &strips, &plds);
    if (threads > 0 && B * (g.d4 / PL_CG) < (1L << 30)) {
      if (plds > 65536) hipFuncSetAttribute((const void*)peg_plane_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)plds);
      hipLaunchKernelGGL(peg_plane_kernel<false>, dim3((unsigned)(B * (g.d4 / PL_CG))), dim3(threads), plds, (hipStream_t)stream,
                         dy, w27, (const float*)nullptr, dx, (bf16_t*)dx_bf16, g, residual, strips);
      CTCLIP_CHECK_LAUNCH();
    }
  }
  const size_t lds = (size_t)27 * d * sizeof(float);
  if (lds > 160 * 1024) return (int)hipErrorInvalidValue;
  if (lds > 65536) hipFuncSetAttribute((const void*)peg_sweep_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(peg_sweep_kernel<false>, dim3(grid_for(B * H * ((W + PEG_CW - 1) / PEG_CW) * g.d4)), dim3(256), lds,
                     (hipStream_t)stream, dy, w27, (const float*)nullptr, dx, (bf16_t*)dx_bf16, g, residual);
  CTCLIP_CHECK_LAUNCH();
}

// data + weight + bias gradient in ONE pass over dy (peg_bwd_fused_kernel); hipErrorInvalidValue when the grid does not take the
// plane tiling (the caller then runs ctclip_peg_bwd_data and ctclip_peg_bwd_weight)
int ctclip_peg_bwd_fused(const float* dy, const float* x, const float* w27, float* dx, void* dx_bf16, float* dw27, float* dbias,
                         long B, int T, int H, int W, int d, int residual, float* partials, void* stream) {
  const long npos = B * T * H * W;
  if (npos <= 0) return 0;
  if ((d & 3) || d / 4 > 256 || !partials || !dx) return (int)hipErrorInvalidValue;
  Grid5 g{B, T, H, W, d / 4};
  hipStream_t st = (hipStream_t)stream;
  constexpr int c2 = 8;
  if ((d / 4) % PL_CG) return (int)hipErrorInvalidValue;
  const int strips = (W + FP - 1) / FP;
  const int threads = (H * strips * c2 + 63) / 64 * 64;
  if (threads > FUSED_MAXT) return (int)hipErrorInvalidValue;
  size_t plds = ((size_t)3 * (H + 2) * fused_pitch(strips) * c2 + 27 * c2) * sizeof(float2);   // the dy ring + the slice's 27 weights
  const size_t red = (size_t)(threads / 64) * c2 * 28 * sizeof(float2);
  if (plds < red) plds = red;
  if (plds > (size_t)160 * 1024) return (int)hipErrorInvalidValue;
  const long rowf = 28L * d;
  const int ncu = ctclip_cu_count8();
  const int nslices = 2 * g.d4 / c2;
  // batch items per workgroup: as ctclip_peg_bwd_weight (one workgroup per CU, the chunking with the fewest item-times)
  const long slots = (long)ncu;
  const long cap = kPartialsFloats / rowf;
  long bchunk = B, best = -1;
  for (int r : {2, 1, 3, 4}) {
    long nc = (r * slots) / nslices;
    if (nc > cap) nc = cap;
    if (nc < 1) nc = 1;
    if (nc > B) nc = B;
    const long bc = (B + nc - 1) / nc;
    nc = (B + bc - 1) / bc;
    const long cost = ((nc * nslices + slots - 1) / slots) * bc;
    if (best < 0 || cost < best) { best = cost; bchunk = bc; }
  }
  const long nchunks = (B + bchunk - 1) / bchunk;
  if (plds > 65536) hipFuncSetAttribute((const void*)peg_bwd_fused_kernel<c2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)plds);
  hipLaunchKernelGGL(peg_bwd_fused_kernel<c2>, dim3((unsigned)(nchunks * nslices)), dim3(threads), plds, st, dy, x, w27, dx,
                     (bf16_t*)dx_bf16, partials, g, strips, (int)bchunk, residual);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return (int)e;
  int r = ctclip_reduce_partials(partials, (int)nchunks, rowf, 27 * d, dw27, st);
  if (r == 0) r = ctclip_reduce_partials(partials + 27L * d, (int)nchunks, rowf, d, dbias, st);
  return r;
}

int ctclip_peg_bwd_weight(const float* dy, const float* x, float* dw27, float* dbias, long B, int T, int H, int W, int d,
                          float* partials, void* stream) {
  const long npos = B * T * H * W;
  if (npos <= 0) return 0;
  if ((d & 3) || d / 4 > 256 || !partials) return (int)hipErrorInvalidValue;
  Grid5 g{B, T, H, W, d / 4};
  hipStream_t st = (hipStream_t)stream;
  const long rowf = 28L * d;                                 // floats per partial row: 27 taps + the bias sum
  auto finish = [&](int nparts) -> int {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    int r = ctclip_reduce_partials(partials, nparts, rowf, 27 * d, dw27, st);
    if (r == 0) r = ctclip_reduce_partials(partials + 27L * d, nparts, rowf, d, dbias, st);
    return r;
  };
  {
    int strips = 0;
    size_t plds = 0;
    constexpr int c2 = 8;
    const int threads = wgrad_plane_threads(H, W, d, c2, &strips, &plds);
    if (threads > 0) {
      static const int ncu = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) return 256;
        return v;
      }();
      const int nslices = 2 * g.d4 / c2;
      // batch items per workgroup: the workgroups run in rounds of one (two) per CU, and a round lasts as long as its longest
      // chunk -- take the chunking with the fewest item-times (rounds x items per chunk) among one to four rounds, two rounds
      // on a tie (the second round's workgroups start while the first round's stragglers finish).  88 items x 32 slices: 8
      // chunks of 11 = exactly one round, 11 item-times, where two rounds of 6-item chunks take 12 (measured 1620 vs 1800 us);
      // 64 items: two rounds of 4 (1150 us; one round of 8: 1185).
      const long slots = (long)ncu * (8 / c2);
      const long cap = kPartialsFloats / rowf;
      long bchunk = B, best = -1;
      for (int r : {2, 1, 3, 4}) {
        long nc = (r * slots) / nslices;
        if (nc > cap) nc = cap;
        if (nc < 1) nc = 1;
        if (nc > B) nc = B;
        const long bc = (B + nc - 1) / nc;
        nc = (B + bc - 1) / bc;
        const long cost = ((nc * nslices + slots - 1) / slots) * bc;
        if (best < 0 || cost < best) { best = cost; bchunk = bc; }
      }
      const long nchunks = (B + bchunk - 1) / bchunk;
      if (plds > 65536) hipFuncSetAttribute((const void*)peg_wgrad_plane_kernel<c2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)plds);
      hipLaunchKernelGGL(peg_wgrad_plane_kernel<c2>, dim3((unsigned)(nchunks * nslices)), dim3(threads), plds, st,
                         dy, x, partials, g, strips, (int)bchunk);
      return finish((int)nchunks);
    }
  }
  // sweep kernel: one partial row per (b, h, w chunk) group, as many groups per launch as the scratch holds
  const long ngroups = B * H * ((W + PEG_CWW - 1) / PEG_CWW);
  long cap = kPartialsFloats / rowf;
  if (cap < 1) return (int)hipErrorInvalidValue;
  for (long g0 = 0; g0 < ngroups; g0 += cap) {
    const int n = (int)((ngroups - g0 < cap) ? ngroups - g0 : cap);
    hipLaunchKernelGGL(peg_bwd_weight_kernel, dim3(grid_for((long)n * g.d4)), dim3(256), 0, st, dy, x, partials, g, g0, n);
    const int r = finish(n);
    if (r) return r;
  }
  return 0;
}

}  // extern "C"
