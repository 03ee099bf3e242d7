// LayerNorm (with / without bias) and per-head cosine normalisation, forward + backward.
// All HBM-bound: one wave per row, float4 / 16-byte accesses, rows kept in registers between the
// statistics pass and the normalise pass (dim <= 4096).  Column reductions (dgamma, dbeta, dscale) are accumulated per
// wave across a grid-stride loop, combined over the workgroup's waves in a fixed order and stored as one row of
// `partials` per workgroup; ctclip_reduce_partials (tail.hip) adds the rows up in index order.  No atomics: the same
// inputs give the same bits.
#include "common.h"

namespace {

constexpr int LN_MAXV = 16;  // float4 per lane -> dim <= 64*4*16 = 4096

// Token re-ordering folded into the row addressing (reference ctvit.py:96,99,101: the rearranges between the spatial and
// the temporal transformer).  Rows seen as [B][A][C]; the other side of the kernel holds them as [B][C][A].  A == 0: none.
__device__ __forceinline__ long swapped_row(int row, int a_ext, int c_ext) {
  if (a_ext <= 0) return row;
  const int ac = a_ext * c_ext, b = row / ac, rem = row - b * ac, a = rem / c_ext, c = rem - a * c_ext;
  return (long)b * ac + (long)c * a_ext + a;
}

// reference src/utils/attention.py:27-34,46 ; src/utils/ctvit.py:49,51 ; transformers BertLayerNorm
template <int LN_NV>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, bf16_t* __restrict__ y16,
                                                            float* __restrict__ y32, float* __restrict__ mean,
                                                            float* __restrict__ rstd, int rows, int dim, float eps,
                                                            int swap_a, int swap_c) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const long orow = swapped_row(row, swap_a, swap_c);   // where the normalised row goes (statistics stay in input order)
  const int nv = dim >> 2;  // float4 count
  const float4* xr = (const float4*)(x + (long)row * dim);
  float4 v[LN_NV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < LN_NV; ++i) {
    const int c = lane + i * 64;
    if (c < nv) {
      v[i] = xr[c];
      s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
  }
  const float mu = wave_sum(s) / (float)dim;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < LN_NV; ++i) {
    const int c = lane + i * 64;
    if (c < nv) {
      const float a = v[i].x - mu, b = v[i].y - mu, cc = v[i].z - mu, d = v[i].w - mu;
      q += (a * a + b * b) + (cc * cc + d * d);
    }
  }
  const float rs = rsqrtf(wave_sum(q) / (float)dim + eps);
  if (lane == 0) {
    if (mean) mean[row] = mu;
    if (rstd) rstd[row] = rs;
  }
#pragma unroll
  for (int i = 0; i < LN_NV; ++i) {
    const int c = lane + i * 64;
    if (c < nv) {
      float4 g = make_float4(1.f, 1.f, 1.f, 1.f), b = make_float4(0.f, 0.f, 0.f, 0.f);
      if (gamma) g = ((const float4*)gamma)[c];            // gamma == NULL: the plain normalised rows (affine part folded into
      if (gamma && beta) b = ((const float4*)beta)[c];     // the projection that follows: ctclip_patch_affine_fold)
      float4 o;
      o.x = (v[i].x - mu) * rs * g.x + b.x;
      o.y = (v[i].y - mu) * rs * g.y + b.y;
      o.z = (v[i].z - mu) * rs * g.z + b.z;
      o.w = (v[i].w - mu) * rs * g.w + b.w;
      if (y32) ((float4*)(y32 + orow * dim))[c] = o;
      if (y16) {
        uint2 p;
        p.x = pack_bf16x2(o.x, o.y);
        p.y = pack_bf16x2(o.z, o.w);
        ((uint2*)(y16 + orow * dim))[c] = p;
      }
    }
  }
}

// dx = dres + rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma
// dgamma += sum_rows dy * xhat ; dbeta += sum_rows dy
// DY16: dy is bf16 (the output of a bf16 data-gradient GEMM) and `dres2` is an optional second, bf16, residual-path term
__device__ __forceinline__ float4 ld_bf16x4(const bf16_t* p) {
  const uint2 v = *(const uint2*)p;
  return make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16),
                     __uint_as_float(v.y & 0xffff0000u));
}

template <int LN_NV, bool DY16>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const void* __restrict__ dy_any, const float* __restrict__ x,
                                                            const float* __restrict__ gamma, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, const float* __restrict__ dres,
                                                            const bf16_t* __restrict__ dres2,
                                                            float* __restrict__ dx, bf16_t* __restrict__ dx16,
                                                            float* __restrict__ partials, int want_beta,
                                                            int rows, int dim, int swap_a, int swap_c) {
  extern __shared__ __attribute__((aligned(16))) float lnred[];   // [4 waves][2][dim]
  const int lane = threadIdx.x & 63;
  const int wave_global = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int nwaves = gridDim.x * 4;
  const int nv = dim >> 2;
  float4 ag[LN_NV], ab[LN_NV];
#pragma unroll
  for (int i = 0; i < LN_NV; ++i) {
    ag[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    ab[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (int row = wave_global; row < rows; row += nwaves) {
    const float mu = mean[row], rs = rstd[row];
    const float4* xr = (const float4*)(x + (long)row * dim);
    const long yrow = swapped_row(row, swap_a, swap_c);           // dy lives in the forward's OUTPUT order
    const float4* dr = (const float4*)((const float*)dy_any + yrow * dim);
    const bf16_t* dr16 = (const bf16_t*)dy_any + yrow * dim;
    float4 xh[LN_NV], gg[LN_NV], rres[LN_NV];
    float s1 = 0.f, s2 = 0.f;
    // the residual-path terms are requested with the row itself, not after the two reductions: one memory round trip per
    // row instead of two
#pragma unroll
    for (int i = 0; i < LN_NV; ++i) {
      const int c = lane + i * 64;
      rres[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (c < nv) {
        if (dres) rres[i] = ((const float4*)(dres + (long)row * dim))[c];
        if (DY16 && dres2) {
          const float4 r2 = ld_bf16x4(dres2 + (long)row * dim + 4 * c);
          rres[i].x += r2.x; rres[i].y += r2.y; rres[i].z += r2.z; rres[i].w += r2.w;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < LN_NV; ++i) {
      const int c = lane + i * 64;
      if (c < nv) {
        const float4 xv = xr[c], dv = DY16 ? ld_bf16x4(dr16 + 4 * c) : dr[c], gm = ((const float4*)gamma)[c];
        xh[i] = make_float4((xv.x - mu) * rs, (xv.y - mu) * rs, (xv.z - mu) * rs, (xv.w - mu) * rs);
        gg[i] = make_float4(dv.x * gm.x, dv.y * gm.y, dv.z * gm.z, dv.w * gm.w);
        s1 += (gg[i].x + gg[i].y) + (gg[i].z + gg[i].w);
        s2 += (gg[i].x * xh[i].x + gg[i].y * xh[i].y) + (gg[i].z * xh[i].z + gg[i].w * xh[i].w);
        ag[i].x += dv.x * xh[i].x; ag[i].y += dv.y * xh[i].y; ag[i].z += dv.z * xh[i].z; ag[i].w += dv.w * xh[i].w;
        ab[i].x += dv.x; ab[i].y += dv.y; ab[i].z += dv.z; ab[i].w += dv.w;
      }
    }
    const float m1 = wave_sum(s1) / (float)dim, m2 = wave_sum(s2) / (float)dim;
#pragma unroll
    for (int i = 0; i < LN_NV; ++i) {
      const int c = lane + i * 64;
      if (c < nv) {
        float4 o;
        o.x = rs * (gg[i].x - m1 - xh[i].x * m2);
        o.y = rs * (gg[i].y - m1 - xh[i].y * m2);
        o.z = rs * (gg[i].z - m1 - xh[i].z * m2);
        o.w = rs * (gg[i].w - m1 - xh[i].w * m2);
        o.x += rres[i].x; o.y += rres[i].y; o.z += rres[i].z; o.w += rres[i].w;
        if (dx) ((float4*)(dx + (long)row * dim))[c] = o;
        if (dx16) {
          uint2 p;
          p.x = pack_bf16x2(o.x, o.y);
          p.y = pack_bf16x2(o.z, o.w);
          ((uint2*)(dx16 + (long)row * dim))[c] = p;
        }
      }
    }
  }
  float* mine = lnred + (threadIdx.x >> 6) * 2 * dim;                // this wave's [gamma | beta] row
#pragma unroll
  for (int i = 0; i < LN_NV; ++i) {
    const int c = lane + i * 64;
    if (c < nv) {
      ((float4*)mine)[c] = ag[i];
      ((float4*)(mine + dim))[c] = ab[i];
    }
  }
  __syncthreads();
  const int width = want_beta ? 2 * dim : dim;
  float* prow = partials + (long)blockIdx.x * width;                 // [dgamma (dim) | dbeta (dim)]
  for (int i = threadIdx.x; i < width; i += 256)
    prow[i] = (lnred[i] + lnred[2 * dim + i]) + (lnred[4 * dim + i] + lnred[6 * dim + i]);
}

// LayerNorm backward from the saved NORMALISED rows (bf16, the operand of the projection that follows -- its affine part folded
// into that projection's weight, so dy here is the gradient w.r.t. xhat and there is no d(gamma) / d(beta) to reduce):
//   dx = dres + dres2 + rstd (dy - mean(dy) - xhat mean(dy xhat))
// Reads 2 + 2 (+ 4 + 2) bytes per element where the form above reads the f32 input row (4) as well, keeps no partial sums and
// needs neither x nor the mean: the f32 input of the LayerNorm is not kept for the backward at all.
template <int LN_NV>
__global__ __launch_bounds__(256) void layernorm_bwd_xhat_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ xhat,
                                                                 const float* __restrict__ rstd, const float* __restrict__ dres,
                                                                 const bf16_t* __restrict__ dres2, float* __restrict__ dx,
                                                                 bf16_t* __restrict__ dx16, int rows, int dim) {
  const int lane = threadIdx.x & 63;
  const int nv = dim >> 2;
  for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += gridDim.x * 4) {
    const float rs = rstd[row];
    const long base = (long)row * dim;
    float4 xh[LN_NV], gg[LN_NV], rres[LN_NV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < LN_NV; ++i) {
      const int c = lane + i * 64;
      rres[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (c < nv) {
        if (dres) rres[i] = ((const float4*)(dres + base))[c];
        if (dres2) {
          const float4 r2 = ld_bf16x4(dres2 + base + 4 * c);
          rres[i].x += r2.x; rres[i].y += r2.y; rres[i].z += r2.z; rres[i].w += r2.w;
        }
        xh[i] = ld_bf16x4(xhat + base + 4 * c);
        gg[i] = ld_bf16x4(dy + base + 4 * c);
        s1 += (gg[i].x + gg[i].y) + (gg[i].z + gg[i].w);
        s2 += (gg[i].x * xh[i].x + gg[i].y * xh[i].y) + (gg[i].z * xh[i].z + gg[i].w * xh[i].w);
      }
    }
    const float m1 = wave_sum(s1) / (float)dim, m2 = wave_sum(s2) / (float)dim;
#pragma unroll
    for (int i = 0; i < LN_NV; ++i) {
      const int c = lane + i * 64;
      if (c < nv) {
        float4 o;
        o.x = rs * (gg[i].x - m1 - xh[i].x * m2) + rres[i].x;
        o.y = rs * (gg[i].y - m1 - xh[i].y * m2) + rres[i].y;
        o.z = rs * (gg[i].z - m1 - xh[i].z * m2) + rres[i].z;
        o.w = rs * (gg[i].w - m1 - xh[i].w * m2) + rres[i].w;
        if (dx) ((float4*)(dx + base))[c] = o;
        if (dx16) {
          uint2 p;
          p.x = pack_bf16x2(o.x, o.y);
          p.y = pack_bf16x2(o.z, o.w);
          ((uint2*)(dx16 + base))[c] = p;
        }
      }
    }
  }
}

// element offset of (row, head) in a [rows, ld] row-major matrix (hm_n == 0) or in the head-major layout
// [sequence][head][token][D] of attention_hm.hip (hm_n = tokens per sequence)
__device__ __forceinline__ long head_off(long row, int head, int H, int D, long ld, int hm_n) {
  if (hm_n <= 0) return row * ld + (long)head * D;
  const long sq = row / hm_n, tok = row - sq * hm_n;
  return ((sq * H + head) * hm_n + tok) * D;
}

// ---- per-head cosine normalisation: y = x / max(|x|,1e-12) * scale[d] * mult --------------------
// reference src/utils/attention.py:151-153 (+ the fixed `scale = 8` of :98,155 folded into q via mult)
// LPH lanes share one (row, head); each lane owns 8 consecutive bf16 (16 bytes).  D = 8*LPH.
template <int LPH>
__global__ __launch_bounds__(256) void headnorm_fwd_kernel(const bf16_t* __restrict__ x, const float* __restrict__ scale,
                                                           bf16_t* __restrict__ y, float* __restrict__ inv_norm,
                                                           long npairs, int H, long ldx, long ldy, float mult, int x_hm,
                                                           int y_hm) {
  const long gid = (long)blockIdx.x * 256 + threadIdx.x;
  const long pair = gid / LPH;
  const int sub = (int)(gid % LPH);
  const bool ok = pair < npairs;
  const long row = ok ? pair / H : 0;
  const int head = ok ? (int)(pair % H) : 0;
  const int D = LPH * 8;
  float f[8];
  uint4 raw = make_uint4(0, 0, 0, 0);
  if (ok) raw = *(const uint4*)(x + head_off(row, head, H, D, ldx, x_hm) + sub * 8);
  const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[2 * i] = __uint_as_float(w[i] << 16);
    f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    ss += f[2 * i] * f[2 * i] + f[2 * i + 1] * f[2 * i + 1];
  }
#pragma unroll
  for (int o = LPH >> 1; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
  const float inv = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
  if (!ok) return;
  if (sub == 0) inv_norm[pair] = inv;
  uint4 o;
  uint32_t* ow = (uint32_t*)&o;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float s0 = scale[sub * 8 + 2 * i] * mult, s1 = scale[sub * 8 + 2 * i + 1] * mult;
    ow[i] = pack_bf16x2(f[2 * i] * inv * s0, f[2 * i + 1] * inv * s1);
  }
  *(uint4*)(y + head_off(row, head, H, D, ldy, y_hm) + sub * 8) = o;
}

// dx = inv * (du - u (u . du)),  u = x*inv,  du = dy * scale * mult ;  dscale[d] += sum dy * u * mult
// x_normed: `x` is not the raw projection but the forward's OUTPUT y = u scale mult (the q / k GEMM normalised in its epilogue,
// ctclip_gemm_bf16_headnorm, and nothing else of the projection was kept): u = y / (scale mult), the raw row is u / inv.  A
// channel whose learned scale is exactly 0 has lost its u (y = 0): it gets u = 0, i.e. no gradient through that channel.
// LNX (ctclip_headnorm_bwd_ln): the eight heads of a row are the 32 lanes of half a wave; besides dx the kernel writes the row
// scaled by rstd and the two row constants of the LayerNorm backward that runs in the next GEMM's epilogue.
struct HeadLnx { const float* rstd; const float* wbar; bf16_t* dxs; long lddxs; float* c1; float* c2; float inv_dim; };
template <int LPH, bool LNX = false>
__global__ __launch_bounds__(256) void headnorm_bwd_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                           const float* __restrict__ inv_norm, const float* __restrict__ scale,
                                                           bf16_t* __restrict__ dx, float* __restrict__ partials,
                                                           long npairs, int H, long lddy, long ldx, long lddx, float mult,
                                                           int x_hm, int x_normed, HeadLnx lx) {
  __shared__ float red[256 / LPH][LPH * 8 + 1];                      // [pair slot of the workgroup][d]
  const int D = LPH * 8;
  const int sub = threadIdx.x % LPH;
  float acc[8], rsc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    acc[i] = 0.f;
    const float sm = scale[sub * 8 + i] * mult;
    rsc[i] = sm != 0.f ? 1.0f / sm : 0.f;
  }
  const long stride = (long)gridDim.x * (256 / LPH);
  // every lane of an LPH group walks the same pair sequence, so the shuffles below stay convergent
  const long iters = (npairs + stride - 1) / stride;
  long pair = (long)blockIdx.x * (256 / LPH) + threadIdx.x / LPH;
  for (long it = 0; it < iters; ++it, pair += stride) {
    const bool ok = pair < npairs;
    const long row = ok ? pair / H : 0;
    const int head = ok ? (int)(pair % H) : 0;
    uint4 rx = make_uint4(0, 0, 0, 0), rd = make_uint4(0, 0, 0, 0);
    float inv = 0.f;
    if (ok) {
      rx = *(const uint4*)(x + head_off(row, head, H, D, ldx, x_hm) + sub * 8);
      rd = *(const uint4*)(dy + row * lddy + head * D + sub * 8);
      inv = inv_norm[pair];
    }
    const uint32_t wx[4] = {rx.x, rx.y, rx.z, rx.w}, wd[4] = {rd.x, rd.y, rd.z, rd.w};
    float u[8], du[8], dot = 0.f;
    float renorm = inv;
    if (x_normed) {
      // y / (scale mult) is the unit row up to the bf16 rounding of y: divide that rounding's radial part out again, so that u
      // is EXACTLY a unit vector and dx below is the exact gradient at a nearby point, as it is for the raw-row form
      float nn = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float x0 = __uint_as_float(wx[i] << 16) * rsc[2 * i], x1 = __uint_as_float(wx[i] & 0xffff0000u) * rsc[2 * i + 1];
        nn += x0 * x0 + x1 * x1;
      }
#pragma unroll
      for (int o = LPH >> 1; o > 0; o >>= 1) nn += __shfl_xor(nn, o, 64);
      renorm = nn > 0.f ? rsqrtf(nn) : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float x0 = __uint_as_float(wx[i] << 16), x1 = __uint_as_float(wx[i] & 0xffff0000u);
      const float d0 = __uint_as_float(wd[i] << 16), d1 = __uint_as_float(wd[i] & 0xffff0000u);
      u[2 * i] = x0 * (x_normed ? rsc[2 * i] * renorm : inv); u[2 * i + 1] = x1 * (x_normed ? rsc[2 * i + 1] * renorm : inv);
      du[2 * i] = d0 * scale[sub * 8 + 2 * i] * mult; du[2 * i + 1] = d1 * scale[sub * 8 + 2 * i + 1] * mult;
      acc[2 * i] += d0 * u[2 * i] * mult; acc[2 * i + 1] += d1 * u[2 * i + 1] * mult;
      dot += u[2 * i] * du[2 * i] + u[2 * i + 1] * du[2 * i + 1];
    }
#pragma unroll
    for (int o = LPH >> 1; o > 0; o >>= 1) dot += __shfl_xor(dot, o, 64);
    uint4 o = make_uint4(0, 0, 0, 0);
    uint32_t* ow = (uint32_t*)&o;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      ow[i] = pack_bf16x2(inv * (du[2 * i] - u[2 * i] * dot), inv * (du[2 * i + 1] - u[2 * i + 1] * dot));
    if (ok) *(uint4*)(dx + row * lddx + head * D + sub * 8) = o;
    if constexpr (LNX) {
      // H * LPH == 32: the lanes of this row are an aligned group of 32
      const float rs = ok ? lx.rstd[row] : 0.f;
      const float nrm = inv > 0.f ? 1.0f / inv : 0.f;                    // |raw row| (x_normed: the raw row is u |row|)
      float s1 = 0.f, s2 = 0.f, v[8];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        v[2 * i] = __uint_as_float(ow[i] << 16); v[2 * i + 1] = __uint_as_float(ow[i] & 0xffff0000u);     // dx as the GEMM will read it
        const float x0 = x_normed ? u[2 * i] * nrm : __uint_as_float(wx[i] << 16);
        const float x1 = x_normed ? u[2 * i + 1] * nrm : __uint_as_float(wx[i] & 0xffff0000u);
        const int k = head * D + sub * 8 + 2 * i;
        s1 += v[2 * i] * lx.wbar[k] + v[2 * i + 1] * lx.wbar[k + 1];
        s2 += v[2 * i] * x0 + v[2 * i + 1] * x1;
      }
#pragma unroll
      for (int m = 1; m < 32; m <<= 1) { s1 += __shfl_xor(s1, m, 64); s2 += __shfl_xor(s2, m, 64); }
      if (ok) {
        uint4 os;
        uint32_t* osw = (uint32_t*)&os;
#pragma unroll
        for (int i = 0; i < 4; ++i) osw[i] = pack_bf16x2(rs * v[2 * i], rs * v[2 * i + 1]);
        *(uint4*)(lx.dxs + row * lx.lddxs + head * D + sub * 8) = os;
        if (head == 0 && sub == 0) { lx.c1[row] = rs * lx.inv_dim * s1; lx.c2[row] = rs * lx.inv_dim * s2; }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) red[threadIdx.x / LPH][sub * 8 + i] = acc[i];
  __syncthreads();
  if (threadIdx.x < D) {                                             // the workgroup's pair slots in slot order
    float s = 0.f;
    for (int g = 0; g < 256 / LPH; ++g) s += red[g][threadIdx.x];
    partials[(long)blockIdx.x * D + threadIdx.x] = s;
  }
}

}  // namespace

// workgroups of the LayerNorm backward: every wave at least 8 rows (the partial rows are summed by a second kernel whose
// time grows with their number), at most 1024, and never more partial rows than the scratch holds
static int ln_bwd_blocks(int rows, int dim) {
  long blocks = (rows + 31) / 32;
  if (blocks > 1024) blocks = 1024;
  const long fit = kPartialsFloats / (2L * dim);
  if (blocks > fit) blocks = fit;
  return blocks < 1 ? 1 : (int)blocks;
}

// second stage: d(gamma) and, if asked for, d(beta) from the [blocks][width] partial rows
static int ln_bwd_finish(const float* partials, int blocks, int dim, float* dgamma, float* dbeta, hipStream_t st) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return (int)e;
  const int width = dbeta ? 2 * dim : dim;
  int r = ctclip_reduce_partials(partials, blocks, width, dim, dgamma, st);
  if (r == 0 && dbeta) r = ctclip_reduce_partials(partials + dim, blocks, width, dim, dbeta, st);
  return r;
}

template <bool DY16>
static int ln_bwd_launch(const void* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                         const float* dres, const bf16_t* dres2, float* dx, bf16_t* dx16, float* dgamma, float* dbeta,
                         float* partials, int rows, int dim, int A, int C, hipStream_t st) {
  if (!partials || !dgamma) return (int)hipErrorInvalidValue;
  const int blocks = ln_bwd_blocks(rows, dim);
  const size_t lnlds = (size_t)8 * dim * sizeof(float);
#define LN_BWD(NV)                                                                                                       \
  do {                                                                                                                   \
    if (lnlds > 65536) hipFuncSetAttribute((const void*)layernorm_bwd_kernel<NV, DY16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lnlds); \
    hipLaunchKernelGGL((layernorm_bwd_kernel<NV, DY16>), dim3(blocks), dim3(256), lnlds, st, dy, x, gamma, mean, rstd, dres, \
                       dres2, dx, dx16, partials, dbeta ? 1 : 0, rows, dim, A, C);                                       \
  } while (0)
  const int nv = (dim / 4 + 63) / 64;
  if (nv <= 1) LN_BWD(1); else if (nv <= 2) LN_BWD(2); else if (nv <= 3) LN_BWD(3); else if (nv <= 4) LN_BWD(4); else LN_BWD(16);
#undef LN_BWD
  return ln_bwd_finish(partials, blocks, dim, dgamma, dbeta, st);
}

extern "C" {

int ctclip_layernorm_fwd(const float* x, const float* gamma, const float* beta, void* y_bf16, float* y_f32,
                         float* mean, float* rstd, int rows, int dim, float eps, void* stream) {
  if (rows <= 0) return 0;
  if ((dim & 3) || dim > 64 * 4 * LN_MAXV) return (int)hipErrorInvalidValue;
#define LN_FWD(NV)                                                                                              \
  hipLaunchKernelGGL(layernorm_fwd_kernel<NV>, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, gamma, \
                     beta, (bf16_t*)y_bf16, y_f32, mean, rstd, rows, dim, eps, 0, 0)
  const int nv = (dim / 4 + 63) / 64;
  if (nv <= 1) LN_FWD(1); else if (nv <= 2) LN_FWD(2); else if (nv <= 3) LN_FWD(3); else if (nv <= 4) LN_FWD(4); else LN_FWD(16);
#undef LN_FWD
  CTCLIP_CHECK_LAUNCH();
}

int ctclip_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                         const float* dres, float* dx, void* dx_bf16, float* dgamma, float* dbeta, int rows, int dim,
                         float* partials, void* stream) {
  if (rows <= 0) return 0;
  if ((dim & 3) || dim > 64 * 4 * LN_MAXV) return (int)hipErrorInvalidValue;
  return ln_bwd_launch<false>(dy, x, gamma, mean, rstd, dres, nullptr, dx, (bf16_t*)dx_bf16, dgamma, dbeta, partials, rows,
                              dim, 0, 0, (hipStream_t)stream);
}

int ctclip_layernorm_swap_fwd(const float* x, const float* gamma, const float* beta, float* y_f32, float* mean, float* rstd,
                              int rows, int dim, float eps, int A, int C, void* stream) {
  if (rows <= 0) return 0;
  if ((dim & 3) || dim > 64 * 4 * LN_MAXV || A <= 0 || C <= 0 || rows % (A * C)) return (int)hipErrorInvalidValue;
#define LN_FWD(NV)                                                                                              \
  hipLaunchKernelGGL(layernorm_fwd_kernel<NV>, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, gamma, \
                     beta, (bf16_t*)nullptr, y_f32, mean, rstd, rows, dim, eps, A, C)
  const int nv = (dim / 4 + 63) / 64;
  if (nv <= 1) LN_FWD(1); else if (nv <= 2) LN_FWD(2); else if (nv <= 3) LN_FWD(3); else if (nv <= 4) LN_FWD(4); else LN_FWD(16);
#undef LN_FWD
  CTCLIP_CHECK_LAUNCH();
}

int ctclip_layernorm_swap_bwd(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                              float* dx, void* dx_bf16, float* dgamma, float* dbeta, int rows, int dim, int A, int C,
                              float* partials, void* stream) {
  if (rows <= 0) return 0;
  if ((dim & 3) || dim > 64 * 4 * LN_MAXV || A <= 0 || C <= 0 || rows % (A * C)) return (int)hipErrorInvalidValue;
  return ln_bwd_launch<false>(dy, x, gamma, mean, rstd, nullptr, nullptr, dx, (bf16_t*)dx_bf16, dgamma, dbeta, partials, rows,
                              dim, A, C, (hipStream_t)stream);
}

int ctclip_layernorm_bwd_bf16(const void* dy_bf16, const float* x, const float* gamma, const float* mean, const float* rstd,
                              const float* dres, const void* dres2_bf16, float* dx, void* dx_bf16, float* dgamma,
                              float* dbeta, int rows, int dim, float* partials, void* stream) {
  if (rows <= 0) return 0;
  if ((dim & 3) || dim > 64 * 4 * LN_MAXV) return (int)hipErrorInvalidValue;
  return ln_bwd_launch<true>(dy_bf16, x, gamma, mean, rstd, dres, (const bf16_t*)dres2_bf16, dx, (bf16_t*)dx_bf16, dgamma,
                             dbeta, partials, rows, dim, 0, 0, (hipStream_t)stream);
}

int ctclip_layernorm_bwd_xhat(const void* dy_bf16, const void* xhat_bf16, const float* rstd, const float* dres,
                              const void* dres2_bf16, float* dx, void* dx_bf16, int rows, int dim, void* stream) {
  if (rows <= 0) return 0;
  if ((dim & 3) || dim > 64 * 4 * LN_MAXV) return (int)hipErrorInvalidValue;
  long blocks = ((long)rows + 3) / 4;
  if (blocks > 8192) blocks = 8192;
#define LN_BWX(NV)                                                                                                         \
  hipLaunchKernelGGL(layernorm_bwd_xhat_kernel<NV>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,            \
                     (const bf16_t*)dy_bf16, (const bf16_t*)xhat_bf16, rstd, dres, (const bf16_t*)dres2_bf16, dx,           \
                     (bf16_t*)dx_bf16, rows, dim)
  const int nv = (dim / 4 + 63) / 64;
  if (nv <= 1) LN_BWX(1); else if (nv <= 2) LN_BWX(2); else if (nv <= 3) LN_BWX(3); else if (nv <= 4) LN_BWX(4); else LN_BWX(16);
#undef LN_BWX
  CTCLIP_CHECK_LAUNCH();
}

int ctclip_headnorm_fwd(const void* x, const float* scale, void* y, float* inv_norm, long rows, int heads, int dhead,
                        long ldx, long ldy, float mult, int x_hm_n, int y_hm_n, void* stream) {
  const long npairs = rows * heads;
  if (npairs <= 0) return 0;
  if (dhead != 32 && dhead != 64) return (int)hipErrorInvalidValue;
  if ((x_hm_n > 0 && rows % x_hm_n) || (y_hm_n > 0 && rows % y_hm_n)) return (int)hipErrorInvalidValue;
  const int lph = dhead / 8;
  const long threads = npairs * lph;
  dim3 grid((unsigned)((threads + 255) / 256)), block(256);
  if (lph == 4)
    hipLaunchKernelGGL(headnorm_fwd_kernel<4>, grid, block, 0, (hipStream_t)stream, (const bf16_t*)x, scale, (bf16_t*)y,
                       inv_norm, npairs, heads, ldx, ldy, mult, x_hm_n, y_hm_n);
  else
    hipLaunchKernelGGL(headnorm_fwd_kernel<8>, grid, block, 0, (hipStream_t)stream, (const bf16_t*)x, scale, (bf16_t*)y,
                       inv_norm, npairs, heads, ldx, ldy, mult, x_hm_n, y_hm_n);
  CTCLIP_CHECK_LAUNCH();
}

int ctclip_headnorm_bwd(const void* dy, const void* x, const float* inv_norm, const float* scale, void* dx,
                        float* dscale, long rows, int heads, int dhead, long lddy, long ldx, long lddx, float mult,
                        int x_hm_n, int x_normed, float* partials, void* stream) {
  const long npairs = rows * heads;
  if (npairs <= 0) return 0;
  if ((dhead != 32 && dhead != 64) || !partials || (x_hm_n > 0 && rows % x_hm_n)) return (int)hipErrorInvalidValue;
  const int lph = dhead / 8;
  long blocks = (npairs + (256 / lph) - 1) / (256 / lph);
  if (blocks > 2048) blocks = 2048;
  dim3 grid((unsigned)blocks), block(256);
  if (lph == 4)
    hipLaunchKernelGGL(headnorm_bwd_kernel<4>, grid, block, 0, (hipStream_t)stream, (const bf16_t*)dy, (const bf16_t*)x,
                       inv_norm, scale, (bf16_t*)dx, partials, npairs, heads, lddy, ldx, lddx, mult, x_hm_n, x_normed, HeadLnx{});
  else
    hipLaunchKernelGGL(headnorm_bwd_kernel<8>, grid, block, 0, (hipStream_t)stream, (const bf16_t*)dy, (const bf16_t*)x,
                       inv_norm, scale, (bf16_t*)dx, partials, npairs, heads, lddy, ldx, lddx, mult, x_hm_n, x_normed, HeadLnx{});
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return (int)e;
  return ctclip_reduce_partials(partials, (int)blocks, dhead, dhead, dscale, (hipStream_t)stream);
}

int ctclip_headnorm_bwd_ln(const void* dy, const void* x, const float* inv_norm, const float* scale, void* dx,
                           float* dscale, long rows, int heads, int dhead, long lddy, long ldx, long lddx, float mult,
                           const float* rstd, const float* wbar, int ln_dim, void* dx_scaled, long lddxs, float* c1, float* c2,
                           int x_hm_n, int x_normed, float* partials, void* stream) {
  const long npairs = rows * heads;
  if (npairs <= 0) return 0;
  if (dhead != 32 || heads != 8 || !partials || !rstd || !wbar || !dx_scaled || !c1 || !c2 || ln_dim <= 0 || (lddxs & 7) ||
      (((uintptr_t)dx_scaled) & 15) || (x_hm_n > 0 && rows % x_hm_n))
    return (int)hipErrorInvalidValue;
  long blocks = (npairs + 63) / 64;
  if (blocks > 2048) blocks = 2048;
  HeadLnx lx{rstd, wbar, (bf16_t*)dx_scaled, lddxs, c1, c2, 1.0f / (float)ln_dim};
  hipLaunchKernelGGL((headnorm_bwd_kernel<4, true>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy,
                     (const bf16_t*)x, inv_norm, scale, (bf16_t*)dx, partials, npairs, heads, lddy, ldx, lddx, mult, x_hm_n, x_normed,
                     lx);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return (int)e;
  return ctclip_reduce_partials(partials, (int)blocks, dhead, dhead, dscale, (hipStream_t)stream);
}

}  // extern "C"
