// CT-ViT tubelet patch embedding, stage 1: gather + LayerNorm(c*pt*p1*p2) -> bf16 GEMM operand.
// reference src/utils/ctvit.py:44-49: Rearrange 'b c (t pt) (h p1) (w p2) -> b t h w (c pt p1 p2)' then
// nn.LayerNorm over the 4000 voxels of a tubelet.  (The Linear(4000,512)+bias is ctclip_gemm_bf16, the
// trailing LayerNorm(512) is ctclip_layernorm_fwd.)
//
// This is the only kernel of the step that touches the raw volume, and it is HBM-bound.  A workgroup
// owns `tpb` horizontally adjacent tubelets of one (b,t,h): it reads c*pt*p1 fully contiguous runs of
// tpb*p2 voxels (each voxel exactly once), keeps them in LDS in FEATURE order (the einops permutation is
// folded into the LDS write address, never into HBM), computes two-pass mean / rstd per tubelet and writes
// normalised rows with coalesced stores.  Feature index = ((c*pt + pt_i)*p1 + p1_i)*p2 + p2_i.
#include "common.h"
#include <stdlib.h>

namespace {

struct PatchGeom {
  int B, C, Dz, Hy, Wx, pt, p;
  int Tt, Ht, Wt, F, tpb, wgroups;
  long ldA;
  float eps;
  unsigned m_p, m_ptp, m_vpr, m_cpt;   // floor(2^32 / d) + 1 for d = p, pt*p, 16-byte vectors per LDS row, chunks per A row:
};                                      // x / d == __umulhi(x, m) for x * d < 2^32 (one v_mul_hi instead of a ~40-instruction divide)

// d == 1 has no 32-bit magic (2^32 + 1): it is encoded as m = 0 and divides by passing x through (one 16-byte vector per
// LDS row, or an 8-wide A row, are legal tiny geometries)
__host__ __device__ inline unsigned magic_of(unsigned d) { return d <= 1u ? 0u : (unsigned)(0x100000000ull / d) + 1u; }
__device__ __forceinline__ int fdiv(int x, unsigned m) { return m ? (int)__umulhi((unsigned)x, m) : x; }

template <typename T> __device__ __forceinline__ float ldf(T v);
template <> __device__ __forceinline__ float ldf<float>(float v) { return v; }
template <> __device__ __forceinline__ float ldf<bf16_t>(bf16_t v) { return bf16_to_f32(v); }

template <typename TIN>
__device__ __forceinline__ void gather_block(TIN* buf, const TIN* __restrict__ vol, const PatchGeom& g, int b, int t, int h,
                                             int w0, int ntok, int tid) {
  const int nrows = g.C * g.pt * g.p, rowlen = ntok * g.p;
  for (int e = tid; e < nrows * rowlen; e += 256) {
    const int rowid = e / rowlen, col = e - rowid * rowlen;
    const int c = rowid / (g.pt * g.p), rem = rowid - c * g.pt * g.p;
    const int pti = rem / g.p, p1i = rem - pti * g.p;
    const long src = ((((long)b * g.C + c) * g.Dz + t * g.pt + pti) * g.Hy + h * g.p + p1i) * g.Wx + (long)w0 * g.p + col;
    const int tok = col / g.p, p2i = col - tok * g.p;
    buf[tok * g.F + rowid * g.p + p2i] = vol[src];
  }
}

template <typename TIN>
__global__ __launch_bounds__(256) void patch_ln_fwd_kernel(const TIN* __restrict__ vol, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, bf16_t* __restrict__ A,
                                                           float* __restrict__ mean, float* __restrict__ rstd, PatchGeom g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  TIN* buf = (TIN*)smem;
  float* smean = (float*)(smem + (size_t)g.tpb * g.F * sizeof(TIN));
  float* srstd = smean + g.tpb;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int bid = blockIdx.x;
  const int wg = bid % g.wgroups; bid /= g.wgroups;
  const int h = bid % g.Ht; bid /= g.Ht;
  const int t = bid % g.Tt;
  const int b = bid / g.Tt;
  const int w0 = wg * g.tpb;
  const int ntok = min(g.tpb, g.Wt - w0);
  gather_block<TIN>(buf, vol, g, b, t, h, w0, ntok, tid);
  __syncthreads();
  const long row0 = (((long)b * g.Tt + t) * g.Ht + h) * g.Wt + w0;
  for (int tok = wave; tok < ntok; tok += 4) {
    const TIN* r = buf + tok * g.F;
    float s = 0.f;
    for (int f = lane; f < g.F; f += 64) s += ldf<TIN>(r[f]);
    const float mu = wave_sum(s) / (float)g.F;
    float q = 0.f;
    for (int f = lane; f < g.F; f += 64) {
      const float d = ldf<TIN>(r[f]) - mu;
      q += d * d;
    }
    const float rs = rsqrtf(wave_sum(q) / (float)g.F + g.eps);
    if (lane == 0) {
      smean[tok] = mu; srstd[tok] = rs;
      mean[row0 + tok] = mu; rstd[row0 + tok] = rs;
    }
  }
  __syncthreads();
  const int half = (int)(g.ldA >> 1);
  for (int e = tid; e < ntok * half; e += 256) {
    const int tok = e / half, f = (e - tok * half) * 2;
    const float mu = smean[tok], rs = srstd[tok];
    float v0 = 0.f, v1 = 0.f;
    if (f < g.F) v0 = (ldf<TIN>(buf[tok * g.F + f]) - mu) * rs * (gamma ? gamma[f] : 1.f) + (gamma ? beta[f] : 0.f);
    if (f + 1 < g.F) v1 = (ldf<TIN>(buf[tok * g.F + f + 1]) - mu) * rs * (gamma ? gamma[f + 1] : 1.f) + (gamma ? beta[f + 1] : 0.f);
    *(uint32_t*)(A + (row0 + tok) * g.ldA + f) = pack_bf16x2(v0, v1);
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Fast path (the production geometry: p even, rows of tpb*p voxels a whole number of 16-byte vectors):
//   phase 1  16-byte coalesced global loads straight into an LDS image kept in VOLUME order  buf[c*pt*p1 row][tpb*p2]
//            (each voxel read from HBM exactly once, no per-element index arithmetic);
//   phase 2  two-pass mean / rstd per tubelet, one wave per tubelet, 8-byte LDS reads;
//   phase 3  the einops permutation happens on the LDS READ side: feature pair (f, f+1) = buf[f / p][tok*p + f % p],
//            normalised, scaled, packed to bf16 and written as 16-byte coalesced rows of the GEMM operand.
// ------------------------------------------------------------------------------------------------------------------
constexpr int PF_THREADS = 512;

template <typename TIN>
__device__ __forceinline__ void load_block_vec(char* buf, const TIN* __restrict__ vol, const PatchGeom& g, int b, int t,
                                               int h, int w0, int tid) {
  const int nrows = g.C * g.pt * g.p;
  const int vpr = (g.tpb * g.p * (int)sizeof(TIN)) / 16;               // 16-byte vectors per row
  for (int e = tid; e < nrows * vpr; e += PF_THREADS) {
    const int rowid = fdiv(e, g.m_vpr), v = e - rowid * vpr;
    const int c = fdiv(rowid, g.m_ptp), rem = rowid - c * g.pt * g.p;
    const int pti = fdiv(rem, g.m_p), p1i = rem - pti * g.p;
    const long src = ((((long)b * g.C + c) * g.Dz + t * g.pt + pti) * g.Hy + h * g.p + p1i) * g.Wx + (long)w0 * g.p;
    *(uint4*)(buf + (size_t)e * 16) = *(const uint4*)((const char*)(vol + src) + v * 16);
  }
}

template <typename TIN>
__device__ __forceinline__ float2 ld_pair(const char* buf, int rlb, int rowid, int col);
template <>
__device__ __forceinline__ float2 ld_pair<bf16_t>(const char* buf, int rlb, int rowid, int col) {
  const uint32_t w = *(const uint32_t*)(buf + (size_t)rowid * rlb + col * 2);
  return make_float2(__uint_as_float(w << 16), __uint_as_float(w & 0xffff0000u));
}
template <>
__device__ __forceinline__ float2 ld_pair<float>(const char* buf, int rlb, int rowid, int col) {
  return *(const float2*)(buf + (size_t)rowid * rlb + col * 4);
}

template <typename TIN>
__device__ __forceinline__ void block_stats(const char* buf, const PatchGeom& g, float* smean, float* srstd, int tid) {
  const int lane = tid & 63, wave = tid >> 6, nw = PF_THREADS / 64;
  const int nrows = g.C * g.pt * g.p, rlb = g.tpb * g.p * (int)sizeof(TIN);
  for (int tok = wave; tok < g.tpb; tok += nw) {
    float s = 0.f;
    for (int r = lane; r < nrows; r += 64)
      for (int c2 = 0; c2 < g.p; c2 += 2) { const float2 v = ld_pair<TIN>(buf, rlb, r, tok * g.p + c2); s += v.x + v.y; }
    const float mu = wave_sum(s) / (float)g.F;
    float q = 0.f;
    for (int r = lane; r < nrows; r += 64)
      for (int c2 = 0; c2 < g.p; c2 += 2) {
        const float2 v = ld_pair<TIN>(buf, rlb, r, tok * g.p + c2);
        q += (v.x - mu) * (v.x - mu) + (v.y - mu) * (v.y - mu);
      }
    const float rs = rsqrtf(wave_sum(q) / (float)g.F + g.eps);
    if (lane == 0) { smean[tok] = mu; srstd[tok] = rs; }
  }
}

template <typename TIN>
__global__ __launch_bounds__(PF_THREADS) void patch_ln_fwd_fast(const TIN* __restrict__ vol, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, bf16_t* __restrict__ A,
                                                               float* __restrict__ mean, float* __restrict__ rstd, PatchGeom g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int nrows = g.C * g.pt * g.p, rlb = g.tpb * g.p * (int)sizeof(TIN);
  char* buf = smem;
  float* smean = (float*)(smem + (size_t)nrows * rlb);
  float* srstd = smean + g.tpb;
  const int tid = threadIdx.x;
  int bid = blockIdx.x;                      // (an XCD-contiguous order of the workgroups that share a volume row: 5050 -> 5130 us)
  const int wg = bid % g.wgroups; bid /= g.wgroups;
  const int h = bid % g.Ht; bid /= g.Ht;
  const int t = bid % g.Tt;
  const int b = bid / g.Tt;
  const int w0 = wg * g.tpb;
  load_block_vec<TIN>(buf, vol, g, b, t, h, w0, tid);
  __syncthreads();
  block_stats<TIN>(buf, g, smean, srstd, tid);
  __syncthreads();
  const long row0 = (((long)b * g.Tt + t) * g.Ht + h) * g.Wt + w0;
  if (tid < g.tpb) { mean[row0 + tid] = smean[tid]; rstd[row0 + tid] = srstd[tid]; }
  const int cpt = (int)(g.ldA >> 3);                                    // 16-byte output chunks per tubelet row
  for (int e = tid; e < g.tpb * cpt; e += PF_THREADS) {
    const int tok = fdiv(e, g.m_cpt), f0 = (e - tok * cpt) * 8;
    const float mu = smean[tok], rs = srstd[tok];
    float o[8];
    if (f0 + 7 < g.F) {
      // eight consecutive features: gamma / beta as two float4 each (they were 16 scalar loads per chunk, the kernel's
      // actual bottleneck), one divide for the first pair and a carry for the rest
      float gm[8] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f}, bt[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (gamma) {                                                     // null: the affine part lives in the projection (folded)
        const float4 g0 = *(const float4*)(gamma + f0), g1 = *(const float4*)(gamma + f0 + 4);
        const float4 b0 = *(const float4*)(beta + f0), b1 = *(const float4*)(beta + f0 + 4);
        gm[0] = g0.x; gm[1] = g0.y; gm[2] = g0.z; gm[3] = g0.w; gm[4] = g1.x; gm[5] = g1.y; gm[6] = g1.z; gm[7] = g1.w;
        bt[0] = b0.x; bt[1] = b0.y; bt[2] = b0.z; bt[3] = b0.w; bt[4] = b1.x; bt[5] = b1.y; bt[6] = b1.z; bt[7] = b1.w;
      }
      int rowid = fdiv(f0, g.m_p), c2 = f0 - rowid * g.p;
#pragma unroll
      for (int k = 0; k < 8; k += 2) {
        const float2 v = ld_pair<TIN>(buf, rlb, rowid, tok * g.p + c2);
        o[k] = (v.x - mu) * rs * gm[k] + bt[k];
        o[k + 1] = (v.y - mu) * rs * gm[k + 1] + bt[k + 1];
        c2 += 2;
        if (c2 >= g.p) { c2 -= g.p; ++rowid; }
      }
    } else {
#pragma unroll
      for (int k = 0; k < 8; k += 2) {
        const int f = f0 + k;
        if (f < g.F) {
          const int rowid = fdiv(f, g.m_p), c2 = f - rowid * g.p;
          const float2 v = ld_pair<TIN>(buf, rlb, rowid, tok * g.p + c2);
          o[k] = (v.x - mu) * rs * (gamma ? gamma[f] : 1.f) + (gamma ? beta[f] : 0.f);
          o[k + 1] = (v.y - mu) * rs * (gamma ? gamma[f + 1] : 1.f) + (gamma ? beta[f + 1] : 0.f);
        } else { o[k] = 0.f; o[k + 1] = 0.f; }
      }
    }
    uint4 pk;
    pk.x = pack_bf16x2(o[0], o[1]); pk.y = pack_bf16x2(o[2], o[3]); pk.z = pack_bf16x2(o[4], o[5]); pk.w = pack_bf16x2(o[6], o[7]);
    *(uint4*)(A + (row0 + tok) * g.ldA + f0) = pk;
  }
}

// ------------------------------------------------------------------------------------------------------------------
// The affine part of LayerNorm(F) folded into the tubelet projection (reference ctvit.py:49-50):
//     z = (xhat * gamma + beta) W^T + b  =  xhat (W * gamma)^T + (b + W beta),      xhat = (x - mean) * rstd
// so the GEMM operand is the plain normalised tubelet row and NOTHING of the 4000-wide backward has to be materialised:
// with G = dz^T xhat (the one weight-gradient GEMM, [N, F] f32) and db = colsum(dz),
//     dW[n][f]  = G[n][f] gamma[f] + db[n] beta[f]
//     dgamma[f] = sum_n W[n][f] G[n][f]            dbeta[f] = sum_n W[n][f] db[n]
// exactly (no division by gamma).  The data-gradient GEMM dz W [tokens, F] and the pass over it that the unfolded form needs
// for d(gamma) / d(beta) (3.6 TFLOP and 2 x 7 GB at 64 pairs) are gone.
// fold: one workgroup per output row n
__global__ __launch_bounds__(256) void patch_affine_fold_kernel(const float* __restrict__ W, const float* __restrict__ bias,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                bf16_t* __restrict__ Wg, float* __restrict__ bfold, int F, long ldw) {
  __shared__ float wsum[4];
  const int n = blockIdx.x;
  float s = 0.f;
  for (int f = threadIdx.x; f < (int)ldw; f += 256) {
    float w = 0.f;
    if (f < F) {
      w = W[(long)n * F + f];
      if (beta) s = fmaf(w, beta[f], s);
      w *= gamma[f];
    }
    Wg[(long)n * ldw + f] = f32_to_bf16(w);
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0 && bfold) bfold[n] = (bias ? bias[n] : 0.f) + ((wsum[0] + wsum[1]) + (wsum[2] + wsum[3]));
}

// backward: a workgroup owns 64 features, its four row lanes split the N output rows (combined in a fixed order)
__global__ __launch_bounds__(256) void patch_affine_bwd_kernel(const float* __restrict__ G, const float* __restrict__ db,
                                                               const float* __restrict__ W, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, float* __restrict__ dW,
                                                               float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                               int N, int F, long ldg, int ncorr) {
  __shared__ float red[2][4][64];
  const int fl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int f = blockIdx.x * 64 + fl;
  float ag = 0.f, ab = 0.f;
  if (f < F) {
    const float gm = gamma[f], bt = beta ? beta[f] : 0.f;
    for (int n = rl; n < N; n += 4) {
      const long e = (long)n * F + f;
      float g = G[(long)n * ldg + f];
      for (int j = 0; j < ncorr; ++j) g -= G[(long)n * ldg + F + j];      // the virtual columns of ctclip_patch_wgrad_fused
      const float w = W[e], d = db ? db[n] : 0.f;
      ag = fmaf(w, g, ag);
      ab = fmaf(w, d, ab);
      dW[e] += fmaf(g, gm, d * bt);
    }
  }
  red[0][rl][fl] = ag;
  red[1][rl][fl] = ab;
  __syncthreads();
  if (rl == 0 && f < F) {
    dgamma[f] += (red[0][0][fl] + red[0][1][fl]) + (red[0][2][fl] + red[0][3][fl]);
    if (dbeta) dbeta[f] += (red[1][0][fl] + red[1][1][fl]) + (red[1][2][fl] + red[1][3][fl]);
  }
}

// d(volume) of gather + LayerNorm (needed only when the input itself is differentiated: integrated gradients,
// reference src/utils/visualizations.py:851-910).  Same block decomposition as patch_ln_fwd_kernel; the tubelets are
// gathered in feature order, dx = rstd * (g - mean(g) - xhat * mean(g * xhat)) with g = dA * gamma is scattered back to
// the voxel each feature came from (every voxel belongs to exactly one tubelet: plain stores, no atomics).
template <typename TIN>
__global__ __launch_bounds__(256) void patch_ln_bwd_dx_kernel(const TIN* __restrict__ vol, const bf16_t* __restrict__ dA, long ldd,
                                                              const float* __restrict__ gamma, const float* __restrict__ mean,
                                                              const float* __restrict__ rstd, float* __restrict__ dvol,
                                                              PatchGeom g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  TIN* buf = (TIN*)smem;
  float* sm1 = (float*)(smem + (size_t)g.tpb * g.F * sizeof(TIN));
  float* sm2 = sm1 + g.tpb;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int bid = blockIdx.x;
  const int wg = bid % g.wgroups; bid /= g.wgroups;
  const int h = bid % g.Ht; bid /= g.Ht;
  const int t = bid % g.Tt;
  const int b = bid / g.Tt;
  const int w0 = wg * g.tpb;
  const int ntok = min(g.tpb, g.Wt - w0);
  gather_block<TIN>(buf, vol, g, b, t, h, w0, ntok, tid);
  __syncthreads();
  const long row0 = (((long)b * g.Tt + t) * g.Ht + h) * g.Wt + w0;
  for (int tok = wave; tok < ntok; tok += 4) {
    const TIN* r = buf + tok * g.F;
    const bf16_t* d = dA + (row0 + tok) * ldd;
    const float mu = mean[row0 + tok], rs = rstd[row0 + tok];
    float s1 = 0.f, s2 = 0.f;
    for (int f = lane; f < g.F; f += 64) {
      const float gg = bf16_to_f32(d[f]) * gamma[f];
      s1 += gg;
      s2 += gg * (ldf<TIN>(r[f]) - mu) * rs;
    }
    s1 = wave_sum(s1) / (float)g.F;
    s2 = wave_sum(s2) / (float)g.F;
    if (lane == 0) { sm1[tok] = s1; sm2[tok] = s2; }
  }
  __syncthreads();
  const int nrows = g.C * g.pt * g.p, rowlen = ntok * g.p;
  for (int e = tid; e < nrows * rowlen; e += 256) {
    const int rowid = e / rowlen, col = e - rowid * rowlen;
    const int c = rowid / (g.pt * g.p), rem = rowid - c * g.pt * g.p;
    const int pti = rem / g.p, p1i = rem - pti * g.p;
    const long dst = ((((long)b * g.C + c) * g.Dz + t * g.pt + pti) * g.Hy + h * g.p + p1i) * g.Wx + (long)w0 * g.p + col;
    const int tok = col / g.p, p2i = col - tok * g.p;
    const int f = rowid * g.p + p2i;
    const float mu = mean[row0 + tok], rs = rstd[row0 + tok];
    const float gg = bf16_to_f32(dA[(row0 + tok) * ldd + f]) * gamma[f];
    const float xh = (ldf<TIN>(buf[tok * g.F + f]) - mu) * rs;
    dvol[dst] = rs * (gg - sm1[tok] - xh * sm2[tok]);
  }
}

// fast-path eligibility + geometry: tpb must divide Wt and give whole 16-byte vectors per row
bool fast_geom(PatchGeom& g, const void* vol, size_t esz, size_t* lds) {
  if ((g.p & 1) || (g.F & 7) || (g.F > 4096) || ((uintptr_t)vol & 15) || (((size_t)g.Wx * esz) & 15)) return false;
  const int nrows = g.C * g.pt * g.p;
  // LDS budget per workgroup: small enough for several workgroups per CU, whose load / statistics / store phases then
  // overlap (one 96 KiB workgroup per CU ran them strictly one after the other).  CTCLIP_PATCH_LDS_KB overrides.
  static const size_t lds_cap = [] { const char* e = CTCLIP_KNOB("CTCLIP_PATCH_LDS_KB"); return (size_t)(e ? atoi(e) : 32) * 1024; }();   // B=16: 128 KiB 1820/2062 us (fwd/bwd), 48 KiB 1113/959, 32 KiB 1033/967
  for (int tpb = g.Wt; tpb >= 1; --tpb) {
    if (g.Wt % tpb) continue;
    const size_t rlb = (size_t)tpb * g.p * esz;
    if ((rlb & 15) || nrows * rlb > lds_cap) continue;
    g.tpb = tpb; g.wgroups = g.Wt / tpb;
    *lds = ((nrows * rlb + (size_t)tpb * 8 + 15) & ~(size_t)15);
    g.m_p = magic_of((unsigned)g.p); g.m_ptp = magic_of((unsigned)(g.pt * g.p));
    g.m_vpr = magic_of((unsigned)(rlb / 16)); g.m_cpt = magic_of((unsigned)(g.ldA >> 3));
    return true;
  }
  return false;
}

int make_geom(PatchGeom& g, int B, int C, int Dz, int Hy, int Wx, int pt, int p, long ldA, float eps, int in_bf16,
              size_t* lds) {
  if (pt <= 0 || p <= 0 || Dz % pt || Hy % p || Wx % p) return (int)hipErrorInvalidValue;
  g.B = B; g.C = C; g.Dz = Dz; g.Hy = Hy; g.Wx = Wx; g.pt = pt; g.p = p;
  g.Tt = Dz / pt; g.Ht = Hy / p; g.Wt = Wx / p; g.F = C * pt * p * p; g.ldA = ldA; g.eps = eps;
  const size_t esz = in_bf16 ? 2 : 4;
  if (ldA < g.F || (ldA & 7)) return (int)hipErrorInvalidValue;
  int tpb = (int)((96 * 1024) / ((size_t)g.F * esz));
  if (tpb < 1) return (int)hipErrorInvalidValue;
  if (tpb > g.Wt) tpb = g.Wt;
  g.tpb = tpb; g.wgroups = (g.Wt + tpb - 1) / tpb;
  *lds = (size_t)tpb * g.F * esz + (size_t)tpb * 8 + 16;
  *lds = (*lds + 15) & ~(size_t)15;
  return 0;
}

}  // namespace

extern "C" {

int ctclip_patch_ln_fwd(const void* volume, int volume_is_bf16, const float* gamma, const float* beta, void* A_bf16,
                        float* mean, float* rstd, int B, int C, int Dz, int Hy, int Wx, int pt, int p, long ldA,
                        float eps, void* stream) {
  PatchGeom g{};
  size_t lds = 0;
  if (int e = make_geom(g, B, C, Dz, Hy, Wx, pt, p, ldA, eps, volume_is_bf16, &lds)) return e;
  {
    PatchGeom gf = g;
    size_t ldsf = 0;
    if (fast_geom(gf, volume, volume_is_bf16 ? 2 : 4, &ldsf) && (((uintptr_t)A_bf16) & 15) == 0) {
      const unsigned nb = (unsigned)((long)B * gf.Tt * gf.Ht * gf.wgroups);
      if (volume_is_bf16) {
        if (ldsf > 65536) hipFuncSetAttribute((const void*)patch_ln_fwd_fast<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsf);
        hipLaunchKernelGGL(patch_ln_fwd_fast<bf16_t>, dim3(nb), dim3(PF_THREADS), ldsf, (hipStream_t)stream,
                           (const bf16_t*)volume, gamma, beta, (bf16_t*)A_bf16, mean, rstd, gf);
      } else {
        if (ldsf > 65536) hipFuncSetAttribute((const void*)patch_ln_fwd_fast<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsf);
        hipLaunchKernelGGL(patch_ln_fwd_fast<float>, dim3(nb), dim3(PF_THREADS), ldsf, (hipStream_t)stream,
                           (const float*)volume, gamma, beta, (bf16_t*)A_bf16, mean, rstd, gf);
      }
      CTCLIP_CHECK_LAUNCH();
    }
  }
  const unsigned nblk = (unsigned)((long)B * g.Tt * g.Ht * g.wgroups);
  if (volume_is_bf16) {
    if (lds > 65536) hipFuncSetAttribute((const void*)patch_ln_fwd_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(patch_ln_fwd_kernel<bf16_t>, dim3(nblk), dim3(256), lds, (hipStream_t)stream, (const bf16_t*)volume,
                       gamma, beta, (bf16_t*)A_bf16, mean, rstd, g);
  } else {
    if (lds > 65536) hipFuncSetAttribute((const void*)patch_ln_fwd_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(patch_ln_fwd_kernel<float>, dim3(nblk), dim3(256), lds, (hipStream_t)stream, (const float*)volume,
                       gamma, beta, (bf16_t*)A_bf16, mean, rstd, g);
  }
  CTCLIP_CHECK_LAUNCH();
}

int ctclip_patch_affine_fold(const float* W, const float* bias, const float* gamma, const float* beta, void* Wg_bf16,
                             float* bias_folded, int N, int F, long ldw, void* stream) {
  if (N <= 0 || F <= 0) return 0;
  if (ldw < F) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(patch_affine_fold_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, W, bias, gamma, beta,
                     (bf16_t*)Wg_bf16, bias_folded, F, ldw);
  CTCLIP_CHECK_LAUNCH();
}

int ctclip_patch_affine_bwd(const float* G, const float* db, const float* W, const float* gamma, const float* beta, float* dW,
                            float* dgamma, float* dbeta, int N, int F, long ldg, int ncorr, void* stream) {
  if (N <= 0 || F <= 0) return 0;
  if (ncorr < 0 || ldg < F + ncorr) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(patch_affine_bwd_kernel, dim3((F + 63) / 64), dim3(256), 0, (hipStream_t)stream, G, db, W, gamma, beta,
                     dW, dgamma, dbeta, N, F, ldg, ncorr);
  CTCLIP_CHECK_LAUNCH();
}

int ctclip_patch_ln_bwd_dx(const void* volume, int volume_is_bf16, const void* dA_bf16, long ldd, const float* gamma,
                           const float* mean, const float* rstd, float* dvolume, int B, int C, int Dz, int Hy, int Wx, int pt,
                           int p, void* stream) {
  PatchGeom g{};
  size_t lds = 0;
  const long F = (long)C * pt * p * p;
  if (int e = make_geom(g, B, C, Dz, Hy, Wx, pt, p, (F + 7) / 8 * 8, 0.f, volume_is_bf16, &lds)) return e;
  const unsigned nblk = (unsigned)((long)B * g.Tt * g.Ht * g.wgroups);
  if (volume_is_bf16) {
    if (lds > 65536) hipFuncSetAttribute((const void*)patch_ln_bwd_dx_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(patch_ln_bwd_dx_kernel<bf16_t>, dim3(nblk), dim3(256), lds, (hipStream_t)stream, (const bf16_t*)volume,
                       (const bf16_t*)dA_bf16, ldd, gamma, mean, rstd, dvolume, g);
  } else {
    if (lds > 65536) hipFuncSetAttribute((const void*)patch_ln_bwd_dx_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(patch_ln_bwd_dx_kernel<float>, dim3(nblk), dim3(256), lds, (hipStream_t)stream, (const float*)volume,
                       (const bf16_t*)dA_bf16, ldd, gamma, mean, rstd, dvolume, g);
  }
  CTCLIP_CHECK_LAUNCH();
}

}  // extern "C"
