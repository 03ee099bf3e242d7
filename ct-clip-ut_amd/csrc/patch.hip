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

namespace {

struct PatchGeom {
  int B, C, Dz, Hy, Wx, pt, p;
  int Tt, Ht, Wt, F, tpb, wgroups;
  long ldA;
  float eps;
};

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
    if (f < g.F) v0 = (ldf<TIN>(buf[tok * g.F + f]) - mu) * rs * gamma[f] + beta[f];
    if (f + 1 < g.F) v1 = (ldf<TIN>(buf[tok * g.F + f + 1]) - mu) * rs * gamma[f + 1] + beta[f + 1];
    *(uint32_t*)(A + (row0 + tok) * g.ldA + f) = pack_bf16x2(v0, v1);
  }
}

// dgamma[f] += sum_tokens dA[token][f] * xhat ; dbeta[f] += sum_tokens dA[token][f]
template <typename TIN>
__global__ __launch_bounds__(256) void patch_ln_bwd_kernel(const TIN* __restrict__ vol, const bf16_t* __restrict__ dA,
                                                           long ldd, const float* __restrict__ mean,
                                                           const float* __restrict__ rstd, float* __restrict__ dgamma,
                                                           float* __restrict__ dbeta, PatchGeom g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  TIN* buf = (TIN*)smem;
  const int tid = threadIdx.x;
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
  for (int f = tid; f < g.F; f += 256) {
    float ag = 0.f, ab = 0.f;
    for (int tok = 0; tok < ntok; ++tok) {
      const float d = bf16_to_f32(dA[(row0 + tok) * ldd + f]);
      const float xh = (ldf<TIN>(buf[tok * g.F + f]) - mean[row0 + tok]) * rstd[row0 + tok];
      ag += d * xh;
      ab += d;
    }
    atomicAdd(dgamma + f, ag);
    atomicAdd(dbeta + f, ab);
  }
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

int ctclip_patch_ln_bwd(const void* volume, int volume_is_bf16, const void* dA_bf16, long ldd, const float* mean,
                        const float* rstd, float* dgamma, float* dbeta, int B, int C, int Dz, int Hy, int Wx, int pt,
                        int p, void* stream) {
  PatchGeom g{};
  size_t lds = 0;
  const long F = (long)C * pt * p * p;
  if (int e = make_geom(g, B, C, Dz, Hy, Wx, pt, p, (F + 7) / 8 * 8, 0.f, volume_is_bf16, &lds)) return e;
  const unsigned nblk = (unsigned)((long)B * g.Tt * g.Ht * g.wgroups);
  if (volume_is_bf16) {
    if (lds > 65536) hipFuncSetAttribute((const void*)patch_ln_bwd_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(patch_ln_bwd_kernel<bf16_t>, dim3(nblk), dim3(256), lds, (hipStream_t)stream, (const bf16_t*)volume,
                       (const bf16_t*)dA_bf16, ldd, mean, rstd, dgamma, dbeta, g);
  } else {
    if (lds > 65536) hipFuncSetAttribute((const void*)patch_ln_bwd_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(patch_ln_bwd_kernel<float>, dim3(nblk), dim3(256), lds, (hipStream_t)stream, (const float*)volume,
                       (const bf16_t*)dA_bf16, ldd, mean, rstd, dgamma, dbeta, g);
  }
  CTCLIP_CHECK_LAUNCH();
}

}  // extern "C"
