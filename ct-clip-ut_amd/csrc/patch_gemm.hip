// CT-ViT tubelet patch embedding as ONE pass over the volume (reference src/utils/ctvit.py:44-52: Rearrange 'b c (t pt) (h p1)
// (w p2) -> b t h w (c pt p1 p2)', LayerNorm(F), Linear(F, dim); the trailing LayerNorm(dim) stays ctclip_layernorm_fwd).
//
// The unfused chain (patch.hip + gemm3) writes the normalised [tokens, F] bf16 operand -- 10.6 GB at 96 pairs -- and reads it
// twice (projection, weight gradient).  Here the MFMA operand is built from the raw voxels inside the GEMM's load block:
//
//   * a workgroup owns 128 consecutive tokens x ALL 512 output columns (8 waves = 2 x 4, 64 x 128 per wave): every voxel is
//     read from HBM once, centred once; the folded weight W' = W o gamma (4 MB, L2 / Infinity-Cache resident) is streamed per
//     tile through the gemm3 ring (global_load_lds, four 32 KiB slots, counted vmcnt);
//   * K runs in FEATURE order, 32 features per step = 8 pieces of 4 voxels per token (a piece never straddles a p2-run because
//     p % 4 == 0): thread (token m = tid / 4, piece j = tid % 4 and j + 4) loads its two 8-byte pieces two K-steps ahead into
//     registers, subtracts the token's centring constant c, rounds to bf16 and writes the k32 tile of gemm_tile.h
//     (ds_write_b64) one K-step ahead of its use -- VALU work that runs in the load block, next to the partner wave's MFMAs;
//   * c = bf16(mean of the token's first p2-run): (x - c) is EXACTLY 0 for a constant tubelet (air = -1 padding), which is what
//     keeps the reference's xhat = 0 there; the statistics ride along, mu' = mean(x - c) and var = mean((x - c)^2) - mu'^2 in f32
//     from the exact differences, and the epilogue finishes the LayerNorm:
//         z = rstd (acc - mu' S[n]) + b'[n],      S[n] = sum_f W'[n][f],  b' = b + W beta        (ctclip_patch_affine_fold)
//     |mu'| is a fraction of the tubelet's own spread, so the correction term carries no cancellation (for a constant tubelet it
//     is exactly 0).  mean = c + mu' and rstd are stored for the backward.
//
// The weight gradient G = dz^T xhat recomputes the centred operand from the volume the same way (patch_wgrad below), so the
// [tokens, F] operand is never allocated in training.
// Preconditions (ctclip_patch_embed_fused returns hipErrorNotSupported otherwise and the caller keeps the unfused chain): bf16
// volume, p % 4 == 0, F % 32 == 0, 128 <= F <= 4096, N == 512, 8-byte aligned runs (Wx % 4 == 0, 16-byte aligned volume).
#include "gemm_tile.h"

namespace pg {
using namespace g3;

constexpr int PBM = 128, PBN = 512, NSB = 4;
constexpr int BSTAGE = PBN * BK * 2;          // 32 KiB: the W' tile of a K-step, [512 rows][32 k] in the k32 layout
constexpr int ASTAGE = PBM * BK * 2;          // 8 KiB: the centred tubelet tile, [128 tokens][32 k]
constexpr int TPPW = BSTAGE / 1024 / 8;       // 1 KiB DMA pieces per wave and K-step: 4
constexpr int MAXF = 4096;
constexpr int OFF_A = NSB * BSTAGE;
constexpr int OFF_TAB = OFF_A + 2 * ASTAGE;
constexpr int OFF_STAT = OFF_TAB + (MAXF / 4) * 4;
constexpr int LDS_BYTES = OFF_STAT + PBM * 2 * 4;

struct Geom {
  int C, Dz, Hy, Wx, pt, p;       // volume [B][C][Dz][Hy][Wx], tubelet pt x p x p
  int Tt, Ht, Wt, F;
};

struct FwdArgs {
  const bf16_t* vol; const bf16_t* W; long ldw;
  const float* S; const float* bias;
  float* Z; long ldz;
  float* cbase; float* mean; float* rstd;
  int M; float eps;
  Geom g;
};

// element offset of token `tok`'s first voxel (c = 0, pti = 0, p1i = 0, p2i = 0)
__device__ __forceinline__ long token_origin(const Geom& g, int tok) {
  const int w = tok % g.Wt; int r = tok / g.Wt;
  const int h = r % g.Ht; r /= g.Ht;
  const int t = r % g.Tt; const int b = r / g.Tt;
  return (((long)b * g.C * g.Dz + (long)t * g.pt) * g.Hy + (long)h * g.p) * g.Wx + (long)w * g.p;
}
// element offset, relative to the token's first voxel, of feature f (a multiple of 4): f = ((c pt + pti) p + p1i) p + p2i
__device__ __forceinline__ int piece_offset(const Geom& g, int f) {
  const int run = f / g.p, p2 = f - run * g.p;
  const int c = run / (g.pt * g.p), rem = run - c * g.pt * g.p;
  const int pti = rem / g.p, p1 = rem - pti * g.p;
  return ((c * g.Dz + pti) * g.Hy + p1) * g.Wx + p2;
}

__device__ __forceinline__ void unpack4(uint2 r, float c, float (&d)[4]) {
  d[0] = __uint_as_float(r.x << 16) - c; d[1] = __uint_as_float(r.x & 0xffff0000u) - c;
  d[2] = __uint_as_float(r.y << 16) - c; d[3] = __uint_as_float(r.y & 0xffff0000u) - c;
}

__global__ __launch_bounds__(512, 2) void patch_gemm_fwd_kernel(FwdArgs a) {
  constexpr int IM = 4, JN = 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* ptab = (int*)(smem + OFF_TAB);
  float* stat = (float*)(smem + OFF_STAT);                 // [128][2]: mu', rstd
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int grp = wave >> 2;                                // role group: SIMD partners are waves w and w + 4
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  const Geom& g = a.g;
  const int nk = g.F / BK;
  const int row0 = (int)blockIdx.x * PBM;

  // ---- the piece table and this thread's token
  for (int i = tid; i < g.F / 4; i += 512) ptab[i] = piece_offset(g, 4 * i);
  const int m = tid >> 2, j0 = tid & 3;
  int tok = row0 + m;
  if (tok >= a.M) tok = a.M - 1;                           // masked in the epilogue
  const bf16_t* tb = a.vol + token_origin(g, tok);
  // c = bf16(mean of the first p2-run): the four lanes of a token share its p / 4 pieces
  float csum = 0.f;
  for (int q = j0; q < g.p / 4; q += 4) {
    float d[4];
    unpack4(*(const uint2*)(tb + 4 * q), 0.f, d);
    csum += (d[0] + d[1]) + (d[2] + d[3]);
  }
  csum += __shfl_xor(csum, 1, 64);
  csum += __shfl_xor(csum, 2, 64);
  const float cb = bf16_to_f32(f32_to_bf16(csum / (float)g.p));

  // ---- W' through the ring: piece q = wave * TPPW + j of a stage = rows 16 q .. 16 q + 15 of the 512
  const bf16_t* srcB[TPPW];
  uint32_t dstB[TPPW];
#pragma unroll
  for (int j = 0; j < TPPW; ++j) {
    const int q = wave * TPPW + j;
    srcB[j] = a.W + piece_src(q, lane, 0, PBN, a.ldw);
    dstB[j] = (uint32_t)(q * 1024);
  }
  auto issue_B = [&](int t) {
    const uint32_t sb = lds0 + (uint32_t)((t % NSB) * BSTAGE);
#pragma unroll
    for (int j = 0; j < TPPW; ++j) G3_GLDS(srcB[j] + (long)t * BK, sb + dstB[j]);
  };
  // this thread's two pieces of a K-step: where they go in the k32 tile (row m, 8-byte half j & 1 of chunk j >> 1)
  const uint32_t wo0 = OFF_A + tile_off(m, j0 >> 1) + 8 * (j0 & 1);
  const uint32_t wo1 = OFF_A + tile_off(m, (j0 + 4) >> 1) + 8 * (j0 & 1);
  float s1 = 0.f, s2 = 0.f;
  uint2 ra[2][2];                                            // [register slot = K-step parity][piece]
  auto load_A = [&](int t, uint2 (&r)[2]) {                  // K-step t's pieces -> registers
    __syncthreads_or_nothing:;
    const int o0 = ptab[t * 8 + j0], o1 = ptab[t * 8 + j0 + 4];
    r[0] = *(const uint2*)(tb + o0);
    r[1] = *(const uint2*)(tb + o1);
  };
  auto convert_A = [&](int t, const uint2 (&r)[2]) {         // registers -> centred bf16 in A slot t & 1
    float d0[4], d1[4];
    unpack4(r[0], cb, d0);
    unpack4(r[1], cb, d1);
    s1 += ((d0[0] + d0[1]) + (d0[2] + d0[3])) + ((d1[0] + d1[1]) + (d1[2] + d1[3]));
    s2 += ((d0[0] * d0[0] + d0[1] * d0[1]) + (d0[2] * d0[2] + d0[3] * d0[3])) +
          ((d1[0] * d1[0] + d1[1] * d1[1]) + (d1[2] * d1[2] + d1[3] * d1[3]));
    const uint32_t base = (uint32_t)((t & 1) * ASTAGE);
    *(uint2*)(smem + base + wo0) = make_uint2(pack_bf16x2(d0[0], d0[1]), pack_bf16x2(d0[2], d0[3]));
    *(uint2*)(smem + base + wo1) = make_uint2(pack_bf16x2(d1[0], d1[1]), pack_bf16x2(d1[2], d1[3]));
  };
  (void)load_A; (void)convert_A;
}

}  // namespace pg
