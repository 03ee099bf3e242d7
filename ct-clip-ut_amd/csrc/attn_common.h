// Shared pieces of the fused attention kernels (attention.hip, attention_ws.hip, attention_hm.hip): argument block, the swizzled
// [row][D] LDS image and the MFMA fragment readers.  gfx950 only.
#pragma once
#include "common.h"

struct CtclipAttnArgs {
  const bf16_t* q; const bf16_t* k; const bf16_t* v;
  bf16_t* o;                 // fwd out
  float* lse;                // [nseq, H, n]
  const float* bias;         // [H, n, n] or null
  const float* mask;         // [nseq, n] additive or null
  const bf16_t* dO;          // bwd
  const bf16_t* oin;         // bwd: forward output (for delta)
  float* delta;              // [nseq, H, n]
  bf16_t* dq; bf16_t* dk; bf16_t* dv;
  float* dbias_dense;        // [H, n, n] or null
  const uint16_t* relidx;    // [n, n] or null
  float* dbias_table;        // [H, R] or null
  int table_size;            // R
  int grid_h, grid_w;        // >0: relidx[i][j] = (yi-yj+h-1)*(2w-1) + (xi-xj+w-1) computed on the fly (i = y*w + x)
  long ldq, ldk, ldv, ldo, lddo, lddq, lddk, lddv;
  int nseq, n, n_pad, heads;
  float scale;
  const uint8_t* drop;       // [nseq, H, n, n] keep flags (1 = keep) of attention-probability dropout, or null
  float drop_scale;          // 1 / (1 - p): kept probabilities are multiplied by it
  int hpb;                   // heads per workgroup (per-sequence kernels): the block is hpb independent groups of waves,
  int lds_per_head;          // each with its own LDS region; heads % hpb == 0
};
typedef CtclipAttnArgs AttnArgs;

// wave-owns-the-sequence kernels for long rows with a shared bias (attention_ws.hip); return -1 when the shape is not
// eligible and the caller must use the per-sequence kernels of attention.hip.
int ctclip_attn_ws_fwd(const CtclipAttnArgs& a, int dhead, hipStream_t st);
int ctclip_attn_ws_bwd(const CtclipAttnArgs& a, int dhead, hipStream_t st);

namespace {

template <int D>
__device__ __forceinline__ uint32_t img_off(int row, int chunk) {
  if (D == 32) return (uint32_t)(row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4));
  return (uint32_t)(row * 128 + ((chunk ^ (((row >> 2) & 3) | (((row >> 1) & 1) << 2))) << 4));
}

// stage rows [0,n) of a [*, ld] matrix (columns head*D..) into a swizzled LDS image; rows n..n_pad-1 = 0
template <int D>
__device__ __forceinline__ void load_image(char* img, const bf16_t* __restrict__ base, long ld, int n, int n_pad,
                                           int tid, int nthreads) {
  constexpr int CH = D / 8;
  for (int id = tid; id < n_pad * CH; id += nthreads) {
    const int r = id / CH, c = id % CH;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (r < n) v = *(const uint4*)(base + (long)r * ld + c * 8);
    *(uint4*)(img + img_off<D>(r, c)) = v;
  }
}

// A/B fragment for MFMA 32x32x16: element j = M[row0 + (lane&31)][16*s + 8*(lane>>5) + j]
template <int D>
__device__ __forceinline__ bf16x8 row_frag(const char* img, int row0, int s, int lane) {
  return *(const bf16x8*)(img + img_off<D>(row0 + (lane & 31), 2 * s + (lane >> 5)));
}

// transposed fragment: element j = M[row0 + 16*s + 8*(j>>2) + 4*(lane>>5) + (j&3)][32*dt + (lane&31)]
// (the k-order an f32 32x32 accumulator has when it is re-used as the other operand)
template <int D>
__device__ __forceinline__ bf16x8 tr_frag(const char* img, int row0, int s, int dt, int lane) {
  const int g = lane >> 4, i16 = lane & 15, qq = i16 >> 2, p = i16 & 3, h = g >> 1;
  const int col = 32 * dt + 16 * (g & 1) + 4 * p;
  const int rlo = row0 + 16 * s + 4 * h + qq;
  const uint32_t sub = (uint32_t)((p & 1) * 8);
  short4v lo = lds_read_tr16(img + img_off<D>(rlo, col >> 3) + sub);
  short4v hi = lds_read_tr16(img + img_off<D>(rlo + 8, col >> 3) + sub);
  return join_tr(lo, hi);
}

// fragment straight from global memory: element j = M[row][16*s + 8*(lane>>5) + j]; lanes that are not `valid` get zeros.
// rowptr must point at a row that EXISTS for every lane (callers clamp the row index): the load is unconditional and the
// zeros are selected afterwards.  With the load inside `if (valid)` every fragment became its own predicated block that ended
// in `s_waitcnt vmcnt(0)` -- ten loads of a wave, ten memory round trips one after the other.
__device__ __forceinline__ bf16x8 gfrag(const bf16_t* __restrict__ rowptr, int s, int lane, bool valid) {
  const short8v z = {0, 0, 0, 0, 0, 0, 0, 0};
  const short8v v = *(const short8v*)(rowptr + 16 * s + 8 * (lane >> 5));
  return as_bf16x8(valid ? v : z);
}

// registers 8s..8s+7 of an accumulator -> bf16 fragment of k-step s
__device__ __forceinline__ bf16x8 acc_frag(const f32x16& a, int s) {
  bf16x8 r;
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = (__bf16)a[8 * s + j];
  return r;
}

__device__ __forceinline__ void zero_acc(f32x16& a) {
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = 0.f;
}

// store one wave's [D x 32] transposed accumulator tile as rows of a [*, ld] bf16 matrix:
// lane (r, h) owns row `row`, columns 32*dt + 8*g4 + 4*h + (0..3)
template <int D>
__device__ __forceinline__ void store_rows(bf16_t* __restrict__ rowptr, const f32x16 (&acc)[D / 32], float mul, int lane) {
  const int h = lane >> 5;
#pragma unroll
  for (int dt = 0; dt < D / 32; ++dt)
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      uint2 p;
      p.x = pack_bf16x2(acc[dt][4 * g4 + 0] * mul, acc[dt][4 * g4 + 1] * mul);
      p.y = pack_bf16x2(acc[dt][4 * g4 + 2] * mul, acc[dt][4 * g4 + 3] * mul);
      *(uint2*)(rowptr + 32 * dt + 8 * g4 + 4 * h) = p;
    }
}

}  // namespace
