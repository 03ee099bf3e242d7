// bf16 MFMA GEMM family for gfx950 (MI355X).
//
//   C[M,N] = alpha * opA(A)[M,K] * opB(B)[K,N]  (+ bias[N]) (+ residual[M,N]) -> act -> bf16 | f32 | f32 atomicAdd
//
// Operand layouts (no transposes are ever materialised in HBM):
//   a_kmajor = 1 : A stored [M][K] (K contiguous)      0 : A stored [K][M] (M contiguous)
//   b_kmajor = 1 : B stored [N][K] (nn.Linear weight)  0 : B stored [K][N]
// which covers the three products of a linear layer:
//   forward  y  = x  W^T   (1,1)      dgrad  dx = dy W   (1,0)      wgrad  dW = dy^T x  (0,0)
//
// Structure: 128x128x64 block tile, 256 threads = 4 waves (2x2), each wave a 64x64 sub-tile as
// 2x2 v_mfma_f32_32x32x16_bf16 accumulators.  Global -> VGPR -> LDS staging, double-buffered
// LDS (64 KiB), next tile's global loads issued before the MFMA block and written to LDS after it
// (one barrier per K-step).  K-major tiles are stored [row][64] with a 16-byte-chunk XOR swizzle
// (chunk ^ ((row>>1)&7)) so ds_read_b128 fragment reads are bank-conflict free; M/N-major tiles are
// stored [k][128] with the dual-use swizzle (chunk ^ ((k&3)<<2 | (k>>2)&3)) and fragments are
// fetched with ds_read_b64_tr_b16 (hardware transpose), also conflict free.
// Block ids are remapped so each XCD's L2 sees a contiguous run of tiles (N fastest: blocks that
// share an A row-panel run on one XCD).  split_k > 1 accumulates with f32 atomics.
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int BM = 128, BN = 128, BKT = 64, NTHREADS = 256;
constexpr int TILE_BYTES = BM * BKT * 2;  // 16 KiB per operand tile

struct GemmArgs {
  const bf16_t* A;
  const bf16_t* B;
  void* C;
  const float* bias;
  const float* resid;
  long lda, ldb, ldc, ldr;
  int M, N, K;
  int tiles_m, tiles_n, split_k, ktiles_per_split;
  int c_fp32, atomic_out, act;
  float alpha;
  float* part_val;  // argmax epilogue: [N][n_parts]
  int* part_idx;
  int n_parts;
  float* part;      // split-K without atomics: split ks stores its tile into part[ks][M][N] (plain stores); null = atomics
};

template <bool KM>
__device__ __forceinline__ uint32_t tile_off(int rk, int chunk) {
  if (KM) return (uint32_t)(rk * 128 + ((chunk ^ ((rk >> 1) & 7)) << 4));
  return (uint32_t)(rk * 256 + ((chunk ^ (((rk & 3) << 2) | ((rk >> 2) & 3))) << 4));
}

template <bool KM>
__device__ __forceinline__ void load_tile_regs(const bf16_t* __restrict__ base, long ld, int row0, int k0,
                                               int R, int K, uint4 (&reg)[4], int tid) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int q = i * NTHREADS + tid;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (KM) {
      const int r = q >> 3, c = q & 7;
      if (row0 + r < R && k0 + c * 8 < K) v = *(const uint4*)(base + (long)(row0 + r) * ld + k0 + c * 8);
    } else {
      const int k = q >> 4, c = q & 15;
      if (k0 + k < K && row0 + c * 8 < R) v = *(const uint4*)(base + (long)(k0 + k) * ld + row0 + c * 8);
    }
    reg[i] = v;
  }
}

template <bool KM>
__device__ __forceinline__ void store_tile_lds(char* tile, const uint4 (&reg)[4], int tid) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int q = i * NTHREADS + tid;
    const uint32_t off = KM ? tile_off<true>(q >> 3, q & 7) : tile_off<false>(q >> 4, q & 15);
    *(uint4*)(tile + off) = reg[i];
  }
}

// MFMA operand fragment for rows rbase..rbase+31 of the tile, k-step s (16 k values).
template <bool KM>
__device__ __forceinline__ bf16x8 read_frag(const char* tile, int rbase, int s, int lane) {
  if (KM) {
    const int r = rbase + (lane & 31);
    return *(const bf16x8*)(tile + tile_off<true>(r, 2 * s + (lane >> 5)));
  } else {
    const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, p = i16 & 3, h = g >> 1;
    const int m0 = rbase + 16 * (g & 1) + 4 * p;
    const int klo = 16 * s + 8 * h + q;
    const uint32_t sub = (uint32_t)((p & 1) * 8);
    short4v lo = lds_read_tr16(tile + tile_off<false>(klo, m0 >> 3) + sub);
    short4v hi = lds_read_tr16(tile + tile_off<false>(klo + 4, m0 >> 3) + sub);
    return join_tr(lo, hi);
  }
}

__device__ __forceinline__ float gelu_erf(float x) { return gelu_erf_fast(x); }

template <bool AKM, bool BKM, int EPI>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_bf16_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  const int nwg = gridDim.x;
  int bid = xcd_remap(blockIdx.x, nwg);
  const int tn = bid % g.tiles_n;
  bid /= g.tiles_n;
  const int tm = bid % g.tiles_m;
  const int ks = bid / g.tiles_m;

  const int row0 = tm * BM, col0 = tn * BN;
  const int nk_total = (g.K + BKT - 1) / BKT;
  const int kt_begin = ks * g.ktiles_per_split;
  const int kt_end = min(nk_total, kt_begin + g.ktiles_per_split);

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (kt_begin < kt_end) {
    uint4 ra[4], rb[4];
    load_tile_regs<AKM>(g.A, g.lda, row0, kt_begin * BKT, g.M, g.K, ra, tid);
    load_tile_regs<BKM>(g.B, g.ldb, col0, kt_begin * BKT, g.N, g.K, rb, tid);
    store_tile_lds<AKM>(smem, ra, tid);
    store_tile_lds<BKM>(smem + TILE_BYTES, rb, tid);
    __syncthreads();

    for (int kt = kt_begin; kt < kt_end; ++kt) {
      const int cur = (kt - kt_begin) & 1;
      const char* ta = smem + cur * 2 * TILE_BYTES;
      const char* tb = ta + TILE_BYTES;
      const bool more = (kt + 1) < kt_end;
      if (more) {
        load_tile_regs<AKM>(g.A, g.lda, row0, (kt + 1) * BKT, g.M, g.K, ra, tid);
        load_tile_regs<BKM>(g.B, g.ldb, col0, (kt + 1) * BKT, g.N, g.K, rb, tid);
      }
#pragma unroll
      for (int s = 0; s < BKT / 16; ++s) {
        bf16x8 fa[2], fb[2];
        fa[0] = read_frag<AKM>(ta, wm * 64, s, lane);
        fa[1] = read_frag<AKM>(ta, wm * 64 + 32, s, lane);
        fb[0] = read_frag<BKM>(tb, wn * 64, s, lane);
        fb[1] = read_frag<BKM>(tb, wn * 64 + 32, s, lane);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = mfma32(fa[i], fb[j], acc[i][j]);
      }
      if (more) {
        char* na = smem + (cur ^ 1) * 2 * TILE_BYTES;
        store_tile_lds<AKM>(na, ra, tid);
        store_tile_lds<BKM>(na + TILE_BYTES, rb, tid);
      }
      __syncthreads();
    }
  }

  const int half = lane >> 5, lc = lane & 31;

  if (EPI == 1) {
    // top-2 over the M (code) rows of this wave's 64x64 sub-tile, per N (token) column.
    // part[((col) * n_parts + tm*2 + wm) * 2 + {0,1}] = (value, row index), best first; ties -> lowest row.
    // Two candidates per 64-code slab let the caller re-rank near-ties in f32 (bf16 scores can flip them).
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = col0 + wn * 64 + j * 32 + lc;
      float b1 = -INFINITY, b2 = -INFINITY;
      int i1 = 0x7fffffff, i2 = 0x7fffffff;
      auto push = [&](float v, int row) {
        if (v > b1 || (v == b1 && row < i1)) { b2 = b1; i2 = i1; b1 = v; i1 = row; }
        else if (v > b2 || (v == b2 && row < i2)) { b2 = v; i2 = row; }
      };
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = row0 + wm * 64 + i * 32 + acc_row(r, half);
          if (row < g.M) push(acc[i][j][r], row);
        }
      const float o1 = __shfl_xor(b1, 32, 64), o2 = __shfl_xor(b2, 32, 64);
      const int oi1 = __shfl_xor(i1, 32, 64), oi2 = __shfl_xor(i2, 32, 64);
      if (oi1 != 0x7fffffff) push(o1, oi1);
      if (oi2 != 0x7fffffff) push(o2, oi2);
      if (half == 0 && col < g.N) {
        const long p = ((long)col * g.n_parts + tm * 2 + wm) * 2;
        g.part_val[p] = b1; g.part_idx[p] = i1;
        g.part_val[p + 1] = b2; g.part_idx[p + 1] = i2;
      }
    }
    return;
  }

  if (g.atomic_out) {
    // split-K / accumulate: f32 atomics straight from the accumulators.  For a fixed register the 64 lanes cover two
    // 128-byte row segments -- the access shape global_atomic_add_f32 runs at full rate with.
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int col = col0 + wn * 64 + j * 32 + lc;
        if (col >= g.N) continue;
        const float bv = (g.bias && ks == 0) ? g.bias[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = row0 + wm * 64 + i * 32 + acc_row(r, half);
          if (row >= g.M) continue;
          float v = acc[i][j][r] * g.alpha + bv;
          if (g.resid && ks == 0) v += g.resid[(long)row * g.ldr + col];
          if (g.part) g.part[((long)ks * g.M + row) * g.N + col] = v;
          else atomicAdd((float*)g.C + (long)row * g.ldc + col, v);
        }
      }
    return;
  }

  // Plain store: the accumulators are lane = column / register = row, which would give 2- or 4-byte stores scattered over
  // 32 rows per instruction (measured: ~45 % of the kernel at K = 512).  Instead the tile is transposed through the
  // (now idle) LDS ring as f32 [128][128] and written out with 16-byte, fully coalesced stores; bias, residual,
  // activation and the bf16 conversion are applied on the way out.
  float* ct = (float*)smem;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        ct[(wm * 64 + i * 32 + acc_row(r, half)) * BN + wn * 64 + j * 32 + lc] = acc[i][j][r];
  __syncthreads();
  const int act = g.act;
  if (g.c_fp32) {
    float* C = (float*)g.C;
    const bool vec = ((g.ldc & 3) == 0) && ((((uintptr_t)C) & 15) == 0) &&
                     (!g.resid || (((g.ldr & 3) == 0) && ((((uintptr_t)g.resid) & 15) == 0)));
#pragma unroll 4
    for (int it = 0; it < (BM * BN / 4) / NTHREADS; ++it) {
      const int id = it * NTHREADS + tid, r = id >> 5, c4 = (id & 31) * 4;
      const int row = row0 + r, col = col0 + c4;
      if (row >= g.M || col >= g.N) continue;
      const float4 t = *(const float4*)(ct + r * BN + c4);
      float v[4] = {t.x * g.alpha, t.y * g.alpha, t.z * g.alpha, t.w * g.alpha};
      if (vec && col + 3 < g.N) {
        if (g.bias) { const float4 b = *(const float4*)(g.bias + col); v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w; }
        if (g.resid) { const float4 q = *(const float4*)(g.resid + (long)row * g.ldr + col); v[0] += q.x; v[1] += q.y; v[2] += q.z; v[3] += q.w; }
        if (act == 1) { v[0] = gelu_erf(v[0]); v[1] = gelu_erf(v[1]); v[2] = gelu_erf(v[2]); v[3] = gelu_erf(v[3]); }
        *(float4*)(C + (long)row * g.ldc + col) = make_float4(v[0], v[1], v[2], v[3]);
      } else {
        for (int e = 0; e < 4 && col + e < g.N; ++e) {
          float x = v[e] + (g.bias ? g.bias[col + e] : 0.f);
          if (g.resid) x += g.resid[(long)row * g.ldr + col + e];
          if (act == 1) x = gelu_erf(x);
          C[(long)row * g.ldc + col + e] = x;
        }
      }
    }
  } else {
    bf16_t* C = (bf16_t*)g.C;
    const bool vec = ((g.ldc & 7) == 0) && ((((uintptr_t)C) & 15) == 0) &&
                     (!g.resid || (((g.ldr & 3) == 0) && ((((uintptr_t)g.resid) & 15) == 0)));
#pragma unroll 4
    for (int it = 0; it < (BM * BN / 8) / NTHREADS; ++it) {
      const int id = it * NTHREADS + tid, r = id >> 4, c8 = (id & 15) * 8;
      const int row = row0 + r, col = col0 + c8;
      if (row >= g.M || col >= g.N) continue;
      const float4 t0 = *(const float4*)(ct + r * BN + c8), t1 = *(const float4*)(ct + r * BN + c8 + 4);
      float v[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] *= g.alpha;
      if (vec && col + 7 < g.N) {
        if (g.bias) {
          const float4 b0 = *(const float4*)(g.bias + col), b1 = *(const float4*)(g.bias + col + 4);
          v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
        }
        if (g.resid) {
          const float4 q0 = *(const float4*)(g.resid + (long)row * g.ldr + col), q1 = *(const float4*)(g.resid + (long)row * g.ldr + col + 4);
          v[0] += q0.x; v[1] += q0.y; v[2] += q0.z; v[3] += q0.w; v[4] += q1.x; v[5] += q1.y; v[6] += q1.z; v[7] += q1.w;
        }
        if (act == 1) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = gelu_erf(v[e]);
        }
        uint4 o;
        o.x = pack_bf16x2(v[0], v[1]); o.y = pack_bf16x2(v[2], v[3]); o.z = pack_bf16x2(v[4], v[5]); o.w = pack_bf16x2(v[6], v[7]);
        *(uint4*)(C + (long)row * g.ldc + col) = o;
      } else {
        for (int e = 0; e < 8 && col + e < g.N; ++e) {
          float x = v[e] + (g.bias ? g.bias[col + e] : 0.f);
          if (g.resid) x += g.resid[(long)row * g.ldr + col + e];
          if (act == 1) x = gelu_erf(x);
          C[(long)row * g.ldc + col + e] = f32_to_bf16(x);
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// VQ nearest-code search: one workgroup owns 128 tokens (the N side) and sweeps ALL code tiles (the M side), keeping a
// running top-4 (value, code) per lane for each of its two token columns.  The [codes x tokens] score matrix never
// exists; the cross-lane work and the stores happen once per token tile.  Each token ends with 4 lanes-groups x 4 = 16
// candidates (disjoint quarters of the codebook), which ctclip_vq_select re-ranks exactly in f32.
// ------------------------------------------------------------------------------------------------------------------
constexpr int VQ_TOP = 4;
__global__ __launch_bounds__(NTHREADS, 2) void gemm_vq_topk_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int col0 = xcd_remap(blockIdx.x, gridDim.x) * BN;      // token tile
  const int nk = (g.K + BKT - 1) / BKT;
  const int steps = g.tiles_m * nk;                              // flattened (code tile, k tile) sequence
  const int half = lane >> 5, lc = lane & 31;

  float bv[2][VQ_TOP];
  int bi[2][VQ_TOP];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int t = 0; t < VQ_TOP; ++t) { bv[j][t] = -INFINITY; bi[j][t] = 0x7fffffff; }

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  uint4 ra[4], rb[4];
  load_tile_regs<true>(g.A, g.lda, 0, 0, g.M, g.K, ra, tid);
  load_tile_regs<true>(g.B, g.ldb, col0, 0, g.N, g.K, rb, tid);
  store_tile_lds<true>(smem, ra, tid);
  store_tile_lds<true>(smem + TILE_BYTES, rb, tid);
  __syncthreads();
  for (int st = 0; st < steps; ++st) {
    const int cur = st & 1;
    const char* ta = smem + cur * 2 * TILE_BYTES;
    const char* tb = ta + TILE_BYTES;
    const bool more = (st + 1) < steps;
    if (more) {
      const int nt = (st + 1) / nk, nkt = (st + 1) % nk;
      load_tile_regs<true>(g.A, g.lda, nt * BM, nkt * BKT, g.M, g.K, ra, tid);
      load_tile_regs<true>(g.B, g.ldb, col0, nkt * BKT, g.N, g.K, rb, tid);
    }
#pragma unroll
    for (int s = 0; s < BKT / 16; ++s) {
      bf16x8 fa[2], fb[2];
      fa[0] = read_frag<true>(ta, wm * 64, s, lane);
      fa[1] = read_frag<true>(ta, wm * 64 + 32, s, lane);
      fb[0] = read_frag<true>(tb, wn * 64, s, lane);
      fb[1] = read_frag<true>(tb, wn * 64 + 32, s, lane);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = mfma32(fa[i], fb[j], acc[i][j]);
    }
    if ((st % nk) == nk - 1) {                                   // a code tile is complete: fold it into the running top-4
      const int row_t = (st / nk) * BM + wm * 64;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float v = acc[i][j][r];
            const int row = row_t + i * 32 + acc_row(r, half);
            if (row < g.M && (v > bv[j][VQ_TOP - 1] || (v == bv[j][VQ_TOP - 1] && row < bi[j][VQ_TOP - 1]))) {
              float cv = v; int ci = row;                        // insertion into the sorted 4-list (rarely taken)
#pragma unroll
              for (int t = 0; t < VQ_TOP; ++t) {
                const bool better = cv > bv[j][t] || (cv == bv[j][t] && ci < bi[j][t]);
                const float tv = bv[j][t]; const int ti = bi[j][t];
                if (better) { bv[j][t] = cv; bi[j][t] = ci; cv = tv; ci = ti; }
              }
            }
            acc[i][j][r] = 0.f;
          }
    }
    if (more) {
      char* na = smem + (cur ^ 1) * 2 * TILE_BYTES;
      store_tile_lds<true>(na, ra, tid);
      store_tile_lds<true>(na + TILE_BYTES, rb, tid);
    }
    __syncthreads();
  }
  // candidates: [token][ (wm*2 + half) * VQ_TOP + t ]
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = col0 + wn * 64 + j * 32 + lc;
    if (col >= g.N) continue;
    const long p = (long)col * g.n_parts + (wm * 2 + half) * VQ_TOP;
#pragma unroll
    for (int t = 0; t < VQ_TOP; ++t) { g.part_val[p + t] = bv[j][t]; g.part_idx[p + t] = bi[j][t]; }
  }
}

template <int EPI>
int launch(const GemmArgs& g, int a_kmajor, int b_kmajor, hipStream_t st) {
  const int nblk = g.tiles_m * g.tiles_n * g.split_k;
  const size_t lds = 4 * TILE_BYTES;
  dim3 grid(nblk), block(NTHREADS);
  if (a_kmajor && b_kmajor) hipLaunchKernelGGL((gemm_bf16_kernel<true, true, EPI>), grid, block, lds, st, g);
  else if (a_kmajor && !b_kmajor) hipLaunchKernelGGL((gemm_bf16_kernel<true, false, EPI>), grid, block, lds, st, g);
  else if (!a_kmajor && b_kmajor) hipLaunchKernelGGL((gemm_bf16_kernel<false, true, EPI>), grid, block, lds, st, g);
  else hipLaunchKernelGGL((gemm_bf16_kernel<false, false, EPI>), grid, block, lds, st, g);
  return (int)hipGetLastError();
}

// 16-byte vector loads: base and row stride must be 8-element aligned.  A k-major operand needs K % 8 == 0 (the K tail is
// zero-filled per vector); an m/n-major operand may have a ragged extent as long as the row stride covers the last
// 8-element chunk (ld >= round_up(extent, 8)): the extra rows/columns it produces are masked in the epilogue.
bool bad_layout(const void* p, long ld, int contiguous_extent, bool kmajor) {
  if ((((uintptr_t)p) & 15) != 0 || (ld & 7) != 0) return true;
  if (kmajor) return (contiguous_extent & 7) != 0;
  return ld < ((contiguous_extent + 7) & ~7);
}

}  // namespace

int ctclip_gemm2_launch(const void* A, const void* B, void* C, const float* bias, const float* resid, int M, int N, int K,
                        long lda, long ldb, long ldc, long ldr, int a_kmajor, int b_kmajor, int c_fp32, int split_k,
                        int accumulate, float alpha, int act, float* part, hipStream_t st);
int ctclip_gemm3_launch(const void* A, const void* B, void* C, const float* bias, const float* resid, int M, int N, int K,
                        long lda, long ldb, long ldc, long ldr, int c_fp32, float alpha, int act, void* G, long ldg,
                        hipStream_t st);
int ctclip_gemm3_launch_hm(const void* A, const void* B, void* C, const float* bias, const float* resid, int M, int N, int K,
                           long lda, long ldb, long ldc, long ldr, int c_fp32, float alpha, int act, void* G, long ldg,
                           int hm_n, int hm_heads, hipStream_t st);
int ctclip_gemm5_launch(const void* A, const void* B, void* C, const float* bias, const float* resid, int M, int N, int K,
                        long lda, long ldb, long ldc, long ldr, int c_fp32, float alpha, int act, void* G, long ldg,
                        int hm_n, int hm_heads, hipStream_t st);
bool ctclip_gemm5_eligible(const void* C, const float* bias, const float* resid, int N, int K, long ldc, long ldr, int c_fp32,
                           int act);
int ctclip_gemm3_launch_ln(const void* A, const void* B, float* C, void* C16, const float* resid, int M, int N, int K, long lda,
                           long ldb, long ldc, long ldc16, long ldr, const void* xhat, long ldx, const float* c1, const float* c2,
                           hipStream_t st);
int ctclip_gemm3_launch_hn(const void* A, const void* B, void* C, int M, int N, int K, long lda, long ldb, long ldc, int hm_n,
                           int hm_heads, const float* scale, float mult, float* inv, int norm_cols, hipStream_t st);
int ctclip_gemm4_launch(const void* A, const void* B, void* C, int M, int N, int K, long lda, long ldb, long ldc, int split_k,
                        float alpha, float* part, hipStream_t st);
int ctclip_vq_topk3_launch(const void* A, const void* B, float* part_val, int* part_idx, int M, int N, int K, long lda,
                           long ldb, int cgroups, hipStream_t st);
extern "C" int ctclip_geglu_fwd(const void* h, void* g, long rows, int inner, int block, long ldh, long ldg, void* stream);
extern "C" int ctclip_geglu_bwd(const void* dg, const void* h, void* dh, long rows, int inner, int block, long lddg, long ldh,
                                void* stream);

namespace {
// gemm5.hip (one wave per SIMD, 128 x 128 per wave, paired full-line LDS-DMA, one request per 8 MFMAs) takes the k-major x k-major
// products on which it measured faster than gemm3.hip's 8 x (128 x 64) tile, interleaved on one box at 32 pairs: FF1 + GEGLU (1831 vs
// 1931 us) and plain products with many column tiles (N >= 2048: FF1 forward 885 vs 830 TFLOP/s) -- profiles/r04_gemm5.txt -- and,
// since its requests are evenly spaced (round 5, profiles/r05_gemm5_even_issue.txt), every plain product with a long matrix loop
// (K >= 1024: FF1 data gradient 1133 vs 1058 TFLOP/s, FF2 forward 817 vs 783); gemm3 stays on the short-K N <= 1408 shapes (kv forward
// 765 vs 788, out-projection 508 vs 534) and on FF2 dgrad + GEGLU backward (1548 vs 1462 us).
// K % 64 == 0 and K >= 192 (six K-steps: the ring fill).  Under the test hook CTCLIP_GEMM_V2_ALL the GEGLU-backward form with
// K >= 1024 goes to gemm5 as well, so that the suite reaches all of its epilogues with small shapes (products with K < 1024 keep
// exercising gemm3); ctclip_gemm5_bf16 calls the kernel directly.
// CTCLIP_GEMM5_MINK (-DCTCLIP_TUNING_KNOBS builds: A/B runs) makes K >= that value the only criterion.
bool gemm5_takes(int N, int K, int act) {
  static const bool v2_all = getenv("CTCLIP_GEMM_V2_ALL") != nullptr;
  static const int mink = [] { const char* e = CTCLIP_KNOB("CTCLIP_GEMM5_MINK"); return e ? atoi(e) : -1; }();
  if ((K % 64) != 0 || K < 192) return false;
  if (mink >= 0) return K >= mink;
  if (v2_all && K >= 1024) return true;
  return act == 2 || (act <= 1 && (N >= 2048 || K >= 1024));
}
// every k-major x k-major launch of the 256 x 256 LDS-DMA kernels goes through here: gemm5 when it takes the shape and the
// output allows the 16-byte register epilogue (asked BEFORE launching: an error of the launch itself is reported, not retried
// on the other kernel)
int launch_kk(const void* A, const void* B, void* C, const float* bias, const float* resid, int M, int N, int K, long lda,
              long ldb, long ldc, long ldr, int c_fp32, float alpha, int act, void* G, long ldg, int hm_n, int hm_heads,
              hipStream_t st) {
  if (gemm5_takes(N, K, act) && ctclip_gemm5_eligible(C, bias, resid, N, K, ldc, ldr, c_fp32, act))
    return ctclip_gemm5_launch(A, B, C, bias, resid, M, N, K, lda, ldb, ldc, ldr, c_fp32, alpha, act, G, ldg, hm_n, hm_heads, st);
  return ctclip_gemm3_launch_hm(A, B, C, bias, resid, M, N, K, lda, ldb, ldc, ldr, c_fp32, alpha, act, G, ldg, hm_n, hm_heads, st);
}
}  // namespace

extern "C" {

// See include/ctclip_hip.h for the contract.
int ctclip_gemm_bf16(const void* A, const void* B, void* C, const float* bias, const float* resid,
                     int M, int N, int K, long lda, long ldb, long ldc, long ldr,
                     int a_kmajor, int b_kmajor, int c_fp32, int split_k, int accumulate, float alpha, int act,
                     float* splitk_ws, long splitk_ws_floats, void* stream) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  if (bad_layout(A, lda, a_kmajor ? K : M, a_kmajor) || bad_layout(B, ldb, b_kmajor ? K : N, b_kmajor)) return (int)hipErrorInvalidValue;
  if (act != 0 && act != 1) return (int)hipErrorInvalidValue;          // 0 none, 1 erf-GELU: nothing else is part of the ABI
  if (split_k > 1 && !accumulate) return (int)hipErrorInvalidValue;
  if (accumulate && (!c_fp32 || act != 0)) return (int)hipErrorInvalidValue;
  if ((long)M * N >= (1L << 31)) splitk_ws = nullptr;
  // Split-K with a workspace: every split stores its [M, N] partial product with plain stores (4-5x the rate of float
  // atomics, which run at ~1.3 TB/s chip-wide: MI355X_MICROARCH.md) and ctclip_reduce_partials adds the partials to C in split
  // order -- C += A B is then bit-reproducible.  The split count is cut to what the workspace holds; one split (or no
  // workspace) accumulates straight into C, where every element has a single writer per launch.
  float* part = nullptr;
  if (accumulate && split_k > 1 && splitk_ws && ldc == N) {
    const long per = (long)M * N;
    const long fit = splitk_ws_floats / per;
    if (split_k > fit) split_k = (int)(fit < 1 ? 1 : fit);
    if (split_k > 1) part = splitk_ws;
  }
  auto finish = [&](int e, int splits) -> int {                        // second stage of the workspace form
    // one split (the launchers clamp the count to the K-steps there are): the kernel has accumulated straight into C and the
    // workspace was never written
    if (e || !part || splits <= 1) return e;
    return ctclip_reduce_partials(part, splits, (long)M * N, (int)((long)M * N), (float*)C, (hipStream_t)stream);
  };
  {
    // Problems with K % 64 == 0 and enough 256x128 tiles to fill the chip go to the pipelined LDS-DMA kernel (gemm2.hip):
    // 969/957 vs 748/750 TFLOP/s at 4096^3/8192^3; at 32 pairs/GPU FF1 forward 622 vs 561, FF1 wgrad 675 vs 526, kv wgrad
    // 753 vs 470 (profiles/r01_gemm_v1_v2.txt).  Ragged-K (K % 8 == 0) and small problems (BERT at M = 128*B) stay on the
    // register-staged 128x128 kernel below.  CTCLIP_GEMM_V2_ALL=1 lowers the size gate (the tests do, to cover every
    // path with small shapes); CTCLIP_GEMM_V1=1 disables the pipelined kernel.
    static const bool v2_all = getenv("CTCLIP_GEMM_V2_ALL") != nullptr;
    static const bool force_v1 = CTCLIP_KNOB("CTCLIP_GEMM_V1") != nullptr;
    // k-major x k-major products with many 256 x 256 tiles (every forward / data-gradient projection over all tokens) go
    // to the deeper, more compute-dense gemm3.hip.  CTCLIP_GEMM_NO_V3=1 disables it.
    static const bool no_v3 = CTCLIP_KNOB("CTCLIP_GEMM_NO_V3") != nullptr;
    const long blocks3 = (long)((M + 255) / 256) * ((N + 255) / 256);
    if (!no_v3 && !force_v1 && a_kmajor && b_kmajor && (K % 32) == 0 && split_k <= 1 && !accumulate &&
        (blocks3 >= 192 || (v2_all && blocks3 >= 4)))
      return launch_kk(A, B, C, bias, resid, M, N, K, lda, ldb, ldc, ldr, c_fp32, alpha, act, nullptr, 0, 0, 0,
                       (hipStream_t)stream);
    // weight gradients (m-major x n-major, split over the tokens) with enough 256 x 256 x split workgroups go to gemm4.hip,
    // the transposed-operand form of the same tile.  CTCLIP_GEMM_NO_V4=1 disables it.
    static const bool no_v4 = CTCLIP_KNOB("CTCLIP_GEMM_NO_V4") != nullptr;
    const long blocks4 = blocks3 * (split_k > 1 ? split_k : 1);
    if (!no_v4 && !force_v1 && !a_kmajor && !b_kmajor && accumulate && c_fp32 && !bias && !resid && (K % 32) == 0 &&
        (blocks4 >= 128 || v2_all)) {
      const int nk = K / 32;
      const int s_eff = split_k < 1 ? 1 : (split_k > nk ? nk : split_k);
      const int kps = (nk + s_eff - 1) / s_eff;
      return finish(ctclip_gemm4_launch(A, B, C, M, N, K, lda, ldb, ldc, split_k, alpha, part, (hipStream_t)stream),
                    (nk + kps - 1) / kps);
    }
    const long blocks2 = (long)((M + 255) / 256) * ((N + 127) / 128) * (split_k > 1 ? split_k : 1);
    const bool eligible = !force_v1 && (K % 64) == 0 && blocks2 >= 192;
    if (eligible || (v2_all && !force_v1 && (K % 64) == 0 && blocks2 >= 8)) {
      const int nk = K / 64;
      const int s_eff = split_k < 1 ? 1 : (split_k > nk ? nk : split_k);
      const int kps = (nk + s_eff - 1) / s_eff;
      return finish(ctclip_gemm2_launch(A, B, C, bias, resid, M, N, K, lda, ldb, ldc, ldr, a_kmajor, b_kmajor, c_fp32, split_k,
                                        accumulate, alpha, act, part, (hipStream_t)stream),
                    (nk + kps - 1) / kps);
    }
  }
  GemmArgs g{};
  g.A = (const bf16_t*)A; g.B = (const bf16_t*)B; g.C = C; g.bias = bias; g.resid = resid;
  g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.ldr = ldr; g.M = M; g.N = N; g.K = K;
  g.tiles_m = (M + BM - 1) / BM; g.tiles_n = (N + BN - 1) / BN;
  const int nk = (K + BKT - 1) / BKT;
  if (split_k < 1) split_k = 1;
  if (split_k > nk) split_k = nk;
  g.ktiles_per_split = (nk + split_k - 1) / split_k;
  g.split_k = (nk + g.ktiles_per_split - 1) / g.ktiles_per_split;
  g.c_fp32 = c_fp32; g.atomic_out = accumulate ? 1 : 0; g.act = act; g.alpha = alpha;
  if (g.split_k <= 1) part = nullptr;
  g.part = part;
  return finish(launch<0>(g, a_kmajor, b_kmajor, (hipStream_t)stream), g.split_k);
}

// The one-wave-per-SIMD kernel of gemm5.hip called directly (k-major x k-major; act 0 / 1 as ctclip_gemm_bf16, 2 = FF1 + GEGLU
// with G = g [M, N / 2], 3 = FF2 data gradient + GEGLU backward with G = h): what the dispatcher does for the shapes
// gemm5_takes() names, for any shape the kernel is eligible for.  hipErrorInvalidValue when it is not (K % 64, K < 192, outputs
// that do not allow the 16-byte epilogue).
int ctclip_gemm5_bf16(const void* A, const void* B, void* C, const float* bias, const float* resid, int M, int N, int K, long lda,
                      long ldb, long ldc, long ldr, int c_fp32, float alpha, int act, void* G, long ldg, void* stream) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  if (bad_layout(A, lda, K, true) || bad_layout(B, ldb, K, true)) return (int)hipErrorInvalidValue;
  return ctclip_gemm5_launch(A, B, C, bias, resid, M, N, K, lda, ldb, ldc, ldr, c_fp32, alpha, act, G, ldg, 0, 0,
                             (hipStream_t)stream);
}

// C = A[M,K] B[N,K]^T written as bf16 in the HEAD-MAJOR layout of attention_hm.hip: [part][sequence][head][token][32]
// with token = row % n_tokens, sequence = row / n_tokens, head = (col / 32) % heads, part = col / (32 heads).
// Always the 256 x 256 LDS-DMA kernel of gemm3.hip (its epilogue has the form): K % 32 == 0, N % 64 == 0, M % n_tokens == 0.
int ctclip_gemm_bf16_headmajor(const void* A, const void* B, void* C, int M, int N, int K, long lda, long ldb, int n_tokens,
                               int heads, void* stream) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  if (bad_layout(A, lda, K, true) || bad_layout(B, ldb, K, true) || (K % 32) || n_tokens <= 0) return (int)hipErrorInvalidValue;
  return launch_kk(A, B, C, nullptr, nullptr, M, N, K, lda, ldb, N, 0, 0, 1.0f, 0, nullptr, 0, n_tokens, heads,
                   (hipStream_t)stream);
}

// C = A[M,K] B[N,K]^T with the per-head cosine normalisation of attention.py:146-153 applied in the epilogue to the heads of the
// first norm_cols columns (include/ctclip_hip.h).  n_tokens > 0: head-major output as ctclip_gemm_bf16_headmajor; 0: row-major
// [M, ldc].  Always the 256 x 256 LDS-DMA kernel of gemm3.hip (its register epilogue has the form).
int ctclip_gemm_bf16_headnorm(const void* A, const void* B, void* C, float* inv_norm, const float* scale, int M, int N, int K,
                              long lda, long ldb, long ldc, int n_tokens, int heads, int norm_cols, float mult, void* stream) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  if (bad_layout(A, lda, K, true) || bad_layout(B, ldb, K, true) || (K % 32) || n_tokens < 0) return (int)hipErrorInvalidValue;
  return ctclip_gemm3_launch_hn(A, B, C, M, N, K, lda, ldb, n_tokens > 0 ? N : ldc, n_tokens, heads, scale, mult, inv_norm,
                                norm_cols, (hipStream_t)stream);
}

// dx = A[M,K] B[N,K]^T - c1[row] - xhat[row][col] c2[row] + resid  (f32, optional bf16 copy): a data-gradient product with the
// LayerNorm backward applied in its epilogue (include/ctclip_hip.h).  Always the LDS-DMA kernel of gemm3.hip.
int ctclip_gemm_bf16_lnbwd(const void* A, const void* B, float* dx, void* dx_bf16, int M, int N, int K, long lda, long ldb,
                           const void* xhat, const float* c1, const float* c2, const float* dres, void* stream) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  if (bad_layout(A, lda, K, true) || bad_layout(B, ldb, K, true) || (K % 32)) return (int)hipErrorInvalidValue;
  return ctclip_gemm3_launch_ln(A, B, dx, dx_bf16, dres, M, N, K, lda, ldb, N, N, N, xhat, N, c1, c2, (hipStream_t)stream);
}

// scores[m][n] = sum_k A[m][k] B[n][k], never materialised; for every column n writes 16 candidates: the top-4 of each
// of 4 disjoint row subsets.  part_val/part_idx are [N][16] (index 0x7fffffff = empty slot).
int ctclip_vq_topk(const void* A, const void* B, float* part_val, int* part_idx, int M, int N, int K, long lda, long ldb,
                   void* stream) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  if (bad_layout(A, lda, K, true) || bad_layout(B, ldb, K, true)) return (int)hipErrorInvalidValue;
  // whole 256-code tiles: the 256 x 256 LDS-DMA form in gemm3.hip (CTCLIP_VQ_NO_V3=1 keeps the kernel below)
  static const bool no_v3 = CTCLIP_KNOB("CTCLIP_VQ_NO_V3") != nullptr;
  if (!no_v3 && (M % 256) == 0 && (K % 32) == 0)
    return ctclip_vq_topk3_launch(A, B, part_val, part_idx, M, N, K, lda, ldb, 1, (hipStream_t)stream);
  GemmArgs g{};
  g.A = (const bf16_t*)A; g.B = (const bf16_t*)B; g.lda = lda; g.ldb = ldb; g.M = M; g.N = N; g.K = K;
  g.tiles_m = (M + BM - 1) / BM; g.tiles_n = (N + BN - 1) / BN;
  g.part_val = part_val; g.part_idx = part_idx; g.n_parts = 4 * VQ_TOP;
  hipLaunchKernelGGL(gemm_vq_topk_kernel, dim3(g.tiles_n), dim3(NTHREADS), 4 * TILE_BYTES, (hipStream_t)stream, g);
  return (int)hipGetLastError();
}

// The same search with the codebook split into `code_groups` groups of whole 256-code tiles (gemm3.hip, "CODE GROUPS"): 16
// candidates per token AND group, part_val / part_idx are [N][16 * code_groups].  code_groups must divide 32 and M / 256;
// M % 256 == 0, K % 32 == 0.  code_groups = 1 is ctclip_vq_topk.
int ctclip_vq_topk_grouped(const void* A, const void* B, float* part_val, int* part_idx, int M, int N, int K, long lda,
                           long ldb, int code_groups, void* stream) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  if (code_groups == 1) return ctclip_vq_topk(A, B, part_val, part_idx, M, N, K, lda, ldb, stream);
  if (bad_layout(A, lda, K, true) || bad_layout(B, ldb, K, true) || (M % 256) || (K % 32)) return (int)hipErrorInvalidValue;
  return ctclip_vq_topk3_launch(A, B, part_val, part_idx, M, N, K, lda, ldb, code_groups, (hipStream_t)stream);
}

// (older per-tile variant, kept for the kernel tests) top-2 over the rows of each 64-row slab:
// part_val/part_idx are [N][n_parts][2], n_parts = 2*ceil(M/128) (index 0x7fffffff = empty).
int ctclip_gemm_argmax_partial(const void* A, const void* B, float* part_val, int* part_idx,
                               int M, int N, int K, long lda, long ldb, void* stream) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  if (bad_layout(A, lda, K, true) || bad_layout(B, ldb, K, true)) return (int)hipErrorInvalidValue;
  GemmArgs g{};
  g.A = (const bf16_t*)A; g.B = (const bf16_t*)B; g.lda = lda; g.ldb = ldb; g.M = M; g.N = N; g.K = K;
  g.tiles_m = (M + BM - 1) / BM; g.tiles_n = (N + BN - 1) / BN;
  g.split_k = 1; g.ktiles_per_split = (K + BKT - 1) / BKT; g.alpha = 1.f;
  g.part_val = part_val; g.part_idx = part_idx; g.n_parts = 2 * g.tiles_m;
  return launch<1>(g, 1, 1, (hipStream_t)stream);
}

int ctclip_gemm_bf16_geglu(const void* A, const void* Bw, void* H, void* G, int M, int inner, int K, long lda, long ldb,
                           long ldh, long ldg, void* stream) {
  if (M <= 0 || inner <= 0) return 0;
  if ((inner & 63) || K <= 0) return (int)hipErrorInvalidValue;
  const int N = 2 * inner;
  static const bool no_v3 = CTCLIP_KNOB("CTCLIP_GEMM_NO_V3") != nullptr;
  static const bool v2_all = getenv("CTCLIP_GEMM_V2_ALL") != nullptr;
  const long blocks3 = (long)((M + 255) / 256) * ((N + 255) / 256);
  const bool aligned = (ldh & 7) == 0 && (ldg & 7) == 0 && ((((uintptr_t)H) | ((uintptr_t)G)) & 15) == 0;
  if (!no_v3 && aligned && (K % 32) == 0 && (N % 256) == 0 && (blocks3 >= 192 || (v2_all && blocks3 >= 4)))
    return launch_kk(A, Bw, H, nullptr, nullptr, M, N, K, lda, ldb, ldh, 0, 0, 1.0f, 2, G, ldg, 0, 0, (hipStream_t)stream);
  // small problems: the plain product, then the gated activation over the same interleaved layout
  if (int e = ctclip_gemm_bf16(A, Bw, H, nullptr, nullptr, M, N, K, lda, ldb, ldh, 0, 1, 1, 0, 1, 0, 1.0f, 0, nullptr, 0, stream)) return e;
  return ctclip_geglu_fwd(H, G, M, inner, 32, ldh, ldg, stream);
}

int ctclip_gemm_bf16_geglu_bwd(const void* dY, const void* W2T, void* H_dH, void* dG_scratch, int M, int inner, int K,
                               long lddy, long ldw, long ldh, long lddg, void* stream) {
  if (M <= 0 || inner <= 0) return 0;
  if ((inner & 63) || K <= 0) return (int)hipErrorInvalidValue;
  static const bool no_v3 = CTCLIP_KNOB("CTCLIP_GEMM_NO_V3") != nullptr;
  static const bool v2_all = getenv("CTCLIP_GEMM_V2_ALL") != nullptr;
  const long blocks3 = (long)((M + 255) / 256) * ((inner + 255) / 256);
  const bool aligned = (ldh & 7) == 0 && (((uintptr_t)H_dH) & 15) == 0;
  if (!no_v3 && aligned && (K % 32) == 0 && (blocks3 >= 192 || (v2_all && blocks3 >= 4)))
    return launch_kk(dY, W2T, nullptr, nullptr, nullptr, M, inner, K, lddy, ldw, 0, 0, 0, 1.0f, 3, H_dH, ldh, 0, 0,
                     (hipStream_t)stream);
  // small problems: dg into the scratch, then the blocked GEGLU backward in place over h
  if (!dG_scratch) return (int)hipErrorInvalidValue;
  if (int e = ctclip_gemm_bf16(dY, W2T, dG_scratch, nullptr, nullptr, M, inner, K, lddy, ldw, lddg, 0, 1, 1, 0, 1, 0, 1.0f, 0, nullptr, 0, stream))
    return e;
  return ctclip_geglu_bwd(dG_scratch, H_dH, H_dH, M, inner, 32, lddg, ldh, stream);
}

}  // extern "C"
