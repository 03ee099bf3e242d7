// bf16 MFMA GEMM, pipelined variant for the large shapes of the CT-CLIP step (gfx950).
//
// Same contract as gemm.hip (operand-layout flags, epilogue, split-K) but restructured around the two things the
// first kernel's profile showed (profiles/r01_*): (1) every K-step drained its global loads (`vmcnt(0)`) so the
// slowest of 8 in-order loads -- an L2 miss ~20 % of the time -- stalled all four waves; (2) only 64 FLOP per operand
// byte.  Here:
//   * block tile 256 x 128 x 64, 512 threads = 8 waves (4 x 2), each wave 64 x 64 as 2x2 MFMA 32x32x16 (85 FLOP/B);
//   * operands go HBM/L2 -> LDS directly with global_load_lds (16 B per lane, no VGPR staging), into a 3-stage ring
//     (3 x 48 KiB); tile t+2 is issued while tile t is consumed and the wait is a COUNTED `s_waitcnt vmcnt(6)`:
//     one whole tile stays in flight across the (raw) s_barrier, one barrier per K-step;
//   * LDS-DMA writes 1 KiB per wave-instruction linearly (base + lane*16), so the bank-conflict swizzle of gemm.hip is
//     applied to the per-lane SOURCE address and again on the fragment reads (same involution), never to the
//     destination;
//   * epilogue: accumulators -> LDS (f32 [256][128], reusing the ring) -> 16-byte coalesced stores with bias /
//     residual / GELU / bf16 conversion fused; split-K keeps the register-direct f32 atomics.
// Preconditions (checked by the dispatcher in gemm.hip): K % 64 == 0.  Rows/columns beyond M/N are fetched from a
// clamped in-range address and masked in the epilogue.
#include "common.h"

namespace g2 {

constexpr int BM = 256, BN = 128, BK = 64, NT = 512, NS = 3;
constexpr int SUB = 16384;                 // one 128-row (or 128-col) operand sub-tile: 128 x 64 bf16
constexpr int STAGE = 3 * SUB;             // A0, A1, B
constexpr int PIECES = STAGE / 1024;       // 48 LDS-DMA pieces of 1 KiB per stage
constexpr int PPW = PIECES / (NT / 64);    // 6 per wave

struct Args {
  const bf16_t* A; const bf16_t* B; void* C; const float* bias; const float* resid;
  long lda, ldb, ldc, ldr;
  int M, N, K, tiles_m, tiles_n, split_k, ktiles_per_split, c_fp32, atomic_out, act;
  float alpha;
  float* part;       // split-K without atomics: split ks stores its tile into part[ks][M][N]; null = atomics
};

template <bool KM>
__device__ __forceinline__ uint32_t tile_off(int rk, int chunk) {
  if (KM) return (uint32_t)(rk * 128 + ((chunk ^ ((rk >> 1) & 7)) << 4));
  return (uint32_t)(rk * 256 + ((chunk ^ (((rk & 3) << 2) | ((rk >> 2) & 3))) << 4));
}

// Transposed LDS read issued as inline asm: with the builtin, hipcc cannot prove the read does not alias the LDS-DMA writes
// still in flight and drains them (`s_waitcnt vmcnt(0)`) in front of every fragment group, which serialises the
// pipeline.  The asm form is invisible to that analysis; its completion is covered by tr_wait() below.
__device__ __forceinline__ short4v tr_read_asm(uint32_t lds_addr) {
  short4v r;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r) : "v"(lds_addr));
  return r;
}
// s_waitcnt lgkmcnt(0) that data-depends on every register the asm reads above wrote (guide 5.7, form ii): nothing that
// consumes them can be scheduled above the wait.
__device__ __forceinline__ void tr_wait(short4v (&p)[8]) {
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7])
               :
               : "memory");
}

// k-major fragment: one ds_read_b128 (compiler-tracked)
__device__ __forceinline__ bf16x8 read_frag_km(const char* tile, int rbase, int s, int lane) {
  const int r = rbase + (lane & 31);
  return *(const bf16x8*)(tile + tile_off<true>(r, 2 * s + (lane >> 5)));
}
// m/n-major fragment, raw halves (join after tr_wait)
__device__ __forceinline__ void read_frag_tr(uint32_t tile_lds, int rbase, int s, int lane, short4v& lo, short4v& hi) {
  const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, p = i16 & 3, h = g >> 1;
  const int m0 = rbase + 16 * (g & 1) + 4 * p;
  const int klo = 16 * s + 8 * h + q;
  const uint32_t sub = (uint32_t)((p & 1) * 8);
  lo = tr_read_asm(tile_lds + tile_off<false>(klo, m0 >> 3) + sub);
  hi = tr_read_asm(tile_lds + tile_off<false>(klo + 4, m0 >> 3) + sub);
}

// Source address (element offset from the operand base, at k-tile 0) of the 16 bytes lane `lane` contributes to 1 KiB
// piece `p` of a sub-tile whose first row/column is `r0`.  LDS destination of that lane is piece_base + lane*16.
template <bool KM>
__device__ __forceinline__ long piece_src(int p, int lane, int r0, int R, long ld) {
  if (KM) {                                        // piece = rows 8p..8p+7, 8 chunks each
    const int r = 8 * p + (lane >> 3), pc = lane & 7, c = pc ^ ((r >> 1) & 7);
    int row = r0 + r;
    if (row >= R) row = R - 1;                     // masked in the epilogue
    return (long)row * ld + c * 8;
  } else {                                         // piece = k rows 4p..4p+3, 16 chunks each
    const int k = 4 * p + (lane >> 4), pc = lane & 15, c = pc ^ (((k & 3) << 2) | ((k >> 2) & 3));
    int col = r0 + c * 8;
    if (col >= R) col = 0;                         // masked in the epilogue
    return (long)k * ld + col;
  }
}

__device__ __forceinline__ float gelu_erf(float x) { return gelu_erf_fast(x); }

#define G2_GLDS(gptr, ldsoff)                                                                                     \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),                          \
                                   (__attribute__((address_space(3))) void*)(uintptr_t)(ldsoff), 16, 0, 0)

template <bool AKM, bool BKM>
__global__ __launch_bounds__(NT, 2) void gemm2_kernel(Args g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;

  int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tn = bid % g.tiles_n; bid /= g.tiles_n;
  const int tm = bid % g.tiles_m;
  const int ks = bid / g.tiles_m;
  const int row0 = tm * BM, col0 = tn * BN;
  const int nk_total = g.K / BK;
  const int kt_begin = ks * g.ktiles_per_split;
  const int kt_end = min(nk_total, kt_begin + g.ktiles_per_split);
  const int nk = kt_end - kt_begin;

  // this wave's 6 pieces of every stage: piece id q = wave*6 + j -> sub-tile q/16 (0,1 = A halves, 2 = B), piece q%16
  const bf16_t* src[PPW];
  long kstep[PPW];
  uint32_t dst[PPW];
#pragma unroll
  for (int j = 0; j < PPW; ++j) {
    const int q = wave * PPW + j, st = q >> 4, p = q & 15;
    if (st < 2) {
      src[j] = g.A + piece_src<AKM>(p, lane, row0 + st * 128, g.M, g.lda) + (AKM ? (long)kt_begin * BK : (long)kt_begin * BK * g.lda);
      kstep[j] = AKM ? BK : (long)BK * g.lda;
    } else {
      src[j] = g.B + piece_src<BKM>(p, lane, col0, g.N, g.ldb) + (BKM ? (long)kt_begin * BK : (long)kt_begin * BK * g.ldb);
      kstep[j] = BKM ? BK : (long)BK * g.ldb;
    }
    dst[j] = (uint32_t)(st * SUB + p * 1024);
  }
  auto issue = [&](int t) {                        // tile t (relative) -> stage t % NS
    const uint32_t sb = lds0 + (uint32_t)((t % NS) * STAGE);
#pragma unroll
    for (int j = 0; j < PPW; ++j) G2_GLDS(src[j] + (long)t * kstep[j], sb + dst[j]);
  };
  auto issue_part = [&](int t, int j0) {           // two of the six pieces: spread between the MFMA groups of a K-step
    const uint32_t sb = lds0 + (uint32_t)((t % NS) * STAGE);
    G2_GLDS(src[j0] + (long)t * kstep[j0], sb + dst[j0]);
    G2_GLDS(src[j0 + 1] + (long)t * kstep[j0 + 1], sb + dst[j0 + 1]);
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (nk > 0) {
    issue(0);
    if (nk > 1) issue(1);
    for (int t = 0; t < nk; ++t) {
      // tile t must have landed; tile t+1 (6 younger LDS-DMAs of this wave) may stay in flight across the barrier
      if (t + 1 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                // also: every wave is done reading stage (t+2) % NS (tile t-1)
      const bool pre = t + 2 < nk;
      const char* sa = smem + (t % NS) * STAGE + (wm >> 1) * SUB;
      const char* sb = smem + (t % NS) * STAGE + 2 * SUB;
#pragma unroll
      for (int s = 0; s < BK / 16; ++s) {
        bf16x8 fa[2], fb[2];
        short4v raw[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) raw[i] = short4v{0, 0, 0, 0};
        const uint32_t sa_l = lds0 + (uint32_t)(sa - smem), sb_l = lds0 + (uint32_t)(sb - smem);
        if (AKM) {
          fa[0] = read_frag_km(sa, (wm & 1) * 64, s, lane);
          fa[1] = read_frag_km(sa, (wm & 1) * 64 + 32, s, lane);
        } else {
          read_frag_tr(sa_l, (wm & 1) * 64, s, lane, raw[0], raw[1]);
          read_frag_tr(sa_l, (wm & 1) * 64 + 32, s, lane, raw[2], raw[3]);
        }
        if (BKM) {
          fb[0] = read_frag_km(sb, wn * 64, s, lane);
          fb[1] = read_frag_km(sb, wn * 64 + 32, s, lane);
        } else {
          read_frag_tr(sb_l, wn * 64, s, lane, raw[4], raw[5]);
          read_frag_tr(sb_l, wn * 64 + 32, s, lane, raw[6], raw[7]);
        }
        if (!AKM || !BKM) {
          tr_wait(raw);
          if (!AKM) { fa[0] = join_tr(raw[0], raw[1]); fa[1] = join_tr(raw[2], raw[3]); }
          if (!BKM) { fb[0] = join_tr(raw[4], raw[5]); fb[1] = join_tr(raw[6], raw[7]); }
        }
        // tile t+2's LDS-DMA is issued two pieces at a time BETWEEN the MFMA groups (not all six right after the
        // barrier, where every wave would sit in ~60-cycle issue slots with the matrix pipe idle)
        if (pre && s < 3) issue_part(t + 2, 2 * s);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = mfma32(fa[i], fb[j], acc[i][j]);
      }
    }
  }

  const int half = lane >> 5, lc = lane & 31;
  if (g.atomic_out) {
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

  __syncthreads();                                 // all fragment reads of the last stage are done: ring becomes C staging
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
    for (int it = 0; it < (BM * BN / 4) / NT; ++it) {
      const int id = it * NT + tid, r = id >> 5, c4 = (id & 31) * 4;
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
    for (int it = 0; it < (BM * BN / 8) / NT; ++it) {
      const int id = it * NT + tid, r = id >> 4, c8 = (id & 15) * 8;
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

}  // namespace g2

// called by ctclip_gemm_bf16 (gemm.hip) when K % 64 == 0 and the problem is large enough
int ctclip_gemm2_launch(const void* A, const void* B, void* C, const float* bias, const float* resid, int M, int N, int K,
                        long lda, long ldb, long ldc, long ldr, int a_kmajor, int b_kmajor, int c_fp32, int split_k,
                        int accumulate, float alpha, int act, float* part, hipStream_t st) {
  using namespace g2;
  Args g{};
  g.A = (const bf16_t*)A; g.B = (const bf16_t*)B; g.C = C; g.bias = bias; g.resid = resid;
  g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.ldr = ldr; g.M = M; g.N = N; g.K = K;
  g.tiles_m = (M + BM - 1) / BM; g.tiles_n = (N + BN - 1) / BN;
  const int nk = K / BK;
  if (split_k < 1) split_k = 1;
  if (split_k > nk) split_k = nk;
  g.ktiles_per_split = (nk + split_k - 1) / split_k;
  g.split_k = (nk + g.ktiles_per_split - 1) / g.ktiles_per_split;
  g.c_fp32 = c_fp32; g.atomic_out = accumulate ? 1 : 0; g.act = act; g.alpha = alpha;
  g.part = g.split_k > 1 ? part : nullptr;
  const int nblk = g.tiles_m * g.tiles_n * g.split_k;
  const size_t lds = (size_t)NS * STAGE;           // 144 KiB: needs the opt-in above 64 KiB
  dim3 grid(nblk), block(NT);
#define G2_LAUNCH(AK, BK_)                                                                                          \
  do {                                                                                                              \
    CTCLIP_LDS_LIMIT_ONCE((gemm2_kernel<AK, BK_>), lds);                                                            \
    hipLaunchKernelGGL((gemm2_kernel<AK, BK_>), grid, block, lds, st, g);                                           \
  } while (0)
  if (a_kmajor && b_kmajor) G2_LAUNCH(true, true);
  else if (a_kmajor && !b_kmajor) G2_LAUNCH(true, false);
  else if (!a_kmajor && b_kmajor) G2_LAUNCH(false, true);
  else G2_LAUNCH(false, false);
#undef G2_LAUNCH
  return (int)hipGetLastError();
}
