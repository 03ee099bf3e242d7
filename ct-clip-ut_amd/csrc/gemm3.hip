// bf16 MFMA GEMM, 256 x 256 tile variant for the k-major x k-major products of the CT-CLIP step (every forward and
// data-gradient projection: C[M,N] = A[M,K] B[N,K]^T with M = all tokens).
//
// gemm2.hip (256 x 128 x 64, 64 x 64 per wave) tops out near 40 % of the MFMA peak: each CU has to pull 48 KiB through
// L2 -> LDS for every 1024 matrix-pipe cycles and only two stages (96 KiB) can be in flight, so with ~1-2 us of loaded
// L2/HBM latency the ring runs dry (profiles/r01_gemm_pmc.txt: matrix pipe 28 % busy, LDS 37 %, waves waiting 43 %).
// This kernel trades tile shape for bytes per flop and depth:
//   * block tile 256 x 256 x 32, 512 threads = 8 waves (2 x 4), each wave 128 x 64 as 4x2 MFMA 32x32x16
//     (128 FLOP per operand byte instead of 85; 6 KiB of LDS fragment reads per 8 MFMAs instead of 4 KiB per 4);
//   * a FOUR-stage ring of 32 KiB stages (128 KiB): three K-steps in flight ahead of the one being consumed;
//   * global_load_lds (16 B per lane) with the bank swizzle on the SOURCE address, counted `s_waitcnt vmcnt(8)` and one
//     raw s_barrier per K-step, as in gemm2.hip;
//   * epilogue through the ring as f32 [128][256] halves -> 16-byte coalesced stores (bias / residual / GELU fused).
// Preconditions (checked by the dispatcher in gemm.hip): both operands k-major, K % 32 == 0, no split-K / accumulate.
#include "common.h"
#include <stdlib.h>

namespace g3 {

#ifdef CTCLIP_G3_STAMPS
// diagnostic build only (hipcc -DCTCLIP_G3_STAMPS; never compiled into the shipped library): per-workgroup phase stamps
// {hw id, xcc id, start, first K-step landed, matrix loop done, epilogue done} in 10 ns ticks, read by tools/gemm_timeline.py
__device__ unsigned long long* g_stamps = nullptr;
__device__ long g_stamp_cap = 0;
#define G3_STAMP(slot)                                                                                            \
  do {                                                                                                            \
    if (g_stamps && threadIdx.x == 0 && (long)blockIdx.x < g_stamp_cap)                                           \
      g_stamps[(long)blockIdx.x * 8 + (slot)] = __builtin_amdgcn_s_memrealtime();                                 \
  } while (0)
#else
#define G3_STAMP(slot) do { } while (0)
#endif

constexpr int BM = 256, BK = 32;
constexpr int SUB = 16384;                 // the A tile of a stage: 256 x 32 bf16 (the B tile follows it)

struct Args {
  const bf16_t* A; const bf16_t* B; void* C; const float* bias; const float* resid;
  long lda, ldb, ldc, ldr;
  int M, N, K, tiles_m, tiles_n, c_fp32, act;
  float alpha;
  bf16_t* G; long ldg;                             // act == 2: gelu(gate) * value of the 64-column interleaved [val|gate] blocks
                                                   // act == 3: G = h (pre-activations, same blocks), overwritten with d(h)
  int stagger, stagger_shift, stagger_limit;       // start-up delay (10 ns ticks) of the workgroups with bit `shift` of blockIdx set
};

// [256 rows][32 k] bf16 tile, 64-byte rows, 16-byte chunks XOR-swizzled so that the 16 lanes of a ds_read_b128 phase
// (16 consecutive rows, same logical chunk) hit 16 different bank groups
// Key = (-(r >> 2)) & 3 serves both fragment shapes: a 32x32x16 read (32 consecutive rows, one chunk per lane half) only
// needs the four row quads of a 16-lane phase on different keys; a 16x16x32 read (16 rows x all four chunks, chunk =
// lane >> 4) puts row quads {0, 3} with chunk c and {1, 2} with chunk c ^ 1 in one phase, and {k0, k3, 1 ^ k1, 1 ^ k2} =
// {0, 1, 2, 3} for this key (the plain key r >> 2 collides there).
__device__ __forceinline__ int swz_key(int r) { return (-(r >> 2)) & 3; }
__device__ __forceinline__ uint32_t tile_off(int r, int chunk) { return (uint32_t)(r * 64 + ((chunk ^ swz_key(r)) << 4)); }

__device__ __forceinline__ bf16x8 read_frag(const char* tile, int rbase, int s, int lane) {
  const int r = rbase + (lane & 31);
  return *(const bf16x8*)(tile + tile_off(r, 2 * s + (lane >> 5)));
}

// 16x16x32 operand fragment: lane l holds row rbase + (l & 15), k = 8 (l >> 4) .. + 7 of the 32-deep K-step
__device__ __forceinline__ bf16x8 read_frag16(const char* tile, int rbase, int lane) {
  return *(const bf16x8*)(tile + tile_off(rbase + (lane & 15), lane >> 4));
}
__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// element offset (from the operand base, at k-tile 0) of the 16 bytes lane `lane` contributes to 1 KiB piece `p`
// (rows 16p .. 16p+15, 4 chunks each) of the tile whose first row is r0; its LDS destination is piece_base + lane*16
__device__ __forceinline__ long piece_src(int p, int lane, int r0, int R, long ld) {
  const int r = 16 * p + (lane >> 2), pc = lane & 3, c = pc ^ swz_key(r);
  int row = r0 + r;
  if (row >= R) row = R - 1;                       // masked in the epilogue
  return (long)row * ld + c * 8;
}

__device__ __forceinline__ float gelu_erf(float x) { return gelu_erf_fast(x); }

// streaming 16-byte stores of the epilogue (gigabytes written once, read by a later kernel): non-temporal by default,
// CTCLIP_GEMM3_NO_NT=1 reverts to plain stores
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
template <bool NT> __device__ __forceinline__ void st16(void* p, uint4 v) {
  const u32x4_t x = {v.x, v.y, v.z, v.w};
  if (NT) __builtin_nontemporal_store(x, (u32x4_t*)p);
  else *(u32x4_t*)p = x;
}
template <bool NT> __device__ __forceinline__ void st16f(void* p, float a, float b, float c, float d) {
  const f32x4_t x = {a, b, c, d};
  if (NT) __builtin_nontemporal_store(x, (f32x4_t*)p);
  else *(f32x4_t*)p = x;
}

#define G3_GLDS(gptr, ldsoff)                                                                                     \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),                          \
                                   (__attribute__((address_space(3))) void*)(uintptr_t)(ldsoff), 16, 0, 0)

// BN = 256: 8 waves (2 x 4), 4-stage ring of 32 KiB, one workgroup per CU.
// BN = 128: 4 waves (2 x 2), 3-stage ring of 24 KiB = 72 KiB, TWO workgroups per CU: vmcnt is an in-order counter, so a
//           wave cannot see its next tile's LDS-DMAs complete before its own epilogue stores have drained to HBM; with two
//           independent workgroups one computes while the other writes back and refills its ring.
// M16: the per-wave 128 x 64 tile as 8 x 4 MFMA 16x16x32 instead of 4 x 2 MFMA 32x32x16 -- the same LDS bytes, registers
//      and matrix-pipe cycles per flop, but the part holds a higher clock on the smaller shape under load
//      (MI355X_MICROARCH.md, DVFS give-back item 7).
// ROLES (BN = 256, M16): the two waves of a SIMD (w and w + 4, i.e. the wm = 0 and wm = 1 halves of the workgroup) run
//      half a K-step apart.  In the plain loop all eight waves leave the K-step barrier together, all read their
//      fragments from LDS together and the matrix pipe idles meanwhile (measured with -DCTCLIP_G3_STAMPS: ~630 of ~1680
//      cycles per K-step, wave 0 parked at the barrier for 540 of them).  Here every barrier interval has one half in its
//      MFMA block and the other half in its load block, then they swap:
//          interval   2k     2k+1    2k+2
//          wm = 0     R_k    M_k     R_k+1        R = read the fragments of K-step k (12 ds_read_b128), issue this wave's
//          wm = 1     M_k-1  R_k     M_k              share of the LDS-DMA of K-step k + NS - 1;  M = the 32 MFMAs of K-step k
//      Stage k is read in intervals 2k (wm 0) and 2k+1 (wm 1), so it is refilled from interval 2k+2 on and must have landed
//      before interval 2k: each wave waits for its own pieces of stage k+1 (counted vmcnt) inside interval 2k+1.
//      sq4096 1171 -> 1261 TFLOP/s, ff1 dgrad (K = 2816) 973 -> 1018, the K = 512 shapes +0..2 %.
template <int BN, int NS, bool NTS, bool M16, bool ROLES = false>
__global__ __launch_bounds__(BN * 2, 2) void gemm3_kernel(Args g) {
  static_assert(!ROLES || (BN == 256 && NS == 4 && M16), "the role-alternating loop is written for the 8-wave 16x16x32 form");
  constexpr int WN = BN / 64, NT = 2 * WN * 64;
  constexpr int STAGE = SUB + BN * BK * 2;
  constexpr int PPW = (STAGE / 1024) / (2 * WN);   // LDS-DMA pieces of 1 KiB per wave and stage: 4 (BN 256) or 6 (BN 128)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;

  int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tn = bid % g.tiles_n;
  const int tm = bid / g.tiles_n;
  const int row0 = tm * BM, col0 = tn * BN;
  const int nk = g.K / BK;

  // piece q = wave * PPW + j of a stage: the first 16 are the A tile, the rest the B tile (stored right behind it)
  const bf16_t* src[PPW];
  uint32_t dst[PPW];
#pragma unroll
  for (int j = 0; j < PPW; ++j) {
    const int q = wave * PPW + j;
    src[j] = (q < 16) ? g.A + piece_src(q, lane, row0, g.M, g.lda) : g.B + piece_src(q - 16, lane, col0, g.N, g.ldb);
    dst[j] = (uint32_t)(q * 1024);
  }
  auto issue_part = [&](int t, int j0) {           // PPW/2 of this wave's pieces of K-step t -> stage t % NS
    const uint32_t sb = lds0 + (uint32_t)((t % NS) * STAGE);
#pragma unroll
    for (int j = j0; j < j0 + PPW / 2; ++j) G3_GLDS(src[j] + (long)t * BK, sb + dst[j]);
  };

  f32x16 acc[4][2];                                 // 32x32x16 form
  f32x4 acc16[8][4];                                // 16x16x32 form (only one of the two is live in an instantiation)
  if (M16) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc16[i][j][r] = 0.f;
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  }

#ifdef CTCLIP_G3_STAMPS
  if (g_stamps && threadIdx.x == 0 && (long)blockIdx.x < g_stamp_cap) {
    g_stamps[(long)blockIdx.x * 8 + 0] = __builtin_amdgcn_s_getreg((31 << 11) | 4);     // HW_REG_HW_ID
    g_stamps[(long)blockIdx.x * 8 + 1] = __builtin_amdgcn_s_getreg((31 << 11) | 20);    // HW_REG_XCC_ID
  }
#endif
  G3_STAMP(2);
  if (g.stagger > 0 && ((blockIdx.x >> g.stagger_shift) & 1) && blockIdx.x < g.stagger_limit) {
    // de-synchronise the two workgroups of a CU (BN = 128): they start together and do identical work, so they stay in
    // lock-step -- both in the matrix loop, then both writing back.  Half a tile period of delay for the second one puts
    // one workgroup's epilogue stores under the other's matrix loop; the pattern then carries itself through the grid.
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();      // 100 MHz
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)g.stagger) __builtin_amdgcn_s_sleep(8);
  }

#pragma unroll
  for (int t = 0; t < NS - 1; ++t)
    if (t < nk) { issue_part(t, 0); issue_part(t, PPW / 2); }

#ifdef CTCLIP_G3_STAMPS
  unsigned long long wait_vm = 0, wait_bar = 0;     // shader cycles wave 0 spends in the counted vmcnt wait / at the barrier
#endif
  if constexpr (ROLES) {
    bf16x8 fa[8], fb[4];
    // own pieces of K-step k+1 landed; up to two younger K-steps (PPW = 4 DMAs each) stay in flight
#define G3_WAIT_NEXT(k)                                                                \
    do {                                                                               \
      if ((k) + 1 < nk) {                                                              \
        const int y_ = nk - 2 - (k);                                                   \
        if (y_ >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                  \
        else if (y_ == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");             \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                          \
      }                                                                                \
    } while (0)
#define G3_BAR()                                                                       \
    do {                                                                               \
      __builtin_amdgcn_sched_barrier(0);                                               \
      __builtin_amdgcn_s_barrier();                                                    \
      __builtin_amdgcn_sched_barrier(0);                                               \
    } while (0)
    // the load block also carries this wave's four LDS-DMA pieces of K-step k + NS - 1: issued between the partner's MFMAs
    // instead (inside the MFMA block) each issue stalls the one wave that is feeding the matrix pipe -- measured slower
    // (sq4096 1141 vs 1261 TFLOP/s)
    auto load_block = [&](int k) {
      const char* sa = smem + (k % NS) * STAGE;
      const char* sb = sa + SUB;
#pragma unroll
      for (int j = 0; j < 4; ++j) fb[j] = read_frag16(sb, wn * 64 + j * 16, lane);
#pragma unroll
      for (int i = 0; i < 8; ++i) fa[i] = read_frag16(sa, wm * 128 + i * 16, lane);
      if (k + NS - 1 < nk) { issue_part(k + NS - 1, 0); issue_part(k + NS - 1, PPW / 2); }
    };
    auto mfma_block = [&]() {
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc16[i][j] = mfma16(fa[i], fb[j], acc16[i][j]);
      __builtin_amdgcn_s_setprio(0);
    };
    {                                               // stage 0 has landed for everybody
      const int y0 = nk - 1;
      if (y0 >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if (y0 == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    G3_BAR();
    G3_STAMP(3);
    if (wm == 0) {
      for (int k = 0; k < nk; ++k) {
        load_block(k);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        G3_BAR();
        mfma_block();
        G3_WAIT_NEXT(k);
        G3_BAR();
      }
      G3_BAR();                                     // the other half's last MFMA block
    } else {
      G3_BAR();                                     // interval 0: the other half reads stage 0
      for (int k = 0; k < nk; ++k) {
        load_block(k);
        G3_WAIT_NEXT(k);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        G3_BAR();
        mfma_block();
        G3_BAR();
      }
    }
#undef G3_WAIT_NEXT
#undef G3_BAR
  } else
  for (int t = 0; t < nk; ++t) {
    // K-step t must have landed; the PPW LDS-DMAs of each of the (up to NS-2) younger steps may stay in flight across the barrier
    const int younger = nk - 1 - t;
#ifdef CTCLIP_G3_STAMPS
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
#endif
    if (NS == 4) {
      if (younger >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if (younger == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      if (younger >= 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
#ifdef CTCLIP_G3_STAMPS
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
#endif
    __builtin_amdgcn_s_barrier();                  // also: every wave is done reading stage (t+3) % NS (K-step t-1)
#ifdef CTCLIP_G3_STAMPS
    if (t > 0) { wait_vm += c1 - c0; wait_bar += __builtin_amdgcn_s_memtime() - c1; }
#endif
    if (t == 0) G3_STAMP(3);
    const bool pre = t + NS - 1 < nk && !(g.act & 0x200);   // 0x200: timing experiment, no operand traffic after the prologue
    const char* sa = smem + (t % NS) * STAGE;
    const char* sb = sa + SUB;
    if (M16) {
      bf16x8 fa[8], fb[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) fb[j] = read_frag16(sb, wn * 64 + j * 16, lane);
#pragma unroll
      for (int i = 0; i < 8; ++i) fa[i] = read_frag16(sa, wm * 128 + i * 16, lane);
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) {
        if (pre) issue_part(t + NS - 1, h2 * (PPW / 2));
#pragma unroll
        for (int i = 4 * h2; i < 4 * h2 + 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc16[i][j] = mfma16(fa[i], fb[j], acc16[i][j]);
      }
    } else {
#pragma unroll
      for (int s = 0; s < BK / 16; ++s) {
        bf16x8 fa[4], fb[2];
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[i] = read_frag(sa, wm * 128 + i * 32, s, lane);
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[j] = read_frag(sb, wn * 64 + j * 32, s, lane);
        if (pre) issue_part(t + NS - 1, s * (PPW / 2));   // the next ring slot, half of the pieces between the MFMA groups
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = mfma32(fa[i], fb[j], acc[i][j]);
      }
    }
  }

  G3_STAMP(4);
#ifdef CTCLIP_G3_STAMPS
  if (g_stamps && threadIdx.x == 0 && (long)blockIdx.x < g_stamp_cap) {
    g_stamps[(long)blockIdx.x * 8 + 6] = wait_vm;
    g_stamps[(long)blockIdx.x * 8 + 7] = wait_bar;
  }
#endif
  // epilogue: two 128-row halves through the ring (f32 [128][BN]: 128 / 64 KiB); half h is owned by the waves with wm == h
  const int half = lane >> 5, lc = lane & 31;
  float* ct = (float*)smem;
  const int act = g.act & 0xff;
#pragma unroll 1
  for (int hh = 0; hh < 2; ++hh) {
    __syncthreads();                               // fragment reads of the last stage / the other half's stores are done
    if (wm == hh) {
      if (M16) {                                   // acc16[i][j][r] = D[16 i + 4 (lane >> 4) + r][16 j + (lane & 15)]
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              ct[(i * 16 + 4 * (lane >> 4) + r) * BN + wn * 64 + j * 16 + (lane & 15)] = acc16[i][j][r];
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) ct[(i * 32 + acc_row(r, half)) * BN + wn * 64 + j * 32 + lc] = acc[i][j][r];
      }
    }
    __syncthreads();
    const int hrow0 = row0 + hh * 128;
    if (g.c_fp32) {
      float* C = (float*)g.C;
      const bool vec = ((g.ldc & 3) == 0) && ((((uintptr_t)C) & 15) == 0) &&
                       (!g.resid || (((g.ldr & 3) == 0) && ((((uintptr_t)g.resid) & 15) == 0)));
#pragma unroll 4
      for (int it = 0; it < (128 * BN / 4) / NT; ++it) {
        const int id = it * NT + tid, r = id / (BN / 4), c4 = (id % (BN / 4)) * 4;
        const int row = hrow0 + r, col = col0 + c4;
        if (row >= g.M || col >= g.N) continue;
        const float4 t = *(const float4*)(ct + r * BN + c4);
        float v[4] = {t.x * g.alpha, t.y * g.alpha, t.z * g.alpha, t.w * g.alpha};
        if (vec && col + 3 < g.N) {
          if (g.bias) { const float4 b = *(const float4*)(g.bias + col); v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w; }
          if (g.resid) { const float4 q = *(const float4*)(g.resid + (long)row * g.ldr + col); v[0] += q.x; v[1] += q.y; v[2] += q.z; v[3] += q.w; }
          if (act == 1) { v[0] = gelu_erf(v[0]); v[1] = gelu_erf(v[1]); v[2] = gelu_erf(v[2]); v[3] = gelu_erf(v[3]); }
          st16f<NTS>(C + (long)row * g.ldc + col, v[0], v[1], v[2], v[3]);
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
      for (int it = 0; it < (128 * BN / 8) / NT; ++it) {
        const int id = it * NT + tid, r = id / (BN / 8), c8 = (id % (BN / 8)) * 8;
        const int row = hrow0 + r, col = col0 + c8;
        if (row >= g.M || col >= g.N) continue;
        const float4 t0 = *(const float4*)(ct + r * BN + c8), t1 = *(const float4*)(ct + r * BN + c8 + 4);
        float v[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= g.alpha;
        if (act == 3) {
          // GEGLU backward in the epilogue of dg = dy W2: this tile's dg never goes to memory; the matching value / gate
          // pre-activations are read from h and replaced by their gradients in place (attention.py:38-41)
          bf16_t* hv = g.G + (long)row * g.ldg + (long)(col >> 6) * 128 + (col & 63);
          const uint4 hval = *(const uint4*)hv, hgate = *(const uint4*)(hv + 64);
          const uint32_t wv[4] = {hval.x, hval.y, hval.z, hval.w}, wg[4] = {hgate.x, hgate.y, hgate.z, hgate.w};
          float dv[8], dt[8];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float v0 = __uint_as_float(wv[e] << 16), v1 = __uint_as_float(wv[e] & 0xffff0000u);
            const float t0g = __uint_as_float(wg[e] << 16), t1g = __uint_as_float(wg[e] & 0xffff0000u);
            dv[2 * e] = v[2 * e] * gelu_erf(t0g); dv[2 * e + 1] = v[2 * e + 1] * gelu_erf(t1g);
            dt[2 * e] = v[2 * e] * v0 * gelu_erf_grad_fast(t0g); dt[2 * e + 1] = v[2 * e + 1] * v1 * gelu_erf_grad_fast(t1g);
          }
          uint4 o0, o1;
          o0.x = pack_bf16x2(dv[0], dv[1]); o0.y = pack_bf16x2(dv[2], dv[3]); o0.z = pack_bf16x2(dv[4], dv[5]); o0.w = pack_bf16x2(dv[6], dv[7]);
          o1.x = pack_bf16x2(dt[0], dt[1]); o1.y = pack_bf16x2(dt[2], dt[3]); o1.z = pack_bf16x2(dt[4], dt[5]); o1.w = pack_bf16x2(dt[6], dt[7]);
          st16<NTS>(hv, o0);
          st16<NTS>(hv + 64, o1);
          continue;
        }
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
          st16<NTS>(C + (long)row * g.ldc + col, o);
          if (act == 2 && ((c8 >> 6) & 1) == 0) {  // a value chunk: its gate sits 64 columns to the right in the same tile
            const float4 g0 = *(const float4*)(ct + r * BN + c8 + 64), g1 = *(const float4*)(ct + r * BN + c8 + 68);
            const float gt[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
            float w[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) w[e] = gelu_erf(gt[e] * g.alpha) * v[e];
            uint4 og;
            og.x = pack_bf16x2(w[0], w[1]); og.y = pack_bf16x2(w[2], w[3]); og.z = pack_bf16x2(w[4], w[5]); og.w = pack_bf16x2(w[6], w[7]);
            st16<NTS>(g.G + (long)row * g.ldg + (col >> 7) * 64 + (col & 63), og);
          }
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
#ifdef CTCLIP_G3_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the stores have left the wave
#endif
  G3_STAMP(5);
}


// ------------------------------------------------------------------------------------------------------------------
// VQ nearest-code search on the same tile and ring (vector_quantize_pytorch cosine-sim codebook, ctvit.py:420-427):
// scores[code][token] = sum_k A[code][k] B[token][k] are never written.  One workgroup owns 256 tokens and sweeps ALL
// 256-code tiles; the (code tile, K-step) sequence is flattened so the ring never drains between tiles.  Every lane
// keeps a running top-4 (value, code) for each of its two token columns over the 64 codes per tile its accumulator
// registers cover; a token ends with 4 lane groups (wm, lane half) x 4 = 16 candidates from disjoint parts of the
// codebook, which ctclip_vq_select re-ranks exactly in f32.  Against the 128 x 128 register-staged kernel of gemm.hip
// this halves the operand bytes pulled through L2 per flop (the codebook is re-streamed once per 256 tokens, not 128).
// Preconditions (gemm.hip): M % 256 == 0, K % 32 == 0.
// ------------------------------------------------------------------------------------------------------------------
struct VqArgs {
  const bf16_t* A; const bf16_t* B;
  long lda, ldb;
  int M, N, K, tiles_m;
  float* part_val; int* part_idx;
};
constexpr int VQ_TOP = 4;                       // the insertion below is written out for exactly four places

__global__ __launch_bounds__(512, 2) void vq_topk3_kernel(VqArgs g) {
  constexpr int BN = 256, NS = 4, WN = 4, STAGE = 2 * SUB, PPW = 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  const int col0 = xcd_remap(blockIdx.x, gridDim.x) * BN;      // token tile
  const int nk = g.K / BK;
  const int steps = g.tiles_m * nk;                            // flattened (code tile, K-step) sequence
  const int half = lane >> 5, lc = lane & 31;

  const bf16_t* src[PPW];
  long tstep[PPW];
  uint32_t dst[PPW];
#pragma unroll
  for (int j = 0; j < PPW; ++j) {
    const int q = wave * PPW + j;
    if (q < 16) { src[j] = g.A + piece_src(q, lane, 0, g.M, g.lda); tstep[j] = (long)BM * g.lda; }
    else { src[j] = g.B + piece_src(q - 16, lane, col0, g.N, g.ldb); tstep[j] = 0; }
    dst[j] = (uint32_t)(q * 1024);
  }
  int pf = 0, pf_tile = 0, pf_k = 0;                           // the next step to request: index, its code tile and K-step
  auto issue_part = [&](int j0) {
    const uint32_t sb = lds0 + (uint32_t)((pf % NS) * STAGE);
#pragma unroll
    for (int j = j0; j < j0 + PPW / 2; ++j) G3_GLDS(src[j] + (long)pf_tile * tstep[j] + (long)pf_k * BK, sb + dst[j]);
  };
  auto advance = [&]() {
    ++pf;
    if (++pf_k == nk) { pf_k = 0; ++pf_tile; }
  };

  float bv[2][VQ_TOP];
  int bi[2][VQ_TOP];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int t = 0; t < VQ_TOP; ++t) { bv[j][t] = -INFINITY; bi[j][t] = 0x7fffffff; }
  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#pragma unroll
  for (int t = 0; t < NS - 1; ++t)
    if (t < steps) { issue_part(0); issue_part(PPW / 2); advance(); }

  int kt = 0, tile = 0;
  for (int st = 0; st < steps; ++st) {
    const int younger = steps - 1 - st;
    if (younger >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (younger == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const bool pre = pf < steps;
    const char* sa = smem + (st % NS) * STAGE;
    const char* sb = sa + SUB;
#pragma unroll
    for (int s = 0; s < BK / 16; ++s) {
      bf16x8 fa[4], fb[2];
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i] = read_frag(sa, wm * 128 + i * 32, s, lane);
#pragma unroll
      for (int j = 0; j < 2; ++j) fb[j] = read_frag(sb, wn * 64 + j * 32, s, lane);
      if (pre) issue_part(s * (PPW / 2));
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = mfma32(fa[i], fb[j], acc[i][j]);
    }
    if (pre) advance();
    if (++kt == nk) {                                          // a code tile is complete: fold it into the running top-4
      kt = 0;
      const int row_t = tile * BM + wm * 128;
      ++tile;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float v = acc[i][j][r];
            const int row = row_t + i * 32 + acc_row(r, half);
            // A lane meets its codes in ascending order (tile, i, r), so a newcomer never outranks an equal score
            // already listed: strict compares alone keep the list sorted by (score desc, code asc).  Entries below the
            // insertion point move down one place.
            if (v > bv[j][3]) {
              const bool c2 = v > bv[j][2], c1 = v > bv[j][1], c0 = v > bv[j][0];
              bv[j][3] = c2 ? bv[j][2] : v;                        bi[j][3] = c2 ? bi[j][2] : row;
              bv[j][2] = c1 ? bv[j][1] : (c2 ? v : bv[j][2]);      bi[j][2] = c1 ? bi[j][1] : (c2 ? row : bi[j][2]);
              bv[j][1] = c0 ? bv[j][0] : (c1 ? v : bv[j][1]);      bi[j][1] = c0 ? bi[j][0] : (c1 ? row : bi[j][1]);
              bv[j][0] = c0 ? v : bv[j][0];                        bi[j][0] = c0 ? row : bi[j][0];
            }
            acc[i][j][r] = 0.f;
          }
    }
  }
  // candidates: [token][ (wm*2 + half) * VQ_TOP + t ]
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = col0 + wn * 64 + j * 32 + lc;
    if (col >= g.N) continue;
    const long p = (long)col * (4 * VQ_TOP) + (wm * 2 + half) * VQ_TOP;
#pragma unroll
    for (int t = 0; t < VQ_TOP; ++t) { g.part_val[p + t] = bv[j][t]; g.part_idx[p + t] = bi[j][t]; }
  }
}

}  // namespace g3

// called by ctclip_gemm_bf16 (gemm.hip): k-major x k-major, K % 32 == 0, plain (non-accumulating) output
int ctclip_gemm3_launch(const void* A, const void* B, void* C, const float* bias, const float* resid, int M, int N, int K,
                        long lda, long ldb, long ldc, long ldr, int c_fp32, float alpha, int act, void* G, long ldg,
                        hipStream_t st) {
  using namespace g3;
  if ((act & 0xff) == 2 && (c_fp32 || bias || resid || !G || (N & 127) || (ldc & 7) || (ldg & 7) || (((uintptr_t)C) & 15) ||
                            (((uintptr_t)G) & 15)))
    return (int)hipErrorInvalidValue;
  if ((act & 0xff) == 3 && (c_fp32 || bias || resid || !G || (N & 63) || (ldg & 7) || (((uintptr_t)G) & 15)))
    return (int)hipErrorInvalidValue;
  // variant: CTCLIP_GEMM3_BN = 256 (one workgroup per CU, 4 stages; default) | 128 (two workgroups per CU, 3 stages);
  // CTCLIP_GEMM3_M16 = 1 selects the 16x16x32 MFMA form; CTCLIP_GEMM3_STAGGER = start-up delay in 10 ns ticks (BN 128).
  static const int variant = [] {
    const char* e = getenv("CTCLIP_GEMM3_BN");
    return (e && atoi(e) == 128) ? 128 : 256;
  }();
  static const bool m16 = [] { const char* e = getenv("CTCLIP_GEMM3_M16"); return !e || atoi(e) != 0; }();   // default on: +2..6 %
  static const int stagger = [] { const char* e = getenv("CTCLIP_GEMM3_STAGGER"); return e ? atoi(e) : 0; }();
  static const int stagger_shift = [] { const char* e = getenv("CTCLIP_GEMM3_STAGGER_SHIFT"); return e ? atoi(e) : 8; }();
  static const bool roles = [] { const char* e = getenv("CTCLIP_GEMM3_ROLES"); return !e || atoi(e) != 0; }();
  const int bn = variant;
  Args g{};
  g.A = (const bf16_t*)A; g.B = (const bf16_t*)B; g.C = C; g.bias = bias; g.resid = resid;
  g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.ldr = ldr; g.M = M; g.N = N; g.K = K;
  g.tiles_m = (M + BM - 1) / BM; g.tiles_n = (N + bn - 1) / bn;
  g.c_fp32 = c_fp32; g.act = act; g.alpha = alpha; g.G = (bf16_t*)G; g.ldg = ldg;
  g.stagger = (bn == 128) ? stagger * (K / 32) / 16 : 0;   // quoted for K = 512, scaled with the length of the matrix loop
  g.stagger_shift = stagger_shift; g.stagger_limit = 512;
#define G3_LAUNCH(BN_, NS_, NT_, M16_, ROLES_, THREADS_, LDS_)                                                            \
  do {                                                                                                                    \
    static bool attr_set = false;                                                                                         \
    if (!attr_set) {                                                                                                      \
      hipError_t e = hipFuncSetAttribute((const void*)gemm3_kernel<BN_, NS_, NT_, M16_, ROLES_>,                          \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)(LDS_));                        \
      if (e != hipSuccess) return (int)e;                                                                                 \
      attr_set = true;                                                                                                    \
    }                                                                                                                     \
    hipLaunchKernelGGL((gemm3_kernel<BN_, NS_, NT_, M16_, ROLES_>), dim3(g.tiles_m * g.tiles_n), dim3(THREADS_), (LDS_),  \
                       st, g);                                                                                            \
  } while (0)
  if (bn == 256) {
    const size_t lds = (size_t)4 * (SUB + 256 * BK * 2);   // 128 KiB
    if (m16 && roles) G3_LAUNCH(256, 4, true, true, true, 512, lds);
    else if (m16) G3_LAUNCH(256, 4, true, true, false, 512, lds);
    else G3_LAUNCH(256, 4, true, false, false, 512, lds);
  } else {
    const size_t lds = (size_t)3 * (SUB + 128 * BK * 2);   // 72 KiB: two workgroups per CU
    if (m16) G3_LAUNCH(128, 3, true, true, false, 256, lds);
    else G3_LAUNCH(128, 3, true, false, false, 256, lds);
  }
#undef G3_LAUNCH
  return (int)hipGetLastError();
}

#ifdef CTCLIP_G3_STAMPS
extern "C" int ctclip_debug_gemm3_occupancy(int bn, int m16) {
  using namespace g3;
  int n = -1;
  hipError_t e;
  if (bn == 128) {
    const size_t lds = (size_t)3 * (SUB + 128 * BK * 2);
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, gemm3_kernel<128, 3, true, true, false>, 256, lds);
  } else {
    const size_t lds = (size_t)4 * (SUB + 256 * BK * 2);
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, gemm3_kernel<256, 4, true, true, true>, 512, lds);
  }
  return e == hipSuccess ? n : -(int)e;
}

extern "C" int ctclip_debug_gemm3_stamps(void* buf, long capacity_blocks) {
  unsigned long long* p = (unsigned long long*)buf;
  hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(g3::g_stamps), &p, sizeof(p));
  if (e != hipSuccess) return (int)e;
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g3::g_stamp_cap), &capacity_blocks, sizeof(capacity_blocks));
}
#endif

// called by ctclip_vq_topk (gemm.hip): codes M % 256 == 0, K % 32 == 0; part_val / part_idx are [N][16]
int ctclip_vq_topk3_launch(const void* A, const void* B, float* part_val, int* part_idx, int M, int N, int K, long lda,
                           long ldb, hipStream_t st) {
  using namespace g3;
  VqArgs g{};
  g.A = (const bf16_t*)A; g.B = (const bf16_t*)B; g.lda = lda; g.ldb = ldb; g.M = M; g.N = N; g.K = K;
  g.tiles_m = M / BM; g.part_val = part_val; g.part_idx = part_idx;
  const size_t lds = (size_t)4 * 2 * SUB;          // 128 KiB
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)vq_topk3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  hipLaunchKernelGGL(vq_topk3_kernel, dim3((N + 255) / 256), dim3(512), lds, st, g);
  return (int)hipGetLastError();
}
