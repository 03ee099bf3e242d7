// bf16 MFMA GEMM, 256 x 256 x 32 tile for the weight gradients of the CT-CLIP step:
//   dW[M,N] += alpha * A[K,M]^T B[K,N]   (A = dy, B = x, both row-major over K = all tokens, i.e. m-major x n-major),
// split over K -- partial tiles into a workspace summed in split order (reproducible), or f32 atomics -- into a zeroed / running dW (torch.nn.Linear's backward, attention.py:55-70, 144-183).
//
// These products have a tiny output and a contraction of ~10^6, so they are pure main loop.  gemm2.hip ran them on its
// 256 x 128 x 64 tile (85 FLOP per operand byte, two 48 KiB stages in flight).  Here the tile of gemm3.hip is used with
// the operands stored the other way round:
//   * block tile 256 x 256 x 32, 512 threads = 8 waves (2 x 4), each wave 128 x 64 as 4x2 MFMA 32x32x16;
//   * a stage holds A[32 k][256 m] and B[32 k][256 n] (512-byte k-rows, 32 KiB), ring of four stages; global_load_lds
//     moves two whole k-rows per wave-instruction, the bank swizzle applied to the SOURCE column chunk;
//   * fragments come out of LDS with ds_read_b64_tr_b16 (the MFMA wants 8 consecutive k per lane, memory has 8
//     consecutive m): twelve transposed reads per 16-wide k-step and wave, issued as inline asm (see gemm2.hip);
//   * role-alternating main loop (the two waves of a SIMD half a K-step apart), as in gemm3.hip;
//   * epilogue: f32 atomics straight from the accumulators (each register covers two 128-byte row segments).
// Preconditions (checked by the dispatcher in gemm.hip): both operands m/n-major, K % 32 == 0, f32 accumulate output.
#include "gemm_tile_t.h"
#include <stdlib.h>

namespace g4 {

#ifdef CTCLIP_G4_STAMPS
// diagnostic build only (hipcc -DCTCLIP_G4_STAMPS; never compiled into the shipped library), read by tools/gemm4_timeline.py:
// per-workgroup phase stamps {hw id, xcc id, start, first K-step landed, matrix loop done, stores drained, stores issued} in
// 10 ns ticks, and per-segment shader-cycle sums of the matrix loop for waves 0 (wm = 0) and 4 (wm = 1)
__device__ unsigned long long* g_stamps = nullptr;
__device__ unsigned long long* g_prof = nullptr;   // [blocks][16]
__device__ long g_stamp_cap = 0;
#define G4_STAMP(slot)                                                                                            \
  do {                                                                                                            \
    if (g_stamps && threadIdx.x == 0 && (long)blockIdx.x < g_stamp_cap)                                           \
      g_stamps[(long)blockIdx.x * 8 + (slot)] = __builtin_amdgcn_s_memrealtime();                                 \
  } while (0)
#define G4_SEG_DECL() unsigned long long seg_[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long c_ = __builtin_amdgcn_s_memtime()
#define G4_SEG(n) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); seg_[n] += t_ - c_; c_ = t_; } while (0)
#define G4_SEG_STORE()                                                                                            \
  do {                                                                                                            \
    if (g_prof && (threadIdx.x == 0 || threadIdx.x == 256) && (long)blockIdx.x < g_stamp_cap)                    \
      for (int n_ = 0; n_ < 8; ++n_) g_prof[(long)blockIdx.x * 16 + (threadIdx.x >> 8) * 8 + n_] = seg_[n_];     \
  } while (0)
#else
#define G4_STAMP(slot) do { } while (0)
#define G4_SEG_DECL() do { } while (0)
#define G4_SEG(n) do { } while (0)
#define G4_SEG_STORE() do { } while (0)
#endif

struct Args {
  const bf16_t* A; const bf16_t* B; float* C;
  long lda, ldb, ldc;
  int M, N, K, tiles_m, tiles_n, split_k, ktiles_per_split;
  float alpha;
  float* part;       // split-K without atomics: split ks stores its tile into part[ks][M][N] (plain stores); null = atomics
};

template <bool MF16>
__global__ __launch_bounds__(NT, 2) void gemm4_kernel(Args g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;

  int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tn = bid % g.tiles_n; bid /= g.tiles_n;
  const int tm = bid % g.tiles_m;
  const int ks = bid / g.tiles_m;
  const int row0 = tm * BM, col0 = tn * BN;
  const int nk_total = g.K / BK;
  const int kt_begin = ks * g.ktiles_per_split;
  const int nk = min(nk_total, kt_begin + g.ktiles_per_split) - kt_begin;
#ifdef CTCLIP_G4_STAMPS
  if (g_stamps && threadIdx.x == 0 && (long)blockIdx.x < g_stamp_cap) {
    g_stamps[(long)blockIdx.x * 8 + 0] = __builtin_amdgcn_s_getreg((31 << 11) | 4);     // HW_REG_HW_ID
    g_stamps[(long)blockIdx.x * 8 + 1] = __builtin_amdgcn_s_getreg((31 << 11) | 20);    // HW_REG_XCC_ID
  }
#endif
  G4_STAMP(2);

  // piece q = wave * PPW + j of a stage: the first 16 are the A tile, the rest the B tile (stored right behind it)
  const bf16_t* src[PPW];
  long kstep[PPW];
  uint32_t dst[PPW];
#pragma unroll
  for (int j = 0; j < PPW; ++j) {
    const int q = wave * PPW + j;
    if (q < 16) {
      src[j] = g.A + piece_src(q, lane, row0, g.M, g.lda) + (long)kt_begin * BK * g.lda;
      kstep[j] = (long)BK * g.lda;
    } else {
      src[j] = g.B + piece_src(q - 16, lane, col0, g.N, g.ldb) + (long)kt_begin * BK * g.ldb;
      kstep[j] = (long)BK * g.ldb;
    }
    dst[j] = (uint32_t)(q * 1024);
  }
  auto issue_part = [&](int t, int j0) {           // half of this wave's pieces of K-step t -> stage t % NS
    const uint32_t sb = lds0 + (uint32_t)((t % NS) * STAGE);
#pragma unroll
    for (int j = j0; j < j0 + PPW / 2; ++j) G4_GLDS(src[j] + (long)t * kstep[j], sb + dst[j]);
  };

  // the same 128 accumulator registers either way: 4 x 2 tiles of 32 x 32 or 8 x 4 tiles of 16 x 16
  f32x16 acc[4][2];
  f32x4 acc16[MF16 ? 8 : 1][MF16 ? 4 : 1];
  if (MF16) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc16[MF16 ? i : 0][MF16 ? j : 0][r] = 0.f;
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  }

#pragma unroll
  for (int t = 0; t < NS - 1; ++t)
    if (t < nk) { issue_part(t, 0); issue_part(t, PPW / 2); }

  // MF16: set s holds A tiles 4 s .. 4 s + 3 and B tiles 2 s, 2 s + 1 (16 wide each); otherwise k-half s of every tile
  auto read_set = [&](short4v (&raw)[12], uint32_t sa_l, int s) {
    if (MF16) {
#pragma unroll
      for (int i = 0; i < 4; ++i) read_frag_tr16(sa_l, wm * 128 + (4 * s + i) * 16, lane, raw[2 * i], raw[2 * i + 1]);
#pragma unroll
      for (int j = 0; j < 2; ++j) read_frag_tr16(sa_l + SUB, wn * 64 + (2 * s + j) * 16, lane, raw[8 + 2 * j], raw[9 + 2 * j]);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) read_frag_tr(sa_l, wm * 128 + i * 32, s, lane, raw[2 * i], raw[2 * i + 1]);
#pragma unroll
      for (int j = 0; j < 2; ++j) read_frag_tr(sa_l + SUB, wn * 64 + j * 32, s, lane, raw[8 + 2 * j], raw[9 + 2 * j]);
    }
  };
  auto mma16_all = [&](short4v (&ra)[12], short4v (&rb)[12]) {
    bf16x8 fa[8], fb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { fa[i] = join_tr(ra[2 * i], ra[2 * i + 1]); fa[4 + i] = join_tr(rb[2 * i], rb[2 * i + 1]); }
#pragma unroll
    for (int j = 0; j < 2; ++j) { fb[j] = join_tr(ra[8 + 2 * j], ra[9 + 2 * j]); fb[2 + j] = join_tr(rb[8 + 2 * j], rb[9 + 2 * j]); }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc16[MF16 ? i : 0][MF16 ? j : 0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc16[MF16 ? i : 0][MF16 ? j : 0], 0, 0, 0);
  };
  auto mma_set = [&](short4v (&raw)[12]) {
    bf16x8 fa[4], fb[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[i] = join_tr(raw[2 * i], raw[2 * i + 1]);
#pragma unroll
    for (int j = 0; j < 2; ++j) fb[j] = join_tr(raw[8 + 2 * j], raw[9 + 2 * j]);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = mfma32(fa[i], fb[j], acc[i][j]);
  };

  // Role-alternating loop, as in gemm3.hip: the two waves of a SIMD (w and w + 4, i.e. wm = 0 / 1) run half a K-step apart,
  // one in its MFMA block (16 MFMA 32x32x16) while the other reads its 24 transposed fragments and issues its four LDS-DMA
  // pieces of K-step k + NS - 1; stage k is read in intervals 2k (wm 0) and 2k+1 (wm 1), refilled from interval 2k+2 on,
  // and each wave waits for its own pieces of stage k+1 (counted vmcnt) inside interval 2k+1.  Against the lock-step loop
  // (one barrier per K-step, fragment reads of the second half under the MFMAs of the first): ff1 wgrad 912 -> 1039
  // TFLOP/s, ff2 847 -> 980, kv 842 -> 958, patch 927 -> 1090 (same box, interleaved).
  short4v r0[12], r1[12];
  auto wait_next = [&](int k) {
    if (k + 1 < nk) {
      const int y = nk - 2 - k;
      if (y >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if (y == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  };
#define G4_BAR()                                                                       \
  do {                                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                 \
    __builtin_amdgcn_s_barrier();                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                 \
  } while (0)
  auto load_block = [&](int k) {
    const uint32_t sa_l = lds0 + (uint32_t)((k % NS) * STAGE);
    read_set(r0, sa_l, 0);
    read_set(r1, sa_l, 1);
    if (k + NS - 1 < nk) { issue_part(k + NS - 1, 0); issue_part(k + NS - 1, PPW / 2); }
  };
  auto mfma_block = [&]() {
    __builtin_amdgcn_s_setprio(1);
    if (MF16) {
      mma16_all(r0, r1);
    } else {
      mma_set(r0);
      mma_set(r1);
    }
    __builtin_amdgcn_s_setprio(0);
  };
  {                                                 // stage 0 has landed for everybody
    const int y0 = nk - 1;
    if (y0 >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (y0 == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  G4_BAR();
  G4_STAMP(3);
  G4_SEG_DECL();
  // profiled segments: 0 load block issue (24 transposed reads + 4 DMA pieces), 1 lgkmcnt wait, 2 barrier after the load block,
  // 3 MFMA block, 4 vmcnt wait, 5 barrier after the MFMA block
  if (wm == 0) {
    for (int k = 0; k < nk; ++k) {
      load_block(k);
      G4_SEG(0);
      tr_wait(r0);
      tr_wait(r1);
      G4_SEG(1);
      G4_BAR();
      G4_SEG(2);
      mfma_block();
      G4_SEG(3);
      wait_next(k);
      G4_SEG(4);
      G4_BAR();
      G4_SEG(5);
    }
    G4_BAR();                                       // the other half's last MFMA block
  } else {
    G4_BAR();                                       // interval 0: the other half reads stage 0
    G4_SEG(6);
    for (int k = 0; k < nk; ++k) {
      load_block(k);
      G4_SEG(0);
      wait_next(k);
      G4_SEG(4);
      tr_wait(r0);
      tr_wait(r1);
      G4_SEG(1);
      G4_BAR();
      G4_SEG(2);
      mfma_block();
      G4_SEG(3);
      G4_BAR();
      G4_SEG(5);
    }
  }
#undef G4_BAR
  G4_SEG_STORE();
  G4_STAMP(4);


  // split-K: f32 atomics straight from the accumulators.  For a fixed register the 64 lanes cover two 128-byte row
  // segments -- the access shape global_atomic_add_f32 runs at full rate with.
  if (MF16) {
    // 16 x 16 tiles: a register covers four 64-byte row segments
    const int q4 = lane >> 4, ml = lane & 15;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int col = col0 + wn * 64 + j * 16 + ml;
        if (col >= g.N) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = row0 + wm * 128 + i * 16 + 4 * q4 + r;
          if (row >= g.M) continue;
          const float v = acc16[MF16 ? i : 0][MF16 ? j : 0][r] * g.alpha;
          if (g.part) g.part[((long)ks * g.M + row) * g.N + col] = v;
          else atomicAdd(g.C + (long)row * g.ldc + col, v);
        }
      }
    G4_STAMP(6);                                      // every store of wave 0 issued
#ifdef CTCLIP_G4_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // ... and drained
#endif
    G4_STAMP(5);
    return;
  }
  const int half = lane >> 5, lc = lane & 31;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = col0 + wn * 64 + j * 32 + lc;
      if (col >= g.N) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = row0 + wm * 128 + i * 32 + acc_row(r, half);
        if (row >= g.M) continue;
        const float v = acc[i][j][r] * g.alpha;
        if (g.part) g.part[((long)ks * g.M + row) * g.N + col] = v;
        else atomicAdd(g.C + (long)row * g.ldc + col, v);
      }
    }
  G4_STAMP(6);
#ifdef CTCLIP_G4_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  G4_STAMP(5);
}

}  // namespace g4

#ifdef CTCLIP_G4_STAMPS
extern "C" int ctclip_debug_gemm4_prof(void* buf) {
  unsigned long long* p = (unsigned long long*)buf;
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g4::g_prof), &p, sizeof(p));
}
extern "C" int ctclip_debug_gemm4_stamps(void* buf, long capacity_blocks) {
  unsigned long long* p = (unsigned long long*)buf;
  hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(g4::g_stamps), &p, sizeof(p));
  if (e != hipSuccess) return (int)e;
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g4::g_stamp_cap), &capacity_blocks, sizeof(capacity_blocks));
}
#endif

// called by ctclip_gemm_bf16 (gemm.hip): m-major x n-major, K % 32 == 0, f32 accumulate output
int ctclip_gemm4_launch(const void* A, const void* B, void* C, int M, int N, int K, long lda, long ldb, long ldc, int split_k,
                        float alpha, float* part, hipStream_t st) {
  using namespace g4;
  Args g{};
  g.A = (const bf16_t*)A; g.B = (const bf16_t*)B; g.C = (float*)C;
  g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.M = M; g.N = N; g.K = K;
  g.tiles_m = (M + BM - 1) / BM; g.tiles_n = (N + BN - 1) / BN;
  const int nk = K / BK;
  if (split_k < 1) split_k = 1;
  if (split_k > nk) split_k = nk;
  g.ktiles_per_split = (nk + split_k - 1) / split_k;
  g.split_k = (nk + g.ktiles_per_split - 1) / g.ktiles_per_split;
  g.alpha = alpha;
  g.part = g.split_k > 1 ? part : nullptr;
  const size_t lds = (size_t)NS * STAGE;           // 128 KiB
  CTCLIP_LDS_LIMIT_ONCE(gemm4_kernel<true>, lds);
  CTCLIP_LDS_LIMIT_ONCE(gemm4_kernel<false>, lds);
  static const bool mf32 = CTCLIP_KNOB("CTCLIP_GEMM4_MFMA32") != nullptr;
  if (mf32) hipLaunchKernelGGL(gemm4_kernel<false>, dim3(g.tiles_m * g.tiles_n * g.split_k), dim3(NT), lds, st, g);
  else hipLaunchKernelGGL(gemm4_kernel<true>, dim3(g.tiles_m * g.tiles_n * g.split_k), dim3(NT), lds, st, g);
  return (int)hipGetLastError();
}
