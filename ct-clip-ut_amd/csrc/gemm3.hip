// bf16 MFMA GEMM, 256 x 256 x 32 (or 256 x 128 x 32) tile, for the k-major x k-major products of the CT-CLIP step (every
// forward and data-gradient projection: C[M,N] = A[M,K] B[N,K]^T with M = all tokens; K = 256 .. 4000, N = 256 .. 2816).
//
//   * 512 threads = 8 waves, each wave a 128 x 64 (or 64 x 64) slab of MFMA 16x16x32 products (the part holds a higher clock
//     on this shape than on 32x32x16 at the same cycles per flop: +2..6 % measured);
//   * a ring of 32 KiB (24 KiB) stages filled by global_load_lds (16 B per lane, bank swizzle on the SOURCE address),
//     counted `s_waitcnt vmcnt`, raw s_barrier: NS - 1 K-steps in flight ahead of the one being consumed;
//   * ROLE-ALTERNATING main loop: the two waves of a SIMD run half a K-step apart, one in its MFMA block while the other
//     reads fragments and issues DMA (see the kernel);
//   * the MFMAs are issued TRANSPOSED (N-fragment as the A operand) and the N-fragment rows are permuted, so a lane's
//     accumulators are contiguous output columns of one row: the epilogue converts and stores 16-byte vectors straight from
//     registers (bias / residual / GELU / both GEGLU forms fused) -- no LDS staging, no barrier.
// What bounds a short-K tile (measured with -DCTCLIP_G3_STAMPS, tools/gemm_timeline.py): a CU's store path moves ~12 bytes
// per clock (~24 GB/s; 256 of them are the chip's ~6 TB/s), so writing a 128 KiB bf16 tile takes >= 5.4 us and an f32 one
// twice that -- 8 workgroups alone on the chip or 19 000 -- next to ~13 us of matrix loop at K = 512.
// Preconditions (checked by the dispatcher in gemm.hip): both operands k-major, K % 32 == 0, no split-K / accumulate.
#include "gemm_tile.h"
#include <stdlib.h>

namespace g3 {

#ifdef CTCLIP_G3_STAMPS
// diagnostic build only (hipcc -DCTCLIP_G3_STAMPS; never compiled into the shipped library), read by tools/gemm_timeline.py:
// per-workgroup phase stamps {hw id, xcc id, start, first K-step landed, matrix loop done, stores drained, stores issued} in
// 10 ns ticks, and per-segment shader-cycle sums of the matrix loop for waves 0 and 4
__device__ unsigned long long* g_stamps = nullptr;
__device__ unsigned long long* g_prof = nullptr;   // [blocks][16]
__device__ long g_stamp_cap = 0;
#define G3_STAMP(slot)                                                                                            \
  do {                                                                                                            \
    if (g_stamps && threadIdx.x == 0 && (long)tile < g_stamp_cap)                                                 \
      g_stamps[(long)tile * 8 + (slot)] = __builtin_amdgcn_s_memrealtime();                                       \
  } while (0)
#define G3_SEG_DECL() unsigned long long seg_[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long c_ = __builtin_amdgcn_s_memtime()
#define G3_SEG(n) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); seg_[n] += t_ - c_; c_ = t_; } while (0)
#define G3_SEG_STORE()                                                                                            \
  do {                                                                                                            \
    if (g_prof && (threadIdx.x == 0 || threadIdx.x == 256) && (long)tile < g_stamp_cap)                          \
      for (int n_ = 0; n_ < 8; ++n_) g_prof[(long)tile * 16 + (threadIdx.x >> 8) * 8 + n_] = seg_[n_];           \
  } while (0)
#else
#define G3_STAMP(slot) do { } while (0)
#define G3_SEG_DECL() do { } while (0)
#define G3_SEG(n) do { } while (0)
#define G3_SEG_STORE() do { } while (0)
#endif

#ifndef G3_NS256
#define G3_NS256 4
#endif
// (tile layout, store helpers and the register epilogue: gemm_tile.h, shared with gemm5.hip)

// Main loop: the two waves of a SIMD (w and w + 4: role groups 0 and 1) run half a K-step apart.  In a plain loop all eight
// waves leave the K-step barrier together, all read their fragments from LDS together and the matrix pipe idles meanwhile
// (measured: ~630 of ~1680 cycles per K-step, wave 0 parked at the barrier for 540 of them).  Here every barrier interval
// has one group in its MFMA block and the other in its load block, then they swap:
//     interval   2k     2k+1    2k+2
//     group 0    R_k    M_k     R_k+1        R = read the fragments of K-step k (ds_read_b128) and issue this wave's share
//     group 1    M_k-1  R_k     M_k              of the LDS-DMA of K-step k + NS - 1;   M = the MFMAs of K-step k
// Stage k is read in intervals 2k (group 0) and 2k+1 (group 1), so it is refilled from interval 2k+2 on and must have landed
// before interval 2k: each wave waits for its own pieces of stage k+1 (counted vmcnt) inside interval 2k+1.
// Per K-step and wave at 256 x 256 (shader cycles, -DCTCLIP_G3_STAMPS): load block issue ~540 (12 ds_read_b128 + 4 DMA
// issues of ~85 each), MFMA block ~576, vmcnt wait ~150, barriers ~300: ~1600 against 1024 of matrix-pipe work; the plain
// loop took ~1680.  Variants measured and dropped: the DMA issues between the MFMAs (they stall the one wave that is
// feeding the matrix pipe: 1141 vs 1261 TFLOP/s at 4096^3); ONE barrier per K-step with group 0 running [R, M] and group 1
// [M, R] (the older wave wins the matrix pipe, group 1's MFMA block stretches over both halves: ~1780 cycles per K-step).
//
// TWO SHAPES of the same kernel:
//   TBN = 256: 256 x 256 tile, waves 2 (M) x 4 (N), 128 x 64 per wave (8 x 4 MFMAs), four 32 KiB stages, ONE workgroup per
//              CU (<= 256 registers).  Least operand traffic per flop.
//   TBN = 128: 256 x 128 tile, waves 4 (M) x 2 (N), 64 x 64 per wave (4 x 4 MFMAs), three 24 KiB stages, TWO workgroups per
//              CU (<= 128 registers, 72 KiB of LDS each), so that one computes while the other writes back and refills.
//              Measured slower on every shape but the f32-output K = 256 one (out-projection: 423 vs 382 TFLOP/s): with
//              64 x 64 per wave a load block (8 reads + 3 DMA issues) is twice as long as an MFMA block (16 MFMAs) and two
//              loops running side by side reach ~50 % of the matrix pipe where one 256 x 256 loop reaches ~64 %.
// EPI selects the epilogue at compile time (one kernel per form keeps the two-workgroup shape inside 128 registers):
//   0 plain -> bf16   1 plain -> f32   (both: alpha, bias, residual, erf-GELU when g.act == 1)   2 FF1 + GEGLU   3 FF2 dgrad + GEGLU backward
template <int TBN, int EPI>
__global__ __launch_bounds__(512, (TBN == 256 ? 2 : 4)) void gemm3_kernel(Args g) {
  constexpr bool F32OUT = EPI == 1 || EPI == 4;     // identity N-fragment rows: a lane holds 4 contiguous f32 columns
  constexpr int WN = TBN / 64, WM = 8 / WN;            // wave grid: 2 x 4 or 4 x 2
  constexpr int IM = BM / WM / 16, JN = 4;             // MFMA tiles per wave: 8 x 4 or 4 x 4 (the slab is always 64 columns)
  constexpr int TNS = (TBN == 256) ? G3_NS256 : 3;     // ring stages
  constexpr int TSTAGE = SUB + TBN * BK * 2;           // 32 / 24 KiB
  constexpr int TPPW = TSTAGE / 1024 / 8;              // DMA pieces per wave and stage: 4 / 3
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int grp = wave >> 2;                           // role group: SIMD partners are waves w and w + 4
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;

  // PERSISTENT over tiles: workgroup b takes the tiles b, b + gridDim.x, ... (gridDim.x = resident workgroups, a multiple
  // of 8, so a workgroup's tiles keep the XCD-contiguous order of xcd_remap).  When a tile's matrix loop is done the whole
  // ring is free, so the first NS - 1 K-steps of the NEXT tile are requested before the epilogue's stores are issued: the
  // ring fill (1.7 us per tile) and the launch gap hide under the ~5 us of write-back.
  const int total = g.tiles_m * g.tiles_n;
  int tile = blockIdx.x;
  if (tile >= total) return;
  const int nk = g.K / BK;
  int row0, col0;
  // piece q = wave * TPPW + j of a stage: the first 16 are the A tile, the rest the B tile (stored right behind it)
  const bf16_t* src[TPPW];
  uint32_t dst[TPPW];
#pragma unroll
  for (int j = 0; j < TPPW; ++j) dst[j] = (uint32_t)((wave * TPPW + j) * 1024);
  auto locate = [&](int t) {
    const int bid = xcd_remap(t, total);
    row0 = (bid / g.tiles_n) * BM;
    col0 = (bid % g.tiles_n) * TBN;
#pragma unroll
    for (int j = 0; j < TPPW; ++j) {
      const int q = wave * TPPW + j;
      src[j] = (q < 16) ? g.A + piece_src(q, lane, row0, g.M, g.lda) : g.B + piece_src(q - 16, lane, col0, g.N, g.ldb);
    }
  };
  auto issue_step = [&](int t) {                   // this wave's pieces of K-step t -> stage t % TNS
    const uint32_t sb = lds0 + (uint32_t)((t % TNS) * TSTAGE);
#pragma unroll
    for (int j = 0; j < TPPW; ++j) G3_GLDS(src[j] + (long)t * BK, sb + dst[j]);
  };
  auto issue_prologue = [&]() {
#pragma unroll
    for (int t = 0; t < TNS - 1; ++t)
      if (t < nk) issue_step(t);
  };

#ifdef CTCLIP_G3_STAMPS
  if (g_stamps && threadIdx.x == 0 && (long)tile < g_stamp_cap) {
    g_stamps[(long)tile * 8 + 0] = __builtin_amdgcn_s_getreg((31 << 11) | 4);     // HW_REG_HW_ID
    g_stamps[(long)tile * 8 + 1] = __builtin_amdgcn_s_getreg((31 << 11) | 20);    // HW_REG_XCC_ID
  }
#endif
  G3_STAMP(2);
  locate(tile);
  issue_prologue();
  bool first = true, counted = false;
  // store instructions per wave of a full tile's register epilogue (EPI 5: at least 2 per row group, 3 in a normalised slab --
  // the smaller count only makes the wait behind the epilogue stricter than it has to be)
  constexpr int NST = IM * ((EPI == 0 || EPI == 5) ? 2 : EPI == 2 ? 3 : 4);

  // fragment addresses inside a stage: M rows 16 i + ml of this wave's rows, (permuted) N rows of its 64
  const int ml = lane & 15, q4 = lane >> 4;
  uint32_t offA, offB[JN];
  offA = tile_off(wm * (IM * 16) + ml, q4);        // + i * 1024: sixteen rows further, same swizzle key
#pragma unroll
  for (int j = 0; j < JN; ++j) offB[j] = SUB + tile_off(wn * 64 + (F32OUT ? 16 * j + ml : nfrag_row(j, ml)), q4);

  bf16x8 fa[IM], fb[JN];
  f32x4 acc[IM][JN];
  // own pieces of K-step k+1 landed; up to TNS - 2 younger K-steps (TPPW DMAs each) stay in flight
  auto wait_next = [&](int k) {
    if (k + 1 < nk) {
      const int y = nk - 2 - k;
      if (TNS >= 5 && y >= 3) wait_vm<3 * TPPW>();
      else if (TNS >= 4 && y >= 2) wait_vm<2 * TPPW>();
      else if (y >= 1) wait_vm<TPPW>();
      else wait_vm<0>();
    }
  };
#define G3_BAR()                                                                       \
  do {                                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                 \
    __builtin_amdgcn_s_barrier();                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                 \
  } while (0)
  auto load_block = [&](int k) {
    const char* st = smem + (k % TNS) * TSTAGE;
#pragma unroll
    for (int j = 0; j < JN; ++j) fb[j] = *(const bf16x8*)(st + offB[j]);
#pragma unroll
    for (int i = 0; i < IM; ++i) fa[i] = *(const bf16x8*)(st + offA + i * 1024);
    if (k + TNS - 1 < nk) issue_step(k + TNS - 1);
  };
  auto mfma_block = [&]() {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < IM; ++i)
#pragma unroll
      for (int j = 0; j < JN; ++j) acc[i][j] = mfma16(fb[j], fa[i], acc[i][j]);      // transposed: see nfrag_row()
    __builtin_amdgcn_s_setprio(0);
  };
  for (;;) {
#pragma unroll
  for (int i = 0; i < IM; ++i)
#pragma unroll
    for (int j = 0; j < JN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
  if (first) {                                      // stage 0 has landed for everybody
    const int y0 = nk - 1;
    if (TNS >= 5 && y0 >= 3) wait_vm<3 * TPPW>();
    else if (TNS >= 4 && y0 >= 2) wait_vm<2 * TPPW>();
    else if (y0 >= 1) wait_vm<TPPW>();
    else wait_vm<0>();
  } else if (counted) {
    // the previous tile's NST epilogue stores are younger than this tile's first stages (vmcnt retires in order): leave
    // them and K-steps 1, 2 in flight.  The in-loop waits of K-steps 0 and 1 then over-wait (they count as if only DMAs
    // were outstanding), which gives the stores two K-steps to drain under the matrix loop instead of blocking here.
    wait_vm<NST + (TNS - 2) * TPPW>();
  } else {
    wait_vm<0>();                                   // ragged previous tile: its store count is not a constant
  }
  const bool stores_pending = !first && counted;
  first = false;
  G3_BAR();
  G3_STAMP(3);
  G3_SEG_DECL();
  // profiled segments: 0 load block issue, 1 lgkmcnt wait, 2 barrier after the load block, 3 MFMA block, 4 vmcnt wait,
  // 5 barrier after the MFMA block
  if (grp == 0) {
    for (int k = 0; k < nk; ++k) {
      load_block(k);
      G3_SEG(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // reads retired before the barrier that frees the stage
      G3_SEG(1);
      G3_BAR();
      G3_SEG(2);
      mfma_block();
      G3_SEG(3);
      wait_next(k);
      G3_SEG(4);
      G3_BAR();
      G3_SEG(5);
    }
    G3_BAR();                                       // the other group's last MFMA block
  } else {
    G3_BAR();                                       // interval 0: the other group reads stage 0
    G3_SEG(6);
    for (int k = 0; k < nk; ++k) {
      load_block(k);
      G3_SEG(0);
      wait_next(k);
      G3_SEG(4);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      G3_SEG(1);
      G3_BAR();
      G3_SEG(2);
      mfma_block();
      G3_SEG(3);
      G3_BAR();
      G3_SEG(5);
    }
  }
  G3_SEG_STORE();
  G3_STAMP(4);

  // every read of the ring was retired before the last barrier: request the next tile's first stages now
  const int erow0 = row0, ecol0 = col0;
  const int next = tile + (int)gridDim.x;
  if (next < total) {
    locate(next);
    issue_prologue();
  }
  counted = g.direct && erow0 + BM <= g.M && ecol0 + TBN <= g.N && nk >= TNS - 1;
  // this wave's 64-column slab; the two-workgroup shape has 128 registers: no second buffer for the epilogue's reads
  epilogue_slab<EPI, IM, JN, 0, TBN == 256>(g, acc, erow0 + wm * (IM * 16), ecol0 + wn * 64, lane);
  G3_STAMP(6);                                        // every store of wave 0 issued
#ifdef CTCLIP_G3_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // ... and drained
#endif
  G3_STAMP(5);
  if (next >= total) break;
  tile = next;
#ifdef CTCLIP_G3_STAMPS
  if (g_stamps && threadIdx.x == 0 && (long)tile < g_stamp_cap) {
    g_stamps[(long)tile * 8 + 0] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    g_stamps[(long)tile * 8 + 1] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
  }
#endif
  G3_STAMP(2);
  }
#undef G3_BAR
}


// ------------------------------------------------------------------------------------------------------------------
// VQ nearest-code search on the same tile and ring (vector_quantize_pytorch cosine-sim codebook, ctvit.py:420-427):
// scores[code][token] = sum_k A[code][k] B[token][k] are never written.  One workgroup owns 256 tokens and sweeps ALL
// 256-code tiles; the (code tile, K-step) sequence is flattened so the ring never drains between tiles.  Every lane
// keeps a running top-4 (value, code) for each of its two token columns over the 64 codes per tile its accumulator
// registers cover; a token ends with 4 lane groups (wm, lane half) x 4 = 16 candidates from disjoint parts of the
// codebook, which ctclip_vq_select re-ranks exactly in f32.  Against the 128 x 128 register-staged kernel of gemm.hip
// this halves the operand bytes pulled through L2 per flop (the codebook is re-streamed once per 256 tokens, not 128).
//
// CODE GROUPS (cgroups > 1; ctclip_vq_topk_grouped): an experiment on WHERE this kernel's fabric traffic comes from (27 GB per
// launch at 64 pairs against 1.4 GB of operands).  A workgroup re-reads its own 256 KiB token tile for every code tile; with
// one workgroup per (token tile, whole codebook) the 32 workgroups resident on an XCD keep 8 MiB of token tiles + the 8 MiB
// codebook moving through a 4 MiB L2.  With code groups the 32 workgroups an XCD holds at a time are `ttiles` token tiles x
// `cgroups` code groups (8 x 4): 2 MiB of token tiles, each shared by four workgroups, and every code tile shared by the
// eight token tiles that sweep it side by side -- 10 MiB per XCD round if the L2 kept them.  MEASURED
// (profiles/r03_vq_code_groups.txt; tools/probe_xcd.py confirms XCD = block % 8 and in-order rounds of 32): FETCH_SIZE per
// launch 33.4 GB with one group, 23.4 / 23.5 / 29.2 GB with 2 / 4 / 8 -- the workgroups of an XCD find only part of each
// other's lines in its L2 however the work is cut (54 MiB per XCD round with 4 groups) -- and the sweep gets slower (8.5 ->
// 9.0 ms at 64 pairs with 4 groups, 9.6 with 8: four / eight times the workgroups, each paying the ring fill and the candidate
// write).  The traffic is served by the 256 MiB Infinity Cache and is not what bounds the kernel (matrix pipe + LDS issue,
// section 4.3 of DESIGN.md).  One group is the default; the grouped form stays as a tested entry point.
// A token ends with 16 candidates PER GROUP, part_val / part_idx are [N][16 * cgroups]; ctclip_vq_select takes any candidate
// count, and the union contains the top-4 of every (wave row, lane half) part of the whole codebook, i.e. everything the
// one-group form keeps.
// Block b runs on XCD b % 8; its XCD-local index b / 8 is (group of 8 x ttiles token tiles, slot, code group).
// Preconditions (gemm.hip): M % 256 == 0, K % 32 == 0, (M / 256) % cgroups == 0.
// ------------------------------------------------------------------------------------------------------------------
struct VqArgs {
  const bf16_t* A; const bf16_t* B;
  long lda, ldb;
  int M, N, K, tiles_m;                  // tiles_m: code tiles per workgroup (= per code group)
  int cgroups, ttiles, ntok_tiles;
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
  int tok_tile, cgrp = 0;
  if (g.cgroups > 1) {
    const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3, per = g.ttiles * g.cgroups;
    const int grp = loc / per, within = loc - grp * per;
    cgrp = within % g.cgroups;
    tok_tile = (grp * 8 + xcd) * g.ttiles + within / g.cgroups;
    if (tok_tile >= g.ntok_tiles) return;                      // whole workgroup, before any barrier
  } else {
    tok_tile = xcd_remap(blockIdx.x, gridDim.x);
  }
  const int col0 = tok_tile * BN;                              // token tile
  const int code0 = cgrp * g.tiles_m * BM;                     // first code of this workgroup's group
  const int nk = g.K / BK;
  const int steps = g.tiles_m * nk;                            // flattened (code tile, K-step) sequence
  const int half = lane >> 5, lc = lane & 31;

  const bf16_t* src[PPW];
  long tstep[PPW];
  uint32_t dst[PPW];
#pragma unroll
  for (int j = 0; j < PPW; ++j) {
    const int q = wave * PPW + j;
    if (q < 16) { src[j] = g.A + piece_src(q, lane, code0, g.M, g.lda); tstep[j] = (long)BM * g.lda; }
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
      const int row_t = code0 + tile * BM + wm * 128;
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
  // candidates: [token][code group][ (wm*2 + half) * VQ_TOP + t ]
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = col0 + wn * 64 + j * 32 + lc;
    if (col >= g.N) continue;
    const long p = ((long)col * g.cgroups + cgrp) * (4 * VQ_TOP) + (wm * 2 + half) * VQ_TOP;
#pragma unroll
    for (int t = 0; t < VQ_TOP; ++t) { g.part_val[p + t] = bv[j][t]; g.part_idx[p + t] = bi[j][t]; }
  }
}

}  // namespace g3

namespace {
// shape: 256 x 256 (one workgroup per CU) everywhere but the f32-output products with a very short matrix loop (the
// out-projection, K = 256: write-back-bound), where two 256 x 128 workgroups per CU overlap one's stores with the other's
// loop.  CTCLIP_GEMM3_BN=128|256 forces one shape (experiments).
int g3_launch(g3::Args& g, int epi, int c_fp32, hipStream_t st) {
  using namespace g3;
  const int M = g.M, N = g.N, K = g.K;
  static const int forced = [] { const char* e = CTCLIP_KNOB("CTCLIP_GEMM3_BN"); return e ? atoi(e) : 0; }();
  const int bn = epi == 5 ? 256 : (forced == 128 || forced == 256) ? forced : ((c_fp32 && K <= 256) ? 128 : 256);
  g.tiles_m = (M + BM - 1) / BM; g.tiles_n = (N + bn - 1) / bn;
#define G3_LAUNCH(BN_, EPI_, LDS_)                                                                                        \
  do {                                                                                                                    \
    CTCLIP_LDS_LIMIT_ONCE((gemm3_kernel<BN_, EPI_>), (LDS_));                                                             \
    hipLaunchKernelGGL((gemm3_kernel<BN_, EPI_>), dim3(grid), dim3(512), (LDS_), st, g);                                  \
  } while (0)
#define G3_SHAPES(EPI_)                                                                                                   \
  do {                                                                                                                    \
    if (bn == 256) G3_LAUNCH(256, EPI_, (size_t)G3_NS256 * (SUB + 256 * BK * 2)); /* 128 KiB: one workgroup per CU */           \
    else G3_LAUNCH(128, EPI_, (size_t)3 * (SUB + 128 * BK * 2));           /* 72 KiB: two workgroups per CU */           \
  } while (0)
  const int cus = ctclip_cu_count8();                  // a multiple of 8 keeps a workgroup's tiles on one XCD's run
  const long total = (long)g.tiles_m * g.tiles_n;
  const long resident = (long)cus * (bn == 256 ? 1 : 2);
  static const bool persist = [] { const char* e = CTCLIP_KNOB("CTCLIP_GEMM3_PERSIST"); return !e || atoi(e) != 0; }();   // 0: one tile per workgroup (A/B)
  const int grid = (int)((total < resident || !persist) ? total : resident);
  if (epi == 0) G3_SHAPES(0);
  else if (epi == 1) G3_SHAPES(1);
  else if (epi == 2) G3_SHAPES(2);
  else if (epi == 3) G3_SHAPES(3);
  else if (epi == 5) G3_LAUNCH(256, 5, (size_t)G3_NS256 * (SUB + 256 * BK * 2));   // bf16 output: always the 256 x 256 shape
  else G3_SHAPES(4);
#undef G3_SHAPES
#undef G3_LAUNCH
  return (int)hipGetLastError();
}
}  // namespace

// called by ctclip_gemm_bf16 (gemm.hip): k-major x k-major, K % 32 == 0, plain (non-accumulating) output
int ctclip_gemm3_launch_hm(const void* A, const void* B, void* C, const float* bias, const float* resid, int M, int N, int K,
                           long lda, long ldb, long ldc, long ldr, int c_fp32, float alpha, int act, void* G, long ldg,
                           int hm_n, int hm_heads, hipStream_t st);

int ctclip_gemm3_launch(const void* A, const void* B, void* C, const float* bias, const float* resid, int M, int N, int K,
                        long lda, long ldb, long ldc, long ldr, int c_fp32, float alpha, int act, void* G, long ldg,
                        hipStream_t st) {
  return ctclip_gemm3_launch_hm(A, B, C, bias, resid, M, N, K, lda, ldb, ldc, ldr, c_fp32, alpha, act, G, ldg, 0, 0, st);
}

// hm_n > 0: bf16 output in the head-major layout [part][sequence][head][token][32] (Args::hm_n); M % hm_n == 0, N % 64 == 0
int ctclip_gemm3_launch_hm(const void* A, const void* B, void* C, const float* bias, const float* resid, int M, int N, int K,
                           long lda, long ldb, long ldc, long ldr, int c_fp32, float alpha, int act, void* G, long ldg,
                           int hm_n, int hm_heads, hipStream_t st) {
  using namespace g3;
  // (the epilogue divides the row by hm_n with a 32-bit multiply-high: exact while row * hm_n < 2^32, and the magic of hm_n = 1
  // does not fit 32 bits)
  if (hm_n > 0 && (c_fp32 || act != 0 || hm_heads <= 0 || (M % hm_n) || (N & 63) || (N / 32) % hm_heads || (((uintptr_t)C) & 15) ||
                   hm_n < 2 || (long)M * hm_n >= (1L << 32)))
    return (int)hipErrorInvalidValue;
  if (act < 0 || act > 3) return (int)hipErrorInvalidValue;
  if (act == 2 && (c_fp32 || bias || resid || !G || (N & 63) || (ldc & 7) || (ldg & 7) || (((uintptr_t)C) & 15) ||
                   (((uintptr_t)G) & 15)))
    return (int)hipErrorInvalidValue;
  if (act == 3 && (c_fp32 || bias || resid || !G || (N & 31) || (ldg & 7) || (((uintptr_t)G) & 15)))
    return (int)hipErrorInvalidValue;
  Args g{};
  g.A = (const bf16_t*)A; g.B = (const bf16_t*)B; g.C = C; g.bias = bias; g.resid = resid;
  g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.ldr = ldr; g.M = M; g.N = N; g.K = K;
  g.act = act; g.alpha = alpha; g.G = (bf16_t*)G; g.ldg = ldg;
  // the register epilogue moves 16-byte vectors of 8 (bf16) / 4 (f32) columns
  g.direct = (act >= 2) ||
             ((N & 7) == 0 && (((uintptr_t)C) & 15) == 0 && (ldc & (c_fp32 ? 3 : 7)) == 0 &&
              (!bias || (((uintptr_t)bias) & 15) == 0) && (!resid || ((((uintptr_t)resid) & 15) == 0 && (ldr & 3) == 0)));
  if (hm_n > 0) {
    if (!g.direct) return (int)hipErrorInvalidValue;
    g.hm_n = hm_n; g.hm_heads = hm_heads;
    g.hm_magic = (uint32_t)(((1ull << 32) + (unsigned long long)hm_n - 1) / (unsigned long long)hm_n);
    g.hm_part = (long)M * hm_heads * 32;
  }
  return g3_launch(g, act >= 2 ? act : (c_fp32 ? 1 : 0), c_fp32, st);
}

// The bf16 product with the per-head cosine normalisation applied in the epilogue (gemm_tile.h, EPI 5): the heads (32 columns
// each) of the first norm_cols columns leave normalised, scaled by scale[d] * mult, and their 1 / norm goes to inv [M, norm_cols /
// 32]; hm_n > 0 writes the head-major layout as ctclip_gemm3_launch_hm does.  norm_cols % 64 == 0, N % 64 == 0, 16-byte aligned C.
int ctclip_gemm3_launch_hn(const void* A, const void* B, void* C, int M, int N, int K, long lda, long ldb, long ldc, int hm_n,
                           int hm_heads, const float* scale, float mult, float* inv, int norm_cols, hipStream_t st) {
  using namespace g3;
  if (!scale || !inv || norm_cols <= 0 || norm_cols > N || (norm_cols & 63) || (N & 63) || (((uintptr_t)C) & 15) || (ldc & 7) ||
      (K % BK))
    return (int)hipErrorInvalidValue;
  if (hm_n > 0 && (hm_heads <= 0 || (M % hm_n) || (N / 32) % hm_heads || hm_n < 2 || (long)M * hm_n >= (1L << 32)))
    return (int)hipErrorInvalidValue;
  Args g{};
  g.A = (const bf16_t*)A; g.B = (const bf16_t*)B; g.C = C;
  g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.M = M; g.N = N; g.K = K; g.alpha = 1.0f; g.direct = 1;
  g.hn_scale = scale; g.hn_mult = mult; g.hn_inv = inv; g.hn_cols = norm_cols;
  if (hm_n > 0) {
    g.hm_n = hm_n; g.hm_heads = hm_heads;
    g.hm_magic = (uint32_t)(((1ull << 32) + (unsigned long long)hm_n - 1) / (unsigned long long)hm_n);
    g.hm_part = (long)M * hm_heads * 32;
  }
  return g3_launch(g, 5, 0, st);
}

// The f32 product with the LayerNorm backward applied in the epilogue (gemm_tile.h, EPI 4): C = A B^T - c1[row] - xhat c2[row]
// + resid, optional bf16 copy.  Same preconditions as the plain launch plus 16-byte aligned C / C16 / resid and 8-byte aligned
// xhat rows (N % 8 == 0): hipErrorInvalidValue otherwise -- there is no element-wise form of this epilogue.
int ctclip_gemm3_launch_ln(const void* A, const void* B, float* C, void* C16, const float* resid, int M, int N, int K, long lda,
                           long ldb, long ldc, long ldc16, long ldr, const void* xhat, long ldx, const float* c1, const float* c2,
                           hipStream_t st) {
  using namespace g3;
  if (!xhat || !c1 || !c2 || (N & 7) || (((uintptr_t)C) & 15) || (ldc & 3) || (((uintptr_t)xhat) & 7) || (ldx & 3) ||
      (C16 && ((((uintptr_t)C16) & 7) || (ldc16 & 3))) || (resid && ((((uintptr_t)resid) & 15) || (ldr & 3))))
    return (int)hipErrorInvalidValue;
  Args g{};
  g.A = (const bf16_t*)A; g.B = (const bf16_t*)B; g.C = C; g.resid = resid;
  g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.ldr = ldr; g.M = M; g.N = N; g.K = K;
  g.alpha = 1.0f; g.direct = 1;
  g.xhat = (const bf16_t*)xhat; g.ldx = ldx; g.c1 = c1; g.c2 = c2; g.C16 = (bf16_t*)C16; g.ldc16 = ldc16;
  return g3_launch(g, 4, 1, st);
}

#ifdef CTCLIP_G3_STAMPS
extern "C" int ctclip_debug_gemm3_occupancy(int bn, int unused) {
  using namespace g3;
  int n = -1;
  hipError_t e = bn == 128 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, gemm3_kernel<128, 0>, 512, (size_t)3 * (SUB + 128 * BK * 2))
                           : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, gemm3_kernel<256, 0>, 512, (size_t)4 * (SUB + 256 * BK * 2));
  return e == hipSuccess ? n : -(int)e;
}

extern "C" int ctclip_debug_gemm3_prof(void* buf) {
  unsigned long long* p = (unsigned long long*)buf;
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g3::g_prof), &p, sizeof(p));
}

extern "C" int ctclip_debug_gemm3_stamps(void* buf, long capacity_blocks) {
  unsigned long long* p = (unsigned long long*)buf;
  hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(g3::g_stamps), &p, sizeof(p));
  if (e != hipSuccess) return (int)e;
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g3::g_stamp_cap), &capacity_blocks, sizeof(capacity_blocks));
}
#endif

// called by ctclip_vq_topk / ctclip_vq_topk_grouped (gemm.hip): codes M % 256 == 0, K % 32 == 0, (M / 256) % cgroups == 0;
// part_val / part_idx are [N][16 * cgroups]
int ctclip_vq_topk3_launch(const void* A, const void* B, float* part_val, int* part_idx, int M, int N, int K, long lda,
                           long ldb, int cgroups, hipStream_t st) {
  using namespace g3;
  if (cgroups < 1 || cgroups > 32 || (32 % cgroups) || (M / BM) % cgroups) return (int)hipErrorInvalidValue;
  VqArgs g{};
  g.A = (const bf16_t*)A; g.B = (const bf16_t*)B; g.lda = lda; g.ldb = ldb; g.M = M; g.N = N; g.K = K;
  g.cgroups = cgroups; g.ttiles = 32 / cgroups; g.ntok_tiles = (N + 255) / 256;
  g.tiles_m = M / BM / cgroups; g.part_val = part_val; g.part_idx = part_idx;
  const size_t lds = (size_t)4 * 2 * SUB;          // 128 KiB
  CTCLIP_LDS_LIMIT_ONCE(vq_topk3_kernel, lds);
  unsigned grid = (unsigned)g.ntok_tiles;
  if (cgroups > 1) {
    const int per_round = 8 * g.ttiles;                               // token tiles the eight XCDs take side by side
    grid = (unsigned)((g.ntok_tiles + per_round - 1) / per_round) * 8u * 32u;
  }
  hipLaunchKernelGGL(vq_topk3_kernel, dim3(grid), dim3(512), lds, st, g);
  return (int)hipGetLastError();
}
