// bf16 MFMA GEMM, 256 x 256 tile, ONE WAVE PER SIMD: 4 waves x (128 x 128).  Takes the k-major x k-major products of the CT-CLIP
// step on which it measured faster than gemm3.hip (dispatcher: csrc/gemm.hip, gemm5_takes): FF1 + GEGLU and plain products with
// N >= 2048 or K >= 1024 (src/utils/attention.py:38-51; every nn.Linear of the path goes through ctclip_gemm_bf16).
//
// Why a second tile shape.  gemm3.hip runs 8 waves x (128 x 64): per 32-deep K-step a wave reads 12 fragments for 32 MFMAs and
// the two waves of a SIMD alternate between a load block and an MFMA block.  The vendor library's fastest kernels on the long-K
// shapes (profiles/r04_gemm_vs_vendor.txt: MT256x256x64, 256 threads, 130 KiB of LDS, 256 + 256 registers) are one wave per SIMD
// with a 128 x 128 wave tile: 16 fragment reads per 64 MFMAs -- half the LDS traffic per flop.  This kernel is that design on
// this repo's ring (measurements: profiles/r04_gemm5.txt):
//   * 256 accumulator registers per lane (8 x 8 MFMA 16x16x32 tiles) pinned to AccVGPRs; row fragments rotate, column
//     fragments are double-buffered (96 registers): the 16 ds_read_b128 of K-step j + 1 go out between the 64 MFMAs of K-step j
//     (one wave can hide them under its own MFMAs), ONE barrier per K-step;
//   * the same k32 LDS tiles, source-side swizzle and permuted N-fragment rows as gemm3.hip (gemm_tile.h), five slots of 32 KiB
//     (all 160 KiB of LDS) filled by global_load_lds with counted vmcnt, PAIRED (both k32 halves of a 128-byte operand line are
//     requested close together so that the L1 sends one request per line) and EVENLY SPACED: one request per 8 MFMAs (see the kernel);
//   * the (tile, K-step) sequence of a workgroup is FLATTENED: the ring never drains between tiles, the loop body is
//     branch-free and every wait count is a constant;
//   * the register epilogue of gemm_tile.h, once per 64-column slab of the wave's 128 columns.
// What bounds it (and gemm3) is neither the matrix pipe nor instruction issue: with the DMAs removed the loop runs at 1600
// TFLOP/s, with operands that hit in L2 a probe of the same K-step hides everything under the MFMAs; the kernel waits on L1 misses
// in flight / their latency (80-100 lines per CU x 330-610 cycles -- the same as the vendor kernel, whose latency is 25 % lower).
// Preconditions (dispatcher in gemm.hip): both operands k-major, K % 64 == 0, K >= 192, 16-byte aligned outputs, no split-K.
#include "gemm_tile.h"
#include <stdlib.h>
#include <type_traits>

namespace g5 {
using namespace g3;

constexpr int G5_NS = 5;   // ring slots of 32 KiB: the whole 160 KiB of LDS

#ifdef CTCLIP_G5_PROF
// diagnostic build only (hipcc -DCTCLIP_G5_PROF, tools/gemm5_prof.py): per wave, shader-cycle sums of the K-step's segments
// {MFMA + issue block, vmcnt wait, lgkmcnt wait, barrier}, the epilogue, the K-step count and the wave's whole life
__device__ unsigned long long* g5_prof = nullptr;      // [blocks][4 waves][8]
#define G5_SEG_DECL() unsigned long long seg_[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long c_ = __builtin_amdgcn_s_memtime(); const unsigned long long c0_ = c_
#define G5_SEG(n) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); seg_[n] += t_ - c_; c_ = t_; } while (0)
#define G5_CNT(n) do { seg_[n] += 1; } while (0)
#define G5_SEG_STORE()                                                                                            \
  do {                                                                                                            \
    seg_[7] = __builtin_amdgcn_s_memtime() - c0_;                                                                 \
    if (g5_prof && lane == 0)                                                                                     \
      for (int n_ = 0; n_ < 8; ++n_) g5_prof[((long)blockIdx.x * 4 + wave) * 8 + n_] = seg_[n_];                  \
  } while (0)
#else
#define G5_SEG_DECL() do { } while (0)
#define G5_SEG(n) do { } while (0)
#define G5_CNT(n) do { } while (0)
#define G5_SEG_STORE() do { } while (0)
#endif

// The accumulators are pinned to the AccVGPR half of the register file by issuing the MFMAs from inline asm ("+a"): left to
// the register allocator, the 256 accumulator + 128 fragment registers of this kernel end in copies between the two halves
// and ~700 spilled dwords.  Hazards the compiler no longer sees: an MFMA that accumulates into the registers the previous
// MFMA wrote needs no wait states when the registers are exactly the same (they are); the epilogue's first v_accvgpr_read
// is kept 18+ wait states behind the last MFMA by the barrier and the s_nops in front of it.
__device__ __forceinline__ void mfma_acc(f32x4& c, bf16x8 a, bf16x8 b) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma_zero(f32x4& c, bf16x8 a, bf16x8 b) {      // first K-step of a tile: C = 0
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(c) : "v"(a), "v"(b));
}

template <int EPI, int NS>
__global__ __launch_bounds__(256) void gemm5_kernel(Args g) {
  constexpr bool F32OUT = EPI == 1 || EPI == 4;     // identity N-fragment rows: a lane holds 4 contiguous f32 columns
  constexpr int IM = 8, JN = 8, BN = 256;
  constexpr int STAGE = 2 * SUB;                     // a slot: the A tile (256 x 32 bf16) and the B tile behind it, 32 KiB
  constexpr int PPW = STAGE / 1024 / 4;              // 1 KiB DMA pieces per wave and slot: 8
  constexpr int NST = 2 * IM * (EPI == 0 ? 2 : EPI == 2 ? 3 : 4);   // 16-byte stores per wave of a full tile's register epilogue
  static_assert(NS == 5, "the paired issue below is written for five slots");
  // PAIRED, EVENLY SPACED ISSUE: the two k32 halves of the same 128-byte operand lines go to two ring slots (K-steps 2P + 4 and
  // 2P + 5); the second request for a line merges with the first in the L1, so the L1 -> L2 request count is that of full lines
  // (profiles/r04_gemm5.txt sections 6, 7).  Round 5: ONE request per row group of 8 MFMAs -- pieces 0-3 of the pair (both halves,
  // alternating) during step 2P, pieces 4-7 during step 2P + 1 -- instead of all sixteen during the even step and none during
  // the odd one: the evenly spaced stream queues less in the memory system (FF1 data gradient 1004 -> 1071 TFLOP/s, FF2 forward
  // 898 -> 932, five interleaved repetitions on one box: profiles/r05_gemm5_even_issue.txt; it is also what the vendor's kernel does,
  // r05_vendor_isa.txt).  A K-step is only known complete when its whole pair is: K-step j + 2 and its partner were requested one
  // and two steps before step j's own requests, so an even step may leave its PPW requests in flight, an odd step its own and the
  // even step's.
  constexpr int NPRO = 4;                            // K-steps of the ring fill: two pairs
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;

  const int total = g.tiles_m * g.tiles_n;
  if ((int)blockIdx.x >= total) return;
  const int nk = g.K / BK;                           // even, >= NS (dispatcher)

  // ---- issue side: the (tile, K-step) sequence of this workgroup is FLATTENED, the ring never drains between tiles.
  // Piece q = wave * PPW + j of a slot: the first 16 are the A tile (waves 0, 1), the rest the B tile (waves 2, 3), so a wave
  // streams ONE operand: a uniform 64-bit base (tile row 0 + the K-step's offset, SGPRs) and a 32-bit byte offset per piece
  // and lane (rows clamped to the operand's last row: masked in the epilogue).  LDS destination of piece j: slot + dst0 + j KiB.
  const bool isA = wave < 2;
  const bf16_t* const opnd = isA ? g.A : g.B;
  const long ld = isA ? g.lda : g.ldb;
  const int R = isA ? g.M : g.N;
  const uint32_t dst0 = (uint32_t)(wave * PPW * 1024);
  const char* ibase;                                 // operand at (issue tile row 0, issue K-step)
  // byte offset of piece i for this lane = min(voff0 + i * rowstep, vlast): sixteen rows further per piece, clamped to the
  // operand's last row (two registers instead of one per piece: every register of this kernel is spoken for)
  uint32_t voff0, vlast;
  const uint32_t rowstep = (uint32_t)(16 * ld * 2);
  uint32_t half1 = 64;                               // the second k32 half of a line, kept in a register: an immediate
  asm volatile("" : "+s"(half1));                    // offset of a global_load_lds is added to the LDS address as well
  int itile = blockIdx.x, ik = 0, islot = 0;
  auto locate_issue = [&](int t) {
    const int bid = xcd_remap(t, total);
    const int r0 = isA ? (bid / g.tiles_n) * BM : (bid % g.tiles_n) * BN;
    ibase = (const char*)(opnd + (long)r0 * ld);
    const int r = 16 * ((wave & 1) * PPW) + (lane >> 2), c = (lane & 3) ^ swz_key(r);     // the key is the same for every piece
    voff0 = (uint32_t)((long)r * ld * 2 + c * 16);
    vlast = (uint32_t)((long)(R - 1 - r0) * ld * 2 + c * 16);
  };
  // after the wave's PPW pieces of a K-step have been issued: next K-step, next tile of this workgroup; past the last tile the
  // last K-step is issued again (into a slot nobody reads any more), which keeps the loop body and the wait counts uniform
  auto advance_issue = [&]() {                       // by a pair of K-steps (nk is even)
    islot = islot + 2 >= NS ? islot + 2 - NS : islot + 2;
    if ((ik += 2) < nk) { ibase += 2 * BK * 2; return; }
    const int nt = itile + (int)gridDim.x;
    if (nt < total) { itile = nt; ik = 0; locate_issue(nt); }
    else ik = nk - 2;
  };
  // piece i of the pair at the issue position: the k32 half at ibase into slot islot and the other half of the same lines (64
  // bytes further) into the next slot, back to back
  auto issue_piece = [&](int i) {
    const uint32_t v = min(voff0 + (uint32_t)i * rowstep, vlast);
    const int s1 = islot + 1 == NS ? 0 : islot + 1;
    G3_GLDS(ibase + v, lds0 + (uint32_t)(islot * STAGE) + dst0 + i * 1024);
    G3_GLDS(ibase + (v + half1), lds0 + (uint32_t)(s1 * STAGE) + dst0 + i * 1024);
  };
  auto issue_half = [&](int i, int h) {               // one half of piece i: the main loop's single request per row group
    const uint32_t v = min(voff0 + (uint32_t)i * rowstep, vlast);
    const int s1 = islot + 1 == NS ? 0 : islot + 1;
    if (h == 0) G3_GLDS(ibase + v, lds0 + (uint32_t)(islot * STAGE) + dst0 + i * 1024);
    else G3_GLDS(ibase + (v + half1), lds0 + (uint32_t)(s1 * STAGE) + dst0 + i * 1024);
  };
  locate_issue(itile);
#pragma unroll
  for (int t = 0; t < NPRO; t += 2) {
#pragma unroll
    for (int j = 0; j < PPW; ++j) issue_piece(j);
    advance_issue();
  }

  // ---- consume side.  Fragment addresses inside a slot: M rows 16 i + ml of this wave's 128 rows; (permuted) N rows of its
  // two 64-column slabs.  N-fragment j = 4 s + jj (slab s, fragment jj): slab rows 16 jj + ml (f32 outputs) or nfrag_row(jj, ml)
  // (bf16): 64 rows (a slab) or 32 rows (jj + 2) further is a constant 4096 / 2048 bytes; jj + 1 changes the swizzle key of the
  // permuted form
  const int ml = lane & 15, q4 = lane >> 4;
  const uint32_t offA = tile_off(wm * 128 + ml, q4);          // + i * 1024: sixteen rows further, same swizzle key
  const uint32_t offB0 = SUB + tile_off(wn * 128 + (F32OUT ? ml : nfrag_row(0, ml)), q4);
  const uint32_t offB1 = SUB + tile_off(wn * 128 + (F32OUT ? 16 + ml : nfrag_row(1, ml)), q4);
  auto offB = [&](int j) { return ((j & 1) ? offB1 : offB0) + (uint32_t)(4096 * (j >> 2) + 2048 * ((j >> 1) & 1)); };

  G5_SEG_DECL();
  f32x4 acc[IM][JN];
  // the row fragments ROTATE (fa[i] is re-loaded with the next K-step's rows as soon as its eight MFMAs have been issued: no
  // second copy), the column fragments are double-buffered: 96 fragment registers, which leaves the allocator the slack that
  // keeps the loop free of spills (a scratch reload waits with vmcnt(0) and would drain the ring)
  bf16x8 fa[IM], fb[2][JN];
  int cslot = 0;                                     // slot of the K-step whose fragments are in registers
  int post = 0;                                      // K-steps after an epilogue whose wait leaves its stores in flight

#define G5_BAR()                                                                       \
  do {                                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                 \
    __builtin_amdgcn_s_barrier();                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                 \
  } while (0)
  // One K-step: 64 MFMAs on (fac, fbc); the 16 fragment reads of the next K-step (slot cslot + 1) and this wave's 8 DMA issues
  // of K-step + NS (into slot cslot, whose fragments are the ones in registers) go out between the row groups.  The last
  // reads are issued one row group early so that their latency is covered.  ZERO: first K-step of a tile (C = 0).
  auto step = [&](auto zero_t, auto odd_t, bf16x8 (&fbc)[JN], bf16x8 (&fbn)[JN]) {
    constexpr bool ZERO = decltype(zero_t)::value, ODD = decltype(odd_t)::value;
    const int nslot = cslot + 1 == NS ? 0 : cslot + 1;
    const char* st = smem + nslot * STAGE;
#pragma unroll
    for (int i = 0; i < IM; ++i) {
#pragma unroll
      for (int jj = 0; jj < JN; ++jj) {                                  // transposed: see nfrag_row()
        if constexpr (ZERO) mfma_zero(acc[i][jj], fbc[jj], fa[i]);
        else mfma_acc(acc[i][jj], fbc[jj], fa[i]);
      }
      fa[i] = *(const bf16x8*)(st + offA + i * 1024);
      if (i < IM - 1) fbn[i] = *(const bf16x8*)(st + offB(i));
      if (i == IM - 2) fbn[JN - 1] = *(const bf16x8*)(st + offB(JN - 1));
      issue_half((ODD ? PPW / 2 : 0) + (i >> 1), i & 1);
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (ODD) advance_issue();
    cslot = nslot;
    G5_SEG(0); G5_CNT(5);
    // own pieces of the K-step after the next landed (it is read during the next step, behind this barrier)
    // (an epilogue's NST stores may stay in flight behind them for the two steps after it; vmcnt holds 6 bits)
    constexpr int C_SP = ODD ? 2 * PPW : PPW, C_SPST = C_SP + NST > 63 ? 63 : C_SP + NST;
    if (post > 0) { wait_vm<C_SPST>(); --post; } else wait_vm<C_SP>();
    G5_SEG(1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the reads of slot `nslot` retired before the barrier that frees it
    G5_SEG(2);
    G5_BAR();
    G5_SEG(3);
  };
  using T_ = std::true_type;
  using F_ = std::false_type;

  // K-steps 0 and 1 landed for everybody; fragments of K-step 0; its slot is free
  wait_vm<(NPRO - 2) * PPW>();
  G5_BAR();
#pragma unroll
  for (int i = 0; i < IM; ++i) fa[i] = *(const bf16x8*)(smem + offA + i * 1024);
#pragma unroll
  for (int j = 0; j < JN; ++j) fb[0][j] = *(const bf16x8*)(smem + offB(j));
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  G5_BAR();

  for (int tile = blockIdx.x; tile < total; tile += (int)gridDim.x) {
    step(T_{}, F_{}, fb[0], fb[1]);
    step(F_{}, T_{}, fb[1], fb[0]);
    for (int k = 2; k < nk; k += 2) {
      step(F_{}, F_{}, fb[0], fb[1]);
      step(F_{}, T_{}, fb[1], fb[0]);
    }
    const int bid = xcd_remap(tile, total);
    const int erow0 = (bid / g.tiles_n) * BM, ecol0 = (bid % g.tiles_n) * BN;
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");   // MFMA results -> v_accvgpr_read (see mfma_acc)
    const int rbase = erow0 + wm * 128, colw = ecol0 + wn * 128;
    // the lane id afresh (mbcnt), not the register live since the kernel's entry: keeps the epilogue's per-lane addresses from
    // being hoisted out of the tile loop, where they would stay live across the matrix loop (every register there is spoken for)
    int lane_e = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    asm volatile("" : "+v"(lane_e));
    epilogue_slab<EPI, IM, JN, 0, true, false>(g, acc, rbase, colw, lane_e);
    epilogue_slab<EPI, IM, JN, 4, true, false>(g, acc, rbase, colw + 64, lane_e);
    if (g.direct && erow0 + BM <= g.M && ecol0 + BN <= g.N) post = 2;     // the next two steps wait for pieces older than the stores
    else wait_vm<0>();                                  // ragged tile: its store count is not the constant NST
    G5_SEG(4);
  }
  G5_SEG_STORE();
  wait_vm<0>();                                         // no LDS-DMA may be in flight when the workgroup's LDS is released
#undef G5_BAR
}

}  // namespace g5

// Does this kernel take the problem at all?  (K a whole number of k64 pairs and at least the ring fill; outputs that allow the
// 16-byte register epilogue.)  The dispatcher asks BEFORE launching, so that an error code of the launch itself is never
// mistaken for "not eligible".
static bool g5_direct(const void* C, const float* bias, const float* resid, int N, long ldc, long ldr, int c_fp32, int act) {
  return (act >= 2) ||
         ((N & 7) == 0 && (((uintptr_t)C) & 15) == 0 && (ldc & (c_fp32 ? 3 : 7)) == 0 &&
          (!bias || (((uintptr_t)bias) & 15) == 0) && (!resid || ((((uintptr_t)resid) & 15) == 0 && (ldr & 3) == 0)));
}
bool ctclip_gemm5_eligible(const void* C, const float* bias, const float* resid, int N, int K, long ldc, long ldr, int c_fp32,
                           int act) {
  using namespace g5;
  return (K % (2 * BK)) == 0 && K / BK >= 6 && g5_direct(C, bias, resid, N, ldc, ldr, c_fp32, act);
}

// called by the dispatcher in gemm.hip: same contract as ctclip_gemm3_launch_hm
int ctclip_gemm5_launch(const void* A, const void* B, void* C, const float* bias, const float* resid, int M, int N, int K,
                        long lda, long ldb, long ldc, long ldr, int c_fp32, float alpha, int act, void* G, long ldg,
                        int hm_n, int hm_heads, hipStream_t st) {
  using namespace g5;
  if (hm_n > 0 && (c_fp32 || act != 0 || hm_heads <= 0 || (M % hm_n) || (N & 63) || (N / 32) % hm_heads || (((uintptr_t)C) & 15) ||
                   hm_n < 2 || (long)M * hm_n >= (1L << 32)))
    return (int)hipErrorInvalidValue;
  if (act < 0 || act > 3) return (int)hipErrorInvalidValue;
  if (act == 2 && (c_fp32 || bias || resid || !G || (N & 63) || (ldc & 7) || (ldg & 7) || (((uintptr_t)C) & 15) ||
                   (((uintptr_t)G) & 15)))
    return (int)hipErrorInvalidValue;
  if (act == 3 && (c_fp32 || bias || resid || !G || (N & 31) || (ldg & 7) || (((uintptr_t)G) & 15)))
    return (int)hipErrorInvalidValue;
  if (!ctclip_gemm5_eligible(C, bias, resid, N, K, ldc, ldr, c_fp32, act)) return (int)hipErrorInvalidValue;
  Args g{};
  g.A = (const bf16_t*)A; g.B = (const bf16_t*)B; g.C = C; g.bias = bias; g.resid = resid;
  g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.ldr = ldr; g.M = M; g.N = N; g.K = K;
  g.act = act; g.alpha = alpha; g.G = (bf16_t*)G; g.ldg = ldg;
  g.direct = 1;                                        // (unaligned outputs stay on gemm3.hip: ctclip_gemm5_eligible)
  if (hm_n > 0) {
    g.hm_n = hm_n; g.hm_heads = hm_heads;
    g.hm_magic = (uint32_t)(((1ull << 32) + (unsigned long long)hm_n - 1) / (unsigned long long)hm_n);
    g.hm_part = (long)M * hm_heads * 32;
  }
  g.tiles_m = (M + BM - 1) / BM; g.tiles_n = (N + 255) / 256;
  const int cus = ctclip_cu_count8();                  // a multiple of 8 keeps a workgroup's tiles on one XCD's run
  const long total = (long)g.tiles_m * g.tiles_n;
  const int grid = (int)(total < cus ? total : cus);
  constexpr size_t lds = (size_t)G5_NS * 2 * SUB;
#define G5_LAUNCH(EPI_)                                                                                                   \
  do {                                                                                                                    \
    CTCLIP_LDS_LIMIT_ONCE((gemm5_kernel<EPI_, G5_NS>), lds);                                                              \
    hipLaunchKernelGGL((gemm5_kernel<EPI_, G5_NS>), dim3(grid), dim3(256), lds, st, g);                                   \
  } while (0)
  const int epi = act >= 2 ? act : (c_fp32 ? 1 : 0);
  if (epi == 0) G5_LAUNCH(0);
  else if (epi == 1) G5_LAUNCH(1);
  else if (epi == 2) G5_LAUNCH(2);
  else G5_LAUNCH(3);
#undef G5_LAUNCH
  return (int)hipGetLastError();
}

#ifdef CTCLIP_G5_PROF
extern "C" int ctclip_debug_gemm5_prof(void* buf) {
  unsigned long long* p = (unsigned long long*)buf;
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g5::g5_prof), &p, sizeof(p));
}
#endif
