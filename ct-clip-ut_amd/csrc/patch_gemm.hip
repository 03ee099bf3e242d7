// CT-ViT tubelet patch embedding as ONE pass over the volume (reference src/utils/ctvit.py:44-52: Rearrange 'b c (t pt) (h p1)
// (w p2) -> b t h w (c pt p1 p2)', LayerNorm(F), Linear(F, dim); the trailing LayerNorm(dim) stays ctclip_layernorm_fwd).
//
// The unfused chain (patch.hip + gemm3) writes the normalised [tokens, F] bf16 operand -- 10.6 GB at 96 pairs -- and reads it
// twice (projection, weight gradient).  Here the MFMA operand is built from the raw voxels inside the GEMM's load block:
//
//   * a workgroup owns 128 consecutive tokens x ALL 512 output columns (8 waves = 2 x 4, 64 x 128 per wave): every voxel is
//     read from HBM once and centred once; the folded weight W' = W o gamma (4 MB, L2 / Infinity-Cache resident) is streamed per
//     tile through the gemm3 ring (global_load_lds, four 32 KiB slots, counted vmcnt, role-alternating main loop);
//   * K runs in FEATURE order, 32 features per step = 8 pieces of 4 voxels per token (a piece never straddles a p2-run because
//     p % 4 == 0): thread (token m = tid / 4, pieces j = tid % 4 and j + 4) loads its two 8-byte pieces two K-steps ahead into
//     registers, subtracts the token's centring constant c, rounds to bf16 and writes the k32 tile of gemm_tile.h
//     (ds_write_b64) one K-step ahead of its use -- VALU work in the load block, next to the partner wave's MFMAs;
//   * c = bf16(mean of the token's first p2-run): (x - c) is EXACTLY 0 for a constant tubelet (air = -1 padding), which is what
//     keeps the reference's xhat = 0 there; the statistics ride along, mu' = mean(x - c) and var = mean((x - c)^2) - mu'^2 in f32
//     from the exact differences, and the epilogue finishes the LayerNorm:
//         z = rstd (acc - mu' S[n]) + b'[n],      S[n] = sum_f W'[n][f],  b' = b + W beta        (ctclip_patch_affine_fold)
//     |mu'| is a fraction of the tubelet's own spread, so the correction term carries no cancellation (for a constant tubelet it
//     is exactly 0).  (c, mu', rstd, mean) are stored per token for the backward.
//
// The weight gradient G = dz^T xhat recomputes the centred operand from the volume the same way (patch_wgrad_kernel below), so
// the [tokens, F] operand is never allocated in training.
// Preconditions (the entry points return hipErrorInvalidValue otherwise; the host layer then keeps the unfused chain): bf16
// volume, p % 4 == 0, F % 32 == 0, 128 <= F <= 4096, N == 512, 8-byte aligned runs (Wx % 4 == 0, 16-byte aligned volume).
#include "gemm_tile.h"
#include "gemm_tile_t.h"

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
  float* tstat;                   // [tokens][4] f32: c, mu' = mean - c, rstd, mean -- what the backward needs per token
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

#define PG_BAR()                                                                       \
  do {                                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                 \
    __builtin_amdgcn_s_barrier();                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                 \
  } while (0)

// vmcnt bookkeeping of the forward kernel.  "Block" j (the prologue's blocks -3, -2, -1 and load_block(j) of the main loop) issues
// first the register loads R(j + 3) and then the LDS-DMA pieces D(j + 3): OPS instructions, always (past the last K-step the
// last one again).  A wave's vmcnt retires in order, so
//   * R(k + 1), converted in block k, has landed when at most  D(k + 1) + [block k - 1]  are outstanding;
//   * D(k + 1), read after the barrier that ends K-step k, has landed when at most  [block k - 1] + [block k]  are.
__global__ __launch_bounds__(512, 2) void patch_gemm_fwd_kernel(FwdArgs a) {
  constexpr int IM = 4, JN = 8;
  constexpr int OPS = 2 + TPPW;                              // vm instructions per block: 2 register loads + 4 DMA pieces
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* ptab = (int*)(smem + OFF_TAB);
  float* stat = (float*)(smem + OFF_STAT);                   // [128][2]: mu', rstd
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int grp = wave >> 2;                                  // role group: SIMD partners are waves w and w + 4
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  const Geom& g = a.g;
  const int nk = g.F / BK;
  const int row0 = (int)blockIdx.x * PBM;

  // ---- the piece table and this thread's token
  for (int i = tid; i < g.F / 4; i += 512) ptab[i] = piece_offset(g, 4 * i);
  const int m = tid >> 2, j0 = tid & 3;
  int tok = row0 + m;
  if (tok >= a.M) tok = a.M - 1;                             // masked in the epilogue
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
  float cb = bf16_to_f32(f32_to_bf16(csum / (float)g.p));
  // every load the COMPILER knows about is retired here: the loads below are inline asm, invisible to its wait-count bookkeeping,
  // so a compiler-computed vmcnt for an older visible load would come out too large
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(cb) : : "memory");

  // ---- W' through the ring: piece q = wave * TPPW + j of a stage = rows 16 q .. 16 q + 15 of the 512
  const bf16_t* srcB[TPPW];
  uint32_t dstB[TPPW];
#pragma unroll
  for (int j = 0; j < TPPW; ++j) {
    const int q = wave * TPPW + j;
    srcB[j] = a.W + piece_src(q, lane, 0, PBN, a.ldw);
    dstB[j] = (uint32_t)(q * 1024);
  }
  auto issue_B = [&](int t) {                                 // t >= nk: the last K-step again (keeps the vmcnt arithmetic uniform)
    const int tt = t < nk ? t : nk - 1;
    const uint32_t sb = lds0 + (uint32_t)((t % NSB) * BSTAGE);
#pragma unroll
    for (int j = 0; j < TPPW; ++j) G3_GLDS(srcB[j] + (long)tt * BK, sb + dstB[j]);
  };
  // this thread's two pieces of a K-step: where they go in the k32 tile (row m, 8-byte half j & 1 of chunk j >> 1)
  const uint32_t wo0 = OFF_A + tile_off(m, j0 >> 1) + 8 * (j0 & 1);
  const uint32_t wo1 = OFF_A + tile_off(m, (j0 + 4) >> 1) + 8 * (j0 & 1);
  float s1 = 0.f, s2 = 0.f;
  uint2 ra0[2], ra1[2];                                       // register slot = K-step parity, [piece]
  int po0, po1;                                               // piece offsets of the NEXT K-step to request (read one block ahead)
  const uint32_t tabaddr = lds0 + OFF_TAB + 4 * j0;
  auto read_tab = [&](int t) {                                // -> po0 / po1, valid behind the block's lgkmcnt wait
    const int tt = t < nk ? t : nk - 1;
    po0 = lds_rd32(tabaddr + (uint32_t)(tt * 32));
    po1 = lds_rd32(tabaddr + (uint32_t)(tt * 32 + 16));
  };
  auto load_A = [&](uint2 (&r)[2]) {                          // the K-step po0 / po1 name: pieces -> registers (asm: see gemm_tile.h)
    r[0] = glb_rd64(tb + po0);
    r[1] = glb_rd64(tb + po1);
  };
  auto convert_A = [&](int t, uint2 (&r)[2]) {                // registers -> centred bf16 in A slot t & 1
    float d0[4], d1[4];
    unpack4(r[0], cb, d0);
    unpack4(r[1], cb, d1);
    s1 += ((d0[0] + d0[1]) + (d0[2] + d0[3])) + ((d1[0] + d1[1]) + (d1[2] + d1[3]));
    s2 += ((d0[0] * d0[0] + d0[1] * d0[1]) + (d0[2] * d0[2] + d0[3] * d0[3])) +
          ((d1[0] * d1[0] + d1[1] * d1[1]) + (d1[2] * d1[2] + d1[3] * d1[3]));
    const uint32_t base = lds0 + (uint32_t)((t & 1) * ASTAGE);
    lds_wr64(base + wo0, make_uint2(pack_bf16x2(d0[0], d0[1]), pack_bf16x2(d0[2], d0[3])));
    lds_wr64(base + wo1, make_uint2(pack_bf16x2(d1[0], d1[1]), pack_bf16x2(d1[2], d1[3])));
  };

  __syncthreads();                                            // the piece table is visible
  // ---- prologue = blocks -3, -2, -1 (nk >= 4: the entry point checks F >= 128)
  read_tab(0); lgkm_wait_tied(po0, po1);
  load_A(ra0); issue_B(0);
  read_tab(1); lgkm_wait_tied(po0, po1);
  load_A(ra1); issue_B(1);
  read_tab(2); lgkm_wait_tied(po0, po1);
  vm_wait_tied<TPPW + OPS>(ra0[0], ra0[1]);                   // R(0) landed (behind it: D(0), R(1), D(1))
  convert_A(0, ra0);
  load_A(ra0); issue_B(2);
  read_tab(3);                                                // for block 0; valid behind the wait in front of the first barrier
  // R(1), R(2) landed (behind them only D(2)): the compiler may copy these registers where the two role loops begin
  asm volatile("s_waitcnt vmcnt(%4)" : "+v"(ra0[0]), "+v"(ra0[1]), "+v"(ra1[0]), "+v"(ra1[1]) : "n"(TPPW) : "memory");

  // fragment addresses: M rows 16 i + ml of this wave's 64 rows; N rows 16 j + ml of its 128 columns (f32 output: identity rows)
  const int ml = lane & 15, q4 = lane >> 4;
  const uint32_t offA = OFF_A + tile_off(wm * 64 + ml, q4);   // + i * 1024: sixteen rows further, same swizzle key
  const uint32_t offB = tile_off(wn * 128 + ml, q4);          // + j * 1024
  bf16x8 fa[IM], fb[JN];
  f32x4 acc[IM][JN];
#pragma unroll
  for (int i = 0; i < IM; ++i)
#pragma unroll
    for (int j = 0; j < JN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;

  // load_block(k): fragments of K-step k; convert R(k + 1) into A slot (k + 1) & 1 (last read for K-step k - 1, two barriers
  // ago); request R(k + 3) into the registers just freed and D(k + 3) into ring slot (k - 1) % 4; read the piece offsets of
  // K-step k + 4.  Everything that touches LDS or registers in flight is inline asm; frag_wait() makes it all valid.
  auto load_block = [&](int k, uint2 (&rs)[2]) {
    const uint32_t stB = lds0 + (uint32_t)((k % NSB) * BSTAGE) + offB;
    const uint32_t stA = lds0 + (uint32_t)((k & 1) * ASTAGE) + offA;
    fb[0] = lds_rd128<0>(stB); fb[1] = lds_rd128<1024>(stB); fb[2] = lds_rd128<2048>(stB); fb[3] = lds_rd128<3072>(stB);
    fb[4] = lds_rd128<4096>(stB); fb[5] = lds_rd128<5120>(stB); fb[6] = lds_rd128<6144>(stB); fb[7] = lds_rd128<7168>(stB);
    fa[0] = lds_rd128<0>(stA); fa[1] = lds_rd128<1024>(stA); fa[2] = lds_rd128<2048>(stA); fa[3] = lds_rd128<3072>(stA);
    // R(k + 1) landed: behind it D(k + 1) and block k - 1.  Every block issues its six instructions -- past the last K-step the
    // last one again, into registers / a slot nobody reads any more -- so the counts are constants and the wait is ONE asm
    // statement whose in / out registers coincide (a wait on one of two branches made the compiler copy the registers in front of
    // it, i.e. while the load was in flight)
    vm_wait_tied<TPPW + OPS>(rs[0], rs[1]);
    if (k + 1 < nk) convert_A(k + 1, rs);
    load_A(rs);
    issue_B(k + 3);
    read_tab(k + 4);
  };
  auto frag_wait = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(fa[0]), "+v"(fa[1]), "+v"(fa[2]), "+v"(fa[3]), "+v"(fb[0]), "+v"(fb[1]), "+v"(fb[2]), "+v"(fb[3]), "+v"(fb[4]),
                   "+v"(fb[5]), "+v"(fb[6]), "+v"(fb[7]), "+v"(po0), "+v"(po1)
                 :
                 : "memory");
  };
  auto mfma_block = [&]() {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < IM; ++i)
#pragma unroll
      for (int j = 0; j < JN; ++j) acc[i][j] = mfma16(fb[j], fa[i], acc[i][j]);      // transposed: gemm_tile.h nfrag_row()
    __builtin_amdgcn_s_setprio(0);
  };
  auto wait_next = [&](int k) { wait_vm<2 * OPS>(); };          // D(k + 1) landed: behind it blocks k - 1 and k

  lgkm_wait_tied(po0, po1);                                   // A slot 0 written, the piece offsets of K-step 3 read
  PG_BAR();
  // the two waves of a SIMD run half a K-step apart (gemm3.hip): even k converts from / loads into ra1, odd k ra0
  if (grp == 0) {
    for (int k = 0; k < nk; k += 2) {
      load_block(k, ra1);
      frag_wait();
      PG_BAR();
      mfma_block();
      wait_next(k);
      PG_BAR();
      if (k + 1 < nk) {
        load_block(k + 1, ra0);
        frag_wait();
        PG_BAR();
        mfma_block();
        wait_next(k + 1);
        PG_BAR();
      }
    }
    PG_BAR();                                                 // the other group's last MFMA block
  } else {
    PG_BAR();                                                 // interval 0: the other group reads stage 0
    for (int k = 0; k < nk; k += 2) {
      load_block(k, ra1);
      wait_next(k);
      frag_wait();
      PG_BAR();
      mfma_block();
      PG_BAR();
      if (k + 1 < nk) {
        load_block(k + 1, ra0);
        wait_next(k + 1);
        frag_wait();
        PG_BAR();
        mfma_block();
        PG_BAR();
      }
    }
  }

  wait_vm<0>();                 // the redundant tail requests: nothing invisible to the compiler may be outstanding from here on
  // ---- the token statistics: every K-step of a token went through exactly one conversion of each of its four lanes
  s1 += __shfl_xor(s1, 1, 64); s1 += __shfl_xor(s1, 2, 64);
  s2 += __shfl_xor(s2, 1, 64); s2 += __shfl_xor(s2, 2, 64);
  if (j0 == 0) {
    const float inv_f = 1.0f / (float)g.F;
    const float mu = s1 * inv_f;
    const float var = fmaxf(s2 * inv_f - mu * mu, 0.f);
    const float rs = rsqrtf(var + a.eps);
    stat[2 * m] = mu; stat[2 * m + 1] = rs;
    if (row0 + m < a.M) *(float4*)(a.tstat + 4L * (row0 + m)) = make_float4(cb, mu, rs, cb + mu);
  }
  __syncthreads();

  // ---- epilogue: z = rstd (acc - mu' S) + b', f32, whole 128-byte lines (gemm_tile.h line_pair)
  const bool upper = ml >= 8;
  const int rsub = ml & 7, hsel = ml >> 3;
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int colw = wn * 128 + 64 * s;
    float4 S4[4], b4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      S4[j] = *(const float4*)(a.S + colw + 16 * j + 4 * q4);
      b4[j] = *(const float4*)(a.bias + colw + 16 * j + 4 * q4);
    }
#pragma unroll
    for (int i = 0; i < IM; ++i) {
      __builtin_amdgcn_sched_barrier(0);
      const int rl = wm * 64 + i * 16 + ml;
      const float mu = stat[2 * rl], rs = stat[2 * rl + 1];
      uint4 pk[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4 v = acc[i][4 * s + j];
        pk[j] = f4_bits(fmaf(rs, v[0] - mu * S4[j].x, b4[j].x), fmaf(rs, v[1] - mu * S4[j].y, b4[j].y),
                        fmaf(rs, v[2] - mu * S4[j].z, b4[j].z), fmaf(rs, v[3] - mu * S4[j].w, b4[j].w));
      }
#pragma unroll
      for (int mm = 0; mm < 2; ++mm) {
        uint4 s0, s1v;
        line_pair(pk[2 * mm], pk[2 * mm + 1], upper, s0, s1v);
        const int col = colw + 32 * mm + 16 * hsel + 4 * q4;
        const int row = row0 + wm * 64 + i * 16 + rsub;
        if (row < a.M) st16(a.Z + (long)row * a.ldz + col, s0);
        if (row + 8 < a.M) st16(a.Z + (long)(row + 8) * a.ldz + col, s1v);
      }
    }
  }
}


}  // namespace pg

// ------------------------------------------------------------------------------------------------------------------
// The tubelet projection's weight gradient from the VOLUME:  G[n][f] = sum_tokens dz[tok][n] xhat[tok][f]  (f32 [N, F]; d(W),
// d(gamma), d(beta) follow from it in ctclip_patch_affine_bwd) with xhat recomputed instead of read back.
// gemm4.hip's transposed-operand tile and role-alternating loop, K = 32 tokens per step, split over the tokens, partial tiles to
// the split-K workspace -- but the tile is ALL 512 output rows x 128 features, so that every voxel is fetched and centred once:
//   * dz [tokens, 512] bf16 goes through the LDS-DMA ring as gemm4's m-major operand (two 256-column sub-tiles, 32 KiB per K-step);
//     the 32 feature windows of one token range are the 32 workgroups of one XCD (xcd_remap), which share that stream in its L2;
//   * the feature operand's [32 tokens][128 features] tile is written by the load block from registers: thread (token kk = tid /
//     16, pieces pl and pl + 16) loads its two 8-byte pieces and the token's statistics one K-step ahead, computes
//     bf16((x - c) rstd) and writes it where the transposing fragment reads expect it (gemm_tile_t.h addressing, 256-byte rows).
// (x - c) rstd = xhat + mu' rstd, so the product carries one extra term  sum_tok dz[tok][n] (mu' rstd)[tok]  in every column:
// it is accumulated by the product itself in two VIRTUAL feature columns F, F + 1 that hold mu' rstd split into a high and a low
// bf16 part (exact to 2^-17), and ctclip_patch_affine_bwd subtracts their sum from every G[n][f].  For a constant tubelet both
// x - c and mu' are exactly 0: it contributes nothing, as in the reference.  Tokens % 32 == 0, N == 512.
// vmcnt: block j = convert R(j + 1) | request R(j + 2): 3 loads | request D(j + 2): 4 DMA pieces -- always (past the last K-step
// the last one is requested again into registers / a slot nobody reads), so every wait is a constant: R(j + 1) has landed when at
// most D(j + 1)'s 4 pieces are outstanding, D(k + 1) when at most the 7 instructions of block k are.
namespace pgw {
using namespace g4;
using pg::Geom;

constexpr int WN = 512, WF = 128, WNS = 3;
constexpr int WB = 32 * WF * 2;                      // the feature tile of a stage: 32 token rows x 256 bytes
constexpr int WSTAGE = 2 * SUB + WB;                 // 40 KiB

struct WgArgs {
  const bf16_t* vol; const bf16_t* dz; long lddz;
  const float* tstat;
  float* part;                 // [splits][512][ldg] partial tiles (plain stores), summed in split order afterwards
  long ldg;
  int Mtok, Fext, tiles_f, split_k, ktiles_per_split;
  unsigned m_wt, m_ht, m_tt;   // multiply-high divisors of Wt, Ht, Tt
  Geom g;
};

__device__ __forceinline__ int mdiv(int x, unsigned m) { return m ? (int)__umulhi((unsigned)x, m) : x; }
__device__ __forceinline__ long token_origin_fast(const WgArgs& a, int tok) {
  const Geom& g = a.g;
  int r = mdiv(tok, a.m_wt); const int w = tok - r * g.Wt;
  int r2 = mdiv(r, a.m_ht); const int h = r - r2 * g.Ht;
  const int b = mdiv(r2, a.m_tt); const int t = r2 - b * g.Tt;
  return (((long)b * g.C * g.Dz + (long)t * g.pt) * g.Hy + (long)h * g.p) * g.Wx + (long)w * g.p;
}
// the feature tile: [32 token rows][128 columns] bf16, 256-byte rows of 16 16-byte chunks, gemm4's XOR on the chunk index
__device__ __forceinline__ uint32_t ftile_off(int k, int chunk) { return (uint32_t)(k * 256) + ((((uint32_t)chunk) ^ (swz(k) & 15u)) << 4); }
__device__ __forceinline__ void read_frag_f16(uint32_t tile_lds, int cbase, int lane, short4v& lo, short4v& hi) {
  const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, p = i16 & 3;
  const int c0 = cbase + 4 * p;
  const int klo = 8 * g + q;
  const uint32_t sub = (uint32_t)((p & 1) * 8);
  lo = tr_read_asm(tile_lds + ftile_off(klo, c0 >> 3) + sub);
  hi = tr_read_asm(tile_lds + ftile_off(klo + 4, c0 >> 3) + sub);
}

__global__ __launch_bounds__(512, 2) void patch_wgrad_kernel(WgArgs a) {
  constexpr int OPS = 7;                                       // vm instructions per block: 2 pieces + 1 statistics load + 4 DMA pieces
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;                     // 4 x 2 waves: 128 rows x 64 features each
  const int grp = wave >> 2;                                   // role group: SIMD partners are waves w and w + 4
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  const Geom& g = a.g;

  // the feature windows of one split are consecutive logical ids: one XCD's workgroups share the dz stream of that token range
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tf = bid % a.tiles_f, ks = bid / a.tiles_f;
  const int col0 = tf * WF;
  const int nk_total = a.Mtok / BK;
  const int kt_begin = ks * a.ktiles_per_split;
  const int nk = min(nk_total, kt_begin + a.ktiles_per_split) - kt_begin;

  // ---- dz through the ring: piece q = 4 wave + j of a stage: sub-tile q / 16 (columns 256 (q / 16) ..), k-rows 2 (q % 16), + 1
  const bf16_t* srcA[4];
  uint32_t dstA[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int q = wave * 4 + j;
    srcA[j] = a.dz + piece_src(q & 15, lane, 256 * (q >> 4), WN, a.lddz) + (long)kt_begin * BK * a.lddz;
    dstA[j] = (uint32_t)((q >> 4) * SUB + (q & 15) * 1024);
  }
  auto issue_D = [&](int t) {
    const int tt = t < nk ? t : nk - 1;
    const uint32_t sb = lds0 + (uint32_t)((t % WNS) * WSTAGE);
#pragma unroll
    for (int j = 0; j < 4; ++j) G4_GLDS(srcA[j] + (long)tt * BK * a.lddz, sb + dstA[j]);
  };

  // ---- the feature operand: this thread's token row kk and its two pieces (fixed for the whole kernel)
  const int kk = tid >> 4, pl = tid & 15;
  int po[2];
  uint32_t wo[2];
  bool special = false;                                        // the window holds virtual / padding columns (the last one or two)
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int col = 4 * (pl + 16 * i), f = col0 + col;
    po[i] = pg::piece_offset(g, f < g.F ? f : 0);
    wo[i] = 2 * SUB + ftile_off(kk, col >> 3) + ((col & 4) ? 8u : 0u);
  }
  special = col0 + WF > g.F;
  uint2 ra[2];
  f32x4 st;                                                    // c, mu', rstd, mean of the token in flight
  auto load_R = [&](int t) {                                   // K-step t of this split: pieces + statistics -> registers
    const int tt = t < nk ? t : nk - 1;
    const int tok = (kt_begin + tt) * BK + kk;
    const bf16_t* tb = a.vol + token_origin_fast(a, tok);
#pragma unroll
    for (int i = 0; i < 2; ++i) ra[i] = g3::glb_rd64(tb + po[i]);
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(st) : "v"(a.tstat + 4L * tok) : "memory");
  };
  auto convert = [&](int t) {                                  // registers -> bf16((x - c) rstd) in stage t % WNS
    const uint32_t base = lds0 + (uint32_t)((t % WNS) * WSTAGE);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      float d[4];
      pg::unpack4(ra[i], st[0], d);
      uint2 o = make_uint2(pack_bf16x2(d[0] * st[2], d[1] * st[2]), pack_bf16x2(d[2] * st[2], d[3] * st[2]));
      if (special) {                                           // uniform over the workgroup
        const int f = col0 + 4 * (pl + 16 * i);
        if (f >= g.F) {
          const float v = st[1] * st[2], hi = bf16_to_f32(f32_to_bf16(v));
          o = f == g.F ? make_uint2(pack_bf16x2(hi, v - hi), 0u) : make_uint2(0u, 0u);
        }
      }
      g3::lds_wr64(base + wo[i], o);
    }
  };
  auto wait_R = [&]() { asm volatile("s_waitcnt vmcnt(4)" : "+v"(ra[0]), "+v"(ra[1]), "+v"(st) : : "memory"); };

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;

  // ---- prologue: D(0) | R(0) -> stage 0 | R(1) | D(1)
  issue_D(0);
  load_R(0);
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(ra[0]), "+v"(ra[1]), "+v"(st) : : "memory");
  convert(0);
  load_R(1);
  issue_D(1);
  wait_R();                                                    // R(1) landed: the compiler may copy these registers where the loops begin

  short4v r0[12], r1[12];
  const int asub = (wm >> 1) * SUB, arow = (wm & 1) * 128;
  auto read_set = [&](short4v (&raw)[12], uint32_t sa_l, int s) {
#pragma unroll
    for (int i = 0; i < 4; ++i) read_frag_tr16(sa_l + asub, arow + (4 * s + i) * 16, lane, raw[2 * i], raw[2 * i + 1]);
#pragma unroll
    for (int j = 0; j < 2; ++j) read_frag_f16(sa_l + 2 * SUB, wn * 64 + (2 * s + j) * 16, lane, raw[8 + 2 * j], raw[9 + 2 * j]);
  };
  auto load_block = [&](int k) {
    const uint32_t sa_l = lds0 + (uint32_t)((k % WNS) * WSTAGE);
    read_set(r0, sa_l, 0);
    read_set(r1, sa_l, 1);
    wait_R();                                                  // R(k + 1): behind it only D(k + 1)
    convert(k + 1);
    load_R(k + 2);
    issue_D(k + 2);
  };
  auto mfma_block = [&]() {
    __builtin_amdgcn_s_setprio(1);
    bf16x8 fa[8], fb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { fa[i] = join_tr(r0[2 * i], r0[2 * i + 1]); fa[4 + i] = join_tr(r1[2 * i], r1[2 * i + 1]); }
#pragma unroll
    for (int j = 0; j < 2; ++j) { fb[j] = join_tr(r0[8 + 2 * j], r0[9 + 2 * j]); fb[2 + j] = join_tr(r1[8 + 2 * j], r1[9 + 2 * j]); }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };
  auto lgkm0 = [&]() { tr_wait(r0); tr_wait(r1); };

  // D(0) and the feature tile of K-step 0 are in place for everybody (the prologue's vmcnt(0) covered D(0))
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  PG_BAR();
  if (grp == 0) {
    for (int k = 0; k < nk; ++k) {
      load_block(k);
      lgkm0();
      PG_BAR();
      mfma_block();
      g3::wait_vm<OPS>();                                      // D(k + 1): behind it block k
      PG_BAR();
    }
    PG_BAR();
  } else {
    PG_BAR();
    for (int k = 0; k < nk; ++k) {
      load_block(k);
      g3::wait_vm<OPS>();
      lgkm0();
      PG_BAR();
      mfma_block();
      PG_BAR();
    }
  }
  g3::wait_vm<0>();                                            // nothing may be in flight into LDS or registers when the wave ends

  // ---- partial tile [512 n][128 features] of split ks -> workspace (16 x 16 tiles: a register covers four 64-byte row segments)
  const int q4 = lane >> 4, ml = lane & 15;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = col0 + wn * 64 + j * 16 + ml;
      if (col >= a.Fext) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = wm * 128 + i * 16 + 4 * q4 + r;
        a.part[((long)ks * WN + row) * a.ldg + col] = acc[i][j][r];
      }
    }
}

}  // namespace pgw

namespace {
bool pg_geom(pg::Geom& g, int C, int Dz, int Hy, int Wx, int pt, int p) {
  if (C <= 0 || pt <= 0 || p <= 0 || Dz % pt || Hy % p || Wx % p || (p & 3) || (Wx & 3)) return false;
  g.C = C; g.Dz = Dz; g.Hy = Hy; g.Wx = Wx; g.pt = pt; g.p = p;
  g.Tt = Dz / pt; g.Ht = Hy / p; g.Wt = Wx / p;
  const long F = (long)C * pt * p * p;
  if ((F % pg::BK) || F < 4 * pg::BK || F > pg::MAXF || (long)C * Dz * Hy * Wx >= (1L << 31)) return false;
  g.F = (int)F;
  return true;
}
}  // namespace

extern "C" {

// See include/ctclip_hip.h.
int ctclip_patch_embed_fused(const void* volume_bf16, const void* Wfold_bf16, long ldw, const float* wsum, const float* bias_folded,
                             float* Z, long ldz, float* tstat, int B, int C, int Dz, int Hy, int Wx, int pt, int p, int N,
                             float eps, void* stream) {
  pg::FwdArgs a{};
  if (B <= 0) return 0;
  if (!pg_geom(a.g, C, Dz, Hy, Wx, pt, p) || N != pg::PBN || !volume_bf16 || !Wfold_bf16 || !wsum || !bias_folded || !Z ||
      !tstat || (((uintptr_t)tstat) & 15) || (((uintptr_t)volume_bf16) & 15) || (((uintptr_t)Wfold_bf16) & 15) || (ldw & 7) || ldw < a.g.F ||
      (((uintptr_t)Z) & 15) || (ldz & 3) || ldz < N || (((uintptr_t)wsum) & 15) || (((uintptr_t)bias_folded) & 15))
    return (int)hipErrorInvalidValue;
  const long M = (long)B * a.g.Tt * a.g.Ht * a.g.Wt;
  if (M >= (1L << 31)) return (int)hipErrorInvalidValue;
  a.vol = (const bf16_t*)volume_bf16; a.W = (const bf16_t*)Wfold_bf16; a.ldw = ldw; a.S = wsum; a.bias = bias_folded;
  a.Z = Z; a.ldz = ldz; a.tstat = tstat; a.M = (int)M; a.eps = eps;
  CTCLIP_LDS_LIMIT_ONCE(pg::patch_gemm_fwd_kernel, pg::LDS_BYTES);
  hipLaunchKernelGGL(pg::patch_gemm_fwd_kernel, dim3((unsigned)((M + pg::PBM - 1) / pg::PBM)), dim3(512), pg::LDS_BYTES,
                     (hipStream_t)stream, a);
  CTCLIP_CHECK_LAUNCH();
}

// See include/ctclip_hip.h.
int ctclip_patch_wgrad_fused(const void* volume_bf16, const void* dz_bf16, long lddz, const float* tstat, float* G, long ldg, int B,
                             int C, int Dz, int Hy, int Wx, int pt, int p, int N, float* splitk_ws, long splitk_ws_floats,
                             void* stream) {
  using namespace g4;
  pgw::WgArgs a{};
  if (B <= 0) return 0;
  if (!pg_geom(a.g, C, Dz, Hy, Wx, pt, p) || N != pgw::WN || !volume_bf16 || !dz_bf16 || !tstat || !G || !splitk_ws ||
      (((uintptr_t)volume_bf16) & 15) || (((uintptr_t)dz_bf16) & 15) || (lddz & 7) || lddz < N || (((uintptr_t)tstat) & 15) ||
      ldg != a.g.F + 2 || (long)N * ldg >= (1L << 31))
    return (int)hipErrorInvalidValue;
  const long M = (long)B * a.g.Tt * a.g.Ht * a.g.Wt;
  if (M >= (1L << 31) || (M % BK)) return (int)hipErrorInvalidValue;
  a.vol = (const bf16_t*)volume_bf16; a.dz = (const bf16_t*)dz_bf16; a.lddz = lddz; a.tstat = tstat; a.ldg = ldg;
  a.Mtok = (int)M; a.Fext = a.g.F + 2; a.tiles_f = (a.Fext + pgw::WF - 1) / pgw::WF;
  auto magic = [](unsigned d) { return d <= 1u ? 0u : (unsigned)(0x100000000ull / d) + 1u; };
  a.m_wt = magic((unsigned)a.g.Wt); a.m_ht = magic((unsigned)a.g.Ht); a.m_tt = magic((unsigned)a.g.Tt);
  // one workgroup per CU: split the tokens so that windows x splits ~ the CU count, at least 8 K-steps per split, and no more
  // partial tiles than the workspace holds
  const int nk = (int)(M / BK);
  int split = ctclip_cu_count8() / a.tiles_f;
  if (split < 1) split = 1;
  if (split > nk / 8) split = nk / 8 < 1 ? 1 : nk / 8;
  const long per = (long)N * ldg;
  if ((long)split * per > splitk_ws_floats) split = (int)(splitk_ws_floats / per);
  if (split < 1) return (int)hipErrorInvalidValue;
  a.ktiles_per_split = (nk + split - 1) / split;
  a.split_k = (nk + a.ktiles_per_split - 1) / a.ktiles_per_split;
  a.part = splitk_ws;
  const size_t lds = (size_t)pgw::WNS * pgw::WSTAGE;
  CTCLIP_LDS_LIMIT_ONCE(pgw::patch_wgrad_kernel, lds);
  hipLaunchKernelGGL(pgw::patch_wgrad_kernel, dim3((unsigned)(a.tiles_f * a.split_k)), dim3(NT), lds, (hipStream_t)stream, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return (int)e;
  // G += the partial tiles in split order (reproducible).  ldg == F + 2 is required so that every element of a partial tile is
  // written (the sum runs over whole rows)
  return ctclip_reduce_partials(splitk_ws, a.split_k, per, (int)per, G, (hipStream_t)stream);
}

}  // extern "C"
