// Shared by the k-major x k-major LDS-DMA GEMM kernels (gemm3.hip: 8 waves x 128 x 64; gemm5.hip: 4 waves x 128 x 128): the
// k32 LDS tile layout with its source-side swizzle, the permuted N-fragment rows, and the REGISTER EPILOGUE of one 64-column
// slab of a wave's accumulators (bias / residual / erf-GELU, FF1 + GEGLU, FF2 data gradient + GEGLU backward, head-major) --
// see the comments at each piece.  Reference: every nn.Linear of src/utils/attention.py:38-51,118-124 and its autograd.
#pragma once
#include "common.h"

namespace g3 {

constexpr int BM = 256, BK = 32;
constexpr int SUB = 16384;                 // the A tile of a stage: 256 x 32 bf16 (the B tile follows it)

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

struct Args {
  const bf16_t* A; const bf16_t* B; void* C; const float* bias; const float* resid;
  long lda, ldb, ldc, ldr;
  int M, N, K, tiles_m, tiles_n, act;
  float alpha;
  bf16_t* G; long ldg;                             // EPI 2: gelu(gate) * value of the 32-column interleaved [val|gate] blocks
                                                   // EPI 3: G = h (pre-activations, same blocks), overwritten with d(h)
  int direct;                                      // pointers / strides allow the 16-byte register epilogue
  // EPI 0, head-major output (hm_n > 0): column c = (part, head, d) with d = c % 32, head = (c / 32) % hm_heads, part =
  // c / (32 hm_heads); row R = (sequence, token) with token = R % hm_n: element (R, c) goes to
  //   C + part * hm_part + ((sequence * hm_heads + head) * hm_n + token) * 32 + d
  // i.e. [part][sequence][head][token][32] -- the operand layout of attention_hm.hip.  hm_magic = ceil(2^32 / hm_n).
  int hm_n, hm_heads;
  uint32_t hm_magic;
  long hm_part;
  // EPI 4 (LayerNorm backward applied to the product, f32 out): C = alpha A B^T - c1[row] - xhat[row][col] c2[row] + resid, with an
  // optional bf16 copy of C in C16 -- the row constants come from the producer of A (ctclip_headnorm_bwd_ln)
  const bf16_t* xhat; long ldx;
  const float* c1; const float* c2;
  bf16_t* C16; long ldc16;
  // EPI 5 (per-head cosine normalisation of the product, bf16 out, row-major or head-major): the heads of the first hn_cols
  // columns (32 columns each) leave as  y = x / max(|x|, 1e-12) * hn_scale[d] * hn_mult  with 1 / max(|x|, 1e-12) stored in
  // hn_inv[row][head] (hn_cols / 32 heads per row); the columns behind them are stored as they are (the v half of a kv product)
  const float* hn_scale; float hn_mult; float* hn_inv; int hn_cols;
};

// [256 rows][32 k] bf16 tile, 64-byte rows, 16-byte chunks XOR-swizzled so that the 16 lanes of a ds_read_b128 phase hit 16
// different bank groups.  Key = (-(r >> 2)) & 3 serves every fragment shape used on this ring: a 32x32x16 read (32
// consecutive rows, one chunk per lane half: vq_topk3 below) only needs the four row quads of a 16-lane phase on different
// keys; a 16x16x32 read (16 rows x all four chunks, chunk = lane >> 4) puts row quads {0, 3} with chunk c and {1, 2} with
// chunk c ^ 1 in one phase, and {k0, k3, 1 ^ k1, 1 ^ k2} = {0, 1, 2, 3} for this key (the plain key r >> 2 collides); the
// permuted N-fragment rows of nfrag_row() below land on {0, 2, 3, 1} / {3, 1, 0, 2}.
__device__ __forceinline__ int swz_key(int r) { return (-(r >> 2)) & 3; }
__device__ __forceinline__ uint32_t tile_off(int r, int chunk) { return (uint32_t)(r * 64 + ((chunk ^ swz_key(r)) << 4)); }

__device__ __forceinline__ bf16x8 read_frag(const char* tile, int rbase, int s, int lane) {
  const int r = rbase + (lane & 31);
  return *(const bf16x8*)(tile + tile_off(r, 2 * s + (lane >> 5)));
}

__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// A wave's slab is IM (i: 16 rows) x 4 (j) MFMA 16x16x32 products issued as D' = Nfrag_j * Mfrag_i^T, so lane l
// (q = l >> 4, ml = l & 15) holds D'[4q + r][ml] = C[16 i + ml][slab column of N-fragment row 4q + r], r = 0..3.
// bf16 outputs: N-fragment j takes the slab's rows   32 (j >> 1) + 8 (a >> 2) + 4 (j & 1) + (a & 3),   a = 0..15,
// so that lane's 16 values of row 16 i + ml are the slab columns  8q .. 8q+7  (j = 0, 1)  and  32 + 8q .. 32 + 8q+7  (j = 2, 3):
// two 16-byte bf16 vectors, and in the GEGLU layouts ([val 32 | gate 32] blocks) a value and its gate.
// f32 outputs use the identity (N-fragment j = slab rows 16 j .. 16 j + 15): the lane's four values of one MFMA are then 4
// contiguous f32 = one 16-byte store, and the four lanes of a row write 64 contiguous bytes per store instruction (with the
// permutation an f32 row would be written in 16-byte pieces 32 bytes apart: measured 2x slower).
__device__ __forceinline__ int nfrag_row(int j, int a) { return 32 * (j >> 1) + 8 * (a >> 2) + 4 * (j & 1) + (a & 3); }

// element offset (from the operand base, at k-tile 0) of the 16 bytes lane `lane` contributes to 1 KiB piece `p`
// (rows 16p .. 16p+15, 4 chunks each) of the tile whose first row is r0; its LDS destination is piece_base + lane*16
__device__ __forceinline__ long piece_src(int p, int lane, int r0, int R, long ld) {
  const int r = 16 * p + (lane >> 2), pc = lane & 3, c = pc ^ swz_key(r);
  int row = r0 + r;
  if (row >= R) row = R - 1;                       // masked in the epilogue
  return (long)row * ld + c * 8;
}

__device__ __forceinline__ float gelu_erf(float x) { return gelu_erf_fast(x); }

// streaming 16-byte stores of the epilogue (gigabytes written once, read by a later kernel): non-temporal, +3..6 % on the
// K = 512 shapes against plain stores
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
__device__ __forceinline__ void st16(void* p, uint4 v) {
  const u32x4_t x = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(x, (u32x4_t*)p);
}
__device__ __forceinline__ void st16f(void* p, float a, float b, float c, float d) {
  const f32x4_t x = {a, b, c, d};
  __builtin_nontemporal_store(x, (f32x4_t*)p);
}
__device__ __forceinline__ uint4 pack8(const float* v) {
  uint4 o;
  o.x = pack_bf16x2(v[0], v[1]); o.y = pack_bf16x2(v[2], v[3]); o.z = pack_bf16x2(v[4], v[5]); o.w = pack_bf16x2(v[6], v[7]);
  return o;
}
__device__ __forceinline__ void unpack8(uint4 w, float* v) {
  const uint32_t u[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
  for (int e = 0; e < 4; ++e) { v[2 * e] = __uint_as_float(u[e] << 16); v[2 * e + 1] = __uint_as_float(u[e] & 0xffff0000u); }
}

// Whole-line stores.  The MFMA layout leaves lane (q4, ml) with two 16-byte pieces of ROW ml, 64 bytes apart (the two
// halves of one 128-byte line), so a store instruction of piece 0 covers 16 rows x 64 bytes and a second one completes the
// lines.  Swapping piece 1 of lanes ml < 8 with piece 0 of lanes ml >= 8 (a DPP row rotate by 8, no LDS) makes the first
// instruction cover rows 0-7 x 128 bytes and the second rows 8-15: measured +5..11 % on the K = 512 products.
// After the call: s0 belongs to row (ml & 7), s1 to row 8 + (ml & 7), both at the half-line selected by ml >> 3.
__device__ __forceinline__ uint32_t ror8(uint32_t x) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x128, 0xf, 0xf, false); }
__device__ __forceinline__ void line_pair(uint4 p0, uint4 p1, bool upper, uint4& s0, uint4& s1) {
  const uint4 send = upper ? p0 : p1;
  uint4 got;
  got.x = ror8(send.x); got.y = ror8(send.y); got.z = ror8(send.z); got.w = ror8(send.w);
  s0 = upper ? got : p0;
  s1 = upper ? p1 : got;
}
__device__ __forceinline__ uint4 f4_bits(float a, float b, float c, float d) {
  return make_uint4(__float_as_uint(a), __float_as_uint(b), __float_as_uint(c), __float_as_uint(d));
}

// x + (x of lanes ^ 16) + (x of lanes ^ 32) + (x of lanes ^ 48): the four lanes (q4 = 0..3) that hold one row's pieces of a
// 32-column head.  v_permlane16_swap exchanges the odd 16-lane rows of one register with the even rows of another, so with both
// operands = x the two results are [r0 r0 r2 r2] and [r1 r1 r3 r3]; v_permlane32_swap does the same for the wave's halves.
__device__ __forceinline__ float sum_over_q4(float x) {
  const uint32_t u = __float_as_uint(x);
  const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  const float s = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  const uint32_t v = __float_as_uint(s);
  const auto b = __builtin_amdgcn_permlane32_swap(v, v, false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// LDS and global accesses as INLINE ASM, for loops whose loads are consumed one or two K-steps after they were issued
// (patch_gemm.hip): a plain C++ load that is loop-carried is, for the compiler's wait-count pass, "a load of unknown age", and it
// waits for it with `s_waitcnt vmcnt(0)` -- which drains the LDS-DMA ring that shares the counter.  An asm access is invisible to
// that pass; what orders it is the loop's own counted wait, and the results are only valid behind a wait that TIES the registers
// (lds_wait / vm_wait_tied: the consumers then cannot be scheduled above it).
template <int OFF>
__device__ __forceinline__ bf16x8 lds_rd128(uint32_t addr) {
  bf16x8 r;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
  return r;
}
__device__ __forceinline__ int lds_rd32(uint32_t addr) {
  int r;
  asm volatile("ds_read_b32 %0, %1" : "=v"(r) : "v"(addr));
  return r;
}
// NOTE for every user: the compiler believes the result is valid the moment the asm has been issued.  Nothing may touch it before
// the tied wait -- and that includes register COPIES the allocator inserts on its own (at a loop's entry, around a second asm's
// tied operands).  After every change of a kernel that keeps such registers in flight, read its ISA: no v_mov of them between the
// load and the wait (patch_gemm.hip states what was checked).
__device__ __forceinline__ uint2 glb_rd64(const void* p) {
  uint2 r;
  asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(r) : "v"(p) : "memory");
  return r;
}
// s_waitcnt vmcnt(N) that the two registers pass through: their consumers stay behind it
template <int N>
__device__ __forceinline__ void vm_wait_tied(uint2& a, uint2& b) {
  asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N) : "memory");
}
__device__ __forceinline__ void lgkm_wait_tied(int& a, int& b) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b) : : "memory"); }
__device__ __forceinline__ void lds_wr64(uint32_t addr, uint2 v) {
  asm volatile("ds_write_b64 %0, %1" : : "v"(addr), "v"(v) : "memory");
}
template <int NA, int NB>
__device__ __forceinline__ void lds_wait(bf16x8 (&a)[NA], bf16x8 (&b)[NB]) {
  static_assert(NA + NB <= 14, "operand count of one asm statement");
  if constexpr (NA == 8 && NB == 4)
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]),
                 "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]) : : "memory");
  else if constexpr (NA == 4 && NB == 4)
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3])
                 : : "memory");
  else if constexpr (NA == 4 && NB == 8)
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]),
                 "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]) : : "memory");
  else
    static_assert(NA == 8 && NB == 4, "add the fragment-count combination");
}

#define G3_GLDS(gptr, ldsoff)                                                                                     \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),                          \
                                   (__attribute__((address_space(3))) void*)(uintptr_t)(ldsoff), 16, 0, 0)

// The epilogue of ONE 64-column slab: accumulators acc[i][J0 .. J0 + 3] (i = 0 .. IM - 1: 16-row groups starting at row
// `rbase`; the slab's first column is `colw`) in the register layout described at nfrag_row().  AHEAD: what the epilogue
// READS (residual, h) for row group i + 1 is requested before row group i is worked on (needs a second buffer).
// GENERIC = false leaves out the element-wise path for unaligned outputs (the caller then only takes g.direct problems).
// EPI: 0 plain -> bf16   1 plain -> f32   (both: alpha, bias, residual, erf-GELU when g.act == 1)   2 FF1 + GEGLU
//      3 FF2 data gradient + GEGLU backward   4 the f32 form with the LayerNorm backward applied (Args::xhat, c1, c2)
//      5 bf16 with the per-head cosine normalisation applied (Args::hn_*; attention.py:146-153)
template <int EPI, int IM, int NJ, int J0, bool AHEAD, bool GENERIC = true>
__device__ __forceinline__ void epilogue_slab(const Args& g, f32x4 (&acc)[IM][NJ], int rbase, int colw, int lane) {
  constexpr bool F32OUT = EPI == 1 || EPI == 4, LNB = EPI == 4;
  const int ml = lane & 15, q4 = lane >> 4;
  const int act = EPI >= 2 ? EPI : g.act;
  if (g.direct) {
    // ---- register epilogue: lane (q4, ml) owns, for each i, row 16 i + ml and (bf16) the slab columns 8 q4 .. +7 and
    //      32 + 8 q4 .. +7, (f32) 16 j + 4 q4 .. + 3.  Stores go out as whole lines (line_pair above); what the epilogue
    //      READS (residual, h) for row group i + 1 is requested before row group i is worked on.
    const bool upper = ml >= 8;
    const int rsub = ml & 7, hsel = ml >> 3;
        if constexpr (F32OUT) {
      // ring of row groups whose reads are in flight: three ahead of the one being worked on (round 5: 819 -> 805 us on FF2 + residual at
      // 32 pairs against one ahead; profiles/r05_epilogue_read_ahead.txt)
      constexpr int RD = AHEAD ? 4 : 1;
      float4 rs[RD][4];
      auto ld_resid = [&](int i, float4 (&dst)[4]) {
        const int row = rbase + i * 16 + ml;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int col = colw + 16 * j + 4 * q4;
          dst[j] = (g.resid && row < g.M && col < g.N) ? *(const float4*)(g.resid + (long)row * g.ldr + col) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      };
      if (AHEAD) {
#pragma unroll
        for (int d = 0; d < RD - 1; ++d)
          if (d < IM) ld_resid(d, rs[d]);
      }
#pragma unroll
      for (int i = 0; i < IM; ++i) {
        __builtin_amdgcn_sched_barrier(0);         // one row group at a time
        if (AHEAD) { if (i + RD - 1 < IM) ld_resid(i + RD - 1, rs[(i + RD - 1) % RD]); }
        else ld_resid(i, rs[0]);
        uint4 pk[4];
        float k1 = 0.f, k2 = 0.f;                    // EPI 4: this lane's row constants
        if constexpr (LNB) {
          const int rowl = rbase + i * 16 + ml;
          if (rowl < g.M) { k1 = g.c1[rowl]; k2 = g.c2[rowl]; }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int col = colw + 16 * j + 4 * q4;
          float x[4] = {acc[i][J0 + j][0] * g.alpha, acc[i][J0 + j][1] * g.alpha, acc[i][J0 + j][2] * g.alpha, acc[i][J0 + j][3] * g.alpha};
          if (g.bias && col < g.N) { const float4 bb = *(const float4*)(g.bias + col); x[0] += bb.x; x[1] += bb.y; x[2] += bb.z; x[3] += bb.w; }
          if constexpr (LNB) {
            const int rowl = rbase + i * 16 + ml;
            uint2 xw = make_uint2(0u, 0u);
            if (rowl < g.M && col < g.N) xw = *(const uint2*)(g.xhat + (long)rowl * g.ldx + col);
            x[0] -= fmaf(__uint_as_float(xw.x << 16), k2, k1);
            x[1] -= fmaf(__uint_as_float(xw.x & 0xffff0000u), k2, k1);
            x[2] -= fmaf(__uint_as_float(xw.y << 16), k2, k1);
            x[3] -= fmaf(__uint_as_float(xw.y & 0xffff0000u), k2, k1);
          }
          const float4 r = rs[i % RD][j];
          x[0] += r.x; x[1] += r.y; x[2] += r.z; x[3] += r.w;
          if (act == 1) { x[0] = gelu_erf(x[0]); x[1] = gelu_erf(x[1]); x[2] = gelu_erf(x[2]); x[3] = gelu_erf(x[3]); }
          pk[j] = f4_bits(x[0], x[1], x[2], x[3]);
          if constexpr (LNB) {
            const int rowl = rbase + i * 16 + ml;
            if (g.C16 && rowl < g.M && col < g.N)
              *(uint2*)(g.C16 + (long)rowl * g.ldc16 + col) = make_uint2(pack_bf16x2(x[0], x[1]), pack_bf16x2(x[2], x[3]));
          }
        }
#pragma unroll
        for (int m = 0; m < 2; ++m) {              // columns 32 m .. 32 m + 31 of the slab: one 128-byte line per row
          uint4 s0, s1;
          line_pair(pk[2 * m], pk[2 * m + 1], upper, s0, s1);
          const int col = colw + 32 * m + 16 * hsel + 4 * q4;
          const int row = rbase + i * 16 + rsub;
          if (col < g.N) {
            if (row < g.M) st16((float*)g.C + (long)row * g.ldc + col, s0);
            if (row + 8 < g.M) st16((float*)g.C + (long)(row + 8) * g.ldc + col, s1);
          }
        }
      }
    } else if constexpr (EPI == 2) {
      // FF1 + GEGLU: the slab is one [val 32 | gate 32] block of h (one line per row); g = gelu(gate) * value
      // (attention.py:38-41) is a 64-byte piece per row and slab, stored as it stands
#pragma unroll
      for (int i = 0; i < IM; ++i) {
        __builtin_amdgcn_sched_barrier(0);
        float v[2][8];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int e = 0; e < 8; ++e) v[h][e] = acc[i][J0 + 2 * h + (e >> 2)][e & 3] * g.alpha;
        float w[8];
#pragma unroll
        for (int e = 0; e < 8; e += 2) {               // pairs: packed-f32 arithmetic (common.h)
          const f32x2 ge = gelu_erf_fast2(f32x2{v[1][e], v[1][e + 1]}) * f32x2{v[0][e], v[0][e + 1]};
          w[e] = ge.x; w[e + 1] = ge.y;
        }
        uint4 s0, s1;
        line_pair(pack8(v[0]), pack8(v[1]), upper, s0, s1);
        const int row = rbase + i * 16 + rsub;
        if (colw < g.N) {
          bf16_t* hp = (bf16_t*)g.C + (long)row * g.ldc + colw + 32 * hsel + 8 * q4;
          if (row < g.M) st16(hp, s0);
          if (row + 8 < g.M) st16(hp + 8 * g.ldc, s1);
          const int rowg = rbase + i * 16 + ml;
          if (rowg < g.M) st16(g.G + (long)rowg * g.ldg + (colw >> 1) + 8 * q4, pack8(w));
        }
      }
    } else if constexpr (EPI == 3) {
      // dg = dy W2 with the GEGLU backward: this tile's dg never goes to memory; the matching value / gate
      // pre-activations are read from h and replaced by their gradients in place.  Column c of dg: its value sits at
      // (c / 32) * 64 + c % 32 of h, the gate 32 further -- value and gate pieces of a lane are the two halves of one line
      constexpr int RD = AHEAD ? 3 : 1;              // two row groups of h in flight ahead (1472 -> 1455 us at 32 pairs against one)
      uint4 hv[RD][4];                               // [buffer][2 h + {value, gate}]
      auto ld_h = [&](int i, uint4 (&dst)[4]) {
        const int row = rbase + i * 16 + ml;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int c = colw + 32 * h + 8 * q4;
          const bf16_t* hp = g.G + (long)row * g.ldg + (long)(c >> 5) * 64 + (c & 31);
          const bool ok = row < g.M && c < g.N;
          dst[2 * h] = ok ? *(const uint4*)hp : make_uint4(0u, 0u, 0u, 0u);
          dst[2 * h + 1] = ok ? *(const uint4*)(hp + 32) : make_uint4(0u, 0u, 0u, 0u);
        }
      };
      if (AHEAD) {
#pragma unroll
        for (int d = 0; d < RD - 1; ++d)
          if (d < IM) ld_h(d, hv[d]);
      }
#pragma unroll
      for (int i = 0; i < IM; ++i) {
        __builtin_amdgcn_sched_barrier(0);
        if (AHEAD) { if (i + RD - 1 < IM) ld_h(i + RD - 1, hv[(i + RD - 1) % RD]); }
        else ld_h(i, hv[0]);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          float val[8], gate[8], dv[8], dt[8];
          unpack8(hv[i % RD][2 * h], val);
          unpack8(hv[i % RD][2 * h + 1], gate);
#pragma unroll
          for (int e = 0; e < 8; e += 2) {             // pairs: packed-f32 arithmetic, one Phi for gelu and its derivative
            const f32x2 dgv = f32x2{acc[i][J0 + 2 * h + (e >> 2)][e & 3], acc[i][J0 + 2 * h + (e >> 2)][(e & 3) + 1]} * pk_splat(g.alpha);
            f32x2 ge, gr;
            gelu_erf_both2(f32x2{gate[e], gate[e + 1]}, ge, gr);
            const f32x2 a = dgv * ge, b = (dgv * f32x2{val[e], val[e + 1]}) * gr;
            dv[e] = a.x; dv[e + 1] = a.y; dt[e] = b.x; dt[e + 1] = b.y;
          }
          uint4 s0, s1;
          line_pair(pack8(dv), pack8(dt), upper, s0, s1);
          const int c = colw + 32 * h + 8 * q4;
          const int row = rbase + i * 16 + rsub;
          if (c < g.N) {
            bf16_t* hp = g.G + (long)row * g.ldg + (long)(c >> 5) * 64 + 32 * hsel + (c & 31);
            if (row < g.M) st16(hp, s0);
            if (row + 8 < g.M) st16(hp + 8 * g.ldg, s1);
          }
        }
      }
    } else if constexpr (EPI == 5) {
      // q / k projection with the per-head cosine normalisation (attention.py:146-153): the slab is two heads; a lane holds 8 of
      // a head's 32 columns for each of its rows, the other 24 sit in the lanes 16, 32 and 48 further -- sum of squares over the
      // four, 1 / max(norm, 1e-12), the learned per-channel scale times `mult`, store.  Nothing of the raw projection is written.
      const bool norm = colw < g.hn_cols;            // uniform over the slab (hn_cols % 64 == 0)
      float sc[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) sc[e] = norm ? g.hn_scale[8 * q4 + e] * g.hn_mult : 1.f;
      const int hpr = g.hn_cols >> 5;                // normalised heads per row
#pragma unroll
      for (int i = 0; i < IM; ++i) {
        __builtin_amdgcn_sched_barrier(0);
        float v[2][8], inv[2] = {1.f, 1.f};
        const int rowl = rbase + i * 16 + ml;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          float ss = 0.f;
#pragma unroll
          for (int e = 0; e < 8; ++e) { v[h][e] = acc[i][J0 + 2 * h + (e >> 2)][e & 3] * g.alpha; ss = fmaf(v[h][e], v[h][e], ss); }
          if (norm) {
            ss = sum_over_q4(ss);
            inv[h] = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[h][e] *= inv[h] * sc[e];
          }
        }
        // lanes q4 = 0 / 1 of a row store the first / second head's 1 / norm: one store instruction per row group
        if (norm && q4 < 2 && rowl < g.M && colw + 32 * q4 < g.N) g.hn_inv[(long)rowl * hpr + ((colw >> 5) + q4)] = q4 ? inv[1] : inv[0];
        if (g.hm_n) {
          if (rowl < g.M) {
            const uint32_t sq = __umulhi((uint32_t)rowl, g.hm_magic), tok = (uint32_t)rowl - sq * (uint32_t)g.hm_n;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const int col = colw + 32 * h + 8 * q4;
              if (col < g.N) {
                const int vh = col >> 5, part = vh / g.hm_heads, hh = vh - part * g.hm_heads;
                st16((bf16_t*)g.C + (long)part * g.hm_part + ((long)(sq * g.hm_heads + hh) * g.hm_n + tok) * 32 + 8 * q4,
                     pack8(v[h]));
              }
            }
          }
        } else {
          uint4 s0, s1;
          line_pair(pack8(v[0]), pack8(v[1]), upper, s0, s1);
          const int col = colw + 32 * hsel + 8 * q4;
          const int row = rbase + i * 16 + rsub;
          if (col < g.N) {
            if (row < g.M) st16((bf16_t*)g.C + (long)row * g.ldc + col, s0);
            if (row + 8 < g.M) st16((bf16_t*)g.C + (long)(row + 8) * g.ldc + col, s1);
          }
        }
      }
    } else {
      float4 rs[1][4];                               // [2 h + half]
      auto ld_resid = [&](int i, float4 (&dst)[4]) {
        const int row = rbase + i * 16 + ml;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int col = colw + 32 * h + 8 * q4;
          const bool ok = g.resid && row < g.M && col < g.N;
          const float* rp = g.resid + (long)row * g.ldr + col;
          dst[2 * h] = ok ? *(const float4*)rp : make_float4(0.f, 0.f, 0.f, 0.f);
          dst[2 * h + 1] = ok ? *(const float4*)(rp + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      };
#pragma unroll
      for (int i = 0; i < IM; ++i) {
        __builtin_amdgcn_sched_barrier(0);
        // (the f32-output form is the one the residual products use.)  Requested AND consumed on every path -- zeros without a
        // residual: a load under `if (g.resid)` that is used under a second `if (g.resid)` is, for the compiler's wait-count
        // pass, possibly still pending when the persistent tile loop comes round, and it then drains the whole DMA ring
        // (`s_waitcnt vmcnt(0)`) in front of the first fragment register it overwrites -- at the top of EVERY K-step (found in round
        // 5 in gemm3_kernel<256, 0>'s ISA; interleaved A/B on q / kv / out-dgrad / FF2-dgrad shapes: no measurable difference,
        // profiles/r05_gemm3_epi0_drain.txt -- the requests of a K-step were issued a whole K-step earlier either way)
        ld_resid(i, rs[0]);
        float v[2][8];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[h][e] = acc[i][J0 + 2 * h + (e >> 2)][e & 3] * g.alpha;
          const int col = colw + 32 * h + 8 * q4;
          if (g.bias && col < g.N) {
            const float4 b0 = *(const float4*)(g.bias + col), b1 = *(const float4*)(g.bias + col + 4);
            v[h][0] += b0.x; v[h][1] += b0.y; v[h][2] += b0.z; v[h][3] += b0.w;
            v[h][4] += b1.x; v[h][5] += b1.y; v[h][6] += b1.z; v[h][7] += b1.w;
          }
          {
            const float4 r0 = rs[0][2 * h], r1 = rs[0][2 * h + 1];
            v[h][0] += r0.x; v[h][1] += r0.y; v[h][2] += r0.z; v[h][3] += r0.w;
            v[h][4] += r1.x; v[h][5] += r1.y; v[h][6] += r1.z; v[h][7] += r1.w;
          }
          if (act == 1) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[h][e] = gelu_erf(v[h][e]);
          }
        }
        if (g.hm_n) {
          // head-major: the lane's two 16-byte pieces belong to two heads of ITS row; the 16 lanes of a row group write 16
          // consecutive tokens x 64 bytes of one head = 1 KiB contiguous per store instruction -- whole lines as they stand
          const int rowl = rbase + i * 16 + ml;
          if (rowl < g.M) {
            const uint32_t sq = __umulhi((uint32_t)rowl, g.hm_magic), tok = (uint32_t)rowl - sq * (uint32_t)g.hm_n;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const int col = colw + 32 * h + 8 * q4;
              if (col < g.N) {
                const int vh = col >> 5, part = vh / g.hm_heads, hh = vh - part * g.hm_heads;
                st16((bf16_t*)g.C + (long)part * g.hm_part + ((long)(sq * g.hm_heads + hh) * g.hm_n + tok) * 32 + 8 * q4,
                     pack8(v[h]));
              }
            }
          }
        } else {
          uint4 s0, s1;
          line_pair(pack8(v[0]), pack8(v[1]), upper, s0, s1);
          const int col = colw + 32 * hsel + 8 * q4;
          const int row = rbase + i * 16 + rsub;
          if (col < g.N) {
            if (row < g.M) st16((bf16_t*)g.C + (long)row * g.ldc + col, s0);
            if (row + 8 < g.M) st16((bf16_t*)g.C + (long)(row + 8) * g.ldc + col, s1);
          }
        }
      }
    }
  } else if constexpr (EPI < 2 && GENERIC) {
    // ---- generic epilogue (unaligned pointers / strides, N % 8 != 0; EPI 0 / 1 only): element-wise from registers
#pragma unroll
    for (int i = 0; i < IM; ++i) {
      const int row = rbase + i * 16 + ml;
      if (row >= g.M) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int col = colw + (F32OUT ? 16 * j + 4 * q4 + r : 32 * (j >> 1) + 8 * q4 + 4 * (j & 1) + r);
          if (col >= g.N) continue;
          float x = acc[i][J0 + j][r] * g.alpha + (g.bias ? g.bias[col] : 0.f);
          if (g.resid) x += g.resid[(long)row * g.ldr + col];
          if (act == 1) x = gelu_erf(x);
          if (F32OUT) ((float*)g.C)[(long)row * g.ldc + col] = x;
          else ((bf16_t*)g.C)[(long)row * g.ldc + col] = f32_to_bf16(x);
        }
    }
  }
}

}  // namespace g3
