// The transposed-operand tile of the weight-gradient GEMMs (gemm4.hip; patch_gemm.hip's weight gradient from the volume): operands
// stored [32 k][256 columns] in 512-byte k-rows, fragments read with ds_read_b64_tr_b16.  Shared so that a producer that WRITES the
// tile from registers (patch_gemm.hip) uses the same addresses the fragment reads do.
#pragma once
#include "common.h"

namespace g4 {

constexpr int BM = 256, BN = 256, BK = 32, NS = 4, NT = 512;
constexpr int SUB = 16384;                 // one operand tile of a stage: 32 k-rows x 256 columns of bf16
constexpr int STAGE = 2 * SUB;
constexpr int PPW = (STAGE / 1024) / (NT / 64);   // 4 LDS-DMA pieces of 1 KiB per wave and stage


// [32 k][256 cols] bf16 tile, 512-byte rows of 32 16-byte chunks.  A transposed read touches, per 16-lane group, four
// k-rows (k & 3) x 32 bytes; the XOR moves those rows to different bank groups and the two k-halves apart.
__device__ __forceinline__ uint32_t swz(int k) { return (uint32_t)(((k & 3) << 2) | ((k >> 2) & 3)); }
__device__ __forceinline__ uint32_t tile_off(int k, int chunk) { return (uint32_t)(k * 512) + (((uint32_t)chunk ^ swz(k)) << 4); }

__device__ __forceinline__ short4v tr_read_asm(uint32_t lds_addr) {
  short4v r;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r) : "v"(lds_addr));
  return r;
}
// s_waitcnt lgkmcnt(0) that data-depends on every register of one fragment set: its consumers cannot move above it
__device__ __forceinline__ void tr_wait(short4v (&p)[12]) {
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]),
                 "+v"(p[8]), "+v"(p[9]), "+v"(p[10]), "+v"(p[11])
               :
               : "memory");
}
// raw halves of the fragment  element j = T[k = 16s + 8*(lane>>5) + j][cbase + (lane&31)]
__device__ __forceinline__ void read_frag_tr(uint32_t tile_lds, int cbase, int s, int lane, short4v& lo, short4v& hi) {
  const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, p = i16 & 3, h = g >> 1;
  const int c0 = cbase + 16 * (g & 1) + 4 * p;
  const int klo = 16 * s + 8 * h + q;
  const uint32_t sub = (uint32_t)((p & 1) * 8);
  lo = tr_read_asm(tile_lds + tile_off(klo, c0 >> 3) + sub);
  hi = tr_read_asm(tile_lds + tile_off(klo + 4, c0 >> 3) + sub);
}

// MFMA 16x16x32 form: element j = T[k = 8*(lane>>4) + j][cbase + (lane&15)] -- the whole 32-deep K-step in one fragment
__device__ __forceinline__ void read_frag_tr16(uint32_t tile_lds, int cbase, int lane, short4v& lo, short4v& hi) {
  const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, p = i16 & 3;
  const int c0 = cbase + 4 * p;
  const int klo = 8 * g + q;
  const uint32_t sub = (uint32_t)((p & 1) * 8);
  lo = tr_read_asm(tile_lds + tile_off(klo, c0 >> 3) + sub);
  hi = tr_read_asm(tile_lds + tile_off(klo + 4, c0 >> 3) + sub);
}

// element offset (from the operand base, at k-tile 0) of the 16 bytes lane `lane` contributes to 1 KiB piece `p`
// (k-rows 2p, 2p+1) of the tile whose first column is c0; its LDS destination is piece_base + lane*16
__device__ __forceinline__ long piece_src(int p, int lane, int c0, int R, long ld) {
  const int k = 2 * p + (lane >> 5), pc = lane & 31, c = pc ^ (int)swz(k);
  int col = c0 + c * 8;
  if (col >= R) col = 0;                           // masked in the epilogue
  return (long)k * ld + col;
}

#define G4_GLDS(gptr, ldsoff)                                                                                     \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),                          \
                                   (__attribute__((address_space(3))) void*)(uintptr_t)(ldsoff), 16, 0, 0)


}  // namespace g4
