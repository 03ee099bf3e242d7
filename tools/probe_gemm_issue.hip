// Probe: what does ONE wave per SIMD pay for the memory instructions of a GEMM K-step when they are issued between its own
// MFMAs?  256 workgroups x 4 waves; one iteration = 64 MFMA 16x16x32 bf16 (1024 matrix-pipe cycles) in 8 groups of 8, and behind
// each group, by variant:
//   mfma        nothing
//   dsr         2 ds_read_b128 (fragment reads of a k32 step: 16 per iteration)
//   dma         1 global_load_lds_dwordx4 (LDS-DMA, 1 KiB per wave instruction: 8 per iteration = the wave's share of a 32 KiB slot)
//   dma2        2 of them (16 per iteration)
//   ld          1 global_load_dwordx4 into registers
//   ldw         1 global_load_dwordx4 + 1 ds_write_b128 of the data loaded one iteration earlier (register staging)
//   dsr+dma     gemm5's K-step
//   dsr+ldw     the register-staged K-step
// SRC = l2: every workgroup re-reads its own 32 KiB window (L2 hits); hbm: streams through a 4 GiB buffer.
// Prints shader cycles (s_memtime) and wall ns per iteration.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/probe_gemm_issue tools/probe_gemm_issue.hip && /tmp/probe_gemm_issue
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d (%s) at %s:%d\n", (int)e_, hipGetErrorString(e_), __FILE__, __LINE__); fflush(stdout); exit(1); } } while (0)

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

#define GLDS(gptr, ldsoff)                                                                                        \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),                         \
                                   (__attribute__((address_space(3))) void*)(uintptr_t)(ldsoff), 16, 0, 0)

__device__ __forceinline__ void mfma_acc(f32x4& c, bf16x8 a, bf16x8 b) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}

enum { DSR = 1, DMA = 2, DMA2 = 4, LD = 8, LDW = 16, DMAH = 32, DMAP = 64 };

template <int V>
__global__ __launch_bounds__(256) void probe(const char* src, long window, long stride_per_iter, int iters, float* sink,
                                             long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  f32x4 acc[64];
#pragma unroll
  for (int i = 0; i < 64; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 fa[2][8], fb[2][8];
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int e = 0; e < 8; ++e) { fa[p][i][e] = (__bf16)(0.001f * (lane + i + e)); fb[p][i][e] = (__bf16)(0.002f * (lane - i + e)); }
  u32x4 stg[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) stg[i] = u32x4{0u, 0u, 0u, 0u};
  const char* base = src + (long)blockIdx.x * window + wave * 8192 + lane * 16;      // this wave's 8 KiB of the window
  long poff[8];                                      // opaque piece offsets: an immediate offset of a global_load_lds is added
#pragma unroll                                       // to the LDS address as well
  for (int i = 0; i < 8; ++i) { poff[i] = i * 1024; asm volatile("" : "+v"(poff[i])); }
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  long off = 0;
  int slot = 0;
  // one K-step: MFMAs on buffer `c`, fragment reads (compiler-generated ds_read_b128: their results are tracked) into buffer `n`
  auto kstep = [&](bool odd, bf16x8 (&fac)[8], bf16x8 (&fbc)[8], bf16x8 (&fan)[8], bf16x8 (&fbn)[8]) {
    const char* g = base + off;
    // half-line forms: the wave's 16 KiB of lines of a K64 step (window offset advances every second step)
    const char* gh = src + (long)blockIdx.x * window + (off & ~65535L) + wave * 16384 + (lane >> 2) * 128 + (lane & 3) * 16;
    const uint32_t sb = lds0 + (uint32_t)(slot * 32768 + wave * 8192);
    const char* rdp = smem + (slot ^ 1) * 32768 + lane * 16 + wave * 1024;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
      for (int j = 0; j < 8; ++j) mfma_acc(acc[i * 8 + j], fbc[j], fac[i]);
      if (V & DSR) {
        fan[i] = *(const bf16x8*)(rdp + i * 512);
        fbn[i] = *(const bf16x8*)(rdp + 16384 + i * 512);
      }
      // half-line pieces (16 rows x 64 B of 16 lines, what a k32 slot takes): DMAH one half per step, the other half of the
      // same lines a step later; DMAP both halves back to back in the even steps, nothing in the odd ones
      if (V & DMAH) GLDS(gh + poff[i] * 2 + (odd ? 64 : 0), sb + i * 1024);
      if ((V & DMAP) && !odd) { GLDS(gh + poff[i] * 2, sb + i * 1024); GLDS(gh + poff[i] * 2 + 64, sb + i * 1024 + 65536); }
      if (V & (DMA | DMA2)) GLDS(g + poff[i], sb + i * 1024);
      if (V & DMA2) GLDS(g + poff[i] + 16384, sb + i * 1024 + 65536);
      if (V & (LD | LDW)) {
        if (V & LDW) *(u32x4*)(smem + (slot * 32768 + wave * 8192 + i * 1024 + lane * 16)) = stg[i];
        stg[i] = *(const u32x4*)(g + i * 1024);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (V & (DMA | DMA2 | LD | LDW | DMAH | DMAP)) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    off += stride_per_iter;
    if (off >= window) off = 0;
    slot = (slot + 1) & 1;
  };
  for (int it = 0; it < iters; it += 2) {
    kstep(false, fa[0], fb[0], fa[1], fb[1]);
    kstep(true, fa[1], fb[1], fa[0], fb[0]);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 64; ++i) s += acc[i][0] + acc[i][3];
#pragma unroll
  for (int i = 0; i < 8; ++i) s += (float)fa[1][i][0] + (float)fb[1][i][1] + (float)stg[i][0];
  if (s == 123.456f) sink[0] = s;
  if (lane == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
}

template <int V>
void run(const char* name, const char* src, long window, long stride, int iters) {
  float* sink; long long* cyc;
  CK(hipMalloc(&sink, 4)); CK(hipMalloc(&cyc, 64));
  CK(hipFuncSetAttribute((const void*)probe<V>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072 + 8192));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  probe<V><<<256, 256, 131072 + 8192>>>(src, window, stride, iters, sink, cyc);
  CK(hipGetLastError()); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  probe<V><<<256, 256, 131072 + 8192>>>(src, window, stride, iters, sink, cyc);
  CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  long long h[4]; CK(hipMemcpy(h, cyc, 32, hipMemcpyDeviceToHost));
  printf("%-10s %-4s %9.1f us = %7.1f ns/iter   s_memtime cycles/iter %7.1f   (MFMA-only floor 1024)   -> %6.0f TFLOP/s equivalent\n", name,
         window > (1L << 20) ? "hbm" : "l2", ms * 1e3, ms * 1e6 / iters, (double)h[0] / iters,
         256.0 * 4 * 64 * 16384.0 / (ms * 1e6 / iters) / 1e3);
  fflush(stdout);
  CK(hipFree(sink)); CK(hipFree(cyc));
}

int main() {
  const int N = 4000;
  char* buf; const long big = 4L << 30;
  CK(hipMalloc(&buf, big + (1 << 20)));
  for (long o = 0; o < big + (1 << 20); o += 1L << 30) {       // 1 GiB pieces
    const long n = big + (1 << 20) - o < (1L << 30) ? big + (1 << 20) - o : (1L << 30);
    CK(hipMemset(buf + o, 1, (size_t)n));
  }
  CK(hipDeviceSynchronize());
  printf("buffer ok\n"); fflush(stdout);
  const long wl2 = 128L << 10;                  // per-workgroup window, re-read every 4 iterations: 32 MiB in all, L2 resident (stride 32 KiB per
                                                // step: a K64 pair covers 64 KiB of lines in the half-line forms)
  const long whbm = big / 256;                  // 16 MiB per workgroup, streamed once
  run<0>("mfma", buf, wl2, 32768, N);
  run<DSR>("dsr", buf, wl2, 32768, N);
  run<DMA>("dma", buf, wl2, 32768, N);
  run<DMA>("dma", buf, whbm, 32768, 500);
  run<DMA2>("dma2", buf, wl2, 32768, N);
  run<LD>("ld", buf, wl2, 32768, N);
  run<LDW>("ldw", buf, wl2, 32768, N);
  run<DSR | DMA>("dsr+dma", buf, wl2, 32768, N);
  run<DSR | DMA>("dsr+dma", buf, whbm, 32768, 500);
  run<DSR | DMAH>("dsr+dmaH", buf, wl2, 32768, N);
  run<DSR | DMAP>("dsr+dmaP", buf, wl2, 32768, N);
  run<DSR | DMAH>("dsr+dmaH", buf, whbm, 32768, 500);
  run<DSR | DMAP>("dsr+dmaP", buf, whbm, 32768, 500);
  run<DSR | LDW>("dsr+ldw", buf, wl2, 32768, N);
  run<DSR | LDW>("dsr+ldw", buf, whbm, 32768, 500);
  return 0;
}
