// Probe: how do the matrix pipe and the VALU of one SIMD share time?  Each workgroup has 8 waves (two per SIMD); the waves
// of the lower half run loop A, the upper half loop B (A, B in {idle, MFMA 32x32x16 bf16, v_exp_f32, v_add_f32,
// v_cvt_pk_bf16_f32, MFMA interleaved with exps in ONE wave}).  Prints shader cycles (s_memtime) per loop iteration.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/probe_overlap tools/probe_overlap.hip && /tmp/probe_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

enum { IDLE = 0, MFMA = 1, EXP = 2, ADD = 3, CVT = 4, MIX = 5, MIX2 = 6 };

// one iteration: MFMA: 8 independent-ish MFMAs (two accumulators alternating); EXP: 32 v_exp_f32; ADD: 32 v_add_f32;
// CVT: 32 v_cvt_pk_bf16_f32; MIX: 8 x (1 MFMA + 4 exps) in one instruction stream; MIX2: 8 x (1 MFMA + 8 exps)
template <int MODE>
__device__ __forceinline__ void body(f32x16& c0, f32x16& c1, bf16x8 a, bf16x8 b, float (&x)[16]) {
  if (MODE == MFMA) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
    }
  } else if (MODE == EXP) {
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
  } else if (MODE == ADD) {
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_add_f32 %0, %0, %0" : "+v"(x[i]));
  } else if (MODE == CVT) {
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %0" : "+v"(x[i]));
  } else if (MODE == MIX || MODE == MIX2) {
    constexpr int PER = MODE == MIX ? 4 : 8;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (i & 1) c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
      else       c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
#pragma unroll
      for (int j = 0; j < PER; ++j) asm volatile("v_exp_f32 %0, %0" : "+v"(x[(i * PER + j) & 15]));
    }
  }
}

template <int MA, int MB>
__global__ __launch_bounds__(512, 1) void probe(int iters, float* sink, long long* cyc) {
  const int w = threadIdx.x >> 6;
  f32x16 c0, c1;
  for (int i = 0; i < 16; ++i) { c0[i] = 0.f; c1[i] = 0.f; }
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)0.f; b[i] = (__bf16)0.f; }
  float x[16];
  for (int i = 0; i < 16; ++i) x[i] = -1.f * i * 1e-3f * threadIdx.x;
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  if (w < 4) { for (int it = 0; it < iters; ++it) body<MA>(c0, c1, a, b, x); }
  else       { for (int it = 0; it < iters; ++it) body<MB>(c0, c1, a, b, x); }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + x[i];
  if (s == 123.456f) sink[0] = s;
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[w] = t1 - t0;
}

template <int MA, int MB>
void run(const char* name, int iters) {
  float* sink; long long* cyc;
  hipMalloc(&sink, 4); hipMalloc(&cyc, 8 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  probe<MA, MB><<<256, 512>>>(iters, sink, cyc);
  hipEventRecord(e0);
  probe<MA, MB><<<256, 512>>>(iters, sink, cyc);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long h[8]; hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
  // s_memtime counts at 100 MHz on this part: report wall time per iteration, and cycles at 2.4 GHz for orientation
  printf("%-34s %8.1f us  = %7.1f ns/iter (%6.1f cycles/iter at 2.4 GHz)   counter: lower %lld upper %lld\n", name, ms * 1e3,
         ms * 1e6 / iters, ms * 1e6 / iters * 2.4, h[0], h[4]);
  hipFree(sink); hipFree(cyc);
}

int main() {
  const int N = 20000;
  run<MFMA, IDLE>("1 wave/SIMD MFMA x8", N);
  run<MFMA, MFMA>("2 waves/SIMD MFMA x8 each", N);
  run<EXP, IDLE>("1 wave/SIMD exp x32", N);
  run<EXP, EXP>("2 waves/SIMD exp x32 each", N);
  run<ADD, IDLE>("1 wave/SIMD add x32", N);
  run<ADD, ADD>("2 waves/SIMD add x32 each", N);
  run<CVT, IDLE>("1 wave/SIMD cvt_pk x32", N);
  run<MFMA, EXP>("MFMA x8 | exp x32 (two waves)", N);
  run<MFMA, ADD>("MFMA x8 | add x32 (two waves)", N);
  run<MIX, IDLE>("1 wave: 8 x (MFMA + 4 exp)", N);
  run<MIX2, IDLE>("1 wave: 8 x (MFMA + 8 exp)", N);
  run<MIX, MIX>("2 waves: 8 x (MFMA + 4 exp) each", N);
  run<MIX2, MIX2>("2 waves: 8 x (MFMA + 8 exp) each", N);
  return 0;
}
