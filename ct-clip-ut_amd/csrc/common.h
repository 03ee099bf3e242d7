// Shared device helpers for the CT-CLIP gfx950 kernels (wave64, MFMA, LDS).
// No CUDA compatibility layer: this code targets CDNA4 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;  // raw bfloat16 storage

// Environment inputs.  The shipped library reads exactly two variables, both TEST HOOKS documented in include/ctclip_hip.h:
//   CTCLIP_GEMM_V2_ALL    lower the size gates of the pipelined GEMM kernels, so the test-suite reaches every kernel with
//                         small shapes;
//   CTCLIP_ATTN_SP_CHUNK  sequences per workgroup of the wave-per-sequence attention kernels (ragged-chunk tests).
// A/B switches and timing ablations of the development rounds exist only in builds made with -DCTCLIP_TUNING_KNOBS
// (CTCLIP_EXTRA_HIPCC_FLAGS of ctclip_hip/build.py, which writes libctclip_hip_diag.so next to the product library).
#include <stdlib.h>
#ifdef CTCLIP_TUNING_KNOBS
#define CTCLIP_KNOB(name) getenv(name)
#else
#define CTCLIP_KNOB(name) ((const char*)nullptr)
#endif

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short short4v;
typedef __attribute__((ext_vector_type(8))) short short8v;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// Scratch every entry point with a `partials` argument may use: CTCLIP_PARTIALS_FLOATS of include/ctclip_hip.h.
constexpr long kPartialsFloats = 1L << 21;
// out[c] += sum_{part < nparts, in index order} partials[part * ld + c]  (tail.hip; the second stage of the two-stage
// reductions: no atomics, bitwise reproducible)
int ctclip_reduce_partials(const float* partials, int nparts, long ld, int width, float* out, hipStream_t st);

// Per-DEVICE process state, all of it idempotent: the CU count (rounded down to a multiple of 8: a persistent workgroup's tiles
// then stay on one XCD's run) and "the dynamic-LDS limit of this kernel was raised" -- kept per device id, so that a process which
// drives several devices through this library gets the attribute on each of them.
inline int ctclip_dev() { int d = 0; (void)hipGetDevice(&d); return d & 63; }
inline int ctclip_cu_count8() {
  static int n[64];
  const int d = ctclip_dev();
  if (!n[d]) {
    int v = 256;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, d) != hipSuccess || v <= 0) v = 256;
    n[d] = v >= 8 ? (v / 8) * 8 : v;
  }
  return n[d];
}
#define CTCLIP_LDS_LIMIT_ONCE(fn, bytes)                                                                              \
  do {                                                                                                                \
    static unsigned long long done_ = 0;                                                                              \
    const int d_ = ctclip_dev();                                                                                      \
    if (!((done_ >> d_) & 1ull)) {                                                                                    \
      hipError_t e_ = hipFuncSetAttribute((const void*)(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)); \
      if (e_ != hipSuccess) return (int)e_;                                                                           \
      done_ |= 1ull << d_;                                                                                            \
    }                                                                                                                 \
  } while (0)

#define CTCLIP_CHECK_LAUNCH() \
  do {                        \
    hipError_t e_ = hipGetLastError(); \
    return (int)e_;           \
  } while (0)

// ----- bf16 <-> f32 -------------------------------------------------------------------
__device__ __forceinline__ float bf16_to_f32(bf16_t h) {
  return __uint_as_float(((uint32_t)h) << 16);
}
// round-to-nearest-even; plain cast lowers to v_cvt_pk_bf16_f32 on gfx950 and keeps NaN a NaN
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(bf16_t, b);
}
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  return (uint32_t)f32_to_bf16(lo) | ((uint32_t)f32_to_bf16(hi) << 16);
}

// ----- erf / exact (erf) GELU -----------------------------------------------------------------
// Abramowitz-Stegun 7.1.26: |error| <= 1.5e-7 over the whole real line -- three orders of magnitude below bf16
// resolution -- in ~14 issue slots (one v_rcp, one v_exp) instead of the ~50-instruction branchy libm erff.  The GELUs
// of this path (GEGLU attention.py:38-41, BertIntermediate) sit in GEMM epilogues and HBM-bound sweeps where that
// difference is exposed time.
__device__ __forceinline__ float erf_fast(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * ax * ax);
  const float y = fmaf(-p * t, e, 1.0f);
  return copysignf(y, x);
}
// The GELUs inside the GEMM epilogues (GEGLU forward / backward: 64-128 evaluations per lane and tile, VALU time the matrix
// pipe waits for) use a transcendental-free form: Phi(x) - 1/2 = x Q(x^2), Q a degree-6 minimax polynomial on |x| <= 3.7
// (beyond it Phi is exactly 0 / 1, so gelu(x) = 0 for x < -3.7 and x for x > 3.7).  |Phi error| <= 6e-5 inside the range and
// <= 1.1e-4 at the jump, |gelu error| <= 4e-4 absolute and <= 1.1e-4 |x| -- a thirtieth of the bf16 rounding of the values
// these epilogues store; 13 issue slots against ~19 with two quarter-rate instructions.
__device__ __forceinline__ float norm_cdf_fast(float x) {
  const float xc = __builtin_amdgcn_fmed3f(x, -3.7f, 3.7f);
  const float w = xc * xc;
  float q = fmaf(4.174217594e-08f, w, -2.483551043e-06f);
  q = fmaf(q, w, 6.402253348e-05f);
  q = fmaf(q, w, -9.565735236e-04f);
  q = fmaf(q, w, 9.406451136e-03f);
  q = fmaf(q, w, -6.585516781e-02f);
  q = fmaf(q, w, 3.987334669e-01f);
  // outside the fitted range Phi saturates to exactly 0 / 1 (x - xc is 0 inside it, and of x's sign beyond): without this
  // gelu(x) would leak 5.9e-5 |x| for strongly negative x
  return __builtin_amdgcn_fmed3f(fmaf(x - xc, 1e30f, fmaf(xc, q, 0.5f)), 0.f, 1.f);
}
__device__ __forceinline__ float gelu_erf_fast(float x) { return x * norm_cdf_fast(x); }
// d/dx [x Phi(x)] = Phi(x) + x phi(x)
__device__ __forceinline__ float gelu_erf_grad_fast(float x) {
  return fmaf(x * 0.3989422804014327f, __builtin_amdgcn_exp2f(-0.72134752044448170f * x * x), norm_cdf_fast(x));
}

// The same two functions on PAIRS of values: the polynomial, the products and the sums become packed-f32 instructions
// (v_pk_fma_f32 / v_pk_mul_f32: two lanes' worth of work per issue slot on CDNA3+), which is what the GEGLU epilogues of the
// GEMM tiles are bound by (64-128 evaluations per lane and tile next to a matrix loop of 16 K-steps).  Operation for operation
// the arithmetic of norm_cdf_fast / gelu_erf_fast / gelu_erf_grad_fast: the results are bit-identical.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 pk_splat(float v) { return f32x2{v, v}; }
__device__ __forceinline__ f32x2 norm_cdf_fast2(f32x2 x) {
  const f32x2 xc = {__builtin_amdgcn_fmed3f(x.x, -3.7f, 3.7f), __builtin_amdgcn_fmed3f(x.y, -3.7f, 3.7f)};
  const f32x2 w = xc * xc;
  f32x2 q = pk_fma(pk_splat(4.174217594e-08f), w, pk_splat(-2.483551043e-06f));
  q = pk_fma(q, w, pk_splat(6.402253348e-05f));
  q = pk_fma(q, w, pk_splat(-9.565735236e-04f));
  q = pk_fma(q, w, pk_splat(9.406451136e-03f));
  q = pk_fma(q, w, pk_splat(-6.585516781e-02f));
  q = pk_fma(q, w, pk_splat(3.987334669e-01f));
  const f32x2 r = pk_fma(x - xc, pk_splat(1e30f), pk_fma(xc, q, pk_splat(0.5f)));
  return f32x2{__builtin_amdgcn_fmed3f(r.x, 0.f, 1.f), __builtin_amdgcn_fmed3f(r.y, 0.f, 1.f)};
}
__device__ __forceinline__ f32x2 gelu_erf_fast2(f32x2 x) { return x * norm_cdf_fast2(x); }
// gelu(x) and d/dx gelu(x) of a pair from ONE evaluation of Phi
__device__ __forceinline__ void gelu_erf_both2(f32x2 x, f32x2& gelu, f32x2& grad) {
  const f32x2 cdf = norm_cdf_fast2(x);
  const f32x2 t = (pk_splat(-0.72134752044448170f) * x) * x;
  const f32x2 e = {__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
  gelu = x * cdf;
  grad = pk_fma(x * pk_splat(0.3989422804014327f), e, cdf);
}

// ----- wave-level reductions (64 lanes) -------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ----- LDS transposed read (gfx950 ds_read_b64_tr_b16) -----------------------------------
// Per 16-lane group: lane 4q+p supplies the address of (row q, cols 4p..4p+3) of a 4x16 block
// of 16-bit elements; lane i receives column i, rows 0..3 in elements 0..3.
__device__ __forceinline__ short4v lds_read_tr16(const void* lds_addr) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (short4v __attribute__((address_space(3)))*)(uintptr_t)(uint32_t)(uintptr_t)lds_addr);
}

__device__ __forceinline__ bf16x8 as_bf16x8(short8v v) { return __builtin_bit_cast(bf16x8, v); }

__device__ __forceinline__ bf16x8 join_tr(short4v lo, short4v hi) {
  short8v r;
  r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
  r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
  return __builtin_bit_cast(bf16x8, r);
}

// MFMA 32x32x16 bf16: D[32x32] += A[32x16] * B[16x32].
//   lane l (r = l&31, h = l>>5) holds A[r][8h+j] and B[8h+j][r], j = 0..7.
//   acc reg i holds D[(i&3) + 8*(i>>2) + 4h][r].
__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ int acc_row(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }

// XCD-aware block-id remap (8 XCDs, round-robin dispatch): gives every XCD a contiguous run of
// logical tile ids so neighbouring tiles share an L2.  Bijective for any nwg.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (bid >> 3);
}
