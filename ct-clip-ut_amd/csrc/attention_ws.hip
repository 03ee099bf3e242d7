// "Wave owns the sequence" fused attention for gfx950: the CT-ViT spatial shape (reference src/utils/attention.py:155-180,
// n = 576 tokens per frame, d_head = 32, one [heads, n, n] relative-position bias shared by EVERY sequence).
//
// attention_sp.hip keeps the bias tiles of a (head, 32-row query block) in registers by splitting the n/32 key tiles of
// every sequence over the eight waves of a workgroup -- and pays for it with a barrier, a partial-result exchange through
// LDS and a combine per sequence: measured ~6700 cycles per (sequence, query block) for ~600 cycles of matrix work and
// ~1200 of softmax arithmetic, because all eight waves walk one short dependency chain in lock-step.
// Here a workgroup still owns one (head, 32-row query block) for a chunk of sequences, but
//   * the bias tiles live in LDS, already in accumulator layout (tile t, register group j, lane l -> one float4), so a
//     wave initialises the score accumulator with four conflict-free ds_read_b128 and the bias still enters the score MFMA
//     as its C operand -- no VALU instruction;
//   * every wave takes WHOLE SEQUENCES of the chunk (wave w: seq0 + w, seq0 + w + 8, ...) and runs a flash-style loop over
//     the n/32 key tiles by itself: K tiles come straight from global memory as MFMA operands, V tiles go through a
//     wave-private 2 KiB transposed image, online softmax with lane-local statistics (S^T = K Q^T, lane = query);
//     no barrier, no exchange, no combine after the bias fill;
//   * K/V of tile t + 2 are requested while tile t is computed.
// The price: each sequence's K/V is read once per query block (n/32 times) instead of once -- from L2, not from HBM.
// Eligible: d_head 32, n % 32 == 0, n / 32 <= 24, no key mask, no dropout.
#include "attn_common.h"

namespace {

constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;
constexpr int WS_NW = 16;                       // waves per workgroup
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

struct WsArgs {
  AttnArgs a;
  int T;        // 32-wide tiles per row (n / 32)
  int chunk;    // sequences per workgroup
  float c1;     // scale * log2(e)
  float inv_scale;
};

__device__ __forceinline__ float exp2_fast(float x) { return __builtin_amdgcn_exp2f(x); }

// ------------------------------------------------------------------------------------------------
// forward: O = softmax(q k^T * scale + bias) v, lse
// ------------------------------------------------------------------------------------------------
// NW waves per workgroup: 16 (four per SIMD, <= 128 registers) hides the L2 latency of the K/V fragment loads better than 8
template <bool HAS_BIAS, int NW>
__global__ __launch_bounds__(NW * 64, NW / 4) void ws_fwd_kernel(WsArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const AttnArgs& a = p.a;
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, half = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  int L = xcd_remap(blockIdx.x, gridDim.x);
  const int blk = L % p.T;
  L /= p.T;
  const int head = L % a.heads, chunk_id = L / a.heads;
  const int q0 = blk * 32;
  const int seq0 = chunk_id * p.chunk, seq1 = min(a.nseq, seq0 + p.chunk);
  const int T = p.T;

  float4* bias_l = (float4*)smem;                                  // [T][4][64] float4: tile, register group, lane
  char* vimg = smem + (size_t)(HAS_BIAS ? T : 0) * 4096 + (size_t)w * 4096;   // wave-private: 2 x [32 keys][32 d] bf16
  if (HAS_BIAS) {
    // registers 4j .. 4j+3 of lane (r, half) of score tile t are keys 32 t + 8 j + 4 half + (0..3) of query q0 + r
    for (int id = tid; id < T * 256; id += NW * 64) {
      const int l = id & 63, j = (id >> 6) & 3, t = id >> 8;
      const float4 v = *(const float4*)(a.bias + ((long)head * a.n + q0 + (l & 31)) * a.n + 32 * t + 8 * j + 4 * (l >> 5));
      bias_l[id] = make_float4(v.x * p.inv_scale, v.y * p.inv_scale, v.z * p.inv_scale, v.w * p.inv_scale);
    }
    __syncthreads();
  }

  // per-lane element offsets inside one sequence (32-bit); the sequence base is uniform
  const int crow = lane >> 2, ccol = lane & 3;                     // 16-byte piece of a V tile: rows crow, crow + 16
  const uint32_t koff = (uint32_t)(r * a.ldk + head * 32 + 8 * half);
  const uint32_t voff = (uint32_t)(crow * a.ldv + head * 32 + ccol * 8);
  const uint32_t vstep = (uint32_t)(16 * a.ldv);
  const uint32_t qoff = (uint32_t)((q0 + r) * a.ldq + head * 32 + 8 * half);
  const uint32_t vst = img_off<32>(crow, ccol);                    // + 1024: sixteen rows further, same swizzle
  const long kseq = (long)a.n * a.ldk, vseq = (long)a.n * a.ldv, qseq = (long)a.n * a.ldq;
  const uint32_t ktile = (uint32_t)(32 * a.ldk), vtile = (uint32_t)(32 * a.ldv);

  for (int seq = seq0 + w; seq < seq1; seq += NW) {
    const bf16_t* kb = a.k + seq * kseq + koff;
    const bf16_t* vb = a.v + seq * vseq + voff;
    const bf16_t* qb = a.q + seq * qseq + qoff;
    const bf16x8 qf0 = as_bf16x8(*(const short8v*)qb), qf1 = as_bf16x8(*(const short8v*)(qb + 16));
    bf16x8 kr[2][2];                                               // K fragments of tiles t, t + 1 (slot = t & 1)
    u32x4 vr[2][2];                                                // V rows of the same tiles
    auto request = [&](int slot, int t) {
      kr[slot][0] = as_bf16x8(*(const short8v*)(kb + (uint32_t)t * ktile));
      kr[slot][1] = as_bf16x8(*(const short8v*)(kb + (uint32_t)t * ktile + 16));
      vr[slot][0] = *(const u32x4*)(vb + (uint32_t)t * vtile);
      vr[slot][1] = *(const u32x4*)(vb + (uint32_t)t * vtile + vstep);
    };
    request(0, 0);
    if (T > 1) request(1, 1);
    float m = -INFINITY, l = 0.f;                                  // running max (in score units) and sum of this lane's keys
    f32x16 O;
    zero_acc(O);

    auto tile = [&](int slot, int t) {
      char* vi = vimg + slot * 2048;
      *(u32x4*)(vi + vst) = vr[slot][0];
      *(u32x4*)(vi + vst + 1024) = vr[slot][1];
      f32x16 S;
      if (HAS_BIAS) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float4 b = bias_l[(t * 4 + j) * 64 + lane];
          S[4 * j] = b.x; S[4 * j + 1] = b.y; S[4 * j + 2] = b.z; S[4 * j + 3] = b.w;
        }
      } else {
        zero_acc(S);
      }
      S = mfma32(kr[slot][0], qf0, S);
      S = mfma32(kr[slot][1], qf1, S);
      if (t + 2 < T) request(slot, t + 2);                         // this slot's registers are free again
      float mt = S[0];
#pragma unroll
      for (int i = 1; i < 16; ++i) mt = fmaxf(mt, S[i]);
      mt = fmaxf(mt, __shfl_xor(mt, 32, 64));                      // lanes r and r + 32 hold the two key halves of query r
      if (__builtin_amdgcn_ballot_w64(mt > m) != 0ull) {           // rare after the first tiles: rescale what was summed
        const float mn = fmaxf(m, mt);
        const float alpha = exp2_fast((m - mn) * p.c1);            // m = -inf: 0, and O, l are 0
        l *= alpha;
#pragma unroll
        for (int i = 0; i < 16; ++i) O[i] *= alpha;
        m = mn;
      }
      const float m2 = m * p.c1;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float e = exp2_fast(fmaf(S[i], p.c1, -m2));
        S[i] = e;
        l += e;
      }
      const bf16x8 p0 = acc_frag(S, 0), p1 = acc_frag(S, 1);
      O = mfma32(tr_frag<32>(vi, 0, 0, 0, lane), p0, O);
      O = mfma32(tr_frag<32>(vi, 0, 1, 0, lane), p1, O);
    };
    int t = 0;
    for (; t + 1 < T; t += 2) {
      tile(0, t);
      tile(1, t + 1);
    }
    if (t < T) tile(0, t);

    l += __shfl_xor(l, 32, 64);
    const long row = (long)seq * a.n + q0 + r;
    const f32x16 oo[1] = {O};
    store_rows<32>(a.o + row * a.ldo + head * 32, oo, 1.0f / l, lane);
    if (half == 0) a.lse[((long)seq * a.heads + head) * a.n + q0 + r] = (m * p.c1 + __log2f(l)) * kLn2;
  }
}

int cu_count_ws() {
  static const int n = [] {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) return 256;
    return v;
  }();
  return n;
}

bool ws_shape_ok(const AttnArgs& a, int dhead) {
  if (CTCLIP_KNOB("CTCLIP_ATTN_NO_WS")) return false;
  if (dhead != 32 || a.mask || a.drop) return false;
  if (a.n % 32) return false;
  const int T = a.n / 32;
  return T >= 4 && T <= 24;
}

WsArgs ws_plan(const AttnArgs& a, int* nblocks) {
  WsArgs p{};
  p.a = a;
  p.T = a.n / 32;
  p.c1 = a.scale * kLog2e;
  p.inv_scale = 1.0f / a.scale;
  // one workgroup per CU is resident (bias tiles + V images: 104 KiB at n = 576): ~8 rounds of workgroups, every wave of a
  // workgroup with the same number of sequences
  const long roles = (long)p.T * a.heads;
  long nchunks = (8L * cu_count_ws() + roles - 1) / roles;
  if (nchunks < 1) nchunks = 1;
  long chunk = (a.nseq + nchunks - 1) / nchunks;
  chunk = (chunk + WS_NW - 1) / WS_NW * WS_NW;
  if (const char* e = getenv("CTCLIP_ATTN_SP_CHUNK")) {            // test hook (include/ctclip_hip.h): ragged chunks
    const int forced = atoi(e);
    if (forced > 0) chunk = forced;
  }
  if (chunk > a.nseq) chunk = a.nseq;
  p.chunk = (int)chunk;
  nchunks = (a.nseq + chunk - 1) / chunk;
  *nblocks = (int)(roles * nchunks);
  return p;
}

}  // namespace

// -1: shape not eligible, the caller falls back to attention_sp.hip / attention.hip
int ctclip_attn_ws_fwd(const CtclipAttnArgs& a, int dhead, hipStream_t st) {
  if (!ws_shape_ok(a, dhead)) return -1;
  int nblocks = 0;
  const WsArgs p = ws_plan(a, &nblocks);
  const bool hb = a.bias != nullptr;
  const size_t lds = (size_t)(hb ? p.T : 0) * 4096 + (size_t)WS_NW * 4096;
  if (lds > 160 * 1024) return -1;
  const void* fn = hb ? (const void*)ws_fwd_kernel<true, WS_NW> : (const void*)ws_fwd_kernel<false, WS_NW>;
  if (lds > 65536) {
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  if (hb) hipLaunchKernelGGL((ws_fwd_kernel<true, WS_NW>), dim3(nblocks), dim3(WS_NW * 64), lds, st, p);
  else hipLaunchKernelGGL((ws_fwd_kernel<false, WS_NW>), dim3(nblocks), dim3(WS_NW * 64), lds, st, p);
  return (int)hipGetLastError();
}
