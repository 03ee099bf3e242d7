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
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

struct WsArgs {
  AttnArgs a;
  int T;        // 32-wide tiles per row (n / 32)
  int chunk;    // sequences per workgroup
  int nchunks;
  float c1;     // scale * log2(e)
  float inv_scale;
};

__device__ __forceinline__ float exp2_fast(float x) { return __builtin_amdgcn_exp2f(x); }

// the value of lane l ^ 32 next to the lane's own, without the LDS crossbar a __shfl_xor(, 32) goes through:
// v_permlane32_swap exchanges the upper half of one register with the lower half of another
__device__ __forceinline__ float max_halves(float x) {
  const unsigned u = __float_as_uint(x);
  const auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);   // {lanes 0-31 in both halves, lanes 32-63 in both}
  return fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
}
__device__ __forceinline__ float sum_halves(float x) {
  const unsigned u = __float_as_uint(x);
  const auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
}

// ------------------------------------------------------------------------------------------------
// forward: O = softmax(q k^T * scale + bias) v, lse
// ------------------------------------------------------------------------------------------------
// QB query blocks (32 rows each) per wave: a K/V tile is fetched once and used for QB score tiles, so the L2 traffic per flop
// falls with QB (measured at QB = 1: 2.3 ms per call, bound by the ~16 GB of half-used 128-byte lines that leave L2, not by
// the softmax arithmetic).  NW waves per workgroup follow from the registers (QB 1: 16 waves, 2 and 3: 8).
// The bias tiles are kept in LDS as fp16 (11-bit mantissa: |error| <= 2.4e-4 |bias|, an order below the bf16 rounding of the
// q.k products they are added to), which is what lets two or three query blocks' tiles fit next to the V images.
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));

template <bool HAS_BIAS, int QB, int NW, int PF>
__global__ __launch_bounds__(NW * 64, NW / 4) void ws_fwd_kernel(WsArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const AttnArgs& a = p.a;
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, half = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int T = p.T;
  const int G = (T + QB - 1) / QB;                                 // query-block groups per row
  // logical order: query group fastest, then chunk, then head -- xcd_remap hands every XCD a contiguous run of it, i.e. (with
  // 8 heads) ONE head: the workgroups of a chunk walk the same sequences of the same head at about the same time on one XCD
  int L = xcd_remap(blockIdx.x, gridDim.x);
  const int grp = L % G;
  L /= G;
  const int chunk_id = L % p.nchunks, head = L / p.nchunks;
  const int seq0 = chunk_id * p.chunk, seq1 = min(a.nseq, seq0 + p.chunk);
  const int nqb = min(QB, T - grp * QB);                           // query blocks of this group (the last one may be short)

  half4_t* bias_l = (half4_t*)smem;                                // [QB][T][4][64] half4: block, tile, register group, lane
  char* vimg = smem + (size_t)(HAS_BIAS ? QB * T : 0) * 2048 + (size_t)w * 4096;   // wave-private: 2 x [32 keys][32 d] bf16
  if (HAS_BIAS) {
    // registers 4j .. 4j+3 of lane (r, half) of score tile t are keys 32 t + 8 j + 4 half + (0..3) of query q0 + r
    for (int id = tid; id < QB * T * 256; id += NW * 64) {
      const int l = id & 63, j = (id >> 6) & 3, bt = id >> 8, t = bt % T, b = bt / T;
      half4_t hv = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
      if (b < nqb) {
        const float4 v = *(const float4*)(a.bias + ((long)head * a.n + (grp * QB + b) * 32 + (l & 31)) * a.n + 32 * t + 8 * j + 4 * (l >> 5));
        hv[0] = (_Float16)(v.x * p.inv_scale); hv[1] = (_Float16)(v.y * p.inv_scale);
        hv[2] = (_Float16)(v.z * p.inv_scale); hv[3] = (_Float16)(v.w * p.inv_scale);
      }
      bias_l[id] = hv;
    }
    __syncthreads();
  }

  // per-lane element offsets inside one sequence (32-bit); the sequence base is uniform
  const int crow = lane >> 2, ccol = lane & 3;                     // 16-byte piece of a V tile: rows crow, crow + 16
  const uint32_t koff = (uint32_t)(r * a.ldk + head * 32 + 8 * half);
  const uint32_t voff = (uint32_t)(crow * a.ldv + head * 32 + ccol * 8);
  const uint32_t vstep = (uint32_t)(16 * a.ldv);
  const uint32_t qoff = (uint32_t)((grp * QB * 32 + r) * a.ldq + head * 32 + 8 * half);
  const uint32_t vst = img_off<32>(crow, ccol);                    // + 1024: sixteen rows further, same swizzle
  const long kseq = (long)a.n * a.ldk, vseq = (long)a.n * a.ldv, qseq = (long)a.n * a.ldq;
  const uint32_t ktile = (uint32_t)(32 * a.ldk), vtile = (uint32_t)(32 * a.ldv), qblk = (uint32_t)(32 * a.ldq);

  for (int seq = seq0 + w; seq < seq1; seq += NW) {
    const bf16_t* kb = a.k + seq * kseq + koff;
    const bf16_t* vb = a.v + seq * vseq + voff;
    const bf16_t* qb = a.q + seq * qseq + qoff;
    bf16x8 qf[QB][2];
#pragma unroll
    for (int b = 0; b < QB; ++b) {
      const bf16_t* qp = qb + (uint32_t)(b < nqb ? b : 0) * qblk;  // a short last group re-reads block 0 (results discarded)
      qf[b][0] = as_bf16x8(*(const short8v*)qp);
      qf[b][1] = as_bf16x8(*(const short8v*)(qp + 16));
    }
    bf16x8 kr[PF][2];                                              // K fragments of the next PF tiles (slot = t % PF)
    u32x4 vr[PF][2];                                               // V rows of the same tiles
    auto request = [&](int slot, int t) {
      kr[slot][0] = as_bf16x8(*(const short8v*)(kb + (uint32_t)t * ktile));
      kr[slot][1] = as_bf16x8(*(const short8v*)(kb + (uint32_t)t * ktile + 16));
      vr[slot][0] = *(const u32x4*)(vb + (uint32_t)t * vtile);
      vr[slot][1] = *(const u32x4*)(vb + (uint32_t)t * vtile + vstep);
    };
    request(0, 0);
    if (PF > 1 && T > 1) request(PF - 1, 1);
    float m[QB], l[QB];                                            // running max (score units) / sum of this lane's keys
    f32x16 O[QB];
#pragma unroll
    for (int b = 0; b < QB; ++b) { m[b] = -INFINITY; l[b] = 0.f; zero_acc(O[b]); }

    auto tile = [&](int slot, int t) {
      char* vi = vimg + (t & 1) * 2048;
      *(u32x4*)(vi + vst) = vr[slot][0];
      *(u32x4*)(vi + vst + 1024) = vr[slot][1];
      const bf16x8 k0 = kr[slot][0], k1 = kr[slot][1];
      if (t + PF < T) request(slot, t + PF);                       // this slot's registers are free again
      const bf16x8 vt0 = tr_frag<32>(vi, 0, 0, 0, lane), vt1 = tr_frag<32>(vi, 0, 1, 0, lane);
#pragma unroll
      for (int b = 0; b < QB; ++b) {
        f32x16 S;
        if (HAS_BIAS) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const half4_t hb = bias_l[((b * T + t) * 4 + j) * 64 + lane];
            S[4 * j] = (float)hb[0]; S[4 * j + 1] = (float)hb[1]; S[4 * j + 2] = (float)hb[2]; S[4 * j + 3] = (float)hb[3];
          }
        } else {
          zero_acc(S);
        }
        S = mfma32(k0, qf[b][0], S);
        S = mfma32(k1, qf[b][1], S);
        float mt = S[0];
#pragma unroll
        for (int i = 1; i < 16; ++i) mt = fmaxf(mt, S[i]);
        mt = max_halves(mt);                                       // lanes r and r + 32 hold the two key halves of query r
        if (__builtin_amdgcn_ballot_w64(mt > m[b]) != 0ull) {      // rare after the first tiles: rescale what was summed
          const float mn = fmaxf(m[b], mt);
          const float alpha = exp2_fast((m[b] - mn) * p.c1);       // m = -inf: 0, and O, l are 0
          l[b] *= alpha;
#pragma unroll
          for (int i = 0; i < 16; ++i) O[b][i] *= alpha;
          m[b] = mn;
        }
        const float m2 = m[b] * p.c1;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const float e = exp2_fast(fmaf(S[i], p.c1, -m2));
          S[i] = e;
          l[b] += e;
        }
        const bf16x8 p0 = acc_frag(S, 0), p1 = acc_frag(S, 1);
        O[b] = mfma32(vt0, p0, O[b]);
        O[b] = mfma32(vt1, p1, O[b]);
      }
    };
    int t = 0;
    for (; t + 1 < T; t += 2) {                                    // two tiles per trip: the V image alternates, slots are static
      tile(0, t);
      tile(PF - 1, t + 1);
    }
    if (t < T) tile(0, t);

#pragma unroll
    for (int b = 0; b < QB; ++b) {
      if (b >= nqb) continue;
      const float lt = sum_halves(l[b]);
      const int q = (grp * QB + b) * 32 + r;
      const long row = (long)seq * a.n + q;
      const f32x16 oo[1] = {O[b]};
      store_rows<32>(a.o + row * a.ldo + head * 32, oo, 1.0f / lt, lane);
      if (half == 0) a.lse[((long)seq * a.heads + head) * a.n + q] = (m[b] * p.c1 + __log2f(lt)) * kLn2;
    }
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

WsArgs ws_plan(const AttnArgs& a, int qb, int nw, int* nblocks) {
  WsArgs p{};
  p.a = a;
  p.T = a.n / 32;
  p.c1 = a.scale * kLog2e;
  p.inv_scale = 1.0f / a.scale;
  // one workgroup per CU is resident (bias tiles + V images in LDS): ~8 rounds of workgroups, every wave of a workgroup
  // with the same number of sequences
  const long roles = (long)((p.T + qb - 1) / qb) * a.heads;
  long nchunks = (8L * cu_count_ws() + roles - 1) / roles;
  if (nchunks < 1) nchunks = 1;
  long chunk = (a.nseq + nchunks - 1) / nchunks;
  chunk = (chunk + nw - 1) / nw * nw;
  if (const char* e = getenv("CTCLIP_ATTN_SP_CHUNK")) {            // test hook (include/ctclip_hip.h): ragged chunks
    const int forced = atoi(e);
    if (forced > 0) chunk = forced;
  }
  if (chunk > a.nseq) chunk = a.nseq;
  p.chunk = (int)chunk;
  nchunks = (a.nseq + chunk - 1) / chunk;
  p.nchunks = (int)nchunks;
  *nblocks = (int)(roles * nchunks);
  return p;
}

template <int QB, int NW, int PF>
int ws_fwd_launch(const AttnArgs& a, hipStream_t st) {
  int nblocks = 0;
  const WsArgs p = ws_plan(a, QB, NW, &nblocks);
  const bool hb = a.bias != nullptr;
  const size_t lds = (size_t)(hb ? QB * p.T : 0) * 2048 + (size_t)NW * 4096;
  if (lds > 160 * 1024) return -1;
  const void* fn = hb ? (const void*)ws_fwd_kernel<true, QB, NW, PF> : (const void*)ws_fwd_kernel<false, QB, NW, PF>;
  if (lds > 65536) {
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  if (hb) hipLaunchKernelGGL((ws_fwd_kernel<true, QB, NW, PF>), dim3(nblocks), dim3(NW * 64), lds, st, p);
  else hipLaunchKernelGGL((ws_fwd_kernel<false, QB, NW, PF>), dim3(nblocks), dim3(NW * 64), lds, st, p);
  return (int)hipGetLastError();
}

}  // namespace

// -1: shape not eligible, the caller falls back to attention_sp.hip / attention.hip
int ctclip_attn_ws_fwd(const CtclipAttnArgs& a, int dhead, hipStream_t st) {
  if (!ws_shape_ok(a, dhead)) return -1;
  static const int qb = [] { const char* e = CTCLIP_KNOB("CTCLIP_ATTN_WS_QB"); return e ? atoi(e) : 2; }();
  // measured at 1536 x 8 x 576 x 32 (same box): QB 1 / 16 waves 2369 us (L2-bound), QB 2 / 8 waves 1762, QB 3 / 8 waves 1734,
  // QB 2 / 12 waves 1820 (attention_sp.hip: 2898).  At QB = 2 the kernel issues 122 VALU instructions per score tile and
  // the VALU is busy 62 % of the time (rocprofv3 SQ counters, profiles/r02_attention_pmc.txt): softmax-arithmetic bound.
#ifdef CTCLIP_TUNING_KNOBS
  if (qb == 1) return ws_fwd_launch<1, 16, 2>(a, st);
  if (qb == 3) return ws_fwd_launch<3, 8, 2>(a, st);
  if (qb == 5) return ws_fwd_launch<2, 12, 1>(a, st);
#endif
  (void)qb;
  return ws_fwd_launch<2, 8, 2>(a, st);
}
