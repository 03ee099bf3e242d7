// "Wave owns the sequence" fused attention for gfx950: spatial attention rows with one [heads, n, n] relative-position bias
// shared by EVERY sequence (reference src/utils/attention.py:155-180), operands in the ROW-MAJOR [token][heads * 32] layout.
//
// WHO REACHES THIS FILE.  The CT-CLIP training step does not: its spatial attention (n = 576 or 256, 8 heads of 32) runs on the
// head-major kernels of attention_hm.hip (ops.py: attn_head_major_ok).  ctclip_attn_fwd / _bwd land here for the spatial shapes
// attention_hm.hip refuses: 21..24 key tiles per row (a non-square patch grid such as 24 x 28), an odd number of heads, or a
// caller that needs the row-major operands.  Shorter rows (temporal attention, n = 24), key masks and dropout (BERT) use the
// per-sequence kernels of attention.hip.  (Round 1's sequence-persistent form, attention_sp.hip, took exactly the shapes this
// file takes and lost to it 2898 vs 1762 us; it was unreachable in a product build and is gone -- DESIGN.md section 4.4 keeps
// its measurements.)
//
// A workgroup owns one (head, 32-row query block) for a chunk of sequences:
//   * the bias tiles live in LDS, already in accumulator layout (tile t, register group j, lane l -> one float4), so a
//     wave initialises the score accumulator with four conflict-free ds_read_b128 and the bias enters the score MFMA
//     as its C operand -- no VALU instruction;
//   * every wave takes WHOLE SEQUENCES of the chunk (wave w: seq0 + w, seq0 + w + 8, ...) and runs a flash-style loop over
//     the n/32 key tiles by itself: K tiles come straight from global memory as MFMA operands, V tiles go through a
//     wave-private 2 KiB transposed image, online softmax with lane-local statistics (S^T = K Q^T, lane = query);
//     no barrier, no exchange, no combine after the bias fill;
//   * K/V of tile t + 2 are requested while tile t is computed.
// The price: each sequence's K/V is read once per query block (n/32 times) instead of once -- from L2, not from HBM.
// Eligible: d_head 32, n % 32 == 0, 4 <= n / 32 <= 24, no key mask, no dropout.
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
  // One workgroup per CU is resident (bias tiles + images in LDS) and every workgroup pays a fixed price (bias fill, d(bias)
  // flush), so: few rounds of workgroups, and the FULL-size chunks fill a whole number of rounds -- what is left over makes
  // one short chunk whose workgroups finish early instead of a full-size tail.  Measured at 1536 sequences (forward /
  // backward with the table, us): 8.4 even rounds 1770 / 4813; 216 sequences per workgroup (7 full chunks = 1.97 rounds of
  // the 72 forward roles, 3.94 of the 144 dQ roles, + 24 left) 1697 / 4600; 192 (exactly 8 full chunks, 2.25 rounds) 1960 / 5328.
  const long roles = (long)((p.T + qb - 1) / qb) * a.heads;
  const long cus = cu_count_ws();
  long nfull = 0;
  double best = 0.0;
  for (long r = 1; r <= 8; ++r) {
    const long nf = r * cus / roles;                               // full-size chunks that fit r rounds
    if (nf < 1) continue;
    const double fill = (double)(roles * nf) / (double)(r * cus);
    if (fill > best + 0.02) { best = fill; nfull = nf; }           // the fewest rounds among (nearly) equally tight fits
  }
  long chunk = nfull > 0 ? a.nseq / nfull : a.nseq;
  chunk = chunk / nw * nw;                                         // every wave of a workgroup the same number of sequences
  if (chunk < nw) {                                                // few sequences: as many chunks as fill ~8 rounds
    long nchunks = (8 * cus + roles - 1) / roles;
    if (nchunks < 1) nchunks = 1;
    chunk = (a.nseq + nchunks - 1) / nchunks;
    chunk = (chunk + nw - 1) / nw * nw;
  }
  if (const char* e = getenv("CTCLIP_ATTN_SP_CHUNK")) {            // test hook (include/ctclip_hip.h): ragged chunks
    const int forced = atoi(e);
    if (forced > 0) chunk = forced;
  }
  if (chunk > a.nseq) chunk = a.nseq;
  if (chunk < 1) chunk = 1;
  p.chunk = (int)chunk;
  const long nchunks = (a.nseq + chunk - 1) / chunk;
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


// ------------------------------------------------------------------------------------------------
// backward pass 1: dQ and delta.  Workgroup = (head, group of QB 32-row query blocks, chunk of sequences); a wave owns whole
// sequences.  Lane = query, accumulator rows = keys (S^T = K Q^T as in the forward), so lse / delta are one scalar per lane.
// ------------------------------------------------------------------------------------------------
// LDS: the bias tiles (fp16, accumulator layout, as the forward) and a 2 x 2 KiB wave-private K image: the K tile arrives
// from global memory as the score MFMA's operand and is written once more in the swizzled row-major layout the transposing
// LDS read wants for dQ^T += K^T dS^T.  One K / V fetch serves the QB query blocks.
// delta[q] = sum_d dO[q][d] O[q][d] is computed here from the dO fragment the wave holds anyway (and stored for the other
// passes).
// DBL (QB = 1 only): d(bias) in the same pass.  The block's T d(bias) tiles live in LDS as f32 in accumulator layout; a wave
// adds its dS tile with a plain read-modify-write under a per-tile LDS lock (one lane's compare-and-swap; ds_add_f32 from
// every lane was measured at ~190 cycles per instruction) and every wave starts its walk over the key tiles somewhere else,
// so two waves rarely want the same tile at the same time.
template <bool HAS_BIAS, int QB, int NW, bool ROWLD, bool DBL>
__global__ __launch_bounds__(NW * 64, 1) void ws_bwd_dq_kernel(WsArgs p) {
  static_assert(!DBL || QB == 1, "the d(bias) tiles of one query block fill the LDS");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const AttnArgs& a = p.a;
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, half = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int T = p.T;
  const int G = (T + QB - 1) / QB;
  int L = xcd_remap(blockIdx.x, gridDim.x);
  const int grp = L % G;
  L /= G;
  const int chunk_id = L % p.nchunks, head = L / p.nchunks;
  const int seq0 = chunk_id * p.chunk, seq1 = min(a.nseq, seq0 + p.chunk);
  const int nqb = min(QB, T - grp * QB);
  const int q0 = grp * QB * 32;

  half4_t* bias_l = (half4_t*)smem;                                // [QB][T][4][64] half4
  char* after_bias = smem + (size_t)(HAS_BIAS ? QB * T : 0) * 2048;
  float4* dbias_l = (float4*)after_bias;                           // DBL: [T][4][64] float4 = registers 4 j .. 4 j + 3 of lane
  unsigned* lock_l = (unsigned*)(after_bias + (size_t)T * 4096);   // DBL: one word per key tile
  // images: [K 2 KiB | V 2 KiB] per buffer; DBL has room for one buffer per wave only
  constexpr int IMG_BUFS = DBL ? 1 : 2;
  constexpr int IMG_WAVE = (ROWLD ? 4096 : 2048) * IMG_BUFS;
  char* kimg = after_bias + (DBL ? (size_t)T * 4096 + 128 : 0) + (size_t)w * IMG_WAVE;
  char* vimg = kimg + 2048 * IMG_BUFS;                             // ROWLD only
  if (DBL) {
    for (int id = tid; id < T * 256; id += NW * 64) dbias_l[id] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int id = tid; id < T; id += NW * 64) lock_l[id] = 0u;
  }
  if (HAS_BIAS) {
    for (int id = tid; id < QB * T * 256; id += NW * 64) {
      const int l = id & 63, j = (id >> 6) & 3, bt = id >> 8, t = bt % T, b = bt / T;
      half4_t hv = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
      if (b < nqb) {
        const float4 v = *(const float4*)(a.bias + ((long)head * a.n + q0 + 32 * b + (l & 31)) * a.n + 32 * t + 8 * j + 4 * (l >> 5));
        hv[0] = (_Float16)(v.x * p.inv_scale); hv[1] = (_Float16)(v.y * p.inv_scale);
        hv[2] = (_Float16)(v.z * p.inv_scale); hv[3] = (_Float16)(v.w * p.inv_scale);
      }
      bias_l[id] = hv;
    }
  }
  if (HAS_BIAS || DBL) __syncthreads();

  // ROWLD: a tile is fetched as 16 whole 64-byte rows per wave-instruction (4 lanes per row) -- half the cache lines per
  // instruction of the fragment pattern (32 rows x 32 bytes) -- and both operands are read back from the LDS image
  const int crow = lane >> 2, ccol = lane & 3;
  const uint32_t koff = ROWLD ? (uint32_t)(crow * a.ldk + head * 32 + ccol * 8) : (uint32_t)(r * a.ldk + head * 32 + 8 * half);
  const uint32_t voff = ROWLD ? (uint32_t)(crow * a.ldv + head * 32 + ccol * 8) : (uint32_t)(r * a.ldv + head * 32 + 8 * half);
  const uint32_t kstep = ROWLD ? (uint32_t)(16 * a.ldk) : 16u, vstep = ROWLD ? (uint32_t)(16 * a.ldv) : 16u;
  const uint32_t qoff = (uint32_t)((q0 + r) * a.ldq + head * 32 + 8 * half);
  const uint32_t dooff = (uint32_t)((q0 + r) * a.lddo + head * 32 + 8 * half);
  const uint32_t ooff = (uint32_t)((q0 + r) * a.ldo + head * 32 + 8 * half);
  const uint32_t kst0 = ROWLD ? img_off<32>(crow, ccol) : img_off<32>(r, half);
  const uint32_t kst1 = ROWLD ? kst0 + 1024 : img_off<32>(r, 2 + half);   // sixteen rows further: same swizzle
  const long kseq = (long)a.n * a.ldk, vseq = (long)a.n * a.ldv, qseq = (long)a.n * a.ldq, doseq = (long)a.n * a.lddo,
             oseq = (long)a.n * a.ldo;
  const uint32_t ktile = (uint32_t)(32 * a.ldk), vtile = (uint32_t)(32 * a.ldv);
  const int rot = DBL ? (w * T) / NW : 0;                          // this wave's first key tile
  auto phys = [&](int it) { const int t = it + rot; return t >= T ? t - T : t; };

  for (int seq = seq0 + w; seq < seq1; seq += NW) {
    const bf16_t* kb = a.k + seq * kseq + koff;
    const bf16_t* vb = a.v + seq * vseq + voff;
    bf16x8 kr[2][2], vr[2][2];
    auto request = [&](int slot, int t) {
      kr[slot][0] = as_bf16x8(*(const short8v*)(kb + (uint32_t)t * ktile));
      kr[slot][1] = as_bf16x8(*(const short8v*)(kb + (uint32_t)t * ktile + kstep));
      vr[slot][0] = as_bf16x8(*(const short8v*)(vb + (uint32_t)t * vtile));
      vr[slot][1] = as_bf16x8(*(const short8v*)(vb + (uint32_t)t * vtile + vstep));
    };
    request(0, phys(0));
    if (T > 1) request(1, phys(1));
    bf16x8 qf[QB][2], df[QB][2];
    float nlse2[QB], delta[QB];
    f32x16 dq[QB];
#pragma unroll
    for (int b = 0; b < QB; ++b) {
      const uint32_t bo = (uint32_t)(b < nqb ? b : 0) * 32;        // a short last group re-reads block 0 (results discarded)
      const short8v q0v = *(const short8v*)(a.q + seq * qseq + qoff + bo * a.ldq), q1v = *(const short8v*)(a.q + seq * qseq + qoff + bo * a.ldq + 16);
      const short8v g0v = *(const short8v*)(a.dO + seq * doseq + dooff + bo * a.lddo), g1v = *(const short8v*)(a.dO + seq * doseq + dooff + bo * a.lddo + 16);
      const short8v o0v = *(const short8v*)(a.oin + seq * oseq + ooff + bo * a.ldo), o1v = *(const short8v*)(a.oin + seq * oseq + ooff + bo * a.ldo + 16);
      const long stat = ((long)seq * a.heads + head) * a.n + q0 + bo + r;
      nlse2[b] = -a.lse[stat] * kLog2e;
      qf[b][0] = as_bf16x8(q0v); qf[b][1] = as_bf16x8(q1v); df[b][0] = as_bf16x8(g0v); df[b][1] = as_bf16x8(g1v);
      float dl = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        dl = fmaf(bf16_to_f32((bf16_t)g0v[j]), bf16_to_f32((bf16_t)o0v[j]), dl);
        dl = fmaf(bf16_to_f32((bf16_t)g1v[j]), bf16_to_f32((bf16_t)o1v[j]), dl);
      }
      delta[b] = sum_halves(dl);
      if (half == 0 && b < nqb) a.delta[stat] = delta[b];
      zero_acc(dq[b]);
    }

    auto tile = [&](int slot, int it) {
      const int t = phys(it);
      char* ki = kimg + (IMG_BUFS == 2 ? (it & 1) * 2048 : 0);
      bf16x8 k0 = kr[slot][0], k1 = kr[slot][1], v0 = vr[slot][0], v1 = vr[slot][1];
      if (IMG_BUFS == 1) asm volatile("" ::: "memory");             // single image: the previous tile's reads stay above
      *(bf16x8*)(ki + kst0) = k0;
      *(bf16x8*)(ki + kst1) = k1;
      if (ROWLD) {
        char* vi = vimg + (IMG_BUFS == 2 ? (it & 1) * 2048 : 0);
        *(bf16x8*)(vi + kst0) = v0;
        *(bf16x8*)(vi + kst1) = v1;
        k0 = row_frag<32>(ki, 0, 0, lane); k1 = row_frag<32>(ki, 0, 1, lane);
        v0 = row_frag<32>(vi, 0, 0, lane); v1 = row_frag<32>(vi, 0, 1, lane);
      }
      const bf16x8 kt0 = tr_frag<32>(ki, 0, 0, 0, lane), kt1 = tr_frag<32>(ki, 0, 1, 0, lane);
      // all score / dP MFMAs of the tile first: block 1's run on the matrix pipe while block 0's exponentials issue
      f32x16 S[QB], dP[QB];
#pragma unroll
      for (int b = 0; b < QB; ++b) {
        if (HAS_BIAS) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const half4_t hb = bias_l[((b * T + t) * 4 + j) * 64 + lane];
            S[b][4 * j] = (float)hb[0]; S[b][4 * j + 1] = (float)hb[1]; S[b][4 * j + 2] = (float)hb[2]; S[b][4 * j + 3] = (float)hb[3];
          }
        } else {
          zero_acc(S[b]);
        }
        zero_acc(dP[b]);
        S[b] = mfma32(k0, qf[b][0], S[b]);
        S[b] = mfma32(k1, qf[b][1], S[b]);
        dP[b] = mfma32(v0, df[b][0], dP[b]);
        dP[b] = mfma32(v1, df[b][1], dP[b]);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int b = 0; b < QB; ++b) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const float pr = exp2_fast(fmaf(S[b][i], p.c1, nlse2[b]));
          S[b][i] = pr * (dP[b][i] - delta[b]);                     // dS^T[key][q]
        }
        const bf16x8 d0 = acc_frag(S[b], 0), d1 = acc_frag(S[b], 1);
        dq[b] = mfma32(kt0, d0, dq[b]);
        dq[b] = mfma32(kt1, d1, dq[b]);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (DBL) {
        // d(bias) tile t += dS: take the tile's lock (bounded spin: a wave never holds a lock across anything but the
        // few LDS instructions below), read-modify-write, release.  LDS executes a wave's instructions in order, so the
        // releasing store is performed after the tile's stores.
        unsigned* lk = lock_l + t;
        for (int spins = 0; spins < (1 << 22); ++spins) {
          unsigned old = 1u;
          if (lane == 0) old = atomicCAS(lk, 0u, 1u);
          if (__builtin_amdgcn_readfirstlane(old) == 0u) break;
          __builtin_amdgcn_s_sleep(1);
        }
        asm volatile("" ::: "memory");
        float4* dt = dbias_l + t * 256 + lane;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float4 v = dt[j * 64];
          v.x += S[0][4 * j]; v.y += S[0][4 * j + 1]; v.z += S[0][4 * j + 2]; v.w += S[0][4 * j + 3];
          dt[j * 64] = v;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) *(volatile unsigned*)lk = 0u;
      }
      if (it + 2 < T) request(slot, phys(it + 2));                 // a whole tile of work ahead of its use
    };
    int t = 0;
    for (; t + 1 < T; t += 2) {
      tile(0, t);
      tile(1, t + 1);
    }
    if (t < T) tile(0, t);

#pragma unroll
    for (int b = 0; b < QB; ++b) {
      if (b >= nqb) continue;
      const f32x16 dd[1] = {dq[b]};
      store_rows<32>(a.dq + ((long)seq * a.n + q0 + 32 * b + r) * a.lddq + head * 32, dd, a.scale, lane);
    }
  }

  if (DBL) {
    __syncthreads();                                               // every wave's tiles are in
    const float* dl = (const float*)dbias_l;                       // element (t, register i, lane l) at ((t*4 + i/4)*64 + l)*4 + i%4
    if (a.dbias_dense) {
      for (int id = tid; id < T * 1024; id += NW * 64) {
        const int e = id & 3, l = (id >> 2) & 63, j = (id >> 8) & 3, t = id >> 10, i = 4 * j + e;
        atomicAdd(a.dbias_dense + ((long)head * a.n + q0 + (l & 31)) * a.n + 32 * t + acc_row(i, l >> 5), dl[id]);
      }
    } else {
      float* table = (float*)(after_bias + (size_t)T * 4096 + 128);   // the images are idle now
      for (int i = tid; i < a.table_size; i += NW * 64) table[i] = 0.f;
      __syncthreads();
      for (int id = tid; id < T * 1024; id += NW * 64) {
        const int e = id & 3, l = (id >> 2) & 63, j = (id >> 8) & 3, t = id >> 10, i = 4 * j + e;
        const int q = q0 + (l & 31), key = 32 * t + acc_row(i, l >> 5);
        int ti;
        if (a.grid_w > 0) {
          const int yq = q / a.grid_w, xq = q % a.grid_w, yk = key / a.grid_w, xk = key % a.grid_w;
          ti = (yq - yk + a.grid_h - 1) * (2 * a.grid_w - 1) + (xq - xk + a.grid_w - 1);
        } else {
          ti = a.relidx[(long)q * a.n + key];
        }
        atomicAdd(&table[ti], dl[id]);
      }
      __syncthreads();
      for (int i = tid; i < a.table_size; i += NW * 64) {
        const float v = table[i];
        if (v != 0.f) atomicAdd(a.dbias_table + (long)head * a.table_size + i, v);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// backward pass 2: dK, dV.  Workgroup = (head, group of KB 32-key blocks, chunk of sequences); a wave owns whole sequences.
// Lane = key, accumulator rows = queries: P and dS come out as the B operands of dV^T += dO^T P and dK^T += Q^T dS.
// ------------------------------------------------------------------------------------------------
// The K / V fragments of the key blocks are fixed for a sequence; Q and dO tiles stream from global memory as MFMA operands
// (one fetch serves the KB key blocks: at KB = 1 the kernel is bound by the half-used 128-byte lines of those fetches, like
// the forward) and are written once more into a wave-private swizzled image for the transposing reads.
// -lse log2(e) and delta vary along the accumulator ROWS here, so each wave first copies its sequence's two stat rows into
// LDS and reads them back four at a time (broadcast reads, shared by the KB blocks).
template <bool HAS_BIAS, int KB, int NW>
__global__ __launch_bounds__(NW * 64, 1) void ws_bwd_dkv_kernel(WsArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const AttnArgs& a = p.a;
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, half = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int T = p.T;
  const int G = (T + KB - 1) / KB;
  int L = xcd_remap(blockIdx.x, gridDim.x);
  const int grp = L % G;
  L /= G;
  const int chunk_id = L % p.nchunks, head = L / p.nchunks;
  const int seq0 = chunk_id * p.chunk, seq1 = min(a.nseq, seq0 + p.chunk);
  const int nkb = min(KB, T - grp * KB);
  const int key0 = grp * KB * 32;

  half4_t* bias_l = (half4_t*)smem;                                // [KB][T][4][64] half4: bias[q rows of the registers][key]
  char* wave_l = smem + (size_t)(HAS_BIAS ? KB * T : 0) * 2048 + (size_t)w * (4096 + (size_t)a.n * 8);
  char* qimg = wave_l;                                             // [32 q][32 d] bf16
  char* doimg = wave_l + 2048;                                     // [32 q][32 d] bf16
  float* stat_l = (float*)(wave_l + 4096);                         // [n] -lse log2e, [n] delta
  if (HAS_BIAS) {
    for (int id = tid; id < KB * T * 256; id += NW * 64) {
      const int l = id & 63, j = (id >> 6) & 3, bt = id >> 8, t = bt % T, b = bt / T;
      half4_t hv = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
      if (b < nkb) {
        const float* bp = a.bias + ((long)head * a.n + 32 * t + 8 * j + 4 * (l >> 5)) * a.n + key0 + 32 * b + (l & 31);
        hv[0] = (_Float16)(bp[0] * p.inv_scale); hv[1] = (_Float16)(bp[a.n] * p.inv_scale);
        hv[2] = (_Float16)(bp[2 * (long)a.n] * p.inv_scale); hv[3] = (_Float16)(bp[3 * (long)a.n] * p.inv_scale);
      }
      bias_l[id] = hv;
    }
    __syncthreads();
  }

  const uint32_t koff = (uint32_t)((key0 + r) * a.ldk + head * 32 + 8 * half);
  const uint32_t voff = (uint32_t)((key0 + r) * a.ldv + head * 32 + 8 * half);
  const uint32_t qoff = (uint32_t)(r * a.ldq + head * 32 + 8 * half);
  const uint32_t dooff = (uint32_t)(r * a.lddo + head * 32 + 8 * half);
  const uint32_t ist0 = img_off<32>(r, half), ist1 = img_off<32>(r, 2 + half);
  const long kseq = (long)a.n * a.ldk, vseq = (long)a.n * a.ldv, qseq = (long)a.n * a.ldq, doseq = (long)a.n * a.lddo;
  const uint32_t qtile = (uint32_t)(32 * a.ldq), dotile = (uint32_t)(32 * a.lddo);
  const uint32_t kblk = (uint32_t)(32 * a.ldk), vblk = (uint32_t)(32 * a.ldv);

  for (int seq = seq0 + w; seq < seq1; seq += NW) {
    const bf16_t* qb = a.q + seq * qseq + qoff;
    const bf16_t* dob = a.dO + seq * doseq + dooff;
    bf16x8 qr[2][2], gr[2][2];
    auto request = [&](int slot, int t) {
      qr[slot][0] = as_bf16x8(*(const short8v*)(qb + (uint32_t)t * qtile));
      qr[slot][1] = as_bf16x8(*(const short8v*)(qb + (uint32_t)t * qtile + 16));
      gr[slot][0] = as_bf16x8(*(const short8v*)(dob + (uint32_t)t * dotile));
      gr[slot][1] = as_bf16x8(*(const short8v*)(dob + (uint32_t)t * dotile + 16));
    };
    bf16x8 kf[KB][2], vf[KB][2];
#pragma unroll
    for (int b = 0; b < KB; ++b) {
      const uint32_t bo = (uint32_t)(b < nkb ? b : 0);             // a short last group re-reads block 0 (results discarded)
      const bf16_t* kp = a.k + seq * kseq + koff + bo * kblk;
      const bf16_t* vp = a.v + seq * vseq + voff + bo * vblk;
      kf[b][0] = as_bf16x8(*(const short8v*)kp);
      kf[b][1] = as_bf16x8(*(const short8v*)(kp + 16));
      vf[b][0] = as_bf16x8(*(const short8v*)vp);
      vf[b][1] = as_bf16x8(*(const short8v*)(vp + 16));
    }
    request(0, 0);
    if (T > 1) request(1, 1);
    {
      const long stat = ((long)seq * a.heads + head) * a.n;
      for (int i = lane; i < a.n; i += 64) {
        stat_l[i] = -a.lse[stat + i] * kLog2e;
        stat_l[a.n + i] = a.delta[stat + i];
      }
    }
    f32x16 dk[KB], dv[KB];
#pragma unroll
    for (int b = 0; b < KB; ++b) { zero_acc(dk[b]); zero_acc(dv[b]); }

    auto tile = [&](int slot, int t) {
      const bf16x8 q0 = qr[slot][0], q1 = qr[slot][1], g0 = gr[slot][0], g1 = gr[slot][1];
      asm volatile("" ::: "memory");                               // the image is single: the previous tile's reads stay above
      *(bf16x8*)(qimg + ist0) = q0;
      *(bf16x8*)(qimg + ist1) = q1;
      *(bf16x8*)(doimg + ist0) = g0;
      *(bf16x8*)(doimg + ist1) = g1;
      const bf16x8 gt0 = tr_frag<32>(doimg, 0, 0, 0, lane), gt1 = tr_frag<32>(doimg, 0, 1, 0, lane);
      const bf16x8 qt0 = tr_frag<32>(qimg, 0, 0, 0, lane), qt1 = tr_frag<32>(qimg, 0, 1, 0, lane);
#pragma unroll
      for (int b = 0; b < KB; ++b) {
        f32x16 S, dP;
        if (HAS_BIAS) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const half4_t hb = bias_l[((b * T + t) * 4 + j) * 64 + lane];
            S[4 * j] = (float)hb[0]; S[4 * j + 1] = (float)hb[1]; S[4 * j + 2] = (float)hb[2]; S[4 * j + 3] = (float)hb[3];
          }
        } else {
          zero_acc(S);
        }
        zero_acc(dP);
        S = mfma32(q0, kf[b][0], S);                               // S[q][key]
        S = mfma32(q1, kf[b][1], S);
        dP = mfma32(g0, vf[b][0], dP);                             // dP[q][key]
        dP = mfma32(g1, vf[b][1], dP);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 l4 = *(const float4*)(stat_l + 32 * t + 8 * g + 4 * half);
          const float4 d4 = *(const float4*)(stat_l + a.n + 32 * t + 8 * g + 4 * half);
          const float ls[4] = {l4.x, l4.y, l4.z, l4.w}, de[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float pr = exp2_fast(fmaf(S[4 * g + i], p.c1, ls[i]));
            S[4 * g + i] = pr;
            dP[4 * g + i] = pr * (dP[4 * g + i] - de[i]);
          }
        }
        const bf16x8 p0 = acc_frag(S, 0), p1 = acc_frag(S, 1), d0 = acc_frag(dP, 0), d1 = acc_frag(dP, 1);
        dv[b] = mfma32(gt0, p0, dv[b]);
        dv[b] = mfma32(gt1, p1, dv[b]);
        dk[b] = mfma32(qt0, d0, dk[b]);
        dk[b] = mfma32(qt1, d1, dk[b]);
        __builtin_amdgcn_sched_barrier(0);                         // one block's temporaries live at a time
      }
      if (t + 2 < T) request(slot, t + 2);                         // a whole tile of work ahead of its use
    };
    int t = 0;
    for (; t + 1 < T; t += 2) {
      tile(0, t);
      tile(1, t + 1);
    }
    if (t < T) tile(0, t);

#pragma unroll
    for (int b = 0; b < KB; ++b) {
      if (b >= nkb) continue;
      const long row = (long)seq * a.n + key0 + 32 * b + r;
      const f32x16 kk[1] = {dk[b]}, vv[1] = {dv[b]};
      store_rows<32>(a.dk + row * a.lddk + head * 32, kk, a.scale, lane);
      store_rows<32>(a.dv + row * a.lddv + head * 32, vv, 1.0f, lane);
    }
  }
}

template <typename K>
int ws_launch(K kernel, const WsArgs& p, int nblocks, int nw, size_t lds, hipStream_t st) {
  if (lds > 160 * 1024) return -1;
  if (lds > 65536) {
    hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(kernel, dim3(nblocks), dim3(nw * 64), lds, st, p);
  return (int)hipGetLastError();
}

template <int NW, int QB, int KB, bool ROWLD>
int ws_bwd_launch(const AttnArgs& a, hipStream_t st) {
  int nb1 = 0, nb2 = 0, nbf = 0, nbf12 = 0;
  const WsArgs p1 = ws_plan(a, QB, NW, &nb1);
  const WsArgs p2 = ws_plan(a, KB, NW, &nb2);
  const WsArgs pf = ws_plan(a, 1, NW, &nbf), pf12 = ws_plan(a, 1, 12, &nbf12);
  const bool table = a.dbias_table != nullptr, dense = a.dbias_dense != nullptr;
  const bool hb = a.bias != nullptr, db = table || dense;
  constexpr size_t IMG = ROWLD ? 4096 : 2048;                      // K (+ V) image bytes per wave and buffer
  const size_t lds1 = (size_t)(hb ? QB * p1.T : 0) * 2048 + (size_t)NW * 2 * IMG;
  const size_t lds2 = (size_t)(hb ? KB * p2.T : 0) * 2048 + (size_t)NW * (4096 + (size_t)a.n * 8);
  // with a bias gradient: one query block per workgroup, its d(bias) tiles in LDS, as many waves as still have room
  const size_t ldsf = (size_t)(hb ? pf.T : 0) * 2048 + (size_t)pf.T * 4096 + 128 + (size_t)NW * IMG;
  const size_t ldsf12 = ldsf + (size_t)(12 - NW) * IMG;
  if (lds2 > 160 * 1024 || (db ? ldsf : lds1) > 160 * 1024) return -1;
  if (table && ((size_t)a.table_size * 4 > (size_t)NW * IMG || (a.grid_w <= 0 && !a.relidx))) return -1;
  int e;
  if (db) {
    // 12 waves (the kernel needs 156 registers): 4846 vs 5210 us per backward call at 1536 x 8 x 576 x 32
    const bool w12 = ldsf12 <= 160 * 1024 && !CTCLIP_KNOB("CTCLIP_ATTN_WS_FUSED_W8");
    if (w12) {
      e = hb ? ws_launch(ws_bwd_dq_kernel<true, 1, 12, ROWLD, true>, pf12, nbf12, 12, ldsf12, st)
             : ws_launch(ws_bwd_dq_kernel<false, 1, 12, ROWLD, true>, pf12, nbf12, 12, ldsf12, st);
    } else {
      e = hb ? ws_launch(ws_bwd_dq_kernel<true, 1, NW, ROWLD, true>, pf, nbf, NW, ldsf, st)
             : ws_launch(ws_bwd_dq_kernel<false, 1, NW, ROWLD, true>, pf, nbf, NW, ldsf, st);
    }
  } else {
    e = hb ? ws_launch(ws_bwd_dq_kernel<true, QB, NW, ROWLD, false>, p1, nb1, NW, lds1, st)
           : ws_launch(ws_bwd_dq_kernel<false, QB, NW, ROWLD, false>, p1, nb1, NW, lds1, st);
  }
  if (e) return e;
  return hb ? ws_launch(ws_bwd_dkv_kernel<true, KB, NW>, p2, nb2, NW, lds2, st)
            : ws_launch(ws_bwd_dkv_kernel<false, KB, NW>, p2, nb2, NW, lds2, st);
}

}  // namespace

// -1: shape not eligible, the caller falls back to attention.hip
int ctclip_attn_ws_fwd(const CtclipAttnArgs& a, int dhead, hipStream_t st) {
  if (!ws_shape_ok(a, dhead)) return -1;
  static const int qb = [] { const char* e = CTCLIP_KNOB("CTCLIP_ATTN_WS_QB"); return e ? atoi(e) : 2; }();
  // measured at 1536 x 8 x 576 x 32 (same box): QB 1 / 16 waves 2369 us (L2-bound), QB 2 / 8 waves 1762, QB 3 / 8 waves 1734,
  // QB 2 / 12 waves 1820 (round 1's sequence-persistent kernels: 2898).  At QB = 2 the kernel issues 122 VALU instructions per score tile and
  // the VALU is busy 62 % of the time (rocprofv3 SQ counters, profiles/r02_attention_pmc.txt): softmax-arithmetic bound.
#ifdef CTCLIP_TUNING_KNOBS
  if (qb == 1) return ws_fwd_launch<1, 16, 2>(a, st);
  if (qb == 3) return ws_fwd_launch<3, 8, 2>(a, st);
  if (qb == 5) return ws_fwd_launch<2, 12, 1>(a, st);
#endif
  (void)qb;
  return ws_fwd_launch<2, 8, 2>(a, st);
}

// dQ, dK, dV, d(bias) and delta for the same shapes; -1 = not eligible (nothing was launched)
int ctclip_attn_ws_bwd(const CtclipAttnArgs& a, int dhead, hipStream_t st) {
  if (!ws_shape_ok(a, dhead)) return -1;
  if (CTCLIP_KNOB("CTCLIP_ATTN_NO_WS_BWD")) return -1;
#ifdef CTCLIP_TUNING_KNOBS
  if (CTCLIP_KNOB("CTCLIP_ATTN_WS_KB1")) return ws_bwd_launch<8, 1, 1, false>(a, st);
  if (CTCLIP_KNOB("CTCLIP_ATTN_WS_FRAGLD")) return ws_bwd_launch<8, 2, 2, false>(a, st);
#endif
  return ws_bwd_launch<8, 2, 2, true>(a, st);
}
