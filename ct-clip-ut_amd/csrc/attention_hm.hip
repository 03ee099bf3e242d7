// Spatial attention of the CT-ViT on HEAD-MAJOR operands (reference src/utils/attention.py:146-182 at n = 576 tokens per
// frame, d_head = 32, one [heads, n, n] relative-position bias shared by every sequence).  Successor of attention_ws.hip
// (which stays as the row-major form behind ctclip_attn_fwd / _bwd); same decomposition -- a workgroup owns a (head, group of
// 32-row query blocks) for a chunk of sequences, A WAVE OWNS WHOLE SEQUENCES -- with the three things its counters asked for
// (profiles/r02_attention_pmc.txt: 12 % MFMA-busy, 85 % VALU-busy, 2.1x / 3.3x the algorithmic HBM bytes):
//
//   1. head-major q / k / v / dO: [sequence][head][token][32], written that way by the producing kernels (head-norm, the
//      kv and out-projection data-gradient GEMM epilogues).  A head's K or V of one sequence is ONE contiguous 36 KiB run, a
//      32-row tile 2 KiB: every fetched 128-byte line is used whole (row-major: the 64-byte head slice of a 1 KiB token row is
//      half a line, and the other half belongs to a head that runs on another XCD).
//   2. logits in the log2 domain, and NO maximum.  The caller folds scale * log2(e) into q (head-norm's multiplier), so a
//      score tile leaves the matrix pipe as log2-logits.  q and k are unit vectors times learned per-channel scales, so
//      |q.k| <= 8 log2(e) max_d |q_scale_d k_scale_d| (11.5 at initialisation), and with the head's largest |bias| entry
//      that is a bound B_h on every |log2-logit|, known BEFORE the kernel runs (ctclip_attn_shift, a function of the
//      parameters only).  While B_h <= 60, p = 2^s lies inside [2^-60, 2^60] -- f32 and bf16 share an 8-bit exponent, the
//      row sums and P V accumulate in f32 -- so the softmax needs no shift at all: no running maximum, no rescale of the
//      accumulator, no exchange between the lane halves, no subtraction (16 v_max, the permlane exchange, 16 fma per tile
//      before).  When the bound is wider (learned scales far beyond anything near initialisation) the SAME launch pair
//      runs the online-softmax kernel instead: both are always enqueued and each returns at once when the flag is not its
//      own (no host sync).
//   3. the bias tile enters through the matrix pipe: S^T[key][q] += sum_j Bias^T[key][j] I[j][q] is an f16 MFMA against an
//      identity fragment, so the fp16 tiles in LDS (same 2 KiB per tile and same 11-bit mantissa as before) are never
//      unpacked by the VALU (16 v_cvt_f32_f16 per tile before).  In the backward the per-row constants (-lse log2(e),
//      -delta) enter the S and dP accumulators before the products too -- in the dQ pass (constant per lane) as a rank-1 f16
//      MFMA, in the dK/dV pass (constant per accumulator row) loaded from LDS straight into the accumulator registers --
//      so p = exp2(S') and dS = p * dP'.
// Per score tile and wave the forward now issues 16 v_exp, 16 v_add (row sum), 8 v_cvt_pk and 6 MFMAs (4 without a bias).
// Where the forward stands (1536 x 8 x 576 x 32, profiles/r03_attention_pmc.txt): 1.11 ms = 470 TFLOP/s; the SQ counters give
// 51.5 VALU instructions and 6 MFMAs per score tile = 270 + 192 cycles of issue per tile and SIMD-resident wave pair, i.e.
// VALU 47 % + matrix pipe 33 % of all SIMD cycles -- the two pipes of a SIMD are shared by its two waves and an MFMA holds
// the vector issue port for 8 of its 32 cycles, so this is close to what two waves per SIMD can interleave.  Measured and
// dropped on the way: 12 waves per workgroup (three per SIMD: 168 registers, 14 spilled: 1275 us), one query block per
// wave with 16 or 12 waves (K / V fetched twice as often: 1437 / 1412 us), touching the wave's next sequence in L2 one
// sequence ahead (1137 us: the first touch is not what the tiles wait for).
#include "attn_common.h"
#ifndef HM_ABL_HOT
#define HM_ABL_HOT 0                  // diagnostic build, timing only: the backward's streamed operand is always L2-hot (profiles/r05_attention_bwd_fetch_ablation.txt)
#endif
#ifndef HM_STAMPS
#define HM_STAMPS 0                   // diagnostic build: per-wave cycle totals of the forward's phases (tools/bench_attn_hm.py STAMPS=1)
#endif

namespace {

constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

struct HmArgs {
  const bf16_t* q; const bf16_t* k; const bf16_t* v;   // [nseq][heads][n][32]
  const bf16_t* dO;                                    // same layout (backward)
  bf16_t* o; const bf16_t* oin; long ldo;              // row-major [nseq * n, ldo], head h in columns 32 h ..
  float* lse; float* delta;                            // [nseq][heads][n], natural log
  const float* bias;                                   // [heads][n][n] f32 (natural-log units) or null
  const float* shift;                                  // [heads + 1]: bound B_h in log2 units, [heads] != 0: unbounded path; or null
  bf16_t* dq; bf16_t* dk; bf16_t* dv; long lddq, lddk, lddv;   // row-major outputs
  float* dbias_dense; const uint16_t* relidx; float* dbias_table;
  int table_size, grid_h, grid_w;
  int nseq, n, heads, T, chunk, nchunks;
};

__device__ __forceinline__ float exp2_fast(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ f32x16 mfma32h(half8 a, half8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ float max_halves(float x) {
  const unsigned u = __float_as_uint(x);
  const auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
}
__device__ __forceinline__ float sum_halves(float x) {
  const unsigned u = __float_as_uint(x);
  const auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
}
__device__ __forceinline__ f32x16 splat16(float x) {
  f32x16 c;
#pragma unroll
  for (int i = 0; i < 16; ++i) c[i] = x;
  return c;
}

// identity fragments of the f16 MFMA: element j of k-step s is I[16 s + 8 half + j][r] (the A and the B fragment of the
// identity hold the same values)
__device__ __forceinline__ void identity_frags(half8 (&idf)[2], int r, int half) {
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int j = 0; j < 8; ++j) idf[s][j] = (r == 16 * s + 8 * half + j) ? (_Float16)1.f : (_Float16)0.f;
}

// Bias fragments in LDS: frag[((b * T + t) * 2 + s) * 64 + lane], element j = bias[q][key] * log2(e) with
//   q   = (QOWN ? own0 + 32 b : 32 t) + 16 s + 8 (lane >> 5) + j      key = (QOWN ? 32 t : own0 + 32 b) + (lane & 31)
// QOWN: the workgroup owns query blocks b and walks the key tiles t (forward, dQ pass: the fragment is the A operand of
// S^T += Bias^T I); else it owns key blocks and walks the query tiles (dK/dV pass: the B operand of S += I Bias).
template <bool QOWN>
__device__ __forceinline__ void fill_bias(half8* bl, const float* __restrict__ bias_h, int n, int own0, int nown, int NB,
                                          int T, int tid, int nthreads) {
  for (int id = tid; id < NB * T * 128; id += nthreads) {
    const int l = id & 63, s = (id >> 6) & 1, bt = id >> 7, t = bt % T, b = bt / T;
    half8 hv;
#pragma unroll
    for (int j = 0; j < 8; ++j) hv[j] = (_Float16)0.f;
    if (b < nown) {
      const int q = (QOWN ? own0 + 32 * b : 32 * t) + 16 * s + 8 * (l >> 5);
      const int key = (QOWN ? 32 * t : own0 + 32 * b) + (l & 31);
      const float* bp = bias_h + (long)q * n + key;
#pragma unroll
      for (int j = 0; j < 8; ++j) hv[j] = (_Float16)(bp[(long)j * n] * kLog2e);
    }
    bl[id] = hv;
  }
}

// The workgroups of the SHORT last chunk go first in dispatch order.  Workgroups are handed out in block order, so as the last
// blocks they started only when round 2 had ended -- 72 workgroups with 3 sequences per wave holding the launch for another
// 90 us of 1215 (stamped: start times 0 / 560-576 / 1122-1158 us at 1536 sequences) -- as the first ones they are gone after
// a tenth of a round and their CUs take full-size chunks.
__device__ __forceinline__ int hm_chunk_order(int logical, int nchunks) { return logical == 0 ? nchunks - 1 : logical - 1; }

// Sequences are handed out to the waves of a workgroup from a counter in LDS: with the static assignment (wave w takes
// seq0 + w, + NW, ...) the waves of a workgroup ended up to 16 % apart (767k .. 1061k cycles of wave life around a mean of
// 902k) and the CU waited for the slowest one.
__device__ __forceinline__ int hm_next_seq(int* counter, int lane) {
  int v = 0;
  if (lane == 0) v = atomicAdd(counter, 1);
  return __builtin_amdgcn_readfirstlane(v);
}

// The same fp16 bias tiles in ACCUMULATOR layout (forward, HM_BIAS_C): cfrag[((b * T + t) * 2 + u) * 64 + lane], element j =
// bias[own0 + 32 b + (lane & 31)][32 t + 8 (2 u + (j >> 2)) + 4 (lane >> 5) + (j & 3)] * log2(e) -- registers 8 u .. 8 u + 7 of
// the lane's S^T accumulator.  The tile is unpacked by the VALU (16 v_cvt_f32_f16) and enters the first score MFMA as its C
// operand: two MFMAs fewer per score tile than the identity product.  The kernels run against the power limit (the shader
// clock sits at 1.6-1.8 GHz and falls when stalls are removed), and a 32x32x16 MFMA costs far more energy than 16 conversions.
__device__ __forceinline__ void fill_bias_c(half8* bl, const float* __restrict__ bias_h, int n, int own0, int nown, int NB, int T,
                                            int tid, int nthreads) {
  for (int id = tid; id < NB * T * 128; id += nthreads) {
    const int l = id & 63, u = (id >> 6) & 1, bt = id >> 7, t = bt % T, b = bt / T;
    half8 hv;
#pragma unroll
    for (int j = 0; j < 8; ++j) hv[j] = (_Float16)0.f;
    if (b < nown) {
      const float* bp = bias_h + (long)(own0 + 32 * b + (l & 31)) * n + 32 * t + 16 * u + 4 * (l >> 5);
#pragma unroll
      for (int j = 0; j < 8; ++j) hv[j] = (_Float16)(bp[8 * (j >> 2) + (j & 3)] * kLog2e);
    }
    bl[id] = hv;
  }
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
// STATIC: p = 2^s, no maximum (the logits are bounded: see the header); else the online-softmax form (running maximum per
// lane, rescale when it moves) -- the fallback when ctclip_attn_shift flags the bound as too wide, and the form used when no
// bound is given.  The score MFMAs of all QB query blocks are issued before any exponential, so one block's matrix work
// runs under the other's VALU work.
template <bool HAS_BIAS, bool STATIC, int QB, int NW>
__global__ __launch_bounds__(NW * 64, 1) void hm_fwd_kernel(HmArgs a) {
  if (a.shift) {                                                   // both kernels of the pair are enqueued: only one runs
    const bool unsafe = a.shift[a.heads] != 0.f;
    if (STATIC == unsafe) return;
  }
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, half = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int T = a.T;
  const int G = (T + QB - 1) / QB;
  // logical order: query group fastest, then chunk, then head -- xcd_remap hands every XCD a contiguous run of it, i.e. (with
  // 8 heads) ONE head: the workgroups of a chunk walk the same sequences of the same head at about the same time on one XCD
  int L = xcd_remap(blockIdx.x, gridDim.x);
  const int grp = L % G;
  L /= G;
  const int chunk_id = hm_chunk_order(L % a.nchunks, a.nchunks), head = L / a.nchunks;
  const int seq0 = chunk_id * a.chunk, seq1 = min(a.nseq, seq0 + a.chunk);
  const int nqb = min(QB, T - grp * QB);
  const int q0 = grp * QB * 32;

  half8* bias_l = (half8*)smem;                                    // [QB][T][2][64] half8
  char* vimg = smem + (size_t)(HAS_BIAS ? QB * T : 0) * 2048 + (size_t)w * 4096;   // wave-private: 2 x [32 keys][32 d] bf16
  __shared__ int seq_counter;
  if (tid == 0) seq_counter = seq0;
  if (HAS_BIAS) fill_bias_c(bias_l, a.bias + (long)head * a.n * a.n, a.n, q0, nqb, QB, T, tid, NW * 64);
  __syncthreads();
  const f32x16 zero16 = splat16(0.f);
  const int crow = lane >> 2, ccol = lane & 3;                     // 16-byte piece of a V tile: rows crow, crow + 16
  const uint32_t koff = (uint32_t)(r * 32 + 8 * half);
  const uint32_t voff = (uint32_t)(crow * 32 + ccol * 8);
  const uint32_t qoff = (uint32_t)((q0 + r) * 32 + 8 * half);
  const uint32_t vst = img_off<32>(crow, ccol);                    // + 1024: sixteen rows further, same swizzle
  const long hstride = (long)a.n * 32;

#if HM_STAMPS
  long long c_pro = 0, c_loop = 0, c_epi = 0, c_n = 0;
  const long long c_start = __builtin_readcyclecounter();
  const long long r_start = __builtin_amdgcn_s_memrealtime();
#endif
  // The K / V tile stream and the q fragments run ACROSS sequences: the last two requests of a sequence fetch the first two
  // tiles of the wave's NEXT sequence (taken from the counter one sequence ahead) and its q rows are requested before the
  // epilogue, so a sequence no longer starts with a memory round trip (stamped: prologue 1.7-2.0k + epilogue 2.1-2.7k of
  // 27-33k cycles per sequence).
  bf16x8 qf[QB][2];
  bf16x8 kr[2][2];
  u32x4 vr[2][2];
  auto seq_base = [&](int sq) { return ((long)sq * a.heads + head) * hstride; };
  auto load_q = [&](long base) {
#pragma unroll
    for (int b = 0; b < QB; ++b) {
      const bf16_t* qp = a.q + base + qoff + (uint32_t)(b < nqb ? b : 0) * 1024;   // a short last group re-reads block 0 (results discarded)
      qf[b][0] = as_bf16x8(*(const short8v*)qp);
      qf[b][1] = as_bf16x8(*(const short8v*)(qp + 16));
    }
  };
  auto request = [&](int slot, long base, int t) {
    const bf16_t* kp = a.k + base + koff + (uint32_t)t * 1024;
    const bf16_t* vp = a.v + base + voff + (uint32_t)t * 1024;
    kr[slot][0] = as_bf16x8(*(const short8v*)kp);
    kr[slot][1] = as_bf16x8(*(const short8v*)(kp + 16));
    vr[slot][0] = *(const u32x4*)vp;
    vr[slot][1] = *(const u32x4*)(vp + 512);
  };
  int seq = hm_next_seq(&seq_counter, lane);
  if (seq < seq1) {
    const long b0 = seq_base(seq);
    load_q(b0);
    request(0, b0, 0);
    request(1, b0, T > 1 ? 1 : 0);
  }
  while (seq < seq1) {
#if HM_STAMPS
    const long long c0 = __builtin_readcyclecounter();
#endif
    const int nxt = hm_next_seq(&seq_counter, lane);
    const long base = seq_base(seq);
    const long nbase = seq_base(nxt < seq1 ? nxt : seq);           // no next sequence: re-read this one's first tiles (dropped)
    // q (requested before the previous epilogue's stores) must be here now; say so ONCE -- left to the compiler's wait-count
    // bookkeeping the q registers count as "a load of unknown age" at the head of the tile loop and every trip waits with
    // vmcnt(1), draining the K / V request it has just issued.  Everything older (both K / V slots) has arrived as well.
    // (This also waits for the previous epilogue's stores, the youngest operations.  Waiting for everything BUT them --
    // asm stores, an exact vmcnt(10) -- does not survive the compiler: it knows nothing of asm stores and puts its own,
    // smaller count in front of the first use of every loaded register.)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int b = 0; b < QB; ++b) asm volatile("" : "+v"(qf[b][0]), "+v"(qf[b][1]));
#pragma unroll
    for (int sl = 0; sl < 2; ++sl) asm volatile("" : "+v"(kr[sl][0]), "+v"(kr[sl][1]), "+v"(vr[sl][0]), "+v"(vr[sl][1]));
    float m[QB], l[QB];
    f32x16 O[QB];
#pragma unroll
    for (int b = 0; b < QB; ++b) { m[b] = -INFINITY; l[b] = 0.f; zero_acc(O[b]); }

    auto tile = [&](int slot, int t) {
      char* vi = vimg + (t & 1) * 2048;
      *(u32x4*)(vi + vst) = vr[slot][0];
      *(u32x4*)(vi + vst + 1024) = vr[slot][1];
      const bf16x8 k0 = kr[slot][0], k1 = kr[slot][1];
      const bf16x8 vt0 = tr_frag<32>(vi, 0, 0, 0, lane), vt1 = tr_frag<32>(vi, 0, 1, 0, lane);
      f32x16 S[QB];
      if (HAS_BIAS) {
#pragma unroll
        for (int b = 0; b < QB; ++b) {
          const half8 c0 = bias_l[((b * T + t) * 2 + 0) * 64 + lane], c1 = bias_l[((b * T + t) * 2 + 1) * 64 + lane];
#pragma unroll
          for (int j = 0; j < 8; ++j) { S[b][j] = (float)c0[j]; S[b][8 + j] = (float)c1[j]; }
        }
#pragma unroll
        for (int b = 0; b < QB; ++b) S[b] = mfma32(k0, qf[b][0], S[b]);
      } else {
#pragma unroll
        for (int b = 0; b < QB; ++b) S[b] = mfma32(k0, qf[b][0], zero16);
      }
#pragma unroll
      for (int b = 0; b < QB; ++b) S[b] = mfma32(k1, qf[b][1], S[b]);
      __builtin_amdgcn_sched_barrier(0);
      // Request this slot's next tile AFTER the score MFMAs have read it (requested before them, the compiler parks k1 in
      // spare registers early -- a copy that waits for the OTHER slot's load, one tile ahead of its use).  UNCONDITIONAL:
      // behind a branch the wait-count bookkeeping no longer knows how many loads are in flight and waits for all of them.
      {
        const bool cur = t + 2 < T;                                // else: tile `slot` of the NEXT sequence lives in this slot
        request(slot, cur ? base : nbase, cur ? t + 2 : (slot < T ? slot : 0));
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int b = 0; b < QB; ++b) {
        if (STATIC) {
          float l0 = 0.f, l1 = 0.f;
#pragma unroll
          for (int i = 0; i < 16; i += 2) {
            const float e0 = exp2_fast(S[b][i]), e1 = exp2_fast(S[b][i + 1]);
            S[b][i] = e0; S[b][i + 1] = e1;
            l0 += e0; l1 += e1;
          }
          l[b] += l0 + l1;
        } else {
          float mt = S[b][0];
#pragma unroll
          for (int i = 1; i < 16; ++i) mt = fmaxf(mt, S[b][i]);
          mt = max_halves(mt);                                     // lanes r and r + 32 hold the two key halves of query r
          if (__builtin_amdgcn_ballot_w64(mt > m[b]) != 0ull) {    // rare after the first tiles: rescale what was summed
            const float mn = fmaxf(m[b], mt);
            const float alpha = exp2_fast(m[b] - mn);              // m = -inf: 0, and O, l are 0
            l[b] *= alpha;
#pragma unroll
            for (int i = 0; i < 16; ++i) O[b][i] *= alpha;
            m[b] = mn;
          }
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const float e = exp2_fast(S[b][i] - m[b]);
            S[b][i] = e;
            l[b] += e;
          }
        }
        const bf16x8 p0 = acc_frag(S[b], 0), p1 = acc_frag(S[b], 1);
        O[b] = mfma32(vt0, p0, O[b]);
        O[b] = mfma32(vt1, p1, O[b]);
      }
    };
#if HM_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long c1 = __builtin_readcyclecounter();
#endif
    int t = 0;
    for (; t + 1 < T; t += 2) {                                    // two tiles per trip: the V image alternates, slots are static
      tile(0, t);
      tile(1, t + 1);
    }
    if (t < T) tile(0, t);
#if HM_STAMPS
    const long long c2 = __builtin_readcyclecounter();
#endif
    load_q(nbase);                                                 // in flight under the epilogue

#pragma unroll
    for (int b = 0; b < QB; ++b) {
      if (b >= nqb) continue;
      const float lt = sum_halves(l[b]);
      const int q = q0 + 32 * b + r;
      const long row = (long)seq * a.n + q;
      const f32x16 oo[1] = {O[b]};
      store_rows<32>(a.o + row * a.ldo + head * 32, oo, 1.0f / lt, lane);
      if (half == 0) a.lse[((long)seq * a.heads + head) * a.n + q] = ((STATIC ? 0.f : m[b]) + __log2f(lt)) * kLn2;
    }
#if HM_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long c3 = __builtin_readcyclecounter();
    c_pro += c1 - c0; c_loop += c2 - c1; c_epi += c3 - c2; c_n += 1;
#endif
    seq = nxt;
  }
#if HM_STAMPS
  if (lane == 0) {                                                 // beyond the real lse: [block][wave][8] floats (the bench allocates it)
    float* st = a.lse + (long)a.nseq * a.heads * a.n + ((long)blockIdx.x * NW + w) * 8;
    st[0] = (float)c_pro; st[1] = (float)c_loop; st[2] = (float)c_epi; st[3] = (float)c_n;
    st[4] = (float)(c_start & 0xffffff); st[5] = (float)(__builtin_readcyclecounter() - c_start); st[6] = (float)(__builtin_amdgcn_s_memrealtime() - r_start); st[7] = (float)(r_start & 0xffffff);
  }
#endif
}

// ------------------------------------------------------------------------------------------------
// backward pass 1: dQ and delta (and, DBL, d(bias)).  Lane = query, accumulator rows = keys, as in the forward.
// ------------------------------------------------------------------------------------------------
// RK1: -lse log2(e) and -delta of the lane's query enter the S and dP accumulators through the matrix pipe as well -- a
// rank-1 f16 MFMA, ones[key][k] x c[k][q] with the constant split into an f16 high and low part on k = 0, 1 (exact to 2^-22
// of its magnitude; a 16-register C block per constant and query block does not fit next to two blocks' accumulators) --
// so the tile's arithmetic is p = exp2(S'), dS = p * dP'.  Without it (the d(bias) form: 12 waves, ~170 registers) they are
// applied by the VALU: p = exp2(S + nlse), dS = p * (dP - delta).
// K / V tiles are fetched as 16 whole 64-byte rows per wave-instruction (1 KiB contiguous in the head-major layout) into a
// wave-private LDS image and read back as MFMA operands: rows for S / dP, transposed for dQ^T += K^T dS^T.
template <bool HAS_BIAS, int QB, int NW, bool DBL, bool RK1, bool ROWLD = true>
__global__ __launch_bounds__(NW * 64, 1) void hm_bwd_dq_kernel(HmArgs a) {
  static_assert(!DBL || QB == 1, "the d(bias) tiles of one query block fill the LDS");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, half = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int T = a.T;
  const int G = (T + QB - 1) / QB;
  int L = xcd_remap(blockIdx.x, gridDim.x);
  const int grp = L % G;
  L /= G;
  const int chunk_id = hm_chunk_order(L % a.nchunks, a.nchunks), head = L / a.nchunks;
  const int seq0 = chunk_id * a.chunk, seq1 = min(a.nseq, seq0 + a.chunk);
  const int nqb = min(QB, T - grp * QB);
  const int q0 = grp * QB * 32;

  half8* bias_l = (half8*)smem;                                    // [QB][T][2][64] half8
  char* after_bias = smem + (size_t)(HAS_BIAS ? QB * T : 0) * 2048;
  float4* dbias_l = (float4*)after_bias;                           // DBL: [T][4][64] float4 = registers 4 j .. 4 j + 3 of lane
  unsigned* lock_l = (unsigned*)(after_bias + (size_t)T * 4096);   // DBL: one word per key tile
  // ROWLD: K and V tiles are fetched as 16 whole rows per instruction into LDS images and every fragment is read back from
  // them; !ROWLD: both arrive as MFMA fragments straight from global memory and only K is written to an image, for the
  // transposed read of the dQ product (6 KiB less LDS traffic per tile: the d(bias) form is LDS-bound)
  constexpr int IMG_BUFS = (DBL && ROWLD) ? 1 : 2;                 // [K 2 KiB | V 2 KiB] per buffer; K alone: two fit
  constexpr int IMG_WAVE = (ROWLD ? 4096 : 2048) * IMG_BUFS;
  char* kimg = after_bias + (DBL ? (size_t)T * 4096 + 128 : 0) + (size_t)w * IMG_WAVE;
  char* vimg = kimg + 2048 * IMG_BUFS;                             // ROWLD only
  if (DBL) {
    for (int id = tid; id < T * 256; id += NW * 64) dbias_l[id] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int id = tid; id < T; id += NW * 64) lock_l[id] = 0u;
  }
  if (HAS_BIAS) fill_bias<true>(bias_l, a.bias + (long)head * a.n * a.n, a.n, q0, nqb, QB, T, tid, NW * 64);
  __shared__ int seq_counter;
  if (tid == 0) seq_counter = seq0;
  __syncthreads();
  half8 idf[2];
  identity_frags(idf, r, half);
  half8 ones01;                                                    // A fragment of ones[key][k], k in {0, 1}
#pragma unroll
  for (int j = 0; j < 8; ++j) ones01[j] = (half == 0 && j < 2) ? (_Float16)1.f : (_Float16)0.f;

  const int crow = lane >> 2, ccol = lane & 3;
  const uint32_t kvoff = ROWLD ? (uint32_t)(crow * 32 + ccol * 8) : (uint32_t)(r * 32 + 8 * half);
  const uint32_t kvstep = ROWLD ? 512u : 16u;                      // second piece: sixteen rows further / the other k-half
  const uint32_t qoff = (uint32_t)((q0 + r) * 32 + 8 * half);
  const uint32_t ooff = (uint32_t)((q0 + r) * a.ldo + head * 32 + 8 * half);
  const uint32_t ist0 = ROWLD ? img_off<32>(crow, ccol) : img_off<32>(r, half);
  const uint32_t ist1 = ROWLD ? ist0 + 1024 : img_off<32>(r, 2 + half);   // sixteen rows further: same swizzle
  const long hstride = (long)a.n * 32, oseq = (long)a.n * a.ldo;
  const int rot = DBL ? (w * T) / NW : 0;                          // this wave's first key tile
  auto phys = [&](int it) { const int t = it + rot; return t >= T ? t - T : t; };

  for (int seq = hm_next_seq(&seq_counter, lane); seq < seq1; seq = hm_next_seq(&seq_counter, lane)) {
    const long base = ((long)seq * a.heads + head) * hstride;
#if HM_ABL_HOT        // timing-only ablation (results wrong): the streamed K/V tiles always come from the chunk's first sequence
    const long sbase = ((long)seq0 * a.heads + head) * hstride;
#else
    const long sbase = base;
#endif
    const bf16_t* kb = a.k + sbase + kvoff;
    const bf16_t* vb = a.v + sbase + kvoff;
    bf16x8 kr[2][2], vr[2][2];
    auto request = [&](int slot, int t) {
      kr[slot][0] = as_bf16x8(*(const short8v*)(kb + (uint32_t)t * 1024));
      kr[slot][1] = as_bf16x8(*(const short8v*)(kb + (uint32_t)t * 1024 + kvstep));
      vr[slot][0] = as_bf16x8(*(const short8v*)(vb + (uint32_t)t * 1024));
      vr[slot][1] = as_bf16x8(*(const short8v*)(vb + (uint32_t)t * 1024 + kvstep));
    };
    request(0, phys(0));
    request(1, phys(T > 1 ? 1 : 0));                               // unconditional: see the forward's request
    bf16x8 qf[QB][2], df[QB][2];
    float nlse2[QB], delta[QB];
    f32x16 dq[QB];
#pragma unroll
    for (int b = 0; b < QB; ++b) {
      const uint32_t bo = (uint32_t)(b < nqb ? b : 0) * 32;        // a short last group re-reads block 0 (results discarded)
      const short8v q0v = *(const short8v*)(a.q + base + qoff + bo * 32), q1v = *(const short8v*)(a.q + base + qoff + bo * 32 + 16);
      const short8v g0v = *(const short8v*)(a.dO + base + qoff + bo * 32), g1v = *(const short8v*)(a.dO + base + qoff + bo * 32 + 16);
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
    half8 cl[QB], cd[QB];                                          // RK1: B fragments c[k][q]: k = 0 high part, k = 1 low part
    if (RK1) {
#pragma unroll
      for (int b = 0; b < QB; ++b) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { cl[b][j] = (_Float16)0.f; cd[b][j] = (_Float16)0.f; }
        const _Float16 lh = (_Float16)nlse2[b], dh = (_Float16)(-delta[b]);
        if (half == 0) {
          cl[b][0] = lh; cl[b][1] = (_Float16)(nlse2[b] - (float)lh);
          cd[b][0] = dh; cd[b][1] = (_Float16)(-delta[b] - (float)dh);
        }
      }
    }
    const f32x16 zero16 = splat16(0.f);

    auto tile = [&](int slot, int it) {
      const int t = phys(it);
      char* ki = kimg + (IMG_BUFS == 2 ? (it & 1) * 2048 : 0);
      char* vi = vimg + (IMG_BUFS == 2 ? (it & 1) * 2048 : 0);
      if (IMG_BUFS == 1) asm volatile("" ::: "memory");             // single image: the previous tile's reads stay above
      *(bf16x8*)(ki + ist0) = kr[slot][0];
      *(bf16x8*)(ki + ist1) = kr[slot][1];
      bf16x8 k0 = kr[slot][0], k1 = kr[slot][1], v0 = vr[slot][0], v1 = vr[slot][1];
      if (ROWLD) {
        *(bf16x8*)(vi + ist0) = vr[slot][0];
        *(bf16x8*)(vi + ist1) = vr[slot][1];
        k0 = row_frag<32>(ki, 0, 0, lane); k1 = row_frag<32>(ki, 0, 1, lane);
        v0 = row_frag<32>(vi, 0, 0, lane); v1 = row_frag<32>(vi, 0, 1, lane);
      }
      const bf16x8 kt0 = tr_frag<32>(ki, 0, 0, 0, lane), kt1 = tr_frag<32>(ki, 0, 1, 0, lane);
      f32x16 dsb;                                                  // DBL: the block's dS tile for the d(bias) update
#pragma unroll
      for (int b = 0; b < QB; ++b) {
        f32x16 S, dP;
        if (RK1) {
          S = mfma32h(ones01, cl[b], zero16);
          dP = mfma32h(ones01, cd[b], zero16);
        } else {
          S = zero16;
          dP = zero16;
        }
        if (HAS_BIAS) {
          const half8 h0 = bias_l[((b * T + t) * 2 + 0) * 64 + lane], h1 = bias_l[((b * T + t) * 2 + 1) * 64 + lane];
          S = mfma32h(h0, idf[0], S);
          S = mfma32h(h1, idf[1], S);
        }
        S = mfma32(k0, qf[b][0], S);
        S = mfma32(k1, qf[b][1], S);
        dP = mfma32(v0, df[b][0], dP);
        dP = mfma32(v1, df[b][1], dP);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          if (RK1) {
            S[i] = exp2_fast(S[i]) * dP[i];                         // dS^T[key][q]
          } else {
            const float pr = exp2_fast(S[i] + nlse2[b]);
            S[i] = pr * (dP[i] - delta[b]);
          }
        }
        const bf16x8 d0 = acc_frag(S, 0), d1 = acc_frag(S, 1);
        dq[b] = mfma32(kt0, d0, dq[b]);
        dq[b] = mfma32(kt1, d1, dq[b]);
        if (DBL) dsb = S;
        __builtin_amdgcn_sched_barrier(0);
      }
      if (DBL) {
        // d(bias) tile t += dS: take the tile's lock (bounded spin: a wave never holds a lock across anything but the
        // few LDS instructions below), read-modify-write, release.  LDS executes a wave's instructions in order, so the
        // releasing store is performed after the tile's stores.
        unsigned* lk = lock_l + t;
        bool got = false;
        for (int spins = 0; spins < (1 << 22); ++spins) {
          unsigned old = 1u;
          if (lane == 0) old = atomicCAS(lk, 0u, 1u);
          if (__builtin_amdgcn_readfirstlane(old) == 0u) { got = true; break; }
          __builtin_amdgcn_s_sleep(1);
        }
        // a lock is held for a dozen LDS instructions; 4 M sleeping spins without getting it means it was never released: abort the
        // launch (the caller sees a launch failure) rather than update the tile unlocked and free a lock this wave does not hold
        if (!got) __builtin_trap();
        asm volatile("" ::: "memory");
        float4* dt = dbias_l + t * 256 + lane;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float4 v = dt[j * 64];
          v.x += dsb[4 * j]; v.y += dsb[4 * j + 1]; v.z += dsb[4 * j + 2]; v.w += dsb[4 * j + 3];
          dt[j * 64] = v;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) *(volatile unsigned*)lk = 0u;
      }
      request(slot, phys(it + 2 < T ? it + 2 : T - 1));            // a whole tile of work ahead of its use (unconditional)
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
      store_rows<32>(a.dq + ((long)seq * a.n + q0 + 32 * b + r) * a.lddq + head * 32, dd, kLn2, lane);
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
// -lse log2(e) and -delta vary along the accumulator ROWS here: a wave copies its sequence's two stat rows into LDS and reads
// them back INTO the S and dP accumulators before their products (4 broadcast ds_read_b128 each per key block).
template <bool HAS_BIAS, int KB, int NW>
__global__ __launch_bounds__(NW * 64, 1) void hm_bwd_dkv_kernel(HmArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, half = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int T = a.T;
  const int G = (T + KB - 1) / KB;
  int L = xcd_remap(blockIdx.x, gridDim.x);
  const int grp = L % G;
  L /= G;
  const int chunk_id = hm_chunk_order(L % a.nchunks, a.nchunks), head = L / a.nchunks;
  const int seq0 = chunk_id * a.chunk, seq1 = min(a.nseq, seq0 + a.chunk);
  const int nkb = min(KB, T - grp * KB);
  const int key0 = grp * KB * 32;

  half8* bias_l = (half8*)smem;                                    // [KB][T][2][64] half8
  char* wave_l = smem + (size_t)(HAS_BIAS ? KB * T : 0) * 2048 + (size_t)w * (4096 + (size_t)a.n * 8);
  char* qimg = wave_l;                                             // [32 q][32 d] bf16
  char* doimg = wave_l + 2048;                                     // [32 q][32 d] bf16
  float* stat_l = (float*)(wave_l + 4096);                         // [n] -lse log2e, [n] -delta
  __shared__ int seq_counter;
  if (tid == 0) seq_counter = seq0;
  if (HAS_BIAS) fill_bias<false>(bias_l, a.bias + (long)head * a.n * a.n, a.n, key0, nkb, KB, T, tid, NW * 64);
  __syncthreads();
  half8 idf[2];
  identity_frags(idf, r, half);

  const uint32_t koff = (uint32_t)((key0 + r) * 32 + 8 * half);
  const uint32_t qoff = (uint32_t)(r * 32 + 8 * half);
  const uint32_t ist0 = img_off<32>(r, half), ist1 = img_off<32>(r, 2 + half);
  const long hstride = (long)a.n * 32;

  for (int seq = hm_next_seq(&seq_counter, lane); seq < seq1; seq = hm_next_seq(&seq_counter, lane)) {
    const long base = ((long)seq * a.heads + head) * hstride;
#if HM_ABL_HOT        // timing-only ablation (results wrong): the streamed Q/dO tiles always come from the chunk's first sequence
    const long sbase = ((long)seq0 * a.heads + head) * hstride;
#else
    const long sbase = base;
#endif
    const bf16_t* qb = a.q + sbase + qoff;
    const bf16_t* dob = a.dO + sbase + qoff;
    bf16x8 qr[2][2], gr[2][2];
    auto request = [&](int slot, int t) {
      qr[slot][0] = as_bf16x8(*(const short8v*)(qb + (uint32_t)t * 1024));
      qr[slot][1] = as_bf16x8(*(const short8v*)(qb + (uint32_t)t * 1024 + 16));
      gr[slot][0] = as_bf16x8(*(const short8v*)(dob + (uint32_t)t * 1024));
      gr[slot][1] = as_bf16x8(*(const short8v*)(dob + (uint32_t)t * 1024 + 16));
    };
    bf16x8 kf[KB][2], vf[KB][2];
#pragma unroll
    for (int b = 0; b < KB; ++b) {
      const uint32_t bo = (uint32_t)(b < nkb ? b : 0) * 1024;      // a short last group re-reads block 0 (results discarded)
      const bf16_t* kp = a.k + base + koff + bo;
      const bf16_t* vp = a.v + base + koff + bo;
      kf[b][0] = as_bf16x8(*(const short8v*)kp);
      kf[b][1] = as_bf16x8(*(const short8v*)(kp + 16));
      vf[b][0] = as_bf16x8(*(const short8v*)vp);
      vf[b][1] = as_bf16x8(*(const short8v*)(vp + 16));
    }
    request(0, 0);
    request(1, T > 1 ? 1 : 0);                                     // unconditional: see the forward's request
    {
      const long stat = ((long)seq * a.heads + head) * a.n;
      for (int i = lane; i < a.n; i += 64) {
        stat_l[i] = -a.lse[stat + i] * kLog2e;
        stat_l[a.n + i] = -a.delta[stat + i];
      }
    }
    f32x16 dk[KB], dv[KB];
#pragma unroll
    for (int b = 0; b < KB; ++b) { zero_acc(dk[b]); zero_acc(dv[b]); }

    auto tile = [&](int slot, int t) {
      asm volatile("" ::: "memory");                               // the image is single: the previous tile's reads stay above
      *(bf16x8*)(qimg + ist0) = qr[slot][0];
      *(bf16x8*)(qimg + ist1) = qr[slot][1];
      *(bf16x8*)(doimg + ist0) = gr[slot][0];
      *(bf16x8*)(doimg + ist1) = gr[slot][1];
      const bf16x8 gt0 = tr_frag<32>(doimg, 0, 0, 0, lane), gt1 = tr_frag<32>(doimg, 0, 1, 0, lane);
      const bf16x8 qt0 = tr_frag<32>(qimg, 0, 0, 0, lane), qt1 = tr_frag<32>(qimg, 0, 1, 0, lane);
#pragma unroll
      for (int b = 0; b < KB; ++b) {
        f32x16 S, dP;                                              // start as the stat rows 8 g + 4 half + (0..3) of query tile t
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 l4 = *(const float4*)(stat_l + 32 * t + 8 * g + 4 * half);
          const float4 d4 = *(const float4*)(stat_l + a.n + 32 * t + 8 * g + 4 * half);
          S[4 * g] = l4.x; S[4 * g + 1] = l4.y; S[4 * g + 2] = l4.z; S[4 * g + 3] = l4.w;
          dP[4 * g] = d4.x; dP[4 * g + 1] = d4.y; dP[4 * g + 2] = d4.z; dP[4 * g + 3] = d4.w;
        }
        if (HAS_BIAS) {
          const half8 h0 = bias_l[((b * T + t) * 2 + 0) * 64 + lane], h1 = bias_l[((b * T + t) * 2 + 1) * 64 + lane];
          S = mfma32h(idf[0], h0, S);                              // S[q][key] = -lse[q] + bias[q][key] + ...
          S = mfma32h(idf[1], h1, S);
        }
        S = mfma32(qr[slot][0], kf[b][0], S);
        S = mfma32(qr[slot][1], kf[b][1], S);
        dP = mfma32(gr[slot][0], vf[b][0], dP);                    // dP[q][key] - delta[q]
        dP = mfma32(gr[slot][1], vf[b][1], dP);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const float pr = exp2_fast(S[i]);
          S[i] = pr;
          dP[i] = pr * dP[i];
        }
        const bf16x8 p0 = acc_frag(S, 0), p1 = acc_frag(S, 1), d0 = acc_frag(dP, 0), d1 = acc_frag(dP, 1);
        dv[b] = mfma32(gt0, p0, dv[b]);
        dv[b] = mfma32(gt1, p1, dv[b]);
        dk[b] = mfma32(qt0, d0, dk[b]);
        dk[b] = mfma32(qt1, d1, dk[b]);
        __builtin_amdgcn_sched_barrier(0);                         // one block's temporaries live at a time
      }
      request(slot, t + 2 < T ? t + 2 : T - 1);                    // a whole tile of work ahead of its use (unconditional)
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
      store_rows<32>(a.dk + row * a.lddk + head * 32, kk, kLn2, lane);
      store_rows<32>(a.dv + row * a.lddv + head * 32, vv, 1.0f, lane);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// B_h: a bound on every |log2-logit| of head h, from the parameters alone.
//   |q'.k| <= qk_mult * max_d |q_scale_d k_scale_d|   (q', k: unit vectors times per-channel scales; q' carries qk_mult =
//   scale * log2 e; +2 % and +0.25 cover the bf16 rounding of both operands), plus the largest |bias| entry in log2 units.
// shift[h] = B_h; shift[heads] = 1 when some B_h exceeds 60 binades (2^s could then leave the range in which f32 row sums
// of 2^s stay finite and normal) or anything is not finite: the no-maximum kernels return at once and the online ones run.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void hm_shift_kernel(const float* __restrict__ qs, const float* __restrict__ ks, int dhead,
                                                        float qk_mult, const float* __restrict__ bias, long count,
                                                        long head_stride, long elem_stride, int heads, float* __restrict__ shift) {
  __shared__ float smax[16], smin[16];
  __shared__ float unsafe;
  float w = 0.f;
  for (int d = 0; d < dhead; ++d) w = fmaxf(w, fabsf(qs[d] * ks[d]));
  const float qkb = qk_mult * w * 1.02f + 0.25f;
  if (threadIdx.x == 0) unsafe = 0.f;
  __syncthreads();
  for (int h = 0; h < heads; ++h) {
    float bmax = bias ? -INFINITY : 0.f, bmin = bias ? INFINITY : 0.f;
    if (bias)
      for (long i = threadIdx.x; i < count; i += 1024) {
        const float v = bias[(long)h * head_stride + i * elem_stride];
        bmax = fmaxf(bmax, v);
        bmin = fminf(bmin, v);
      }
    bmax = wave_max(bmax);
    bmin = -wave_max(-bmin);
    if ((threadIdx.x & 63) == 0) { smax[threadIdx.x >> 6] = bmax; smin[threadIdx.x >> 6] = bmin; }
    __syncthreads();
    if (threadIdx.x == 0) {
      for (int i = 1; i < 16; ++i) { bmax = fmaxf(bmax, smax[i]); bmin = fminf(bmin, smin[i]); }
      const float B = qkb + fmaxf(fabsf(bmax), fabsf(bmin)) * kLog2e;
      shift[h] = B;
      if (!(B <= 60.f)) unsafe = 1.f;                                // also catches NaN
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) shift[heads] = unsafe;
}

int cu_count_hm() {
  static const int n = [] {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) return 256;
    return v;
  }();
  return n;
}

// One workgroup per CU is resident (bias tiles + images in LDS) and every workgroup pays a fixed price (bias fill, d(bias)
// flush), so: few rounds of workgroups, and the FULL-size chunks fill a whole number of rounds -- what is left over makes
// one short chunk whose workgroups finish early instead of a full-size tail (measured on the row-major kernels at 1536
// sequences: 8.4 even rounds 1770 / 4813 us forward / backward, 216 per workgroup 1697 / 4600, exactly 8 chunks 1960 / 5328).
void hm_plan(HmArgs& p, int qb, int nw, int* nblocks) {
  p.T = p.n / 32;
  const long roles = (long)((p.T + qb - 1) / qb) * p.heads;
  const long cus = cu_count_hm();
  long nfull = 0;
  double best = 0.0;
  for (long rr = 1; rr <= 8; ++rr) {
    const long nf = rr * cus / roles;                              // full-size chunks that fit rr rounds
    if (nf < 1) continue;
    const double fill = (double)(roles * nf) / (double)(rr * cus);
    if (fill > best + 0.02) { best = fill; nfull = nf; }           // the fewest rounds among (nearly) equally tight fits
  }
  // sequences are handed out to the waves dynamically (hm_next_seq), so a chunk need not be a multiple of the wave count:
  // nfull (nearly) equal chunks and NO short one.  With chunks rounded down to whole waves (216 of 1536 / 7 = 219.4) the
  // remainder made an eighth, short chunk whose workgroups either ended the launch alone or, dispatched first, pushed a
  // quarter of the full-size ones 78 us back: 1157 us where two rounds of 7 x 220 take ~1030
  long chunk = nfull > 0 ? (p.nseq + nfull - 1) / nfull : p.nseq;
  if (chunk < nw) {                                                // few sequences: as many chunks as fill ~8 rounds
    long nchunks = (8 * cus + roles - 1) / roles;
    if (nchunks < 1) nchunks = 1;
    chunk = (p.nseq + nchunks - 1) / nchunks;
    chunk = (chunk + nw - 1) / nw * nw;
  }
  if (const char* e = getenv("CTCLIP_ATTN_SP_CHUNK")) {            // test hook (include/ctclip_hip.h): ragged chunks
    const int forced = atoi(e);
    if (forced > 0) chunk = forced;
  }
  if (chunk > p.nseq) chunk = p.nseq;
  if (chunk < 1) chunk = 1;
  p.chunk = (int)chunk;
  p.nchunks = (int)((p.nseq + chunk - 1) / chunk);
  *nblocks = (int)(roles * p.nchunks);
}

template <typename K>
int hm_launch(K kernel, const HmArgs& p, int nblocks, int nw, size_t lds, hipStream_t st) {
  if (lds > 160 * 1024) return (int)hipErrorInvalidValue;
  if (lds > 65536) {
    hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(kernel, dim3(nblocks), dim3(nw * 64), lds, st, p);
  return (int)hipGetLastError();
}

template <int QB, int NW>
int hm_fwd_launch(HmArgs a, hipStream_t st) {
  int nblocks = 0;
  hm_plan(a, QB, NW, &nblocks);
  const bool hb = a.bias != nullptr;
  const size_t lds = (size_t)(hb ? QB * a.T : 0) * 2048 + (size_t)NW * 4096;
  int e = 0;
  if (a.shift) {                                                   // the no-maximum kernel; returns at once when flagged
    e = hb ? hm_launch(hm_fwd_kernel<true, true, QB, NW>, a, nblocks, NW, lds, st)
           : hm_launch(hm_fwd_kernel<false, true, QB, NW>, a, nblocks, NW, lds, st);
    if (e) return e;
  }
  // the online kernel: the only one without a bound, and with one it returns at once unless the flag is set
  return hb ? hm_launch(hm_fwd_kernel<true, false, QB, NW>, a, nblocks, NW, lds, st)
            : hm_launch(hm_fwd_kernel<false, false, QB, NW>, a, nblocks, NW, lds, st);
}

bool hm_shape_ok(int n, int nseq, int heads) {
  if (n % 32 || nseq <= 0 || heads <= 0) return false;
  const int T = n / 32;
  return T >= 1 && T <= 24;
}

}  // namespace

extern "C" {

int ctclip_attn_shift(const float* q_scale, const float* k_scale, int dhead, float qk_mult, const float* bias, long bias_count,
                      long bias_head_stride, long bias_elem_stride, int heads, float* shift, void* stream) {
  if (dhead <= 0 || heads <= 0 || !q_scale || !k_scale || !shift) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(hm_shift_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, q_scale, k_scale, dhead, qk_mult, bias,
                     bias_count, bias_head_stride, bias_elem_stride, heads, shift);
  CTCLIP_CHECK_LAUNCH();
}

int ctclip_attn_hm_fwd(const void* q, const void* k, const void* v, void* o, float* lse, const float* bias,
                       const float* shift, int nseq, int n, int heads, long ldo, void* stream) {
  if (!hm_shape_ok(n, nseq, heads) || (ldo & 7)) return (int)hipErrorInvalidValue;
  HmArgs a{};
  a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.o = (bf16_t*)o; a.ldo = ldo; a.lse = lse;
  a.bias = bias; a.shift = shift; a.nseq = nseq; a.n = n; a.heads = heads;
#ifdef CTCLIP_TUNING_KNOBS
  if (CTCLIP_KNOB("CTCLIP_HM_FWD_NW12")) return hm_fwd_launch<2, 12>(a, (hipStream_t)stream);
#endif
  return hm_fwd_launch<2, 8>(a, (hipStream_t)stream);
}

int ctclip_attn_hm_bwd(const void* q, const void* k, const void* v, const void* o, const void* dO, const float* lse,
                       float* delta, void* dq, void* dk, void* dv, const float* bias, float* dbias_dense,
                       const uint16_t* relidx, float* dbias_table, int table_size, int grid_h, int grid_w, int nseq, int n,
                       int heads, long ldo, long lddq, long lddk, long lddv, void* stream) {
  if (!hm_shape_ok(n, nseq, heads) || ((ldo | lddq | lddk | lddv) & 3)) return (int)hipErrorInvalidValue;
  HmArgs a{};
  a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.oin = (const bf16_t*)o; a.ldo = ldo;
  a.dO = (const bf16_t*)dO; a.lse = (float*)lse; a.delta = delta; a.dq = (bf16_t*)dq; a.dk = (bf16_t*)dk; a.dv = (bf16_t*)dv;
  a.lddq = lddq; a.lddk = lddk; a.lddv = lddv; a.bias = bias; a.dbias_dense = dbias_dense; a.relidx = relidx;
  const bool grid_ok = grid_h > 0 && grid_w > 0 && grid_h * grid_w == n && table_size == (2 * grid_h - 1) * (2 * grid_w - 1);
  if ((grid_h > 0 || grid_w > 0) && !grid_ok) return (int)hipErrorInvalidValue;
  a.dbias_table = ((relidx || grid_ok) && !dbias_dense) ? dbias_table : nullptr;
  a.table_size = table_size; a.grid_h = grid_ok ? grid_h : 0; a.grid_w = grid_ok ? grid_w : 0;
  a.nseq = nseq; a.n = n; a.heads = heads;
  const bool hb = bias != nullptr, db = a.dbias_table != nullptr || dbias_dense != nullptr;
  hipStream_t st = (hipStream_t)stream;
  int e;
  if (db) {
    // with a bias gradient: one query block per workgroup, its d(bias) tiles in LDS, 12 waves -- 8 when the tiles of a long
    // row (19 or 20 of them) leave no room for twelve wave images
#ifdef CTCLIP_TUNING_KNOBS
    const bool rk1 = CTCLIP_KNOB("CTCLIP_HM_DBL_RK1") != nullptr;
#else
    const bool rk1 = false;
#endif
    constexpr bool DBL_RK1 = false;
#ifdef CTCLIP_TUNING_KNOBS
    const bool rowld = CTCLIP_KNOB("CTCLIP_HM_DBL_ROWLD") != nullptr;
#else
    const bool rowld = false;
#endif
    auto lds_for = [&](int T, int nw) { return (size_t)(hb ? T : 0) * 2048 + (size_t)T * 4096 + 128 + (size_t)nw * 4096; };
    const int T = n / 32;
    if (lds_for(T, 12) <= 160 * 1024) {
      constexpr int NWF = 12;
      HmArgs p = a;
      int nb = 0;
      hm_plan(p, 1, NWF, &nb);
      if (a.dbias_table && (size_t)table_size * 4 > (size_t)NWF * 4096) return (int)hipErrorInvalidValue;
#ifdef CTCLIP_TUNING_KNOBS
      if (rk1)
        e = hb ? hm_launch(hm_bwd_dq_kernel<true, 1, NWF, true, true>, p, nb, NWF, lds_for(T, NWF), st)
               : hm_launch(hm_bwd_dq_kernel<false, 1, NWF, true, true>, p, nb, NWF, lds_for(T, NWF), st);
      else
#endif
      if (!rowld)
        e = hb ? hm_launch(hm_bwd_dq_kernel<true, 1, NWF, true, DBL_RK1, false>, p, nb, NWF, lds_for(T, NWF), st)
               : hm_launch(hm_bwd_dq_kernel<false, 1, NWF, true, DBL_RK1, false>, p, nb, NWF, lds_for(T, NWF), st);
      else
      e = hb ? hm_launch(hm_bwd_dq_kernel<true, 1, NWF, true, DBL_RK1>, p, nb, NWF, lds_for(T, NWF), st)
             : hm_launch(hm_bwd_dq_kernel<false, 1, NWF, true, DBL_RK1>, p, nb, NWF, lds_for(T, NWF), st);
    } else {
      constexpr int NWF = 8;
      HmArgs p = a;
      int nb = 0;
      hm_plan(p, 1, NWF, &nb);
      if (a.dbias_table && (size_t)table_size * 4 > (size_t)NWF * 4096) return (int)hipErrorInvalidValue;
      e = hb ? hm_launch(hm_bwd_dq_kernel<true, 1, NWF, true, DBL_RK1>, p, nb, NWF, lds_for(T, NWF), st)
             : hm_launch(hm_bwd_dq_kernel<false, 1, NWF, true, DBL_RK1>, p, nb, NWF, lds_for(T, NWF), st);
    }
  } else {
    constexpr int QB = 2, NW = 8;
    HmArgs p = a;
    int nb = 0;
    hm_plan(p, QB, NW, &nb);
    const size_t lds = (size_t)(hb ? QB * p.T : 0) * 2048 + (size_t)NW * 8192;
    e = hb ? hm_launch(hm_bwd_dq_kernel<true, QB, NW, false, true>, p, nb, NW, lds, st)
           : hm_launch(hm_bwd_dq_kernel<false, QB, NW, false, true>, p, nb, NW, lds, st);
  }
  if (e) return e;
  constexpr int KB = 2, NW2 = 8;
  HmArgs p2 = a;
  int nb2 = 0;
  hm_plan(p2, KB, NW2, &nb2);
  const size_t lds2 = (size_t)(hb ? KB * p2.T : 0) * 2048 + (size_t)NW2 * (4096 + (size_t)n * 8);
  return hb ? hm_launch(hm_bwd_dkv_kernel<true, KB, NW2>, p2, nb2, NW2, lds2, st)
            : hm_launch(hm_bwd_dkv_kernel<false, KB, NW2>, p2, nb2, NW2, lds2, st);
}

}  // extern "C"
