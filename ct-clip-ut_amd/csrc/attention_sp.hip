// Sequence-persistent fused attention for gfx950: the CT-ViT spatial shape (reference src/utils/attention.py:155-180,
// n = 576 tokens per frame, d_head = 32, one [heads, n, n] relative-position bias shared by EVERY sequence).
//
// The per-sequence kernels of attention.hip re-read a 32x32 bias tile from L2 for every score tile and reduce d(bias)
// through LDS + global atomics once per (sequence, head) workgroup; at 1536 sequences per step that is ~16 GB of bias
// traffic per pass and ~80 M global atomics.  Here a workgroup owns one (head, 32-row block) of the score matrix for a
// whole CHUNK of sequences instead:
//   * wave w owns `ntw` (<= 3) of the n/32 column tiles of that block; its bias tiles live in registers for the whole
//     chunk (they enter the score MFMA as the C operand, so adding the bias costs no VALU instruction) and so do its
//     d(bias) sums, which are flushed ONCE per workgroup (LDS table -> a few hundred global atomics);
//   * per sequence the waves' partial results (O with its softmax statistics, dQ, dK/dV) are combined through LDS by a
//     conflict-free float4 exchange and written with 8-byte row stores;
//   * everything of sequence i+1 is prefetched a whole iteration ahead: operand fragments that are consumed straight
//     from registers (K rows in forward, V rows in dQ) are re-loaded in place right after their last MFMA, rows that need
//     the transposed LDS read go global -> registers -> wave-private LDS image.
// Orientation / fragment conventions are those of attention.hip (scores key-major in forward and dQ, query-major in
// dK/dV; the f32 accumulator of the first product is the bf16 operand of the second).
//
// Eligible: d_head 32, n % 32 == 0, 4 <= n/32 <= 24, no key mask.  Everything else uses attention.hip.
#include "attn_common.h"
#include <stdlib.h>

namespace {

constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;
constexpr int kMaxWaves = 8;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;  // one 16-byte piece of an image row

// timing ablations (drop a barrier / the combine / the next-sequence fetch): compiled in for tuning builds only
#ifdef CTCLIP_TUNING_KNOBS
#define SP_DBG(bit) (p.dbg & (bit))
#else
#define SP_DBG(bit) false
#endif

struct SpArgs {
  AttnArgs a;
  int T;        // 32-wide tiles per row (n / 32)
  int W;        // waves per workgroup
  int base;     // tiles per wave: base, +1 for the first `rem` waves
  int rem;
  int chunk;    // sequences per workgroup
  float c1;     // scale * log2(e)
  float inv_scale;
  int dbg;      // timing ablations, -DCTCLIP_TUNING_KNOBS builds only (always 0 in the product library)
};

__device__ __forceinline__ float exp2_fast(float x) { return __builtin_amdgcn_exp2f(x); }

// Workgroup barrier that orders LDS traffic only.  __syncthreads() would also drain vmcnt, i.e. wait for the global
// loads of the NEXT sequence that are deliberately left in flight across the barrier.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// partial-result exchange: wave w, lane l stores accumulator registers 4j..4j+3 as one float4 at
// ((w*64 + l)*16 + 4*(j ^ ((l>>2)&3))) -- conflict-free for ds_write_b128 and for the combining read below.
__device__ __forceinline__ void part_store(float* part, int w, int lane, const f32x16& acc) {
  float* base = part + (size_t)(w * 64 + lane) * 16;
  const int sw = (lane >> 2) & 3;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float4 v = make_float4(acc[4 * j], acc[4 * j + 1], acc[4 * j + 2], acc[4 * j + 3]);
    *(float4*)(base + 4 * (j ^ sw)) = v;
  }
}
__device__ __forceinline__ float4 part_load(const float* part, int w, int lane, int j) {
  return *(const float4*)(part + (size_t)(w * 64 + lane) * 16 + 4 * (j ^ ((lane >> 2) & 3)));
}
__device__ __forceinline__ void store_bf16x4(bf16_t* p, float4 v, float mul) {
  uint2 o;
  o.x = pack_bf16x2(v.x * mul, v.y * mul);
  o.y = pack_bf16x2(v.z * mul, v.w * mul);
  *(uint2*)p = o;
}

struct SpIds {
  int blk, head, chunk_id, ntw, t0;
};
// w must be wave-uniform (readfirstlane) so that everything derived from it lives in SGPRs and `u < ntw` is a scalar branch
__device__ __forceinline__ SpIds sp_ids(const SpArgs& p, int w) {
  SpIds r;
  int L = xcd_remap(blockIdx.x, gridDim.x);
  r.blk = L % p.T;
  L /= p.T;
  r.head = L % p.a.heads;
  r.chunk_id = L / p.a.heads;
  r.ntw = p.base + (w < p.rem ? 1 : 0);
  r.t0 = w * p.base + (w < p.rem ? w : p.rem);
  return r;
}

__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.f;
  return z;
}

// bias[head][q0 + r][32*tile + acc_row(i, half)] / scale for the key-major score tile (lane = query)
__device__ __forceinline__ f32x16 bias_tile_km(const AttnArgs& a, int head, int q, int tile, int half, float inv_scale) {
  f32x16 b;
  const float* brow = a.bias + ((long)head * a.n + q) * a.n + 32 * tile + 4 * half;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const float4 t = *(const float4*)(brow + 8 * g);
    b[4 * g] = t.x * inv_scale; b[4 * g + 1] = t.y * inv_scale; b[4 * g + 2] = t.z * inv_scale; b[4 * g + 3] = t.w * inv_scale;
  }
  return b;
}

// sum of the W partial float4s of (lane cl, register group rg)
__device__ __forceinline__ float4 part_sum(const float* pbuf, int W, int cl, int rg) {
  float4 tw[kMaxWaves];
#pragma unroll
  for (int ww = 0; ww < kMaxWaves; ++ww) tw[ww] = ww < W ? part_load(pbuf, ww, cl, rg) : make_float4(0.f, 0.f, 0.f, 0.f);
  float4 acc = tw[0];
#pragma unroll
  for (int ww = 1; ww < kMaxWaves; ++ww) { acc.x += tw[ww].x; acc.y += tw[ww].y; acc.z += tw[ww].z; acc.w += tw[ww].w; }
  return acc;
}

// ------------------------------------------------------------------------------------------------
// forward: O = softmax(q k^T * scale + bias) v, lse
// ------------------------------------------------------------------------------------------------
template <int TPW, bool HAS_BIAS>
__global__ __launch_bounds__(512) void sp_fwd_kernel(SpArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const AttnArgs& a = p.a;
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, half = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nthreads = blockDim.x;
  const SpIds id = sp_ids(p, w);
  const int head = id.head, ntw = id.ntw, t0 = id.t0, q0 = id.blk * 32;
  const int seq0 = id.chunk_id * p.chunk, seq1 = min(a.nseq, seq0 + p.chunk);
  if (seq0 >= seq1) return;

  char* vimg = smem;                                   // [n][32] bf16, rows of a wave's own tiles are wave-private
  char* qbuf = vimg + (size_t)a.n * 64;                // 2 x [32][32] bf16
  float* part = (float*)(qbuf + 4096);                 // 2 x W x 1024 f32
  float* ml = part + (size_t)2 * p.W * 1024;           // 2 x W x {m2[32], l[32]}

  f32x16 bias[TPW];
#pragma unroll
  for (int u = 0; u < TPW; ++u) bias[u] = (HAS_BIAS && u < ntw) ? bias_tile_km(a, head, q0 + r, t0 + u, half, p.inv_scale) : zero16();

  // per-lane element offsets inside one sequence (32-bit); the sequence base is uniform
  const int crow = lane >> 2, ccol = lane & 3;         // 16-byte piece of an image tile: rows crow, crow+16
  const uint32_t voff = (uint32_t)((32 * t0 + crow) * a.ldv + head * 32 + ccol * 8);
  const uint32_t vstep = (uint32_t)(16 * a.ldv);
  const uint32_t koff = (uint32_t)((32 * t0 + r) * a.ldk + head * 32 + 8 * half);
  const uint32_t qoff = (uint32_t)((q0 + (tid >> 2)) * a.ldq + head * 32 + (tid & 3) * 8);   // waves 0,1 only
  const uint32_t vimg_off = img_off<32>(32 * t0 + crow, ccol);    // + 1024 for row +16, + 2048 per tile (swizzle-invariant)
  const long kseq = (long)a.n * a.ldk, vseq = (long)a.n * a.ldv, qseq = (long)a.n * a.ldq;

  u32x4 vrow[TPW][2];
  u32x4 qreg = {0u, 0u, 0u, 0u};
  bf16x8 kf[TPW][2];
  auto ld_all = [&](long seq) {
    const bf16_t* kb = a.k + seq * kseq;
    const bf16_t* vb = a.v + seq * vseq;
#pragma unroll
    for (int u = 0; u < TPW; ++u)
      if (u < ntw) {
        kf[u][0] = as_bf16x8(*(const short8v*)(kb + koff + (uint32_t)(32 * u * a.ldk)));
        kf[u][1] = as_bf16x8(*(const short8v*)(kb + koff + (uint32_t)(32 * u * a.ldk) + 16));
      }
#pragma unroll
    for (int u = 0; u < TPW; ++u)
      if (u < ntw) {
        vrow[u][0] = *(const u32x4*)(vb + voff + (uint32_t)(32 * u * a.ldv));
        vrow[u][1] = *(const u32x4*)(vb + voff + (uint32_t)(32 * u * a.ldv) + vstep);
      }
    if (w < 2) qreg = *(const u32x4*)(a.q + seq * qseq + qoff);
  };
  auto st_staged = [&](int buf) {
#pragma unroll
    for (int u = 0; u < TPW; ++u)
      if (u < ntw) {
        *(u32x4*)(vimg + vimg_off + 2048 * u) = vrow[u][0];
        *(u32x4*)(vimg + vimg_off + 2048 * u + 1024) = vrow[u][1];
      }
    if (w < 2) *(u32x4*)(qbuf + buf * 2048 + img_off<32>(tid >> 2, tid & 3)) = qreg;
  };

  ld_all(seq0);
  st_staged(0);
  __syncthreads();

  for (int seq = seq0; seq < seq1; ++seq) {
    const int buf = (seq - seq0) & 1;
    const long sn = SP_DBG(1) ? seq0 : (seq + 1 < seq1 ? seq + 1 : seq);   // next sequence (the last re-loads its own)
    const bf16x8 qf0 = row_frag<32>(qbuf + buf * 2048, 0, 0, lane), qf1 = row_frag<32>(qbuf + buf * 2048, 0, 1, lane);
    f32x16 S[TPW];
#pragma unroll
    for (int u = 0; u < TPW; ++u)
      if (u < ntw) {
        S[u] = mfma32(kf[u][0], qf0, bias[u]);
        S[u] = mfma32(kf[u][1], qf1, S[u]);
      }
    // every global load of the next sequence is issued here and consumed at the end of this iteration, so no load is
    // in flight across the loop edge (vmcnt is in-order and also counts the combine's stores)
    ld_all(sn);
    float m = -INFINITY;
#pragma unroll
    for (int u = 0; u < TPW; ++u)
      if (u < ntw) {
#pragma unroll
        for (int i = 0; i < 16; ++i) m = fmaxf(m, S[u][i]);
      }
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    const float m2 = m * p.c1;
    float l = 0.f;
#pragma unroll
    for (int u = 0; u < TPW; ++u)
      if (u < ntw) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const float e = exp2_fast(fmaf(S[u][i], p.c1, -m2));
          S[u][i] = e;
          l += e;
        }
      }
    l += __shfl_xor(l, 32, 64);
    f32x16 oacc;
    {
      const bf16x8 p0 = acc_frag(S[0], 0), p1 = acc_frag(S[0], 1);
      oacc = mfma32(tr_frag<32>(vimg, 32 * t0, 0, 0, lane), p0, zero16());
      oacc = mfma32(tr_frag<32>(vimg, 32 * t0, 1, 0, lane), p1, oacc);
    }
#pragma unroll
    for (int u = 1; u < TPW; ++u)
      if (u < ntw) {
        const bf16x8 p0 = acc_frag(S[u], 0), p1 = acc_frag(S[u], 1);
        oacc = mfma32(tr_frag<32>(vimg, 32 * (t0 + u), 0, 0, lane), p0, oacc);
        oacc = mfma32(tr_frag<32>(vimg, 32 * (t0 + u), 1, 0, lane), p1, oacc);
      }
    float* pbuf = part + (size_t)buf * p.W * 1024;
    float* mbuf = ml + (size_t)buf * p.W * 64;
    part_store(pbuf, w, lane, oacc);
    if (half == 0) {
      mbuf[w * 64 + r] = m2;
      mbuf[w * 64 + 32 + r] = l;
    }
    st_staged(buf ^ 1);                                  // own V rows (own transposed reads are done) + the next Q block
    // the K fragments of the next sequence are consumed here too (a register use the compiler waits for): nothing loaded
    // is then in flight across the loop edge, where the only vmcnt wait it could place would also cover the combine's stores
#pragma unroll
    for (int u = 0; u < TPW; ++u) asm volatile("" : "+v"(kf[u][0]), "+v"(kf[u][1]));
    if (!SP_DBG(4)) lds_barrier();
    // combine the W partial rows: O = sum_w 2^(m_w - m) O_w / sum_w 2^(m_w - m) l_w.  Done by the LAST four waves: when
    // the tiles do not divide evenly the first waves carry the extra tile.
    const int idx = (p.W == 8) ? (((w & 2) ? (((w >> 2) << 1) | (w & 1)) : -1) * 64 + lane) : tid - (nthreads - 256);
    if (idx >= 0 && !SP_DBG(8)) {
      const int cl = idx & 63, rg = idx >> 6, q = cl & 31, ch = cl >> 5;
      float mw[kMaxWaves], lw[kMaxWaves];
      float4 tw[kMaxWaves];
#pragma unroll
      for (int ww = 0; ww < kMaxWaves; ++ww) {
        const bool on = ww < p.W;
        mw[ww] = on ? mbuf[ww * 64 + q] : -INFINITY;
        lw[ww] = on ? mbuf[ww * 64 + 32 + q] : 0.f;
        tw[ww] = on ? part_load(pbuf, ww, cl, rg) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      float mm = mw[0];
#pragma unroll
      for (int ww = 1; ww < kMaxWaves; ++ww) mm = fmaxf(mm, mw[ww]);
      float L = 0.f;
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int ww = 0; ww < kMaxWaves; ++ww) {
        const float f = exp2_fast(mw[ww] - mm);
        L = fmaf(f, lw[ww], L);
        acc.x = fmaf(f, tw[ww].x, acc.x); acc.y = fmaf(f, tw[ww].y, acc.y);
        acc.z = fmaf(f, tw[ww].z, acc.z); acc.w = fmaf(f, tw[ww].w, acc.w);
      }
      const long row = (long)seq * a.n + q0 + q;
      store_bf16x4(a.o + row * a.ldo + head * 32 + 8 * rg + 4 * ch, acc, 1.0f / L);
      if (rg == 0 && ch == 0) a.lse[((long)seq * a.heads + head) * a.n + q0 + q] = (mm + __log2f(L)) * kLn2;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// delta[seq, h, q] = sum_d dO[row, h*32+d] * O[row, h*32+d]
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sp_delta_kernel(const bf16_t* __restrict__ dO, const bf16_t* __restrict__ o,
                                                       float* __restrict__ delta, long rows, int n, int heads, long lddo,
                                                       long ldo) {
  // 4 threads per (row, head): 16 bytes of each operand per thread
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long item = gid >> 2;
  const int piece = (int)(gid & 3);
  const bool ok = item < rows * heads;
  float s = 0.f;
  long row = 0;
  int h = 0;
  if (ok) {
    row = item / heads;
    h = (int)(item % heads);
    const short8v x = *(const short8v*)(dO + row * lddo + h * 32 + piece * 8);
    const short8v y = *(const short8v*)(o + row * ldo + h * 32 + piece * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) s = fmaf(bf16_to_f32((bf16_t)x[j]), bf16_to_f32((bf16_t)y[j]), s);
  }
  s += __shfl_xor(s, 1, 64);
  s += __shfl_xor(s, 2, 64);
  if (ok && piece == 0) {
    const long seq = row / n;
    const int q = (int)(row % n);
    delta[(seq * heads + h) * n + q] = s;
  }
}

// ------------------------------------------------------------------------------------------------
// backward pass 1: dQ and d(bias).  Lane = query, accumulator rows = keys (as attn_bwd_dq_kernel).
// ------------------------------------------------------------------------------------------------
template <int TPW, bool HAS_BIAS, bool DBIAS>
__global__ __launch_bounds__(512) void sp_bwd_dq_kernel(SpArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const AttnArgs& a = p.a;
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, half = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nthreads = blockDim.x;
  const SpIds id = sp_ids(p, w);
  const int head = id.head, ntw = id.ntw, t0 = id.t0, q0 = id.blk * 32;
  const int seq0 = id.chunk_id * p.chunk, seq1 = min(a.nseq, seq0 + p.chunk);
  if (seq0 >= seq1) return;

  char* kimg = smem;                                   // [n][32] bf16
  char* qd = kimg + (size_t)a.n * 64;                  // 2 x {Q [32][32], dO [32][32]} bf16
  float* part = (float*)(qd + 8192);                   // 2 x W x 1024 f32 (re-used as the d(bias) table at the end)

  f32x16 bias[TPW], dbacc[TPW];
#pragma unroll
  for (int u = 0; u < TPW; ++u) {
    bias[u] = (HAS_BIAS && u < ntw) ? bias_tile_km(a, head, q0 + r, t0 + u, half, p.inv_scale) : zero16();
    dbacc[u] = zero16();
  }

  const int crow = lane >> 2, ccol = lane & 3;
  const uint32_t koff = (uint32_t)((32 * t0 + crow) * a.ldk + head * 32 + ccol * 8);
  const uint32_t kstep = (uint32_t)(16 * a.ldk);
  const uint32_t voff = (uint32_t)((32 * t0 + r) * a.ldv + head * 32 + 8 * half);
  // waves 0,1 stage the Q block, waves 2,3 the dO block of the next sequence
  const uint32_t qdoff = w < 2 ? (uint32_t)((q0 + (tid >> 2)) * a.ldq + head * 32 + (tid & 3) * 8)
                               : (uint32_t)((q0 + ((tid - 128) >> 2)) * a.lddo + head * 32 + (tid & 3) * 8);
  const uint32_t kimg_off = img_off<32>(32 * t0 + crow, ccol);
  const long kseq = (long)a.n * a.ldk, vseq = (long)a.n * a.ldv, qseq = (long)a.n * a.ldq, doseq = (long)a.n * a.lddo;
  const long stat0 = (long)head * a.n + q0 + r, statseq = (long)a.heads * a.n;

  u32x4 krow[TPW][2];
  u32x4 qdreg = {0u, 0u, 0u, 0u};
  bf16x8 vf[TPW][2];
  float nlse2 = 0.f, delta = 0.f, nlse2_n = 0.f, delta_n = 0.f;
  auto ld_staged = [&](long seq) {
    const bf16_t* kb = a.k + seq * kseq;
#pragma unroll
    for (int u = 0; u < TPW; ++u)
      if (u < ntw) {
        krow[u][0] = *(const u32x4*)(kb + koff + (uint32_t)(32 * u * a.ldk));
        krow[u][1] = *(const u32x4*)(kb + koff + (uint32_t)(32 * u * a.ldk) + kstep);
      }
    if (w < 2) qdreg = *(const u32x4*)(a.q + seq * qseq + qdoff);
    else if (w < 4) qdreg = *(const u32x4*)(a.dO + seq * doseq + qdoff);
    nlse2_n = -a.lse[seq * statseq + stat0] * kLog2e;
    delta_n = a.delta[seq * statseq + stat0];
  };
  auto st_staged = [&](int buf) {
#pragma unroll
    for (int u = 0; u < TPW; ++u)
      if (u < ntw) {
        *(u32x4*)(kimg + kimg_off + 2048 * u) = krow[u][0];
        *(u32x4*)(kimg + kimg_off + 2048 * u + 1024) = krow[u][1];
      }
    if (w < 4) *(u32x4*)(qd + buf * 4096 + (w >> 1) * 2048 + img_off<32>((tid & 127) >> 2, tid & 3)) = qdreg;
    nlse2 = nlse2_n;
    delta = delta_n;
  };
  auto ld_vf = [&](int u, long seq) {
    const bf16_t* vb = a.v + seq * vseq;
    vf[u][0] = as_bf16x8(*(const short8v*)(vb + voff + (uint32_t)(32 * u * a.ldv)));
    vf[u][1] = as_bf16x8(*(const short8v*)(vb + voff + (uint32_t)(32 * u * a.ldv) + 16));
  };

  ld_staged(seq0);
#pragma unroll
  for (int u = 0; u < TPW; ++u)
    if (u < ntw) ld_vf(u, seq0);
  st_staged(0);
  __syncthreads();

  for (int seq = seq0; seq < seq1; ++seq) {
    const int buf = (seq - seq0) & 1;
    const long sn = SP_DBG(1) ? seq0 : (seq + 1 < seq1 ? seq + 1 : seq);
    const char* qdb = qd + buf * 4096;
    const bf16x8 qf0 = row_frag<32>(qdb, 0, 0, lane), qf1 = row_frag<32>(qdb, 0, 1, lane);
    const bf16x8 df0 = row_frag<32>(qdb + 2048, 0, 0, lane), df1 = row_frag<32>(qdb + 2048, 0, 1, lane);
    ld_staged(sn);                                       // consumed at the end of this iteration
    f32x16 dqacc = zero16();
#pragma unroll
    for (int u = 0; u < TPW; ++u)
      if (u < ntw) {
        const int krow0 = 32 * (t0 + u);
        f32x16 s = mfma32(row_frag<32>(kimg, krow0, 0, lane), qf0, bias[u]);
        s = mfma32(row_frag<32>(kimg, krow0, 1, lane), qf1, s);
        f32x16 dp = mfma32(vf[u][0], df0, zero16());
        dp = mfma32(vf[u][1], df1, dp);
        ld_vf(u, sn);                                    // in-place reload for the next sequence
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const float pr = exp2_fast(fmaf(s[i], p.c1, nlse2));
          const float ds = pr * (dp[i] - delta);          // dS^T[key][q]
          if (DBIAS) dbacc[u][i] += ds;
          s[i] = ds;
        }
        const bf16x8 d0 = acc_frag(s, 0), d1 = acc_frag(s, 1);
        dqacc = mfma32(tr_frag<32>(kimg, krow0, 0, 0, lane), d0, dqacc);
        dqacc = mfma32(tr_frag<32>(kimg, krow0, 1, 0, lane), d1, dqacc);
        __builtin_amdgcn_sched_barrier(0);               // keep one tile's temporaries live at a time
      }
    float* pbuf = part + (size_t)buf * p.W * 1024;
    part_store(pbuf, w, lane, dqacc);
    st_staged(buf ^ 1);                                  // own K rows (own reads are done), next Q / dO block, stats
#pragma unroll
    for (int u = 0; u < TPW; ++u) {
      asm volatile("" : "+v"(vf[u][0]), "+v"(vf[u][1]));
    }
    lds_barrier();
    const int idx = tid - (nthreads - 256);
    if (idx >= 0) {
      const int cl = idx & 63, rg = idx >> 6, q = cl & 31, ch = cl >> 5;
      const float4 acc = part_sum(pbuf, p.W, cl, rg);
      const long row = (long)seq * a.n + q0 + q;
      store_bf16x4(a.dq + row * a.lddq + head * 32 + 8 * rg + 4 * ch, acc, a.scale);
    }
  }

  if (DBIAS) {
    const int q = q0 + r;
    if (a.dbias_dense) {
#pragma unroll
      for (int u = 0; u < TPW; ++u)
        if (u < ntw) {
#pragma unroll
          for (int i = 0; i < 16; ++i)
            atomicAdd(a.dbias_dense + ((long)head * a.n + q) * a.n + 32 * (t0 + u) + acc_row(i, half), dbacc[u][i]);
        }
    } else {
      __syncthreads();                                   // every wave is done with `part`
      float* table = part;
      for (int i = tid; i < a.table_size; i += nthreads) table[i] = 0.f;
      __syncthreads();
      const int yq = a.grid_w > 0 ? q / a.grid_w : 0, xq = a.grid_w > 0 ? q % a.grid_w : 0;
#pragma unroll
      for (int u = 0; u < TPW; ++u)
        if (u < ntw) {
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int key = 32 * (t0 + u) + acc_row(i, half);
            int ti;
            if (a.grid_w > 0) {
              const int yk = key / a.grid_w, xk = key % a.grid_w;
              ti = (yq - yk + a.grid_h - 1) * (2 * a.grid_w - 1) + (xq - xk + a.grid_w - 1);
            } else {
              ti = a.relidx[(long)q * a.n + key];
            }
            atomicAdd(&table[ti], dbacc[u][i]);
          }
        }
      __syncthreads();
      for (int i = tid; i < a.table_size; i += nthreads) {
        const float v = table[i];
        if (v != 0.f) atomicAdd(a.dbias_table + (long)head * a.table_size + i, v);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// backward pass 2: dK, dV.  Workgroup = (head, 32-key block); waves own query tiles.  Lane = key.
// ------------------------------------------------------------------------------------------------
template <int TPW, bool HAS_BIAS>
__global__ __launch_bounds__(512) void sp_bwd_dkv_kernel(SpArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const AttnArgs& a = p.a;
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, half = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nthreads = blockDim.x;
  const SpIds id = sp_ids(p, w);
  const int head = id.head, ntw = id.ntw, t0 = id.t0, key0 = id.blk * 32;
  const int seq0 = id.chunk_id * p.chunk, seq1 = min(a.nseq, seq0 + p.chunk);
  if (seq0 >= seq1) return;

  char* qimg = smem;                                   // [n][32] bf16 (wave-private rows)
  char* doimg = qimg + (size_t)a.n * 64;
  char* kvbuf = doimg + (size_t)a.n * 64;              // 2 x {K [32][32], V [32][32]}
  float* stat = (float*)(kvbuf + 8192);                // [-lse*log2e [n], delta [n]] (wave-private rows)
  float* partk = stat + 2 * a.n;                       // W x 1024 f32
  float* partv = partk + (size_t)p.W * 1024;

  f32x16 bias[TPW];                                    // bias[h][q][key] / scale in query-major orientation
#pragma unroll
  for (int u = 0; u < TPW; ++u) {
    bias[u] = zero16();
    if (HAS_BIAS && u < ntw) {
#pragma unroll
      for (int i = 0; i < 16; ++i)
        bias[u][i] = a.bias[((long)head * a.n + 32 * (t0 + u) + acc_row(i, half)) * a.n + key0 + r] * p.inv_scale;
    }
  }

  const int crow = lane >> 2, ccol = lane & 3;
  const int nrows = 32 * ntw;
  const uint32_t qoff = (uint32_t)((32 * t0 + crow) * a.ldq + head * 32 + ccol * 8), qstep = (uint32_t)(16 * a.ldq);
  const uint32_t dooff = (uint32_t)((32 * t0 + crow) * a.lddo + head * 32 + ccol * 8), dostep = (uint32_t)(16 * a.lddo);
  // waves 0,1 stage the K block, waves 2,3 the V block of the next sequence
  const uint32_t kvoff = w < 2 ? (uint32_t)((key0 + (tid >> 2)) * a.ldk + head * 32 + (tid & 3) * 8)
                               : (uint32_t)((key0 + ((tid - 128) >> 2)) * a.ldv + head * 32 + (tid & 3) * 8);
  const uint32_t img0 = img_off<32>(32 * t0 + crow, ccol);
  const long kseq = (long)a.n * a.ldk, vseq = (long)a.n * a.ldv, qseq = (long)a.n * a.ldq, doseq = (long)a.n * a.lddo;
  const long stat0 = (long)head * a.n + 32 * t0 + lane, statseq = (long)a.heads * a.n;

  u32x4 qrow[TPW][2], dorow[TPW][2];
  u32x4 kvreg = {0u, 0u, 0u, 0u};
  float st_l[2] = {0.f, 0.f}, st_d[2] = {0.f, 0.f};
  auto ld_staged = [&](long seq) {
    const bf16_t* qb = a.q + seq * qseq;
    const bf16_t* dob = a.dO + seq * doseq;
#pragma unroll
    for (int u = 0; u < TPW; ++u)
      if (u < ntw) {
        qrow[u][0] = *(const u32x4*)(qb + qoff + (uint32_t)(32 * u * a.ldq));
        qrow[u][1] = *(const u32x4*)(qb + qoff + (uint32_t)(32 * u * a.ldq) + qstep);
        dorow[u][0] = *(const u32x4*)(dob + dooff + (uint32_t)(32 * u * a.lddo));
        dorow[u][1] = *(const u32x4*)(dob + dooff + (uint32_t)(32 * u * a.lddo) + dostep);
      }
#pragma unroll
    for (int j = 0; j < 2; ++j)
      if (lane + 64 * j < nrows) {
        st_l[j] = -a.lse[seq * statseq + stat0 + 64 * j] * kLog2e;
        st_d[j] = a.delta[seq * statseq + stat0 + 64 * j];
      }
    if (w < 2) kvreg = *(const u32x4*)(a.k + seq * kseq + kvoff);
    else if (w < 4) kvreg = *(const u32x4*)(a.v + seq * vseq + kvoff);
  };
  auto st_staged = [&](int buf) {
#pragma unroll
    for (int u = 0; u < TPW; ++u)
      if (u < ntw) {
        *(u32x4*)(qimg + img0 + 2048 * u) = qrow[u][0];
        *(u32x4*)(qimg + img0 + 2048 * u + 1024) = qrow[u][1];
        *(u32x4*)(doimg + img0 + 2048 * u) = dorow[u][0];
        *(u32x4*)(doimg + img0 + 2048 * u + 1024) = dorow[u][1];
      }
#pragma unroll
    for (int j = 0; j < 2; ++j)
      if (lane + 64 * j < nrows) {
        stat[32 * t0 + lane + 64 * j] = st_l[j];
        stat[a.n + 32 * t0 + lane + 64 * j] = st_d[j];
      }
    if (w < 4) *(u32x4*)(kvbuf + buf * 4096 + (w >> 1) * 2048 + img_off<32>((tid & 127) >> 2, tid & 3)) = kvreg;
  };

  ld_staged(seq0);
  st_staged(0);
  __syncthreads();

  for (int seq = seq0; seq < seq1; ++seq) {
    const int buf = (seq - seq0) & 1;
    const long sn = SP_DBG(1) ? seq0 : (seq + 1 < seq1 ? seq + 1 : seq);
    const char* kvb = kvbuf + buf * 4096;
    const bf16x8 kf0 = row_frag<32>(kvb, 0, 0, lane), kf1 = row_frag<32>(kvb, 0, 1, lane);
    const bf16x8 vf0 = row_frag<32>(kvb + 2048, 0, 0, lane), vf1 = row_frag<32>(kvb + 2048, 0, 1, lane);
    ld_staged(sn);                                       // consumed at the end of this iteration
    f32x16 dkacc = zero16(), dvacc = zero16();
#pragma unroll
    for (int u = 0; u < TPW; ++u)
      if (u < ntw) {
        const int qrow0 = 32 * (t0 + u);
        f32x16 s = mfma32(row_frag<32>(qimg, qrow0, 0, lane), kf0, bias[u]);     // S[q][key] + bias
        s = mfma32(row_frag<32>(qimg, qrow0, 1, lane), kf1, s);
        f32x16 dp = mfma32(row_frag<32>(doimg, qrow0, 0, lane), vf0, zero16()); // dP[q][key]
        dp = mfma32(row_frag<32>(doimg, qrow0, 1, lane), vf1, dp);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int qb = qrow0 + 8 * g + 4 * half;
          const float4 l4 = *(const float4*)(stat + qb);
          const float4 d4 = *(const float4*)(stat + a.n + qb);
          const float ls[4] = {l4.x, l4.y, l4.z, l4.w}, de[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float pr = exp2_fast(fmaf(s[4 * g + i], p.c1, ls[i]));
            s[4 * g + i] = pr;
            dp[4 * g + i] = pr * (dp[4 * g + i] - de[i]);
          }
        }
        const bf16x8 p0 = acc_frag(s, 0), p1 = acc_frag(s, 1), d0 = acc_frag(dp, 0), d1 = acc_frag(dp, 1);
        dvacc = mfma32(tr_frag<32>(doimg, qrow0, 0, 0, lane), p0, dvacc);
        dvacc = mfma32(tr_frag<32>(doimg, qrow0, 1, 0, lane), p1, dvacc);
        dkacc = mfma32(tr_frag<32>(qimg, qrow0, 0, 0, lane), d0, dkacc);
        dkacc = mfma32(tr_frag<32>(qimg, qrow0, 1, 0, lane), d1, dkacc);
        __builtin_amdgcn_sched_barrier(0);
      }
    if (!SP_DBG(16)) lds_barrier();                    // the previous sequence's combine has left partk / partv
    part_store(partk, w, lane, dkacc);
    part_store(partv, w, lane, dvacc);
    st_staged(buf ^ 1);                                  // own Q / dO rows + stats (own reads are done), next K / V block
    if (!SP_DBG(4)) lds_barrier();
    if (!SP_DBG(8))
    for (int idx = tid; idx < 512; idx += nthreads) {
      const int which = idx >> 8, cl = idx & 63, rg = (idx >> 6) & 3, key = cl & 31, ch = cl >> 5;
      const float4 acc = part_sum(which ? partv : partk, p.W, cl, rg);
      const long row = (long)seq * a.n + key0 + key;
      if (which) store_bf16x4(a.dv + row * a.lddv + head * 32 + 8 * rg + 4 * ch, acc, 1.0f);
      else store_bf16x4(a.dk + row * a.lddk + head * 32 + 8 * rg + 4 * ch, acc, a.scale);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
bool sp_enabled() {
  static const bool on = [] {
    const char* e = CTCLIP_KNOB("CTCLIP_ATTN_SP");
    return !(e && e[0] == '0');
  }();
  return on;
}

int cu_count() {
  static const int n = [] {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) return 256;
    return v;
  }();
  return n;
}

bool sp_shape_ok(const AttnArgs& a, int dhead) {
  if (!sp_enabled() || dhead != 32 || a.mask || a.drop) return false;
  if (a.n % 32) return false;
  const int T = a.n / 32;
  return T >= 4 && T <= 3 * kMaxWaves;
}

SpArgs sp_plan(const AttnArgs& a, int* nblocks) {
  SpArgs p{};
  p.a = a;
  p.T = a.n / 32;
  p.W = p.T < kMaxWaves ? p.T : kMaxWaves;
  p.base = p.T / p.W;
  p.rem = p.T % p.W;
  p.c1 = a.scale * kLog2e;
  p.inv_scale = 1.0f / a.scale;
  // one workgroup per CU is resident (8 waves, <= 256 registers each): aim at ~4 rounds of workgroups
  const long pairs = (long)p.T * a.heads;
  long nchunks = (4L * cu_count()) / pairs;
  if (nchunks < 1) nchunks = 1;
  if (nchunks > a.nseq) nchunks = a.nseq;
  p.chunk = (int)((a.nseq + nchunks - 1) / nchunks);
  // ... but no more than 32 sequences per workgroup: measured at 1536 sequences, 32-48 per workgroup (27 rounds) is 13 %
  // faster forward and 2 % faster backward than 220 (4 rounds) -- the tail of a 4-round launch costs more than the extra
  // bias loads / d(bias) flushes
  if (p.chunk > 32) p.chunk = 32;
  if (const char* e = CTCLIP_KNOB("CTCLIP_ATTN_SP_DBG")) p.dbg = atoi(e);
  if (const char* e = getenv("CTCLIP_ATTN_SP_CHUNK")) {  // test knob: force the number of sequences per workgroup
    const int forced = atoi(e);
    if (forced > 0) p.chunk = forced < a.nseq ? forced : a.nseq;
  }
  nchunks = (a.nseq + p.chunk - 1) / p.chunk;
  *nblocks = (int)(pairs * nchunks);
  return p;
}

template <typename K>
int sp_launch(K kernel, const SpArgs& p, int nblocks, size_t lds, hipStream_t st) {
  if (lds > 160 * 1024) return (int)hipErrorInvalidValue;
  if (lds > 65536) {
    hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(kernel, dim3(nblocks), dim3(p.W * 64), lds, st, p);
  return (int)hipGetLastError();
}

}  // namespace

int ctclip_attn_sp_fwd(const CtclipAttnArgs& a, int dhead, hipStream_t st) {
  if (!sp_shape_ok(a, dhead)) return -1;
  int nblocks = 0;
  const SpArgs p = sp_plan(a, &nblocks);
  const size_t lds = (size_t)a.n * 64 + 4096 + (size_t)2 * p.W * 4096 + (size_t)2 * p.W * 256;
  const int tpw = p.base + (p.rem ? 1 : 0);
  const bool hb = a.bias != nullptr;
#define SP_FWD(TPW_) (hb ? sp_launch(sp_fwd_kernel<TPW_, true>, p, nblocks, lds, st) : sp_launch(sp_fwd_kernel<TPW_, false>, p, nblocks, lds, st))
  return tpw == 1 ? SP_FWD(1) : tpw == 2 ? SP_FWD(2) : SP_FWD(3);
#undef SP_FWD
}

int ctclip_attn_sp_bwd(const CtclipAttnArgs& a, int dhead, hipStream_t st) {
  if (!sp_shape_ok(a, dhead)) return -1;
  int nblocks = 0;
  const SpArgs p = sp_plan(a, &nblocks);
  const bool table = a.dbias_table != nullptr, dense = a.dbias_dense != nullptr;
  const size_t part_bytes = (size_t)2 * p.W * 4096;
  if (table && (size_t)a.table_size * 4 > part_bytes) return -1;
  if (table && a.grid_w <= 0 && !a.relidx) return -1;
  {
    // both passes must fit the CU's LDS (the dK/dV pass keeps Q and dO images of the whole sequence): decide before
    // anything is launched, the per-sequence kernels of attention.hip take the shape otherwise
    const size_t l1 = (size_t)a.n * 64 + 8192 + part_bytes;
    const size_t l2 = (size_t)2 * a.n * 64 + 8192 + (size_t)2 * a.n * 4 + (size_t)2 * p.W * 4096;
    if (l1 > 160 * 1024 || l2 > 160 * 1024) return -1;
  }
  {
    const long items = (long)a.nseq * a.n * a.heads * 4;
    hipLaunchKernelGGL(sp_delta_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, st, a.dO, a.oin, a.delta,
                       (long)a.nseq * a.n, a.n, a.heads, a.lddo, a.ldo);
  }
  const size_t lds1 = (size_t)a.n * 64 + 8192 + part_bytes;
  const size_t lds2 = (size_t)2 * a.n * 64 + 8192 + (size_t)2 * a.n * 4 + (size_t)2 * p.W * 4096;
  const int tpw = p.base + (p.rem ? 1 : 0);
  const bool hb = a.bias != nullptr, db = table || dense;
  int e;
#define SP_DQ(TPW_)                                                                                                   \
  (hb ? (db ? sp_launch(sp_bwd_dq_kernel<TPW_, true, true>, p, nblocks, lds1, st)                                     \
            : sp_launch(sp_bwd_dq_kernel<TPW_, true, false>, p, nblocks, lds1, st))                                   \
      : (db ? sp_launch(sp_bwd_dq_kernel<TPW_, false, true>, p, nblocks, lds1, st)                                    \
            : sp_launch(sp_bwd_dq_kernel<TPW_, false, false>, p, nblocks, lds1, st)))
#define SP_DKV(TPW_) (hb ? sp_launch(sp_bwd_dkv_kernel<TPW_, true>, p, nblocks, lds2, st) : sp_launch(sp_bwd_dkv_kernel<TPW_, false>, p, nblocks, lds2, st))
  e = tpw == 1 ? SP_DQ(1) : tpw == 2 ? SP_DQ(2) : SP_DQ(3);
  if (e) return e;
  e = tpw == 1 ? SP_DKV(1) : tpw == 2 ? SP_DKV(2) : SP_DKV(3);
#undef SP_DQ
#undef SP_DKV
  return e;
}
