// Fused attention for gfx950: softmax(q k^T * scale + bias[h,i,j] + mask[seq,j]) v, forward and backward.
//
// Serves the three attention shapes of the CT-CLIP step:
//   CT-ViT spatial  (reference src/utils/attention.py:155-180): n = 576, d_head = 32, bias [8,576,576]
//   CT-ViT temporal (same code, no bias):                       n = 24,  d_head = 32
//   BERT self-attention (transformers BertSelfAttention):       n = L,   d_head = 64, additive key mask
// The cosine-sim l2norm / q_scale / k_scale / `scale = 8` of attention.py:151-155 are applied to q,k
// by ctclip_headnorm_* before this kernel, so here `scale` is 1 for CT-ViT and 1/sqrt(d) for BERT.
//
// Layout: q,k,v,o,do are [nseq*n, ld] bf16 row-major, head h occupying columns h*D .. h*D+D-1
// (the 'b n (h d)' layout the projections produce, attention.py:144,180) -- heads are never split
// into separate tensors.
//
// Structure (one workgroup per (row-block, head, sequence); each wave owns 32 rows):
//   * K and V (forward, dQ pass) or Q and dO (dK/dV pass) of the whole sequence are staged once in
//     LDS as swizzled [n][D] images that serve both ds_read_b128 row reads and ds_read_b64_tr_b16
//     transposed reads without bank conflicts.
//   * scores are computed "key-major" (S^T = K Q^T) in the forward/dQ passes so that a query's
//     row statistics live in one lane (+ its lane^32 partner): softmax needs one cross-lane op.
//   * every second product consumes the f32 accumulator of the first directly as an MFMA operand
//     (v_cvt to bf16, no LDS round trip): O^T = V^T P^T, dQ^T = K^T dS^T, dV^T = dO^T P, dK^T = Q^T dS.
//   * the n x n probability matrix is never written; backward recomputes it from the saved
//     log-sum-exp.  d(bias) is reduced on chip into a [heads][R] relative-position table through
//     LDS atomics (the CT-ViT bias has only (2h-1)(2w-1) distinct values per head), or, for
//     arbitrary biases, added to a dense [heads,n,n] buffer with global atomics.
#include "attn_common.h"
#include <stdlib.h>

namespace {

// additive terms of one 32x32 score tile in "row of accumulator = key, lane = query" orientation:
// add[reg] = bias[head][q][key] + mask[key]  (0 where absent).  Issued one tile AHEAD of its use so the L2
// latency of the bias rows hides behind the previous tile's MFMA + softmax work.
__device__ __forceinline__ void load_addend(float (&add)[16], const AttnArgs& a, int head, int seq, int qrow_c, int key_base,
                                            int half) {
  const float* brow = a.bias ? a.bias + ((long)head * a.n + qrow_c) * a.n : nullptr;
  const float* mrow = a.mask ? a.mask + (long)seq * a.n : nullptr;
#pragma unroll
  for (int g4 = 0; g4 < 4; ++g4) {
    const int k0 = key_base + 8 * g4 + 4 * half;
    float b[4] = {0.f, 0.f, 0.f, 0.f};
    if (brow) {
      if (((a.n & 3) == 0) && k0 + 3 < a.n) {
        const float4 t = *(const float4*)(brow + k0);
        b[0] = t.x; b[1] = t.y; b[2] = t.z; b[3] = t.w;
      } else {
        // unconditional loads from a clamped index, zeros selected afterwards: a load inside `if (key < n)` is a predicated
        // block of its own ending in a full wait (see gfrag)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float t = brow[k0 + i < a.n ? k0 + i : a.n - 1];
          b[i] = (k0 + i < a.n) ? t : 0.f;
        }
      }
    }
    if (mrow) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float t = mrow[k0 + i < a.n ? k0 + i : a.n - 1];
        b[i] += (k0 + i < a.n) ? t : 0.f;
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) add[4 * g4 + i] = b[i];
  }
}

// v[reg] = acc*scale + add[reg]; keys >= n -> -inf
__device__ __forceinline__ void apply_scores(f32x16& s, const float (&add)[16], const AttnArgs& a, int key_base, int half) {
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int key = key_base + acc_row(i, half);
    s[i] = (key < a.n) ? s[i] * a.scale + add[i] : -INFINITY;
  }
}

// attention-probability dropout (transformers BertSelfAttention: `attention_probs = self.dropout(attention_probs)`): the
// caller draws the keep flags, the kernels only apply them, so forward and backward see the same mask by construction.
// keep[reg] = drop_scale where the flag of (query row, key key_base + acc_row(reg)) is set, else 0  ("lane = query").
__device__ __forceinline__ void load_keep(float (&keep)[16], const AttnArgs& a, int head, int seq, int qrow_c, int key_base,
                                          int half) {
  const uint8_t* row = a.drop + (((long)seq * a.heads + head) * a.n + qrow_c) * a.n;
#pragma unroll
  for (int g4 = 0; g4 < 4; ++g4) {
    const int k0 = key_base + 8 * g4 + 4 * half;
    if (((a.n & 3) == 0) && k0 + 3 < a.n) {
      const uint32_t w = *(const uint32_t*)(row + k0);
#pragma unroll
      for (int i = 0; i < 4; ++i) keep[4 * g4 + i] = ((w >> (8 * i)) & 0xffu) ? a.drop_scale : 0.f;
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) keep[4 * g4 + i] = (k0 + i < a.n && row[k0 + i]) ? a.drop_scale : 0.f;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(D == 32 ? 576 : 256) void attn_fwd_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_base[];
  char* smem = smem_base;
  constexpr int KS = D / 16, DT = D / 32;
  // the block is a.hpb independent groups of waves, one head each (short sequences: one wave per head would otherwise
  // mean one 64-thread workgroup per (sequence, head) and half-used 128-byte lines)
  const int nthr = blockDim.x / a.hpb, sub = threadIdx.x / nthr;
  const int tid = threadIdx.x - sub * nthr, lane = tid & 63, wave = tid >> 6, nwaves = nthr >> 6;
  const int head = blockIdx.y * a.hpb + sub, seq = blockIdx.z;
  smem += (size_t)sub * a.lds_per_head;
  const int half = lane >> 5;
  char* vimg = smem;
  const long row_base = (long)seq * a.n;
  load_image<D>(vimg, a.v + row_base * a.ldv + head * D, a.ldv, a.n, a.n_pad, tid, nthr);
  __syncthreads();

  const int q0 = (blockIdx.x * nwaves + wave) * 32;
  if (q0 >= a.n) return;  // no barrier after this point
  const int qrow = q0 + (lane & 31);
  const bool valid = qrow < a.n;
  const int qrow_c = valid ? qrow : a.n - 1;
  bf16x8 qf[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) qf[s] = gfrag(a.q + (row_base + qrow_c) * a.ldq + head * D, s, lane, valid);

  float m = -INFINITY, l = 0.f;
  f32x16 oacc[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) zero_acc(oacc[dt]);

  // K is only read row-wise: its fragments come straight from L2 (every wave of the workgroup reads the same rows),
  // one tile ahead, which leaves LDS to the V image and doubles the workgroups resident per CU.
  const bf16_t* kbase = a.k + row_base * a.ldk + head * D;
  auto load_k = [&](bf16x8 (&dst)[KS], int kt) {
    const int key = kt + (lane & 31);
    const bool ok = key < a.n;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) dst[ks] = gfrag(kbase + (long)(ok ? key : 0) * a.ldk, ks, lane, ok);
  };
  // Prefetch without double buffers: each buffer is re-loaded for tile kt+32 immediately after its last use in tile
  // kt, so the loads have the whole softmax + PV phase to land and the kernel stays under 96 VGPRs (2 workgroups/CU).
  float add[16];
  bf16x8 kf[KS];
  load_addend(add, a, head, seq, qrow_c, 0, half);
  load_k(kf, 0);
  for (int kt = 0; kt < a.n_pad; kt += 32) {
    f32x16 s;
    zero_acc(s);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) s = mfma32(kf[ks], qf[ks], s);
    if (kt + 32 < a.n_pad) load_k(kf, kt + 32);
    apply_scores(s, add, a, kt, half);
    if (kt + 32 < a.n_pad) load_addend(add, a, head, seq, qrow_c, kt + 32, half);
    float tmax = s[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) tmax = fmaxf(tmax, s[i]);
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float mnew = fmaxf(m, tmax);
    const float alpha = (m == -INFINITY) ? 0.f : __expf(m - mnew);
    float psum = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float p = (mnew == -INFINITY) ? 0.f : __expf(s[i] - mnew);
      s[i] = p;
      psum += p;
    }
    psum += __shfl_xor(psum, 32, 64);
    l = l * alpha + psum;                                  // the softmax normaliser is over ALL keys, dropped or not
    m = mnew;
    if (a.drop) {
      float keep[16];
      load_keep(keep, a, head, seq, qrow_c, kt, half);
#pragma unroll
      for (int i = 0; i < 16; ++i) s[i] *= keep[i];
    }
    const bf16x8 p0 = acc_frag(s, 0), p1 = acc_frag(s, 1);
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
#pragma unroll
      for (int i = 0; i < 16; ++i) oacc[dt][i] *= alpha;
      oacc[dt] = mfma32(tr_frag<D>(vimg, kt, 0, dt, lane), p0, oacc[dt]);
      oacc[dt] = mfma32(tr_frag<D>(vimg, kt, 1, dt, lane), p1, oacc[dt]);
    }
  }
  if (valid) {
    const float inv = 1.0f / l;
    store_rows<D>(a.o + (row_base + qrow) * a.ldo + head * D, oacc, inv, lane);
    if (half == 0) a.lse[((long)seq * a.heads + head) * a.n + qrow] = m + __logf(l);
  }
}

// ------------------------------------------------------------------------------------------------
// backward pass 1: dQ, delta = rowsum(dO * O), d(bias)
// ------------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(D == 32 ? 384 : 256) void attn_bwd_dq_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_base[];
  char* smem = smem_base;
  constexpr int KS = D / 16, DT = D / 32;
  // the block is a.hpb independent groups of waves, one head each (short sequences: one wave per head would otherwise
  // mean one 64-thread workgroup per (sequence, head) and half-used 128-byte lines)
  const int nthr = blockDim.x / a.hpb, sub = threadIdx.x / nthr;
  const int tid = threadIdx.x - sub * nthr, lane = tid & 63, wave = tid >> 6, nwaves = nthr >> 6;
  const int head = blockIdx.y * a.hpb + sub, seq = blockIdx.z;
  smem += (size_t)sub * a.lds_per_head;
  const int half = lane >> 5;
  char* kimg = smem;
  float* table = (float*)(smem + (size_t)a.n_pad * D * 2);
  int* keyoff = (int*)(table + ((a.table_size + 3) & ~3));
  const long row_base = (long)seq * a.n;
  load_image<D>(kimg, a.k + row_base * a.ldk + head * D, a.ldk, a.n, a.n_pad, tid, nthr);
  if (a.dbias_table) {
    for (int i = tid; i < a.table_size; i += nthr) table[i] = 0.f;
    if (a.grid_w > 0)
      for (int i = tid; i < a.n_pad; i += nthr) keyoff[i] = (i / a.grid_w) * (2 * a.grid_w - 1) + (i % a.grid_w);
  }
  __syncthreads();

  const int q0 = (blockIdx.x * nwaves + wave) * 32;
  const bool wave_active = q0 < a.n;  // inactive waves still reach the final barrier
  if (wave_active) {
    const int qrow = q0 + (lane & 31);
    const bool valid = qrow < a.n;
    const int qrow_c = valid ? qrow : a.n - 1;
    const long stat = ((long)seq * a.heads + head) * a.n + qrow_c;
    bf16x8 qf[KS], dof[KS];
    float dsum = 0.f;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      qf[s] = gfrag(a.q + (row_base + qrow_c) * a.ldq + head * D, s, lane, valid);
      dof[s] = gfrag(a.dO + (row_base + qrow_c) * a.lddo + head * D, s, lane, valid);
      const bf16x8 of = gfrag(a.oin + (row_base + qrow_c) * a.ldo + head * D, s, lane, valid);
#pragma unroll
      for (int j = 0; j < 8; ++j) dsum += (float)dof[s][j] * (float)of[j];
    }
    dsum += __shfl_xor(dsum, 32, 64);
    if (valid && half == 0) a.delta[stat] = dsum;
    const float lse = a.lse[stat];
    const int xq = (a.grid_w > 0) ? qrow_c % a.grid_w : 0;
    const int qoff = (a.grid_w > 0) ? (qrow_c / a.grid_w + a.grid_h - 1) * (2 * a.grid_w - 1) + (xq + a.grid_w - 1) : 0;

    f32x16 dqacc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) zero_acc(dqacc[dt]);

    // V is only read row-wise here: fragments straight from L2, one tile ahead (see attn_fwd_kernel)
    const bf16_t* vbase = a.v + row_base * a.ldv + head * D;
    auto load_v = [&](bf16x8 (&dst)[KS], int kt) {
      const int key = kt + (lane & 31);
      const bool ok = key < a.n;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) dst[ks] = gfrag(vbase + (long)(ok ? key : 0) * a.ldv, ks, lane, ok);
    };
    float add[16];
    bf16x8 vf[KS];
    load_addend(add, a, head, seq, qrow_c, 0, half);
    load_v(vf, 0);
    for (int kt = 0; kt < a.n_pad; kt += 32) {
      f32x16 s, dp;
      zero_acc(s);
      zero_acc(dp);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        s = mfma32(row_frag<D>(kimg, kt, ks, lane), qf[ks], s);
        dp = mfma32(vf[ks], dof[ks], dp);
      }
      if (kt + 32 < a.n_pad) load_v(vf, kt + 32);          // re-load in place right after the last use
      apply_scores(s, add, a, kt, half);
      if (kt + 32 < a.n_pad) load_addend(add, a, head, seq, qrow_c, kt + 32, half);
      if (a.drop) {                                         // d(P) reaches only the kept entries, scaled like them
        float keep[16];
        load_keep(keep, a, head, seq, qrow_c, kt, half);
#pragma unroll
        for (int i = 0; i < 16; ++i) dp[i] *= keep[i];
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float p = valid ? __expf(s[i] - lse) : 0.f;  // -inf scores -> 0
        s[i] = p * (dp[i] - dsum);                          // dS^T[key][q]
      }
      if (a.dbias_table && a.grid_w > 0 && (a.grid_w & 3) == 0) {
        // d(table)[idx(q,key)] += dS.  LDS float atomics are slow (~150 cycles per wave-instruction, measured), so
        // contributions that share a table entry are summed across lanes first: idx(q+1, key+1) == idx(q, key) while q
        // and key stay inside their image rows, so for each group of 4 consecutive keys (never straddling a key row:
        // grid_w % 4 == 0) lane L collects element j from lane L+j -- one full-width atomic per group instead of four.
        // Elements whose collector does not exist (wave edge / start of an image row) are added by their owner.
        const int L = lane & 31;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int k0 = kt + 8 * g4 + 4 * half;
          if (k0 >= a.n) continue;
          const int base = qoff - keyoff[k0];
          float u = s[4 * g4];
#pragma unroll
          for (int j = 1; j < 4; ++j) {
            const float other = __shfl_down(s[4 * g4 + j], j, 32);
            if (L + j < 32 && xq + j < a.grid_w) u += other;
          }
          if (valid) atomicAdd(&table[base], u);
#pragma unroll
          for (int j = 1; j < 4; ++j)
            if (valid && (L < j || xq < j)) atomicAdd(&table[base - j], s[4 * g4 + j]);
        }
      } else if (valid && a.dbias_table && a.grid_w > 0) {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int k0 = kt + 8 * g4 + 4 * half;
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (k0 + i < a.n) atomicAdd(&table[qoff - keyoff[k0 + i]], s[4 * g4 + i]);
        }
      } else if (valid && (a.dbias_dense || a.dbias_table)) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int key = kt + acc_row(i, half);
          if (key < a.n) {
            if (a.dbias_dense) atomicAdd(a.dbias_dense + ((long)head * a.n + qrow) * a.n + key, s[i]);
            else atomicAdd(&table[a.relidx[(long)qrow * a.n + key]], s[i]);
          }
        }
      }
      const bf16x8 d0 = acc_frag(s, 0), d1 = acc_frag(s, 1);
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        dqacc[dt] = mfma32(tr_frag<D>(kimg, kt, 0, dt, lane), d0, dqacc[dt]);
        dqacc[dt] = mfma32(tr_frag<D>(kimg, kt, 1, dt, lane), d1, dqacc[dt]);
      }
    }
    if (valid) store_rows<D>(a.dq + (row_base + qrow) * a.lddq + head * D, dqacc, a.scale, lane);
  }
  if (a.dbias_table) {
    __syncthreads();
    for (int i = tid; i < a.table_size; i += nthr) {
      const float v = table[i];
      if (v != 0.f) atomicAdd(a.dbias_table + (long)head * a.table_size + i, v);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// backward pass 2: dK, dV (each wave owns 32 keys, sweeps all query tiles)
// ------------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(D == 32 ? 384 : 256) void attn_bwd_dkv_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_base[];
  char* smem = smem_base;
  constexpr int KS = D / 16, DT = D / 32;
  // the block is a.hpb independent groups of waves, one head each (short sequences: one wave per head would otherwise
  // mean one 64-thread workgroup per (sequence, head) and half-used 128-byte lines)
  const int nthr = blockDim.x / a.hpb, sub = threadIdx.x / nthr;
  const int tid = threadIdx.x - sub * nthr, lane = tid & 63, wave = tid >> 6, nwaves = nthr >> 6;
  const int head = blockIdx.y * a.hpb + sub, seq = blockIdx.z;
  smem += (size_t)sub * a.lds_per_head;
  const int half = lane >> 5;
  char* qimg = smem;
  char* doimg = smem + (size_t)a.n_pad * D * 2;
  float* lse_s = (float*)(smem + (size_t)a.n_pad * D * 4);
  float* del_s = lse_s + a.n_pad;
  const long row_base = (long)seq * a.n;
  load_image<D>(qimg, a.q + row_base * a.ldq + head * D, a.ldq, a.n, a.n_pad, tid, nthr);
  load_image<D>(doimg, a.dO + row_base * a.lddo + head * D, a.lddo, a.n, a.n_pad, tid, nthr);
  const long stat_base = ((long)seq * a.heads + head) * a.n;
  for (int i = tid; i < a.n_pad; i += nthr) {
    lse_s[i] = (i < a.n) ? a.lse[stat_base + i] : INFINITY;   // exp(v - inf) = 0 for padded queries
    del_s[i] = (i < a.n) ? a.delta[stat_base + i] : 0.f;
  }
  __syncthreads();

  const int key0 = (blockIdx.x * nwaves + wave) * 32;
  if (key0 >= a.n) return;
  const int key = key0 + (lane & 31);
  const bool valid = key < a.n;
  const int key_c = valid ? key : a.n - 1;
  bf16x8 kf[KS], vf[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    kf[s] = gfrag(a.k + (row_base + key_c) * a.ldk + head * D, s, lane, valid);
    vf[s] = gfrag(a.v + (row_base + key_c) * a.ldv + head * D, s, lane, valid);
  }
  const float mval = (a.mask && valid) ? a.mask[(long)seq * a.n + key] : 0.f;
  const float* bcol = a.bias ? a.bias + (long)head * a.n * a.n + key_c : nullptr;
  const uint8_t* drow = a.drop ? a.drop + ((long)seq * a.heads + head) * a.n * a.n + key_c : nullptr;

  f32x16 dkacc[DT], dvacc[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) {
    zero_acc(dkacc[dt]);
    zero_acc(dvacc[dt]);
  }
  float bcur[16], bnxt[16];
  auto load_bcol = [&](float (&dst)[16], int qt) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int q = qt + acc_row(i, half);
      dst[i] = (bcol && q < a.n) ? bcol[(long)q * a.n] : 0.f;
    }
  };
  load_bcol(bcur, 0);
  for (int qt = 0; qt < a.n_pad; qt += 32) {
    if (qt + 32 < a.n_pad) load_bcol(bnxt, qt + 32);
    f32x16 s, dp;
    zero_acc(s);
    zero_acc(dp);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      s = mfma32(row_frag<D>(qimg, qt, ks, lane), kf[ks], s);     // S[q][key]
      dp = mfma32(row_frag<D>(doimg, qt, ks, lane), vf[ks], dp);  // dP[q][key]
    }
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const int qb = qt + 8 * g4 + 4 * half;
      const float4 l4 = *(const float4*)(lse_s + qb);
      const float4 d4 = *(const float4*)(del_s + qb);
      const float ls[4] = {l4.x, l4.y, l4.z, l4.w}, de[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float v = s[4 * g4 + i] * a.scale + mval + bcur[4 * g4 + i];
        const float p = valid ? __expf(v - ls[i]) : 0.f;
        float keep = 1.f;                                   // dropout flag of (query qb + i, this lane's key)
        if (a.drop) keep = (qb + i < a.n && drow[(long)(qb + i) * a.n]) ? a.drop_scale : 0.f;
        s[4 * g4 + i] = p * keep;                           // what multiplied V in the forward
        dp[4 * g4 + i] = p * (dp[4 * g4 + i] * keep - de[i]);
      }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) bcur[i] = bnxt[i];
    const bf16x8 p0 = acc_frag(s, 0), p1 = acc_frag(s, 1), d0 = acc_frag(dp, 0), d1 = acc_frag(dp, 1);
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      dvacc[dt] = mfma32(tr_frag<D>(doimg, qt, 0, dt, lane), p0, dvacc[dt]);
      dvacc[dt] = mfma32(tr_frag<D>(doimg, qt, 1, dt, lane), p1, dvacc[dt]);
      dkacc[dt] = mfma32(tr_frag<D>(qimg, qt, 0, dt, lane), d0, dkacc[dt]);
      dkacc[dt] = mfma32(tr_frag<D>(qimg, qt, 1, dt, lane), d1, dkacc[dt]);
    }
  }
  if (valid) {
    store_rows<D>(a.dk + (row_base + key) * a.lddk + head * D, dkacc, a.scale, lane);
    store_rows<D>(a.dv + (row_base + key) * a.lddv + head * D, dvacc, 1.0f, lane);
  }
}

// ------------------------------------------------------------------------------------------------
// materialise probabilities (only for callers that ask Attention.forward for them, attention.py:182)
// ------------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(D == 32 ? 576 : 256) void attn_probs_kernel(AttnArgs a, float* __restrict__ probs) {
  extern __shared__ __attribute__((aligned(16))) char smem_base[];
  char* smem = smem_base;
  constexpr int KS = D / 16;
  // the block is a.hpb independent groups of waves, one head each (short sequences: one wave per head would otherwise
  // mean one 64-thread workgroup per (sequence, head) and half-used 128-byte lines)
  const int nthr = blockDim.x / a.hpb, sub = threadIdx.x / nthr;
  const int tid = threadIdx.x - sub * nthr, lane = tid & 63, wave = tid >> 6, nwaves = nthr >> 6;
  const int head = blockIdx.y * a.hpb + sub, seq = blockIdx.z;
  smem += (size_t)sub * a.lds_per_head;
  const int half = lane >> 5;
  char* kimg = smem;
  const long row_base = (long)seq * a.n;
  load_image<D>(kimg, a.k + row_base * a.ldk + head * D, a.ldk, a.n, a.n_pad, tid, nthr);
  __syncthreads();
  const int q0 = (blockIdx.x * nwaves + wave) * 32;
  if (q0 >= a.n) return;
  const int qrow = q0 + (lane & 31);
  const bool valid = qrow < a.n;
  const int qrow_c = valid ? qrow : a.n - 1;
  bf16x8 qf[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) qf[s] = gfrag(a.q + (row_base + qrow_c) * a.ldq + head * D, s, lane, valid);
  const float lse = a.lse[((long)seq * a.heads + head) * a.n + qrow_c];
  float* prow = probs + (((long)seq * a.heads + head) * a.n + qrow_c) * a.n;
  for (int kt = 0; kt < a.n_pad; kt += 32) {
    f32x16 s;
    zero_acc(s);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) s = mfma32(row_frag<D>(kimg, kt, ks, lane), qf[ks], s);
    float add[16];
    load_addend(add, a, head, seq, qrow_c, kt, half);
    apply_scores(s, add, a, kt, half);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int key = kt + acc_row(i, half);
      if (valid && key < a.n) prow[key] = __expf(s[i] - lse);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// whole backward in one launch for sequences of at most 32 rows and d_head 32 (the temporal attention of the CT-ViT,
// n = 24, ctvit.py:283-297 -> attention.py:144-183): one wave owns one (sequence, head).  With a single 32x32 score
// tile there is nothing to sweep, so the two-pass split (and its second read of q/k/v/dO, the delta round trip through
// HBM and the second launch) buys nothing; the wave instead forms the tile in both orientations -- [key][q] feeds dQ,
// [q][key] feeds dK and dV as the accumulator-as-operand trick needs -- from the same ten operand fragments, which
// serve as A or B operand alike.  Three wave-private 2 KiB images (K, Q, dO) supply the transposed reads; no block
// barrier anywhere.  Not used when a bias gradient is requested.
// ------------------------------------------------------------------------------------------------
constexpr int SMALL_WAVE_LDS = 3 * 2048 + 256;
__global__ __launch_bounds__(512) void attn_small_bwd_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_base[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const long item = (long)blockIdx.x * (blockDim.x >> 6) + wave;
  if (item >= (long)a.nseq * a.heads) return;
  const int seq = (int)(item / a.heads), head = (int)(item % a.heads);
  const int r = lane & 31, half = lane >> 5;
  char* kimg = smem_base + (size_t)wave * SMALL_WAVE_LDS;
  char* qimg = kimg + 2048;
  char* doimg = qimg + 2048;
  float* lse_s = (float*)(doimg + 2048);
  float* del_s = lse_s + 32;
  const bool valid = r < a.n;                       // lane's row: a query in the first orientation, a key in the second
  const int rc = valid ? r : a.n - 1;
  const long row = (long)seq * a.n + rc;
  const long hoff = (long)head * 32;
  bf16x8 qf[2], kf[2], vf[2], dof[2], of[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    qf[s] = gfrag(a.q + row * a.ldq + hoff, s, lane, valid);
    kf[s] = gfrag(a.k + row * a.ldk + hoff, s, lane, valid);
    vf[s] = gfrag(a.v + row * a.ldv + hoff, s, lane, valid);
    dof[s] = gfrag(a.dO + row * a.lddo + hoff, s, lane, valid);
    of[s] = gfrag(a.oin + row * a.ldo + hoff, s, lane, valid);
  }
  const long stat = ((long)seq * a.heads + head) * a.n;
  const float lse = valid ? a.lse[stat + r] : INFINITY;    // exp(v - inf) = 0 for padded queries
  float dsum = 0.f;
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int j = 0; j < 8; ++j) dsum += (float)dof[s][j] * (float)of[s][j];
  dsum += __shfl_xor(dsum, 32, 64);
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    *(bf16x8*)(kimg + img_off<32>(r, 2 * s + half)) = kf[s];
    *(bf16x8*)(qimg + img_off<32>(r, 2 * s + half)) = qf[s];
    *(bf16x8*)(doimg + img_off<32>(r, 2 * s + half)) = dof[s];
  }
  if (half == 0) {
    lse_s[r] = lse;
    del_s[r] = dsum;
    if (valid && a.delta) a.delta[stat + r] = dsum;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

  f32x16 s, dp, out[1];
  // ---- [key][q]: lane = query r -> dQ
  zero_acc(s);
  zero_acc(dp);
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    s = mfma32(kf[ks], qf[ks], s);
    dp = mfma32(vf[ks], dof[ks], dp);
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int key = acc_row(i, half), key_c = key < a.n ? key : a.n - 1;
    float v = s[i] * a.scale;
    if (a.bias) v += a.bias[((long)head * a.n + rc) * a.n + key_c];
    if (a.mask) v += a.mask[(long)seq * a.n + key_c];
    const float p = (valid && key < a.n) ? __expf(v - lse) : 0.f;
    s[i] = p * (dp[i] - dsum);                      // dS^T[key][q]
  }
  {
    const bf16x8 d0 = acc_frag(s, 0), d1 = acc_frag(s, 1);
    zero_acc(out[0]);
    out[0] = mfma32(tr_frag<32>(kimg, 0, 0, 0, lane), d0, out[0]);
    out[0] = mfma32(tr_frag<32>(kimg, 0, 1, 0, lane), d1, out[0]);
    if (valid) store_rows<32>(a.dq + ((long)seq * a.n + r) * a.lddq + hoff, out, a.scale, lane);
  }
  // ---- [q][key]: lane = key r -> dV, dK
  zero_acc(s);
  zero_acc(dp);
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    s = mfma32(qf[ks], kf[ks], s);
    dp = mfma32(dof[ks], vf[ks], dp);
  }
  const float mval = (a.mask && valid) ? a.mask[(long)seq * a.n + r] : 0.f;
#pragma unroll
  for (int g4 = 0; g4 < 4; ++g4) {
    const int qb = 8 * g4 + 4 * half;
    const float4 l4 = *(const float4*)(lse_s + qb);
    const float4 d4 = *(const float4*)(del_s + qb);
    const float ls[4] = {l4.x, l4.y, l4.z, l4.w}, de[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float v = s[4 * g4 + i] * a.scale + mval;
      if (a.bias) v += a.bias[((long)head * a.n + (qb + i < a.n ? qb + i : a.n - 1)) * a.n + rc];
      const float p = valid ? __expf(v - ls[i]) : 0.f;
      s[4 * g4 + i] = p;
      dp[4 * g4 + i] = p * (dp[4 * g4 + i] - de[i]);
    }
  }
  const bf16x8 p0 = acc_frag(s, 0), p1 = acc_frag(s, 1), d0 = acc_frag(dp, 0), d1 = acc_frag(dp, 1);
  zero_acc(out[0]);
  out[0] = mfma32(tr_frag<32>(doimg, 0, 0, 0, lane), p0, out[0]);
  out[0] = mfma32(tr_frag<32>(doimg, 0, 1, 0, lane), p1, out[0]);
  if (valid) store_rows<32>(a.dv + ((long)seq * a.n + r) * a.lddv + hoff, out, 1.0f, lane);
  zero_acc(out[0]);
  out[0] = mfma32(tr_frag<32>(qimg, 0, 0, 0, lane), d0, out[0]);
  out[0] = mfma32(tr_frag<32>(qimg, 0, 1, 0, lane), d1, out[0]);
  if (valid) store_rows<32>(a.dk + ((long)seq * a.n + r) * a.lddk + hoff, out, a.scale, lane);
}

int check(const AttnArgs& a, int dhead) {
  if (dhead != 32 && dhead != 64) return (int)hipErrorInvalidValue;
  if (a.n <= 0 || a.nseq <= 0 || a.heads <= 0) return (int)hipErrorInvalidValue;
  if (a.heads > 65535 || a.nseq > 65535 * 32) return (int)hipErrorInvalidValue;
  return 0;
}

// waves (32-row tiles) per workgroup: all of them share the LDS images of one (sequence, head), so more waves per
// workgroup = fewer image loads and more waves per SIMD to hide the bias / LDS latency.  d_head 32 kernels are
// compiled for up to 9 waves (n = 576 -> 2 workgroups of 9), d_head 64 for 4.
inline int waves_for(int n, int dhead, int cap32 = 9) {
  const int tiles = (n + 31) / 32, cap = (dhead == 32) ? cap32 : 4;
  if (tiles <= cap) return tiles;
  for (int w = cap; w >= 4; --w)
    if (tiles % w == 0) return w;
  return cap < 8 ? cap : 8;
}

// heads per workgroup for short sequences (all row tiles of a head fit one group of `nw` waves): the largest divisor of
// `heads` whose waves fit the kernel's launch bound and whose LDS regions stay under the default 64 KiB
inline int heads_per_block(int n_pad, int nw, int heads, int max_waves, size_t lds_per_head) {
  static const bool off = CTCLIP_KNOB("CTCLIP_ATTN_HPB1") != nullptr;
  if (off || n_pad / 32 > nw) return 1;
  for (int d = heads; d >= 1; --d)
    if (heads % d == 0 && d * nw <= max_waves && d * lds_per_head <= 65536) return d;
  return 1;
}

}  // namespace

extern "C" {

int ctclip_attn_fwd_dropout(const void* q, const void* k, const void* v, void* o, float* lse, const float* bias,
                            const float* mask, const uint8_t* keep, float keep_scale, int nseq, int n, int heads, int dhead,
                            long ldq, long ldk, long ldv, long ldo, float scale, void* stream) {
  AttnArgs a{};
  a.drop = keep; a.drop_scale = keep_scale;
  a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.o = (bf16_t*)o; a.lse = lse;
  a.bias = bias; a.mask = mask; a.nseq = nseq; a.n = n; a.n_pad = (n + 31) / 32 * 32; a.heads = heads;
  a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo; a.scale = scale;
  if (int e = check(a, dhead)) return e;
  {
    const int ew = ctclip_attn_ws_fwd(a, dhead, (hipStream_t)stream);  // long rows with a shared bias: a wave per sequence
    if (ew >= 0) return ew;
  }
  const int nw = waves_for(n, dhead);
  const size_t lds1h = (size_t)a.n_pad * dhead * 2;
  a.hpb = heads_per_block(a.n_pad, nw, heads, dhead == 32 ? 9 : 4, lds1h);
  a.lds_per_head = (int)lds1h;
  dim3 grid((a.n_pad / 32 + nw - 1) / nw, heads / a.hpb, nseq), block(nw * 64 * a.hpb);
  const size_t lds = lds1h * a.hpb;
  if (lds > 160 * 1024) return (int)hipErrorInvalidValue;
  if (dhead == 32) {
    if (lds > 65536) hipFuncSetAttribute((const void*)attn_fwd_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(attn_fwd_kernel<32>, grid, block, lds, (hipStream_t)stream, a);
  } else {
    if (lds > 65536) hipFuncSetAttribute((const void*)attn_fwd_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(attn_fwd_kernel<64>, grid, block, lds, (hipStream_t)stream, a);
  }
  CTCLIP_CHECK_LAUNCH();
}

int ctclip_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse, const float* bias,
                    const float* mask, int nseq, int n, int heads, int dhead, long ldq, long ldk, long ldv, long ldo,
                    float scale, void* stream) {
  return ctclip_attn_fwd_dropout(q, k, v, o, lse, bias, mask, nullptr, 1.0f, nseq, n, heads, dhead, ldq, ldk, ldv, ldo, scale,
                                 stream);
}

int ctclip_attn_bwd_dropout(const void* q, const void* k, const void* v, const void* o, const void* dO, const float* lse,
                            float* delta, void* dq, void* dk, void* dv, const float* bias, const float* mask,
                            const uint8_t* keep, float keep_scale, float* dbias_dense, const uint16_t* relidx,
                            float* dbias_table, int table_size, int grid_h, int grid_w, int nseq, int n, int heads, int dhead,
                            long ldq, long ldk, long ldv, long ldo, long lddo, long lddq, long lddk, long lddv, float scale,
                            void* stream) {
  AttnArgs a{};
  a.drop = keep; a.drop_scale = keep_scale;
  a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.oin = (const bf16_t*)o;
  a.dO = (const bf16_t*)dO; a.lse = (float*)lse; a.delta = delta; a.dq = (bf16_t*)dq; a.dk = (bf16_t*)dk;
  a.dv = (bf16_t*)dv; a.bias = bias; a.mask = mask; a.dbias_dense = dbias_dense; a.relidx = relidx;
  const bool grid_ok = grid_h > 0 && grid_w > 0 && grid_h * grid_w == n && table_size == (2 * grid_h - 1) * (2 * grid_w - 1);
  if ((grid_h > 0 || grid_w > 0) && !grid_ok) return (int)hipErrorInvalidValue;
  a.dbias_table = ((relidx || grid_ok) && !dbias_dense) ? dbias_table : nullptr; a.table_size = table_size;
  a.grid_h = grid_ok ? grid_h : 0; a.grid_w = grid_ok ? grid_w : 0;
  a.nseq = nseq; a.n = n; a.n_pad = (n + 31) / 32 * 32; a.heads = heads;
  a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo; a.lddo = lddo; a.lddq = lddq; a.lddk = lddk; a.lddv = lddv;
  a.scale = scale;
  if (int e = check(a, dhead)) return e;
  {
    const int ew = ctclip_attn_ws_bwd(a, dhead, (hipStream_t)stream);  // a wave per sequence (attention_ws.hip)
    if (ew >= 0) return ew;
  }
  static const bool no_small = CTCLIP_KNOB("CTCLIP_ATTN_NO_SMALL") != nullptr;
  if (dhead == 32 && n <= 32 && !a.dbias_dense && !a.dbias_table && !a.drop && !no_small) {
    const long items = (long)nseq * heads;
    const int wpb = 8;
    hipLaunchKernelGGL(attn_small_bwd_kernel, dim3((unsigned)((items + wpb - 1) / wpb)), dim3(wpb * 64),
                       (size_t)wpb * SMALL_WAVE_LDS, (hipStream_t)stream, a);
    CTCLIP_CHECK_LAUNCH();
  }
  // both backward passes use 6-wave workgroups: two of them fit per CU at their register budgets (12 resident waves)
  const int nw = waves_for(n, dhead, 6);
  const size_t lds1h = ((size_t)a.n_pad * dhead * 2 +
                        (a.dbias_table ? (size_t)((table_size + 3) & ~3) * 4 + (size_t)a.n_pad * 4 : 0) + 15) & ~(size_t)15;
  const size_t lds2h = (size_t)a.n_pad * dhead * 4 + (size_t)a.n_pad * 8;
  a.hpb = heads_per_block(a.n_pad, nw, heads, dhead == 32 ? 6 : 4, lds1h > lds2h ? lds1h : lds2h);
  dim3 grid((a.n_pad / 32 + nw - 1) / nw, heads / a.hpb, nseq), block(nw * 64 * a.hpb);
  const int nw2 = nw;
  dim3 grid2 = grid, block2 = block;
  const size_t lds1 = lds1h * a.hpb, lds2 = lds2h * a.hpb;
  if (lds1 > 160 * 1024 || lds2 > 160 * 1024) return (int)hipErrorInvalidValue;
  hipStream_t st = (hipStream_t)stream;
  AttnArgs a2 = a;                                  // the two passes carve different LDS regions per head
  a.lds_per_head = (int)lds1h;
  a2.lds_per_head = (int)lds2h;
  if (dhead == 32) {
    if (lds1 > 65536) hipFuncSetAttribute((const void*)attn_bwd_dq_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1);
    if (lds2 > 65536) hipFuncSetAttribute((const void*)attn_bwd_dkv_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
    hipLaunchKernelGGL(attn_bwd_dq_kernel<32>, grid, block, lds1, st, a);
    hipLaunchKernelGGL(attn_bwd_dkv_kernel<32>, grid2, block2, lds2, st, a2);
  } else {
    if (lds1 > 65536) hipFuncSetAttribute((const void*)attn_bwd_dq_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1);
    if (lds2 > 65536) hipFuncSetAttribute((const void*)attn_bwd_dkv_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
    hipLaunchKernelGGL(attn_bwd_dq_kernel<64>, grid, block, lds1, st, a);
    hipLaunchKernelGGL(attn_bwd_dkv_kernel<64>, grid2, block2, lds2, st, a2);
  }
  CTCLIP_CHECK_LAUNCH();
}

int ctclip_attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* dO, const float* lse,
                    float* delta, void* dq, void* dk, void* dv, const float* bias, const float* mask,
                    float* dbias_dense, const uint16_t* relidx, float* dbias_table, int table_size, int grid_h, int grid_w,
                    int nseq, int n, int heads, int dhead, long ldq, long ldk, long ldv, long ldo, long lddo, long lddq,
                    long lddk, long lddv, float scale, void* stream) {
  return ctclip_attn_bwd_dropout(q, k, v, o, dO, lse, delta, dq, dk, dv, bias, mask, nullptr, 1.0f, dbias_dense, relidx,
                                 dbias_table, table_size, grid_h, grid_w, nseq, n, heads, dhead, ldq, ldk, ldv, ldo, lddo,
                                 lddq, lddk, lddv, scale, stream);
}

int ctclip_attn_probs(const void* q, const void* k, const float* lse, const float* bias, const float* mask,
                      float* probs, int nseq, int n, int heads, int dhead, long ldq, long ldk, float scale,
                      void* stream) {
  AttnArgs a{};
  a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.lse = (float*)lse; a.bias = bias; a.mask = mask;
  a.nseq = nseq; a.n = n; a.n_pad = (n + 31) / 32 * 32; a.heads = heads; a.ldq = ldq; a.ldk = ldk; a.scale = scale;
  if (int e = check(a, dhead)) return e;
  const int nw = waves_for(n, dhead);
  a.hpb = 1;
  a.lds_per_head = 0;
  dim3 grid((a.n_pad / 32 + nw - 1) / nw, heads, nseq), block(nw * 64);
  const size_t lds = (size_t)a.n_pad * dhead * 2;
  if (lds > 160 * 1024) return (int)hipErrorInvalidValue;
  if (dhead == 32) {
    if (lds > 65536) hipFuncSetAttribute((const void*)attn_probs_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(attn_probs_kernel<32>, grid, block, lds, (hipStream_t)stream, a, probs);
  } else {
    if (lds > 65536) hipFuncSetAttribute((const void*)attn_probs_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(attn_probs_kernel<64>, grid, block, lds, (hipStream_t)stream, a, probs);
  }
  CTCLIP_CHECK_LAUNCH();
}

}  // extern "C"
