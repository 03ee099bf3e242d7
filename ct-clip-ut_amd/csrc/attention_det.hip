// Reproducible d(bias) of the shared-bias attention (reference src/utils/attention.py:155-180: the [heads, n, n] relative-
// position bias of the CT-ViT spatial transformer; its attribution code asks for torch.use_deterministic_algorithms(True),
// src/utils/visualizations.py:29-39).
//
// The training kernels (attention_hm.hip / attention_ws.hip / attention.hip) sum dS = P * (dP - delta) over the sequences
// inside the dQ pass, in whatever order the waves reach a tile (LDS locks / float atomics): fast, and different in the last
// bits from run to run.  This is the ORDERED form, used when deterministic algorithms are requested: a workgroup owns ONE
// (head, 32-query tile, 32-key tile) and walks ALL sequences -- wave w takes sequences w, w + NW, ... in ascending order and
// keeps its partial tile in registers; the NW partials are then added in wave order and the tile is written by its single
// owner.  No atomics, a fixed association: bit-identical from run to run.  It recomputes S and dP for every tile (the work
// of a dQ pass), so it costs about as much as one; the dQ / dK / dV passes then run without a bias gradient.
//
// Operands are addressed by (sequence, head, row) strides, so both the row-major ([nseq * n, ld], head h in columns 32 h ..)
// and the head-major ([nseq][heads][n][32]) layouts are served.  d_head 32 (narrower heads zero-padded to 32, as everywhere);
// any n: rows and keys beyond it are masked out of the last tiles.
#include "attn_common.h"

namespace {

constexpr float kLog2e = 1.4426950408889634f;

struct DetArgs {
  const bf16_t* q; const bf16_t* k; const bf16_t* v; const bf16_t* dO;
  long q_seq, q_head, q_row, k_seq, k_head, k_row, v_seq, v_head, v_row, do_seq, do_head, do_row;   // element strides
  const float* lse; const float* delta;          // [nseq, heads, n]
  const float* bias;                             // [heads, n, n] or null
  float* dbias;                                  // [heads, n, n], +=
  int nseq, n, heads, T;
  float c1;                                      // scale * log2(e): log2-logit = c1 * (q . k) + bias * log2(e)
};

template <int NW>
__global__ __launch_bounds__(NW * 64) void dbias_ordered_kernel(DetArgs a) {
  __shared__ float part[NW][16][64 + 1];
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, half = lane >> 5;
  const int w = tid >> 6;
  int L = blockIdx.x;
  const int kt = L % a.T; L /= a.T;
  const int qt = L % a.T;
  const int head = L / a.T;
  const int q0 = qt * 32, key0 = kt * 32;

  const bool qok = q0 + r < a.n, kok = key0 + r < a.n;             // this lane's query row (S^T columns) / key row (operand rows)
  float bias2[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int key = key0 + acc_row(i, half);
    bias2[i] = (key < a.n && qok) ? (a.bias ? a.bias[((long)head * a.n + q0 + r) * a.n + key] * kLog2e : 0.f) : -INFINITY;
  }
  const short8v zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
  f32x16 sum;
  zero_acc(sum);
  for (int seq = w; seq < a.nseq; seq += NW) {
    const bf16_t* qp = a.q + seq * a.q_seq + head * a.q_head + (long)(q0 + r) * a.q_row + 8 * half;
    const bf16_t* kp = a.k + seq * a.k_seq + head * a.k_head + (long)(key0 + r) * a.k_row + 8 * half;
    const bf16_t* vp = a.v + seq * a.v_seq + head * a.v_head + (long)(key0 + r) * a.v_row + 8 * half;
    const bf16_t* gp = a.dO + seq * a.do_seq + head * a.do_head + (long)(q0 + r) * a.do_row + 8 * half;
    const bf16x8 qf0 = as_bf16x8(qok ? *(const short8v*)qp : zero8), qf1 = as_bf16x8(qok ? *(const short8v*)(qp + 16) : zero8);
    const bf16x8 kf0 = as_bf16x8(kok ? *(const short8v*)kp : zero8), kf1 = as_bf16x8(kok ? *(const short8v*)(kp + 16) : zero8);
    const bf16x8 vf0 = as_bf16x8(kok ? *(const short8v*)vp : zero8), vf1 = as_bf16x8(kok ? *(const short8v*)(vp + 16) : zero8);
    const bf16x8 gf0 = as_bf16x8(qok ? *(const short8v*)gp : zero8), gf1 = as_bf16x8(qok ? *(const short8v*)(gp + 16) : zero8);
    const long stat = ((long)seq * a.heads + head) * a.n + q0 + r;
    const float nlse2 = qok ? -a.lse[stat] * kLog2e : 0.f, dl = qok ? a.delta[stat] : 0.f;
    f32x16 S, dP;
    zero_acc(S);
    zero_acc(dP);
    S = mfma32(kf0, qf0, S);                                       // S^T[key][q]: lane = query, registers = keys
    S = mfma32(kf1, qf1, S);
    dP = mfma32(vf0, gf0, dP);
    dP = mfma32(vf1, gf1, dP);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float p = __builtin_amdgcn_exp2f(fmaf(S[i], a.c1, bias2[i] + nlse2));
      sum[i] += p * (dP[i] - dl);
    }
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) part[w][i][lane] = sum[i];
  __syncthreads();
  for (int e = tid; e < 16 * 64; e += NW * 64) {
    const int i = e >> 6, l = e & 63;
    float t = 0.f;
#pragma unroll
    for (int ww = 0; ww < NW; ++ww) t += part[ww][i][l];           // wave order: a fixed association
    const int qq = q0 + (l & 31), key = key0 + acc_row(i, l >> 5);
    if (qq < a.n && key < a.n) a.dbias[((long)head * a.n + qq) * a.n + key] += t;
  }
}

// table[h][ti] += sum of dense[h][q][key] over the (q, key) pairs at relative position ti of a gh x gw grid, q ascending:
// every table entry has one owner and a fixed order.  ti = (yq - yk + gh - 1) (2 gw - 1) + (xq - xk + gw - 1).
__global__ __launch_bounds__(256) void dbias_table_kernel(const float* __restrict__ dense, float* __restrict__ table, int heads,
                                                          int gh, int gw) {
  const int R = (2 * gh - 1) * (2 * gw - 1), n = gh * gw;
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= heads * R) return;
  const int h = e / R, ti = e % R;
  const int dy = ti / (2 * gw - 1) - (gh - 1), dx = ti % (2 * gw - 1) - (gw - 1);
  float t = 0.f;
  for (int yq = 0; yq < gh; ++yq) {
    const int yk = yq - dy;
    if (yk < 0 || yk >= gh) continue;
    for (int xq = 0; xq < gw; ++xq) {
      const int xk = xq - dx;
      if (xk < 0 || xk >= gw) continue;
      t += dense[((long)h * n + yq * gw + xq) * n + yk * gw + xk];
    }
  }
  table[(long)h * R + ti] += t;
}

}  // namespace

extern "C" {

// dbias[heads, n, n] += sum over the sequences of dS, in a fixed order (see the header of this file).  layout_hm = 1: q / k /
// v / dO are head-major [nseq][heads][n][32] (ld* ignored); 0: row-major [nseq * n, ld*], head h in columns 32 h ...
int ctclip_attn_dbias_ordered(const void* q, const void* k, const void* v, const void* dO, const float* lse,
                              const float* delta, const float* bias, float* dbias, int nseq, int n, int heads, int layout_hm,
                              long ldq, long ldk, long ldv, long lddo, float scale, void* stream) {
  if (nseq <= 0 || n <= 0 || heads <= 0 || !dbias) return (int)hipErrorInvalidValue;
  DetArgs a{};
  a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.dO = (const bf16_t*)dO;
  auto set = [&](long& s_seq, long& s_head, long& s_row, long ld) {
    if (layout_hm) { s_row = 32; s_head = (long)n * 32; s_seq = (long)heads * n * 32; }
    else { s_row = ld; s_head = 32; s_seq = (long)n * ld; }
  };
  set(a.q_seq, a.q_head, a.q_row, ldq);
  set(a.k_seq, a.k_head, a.k_row, ldk);
  set(a.v_seq, a.v_head, a.v_row, ldv);
  set(a.do_seq, a.do_head, a.do_row, lddo);
  a.lse = lse; a.delta = delta; a.bias = bias; a.dbias = dbias; a.nseq = nseq; a.n = n; a.heads = heads; a.T = (n + 31) / 32;
  a.c1 = scale * kLog2e;
  constexpr int NW = 8;
  hipLaunchKernelGGL(dbias_ordered_kernel<NW>, dim3((unsigned)(heads * a.T * a.T)), dim3(NW * 64), 0, (hipStream_t)stream, a);
  CTCLIP_CHECK_LAUNCH();
}

// table[heads][(2 gh - 1)(2 gw - 1)] += the dense gradient gathered by 2-D relative position (n = gh * gw), one owner per entry
int ctclip_attn_dbias_table(const float* dbias_dense, float* dbias_table, int heads, int grid_h, int grid_w, void* stream) {
  if (heads <= 0 || grid_h <= 0 || grid_w <= 0) return (int)hipErrorInvalidValue;
  const int R = (2 * grid_h - 1) * (2 * grid_w - 1);
  hipLaunchKernelGGL(dbias_table_kernel, dim3((unsigned)((heads * R + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dbias_dense,
                     dbias_table, heads, grid_h, grid_w);
  CTCLIP_CHECK_LAUNCH();
}

}  // extern "C"
