// small_attention.cuh -- the decoder's dense self-attention over a few hundred queries, one launch each way (gfx950).
//
// reference models/deformable_transformer.py:282-287: nn.MultiheadAttention over the nq * T object queries of a sample
// (240 at 60 queries x 4 frames), 8 heads of 48 channels, dropout on the probabilities.  PyTorch's fused attention
// kernels are built for long sequences: at this size one call is 34 us forward and 41 + 36 us backward, plus two fill
// launches, all latency.  Here:
//   forward   workgroup = (batch, head, 16 queries) x 16 lanes per query: K (rows padded to hd + 4 floats: the 16 lanes of
//             a query read 16 consecutive keys conflict-free) and V of the head in LDS; lane c scores keys c, c + 16, ...;
//             row max / sum by lane shuffles over the 16 lanes; probabilities (before dropout) are SAVED ([bs, H, L, L]
//             float32, 3.7 MB at the benchmark size -- cheaper than recomputing them) and parked in LDS; lanes
//             0 .. HD/4 - 1 then own 4 channels each of the output row (16-byte row reads, 4 keys per trip).
//   backward  ONE launch, two kinds of workgroups.  Row kind (16 queries): dS_ij = P_ij (dP_ij - delta_i) with
//             dP = keep / (1 - p) * (dO_i . V_j) and delta_i = dO_i . O_i (which equals sum_j dP_ij P_ij, dropout
//             included), then dQ_i = scale * sum_j dS_ij K_j.  Column kind (16 keys): the same dS_ij and the dropped
//             probabilities for its keys, then dK_j = scale * sum_i dS_ij Q_i and dV_j = sum_i Pd_ij dO_i.
// Dropout: counter-based hash of (seed, element index) as in gemm_bf16.cuh.  q / k are the two halves of the packed
// projection output [bs, L, 2E] and are read (and their gradients written) in place through strides.
// L <= kSaMaxL (one staging of both row sets) or <= kSaMaxL2 (sequential staging, below), head dimension = HD (48 or 32);
// anything else takes the library path in Python.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "gemm_bf16.cuh"

namespace snipper {

constexpr int kSaMaxL = 256, kSaRows = 16, kSaThreads = 256;

struct SmallAttnArgs {
  const float *q; long long q_ld, q_bs;      // element (b, i, h, e) at q[b * q_bs + i * q_ld + h * HD + e]
  const float *k; long long k_ld, k_bs;
  const float *v; long long v_ld, v_bs;
  float *out; long long o_ld, o_bs;          // forward: the attention output; backward: the saved output (read)
  float *P;                                  // [bs][H][L][L] probabilities before dropout (forward writes, backward reads)
  // backward only
  const float *dout; long long do_ld, do_bs;
  float *dq; long long dq_ld, dq_bs;
  float *dk; long long dk_ld, dk_bs;
  float *dv; long long dv_ld, dv_bs;
  int bs, H, L;
  float scale, drop_p;
  uint32_t seed_lo, seed_hi;
};

__device__ __forceinline__ float sa_row16_max(float v) {
  v = fmaxf(v, __shfl_xor(v, 1, 64)); v = fmaxf(v, __shfl_xor(v, 2, 64));
  v = fmaxf(v, __shfl_xor(v, 4, 64)); v = fmaxf(v, __shfl_xor(v, 8, 64));
  return v;
}
__device__ __forceinline__ float sa_row16_sum(float v) {
  v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
  return v;
}
// keep-scale of probability element (bh, i, j): 1 / (1 - p) if kept, 0 if dropped (1 when p == 0)
__device__ __forceinline__ float sa_keep(const SmallAttnArgs &g, uint32_t thresh, float keep_scale, int bh, int i, int j) {
  if (g.drop_p <= 0.f) return 1.f;
  const uint32_t e = ((uint32_t)bh * (uint32_t)g.L + (uint32_t)i) * (uint32_t)g.L + (uint32_t)j;
  return gemm_rand(e, g.seed_lo, g.seed_hi) >= thresh ? keep_scale : 0.f;
}

template <int HD> struct SaLds {
  static constexpr int kPad = HD + 4;
  float a[kSaMaxL * kPad];      // rows read 16-at-a-time by the 16 lanes of a group (padded)
  float b[kSaMaxL * kPad];
  float p[kSaRows][kSaMaxL];    // one row of coefficients per group
  float p2[kSaRows][kSaMaxL];
  float delta[kSaMaxL];
};

// rows [0, L) of a strided [L][HD] matrix -> LDS rows of kPad floats
// (rows [L, round-up-to-4 of L) are zeroed: the combination loops below walk the rows four at a time)
template <int HD, int THREADS = kSaThreads>
__device__ __forceinline__ void sa_stage(float *dst, const float *src, long long ld, int L, int tid) {
  constexpr int kSaThreads = THREADS;
  constexpr int kPad = HD + 4, kV = HD / 4, kIter = (kSaMaxL * kV + kSaThreads - 1) / kSaThreads;
  const int L4 = (L + 3) & ~3;
  float4 t[kIter];
#pragma unroll
  for (int it = 0; it < kIter; ++it) {          // all loads in flight before the first LDS store (one workgroup per CU:
    const int x = tid + it * kSaThreads;        // nothing else would hide their latency)
    const int r = x / kV, c = x - r * kV;
    t[it] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < L) t[it] = *reinterpret_cast<const float4 *>(src + (long long)r * ld + 4 * c);
  }
#pragma unroll
  for (int it = 0; it < kIter; ++it) {
    const int x = tid + it * kSaThreads;
    const int r = x / kV, c = x - r * kV;
    if (r < L4) *reinterpret_cast<float4 *>(dst + r * kPad + 4 * c) = t[it];
  }
}
// acc[0..3] += sum_y coef[y] * rows[y][4c .. 4c+3] over y < L4 (coefficients four at a time, one 16-byte row read each)
template <int HD>
__device__ __forceinline__ void sa_combine(float (&acc)[4], const float *coef, const float *rows, int L4, int c) {
  constexpr int kPad = HD + 4;
  for (int y = 0; y < L4; y += 4) {
    const float4 w = *reinterpret_cast<const float4 *>(coef + y);
    const float4 r0 = *reinterpret_cast<const float4 *>(rows + (y + 0) * kPad + 4 * c);
    const float4 r1 = *reinterpret_cast<const float4 *>(rows + (y + 1) * kPad + 4 * c);
    const float4 r2 = *reinterpret_cast<const float4 *>(rows + (y + 2) * kPad + 4 * c);
    const float4 r3 = *reinterpret_cast<const float4 *>(rows + (y + 3) * kPad + 4 * c);
    acc[0] = fmaf(w.x, r0.x, acc[0]); acc[1] = fmaf(w.x, r0.y, acc[1]); acc[2] = fmaf(w.x, r0.z, acc[2]); acc[3] = fmaf(w.x, r0.w, acc[3]);
    acc[0] = fmaf(w.y, r1.x, acc[0]); acc[1] = fmaf(w.y, r1.y, acc[1]); acc[2] = fmaf(w.y, r1.z, acc[2]); acc[3] = fmaf(w.y, r1.w, acc[3]);
    acc[0] = fmaf(w.z, r2.x, acc[0]); acc[1] = fmaf(w.z, r2.y, acc[1]); acc[2] = fmaf(w.z, r2.z, acc[2]); acc[3] = fmaf(w.z, r2.w, acc[3]);
    acc[0] = fmaf(w.w, r3.x, acc[0]); acc[1] = fmaf(w.w, r3.y, acc[1]); acc[2] = fmaf(w.w, r3.z, acc[2]); acc[3] = fmaf(w.w, r3.w, acc[3]);
  }
}

template <int HD>
__global__ __launch_bounds__(kSaThreads) void small_attn_fwd_kernel(SmallAttnArgs g) {
  __shared__ SaLds<HD> S;
  constexpr int kPad = HD + 4, kLanes = HD / 4;       // output phase: lanes 0 .. HD/4 - 1 of a group own 4 channels each
  const int tid = threadIdx.x, grp = tid >> 4, c = tid & 15;
  const int L4 = (g.L + 3) & ~3;
  const int nblk = (g.L + kSaRows - 1) / kSaRows;
  const int blk = blockIdx.x % nblk, bh = blockIdx.x / nblk, h = bh % g.H, b = bh / g.H;
  const int i = blk * kSaRows + grp;
  const bool row_ok = i < g.L;
  sa_stage<HD>(S.a, g.k + b * g.k_bs + h * HD, g.k_ld, g.L, tid);
  sa_stage<HD>(S.b, g.v + b * g.v_bs + h * HD, g.v_ld, g.L, tid);
  float qv[HD];
  {
    const float *qp = g.q + b * g.q_bs + (long long)(row_ok ? i : 0) * g.q_ld + h * HD;
#pragma unroll
    for (int e = 0; e < HD; e += 4) {
      const float4 t = *reinterpret_cast<const float4 *>(qp + e);
      qv[e] = t.x * g.scale; qv[e + 1] = t.y * g.scale; qv[e + 2] = t.z * g.scale; qv[e + 3] = t.w * g.scale;
    }
  }
  __syncthreads();
  // scores of keys c, c + 16, ... ; running max
  float mx = -INFINITY;
  for (int j = c; j < g.L; j += 16) {
    const float *kr = S.a + j * kPad;
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < HD; e += 4) {
      const float4 t = *reinterpret_cast<const float4 *>(kr + e);
      s = fmaf(qv[e], t.x, s); s = fmaf(qv[e + 1], t.y, s); s = fmaf(qv[e + 2], t.z, s); s = fmaf(qv[e + 3], t.w, s);
    }
    S.p[grp][j] = s;
    mx = fmaxf(mx, s);
  }
  mx = sa_row16_max(mx);
  float sum = 0.f;
  for (int j = c; j < g.L; j += 16) {
    const float e = __expf(S.p[grp][j] - mx);
    S.p[grp][j] = e;
    sum += e;
  }
  sum = sa_row16_sum(sum);
  const float inv = 1.f / sum;
  const float keep_scale = g.drop_p > 0.f ? 1.f / (1.f - g.drop_p) : 1.f;
  const uint32_t thresh = (uint32_t)fminf(g.drop_p * 4294967296.f, 4294967040.f);
  float *Prow = g.P + ((long long)bh * g.L + (row_ok ? i : 0)) * g.L;
  for (int j = c; j < g.L; j += 16) {
    const float p = S.p[grp][j] * inv;
    if (row_ok) Prow[j] = p;
    S.p[grp][j] = p * sa_keep(g, thresh, keep_scale, bh, i, j);
  }
  if (c < L4 - g.L) S.p[grp][g.L + c] = 0.f;
  __syncthreads();         // (the group's row is complete; groups straddle no wave, but a workgroup barrier is cheap here)
  float o[4] = {0.f, 0.f, 0.f, 0.f};
  if (c < kLanes) sa_combine<HD>(o, S.p[grp], S.b, L4, c);
  if (row_ok && c < kLanes)
    *reinterpret_cast<float4 *>(g.out + b * g.o_bs + (long long)i * g.o_ld + h * HD + 4 * c) = make_float4(o[0], o[1], o[2], o[3]);
}

// backward: blocks [0, nblk) of a (b, h) are the row kind, [nblk, 2 nblk) the column kind.
// Round 6: ROWS = 32 queries (or keys) per workgroup of 512 threads.  With 16 rows a launch at L = 240 was 480 workgroups of
// 139 KB of LDS -- one per CU, i.e. TWO residency rounds of ~20 us each on 256 CUs (46 us per launch, profiles/r05 kernel
// table); 32 rows make it 240 workgroups = one round, and the staging of K / V (or dO / Q), which every workgroup repeats, is
// paid half as often.  The column kind's second coefficient row (the dropped probabilities, for dV) no longer has an LDS row
// of its own: the lane keeps its <= 16 values in registers and writes them over the dS row once dK has been combined.
template <int HD, int ROWS> struct SaLdsB {
  static constexpr int kPad = HD + 4;
  float a[kSaMaxL * kPad];
  float b[kSaMaxL * kPad];
  float p[ROWS][kSaMaxL];
  float delta[kSaMaxL];
};

template <int HD, int ROWS>
__global__ __launch_bounds__(ROWS * 16) void small_attn_bwd_kernel(SmallAttnArgs g) {
  __shared__ SaLdsB<HD, ROWS> S;
  constexpr int kPad = HD + 4, kLanes = HD / 4, kThreads = ROWS * 16;
  const int tid = threadIdx.x, grp = tid >> 4, c = tid & 15;
  const int L4 = (g.L + 3) & ~3;
  const int nblk = (g.L + ROWS - 1) / ROWS;
  const int per_bh = 2 * nblk;
  const int bh = blockIdx.x / per_bh, r = blockIdx.x % per_bh, h = bh % g.H, b = bh / g.H;
  const bool col_kind = r >= nblk;
  const int x0 = (col_kind ? r - nblk : r) * ROWS + grp;            // this group's query (row kind) or key (column kind)
  const bool x_ok = x0 < g.L;
  const float keep_scale = g.drop_p > 0.f ? 1.f / (1.f - g.drop_p) : 1.f;
  const uint32_t thresh = (uint32_t)fminf(g.drop_p * 4294967296.f, 4294967040.f);
  const float *dO = g.dout + b * g.do_bs + h * HD;
  const float *O = g.out + b * g.o_bs + h * HD;
  // delta_i = dO_i . O_i for every query
  for (int i = tid; i < g.L; i += kThreads) {
    float d = 0.f;
#pragma unroll
    for (int e = 0; e < HD; e += 4) {
      const float4 a4 = *reinterpret_cast<const float4 *>(dO + (long long)i * g.do_ld + e);
      const float4 b4 = *reinterpret_cast<const float4 *>(O + (long long)i * g.o_ld + e);
      d = fmaf(a4.x, b4.x, d); d = fmaf(a4.y, b4.y, d); d = fmaf(a4.z, b4.z, d); d = fmaf(a4.w, b4.w, d);
    }
    S.delta[i] = d;
  }
  float xv[HD];      // row kind: dO_i; column kind: V_j
  if (!col_kind) {
    sa_stage<HD, kThreads>(S.a, g.v + b * g.v_bs + h * HD, g.v_ld, g.L, tid);      // V rows (scored against dO_i)
    sa_stage<HD, kThreads>(S.b, g.k + b * g.k_bs + h * HD, g.k_ld, g.L, tid);      // K rows (combined into dQ)
    const float *p = dO + (long long)(x_ok ? x0 : 0) * g.do_ld;
#pragma unroll
    for (int e = 0; e < HD; e += 4) {
      const float4 t = *reinterpret_cast<const float4 *>(p + e);
      xv[e] = t.x; xv[e + 1] = t.y; xv[e + 2] = t.z; xv[e + 3] = t.w;
    }
  } else {
    sa_stage<HD, kThreads>(S.a, dO, g.do_ld, g.L, tid);                             // dO rows (scored against V_j; combined into dV)
    sa_stage<HD, kThreads>(S.b, g.q + b * g.q_bs + h * HD, g.q_ld, g.L, tid);      // Q rows (combined into dK)
    const float *p = g.v + b * g.v_bs + (long long)(x_ok ? x0 : 0) * g.v_ld + h * HD;
#pragma unroll
    for (int e = 0; e < HD; e += 4) {
      const float4 t = *reinterpret_cast<const float4 *>(p + e);
      xv[e] = t.x; xv[e + 1] = t.y; xv[e + 2] = t.z; xv[e + 3] = t.w;
    }
  }
  __syncthreads();
  // coefficients of this group's row (of dS) or column (of dS and of the dropped probabilities).  The saved
  // probabilities of the lane's <= 16 elements are fetched up front (a load inside the loop would expose its latency
  // once per element: one workgroup per CU, nothing to switch to)
  const float *Pbh = g.P + (long long)bh * g.L * g.L;
  constexpr int kPer = kSaMaxL / 16;
  float pv[kPer];
#pragma unroll
  for (int t = 0; t < kPer; ++t) {
    const int y = c + 16 * t;
    const int i = col_kind ? y : x0, j = col_kind ? x0 : y;
    pv[t] = (x_ok && y < g.L) ? Pbh[(long long)i * g.L + j] : 0.f;
  }
#pragma unroll
  for (int t = 0; t < kPer; ++t) {
    const int y = c + 16 * t;
    if (y < g.L) {
      const float *yr = S.a + y * kPad;
      float dot = 0.f;
#pragma unroll
      for (int e = 0; e < HD; e += 4) {
        const float4 v4 = *reinterpret_cast<const float4 *>(yr + e);
        dot = fmaf(xv[e], v4.x, dot); dot = fmaf(xv[e + 1], v4.y, dot); dot = fmaf(xv[e + 2], v4.z, dot); dot = fmaf(xv[e + 3], v4.w, dot);
      }
      const int i = col_kind ? y : x0, j = col_kind ? x0 : y;
      const float p = pv[t];
      const float ks = sa_keep(g, thresh, keep_scale, bh, i, j);
      const float ds = p * (ks * dot - S.delta[i]);
      S.p[grp][y] = ds * g.scale;
      pv[t] = p * ks;                               // (column kind: the dropped probability, written over the dS row below)
    }
  }
  if (c < L4 - g.L) S.p[grp][g.L + c] = 0.f;
  __syncthreads();
  if (!col_kind) {
    if (c >= kLanes) return;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    sa_combine<HD>(acc, S.p[grp], S.b, L4, c);                       // dQ_i = scale * sum_j dS_ij K_j
    if (x_ok)
      *reinterpret_cast<float4 *>(g.dq + b * g.dq_bs + (long long)x0 * g.dq_ld + h * HD + 4 * c) =
          make_float4(acc[0], acc[1], acc[2], acc[3]);
    return;
  }
  float ak[4] = {0.f, 0.f, 0.f, 0.f}, av[4] = {0.f, 0.f, 0.f, 0.f};
  if (c < kLanes) sa_combine<HD>(ak, S.p[grp], S.b, L4, c);          // dK_j = scale * sum_i dS_ij Q_i
  __syncthreads();                                                   // (every lane of the group has read the dS row)
#pragma unroll
  for (int t = 0; t < kPer; ++t) {
    const int y = c + 16 * t;
    if (y < g.L) S.p[grp][y] = pv[t];
  }
  __syncthreads();
  if (c >= kLanes) return;
  sa_combine<HD>(av, S.p[grp], S.a, L4, c);                          // dV_j = sum_i Pd_ij dO_i
  if (x_ok) {
    *reinterpret_cast<float4 *>(g.dk + b * g.dk_bs + (long long)x0 * g.dk_ld + h * HD + 4 * c) =
        make_float4(ak[0], ak[1], ak[2], ak[3]);
    *reinterpret_cast<float4 *>(g.dv + b * g.dv_bs + (long long)x0 * g.dv_ld + h * HD + 4 * c) =
        make_float4(av[0], av[1], av[2], av[3]);
  }
}


// ---- 256 < L <= 384 (round 6: the forecast model's 360 object queries, 60 queries x (4 + 2) frames, reference main.py:88-96) --
// K and V of 384 keys do not fit the LDS side by side (2 x 80 KB + the coefficient rows): ONE row buffer is staged twice --
// K for the scores, then V for the combination (forward); V / K, or dO / Q (backward) -- with 32 rows per 512-thread workgroup.
// The column kind of the backward needs dO twice (scores, dV) and Q once (dK): it keeps its dS values in registers while the
// dropped probabilities sit in the coefficient row for dV, then swaps.  Same arithmetic, same dropout stream, same saved
// probabilities as the kernels above; the library's fused attention took 63 + 59 + 56 us per decoder layer at this size.
constexpr int kSaMaxL2 = 384, kSaRows2 = 32, kSaThreads2 = kSaRows2 * 16;

template <int HD> struct SaLds2 {
  static constexpr int kPad = HD + 4;
  float buf[kSaMaxL2 * kPad];
  float p[kSaRows2][kSaMaxL2];
  float delta[kSaMaxL2];
};

template <int HD>
__device__ __forceinline__ void sa_stage2(float *dst, const float *src, long long ld, int L, int tid) {
  constexpr int kPad = HD + 4, kV = HD / 4, kIter = (kSaMaxL2 * kV + kSaThreads2 - 1) / kSaThreads2;
  const int L4 = (L + 3) & ~3;
  float4 t[kIter];
#pragma unroll
  for (int it = 0; it < kIter; ++it) {
    const int x = tid + it * kSaThreads2;
    const int r = x / kV, c = x - r * kV;
    t[it] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < L) t[it] = *reinterpret_cast<const float4 *>(src + (long long)r * ld + 4 * c);
  }
#pragma unroll
  for (int it = 0; it < kIter; ++it) {
    const int x = tid + it * kSaThreads2;
    const int r = x / kV, c = x - r * kV;
    if (r < L4) *reinterpret_cast<float4 *>(dst + r * kPad + 4 * c) = t[it];
  }
}

template <int HD>
__global__ __launch_bounds__(kSaThreads2) void small_attn_fwd_seq_kernel(SmallAttnArgs g) {
  __shared__ SaLds2<HD> S;
  constexpr int kPad = HD + 4, kLanes = HD / 4;
  const int tid = threadIdx.x, grp = tid >> 4, c = tid & 15;
  const int L4 = (g.L + 3) & ~3;
  const int nblk = (g.L + kSaRows2 - 1) / kSaRows2;
  const int blk = blockIdx.x % nblk, bh = blockIdx.x / nblk, h = bh % g.H, b = bh / g.H;
  const int i = blk * kSaRows2 + grp;
  const bool row_ok = i < g.L;
  sa_stage2<HD>(S.buf, g.k + b * g.k_bs + h * HD, g.k_ld, g.L, tid);
  float qv[HD];
  {
    const float *qp = g.q + b * g.q_bs + (long long)(row_ok ? i : 0) * g.q_ld + h * HD;
#pragma unroll
    for (int e = 0; e < HD; e += 4) {
      const float4 t = *reinterpret_cast<const float4 *>(qp + e);
      qv[e] = t.x * g.scale; qv[e + 1] = t.y * g.scale; qv[e + 2] = t.z * g.scale; qv[e + 3] = t.w * g.scale;
    }
  }
  __syncthreads();
  float mx = -INFINITY;
  for (int j = c; j < g.L; j += 16) {
    const float *kr = S.buf + j * kPad;
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < HD; e += 4) {
      const float4 t = *reinterpret_cast<const float4 *>(kr + e);
      s = fmaf(qv[e], t.x, s); s = fmaf(qv[e + 1], t.y, s); s = fmaf(qv[e + 2], t.z, s); s = fmaf(qv[e + 3], t.w, s);
    }
    S.p[grp][j] = s;
    mx = fmaxf(mx, s);
  }
  __syncthreads();                                  // every group has read K: the buffer may take V
  sa_stage2<HD>(S.buf, g.v + b * g.v_bs + h * HD, g.v_ld, g.L, tid);
  mx = sa_row16_max(mx);
  float sum = 0.f;
  for (int j = c; j < g.L; j += 16) {
    const float e = __expf(S.p[grp][j] - mx);
    S.p[grp][j] = e;
    sum += e;
  }
  sum = sa_row16_sum(sum);
  const float inv = 1.f / sum;
  const float keep_scale = g.drop_p > 0.f ? 1.f / (1.f - g.drop_p) : 1.f;
  const uint32_t thresh = (uint32_t)fminf(g.drop_p * 4294967296.f, 4294967040.f);
  float *Prow = g.P + ((long long)bh * g.L + (row_ok ? i : 0)) * g.L;
  for (int j = c; j < g.L; j += 16) {
    const float p = S.p[grp][j] * inv;
    if (row_ok) Prow[j] = p;
    S.p[grp][j] = p * sa_keep(g, thresh, keep_scale, bh, i, j);
  }
  if (c < L4 - g.L) S.p[grp][g.L + c] = 0.f;
  __syncthreads();
  float o[4] = {0.f, 0.f, 0.f, 0.f};
  if (c < kLanes) sa_combine<HD>(o, S.p[grp], S.buf, L4, c);
  if (row_ok && c < kLanes)
    *reinterpret_cast<float4 *>(g.out + b * g.o_bs + (long long)i * g.o_ld + h * HD + 4 * c) = make_float4(o[0], o[1], o[2], o[3]);
}

template <int HD>
__global__ __launch_bounds__(kSaThreads2) void small_attn_bwd_seq_kernel(SmallAttnArgs g) {
  __shared__ SaLds2<HD> S;
  constexpr int kPad = HD + 4, kLanes = HD / 4;
  const int tid = threadIdx.x, grp = tid >> 4, c = tid & 15;
  const int L4 = (g.L + 3) & ~3;
  const int nblk = (g.L + kSaRows2 - 1) / kSaRows2;
  const int per_bh = 2 * nblk;
  const int bh = blockIdx.x / per_bh, r = blockIdx.x % per_bh, h = bh % g.H, b = bh / g.H;
  const bool col_kind = r >= nblk;
  const int x0 = (col_kind ? r - nblk : r) * kSaRows2 + grp;
  const bool x_ok = x0 < g.L;
  const float keep_scale = g.drop_p > 0.f ? 1.f / (1.f - g.drop_p) : 1.f;
  const uint32_t thresh = (uint32_t)fminf(g.drop_p * 4294967296.f, 4294967040.f);
  const float *dO = g.dout + b * g.do_bs + h * HD;
  const float *O = g.out + b * g.o_bs + h * HD;
  for (int i = tid; i < g.L; i += kSaThreads2) {
    float d = 0.f;
#pragma unroll
    for (int e = 0; e < HD; e += 4) {
      const float4 a4 = *reinterpret_cast<const float4 *>(dO + (long long)i * g.do_ld + e);
      const float4 b4 = *reinterpret_cast<const float4 *>(O + (long long)i * g.o_ld + e);
      d = fmaf(a4.x, b4.x, d); d = fmaf(a4.y, b4.y, d); d = fmaf(a4.z, b4.z, d); d = fmaf(a4.w, b4.w, d);
    }
    S.delta[i] = d;
  }
  // the rows scored against this group's vector: V (row kind: against dO_i) or dO (column kind: against V_j)
  sa_stage2<HD>(S.buf, col_kind ? dO : g.v + b * g.v_bs + h * HD, col_kind ? g.do_ld : g.v_ld, g.L, tid);
  float xv[HD];
  {
    const float *p = col_kind ? g.v + b * g.v_bs + (long long)(x_ok ? x0 : 0) * g.v_ld + h * HD : dO + (long long)(x_ok ? x0 : 0) * g.do_ld;
#pragma unroll
    for (int e = 0; e < HD; e += 4) {
      const float4 t = *reinterpret_cast<const float4 *>(p + e);
      xv[e] = t.x; xv[e + 1] = t.y; xv[e + 2] = t.z; xv[e + 3] = t.w;
    }
  }
  __syncthreads();
  const float *Pbh = g.P + (long long)bh * g.L * g.L;
  constexpr int kPer = kSaMaxL2 / 16;
  float pv[kPer], dsv[kPer];
#pragma unroll
  for (int t = 0; t < kPer; ++t) {
    const int y = c + 16 * t;
    const int i = col_kind ? y : x0, j = col_kind ? x0 : y;
    pv[t] = (x_ok && y < g.L) ? Pbh[(long long)i * g.L + j] : 0.f;
    dsv[t] = 0.f;
  }
#pragma unroll
  for (int t = 0; t < kPer; ++t) {
    const int y = c + 16 * t;
    if (y < g.L) {
      const float *yr = S.buf + y * kPad;
      float dot = 0.f;
#pragma unroll
      for (int e = 0; e < HD; e += 4) {
        const float4 v4 = *reinterpret_cast<const float4 *>(yr + e);
        dot = fmaf(xv[e], v4.x, dot); dot = fmaf(xv[e + 1], v4.y, dot); dot = fmaf(xv[e + 2], v4.z, dot); dot = fmaf(xv[e + 3], v4.w, dot);
      }
      const int i = col_kind ? y : x0, j = col_kind ? x0 : y;
      const float p = pv[t];
      const float ks = sa_keep(g, thresh, keep_scale, bh, i, j);
      dsv[t] = p * (ks * dot - S.delta[i]) * g.scale;
      pv[t] = p * ks;
      S.p[grp][y] = col_kind ? pv[t] : dsv[t];      // column kind: the dropped probabilities first (dV combines them with dO)
    }
  }
  if (c < L4 - g.L) S.p[grp][g.L + c] = 0.f;
  __syncthreads();
  if (!col_kind) {
    // row kind: the buffer takes K for dQ_i = scale * sum_j dS_ij K_j (every group has finished reading V)
    sa_stage2<HD>(S.buf, g.k + b * g.k_bs + h * HD, g.k_ld, g.L, tid);
    __syncthreads();
    if (c >= kLanes) return;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    sa_combine<HD>(acc, S.p[grp], S.buf, L4, c);
    if (x_ok)
      *reinterpret_cast<float4 *>(g.dq + b * g.dq_bs + (long long)x0 * g.dq_ld + h * HD + 4 * c) =
          make_float4(acc[0], acc[1], acc[2], acc[3]);
    return;
  }
  float ak[4] = {0.f, 0.f, 0.f, 0.f}, av[4] = {0.f, 0.f, 0.f, 0.f};
  if (c < kLanes) sa_combine<HD>(av, S.p[grp], S.buf, L4, c);            // dV_j = sum_i Pd_ij dO_i
  __syncthreads();                                                        // dO and the Pd rows have been read by every lane
  sa_stage2<HD>(S.buf, g.q + b * g.q_bs + h * HD, g.q_ld, g.L, tid);     // Q rows for dK
#pragma unroll
  for (int t = 0; t < kPer; ++t) {
    const int y = c + 16 * t;
    if (y < g.L) S.p[grp][y] = dsv[t];
  }
  __syncthreads();
  if (c >= kLanes) return;
  sa_combine<HD>(ak, S.p[grp], S.buf, L4, c);                            // dK_j = scale * sum_i dS_ij Q_i
  if (x_ok) {
    *reinterpret_cast<float4 *>(g.dk + b * g.dk_bs + (long long)x0 * g.dk_ld + h * HD + 4 * c) =
        make_float4(ak[0], ak[1], ak[2], ak[3]);
    *reinterpret_cast<float4 *>(g.dv + b * g.dv_bs + (long long)x0 * g.dv_ld + h * HD + 4 * c) =
        make_float4(av[0], av[1], av[2], av[3]);
  }
}

}  // namespace snipper
