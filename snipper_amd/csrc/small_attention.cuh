// small_attention.cuh -- the decoder's dense self-attention over a few hundred queries, one launch each way (gfx950).
//
// reference models/deformable_transformer.py:282-287: nn.MultiheadAttention over the nq * T object queries of a sample
// (240 at 60 queries x 4 frames), 8 heads of 48 channels, dropout on the probabilities.  PyTorch's fused attention
// kernels are built for long sequences: at this size one call is 34 us forward and 41 + 36 us backward, plus two fill
// launches, all latency.  Here:
//   forward   workgroup = (batch, head, 16 queries) x 16 lanes per query: K (rows padded to hd + 4 floats: the 16 lanes of
//             a query read 16 consecutive keys conflict-free) and V of the head in LDS; lane c scores keys c, c + 16, ...;
//             row max / sum by DPP over the 16 lanes; probabilities (before dropout) are SAVED ([bs, H, L, L] float32,
//             3.7 MB at the benchmark size -- cheaper than recomputing them) and parked in LDS; lane c then owns
//             channels c, c + 16, c + 32 of the output row.
//   backward  ONE launch, two kinds of workgroups.  Row kind (16 queries): dS_ij = P_ij (dP_ij - delta_i) with
//             dP = keep / (1 - p) * (dO_i . V_j) and delta_i = dO_i . O_i (which equals sum_j dP_ij P_ij, dropout
//             included), then dQ_i = scale * sum_j dS_ij K_j.  Column kind (16 keys): the same dS_ij and the dropped
//             probabilities for its keys, then dK_j = scale * sum_i dS_ij Q_i and dV_j = sum_i Pd_ij dO_i.
// Dropout: counter-based hash of (seed, element index) as in gemm_bf16.cuh.  q / k are the two halves of the packed
// projection output [bs, L, 2E] and are read (and their gradients written) in place through strides.
// L <= kSaMaxL, head dimension = HD (48 or 32); anything else takes the library path in Python.
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
template <int HD>
__device__ __forceinline__ void sa_stage(float *dst, const float *src, long long ld, int L, int tid) {
  constexpr int kPad = HD + 4, kV = HD / 4;
  for (int x = tid; x < L * kV; x += kSaThreads) {
    const int r = x / kV, c = x - r * kV;
    *reinterpret_cast<float4 *>(dst + r * kPad + 4 * c) = *reinterpret_cast<const float4 *>(src + (long long)r * ld + 4 * c);
  }
}

template <int HD>
__global__ __launch_bounds__(kSaThreads) void small_attn_fwd_kernel(SmallAttnArgs g) {
  __shared__ SaLds<HD> S;
  constexpr int kPad = HD + 4, kC = HD / 16;          // channels per lane in the output phase (lane c: c, c + 16, ...)
  const int tid = threadIdx.x, grp = tid >> 4, c = tid & 15;
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
  __syncthreads();         // (the group's row is complete; a workgroup barrier because groups straddle no wave, but cheap)
  float o[kC];
#pragma unroll
  for (int u = 0; u < kC; ++u) o[u] = 0.f;
  for (int j = 0; j < g.L; ++j) {
    const float p = S.p[grp][j];
    const float *vr = S.b + j * kPad;
#pragma unroll
    for (int u = 0; u < kC; ++u) o[u] = fmaf(p, vr[c + 16 * u], o[u]);
  }
  if (row_ok) {
    float *op = g.out + b * g.o_bs + (long long)i * g.o_ld + h * HD;
#pragma unroll
    for (int u = 0; u < kC; ++u) op[c + 16 * u] = o[u];
  }
}

// backward: blocks [0, nblk) of a (b, h) are the row kind, [nblk, 2 nblk) the column kind
template <int HD>
__global__ __launch_bounds__(kSaThreads) void small_attn_bwd_kernel(SmallAttnArgs g) {
  __shared__ SaLds<HD> S;
  constexpr int kPad = HD + 4, kC = HD / 16;
  const int tid = threadIdx.x, grp = tid >> 4, c = tid & 15;
  const int nblk = (g.L + kSaRows - 1) / kSaRows;
  const int per_bh = 2 * nblk;
  const int bh = blockIdx.x / per_bh, r = blockIdx.x % per_bh, h = bh % g.H, b = bh / g.H;
  const bool col_kind = r >= nblk;
  const int x0 = (col_kind ? r - nblk : r) * kSaRows + grp;         // this group's query (row kind) or key (column kind)
  const bool x_ok = x0 < g.L;
  const float keep_scale = g.drop_p > 0.f ? 1.f / (1.f - g.drop_p) : 1.f;
  const uint32_t thresh = (uint32_t)fminf(g.drop_p * 4294967296.f, 4294967040.f);
  const float *dO = g.dout + b * g.do_bs + h * HD;
  const float *O = g.out + b * g.o_bs + h * HD;
  // delta_i = dO_i . O_i for every query
  for (int i = tid; i < g.L; i += kSaThreads) {
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
    sa_stage<HD>(S.a, g.v + b * g.v_bs + h * HD, g.v_ld, g.L, tid);      // V rows (scored against dO_i)
    sa_stage<HD>(S.b, g.k + b * g.k_bs + h * HD, g.k_ld, g.L, tid);      // K rows (combined into dQ)
    const float *p = dO + (long long)(x_ok ? x0 : 0) * g.do_ld;
#pragma unroll
    for (int e = 0; e < HD; e += 4) {
      const float4 t = *reinterpret_cast<const float4 *>(p + e);
      xv[e] = t.x; xv[e + 1] = t.y; xv[e + 2] = t.z; xv[e + 3] = t.w;
    }
  } else {
    sa_stage<HD>(S.a, dO, g.do_ld, g.L, tid);                             // dO rows (scored against V_j; combined into dV)
    sa_stage<HD>(S.b, g.q + b * g.q_bs + h * HD, g.q_ld, g.L, tid);      // Q rows (combined into dK)
    const float *p = g.v + b * g.v_bs + (long long)(x_ok ? x0 : 0) * g.v_ld + h * HD;
#pragma unroll
    for (int e = 0; e < HD; e += 4) {
      const float4 t = *reinterpret_cast<const float4 *>(p + e);
      xv[e] = t.x; xv[e + 1] = t.y; xv[e + 2] = t.z; xv[e + 3] = t.w;
    }
  }
  __syncthreads();
  // coefficients of this group's row (of dS) or column (of dS and of the dropped probabilities)
  const float *Pbh = g.P + (long long)bh * g.L * g.L;
  for (int y = c; y < g.L; y += 16) {
    const float *yr = S.a + y * kPad;
    float dot = 0.f;
#pragma unroll
    for (int e = 0; e < HD; e += 4) {
      const float4 t = *reinterpret_cast<const float4 *>(yr + e);
      dot = fmaf(xv[e], t.x, dot); dot = fmaf(xv[e + 1], t.y, dot); dot = fmaf(xv[e + 2], t.z, dot); dot = fmaf(xv[e + 3], t.w, dot);
    }
    const int i = col_kind ? y : x0, j = col_kind ? x0 : y;
    const float p = x_ok ? Pbh[(long long)i * g.L + j] : 0.f;
    const float ks = sa_keep(g, thresh, keep_scale, bh, i, j);
    const float ds = p * (ks * dot - S.delta[i]);
    S.p[grp][y] = ds * g.scale;
    if (col_kind) S.p2[grp][y] = p * ks;
  }
  __syncthreads();
  if (!col_kind) {
    float acc[kC];
#pragma unroll
    for (int u = 0; u < kC; ++u) acc[u] = 0.f;
    for (int j = 0; j < g.L; ++j) {
      const float w = S.p[grp][j];
      const float *kr = S.b + j * kPad;
#pragma unroll
      for (int u = 0; u < kC; ++u) acc[u] = fmaf(w, kr[c + 16 * u], acc[u]);
    }
    if (x_ok) {
      float *o = g.dq + b * g.dq_bs + (long long)x0 * g.dq_ld + h * HD;
#pragma unroll
      for (int u = 0; u < kC; ++u) o[c + 16 * u] = acc[u];
    }
  } else {
    float ak[kC], av[kC];
#pragma unroll
    for (int u = 0; u < kC; ++u) { ak[u] = 0.f; av[u] = 0.f; }
    for (int i = 0; i < g.L; ++i) {
      const float w = S.p[grp][i], pd = S.p2[grp][i];
      const float *qr = S.b + i * kPad, *gr = S.a + i * kPad;
#pragma unroll
      for (int u = 0; u < kC; ++u) {
        ak[u] = fmaf(w, qr[c + 16 * u], ak[u]);
        av[u] = fmaf(pd, gr[c + 16 * u], av[u]);
      }
    }
    if (x_ok) {
      float *ok_ = g.dk + b * g.dk_bs + (long long)x0 * g.dk_ld + h * HD;
      float *ov = g.dv + b * g.dv_bs + (long long)x0 * g.dv_ld + h * HD;
#pragma unroll
      for (int u = 0; u < kC; ++u) { ok_[c + 16 * u] = ak[u]; ov[c + 16 * u] = av[u]; }
    }
  }
}

}  // namespace snipper
