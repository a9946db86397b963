// small_ln.cuh -- the decoder's residual + dropout + LayerNorm chain for a few hundred float32 rows (gfx950), and the two
// other element-wise pieces of a decoder layer that used to be ATen launches.
//
// reference: models/deformable_transformer.py:266-300 (``tgt = tgt + self.dropoutN(tgt2); tgt = self.normN(tgt)`` three times
// per decoder layer), :252-254 (``with_pos_embed``: tgt + query_pos in front of the self-attention's q / k projection and of
// the cross attention's offset / weight projections), :329-333 (iterative reference-point refinement).
//
// The decoder sees 480 rows (60 queries x 4 frames x 2 samples; 720 with two forecast frames).  At that size every kernel is
// one residency wave of a fraction of the chip and costs its launch + a dependent boundary (MI355X_MICROARCH.md), so what
// matters is the NUMBER of launches in the chain, not bytes:
//   * forward: ONE launch writes y = LayerNorm(x + dropout(z)) and, with `pos`, yq = y + pos (float32) -- the two
//     ``with_pos_embed`` adds of a layer disappear (csrc/ln_fused.cuh offers the bf16 form of the same output for the
//     encoder);
//   * backward: ONE launch.  Its gradient is the sum of up to kSmallLnMaxSrc float32 sources (a LayerNorm output feeds the
//     next residual, a projection, the position-added copy, the prediction heads: autograd would add them pairwise, one
//     launch each), and the parameter gradients dgamma / dbeta come from the SAME launch: the workgroups after the row
//     workgroups own 16 columns each and sum over all rows -- h = (s - mean) * rstd is available from the saved statistics, so
//     the column sums need no result of the row pass, no second kernel and no cross-workgroup fence.  Deterministic.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "ln_fused.cuh"

namespace snipper {

constexpr int kSmallLnMaxSrc = 4;
constexpr int kSmallLnMaxRows = 16384;     // above this the encoder-size pair (csrc/ln_fused.cuh) is the right tool
constexpr int kSmallLnColsPerWg = 4;       // columns per column-workgroup: one float4 chunk x 256 row slices

struct SmallLnFwdArgs {
  const float *x, *z, *pos;          // [rows][C]; z, pos may be nullptr
  const float *gamma, *beta;         // [C]
  float *s_save, *mean, *rstd;       // [rows][C], [rows], [rows] (nullptr: no backward will run)
  uint8_t *keep;                     // [rows][C/4] or nullptr
  float *y, *yq;                     // [rows][C]; yq = y + pos or nullptr
  int rows, C;
  float p, eps;
  uint32_t seed_lo, seed_hi;
};

// one wave per row, 4-element chunks per lane (the layout and the dropout stream of ln_fused_fwd_kernel: same hash of
// (seed, element index), so a test can compare the two kernels bit for bit)
__global__ __launch_bounds__(kLnThreads) void small_ln_fwd_kernel(SmallLnFwdArgs a) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (kLnThreads / 64) + (threadIdx.x >> 6);
  if (row >= a.rows) return;
  const long long base = (long long)row * a.C;
  const float keep_scale = a.p > 0.f ? 1.f / (1.f - a.p) : 1.f;
  const uint32_t thresh16 = (uint32_t)fminf(a.p * 65536.f + 0.5f, 65535.f);
  float4 v[kLnMaxIter];
  float sum = 0.f;
#pragma unroll
  for (int it = 0; it < kLnMaxIter; ++it) {
    const int c = (lane + 64 * it) * 4;
    v[it] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < a.C) {
      float4 x = *reinterpret_cast<const float4 *>(a.x + base + c);
      if (a.z) {
        float4 z = *reinterpret_cast<const float4 *>(a.z + base + c);
        if (a.p > 0.f) {
          uint32_t h1 = ln_rand((uint32_t)(base + c), a.seed_lo, a.seed_hi) | 1u, h2 = h1;
          h2 ^= h2 << 13; h2 ^= h2 >> 17; h2 ^= h2 << 5;
          const bool k0 = (h1 & 0xffffu) >= thresh16, k1 = (h1 >> 16) >= thresh16;
          const bool k2 = (h2 & 0xffffu) >= thresh16, k3 = (h2 >> 16) >= thresh16;
          z.x = k0 ? z.x * keep_scale : 0.f; z.y = k1 ? z.y * keep_scale : 0.f;
          z.z = k2 ? z.z * keep_scale : 0.f; z.w = k3 ? z.w * keep_scale : 0.f;
          if (a.keep) a.keep[(base + c) >> 2] = (uint8_t)(k0 | (k1 << 1) | (k2 << 2) | (k3 << 3));
        }
        x.x += z.x; x.y += z.y; x.z += z.z; x.w += z.w;
      }
      v[it] = x;
      sum += (x.x + x.y) + (x.z + x.w);
      if (a.s_save) *reinterpret_cast<float4 *>(a.s_save + base + c) = x;
    }
  }
  const float mean = ln_wave_sum(sum) / (float)a.C;
  float sq = 0.f;
#pragma unroll
  for (int it = 0; it < kLnMaxIter; ++it) {
    const int c = (lane + 64 * it) * 4;
    if (c < a.C) {
      const float dx = v[it].x - mean, dy = v[it].y - mean, dz = v[it].z - mean, dw = v[it].w - mean;
      sq += (dx * dx + dy * dy) + (dz * dz + dw * dw);
    }
  }
  const float rstd = rsqrtf(ln_wave_sum(sq) / (float)a.C + a.eps);
  if (lane == 0 && a.mean) { a.mean[row] = mean; a.rstd[row] = rstd; }
#pragma unroll
  for (int it = 0; it < kLnMaxIter; ++it) {
    const int c = (lane + 64 * it) * 4;
    if (c < a.C) {
      const float4 gm = *reinterpret_cast<const float4 *>(a.gamma + c), bt = *reinterpret_cast<const float4 *>(a.beta + c);
      float4 y;
      y.x = (v[it].x - mean) * rstd * gm.x + bt.x; y.y = (v[it].y - mean) * rstd * gm.y + bt.y;
      y.z = (v[it].z - mean) * rstd * gm.z + bt.z; y.w = (v[it].w - mean) * rstd * gm.w + bt.w;
      *reinterpret_cast<float4 *>(a.y + base + c) = y;
      if (a.yq) {
        const float4 ps = *reinterpret_cast<const float4 *>(a.pos + base + c);
        *reinterpret_cast<float4 *>(a.yq + base + c) = make_float4(y.x + ps.x, y.y + ps.y, y.z + ps.z, y.w + ps.w);
      }
    }
  }
}

struct SmallLnBwdArgs {
  const float *g[kSmallLnMaxSrc];    // gradient sources of y (and of yq: d(y + pos)/dy = 1), nullptr = absent
  const float *s_save, *mean, *rstd, *gamma;
  const uint8_t *keep;               // or nullptr
  float *dx, *dz;                    // [rows][C], each may be nullptr
  float *dgamma, *dbeta;             // [C]
  int rows, C, row_blocks;
  float p;
};

__device__ __forceinline__ float4 small_ln_gsum(const SmallLnBwdArgs &a, long long e) {
  float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int k = 0; k < kSmallLnMaxSrc; ++k)
    if (a.g[k]) {                              // (uniform: a kernel argument)
      const float4 t = *reinterpret_cast<const float4 *>(a.g[k] + e);
      g.x += t.x; g.y += t.y; g.z += t.z; g.w += t.w;
    }
  return g;
}

// blocks [0, row_blocks): one wave per row -> dx, dz.  blocks [row_blocks, ...): kSmallLnColsPerWg columns each -> dgamma, dbeta
// (256 row slices; the slices meet by wave shuffles and LDS in a fixed order).
__global__ __launch_bounds__(kLnThreads) void small_ln_bwd_kernel(SmallLnBwdArgs a) {
  if ((int)blockIdx.x < a.row_blocks) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (kLnThreads / 64) + (threadIdx.x >> 6);
    if (row >= a.rows) return;
    const long long base = (long long)row * a.C;
    const float mean = a.mean[row], rstd = a.rstd[row];
    const float keep_scale = a.p > 0.f ? 1.f / (1.f - a.p) : 1.f;
    const float inv_c = 1.f / (float)a.C;
    float4 gg[kLnMaxIter], xh[kLnMaxIter];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int it = 0; it < kLnMaxIter; ++it) {
      const int c = (lane + 64 * it) * 4;
      gg[it] = xh[it] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (c < a.C) {
        float4 g = small_ln_gsum(a, base + c);
        const float4 s = *reinterpret_cast<const float4 *>(a.s_save + base + c);
        const float4 gm = *reinterpret_cast<const float4 *>(a.gamma + c);
        const float4 h = make_float4((s.x - mean) * rstd, (s.y - mean) * rstd, (s.z - mean) * rstd, (s.w - mean) * rstd);
        g.x *= gm.x; g.y *= gm.y; g.z *= gm.z; g.w *= gm.w;
        s1 += (g.x + g.y) + (g.z + g.w);
        s2 += (g.x * h.x + g.y * h.y) + (g.z * h.z + g.w * h.w);
        gg[it] = g; xh[it] = h;
      }
    }
    const float m1 = ln_wave_sum(s1) * inv_c, m2 = ln_wave_sum(s2) * inv_c;
#pragma unroll
    for (int it = 0; it < kLnMaxIter; ++it) {
      const int c = (lane + 64 * it) * 4;
      if (c < a.C) {
        float4 d;
        d.x = rstd * (gg[it].x - m1 - xh[it].x * m2); d.y = rstd * (gg[it].y - m1 - xh[it].y * m2);
        d.z = rstd * (gg[it].z - m1 - xh[it].z * m2); d.w = rstd * (gg[it].w - m1 - xh[it].w * m2);
        if (a.dx) *reinterpret_cast<float4 *>(a.dx + base + c) = d;
        if (a.dz) {
          if (a.keep) {
            const uint32_t k = a.keep[(base + c) >> 2];
            d.x = (k & 1) ? d.x * keep_scale : 0.f; d.y = (k & 2) ? d.y * keep_scale : 0.f;
            d.z = (k & 4) ? d.z * keep_scale : 0.f; d.w = (k & 8) ? d.w * keep_scale : 0.f;
          }
          *reinterpret_cast<float4 *>(a.dz + base + c) = d;
        }
      }
    }
    return;
  }
  // ---- column workgroups: 4 columns each, thread = row slice (256 slices): at 480 rows a thread sums TWO rows -- the first
  // version (16 columns x 64 slices) walked 8 rows per thread, eight dependent global round trips in a latency-bound launch:
  // 12 us per launch against 6 for the row part alone.  The slices meet by wave shuffles, then the four waves through LDS, in a
  // fixed order.
  __shared__ float red[kLnThreads / 64][8];
  const int slice = threadIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c0 = ((int)blockIdx.x - a.row_blocks) * kSmallLnColsPerWg;
  float4 dg = make_float4(0.f, 0.f, 0.f, 0.f), db = dg;
  if (c0 < a.C)
#pragma unroll 2
    for (int row = slice; row < a.rows; row += kLnThreads) {
      const long long e = (long long)row * a.C + c0;
      const float4 g = small_ln_gsum(a, e);
      const float4 s = *reinterpret_cast<const float4 *>(a.s_save + e);
      const float mean = a.mean[row], rstd = a.rstd[row];
      dg.x += g.x * ((s.x - mean) * rstd); dg.y += g.y * ((s.y - mean) * rstd);
      dg.z += g.z * ((s.z - mean) * rstd); dg.w += g.w * ((s.w - mean) * rstd);
      db.x += g.x; db.y += g.y; db.z += g.z; db.w += g.w;
    }
  float v8[8] = {dg.x, dg.y, dg.z, dg.w, db.x, db.y, db.z, db.w};
#pragma unroll
  for (int k = 0; k < 8; ++k) v8[k] = ln_wave_sum(v8[k]);
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 8; ++k) red[wave][k] = v8[k];
  }
  __syncthreads();
  if (threadIdx.x < 8 && c0 + (int)(threadIdx.x & 3) < a.C) {
    float sum = 0.f;
#pragma unroll
    for (int w = 0; w < kLnThreads / 64; ++w) sum += red[w][threadIdx.x];
    (threadIdx.x >= 4 ? a.dbeta : a.dgamma)[c0 + (threadIdx.x & 3)] = sum;
  }
}

// ---- out = sum of up to kSumF32MaxSrc float32 tensors in one pass (the gradients of the aliases fused.FanOut hands out:
// query_pos feeds two additions per decoder layer, reference models/deformable_transformer.py:252-254) ----------------------
constexpr int kSumF32MaxSrc = 16;
struct SumF32Srcs { const float *p[kSumF32MaxSrc]; int n; };
__global__ __launch_bounds__(256) void sum_f32_kernel(SumF32Srcs s, float *__restrict__ out, long long n4) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int k = 0; k < s.n; ++k) {
    const float4 v = reinterpret_cast<const float4 *>(s.p[k])[i];
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  reinterpret_cast<float4 *>(out)[i] = acc;
}

// ---- reference-point refinement with the root head's first two outputs computed in place (reference
// models/deformable_transformer.py:329-333: ``tmp = self.root_embed[lid](output)``, ``new = (tmp[..., :2] +
// inverse_sigmoid(reference)).sigmoid().detach()``): the head is ONE Linear(C -> 4) in the reference's model (models/model.py:95,
// MLP(hidden, hidden, 4, 1)); only its rows 0 and 1 reach the refinement, and the result carries no gradient.  One wave per
// decoder row: two dot products of C elements, then the arithmetic of refine_reference_kernel (csrc/match_cost.cuh). ----------
__global__ __launch_bounds__(256) void refine_reference_linear_kernel(const float *__restrict__ x, const float *__restrict__ W,
                                                                      const float *__restrict__ b, const float *__restrict__ ref,
                                                                      const float *__restrict__ valid_ratios, int rows, int C,
                                                                      int rows_per_batch, int L, float eps,
                                                                      float *__restrict__ new_ref, float *__restrict__ ref_in) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float d0 = 0.f, d1 = 0.f;
  for (int c = lane * 4; c < C; c += 256) {
    const float4 xv = *reinterpret_cast<const float4 *>(x + (long long)row * C + c);
    const float4 w0 = *reinterpret_cast<const float4 *>(W + c), w1 = *reinterpret_cast<const float4 *>(W + C + c);
    d0 += (xv.x * w0.x + xv.y * w0.y) + (xv.z * w0.z + xv.w * w0.w);
    d1 += (xv.x * w1.x + xv.y * w1.y) + (xv.z * w1.z + xv.w * w1.w);
  }
  d0 = ln_wave_sum(d0); d1 = ln_wave_sum(d1);
  if (lane < 2) {
    const int c = lane;
    const float delta = (c ? d1 : d0) + (b ? b[c] : 0.f);
    const int i = row * 2 + c;
    const float xr = fminf(fmaxf(ref[i], 0.f), 1.f);
    const float x1 = fmaxf(xr, eps), x2 = fmaxf(1.f - xr, eps);
    const float z = delta + logf(x1 / x2);
    const float r = 1.f / (1.f + expf(-z));
    new_ref[i] = r;
    const float *vr = valid_ratios + (long long)(row / rows_per_batch) * L * 2 + c;
    for (int l = 0; l < L; ++l) ref_in[((long long)row * L + l) * 2 + c] = r * vr[2 * l];
  }
}

}  // namespace snipper
