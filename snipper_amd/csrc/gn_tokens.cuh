// gn_tokens.cuh -- GroupNorm of the input projections, computed on token rows (gfx950).
//
// The model projects every backbone level with Conv2d(1x1) + GroupNorm(32, C) (reference models/model.py:62-84) and
// the transformer then flattens each [b, c, t, h, w] map into token rows [b, t, h*w, c] and concatenates the levels
// (models/deformable_transformer.py:103-122).  With NHWC activations the 1x1 convolution already IS a GEMM whose
// output rows are those tokens; what is left is a GroupNorm whose statistics run over (h*w, C/G channels) of one
// image.  These kernels do that on the [n, hw, C] bf16 rows and write the result straight into the level's slice of
// the concatenated [b, t, S, C] buffers -- float32 (residual stream), bf16 (input of the value projection) and
// bf16(y + pos) (query of the first encoder layer) -- so no layout copy, concatenation, cast or add kernel follows.
//
//   forward : gn_stats_kernel  -> per-workgroup partial (sum, sum of squares) per (image, group)   [deterministic]
//             gn_apply_kernel  -> finishes the statistics, normalises, writes the three views + mean / rstd
//   backward: gn_bwd_stats_kernel -> partial (sum g*gamma, sum g*gamma*xhat) per (image, group) and partial
//                                    dgamma / dbeta per channel (summed by ln_param_grad_kernel)
//             gn_bwd_apply_kernel -> dx = rstd * (g*gamma - (s1 + xhat*s2) / count), bf16
//
// Thread mapping (all four): a thread owns one 4-channel chunk (4 | C/G, so a chunk never straddles groups) and
// every RPP-th row of its workgroup's row range; blockDim = (C/4) * RPP <= 256.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "ln_fused.cuh"

namespace snipper {

constexpr int kGnMaxGroups = 64;

struct GnArgs {
  const uint16_t *x;          // [n][hw][C] bf16 (the projection's output)
  const float *gamma, *beta;  // [C]
  float *part;                // [n][nblk][G][2] partial sums (forward) / partial (s1, s2) (backward)
  float *stats;               // [n][G][2] mean, rstd
  // forward outputs: rows of image i start at row (i * dst_rows_per_image + dst_row_offset) of [*, C] buffers
  float *y32; uint16_t *y16; uint16_t *yq16;
  const void *pos; int pos_dt;            // same row addressing as the outputs; only for yq16
  // backward inputs (same row addressing) and outputs
  const float *g32; const uint16_t *g16; const uint16_t *gq16;
  uint16_t *dx;               // [n][hw][C] bf16
  float *part_param;          // [n * nblk][2][C] partial dgamma / dbeta
  long long dst_rows_per_image, dst_row_offset;
  int n, hw, C, G, nblk, rpp;
  float eps;
};

__device__ __forceinline__ void gn_block_group_reduce(float a, float b, int G, int chunks_per_group, int chunks, int rpp,
                                                      float *lds /* [256][2] */, float *out /* [G][2] */) {
  lds[2 * threadIdx.x] = a;
  lds[2 * threadIdx.x + 1] = b;
  __syncthreads();
  if ((int)threadIdx.x < G) {
    float sa = 0.f, sb = 0.f;
    for (int r = 0; r < rpp; ++r)
      for (int c = 0; c < chunks_per_group; ++c) {
        const int t = r * chunks + threadIdx.x * chunks_per_group + c;
        sa += lds[2 * t];
        sb += lds[2 * t + 1];
      }
    out[2 * threadIdx.x] = sa;
    out[2 * threadIdx.x + 1] = sb;
  }
}

__global__ __launch_bounds__(256) void gn_stats_kernel(GnArgs a) {
  __shared__ float lds[512];
  const int chunks = a.C / 4, chunk = threadIdx.x % chunks, rsub = threadIdx.x / chunks;
  const int img = blockIdx.y;
  const uint16_t *x = a.x + (long long)img * a.hw * a.C + chunk * 4;
  float s = 0.f, q = 0.f;
  for (int r = blockIdx.x * a.rpp + rsub; r < a.hw; r += a.nblk * a.rpp) {
    const float4 v = ln_load4(x, 1, (long long)r * a.C);
    s += (v.x + v.y) + (v.z + v.w);
    q += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
  }
  gn_block_group_reduce(s, q, a.G, (a.C / a.G) / 4, chunks, a.rpp, lds,
                        a.part + ((long long)img * a.nblk + blockIdx.x) * a.G * 2);
}

// finishes per-(image, group) sums from the partials: returns (A, B) for this thread's group via LDS
__device__ __forceinline__ void gn_finish(const float *part, int img, int nblk, int G, float *lds_out /* [G][2] */) {
  if ((int)threadIdx.x < G) {
    float sa = 0.f, sb = 0.f;
    const float *p = part + (long long)img * nblk * G * 2 + threadIdx.x * 2;
    for (int b = 0; b < nblk; ++b) { sa += p[(long long)b * G * 2]; sb += p[(long long)b * G * 2 + 1]; }
    lds_out[2 * threadIdx.x] = sa;
    lds_out[2 * threadIdx.x + 1] = sb;
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void gn_apply_kernel(GnArgs a) {
  __shared__ float grp[2 * kGnMaxGroups];
  const int chunks = a.C / 4, chunk = threadIdx.x % chunks, rsub = threadIdx.x / chunks;
  const int img = blockIdx.y, cpg = a.C / a.G, g = (chunk * 4) / cpg;
  gn_finish(a.part, img, a.nblk, a.G, grp);
  const float cnt = (float)a.hw * (float)cpg;
  const float mean = grp[2 * g] / cnt;
  const float var = fmaxf(grp[2 * g + 1] / cnt - mean * mean, 0.f);
  const float rstd = rsqrtf(var + a.eps);
  if (blockIdx.x == 0 && (int)threadIdx.x < a.G) {
    const float m = grp[2 * threadIdx.x] / cnt;
    const float v = fmaxf(grp[2 * threadIdx.x + 1] / cnt - m * m, 0.f);
    a.stats[((long long)img * a.G + threadIdx.x) * 2] = m;
    a.stats[((long long)img * a.G + threadIdx.x) * 2 + 1] = rsqrtf(v + a.eps);
  }
  const float4 gm = *reinterpret_cast<const float4 *>(a.gamma + chunk * 4), bt = *reinterpret_cast<const float4 *>(a.beta + chunk * 4);
  const uint16_t *x = a.x + (long long)img * a.hw * a.C + chunk * 4;
  const long long drow0 = (long long)img * a.dst_rows_per_image + a.dst_row_offset;
  for (int r = blockIdx.x * a.rpp + rsub; r < a.hw; r += a.nblk * a.rpp) {
    const float4 v = ln_load4(x, 1, (long long)r * a.C);
    float4 y;
    y.x = (v.x - mean) * rstd * gm.x + bt.x; y.y = (v.y - mean) * rstd * gm.y + bt.y;
    y.z = (v.z - mean) * rstd * gm.z + bt.z; y.w = (v.w - mean) * rstd * gm.w + bt.w;
    const long long e = (drow0 + r) * a.C + chunk * 4;
    if (a.y32) *reinterpret_cast<float4 *>(a.y32 + e) = y;
    if (a.y16) ln_store4(a.y16, 1, e, y);
    if (a.yq16) {
      const float4 ps = ln_load4(a.pos, a.pos_dt, e);
      ln_store4(a.yq16, 1, e, make_float4(y.x + ps.x, y.y + ps.y, y.z + ps.z, y.w + ps.w));
    }
  }
}

__device__ __forceinline__ float4 gn_load_grad(const GnArgs &a, long long e) {
  float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
  if (a.g32) g = *reinterpret_cast<const float4 *>(a.g32 + e);
  if (a.g16) { const float4 t = ln_load4(a.g16, 1, e); g.x += t.x; g.y += t.y; g.z += t.z; g.w += t.w; }
  if (a.gq16) { const float4 t = ln_load4(a.gq16, 1, e); g.x += t.x; g.y += t.y; g.z += t.z; g.w += t.w; }
  return g;
}

__global__ __launch_bounds__(256) void gn_bwd_stats_kernel(GnArgs a) {
  __shared__ float lds[512];
  __shared__ float pg[256 * 8];
  const int chunks = a.C / 4, chunk = threadIdx.x % chunks, rsub = threadIdx.x / chunks;
  const int img = blockIdx.y, cpg = a.C / a.G, g = (chunk * 4) / cpg;
  const float mean = a.stats[((long long)img * a.G + g) * 2], rstd = a.stats[((long long)img * a.G + g) * 2 + 1];
  const float4 gm = *reinterpret_cast<const float4 *>(a.gamma + chunk * 4);
  const uint16_t *x = a.x + (long long)img * a.hw * a.C + chunk * 4;
  const long long drow0 = (long long)img * a.dst_rows_per_image + a.dst_row_offset;
  float s1 = 0.f, s2 = 0.f;
  float4 dg = make_float4(0.f, 0.f, 0.f, 0.f), db = dg;
  for (int r = blockIdx.x * a.rpp + rsub; r < a.hw; r += a.nblk * a.rpp) {
    const float4 v = ln_load4(x, 1, (long long)r * a.C);
    const float4 gr = gn_load_grad(a, (drow0 + r) * a.C + chunk * 4);
    const float4 h = make_float4((v.x - mean) * rstd, (v.y - mean) * rstd, (v.z - mean) * rstd, (v.w - mean) * rstd);
    dg.x += gr.x * h.x; dg.y += gr.y * h.y; dg.z += gr.z * h.z; dg.w += gr.w * h.w;
    db.x += gr.x; db.y += gr.y; db.z += gr.z; db.w += gr.w;
    const float4 gg = make_float4(gr.x * gm.x, gr.y * gm.y, gr.z * gm.z, gr.w * gm.w);
    s1 += (gg.x + gg.y) + (gg.z + gg.w);
    s2 += (gg.x * h.x + gg.y * h.y) + (gg.z * h.z + gg.w * h.w);
  }
  // per-channel partials: the rpp row-lanes of a chunk meet in LDS, lane 0 of each chunk writes
  float *mine = pg + threadIdx.x * 8;
  mine[0] = dg.x; mine[1] = dg.y; mine[2] = dg.z; mine[3] = dg.w; mine[4] = db.x; mine[5] = db.y; mine[6] = db.z; mine[7] = db.w;
  gn_block_group_reduce(s1, s2, a.G, cpg / 4, chunks, a.rpp, lds, a.part + ((long long)img * a.nblk + blockIdx.x) * a.G * 2);
  // (gn_block_group_reduce has synchronised the workgroup after the LDS writes above)
  if (rsub == 0) {
    float acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = 0.f;
    for (int r = 0; r < a.rpp; ++r)
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] += pg[(r * chunks + chunk) * 8 + k];
    float *pp = a.part_param + ((long long)img * a.nblk + blockIdx.x) * 2 * a.C;
    *reinterpret_cast<float4 *>(pp + chunk * 4) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    *reinterpret_cast<float4 *>(pp + a.C + chunk * 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
  }
}

__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(GnArgs a) {
  __shared__ float grp[2 * kGnMaxGroups];
  const int chunks = a.C / 4, chunk = threadIdx.x % chunks, rsub = threadIdx.x / chunks;
  const int img = blockIdx.y, cpg = a.C / a.G, g = (chunk * 4) / cpg;
  gn_finish(a.part, img, a.nblk, a.G, grp);
  const float inv_cnt = 1.f / ((float)a.hw * (float)cpg);
  const float m1 = grp[2 * g] * inv_cnt, m2 = grp[2 * g + 1] * inv_cnt;
  const float mean = a.stats[((long long)img * a.G + g) * 2], rstd = a.stats[((long long)img * a.G + g) * 2 + 1];
  const float4 gm = *reinterpret_cast<const float4 *>(a.gamma + chunk * 4);
  const uint16_t *x = a.x + (long long)img * a.hw * a.C + chunk * 4;
  uint16_t *dx = a.dx + (long long)img * a.hw * a.C + chunk * 4;
  const long long drow0 = (long long)img * a.dst_rows_per_image + a.dst_row_offset;
  for (int r = blockIdx.x * a.rpp + rsub; r < a.hw; r += a.nblk * a.rpp) {
    const float4 v = ln_load4(x, 1, (long long)r * a.C);
    const float4 gr = gn_load_grad(a, (drow0 + r) * a.C + chunk * 4);
    float4 d;
    d.x = rstd * (gr.x * gm.x - m1 - (v.x - mean) * rstd * m2);
    d.y = rstd * (gr.y * gm.y - m1 - (v.y - mean) * rstd * m2);
    d.z = rstd * (gr.z * gm.z - m1 - (v.z - mean) * rstd * m2);
    d.w = rstd * (gr.w * gm.w - m1 - (v.w - mean) * rstd * m2);
    ln_store4(dx, 1, (long long)r * a.C, d);
  }
}

// ---- column sums of row segments: out[c] = sum over images i and rows r < rows_per_seg of x[i * image_stride + r][c]
// (the gradient of a per-level embedding that was broadcast over a level's tokens: reference
// models/deformable_transformer.py:118 ``lvl_pos_embed = pos_embed + self.level_embed[lvl]``).  x bf16, out f32.
// Stage 1: workgroup (blk, image) sums every nblk-th group of rows into part[image * nblk + blk][C]; stage 2 adds the
// partials in a fixed order.
__global__ __launch_bounds__(256) void colsum_partial_kernel(const uint16_t *x, long long image_stride, int rows_per_seg,
                                                             int C, int nblk, int rpp, float *part) {
  __shared__ float lds[256 * 4];
  const int chunks = C / 4, chunk = threadIdx.x % chunks, rsub = threadIdx.x / chunks;
  const uint16_t *xi = x + (long long)blockIdx.y * image_stride + chunk * 4;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int r = blockIdx.x * rpp + rsub; r < rows_per_seg; r += nblk * rpp) {
    const float4 v = ln_load4(xi, 1, (long long)r * C);
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  *reinterpret_cast<float4 *>(lds + threadIdx.x * 4) = s;
  __syncthreads();
  if (rsub == 0) {
    for (int r = 1; r < rpp; ++r) {
      const float4 t = *reinterpret_cast<const float4 *>(lds + (r * chunks + chunk) * 4);
      s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
    }
    *reinterpret_cast<float4 *>(part + ((long long)blockIdx.y * nblk + blockIdx.x) * C + chunk * 4) = s;
  }
}

// the same over SEVERAL source tensors of one shape (blockIdx.y = source * n_images + image): the column sums of a sum of
// tensors without forming the sum -- level_embed's gradient from the six gradients of its six uses (fused.LevelPosTokens)
struct ColsumSrcs { const uint16_t *p[8]; };
__global__ __launch_bounds__(256) void colsum_partial_multi_kernel(ColsumSrcs srcs, int n_src, long long image_stride,
                                                                   int rows_per_seg, int C, int nblk, int rpp, float *part) {
  // (blockIdx.y = image; the sources are summed HERE, row by row, so the second pass sees as many partial rows as for one
  //  source -- with a partial row per (source, image, block) its 16 slices walked 3 072 rows serially: 82 us)
  __shared__ float lds[256 * 4];
  const int chunks = C / 4, chunk = threadIdx.x % chunks, rsub = threadIdx.x / chunks;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int r = blockIdx.x * rpp + rsub; r < rows_per_seg; r += nblk * rpp) {
    for (int k = 0; k < n_src; ++k) {
      const float4 v = ln_load4(srcs.p[k] + (long long)blockIdx.y * image_stride + chunk * 4, 1, (long long)r * C);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  }
  *reinterpret_cast<float4 *>(lds + threadIdx.x * 4) = s;
  __syncthreads();
  if (rsub == 0) {
    for (int r = 1; r < rpp; ++r) {
      const float4 t = *reinterpret_cast<const float4 *>(lds + (r * chunks + chunk) * 4);
      s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
    }
    *reinterpret_cast<float4 *>(part + ((long long)blockIdx.y * nblk + blockIdx.x) * C + chunk * 4) = s;
  }
}

__global__ __launch_bounds__(256) void colsum_final_kernel(const float *part, int nparts, int C, float *out) {
  __shared__ float red[16][17];
  const int ci = threadIdx.x & 15, slice = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + ci;
  float sum = 0.f;
  if (c < C)
    for (int b = slice; b < nparts; b += 16) sum += part[(long long)b * C + c];
  red[slice][ci] = sum;
  __syncthreads();
  if (slice == 0 && c < C) {
    sum = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) sum += red[k][ci];
    out[c] = sum;
  }
}

// ---- ResNet stem tail: frozen-BN shift + ReLU + 3x3 / stride 2 / pad 1 max-pool in one pass (NHWC bf16) -------------
// reference models/backbone.py:27-64 (FrozenBatchNorm2d after conv1) + torchvision's stem (ReLU, MaxPool2d(3, 2, 1)).
// y = conv1(x, w * scale) [N, H, W, C] without the shift; out[n, oh, ow, c] = relu(max_{3x3 window}(y) + shift[c]) --
// adding a per-channel constant and clamping at 0 are monotone, so they commute with the max.  PyTorch runs this as
// an add (48 us), a clamp (37 us) and the pooling kernel (72 us) over the 123 MB stem output; here the output is read
// once (window overlap through the caches) and only the pooled quarter is written.  One thread = 8 channels of one
// output pixel (16-byte accesses); padding cells take no part (as -inf does in the reference).
__global__ __launch_bounds__(256) void stem_pool_kernel(const uint16_t *__restrict__ y, const float *__restrict__ shift,
                                                        int N, int H, int W, int C, int OH, int OW,
                                                        uint16_t *__restrict__ out) {
  const int c8 = C >> 3;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long total = (long long)N * OH * OW * c8;
  if (idx >= total) return;
  const int cc = (int)(idx % c8);
  const int ow = (int)((idx / c8) % OW);
  const int oh = (int)((idx / ((long long)c8 * OW)) % OH);
  const int n = (int)(idx / ((long long)c8 * OW * OH));
  float m[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) m[i] = -3.0e38f;
  const uint16_t *img = y + (long long)n * H * W * C + cc * 8;
#pragma unroll
  for (int dy = 0; dy < 3; ++dy) {
    const int ih = oh * 2 - 1 + dy;
    if (ih < 0 || ih >= H) continue;
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int iw = ow * 2 - 1 + dx;
      if (iw < 0 || iw >= W) continue;
      const uint4 v = *reinterpret_cast<const uint4 *>(img + ((long long)ih * W + iw) * C);
      const unsigned w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        m[2 * i] = fmaxf(m[2 * i], __uint_as_float(w4[i] << 16));
        m[2 * i + 1] = fmaxf(m[2 * i + 1], __uint_as_float(w4[i] & 0xffff0000u));
      }
    }
  }
  const float4 s0 = *reinterpret_cast<const float4 *>(shift + cc * 8), s1 = *reinterpret_cast<const float4 *>(shift + cc * 8 + 4);
  const float sh[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
  unsigned o[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    // (the convolution output arrives rounded to bf16 WITHOUT the shift, so conv + shift is rounded twice here where a
    //  biased convolution rounds once: within one bf16 ulp of it)
    const float a = fmaxf(m[2 * i] + sh[2 * i], 0.f), b = fmaxf(m[2 * i + 1] + sh[2 * i + 1], 0.f);
    o[i] = (unsigned)__builtin_bit_cast(uint16_t, (__bf16)a) | ((unsigned)__builtin_bit_cast(uint16_t, (__bf16)b) << 16);
  }
  *reinterpret_cast<uint4 *>(out + (((long long)n * OH + oh) * OW + ow) * C + cc * 8) = make_uint4(o[0], o[1], o[2], o[3]);
}

}  // namespace snipper
