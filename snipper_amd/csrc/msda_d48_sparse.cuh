// msda_d48_sparse.cuh -- grad_value of the core op for FEW queries on a large bf16 value (the decoder's cross attention),
// without float atomics, without a float32 accumulation buffer and without a cast pass (gfx950).
//
// reference: models/ops/src/cuda/ms_deform_im2col_cuda.cuh:87-159 (col2im: every tap adds weight * attention * grad_out to
// its pixel's grad_value row with atomicAdd).  The decoder samples 60 queries x 8 heads x 12 points per frame out of 9 875
// positions (models/deformable_transformer.py:290-295): 184 000 taps per launch touch at most 29 % of the 632 000
// (position, head) rows, but the atomic formulation paid for all of them three times -- a 121 MB float32 memset, the atomics,
// and a 121 -> 60 MB cast to the bf16 gradient its consumers (the value projection's weight / data gradient) read:
// 83 us per decoder layer.  Here one workgroup owns one (sample, head, level):
//   1. its <= 1 024 taps (queries x points x 4 corners) become 32-bit keys  pixel << 10 | tap  in LDS (out-of-map taps: ~0);
//   2. a bitonic sort in LDS brings the taps of a pixel together, in tap order;
//   3. the thread that holds the first tap of a pixel adds up its taps' weight * attention * grad_out rows (float32, the
//      grad_out rows of the (sample, head) staged in LDS) and stores the pixel's 48 channels ONCE, as bf16.
// grad_value is zeroed by the launcher as bf16 (60 MB); untouched rows stay zero.  Deterministic (sorted order), one rounding
// per element.  grad_loc / grad_attn still come from msda_bwd_d48_f32_kernel (launched without its atomics).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "msda_d48.cuh"

namespace snipper {

constexpr int kSpTaps = 1024, kSpMaxLq = 64, kSpD = 48;

__global__ __launch_bounds__(256) void msda_bwd_d48_sparse_gv_kernel(
    const uint16_t *__restrict__ grad_out,      // [N][Lq][M][48] bf16
    const int64_t *__restrict__ shapes, const int64_t *__restrict__ level_start,
    const float *__restrict__ loc, const float *__restrict__ attn, CoreDims d,
    uint16_t *__restrict__ grad_value) {        // [N][S][M][48] bf16, zeroed
  __shared__ __attribute__((aligned(16))) float g[kSpMaxLq * kSpD];
  __shared__ unsigned keys[kSpTaps];
  __shared__ float wa[kSpTaps];
  const int tid = threadIdx.x;
  const int l = blockIdx.x % d.L;
  const int nm = blockIdx.x / d.L;
  const int m = nm % d.M, n = nm / d.M;
  const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1], start = (int)level_start[l];
  const int ntap = d.Lq * d.P * 4;

  // ---- grad_out rows of this (sample, head)
  for (int e = tid; e < d.Lq * kSpD; e += 256) {
    const int q = e / kSpD, c = e - q * kSpD;
    g[e] = bf16_bits_to_f32(grad_out[(((size_t)n * d.Lq + q) * d.M + m) * kSpD + c]);
  }
  // ---- taps -> keys (the arithmetic of msda_bwd_d48_f32_kernel: same in-map tests, same weights)
  for (int tp = tid; tp < kSpTaps; tp += 256) {
    unsigned key = 0xffffffffu;
    float w = 0.f;
    if (tp < ntap) {
      const int corner = tp & 3, sp = tp >> 2, p = sp % d.P, q = sp / d.P;
      const long long li = ((((long long)n * d.Lq + q) * d.M + m) * d.L + l) * d.P + p;
      const float lx = loc[2 * li], ly = loc[2 * li + 1], a = attn[li];
      const float y = px_coord(ly, H), x = px_coord(lx, W);
      const bool inside = (y > -1.f) && (x > -1.f) && (y < (float)H) && (x < (float)W);
      const float yf = floorf(y), xf = floorf(x);
      const float lh = y - yf, lw = x - xf;
      const int yy = (int)yf + (corner >> 1), xx = (int)xf + (corner & 1);
      if (inside && yy >= 0 && yy <= H - 1 && xx >= 0 && xx <= W - 1) {
        w = ((corner >> 1) ? lh : 1.f - lh) * ((corner & 1) ? lw : 1.f - lw) * a;
        key = ((unsigned)(yy * W + xx) << 10) | (unsigned)tp;
      }
    }
    keys[tp] = key;
    wa[tp] = w;
  }
  __syncthreads();
  // ---- bitonic sort of the 1 024 keys (ascending): 55 stages, two compare-exchanges per thread and stage
  for (unsigned k = 2; k <= kSpTaps; k <<= 1)
    for (unsigned j = k >> 1; j > 0; j >>= 1) {
#pragma unroll
      for (int rep = 0; rep < 2; ++rep) {
        const unsigned t = (unsigned)tid + 256u * rep;            // pair index 0 .. 511
        const unsigned i = ((t & ~(j - 1)) << 1) | (t & (j - 1)); // the lower element of the pair
        const unsigned ixj = i | j;
        const unsigned a = keys[i], b = keys[ixj];
        const bool up = (i & k) == 0;
        if ((a > b) == up) { keys[i] = b; keys[ixj] = a; }
      }
      __syncthreads();
    }
  // ---- one store per touched pixel
  const int P4 = d.P * 4;
  for (int i = tid; i < kSpTaps; i += 256) {
    const unsigned key = keys[i];
    if (key == 0xffffffffu) continue;
    const unsigned pix = key >> 10;
    if (i > 0 && (keys[i - 1] >> 10) == pix) continue;            // not the first tap of its pixel
    float acc[kSpD];
#pragma unroll
    for (int c = 0; c < kSpD; ++c) acc[c] = 0.f;
    for (int j = i; j < kSpTaps; ++j) {
      const unsigned kj = keys[j];
      if (kj == 0xffffffffu || (kj >> 10) != pix) break;
      const unsigned tp = kj & 1023u;
      const float w = wa[tp];
      const float4 *row = reinterpret_cast<const float4 *>(g + (tp / P4) * kSpD);
#pragma unroll
      for (int c4 = 0; c4 < kSpD / 4; ++c4) {
        const float4 v = row[c4];
        acc[4 * c4] = fmaf(w, v.x, acc[4 * c4]); acc[4 * c4 + 1] = fmaf(w, v.y, acc[4 * c4 + 1]);
        acc[4 * c4 + 2] = fmaf(w, v.z, acc[4 * c4 + 2]); acc[4 * c4 + 3] = fmaf(w, v.w, acc[4 * c4 + 3]);
      }
    }
    uint4 *dst = reinterpret_cast<uint4 *>(grad_value + (((size_t)n * d.S + start + pix) * d.M + m) * kSpD);
#pragma unroll
    for (int c8 = 0; c8 < kSpD / 8; ++c8) {
      auto pk = [](float x, float y) { return (unsigned)f32_to_bf16_bits(x) | ((unsigned)f32_to_bf16_bits(y) << 16); };
      uint4 o;
      o.x = pk(acc[8 * c8], acc[8 * c8 + 1]); o.y = pk(acc[8 * c8 + 2], acc[8 * c8 + 3]);
      o.z = pk(acc[8 * c8 + 4], acc[8 * c8 + 5]); o.w = pk(acc[8 * c8 + 6], acc[8 * c8 + 7]);
      dst[c8] = o;
    }
  }
}

}  // namespace snipper
