// misc_kernels.cuh -- small bandwidth-bound helpers that replace chains of ATen launches around the hot path (gfx950).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "msda_common.cuh"

namespace snipper {

constexpr int kSumMaxSrc = 8;
struct SumSrcs { const uint16_t *p[kSumMaxSrc]; int n; };

// out = bf16(sum_i float(src_i)), 8 elements per thread (16-byte accesses).  The gradient of a tensor with several
// consumers (the encoder memory's bf16 twin: six decoder layers' value projections, reference
// models/deformable_transformer.py:290-295) arrives as one tensor per consumer; autograd adds them pairwise -- n - 1
// launches of 3 x 60 MB each, every partial sum rounded to bf16.  Here: one pass, float32 accumulation, one rounding.
__global__ __launch_bounds__(256) void sum_bf16_kernel(SumSrcs s, uint16_t *__restrict__ out, long long n8) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n8) return;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int k = 0; k < s.n; ++k) {
    const uint4 v = reinterpret_cast<const uint4 *>(s.p[k])[i];
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      acc[2 * c] += __uint_as_float(w[c] << 16);
      acc[2 * c + 1] += __uint_as_float(w[c] & 0xffff0000u);
    }
  }
  uint4 o;
  o.x = (unsigned)f32_to_bf16_bits(acc[0]) | ((unsigned)f32_to_bf16_bits(acc[1]) << 16);
  o.y = (unsigned)f32_to_bf16_bits(acc[2]) | ((unsigned)f32_to_bf16_bits(acc[3]) << 16);
  o.z = (unsigned)f32_to_bf16_bits(acc[4]) | ((unsigned)f32_to_bf16_bits(acc[5]) << 16);
  o.w = (unsigned)f32_to_bf16_bits(acc[6]) | ((unsigned)f32_to_bf16_bits(acc[7]) << 16);
  reinterpret_cast<uint4 *>(out)[i] = o;
}

// Frozen 7x7 stem's input: float32 images [N, 3, H, W] (planar, as the data loader delivers them: values in [0, 1],
// datasets/transforms.py:143) -> bf16 [N, H, W, 4] with a zero fourth channel, the layout csrc/gemm_bf16.cuh's stem kernel
// reads (a tap row of a pixel pair = one aligned 16-byte piece).  Replaces a channels_last copy, a fill and a strided copy.
// One thread = 4 consecutive pixels of a row (W % 4 == 0): three 16-byte loads, two 16-byte stores.
__global__ __launch_bounds__(256) void stem_pack_kernel(const float *__restrict__ x, long long plane, long long quads,
                                                        uint16_t *__restrict__ out) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;      // quad index over N * H * W / 4
  if (i >= quads) return;
  const long long per_image = plane / 4, n = i / per_image, q = i - n * per_image;
  const float *base = x + n * 3 * plane + q * 4;
  const float4 r = *reinterpret_cast<const float4 *>(base), g = *reinterpret_cast<const float4 *>(base + plane),
               b = *reinterpret_cast<const float4 *>(base + 2 * plane);
  auto px = [](float a, float c) { return (unsigned)f32_to_bf16_bits(a) | ((unsigned)f32_to_bf16_bits(c) << 16); };
  uint4 o0, o1;
  o0.x = px(r.x, g.x); o0.y = px(b.x, 0.f); o0.z = px(r.y, g.y); o0.w = px(b.y, 0.f);
  o1.x = px(r.z, g.z); o1.y = px(b.z, 0.f); o1.z = px(r.w, g.w); o1.w = px(b.w, 0.f);
  uint4 *dst = reinterpret_cast<uint4 *>(out) + i * 2;
  dst[0] = o0; dst[1] = o1;
}

// ---- pos16[b, t, S, C] = bf16(cat_l(pos_l[b, t, hw_l, C] + level_embed[l])) in one launch --------------------------------
// (reference models/deformable_transformer.py:118-121: lvl_pos_embed = pos_embed + level_embed[lvl], flattened and
// concatenated.)  Three torch.add(..., out=strided bf16 slice) launches ran on the generic strided kernel: 92 us per step for
// 121 MB read + 60 MB written; 8 channels per thread here.
constexpr int kLpMaxLevels = 4;
struct LevelPosArgs {
  const float *pos[kLpMaxLevels];      // level l: [bt][hw_l][C] float32
  const float *level_embed;            // [levels][C]
  uint16_t *out;                       // [bt][S][C] bf16
  int hw[kLpMaxLevels], start[kLpMaxLevels];
  int levels, bt, S, C;
};
__global__ __launch_bounds__(256) void level_pos_kernel(LevelPosArgs g) {
  const int c8 = g.C >> 3;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)g.bt * g.S * c8) return;
  const long long r = i / c8;
  const int c = (int)(i - r * c8) * 8;
  const int s = (int)(r % g.S);
  const long long bt = r / g.S;
  int l = 0;
#pragma unroll
  for (int k = 1; k < kLpMaxLevels; ++k)
    if (k < g.levels && s >= g.start[k]) l = k;
  const float *p = g.pos[l] + ((bt * g.hw[l] + (s - g.start[l])) * g.C + c);
  const float *e = g.level_embed + l * g.C + c;
  const float4 a = *reinterpret_cast<const float4 *>(p), b = *reinterpret_cast<const float4 *>(p + 4);
  const float4 ea = *reinterpret_cast<const float4 *>(e), eb = *reinterpret_cast<const float4 *>(e + 4);
  auto pk = [](float x, float y) { return (unsigned)f32_to_bf16_bits(x) | ((unsigned)f32_to_bf16_bits(y) << 16); };
  uint4 o;
  o.x = pk(a.x + ea.x, a.y + ea.y); o.y = pk(a.z + ea.z, a.w + ea.w);
  o.z = pk(b.x + eb.x, b.y + eb.y); o.w = pk(b.z + eb.z, b.w + eb.w);
  *reinterpret_cast<uint4 *>(g.out + r * g.C + c) = o;
}

// ---- bf16 working copies of many float32 weights in ONE launch, a per-output-channel scale folded in ----------------------
// dst[e] = bf16(src[e] * scale[e / inner]) (scale == nullptr: plain cast) for a TABLE of tensors that lives in device memory
// (the per-step refresh of the weight shadows, snipper_amd/shadow.py: ~160 tensors, 41 M elements; the table is a constant of
// the model and is uploaded once).  PyTorch's multi-tensor path for the same work is a _foreach_mul by full-size copies of the
// frozen-BN scales into temporaries plus _foreach_copy_ casts: 162 us and 93 MB of scale copies per step; this reads every
// weight once and writes its bf16 copy: ~45 us.  Block b belongs to the item i with block_end[i - 1] <= b < block_end[i]
// (binary search); 2 048 elements per block, 8 per thread; numel % 8 == 0, inner % 8 == 0 or no scale.
struct CastItem {
  const float *src; uint16_t *dst; const float *scale;
  long long numel;
  int inner, pad;
};
__global__ __launch_bounds__(256) void cast_scale_table_kernel(const CastItem *__restrict__ items, const int *__restrict__ block_end,
                                                                int n_items) {
  int lo = 0, hi = n_items - 1;
  while (lo < hi) {                                   // first item whose block_end exceeds this block
    const int mid = (lo + hi) >> 1;
    if ((int)blockIdx.x < block_end[mid]) hi = mid; else lo = mid + 1;
  }
  const CastItem it = items[lo];
  const long long e = ((long long)((int)blockIdx.x - (lo ? block_end[lo - 1] : 0)) * 256 + threadIdx.x) * 8;
  if (e >= it.numel) return;
  const float4 a = *reinterpret_cast<const float4 *>(it.src + e), b = *reinterpret_cast<const float4 *>(it.src + e + 4);
  const float sc = it.scale ? it.scale[e / it.inner] : 1.f;
  typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
  auto pk = [](float x, float y) {
    return (unsigned)__builtin_bit_cast(uint16_t, (__bf16)x) | ((unsigned)__builtin_bit_cast(uint16_t, (__bf16)y) << 16);
  };
  const u32x4 o = {pk(a.x * sc, a.y * sc), pk(a.z * sc, a.w * sc), pk(b.x * sc, b.y * sc), pk(b.z * sc, b.w * sc)};
  *reinterpret_cast<u32x4 *>(it.dst + e) = o;
}

}  // namespace snipper
