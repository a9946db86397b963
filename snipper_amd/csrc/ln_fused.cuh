// ln_fused.cuh -- residual + dropout + LayerNorm in one pass, forward and backward (gfx950).
//
//   s = x + dropout_p(z)            y = LayerNorm(s) * gamma + beta
//
// This is the tail of every sub-layer of the transformer (reference models/deformable_transformer.py:200-216 for the
// encoder layer, :266-300 for the decoder layer: ``src = src + self.dropoutN(src2); src = self.normN(src)``).  On
// the encoder's 79 000 x 384 token matrix PyTorch runs it as dropout, add, LayerNorm and a cast for the next
// Linear -- four passes over HBM forward and five backward; here it is one pass each way:
//
//   forward : reads x (f32 / bf16) and z (bf16 / f32) once; writes the pre-norm sum s (f32, for the backward), the
//             per-row mean / rstd, one byte of keep-bits per 4 elements, and up to three views of the result -- y in
//             f32 (the residual stream), y in bf16 (input of the next Linear) and bf16(y + pos) (the query of the
//             next deformable attention) -- so no separate cast or add kernel follows.
//   backward: takes the gradients of those three outputs (any subset), adds them in registers, and writes dL/dx
//             (= dL/ds), dL/dz (masked, rescaled) and per-workgroup partial sums of dL/dgamma, dL/dbeta, which a
//             small second kernel adds up in a fixed order.
//
// One wave per row, 4-element chunks per lane (16-byte f32 / 8-byte bf16 accesses), C <= 1024, C % 4 == 0.
// Dropout uses a counter-based hash of (seed, element index): the mask is reproducible from the seed and is also
// stored (bits) so the backward does not depend on it.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace snipper {

constexpr int kLnMaxIter = 4;        // 64 lanes x 4 elements x 4 = C <= 1024
constexpr int kLnThreads = 256;      // 4 rows per workgroup

struct LnFwdArgs {
  // "lazy" residual: x is the PREVIOUS LayerNorm's saved pre-norm sum and its output is recomputed on load,
  //   x := (x - x_mean[row]) * x_rstd[row] * x_gamma + x_beta      (float32 x only)
  // -- the expression that kernel would have written as y32, so a chain of sub-layers never materialises the float32
  // residual stream between two LayerNorms (121 MB written and 121 MB allocated less per encoder sub-layer).  NULL = plain x.
  const float *x_mean, *x_rstd, *x_gamma, *x_beta;
  const void *x; int x_dt;           // [rows][C] residual; dtype code 0 = f32, 1 = bf16
  const void *z; int z_dt;           // [rows][C] branch output, or nullptr (plain LayerNorm of x)
  const void *pos; int pos_dt;       // [rows][C] addend of the third output, or nullptr
  const float *gamma, *beta;         // [C]
  float *s_save;                     // [rows][C] or nullptr
  float *mean, *rstd;                // [rows] or nullptr
  uint8_t *keep;                     // [rows][C/4] low 4 bits = keep flags, or nullptr when p == 0
  float *y32; uint16_t *y16; uint16_t *yq16;    // outputs, each may be nullptr
  int rows, C;
  float p, eps;
  uint32_t seed_lo, seed_hi;
};

struct LnBwdArgs {
  const float *g32; const uint16_t *g16; const uint16_t *gq16;   // gradients of y32 / y16 / yq16, each may be nullptr
  const float *s_save, *mean, *rstd, *gamma;
  const uint8_t *keep;               // or nullptr (p == 0 or no z)
  void *dx; int dx_dt;               // [rows][C] or nullptr
  void *dz; int dz_dt;               // [rows][C] or nullptr
  float *part;                       // [gridDim.x][2][C] partial dgamma / dbeta
  int rows, C;
  float p;
};

__device__ __forceinline__ uint32_t ln_hash(uint32_t v) {
  v ^= v >> 16; v *= 0x7feb352du; v ^= v >> 15; v *= 0x846ca68bu; v ^= v >> 16;
  return v;
}
__device__ __forceinline__ uint32_t ln_rand(uint32_t idx, uint32_t seed_lo, uint32_t seed_hi) {
  return ln_hash(ln_hash(idx + seed_lo * 0x9e3779b9u) ^ seed_hi);
}
__device__ __forceinline__ float4 ln_load4(const void *base, int dt, long long e) {
  if (dt == 0) return *reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(base) + e);
  const uint2 r = *reinterpret_cast<const uint2 *>(reinterpret_cast<const uint16_t *>(base) + e);
  return make_float4(__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u),
                     __uint_as_float(r.y << 16), __uint_as_float(r.y & 0xffff0000u));
}
__device__ __forceinline__ uint32_t ln_pack2(float a, float b) {
  return (uint32_t)__builtin_bit_cast(uint16_t, (__bf16)a) | ((uint32_t)__builtin_bit_cast(uint16_t, (__bf16)b) << 16);
}
__device__ __forceinline__ void ln_store4(void *base, int dt, long long e, float4 v) {
  if (dt == 0) {
    *reinterpret_cast<float4 *>(reinterpret_cast<float *>(base) + e) = v;
  } else {
    *reinterpret_cast<uint2 *>(reinterpret_cast<uint16_t *>(base) + e) = make_uint2(ln_pack2(v.x, v.y), ln_pack2(v.z, v.w));
  }
}
__device__ __forceinline__ float ln_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__global__ __launch_bounds__(kLnThreads) void ln_fused_fwd_kernel(LnFwdArgs a) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (kLnThreads / 64) + (threadIdx.x >> 6);
  if (row >= a.rows) return;                       // whole waves leave together: no partial-EXEC shuffles below
  const long long base = (long long)row * a.C;
  const float keep_scale = a.p > 0.f ? 1.f / (1.f - a.p) : 1.f;
  const uint32_t thresh16 = (uint32_t)fminf(a.p * 65536.f + 0.5f, 65535.f);
  float4 v[kLnMaxIter];
  float sum = 0.f;
  const float xm = a.x_mean ? a.x_mean[row] : 0.f, xr = a.x_mean ? a.x_rstd[row] : 1.f;
#pragma unroll
  for (int it = 0; it < kLnMaxIter; ++it) {
    const int c = (lane + 64 * it) * 4;
    v[it] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < a.C) {
      float4 x = ln_load4(a.x, a.x_dt, base + c);
      if (a.x_mean) {
        const float4 gm = *reinterpret_cast<const float4 *>(a.x_gamma + c), bt = *reinterpret_cast<const float4 *>(a.x_beta + c);
        x.x = (x.x - xm) * xr * gm.x + bt.x; x.y = (x.y - xm) * xr * gm.y + bt.y;
        x.z = (x.z - xm) * xr * gm.z + bt.z; x.w = (x.w - xm) * xr * gm.w + bt.w;
      }
      if (a.z) {
        float4 z = ln_load4(a.z, a.z_dt, base + c);
        if (a.p > 0.f) {
          // ONE counter-based hash per 4-element chunk (a function of (seed, element index), as before) + one xorshift32 step
          // give the chunk's four 16-bit uniforms, compared with p * 2^16 (csrc/wres_gemm_bf16.cuh's scheme).  Round 1-4
          // hashed every element twice over: 16 v_mul_lo_u32 per chunk, a quarter-rate instruction -- a third of this
          // kernel's vector time for bits the backward reads from `keep` anyway.
          uint32_t h1 = ln_rand((uint32_t)(base + c), a.seed_lo, a.seed_hi) | 1u, h2 = h1;
          h2 ^= h2 << 13; h2 ^= h2 >> 17; h2 ^= h2 << 5;
          const bool k0 = (h1 & 0xffffu) >= thresh16, k1 = (h1 >> 16) >= thresh16;
          const bool k2 = (h2 & 0xffffu) >= thresh16, k3 = (h2 >> 16) >= thresh16;
          z.x = k0 ? z.x * keep_scale : 0.f; z.y = k1 ? z.y * keep_scale : 0.f;
          z.z = k2 ? z.z * keep_scale : 0.f; z.w = k3 ? z.w * keep_scale : 0.f;
          if (a.keep) a.keep[((long long)row * a.C + c) >> 2] = (uint8_t)(k0 | (k1 << 1) | (k2 << 2) | (k3 << 3));
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
      if (a.y32) *reinterpret_cast<float4 *>(a.y32 + base + c) = y;
      if (a.y16) ln_store4(a.y16, 1, base + c, y);
      if (a.yq16) {
        const float4 ps = ln_load4(a.pos, a.pos_dt, base + c);
        ln_store4(a.yq16, 1, base + c, make_float4(y.x + ps.x, y.y + ps.y, y.z + ps.z, y.w + ps.w));
      }
    }
  }
}

// grid-stride over rows: every wave keeps its lanes' dgamma / dbeta in registers; one partial row per workgroup
// NIT = ceil(C / 256) register chunks per lane (C = 384 -> 2: 70 VGPRs instead of 130, twice the waves in flight).
template <int NIT>
__global__ __launch_bounds__(kLnThreads) __attribute__((amdgpu_waves_per_eu(NIT <= 2 ? 6 : 1, 8))) void ln_fused_bwd_kernel(LnBwdArgs a) {
  __shared__ float red[(kLnThreads / 64) * 2 * NIT * 256];               // [wave][2][NIT * 256 >= C]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float keep_scale = a.p > 0.f ? 1.f / (1.f - a.p) : 1.f;
  float4 dg[NIT], db[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) dg[it] = db[it] = make_float4(0.f, 0.f, 0.f, 0.f);
  const float inv_c = 1.f / (float)a.C;
  for (int row = blockIdx.x * (kLnThreads / 64) + wave; row < a.rows; row += gridDim.x * (kLnThreads / 64)) {
    const long long base = (long long)row * a.C;
    const float mean = a.mean[row], rstd = a.rstd[row];
    float4 gg[NIT], xh[NIT];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int c = (lane + 64 * it) * 4;
      gg[it] = xh[it] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (c < a.C) {
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a.g32) g = *reinterpret_cast<const float4 *>(a.g32 + base + c);
        if (a.g16) { const float4 t = ln_load4(a.g16, 1, base + c); g.x += t.x; g.y += t.y; g.z += t.z; g.w += t.w; }
        if (a.gq16) { const float4 t = ln_load4(a.gq16, 1, base + c); g.x += t.x; g.y += t.y; g.z += t.z; g.w += t.w; }
        const float4 s = *reinterpret_cast<const float4 *>(a.s_save + base + c);
        const float4 gm = *reinterpret_cast<const float4 *>(a.gamma + c);
        const float4 h = make_float4((s.x - mean) * rstd, (s.y - mean) * rstd, (s.z - mean) * rstd, (s.w - mean) * rstd);
        dg[it].x += g.x * h.x; dg[it].y += g.y * h.y; dg[it].z += g.z * h.z; dg[it].w += g.w * h.w;
        db[it].x += g.x; db[it].y += g.y; db[it].z += g.z; db[it].w += g.w;
        g.x *= gm.x; g.y *= gm.y; g.z *= gm.z; g.w *= gm.w;
        s1 += (g.x + g.y) + (g.z + g.w);
        s2 += (g.x * h.x + g.y * h.y) + (g.z * h.z + g.w * h.w);
        gg[it] = g; xh[it] = h;
      }
    }
    const float m1 = ln_wave_sum(s1) * inv_c, m2 = ln_wave_sum(s2) * inv_c;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int c = (lane + 64 * it) * 4;
      if (c < a.C) {
        float4 d;
        d.x = rstd * (gg[it].x - m1 - xh[it].x * m2); d.y = rstd * (gg[it].y - m1 - xh[it].y * m2);
        d.z = rstd * (gg[it].z - m1 - xh[it].z * m2); d.w = rstd * (gg[it].w - m1 - xh[it].w * m2);
        if (a.dx) ln_store4(a.dx, a.dx_dt, base + c, d);
        if (a.dz) {
          if (a.keep) {
            const uint32_t k = a.keep[(base + c) >> 2];
            d.x = (k & 1) ? d.x * keep_scale : 0.f; d.y = (k & 2) ? d.y * keep_scale : 0.f;
            d.z = (k & 4) ? d.z * keep_scale : 0.f; d.w = (k & 8) ? d.w * keep_scale : 0.f;
          }
          ln_store4(a.dz, a.dz_dt, base + c, d);
        }
      }
    }
  }
  // the 4 waves' partials meet in LDS; wave 0 writes the workgroup's row of the partial buffer
  float *mine = red + wave * 2 * NIT * 256;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {          // (channels >= C are never read back)
    *reinterpret_cast<float4 *>(mine + (lane + 64 * it) * 4) = dg[it];
    *reinterpret_cast<float4 *>(mine + NIT * 256 + (lane + 64 * it) * 4) = db[it];
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 2 * a.C; e += kLnThreads) {
    const int which = e >= a.C, c = which ? e - a.C : e;
    float sum = 0.f;
#pragma unroll
    for (int w = 0; w < kLnThreads / 64; ++w) sum += red[w * 2 * NIT * 256 + which * NIT * 256 + c];
    a.part[((long long)blockIdx.x * 2 + which) * a.C + c] = sum;
  }
}

// dgamma[c] = sum_b part[b][0][c], dbeta[c] = sum_b part[b][1][c]: kLnPgCh channels x (256 / kLnPgCh) slices of b per workgroup
// (the lanes of a slice read 32 contiguous bytes; the slices meet in LDS), summed in a fixed order.  Round 5: 8 channels x 32
// slices instead of 16 x 16 -- twice the workgroups (96 for C = 384) and half the dependent loads per thread: the 4.7 MB of
// partial rows of an encoder LayerNorm were read at 0.7 TB/s by 48 workgroups.
constexpr int kLnPgCh = 8, kLnPgSlices = 256 / kLnPgCh;
__global__ __launch_bounds__(256) void ln_param_grad_kernel(const float *part, int nparts, int C, float *dgamma, float *dbeta) {
  __shared__ float red[kLnPgSlices][kLnPgCh + 1];
  const int ci = threadIdx.x % kLnPgCh, slice = threadIdx.x / kLnPgCh;
  const int e = blockIdx.x * kLnPgCh + ci;                  // over 2*C
  float sum = 0.f;
  if (e < 2 * C) {
    const int which = e >= C, c = which ? e - C : e;
    const float *p = part + (long long)which * C + c;
#pragma unroll 8
    for (int b = slice; b < nparts; b += kLnPgSlices) sum += p[(long long)b * 2 * C];
  }
  red[slice][ci] = sum;
  __syncthreads();
  if (slice == 0 && e < 2 * C) {
    sum = 0.f;
#pragma unroll
    for (int k = 0; k < kLnPgSlices; ++k) sum += red[k][ci];
    if (e >= C) dbeta[e - C] = sum; else dgamma[e] = sum;
  }
}

}  // namespace snipper
