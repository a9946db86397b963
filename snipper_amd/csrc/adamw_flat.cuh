// adamw_flat.cuh -- global-norm gradient clipping + AdamW on ONE flat float32 buffer (gfx950).
//
// The reference's optimizer phase is `torch.nn.utils.clip_grad_norm_(model.parameters(), 0.1)` (engine.py:74) followed by
// `torch.optim.AdamW` over three parameter groups (main.py:201-221).  With the parameters and gradients already in one flat
// buffer (flat_params.py / grad_sync.py) PyTorch runs it as 3 norm launches + 7 small ones + 3 scale launches + 4 fused
// AdamW launches = 0.49 ms per step; element for element it is
//
//     coef = min(1, max_norm / (||g||_2 + 1e-6))                     g' = coef * g
//     p   *= 1 - lr * weight_decay                                    (decoupled decay, per group)
//     m    = m + (g' - m) * (1 - beta1)        v = beta2 * v + (1 - beta2) * g' * g'
//     p   -= (lr / (1 - beta1^t)) * m / (sqrt(v) / sqrt(1 - beta2^t) + eps)
//
// i.e. one read of the gradient for the norm and one pass over p, g, m, v.  Two launches:
//   gradnorm_partials_kernel   per-workgroup partial sums of g^2 (fixed order inside a workgroup)
//   adamw_clip_kernel          every workgroup adds the partials in the same fixed order (a few KB), derives coef, and updates
//                              its grid-stride share of the buffer, 16 bytes per lane and array; the scaled gradient is not
//                              written back (nothing reads it after the step)
// The result is deterministic (no atomics).  Groups are contiguous ranges of the flat buffer that start on 64-element
// boundaries (grad_sync.FLAT_ALIGN), so a 4-element chunk never straddles two groups.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace snipper {

constexpr int kAdamThreads = 256;
constexpr int kAdamMaxSeg = 8;
constexpr int kAdamMaxParts = 4096;

struct AdamSeg { long long begin, end; float lr, wd; };
struct AdamArgs {
  float *p; const float *g; float *m; float *v;
  long long n;                       // elements (multiple of 4)
  AdamSeg seg[kAdamMaxSeg];
  int nseg;
  float beta1, beta2, eps, inv_bc1, inv_bc2_sqrt;      // 1 / (1 - beta1^t), 1 / sqrt(1 - beta2^t)
  const float *partials; int nparts;                   // of g^2 (nullptr / 0: no clipping)
  float max_norm;
  float *norm_out;                                     // [1] or nullptr: ||g||_2 before clipping
};

__device__ __forceinline__ float adam_block_sum(float v, float *red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float s = 0.f;
#pragma unroll
  for (int w = 0; w < kAdamThreads / 64; ++w) s += red[w];
  __syncthreads();
  return s;
}

__global__ __launch_bounds__(kAdamThreads) void gradnorm_partials_kernel(const float *__restrict__ g, long long n4,
                                                                         float *__restrict__ partials) {
  __shared__ float red[kAdamThreads / 64];
  const float4 *g4 = reinterpret_cast<const float4 *>(g);
  float s = 0.f;
  for (long long i = (long long)blockIdx.x * kAdamThreads + threadIdx.x; i < n4; i += (long long)gridDim.x * kAdamThreads) {
    const float4 x = g4[i];
    s += (x.x * x.x + x.y * x.y) + (x.z * x.z + x.w * x.w);
  }
  s = adam_block_sum(s, red);
  if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

__global__ __launch_bounds__(kAdamThreads) void adamw_clip_kernel(AdamArgs a) {
  __shared__ float red[kAdamThreads / 64];
  float coef = 1.f;
  if (a.partials) {
    float s = 0.f;
    for (int i = threadIdx.x; i < a.nparts; i += kAdamThreads) s += a.partials[i];
    s = adam_block_sum(s, red);
    const float norm = sqrtf(s);
    if (a.norm_out && blockIdx.x == 0 && threadIdx.x == 0) *a.norm_out = norm;
    if (a.max_norm > 0.f) coef = fminf(1.f, a.max_norm / (norm + 1e-6f));
  }
  float4 *p4 = reinterpret_cast<float4 *>(a.p), *m4 = reinterpret_cast<float4 *>(a.m), *v4 = reinterpret_cast<float4 *>(a.v);
  const float4 *g4 = reinterpret_cast<const float4 *>(a.g);
  const long long n4 = a.n >> 2;
  const float omb1 = 1.f - a.beta1, omb2 = 1.f - a.beta2;
  for (long long i = (long long)blockIdx.x * kAdamThreads + threadIdx.x; i < n4; i += (long long)gridDim.x * kAdamThreads) {
    const long long e = i << 2;
    float lr = 0.f, wd = 0.f;
    bool in = false;
#pragma unroll
    for (int k = 0; k < kAdamMaxSeg; ++k) {
      if (k < a.nseg && e >= a.seg[k].begin && e < a.seg[k].end) { lr = a.seg[k].lr; wd = a.seg[k].wd; in = true; }
    }
    if (!in) continue;
    float4 p = p4[i], m = m4[i], v = v4[i];
    const float4 g = g4[i];
    const float decay = 1.f - lr * wd, step = lr * a.inv_bc1;
    auto upd = [&](float &pp, float &mm, float &vv, float gg) {
      gg *= coef;
      pp *= decay;
      mm = mm + (gg - mm) * omb1;
      vv = a.beta2 * vv + omb2 * gg * gg;
      const float denom = sqrtf(vv) * a.inv_bc2_sqrt + a.eps;
      pp -= step * (mm / denom);
    };
    upd(p.x, m.x, v.x, g.x); upd(p.y, m.y, v.y, g.y); upd(p.z, m.z, v.z, g.z); upd(p.w, m.w, v.w, g.w);
    p4[i] = p; m4[i] = m; v4[i] = v;
  }
}

}  // namespace snipper
