// heatmap_loss.cuh -- the heat-map targets' scatter and the heat-map loss of the training criterion (gfx950).
//
// reference models/model.py:447-483 (`loss_heatmap`): per level, a one-hot joint map per (sample, joint, frame) is blurred
// and compared (sum of squared differences / n_heads) with the first K channels of every head of the encoder memory
// (models/deformable_transformer.py:141-149: the "heatmaps" are strided VIEWS of the memory [bs, T, S, C]).
//
// In PyTorch the targets' index arithmetic is ~30 tiny launches on [levels, persons, T, K] tensors, the loss three
// mse_loss + sum + scale + add chains on expanded / strided views, and the backward per level an mse_backward, a zero-fill of
// the level's full [.., heads, head_dim] grid, a strided copy into it, and a concatenation of the levels into the memory's
// gradient: ~75 launches of a few microseconds each in the host-bound stretch of the step (decoder -> criterion -> decoder
// backward), 0.2 ms of GPU time.  Here: one scatter launch (a valid joint stores 1.0: the blur clamps at 1, so the count
// of joints on a pixel does not matter), one forward launch over all levels (deterministic block sums) and one backward
// launch that writes the WHOLE gradient of the memory (zeros outside the K heat-map channels of each head).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace snipper {

constexpr int kHlMaxLevels = 4;

struct HeatmapScatterArgs {
  const float *kpts;           // [n_person][Tk][K][3] (x, y in [0, 1), visibility)
  const long long *sample;     // [n_person] sample index of the person
  float *out;                  // level l's maps [bs][K][T][h][w] start at base[l]; zeroed by the caller
  long long base[kHlMaxLevels];
  int h[kHlMaxLevels], w[kHlMaxLevels];
  int levels, n_person, Tk, T, K;
};

__global__ __launch_bounds__(256) void heatmap_scatter_kernel(HeatmapScatterArgs g) {
  const long long per_level = (long long)g.n_person * g.T * g.K;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= per_level * g.levels) return;
  const int l = (int)(i / per_level);
  long long r = i - l * per_level;
  const int k = (int)(r % g.K); r /= g.K;
  const int t = (int)(r % g.T);
  const int n = (int)(r / g.T);
  const float *p = g.kpts + (((long long)n * g.Tk + t) * g.K + k) * 3;
  // the same arithmetic as the tensor formulation: float32 product, truncation toward zero, range test, then a clamp
  // that can no longer change anything
  const long long x = (long long)(p[0] * (float)g.w[l]), y = (long long)(p[1] * (float)g.h[l]);
  if (!(p[2] > 0.f) || x < 0 || x >= g.w[l] || y < 0 || y >= g.h[l]) return;
  const long long plane = (g.sample[n] * g.K + k) * g.T + t;
  g.out[g.base[l] + (plane * g.h[l] + y) * g.w[l] + x] = 1.0f;
}

struct HeatmapLossArgs {
  const float *mem;            // [bs * T * S][C]: the encoder memory
  const float *tm[kHlMaxLevels];       // level l: blurred targets [bs][K][T][h][w]
  int hw[kHlMaxLevels], start[kHlMaxLevels];     // pixels of a level's map, its first position in S
  int levels, bs, T, S, C, nhead, D, K;
  float *partial;              // forward: [gridDim.x] block sums of squared differences
  const float *gscale;         // backward: dL / d(sum of squared differences), a device scalar
  float *gmem;                 // backward: [bs * T * S][C], every element written
};

__device__ __forceinline__ int hl_level(const HeatmapLossArgs &g, int s) {
  int l = 0;
#pragma unroll
  for (int i = 1; i < kHlMaxLevels; ++i)
    if (i < g.levels && s >= g.start[i]) l = i;
  return l;
}

// one wave per row (position of one frame of one sample), grid-stride; lanes over (head, joint)
__global__ __launch_bounds__(256) void heatmap_loss_fwd_kernel(HeatmapLossArgs g) {
  __shared__ float red[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long rows = (long long)g.bs * g.T * g.S;
  const int ne = g.nhead * g.K;
  float sum = 0.f;
  for (long long r = (long long)blockIdx.x * 4 + wave; r < rows; r += (long long)gridDim.x * 4) {
    const int s = (int)(r % g.S);
    const long long bt = r / g.S;
    const int t = (int)(bt % g.T), b = (int)(bt / g.T);
    const int l = hl_level(g, s);
    const int p = s - g.start[l];
    const float *tm = g.tm[l] + ((long long)b * g.K * g.T + t) * g.hw[l] + p;       // + k * T * hw
    const float *m = g.mem + r * g.C;
    for (int e = lane; e < ne; e += 64) {
      const int head = e / g.K, k = e - head * g.K;
      const float d = tm[(long long)k * g.T * g.hw[l]] - m[head * g.D + k];
      sum = fmaf(d, d, sum);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
  if (lane == 0) red[wave] = sum;
  __syncthreads();
  if (threadIdx.x == 0) g.partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// one thread per 4 channels of a row: d/dmem of sum (tm - mem)^2 = 2 (mem - tm) on the heat-map channels, 0 elsewhere
__global__ __launch_bounds__(256) void heatmap_loss_bwd_kernel(HeatmapLossArgs g) {
  const int c4 = g.C >> 2;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long rows = (long long)g.bs * g.T * g.S;
  if (i >= rows * c4) return;
  const long long r = i / c4;
  const int c0 = (int)(i - r * c4) * 4;
  float4 out = make_float4(0.f, 0.f, 0.f, 0.f);
  const int head = c0 / g.D, k0 = c0 - head * g.D;       // (D % 4 == 0: a chunk never straddles two heads)
  if (k0 < g.K) {
    const float gs = 2.f * g.gscale[0];
    const int s = (int)(r % g.S);
    const long long bt = r / g.S;
    const int t = (int)(bt % g.T), b = (int)(bt / g.T);
    const int l = hl_level(g, s);
    const float *tm = g.tm[l] + ((long long)b * g.K * g.T + t) * g.hw[l] + (s - g.start[l]);
    const float4 m = *reinterpret_cast<const float4 *>(g.mem + r * g.C + c0);
    const float mv[4] = {m.x, m.y, m.z, m.w};
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = k0 + j < g.K ? gs * (mv[j] - tm[(long long)(k0 + j) * g.T * g.hw[l]]) : 0.f;
    out = make_float4(o[0], o[1], o[2], o[3]);
  }
  *reinterpret_cast<float4 *>(g.gmem + r * g.C + c0) = out;
}

}  // namespace snipper
