// heatmap_blur.cuh -- Gaussian blur of the joint heat-map targets (gfx950).
//
// reference models/model.py:447-483: the target of the heat-map loss is a one-hot joint map per (sample, joint, frame)
// blurred with torchvision's gaussian_blur (separable kernel, sigma = 0.3 ((k - 1) / 2 - 1) + 0.8, reflect padding).
// In PyTorch that is a padding kernel plus two float32 convolutions through the vendor library per level -- the last
// library convolutions of the training step (50 - 60 us per level, and a solver search in the first steps).  The maps are
// small (75 x 100 at most) and the kernel has k <= 31 taps per axis: one thread per output pixel evaluates
//     out[y][x] = sum_kx w[kx] * ( sum_ky w[ky] * min(in[refl(y + ky - r)][refl(x + kx - r)], clamp) )
// -- the vertical pass inside the horizontal one, the association of the two-pass formulation -- reading the image through
// the cache; the clamp of the scatter-added one-hot map (several joints on one pixel still give 1) is applied on the fly.
#pragma once
#include <hip/hip_runtime.h>

namespace snipper {

constexpr int kBlurMaxTaps = 31;
struct BlurArgs {
  const float *in;   // [n_images][H][W]
  float *out;        // [n_images][H][W]
  int n_images, H, W, k;
  float clamp_max;   // inputs are read as min(in, clamp_max); +inf = no clamp
  float w[kBlurMaxTaps];
};

__device__ __forceinline__ int blur_reflect(int i, int n) {      // torch "reflect": -1 -> 1, n -> n - 2 (needs pad < n)
  i = i < 0 ? -i : i;
  return i >= n ? 2 * (n - 1) - i : i;
}

__global__ __launch_bounds__(256) void heatmap_blur_kernel(BlurArgs g) {
  const long long total = (long long)g.n_images * g.H * g.W;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int x = (int)(i % g.W);
  const long long t = i / g.W;
  const int y = (int)(t % g.H);
  const float *img = g.in + (t / g.H) * (long long)g.H * g.W;
  const int r = g.k / 2;
  float acc = 0.f;
  for (int kx = 0; kx < g.k; ++kx) {
    const int xx = blur_reflect(x + kx - r, g.W);
    float col = 0.f;
    for (int ky = 0; ky < g.k; ++ky) {
      const int yy = blur_reflect(y + ky - r, g.H);
      col = fmaf(g.w[ky], fminf(img[(long long)yy * g.W + xx], g.clamp_max), col);
    }
    acc = fmaf(g.w[kx], col, acc);
  }
  g.out[i] = acc;
}

}  // namespace snipper
