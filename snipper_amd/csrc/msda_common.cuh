// msda_common.cuh -- shared device helpers for the gfx950 deformable-attention kernels.
//
// Written for CDNA4 only (wave64, MI355X).  Semantics restated from the reference's
// sampling rule (/root/reference/models/ops/src/cuda/ms_deform_im2col_cuda.cuh:285-291
// for the pixel mapping / in-range test, :33-84 for the four-tap read, :87-159 for the
// tap gradients); no code is shared with it.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace snipper {

constexpr int kWave = 64;  // gfx950 wavefront

// ---- storage <-> compute conversions -------------------------------------------------
// bf16 tensors cross the ABI as raw uint16_t bits.
__device__ __forceinline__ float bf16_bits_to_f32(uint16_t b) {
  return __uint_as_float(static_cast<uint32_t>(b) << 16);
}
__device__ __forceinline__ uint16_t f32_to_bf16_bits(float f) {
  // plain cast: hipcc emits v_cvt_pk_bf16_f32 (RNE, NaN stays NaN) on gfx950
  return __builtin_bit_cast(uint16_t, static_cast<__bf16>(f));
}

template <typename VT, typename CT> struct Conv;
template <> struct Conv<float, float> {
  static __device__ __forceinline__ float ld(const float *p) { return *p; }
  static __device__ __forceinline__ void st(float *p, float v) { *p = v; }
};
template <> struct Conv<double, double> {
  static __device__ __forceinline__ double ld(const double *p) { return *p; }
  static __device__ __forceinline__ void st(double *p, double v) { *p = v; }
};
template <> struct Conv<uint16_t, float> {
  static __device__ __forceinline__ float ld(const uint16_t *p) { return bf16_bits_to_f32(*p); }
  static __device__ __forceinline__ void st(uint16_t *p, float v) { *p = f32_to_bf16_bits(v); }
};

// ---- one sampling point, decoded once and shared by the lanes that work on its row ---
// pix[k] is the element offset of tap k's head-row relative to the (batch, head) base of
// `value` (so value[batch_base + head*D + pix[k] + c] is channel c), or -1 when the tap
// lies outside the feature map.  Tap order: (y0,x0) (y0,x1) (y1,x0) (y1,x1).
template <typename CT> struct SamplePoint {
  CT lh, lw;   // fractional parts along y / x
  CT a;        // attention weight (0 when the sample is skipped)
  CT Hf, Wf;   // level size as CT (scales grad_loc back to normalised units)
  int pix[4];
};

template <typename CT>
__device__ __forceinline__ SamplePoint<CT> decode_sample(CT lx, CT ly, CT a, int H, int W,
                                                         int level_start, int row_stride /* M*D */) {
  SamplePoint<CT> sp;
  const CT y = ly * CT(H) - CT(0.5);
  const CT x = lx * CT(W) - CT(0.5);
  const bool inside = (y > CT(-1)) && (x > CT(-1)) && (y < CT(H)) && (x < CT(W));
  const CT yf = floor(y), xf = floor(x);
  const int y0 = static_cast<int>(yf), x0 = static_cast<int>(xf);
  sp.lh = inside ? y - yf : CT(0);   // skipped sample: all-zero weights, even for NaN/inf loc
  sp.lw = inside ? x - xf : CT(0);
  sp.a = inside ? a : CT(0);
  sp.Hf = CT(H);
  sp.Wf = CT(W);
  const bool yok0 = inside && (y0 >= 0), yok1 = inside && (y0 + 1 <= H - 1);
  const bool xok0 = (x0 >= 0), xok1 = (x0 + 1 <= W - 1);
  const int p00 = (level_start + y0 * W + x0) * row_stride;
  sp.pix[0] = (yok0 && xok0) ? p00 : -1;
  sp.pix[1] = (yok0 && xok1) ? p00 + row_stride : -1;
  sp.pix[2] = (yok1 && xok0) ? p00 + W * row_stride : -1;
  sp.pix[3] = (yok1 && xok1) ? p00 + (W + 1) * row_stride : -1;
  return sp;
}

// Sum over the G (power of two, <= 64) consecutive lanes of an aligned lane group.
template <int G, typename CT> __device__ __forceinline__ CT group_sum(CT v) {
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, G);
  return v;
}

}  // namespace snipper
