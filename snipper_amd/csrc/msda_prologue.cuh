// msda_prologue.cuh -- the element-wise work around the deformable-attention core op, fused.
//
// In the reference every line below is its own PyTorch kernel over tensors of N*T*S*C (121 MB) or
// N*T*Lq*M*L*P*2 (61 MB) elements, forward and backward
// (/root/reference/models/ops/modules/ms_deform_attn.py):
//   :116      value.masked_fill(mask, 0)                      \
//   :172-225  one core op per neighbouring value frame, summed  > temporal_mix_kernel: one pass that
//             (here: the temporal mean of the frames, cast)    /  reads every value frame once
//   :164      sampling_offsets / normalizer                   \
//   :165      reference_points + offsets                        > prologue_fwd_kernel / prologue_bwd_kernel
//   :149      softmax over the L*P logits of a (query, head)  /
// All are HBM-bound: the kernels move each byte once, 16 B per lane where the layout allows.
#pragma once
#include "msda_common.cuh"

namespace snipper {

constexpr int kMixMaxFrames = 8;
struct MixMatrix { float w[kMixMaxFrames][kMixMaxFrames]; };   // [out frame][in frame]

template <typename T> struct Vec4IO;
template <> struct Vec4IO<float> {
  static __device__ __forceinline__ void ld(const float *p, float (&v)[4]) {
    const float4 t = *reinterpret_cast<const float4 *>(p); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  }
  static __device__ __forceinline__ void st(float *p, const float (&v)[4]) {
    *reinterpret_cast<float4 *>(p) = make_float4(v[0], v[1], v[2], v[3]);
  }
};
template <> struct Vec4IO<uint16_t> {
  static __device__ __forceinline__ void ld(const uint16_t *p, float (&v)[4]) {
    const uint2 t = *reinterpret_cast<const uint2 *>(p);
    v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
    v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
  }
  static __device__ __forceinline__ void st(uint16_t *p, const float (&v)[4]) {
    uint2 t;
    t.x = (unsigned)f32_to_bf16_bits(v[0]) | ((unsigned)f32_to_bf16_bits(v[1]) << 16);
    t.y = (unsigned)f32_to_bf16_bits(v[2]) | ((unsigned)f32_to_bf16_bits(v[3]) << 16);
    *reinterpret_cast<uint2 *>(p) = t;
  }
};

// W consecutive channels per lane: 4 (the types above) or 8 (two of them; bf16: ONE 16-byte access per lane)
template <typename T, int W> struct VecIO {
  static __device__ __forceinline__ void ld(const T *p, float (&v)[W]) {
    if constexpr (W == 4) {
      Vec4IO<T>::ld(p, v);
    } else if constexpr (sizeof(T) == 2) {
      const uint4 t = *reinterpret_cast<const uint4 *>(p);
      v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
      v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
      v[4] = __uint_as_float(t.z << 16); v[5] = __uint_as_float(t.z & 0xffff0000u);
      v[6] = __uint_as_float(t.w << 16); v[7] = __uint_as_float(t.w & 0xffff0000u);
    } else {
      const float4 a = reinterpret_cast<const float4 *>(p)[0], b = reinterpret_cast<const float4 *>(p)[1];
      v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    }
  }
  static __device__ __forceinline__ void st(T *p, const float (&v)[W]) {
    if constexpr (W == 4) {
      Vec4IO<T>::st(p, v);
    } else if constexpr (sizeof(T) == 2) {
      uint4 t;
      t.x = (unsigned)f32_to_bf16_bits(v[0]) | ((unsigned)f32_to_bf16_bits(v[1]) << 16);
      t.y = (unsigned)f32_to_bf16_bits(v[2]) | ((unsigned)f32_to_bf16_bits(v[3]) << 16);
      t.z = (unsigned)f32_to_bf16_bits(v[4]) | ((unsigned)f32_to_bf16_bits(v[5]) << 16);
      t.w = (unsigned)f32_to_bf16_bits(v[6]) | ((unsigned)f32_to_bf16_bits(v[7]) << 16);
      *reinterpret_cast<uint4 *>(p) = t;
    } else {
      reinterpret_cast<float4 *>(p)[0] = make_float4(v[0], v[1], v[2], v[3]);
      reinterpret_cast<float4 *>(p)[1] = make_float4(v[4], v[5], v[6], v[7]);
    }
  }
};

// out[n, to, s, c] = sum_ti mix[to][ti] * in[n, ti, s, c], masked positions (mask[n, t, s] != 0) reading as 0
// (MASK_IN: the mask belongs to the input frames -- forward) or being written as 0 (the mask belongs to
// the output frames -- backward, where `in` is the gradient of the mixed frames).
// F = frames the instantiation holds in registers (4 or 8 >= Ti, To), W = channels per lane (8 when C % 8 == 0: a bf16
// frame row then moves as 16 bytes per lane; measured 47 -> see DESIGN 3.5).  One position per thread: the grid covers
// N * S * C / W exactly (< 2^31), so the index arithmetic is 32-bit.
// HM: one side (or both) is HEAD-MAJOR, [frames][M][S][Dh] instead of [frames][S][M * Dh] (snipper_msda_config.value_layout):
// the threads are then dealt in head-major order (n, head, position, vector of the head row), so that side moves as
// contiguous runs and the other one as whole head rows (Dh elements: 96 or 192 bytes).
template <typename TI, typename TO, bool MASK_IN, int F, int W, bool HM = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void temporal_mix_kernel(
    const TI *__restrict__ in, const unsigned char *__restrict__ mask, MixMatrix mix, int N, int Ti, int To, int S, int C,
    TO *__restrict__ out, int Dh = 0, int in_hm = 0, int out_hm = 0) {
  const unsigned XW = (unsigned)S * (unsigned)C / W;    // vectors of W channels per frame
  const unsigned i = blockIdx.x * 256u + threadIdx.x;
  if (i >= (unsigned)N * XW) return;
  const unsigned n = i / XW;
  unsigned xw = i - n * XW, s, xw_hm = 0;
  if constexpr (HM) {
    const unsigned DV = (unsigned)Dh / W, per_head = (unsigned)S * DV;
    const unsigned m = xw / per_head, r = xw - m * per_head;
    s = r / DV;
    xw_hm = xw;                                                        // ((m S + s) Dh + dv W) / W
    xw = s * ((unsigned)C / W) + m * DV + (r - s * DV);                // ((s M + m) Dh + dv W) / W
  } else {
    s = xw / ((unsigned)C / W);
  }
  const unsigned xw_in = (HM && in_hm) ? xw_hm : xw, xw_out = (HM && out_hm) ? xw_hm : xw;
  float v[F][W];
#pragma unroll
  for (int ti = 0; ti < F; ++ti) {
    if (ti < Ti) {
      VecIO<TI, W>::ld(in + ((size_t)(n * Ti + ti) * XW + xw_in) * W, v[ti]);
      if (MASK_IN && mask && mask[(size_t)(n * Ti + ti) * S + s]) {
#pragma unroll
        for (int k = 0; k < W; ++k) v[ti][k] = 0.f;
      }
    }
  }
#pragma unroll
  for (int to = 0; to < F; ++to) {
    if (to < To) {
      float o[W];
#pragma unroll
      for (int k = 0; k < W; ++k) o[k] = 0.f;
#pragma unroll
      for (int ti = 0; ti < F; ++ti) {
        if (ti < Ti) {
          const float w = mix.w[to][ti];
#pragma unroll
          for (int k = 0; k < W; ++k) o[k] = fmaf(w, v[ti][k], o[k]);
        }
      }
      if (!MASK_IN && mask && mask[(size_t)(n * To + to) * S + s]) {
#pragma unroll
        for (int k = 0; k < W; ++k) o[k] = 0.f;
      }
      VecIO<TO, W>::st(out + ((size_t)(n * To + to) * XW + xw_out) * W, o);
    }
  }
}

// ---- sampling locations + attention probabilities ---------------------------------------------------
constexpr int kPrologueMaxLP = 16, kPrologueMaxL = 8;
struct LevelScale { float inv_w[kPrologueMaxL], inv_h[kPrologueMaxL]; };

template <typename T> __device__ __forceinline__ float ld_scalar(const T *p);
template <> __device__ __forceinline__ float ld_scalar<float>(const float *p) { return *p; }
template <> __device__ __forceinline__ float ld_scalar<uint16_t>(const uint16_t *p) { return bf16_bits_to_f32(*p); }
template <typename T> __device__ __forceinline__ void st_scalar(T *p, float v);
template <> __device__ __forceinline__ void st_scalar<float>(float *p, float v) { *p = v; }
template <> __device__ __forceinline__ void st_scalar<uint16_t>(uint16_t *p, float v) { *p = f32_to_bf16_bits(v); }

// One thread per row (n, t, q, m):  loc = ref + off * (1/W_l, 1/H_l),  prob = softmax(logits).
// off / logit are addressed per QUERY with leading dimensions (elements): row (q, m) reads off + q * off_ld + m * 2LP
// and logit + q * logit_ld + m * LP, so both may be column slices of one merged projection output.  VEC: 4-element
// vector accesses (LP % 4 == 0 and 4-element aligned slices); otherwise scalar.
// CL / CP > 0 fix L and P at compile time (every loop bound and level index folds: registers only, no scratch);
// CL = 0 takes them from the arguments.
template <typename TI, bool VEC, int CL, int CP>
__global__ __launch_bounds__(256) void prologue_fwd_kernel(const TI *__restrict__ off, long long off_ld,
                                                           const TI *__restrict__ logit, long long logit_ld,
                                                           const float *__restrict__ ref, LevelScale sc,
                                                           long long rows, int M, int L_rt, int P_rt,
                                                           float *__restrict__ loc, float *__restrict__ prob,
                                                           const float *__restrict__ off_bias) {
  // off_bias [M * L * P * 2] float32 or nullptr: added to the offsets HERE, in float32.  The offsets arrive as the bf16
  // output of a projection under autocast; their bias -- the reference initialisation's grid of 1 .. P pixels along the head's
  // direction (ms_deform_attn.py:82-90) -- is the large part of them, and rounding (W q + b) to bf16 moves a sampling
  // location by up to 2^-9 of 4 px.  The gradient with respect to a location is piecewise constant per pixel cell
  // (ms_deform_im2col_cuda.cuh:87-159), so that displacement puts a few per cent of the samples into the neighbouring cell:
  // 6-9 % relative error in the offset projection's gradients against 1 % when the projection's GEMM carries no bias and the
  // bias is added here (tests/test_timed_path_gpu.py).
  const long long row = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= rows) return;
  const int L = CL > 0 ? CL : L_rt, P = CL > 0 ? CP : P_rt;
  const int LP = L * P;
  const long long q = row / M;
  const int m = (int)(row - q * M);
  const TI *o = off + q * off_ld + (long long)m * LP * 2;
  const TI *g = logit + q * logit_ld + (long long)m * LP;
  const float *r = ref + q * L * 2;
  float *lo = loc + row * LP * 2;
  float *pr = prob + row * LP;
  float z[kPrologueMaxLP], ov[2 * kPrologueMaxLP];
  if (VEC) {
#pragma unroll
    for (int i = 0; i < kPrologueMaxLP / 4; ++i)
      if (i * 4 < LP) { float t[4]; Vec4IO<TI>::ld(g + i * 4, t); z[4 * i] = t[0]; z[4 * i + 1] = t[1]; z[4 * i + 2] = t[2]; z[4 * i + 3] = t[3]; }
#pragma unroll
    for (int i = 0; i < kPrologueMaxLP / 2; ++i)
      if (i * 4 < 2 * LP) { float t[4]; Vec4IO<TI>::ld(o + i * 4, t); ov[4 * i] = t[0]; ov[4 * i + 1] = t[1]; ov[4 * i + 2] = t[2]; ov[4 * i + 3] = t[3]; }
  } else {
#pragma unroll
    for (int i = 0; i < kPrologueMaxLP; ++i)
      if (i < LP) { z[i] = ld_scalar<TI>(g + i); ov[2 * i] = ld_scalar<TI>(o + 2 * i); ov[2 * i + 1] = ld_scalar<TI>(o + 2 * i + 1); }
  }
  if (off_bias) {
    const float *ob = off_bias + (long long)m * LP * 2;
#pragma unroll
    for (int i = 0; i < 2 * kPrologueMaxLP; ++i)
      if (i < 2 * LP) ov[i] += ob[i];
  }
  float zmax = -INFINITY;
#pragma unroll
  for (int i = 0; i < kPrologueMaxLP; ++i)
    if (i < LP) zmax = fmaxf(zmax, z[i]);
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < kPrologueMaxLP; ++i) {
    if (i < LP) {
      z[i] = __expf(z[i] - zmax);
      sum += z[i];
    }
  }
  const float inv = 1.f / sum;
#pragma unroll
  for (int i = 0; i < kPrologueMaxLP; ++i) {
    if (i < LP) {
      const int l = i / P;
      z[i] *= inv;
      ov[2 * i] = fmaf(ov[2 * i], sc.inv_w[l], r[2 * l]);
      ov[2 * i + 1] = fmaf(ov[2 * i + 1], sc.inv_h[l], r[2 * l + 1]);
    }
  }
  if (VEC) {
#pragma unroll
    for (int i = 0; i < kPrologueMaxLP / 4; ++i)
      if (i * 4 < LP) *reinterpret_cast<float4 *>(pr + 4 * i) = make_float4(z[4 * i], z[4 * i + 1], z[4 * i + 2], z[4 * i + 3]);
#pragma unroll
    for (int i = 0; i < kPrologueMaxLP / 2; ++i)
      if (i * 4 < 2 * LP) *reinterpret_cast<float4 *>(lo + 4 * i) = make_float4(ov[4 * i], ov[4 * i + 1], ov[4 * i + 2], ov[4 * i + 3]);
  } else {
#pragma unroll
    for (int i = 0; i < kPrologueMaxLP; ++i)
      if (i < LP) { pr[i] = z[i]; lo[2 * i] = ov[2 * i]; lo[2 * i + 1] = ov[2 * i + 1]; }
  }
}

// grad_off = grad_loc * (1/W, 1/H);  grad_logit = prob * (grad_prob - <prob, grad_prob>);
// grad_ref[n,t,q,l,:] = sum over heads and points of grad_loc   (M a power of two <= 64, rows of a query adjacent).
// goff / glogit use the same per-query leading dimensions as the forward's inputs.
template <typename TI, bool VEC, int CL, int CP>
__global__ __launch_bounds__(256) void prologue_bwd_kernel(const float *__restrict__ gloc, const float *__restrict__ gprob,
                                                           const float *__restrict__ prob, LevelScale sc,
                                                           long long rows, int M, int L_rt, int P_rt,
                                                           TI *__restrict__ goff, long long goff_ld,
                                                           TI *__restrict__ glogit, long long glogit_ld,
                                                           float *__restrict__ gref /* or nullptr */) {
  const long long row = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = row < rows;
  const long long rr = live ? row : 0;
  const int L = CL > 0 ? CL : L_rt, P = CL > 0 ? CP : P_rt;
  const int LP = L * P;
  const long long q = rr / M;
  const int m = (int)(rr - q * M);
  const float *gl = gloc + rr * LP * 2, *gp = gprob + rr * LP, *pr = prob + rr * LP;
  TI *go = goff + q * goff_ld + (long long)m * LP * 2;
  TI *gg = glogit + q * glogit_ld + (long long)m * LP;
  float p[kPrologueMaxLP], g[kPrologueMaxLP], gx[2 * kPrologueMaxLP];
  if (VEC) {
#pragma unroll
    for (int i = 0; i < kPrologueMaxLP / 4; ++i) {
      if (i * 4 < LP) {
        const float4 a = *reinterpret_cast<const float4 *>(pr + 4 * i), b = *reinterpret_cast<const float4 *>(gp + 4 * i);
        p[4 * i] = a.x; p[4 * i + 1] = a.y; p[4 * i + 2] = a.z; p[4 * i + 3] = a.w;
        g[4 * i] = b.x; g[4 * i + 1] = b.y; g[4 * i + 2] = b.z; g[4 * i + 3] = b.w;
      }
    }
#pragma unroll
    for (int i = 0; i < kPrologueMaxLP / 2; ++i) {
      if (i * 4 < 2 * LP) {
        const float4 a = *reinterpret_cast<const float4 *>(gl + 4 * i);
        gx[4 * i] = a.x; gx[4 * i + 1] = a.y; gx[4 * i + 2] = a.z; gx[4 * i + 3] = a.w;
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < kPrologueMaxLP; ++i)
      if (i < LP) { p[i] = pr[i]; g[i] = gp[i]; gx[2 * i] = gl[2 * i]; gx[2 * i + 1] = gl[2 * i + 1]; }
  }
  float dotp = 0.f;
#pragma unroll
  for (int i = 0; i < kPrologueMaxLP; ++i)
    if (i < LP) dotp = fmaf(p[i], g[i], dotp);
  float rx[kPrologueMaxL], ry[kPrologueMaxL];
#pragma unroll
  for (int l = 0; l < kPrologueMaxL; ++l) rx[l] = ry[l] = 0.f;
#pragma unroll
  for (int i = 0; i < kPrologueMaxLP; ++i) {
    if (i < LP) {
      const int l = i / P;
#pragma unroll
      for (int ll = 0; ll < kPrologueMaxL; ++ll) {
        if (ll == l) { rx[ll] += live ? gx[2 * i] : 0.f; ry[ll] += live ? gx[2 * i + 1] : 0.f; }
      }
      g[i] = p[i] * (g[i] - dotp);
      gx[2 * i] *= sc.inv_w[l];
      gx[2 * i + 1] *= sc.inv_h[l];
    }
  }
  if (live) {
    if (VEC) {
#pragma unroll
      for (int i = 0; i < kPrologueMaxLP / 4; ++i)
        if (i * 4 < LP) { const float t[4] = {g[4 * i], g[4 * i + 1], g[4 * i + 2], g[4 * i + 3]}; Vec4IO<TI>::st(gg + 4 * i, t); }
#pragma unroll
      for (int i = 0; i < kPrologueMaxLP / 2; ++i)
        if (i * 4 < 2 * LP) { const float t[4] = {gx[4 * i], gx[4 * i + 1], gx[4 * i + 2], gx[4 * i + 3]}; Vec4IO<TI>::st(go + 4 * i, t); }
    } else {
#pragma unroll
      for (int i = 0; i < kPrologueMaxLP; ++i)
        if (i < LP) { st_scalar<TI>(gg + i, g[i]); st_scalar<TI>(go + 2 * i, gx[2 * i]); st_scalar<TI>(go + 2 * i + 1, gx[2 * i + 1]); }
    }
  }
  if (gref) {
#pragma unroll
    for (int l = 0; l < kPrologueMaxL; ++l) {
      if (l < L) {
        float ax = rx[l], ay = ry[l];
        for (int o = M >> 1; o > 0; o >>= 1) {          // the M heads of a query sit in M adjacent lanes
          ax += __shfl_xor(ax, o, 64);
          ay += __shfl_xor(ay, o, 64);
        }
        if (live && m == 0) {
          gref[q * L * 2 + 2 * l] = ax;
          gref[q * L * 2 + 2 * l + 1] = ay;
        }
      }
    }
  }
}

}  // namespace snipper
