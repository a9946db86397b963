// msda_generic.cuh -- any-(D, L, P), any-dtype forward / backward kernels.
//
// This is the correctness-first path: it covers what the reference's launcher covers
// (every channel count of models/ops/test.py:85-86, float and double) plus bf16 storage.
// The tuned D=48 kernels live in msda_d48.cuh; the dispatcher in msda_capi.hip picks.
//
// Mapping (wave64): a "row" is one (n, q, m) triple = one D-vector of the output.  G
// consecutive lanes (G in {4,16,64}, the smallest that covers D, so a group never
// straddles a wave) cooperate on a row: lane j owns channels j, j+G, ...  so the lanes of
// a group read one contiguous head-row of `value` per tap (coalesced).  The L*P sampling
// points of the row are decoded ONCE, spread over the group's lanes, and parked in LDS;
// afterwards every lane replays them from LDS (same-address reads broadcast).  In the
// reference every one of the D threads of a row re-decodes all points and re-reads
// loc/attn from memory (.cuh:255-291).
#pragma once
#include "msda_common.cuh"

namespace snipper {

constexpr int kGenericBlock = 256;

struct CoreDims {
  int N, S, M, D, L, Lq, P;
  // layout of value / grad_value: 0 = the reference's [N, S, M, D]; 1 = head-major [N, M, S, D] (snipper_msda_config
  // .value_layout, the tuned D = 48 / 24 kernels only): a head's rows of neighbouring pixels are contiguous, so the two
  // x-neighbour taps of a sample are 2 D contiguous elements
  int head_major;
};
// byte offset of row (n, pixel s, head m) = msda_row_base(n, m) + s * msda_px_stride(), rows of row_bytes bytes
__device__ __forceinline__ unsigned msda_px_stride(const CoreDims &d, unsigned row_bytes) {
  return d.head_major ? row_bytes : (unsigned)d.M * row_bytes;
}
__device__ __forceinline__ unsigned msda_row_base(const CoreDims &d, unsigned n, unsigned m, unsigned row_bytes) {
  return d.head_major ? (n * (unsigned)d.M + m) * (unsigned)d.S * row_bytes : n * (unsigned)d.S * (unsigned)d.M * row_bytes + m * row_bytes;
}

template <typename VT, typename CT, int G>
__global__ __launch_bounds__(kGenericBlock) void msda_fwd_generic_kernel(
    const VT *__restrict__ value, const int64_t *__restrict__ shapes,
    const int64_t *__restrict__ level_start, const CT *__restrict__ loc,
    const CT *__restrict__ attn, CoreDims d, VT *__restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  auto *pts = reinterpret_cast<SamplePoint<CT> *>(smem_raw);
  constexpr int kRows = kGenericBlock / G;
  const int LP = d.L * d.P;
  const int grp = threadIdx.x / G, lane = threadIdx.x % G;
  const long long total_rows = (long long)d.N * d.Lq * d.M;
  const int row_stride = d.M * d.D;

  for (long long row0 = (long long)blockIdx.x * kRows; row0 < total_rows;
       row0 += (long long)gridDim.x * kRows) {
    const long long row = row0 + grp;
    const bool live = row < total_rows;
    SamplePoint<CT> *mine = pts + (size_t)grp * LP;
    if (live) {
      for (int s = lane; s < LP; s += G) {
        const int l = s / d.P;
        const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
        const long long li = (row * LP + s);
        mine[s] = decode_sample<CT>(loc[2 * li], loc[2 * li + 1], attn[li], H, W,
                                    (int)level_start[l], row_stride);
      }
    }
    __syncthreads();
    if (live) {
      const int m = (int)(row % d.M);
      const long long n = row / ((long long)d.M * d.Lq);
      const VT *vb = value + (size_t)n * d.S * row_stride + (size_t)m * d.D;
      for (int c = lane; c < d.D; c += G) {
        CT acc = CT(0);
        for (int s = 0; s < LP; ++s) {
          const SamplePoint<CT> sp = mine[s];
          const CT hh = CT(1) - sp.lh, hw = CT(1) - sp.lw;
          const CT v0 = sp.pix[0] >= 0 ? Conv<VT, CT>::ld(vb + sp.pix[0] + c) : CT(0);
          const CT v1 = sp.pix[1] >= 0 ? Conv<VT, CT>::ld(vb + sp.pix[1] + c) : CT(0);
          const CT v2 = sp.pix[2] >= 0 ? Conv<VT, CT>::ld(vb + sp.pix[2] + c) : CT(0);
          const CT v3 = sp.pix[3] >= 0 ? Conv<VT, CT>::ld(vb + sp.pix[3] + c) : CT(0);
          const CT smp = hh * hw * v0 + hh * sp.lw * v1 + sp.lh * hw * v2 + sp.lh * sp.lw * v3;
          acc += smp * sp.a;
        }
        Conv<VT, CT>::st(out + (size_t)row * d.D + c, acc);
      }
    }
    __syncthreads();
  }
}

// GVT = storage type of grad_value (float for bf16 inputs: gradients accumulate in f32).
template <typename VT, typename CT, typename GVT, int G>
__global__ __launch_bounds__(kGenericBlock) void msda_bwd_generic_kernel(
    const VT *__restrict__ grad_out, const VT *__restrict__ value,
    const int64_t *__restrict__ shapes, const int64_t *__restrict__ level_start,
    const CT *__restrict__ loc, const CT *__restrict__ attn, CoreDims d,
    GVT *__restrict__ grad_value, CT *__restrict__ grad_loc, CT *__restrict__ grad_attn) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  auto *pts = reinterpret_cast<SamplePoint<CT> *>(smem_raw);
  constexpr int kRows = kGenericBlock / G;
  const int LP = d.L * d.P;
  const int grp = threadIdx.x / G, lane = threadIdx.x % G;
  const long long total_rows = (long long)d.N * d.Lq * d.M;
  const int row_stride = d.M * d.D;

  for (long long row0 = (long long)blockIdx.x * kRows; row0 < total_rows;
       row0 += (long long)gridDim.x * kRows) {
    const long long row = row0 + grp;
    const bool live = row < total_rows;
    SamplePoint<CT> *mine = pts + (size_t)grp * LP;
    if (live) {
      for (int s = lane; s < LP; s += G) {
        const int l = s / d.P;
        const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
        const long long li = (row * LP + s);
        mine[s] = decode_sample<CT>(loc[2 * li], loc[2 * li + 1], attn[li], H, W,
                                    (int)level_start[l], row_stride);
      }
    }
    __syncthreads();
    // Every lane of the block runs the sample loop (the group reduction needs all G lanes);
    // dead rows simply carry zeros.
    const int m = live ? (int)(row % d.M) : 0;
    const long long n = live ? row / ((long long)d.M * d.Lq) : 0;
    const size_t vbase = (size_t)n * d.S * row_stride + (size_t)m * d.D;
    const VT *vb = value + vbase;
    GVT *gvb = grad_value + vbase;
    const VT *gb = grad_out + (size_t)(live ? row : 0) * d.D;
    for (int s = 0; s < LP; ++s) {
      CT acc_a = CT(0), acc_x = CT(0), acc_y = CT(0);
      if (live) {
        const SamplePoint<CT> sp = mine[s];
        const CT hh = CT(1) - sp.lh, hw = CT(1) - sp.lw;
        const CT w0 = hh * hw, w1 = hh * sp.lw, w2 = sp.lh * hw, w3 = sp.lh * sp.lw;
        for (int c = lane; c < d.D; c += G) {
          const CT g = Conv<VT, CT>::ld(gb + c);
          const CT ga = g * sp.a;
          CT v0 = CT(0), v1 = CT(0), v2 = CT(0), v3 = CT(0);
          if (sp.pix[0] >= 0) {
            v0 = Conv<VT, CT>::ld(vb + sp.pix[0] + c);
            atomicAdd(gvb + sp.pix[0] + c, GVT(w0 * ga));
          }
          if (sp.pix[1] >= 0) {
            v1 = Conv<VT, CT>::ld(vb + sp.pix[1] + c);
            atomicAdd(gvb + sp.pix[1] + c, GVT(w1 * ga));
          }
          if (sp.pix[2] >= 0) {
            v2 = Conv<VT, CT>::ld(vb + sp.pix[2] + c);
            atomicAdd(gvb + sp.pix[2] + c, GVT(w2 * ga));
          }
          if (sp.pix[3] >= 0) {
            v3 = Conv<VT, CT>::ld(vb + sp.pix[3] + c);
            atomicAdd(gvb + sp.pix[3] + c, GVT(w3 * ga));
          }
          const CT dy = hw * (v2 - v0) + sp.lw * (v3 - v1);   // d sample / d y  (pixels)
          const CT dx = hh * (v1 - v0) + sp.lh * (v3 - v2);   // d sample / d x  (pixels)
          acc_a += g * (w0 * v0 + w1 * v1 + w2 * v2 + w3 * v3);
          acc_x += sp.Wf * dx * ga;
          acc_y += sp.Hf * dy * ga;
        }
      }
      acc_a = group_sum<G>(acc_a);
      acc_x = group_sum<G>(acc_x);
      acc_y = group_sum<G>(acc_y);
      if (live && lane == 0) {
        const long long li = row * LP + s;
        grad_attn[li] = acc_a;
        grad_loc[2 * li] = acc_x;
        grad_loc[2 * li + 1] = acc_y;
      }
    }
    __syncthreads();
  }
}

}  // namespace snipper
