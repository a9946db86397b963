// pair_losses.cuh -- the nine keypoint / depth / continuity loss terms of SetCriterion for all matched
// (prediction, target) pairs of all decoder layers in ONE launch each way (gfx950).
//
// Reference: models/model.py:289-427 (loss_root, loss_joint, loss_joint_disp, loss_joint_cont) -- there a dozen masked
// L1 / L2 reductions over [T, K] keypoints per pair, written as ~100 element-wise / reduction launches forward and
// ~200 backward on tensors of a few thousand elements (launch-bound: 2.6 ms of host time per step).  One wave handles
// one pair: its T x K keypoints sit in LDS, lane e = t * K + k owns keypoint (t, k), the per-pair normalisers are
// wave reductions.
//
//   sk [P, T, K, 3]  predicted (x, y, vis)          tk [P, T, K, 3]  target (x, y, vis)
//   sd [P, T, K, 1]  predicted depth                td [P, T, K, 2]  target (depth, valid)        P = n_layers * pairs
//   out [P, 9]:  0 root  1 root_depth  2 root_vis  3 joint_disp  4 joint_depth_disp  5 joint  6 joint_depth
//                7 joint_vis  8 cont           (per pair; the caller sums over pairs and divides by num_traj)
//
// Term definitions (k = 0 is the root joint, v = target visibility, ok = target depth validity):
//   root             sum_{t,c} v(t,0) |S(t,0,c) - G(t,0,c)| / (sum_t v(t,0) + eps)                        c in {x, y}
//   root_depth       sum_t ok(t,0) |Gd(t,0) - D(t,0)| / (sum_t ok(t,0) + eps)
//   root_vis         mean_t (S(t,0,vis) - v(t,0))^2
//   joint_disp       sum_{t,k>0,c} v(t,k) v(t,0) |S(t,k,c) - (G(t,k,c) - G(t,0,c))| / (sum v(t,k) v(t,0) + eps)
//   joint_depth_disp sum_{t,k>0} ok(t,k) ok(t,0) |D(t,k) - (Gd(t,k) - Gd(t,0))| / (sum ok(t,k) ok(t,0) + eps)
//   joint            sum_{t,k>0,c} v(t,k) |S(t,k,c) + S(t,0,c) - G(t,k,c)| / (sum_{t,k>0} v(t,k) + eps)
//   joint_depth      sum_{t,k>0} ok(t,k) |D(t,0) + D(t,k)/md - Gd(t,k)| / (sum_{t,k>0} ok(t,k) + eps)
//   joint_vis        mean_{t,k>0} (S(t,k,vis) - v(t,k))^2
//   cont             sum_{c in {x,y,d}} sum_{t<T-1,k} w(k) v(t+1,k) v(t,k) (kp(t+1,k,c) - kp(t,k,c))^2 / (sum v v + eps)
//                    kp(t,0,.) = (S(t,0,x), S(t,0,y), D(t,0));  kp(t,k>0,xy) = S(t,k,xy) - stopgrad(S(t,0,xy));
//                    kp(t,k>0,d) = D(t,0) + D(t,k)/md - stopgrad(D(t,0))
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace snipper {

constexpr int kPairMaxTK = 128;      // T * K keypoints per pair (two per lane)
constexpr int kPairTerms = 9;

struct PairLossArgs {
  const float *sk, *sd, *tk, *td;     // see above
  const float *cont_w;                // [K]
  const float *max_depth;             // device scalar
  float *out;                         // [P, 9]                      (forward)
  const float *gw;                    // [P, 9] dL/d(per-pair term)                                   (backward)
  float *dsk, *dsd;                   // [P, T, K, 3], [P, T, K, 1]  (backward)
  int P, T, K, pairs_per_layer;
  float eps;
};

__device__ __forceinline__ float pair_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float pair_sign(float r) { return (float)(r > 0.f) - (float)(r < 0.f); }

struct PairTile {          // one pair's data in LDS
  float S[kPairMaxTK][3], D[kPairMaxTK], G[kPairMaxTK][3], Gd[kPairMaxTK][2];
};

__device__ __forceinline__ void pair_load(const PairLossArgs &a, int p, int lane, PairTile &s) {
  const int n = a.T * a.K;
  for (int e = lane; e < n; e += 64) {
    const long long o = (long long)p * n + e;
    s.S[e][0] = a.sk[o * 3]; s.S[e][1] = a.sk[o * 3 + 1]; s.S[e][2] = a.sk[o * 3 + 2];
    s.D[e] = a.sd[o];
    s.G[e][0] = a.tk[o * 3]; s.G[e][1] = a.tk[o * 3 + 1]; s.G[e][2] = a.tk[o * 3 + 2];
    s.Gd[e][0] = a.td[o * 2]; s.Gd[e][1] = a.td[o * 2 + 1];
  }
}

// the normalisers shared by forward and backward: 0 sum v_root, 1 sum ok_root, 2 sum v_k v_root, 3 sum ok_k ok_root,
// 4 sum v_k (k>0), 5 sum ok_k (k>0), 6 sum v(t+1) v(t)
__device__ __forceinline__ void pair_norms(const PairLossArgs &a, const PairTile &s, int lane, float (&den)[7]) {
  const int n = a.T * a.K;
  float d[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int e = lane; e < n; e += 64) {
    const int t = e / a.K, k = e - t * a.K, r = t * a.K;
    const float v = s.G[e][2], ok = s.Gd[e][1], vr = s.G[r][2], okr = s.Gd[r][1];
    if (k == 0) { d[0] += v; d[1] += ok; }
    else { d[2] += v * vr; d[3] += ok * okr; d[4] += v; d[5] += ok; }
    if (t + 1 < a.T) d[6] += s.G[e + a.K][2] * v;
  }
#pragma unroll
  for (int i = 0; i < 7; ++i) den[i] = pair_wave_sum(d[i]);
}

// kp(t, k, c) of the continuity term (values only)
__device__ __forceinline__ float pair_kp(const PairLossArgs &a, const PairTile &s, int e, int c, float inv_md) {
  const int t = e / a.K, k = e - t * a.K, r = t * a.K;
  if (c < 2) return k == 0 ? s.S[e][c] : s.S[e][c] - s.S[r][c];
  return k == 0 ? s.D[e] : s.D[e] * inv_md;           // D(t,0) + D(t,k)/md - D(t,0)
}

__global__ __launch_bounds__(256) void pair_losses_fwd_kernel(PairLossArgs a) {
  __shared__ PairTile tiles[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int p = blockIdx.x * 4 + wave;
  if (p >= a.P) return;
  PairTile &s = tiles[wave];
  pair_load(a, p, lane, s);
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0);          // the wave's own LDS writes are visible to all its lanes
  float den[7];
  pair_norms(a, s, lane, den);
  const float inv_md = 1.f / a.max_depth[0];
  const int n = a.T * a.K;
  float acc[kPairTerms] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int e = lane; e < n; e += 64) {
    const int t = e / a.K, k = e - t * a.K, r = t * a.K;
    const float v = s.G[e][2], ok = s.Gd[e][1], vr = s.G[r][2], okr = s.Gd[r][1];
    if (k == 0) {
      acc[0] += v * (fabsf(s.S[e][0] - s.G[e][0]) + fabsf(s.S[e][1] - s.G[e][1]));
      acc[1] += ok * fabsf(s.Gd[e][0] - s.D[e]);
      const float dv = s.S[e][2] - v;
      acc[2] += dv * dv;
    } else {
      const float vv = v * vr;
      acc[3] += vv * (fabsf(s.S[e][0] - (s.G[e][0] - s.G[r][0])) + fabsf(s.S[e][1] - (s.G[e][1] - s.G[r][1])));
      acc[4] += ok * okr * fabsf(s.D[e] - (s.Gd[e][0] - s.Gd[r][0]));
      acc[5] += v * (fabsf(s.S[e][0] + s.S[r][0] - s.G[e][0]) + fabsf(s.S[e][1] + s.S[r][1] - s.G[e][1]));
      acc[6] += ok * fabsf(s.D[r] + s.D[e] * inv_md - s.Gd[e][0]);
      const float dv = s.S[e][2] - v;
      acc[7] += dv * dv;
    }
    if (t + 1 < a.T) {
      const float cv = a.cont_w[k] * s.G[e + a.K][2] * v;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float df = pair_kp(a, s, e + a.K, c, inv_md) - pair_kp(a, s, e, c, inv_md);
        acc[8] += cv * df * df;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < kPairTerms; ++i) acc[i] = pair_wave_sum(acc[i]);
  if (lane == 0) {
    float *o = a.out + (long long)p * kPairTerms;
    o[0] = acc[0] / (den[0] + a.eps);
    o[1] = acc[1] / (den[1] + a.eps);
    o[2] = acc[2] / (float)a.T;
    o[3] = acc[3] / (den[2] + a.eps);
    o[4] = acc[4] / (den[3] + a.eps);
    o[5] = acc[5] / (den[4] + a.eps);
    o[6] = acc[6] / (den[5] + a.eps);
    o[7] = a.K > 1 ? acc[7] / (float)(a.T * (a.K - 1)) : 0.f;
    o[8] = acc[8] / (den[6] + a.eps);
  }
}

// Gradients w.r.t. sk and sd.  Every lane first writes the direct contributions of ITS keypoint into an LDS gradient
// tile; the contributions that land on the root joint of the same frame (joint, joint_depth, cont's depth term) and on
// the neighbouring frame (cont) are added with LDS float atomics (a handful per lane).
__global__ __launch_bounds__(256) void pair_losses_bwd_kernel(PairLossArgs a) {
  __shared__ PairTile tiles[4];
  __shared__ float gS[4][kPairMaxTK][3], gD[4][kPairMaxTK];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int p = blockIdx.x * 4 + wave;
  if (p >= a.P) return;
  PairTile &s = tiles[wave];
  float (*dS)[3] = gS[wave];
  float *dD = gD[wave];
  const int n = a.T * a.K;
  pair_load(a, p, lane, s);
  for (int e = lane; e < n; e += 64) { dS[e][0] = dS[e][1] = dS[e][2] = 0.f; dD[e] = 0.f; }
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0);
  float den[7];
  pair_norms(a, s, lane, den);
  const float md = a.max_depth[0], inv_md = 1.f / md;
  const float *w = a.gw + (long long)p * kPairTerms;
  const float c_root = w[0] / (den[0] + a.eps), c_rd = w[1] / (den[1] + a.eps), c_rv = w[2] * 2.f / (float)a.T;
  const float c_jd = w[3] / (den[2] + a.eps), c_jdd = w[4] / (den[3] + a.eps), c_j = w[5] / (den[4] + a.eps);
  const float c_jz = w[6] / (den[5] + a.eps);
  const float c_jv = a.K > 1 ? w[7] * 2.f / (float)(a.T * (a.K - 1)) : 0.f;
  const float c_ct = w[8] * 2.f / (den[6] + a.eps);
  for (int e = lane; e < n; e += 64) {
    const int t = e / a.K, k = e - t * a.K, r = t * a.K;
    const float v = s.G[e][2], ok = s.Gd[e][1], vr = s.G[r][2], okr = s.Gd[r][1];
    float gx = 0.f, gy = 0.f, gv = 0.f, gd = 0.f;
    if (k == 0) {
      gx += c_root * v * pair_sign(s.S[e][0] - s.G[e][0]);
      gy += c_root * v * pair_sign(s.S[e][1] - s.G[e][1]);
      gd += c_rd * ok * pair_sign(s.D[e] - s.Gd[e][0]);
      gv += c_rv * (s.S[e][2] - v);
    } else {
      const float vv = v * vr;
      gx += c_jd * vv * pair_sign(s.S[e][0] - (s.G[e][0] - s.G[r][0]));
      gy += c_jd * vv * pair_sign(s.S[e][1] - (s.G[e][1] - s.G[r][1]));
      gd += c_jdd * ok * okr * pair_sign(s.D[e] - (s.Gd[e][0] - s.Gd[r][0]));
      const float jx = c_j * v * pair_sign(s.S[e][0] + s.S[r][0] - s.G[e][0]);
      const float jy = c_j * v * pair_sign(s.S[e][1] + s.S[r][1] - s.G[e][1]);
      gx += jx; gy += jy;
      atomicAdd(&dS[r][0], jx);                                  // the root of the same frame is part of the sum
      atomicAdd(&dS[r][1], jy);
      const float jz = c_jz * ok * pair_sign(s.D[r] + s.D[e] * inv_md - s.Gd[e][0]);
      gd += jz * inv_md;
      atomicAdd(&dD[r], jz);
      gv += c_jv * (s.S[e][2] - v);
    }
    if (t + 1 < a.T) {      // the pair (t, t+1) of this keypoint: +q on frame t+1, -q on frame t
      const float cv = c_ct * a.cont_w[k] * s.G[e + a.K][2] * v;
      const float qx = cv * (pair_kp(a, s, e + a.K, 0, inv_md) - pair_kp(a, s, e, 0, inv_md));
      const float qy = cv * (pair_kp(a, s, e + a.K, 1, inv_md) - pair_kp(a, s, e, 1, inv_md));
      const float qz = cv * (pair_kp(a, s, e + a.K, 2, inv_md) - pair_kp(a, s, e, 2, inv_md));
      gx -= qx; gy -= qy;
      atomicAdd(&dS[e + a.K][0], qx);
      atomicAdd(&dS[e + a.K][1], qy);
      if (k == 0) {
        gd -= qz;
        atomicAdd(&dD[e + a.K], qz);
      } else {              // kp_d(t,k) = D(t,0) + D(t,k)/md - stopgrad(D(t,0)): gradient 1 to D(t,0), 1/md to D(t,k)
        gd -= qz * inv_md;
        atomicAdd(&dD[e + a.K], qz * inv_md);
        atomicAdd(&dD[r], -qz);
        atomicAdd(&dD[r + a.K], qz);
      }
    }
    atomicAdd(&dS[e][0], gx);
    atomicAdd(&dS[e][1], gy);
    atomicAdd(&dS[e][2], gv);
    atomicAdd(&dD[e], gd);
  }
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0);
  for (int e = lane; e < n; e += 64) {
    const long long o = (long long)p * n + e;
    a.dsk[o * 3] = dS[e][0]; a.dsk[o * 3 + 1] = dS[e][1]; a.dsk[o * 3 + 2] = dS[e][2];
    a.dsd[o] = dD[e];
  }
}

}  // namespace snipper
