// match_cost.cuh -- the Hungarian matcher's cost matrix in one launch (gfx950).
//
// Reference: /root/reference/models/matcher.py:60-127 builds, per sample, seven [n_queries, n_targets] terms from
// broadcast [n, m, T, K, c] tensors (~60 element-wise / reduction launches per sample and call; the build's criterion
// evaluates all decoder layers at once, so per sample and step).  Every entry of the matrix is a function of ONE
// prediction (layer l, query q) and ONE target m:
//
//   class : -sum_t prob[t] vis[t] / (sum_t vis[t] + eps),  prob = softmax(logits)[..., 1],  vis[t] = any joint visible
//   joint : sum_{t,k>=1} v |o_root.xy + o_k.xy - t_k.xy|_1 / (sum v + eps)        v = target joint visibility
//   jvis  : mean_{t,k>=1} (o_k.vis - v)^2
//   jdep  : sum_{t,k>=1} e |o_root.d + o_k.d / max_depth - t_k.d| / (sum e + eps)  e = target depth-exists flag
//   root / rvis / rdep: the same three for k = 0 (no displacement)
//
// so one THREAD evaluates one entry with a loop over the T x K keypoints (60 at T = 4): no intermediate tensor, no
// reduction tree.  The matrix is tiny (6 layers x 60 queries x <= 20 targets); the point is the ~120 launches it
// replaces, not bandwidth.  Sums run in (t, k) order in float32 -- the reference's reduction order is PyTorch's, so
// entries agree to rounding (tests: 1e-5 relative), and the assignment is compared on the same matrix.
#pragma once
#include <hip/hip_runtime.h>

namespace snipper {

struct MatchCostArgs {
  const float *kpts; long long kp_sl, kp_sq;      // [L][.][Q][T][K][3]: strides (elements) of the layer / query axes
  int kp_sk;                                      //   and of the keypoint axis (3, or 4 for a slice of [..., K, 4])
  const float *depth; long long d_sl, d_sq;       // [L][.][Q][T][K][1]
  int d_sk;
  const float *logits; long long lg_sl, lg_sq;    // [L][.][Q][T][2]
  const float *tk;                                // [M][T][K][3] target keypoints (x, y, visibility)
  const float *td;                                // [M][T][K][2] target depth (value, exists)
  const float *max_depth;                         // device scalar
  int L, Q, M, T, K;
  float w_class, w_root, w_root_vis, w_root_depth, w_joint, w_joint_vis, w_joint_depth, eps;
  float *out;                                     // [L][Q][M]
};

__global__ __launch_bounds__(256) void match_cost_kernel(MatchCostArgs a) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long total = (long long)a.L * a.Q * a.M;
  if (idx >= total) return;
  const int m = (int)(idx % a.M);
  const int q = (int)((idx / a.M) % a.Q);
  const int l = (int)(idx / ((long long)a.M * a.Q));
  const float *ok = a.kpts + l * a.kp_sl + q * a.kp_sq;
  const float *od = a.depth + l * a.d_sl + q * a.d_sq;
  const float *lg = a.logits + l * a.lg_sl + q * a.lg_sq;
  const float *tk = a.tk + (long long)m * a.T * a.K * 3;
  const float *td = a.td + (long long)m * a.T * a.K * 2;
  const float md = *a.max_depth;

  float cls_num = 0.f, cls_den = 0.f;
  float j_num = 0.f, j_den = 0.f, jv = 0.f, jd_num = 0.f, jd_den = 0.f;
  float r_num = 0.f, r_den = 0.f, rv = 0.f, rd_num = 0.f, rd_den = 0.f;
  for (int t = 0; t < a.T; ++t) {
    const float *okt = ok + (long long)t * a.K * a.kp_sk, *odt = od + (long long)t * a.K * a.d_sk;
    const float *tkt = tk + (long long)t * a.K * 3, *tdt = td + (long long)t * a.K * 2;
    // root (k = 0)
    const float orx = okt[0], ory = okt[1], orv = okt[2], ord_ = odt[0];
    const float trv = tkt[2];
    r_num += fabsf(trv * (orx - tkt[0])) + fabsf(trv * (ory - tkt[1]));
    r_den += trv;
    rv += (orv - trv) * (orv - trv);
    rd_num += fabsf(tdt[1] * (ord_ - tdt[0]));
    rd_den += tdt[1];
    float vis_sum = 0.f;
    for (int k = 1; k < a.K; ++k) {
      const float v = tkt[3 * k + 2];
      const float *okk = okt + k * a.kp_sk;
      const float jx = okk[0] + orx, jy = okk[1] + ory;
      j_num += fabsf(v * (jx - tkt[3 * k])) + fabsf(v * (jy - tkt[3 * k + 1]));
      j_den += v;
      vis_sum += v;
      const float dv = okk[2] - v;
      jv += dv * dv;
      const float e = tdt[2 * k + 1];
      jd_num += fabsf(e * ((ord_ + odt[k * a.d_sk] / md) - tdt[2 * k]));
      jd_den += e;
    }
    // softmax over the two classes, probability of class 1
    const float l0 = lg[2 * t], l1 = lg[2 * t + 1], mx = fmaxf(l0, l1);
    const float e0 = expf(l0 - mx), e1 = expf(l1 - mx);
    const float vis = vis_sum > 0.f ? 1.f : 0.f;
    cls_num += (e1 / (e0 + e1)) * vis;
    cls_den += vis;
  }
  const float n_joint = (float)(a.T * (a.K - 1)), n_root = (float)a.T;
  const float cost = a.w_class * (-cls_num / (cls_den + a.eps)) + a.w_root * (r_num / (r_den + a.eps)) +
                     a.w_root_vis * (rv / n_root) + a.w_root_depth * (rd_num / (rd_den + a.eps)) +
                     a.w_joint * (j_num / (j_den + a.eps)) + a.w_joint_vis * (jv / n_joint) +
                     a.w_joint_depth * (jd_num / (jd_den + a.eps));
  a.out[idx] = cost;
}

// ---- iterative refinement of the decoder's reference points (reference models/deformable_transformer.py:329-333 with
// util/misc.py:481-485): new_ref = sigmoid(delta + inverse_sigmoid(ref)), detached, and the next layer's
// ref_in[row][l] = new_ref[row] * valid_ratios[batch(row)][l] (:319-321).  Eight element-wise launches on a few hundred
// values per decoder layer in PyTorch; one here.  No gradient flows through it (the reference detaches the result).
__global__ __launch_bounds__(256) void refine_reference_kernel(const float *__restrict__ delta, long long ld_delta,
                                                               const float *__restrict__ ref,
                                                               const float *__restrict__ valid_ratios, int rows,
                                                               int rows_per_batch, int L, float eps,
                                                               float *__restrict__ new_ref, float *__restrict__ ref_in) {
  const int i = blockIdx.x * 256 + threadIdx.x;          // one thread per (row, coordinate)
  if (i >= rows * 2) return;
  const int row = i >> 1, c = i & 1;
  float x = fminf(fmaxf(ref[i], 0.f), 1.f);
  const float x1 = fmaxf(x, eps), x2 = fmaxf(1.f - x, eps);
  const float z = delta[(long long)row * ld_delta + c] + logf(x1 / x2);
  const float r = 1.f / (1.f + expf(-z));
  new_ref[i] = r;
  const float *vr = valid_ratios + (long long)(row / rows_per_batch) * L * 2 + c;
  for (int l = 0; l < L; ++l) ref_in[((long long)row * L + l) * 2 + c] = r * vr[2 * l];
}

}  // namespace snipper
