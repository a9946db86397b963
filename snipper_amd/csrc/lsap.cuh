// lsap.cuh -- linear sum assignment on the device, one wavefront per problem.
//
// The reference solves one 60 x m assignment per sample and decoder layer with SciPy on the HOST:
// models/matcher.py:132 `linear_sum_assignment(cost.cpu())`, i.e. 12 device-to-host synchronisations per training
// step.  The problems are tiny (n_query = 60 <= 64 predictions, m <= a few dozen persons), so one wave solves one
// problem with the shortest-augmenting-path (Hungarian) algorithm in O(m^2) wave steps: lane j owns column
// (prediction) j -- its dual v[j], its slack minv[j], its back pointer way[j] and the row p[j] assigned to it;
// the row duals u[] sit in LDS; the arg-min over the unvisited columns is a wave reduction.  Arithmetic is
// float64 like SciPy's (the costs are float32, so they convert exactly): the optimum of a tie-free problem is
// unique, so the pairs are SciPy's.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace snipper {

constexpr int kLsapMaxCols = 64, kLsapMaxRows = 64;

// cost [P][n][m] (prediction-major, as the matcher builds it), m <= n <= 64.
// out_src / out_tgt [P][m]: the m matched (prediction, target) pairs, predictions ascending.
__global__ __launch_bounds__(64) void lsap_kernel(const float *__restrict__ cost, int n, int m,
                                                  long long *__restrict__ out_src, long long *__restrict__ out_tgt) {
  __shared__ double u[kLsapMaxRows];
  const int lane = threadIdx.x;
  const float *C = cost + (size_t)blockIdx.x * n * m;
  const bool real = lane < n;
  if (lane < m) u[lane] = 0.0;
  double v = 0.0;
  int p = -1;                               // row assigned to my column
  __syncthreads();
  const double INF = 1e300;
  for (int i = 0; i < m; ++i) {
    double minv = INF;
    bool used = !real;
    int way = -2;
    int j0 = -1, i0 = i;                    // j0 = -1: the virtual column that holds row i
    double u_virtual_add = 0.0;
    while (true) {
      if (lane == j0) used = true;
      const double ui0 = u[i0];
      if (!used) {
        const double cur = (double)C[(size_t)lane * m + i0] - ui0 - v;
        if (cur < minv) { minv = cur; way = j0; }
      }
      // arg-min of minv over the unvisited columns (lowest lane wins ties)
      double best = used ? INF : minv;
      int bj = lane;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const double ob = __shfl_xor(best, o, 64);
        const int oj = __shfl_xor(bj, o, 64);
        if (ob < best || (ob == best && oj < bj)) { best = ob; bj = oj; }
      }
      const double delta = best;
      // dual update: visited columns move their row's u and their own v; the virtual column moves u[i]
      if (used && real && p >= 0) u[p] += delta;          // distinct rows: distinct LDS words
      if (lane == 0) u[i] += delta;                        // row i sits in the virtual column (always visited)
      if (used && real) v -= delta; else if (real) minv -= delta;
      __syncthreads();
      j0 = bj;
      const int pj0 = __shfl(p, j0, 64);
      if (pj0 < 0) break;                                  // reached a free column: augment
      i0 = pj0;
    }
    (void)u_virtual_add;
    // augment along the back pointers: column j takes the row of column way[j], the first one takes row i
    int j = j0;
    while (j >= 0) {
      const int jprev = __shfl(way, j, 64);
      const int newp = jprev < 0 ? i : __shfl(p, jprev < 0 ? 0 : jprev, 64);
      if (lane == j) p = newp;
      j = jprev;
    }
    __syncthreads();
  }
  // compact the matched columns in ascending order
  const unsigned long long mask = __ballot(real && p >= 0);
  if (real && p >= 0) {
    const int pos = __popcll(mask & ((1ull << lane) - 1ull));
    out_src[(size_t)blockIdx.x * m + pos] = lane;
    out_tgt[(size_t)blockIdx.x * m + pos] = p;
  }
}

}  // namespace snipper
