// small_linear.cuh -- backward of y = x . W^T + b for a few hundred float32 rows, ONE launch (gfx950).
//
// The decoder's Linears (reference models/deformable_transformer.py:244-343: self-attention projections, the offset /
// weight / output projections of the cross attention, the feed-forward block, 60 queries x T frames x batch = a few
// hundred rows) each need three products in the backward pass
//     dX[M,K] = G[M,N] . W[N,K]        dW[N,K] = G^T . X[M,K]        db[N] = sum_m G[m,:]
// which PyTorch issues as three library GEMM launches of ~9 us each (the third one a [1,M] x [M,N] product) -- 36 such
// triples per training step, all latency-bound.  Here the three are tiles of one grid: workgroup b owns one 32 x 32 tile
// of dX or of dW; its four waves split the reduction axis (chunks of 32, interleaved), each accumulates the tile with
// v_mfma_f32_32x32x2_f32 (float32 in, float32 accumulate: bitwise an fmaf chain, so the result is float32-exact like
// the library's), and the four partial tiles meet in LDS in a fixed order.  db is the column sum of the G chunks that the
// dW tiles of the first K column stage anyway.  Everything is deterministic.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace snipper {

typedef __attribute__((ext_vector_type(16))) float sl_f32x16;

struct SmallLinearBwdArgs {
  const float *G; long long ldg;     // [M][N]   dL/dy
  const float *X; long long ldx;     // [M][K]   the layer's input
  const float *W; long long ldw;     // [N][K]   the weight
  float *dX; long long lddx;         // [M][K] or nullptr
  float *dW; long long lddw;         // [N][K] or nullptr
  float *db;                         // [N] or nullptr
  int M, N, K;
  int tiles_dx;                      // workgroups [0, tiles_dx): dX tiles; the rest: dW tiles
};

constexpr int kSlTile = 32, kSlStride = 33, kSlThreads = 256;

// a 32 x 32 float32 tile, rows r0.., columns c0.. of a row-major matrix with `rows` x `cols` valid elements -> LDS
// [32][33] (zeros outside); 64 lanes x 4 float4
__device__ __forceinline__ void sl_load_tile(const float *__restrict__ A, long long ld, int rows, int cols, int r0, int c0,
                                             float4 (&v)[4], int lane) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + (lane >> 3) + 8 * i, c = c0 + (lane & 7) * 4;
    v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < rows && c < cols) v[i] = *reinterpret_cast<const float4 *>(A + (long long)r * ld + c);     // (cols % 4 == 0)
  }
}
__device__ __forceinline__ void sl_store_tile(float *T, const float4 (&v)[4], int lane) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float *p = T + ((lane >> 3) + 8 * i) * kSlStride + (lane & 7) * 4;
    p[0] = v[i].x; p[1] = v[i].y; p[2] = v[i].z; p[3] = v[i].w;
  }
}

__global__ __launch_bounds__(kSlThreads) void small_linear_bwd_f32_kernel(SmallLinearBwdArgs g) {
  __shared__ float At[4][kSlTile * kSlStride], Bt[4][kSlTile * kSlStride];      // per wave: its chunk's operand tiles
  __shared__ float bsum[4][kSlTile];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool is_dx = (int)blockIdx.x < g.tiles_dx;
  const int tiles_k = (g.K + kSlTile - 1) / kSlTile;
  const int t = is_dx ? (int)blockIdx.x : (int)blockIdx.x - g.tiles_dx;
  const int tr = t / tiles_k, tk = t - tr * tiles_k;          // tile row (m for dX, n for dW), tile column (k)
  const int r0 = tr * kSlTile, c0 = tk * kSlTile;
  const int R = is_dx ? g.N : g.M;                            // reduction length
  const int nchunk = (R + kSlTile - 1) / kSlTile, per_wave = (nchunk + 3) / 4;
  const bool do_bias = !is_dx && g.db != nullptr && tk == 0;

  sl_f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  float bias_part = 0.f;
  float *A = At[wave], *B = Bt[wave];
  const int li = lane & 31, lk = lane >> 5;
  for (int it = 0; it < per_wave; ++it) {
    const int q0 = (it * 4 + wave) * kSlTile;                 // this wave's chunk of the reduction axis (may lie past R: zeros)
    float4 va[4], vb[4];
    if (is_dx) {
      sl_load_tile(g.G, g.ldg, g.M, g.N, r0, q0, va, lane);    // A[i = m][k = n]: rows of G
      sl_load_tile(g.W, g.ldw, g.N, g.K, q0, c0, vb, lane);    // B[k = n][j = kc]: rows of W
    } else {
      sl_load_tile(g.G, g.ldg, g.M, g.N, q0, r0, va, lane);    // stored [m][n]; read as A[i = n][k = m]
      sl_load_tile(g.X, g.ldx, g.M, g.K, q0, c0, vb, lane);    // B[k = m][j = kc]: rows of X
    }
    __syncthreads();                                           // the previous chunk's reads are done
    sl_store_tile(A, va, lane);
    sl_store_tile(B, vb, lane);
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < kSlTile; kk += 2) {
      const float a = is_dx ? A[li * kSlStride + kk + lk] : A[(kk + lk) * kSlStride + li];
      const float b = B[(kk + lk) * kSlStride + li];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    if (do_bias && lane < kSlTile) {
#pragma unroll 8
      for (int m = 0; m < kSlTile; ++m) bias_part += A[m * kSlStride + lane];
    }
  }
  __syncthreads();
  // the four partial tiles -> LDS (the operand tiles are free), summed in a fixed order.  C/D layout of the 32x32 MFMA:
  // column = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
  float *P = &At[0][0];            // 4 x 32 x 33 floats = exactly At
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * lk;
    P[wave * kSlTile * kSlStride + row * kSlStride + li] = acc[r];
  }
  if (do_bias && lane < kSlTile) bsum[wave][lane] = bias_part;
  __syncthreads();
  float *out = is_dx ? g.dX : g.dW;
  const long long ldo = is_dx ? g.lddx : g.lddw;
  const int rows_out = is_dx ? g.M : g.N;
  if (out) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = tid + kSlThreads * i, row = e >> 5, col = e & 31;
      const int o = row * kSlStride + col;
      const float s = (P[o] + P[kSlTile * kSlStride + o]) + (P[2 * kSlTile * kSlStride + o] + P[3 * kSlTile * kSlStride + o]);
      if (r0 + row < rows_out && c0 + col < g.K) out[(long long)(r0 + row) * ldo + c0 + col] = s;
    }
  }
  if (do_bias && tid < kSlTile && r0 + tid < g.N)
    g.db[r0 + tid] = (bsum[0][tid] + bsum[1][tid]) + (bsum[2][tid] + bsum[3][tid]);
}

}  // namespace snipper
