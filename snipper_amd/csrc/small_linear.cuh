// small_linear.cuh -- float32 products of decoder size, several per launch (gfx950).
//
// The decoder's Linears (reference models/deformable_transformer.py:244-343: self-attention projections, the offset /
// weight / output projections of the cross attention, the feed-forward block; 60 queries x T frames x batch = a few
// hundred rows) are latency-bound library GEMM launches in PyTorch: ~9 us of GPU time and 21-27 us of host time each,
// and a Linear's backward needs three of them (dX = G . W, dW = G^T . X, db = column sums of G, the last one issued as
// a [1, M] x [M, N] product).  Here a launch takes a LIST of up to kSgMaxProblems products
//     out[I, J] = opA(A)[I, R] . opB(B)[R, J] (+ bias[J]),      optionally  colsum[I] = sum_r opA(A)[i, r]
// with each operand read as stored or transposed (flags), so that one launch is a Linear's whole backward (dX, dW + db),
// a packed q|k / v projection pair, or that pair's backward (two dX, two dW written into the row blocks of the packed
// gradient, two db) -- no concatenations, no transposed copies.
// Workgroup b owns one 32 x 32 tile of one product; its four waves split the reduction axis (chunks of 32, interleaved),
// each accumulates the tile with v_mfma_f32_32x32x2_f32 (float32 in, float32 accumulate: bitwise an fmaf chain, so the
// result is float32-exact like the library's), and the four partial tiles meet in LDS in a fixed order: deterministic.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "gemm_bf16.cuh"      // gemm_rand: the counter-based hash shared by the dropout epilogues

namespace snipper {

typedef __attribute__((ext_vector_type(16))) float sg_f32x16;

constexpr int kSgMaxProblems = 6;
struct SmallGemmProblem {
  const float *A; long long lda;     // a_tr == 0: [I][R]   a_tr == 1: [R][I]
  const float *B; long long ldb;     // b_tr == 0: [R][J]   b_tr == 1: [J][R]
  float *out; long long ldo;         // [I][J] or nullptr
  const float *bias;                 // [J] added to every row, or nullptr
  float *colsum;                     // [I] = sum over r of opA(A)[i][r], or nullptr (needs a_tr == 1 or 0 alike)
  int I, J, R, a_tr, b_tr;
  int tile_end;                      // workgroups [previous tile_end, tile_end) belong to this product
  // B split along the reduction axis (b_tr == 0 only): rows r >= r_split of opB come from B2[r - r_split] -- the data
  // gradient of two Linears that share their input, g = [ga | gb], W = [Wa; Wb] stored apart.  r_split % 32 == 0.
  const float *B2; long long ldb2; int r_split;
  // epilogue: out = gate(dropout(relu(acc + bias))) -- relu / dropout for a forward Linear + ReLU + Dropout, gate for the
  // data gradient that flows back through such a layer (gate = the layer's OUTPUT h: kept and active <=> h > 0)
  int relu; float drop_p; uint32_t seed_lo, seed_hi;
  const float *gate; long long ldgate; float gate_scale;
};
struct SmallGemmBatch {
  SmallGemmProblem p[kSgMaxProblems];
  int count;
};

constexpr int kSgTile = 32, kSgStride = 33, kSgThreads = 256;

// a 32 x 32 float32 tile, rows r0.., columns c0.. of a row-major matrix with `rows` x `cols` valid elements (zeros
// outside; cols % 4 == 0 and 16-byte aligned rows): 64 lanes x 4 float4
__device__ __forceinline__ void sg_load_tile(const float *__restrict__ A, long long ld, int rows, int cols, int r0, int c0,
                                             float4 (&v)[4], int lane) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + (lane >> 3) + 8 * i, c = c0 + (lane & 7) * 4;
    v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < rows && c < cols) v[i] = *reinterpret_cast<const float4 *>(A + (long long)r * ld + c);
  }
}
__device__ __forceinline__ void sg_store_tile(float *T, const float4 (&v)[4], int lane) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float *p = T + ((lane >> 3) + 8 * i) * kSgStride + (lane & 7) * 4;
    p[0] = v[i].x; p[1] = v[i].y; p[2] = v[i].z; p[3] = v[i].w;
  }
}

__global__ __launch_bounds__(kSgThreads) void small_gemm_batch_f32_kernel(SmallGemmBatch batch) {
  __shared__ float At[4][kSgTile * kSgStride], Bt[4][kSgTile * kSgStride];      // per wave: its chunk's operand tiles
  __shared__ float csum[4][kSgTile];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // which product (a select chain: a run-time index into the kernel argument would copy it to scratch)
  SmallGemmProblem g = batch.p[0];
  int first = 0;
#pragma unroll
  for (int i = 1; i < kSgMaxProblems; ++i) {
    if (i < batch.count && (int)blockIdx.x >= batch.p[i - 1].tile_end) { g = batch.p[i]; first = batch.p[i - 1].tile_end; }
  }
  const int tiles_j = (g.J + kSgTile - 1) / kSgTile;
  const int t = (int)blockIdx.x - first;
  const int ti = t / tiles_j, tj = t - ti * tiles_j;
  const int r0 = ti * kSgTile, c0 = tj * kSgTile;
  const int nchunk = (g.R + kSgTile - 1) / kSgTile, per_wave = (nchunk + 3) / 4;
  const bool do_sum = g.colsum != nullptr && tj == 0;
  const bool a_tr = g.a_tr != 0, b_tr = g.b_tr != 0;

  sg_f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  float sum_part = 0.f;
  float *A = At[wave], *B = Bt[wave];
  const int li = lane & 31, lk = lane >> 5;
  // operand tiles of chunk `it` of this wave (a chunk past R reads as zeros): the stored orientation decides which axis
  // of the tile is the reduction
  const int a_rows = a_tr ? g.R : g.I, a_cols = a_tr ? g.I : g.R, b_rows = b_tr ? g.J : g.R, b_cols = b_tr ? g.R : g.J;
  float4 va[4], vb[4];
  {
    const int q0 = wave * kSgTile;
    sg_load_tile(g.A, g.lda, a_rows, a_cols, a_tr ? q0 : r0, a_tr ? r0 : q0, va, lane);
    if (!b_tr && g.B2 && q0 >= g.r_split) sg_load_tile(g.B2, g.ldb2, g.R - g.r_split, b_cols, q0 - g.r_split, c0, vb, lane);
    else sg_load_tile(g.B, g.ldb, b_tr ? b_rows : (g.B2 ? g.r_split : b_rows), b_cols, b_tr ? c0 : q0, b_tr ? q0 : c0, vb, lane);
  }
  for (int it = 0; it < per_wave; ++it) {
    __syncthreads();                                                      // the previous chunk's reads are done
    sg_store_tile(A, va, lane);
    sg_store_tile(B, vb, lane);
    __syncthreads();
    {                                                 // the next chunk: in flight while this one is multiplied (one or
      const int q0 = ((it + 1) * 4 + wave) * kSgTile; // two workgroups per CU: nothing else would hide the latency)
      sg_load_tile(g.A, g.lda, a_rows, a_cols, a_tr ? q0 : r0, a_tr ? r0 : q0, va, lane);
      if (!b_tr && g.B2 && q0 >= g.r_split) sg_load_tile(g.B2, g.ldb2, g.R - g.r_split, b_cols, q0 - g.r_split, c0, vb, lane);
      else sg_load_tile(g.B, g.ldb, b_tr ? b_rows : (g.B2 ? g.r_split : b_rows), b_cols, b_tr ? c0 : q0, b_tr ? q0 : c0, vb, lane);
    }
    // MFMA operands: lane l holds A[i = l & 31][k = l >> 5] and B[k = l >> 5][j = l & 31] of each 2-step
#pragma unroll
    for (int kk = 0; kk < kSgTile; kk += 2) {
      const float a = a_tr ? A[(kk + lk) * kSgStride + li] : A[li * kSgStride + kk + lk];
      const float b = b_tr ? B[li * kSgStride + kk + lk] : B[(kk + lk) * kSgStride + li];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    if (do_sum && lane < kSgTile) {
#pragma unroll 8
      for (int q = 0; q < kSgTile; ++q) sum_part += a_tr ? A[q * kSgStride + lane] : A[lane * kSgStride + q];
    }
  }
  __syncthreads();
  // the four partial tiles -> LDS (the operand tiles are free), summed in a fixed order.  C/D layout of the 32x32 MFMA:
  // column = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
  float *P = &At[0][0];            // 4 x 32 x 33 floats = exactly At
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * lk;
    P[wave * kSgTile * kSgStride + row * kSgStride + li] = acc[r];
  }
  if (do_sum && lane < kSgTile) csum[wave][lane] = sum_part;
  __syncthreads();
  if (g.out) {
    const bool drop = g.drop_p > 0.f;
    const float keep_scale = drop ? 1.f / (1.f - g.drop_p) : 1.f;
    const uint32_t thresh = (uint32_t)fminf(g.drop_p * 4294967296.f, 4294967040.f);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = tid + kSgThreads * i, row = e >> 5, col = e & 31;
      const int o = row * kSgStride + col;
      float s = (P[o] + P[kSgTile * kSgStride + o]) + (P[2 * kSgTile * kSgStride + o] + P[3 * kSgTile * kSgStride + o]);
      if (r0 + row < g.I && c0 + col < g.J) {
        if (g.bias) s += g.bias[c0 + col];
        if (g.relu) s = fmaxf(s, 0.f);
        if (drop) {
          const uint32_t idx = (uint32_t)(r0 + row) * (uint32_t)g.J + (uint32_t)(c0 + col);
          s = gemm_rand(idx, g.seed_lo, g.seed_hi) >= thresh ? s * keep_scale : 0.f;
        }
        if (g.gate) s = g.gate[(long long)(r0 + row) * g.ldgate + c0 + col] > 0.f ? s * g.gate_scale : 0.f;
        g.out[(long long)(r0 + row) * g.ldo + c0 + col] = s;
      }
    }
  }
  if (do_sum && tid < kSgTile && r0 + tid < g.I)
    g.colsum[r0 + tid] = (csum[0][tid] + csum[1][tid]) + (csum[2][tid] + csum[3][tid]);
}

}  // namespace snipper
