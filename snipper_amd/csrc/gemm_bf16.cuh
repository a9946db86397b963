// gemm_bf16.cuh -- bf16 MFMA kernel for the path's dense layers (gfx950).
//
//   Y[M,N] = act( X[M,K] . W[N,K]^T + bias[N] (+ R[M,N]) )        X, W, R, Y bf16; bias f32; f32 accumulate
//
// This is the shape of every projection on the hot path -- MSDeformAttn's value / output / offset /
// weight Linears and the FFN of the encoder layer (reference models/ops/modules/ms_deform_attn.py:114,
// 143-163,237 and models/deformable_transformer.py:194-198), M = B*T*S = 79 000 tokens, K, N <= 1024 --
// and, in NHWC, of the ResNet bottleneck's 1x1 convolutions with the frozen BatchNorm folded into W /
// bias, the ReLU and the residual add fused in the epilogue (reference models/backbone.py:54-64 plus
// torchvision's Bottleneck).  K is short, so the kernel lives on the HBM side of the roofline: what it
// has to do is stream X once, keep W in L2/LDS, and write Y once with the whole epilogue applied.
//
// Tiling (wave64): 256 threads = 2 x 2 waves, block tile 128 (M) x 128 (N), K step 64, each wave a
// 64 x 64 sub-tile = 4 x 4 v_mfma_f32_16x16x32_bf16 tiles (16 accumulators x 4 VGPRs).  The product is
// evaluated as Y^T = W . X^T, i.e. W is the MFMA "A" operand: the C/D layout (col = lane & 15,
// row = 4 * (lane >> 4) + reg) then gives every lane 4 CONSECUTIVE n of one output row m, so the
// epilogue stores 8 bytes per lane instead of scattered bf16s.  Both operands are K-contiguous in
// memory and in LDS (rows padded 64 -> 72 elements: the 16 lanes of a ds_read_b128 group land on 16
// distinct 4-bank slots), fetched with 16-byte loads, one K-step ahead in registers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace snipper {

// Workgroup -> (M tile, N tile) for a 1-D grid of tiles_m * tiles_n workgroups.  Consecutive workgroup ids go
// round-robin over the 8 XCDs, each with its own L2; the N tiles of one M tile all read the same X rows, so they are
// given to ONE XCD back to back: the first fetches the X tile from HBM, the others hit that XCD's L2.  (With the
// plain (M tile, N tile) = (blockIdx.x, blockIdx.y) order X came from HBM once per N tile: 3x for N = 384.)
// The grid has tiles_n * 8 * ceil(tiles_m / 8) workgroups; those whose M tile does not exist return at once.
__device__ __forceinline__ void gemm_tile_of_block(int tiles_n, int &tm, int &tn) {
  const int b = blockIdx.x;
  const int xcd = b & 7, j = b >> 3;
  tm = xcd + 8 * (j / tiles_n);
  tn = j % tiles_n;
}
inline unsigned gemm_grid_size(long long M, int N) {
  const long long tiles_m = (M + 127) / 128, tiles_n = (N + 127) / 128;
  return (unsigned)(tiles_n * 8 * ((tiles_m + 7) / 8));
}

// Output staging: the MFMA accumulator layout gives a lane 4 consecutive n of ONE row, so a direct store writes
// 8-byte pieces of 16 different rows per instruction (32-byte runs in memory).  The finished bf16 tile is therefore
// parked in LDS ([128][136], the operand buffers are free by then) and written out as whole 256-byte row segments,
// 16 bytes per lane.  Needs N % 8 == 0, ldy % 8 == 0 and a 16-byte aligned Y; otherwise the direct stores are used.
constexpr int kGemmCtStride = 136;
__device__ __forceinline__ bool gemm_wide_ok(const uint16_t *Y, long long ldy, int N) {
  return (N % 8) == 0 && (ldy % 8) == 0 && ((uintptr_t)Y % 16) == 0;
}
// `gate` (optional, [M][N] bf16, 16-byte aligned rows): elements whose gate value is not > 0 are written as 0 -- the
// sign test of a ReLU (+ dropout) backward, done here on whole 16-byte chunks instead of per accumulator fragment.
__device__ __forceinline__ void gemm_flush_tile(const uint16_t *Ct, uint16_t *Y, long long ldy, int m0, int n0, long long M, int N,
                                                const uint16_t *gate = nullptr, long long ldg = 0) {
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int idx = threadIdx.x + 256 * i, row = idx >> 4, ch = idx & 15;
    const long long m = (long long)m0 + row;
    const int n = n0 + ch * 8;
    if (m < M && n < N) {
      uint4 v = *reinterpret_cast<const uint4 *>(Ct + row * kGemmCtStride + ch * 8);
      if (gate) {
        const uint4 a = *reinterpret_cast<const uint4 *>(gate + m * ldg + n);
        // a bf16 is > 0 iff its sign bit is clear and it is not +0 (NaN gates do not occur: the gate is a ReLU output)
        auto keep = [](unsigned av, unsigned vv) {
          const unsigned lo = ((av & 0x8000u) == 0u && (av & 0x7fffu) != 0u) ? 0x0000ffffu : 0u;
          const unsigned hi = ((av & 0x80000000u) == 0u && (av & 0x7fff0000u) != 0u) ? 0xffff0000u : 0u;
          return vv & (lo | hi);
        };
        v.x = keep(a.x, v.x); v.y = keep(a.y, v.y); v.z = keep(a.z, v.z); v.w = keep(a.w, v.w);
      }
      *reinterpret_cast<uint4 *>(Y + m * ldy + n) = v;
    }
  }
}

// the same for a tile of BN columns staged with rows of BN + 8 elements
template <int BN>
__device__ __forceinline__ void gemm_flush_tile_n(const uint16_t *Ct, uint16_t *Y, long long ldy, int m0, int n0, long long M,
                                                  int N, const uint16_t *gate = nullptr, long long ldg = 0) {
  constexpr int CH = BN / 8, ITER = 128 * CH / 256, CTS = BN + 8;
  __syncthreads();
#pragma unroll
  for (int i = 0; i < ITER; ++i) {
    const int idx = threadIdx.x + 256 * i, row = idx / CH, ch = idx % CH;
    const long long m = (long long)m0 + row;
    const int n = n0 + ch * 8;
    if (m < M && n < N) {
      uint4 v = *reinterpret_cast<const uint4 *>(Ct + row * CTS + ch * 8);
      if (gate) {
        const uint4 a = *reinterpret_cast<const uint4 *>(gate + m * ldg + n);
        auto keep = [](unsigned av, unsigned vv) {
          const unsigned lo = ((av & 0x8000u) == 0u && (av & 0x7fffu) != 0u) ? 0x0000ffffu : 0u;
          const unsigned hi = ((av & 0x80000000u) == 0u && (av & 0x7fff0000u) != 0u) ? 0xffff0000u : 0u;
          return vv & (lo | hi);
        };
        v.x = keep(a.x, v.x); v.y = keep(a.y, v.y); v.z = keep(a.z, v.z); v.w = keep(a.w, v.w);
      }
      *reinterpret_cast<uint4 *>(Y + m * ldy + n) = v;
    }
  }
}

// the same with the tile's rows stored at MAPPED rows of Y (and of the gate): rowmap(m) -> row.  The stride-2 data gradient's
// parity classes write every other pixel of every other image row; as 8-byte pieces straight from the accumulators (each
// behind its own 8-byte gate load) that store phase outweighed the 4-16 K-steps of a class.
template <int BN, typename RowMap>
__device__ __forceinline__ void gemm_flush_tile_n_map(const uint16_t *Ct, uint16_t *Y, long long ldy, int m0, int n0, long long M,
                                                      int N, const uint16_t *gate, long long ldg, RowMap rowmap) {
  constexpr int CH = BN / 8, ITER = 128 * CH / 256, CTS = BN + 8;
  __syncthreads();
#pragma unroll
  for (int i = 0; i < ITER; ++i) {
    const int idx = threadIdx.x + 256 * i, row = idx / CH, ch = idx % CH;
    const long long m = (long long)m0 + row;
    const int n = n0 + ch * 8;
    if (m < M && n < N) {
      const long long dst = rowmap(m);
      uint4 v = *reinterpret_cast<const uint4 *>(Ct + row * CTS + ch * 8);
      if (gate) {
        const uint4 a = *reinterpret_cast<const uint4 *>(gate + dst * ldg + n);
        auto keep = [](unsigned av, unsigned vv) {
          const unsigned lo = ((av & 0x8000u) == 0u && (av & 0x7fffu) != 0u) ? 0x0000ffffu : 0u;
          const unsigned hi = ((av & 0x80000000u) == 0u && (av & 0x7fff0000u) != 0u) ? 0xffff0000u : 0u;
          return vv & (lo | hi);
        };
        v.x = keep(a.x, v.x); v.y = keep(a.y, v.y); v.z = keep(a.z, v.z); v.w = keep(a.w, v.w);
      }
      *reinterpret_cast<uint4 *>(Y + dst * ldy + n) = v;
    }
  }
}

typedef __attribute__((ext_vector_type(8))) __bf16 gemm_bf16x8;
typedef __attribute__((ext_vector_type(4))) float gemm_f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int gemm_u32x4;

constexpr int kGemmBM = 128, kGemmBN = 128, kGemmBK = 64, kGemmPad = 72, kGemmThreads = 256;

struct GemmArgs {
  const uint16_t *X; long long ldx;      // [M][K]
  const uint16_t *W;                     // [N][K]
  const float *bias;                     // [N] or nullptr
  const uint16_t *R; long long ldr;      // [M][N] residual or nullptr
  uint16_t *Y; long long ldy;            // [M][N]
  int M, N, K;
  float drop_p;                          // dropout after the activation (0 = none); element index m * N + n
  uint32_t seed_lo, seed_hi;
};

// counter-based hash RNG shared with csrc/ln_fused.cuh (same constants): keep iff rand >= p * 2^32
__device__ __forceinline__ uint32_t gemm_hash(uint32_t v) {
  v ^= v >> 16; v *= 0x7feb352du; v ^= v >> 15; v *= 0x846ca68bu; v ^= v >> 16;
  return v;
}
__device__ __forceinline__ uint32_t gemm_rand(uint32_t idx, uint32_t seed_lo, uint32_t seed_hi) {
  return gemm_hash(gemm_hash(idx + seed_lo * 0x9e3779b9u) ^ seed_hi);
}

__device__ __forceinline__ float gemm_bf16_to_f32(uint16_t b) { return __uint_as_float((unsigned)b << 16); }
// two floats -> packed bf16 pair (a in the low half): ONE v_cvt_pk_bf16_f32 (the scalar casts compiled to two of them plus
// a shift and an or; same round-to-nearest-even, NaN stays NaN)
typedef __attribute__((ext_vector_type(2))) float gemm_f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 gemm_bf16x2;
__device__ __forceinline__ unsigned gemm_pack2(float a, float b) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector(gemm_f32x2{a, b}, gemm_bf16x2));
}

// The grid may be smaller than the tile count: workgroup (xcd, w) then walks the tiles j = w, w + W, w + 2W, ... of its
// XCD's list (W = gridDim.x / 8 workgroups per XCD) with the first K-step of the NEXT tile in flight behind the epilogue
// of the current one.  Measured with 3 workgroups per CU resident for the whole launch: no gain over one tile per
// workgroup (79 000 x 384 x 384: 44.4 vs 44.6 us; x1024: 96 vs 92 us) -- the dispatcher already starts a new workgroup
// while its two neighbours compute -- so the launcher uses gemm_grid_size() workgroups, one tile each.
// Also measured without gain: a cross-workgroup L2 prefetch (the first N tile's workgroup touches the X rows of the M tile
// its XCD starts 4 / 12 / 32 tiles later): 79 000 x 384 x 384 45.2 -> 46.5 - 46.7 us, every other shape equal or slower.
template <bool RELU>
__global__ __launch_bounds__(kGemmThreads) __attribute__((amdgpu_waves_per_eu(3, 3))) void linear_bf16_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) uint16_t smem[(kGemmBM + kGemmBN) * kGemmPad];
  uint16_t *Xs = smem, *Ws = smem + kGemmBM * kGemmPad;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1;
  const int tiles_n = (g.N + kGemmBN - 1) / kGemmBN, tiles_m = (g.M + kGemmBM - 1) / kGemmBM;
  const int xcd = blockIdx.x & 7, per_xcd = (int)(gridDim.x >> 3);
  const int ntile = ((tiles_m - xcd + 7) >> 3) * tiles_n;        // tiles of this XCD (M tiles xcd, xcd + 8, ...)
  int j = (int)(blockIdx.x >> 3);
  if (j >= ntile) return;

  // loader: 4 x 16 B per operand per thread and K-step; row = 32 * i + tid / 8, 8-element chunk = tid % 8.  Raw buffer
  // loads: the descriptor covers the tile's rows only (base = the tile's first row, a uniform value), so a row past the
  // end of the matrix reads as 0 without a clamp and the per-lane state is ONE 32-bit offset per operand for all tiles
  // (+ a multiple of the row-group stride, added per load); the K-step goes into the scalar offset.  (The range check
  // of a raw buffer looks at the vector offset only, which is why the row group must be part of it.)
  const int lrow = tid >> 3, kc = tid & 7;
  const unsigned x_voff = ((unsigned)lrow * (unsigned)g.ldx + kc * 8) * 2u, w_voff = ((unsigned)lrow * (unsigned)g.K + kc * 8) * 2u;
  const unsigned x_step = 32u * (unsigned)g.ldx * 2u, w_step = 32u * (unsigned)g.K * 2u;      // bytes between row groups
  int lds_off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) lds_off[i] = (lrow + 32 * i) * kGemmPad + kc * 8;
  gemm_u32x4 xr[4], wr[4];
  __amdgpu_buffer_rsrc_t xsrc, wsrc;
  auto load_step = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      xr[i] = __builtin_amdgcn_raw_buffer_load_b128(xsrc, x_voff + i * x_step, (unsigned)k0 * 2u, 0);
      wr[i] = __builtin_amdgcn_raw_buffer_load_b128(wsrc, w_voff + i * w_step, (unsigned)k0 * 2u, 0);
    }
  };
  auto start_tile = [&](int jj) {          // descriptors of tile jj, its first K-step in flight
    const int m0 = (xcd + 8 * (jj / tiles_n)) * kGemmBM, n0 = (jj % tiles_n) * kGemmBN;
    const int mrows = min(kGemmBM, g.M - m0), nrows = min(kGemmBN, g.N - n0);
    xsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(g.X + (long long)m0 * g.ldx), 0,
                                             (int)(((long long)(mrows - 1) * g.ldx + g.K) * 2), 0x00020000);
    wsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(g.W + (long long)n0 * g.K), 0,
                                             (int)((long long)nrows * g.K * 2), 0x00020000);
    load_step(0);
  };
  start_tile(j);
  const int frag_row = lane & 15, frag_k = (lane >> 4) * 8;
  const bool wide = gemm_wide_ok(g.Y, g.ldy, g.N);
  const bool drop = g.drop_p > 0.f;
  const float keep_scale = drop ? 1.f / (1.f - g.drop_p) : 1.f;
  const uint32_t thresh = (uint32_t)fminf(g.drop_p * 4294967296.f, 4294967040.f);

  for (;;) {
    const int m0 = (xcd + 8 * (j / tiles_n)) * kGemmBM, n0 = (j % tiles_n) * kGemmBN;
    gemm_f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) acc[i][jj] = gemm_f32x4{0.f, 0.f, 0.f, 0.f};

    for (int k0 = 0; k0 < g.K; k0 += kGemmBK) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        *reinterpret_cast<gemm_u32x4 *>(Xs + lds_off[i]) = xr[i];
        *reinterpret_cast<gemm_u32x4 *>(Ws + lds_off[i]) = wr[i];
      }
      __syncthreads();
      if (k0 + kGemmBK < g.K) load_step(k0 + kGemmBK);     // next K-step in flight while this one is multiplied
#pragma unroll
      for (int kk = 0; kk < kGemmBK; kk += 32) {
        gemm_bf16x8 wf[4], xf[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          wf[i] = *reinterpret_cast<const gemm_bf16x8 *>(Ws + (wn * 64 + i * 16 + frag_row) * kGemmPad + kk + frag_k);
          xf[i] = *reinterpret_cast<const gemm_bf16x8 *>(Xs + (wm * 64 + i * 16 + frag_row) * kGemmPad + kk + frag_k);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int jj = 0; jj < 4; ++jj)
            acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf[jj], acc[i][jj], 0, 0, 0);
      }
      __syncthreads();
    }
    const int jn = j + per_xcd;
    const bool more = jn < ntile;
    if (more) start_tile(jn);      // in flight behind this tile's epilogue

    // epilogue: lane holds n = nb + 4*(lane>>4) + r (r = 0..3) of output row m = mb + (lane & 15)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int n = n0 + wn * 64 + i * 16 + (lane >> 4) * 4;
      if (n >= g.N) continue;
      gemm_f32x4 b = {0.f, 0.f, 0.f, 0.f};
      if (g.bias) b = *reinterpret_cast<const gemm_f32x4 *>(g.bias + n);
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const int m = m0 + wm * 64 + jj * 16 + (lane & 15);
        if (m >= g.M) continue;
        gemm_f32x4 v = acc[i][jj] + b;
        if (g.R) {
          const uint2 r = *reinterpret_cast<const uint2 *>(g.R + (long long)m * g.ldr + n);
          v.x += __uint_as_float(r.x << 16); v.y += __uint_as_float(r.x & 0xffff0000u);
          v.z += __uint_as_float(r.y << 16); v.w += __uint_as_float(r.y & 0xffff0000u);
        }
        if (RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        if (drop) {
          const uint32_t e = (uint32_t)m * (uint32_t)g.N + (uint32_t)n;
          v.x = gemm_rand(e, g.seed_lo, g.seed_hi) >= thresh ? v.x * keep_scale : 0.f;
          v.y = gemm_rand(e + 1, g.seed_lo, g.seed_hi) >= thresh ? v.y * keep_scale : 0.f;
          v.z = gemm_rand(e + 2, g.seed_lo, g.seed_hi) >= thresh ? v.z * keep_scale : 0.f;
          v.w = gemm_rand(e + 3, g.seed_lo, g.seed_hi) >= thresh ? v.w * keep_scale : 0.f;
        }
        uint2 o;
        o.x = gemm_pack2(v.x, v.y);
        o.y = gemm_pack2(v.z, v.w);
        if (wide) *reinterpret_cast<uint2 *>(smem + (m - m0) * kGemmCtStride + (n - n0)) = o;
        else *reinterpret_cast<uint2 *>(g.Y + (long long)m * g.ldy + n) = o;
      }
    }
    if (wide) gemm_flush_tile(smem, g.Y, g.ldy, m0, n0, g.M, g.N);
    if (!more) break;
    j = jn;
    __syncthreads();               // the staged tile has been read out before the next tile's operands overwrite it
  }
}

// ---- the same product on 128 x 64 tiles -----------------------------------------------------------------------------
// Each wave owns 64 x 32: half the accumulators, 94 VGPRs, 27.6 KB of LDS -> FIVE workgroups per CU instead of three.
// The phases of a short-K tile (first load, LDS writes, MFMA, store) do not overlap within a workgroup
// (profiles/r02_gemm_phase_ablation.json), so residency is what hides them; the price is X read from L2 once per 64
// instead of per 128 output columns.  Measured (tools/gemmbench.py, us, 128x128 -> 128x64): 240000x256x64 34 -> 27,
// 240000x64x64 22 -> 16, 3800x2048x512 32 -> 24, 60000x128x512 28 -> 24, 79000x384x384 45 -> 45, but 79000x384x1024
// 92 -> 102 and 79000x1024x384 84 -> 95: the launcher takes these tiles unless the product is both long and wide
// (gemm_use_n64).
constexpr int kGemmBN64 = 64, kGemmCt64 = 72;
inline bool gemm_use_n64(long long M, int N, int K) { return !(M >= 32768 && (K >= 1024 || N >= 1024)); }
template <bool RELU>
__global__ __launch_bounds__(kGemmThreads) __attribute__((amdgpu_waves_per_eu(5, 5))) void linear_bf16_n64_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) uint16_t smem[(kGemmBM + kGemmBN64) * kGemmPad];
  uint16_t *Xs = smem, *Ws = smem + kGemmBM * kGemmPad;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1;
  const int tiles_n = (g.N + kGemmBN64 - 1) / kGemmBN64;
  const int xcd = blockIdx.x & 7, j = (int)(blockIdx.x >> 3);
  const int tm = xcd + 8 * (j / tiles_n), tn = j % tiles_n;
  if ((long long)tm * kGemmBM >= g.M) return;
  const int m0 = tm * kGemmBM, n0 = tn * kGemmBN64;
  const int lrow = tid >> 3, kc = tid & 7;
  const unsigned x_voff = ((unsigned)lrow * (unsigned)g.ldx + kc * 8) * 2u, w_voff = ((unsigned)lrow * (unsigned)g.K + kc * 8) * 2u;
  const unsigned x_step = 32u * (unsigned)g.ldx * 2u, w_step = 32u * (unsigned)g.K * 2u;
  int lds_off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) lds_off[i] = (lrow + 32 * i) * kGemmPad + kc * 8;
  const int mrows = min(kGemmBM, g.M - m0), nrows = min(kGemmBN64, g.N - n0);
  const __amdgpu_buffer_rsrc_t xsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint16_t *>(g.X + (long long)m0 * g.ldx), 0, (int)(((long long)(mrows - 1) * g.ldx + g.K) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t wsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint16_t *>(g.W + (long long)n0 * g.K), 0, (int)((long long)nrows * g.K * 2), 0x00020000);
  gemm_u32x4 xr[4], wr[2];
  auto load_step = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) xr[i] = __builtin_amdgcn_raw_buffer_load_b128(xsrc, x_voff + i * x_step, (unsigned)k0 * 2u, 0);
#pragma unroll
    for (int i = 0; i < 2; ++i) wr[i] = __builtin_amdgcn_raw_buffer_load_b128(wsrc, w_voff + i * w_step, (unsigned)k0 * 2u, 0);
  };
  load_step(0);
  gemm_f32x4 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) acc[i][jj] = gemm_f32x4{0.f, 0.f, 0.f, 0.f};
  const int frag_row = lane & 15, frag_k = (lane >> 4) * 8;
  for (int k0 = 0; k0 < g.K; k0 += kGemmBK) {
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<gemm_u32x4 *>(Xs + lds_off[i]) = xr[i];
#pragma unroll
    for (int i = 0; i < 2; ++i) *reinterpret_cast<gemm_u32x4 *>(Ws + lds_off[i]) = wr[i];
    __syncthreads();
    if (k0 + kGemmBK < g.K) load_step(k0 + kGemmBK);
#pragma unroll
    for (int kk = 0; kk < kGemmBK; kk += 32) {
      gemm_bf16x8 wf[2], xf[4];
#pragma unroll
      for (int i = 0; i < 2; ++i)
        wf[i] = *reinterpret_cast<const gemm_bf16x8 *>(Ws + (wn * 32 + i * 16 + frag_row) * kGemmPad + kk + frag_k);
#pragma unroll
      for (int i = 0; i < 4; ++i)
        xf[i] = *reinterpret_cast<const gemm_bf16x8 *>(Xs + (wm * 64 + i * 16 + frag_row) * kGemmPad + kk + frag_k);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
          acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf[jj], acc[i][jj], 0, 0, 0);
    }
    __syncthreads();
  }
  const bool wide = gemm_wide_ok(g.Y, g.ldy, g.N);
  const bool drop = g.drop_p > 0.f;
  const float keep_scale = drop ? 1.f / (1.f - g.drop_p) : 1.f;
  const uint32_t thresh = (uint32_t)fminf(g.drop_p * 4294967296.f, 4294967040.f);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int n = n0 + wn * 32 + i * 16 + (lane >> 4) * 4;
    if (n >= g.N) continue;
    gemm_f32x4 b = {0.f, 0.f, 0.f, 0.f};
    if (g.bias) b = *reinterpret_cast<const gemm_f32x4 *>(g.bias + n);
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int m = m0 + wm * 64 + jj * 16 + (lane & 15);
      if (m >= g.M) continue;
      gemm_f32x4 v = acc[i][jj] + b;
      if (g.R) {
        const uint2 r = *reinterpret_cast<const uint2 *>(g.R + (long long)m * g.ldr + n);
        v.x += __uint_as_float(r.x << 16); v.y += __uint_as_float(r.x & 0xffff0000u);
        v.z += __uint_as_float(r.y << 16); v.w += __uint_as_float(r.y & 0xffff0000u);
      }
      if (RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      if (drop) {
        const uint32_t e = (uint32_t)m * (uint32_t)g.N + (uint32_t)n;
        v.x = gemm_rand(e, g.seed_lo, g.seed_hi) >= thresh ? v.x * keep_scale : 0.f;
        v.y = gemm_rand(e + 1, g.seed_lo, g.seed_hi) >= thresh ? v.y * keep_scale : 0.f;
        v.z = gemm_rand(e + 2, g.seed_lo, g.seed_hi) >= thresh ? v.z * keep_scale : 0.f;
        v.w = gemm_rand(e + 3, g.seed_lo, g.seed_hi) >= thresh ? v.w * keep_scale : 0.f;
      }
      uint2 o;
      o.x = gemm_pack2(v.x, v.y);
      o.y = gemm_pack2(v.z, v.w);
      if (wide) *reinterpret_cast<uint2 *>(smem + (m - m0) * kGemmCt64 + (n - n0)) = o;
      else *reinterpret_cast<uint2 *>(g.Y + (long long)m * g.ldy + n) = o;
    }
  }
  if (wide) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid + 256 * i, row = idx >> 3, ch = idx & 7;
      const long long m = (long long)m0 + row;
      const int n = n0 + ch * 8;
      if (m < g.M && n < g.N)
        *reinterpret_cast<uint4 *>(g.Y + m * g.ldy + n) = *reinterpret_cast<const uint4 *>(smem + row * kGemmCt64 + ch * 8);
    }
  }
}

// ---- data gradient: Y[M,N] = X[M,K] . W[K,N]  (W row-major with the REDUCTION index as its slow axis) ------------
// dX = dY . W for a Linear / 1x1 convolution whose weight is stored [out, in] = [K, N]: the kernel above would need
// W transposed.  Here the W tile is staged as it lies in memory ([k][n], 16-byte loads along n) and the MFMA operand
// is read with ds_read_b64_tr_b16 (the transposing LDS read, see wgrad_bf16.cuh): lane group g receives k-rows
// {4g..4g+3} and {16+4g..16+4g+3} of each 32-step; the X operand is read in the SAME k order (two 8-byte reads
// instead of one 16-byte read), so the products pair up correctly.  No epilogue besides the bf16 rounding.
typedef __attribute__((__vector_size__(4 * sizeof(__bf16)))) __bf16 gemm_bf16x4;
constexpr int kGemmTrStride = 144;       // [64 k][128 n + 16]: 288-byte rows, conflict-free transposed reads

struct GemmNNArgs {
  const uint16_t *X; long long ldx;      // [M][K]
  const uint16_t *W; long long ldw;      // [K][N]
  uint16_t *Y; long long ldy;            // [M][N]
  int M, N, K;
  const uint16_t *R; long long ldr;      // [M][N] addend (e.g. the gradient arriving over a skip connection) or nullptr
  const uint16_t *A; long long lda;      // [M][N] activation whose sign gates the result (ReLU / ReLU+dropout
  float gate_scale;                      //   backward): Y = A > 0 ? Y * gate_scale : 0;  nullptr = no gate
};

template <int TRS>
__device__ __forceinline__ gemm_bf16x8 gemm_tr_frag(const uint16_t *tile, int byte_off) {
  typedef __attribute__((address_space(3))) gemm_bf16x4 lds_v4;
  const char *base = reinterpret_cast<const char *>(tile) + byte_off;
  const gemm_bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4 *)(base));
  const gemm_bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4 *)(base + 16 * TRS * 2));
  return gemm_bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

// BN = 128: the tile of linear_bf16_kernel (3 workgroups per CU), the only instantiation.  BN = 64 (each wave 64 x 32,
// five workgroups per CU, as linear_bf16_n64_kernel) was built and measured SLOWER here at every shape of the step
// (79000x384x384 41 -> 50 us, 60000x512x256 28 -> 33 us, 3800x2048x1024 29 -> 32 us; 6 VGPRs spilled at 5 waves): this
// kernel's X operand costs two 8-byte LDS reads per fragment, and halving the W reuse doubles their share.
// W rows in LDS: BN + 16 elements (288 bytes: the 8 k-rows a 32-lane half reads with ds_read_b64_tr_b16 land on 8
// distinct groups of 8 banks).
template <int BN>
__global__ __launch_bounds__(kGemmThreads) __attribute__((amdgpu_waves_per_eu(BN == 128 ? 3 : 5, BN == 128 ? 3 : 5)))
void linear_bf16_nn_kernel(GemmNNArgs g) {
  constexpr int TRS = BN + 16;            // W tile row stride (elements)
  constexpr int NI = BN / 32;             // 16-column blocks per wave
  constexpr int WCH = BN / 8;             // 16-byte chunks per W k-row
  constexpr int WROWS = kGemmThreads / WCH, WPASS = kGemmBK / WROWS;      // k-rows per loader pass, passes per K-step
  constexpr int CTS = BN + 8;             // staged output row stride
  __shared__ __attribute__((aligned(16))) uint16_t smem[kGemmBM * kGemmPad + kGemmBK * TRS];
  uint16_t *Xs = smem, *Ws = smem + kGemmBM * kGemmPad;
  const bool wide = gemm_wide_ok(g.Y, g.ldy, g.N);
  const bool gate_in_flush = wide && g.A && (g.lda % 8) == 0 && ((uintptr_t)g.A % 16) == 0;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1;
  const int tiles_n = (g.N + BN - 1) / BN;
  const int xcd = blockIdx.x & 7, jb = (int)(blockIdx.x >> 3);
  const int tm = xcd + 8 * (jb / tiles_n), tn = jb % tiles_n;
  if ((long long)tm * kGemmBM >= g.M) return;
  const int m0 = tm * kGemmBM, n0 = tn * BN;

  // loaders (raw buffer loads as in linear_bf16_kernel).  X: 128 rows x 8 chunks of 8 k, row = 32 * i + tid / 8; the
  // descriptor covers the tile's rows, so rows past the end read as 0.  W: 64 k-rows x WCH chunks of 8 n, k-row =
  // WROWS * i + tid / WCH; a lane whose columns lie past N gets an offset outside the descriptor (reads as 0).
  const int lrow = tid >> 3, kc = tid & 7, krow = tid / WCH, nc = tid % WCH;
  const unsigned x_voff = ((unsigned)lrow * (unsigned)g.ldx + kc * 8) * 2u, x_step = 32u * (unsigned)g.ldx * 2u;
  const unsigned w_step = (unsigned)WROWS * (unsigned)g.ldw * 2u;
  const unsigned w_voff = n0 + nc * 8 < g.N ? ((unsigned)krow * (unsigned)g.ldw + nc * 8) * 2u : 0x80000000u;
  int x_off[4], w_off[WPASS];
#pragma unroll
  for (int i = 0; i < 4; ++i) x_off[i] = (lrow + 32 * i) * kGemmPad + kc * 8;
#pragma unroll
  for (int i = 0; i < WPASS; ++i) w_off[i] = (krow + WROWS * i) * TRS + nc * 8;
  const int mrows = min(kGemmBM, g.M - m0);
  const __amdgpu_buffer_rsrc_t xsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint16_t *>(g.X + (long long)m0 * g.ldx), 0, (int)(((long long)(mrows - 1) * g.ldx + g.K) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t wsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint16_t *>(g.W + n0), 0, (int)(((long long)(g.K - 1) * g.ldw + (g.N - n0)) * 2), 0x00020000);
  gemm_u32x4 xr[4], wr[WPASS];
  auto load_step = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) xr[i] = __builtin_amdgcn_raw_buffer_load_b128(xsrc, x_voff + i * x_step, (unsigned)k0 * 2u, 0);
#pragma unroll
    for (int i = 0; i < WPASS; ++i)
      wr[i] = __builtin_amdgcn_raw_buffer_load_b128(wsrc, w_voff + i * w_step, (unsigned)k0 * (unsigned)g.ldw * 2u, 0);
  };
  load_step(0);
  gemm_f32x4 acc[NI][4];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = gemm_f32x4{0.f, 0.f, 0.f, 0.f};

  const int grp = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
  const int tr_base = ((grp * 4 + q) * TRS + 4 * p) * 2;       // bytes (transposed W reads)
  const int frag_row = lane & 15;
  for (int k0 = 0; k0 < g.K; k0 += kGemmBK) {
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<gemm_u32x4 *>(Xs + x_off[i]) = xr[i];
#pragma unroll
    for (int i = 0; i < WPASS; ++i) *reinterpret_cast<gemm_u32x4 *>(Ws + w_off[i]) = wr[i];
    __syncthreads();
    if (k0 + kGemmBK < g.K) load_step(k0 + kGemmBK);
#pragma unroll
    for (int kk = 0; kk < kGemmBK; kk += 32) {
      gemm_bf16x8 wf[NI], xf[4];
#pragma unroll
      for (int i = 0; i < NI; ++i) wf[i] = gemm_tr_frag<TRS>(Ws, tr_base + (kk * TRS + wn * (BN / 2) + i * 16) * 2);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const uint16_t *xrow = Xs + (wm * 64 + i * 16 + frag_row) * kGemmPad + kk + grp * 4;
        const gemm_bf16x4 lo = *reinterpret_cast<const gemm_bf16x4 *>(xrow);
        const gemm_bf16x4 hi = *reinterpret_cast<const gemm_bf16x4 *>(xrow + 16);
        xf[i] = gemm_bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int n = n0 + wn * (BN / 2) + i * 16 + (lane >> 4) * 4;
    if (n >= g.N) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = m0 + wm * 64 + j * 16 + (lane & 15);
      if (m >= g.M) continue;
      gemm_f32x4 v = acc[i][j];
      if (g.R) {
        const uint2 r = *reinterpret_cast<const uint2 *>(g.R + (long long)m * g.ldr + n);
        v.x += __uint_as_float(r.x << 16); v.y += __uint_as_float(r.x & 0xffff0000u);
        v.z += __uint_as_float(r.y << 16); v.w += __uint_as_float(r.y & 0xffff0000u);
      }
      if (g.A) {
        const float s = g.gate_scale;
        v.x *= s; v.y *= s; v.z *= s; v.w *= s;
        if (!gate_in_flush) {      // narrow path: sign test here, per fragment
          const uint2 a = *reinterpret_cast<const uint2 *>(g.A + (long long)m * g.lda + n);
          v.x = __uint_as_float(a.x << 16) > 0.f ? v.x : 0.f;
          v.y = __uint_as_float(a.x & 0xffff0000u) > 0.f ? v.y : 0.f;
          v.z = __uint_as_float(a.y << 16) > 0.f ? v.z : 0.f;
          v.w = __uint_as_float(a.y & 0xffff0000u) > 0.f ? v.w : 0.f;
        }
      }
      uint2 o;
      o.x = gemm_pack2(v.x, v.y);
      o.y = gemm_pack2(v.z, v.w);
      if (wide) *reinterpret_cast<uint2 *>(smem + (m - m0) * CTS + (n - n0)) = o;
      else *reinterpret_cast<uint2 *>(g.Y + (long long)m * g.ldy + n) = o;
    }
  }
  if (wide) gemm_flush_tile_n<BN>(smem, g.Y, g.ldy, m0, n0, g.M, g.N, gate_in_flush ? g.A : nullptr, g.lda);
}

// Backward of (ReLU -> dropout) given only the layer's OUTPUT y: a kept, active element has y > 0, a dropped or
// inactive one y == 0, so  dL/dpre = y > 0 ? g * scale : 0  (scale = 1 / (1 - p); p = 0: plain ReLU backward).
// bf16 in / out, 8 elements per thread.
__global__ __launch_bounds__(256) void relu_dropout_bwd_kernel(const uint16_t *__restrict__ gy, const uint16_t *__restrict__ y,
                                                               uint16_t *__restrict__ out, long long n8, float scale) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n8) return;
  const gemm_u32x4 a = reinterpret_cast<const gemm_u32x4 *>(gy)[i], b = reinterpret_cast<const gemm_u32x4 *>(y)[i];
  const unsigned av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w};
  unsigned o[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float g0 = __uint_as_float(av[k] << 16) * scale, g1 = __uint_as_float(av[k] & 0xffff0000u) * scale;
    const float y0 = __uint_as_float(bv[k] << 16), y1 = __uint_as_float(bv[k] & 0xffff0000u);
    o[k] = gemm_pack2(y0 > 0.f ? g0 : 0.f, y1 > 0.f ? g1 : 0.f);
  }
  reinterpret_cast<gemm_u32x4 *>(out)[i] = gemm_u32x4{o[0], o[1], o[2], o[3]};
}

// ---- 3x3 convolution, NHWC, as an implicit GEMM on the same tile machinery --------------------------------
//   Y[b, oy, ox, :] = act( sum_{ky,kx} X[b, oy*s + ky - 1, ox*s + kx - 1, :] . W[:, ky, kx, :]^T + bias )
// X [B, H, W, Cin] bf16, W [Cout, 3, 3, Cin] bf16 (the channels_last layout of a PyTorch conv weight), padding 1,
// stride s in {1, 2}, Cin % 64 == 0.  The K loop runs over (tap, 64-channel slice): a row of the A tile is the
// input pixel under that tap for one output pixel (zeros outside the image), so nothing is materialised.
struct Conv3x3Args {
  const uint16_t *X;   // [B][H][W][Cin]
  const uint16_t *W;   // [Cout][3][3][Cin]
  const float *bias;   // [Cout] or nullptr
  uint16_t *Y;         // [B][Ho][Wo][Cout]
  int B, H, Wd, Cin, Cout, Ho, Wo, stride;
  // dgrad2 != 0: the data gradient of a STRIDE-2 convolution, one launch per parity class (cy, cx) of the input pixel
  // (iy, ix) = (2a + cy, 2b + cx):  dX[n, iy, ix, :] = sum over the taps with (iy + 1 - ky) and (ix + 1 - kx) even of
  // G[n, (iy + 1 - ky) / 2, (ix + 1 - kx) / 2, :] . W'[:, ky, kx, :]^T  -- 1, 2, 2 or 4 taps per class instead of 9, every
  // MFMA useful.  Here X = G [B][H][W][Cin] (the output gradient, "Cin" = the convolution's Cout), W = W' [Cout'][3][3][Cin]
  // (the weight with its channel roles swapped), Y = dX [B][Hy][Wy][Cout]; a row of the launch is (n, a, b) over the
  // class's grid Ho x Wo and is stored at pixel (2a + cy, 2b + cx) of Y.
  // dgrad2 == 2: ALL FOUR classes in one launch -- class c = 0..3 = (cy, cx) = (1,1), (1,0), (0,1), (0,0) (most taps first) owns
  // the blocks [cls_end[c - 1], cls_end[c]) of the grid (multiples of 8: the XCD map works per class); the kernel fills in cy,
  // cx, Ho, Wo from its block index.  Four separate launches of 4-16 K-steps each were mostly ramp and tail.
  int dgrad2, cy, cx, Hy, Wy;
  // gate (optional, the layout of Y): outputs whose gate value is not > 0 are written as 0 -- when this launch is a data
  // gradient and Y's activation came out of a ReLU, that ReLU's backward happens here, in the store phase
  const uint16_t *gate;
  // flip != 0: the weight is read with its taps reversed (tap (ky, kx) uses W[:, 2 - ky, 2 - kx, :]) -- the stride-1 data
  // gradient is the same convolution with reversed taps and swapped channel roles, so the caller only has to swap the
  // channel axes of the weight, not to flip it as well
  int flip;
  int cls_end[4];
};

// the merged stride-2 data gradient: this block's class and its block index within the class
__device__ __forceinline__ int conv_dgrad2_class(Conv3x3Args &g) {
  int bid = (int)blockIdx.x;
  if (g.dgrad2 == 2) {
    int c = 0;
    while (c < 3 && bid >= g.cls_end[c]) ++c;
    bid -= c ? g.cls_end[c - 1] : 0;
    g.cy = c < 2 ? 1 : 0;
    g.cx = (c & 1) ? 0 : 1;
    g.Ho = (g.Hy - g.cy + 1) / 2;
    g.Wo = (g.Wy - g.cx + 1) / 2;
  }
  return bid;
}

// BN = 128 or 64 output channels per workgroup (64: each wave 64 x 32, more workgroups per CU; the launcher takes it when
// the 128-wide grid would not fill the chip -- layer3 / layer4 of the ResNet have 238 / 120 such tiles for 768 slots).
template <bool RELU, int BN>
__global__ __launch_bounds__(kGemmThreads) __attribute__((amdgpu_waves_per_eu(BN == 128 ? 3 : 4, BN == 128 ? 3 : 4)))
void conv3x3_bf16_kernel(Conv3x3Args g) {
  constexpr int NI = BN / 32, WL = BN / 32, CTS = BN + 8;      // 16-column blocks per wave, W loads per thread and step
  __shared__ __attribute__((aligned(16))) uint16_t smem[(kGemmBM + BN) * kGemmPad];
  uint16_t *Xs = smem, *Ws = smem + kGemmBM * kGemmPad;
  const int bid = conv_dgrad2_class(g);
  const bool wide = gemm_wide_ok(g.Y, g.Cout, g.Cout) && (!g.gate || ((uintptr_t)g.gate % 16) == 0);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1;
  const int M = g.B * g.Ho * g.Wo;
  const int tiles_n = (g.Cout + BN - 1) / BN;
  const int xcd = bid & 7, jb = bid >> 3;
  const int tm = xcd + 8 * (jb / tiles_n), tn = jb % tiles_n;
  if ((long long)tm * kGemmBM >= M) return;
  const int m0 = tm * kGemmBM, n0 = tn * BN;
  // taps of this launch: all nine, or the parity class's (dgrad2): ky in {1} / {0, 2} for cy = 0 / 1, kx likewise
  const int nty = g.dgrad2 ? (g.cy ? 2 : 1) : 3, ntx = g.dgrad2 ? (g.cx ? 2 : 1) : 3;
  const int kslices = g.Cin / kGemmBK, steps = nty * ntx * kslices;

  // loader: rows 32 * i + tid / 8 of the tile, 8-element chunk tid % 8.  Raw buffer loads: the X descriptor covers the
  // whole activation (< 2 GiB, checked by the launcher) and a tap that falls outside the image gets an offset outside the
  // descriptor -- it reads as 0 without a select; the W descriptor covers the tile's Cout rows (rows past Cout read as
  // 0), one lane offset for all taps, the tap / channel slice in the scalar offset.
  const int lrow = tid >> 3, kc = tid & 7;
  int py[4], px[4], pix[4], lds_off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = lrow + 32 * i;
    const int m = min(m0 + row, M - 1);
    const int b = m / (g.Ho * g.Wo);
    const int r = m - b * (g.Ho * g.Wo);
    py[i] = g.dgrad2 ? r / g.Wo : (r / g.Wo) * g.stride - 1;
    px[i] = g.dgrad2 ? r % g.Wo : (r % g.Wo) * g.stride - 1;
    pix[i] = (b * g.H + py[i]) * g.Wd + px[i];           // (may be "negative" by up to a row + 1: only used when in range)
    lds_off[i] = row * kGemmPad + kc * 8;
  }
  const __amdgpu_buffer_rsrc_t xsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint16_t *>(g.X), 0, (int)((long long)g.B * g.H * g.Wd * g.Cin * 2), 0x00020000);
  const int nrows = min(BN, g.Cout - n0);
  const __amdgpu_buffer_rsrc_t wsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint16_t *>(g.W + (long long)n0 * 9 * g.Cin), 0, (int)((long long)nrows * 9 * g.Cin * 2), 0x00020000);
  const unsigned w_voff = ((unsigned)lrow * 9u * (unsigned)g.Cin + kc * 8) * 2u, w_step = 32u * 9u * (unsigned)g.Cin * 2u;
  auto load_step = [&](int s, gemm_u32x4 (&xr)[4], gemm_u32x4 (&wr)[WL]) {
    const int t = s / kslices, k0 = (s - t * kslices) * kGemmBK;
    const int ty = t / ntx, tx = t - ty * ntx;
    // forward: tap (ky, kx) reads input pixel (oy * s - 1 + ky, ...); dgrad2: tap ky = cy ? 2 * ty : 1 reads gradient
    // pixel a + (cy + 1 - ky) / 2
    const int ky = g.dgrad2 ? (g.cy ? 2 * ty : 1) : ty, kx = g.dgrad2 ? (g.cx ? 2 * tx : 1) : tx;
    const int dy = g.dgrad2 ? (g.cy + 1 - ky) / 2 : ky, dx = g.dgrad2 ? (g.cx + 1 - kx) / 2 : kx;
    const int tap = g.flip ? 8 - (ky * 3 + kx) : ky * 3 + kx;      // (which slice of W this step multiplies)
    const int dpix = dy * g.Wd + dx;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int iy = py[i] + dy, ix = px[i] + dx;
      const bool ok = iy >= 0 && iy < g.H && ix >= 0 && ix < g.Wd;
      const unsigned xoff = ok ? ((unsigned)(pix[i] + dpix) * (unsigned)g.Cin + (unsigned)(k0 + kc * 8)) * 2u : 0x80000000u;
      xr[i] = __builtin_amdgcn_raw_buffer_load_b128(xsrc, xoff, 0, 0);
      if (i < WL) wr[i] = __builtin_amdgcn_raw_buffer_load_b128(wsrc, w_voff + i * w_step, (unsigned)(tap * g.Cin + k0) * 2u, 0);
    }
  };
  gemm_u32x4 xr[4], wr[WL];
  load_step(0, xr, wr);
  gemm_f32x4 acc[NI][4];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = gemm_f32x4{0.f, 0.f, 0.f, 0.f};
  const int frag_row = lane & 15, frag_k = (lane >> 4) * 8;
  for (int s = 0; s < steps; ++s) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<gemm_u32x4 *>(Xs + lds_off[i]) = xr[i];
      if (i < WL) *reinterpret_cast<gemm_u32x4 *>(Ws + lds_off[i]) = wr[i];
    }
    __syncthreads();
    if (s + 1 < steps) load_step(s + 1, xr, wr);
#pragma unroll
    for (int kk = 0; kk < kGemmBK; kk += 32) {
      gemm_bf16x8 wf[NI], xf[4];
#pragma unroll
      for (int i = 0; i < NI; ++i)
        wf[i] = *reinterpret_cast<const gemm_bf16x8 *>(Ws + (wn * (BN / 2) + i * 16 + frag_row) * kGemmPad + kk + frag_k);
#pragma unroll
      for (int i = 0; i < 4; ++i)
        xf[i] = *reinterpret_cast<const gemm_bf16x8 *>(Xs + (wm * 64 + i * 16 + frag_row) * kGemmPad + kk + frag_k);
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int n = n0 + wn * (BN / 2) + i * 16 + (lane >> 4) * 4;
    if (n >= g.Cout) continue;
    gemm_f32x4 b = {0.f, 0.f, 0.f, 0.f};
    if (g.bias) b = *reinterpret_cast<const gemm_f32x4 *>(g.bias + n);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = m0 + wm * 64 + j * 16 + (lane & 15);
      if (m >= M) continue;
      gemm_f32x4 v = acc[i][j] + b;
      if (RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      uint2 o;
      o.x = gemm_pack2(v.x, v.y);
      o.y = gemm_pack2(v.z, v.w);
      if (wide) *reinterpret_cast<uint2 *>(smem + (m - m0) * CTS + (n - n0)) = o;
      else {
        long long row = m;
        if (g.dgrad2) {
          const int b = m / (g.Ho * g.Wo), r = m - b * (g.Ho * g.Wo);
          row = ((long long)b * g.Hy + 2 * (r / g.Wo) + g.cy) * g.Wy + 2 * (r % g.Wo) + g.cx;
        }
        if (g.gate) {
          const uint2 a = *reinterpret_cast<const uint2 *>(g.gate + row * g.Cout + n);
          o.x &= (__uint_as_float(a.x << 16) > 0.f ? 0x0000ffffu : 0u) | (__uint_as_float(a.x & 0xffff0000u) > 0.f ? 0xffff0000u : 0u);
          o.y &= (__uint_as_float(a.y << 16) > 0.f ? 0x0000ffffu : 0u) | (__uint_as_float(a.y & 0xffff0000u) > 0.f ? 0xffff0000u : 0u);
        }
        *reinterpret_cast<uint2 *>(g.Y + row * g.Cout + n) = o;
      }
    }
  }
  if (wide) {
    if (g.dgrad2) {
      const int hw = g.Ho * g.Wo;
      gemm_flush_tile_n_map<BN>(smem, g.Y, g.Cout, m0, n0, M, g.Cout, g.gate, g.Cout, [&](long long m) {
        const int b = (int)m / hw, r = (int)m - b * hw, a = r / g.Wo;
        return ((long long)b * g.Hy + 2 * a + g.cy) * g.Wy + 2 * (r - a * g.Wo) + g.cx;
      });
    } else {
      gemm_flush_tile_n<BN>(smem, g.Y, g.Cout, m0, n0, M, g.Cout, g.gate, g.Cout);
    }
  }
}

// ---- the ResNet stem: 7x7 convolution, stride 2, padding 3, 3 input channels -> 64, NHWC bf16 ----------------
//   Y[b, oy, ox, :] = sum_{ky,kx,c} X4[b, 2*oy - 3 + ky, 2*ox - 3 + kx, c] * Wp[:, ky*32 + kx*4 + c]
// X4 [B][H][W][4] bf16 = the image with its 3 channels padded to 4 (8 B per pixel, so a tap row of a pixel pair is one
// aligned 16-byte piece), Wp [64][256] bf16 = the weight (BN scale folded in) in the matching K order, zero in the padding
// slots (c = 3, kx = 7, ky = 7).  Implicit GEMM on the tile machinery above: 128 output pixels x 64 channels per
// workgroup, K = 256 in four steps of two tap rows; each wave owns 32 pixels x 64 channels (2 x 4 MFMA tiles).
// The stem is frozen in every Snipper recipe (reference backbone.py:71-73): forward only.
struct StemArgs {
  const uint16_t *X4;   // [B][H][W][4]
  const uint16_t *Wp;   // [64][256]
  uint16_t *Y;          // [B][Ho][Wo][64]
  int B, H, Wd, Ho, Wo;
};
constexpr int kStemN = 64, kStemK = 256;

__global__ __launch_bounds__(kGemmThreads) void stem7x7_bf16_kernel(StemArgs g) {
  __shared__ __attribute__((aligned(16))) uint16_t smem[(kGemmBM + kStemN) * kGemmPad];
  uint16_t *Xs = smem, *Ws = smem + kGemmBM * kGemmPad;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long long M = (long long)g.B * g.Ho * g.Wo;
  const long long m0 = (long long)blockIdx.x * kGemmBM;
  if (m0 >= M) return;
  int pb[4], py[4], px[4], lds_off[4], kyl[4], kx0[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = tid + kGemmThreads * i, row = idx >> 3, kc = idx & 7;
    const long long m = min(m0 + row, M - 1);
    pb[i] = (int)(m / (g.Ho * g.Wo));
    const int r = (int)(m - (long long)pb[i] * (g.Ho * g.Wo));
    py[i] = (r / g.Wo) * 2 - 3;
    px[i] = (r % g.Wo) * 2 - 3;
    kyl[i] = kc >> 2;               // tap row within the step (two per step)
    kx0[i] = (kc & 3) * 2;          // first tap column of this 16-byte piece (a pixel pair)
    lds_off[i] = row * kGemmPad + kc * 8;
  }
  const int wrow = tid >> 2, wkc = (tid & 3) * 2;        // weight tile: 64 rows x 8 pieces, two pieces per thread
  typedef unsigned int stem_u32x2 __attribute__((ext_vector_type(2)));
  auto load_step = [&](int s, gemm_u32x4 (&xr)[4], gemm_u32x4 (&wr)[2]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int ky = 2 * s + kyl[i];
      const int iy = py[i] + ky, ix = px[i] + kx0[i];
      const bool oky = ky < 7 && iy >= 0 && iy < g.H;
      const bool ok0 = oky && ix >= 0 && ix < g.Wd, ok1 = oky && ix + 1 >= 0 && ix + 1 < g.Wd;
      const uint16_t *base = g.X4 + (((long long)pb[i] * g.H + (oky ? iy : 0)) * g.Wd) * 4;
      const stem_u32x2 a = *reinterpret_cast<const stem_u32x2 *>(base + (long long)(ok0 ? ix : 0) * 4);
      const stem_u32x2 b = *reinterpret_cast<const stem_u32x2 *>(base + (long long)(ok1 ? ix + 1 : 0) * 4);
      xr[i] = gemm_u32x4{ok0 ? a.x : 0u, ok0 ? a.y : 0u, ok1 ? b.x : 0u, ok1 ? b.y : 0u};
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
      wr[i] = *reinterpret_cast<const gemm_u32x4 *>(g.Wp + (long long)wrow * kStemK + s * kGemmBK + (wkc + i) * 8);
  };
  gemm_u32x4 xr[4], wr[2];
  load_step(0, xr, wr);
  gemm_f32x4 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = gemm_f32x4{0.f, 0.f, 0.f, 0.f};
  const int frag_row = lane & 15, frag_k = (lane >> 4) * 8;
  for (int s = 0; s < kStemK / kGemmBK; ++s) {
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<gemm_u32x4 *>(Xs + lds_off[i]) = xr[i];
#pragma unroll
    for (int i = 0; i < 2; ++i) *reinterpret_cast<gemm_u32x4 *>(Ws + wrow * kGemmPad + (wkc + i) * 8) = wr[i];
    __syncthreads();
    if (s + 1 < kStemK / kGemmBK) load_step(s + 1, xr, wr);
#pragma unroll
    for (int kk = 0; kk < kGemmBK; kk += 32) {
      gemm_bf16x8 wf[4], xf[2];
#pragma unroll
      for (int i = 0; i < 4; ++i)
        wf[i] = *reinterpret_cast<const gemm_bf16x8 *>(Ws + (i * 16 + frag_row) * kGemmPad + kk + frag_k);
#pragma unroll
      for (int j = 0; j < 2; ++j)
        xf[j] = *reinterpret_cast<const gemm_bf16x8 *>(Xs + (wave * 32 + j * 16 + frag_row) * kGemmPad + kk + frag_k);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int n = i * 16 + (lane >> 4) * 4;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const long long m = m0 + wave * 32 + j * 16 + (lane & 15);
      if (m >= M) continue;
      uint2 o;
      o.x = gemm_pack2(acc[i][j].x, acc[i][j].y);
      o.y = gemm_pack2(acc[i][j].z, acc[i][j].w);
      *reinterpret_cast<uint2 *>(g.Y + m * kStemN + n) = o;
    }
  }
}

}  // namespace snipper
