// wgrad_bf16.cuh -- weight (and bias) gradient of the path's dense layers on gfx950 MFMA.
//
//   dW[N, Kc] = sum_m G[m, n] * X[m, kc]        db[N] = sum_m G[m, n]        G, X bf16 row-major; dW, db f32
//
// G is the gradient of a Linear's (or NHWC 1x1 convolution's) output and X its input; M = B*T*S = 79 000 tokens
// for the encoder's projections while N, Kc <= 1024: the reduction runs over the LONG axis and the output is tiny.
// A library GEMM tiles the output (9 tiles of 128 x 128 for a 384 x 384 weight) and so keeps a handful of CUs busy
// for ~0.3 ms; here the reduction axis is split over the whole chip instead:
//
//   * grid = output tiles x S row-ranges; each workgroup streams its range of G[:, 128 cols] and X[:, 128 cols]
//     once, accumulates a 128 x 128 f32 tile in MFMA accumulators (4 waves x 4 x 4 v_mfma_f32_16x16x32_bf16), and
//     writes it to a [S][N][Kc] partial buffer; a second kernel sums the S partials in a fixed order (deterministic,
//     unlike atomics).  The workgroups of one row-range are placed on one XCD so that the re-reads of G / X by the
//     other output tiles hit in that XCD's L2.
//   * both operands have the reduction index m as their SLOW axis, but an MFMA lane needs 8 consecutive reduction
//     elements of one output row.  The tiles are stored in LDS exactly as they arrive ([m][n], coalesced 16-byte
//     loads) and read back with ds_read_b64_tr_b16, gfx950's transposing LDS read: a 16-lane group fetches a
//     4 (m) x 16 (n) block and every lane receives one column = 4 reduction elements of its row; two reads make one
//     operand.  Which m lands in which reduction slot is irrelevant as long as G and X agree, so lane group g takes
//     rows {4g..4g+3} and {16+4g..16+4g+3} of each 32-row step: the two groups of a 32-lane half then read 8
//     CONSECUTIVE rows, which with a row stride of 288 B (128 columns + 16 pad) covers the 64 banks exactly once.
//   * the bias gradient is the column sum of the same G tile: the workgroups of the first output-tile column add up
//     the values they stage anyway.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "gemm_bf16.cuh"

namespace snipper {

typedef __attribute__((__vector_size__(4 * sizeof(__bf16)))) __bf16 wgrad_bf16x4;

constexpr int kWgTile = 128, kWgRows = 64, kWgStride = 144, kWgThreads = 256;    // (128-row steps: 2 waves/SIMD, measured slower)
constexpr int kWgLoads = kWgRows / 16;      // 16-byte loads per operand, thread and step

struct WgradArgs {
  const uint16_t *G; long long ldg;   // [M][N]
  const uint16_t *X; long long ldx;   // [M][Kc]
  float *P;                           // [S][N][Kc] partial sums
  float *Pb;                          // [S][N] partial column sums of G, or nullptr
  int M, N, Kc, S, rows_per_split, tiles_n, tiles_k;
  // 3x3-convolution mode (conv != 0): X is the convolution's INPUT [B][H][W][Cin] (NHWC), G its output gradient
  // [B][Ho][Wo][N]; the reduction row m = (b, oy, ox) pairs G[m] with the input pixel under tap (ky, kx) of that output
  // pixel (zeros outside the image), and column kc of dW is (tap, ci): dW = [N][3][3][Cin], the channels_last layout of a
  // convolution weight.  Kc = 9 * Cin, Cin % 128 == 0 (a 128-column tile never straddles two taps).
  int conv, H, Wd, Cin, Ho, Wo, stride;
};

template <int STRIDE>
__device__ __forceinline__ gemm_bf16x8 wgrad_frag(const uint16_t *tile, int byte_off) {
  typedef __attribute__((address_space(3))) wgrad_bf16x4 lds_v4;
  const char *base = reinterpret_cast<const char *>(tile) + byte_off;
  const wgrad_bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4 *)(base));
  const wgrad_bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4 *)(base + 16 * STRIDE * 2));
  return gemm_bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

// WK = columns of X (of dW) per workgroup: 128 is the only instantiation.  WK = 64 (each wave 64 x 32 of the tile, 104
// VGPRs, 4 waves per SIMD; the G column tile re-read by twice as many workgroups) was built, is parity-green, and made the
// training step SLOWER at every grid size tried: 27.78 / 27.65 / 27.90 ms at 512 / 768 / 1024 workgroups against 27.47 ms.
template <int WK>
__global__ __launch_bounds__(kWgThreads) __attribute__((amdgpu_waves_per_eu(WK == 128 ? 3 : 4, WK == 128 ? 3 : 4)))
void wgrad_bf16_kernel(WgradArgs g) {
  constexpr int XSTRIDE = WK + 16;                 // 288 / 160-byte rows: conflict-free transposed reads either way
  constexpr int XCH = WK / 8;                      // 16-byte chunks per X row
  constexpr int XROWS = kWgThreads / XCH;          // X rows per loader pass
  constexpr int XLOADS = kWgRows / XROWS;          // X loads per thread and step
  constexpr int NJ = WK / 32;                      // 16-column blocks of X per wave
  __shared__ __attribute__((aligned(16))) uint16_t Gs[kWgRows * kWgStride];
  __shared__ __attribute__((aligned(16))) uint16_t Xs[kWgRows * XSTRIDE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave & 1, wk = wave >> 1;

  // workgroup -> (row-range s, output tile): consecutive workgroups go round-robin over the 8 XCDs, so the
  // workgroups with the same (id % 8) run on one XCD; all output tiles of a row-range are given to one XCD
  const int tiles = g.tiles_n * g.tiles_k, total = tiles * g.S;
  int s, t;
  if (g.S % 8 == 0) {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    s = xcd + 8 * (j / tiles);
    t = j % tiles;
  } else {
    s = blockIdx.x / tiles;
    t = blockIdx.x % tiles;
  }
  (void)total;
  const int tn = t / g.tiles_k, tk = t % g.tiles_k;
  const int n0 = tn * kWgTile, k0 = tk * WK;
  const int m_begin = s * g.rows_per_split, m_end = min(g.M, m_begin + g.rows_per_split);
  const bool do_bias = g.Pb != nullptr && tk == 0;

  // loader: 4 x 16 B per operand per thread and step; row = idx / 16, 8-column chunk = idx % 16 (fixed per thread)
  const int chunk = tid & 15, row0 = tid >> 4;                 // G: 16 chunks per row, 16 rows per pass
  const int xchunk = tid % XCH, xrow0 = tid / XCH;             // X: XCH chunks per row, XROWS rows per pass
  const bool g_col_ok = n0 + chunk * 8 < g.N, x_col_ok = k0 + xchunk * 8 < g.Kc;
  const int tap = g.conv ? k0 / g.Cin : 0, ky = tap / 3 - 1, kx = tap - (tap / 3) * 3 - 1;     // (conv mode)
  const unsigned conv_col = (unsigned)((g.conv ? k0 - tap * g.Cin : k0) + xchunk * 8);
  // G (and X outside conv mode) by raw buffer loads: the descriptor covers this workgroup's row range and column tile
  // (base = its first element, a uniform value), the per-lane offset is one 32-bit register per operand (+ the row
  // group and the step's row offset, added per load -- they must be part of the VECTOR offset, the only one the range
  // check sees), rows past the range and lanes whose columns lie past the matrix read as 0 without selects.
  const int nrows = max(m_end - m_begin, 1);
  const __amdgpu_buffer_rsrc_t gsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint16_t *>(g.G + (long long)m_begin * g.ldg + n0), 0,
      (int)(((long long)(nrows - 1) * g.ldg + (g.N - n0)) * 2), 0x00020000);
  // (conv mode: X is the [B][H][W][Cin] activation, < 2 GiB by the launcher's check; the descriptor covers all of it and a
  //  tap outside the image gets an offset outside the descriptor)
  const __amdgpu_buffer_rsrc_t xsrc = g.conv
      ? __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(g.X), 0, (int)((long long)(g.M / (g.Ho * g.Wo)) * g.H * g.Wd * g.ldx * 2), 0x00020000)
      : __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(g.X + (long long)m_begin * g.ldx + k0), 0,
                                          (int)(((long long)(nrows - 1) * g.ldx + (g.Kc - k0)) * 2), 0x00020000);
  const unsigned g_voff = g_col_ok ? ((unsigned)row0 * (unsigned)g.ldg + chunk * 8) * 2u : 0x80000000u;
  const unsigned x_voff = x_col_ok ? ((unsigned)xrow0 * (unsigned)g.ldx + xchunk * 8) * 2u : 0x80000000u;
  const unsigned g_grp = 16u * (unsigned)g.ldg * 2u, x_grp = (unsigned)XROWS * (unsigned)g.ldx * 2u;
  auto load_step = [&](int m, gemm_u32x4 (&gr)[kWgLoads], gemm_u32x4 (&xr)[XLOADS]) {
    const unsigned mrel = (unsigned)(m - m_begin);
#pragma unroll
    for (int i = 0; i < kWgLoads; ++i)
      gr[i] = __builtin_amdgcn_raw_buffer_load_b128(gsrc, g_voff + i * g_grp + mrel * (unsigned)g.ldg * 2u, 0, 0);
#pragma unroll
    for (int i = 0; i < XLOADS; ++i) {
      if (g.conv) {
        const int r = m + xrow0 + XROWS * i;
        const bool ok = r < m_end;
        const int rr = ok ? r : m_begin;
        const int b = rr / (g.Ho * g.Wo), rem = rr - b * (g.Ho * g.Wo);
        const int oy = rem / g.Wo, ox = rem - oy * g.Wo;
        const int iy = oy * g.stride + ky, ix = ox * g.stride + kx;
        const bool in = ok && x_col_ok && iy >= 0 && iy < g.H && ix >= 0 && ix < g.Wd;
        const unsigned xo = in ? ((unsigned)((b * g.H + iy) * g.Wd + ix) * (unsigned)g.ldx + conv_col) * 2u : 0x80000000u;
        xr[i] = __builtin_amdgcn_raw_buffer_load_b128(xsrc, xo, 0, 0);
      } else {
        xr[i] = __builtin_amdgcn_raw_buffer_load_b128(xsrc, x_voff + i * x_grp + mrel * (unsigned)g.ldx * 2u, 0, 0);
      }
    }
  };
  gemm_u32x4 gr[kWgLoads], xr[XLOADS];
  if (m_begin < m_end) load_step(m_begin, gr, xr);
  gemm_f32x4 acc[4][NJ];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = gemm_f32x4{0.f, 0.f, 0.f, 0.f};
  float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  // transposed-read addressing: lane 4q+p of a 16-lane group supplies row q, columns 4p..4p+3 of the block
  const int grp = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
  const int frag_base = ((grp * 4 + q) * kWgStride + 4 * p) * 2;      // bytes; + 32-row step + column offset
  const int xfrag_base = ((grp * 4 + q) * XSTRIDE + 4 * p) * 2;

  for (int m = m_begin; m < m_end; m += kWgRows) {
#pragma unroll
    for (int i = 0; i < kWgLoads; ++i)
      *reinterpret_cast<gemm_u32x4 *>(Gs + (row0 + 16 * i) * kWgStride + chunk * 8) = gr[i];
#pragma unroll
    for (int i = 0; i < XLOADS; ++i)
      *reinterpret_cast<gemm_u32x4 *>(Xs + (xrow0 + XROWS * i) * XSTRIDE + xchunk * 8) = xr[i];
    if (do_bias) {
#pragma unroll
      for (int i = 0; i < kWgLoads; ++i) {
        const unsigned w[4] = {gr[i].x, gr[i].y, gr[i].z, gr[i].w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          bsum[2 * c] += __uint_as_float(w[c] << 16);
          bsum[2 * c + 1] += __uint_as_float(w[c] & 0xffff0000u);
        }
      }
    }
    __syncthreads();
    if (m + kWgRows < m_end) load_step(m + kWgRows, gr, xr);
#pragma unroll
    for (int kk = 0; kk < kWgRows; kk += 32) {
      gemm_bf16x8 gf[4], xf[NJ];
#pragma unroll
      for (int i = 0; i < 4; ++i) gf[i] = wgrad_frag<kWgStride>(Gs, frag_base + (kk * kWgStride + wn * 64 + i * 16) * 2);
#pragma unroll
      for (int j = 0; j < NJ; ++j) xf[j] = wgrad_frag<XSTRIDE>(Xs, xfrag_base + (kk * XSTRIDE + wk * (WK / 2) + j * 16) * 2);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf[i], xf[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }

  // partial tile, in ACCUMULATOR order: the workspace is scratch, so each lane stores its 16 x 4 floats as 16 coalesced
  // 16-byte pieces (one KB per wave-instruction) and wgrad_reduce_kernel undoes the permutation.  Written as dW is laid
  // out ([n][kc], 4-byte pieces of 4 rows per instruction) the store epilogue alone took 9 us of a 40 us launch (round 3,
  // profiles/r03_wgrad_ring_experiment.json).  Position: ((((s * tiles + t) * 4 + wave) * 16 + 4 i + j) * 64 + lane) * 4 + reg.
  static_assert(NJ == 4, "the partial layout assumes 4 x 4 accumulator tiles per wave");
  gemm_f32x4 *Pq = reinterpret_cast<gemm_f32x4 *>(g.P) + ((((long long)s * tiles + t) * 4 + wave) * 16) * 64 + lane;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) Pq[(i * 4 + j) * 64] = acc[i][j];
  if (do_bias) {                                   // 16 row-groups x 128 columns -> 128 column sums
    float *red = reinterpret_cast<float *>(Gs);    // 8 KB of the 18 KB tile; all reads of it are behind a barrier
#pragma unroll
    for (int c = 0; c < 8; ++c) red[row0 * kWgTile + chunk * 8 + c] = bsum[c];
    __syncthreads();
    if (tid < kWgTile && n0 + tid < g.N) {
      float sum = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) sum += red[r * kWgTile + tid];
      g.Pb[(long long)s * g.N + n0 + tid] = sum;
    }
  }
}

// out[n][kc] (+)= scale[n] * sum_s P[s][...];  db[n] (+)= sum_s Pb[s][n].  Fixed summation order.  The partials lie in
// accumulator order (see the store at the end of wgrad_bf16_kernel): quad q of a partial = (tile t, wave, 4 i + j, lane) holds
// rows n .. n + 3 of ONE column kc.
struct WgradReduceArgs {
  const float *P; const float *Pb;
  float *dW; long long lddw;     // [N][Kc] with leading dimension
  float *db;                     // [N] or nullptr
  const float *scale;            // [N] or nullptr: per-output-row factor (the folded BatchNorm scale)
  int N, Kc, S, accumulate, tiles_n, tiles_k;
  // transpose != 0 (csrc/wgrad_wide_bf16.cuh, "swapped" use): the partials hold dW^T -- N / Kc / tiles describe THAT matrix,
  // element (n, kc) of it is stored at dW[kc][n], `scale` is indexed by kc.  bias_len: length of db and row stride of Pb
  // (0 = N; the swapped use sums the columns of the other operand: Kc).
  int transpose, bias_len;
};

// 256 threads = 64 quads x 4 slices of the S partials; the slices meet in LDS
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(WgradReduceArgs g) {
  __shared__ gemm_f32x4 red[4][64];
  const long long total = (long long)g.tiles_n * g.tiles_k * 16384;        // floats per partial
  const int qi = threadIdx.x & 63, slice = threadIdx.x >> 6;
  const long long q = (long long)blockIdx.x * 64 + qi, e = q * 4;
  gemm_f32x4 sum = {0.f, 0.f, 0.f, 0.f};
  if (e < total) {
    const float *p = g.P + e;
    // 8 partials of this slice in flight per thread (2.25 workgroups per CU at 384 x 384: the kernel is bound by the loads
    // each wave keeps outstanding, not by bandwidth); the summation order stays fixed: s ascending within the slice
    int s = slice;
    for (; s + 28 < g.S; s += 32) {
      gemm_f32x4 v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<const gemm_f32x4 *>(p + (long long)(s + 4 * i) * total);
#pragma unroll
      for (int i = 0; i < 8; ++i) sum += v[i];
    }
    for (; s < g.S; s += 4) sum += *reinterpret_cast<const gemm_f32x4 *>(p + (long long)s * total);
  }
  red[slice][qi] = sum;
  __syncthreads();
  if (slice == 0 && e < total) {
    sum = (red[0][qi] + red[1][qi]) + (red[2][qi] + red[3][qi]);
    const int lane = (int)(q & 63), ij = (int)(q >> 6) & 15, wave = (int)(q >> 10) & 3, t = (int)(q >> 12);
    const int tn = t / g.tiles_k, tk = t - tn * g.tiles_k;
    const int n = tn * 128 + (wave & 1) * 64 + (ij >> 2) * 16 + (lane >> 4) * 4;
    const int kc = tk * 128 + (wave >> 1) * 64 + (ij & 3) * 16 + (lane & 15);
    if (g.transpose) {
      if (kc < g.Kc && n < g.N) {                    // rows n .. n + 3 of dW^T are 4 consecutive elements of row kc of dW
        float *dst = g.dW + (long long)kc * g.lddw + n;
        const float sc = g.scale ? g.scale[kc] : 1.f;
        if (n + 3 < g.N) {
          gemm_f32x4 o = {sum.x * sc, sum.y * sc, sum.z * sc, sum.w * sc};
          if (g.accumulate) o += *reinterpret_cast<const gemm_f32x4 *>(dst);
          *reinterpret_cast<gemm_f32x4 *>(dst) = o;
        } else {
          const float v[4] = {sum.x, sum.y, sum.z, sum.w};
          for (int r = 0; r < 4 && n + r < g.N; ++r) dst[r] = g.accumulate ? dst[r] + v[r] * sc : v[r] * sc;
        }
      }
    } else if (kc < g.Kc) {
      const float v[4] = {sum.x, sum.y, sum.z, sum.w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (n + r < g.N) {
          float o = g.scale ? v[r] * g.scale[n + r] : v[r];
          float *dst = g.dW + (long long)(n + r) * g.lddw + kc;
          if (g.accumulate) o += *dst;
          *dst = o;
        }
      }
    }
  }
  // bias gradient: column n = 64 blockIdx + qi, the S partial sums split over the 4 slices like the tiles above.  (One
  // thread per column walking all S partials serially -- 57 dependent round trips in two workgroups -- made this 12-KB
  // reduction the longest thing in the launch: 19.8 us of kernel time at 384 x 384, S = 57.)
  if (g.db && g.Pb) {
    const long long n = (long long)blockIdx.x * 64 + qi;
    const int nb = g.bias_len > 0 ? g.bias_len : g.N;
    float bs = 0.f;
    if (n < nb) {
      int s = slice;
      for (; s + 28 < g.S; s += 32) {
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = g.Pb[(long long)(s + 4 * i) * nb + n];
#pragma unroll
        for (int i = 0; i < 8; ++i) bs += v[i];
      }
      for (; s < g.S; s += 4) bs += g.Pb[(long long)s * nb + n];
    }
    __syncthreads();                               // (the tile sums above have been read out of `red`)
    red[slice][qi].x = bs;
    __syncthreads();
    if (slice == 0 && n < nb) {
      const float tot = (red[0][qi].x + red[1][qi].x) + (red[2][qi].x + red[3][qi].x);
      g.db[n] = g.accumulate ? g.db[n] + tot : tot;
    }
  }
}

}  // namespace snipper
