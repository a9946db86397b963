// wgrad_conv_patch_bf16.cuh -- weight gradient of a stride-1, padding-1 3x3 convolution with the input patch RESIDENT in LDS.
//
//   dW[co][ky][kx][ci] = sum over (b, y, x) of G[b][y][x][co] * X[b][y + ky - 1][x + kx - 1][ci]      (zeros outside the image)
//
// (the 3x3 convolutions of torchvision's Bottleneck behind reference models/backbone.py:67-111; G, X bf16 NHWC, dW f32.)
//
// The conv mode of wgrad_bf16_kernel treats every tap as its own 128-column tile of a [Cout] x [9 Cin] product: a G tile is
// staged once per (tap, ci tile) and the nine shifted views of X are nine separate streams -- 0.13-0.15 of the MFMA peak.
// Here a workgroup owns a (co tile, 64-channel ci tile) for ALL NINE taps:
//
//   * the pixels of the batch are walked in PADDED raster order: position q = (b, py, px) over images of (H+2) x (W+2) with
//     a zero border.  In that space a tap is a constant displacement sh = (ky-1)(W+2) + (kx-1) -- no row wrap, no edge case:
//     border positions hold zeros (the loaders' raw-buffer offset is out of range there), so out-of-image taps, the row ends
//     and the seams between images all come out as products with zero;
//   * X positions stream through an LDS RING of 2*lead + 128 positions (lead >= W+3, the largest |sh|): each position is
//     staged ONCE per workgroup and read by nine taps; G arrives in chunks of 128 positions;
//   * per 32-position step a wave reads CB G fragments and nine X fragments (ds_read_b64_tr_b16: both operands have the
//     reduction index as their slow axis, see wgrad_bf16.cuh) for 9 * CB v_mfma_f32_16x16x32_bf16;
//   * the window of a (step, tap) starts at a UNIFORM ring index (scalar arithmetic, one conditional subtract); the first 32
//     ring rows are mirrored behind the ring's end so that a window never wraps inside a wave: a lane's address is one add;
//   * the reduction (positions) is split over S row-ranges; partial tiles go out in the accumulator order wgrad_reduce_kernel
//     already understands (a 16 x 16 block of (co, tap * Cin + ci) is a definite (tile, wave, block) slot of its 128 x 128
//     images), so the second pass, the BatchNorm-scale fold and the deterministic summation order are shared.
#pragma once
#include "wgrad_bf16.cuh"

namespace snipper {

constexpr int kWcThreads = 256, kWcChunk = 128, kWcCi = 64, kWcXStride = 80, kWcMirror = 32;   // (80 elements = 160 B rows)

struct WgradConvArgs {
  const uint16_t *G;                  // [B][H][W][Cout]
  const uint16_t *X;                  // [B][H][W][Cin]
  float *P;                           // [S][tiles128][4 waves][16 blocks][64 lanes][4] (wgrad_reduce_kernel's layout)
  int B, H, Wd, Cin, Cout;
  int S, span;                        // row-ranges and padded positions per range (a multiple of kWcChunk)
  int tiles_co, tiles_ci;             // Cout / (16 CB), Cin / 64
  int lead, ring;                     // lead % 32 == 0, lead >= W + 3; ring = 2 lead + kWcChunk
  int tiles_k128, tiles128;           // 9 Cin / 128 and ceil(Cout / 128) * tiles_k128
  unsigned mag_row, mag_img;          // ceil(2^32 / (W + 2)), ceil(2^32 / (H + 2))
  int debug;                          // timing ablations (WRONG results): 1 loads out of range, 2 no fragment reads / MFMAs, 4 no partial store, 8 no LDS stores, 16 no loads issued in the loop
};

inline int wgrad_conv_lds_bytes(int ring, int cb) {
  return ((ring + kWcMirror) * kWcXStride + kWcChunk * (16 * cb + 16)) * 2;
}

template <int CB>
__global__ __launch_bounds__(kWcThreads) __attribute__((amdgpu_waves_per_eu(2, 2)))
void wgrad_conv_patch_kernel(WgradConvArgs g) {
  constexpr int CO = 16 * CB, GSTR = CO + 16;            // G rows: 96 B (CB = 2) / 160 B (CB = 4): 24 / 40 dwords, conflict-free
  constexpr int GCH = CO / 8, GROWS = kWcThreads / GCH, GPASS = kWcChunk / GROWS;
  constexpr int XPASS = kWcChunk / 32;
  extern __shared__ __attribute__((aligned(16))) uint16_t wc_lds[];
  uint16_t *Xs = wc_lds;
  uint16_t *Gs = wc_lds + (g.ring + kWcMirror) * kWcXStride;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  const int tiles = g.tiles_co * g.tiles_ci;
  int s, t;
  if (g.S % 8 == 0) {                                    // the tiles of one row-range on one XCD (they share G and X in its L2)
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    s = xcd + 8 * (j / tiles);
    t = j % tiles;
  } else {
    s = blockIdx.x / tiles;
    t = blockIdx.x % tiles;
  }
  const int tco = t / g.tiles_ci, tci = t - tco * g.tiles_ci;
  const int co0 = tco * CO, ci0 = tci * kWcCi;
  const int Wp = g.Wd + 2, Hp = g.H + 2, Q = Hp * Wp;
  const long long total = (long long)g.B * Q;
  const long long begin = (long long)s * g.span;
  const int npos = (int)max(0LL, min(total, begin + g.span) - begin);
  const int nchunks = (npos + kWcChunk - 1) / kWcChunk;
  // positions below are SHIFTED by one (all-border) image, so that the lead before the first image is a valid coordinate
  const unsigned g0 = (unsigned)(begin + Q), x0 = g0 - (unsigned)g.lead;

  const __amdgpu_buffer_rsrc_t xsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint16_t *>(g.X + ci0), 0, (int)(((long long)g.B * g.H * g.Wd * g.Cin - ci0) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t gsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint16_t *>(g.G + co0), 0, (int)(((long long)g.B * g.H * g.Wd * g.Cout - co0) * 2), 0x00020000);
  // padded position -> byte offset of its pixel's first channel (row stride `row_bytes`), or an offset outside the descriptor
  // on the border / outside the batch.  Branch-free: "coordinate - 1 < extent" as unsigned covers both ends at once.
  auto offset_of = [&](unsigned q, unsigned row_bytes) -> unsigned {
    const unsigned gy = __umulhi(q, g.mag_row), px = q - gy * (unsigned)Wp - 1u;
    const unsigned bb = __umulhi(gy, g.mag_img), py = gy - bb * (unsigned)Hp - 1u, b1 = bb - 1u;
    const bool real = (b1 < (unsigned)g.B) & (py < (unsigned)g.H) & (px < (unsigned)g.Wd) & !(g.debug & 1);
    const unsigned pix = (b1 * (unsigned)g.H + py) * (unsigned)g.Wd + px;
    return real ? pix * row_bytes : 0x80000000u;
  };
  // Loaders.  X: 8 lanes (16-byte chunks of 64 channels) per position, 32 positions per pass; G: GCH lanes per position.  A wave's
  // 6 loads of a chunk touch 32 X positions and 32 G positions: lane L decodes ONE of them (L < 32: X slot L, else G slot L - 32)
  // and ds_bpermute hands every loader lane its position's offset -- one decode per wave and chunk instead of six (the decode
  // is ~20 vector instructions; six of them per chunk cost as much issue time as the chunk's MFMAs).
  constexpr int PW = 64 / GCH;                           // G positions per wave and pass
  const int xch = tid & 7, xrow = tid >> 3;
  const int gch = tid % GCH, grow = tid / GCH;
  const unsigned slot_pos = lane < 32 ? 32u * (lane >> 3) + 8u * wave + (lane & 7)
                                      : (unsigned)(GROWS * ((lane - 32) / PW) + PW * wave + (lane - 32) % PW);
  const unsigned slot_rb = lane < 32 ? 2u * g.Cin : 2u * g.Cout;
  const int xsel = (lane >> 3) * 4, gsel = (32 + lane / GCH) * 4;          // bpermute byte addresses of pass 0
  auto decode = [&](unsigned xq, unsigned gq) -> unsigned { return offset_of((lane < 32 ? xq : gq) + slot_pos, slot_rb); };
  auto load_x = [&](unsigned v, gemm_u32x4 (&xr)[XPASS], int passes) {
#pragma unroll
    for (int i = 0; i < XPASS; ++i) {
      if (i < passes) {
        const unsigned off = (unsigned)__builtin_amdgcn_ds_bpermute(xsel + 32 * i, (int)v) + 16u * xch;      // (stays out of range)
        xr[i] = __builtin_amdgcn_raw_buffer_load_b128(xsrc, off, 0, 0);
      }
    }
  };
  auto load_g = [&](unsigned v, gemm_u32x4 (&gr)[GPASS]) {
#pragma unroll
    for (int i = 0; i < GPASS; ++i) {
      const unsigned off = (unsigned)__builtin_amdgcn_ds_bpermute(gsel + 4 * PW * i, (int)v) + 16u * gch;
      gr[i] = __builtin_amdgcn_raw_buffer_load_b128(gsrc, off, 0, 0);
    }
  };
  auto store_x = [&](int wr, const gemm_u32x4 (&xr)[XPASS], int passes) {   // wr = ring index of the first position
    // (only a piece that starts at ring index 0 or wraps touches the 32 mirrored rows: a uniform test keeps the second store
    //  out of the common case -- a 16-byte LDS store costs 13 cycles of the store path per wave)
    const bool mirror = wr == 0 || wr + 32 * passes > g.ring;
#pragma unroll
    for (int i = 0; i < XPASS; ++i) {
      if (i < passes) {
        int idx = wr + 32 * i + xrow;
        idx = idx >= g.ring ? idx - g.ring : idx;
        *reinterpret_cast<gemm_u32x4 *>(Xs + idx * kWcXStride + xch * 8) = xr[i];
        if (mirror && idx < kWcMirror) *reinterpret_cast<gemm_u32x4 *>(Xs + (idx + g.ring) * kWcXStride + xch * 8) = xr[i];
      }
    }
  };

  gemm_f32x4 acc[9][CB];
#pragma unroll
  for (int tp = 0; tp < 9; ++tp)
#pragma unroll
    for (int i = 0; i < CB; ++i) acc[tp][i] = gemm_f32x4{0.f, 0.f, 0.f, 0.f};

  if (nchunks > 0) {
    gemm_u32x4 xr[XPASS], gr[GPASS];
    // the ring's first 2 lead positions (stream number n = shifted position x0 + n), the first chunk's piece and its G rows: all
    // loads are issued before the first store, so that the workgroup starts after ONE memory latency (lead <= 4 pieces' worth:
    // wider images stage the rest piece by piece)
    {
      gemm_u32x4 pr[2][XPASS];
      const int n_pro = 2 * g.lead, first = min(n_pro, 2 * kWcChunk);
#pragma unroll
      for (int h = 0; h < 2; ++h) load_x(decode(x0 + h * kWcChunk, 0u), pr[h], min(XPASS, max(0, (first - h * kWcChunk) / 32)));
      const unsigned v0 = decode(x0 + n_pro, g0);
      if (n_pro <= first) {
        load_x(v0, xr, XPASS);
        load_g(v0, gr);
      }
#pragma unroll
      for (int h = 0; h < 2; ++h) store_x(h * kWcChunk, pr[h], min(XPASS, max(0, (first - h * kWcChunk) / 32)));   // < ring: no wrap
      for (int n = first; n < n_pro; n += kWcChunk) {
        const int passes = min(XPASS, (n_pro - n) / 32);
        load_x(decode(x0 + n, 0u), xr, passes);
        store_x(n, xr, passes);
      }
      if (n_pro > first) {
        load_x(v0, xr, XPASS);
        load_g(v0, gr);
      }
    }

    const int grp = lane >> 4, q4 = (lane >> 2) & 3, p4 = lane & 3;
    const int gfrag_lane = ((grp * 4 + q4) * GSTR + 4 * p4) * 2;
    const int xfrag_lane = ((grp * 4 + q4) * kWcXStride + 4 * p4 + 16 * wave) * 2;
    int cb = 0;                                          // ring index of stream number c * 128
    int wr = 2 * g.lead;                                 // ring index the chunk's X piece is written at (< ring)
    for (int c = 0; c < nchunks; ++c) {
      if (!(g.debug & 8)) {
#pragma unroll
        for (int i = 0; i < GPASS; ++i)
          *reinterpret_cast<gemm_u32x4 *>(Gs + (GROWS * i + grow) * GSTR + gch * 8) = gr[i];
        store_x(wr, xr, XPASS);
      }
      __syncthreads();
      if (c + 1 < nchunks && !(g.debug & 16)) {
        const unsigned v = decode(x0 + 2 * g.lead + (c + 1) * kWcChunk, g0 + (c + 1) * kWcChunk);
        load_x(v, xr, XPASS);
        load_g(v, gr);
      }
      // fragments of step kk + 1 are read while the MFMAs of step kk run (two register sets).  The 36 window starts of the
      // chunk (4 steps x 9 taps, uniform, wrapped into the ring) are computed ONCE, one per lane, and fetched with v_readlane:
      // as scalar arithmetic they were five instructions per (step, tap) in a stream whose issue slots, not the matrix pipe,
      // bound the loop.
      int win;
      {
        const int kk_l = lane / 9, tp_l = lane - 9 * kk_l;
        int w0 = cb + g.lead + kk_l * 32 + (tp_l / 3 - 1) * Wp + (tp_l % 3 - 1);      // in [0, 2 ring) for lane < 36
        w0 = w0 >= g.ring ? w0 - g.ring : w0;
        win = w0 * (kWcXStride * 2);
      }
      auto read_frags = [&](int kk, gemm_bf16x8 (&gf)[CB], gemm_bf16x8 (&xf)[9]) {
#pragma unroll
        for (int i = 0; i < CB; ++i) gf[i] = wgrad_frag<GSTR>(Gs, gfrag_lane + (kk * 32 * GSTR + i * 16) * 2);
#pragma unroll
        for (int tp = 0; tp < 9; ++tp)
          xf[tp] = wgrad_frag<kWcXStride>(Xs, xfrag_lane + __builtin_amdgcn_readlane(win, kk * 9 + tp));
      };
      gemm_bf16x8 gf[2][CB], xf[2][9];
      if (!(g.debug & 2)) {
      read_frags(0, gf[0], xf[0]);
#pragma unroll
      for (int kk = 0; kk < kWcChunk / 32; ++kk) {
        if (kk + 1 < kWcChunk / 32) read_frags(kk + 1, gf[(kk + 1) & 1], xf[(kk + 1) & 1]);
#pragma unroll
        for (int tp = 0; tp < 9; ++tp)
#pragma unroll
          for (int i = 0; i < CB; ++i)
            acc[tp][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf[kk & 1][i], xf[kk & 1][tp], acc[tp][i], 0, 0, 0);
      }
      }
      __syncthreads();
      cb += kWcChunk; cb = cb >= g.ring ? cb - g.ring : cb;
      wr += kWcChunk; wr = wr >= g.ring ? wr - g.ring : wr;
    }
  }

  if (g.debug & 4) return;
  // partial blocks into the slots of the 128 x 128 accumulator images (see the store at the end of wgrad_bf16_kernel)
  gemm_f32x4 *Pq = reinterpret_cast<gemm_f32x4 *>(g.P) + (long long)s * g.tiles128 * 4096 + lane;
#pragma unroll
  for (int tp = 0; tp < 9; ++tp) {
    const int kc0 = tp * g.Cin + ci0 + 16 * wave;
    const int tk = kc0 >> 7, wk_r = (kc0 >> 6) & 1, j_r = (kc0 >> 4) & 3;
#pragma unroll
    for (int i = 0; i < CB; ++i) {
      const int n0 = co0 + 16 * i;
      const int tn = n0 >> 7, wn_r = (n0 >> 6) & 1, i_r = (n0 >> 4) & 3;
      Pq[((((long long)tn * g.tiles_k128 + tk) * 4 + (wn_r + 2 * wk_r)) * 16 + i_r * 4 + j_r) * 64] = acc[tp][i];
    }
  }
}

}  // namespace snipper
