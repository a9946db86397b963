// conv3x3_ring_bf16.cuh -- the 3x3 convolution of csrc/gemm_bf16.cuh (conv3x3_bf16_kernel: same arguments, same tiles, same
// arithmetic and epilogue) with the operands streamed through an LDS-DMA ring (gfx950).
//
// Reference: the 3x3 convolutions of the ResNet-50 body behind /root/reference/models/backbone.py:67-111 (torchvision
// Bottleneck.conv2, 16 of the 53 convolutions; forward, and -- with the taps flipped / per parity class -- their data gradients).
//
// Why: conv3x3_bf16_kernel keeps ONE K-step (one tap x 64 input channels) of operands in flight per workgroup (register
// prefetch, store to LDS, barrier, 16 MFMAs per wave, barrier).  Its grid only fills two workgroups per CU on the layers that
// matter (layer3: 476 tiles for 1 024 slots), so a step costs most of a memory round trip: ~3 000 cycles for 256 cycles of
// matrix work per wave -- 10-18 % of the bf16 MFMA peak (profiles/r04_backbone_roofline.csv).  Here, as in
// csrc/wgrad_ring_bf16.cuh:
//   * a ring of THREE 24 KB slots per workgroup (the step's 128 x 64 activation tile + its 64 x 64 weight tile), written by
//     LDS-DMA (buffer_load ... lds, 16 B per lane, six per thread and step): two steps are in flight behind the one being
//     multiplied, across the barriers, with a counted s_waitcnt vmcnt and ONE raw s_barrier per step;
//   * 72 KB of LDS: two workgroups per CU, which drift apart and cover each other's issue / read phases;
//   * plain 128-byte rows with the 16-byte chunks XOR-swizzled by (row >> 1) & 7 -- the DMA writes LDS linearly, so the
//     permutation is applied to the SOURCE chunk -- which makes the 16 rows of a fragment read (ds_read_b128, one 16-lane
//     group = 16 rows x the same k chunk) cover the 64 banks exactly once;
//   * a tap outside the image, a row past the last pixel and every piece of a step past the last one get an offset outside
//     the buffer descriptor: the DMA writes zeros, nothing is selected or branched on, and every step issues exactly six
//     vector-memory instructions per thread, so the counted waits are exact.
// The accumulator layout, the tile -> workgroup map (XCD bands) and the store phase (bias, ReLU, gate, staging of whole rows,
// the stride-2 data gradient's parity classes) are those of conv3x3_bf16_kernel<., 64>: results are bit-identical to it.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "gemm_bf16.cuh"

namespace snipper {

constexpr int kCrSlots = 3, kCrBN = 64;
constexpr int kCrXB = kGemmBM * 128, kCrWB = kCrBN * 128, kCrSlotB = kCrXB + kCrWB;      // 16 KB + 8 KB per step

template <bool RELU>
__global__ __launch_bounds__(kGemmThreads) __attribute__((amdgpu_waves_per_eu(2, 2)))
void conv3x3_ring_kernel(Conv3x3Args g) {
  constexpr int BN = kCrBN, NI = BN / 32, CTS = BN + 8;
  __shared__ __attribute__((aligned(16))) unsigned char smem[kCrSlots * kCrSlotB];      // ONE LDS object (see wres_gemm_bf16.cuh)
  const int bid = conv_dgrad2_class(g);
  const bool wide = gemm_wide_ok(g.Y, g.Cout, g.Cout) && (!g.gate || ((uintptr_t)g.gate % 16) == 0);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 1, wn = wave >> 1;
  const int M = g.B * g.Ho * g.Wo;
  const int tiles_n = (g.Cout + BN - 1) / BN;
  const int xcd = bid & 7, jb = bid >> 3;
  const int tm = xcd + 8 * (jb / tiles_n), tn = jb % tiles_n;
  if ((long long)tm * kGemmBM >= M) return;
  const int m0 = tm * kGemmBM, n0 = tn * BN;
  const int nty = g.dgrad2 ? (g.cy ? 2 : 1) : 3, ntx = g.dgrad2 ? (g.cx ? 2 : 1) : 3;
  const int kslices = g.Cin / kGemmBK, steps = nty * ntx * kslices;

  // ---- DMA geometry: piece i = tid + 256 j of a tile image; row r = i / 8, chunk slot i % 8 holds source chunk
  // c = (i % 8) ^ ((r >> 1) & 7).  The pixel arithmetic is conv3x3_bf16_kernel's (rows 32 j + tid / 8).
  const int lrow = tid >> 3;
  int py[4], px[4], pix[4];
  unsigned xc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = lrow + 32 * j;
    const int m = m0 + row;
    const int mm = min(m, M - 1);
    const int b = mm / (g.Ho * g.Wo);
    const int r = mm - b * (g.Ho * g.Wo);
    py[j] = g.dgrad2 ? r / g.Wo : (r / g.Wo) * g.stride - 1;
    px[j] = g.dgrad2 ? r % g.Wo : (r % g.Wo) * g.stride - 1;
    if (m >= M) py[j] = -(1 << 20);                      // a row past the last pixel: every tap is "outside the image"
    pix[j] = (b * g.H + py[j]) * g.Wd + px[j];
    xc[j] = (unsigned)(((tid & 7) ^ ((row >> 1) & 7)) * 8);
  }
  const __amdgpu_buffer_rsrc_t xsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint16_t *>(g.X), 0, (int)((long long)g.B * g.H * g.Wd * g.Cin * 2), 0x00020000);
  const int nrows = min(BN, g.Cout - n0);
  const __amdgpu_buffer_rsrc_t wsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint16_t *>(g.W + (long long)n0 * 9 * g.Cin), 0, (int)((long long)nrows * 9 * g.Cin * 2), 0x00020000);
  unsigned w_voff[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int row = lrow + 32 * j;
    w_voff[j] = ((unsigned)row * 9u * (unsigned)g.Cin + (unsigned)(((tid & 7) ^ ((row >> 1) & 7)) * 8)) * 2u;
  }
  const int lds_piece = (tid - lane) * 16;                // (+ 4096 j: 256 pieces further on)
  typedef __attribute__((address_space(3))) void lds_void;
  auto issue = [&](int s) {                               // ALWAYS six instructions
    const bool live = s < steps;
    const int t = s / kslices, k0 = (s - t * kslices) * kGemmBK;
    const int ty = t / ntx, tx = t - ty * ntx;
    const int ky = g.dgrad2 ? (g.cy ? 2 * ty : 1) : ty, kx = g.dgrad2 ? (g.cx ? 2 * tx : 1) : tx;
    const int dy = g.dgrad2 ? (g.cy + 1 - ky) / 2 : ky, dx = g.dgrad2 ? (g.cx + 1 - kx) / 2 : kx;
    const int tap = g.flip ? 8 - (ky * 3 + kx) : ky * 3 + kx;
    const int dpix = dy * g.Wd + dx;
    unsigned char *slot = smem + (s % kCrSlots) * kCrSlotB;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int iy = py[j] + dy, ix = px[j] + dx;
      const bool ok = live && iy >= 0 && iy < g.H && ix >= 0 && ix < g.Wd;
      const unsigned xoff = ok ? ((unsigned)(pix[j] + dpix) * (unsigned)g.Cin + (unsigned)k0 + xc[j]) * 2u : 0x80000000u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrc, (lds_void *)(slot + lds_piece + 4096 * j), 16, xoff, 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wsrc, (lds_void *)(slot + kCrXB + lds_piece + 4096 * j), 16,
                                               live ? w_voff[j] : 0x80000000u, live ? (unsigned)(tap * g.Cin + k0) * 2u : 0u, 0, 0);
  };

  // ---- fragment addressing: lane (row = lane & 15, k group = lane >> 4) reads 16 B at chunk (4 kk/32 + k group) ^ swizzle
  // of its row; 16-row blocks and the two waves' halves are immediate offsets.  Inline assembly, as in wgrad_ring_bf16.cuh:
  // next to a pending LDS-DMA hipcc puts vmcnt(0) before a compiler-visible LDS read.
  const int frag_row = lane & 15, g4 = lane >> 4;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem;
  const unsigned swz = (unsigned)((frag_row >> 1) & 7);
  const unsigned off_k0 = 16u * ((unsigned)g4 ^ swz), off_k1 = 16u * (((unsigned)g4 ^ 4u) ^ swz);
  const unsigned x_base = lds0 + (unsigned)((wm * 64 + frag_row) * 128), w_base = lds0 + (unsigned)(kCrXB + (wn * (BN / 2) + frag_row) * 128);
#define CR_READ(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:" #off : "=v"(dst) : "v"(addr))

  gemm_f32x4 acc[NI][4];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = gemm_f32x4{0.f, 0.f, 0.f, 0.f};

  for (int s = 0; s < kCrSlots - 1; ++s) issue(s);
  for (int s = 0; s < steps; ++s) {
    // vector-memory instructions younger than step s's six DMAs: step s + 1's six
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    __builtin_amdgcn_s_barrier();                 // step s has landed for every wave; every wave is done with step s - 1
    const unsigned sb = (unsigned)((s % kCrSlots) * kCrSlotB);
    gemm_u32x4 wf[2][NI], xf[2][4];
    {
      const unsigned a0 = w_base + sb + off_k0, a1 = w_base + sb + off_k1;
      CR_READ(wf[0][0], a0, 0); CR_READ(wf[0][1], a0, 2048);
      CR_READ(wf[1][0], a1, 0); CR_READ(wf[1][1], a1, 2048);
      const unsigned b0 = x_base + sb + off_k0, b1 = x_base + sb + off_k1;
      CR_READ(xf[0][0], b0, 0); CR_READ(xf[0][1], b0, 2048); CR_READ(xf[0][2], b0, 4096); CR_READ(xf[0][3], b0, 6144);
      CR_READ(xf[1][0], b1, 0); CR_READ(xf[1][1], b1, 2048); CR_READ(xf[1][2], b1, 4096); CR_READ(xf[1][3], b1, 6144);
    }
    __builtin_amdgcn_sched_barrier(0);
    issue(s + kCrSlots - 1);                      // into the slot of step s - 1, behind this step's LDS reads
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(gemm_bf16x8, wf[kk][i]),
                                                              __builtin_bit_cast(gemm_bf16x8, xf[kk][j]), acc[i][j], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);            // (nothing of the next step's reads above this step's last MFMA)
  }
#undef CR_READ
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the empty DMAs of the steps past the end
  __syncthreads();                                   // the staging below overwrites slot 0

  // ---- store phase: conv3x3_bf16_kernel's ----
  uint16_t *stage = reinterpret_cast<uint16_t *>(smem);
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
      if (wide) *reinterpret_cast<uint2 *>(stage + (m - m0) * CTS + (n - n0)) = o;
      else {
        long long row = m;
        if (g.dgrad2) {
          const int b2 = m / (g.Ho * g.Wo), r = m - b2 * (g.Ho * g.Wo);
          row = ((long long)b2 * g.Hy + 2 * (r / g.Wo) + g.cy) * g.Wy + 2 * (r % g.Wo) + g.cx;
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
      gemm_flush_tile_n_map<BN>(stage, g.Y, g.Cout, m0, n0, M, g.Cout, g.gate, g.Cout, [&](long long m) {
        const int b = (int)m / hw, r = (int)m - b * hw, a = r / g.Wo;
        return ((long long)b * g.Hy + 2 * a + g.cy) * g.Wy + 2 * (r - a * g.Wo) + g.cx;
      });
    } else {
      gemm_flush_tile_n<BN>(stage, g.Y, g.Cout, m0, n0, M, g.Cout, g.gate, g.Cout);
    }
  }
}

}  // namespace snipper
