// wgrad_ring_bf16.cuh -- split-reduction weight gradient with the operands streamed through an LDS-DMA ring (gfx950).
//
//   dW[N, Kc] = sum_m G[m, n] * X[m, kc]        db[N] = sum_m G[m, n]        G, X bf16 row-major; partial sums f32
//
// The same decomposition as csrc/wgrad_bf16.cuh (output tiles of 128 x 128 x S row ranges, two workgroups per CU, partial
// tiles summed by wgrad_reduce_kernel in a fixed order: deterministic), for the dense layers on the encoder's 79 000 token
// rows (reference: the weight gradients of every nn.Linear in models/ops/modules/ms_deform_attn.py:60-66 and
// models/deformable_transformer.py:180-198).  What changes is how the operands reach the MFMAs.  That kernel keeps ONE
// 64-row step of G / X in flight per workgroup (register prefetch): every step pays most of a memory round trip (measured
// 40 us for 384 x 384 at M = 79 000 = 22 steps of 1.8 us per workgroup, the matrix pipe 27 % busy; the HBM floor is 19 us).
// Here:
//   * a ring of FOUR 16 KB slots per workgroup (32 rows of the G tile + 32 rows of the X tile): three steps are in flight
//     per workgroup, issued by LDS-DMA (buffer_load ... lds, 16 B per lane) with a counted s_waitcnt vmcnt and ONE raw
//     s_barrier per step (the barrier that publishes step s also frees the slot of step s - 1);
//   * TWO workgroups per CU (64 KB of LDS each): a first version with one 4-wave workgroup per CU was slower than the
//     register-prefetch kernel (49 us) although memory was not its limiter -- with one wave per SIMD the MFMA block, the
//     LDS-DMA issues (~60 cycles each for the issuing wave) and the LDS reads of a step are one serial instruction stream
//     (profiles/r03_wgrad_ring_experiment.json); two workgroups drift apart and cover each other's non-matrix phases;
//   * the tiles lie in LDS as plain 256-byte rows with the 16-byte chunks XOR-swizzled by the row (the DMA writes LDS
//     linearly, so the permutation is applied to the SOURCE address): chunk c of row r sits at chunk slot
//     c ^ (((r & 3) << 2) | ((r >> 2) & 3)).  The fragments are read with ds_read_b64_tr_b16, lane group g taking rows
//     {0, 8, 4, 12}[g] + q (and + 16): the two groups of a 32-lane half read blocks 8 rows apart, which with this swizzle
//     covers the 64 banks exactly once (conflict-free; the reduction order is free as long as G and X agree);
//   * the bias gradient is one more MFMA column: G^T times a fragment of ones (exact: bf16 x 1.0 summed in f32).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "gemm_bf16.cuh"
#include "wgrad_bf16.cuh"

namespace snipper {

constexpr int kWrgThreads = 256, kWrgRows = 32, kWrgTile = 128, kWrgSlots = 4;
constexpr int kWrgRowB = 256, kWrgTileB = kWrgRows * kWrgRowB, kWrgSlotB = 2 * kWrgTileB;      // 8 KB per operand, 16 KB per slot
constexpr int kWrgPerThread = kWrgRows * 16 / kWrgThreads;                                     // 2 DMAs per operand, thread and step

struct WgradRingArgs {
  const uint16_t *G; long long ldg;   // [M][N]
  const uint16_t *X; long long ldx;   // [M][Kc]
  float *P;                           // [S][N][Kc] partial sums
  float *Pb;                          // [S][N] partial column sums of G, or nullptr
  int M, N, Kc, S, rows_per_split, tiles_n, tiles_k;
  int debug;     // timing ablations (WRONG results): 1 no memory reads (empty descriptors), 2 no MFMA, 8 no DMA instructions
};

__device__ __forceinline__ int wrg_swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

__global__ __launch_bounds__(kWrgThreads) __attribute__((amdgpu_waves_per_eu(2, 2))) void wgrad_ring_kernel(WgradRingArgs g) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[kWrgSlots * kWrgSlotB];      // ONE LDS object (see wres_gemm_bf16.cuh)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave & 1, wk = wave >> 1;

  // workgroup -> (row range s, output tile t) as wgrad_bf16_kernel: the tiles of a row range share an XCD
  const int tiles = g.tiles_n * g.tiles_k;
  int s, t;
  if (g.S % 8 == 0) {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    s = xcd + 8 * (j / tiles);
    t = j % tiles;
  } else {
    s = blockIdx.x / tiles;
    t = blockIdx.x % tiles;
  }
  const int tn = t / g.tiles_k, tk = t - tn * g.tiles_k;
  const int n0 = tn * kWrgTile, k0 = tk * kWrgTile;
  const int m_begin = s * g.rows_per_split, m_end = min(g.M, m_begin + g.rows_per_split);
  const int n_steps = (m_end - m_begin + kWrgRows - 1) / kWrgRows;
  const bool do_bias = g.Pb != nullptr && tk == 0;

  // ---- DMA geometry: piece i = tid + 256 j of a 32 x 16-chunk tile image; row r = i / 16, slot sl = i % 16 holds chunk
  // c = sl ^ swz(r) of the row.  Rows past the range and chunks past the matrix get an offset outside the descriptor (0).
  unsigned g_voff[kWrgPerThread], x_voff[kWrgPerThread];
  int lds_piece[kWrgPerThread];
#pragma unroll
  for (int j = 0; j < kWrgPerThread; ++j) {
    const int i = tid + kWrgThreads * j, r = i >> 4, c = (i & 15) ^ wrg_swz(r);
    g_voff[j] = n0 + c * 8 < g.N ? ((unsigned)r * (unsigned)g.ldg + (unsigned)c * 8u) * 2u : 0x80000000u;
    x_voff[j] = k0 + c * 8 < g.Kc ? ((unsigned)r * (unsigned)g.ldx + (unsigned)c * 8u) * 2u : 0x80000000u;
    lds_piece[j] = (i - lane) * 16;
  }
  auto issue = [&](int st) {                       // ALWAYS 4 instructions (empty descriptors past the last step)
    const int m = m_begin + st * kWrgRows;
    const int rows = (st < n_steps && !(g.debug & 1)) ? min(kWrgRows, m_end - m) : 0;
    if (g.debug & 8) return;
    const __amdgpu_buffer_rsrc_t gsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint16_t *>(g.G + (long long)(rows > 0 ? m : 0) * g.ldg + n0), 0,
        rows > 0 ? (int)(((long long)(rows - 1) * g.ldg + (g.N - n0)) * 2) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t xsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint16_t *>(g.X + (long long)(rows > 0 ? m : 0) * g.ldx + k0), 0,
        rows > 0 ? (int)(((long long)(rows - 1) * g.ldx + (g.Kc - k0)) * 2) : 0, 0x00020000);
    unsigned char *slot = smem + (st % kWrgSlots) * kWrgSlotB;
#pragma unroll
    for (int j = 0; j < kWrgPerThread; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(gsrc, (__attribute__((address_space(3))) void *)(slot + lds_piece[j]), 16, g_voff[j], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < kWrgPerThread; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrc, (__attribute__((address_space(3))) void *)(slot + kWrgTileB + lds_piece[j]), 16,
                                               x_voff[j], 0, 0, 0);
  };

  // ---- fragment addressing (transposing LDS read): lane 4q + p of 16-lane group grp supplies row rowmap[grp] + q (+ 16),
  // 8 bytes at columns 4p .. 4p + 3 of a 16-column block: chunk 2 coltile + (p >> 1), byte 8 (p & 1) within it.  The swizzle
  // term only looks at row bits 0-3, which the + 16 (second read) leaves alone: ONE address register per 16-column block.
  // The reads are inline assembly: before a compiler-visible LDS read next to a pending LDS-DMA hipcc waits vmcnt(0),
  // which would drain the ring every step; the waits are placed by hand (with the sched_barrier hipcc needs to respect them).
  const int grp = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
  const int frow = ((grp & 1) << 3) + ((grp >> 1) << 2) + q;          // {0, 8, 4, 12}[grp] + q
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem;
  unsigned addr_g[4], addr_x[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    addr_g[i] = lds0 + (unsigned)(frow * kWrgRowB + 16 * ((2 * (wn * 4 + i) + (p >> 1)) ^ wrg_swz(frow)) + 8 * (p & 1));
    addr_x[i] = lds0 + (unsigned)(kWrgTileB + frow * kWrgRowB + 16 * ((2 * (wk * 4 + i) + (p >> 1)) ^ wrg_swz(frow)) + 8 * (p & 1));
  }
#define WRG_TR_READ(dst, addr, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:" #off : "=v"(dst) : "v"(addr))
  auto join = [](wgrad_bf16x4 lo, wgrad_bf16x4 hi) { return gemm_bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]}; };

  gemm_f32x4 acc[4][4], accb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    accb[i] = gemm_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = gemm_f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const __bf16 one = (__bf16)1.0f;
  const gemm_bf16x8 ones = {one, one, one, one, one, one, one, one};
  const bool bias_wave = do_bias && wk == 0;       // the two waves with wk == 0 cover the tile's 128 rows n

  if (g.debug & 32) return;
  for (int st = 0; st < kWrgSlots - 1; ++st) issue(st);
  for (int st = 0; st < n_steps; ++st) {
    // vector-memory instructions younger than step st's four DMAs: steps st + 1 and st + 2 = 8
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();                 // step st has landed for every wave; every wave is done with step st - 1
    const unsigned sb = (unsigned)((st % kWrgSlots) * kWrgSlotB);
    wgrad_bf16x4 g0[4], g1[4], x0[4], x1[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {                  // G fragments first: the bias product can start on them
      const unsigned ag = addr_g[i] + sb;
      WRG_TR_READ(g0[i], ag, 0); WRG_TR_READ(g1[i], ag, 4096);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned ax = addr_x[i] + sb;
      WRG_TR_READ(x0[i], ax, 0); WRG_TR_READ(x1[i], ax, 4096);
    }
    __builtin_amdgcn_sched_barrier(0);
    issue(st + kWrgSlots - 1);                    // into the slot of step st - 1, behind this step's LDS reads
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (!(g.debug & 2)) {
      gemm_bf16x8 gf[4], xf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { gf[i] = join(g0[i], g1[i]); xf[i] = join(x0[i], x1[i]); }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf[i], xf[j], acc[i][j], 0, 0, 0);
      if (bias_wave) {
#pragma unroll
        for (int i = 0; i < 4; ++i) accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf[i], ones, accb[i], 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);             // (nothing of the next step's reads above this step's last MFMA)
  }
#undef WRG_TR_READ

  // partial tile in accumulator order, 16 coalesced 16-byte stores per lane (layout and reason: wgrad_bf16_kernel)
  if (g.debug & 16) return;
  gemm_f32x4 *Pq = reinterpret_cast<gemm_f32x4 *>(g.P) + ((((long long)s * tiles + t) * 4 + wave) * 16) * 64 + lane;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) Pq[(i * 4 + j) * 64] = acc[i][j];
  if (bias_wave && (lane & 15) == 0) {             // every column of the ones product holds the sum: take column 0
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int n = n0 + wn * 64 + i * 16 + (lane >> 4) * 4;
      const float v[4] = {accb[i].x, accb[i].y, accb[i].z, accb[i].w};
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (n + r < g.N) g.Pb[(long long)s * g.N + n + r] = v[r];
    }
  }
}

}  // namespace snipper
