// wgrad_wide_bf16.cuh -- split-reduction weight gradient on 128 x 384 output tiles, one 8-wave workgroup per CU (gfx950).
//
//   dW[N, Kc] = sum_m A[m, n] * B[m, kc]        A, B bf16 row-major over the M = B*T*S = 79 000 token rows; partials f32
//
// For the weight gradients of the encoder's dense layers (reference: every nn.Linear of
// models/ops/modules/ms_deform_attn.py:60-66 and the FFN of models/deformable_transformer.py:180-198): 384 x 384,
// 288 x 384, 1024 x 384 and 384 x 1024 outputs over 79 000 rows.  Round 4's kernels (csrc/wgrad_bf16.cuh,
// csrc/wgrad_ring_bf16.cuh) cut the output into 128 x 128 tiles: a workgroup streams 16 KB of operands per 32 rows for
// 128 x 128 x 32 products, so a 384 x 384 gradient moves its 121 MB of operands three times from L2 into LDS (363 MB; the
// FFN's 970 MB for 222 MB) -- and the per-CU rate of that path, not HBM and not the matrix pipe, is what bounds them
// (profiles/r04_backbone_roofline.csv: 0.27-0.30 of the HBM roofline; ~40 GB/s per CU against the ~65 an LDS-DMA ring
// sustains).  Here a workgroup owns a 128 x 384 tile:
//   * 32 KB per 32-row step (A: 32 x 128, B: 32 x 384) for THREE times the products: 242 MB through L2 for 384 x 384, 647 MB
//     for the FFN's;
//   * eight waves (2 x 4: 64 rows x 96 columns each = 4 x 6 v_mfma_f32_16x16x32_bf16 accumulators, 96 VGPRs), ONE workgroup
//     per CU: 128 KB of LDS = a ring of four 32 KB slots filled by LDS-DMA, two steps in flight behind the one being
//     multiplied, counted s_waitcnt vmcnt, one raw s_barrier per step;
//   * the two waves of a SIMD run half a step apart (one multiplies while the other issues DMAs and reads fragments): in
//     lockstep the phases of a step added up (see the main loop);
//   * "swapped" use: dW^T = B^T . A through the same kernel (the caller passes the operands exchanged) when that orientation
//     tiles better -- the FFN's 384 x 1024 gradient is eight 128-row tiles of dW2^T (8 tiles x 32 row ranges = 256
//     workgroups) instead of nine padded tiles of dW2; wgrad_reduce_kernel then stores the transpose.
// Partial tiles go to the workspace as 128 x 128 accumulator images in the layout of wgrad_bf16.cuh (a wide tile = three of
// them), so that kernel's deterministic second pass sums them unchanged.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "gemm_bf16.cuh"
#include "wgrad_bf16.cuh"
#include "wgrad_ring_bf16.cuh"      // wrg_swz

namespace snipper {

constexpr int kWwThreads = 512, kWwRows = 32, kWwSlots = 4;
constexpr int kWwTileN = 128, kWwTileK = 384;
constexpr int kWwARowB = 256, kWwBRowB = 768;                                     // bytes per LDS row of the A / B image
constexpr int kWwATileB = kWwRows * kWwARowB, kWwBTileB = kWwRows * kWwBRowB;     // 8 KB + 24 KB
constexpr int kWwSlotB = kWwATileB + kWwBTileB;                                   // 32 KB

struct WgradWideArgs {
  const uint16_t *A; long long lda;   // [M][N]   (rows of the output)
  const uint16_t *B; long long ldb;   // [M][Kc]  (columns of the output)
  float *P;                           // partial sums: [S][tiles_n * tiles_kv] accumulator images of 128 x 128
  float *Pb;                          // [S][N] partial column sums of A (bias_side 1) / [S][Kc] of B (bias_side 2), or nullptr
  int M, N, Kc, S, rows_per_split, tiles_n, tiles_kw;      // tiles_kw wide column tiles (384); tiles_kv = 3 tiles_kw
  int bias_side;                      // 0 none, 1 column sums of A, 2 column sums of B
  int debug;     // timing ablations (WRONG results; wres_debug()): 1 no memory reads (empty descriptors), 2 no MFMA, 4 no fragment
                 // reads, 16 no partial stores
};

__global__ __launch_bounds__(kWwThreads) __attribute__((amdgpu_waves_per_eu(2, 2))) void wgrad_wide_kernel(WgradWideArgs g) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[kWwSlots * kWwSlotB];       // ONE LDS object (see wres_gemm_bf16.cuh)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave & 1, wk = wave >> 1;                 // rows [64 wn, +64), columns [96 wk, +96) of the tile

  // workgroup -> (row range s, tile t): the tiles of a row range share an XCD (they re-read the same rows of A / B from its L2)
  const int tiles = g.tiles_n * g.tiles_kw;
  int s, t;
  if (g.S % 8 == 0) {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    s = xcd + 8 * (j / tiles);
    t = j % tiles;
  } else {
    s = blockIdx.x / tiles;
    t = blockIdx.x % tiles;
  }
  const int tn = t / g.tiles_kw, tkw = t - tn * g.tiles_kw;
  const int n0 = tn * kWwTileN, k0 = tkw * kWwTileK;
  const int m_begin = s * g.rows_per_split, m_end = min(g.M, m_begin + g.rows_per_split);
  const int n_steps = max(0, (m_end - m_begin + kWwRows - 1) / kWwRows);

  // ---- DMA geometry.  A image: piece i = tid of 32 rows x 16 chunks; B image: pieces i = tid + 512 j of 32 rows x 48
  // chunks.  Slot sl of row r holds chunk (sl & ~15) | ((sl & 15) ^ swz(r)) of the row: the XOR stays inside a 256-byte
  // block, and the DMA writes LDS linearly, so the permutation is applied to the SOURCE address.
  unsigned a_voff, b_voff[3];
  int a_piece, b_piece[3];
  {
    const int r = tid >> 4, c = (tid & 15) ^ wrg_swz(r);
    a_voff = n0 + c * 8 < g.N ? ((unsigned)r * (unsigned)g.lda + (unsigned)c * 8u) * 2u : 0x80000000u;
    a_piece = (tid - lane) * 16;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int i = tid + kWwThreads * j, rb = i / 48, sl = i - rb * 48, cb = (sl & ~15) | ((sl & 15) ^ wrg_swz(rb));
      b_voff[j] = k0 + cb * 8 < g.Kc ? ((unsigned)rb * (unsigned)g.ldb + (unsigned)cb * 8u) * 2u : 0x80000000u;
      b_piece[j] = (i - lane) * 16;
    }
  }
  // descriptors: a full 32-row step has a fixed size and its base advances by a constant -- issue() is called with
  // consecutive steps, so the bases are running scalars; only the last (short) step and the empty ones past it are
  // computed in full.  (The first version rebuilt both descriptors from (m, rows) in every call: ~60 scalar instructions
  // in front of every step's DMAs, in both waves of every SIMD.)
  const uint16_t *a_cur = g.A + (long long)m_begin * g.lda + n0, *b_cur = g.B + (long long)m_begin * g.ldb + k0;
  const long long a_adv = (long long)kWwRows * g.lda, b_adv = (long long)kWwRows * g.ldb;
  const int a_full = (int)(((long long)(kWwRows - 1) * g.lda + (g.N - n0)) * 2), b_full = (int)(((long long)(kWwRows - 1) * g.ldb + (g.Kc - k0)) * 2);
  const int n_full = (m_end - m_begin) / kWwRows;          // steps with all 32 rows
  auto issue = [&](int st) {                       // ALWAYS 4 instructions (empty descriptors past the last step)
    int abytes = a_full, bbytes = b_full;
    if (st >= n_full || (g.debug & 1)) {
      const int rows = (st < n_steps && !(g.debug & 1)) ? m_end - m_begin - st * kWwRows : 0;
      abytes = rows > 0 ? (int)(((long long)(rows - 1) * g.lda + (g.N - n0)) * 2) : 0;
      bbytes = rows > 0 ? (int)(((long long)(rows - 1) * g.ldb + (g.Kc - k0)) * 2) : 0;
    }
    const __amdgpu_buffer_rsrc_t asrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(a_cur), 0, abytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t bsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(b_cur), 0, bbytes, 0x00020000);
    if (st < n_steps) { a_cur += a_adv; b_cur += b_adv; }
    unsigned char *slot = smem + (st % kWwSlots) * kWwSlotB;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(asrc, (__attribute__((address_space(3))) void *)(slot + a_piece), 16, a_voff, 0, 0, 0);
#pragma unroll
    for (int j = 0; j < 3; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(bsrc, (__attribute__((address_space(3))) void *)(slot + kWwATileB + b_piece[j]), 16,
                                               b_voff[j], 0, 0, 0);
  };

  // ---- fragment addressing (transposing LDS read, as wgrad_ring_bf16.cuh): lane 4q + p of 16-lane group grp supplies row
  // {0, 8, 4, 12}[grp] + q (and + 16 by the second read), 8 bytes at columns 4p .. 4p + 3 of a 16-column block = chunk
  // 2 block + (p >> 1), byte 8 (p & 1).  Both images have rows of a multiple of 256 bytes, so a 32-lane half reads eight
  // 32-byte spans whose positions within the 256-byte bank row the swizzle makes distinct: conflict-free.
  const int grp = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
  const int frow = ((grp & 1) << 3) + ((grp >> 1) << 2) + q;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem;
  unsigned addr_a[4], addr_b[6];
#pragma unroll
  for (int i = 0; i < 4; ++i)
    addr_a[i] = lds0 + (unsigned)(frow * kWwARowB + 16 * ((2 * (wn * 4 + i) + (p >> 1)) ^ wrg_swz(frow)) + 8 * (p & 1));
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const int ch = 2 * (wk * 6 + j) + (p >> 1);
    addr_b[j] = lds0 + (unsigned)(kWwATileB + frow * kWwBRowB + 16 * ((ch & ~15) | ((ch & 15) ^ wrg_swz(frow))) + 8 * (p & 1));
  }
  // (inline assembly: before a compiler-visible LDS read next to a pending LDS-DMA hipcc waits vmcnt(0), which would drain
  //  the ring every step; the waits are placed by hand)
#define WW_TR_READ(dst, addr, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:" #off : "=v"(dst) : "v"(addr))
  auto join = [](wgrad_bf16x4 lo, wgrad_bf16x4 hi) { return gemm_bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]}; };

  gemm_f32x4 acc[4][6], accb[6];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 6; ++j) acc[i][j] = gemm_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 6; ++j) accb[j] = gemm_f32x4{0.f, 0.f, 0.f, 0.f};
  const __bf16 one = (__bf16)1.0f;
  const gemm_bf16x8 ones = {one, one, one, one, one, one, one, one};
  // column sums as one more MFMA row / column against a fragment of ones (exact: bf16 x 1.0 summed in f32): of A by the two
  // waves with wk == 0 of the tiles with tkw == 0 (their 4 A fragments cover the tile's 128 rows n), of B by the four waves
  // with wn == 0 of the tiles with tn == 0 (their 6 B fragments each cover the tile's 384 columns)
  const bool bias_a = g.Pb != nullptr && g.bias_side == 1 && tkw == 0 && wk == 0;
  const bool bias_b = g.Pb != nullptr && g.bias_side == 2 && tn == 0 && wn == 0;

  struct Frags { wgrad_bf16x4 a0[4], a1[4], b0[6], b1[6]; };
  auto read_frags = [&](int st, Frags &f) {
    const unsigned sb = (unsigned)((st % kWwSlots) * kWwSlotB);
    if (g.debug & 4) return;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned aa = addr_a[i] + sb;
      WW_TR_READ(f.a0[i], aa, 0); WW_TR_READ(f.a1[i], aa, 4096);          // + 16 rows x 256 B
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const unsigned ab = addr_b[j] + sb;
      WW_TR_READ(f.b0[j], ab, 0); WW_TR_READ(f.b1[j], ab, 12288);         // + 16 rows x 768 B
    }
  };
  auto multiply = [&](const Frags &f) {
    if (g.debug & 2) return;
    gemm_bf16x8 af[4], bf[6];
#pragma unroll
    for (int i = 0; i < 4; ++i) af[i] = join(f.a0[i], f.a1[i]);
#pragma unroll
    for (int j = 0; j < 6; ++j) bf[j] = join(f.b0[j], f.b1[j]);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 6; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
    if (bias_a) {
#pragma unroll
      for (int i = 0; i < 4; ++i) accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], ones, accb[i], 0, 0, 0);
    }
    if (bias_b) {
#pragma unroll
      for (int j = 0; j < 6; ++j) accb[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, bf[j], accb[j], 0, 0, 0);
    }
  };
  // ---- main loop.  The eight waves run ONE program; left in lockstep (round 5's first version: every wave issues its
  // DMAs, waits, passes the barrier, reads its fragments, multiplies) the phases of a step ADD UP -- 36 us for 79 000 x 384 x
  // 384 = 24 us with the MFMAs and fragment reads compiled out + 11 (tools/run_wgrad_ablation.sh): the two waves of a SIMD
  // want the matrix pipe at the same time and the DMA issue slots at the same time.  So the two waves of a SIMD run HALF A
  // STEP APART (MI355X_MICROARCH.md "Two waves per SIMD", items 5 and 9; the pairing csrc/wres_gemm_bf16.cuh uses): group 0
  // (waves 0-3, one per SIMD) multiplies step s while group 1 (waves 4-7) issues its DMAs and reads its fragments, then the
  // roles swap; one barrier per half-step.
  //   half-step 2s    : G0 multiply(s), then wait for its pieces of step s + 1 | G1 issue(s + 3), read(s), wait for step s + 1
  //   half-step 2s + 1: G0 issue(s + 4), read(s + 1)                           | G1 multiply(s)
  // A slot is rewritten only after both groups' reads of it lie behind a barrier (G0's issue(s + 4) -> slot of step s: read
  // by G0 in half-step 2s - 1, by G1 in 2s; G1's issue(s + 3) -> slot of step s - 1: read in 2s - 3 / 2s - 2); a step's
  // fragments are read only after every wave's counted wait for its own pieces of that step and a barrier.  The DMAs a thread
  // has in flight behind the step it waits for are two steps' = 8 instructions in both groups.
  Frags f;
  auto lds_wait = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  if (wave < 4) {
    for (int st = 0; st < kWwSlots; ++st) issue(st);
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");      // step 0 (younger: steps 1, 2, 3)
    __builtin_amdgcn_s_barrier();
    read_frags(0, f);
    lds_wait();
    for (int st = 0; st < n_steps; ++st) {
      multiply(f);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");     // step st + 1 (younger: st + 2, st + 3)
      __builtin_amdgcn_s_barrier();
      issue(st + kWwSlots);
      read_frags(st + 1, f);
      lds_wait();
      __builtin_amdgcn_s_barrier();
    }
  } else {
    for (int st = 0; st < kWwSlots - 1; ++st) issue(st);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");       // step 0 (younger: steps 1, 2)
    __builtin_amdgcn_s_barrier();
    for (int st = 0; st < n_steps; ++st) {
      issue(st + kWwSlots - 1);
      read_frags(st, f);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");     // step st + 1 (younger: st + 2, st + 3)
      lds_wait();
      __builtin_amdgcn_s_barrier();
      multiply(f);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
    }
  }
#undef WW_TR_READ
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // (the empty DMAs past the last step)

  // ---- partial tile as three 128 x 128 accumulator images in the layout wgrad_reduce_kernel reads: image v = column tile
  // / 8 of the wide tile; virtual wave = wn + 2 (column tile % 8) / 4; accumulator 4 i + (column tile % 4)
  const int tiles_kv = 3 * g.tiles_kw, tiles_v = g.tiles_n * tiles_kv;
  if (g.debug & 16) return;
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const int ct = wk * 6 + j, v = ct >> 3, vw = wn + 2 * ((ct & 7) >> 2), jj = ct & 3;
    const int tv = tn * tiles_kv + 3 * tkw + v;
    gemm_f32x4 *Pq = reinterpret_cast<gemm_f32x4 *>(g.P) + ((((long long)s * tiles_v + tv) * 4 + vw) * 16) * 64 + lane;
#pragma unroll
    for (int i = 0; i < 4; ++i) Pq[(i * 4 + jj) * 64] = acc[i][j];
  }
  if (bias_a && (lane & 15) == 0) {                // D[row n][any column] = sum_m A[m][n]: take column 0
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int n = n0 + wn * 64 + i * 16 + (lane >> 4) * 4;
      const float vv[4] = {accb[i].x, accb[i].y, accb[i].z, accb[i].w};
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (n + r < g.N) g.Pb[(long long)s * g.N + n + r] = vv[r];
    }
  }
  if (bias_b && lane < 16) {                       // D[any row][column kc] = sum_m B[m][kc]: take row 0 (lanes 0-15, reg 0)
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int kc = k0 + (wk * 6 + j) * 16 + lane;
      if (kc < g.Kc) g.Pb[(long long)s * g.Kc + kc] = accb[j].x;
    }
  }
}

}  // namespace snipper
