// wres_gemm_bf16.cuh -- weight-stationary bf16 MFMA kernel for the encoder's short-reduction dense layers (gfx950).
//
//   Y[M, N] = gate( dropout( act( X[M, K] . W^T + bias ) ) )     K = 32 * KS <= 384,  M = tens of thousands of token rows
//
// Reference shapes: every nn.Linear of MSDeformAttn (models/ops/modules/ms_deform_attn.py:60-66: value / output / offset /
// weight projections, 384 -> 384 / 288), linear1 of the encoder FFN (models/deformable_transformer.py:180-198, 384 -> 1024)
// and the data gradients whose REDUCTION is that short axis (dX = dY . W for the 384- and 288-wide outputs, and the FFN's
// hidden gradient dH = dZ . W2 with the ReLU / dropout gate), on M = B*T*S = 79 000 rows.
//
// Why another kernel.  With K = 384 a 128 x 128 output tile has six K-steps; csrc/gemm_bf16.cuh keeps one K-step of
// operands in flight per workgroup, so every step pays a memory round trip (profiles/r02_gemm_phase_ablation.json: the
// phases of a tile are additive, 44 us for 79 000 x 384 x 384 against a 19 us HBM floor), and X is read once per 128 output
// columns.  Here the roles are turned round:
//   * W is STATIONARY IN REGISTERS.  A workgroup of 8 waves covers up to 384 output columns, wave w the 48 columns
//     [48w, 48w + 48): its W slice, 48 x K bf16 = 36 KB, is 3 x KS MFMA A-fragments = 144 VGPRs, loaded once per launch.
//     There is no W traffic in the main loop at all (not from L2, not from LDS).
//   * X is streamed through LDS by LDS-DMA (buffer_load ... lds) in chunks of 32 rows x K (24 KB), a ring of 3-4 slots per
//     workgroup, ONE workgroup per CU walking a strided list of chunks: 48-72 KB of HBM reads are in flight per CU at any
//     time, across the barriers (counted s_waitcnt vmcnt, raw s_barrier).  X is read from HBM exactly once per 384 columns.
//   * each wave multiplies the chunk's 2 x KS X fragments (ds_read_b128, XOR-swizzled image: conflict-free) with its
//     resident W fragments: 72 v_mfma_f32_16x16x32_bf16 per chunk and wave, Y^T = W . X^T so that a lane owns 4
//     consecutive columns of one row; the finished bf16 chunk is parked in LDS and written as whole rows, 16 B per lane.
//   * THE TWO WAVES OF A SIMD RUN HALF AN ITERATION APART.  Waves 0-3 (group A, one per SIMD) multiply chunk i while waves
//     4-7 (group B) convert / park / store chunk i - 1 and issue the DMAs, then the roles swap; one workgroup barrier per
//     half-step.  The first version of this kernel ran all eight waves in lockstep and its phases were additive again
//     (measured: MFMA 11.6 us + everything else 27.6 us = 39.2 us for 79 000 x 384 x 384); the matrix pipe of a SIMD is
//     shared by its two waves, so matrix work beside the partner's vector / memory work is the pairing that nets
//     (MI355X_MICROARCH.md, "Two waves per SIMD", items 5 and 9).
//   * the gate operand (the FFN's hidden activation, whose sign gates dH) rides the same DMA mechanism in a slot of its own.
// Every vector-memory instruction of the loop is an LDS-DMA or a range-checked buffer store issued by ALL lanes, so the
// counted waits are exact; rows / columns past the matrix are handled by the buffer range check (reads 0, stores dropped).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "gemm_bf16.cuh"

namespace snipper {

constexpr int kWrThreads = 512, kWrRows = 32, kWrPieces = 48, kWrRowB = kWrPieces * 16;     // LDS image: 768-byte rows
constexpr int kWrSlotB = kWrRows * kWrRowB;                                                // 24 576
constexpr int kWrStageStrideB = kWrRowB + 16, kWrStageB = kWrRows * kWrStageStrideB;       // 784-byte rows (2-way writes)
constexpr int kWrCols = 384;                                                               // output columns per workgroup
constexpr int kWrPerThread = 3;            // 16-byte pieces per thread: 1536 per chunk over 512 threads, 768 per half over 256

struct WresArgs {
  const uint16_t *X; long long ldx;      // [M][K]
  const uint16_t *W; long long ldw;      // [N][K] (row stride ldw): a data gradient passes the TRANSPOSED weight
  const float *bias;                     // [N] or nullptr
  const uint16_t *A; long long lda;      // [M][N] gate activation or nullptr (GATE instantiation only)
  float gate_scale;
  uint16_t *Y; long long ldy;            // [M][N]
  int M, N, K;
  int relu;
  float drop_p; uint32_t seed_lo, seed_hi;
  int n_series;                          // chunk lists: gridDim.x = n_series * ceil(N / 384)
  unsigned long long *stamps;            // diagnostic builds (-DWRES_STAMPS): s_memtime stamps of workgroup 0, waves 0 and 4
  int debug;                             // timing ablations (WRONG results): 1 no W loads, 2 no MFMA, 4 no stores, 8 no X reads,
                                         // 16 no epilogue, 32 no flush, 64 no counted waits, 128 no X DMA instructions
};

typedef __attribute__((address_space(3))) void wres_lds_void;
typedef __attribute__((ext_vector_type(2))) unsigned int wres_u32x2;

__device__ __forceinline__ void wres_dma16(__amdgpu_buffer_rsrc_t src, unsigned char *lds_wave_base, unsigned voff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(src, (wres_lds_void *)lds_wave_base, 16, voff, 0, 0, 0);
}

// LDS: NS ring slots of 24 KB (+ one gate slot) + two 24.5 KB staging images = 145 KB either way: one workgroup per CU.
// ACT: a ReLU and / or dropout epilogue is compiled in (the plain projections carry no trace of it).
#ifdef WRES_STAMPS
#define WRES_STAMP(slot) do { if (blockIdx.x == 0 && lane == 0 && (wave & 3) == 0 && stamp_n < 8 * 40) { stamp_buf[(wave >> 2) * 320 + stamp_n] = ((unsigned long long)(slot) << 56) | (__builtin_amdgcn_s_memtime() & 0x00ffffffffffffffull); ++stamp_n; } } while (0)
#else
#define WRES_STAMP(slot) do { } while (0)
#endif

template <int KS, bool GATE, bool ACT>
__global__ __launch_bounds__(kWrThreads) __attribute__((amdgpu_waves_per_eu(2, 2))) void wres_gemm_kernel(WresArgs g) {
  constexpr int NS = GATE ? 3 : 4;
  constexpr int kRingB = NS * kWrSlotB, kGateB = GATE ? kWrSlotB : 0;
  // ONE LDS object (a second one beside an LDS-DMA target makes hipcc drain vmcnt before every ds_read)
  __shared__ __attribute__((aligned(16))) unsigned char smem[kRingB + kGateB + 2 * kWrStageB];
  unsigned char *ring = smem, *gslot = smem + kRingB, *stage = smem + kRingB + kGateB;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);          // scalar: the group branch below must be a real branch
  const int group = wave >> 2, tidg = tid & 255;
#ifdef WRES_STAMPS
  unsigned long long *stamp_buf = g.stamps;
  int stamp_n = 0;
#endif

  // workgroup -> (column block cb, chunk list `series`): the ncb workgroups that share a chunk list sit on one XCD
  // (ids b, b + 8, b + 16: round-robin placement), so a chunk comes from HBM once and from that XCD's L2 after
  const int ncb = (g.N + kWrCols - 1) / kWrCols, per_group = 8 * ncb;
  const int b = blockIdx.x, grp = b / per_group, within = b - grp * per_group;
  const int cb = within >> 3, series = grp * 8 + (within & 7);
  const int n0 = cb * kWrCols;
  const int NC = (g.M + kWrRows - 1) / kWrRows;
  const int n_iter = series < NC ? (NC - series + g.n_series - 1) / g.n_series : 0;
  if (n_iter == 0) return;

  // ---- X DMA geometry (all 512 threads): piece i = tid + 512 t of a chunk image; row r = i / 48, slot s = i % 48 holds the
  // 16-byte piece c = s ^ (r & 15) of the row (the XOR makes the fragment reads below conflict-free; rows are 48 pieces =
  // 3 blocks of 16, the XOR stays inside a block)
  unsigned x_voff[kWrPerThread];
  int lds_piece[kWrPerThread];
#pragma unroll
  for (int t = 0; t < kWrPerThread; ++t) {
    const int i = tid + kWrThreads * t, r = i / kWrPieces, s = i - r * kWrPieces, c = s ^ (r & 15);
    x_voff[t] = c * 8 < g.K ? ((unsigned)r * (unsigned)g.ldx + (unsigned)c * 8u) * 2u : 0x80000000u;
    lds_piece[t] = (i - lane) * 16;                 // wave-uniform: the hardware adds lane * 16
  }
  auto chunk_of = [&](int it) { return series + it * g.n_series; };
  auto issue_x = [&](int it) {                       // ALWAYS 3 instructions (an empty descriptor outside the chunk list)
    const int chunk = (it < n_iter && !(g.debug & 8)) ? chunk_of(it) : NC;
    const int rows = chunk < NC ? min(kWrRows, g.M - chunk * kWrRows) : 0;
    const long long bytes = rows > 0 ? ((long long)(rows - 1) * g.ldx + g.K) * 2 : 0;
    const __amdgpu_buffer_rsrc_t src = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint16_t *>(g.X + (long long)(rows > 0 ? chunk : 0) * kWrRows * g.ldx), 0, (int)bytes, 0x00020000);
    unsigned char *slot = ring + ((it + NS) % NS) * kWrSlotB;
    if (g.debug & 128) return;
#pragma unroll
    for (int t = 0; t < kWrPerThread; ++t) wres_dma16(src, slot + lds_piece[t], x_voff[t]);
  };
  // ---- output / gate geometry: a GROUP writes out 16 of the chunk's 32 rows (A rows 0-15, B rows 16-31), piece
  // p = tidg + 256 t of that half, row = 16 group + p / 48, 16-byte piece c = p % 48, unswizzled.  A thread DMA-writes exactly
  // the gate pieces it reads back in its flush, so the gate slot needs no barrier of its own.
  unsigned a_voff[kWrPerThread], y_voff[kWrPerThread];
  int stage_off[kWrPerThread], gate_off[kWrPerThread];
#pragma unroll
  for (int t = 0; t < kWrPerThread; ++t) {
    const int p = tidg + 256 * t, rl = p / kWrPieces, c = p - rl * kWrPieces, row = 16 * group + rl;
    const bool col_ok = n0 + c * 8 < g.N;
    stage_off[t] = row * kWrStageStrideB + c * 16;
    gate_off[t] = (row * kWrPieces + c) * 16;
    a_voff[t] = (GATE && col_ok) ? ((unsigned)row * (unsigned)g.lda + (unsigned)c * 8u) * 2u : 0x80000000u;
    y_voff[t] = col_ok ? ((unsigned)row * (unsigned)g.ldy + (unsigned)c * 8u) * 2u : 0x80000000u;
  }
  auto issue_gate = [&](int it) {                    // ALWAYS 3 instructions
    const int chunk = (it >= 0 && it < n_iter) ? chunk_of(it) : NC;
    const int rows = chunk < NC ? min(kWrRows, g.M - chunk * kWrRows) : 0;
    const long long bytes = rows > 0 ? ((long long)(rows - 1) * g.lda + (g.N - n0)) * 2 : 0;
    const __amdgpu_buffer_rsrc_t src = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint16_t *>(g.A + (long long)(rows > 0 ? chunk : 0) * kWrRows * g.lda + n0), 0, (int)bytes, 0x00020000);
#pragma unroll
    for (int t = 0; t < kWrPerThread; ++t) wres_dma16(src, gslot + (gate_off[t] - lane * 16), a_voff[t]);
  };

  // ---- prologue.  The counted waits assume the steady-state instruction stream of a thread -- per iteration: 3 stores,
  // (3 gate DMAs,) 3 X DMAs (chunk it + NS - 1) -- so the NS - 1 iterations "before the first" are issued in exactly that
  // shape, with dropped stores and an empty gate descriptor.
  const __amdgpu_buffer_rsrc_t nowhere = __builtin_amdgcn_make_buffer_rsrc(g.Y, 0, 0, 0x00020000);
  auto dummy_stores = [&]() {
#pragma unroll
    for (int t = 0; t < kWrPerThread; ++t)
      __builtin_amdgcn_raw_buffer_store_b128(gemm_u32x4{0u, 0u, 0u, 0u}, nowhere, 0x80000000u, 0, 0);
  };
#pragma unroll
  for (int j = -(NS - 1); j < 0; ++j) {
    dummy_stores();
    if constexpr (GATE) issue_gate(-1);
    issue_x(j + NS - 1);
  }
  // ---- W slice of this wave into registers: fragment (nt, ks) = rows n = n0 + 48 wave + 16 nt + (lane & 15), reduction
  // elements k = 32 ks + 8 (lane >> 4) .. + 7 (the A operand of v_mfma_f32_16x16x32_bf16)
  gemm_bf16x8 wf[3][KS];
  const int wn0 = n0 + 48 * wave;
  {
    const __amdgpu_buffer_rsrc_t wsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint16_t *>(g.W), 0, (int)(((long long)(g.N - 1) * g.ldw + g.K) * 2), 0x00020000);
#pragma unroll
    for (int nt = 0; nt < 3; ++nt) {
      const int n = wn0 + 16 * nt + (lane & 15);
      const unsigned base = (n < g.N && !(g.debug & 1)) ? ((unsigned)n * (unsigned)g.ldw + 8u * (unsigned)(lane >> 4)) * 2u : 0x80000000u;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        wf[nt][ks] = __builtin_bit_cast(gemm_bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wsrc, base + ks * 64u, 0, 0));
    }
  }
  // bias of this lane's 4 consecutive columns per n-tile
  gemm_f32x4 bias[3];
#pragma unroll
  for (int nt = 0; nt < 3; ++nt) {
    const int n = wn0 + 16 * nt + 4 * (lane >> 4);
    bias[nt] = gemm_f32x4{0.f, 0.f, 0.f, 0.f};
    if (g.bias && n < g.N) bias[nt] = *reinterpret_cast<const gemm_f32x4 *>(g.bias + n);
  }
  // the compiler's own waits for these register loads belong HERE, not at their first use inside the loop (where its
  // conservative vmcnt(0) would drain the DMA ring every iteration): consume every value once
#pragma unroll
  for (int nt = 0; nt < 3; ++nt) {
    asm volatile("" ::"v"(bias[nt]));
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) asm volatile("" ::"v"(wf[nt][ks]));
  }
  // n-tiles of this wave that hold real columns (wave-uniform)
  const int nta = wn0 >= g.N ? 0 : min(3, (g.N - wn0 + 15) / 16);
  const bool drop = g.drop_p > 0.f;
  const float keep_scale = drop ? 1.f / (1.f - g.drop_p) : 1.f;
  const uint32_t thresh16 = (uint32_t)fminf(g.drop_p * 65536.f + 0.5f, 65535.f);
  uint32_t rng = gemm_rand((uint32_t)blockIdx.x * kWrThreads + (uint32_t)tid, g.seed_lo, g.seed_hi) | 1u;     // dropout stream of this lane
  const int frow = lane & 15, fk = lane >> 4;
  const unsigned stage_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)stage;

  gemm_f32x4 acc[3][2];
  // ---- C: the chunk's fragments against the resident W
  auto compute = [&](int it) {
    const unsigned char *slot = ring + (it % NS) * kWrSlotB;
#pragma unroll
    for (int nt = 0; nt < 3; ++nt)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) acc[nt][mt] = bias[nt];        // the sums start at the bias
    if (nta > 0 && !(g.debug & 2)) {          // (a wave whose 48 columns lie past N only keeps the barriers company)
      // the X fragments of k-step ks + 1 are requested before the MFMAs of k-step ks: this wave is the only one of its
      // SIMD in the matrix phase (its partner is in the store phase), so nobody else covers its LDS latency
      auto xread = [&](int ks, int mt) {
        return *reinterpret_cast<const gemm_bf16x8 *>(slot + (16 * mt + frow) * kWrRowB + 16 * ((4 * ks + fk) ^ frow));
      };
      gemm_bf16x8 xa = xread(0, 0), xb = xread(0, 1);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        gemm_bf16x8 na = xa, nb = xb;
        if (ks + 1 < KS) { na = xread(ks + 1, 0); nb = xread(ks + 1, 1); }
        __builtin_amdgcn_sched_barrier(0);          // (hipcc otherwise sinks the two reads back down to their first use)
#pragma unroll
        for (int nt = 0; nt < 3; ++nt) acc[nt][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt][ks], xa, acc[nt][0], 0, 0, 0);
#pragma unroll
        for (int nt = 0; nt < 3; ++nt) acc[nt][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt][ks], xb, acc[nt][1], 0, 0, 0);
        xa = na; xb = nb;
      }
    }
  };
  // ---- E: bias / ReLU / dropout / scale, rounded to bf16, into staging image it & 1: a lane holds columns n .. n + 3 of
  // row 16 mt + (lane & 15).  The LDS stores are inline assembly: before a compiler-visible ds_write hipcc drains vmcnt(0)
  // (a pending LDS-DMA might alias it), which would empty the ring every iteration; the s_waitcnt lgkmcnt(0) in front of
  // the next barrier covers them.
  auto epilogue = [&](int it) {
    if (g.debug & 16) return;
    const unsigned base = stage_lds + (unsigned)((it & 1) * kWrStageB);
#pragma unroll
    for (int nt = 0; nt < 3; ++nt) {
      if (nt >= nta) continue;
      const int nl = 48 * wave + 16 * nt + 4 * (lane >> 4);          // column within the 384-wide block
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        gemm_f32x4 v = acc[nt][mt];
        if (ACT && g.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        if (ACT && drop) {
          // two xorshift32 steps of this lane's own stream (seeded from the double hash of (seed, global thread id)) give
          // the four 16-bit uniforms of the quad, compared with p * 2^16.  Shifts and xors only: v_mul_lo_u32 runs at a
          // quarter of the vector rate, and the per-element double hash of csrc/gemm_bf16.cuh (4 of them) made this epilogue
          // -- which here runs beside the partner wave's matrix phase -- the longer half of every half-step (124 us for
          // 79 000 x 384 x 1024 against 64 without dropout).  The mask is a function of (seed, launch geometry), not of the
          // element index alone; the backward of dropout(relu(.)) reads it off the output (y > 0).
          uint32_t h = rng;
          h ^= h << 13; h ^= h >> 17; h ^= h << 5;
          uint32_t h2 = h;
          h2 ^= h2 << 13; h2 ^= h2 >> 17; h2 ^= h2 << 5;
          rng = h2;
          v.x = (h & 0xffffu) >= thresh16 ? v.x * keep_scale : 0.f;
          v.y = (h >> 16) >= thresh16 ? v.y * keep_scale : 0.f;
          v.z = (h2 & 0xffffu) >= thresh16 ? v.z * keep_scale : 0.f;
          v.w = (h2 >> 16) >= thresh16 ? v.w * keep_scale : 0.f;
        }
        if constexpr (GATE) { v.x *= g.gate_scale; v.y *= g.gate_scale; v.z *= g.gate_scale; v.w *= g.gate_scale; }
        wres_u32x2 o;
        o[0] = gemm_pack2(v.x, v.y);
        o[1] = gemm_pack2(v.z, v.w);
        const unsigned addr = base + (unsigned)((16 * mt + frow) * kWrStageStrideB + nl * 2);
        asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(o) : "memory");
      }
    }
  };
  // ---- F: this group's 16 rows of chunk `it` from staging image it & 1 to memory, whole rows, 16 B per lane (ALWAYS 3
  // store instructions: an empty descriptor outside the chunk list, rows past M and columns past N dropped by the range
  // check); the gate pieces were DMA'd by this very thread
  auto flush = [&](int it) {
    if (g.debug & 32) return;
    const bool valid = it >= 0 && it < n_iter && !(g.debug & 4);
    const int m0 = valid ? chunk_of(it) * kWrRows : 0;
    const int rows = valid ? min(kWrRows, g.M - m0) : 0;
    const __amdgpu_buffer_rsrc_t ysrc = __builtin_amdgcn_make_buffer_rsrc(
        g.Y + (long long)m0 * g.ldy + n0, 0, rows > 0 ? (int)(((long long)(rows - 1) * g.ldy + (g.N - n0)) * 2) : 0, 0x00020000);
    const unsigned char *img = stage + (it & 1) * kWrStageB;
    if constexpr (GATE) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");   // younger than this thread's gate DMAs: 3 X DMAs
#pragma unroll
    for (int t = 0; t < kWrPerThread; ++t) {
      gemm_u32x4 v = *reinterpret_cast<const gemm_u32x4 *>(img + stage_off[t]);
      if constexpr (GATE) {
        const gemm_u32x4 a = *reinterpret_cast<const gemm_u32x4 *>(gslot + gate_off[t]);
        auto keep = [](unsigned av, unsigned vv) {
          const unsigned lo = ((av & 0x8000u) == 0u && (av & 0x7fffu) != 0u) ? 0x0000ffffu : 0u;
          const unsigned hi = ((av & 0x80000000u) == 0u && (av & 0x7fff0000u) != 0u) ? 0xffff0000u : 0u;
          return vv & (lo | hi);
        };
        v.x = keep(a.x, v.x); v.y = keep(a.y, v.y); v.z = keep(a.z, v.z); v.w = keep(a.w, v.w);
      }
      __builtin_amdgcn_raw_buffer_store_b128(v, ysrc, y_voff[t], 0, 0);
    }
  };
  // vector-memory instructions of a thread younger than the X DMAs of the chunk it is about to wait for: the NS - 2 full
  // iterations issued since (plain: 2 x (3 stores + 3 DMAs) = 12;  gate: 1 x (3 + 3 + 3) = 9)
  auto wait_chunk = [&]() {
    if (g.debug & 64) return;
    if constexpr (GATE) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  };
  auto lds_done_barrier = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  };

  wait_chunk();                                   // chunk 0 (this thread's pieces; everything older has completed too)
  __builtin_amdgcn_s_barrier();
  if (group == 0) {
    // group A: half-step 2 it = C(it); half-step 2 it + 1 = E(it), F(it - 1), DMAs, wait for chunk it + 1
    for (int it = 0; it < n_iter; ++it) {
      WRES_STAMP(1);
      compute(it);
      WRES_STAMP(2);
      lds_done_barrier();
      WRES_STAMP(3);
      epilogue(it);
      WRES_STAMP(4);
      flush(it - 1);
      WRES_STAMP(5);
      if constexpr (GATE) issue_gate(it);
      issue_x(it + NS - 1);                       // slot (it - 1) % NS: last read in half-step 2 it - 1
      WRES_STAMP(6);
      wait_chunk();
      WRES_STAMP(7);
      lds_done_barrier();
    }
    __builtin_amdgcn_s_barrier();                 // half-step 2 n: group B parks chunk n - 1
    flush(n_iter - 1);
  } else {
    // group B, half an iteration behind: half-step 2 it = E(it - 1), F(it - 2), DMAs; half-step 2 it + 1 = C(it), wait
    for (int it = 0; it < n_iter; ++it) {
      WRES_STAMP(3);
      if (it > 0) epilogue(it - 1);
      WRES_STAMP(4);
      flush(it - 2);
      WRES_STAMP(5);
      if constexpr (GATE) issue_gate(it - 1);
      issue_x(it + NS - 1);
      WRES_STAMP(6);
      lds_done_barrier();
      WRES_STAMP(1);
      compute(it);
      WRES_STAMP(2);
      wait_chunk();
      WRES_STAMP(7);
      lds_done_barrier();
    }
    epilogue(n_iter - 1);
    flush(n_iter - 2);
    if constexpr (GATE) { issue_gate(n_iter - 1); issue_x(n_iter + NS - 1); }   // (the X DMAs keep flush's count exact)
    lds_done_barrier();
    flush(n_iter - 1);
  }
}

// ---- batched bf16 transpose: dst_i[c][r] = src_i[r][c] for a short list of small matrices in ONE launch -----------------
// (the data gradients above want W^T; the per-step weight refresh, snipper_amd/shadow.py, writes the transposed bf16
// copies of ~30 weights with this kernel instead of one strided-copy launch per weight)
constexpr int kTrMaxItems = 48;
struct TransposeItem { const uint16_t *src; uint16_t *dst; int rows, cols, ld_src, ld_dst, tile0, tiles_c; };
struct TransposeBatch { TransposeItem it[kTrMaxItems]; int count; };

__global__ __launch_bounds__(256) void transpose_batch_bf16_kernel(TransposeBatch b) {
  __shared__ uint16_t tile[64][66];
  int i = 0;
  while (i + 1 < b.count && (int)blockIdx.x >= b.it[i + 1].tile0) ++i;          // tiny list: linear search
  const TransposeItem &t = b.it[i];
  const int local = blockIdx.x - t.tile0, tr = local / t.tiles_c, tc = local - tr * t.tiles_c;
  const int r0 = tr * 64, c0 = tc * 64;
  for (int e = threadIdx.x; e < 64 * 64; e += 256) {
    const int r = e >> 6, c = e & 63;
    tile[r][c] = (r0 + r < t.rows && c0 + c < t.cols) ? t.src[(long long)(r0 + r) * t.ld_src + c0 + c] : (uint16_t)0;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 64 * 64; e += 256) {
    const int c = e >> 6, r = e & 63;                                            // dst row = source column
    if (c0 + c < t.cols && r0 + r < t.rows) t.dst[(long long)(c0 + c) * t.ld_dst + r0 + r] = tile[r][c];
  }
}

}  // namespace snipper
