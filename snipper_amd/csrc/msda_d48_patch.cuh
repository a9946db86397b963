// msda_d48_patch.cuh -- "owner-computes" backward for the encoder shape (D = 48, float32 or bfloat16 value, P = 4,
// L <= 4, Lq == S, level shapes known on the host).  gfx950 only.
//
// Why: the straightforward backward scatters every tap with a float atomic to HBM (the reference does exactly that,
// /root/reference/models/ops/src/cuda/ms_deform_im2col_cuda.cuh:125-152).  On MI355X float atomics execute at the memory
// side at ~1.3-1.9 TB/s chip-wide whatever the locality; an encoder launch adds N*S*M*L*P*4*D*4 B = 728 MB per sample:
// 4.1 ms per launch at N = 8, 1.6 % of the HBM roofline, the largest kernel of the training step.
//
// Idea: in the encoder the queries ARE the pixels of the L feature maps and a query samples each level near its own
// position rescaled to that level (its "anchor").  grad_value is cut into tiles (one level, one (batch, head), 16 x 16 /
// 8 x 8 / 4 x 4 pixels); every tile has ONE owner workgroup that sums the taps landing in it and writes the tile to HBM with
// plain stores (the tiles cover the maps: grad_value needs no zeroing beforehand).  Three kernels:
//   msda_bwd_d48_patchbin_kernel  query side, one workgroup per (batch, head, 8 x 8 block of queries of one level):
//                                 grad_loc / grad_attn for every sample (64 rows x 4 points decoded by 256 threads, then
//                                 32 rows x 8 lanes gather the tap rows -- lane j owns channels 4j..4j+3 and 32+2j, 33+2j
//                                 -- with v_dot2c_f32_bf16 when value and grad_out are both bf16); "which queries of the
//                                 block have a near tap in which tile" as ONE 64-bit word per (tile, block), OR-ed in LDS
//                                 and stored once (4 MB of marks per launch at N = 8; round 1's byte map was 58 MB and
//                                 had to be scanned); a sample with a tap no tile owns is appended to the far list.
//   msda_bwd_d48_tile2_kernel     grad_value side, one workgroup per (n, m, tile): expands the marks of its <= 256
//                                 candidate blocks into a hit list (prefix over popcounts), re-decodes the hits' samples
//                                 in rounds of 128 (the next round's locations prefetched behind the accumulate phase),
//                                 ranks every owned tap within its (wave, pixel) counter and places it with a prefix over
//                                 pixels, accumulates every pixel in registers from float32 rows staged in LDS, adds the
//                                 tile.  No float sum depends on an arrival order: grad_value is BIT-REPRODUCIBLE from
//                                 launch to launch for every tap the tiles own.  (bf16 grad_out rows: the matrix-pipe
//                                 kernels of msda_d48_tilemm.cuh take this kernel's place.)
//   msda_bwd_d48_far_kernel       after the tile kernels: the far list's taps, HBM float atomics on top of the stored
//                                 tiles, 16 lanes per sample so that every atomic instruction adds 64 contiguous bytes.
// "near" = inside the map and |pixel - anchor| <= R on both axes, anchor a fixed function of the query INDEX; "owned" =
// near, in the map, and the block is among the tile's candidates.  Both kernels evaluate them with the pinned arithmetic of
// msda_d48.cuh, so owned + unowned is a partition of the taps for ANY input; locality only decides how many taps go the
// fast way.  Measured against round 1's pair (byte map + LDS-atomic ranks, msda_d48_owner.cuh, removed): whole backward at
// N = 8, bf16 rows, sigma 1.5 / 8 px: 1.07 / 2.33 ms against 1.15 / 2.40 ms; with a bfloat16 value 0.97 ms.
//
// What is NOT here: round 2 also built the forward and this query side around LDS-staged value neighbourhoods (anchor range
// +- 5 px of every level staged by LDS-DMA, gathers from LDS), one-head-per-workgroup and software-pipelined.  Both were
// parity-green and both were slower than the texture-path kernels (forward 311 / 395 us against 272 us): a 62 KB float32
// window leaves two workgroups per CU, every phase becomes latency-bound, and LDS-DMA with per-lane addresses costs the
// issuing wave as much as the register loads it replaces (profiles/r02_lds_staging_experiment_kernel_avgs.csv, DESIGN.md
// section 3.4b); the kernels were removed.
//
// Semantics restated from /root/reference/models/ops/src/cuda/ms_deform_im2col_cuda.cuh:87-159 (tap gradients) and
// :513-616 (backward, D = 48 branch); no code is shared with it.
#pragma once
#include "msda_d48.cuh"

namespace snipper {

constexpr int kPatchMaxLevels = 4;
constexpr int kPatchP = 4;
constexpr int kPatchB = 8;                       // query block edge
constexpr int kPatchThreads = 256;               // = 64 rows x 4 points (decode) = 32 rows x 8 lanes (gather) = 16 rows x 16 lanes (atomics)
constexpr int kPatchRowBytes = kD48 * 4;         // 192
constexpr int kPatchMaxTiles = 152;              // tiles one (block, level) may mark: a block of the coarsest of three levels
                                                 // (8 queries = 32 px of the finest) at radius 24 px reaches 12 x 12 tiles of 8 x 8
                                                 // there; tiles past this bound are nobody's (their taps take the far list)
constexpr int kPatchMaxCand = 256;               // candidate blocks per tile (one per thread of the tile kernel)

struct PatchLevel {
  int H, W, start;          // level geometry
  int nbx, nby, blk_base;   // 8 x 8 query blocks of this level, index of its first block
  int shift, ntx, nty, tile_base;   // grad_value tiles (edge = 1 << shift)
};

struct PatchPlan {
  PatchLevel lv[kPatchMaxLevels];
  // marks: tile T of sampled level l has, for every query level lq, a rectangle of candidate blocks (the blocks holding
  // queries whose near taps can reach T: a pure function of T, l, lq, radius).  Block (by, bx) of that rectangle owns the
  // word  lvl_base[l] + T * tstride[l] + coff[l][lq] + (by - by0) * cbw[l][lq] + (bx - bx0)  of its (n, m) slab.
  int cbw[kPatchMaxLevels][kPatchMaxLevels], cbh[kPatchMaxLevels][kPatchMaxLevels];
  int coff[kPatchMaxLevels][kPatchMaxLevels];
  int tstride[kPatchMaxLevels];
  long long lvl_base[kPatchMaxLevels];
  long long words_per_nm;
  unsigned long long *marks;    // [N*M][words_per_nm], zeroed per backward call
  // taps no tile owns (their sample is not near its anchor, or the block is not among the tile's candidates): the query
  // side appends the SAMPLE to this list (far_count zeroed with the marks) and msda_bwd_d48_far_kernel, launched after the
  // tile kernels, adds its far taps with HBM float atomics -- so the tile kernels write every pixel of grad_value with plain
  // stores, nothing has to be zeroed beforehand (121 MB per launch at N = 8) and no tile reads grad_value back.
  unsigned *far_count;          // entries in far_list
  uint2 *far_list;              // {sample index ((n Lq + q) M + m) L P + l P + p, bit k: tap k is far}; capacity = all samples
  // size ratios rw[a][b] = (float)W_a / (float)W_b (rh likewise), computed on the host with IEEE float division: the
  // anchor arithmetic both kernels must agree on, without per-thread divisions (bit-identical to __fdiv_rn)
  float rw[kPatchMaxLevels][kPatchMaxLevels], rh[kPatchMaxLevels][kPatchMaxLevels];
  int L, nblocks, total_tiles;
  float radius;    // a sample is "near" when |pixel - anchor| <= radius on both axes
  int debug;       // timing ablations (config.reserved[0]; results are WRONG when != 0): 1 no row staging, 2 no gather /
                   // accumulate, 4 no decode, 8 skeleton only
  unsigned long long *stamps;   // diagnostic builds (-DTILE2_STAMPS, tools/tile2_stamps.py): s_memtime stamps of sampled workgroups
};

struct PatchBlock { int n, m, lq, qy0, qx0, bh, bw; };

struct PatchRec {   // 32 B per (row, point) of the level in flight
  f32x4 w;          // lh, lw, a, bits (bit 4 + k: no tile owns tap k -> HBM atomic here)
  u32x4 g;          // byte offset of each tap's row in value / grad_value, or kOobOffset
};

__device__ __forceinline__ bool patch_block(const PatchPlan &p, const CoreDims &d, int nblk_padded, PatchBlock &b) {
  // (32-bit: the launcher's grid is an int, and 64-bit divisions by run-time values are ~100 scalar instructions each)
  const unsigned id = (unsigned)xcd_band_block(nblk_padded);     // (n, block, m), m fastest: the 8 heads of a block share lines
  const unsigned total = (unsigned)d.N * (unsigned)p.nblocks * (unsigned)d.M;
  if (id >= total) return false;
  const unsigned t = id / (unsigned)d.M;
  b.m = (int)(id - t * (unsigned)d.M);
  b.n = (int)(t / (unsigned)p.nblocks);
  const int blk = (int)(t - (unsigned)b.n * (unsigned)p.nblocks);
  int lq = 0;
  for (int i = 1; i < p.L; ++i) lq = (blk >= p.lv[i].blk_base) ? i : lq;
  const int r = blk - p.lv[lq].blk_base;
  const int by = r / p.lv[lq].nbx, bx = r - by * p.lv[lq].nbx;
  b.lq = lq;
  b.qy0 = by * kPatchB;
  b.qx0 = bx * kPatchB;
  b.bh = min(kPatchB, p.lv[lq].H - b.qy0);
  b.bw = min(kPatchB, p.lv[lq].W - b.qx0);
  return true;
}

// workgroup barrier that orders LDS traffic only: unlike __syncthreads() it does not drain the vector-memory counter, so
// LDS-DMA transfers and prefetched global loads stay in flight across it
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void vm_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// all-reduce over the 8 lanes of a row group (DPP: quad xor 1, quad xor 2, mirror within the half row)
__device__ __forceinline__ float row8_sum(float v) {
  v += dpp_f32<0xB1>(v);
  v += dpp_f32<0x4E>(v);
  v += dpp_f32<0x141>(v);
  return v;
}

// lane j of an 8-lane row group owns channels 4j..4j+3 (one 16-B access) and 32+2j, 33+2j (one 8-B access)
struct Row6 { f32x4 a; float2 b; };
__device__ __forceinline__ Row6 lds_row(const unsigned char *win, unsigned off, int j) {
  Row6 r;
  r.a = *reinterpret_cast<const f32x4 *>(win + off + 16 * j);
  r.b = *reinterpret_cast<const float2 *>(win + off + 128 + 8 * j);
  return r;
}
__device__ __forceinline__ Row6 buf_row(__amdgpu_buffer_rsrc_t rsrc, unsigned off, int j) {
  Row6 r;
  const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + 16u * j, 0, 0);
  typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
  const u32x2 b = __builtin_amdgcn_raw_buffer_load_b64(rsrc, off + 128u + 8u * j, 0, 0);
  r.a.x = __uint_as_float(a.x); r.a.y = __uint_as_float(a.y); r.a.z = __uint_as_float(a.z); r.a.w = __uint_as_float(a.w);
  r.b.x = __uint_as_float(b.x); r.b.y = __uint_as_float(b.y);
  return r;
}

// ------------------------------------------------------------------------------------------------------------------
// Marks geometry shared by the two backward kernels
// ------------------------------------------------------------------------------------------------------------------
// conservative index range of the queries of an axis with n_lq cells whose anchor, expressed in the sampled level
// (n_l cells), can fall in [lo_px, hi_px]
__device__ __forceinline__ void patch_anchor_range(float lo_px, float hi_px, float inv, int n_lq, int &i0, int &i1) {
  i0 = (int)floorf(__fmaf_rn(lo_px + 0.5f, inv, -0.5f)) - 1;
  i1 = (int)ceilf(__fmaf_rn(hi_px + 0.5f, inv, -0.5f)) + 1;
  i0 = i0 < 0 ? 0 : i0;
  i1 = i1 > n_lq - 1 ? n_lq - 1 : i1;
}
// candidate BLOCK rectangle of query level lq for the tile with origin (ty0, tx0), edge `edge`, of level l: a near tap
// in the tile has |sample - anchor| <= R and tap in [sample - 1, sample + 1], so the anchor lies in [t0 - 1 - R, t0 + edge + R]
__device__ __forceinline__ void patch_tile_cand(const PatchPlan &p, int l, int lq, int ty0, int tx0, int edge,
                                                int &bx0, int &bx1, int &by0, int &by1) {
  int qx0, qx1, qy0, qy1;
  patch_anchor_range((float)(tx0 - 1) - p.radius, (float)(tx0 + edge) + p.radius, p.rw[lq][l], p.lv[lq].W, qx0, qx1);
  patch_anchor_range((float)(ty0 - 1) - p.radius, (float)(ty0 + edge) + p.radius, p.rh[lq][l], p.lv[lq].H, qy0, qy1);
  bx0 = qx0 >> 3; bx1 = qx1 >> 3; by0 = qy0 >> 3; by1 = qy1 >> 3;
}
// tiles of level l that near taps of block b can touch (anchor range +- (R + 1), clipped to the map)
struct PatchTileBox { int tx0, ty0, ntx, nty; };
__device__ __forceinline__ PatchTileBox patch_tile_box(const PatchPlan &p, const PatchBlock &b, int l) {
  const PatchLevel &s = p.lv[l];
  const float rw = p.rw[l][b.lq], rh = p.rh[l][b.lq];
  const float ax0 = anchor_from_ratio(b.qx0, rw), ax1 = anchor_from_ratio(b.qx0 + b.bw - 1, rw);
  const float ay0 = anchor_from_ratio(b.qy0, rh), ay1 = anchor_from_ratio(b.qy0 + b.bh - 1, rh);
  const int x0 = max(0, (int)floorf(ax0 - p.radius) - 1), x1 = min(s.W - 1, (int)ceilf(ax1 + p.radius) + 1);
  const int y0 = max(0, (int)floorf(ay0 - p.radius) - 1), y1 = min(s.H - 1, (int)ceilf(ay1 + p.radius) + 1);
  PatchTileBox t;
  t.tx0 = x0 >> s.shift; t.ty0 = y0 >> s.shift;
  t.ntx = x1 >= x0 ? (x1 >> s.shift) - t.tx0 + 1 : 0;
  t.nty = y1 >= y0 ? (y1 >> s.shift) - t.ty0 + 1 : 0;
  return t;
}
// word of block b in the marks of tile (tyi, txi) of level l, or -1 when the block is not among the tile's candidates
__device__ __forceinline__ long long patch_slot(const PatchPlan &p, const PatchBlock &b, int M, int l, int tyi, int txi) {
  const PatchLevel &s = p.lv[l];
  int bx0, bx1, by0, by1;
  patch_tile_cand(p, l, b.lq, tyi << s.shift, txi << s.shift, 1 << s.shift, bx0, bx1, by0, by1);
  const int bx = b.qx0 >> 3, by = b.qy0 >> 3;
  const int dx = bx - bx0, dy = by - by0;
  if (dx < 0 || dy < 0 || bx > bx1 || by > by1 || dx >= p.cbw[l][b.lq] || dy >= p.cbh[l][b.lq]) return -1;
  return ((long long)b.n * M + b.m) * p.words_per_nm + p.lvl_base[l] + (long long)(tyi * s.ntx + txi) * p.tstride[l] +
         p.coff[l][b.lq] + dy * p.cbw[l][b.lq] + dx;
}

// ------------------------------------------------------------------------------------------------------------------
// Backward, query side.  VT = storage type of `value` (float, or uint16_t = bfloat16 bits: under bf16 autocast the
// temporal mean of the projected memory can be kept in bf16, which halves the bytes of every tap row -- the gathers are
// bound by the ~20 TB/s the texture path delivers, profiles/r02_lds_staging_experiment_kernel_avgs.csv); coordinates,
// weights, all gradients and all sums stay float32.
// ------------------------------------------------------------------------------------------------------------------
typedef __bf16 patch_bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int patch_u32x2 __attribute__((ext_vector_type(2)));

// <grad_out slice, value tap row> over this lane's 6 channels (4j..4j+3, 32+2j, 33+2j)
template <typename VT, bool GO_BF16> struct PatchDot;
template <bool GO_BF16> struct PatchDot<float, GO_BF16> {
  static constexpr unsigned kRowV = 192;
  static constexpr bool kContig = false;      // lane j owns channels 4j..4j+3 and 32+2j, 33+2j (a 16-byte + an 8-byte load)
  typedef Row6 Raw;
  static __device__ __forceinline__ Raw load(__amdgpu_buffer_rsrc_t vsrc, unsigned off, int j) { return buf_row(vsrc, off, j); }
  static __device__ __forceinline__ float dot(const Raw &v, const float (&g)[6], const unsigned (&)[3]) {
    return g[0] * v.a.x + g[1] * v.a.y + g[2] * v.a.z + g[3] * v.a.w + g[4] * v.b.x + g[5] * v.b.y;
  }
};
template <bool GO_BF16> struct PatchDot<uint16_t, GO_BF16> {
  static constexpr unsigned kRowV = 96;
  // bf16 value rows are 96 bytes: lane j owns the 6 CONTIGUOUS channels 6j .. 6j+5, so a tap is ONE 12-byte load per lane
  // (the float32 layout needs a 16-byte and an 8-byte one); the texture path's cost is per instruction and per 64-byte
  // segment touched, and this halves the former (the forward kernel of msda_d48.cuh gathers the same way)
  static constexpr bool kContig = true;
  typedef unsigned int Raw __attribute__((ext_vector_type(3)));
  static __device__ __forceinline__ Raw load(__amdgpu_buffer_rsrc_t vsrc, unsigned off, int j) {
    return __builtin_amdgcn_raw_buffer_load_b96(vsrc, off + 12u * j, 0, 0);
  }
  static __device__ __forceinline__ float dot(const Raw &a, const float (&g)[6], const unsigned (&gp)[3]) {
    // (elements by INDEX: with `.x` / `.y` on a loaded pair, hipcc of ROCm 7.2 narrowed a 64-bit load to one dword and fed
    //  the first element to both products -- reproduced in isolation, round 2)
    const unsigned ax = a[0], ay = a[1], az = a[2];
    if constexpr (GO_BF16) {        // both operands are bf16 pairs: v_dot2c_f32_bf16, products exact, float32 sums
      float acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(patch_bf16x2, ax), __builtin_bit_cast(patch_bf16x2, gp[0]), 0.f, false);
      acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(patch_bf16x2, ay), __builtin_bit_cast(patch_bf16x2, gp[1]), acc, false);
      return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(patch_bf16x2, az), __builtin_bit_cast(patch_bf16x2, gp[2]), acc, false);
    } else {
      return g[0] * __uint_as_float(ax << 16) + g[1] * __uint_as_float(ax & 0xffff0000u) +
             g[2] * __uint_as_float(ay << 16) + g[3] * __uint_as_float(ay & 0xffff0000u) +
             g[4] * __uint_as_float(az << 16) + g[5] * __uint_as_float(az & 0xffff0000u);
    }
  }
};

// x + (x of the DPP partner lane), ONE instruction (v_add_f32_dpp).  Inline assembly: from `v + dpp(v)` hipcc builds
// v_mov_b32_dpp pairs feeding v_pk_add_f32 (1.5 instructions per sum).  A DPP operand written by the preceding VALU
// instruction needs two wait states the assembler does not insert: callers pass the inputs through patch_dpp_fence first.
__device__ __forceinline__ float patch_add_xor1(float x) {
  float r;
  asm volatile("v_add_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(x));
  return r;
}
__device__ __forceinline__ float patch_add_xor2(float x) {
  float r;
  asm volatile("v_add_f32_dpp %0, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(x));
  return r;
}
__device__ __forceinline__ float patch_add_mirror8(float x) {
  float r;
  asm volatile("v_add_f32_dpp %0, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(x));
  return r;
}
template <int N> __device__ __forceinline__ void patch_dpp_fence(float (&v)[N]);
template <> __device__ __forceinline__ void patch_dpp_fence<3>(float (&v)[3]) {
  asm volatile("s_nop 1" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]));
}
template <> __device__ __forceinline__ void patch_dpp_fence<6>(float (&v)[6]) {
  asm volatile("s_nop 1" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]));
}
template <> __device__ __forceinline__ void patch_dpp_fence<12>(float (&v)[12]) {
  asm volatile("s_nop 1" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]),
               "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]));
}

// the rows of a wave (16 of them, 4 point lanes each) that take part, as a 16-bit set: OR over each row's 4 lanes of a ballot,
// then every 4th bit gathered (scalar arithmetic: the ballot is wave-uniform)
__device__ __forceinline__ unsigned long long patch_rows_of(unsigned long long bal) {
  unsigned long long x = (bal | (bal >> 1) | (bal >> 2) | (bal >> 3)) & 0x1111111111111111ull;
  x = (x | (x >> 3)) & 0x0303030303030303ull;
  x = (x | (x >> 6)) & 0x000F000F000F000Full;
  x = (x | (x >> 12)) & 0x000000FF000000FFull;
  return (x | (x >> 24)) & 0xFFFFull;
}

// One workgroup = one (n, m, 8 x 8 block of queries of one level).  Three phases, TWO barriers (round 4; before, every
// sampled level ran decode -> barrier -> marks -> gather -> barrier in turn, and the phase stamps of tools/tile2_stamps.py
// showed 62 % of a workgroup's life outside the gathers: profiles/r04_patchbin_stamps.txt):
//   0. issue every global load the workgroup needs (loc / attn of all levels, its grad_out rows) and, one thread per
//      (level, tile), the mark-word slots of all levels at once;
//   1. decode the samples of ALL levels (thread = (row, point), one sample per level) into LDS records and OR the rows into
//      the tiles' mark words -- per wave and tile ONE LDS atomic carrying the rows of all its lanes (64 lanes OR-ing bits
//      into the same word serialise in the LDS atomic unit);
//   2. store the marks, gather + dot + reduce (32 rows x 8 lanes, twice) and store grad_loc / grad_attn for all levels,
//      then the far taps' HBM atomics of the levels that have any.
// NL = levels the LDS is sized for (3: the usual three-level plan -- 31.5 KB, four workgroups per CU; 4: 42 KB, three)
template <typename VT, bool GO_BF16, int NL = kPatchMaxLevels>
__global__ __launch_bounds__(kPatchThreads) void msda_bwd_d48_patchbin_kernel(
    const void *__restrict__ grad_out, const VT *__restrict__ value, const float *__restrict__ loc,
    const float *__restrict__ attn, CoreDims d, PatchPlan plan, float *__restrict__ grad_value,
    float *__restrict__ grad_loc, float *__restrict__ grad_attn, int nblk_padded) {
  using DOT = PatchDot<VT, GO_BF16>;
  constexpr unsigned kRowV = DOT::kRowV;                 // bytes of one head row of `value`
#ifdef TILE2_STAMPS
  constexpr int kStampBytes = 64 * 8;
#else
  constexpr int kStampBytes = 0;
#endif
  __shared__ __attribute__((aligned(16))) unsigned char smem[NL * kPatchThreads * sizeof(PatchRec) + NL * kPatchMaxTiles * 16 + kStampBytes];
  PatchRec *recs = reinterpret_cast<PatchRec *>(smem);                                                   // [level][row][point]
  unsigned long long *s_mask = reinterpret_cast<unsigned long long *>(recs + NL * kPatchThreads);         // [level][tile]
  long long *s_slot = reinterpret_cast<long long *>(s_mask + NL * kPatchMaxTiles);                       // [level][tile]
  PatchBlock b;
  if (!patch_block(plan, d, nblk_padded, b)) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // (uniform: the level tables below are then SCALAR loads)
#ifdef TILE2_STAMPS
  // (tools/tile2_stamps.py: the second half of the stamp buffer, wave 0 of every 97th workgroup)
  unsigned long long *pst = reinterpret_cast<unsigned long long *>(s_slot + NL * kPatchMaxTiles);
  unsigned long long *pst_out = (plan.stamps && blockIdx.x % 97 == 0 && blockIdx.x / 97 < 256) ? plan.stamps + 256 * 128 + (blockIdx.x / 97) * 64 : nullptr;
  int pst_n = 0;
#define PATCH_STAMP(slot) do { if (pst_out && tid == 0 && pst_n < 64) { pst[pst_n] = ((unsigned long long)(slot) << 56) | (__builtin_amdgcn_s_memtime() & 0x00ffffffffffffffull); ++pst_n; } } while (0)
#else
#define PATCH_STAMP(slot) do { } while (0)
#endif
  PATCH_STAMP(0);
  const PatchLevel lvq = plan.lv[b.lq];
  const int LP = d.L * kPatchP;
  const int rd = tid >> 2, pd = tid & 3;
  const int rdy = rd >> 3, rdx = rd & 7;
  const bool rd_ok = rdy < b.bh && rdx < b.bw;
  const int qd = rd_ok ? lvq.start + (b.qy0 + rdy) * lvq.W + b.qx0 + rdx : lvq.start;
  const long long rowd = ((long long)b.n * d.Lq + qd) * d.M + b.m;
  const unsigned px_stride = msda_px_stride(d, kRowV);
  const unsigned gbase = msda_row_base(d, (unsigned)b.n, (unsigned)b.m, kRowV);
  const int j = tid & 7, rg = tid >> 3;
  const unsigned value_bytes = (unsigned)((size_t)d.N * d.S * d.M * kRowV);
  const auto vsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<VT *>(value), 0, (int)value_bytes, 0x00020000);

  // ---- phase 0: loads.  loc / attn of this thread's sample in every level first (phase 1 waits for them only) ...
  float2 xy_l[kPatchMaxLevels];
  float a_l[kPatchMaxLevels];
#pragma unroll
  for (int l = 0; l < kPatchMaxLevels; ++l) {
    xy_l[l] = make_float2(0.f, 0.f); a_l[l] = 0.f;
    if (l < plan.L) {
      const long long li = rowd * LP + l * kPatchP + pd;
      xy_l[l] = *reinterpret_cast<const float2 *>(loc + 2 * li);
      a_l[l] = attn[li];
    }
  }
  // ... then this thread's slices of the two grad_out rows it gathers for (channels 4j..4j+3, 32+2j, 33+2j, or 6j..6j+5),
  // left in flight until phase 1 is done
  unsigned graw[2][6];
  long long rowg[2];
  bool rg_ok[2];
  constexpr int kCa = DOT::kContig ? 6 : 4;
#pragma unroll
  for (int ps = 0; ps < 2; ++ps) {
    const int r = ps * 32 + rg, ry = r >> 3, rx = r & 7;
    rg_ok[ps] = ry < b.bh && rx < b.bw;
    const int q = rg_ok[ps] ? lvq.start + (b.qy0 + ry) * lvq.W + b.qx0 + rx : lvq.start;
    rowg[ps] = ((long long)b.n * d.Lq + q) * d.M + b.m;
    const int ca = kCa * j, cb = DOT::kContig ? 6 * j + 4 : 32 + 2 * j;
#pragma unroll
    for (int c = 0; c < 6; ++c) graw[ps][c] = 0u;
    if constexpr (GO_BF16) {
      const uint16_t *gr = reinterpret_cast<const uint16_t *>(grad_out) + rowg[ps] * kD48;
      if constexpr (DOT::kContig) {            // 12 bytes at 12 j: 4-byte aligned
        const unsigned *g32 = reinterpret_cast<const unsigned *>(gr + ca);
        graw[ps][0] = g32[0]; graw[ps][1] = g32[1]; graw[ps][2] = g32[2];
      } else {
        const uint2 pa = *reinterpret_cast<const uint2 *>(gr + ca);
        graw[ps][0] = pa.x; graw[ps][1] = pa.y; graw[ps][2] = *reinterpret_cast<const unsigned *>(gr + cb);
      }
    } else {
      const float *gr = reinterpret_cast<const float *>(grad_out) + rowg[ps] * kD48;
      if constexpr (DOT::kContig) {
        const uint2 a0 = *reinterpret_cast<const uint2 *>(gr + ca), a1 = *reinterpret_cast<const uint2 *>(gr + ca + 2);
        const uint2 a2 = *reinterpret_cast<const uint2 *>(gr + cb);
        graw[ps][0] = a0.x; graw[ps][1] = a0.y; graw[ps][2] = a1.x; graw[ps][3] = a1.y; graw[ps][4] = a2.x; graw[ps][5] = a2.y;
      } else {
        const u32x4 pa = *reinterpret_cast<const u32x4 *>(gr + ca);
        const uint2 pb = *reinterpret_cast<const uint2 *>(gr + cb);
        graw[ps][0] = pa.x; graw[ps][1] = pa.y; graw[ps][2] = pa.z; graw[ps][3] = pa.w; graw[ps][4] = pb.x; graw[ps][5] = pb.y;
      }
    }
  }
  // mark-word slots of every level: thread (wave = level, lane = tile of the level's box)
  // (wave-uniform level: the plan's tables stay scalar loads.  A wave without a level of its own -- wave 3 of a three-level
  //  plan -- takes every other 64 tiles of level 0, whose box is the largest: 81 tiles of 8 x 8 for a block of the finest level)
  const int slot_lv = wave < plan.L ? wave : 0;
  const int slot_first = wave < plan.L ? lane : lane + 64;
  const int slot_step = (slot_lv == 0 && plan.L < kPatchThreads / 64) ? 128 : 64;
  {
    const PatchTileBox tb = patch_tile_box(plan, b, slot_lv);
    const int nt = min(tb.ntx * tb.nty, kPatchMaxTiles);
    for (int ti = slot_first; ti < nt; ti += slot_step) {
      const int tiy = ti / tb.ntx, tix = ti - tiy * tb.ntx;
      s_slot[slot_lv * kPatchMaxTiles + ti] = patch_slot(plan, b, d.M, slot_lv, tb.ty0 + tiy, tb.tx0 + tix);
      s_mask[slot_lv * kPatchMaxTiles + ti] = 0ull;
    }
  }
  PATCH_STAMP(1);
  lds_barrier();                 // slots and cleared masks visible (the global loads stay in flight)
  PATCH_STAMP(2);

  // ---- phase 1: decode + marks, all levels ----
#pragma unroll
  for (int l = 0; l < kPatchMaxLevels; ++l) {
    if (l >= plan.L) break;
    const PatchLevel lvl = plan.lv[l];
    const PatchTileBox tb = patch_tile_box(plan, b, l);
    unsigned long long *mask_l = s_mask + l * kPatchMaxTiles;
    const long long *slot_l = s_slot + l * kPatchMaxTiles;
    unsigned my_bits = 0u;
    const float2 xy = xy_l[l];
    const float a_in = a_l[l];
    const float y = px_coord(xy.y, lvl.H), x = px_coord(xy.x, lvl.W);
    const bool inside = rd_ok && (y > -1.f) && (x > -1.f) && (y < (float)lvl.H) && (x < (float)lvl.W);
    const bool near = inside && near_anchor(x, y, anchor_from_ratio(b.qx0 + rdx, plan.rw[l][b.lq]),
                                            anchor_from_ratio(b.qy0 + rdy, plan.rh[l][b.lq]), plan.radius);
    const float yf = floorf(y), xf = floorf(x);
    const int y0 = (int)yf, x0 = (int)xf;
    PatchRec r;
    r.w.x = inside ? y - yf : 0.f; r.w.y = inside ? x - xf : 0.f; r.w.z = inside ? a_in : 0.f;
    unsigned go[4];
    int ti_k[4];                 // the tile (index in the level's box) that owns tap k, or -1
    // per AXIS first (two columns, two rows): in the map, tile index within the level's box, inside the box -- one unsigned
    // compare each; a tap then only combines its row's and its column's results
    bool okx[2], oky[2], boxx[2], boxy[2];
    int tixa[2], tiya[2];
    unsigned rowoff[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      okx[a] = inside && (unsigned)(x0 + a) < (unsigned)lvl.W;
      oky[a] = inside && (unsigned)(y0 + a) < (unsigned)lvl.H;
      tixa[a] = ((x0 + a) >> lvl.shift) - tb.tx0;
      tiya[a] = ((y0 + a) >> lvl.shift) - tb.ty0;
      boxx[a] = near && (unsigned)tixa[a] < (unsigned)tb.ntx;
      boxy[a] = (unsigned)tiya[a] < (unsigned)tb.nty;
      rowoff[a] = gbase + (unsigned)(lvl.start + (y0 + a) * lvl.W + x0) * px_stride;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const bool in_map = oky[k >> 1] && okx[k & 1];
      go[k] = in_map ? rowoff[k >> 1] + (unsigned)(k & 1) * px_stride : kOobOffset;
      ti_k[k] = -1;
      if (in_map && boxy[k >> 1] && boxx[k & 1]) {
        const int ti = tiya[k >> 1] * tb.ntx + tixa[k & 1];
        if (ti < kPatchMaxTiles && slot_l[ti] >= 0) ti_k[k] = ti;
      }
      my_bits |= (in_map && ti_k[k] < 0) ? (1u << k) : 0u;    // no tile owns the tap: msda_bwd_d48_far_kernel adds it
    }
    // samples with a far tap go to the far list: one counter bump per wave that has any (ballot + prefix count)
    {
      const unsigned long long fb = __ballot(my_bits != 0u);
      if (fb) {
        unsigned base = 0u;
        if (lane == 0) base = atomicAdd(plan.far_count, (unsigned)__popcll(fb));
        base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
        if (my_bits)
          plan.far_list[base + __builtin_amdgcn_mbcnt_hi((unsigned)(fb >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)fb, 0u))] =
              make_uint2((unsigned)(rowd * LP + l * kPatchP + pd), my_bits);
      }
    }
    // marks: a tap whose tile an earlier tap of the sample already marks adds nothing
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      bool want = ti_k[k] >= 0;
#pragma unroll
      for (int e = 0; e < k; ++e) want = want && ti_k[e] != ti_k[k];
      if (want && !(plan.debug & 16)) atomicOr(&mask_l[ti_k[k]], 1ull << rd);      // (debug 16: timing ablation, WRONG results)
    }
    r.w.w = 0.f;
    r.g.x = go[0]; r.g.y = go[1]; r.g.z = go[2]; r.g.w = go[3];
    recs[l * kPatchThreads + tid] = r;
  }
  // the grad_out rows have landed by now
  float g[2][6];
  unsigned gp[2][3];
#pragma unroll
  for (int ps = 0; ps < 2; ++ps) {
    if constexpr (GO_BF16) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        gp[ps][c] = rg_ok[ps] ? graw[ps][c] : 0u;
        g[ps][2 * c] = __uint_as_float(gp[ps][c] << 16); g[ps][2 * c + 1] = __uint_as_float(gp[ps][c] & 0xffff0000u);
      }
    } else {
#pragma unroll
      for (int c = 0; c < 3; ++c) gp[ps][c] = 0u;
#pragma unroll
      for (int c = 0; c < 6; ++c) g[ps][c] = rg_ok[ps] ? __uint_as_float(graw[ps][c]) : 0.f;
    }
  }
  PATCH_STAMP(3);
  lds_barrier();                 // records and masks complete
  PATCH_STAMP(4);

  // ---- phase 2: marks out, gathers ----
  {
    const PatchTileBox tb = patch_tile_box(plan, b, slot_lv);
    const int nt = min(tb.ntx * tb.nty, kPatchMaxTiles);
    for (int ti = slot_first; ti < nt; ti += slot_step) {
      const long long sl = s_slot[slot_lv * kPatchMaxTiles + ti];
      const unsigned long long mk = s_mask[slot_lv * kPatchMaxTiles + ti];
      if (sl >= 0 && mk != 0ull) plan.marks[sl] = mk;
    }
  }
  PATCH_STAMP(5);
  // The gathers: per (level, pass) 16 tap loads in flight, then the dots and the reduction.  (Measured and dropped, round 4:
  // a software pipeline over points that keeps 8 or 16 loads of the NEXT points in flight across passes and levels --
  // 294 / 297 us against 282: the phase is bound by the texture path / L1 footprint of three workgroups per CU, more loads
  // in flight thrash it; the same reason four waves per SIMD are slower.)
  // 12 sums over the row's 8 lanes of which lane p < 4 needs those of point p only: a transposing reduction, every
  // stage halves what a lane carries (12 -> 6 -> 3 values: 30 instructions against 4 x 3 x 3 all-reduce steps).
  // Lane i ends with point tgt(i) = {0, 1, 2, 3, 3, 2, 1, 0}[i]; stage 1 (lane ^ 1) keeps the points of tgt's parity,
  // stage 2 (lane ^ 2) those of tgt's half, stage 3 adds lane 7 - i (same target).  Order of every sum fixed.
  const bool keep_even = ((j ^ (j >> 2)) & 1) == 0, keep_h0 = (((j >> 1) ^ (j >> 2)) & 1) == 0;
#pragma unroll
  for (int l = 0; l < kPatchMaxLevels; ++l) {
    if (l >= plan.L || (plan.debug & 2)) break;
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
      const PatchRec *mine = recs + l * kPatchThreads + (ps * 32 + rg) * kPatchP;
      // per point: this lane's share (6 channels) of the four tap dots, folded into its share of
      //   A = sum_k w_k dot_k,  X = hh (d1 - d0) + lh (d3 - d2),  Y = hw (d2 - d0) + lw (d3 - d1)
      // in lerp form (10 operations: with f = (d3 - d2) - (d1 - d0), X = e01 + lh f, Y = e02 + lw f)
      float q[12];             // [quantity A / X / Y][point]
#pragma unroll
      for (int p = 0; p < kPatchP; ++p) {
        const PatchRec r = mine[p];
        const float lh = r.w.x, lw = r.w.y;
        const unsigned go[4] = {r.g.x, r.g.y, r.g.z, r.g.w};
        float dot[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)            // (a tap outside the map has offset kOobOffset: the buffer load returns 0)
          dot[k] = DOT::dot(DOT::load(vsrc, go[k], j), g[ps], gp[ps]);
        const float e01 = dot[1] - dot[0], e23 = dot[3] - dot[2], e02 = dot[2] - dot[0], f = e23 - e01;
        const float t0 = __fmaf_rn(lw, e01, dot[0]), t1 = __fmaf_rn(lw, e23, dot[2]);
        q[p] = __fmaf_rn(lh, t1 - t0, t0);
        q[4 + p] = __fmaf_rn(lh, f, e01);
        q[8 + p] = __fmaf_rn(lw, f, e02);
      }
      patch_dpp_fence(q);
      float r6[6];             // [quantity][pair h]: point 2 h + parity
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int hh2 = 0; hh2 < 2; ++hh2) {
          const float se = patch_add_xor1(q[4 * c + 2 * hh2]), so = patch_add_xor1(q[4 * c + 2 * hh2 + 1]);
          r6[2 * c + hh2] = keep_even ? se : so;
        }
      patch_dpp_fence(r6);
      float r3[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float s0 = patch_add_xor2(r6[2 * c]), s1 = patch_add_xor2(r6[2 * c + 1]);
        r3[c] = keep_h0 ? s0 : s1;
      }
      patch_dpp_fence(r3);
#pragma unroll
      for (int c = 0; c < 3; ++c) r3[c] = patch_add_mirror8(r3[c]);
      if (rg_ok[ps] && j < kPatchP) {
        const float a = mine[j].w.z;
        const long long li = rowg[ps] * LP + l * kPatchP + j;
        grad_attn[li] = r3[0];
        *reinterpret_cast<float2 *>(grad_loc + 2 * li) =
            make_float2(r3[1] * (a * (float)plan.lv[l].W), r3[2] * (a * (float)plan.lv[l].H));
      }
    }
  }
  PATCH_STAMP(6);
#ifdef TILE2_STAMPS
  if (pst_out && tid == 0)
    for (int i = 0; i < pst_n; ++i) pst_out[i] = pst[i];
#endif
#undef PATCH_STAMP
}

// The far taps: the samples the query side put on the far list, their marked taps added to grad_value with HBM float atomics
// (the reference's own formulation, ms_deform_im2col_cuda.cuh:125-152, for these few).  Launched AFTER the tile kernels, which
// have written every pixel by then.  16 lanes per sample, lane i adding channels {i, i + 16, i + 32} of w_k * grad_out: every
// atomic instruction adds 64 contiguous bytes per row (the shape the memory-side atomic units take at full rate).  A fixed
// grid walks the list (its length is only known on the device); with no far taps the kernel is ~3 us of launch.
constexpr int kFarBlocks = 1024;
template <bool GO_BF16>
__global__ __launch_bounds__(kPatchThreads) void msda_bwd_d48_far_kernel(
    const void *__restrict__ grad_out, const float *__restrict__ loc, const float *__restrict__ attn, CoreDims d,
    PatchPlan plan, float *__restrict__ grad_value) {
  const unsigned count = *plan.far_count;
  const int tid = threadIdx.x, ai = tid & 15;
  const unsigned LP = (unsigned)(d.L * kPatchP);
  const unsigned gv_bytes = (unsigned)((size_t)d.N * d.S * d.M * kPatchRowBytes);
  const auto gsrc = __builtin_amdgcn_make_buffer_rsrc(grad_value, 0, (int)gv_bytes, 0x00020000);
  for (unsigned e = blockIdx.x * (kPatchThreads / 16) + (tid >> 4); e < count; e += gridDim.x * (kPatchThreads / 16)) {
    const uint2 ent = plan.far_list[e];
    const unsigned li = ent.x, bits = ent.y;
    const unsigned row = li / LP, s = li - row * LP;            // row = (n Lq + q) M + m
    const int l = (int)(s / kPatchP);
    const unsigned nq = row / (unsigned)d.M, m = row - nq * (unsigned)d.M, n = nq / (unsigned)d.Lq;
    PatchLevel lvl = plan.lv[0];
#pragma unroll
    for (int i = 1; i < kPatchMaxLevels; ++i)
      if (l == i) lvl = plan.lv[i];
    const float2 xy = *reinterpret_cast<const float2 *>(loc + 2 * (size_t)li);
    const float a = attn[li];
    const float y = px_coord(xy.y, lvl.H), x = px_coord(xy.x, lvl.W);
    const float yf = floorf(y), xf = floorf(x);
    const int y0 = (int)yf, x0 = (int)xf;
    const float lh = y - yf, lw = x - xf, hh = 1.f - lh, hw = 1.f - lw;
    const float wk[4] = {hh * hw * a, hh * lw * a, lh * hw * a, lh * lw * a};
    float g0, g1, g2;
    if constexpr (GO_BF16) {
      const uint16_t *gr = reinterpret_cast<const uint16_t *>(grad_out) + (size_t)row * kD48;
      g0 = __uint_as_float((unsigned)gr[ai] << 16); g1 = __uint_as_float((unsigned)gr[16 + ai] << 16);
      g2 = __uint_as_float((unsigned)gr[32 + ai] << 16);
    } else {
      const float *gr = reinterpret_cast<const float *>(grad_out) + (size_t)row * kD48;
      g0 = gr[ai]; g1 = gr[16 + ai]; g2 = gr[32 + ai];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      // (a far tap is inside the map by construction: the query side only lists those)
      const unsigned pix = (unsigned)(lvl.start + (y0 + (k >> 1)) * lvl.W + x0 + (k & 1));
      const unsigned o = ((bits >> k) & 1u) ? msda_row_base(d, n, m, kPatchRowBytes) + pix * msda_px_stride(d, kPatchRowBytes) + 4u * ai : kOobOffset;
      __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(wk[k] * g0, gsrc, o, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(wk[k] * g1, gsrc, o + 64u, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(wk[k] * g2, gsrc, o + 128u, 0, 0);
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Backward, grad_value side: one workgroup per (n, m, tile)
// ------------------------------------------------------------------------------------------------------------------
constexpr int kTile2MaxPx = 256;
constexpr int kTile2HitList = 1536;   // hits expanded per pass over a tile (a tile of the 600x800 geometry has ~700)
template <int HITCAP, bool GO_BF16> struct Tile2Lds {
  // grad_out rows of the round's hits, float32 (bf16 rows are widened once while they are staged: 6 conversions per
  // row and lane instead of 6 per tap)
  __attribute__((aligned(16))) unsigned char rows[HITCAP * 192];
  int2 tap[HITCAP * 16];              // sorted taps of the round: (byte offset of the hit's row in `rows`, weight bits)
  unsigned hits[kTile2HitList];       // queries with a mark in this tile, packed (level << 30 | qy << 15 | qx): the decode of a
                                      // sample needs the query's grid position, not only its index
  __attribute__((aligned(16))) int cntw[4][kTile2MaxPx];           // per wave and pixel: taps counted, then the wave's first position
  int off[kTile2MaxPx + 1];           // exclusive prefix over pixels
  int wsum[4];
  int total_hits;
#ifdef TILE2_STAMPS
  unsigned long long st[128];         // (stamps are kept in LDS and copied out at the end: a global store per stamp made hipcc
                                      //  drain the vector-memory counter at every stamp, i.e. undid the prefetches)
#endif
};


__device__ __forceinline__ int block_incl_scan(int v, int *wsum, int tid) {
  int inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int up = __shfl_up(inc, o, 64);
    inc += ((tid & 63) >= o) ? up : 0;
  }
  if ((tid & 63) == 63) wsum[tid >> 6] = inc;
  lds_barrier();
  int before = 0;
  for (int w = 0; w < (tid >> 6); ++w) before += wsum[w];
  return before + inc;
}

// one tap: acc += w * row, the row held in LDS as bf16 (96 B) or f32 (192 B); lane j owns channels 4j..4j+3, 32+2j, 33+2j
template <bool GO_BF16>
__device__ __forceinline__ void tile2_fma(float (&a6)[6], float w, const unsigned char *rows, int row_off, int j) {
  const Row6 v = lds_row(rows, (unsigned)row_off, j);
  a6[0] = fmaf(w, v.a.x, a6[0]); a6[1] = fmaf(w, v.a.y, a6[1]); a6[2] = fmaf(w, v.a.z, a6[2]);
  a6[3] = fmaf(w, v.a.w, a6[3]); a6[4] = fmaf(w, v.b.x, a6[4]); a6[5] = fmaf(w, v.b.y, a6[5]);
}

// The query levels' geometry and the size ratios of the tile's level over them, read from the kernel argument ONCE (uniform
// scalar loads) and pinned in registers.  (Round 2 selected `plan.lv[i].start` etc. per hit: hipcc turned the chain of
// selected loads into a load from a selected ADDRESS -- two dependent vector loads from the kernel-argument segment in front
// of every gather of the loc / attn / grad_out rows of a hit; found in the ISA in round 3.)
struct Tile2Levels {
  int start[kPatchMaxLevels], W[kPatchMaxLevels];
  float rw[kPatchMaxLevels], rh[kPatchMaxLevels];
};
__device__ __forceinline__ Tile2Levels tile2_levels(const PatchPlan &p, int l) {
  Tile2Levels t;
#pragma unroll
  for (int i = 0; i < kPatchMaxLevels; ++i) {
    t.start[i] = p.lv[i].start; t.W[i] = p.lv[i].W; t.rw[i] = p.rw[l][i]; t.rh[i] = p.rh[l][i];
    asm volatile("" : "+s"(t.start[i]), "+s"(t.W[i]), "+s"(t.rw[i]), "+s"(t.rh[i]));
  }
  return t;
}
// (level, qy, qx) of a packed hit -> query index
__device__ __forceinline__ int hit_query(const Tile2Levels &t, unsigned hit, int &lq, int &qy, int &qx) {
  lq = (int)(hit >> 30); qy = (int)((hit >> 15) & 0x7fffu); qx = (int)(hit & 0x7fffu);
  int start = t.start[0], Wq = t.W[0];
#pragma unroll
  for (int i = 1; i < kPatchMaxLevels; ++i) {
    start = lq == i ? t.start[i] : start;
    Wq = lq == i ? t.W[i] : Wq;
  }
  return start + qy * Wq + qx;
}
__device__ __forceinline__ int hit_query(const Tile2Levels &t, unsigned hit) {
  int lq, qy, qx;
  return hit_query(t, hit, lq, qy, qx);
}
// size ratios of the tile's level over the hit's query level lq (per lane)
__device__ __forceinline__ void hit_ratios(const Tile2Levels &t, int lq, float &rw, float &rh) {
  rw = t.rw[0]; rh = t.rh[0];
#pragma unroll
  for (int i = 1; i < kPatchMaxLevels; ++i) {
    rw = lq == i ? t.rw[i] : rw;
    rh = lq == i ? t.rh[i] : rh;
  }
}

// phase stamps of every 61st workgroup (wave 0): 128 slots per sampled workgroup, slot id in the top byte
#ifdef TILE2_STAMPS
#define TILE2_STAMP(slot) do { if (stamp_buf && tid == 0 && stamp_n < 128) { S.st[stamp_n] = ((unsigned long long)(slot) << 56) | (__builtin_amdgcn_s_memtime() & 0x00ffffffffffffffull); ++stamp_n; } } while (0)
#else
#define TILE2_STAMP(slot) do { } while (0)
#endif

template <int HITCAP, bool GO_BF16>
__global__ __launch_bounds__(kPatchThreads) __attribute__((amdgpu_waves_per_eu(3, 3))) void msda_bwd_d48_tile2_kernel(
    const void *__restrict__ grad_out, const float *__restrict__ loc, const float *__restrict__ attn, CoreDims d,
    PatchPlan plan, float *__restrict__ grad_value) {
  __shared__ Tile2Lds<HITCAP, GO_BF16> S;
  constexpr int kItems = HITCAP * kPatchP / kPatchThreads;
  constexpr int kRowB = GO_BF16 ? 96 : 192;
  // XCD-major: all tiles of one (n, m) go to ONE XCD back to back, so that (n, m)'s grad_out rows / locations (~1-2 MB),
  // which every tile a query's samples touch reads again, stay in that XCD's L2; an XCD takes CONSECUTIVE (n, m) pairs,
  // i.e. the heads of one sample one after the other: a head's 96-byte grad_out row shares its 128-byte lines with its
  // neighbours' rows (the rows of a query's 8 heads are contiguous).  Whole backward 0.774 -> 0.76 ms (round 3; walking the
  // heads innermost -- the 8 heads of ONE tile together -- measured 0.774 / 0.87 / 0.92 ms at sigma 0 / 3 / 8 px against
  // 0.76 / 0.88 / 0.91 ms for this order)
  const int tiles = plan.total_tiles;
  const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
  const int nm = xcd * ((d.N * d.M + 7) >> 3) + jb / tiles;
  if (nm >= d.N * d.M) return;
  const int tile_id = jb % tiles;
  const int m = nm % d.M, n = nm / d.M;
  int l = 0;
  for (int i = 1; i < plan.L; ++i) l = (tile_id >= plan.lv[i].tile_base) ? i : l;
  const PatchLevel me = plan.lv[l];
  const Tile2Levels lv = tile2_levels(plan, l);
  const int t = tile_id - me.tile_base;
  const int edge = 1 << me.shift, tpx = edge * edge, tsh = 2 * me.shift;
  const int tyi = t / me.ntx, txi = t - tyi * me.ntx;
  const int ty0 = tyi << me.shift, tx0 = txi << me.shift;
  const int tid = threadIdx.x, wave = tid >> 6;
  const int LP = d.L * kPatchP;
  const size_t row_base = (size_t)n * d.Lq;
#ifdef TILE2_STAMPS
  unsigned long long *stamp_buf = (plan.stamps && blockIdx.x % 61 == 0 && blockIdx.x / 61 < 256) ? plan.stamps + (blockIdx.x / 61) * 128 : nullptr;
  int stamp_n = 0;
#endif
  TILE2_STAMP(0);

  // ---- A. this thread's candidate block: its mark word ----
  unsigned long long mask = 0ull;
  int c_lq = 0, c_by = 0, c_bx = 0;
  {
    int c = tid;
    const unsigned long long *slab = plan.marks + ((long long)n * d.M + m) * plan.words_per_nm + plan.lvl_base[l] +
                                     (long long)t * plan.tstride[l];
    for (int lq = 0; lq < plan.L; ++lq) {
      int bx0, bx1, by0, by1;
      patch_tile_cand(plan, l, lq, ty0, tx0, edge, bx0, bx1, by0, by1);
      const int cw = min(bx1 - bx0 + 1, plan.cbw[l][lq]), ch = min(by1 - by0 + 1, plan.cbh[l][lq]);
      const int cnt = (cw > 0 && ch > 0) ? cw * ch : 0;
      if (c >= 0 && c < cnt) {
        const int dy = c / cw, dx = c - dy * cw;
        mask = slab[plan.coff[l][lq] + dy * plan.cbw[l][lq] + dx];
        c_lq = lq; c_by = by0 + dy; c_bx = bx0 + dx;
      }
      c -= cnt;          // (negative once this thread has found its block)
    }
  }
  const int my_cnt = __popcll(mask);
  const int my_excl = block_incl_scan(my_cnt, S.wsum, tid) - my_cnt;
  if (tid == kPatchThreads - 1) S.total_hits = my_excl + my_cnt;
  reinterpret_cast<u32x4 *>(&S.cntw[0][0])[tid] = u32x4{0u, 0u, 0u, 0u};       // 4 x 256 counters = 256 x 16 B
  lds_barrier();
  const int total_hits = (plan.debug & 8) ? 0 : S.total_hits;
  TILE2_STAMP(1);
#ifdef TILE2_STAMPS
  if (stamp_buf && tid == 0) S.st[stamp_n++] = (200ull << 56) | (unsigned long long)total_hits;
#endif

  // accumulators: 32 groups of 8 lanes own the tile's pixels
  const int grp = tid >> 3, j = tid & 7;
  constexpr int kMaxU = kTile2MaxPx / 32;
  float acc[kMaxU][6];
#pragma unroll
  for (int u = 0; u < kMaxU; ++u)
#pragma unroll
    for (int c = 0; c < 6; ++c) acc[u][c] = 0.f;
  const unsigned char *go_nm = reinterpret_cast<const unsigned char *>(grad_out) + (row_base * d.M + m) * kRowB;
  const size_t q_stride = (size_t)d.M * kRowB;
  const float *loc_nm = loc + ((row_base * d.M + m) * LP + l * kPatchP + (tid & 3)) * 2;
  const float *attn_nm = attn + (row_base * d.M + m) * LP + l * kPatchP + (tid & 3);
  const size_t s_stride = (size_t)d.M * LP;

  for (int pass0 = 0; pass0 < total_hits; pass0 += kTile2HitList) {
    const int pass1 = min(pass0 + kTile2HitList, total_hits), np = pass1 - pass0;
    // ---- expand the marks into the hit list (order: candidate, then bit -- fixed) ----
    //      A wave takes its candidates with marks one after the other (a wave-uniform loop); lane i expands bit i = query
    //      (i >> 3, i & 7) of the block: its place is the candidate's prefix + the set bits below i.  (Round 2 had every
    //      thread walk its own word bit by bit: a serial loop of up to 64 trips in a few lanes, 13 % of a workgroup's life.)
    {
      unsigned long long todo = __ballot(my_cnt && my_excl < pass1 && my_excl + my_cnt > pass0);
      const int lane = tid & 63;
      while (todo) {
        const int src = __builtin_ctzll(todo);
        todo &= todo - 1;
        const unsigned mlo = __builtin_amdgcn_readlane((unsigned)mask, src);
        const unsigned mhi = __builtin_amdgcn_readlane((unsigned)(mask >> 32), src);
        const int ex = __builtin_amdgcn_readlane(my_excl, src);
        const unsigned lqv = __builtin_amdgcn_readlane(c_lq, src);
        const unsigned byv = __builtin_amdgcn_readlane(c_by, src), bxv = __builtin_amdgcn_readlane(c_bx, src);
        const bool bit = (((lane < 32 ? mlo : mhi) >> (lane & 31)) & 1u) != 0u;
        const int gi = ex + (int)__builtin_amdgcn_mbcnt_hi(mhi, __builtin_amdgcn_mbcnt_lo(mlo, 0u));
        if (bit && gi >= pass0 && gi < pass1)
          S.hits[gi - pass0] = (lqv << 30) | ((byv * kPatchB + (unsigned)(lane >> 3)) << 15) | (bxv * kPatchB + (unsigned)(lane & 7));
      }
    }
    lds_barrier();
    // this thread's samples of the first round (later rounds: fetched behind the previous round's accumulate phase)
    float2 n_xy[kItems];
    float n_a[kItems];
    auto fetch = [&](int lo_, int nh_) {
#pragma unroll
      for (int it = 0; it < kItems; ++it) {
        const int h = (tid + it * kPatchThreads) >> 2;
        n_xy[it] = make_float2(-4.f, -4.f); n_a[it] = 0.f;      // (a location outside every map: decodes to "not near")
        if (h < nh_ && !(plan.debug & 4)) {
          const size_t so = (size_t)hit_query(lv, S.hits[lo_ + h]) * s_stride;
          n_xy[it] = *reinterpret_cast<const float2 *>(loc_nm + 2 * so);
          n_a[it] = attn_nm[so];
        }
      }
    };
    fetch(0, min(HITCAP, np));
    TILE2_STAMP(2);

    for (int lo = 0; lo < np; lo += HITCAP) {
      const int nh = min(HITCAP, np - lo);
      TILE2_STAMP(3);
      // ---- the hits' grad_out rows -> LDS: float32 rows by LDS-DMA, bfloat16 rows through registers (issued here,
      //      widened and written behind decode / prefix / scatter) ----
      constexpr int kPieces = HITCAP * 6 / kPatchThreads;       // bf16: 16-B pieces (8 channels) per thread and round
      u32x4 pc[kPieces];
      if constexpr (GO_BF16) {
#pragma unroll
        for (int i = 0; i < kPieces; ++i) {
          const int g = tid + i * kPatchThreads, h = g / 6, piece = g - h * 6;
          pc[i] = u32x4{0u, 0u, 0u, 0u};
          if (h < nh && !(plan.debug & 1)) {
            pc[i] = *reinterpret_cast<const u32x4 *>(go_nm + (size_t)hit_query(lv, S.hits[lo + h]) * q_stride + piece * 16);
          }
        }
      } else if (!(plan.debug & 1)) {
        constexpr int gpr = kRowB / 16;
        const int G = nh * gpr;
        for (int g0 = 0; g0 < G; g0 += kPatchThreads) {
          const int g = g0 + tid;
          if (g < G) {
            const int h = g / gpr, piece = g - h * gpr;
            const unsigned char *src = go_nm + (size_t)hit_query(lv, S.hits[lo + h]) * q_stride + piece * 16;
            unsigned char *dst = S.rows + (size_t)(g - (tid & 63)) * 16;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
          }
        }
      }
      // ---- decode: item = (hit h, point p); rank every tap of this tile within its (wave, pixel) ----
      TILE2_STAMP(4);
      unsigned t_pix[kItems], t_ok[kItems];   // 4 x 8 bits: pixel of each tap; bit k: tap k is in this tile
      unsigned t_rank[kItems][2];             // 4 x 16 bits
      float t_w[kItems][4];
#pragma unroll
      for (int it = 0; it < kItems; ++it) {
        const int h = (tid + it * kPatchThreads) >> 2;
        t_pix[it] = 0u; t_ok[it] = 0u; t_rank[it][0] = t_rank[it][1] = 0u;
#pragma unroll
        for (int k = 0; k < 4; ++k) t_w[it][k] = 0.f;
        const float x = px_coord(n_xy[it].x, me.W), y = px_coord(n_xy[it].y, me.H);
        const bool inside = h < nh && (y > -1.f) && (x > -1.f) && (y < (float)me.H) && (x < (float)me.W);
        if (inside) {
          int lq, qy, qx;
          hit_query(lv, S.hits[lo + h], lq, qy, qx);
          float rwq, rhq;
          hit_ratios(lv, lq, rwq, rhq);
          if (near_anchor(x, y, anchor_from_ratio(qx, rwq), anchor_from_ratio(qy, rhq), plan.radius)) {
            const float a = n_a[it];
            const float yf = floorf(y), xf = floorf(x);
            const int y0 = (int)yf, x0 = (int)xf;
            const float lh = y - yf, lw = x - xf, hh = 1.f - lh, hw = 1.f - lw;
            const float w4[4] = {hh * hw * a, hh * lw * a, lh * hw * a, lh * lw * a};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const int ty = y0 + (k >> 1), tx = x0 + (k & 1);
              const bool in_map = ty >= 0 && ty <= me.H - 1 && tx >= 0 && tx <= me.W - 1;
              const bool mine = in_map && (ty >> me.shift) == tyi && (tx >> me.shift) == txi;
              if (mine) {
                const int pix = ((ty - ty0) << me.shift) + (tx - tx0);
                const int rank = atomicAdd(&S.cntw[wave][pix], 1);
                t_pix[it] |= (unsigned)pix << (8 * k);
                t_ok[it] |= 1u << k;
                t_rank[it][k >> 1] |= (unsigned)rank << (16 * (k & 1));
                t_w[it][k] = w4[k];
              }
            }
          }
        }
      }
      TILE2_STAMP(5);
      lds_barrier();     // ranks complete
      TILE2_STAMP(6);
      // ---- exclusive prefix over pixels; per wave the position of its first tap of every pixel ----
      {
        int c[4] = {0, 0, 0, 0};
        if (tid < tpx) {
#pragma unroll
          for (int w = 0; w < 4; ++w) c[w] = S.cntw[w][tid];
        }
        const int tot = c[0] + c[1] + c[2] + c[3];
        const int incl = block_incl_scan(tot, S.wsum, tid);
        if (tid < tpx) {
          int base = incl - tot;
          S.off[tid] = base;
#pragma unroll
          for (int w = 0; w < 4; ++w) { S.cntw[w][tid] = base; base += c[w]; }
        }
        if (tid == kPatchThreads - 1) S.off[tpx] = incl;      // (threads >= tpx carry the total)
      }
      lds_barrier();
      TILE2_STAMP(7);
      // ---- scatter the taps into pixel order ----
#pragma unroll
      for (int it = 0; it < kItems; ++it) {
        const int h = (tid + it * kPatchThreads) >> 2;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if ((t_ok[it] >> k) & 1u) {
            const unsigned pix = (t_pix[it] >> (8 * k)) & 0xffu;
            const int pos = S.cntw[wave][pix] + (int)((t_rank[it][k >> 1] >> (16 * (k & 1))) & 0xffffu);
            S.tap[pos] = make_int2(h * kPatchRowBytes, __float_as_int(t_w[it][k]));
          }
        }
      }
      TILE2_STAMP(8);
      if constexpr (GO_BF16) {
#pragma unroll
        for (int i = 0; i < kPieces; ++i) {
          const int g = tid + i * kPatchThreads, h = g / 6, piece = g - h * 6;
          if (h < nh) {
            f32x4 lo4, hi4;
            lo4.x = __uint_as_float(pc[i].x << 16); lo4.y = __uint_as_float(pc[i].x & 0xffff0000u);
            lo4.z = __uint_as_float(pc[i].y << 16); lo4.w = __uint_as_float(pc[i].y & 0xffff0000u);
            hi4.x = __uint_as_float(pc[i].z << 16); hi4.y = __uint_as_float(pc[i].z & 0xffff0000u);
            hi4.z = __uint_as_float(pc[i].w << 16); hi4.w = __uint_as_float(pc[i].w & 0xffff0000u);
            *reinterpret_cast<f32x4 *>(S.rows + h * 192 + piece * 32) = lo4;
            *reinterpret_cast<f32x4 *>(S.rows + h * 192 + piece * 32 + 16) = hi4;
          }
        }
      } else {
        vm_drain();      // this wave's share of the rows has landed
      }
      TILE2_STAMP(9);
      lds_barrier();     // taps sorted, every wave's rows in LDS
      TILE2_STAMP(10);
      // the next round's samples: in flight while this round accumulates
      if (lo + HITCAP < np) fetch(lo + HITCAP, min(HITCAP, np - lo - HITCAP));
      reinterpret_cast<u32x4 *>(&S.cntw[0][0])[tid] = u32x4{0u, 0u, 0u, 0u};                          // for the next round
      TILE2_STAMP(11);
      // ---- accumulate: every pixel by one group (tiles of < 32 pixels: 32 / tpx groups share a pixel's taps) ----
      if (!(plan.debug & 2)) {       // pixels one after the other, 4 taps per trip
        auto run = [&](float (&a6)[6], int b0, int e0) {
          const int last = e0 - 1;
          for (int e = b0; e < e0; e += 4) {
            const int2 r0 = S.tap[e], r1 = S.tap[min(e + 1, last)], r2 = S.tap[min(e + 2, last)], r3 = S.tap[min(e + 3, last)];
            const float w0 = __int_as_float(r0.y), w1 = e + 1 < e0 ? __int_as_float(r1.y) : 0.f;
            const float w2 = e + 2 < e0 ? __int_as_float(r2.y) : 0.f, w3 = e + 3 < e0 ? __int_as_float(r3.y) : 0.f;
            tile2_fma<GO_BF16>(a6, w0, S.rows, r0.x, j);
            tile2_fma<GO_BF16>(a6, w1, S.rows, r1.x, j);
            tile2_fma<GO_BF16>(a6, w2, S.rows, r2.x, j);
            tile2_fma<GO_BF16>(a6, w3, S.rows, r3.x, j);
          }
        };
        if (tpx >= 32) {
#pragma unroll
          for (int u = 0; u < kMaxU; ++u) {
            if (u < (tpx >> 5)) run(acc[u], S.off[grp + 32 * u], S.off[grp + 32 * u + 1]);
          }
        } else {
          const int pix = grp & (tpx - 1), s = grp >> tsh, gpp = 32 >> tsh;
          const int pb = S.off[pix], nt = S.off[pix + 1] - pb;
          run(acc[0], pb + (nt * s) / gpp, pb + (nt * (s + 1)) / gpp);
        }
      }
      TILE2_STAMP(12);
      lds_barrier();     // before the next round overwrites rows / taps
      TILE2_STAMP(13);
    }
  }
  TILE2_STAMP(14);

  // ---- tiles of fewer than 32 pixels: add the groups' partial sums in a fixed order ----
  if (tpx < 32) {
    float *comb = reinterpret_cast<float *>(S.tap);               // [32 groups][8 lanes][6]
    const int s = grp >> tsh, gpp = 32 >> tsh;
    if (s > 0) {
#pragma unroll
      for (int c = 0; c < 6; ++c) comb[(grp * 8 + j) * 6 + c] = acc[0][c];
    }
    lds_barrier();
    if (s == 0) {
      for (int o = 1; o < gpp; ++o) {
#pragma unroll
        for (int c = 0; c < 6; ++c) acc[0][c] += comb[(((o << tsh) + grp) * 8 + j) * 6 + c];
      }
    }
  }
  // ---- store the tile: plain stores (tiles are disjoint and cover the map; the far taps are added afterwards by
  //      msda_bwd_d48_far_kernel, so nothing was in grad_value before) ----
  const size_t gv_base = msda_row_base(d, (unsigned)n, (unsigned)m, kPatchRowBytes), gv_px = msda_px_stride(d, kPatchRowBytes);
#pragma unroll
  for (int u = 0; u < kMaxU; ++u) {
    const bool on = tpx >= 32 ? (u < (tpx >> 5)) : (u == 0 && (grp >> tsh) == 0);
    if (on) {
      const int pix = tpx >= 32 ? grp + 32 * u : (grp & (tpx - 1));
      const int ty = ty0 + (pix >> me.shift), tx = tx0 + (pix & (edge - 1));
      if (ty < me.H && tx < me.W) {
        float *dst = reinterpret_cast<float *>(reinterpret_cast<unsigned char *>(grad_value) + gv_base +
                                               (size_t)(me.start + ty * me.W + tx) * gv_px);
        *reinterpret_cast<f32x4 *>(dst + 4 * j) = f32x4{acc[u][0], acc[u][1], acc[u][2], acc[u][3]};
        *reinterpret_cast<float2 *>(dst + 32 + 2 * j) = make_float2(acc[u][4], acc[u][5]);
      }
    }
  }
  TILE2_STAMP(15);
#ifdef TILE2_STAMPS
  if (stamp_buf && tid == 0)
    for (int i = 0; i < stamp_n; ++i) stamp_buf[i] = S.st[i];
#endif
}

}  // namespace snipper
