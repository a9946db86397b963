// msda_d48_tilemm.cuh -- grad_value side of the owner-computes backward (msda_d48_patch.cuh) on the MATRIX pipe.
// gfx950 only; bfloat16 grad_out rows (the training step's case), any `value` storage type (this kernel never reads it).
//
// Why: msda_bwd_d48_tile2_kernel sums a tile's taps on the vector / LDS path: every round it ranks each tap within its
// pixel (LDS atomics), builds a prefix over pixels, scatters (row, weight) records into pixel order and lets 8-lane groups
// walk their pixels' lists.  Its own counters and phase stamps (profiles/r03_tile2_experiments.md) say it is neither
// HBM- nor LDS-bandwidth-bound but issue-bound on that bookkeeping: 51 % of a workgroup's life is rank + prefix + scatter
// + accumulate, whose trip count is the maximum over a wave's groups; seven restructurings of that loop did not move it.
//
// Here the scatter is made DENSE.  For a round of 64 hits (queries with a tap in the tile)
//     tile[pixel][channel] += sum_hit  Wt[pixel][hit] * G[hit][channel]
// where G is the hits' bfloat16 grad_out rows exactly as they lie in memory and Wt[pixel][hit] is the sum of the (bilinear
// x attention) weights of the hit's taps on that pixel -- a [pixels x 64] float32 matrix in LDS of which a hit's lanes
// write ONLY ITS OWN COLUMN: no atomics, no ranks, no prefix, no sort, no divergent trip counts.  The product runs on
// v_mfma_f32_16x16x32_bf16 with Wt split into bf16 hi + lo parts when its fragment is loaded (hi = rne(w), lo = rne(w - hi):
// relative error <= 2^-17 per weight, float32-class; G is bf16 already, so the products are exact and the sums float32).
// Most of the 48 x 4 096 multiply-adds per round multiply a structural zero: the matrix pipe is otherwise idle in this
// kernel and executes them in ~400 cycles per wave, against ~14 000 cycles of list bookkeeping per 128-hit round before.
// (north_star reserves MFMA for the dense projections; it also asks for choices "evidenced by rocprof", and the counters
// say the vector formulation is issue-bound at 7 % of the HBM roofline.  The vector kernel stays: float32 grad_out rows
// take it, and `snipper_msda_config.tile_kernel = 1` selects it for A/B runs.)
//
// Order of every float sum is fixed (hit list order = candidate block, then query bit; k order inside an MFMA; hi before
// lo), so grad_value stays BIT-REPRODUCIBLE from launch to launch for every tap the tiles own.
//
// LDS images (brute-forced against the lane groups of MI355X_MICROARCH.md, LDS table):
//   Wt  4 planes x [pixels][8 floats], plane stride pixels * 32 + 16 bytes: lane (pixel r = lane & 15, k-group g = lane >> 4)
//       of a Wt fragment reads the 32 contiguous bytes of pixel 16 pb + r in plane g with two ds_read_b128, conflict-free;
//   G   [32 hits][96 B] unpadded; B fragments by ds_read_b64_tr_b16, lane group g taking hit rows {4g..4g+3} and
//       {16+4g..16+4g+3} (the 8 rows a 32-lane half reads are consecutive: conflict-free at a 96-byte stride), so hit h
//       sits in plane (h >> 2) & 3, slot (h & 3) + 4 (h >> 4) of Wt -- the same k order on both operands.
//
// Non-finite grad_out: a NaN / inf in a hit's row reaches every pixel of the tile (0 x NaN), not only the pixels the hit
// taps as in the reference (/root/reference/models/ops/src/cuda/ms_deform_im2col_cuda.cuh:125-152); a step with a
// non-finite output gradient is lost either way.
//
// Semantics restated from ms_deform_im2col_cuda.cuh:87-159; no code is shared with it.
#pragma once
#include "gemm_bf16.cuh"
#include "msda_d48_patch.cuh"

namespace snipper {

constexpr int kT3Hits = 64;                         // hits per round = two k-steps of v_mfma_f32_16x16x32_bf16
constexpr int kT3List = 512;                        // hits expanded per pass over a tile
constexpr int kT3PlaneMax = kTile2MaxPx * 32 + 16;  // bytes of one Wt plane of a 256-pixel tile
constexpr int kT3RowB = 96;                         // one bfloat16 grad_out head row

// one hit = one query with a mark in this tile: what every use of it needs, computed ONCE when the marks are expanded (a
// wave expands one candidate block per trip, lane i = query i of the block, so the level's constants are uniform there)
struct T3Hit { int q; float ax, ay; int pad; };     // query index; its anchor in the tile's level

// WPX = pixels of the largest tile a kernel instance serves: 256 (16 x 16 tiles: 65 664 B of Wt, two workgroups per CU) or 64
// (8 x 8 and smaller: 16 512 B, 32 KB in all -- the kernel is LATENCY-bound: padded to one workgroup per CU it takes 1.72x
// as long (profiles/r04_tile3_ablations.txt), so the small tiles, 2/3 of all rounds, get an instance of their own at five
// workgroups per CU)
template <int WPX>
struct Tile3Lds {
  __attribute__((aligned(16))) unsigned char W[2 * 4 * (WPX * 32 + 16)];  // two k-steps x four planes
  __attribute__((aligned(16))) unsigned char G[kT3Hits * kT3RowB];    // 6 144 B
  __attribute__((aligned(16))) T3Hit hits[kT3List];                   // 8 192 B
  float trash[kPatchThreads];                                         // where the taps this tile does not own go
  __attribute__((aligned(16))) unsigned dirty[4];                     // per wave: (k-step, pixel block) pairs its hits tapped this round
  int wsum[4];
  int total_hits;
#ifdef T3_LDS_PAD       // occupancy experiment (diagnostic builds only): pad the footprint to one workgroup per CU
  unsigned char pad[T3_LDS_PAD];
#endif
};

__device__ __forceinline__ gemm_bf16x8 tile3_bfrag(const unsigned char *base) {
  typedef __attribute__((address_space(3))) gemm_bf16x4 lds_v4;
  const gemm_bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4 *)(base));
  const gemm_bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4 *)(base + 16 * kT3RowB));
  return gemm_bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

// 8 float32 weights -> their bf16 hi and lo parts (hi = rne(w), lo = rne(w - hi))
__device__ __forceinline__ void tile3_split(const f32x4 &r0, const f32x4 &r1, gemm_bf16x8 &hi, gemm_bf16x8 &lo) {
  const float w[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
  gemm_u32x4 h, q;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const unsigned hp = gemm_pack2(w[2 * i], w[2 * i + 1]);
    h[i] = hp;
    q[i] = gemm_pack2(w[2 * i] - __uint_as_float(hp << 16), w[2 * i + 1] - __uint_as_float(hp & 0xffff0000u));
  }
  hi = __builtin_bit_cast(gemm_bf16x8, h);
  lo = __builtin_bit_cast(gemm_bf16x8, q);
}

// LDS accesses of the Wt build as inline assembly: the ORDER of the four points' read-add-write passes is what makes the
// sums right, and hipcc may merge / reorder identical exec-masked bodies (it did: the four passes became one).
// (LDS float atomics -- ds_add_f32 without return, no wait at all -- were measured: 527 us per launch against 379 us for
//  the read-add-write passes; the LDS executes them an order of magnitude slower than plain accesses.)
__device__ __forceinline__ float t3_lds_read(unsigned addr) {
  float v;
  asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(addr) : "memory");
  return v;
}
__device__ __forceinline__ void t3_lds_write(unsigned addr, float v) {
  asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
// (the values pass THROUGH the wait: hipcc does not count inline-assembly loads, and nothing else would keep a use of
//  them below it)
__device__ __forceinline__ void t3_lds_wait(float (&v)[4]) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3])::"memory");
}

// NPB = 16-pixel blocks of the tile (16 x 16 tiles: 16; 8 x 8: 4; smaller: 1 block padded with never-written rows)
template <int NPB, typename LDS>
__device__ __forceinline__ void tile3_body(LDS &S, const unsigned char *__restrict__ grad_out,
                                           const float *__restrict__ loc, const float *__restrict__ attn,
                                           const CoreDims &d, const PatchPlan &plan, float *__restrict__ grad_value,
                                           int n, int m, int l, int t) {
  constexpr int NPX = 16 * NPB, PLANE = NPX * 32 + 16;
  constexpr int NACC = NPB >= 4 ? NPB / 4 : 1;          // pixel blocks per wave (16-pixel tiles: wave 0 only)
  const PatchLevel me = plan.lv[l];
  const Tile2Levels lv = tile2_levels(plan, l);
  const int edge = 1 << me.shift, tpx = edge * edge;
  const int tyi = t / me.ntx, txi = t - tyi * me.ntx;
  const int ty0 = tyi << me.shift, tx0 = txi << me.shift;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int LP = d.L * kPatchP;
  const size_t row_base = (size_t)n * d.Lq;
  const unsigned th = (unsigned)min(edge, me.H - ty0), tw_ = (unsigned)min(edge, me.W - tx0);     // the tile's extent inside the map

  // ---- A. this thread's candidate block: its mark word (as msda_bwd_d48_tile2_kernel) ----
  unsigned long long mask = 0ull;
  int c_pack = 0;          // this thread's candidate block: level << 28 | block row << 14 | block column (maps < 32 768 pixels a side)
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
        c_pack = (lq << 28) | ((by0 + dy) << 14) | (bx0 + dx);
      }
      c -= cnt;
    }
  }
  // The accumulators start from zero and the finished tile is stored plainly: the tiles cover the map, the taps no tile owns
  // are added afterwards (msda_bwd_d48_far_kernel), so grad_value is neither zeroed beforehand nor read here.  The product is
  // evaluated transposed (G^T as the A operand, Wt^T as B), so lane (column c = lane & 15, k-group g) holds the FOUR
  // CONSECUTIVE channels 16 cb + 4 g .. + 3 of pixel 16 pb + c: one 16-byte access per accumulator, the four lanes of a
  // pixel covering 64 contiguous bytes.
  const int r16 = lane & 15, g4 = lane >> 4;
  gemm_f32x4 acc[NACC][3];
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int cb = 0; cb < 3; ++cb) acc[i][cb] = gemm_f32x4{0.f, 0.f, 0.f, 0.f};
  // where accumulator i of this lane goes (computed when the tile is stored: pointers held across the rounds cost registers
  // this kernel does not have -- at 96 it spilled three, and the reload sat in front of every round's prefetch)
  auto tile_dst = [&](int i) -> float * {
    const int pix = 16 * (NPB >= 4 ? 4 * i + wave : 0) + r16;
    const int ty = ty0 + (pix >> me.shift), tx = tx0 + (pix & (edge - 1));
    const bool on = (NPB >= 4 || wave == 0) && pix < tpx && ty < me.H && tx < me.W;
    return on ? reinterpret_cast<float *>(reinterpret_cast<unsigned char *>(grad_value) + msda_row_base(d, (unsigned)n, (unsigned)m, kPatchRowBytes) +
                                         (size_t)(me.start + ty * me.W + tx) * msda_px_stride(d, kPatchRowBytes)) + 4 * g4 : nullptr;
  };
  // Wt starts as zeros and a column is cleared again by the lanes that wrote it; rows of G beyond a round's last hit are
  // written as zeros by the loader (0 x stale bits must not make a NaN).  (Placed between the mark load and its first use.)
  for (int i = tid; i < 2 * 4 * PLANE / 16; i += kPatchThreads) reinterpret_cast<u32x4 *>(S.W)[i] = u32x4{0u, 0u, 0u, 0u};
  for (int i = tid; i < kT3Hits * kT3RowB / 16; i += kPatchThreads) reinterpret_cast<u32x4 *>(S.G)[i] = u32x4{0u, 0u, 0u, 0u};
  const int my_cnt = __popcll(mask);
  const int my_excl = block_incl_scan(my_cnt, S.wsum, tid) - my_cnt;
  if (tid == kPatchThreads - 1) S.total_hits = my_excl + my_cnt;
  lds_barrier();
  const int total_hits = (plan.debug & 8) ? 0 : S.total_hits;      // (plan.debug: timing ablations, WRONG results)

  // roles of a round: item = (hit h, point p), its four taps; G piece g = (hit g / 6, 16-byte part g % 6), 384 per round
  const int h = tid >> 2, p = tid & 3, h5 = h & 31;
  const unsigned w_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)S.W;
  const unsigned wcol = w_lds + (unsigned)(((h >> 5) * 4 + ((h5 >> 2) & 3)) * PLANE + ((h5 & 3) + ((h5 >> 4) << 2)) * 4);
  const unsigned trash = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float *)(S.trash + tid);
  // loc / attn / grad_out of this sample (n): raw buffers with 32-bit offsets (the launcher checked the sizes)
  const unsigned q_loc = (unsigned)(d.M * LP) * 8u, q_attn = (unsigned)(d.M * LP) * 4u, q_go = (unsigned)d.M * kT3RowB;
  const auto loc_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(loc + row_base * d.M * LP * 2), 0,
                                                         (int)((unsigned)d.Lq * q_loc), 0x00020000);
  const auto attn_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(attn + row_base * d.M * LP), 0,
                                                          (int)((unsigned)d.Lq * q_attn), 0x00020000);
  const auto go_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char *>(grad_out + row_base * d.M * kT3RowB), 0,
                                                        (int)((unsigned)d.Lq * q_go), 0x00020000);
  const unsigned my_item = (unsigned)((m * LP + l * kPatchP + p) * 4);          // (x 2 for loc)
  const unsigned char *gfrag = S.G + (4 * g4 + ((lane >> 2) & 3)) * kT3RowB + 8 * (lane & 3);

  for (int pass0 = 0; pass0 < total_hits; pass0 += kT3List) {
    const int pass1 = min(pass0 + kT3List, total_hits), np = pass1 - pass0;
    // ---- expand the marks into the hit list (order: candidate, then bit -- fixed; see msda_bwd_d48_tile2_kernel): a wave
    //      takes its candidates with marks one after the other, lane i = query (i >> 3, i & 7) of the block ----
    {
      unsigned long long todo = __ballot(my_cnt && my_excl < pass1 && my_excl + my_cnt > pass0);
      while (todo) {
        const int src = __builtin_ctzll(todo);
        todo &= todo - 1;
        const unsigned mlo = __builtin_amdgcn_readlane((unsigned)mask, src);
        const unsigned mhi = __builtin_amdgcn_readlane((unsigned)(mask >> 32), src);
        const int ex = __builtin_amdgcn_readlane(my_excl, src);
        const int cpv = __builtin_amdgcn_readlane(c_pack, src);
        const int lqv = (int)((unsigned)cpv >> 28), byv = (cpv >> 14) & 0x3fff, bxv = cpv & 0x3fff;
        int start = lv.start[0], Wq = lv.W[0];
        float rwq = lv.rw[0], rhq = lv.rh[0];
#pragma unroll
        for (int i = 1; i < kPatchMaxLevels; ++i) {
          start = lqv == i ? lv.start[i] : start; Wq = lqv == i ? lv.W[i] : Wq;
          rwq = lqv == i ? lv.rw[i] : rwq; rhq = lqv == i ? lv.rh[i] : rhq;
        }
        const bool bit = (((lane < 32 ? mlo : mhi) >> (lane & 31)) & 1u) != 0u;
        const int gi = ex + (int)__builtin_amdgcn_mbcnt_hi(mhi, __builtin_amdgcn_mbcnt_lo(mlo, 0u));
        if (bit && gi >= pass0 && gi < pass1) {
          const int qy = byv * kPatchB + (lane >> 3), qx = bxv * kPatchB + (lane & 7);
          T3Hit r;
          r.q = start + qy * Wq + qx; r.ax = anchor_from_ratio(qx, rwq); r.ay = anchor_from_ratio(qy, rhq); r.pad = 0;
          S.hits[gi - pass0] = r;
        }
      }
    }
    lds_barrier();

    // this thread's item and G pieces of the first round (later rounds: in flight behind the previous round)
    float2 n_xy = make_float2(-4.f, -4.f), n_an = make_float2(0.f, 0.f);
    float n_a = 0.f;
    u32x4 n_g[2];
    auto fetch = [&](int lo_, int nh_) {
      n_xy = make_float2(-4.f, -4.f); n_a = 0.f;       // (a location outside every map decodes to "not near")
      if (h < nh_ && !(plan.debug & 4)) {
        const T3Hit r = S.hits[lo_ + h];
        n_an = make_float2(r.ax, r.ay);
        typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
        const u32x2_t xy = __builtin_amdgcn_raw_buffer_load_b64(loc_src, (unsigned)r.q * q_loc + 2u * my_item, 0, 0);
        n_xy = make_float2(__uint_as_float(xy[0]), __uint_as_float(xy[1]));
        n_a = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(attn_src, (unsigned)r.q * q_attn + my_item, 0, 0));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int g = tid + i * kPatchThreads, gh = g / 6, gpart = g - gh * 6;
        n_g[i] = u32x4{0u, 0u, 0u, 0u};
        if (gh < nh_ && !(plan.debug & 1))       // (gh < 64 follows: nh_ <= 64)
          n_g[i] = __builtin_amdgcn_raw_buffer_load_b128(go_src, (unsigned)S.hits[lo_ + gh].q * q_go + (unsigned)(m * kT3RowB + gpart * 16), 0, 0);
      }
    };
    fetch(0, min(kT3Hits, np));

    for (int lo = 0; lo < np; lo += kT3Hits) {
      // ---- decode this thread's item: the four taps of sample (h, p); a tap this tile does not own adds 0 to `trash` ----
      unsigned ta[4] = {trash, trash, trash, trash};
      float tw[4] = {0.f, 0.f, 0.f, 0.f};
      unsigned dm = 0u;        // bit 16 s + pb: this item taps pixel block pb in k-step s
      {
        const float x = px_coord(n_xy.x, me.W), y = px_coord(n_xy.y, me.H);
        const bool inside = (y > -1.f) && (x > -1.f) && (y < (float)me.H) && (x < (float)me.W);
        if (inside && near_anchor(x, y, n_an.x, n_an.y, plan.radius)) {
          const float yf = floorf(y), xf = floorf(x);
          const int y0 = (int)yf, x0 = (int)xf;
          const float lh = y - yf, lw = x - xf, hh = 1.f - lh, hw = 1.f - lw;
          const float w4[4] = {hh * hw * n_a, hh * lw * n_a, lh * hw * n_a, lh * lw * n_a};
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            // in the map AND in this tile <=> inside the tile's extent within the map (two unsigned compares)
            const unsigned ry = (unsigned)(y0 + (k >> 1) - ty0), rx = (unsigned)(x0 + (k & 1) - tx0);
            if (ry < th && rx < tw_) {
              const unsigned pix = (ry << me.shift) + rx;
              ta[k] = wcol + pix * 32u;
              tw[k] = w4[k];
              if constexpr (NPB >= 16) dm |= 1u << ((pix >> 4) + 16u * (unsigned)(h >> 5));
            }
          }
        }
      }
      // ---- this round's G rows -> LDS (the previous round's fragment reads ended at its second barrier) ----
      reinterpret_cast<u32x4 *>(S.G)[tid] = n_g[0];
      if (tid < kT3Hits * 6 - kPatchThreads) reinterpret_cast<u32x4 *>(S.G)[tid + kPatchThreads] = n_g[1];
      // ---- the next round's item and G pieces: in flight behind everything below ----
      if (lo + kT3Hits < np) fetch(lo + kT3Hits, min(kT3Hits, np - lo - kT3Hits));
      // ---- build Wt: the four points of a hit may tap the same pixel, so they add one after the other (the LDS executes
      //      a wave's instructions in order; point 0 finds its column clear and only writes); the taps of ONE point are
      //      four different pixels and different hits are different columns: no two lanes of an instruction meet.
      //      (The phase numbers are made opaque: hipcc otherwise sees four mutually exclusive branches on p and is free to
      //      run them in ANY order -- it built a decision tree that ran point 3 first and point 0's plain store third.) ----
#pragma unroll
      for (int ph = 0; ph < kPatchP; ++ph) {
        int phv = ph;
        asm volatile("" : "+s"(phv));
        if (p == phv) {
          if (ph == 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) t3_lds_write(ta[k], tw[k]);
          } else {
            float v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = t3_lds_read(ta[k]);
            t3_lds_wait(v);
#pragma unroll
            for (int k = 0; k < 4; ++k) t3_lds_write(ta[k], v[k] + tw[k]);
          }
        }
      }
      // which (k-step, pixel block) fragments are not all zeros: OR over the wave (DPP within rows of 16, then the 4 rows).
      // (16 x 16 tiles only: a small tile has one or two fragments per wave and round, the bookkeeping costs more than it saves)
      if constexpr (NPB >= 16) {
        unsigned v = dm;
        v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);     // quad_perm [1,0,3,2]
        v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);     // quad_perm [2,3,0,1]
        v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true);    // row_half_mirror
        v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, true);    // row_mirror
        const unsigned wv = __builtin_amdgcn_readlane(v, 0) | __builtin_amdgcn_readlane(v, 16) |
                            __builtin_amdgcn_readlane(v, 32) | __builtin_amdgcn_readlane(v, 48);
        if (lane == 0) S.dirty[wave] = wv;
      }
      lds_barrier();      // Wt, G and the dirty words complete
      // ---- tile^T += G^T . Wt^T on the matrix pipe ----
      if ((NPB >= 4 || wave == 0) && !(plan.debug & 2)) {
        unsigned dirty = 0xffffffffu;
        if constexpr (NPB >= 16) {
          const u32x4 dw = *reinterpret_cast<const u32x4 *>(S.dirty);
          dirty = __builtin_amdgcn_readfirstlane(dw.x | dw.y | dw.z | dw.w);
        } else if (min(kT3Hits, np - lo) <= 32) {
          dirty = 0x0000ffffu;                                        // a last round of <= 32 hits: no second k-step
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          if (((dirty >> (16 * s)) & 0xffffu) == 0u) continue;        // (covers a last round of <= 32 hits)
          gemm_bf16x8 bf[3];
#pragma unroll
          for (int cb = 0; cb < 3; ++cb) bf[cb] = tile3_bfrag(gfrag + s * 32 * kT3RowB + 32 * cb);
#pragma unroll
          for (int i = 0; i < NACC; ++i) {
            const int pb = NPB >= 4 ? 4 * i + wave : 0;
            if (!((dirty >> (16 * s + pb)) & 1u)) continue;           // every weight of this fragment is zero
            const unsigned char *wp = S.W + (s * 4 + g4) * PLANE + (16 * pb + r16) * 32;
            const f32x4 r0 = *reinterpret_cast<const f32x4 *>(wp), r1 = *reinterpret_cast<const f32x4 *>(wp + 16);
            gemm_bf16x8 ahi, alo;
            tile3_split(r0, r1, ahi, alo);
#pragma unroll
            for (int cb = 0; cb < 3; ++cb) acc[i][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[cb], ahi, acc[i][cb], 0, 0, 0);
#pragma unroll
            for (int cb = 0; cb < 3; ++cb) acc[i][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[cb], alo, acc[i][cb], 0, 0, 0);
          }
        }
      }
      lds_barrier();      // fragment reads done: columns may be cleared, G overwritten
#pragma unroll
      for (int k = 0; k < 4; ++k) t3_lds_write(ta[k], 0.f);
    }
    lds_barrier();        // before the next pass overwrites the hit list
  }

  // ---- store the tile (tiles are disjoint; the accumulators started from grad_value) ----
#pragma unroll
  for (int i = 0; i < NACC; ++i) {
    float *dst = tile_dst(i);
    if (dst) {
#pragma unroll
      for (int cb = 0; cb < 3; ++cb) *reinterpret_cast<gemm_f32x4 *>(dst + 16 * cb) = acc[i][cb];
    }
  }
}

// the tiles one kernel instance serves: levels whose tiles have <= WPX pixels (and, for WPX = 256, more than 64)
constexpr int kT3OrderMax = 768;
struct T3Class {
  int tiles;                        // per (n, m)
  int base[kPatchMaxLevels];        // class-local index of the level's first tile, or -1 when the level is not in the class
  // order[k] = class-local index of the k-th tile of the walk when `ordered` (t3_region_order below), else the walk is
  // tiles - 1 ... 0 (coarse levels first)
  int ordered;
  uint16_t order[kT3OrderMax];
};

// The walk of a class with several levels, REGION by region (round 5).  A query's grad_out / loc / attn rows are needed by
// the tiles around its reference point on EVERY level; walking level after level touched each row in as many separate
// time windows as there are levels, a sample's 19 MB apart (L2: 4 MB), so the tile kernels fetched 2.7 x their input from
// HBM even with the heads of a tile back to back.  Regions = the tiles of the class's coarsest level; a finer tile belongs
// to the region its centre falls in; the walk takes a region's coarse tile, then its finer tiles level by level, then the
// next region -- all uses of a row fall within one or two regions of the walk.
// MEASURED SLOWER than level after level (N = 8, 600 x 800, sigma 0 / 3 / 8 px: 198-202 against 177-181 us, 272 against 247-258,
// 328-336 against 306-310; profiles/r05_region_order_ab.txt): kept behind debug bit 128 for A/B runs only.
inline void t3_region_order(const PatchPlan &plan, T3Class &c) {
  c.ordered = 0;
  int lv[kPatchMaxLevels], nl = 0;
  for (int l = 0; l < plan.L; ++l)
    if (c.base[l] >= 0) lv[nl++] = l;
  if (nl < 2 || c.tiles > kT3OrderMax) return;
  for (int i = 1; i < nl; ++i)                      // coarse (fewest tiles) first; ties keep the higher level first
    for (int j = i; j > 0; --j) {
      const PatchLevel &a = plan.lv[lv[j - 1]], &b = plan.lv[lv[j]];
      if (b.ntx * b.nty < a.ntx * a.nty || (b.ntx * b.nty == a.ntx * a.nty && lv[j] > lv[j - 1])) std::swap(lv[j - 1], lv[j]);
      else break;
    }
  const PatchLevel &C = plan.lv[lv[0]];
  if (C.nty > 64 || C.ntx > 64) return;
  const long long ec = 1 << C.shift;
  // lo[i][r] = first tile row / column of level lv[i] whose centre lies in region row / column >= r (monotone in the tile index)
  int ylo[kPatchMaxLevels][66], xlo[kPatchMaxLevels][66];
  for (int i = 0; i < nl; ++i) {
    const PatchLevel &A = plan.lv[lv[i]];
    const long long e = 1 << A.shift;
    auto fill = [&](int *lo, int n_tiles, long long size_a, long long size_c, int n_regions) {
      int r = 0;
      lo[0] = 0;
      for (int t = 0; t < n_tiles; ++t) {
        long long rt = ((2 * t * e + e) * size_c) / (2 * size_a * ec);
        if (rt > n_regions - 1) rt = n_regions - 1;
        while (r < rt) lo[++r] = t;
      }
      while (r < n_regions) lo[++r] = n_tiles;
    };
    fill(ylo[i], A.nty, A.H, C.H, C.nty);
    fill(xlo[i], A.ntx, A.W, C.W, C.ntx);
  }
  int k = 0;
  for (int ry = 0; ry < C.nty; ++ry)
    for (int rx = 0; rx < C.ntx; ++rx)
      for (int i = 0; i < nl; ++i) {
        const PatchLevel &A = plan.lv[lv[i]];
        for (int ty = ylo[i][ry]; ty < ylo[i][ry + 1]; ++ty)
          for (int tx = xlo[i][rx]; tx < xlo[i][rx + 1]; ++tx) c.order[k++] = (uint16_t)(c.base[lv[i]] + ty * A.ntx + tx);
      }
  c.ordered = (k == c.tiles) ? 1 : 0;
}

// Which (n, m) pair and which tile workgroup jb of XCD `xcd` takes.  An XCD owns a contiguous run of (n, m) pairs (N * M / 8:
// at the bench geometry the eight heads of one sample).  Round 3-4 walked it pair by pair -- all tiles of (n, 0), then all
// tiles of (n, 1) ... -- so the eight workgroups that need the SAME query rows of grad_out / loc / attn (a query's eight
// 96-byte head rows are one 768-byte run: six 128-byte lines shared by neighbouring heads) ran a whole pair's worth of
// tiles apart, and by then the lines had left the 4 MB L2 (one sample's grad_out + loc + attn is 19 MB): FETCH_SIZE of the
// tile kernels was 5.4x the algorithmic input.  Round 5: TILE-major -- the pairs of the XCD take the
// same tile back to back, so a tile's hit rows are fetched once for all heads.  Measured (N = 8, head-major bf16 value, rocprofv3,
// sigma 0 / 3 px): the 8 x 8-tile instance 162 -> 119 us / 226 -> 178 us; the 16 x 16-tile instance (two workgroups per CU: 64 per
// XCD were already most of two heads' tiles) 110 -> 112 / 152 -> 151 us: it keeps the pair-major walk.  (plan.debug bit 64 flips
// either choice for A/B runs: results are the same.)
__device__ __forceinline__ void t3_order(const CoreDims &d, const PatchPlan &plan, int tiles, int xcd, int jb, bool tile_major,
                                         int &nm, int &ct) {
  const int per_xcd = (d.N * d.M + 7) >> 3;
  if (tile_major == ((plan.debug & 64) != 0)) {
    nm = xcd * per_xcd + jb / tiles;
    ct = tiles - 1 - jb % tiles;
  } else {
    nm = xcd * per_xcd + jb % per_xcd;
    ct = tiles - 1 - jb / per_xcd;
  }
}

// ---- 16 x 16 tiles on EIGHT waves (512 threads) ----------------------------------------------------------------------
// The 256-pixel instance is held to two workgroups per CU by its 64 KB of Wt, and the kernel is latency-bound (above): a
// workgroup of 8 waves doubles the waves per CU at the same footprint.  Item = (hit, point, tap ROW): 64 x 4 x 2 = 512
// threads, two taps each; waves 0-3 multiply k-step 0 (hits 0-31) and waves 4-7 k-step 1 (hits 32-63) of every round, each
// for the pixel blocks {4 i + (wave & 3)}; the two halves meet once, through LDS, in a fixed order, before the tile is stored.
constexpr int kT3WideThreads = 512;
constexpr int kT3WideList = 512;                    // hits per pass (two workgroups fit the 160 KB of a CU: 80 336 B each; 448 with a
                                                    // trash word per thread left most 16 x 16 tiles a second, nearly empty pass: 114 -> 111 us)
struct Tile3LdsWide {
  __attribute__((aligned(16))) unsigned char W[2 * 4 * kT3PlaneMax];
  __attribute__((aligned(16))) unsigned char G[kT3Hits * kT3RowB];
  __attribute__((aligned(16))) T3Hit hits[kT3WideList];
  float trash[64];                                                    // (one word per LANE: what lands here is never read)
  __attribute__((aligned(16))) unsigned dirty[8];
  int wsum[8];
  int total_hits;
};

__device__ __forceinline__ void tile3_body_wide(Tile3LdsWide &S, const unsigned char *__restrict__ grad_out,
                                                const float *__restrict__ loc, const float *__restrict__ attn,
                                                const CoreDims &d, const PatchPlan &plan, float *__restrict__ grad_value,
                                                int n, int m, int l, int t) {
  constexpr int NPX = 256, PLANE = NPX * 32 + 16, NACC = 4;
  const PatchLevel me = plan.lv[l];
  const Tile2Levels lv = tile2_levels(plan, l);
  const int edge = 1 << me.shift, tpx = edge * edge;
  const int tyi = t / me.ntx, txi = t - tyi * me.ntx;
  const int ty0 = tyi << me.shift, tx0 = txi << me.shift;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int kstep = wave >> 2, wq = wave & 3;
  const int LP = d.L * kPatchP;
  const size_t row_base = (size_t)n * d.Lq;
  const unsigned th = (unsigned)min(edge, me.H - ty0), tw_ = (unsigned)min(edge, me.W - tx0);

  // ---- this thread's candidate block: its mark word (threads 256.. have none: <= 256 candidates per tile) ----
  unsigned long long mask = 0ull;
  int c_pack = 0;          // this thread's candidate block: level << 28 | block row << 14 | block column (maps < 32 768 pixels a side)
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
        c_pack = (lq << 28) | ((by0 + dy) << 14) | (bx0 + dx);
      }
      c -= cnt;
    }
  }
  const int r16 = lane & 15, g4 = lane >> 4;
  gemm_f32x4 acc[NACC][3];
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int cb = 0; cb < 3; ++cb) acc[i][cb] = gemm_f32x4{0.f, 0.f, 0.f, 0.f};
  auto tile_dst = [&](int i) -> float * {     // (computed when the tile is stored, see tile3_body)
    const int pix = 16 * (4 * i + wq) + r16;
    const int ty = ty0 + (pix >> me.shift), tx = tx0 + (pix & (edge - 1));
    const bool on = kstep == 0 && pix < tpx && ty < me.H && tx < me.W;         // (waves 0-3 own the store)
    return on ? reinterpret_cast<float *>(reinterpret_cast<unsigned char *>(grad_value) + msda_row_base(d, (unsigned)n, (unsigned)m, kPatchRowBytes) +
                                         (size_t)(me.start + ty * me.W + tx) * msda_px_stride(d, kPatchRowBytes)) + 4 * g4 : nullptr;
  };
  for (int i = tid; i < 2 * 4 * PLANE / 16; i += kT3WideThreads) reinterpret_cast<u32x4 *>(S.W)[i] = u32x4{0u, 0u, 0u, 0u};
  const int my_cnt = __popcll(mask);
  const int my_excl = block_incl_scan(my_cnt, S.wsum, tid) - my_cnt;
  if (tid == kT3WideThreads - 1) S.total_hits = my_excl + my_cnt;
  lds_barrier();
  const int total_hits = (plan.debug & 8) ? 0 : S.total_hits;

  // item = (hit h, point p, tap row dy): the taps (y0 + dy, x0) and (y0 + dy, x0 + 1); G piece = (hit tid / 6, part tid % 6)
  const int h = tid >> 3, p = (tid >> 1) & 3, dy = tid & 1, h5 = h & 31;
  const unsigned w_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)S.W;
  const unsigned wcol = w_lds + (unsigned)(((h >> 5) * 4 + ((h5 >> 2) & 3)) * PLANE + ((h5 & 3) + ((h5 >> 4) << 2)) * 4);
  const unsigned trash = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float *)(S.trash + lane);
  const unsigned q_loc = (unsigned)(d.M * LP) * 8u, q_attn = (unsigned)(d.M * LP) * 4u, q_go = (unsigned)d.M * kT3RowB;
  const auto loc_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(loc + row_base * d.M * LP * 2), 0,
                                                         (int)((unsigned)d.Lq * q_loc), 0x00020000);
  const auto attn_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(attn + row_base * d.M * LP), 0,
                                                          (int)((unsigned)d.Lq * q_attn), 0x00020000);
  const auto go_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char *>(grad_out + row_base * d.M * kT3RowB), 0,
                                                        (int)((unsigned)d.Lq * q_go), 0x00020000);
  const unsigned my_item = (unsigned)((m * LP + l * kPatchP + p) * 4);
  const int gh = tid / 6, gpart = tid - gh * 6;
  const bool g_thread = tid < kT3Hits * 6;
  const unsigned char *gfrag = S.G + (kstep * 32 + 4 * g4 + ((lane >> 2) & 3)) * kT3RowB + 8 * (lane & 3);

  for (int pass0 = 0; pass0 < total_hits; pass0 += kT3WideList) {
    const int pass1 = min(pass0 + kT3WideList, total_hits), np = pass1 - pass0;
    {
      unsigned long long todo = __ballot(my_cnt && my_excl < pass1 && my_excl + my_cnt > pass0);
      while (todo) {
        const int src = __builtin_ctzll(todo);
        todo &= todo - 1;
        const unsigned mlo = __builtin_amdgcn_readlane((unsigned)mask, src);
        const unsigned mhi = __builtin_amdgcn_readlane((unsigned)(mask >> 32), src);
        const int ex = __builtin_amdgcn_readlane(my_excl, src);
        const int cpv = __builtin_amdgcn_readlane(c_pack, src);
        const int lqv = (int)((unsigned)cpv >> 28), byv = (cpv >> 14) & 0x3fff, bxv = cpv & 0x3fff;
        int start = lv.start[0], Wq = lv.W[0];
        float rwq = lv.rw[0], rhq = lv.rh[0];
#pragma unroll
        for (int i = 1; i < kPatchMaxLevels; ++i) {
          start = lqv == i ? lv.start[i] : start; Wq = lqv == i ? lv.W[i] : Wq;
          rwq = lqv == i ? lv.rw[i] : rwq; rhq = lqv == i ? lv.rh[i] : rhq;
        }
        const bool bit = (((lane < 32 ? mlo : mhi) >> (lane & 31)) & 1u) != 0u;
        const int gi = ex + (int)__builtin_amdgcn_mbcnt_hi(mhi, __builtin_amdgcn_mbcnt_lo(mlo, 0u));
        if (bit && gi >= pass0 && gi < pass1) {
          const int qy = byv * kPatchB + (lane >> 3), qx = bxv * kPatchB + (lane & 7);
          T3Hit r;
          r.q = start + qy * Wq + qx; r.ax = anchor_from_ratio(qx, rwq); r.ay = anchor_from_ratio(qy, rhq); r.pad = 0;
          S.hits[gi - pass0] = r;
        }
      }
    }
    lds_barrier();

    float2 n_xy = make_float2(-4.f, -4.f), n_an = make_float2(0.f, 0.f);
    float n_a = 0.f;
    u32x4 n_g = u32x4{0u, 0u, 0u, 0u};
    auto fetch = [&](int lo_, int nh_) {
      n_xy = make_float2(-4.f, -4.f); n_a = 0.f;
      if (h < nh_ && !(plan.debug & 4)) {
        const T3Hit r = S.hits[lo_ + h];
        n_an = make_float2(r.ax, r.ay);
        typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
        const u32x2_t xy = __builtin_amdgcn_raw_buffer_load_b64(loc_src, (unsigned)r.q * q_loc + 2u * my_item, 0, 0);
        n_xy = make_float2(__uint_as_float(xy[0]), __uint_as_float(xy[1]));
        n_a = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(attn_src, (unsigned)r.q * q_attn + my_item, 0, 0));
      }
      n_g = u32x4{0u, 0u, 0u, 0u};
      if (g_thread && gh < nh_ && !(plan.debug & 1))
        n_g = __builtin_amdgcn_raw_buffer_load_b128(go_src, (unsigned)S.hits[lo_ + gh].q * q_go + (unsigned)(m * kT3RowB + gpart * 16), 0, 0);
    };
    fetch(0, min(kT3Hits, np));

    for (int lo = 0; lo < np; lo += kT3Hits) {
      unsigned ta[2] = {trash, trash};
      float tw[2] = {0.f, 0.f};
      unsigned dm = 0u;
      {
        const float x = px_coord(n_xy.x, me.W), y = px_coord(n_xy.y, me.H);
        const bool inside = (y > -1.f) && (x > -1.f) && (y < (float)me.H) && (x < (float)me.W);
        if (inside && near_anchor(x, y, n_an.x, n_an.y, plan.radius)) {
          const float yf = floorf(y), xf = floorf(x);
          const int y0 = (int)yf, x0 = (int)xf;
          const float lh = y - yf, lw = x - xf, hh = 1.f - lh, hw = 1.f - lw;
          const float wr = dy ? lh : hh;
          const float w2[2] = {wr * hw * n_a, wr * lw * n_a};
          const unsigned ry = (unsigned)(y0 + dy - ty0);
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            const unsigned rx = (unsigned)(x0 + k - tx0);
            if (ry < th && rx < tw_) {         // in the map AND in this tile
              const unsigned pix = (ry << me.shift) + rx;
              ta[k] = wcol + pix * 32u;
              tw[k] = w2[k];
              dm |= 1u << ((pix >> 4) + 16u * (unsigned)(h >> 5));
            }
          }
        }
      }
      if (g_thread) reinterpret_cast<u32x4 *>(S.G)[tid] = n_g;
      if (lo + kT3Hits < np) fetch(lo + kT3Hits, min(kT3Hits, np - lo - kT3Hits));
      // the four points of a hit add one after the other (see tile3_body); point 0 finds its column clear
#pragma unroll
      for (int ph = 0; ph < kPatchP; ++ph) {
        int phv = ph;
        asm volatile("" : "+s"(phv));
        if (p == phv) {
          if (ph == 0) {
            t3_lds_write(ta[0], tw[0]); t3_lds_write(ta[1], tw[1]);
          } else {
            float v[4] = {t3_lds_read(ta[0]), t3_lds_read(ta[1]), 0.f, 0.f};
            t3_lds_wait(v);
            t3_lds_write(ta[0], v[0] + tw[0]); t3_lds_write(ta[1], v[1] + tw[1]);
          }
        }
      }
      {
        unsigned v = dm;
        v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);
        v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);
        v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true);
        v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, true);
        const unsigned wv = __builtin_amdgcn_readlane(v, 0) | __builtin_amdgcn_readlane(v, 16) |
                            __builtin_amdgcn_readlane(v, 32) | __builtin_amdgcn_readlane(v, 48);
        if (lane == 0) S.dirty[wave] = wv;
      }
      lds_barrier();
      if (!(plan.debug & 2)) {
        const u32x4 d0 = *reinterpret_cast<const u32x4 *>(S.dirty), d1 = *reinterpret_cast<const u32x4 *>(S.dirty + 4);
        const unsigned dirty = __builtin_amdgcn_readfirstlane(d0.x | d0.y | d0.z | d0.w | d1.x | d1.y | d1.z | d1.w) >> (16 * kstep);
        if (dirty & 0xffffu) {
          gemm_bf16x8 bf[3];
#pragma unroll
          for (int cb = 0; cb < 3; ++cb) bf[cb] = tile3_bfrag(gfrag + 32 * cb);
#pragma unroll
          for (int i = 0; i < NACC; ++i) {
            const int pb = 4 * i + wq;
            if (!((dirty >> pb) & 1u)) continue;
            const unsigned char *wp = S.W + (kstep * 4 + g4) * PLANE + (16 * pb + r16) * 32;
            const f32x4 r0 = *reinterpret_cast<const f32x4 *>(wp), r1 = *reinterpret_cast<const f32x4 *>(wp + 16);
            gemm_bf16x8 ahi, alo;
            tile3_split(r0, r1, ahi, alo);
#pragma unroll
            for (int cb = 0; cb < 3; ++cb) acc[i][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[cb], ahi, acc[i][cb], 0, 0, 0);
#pragma unroll
            for (int cb = 0; cb < 3; ++cb) acc[i][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[cb], alo, acc[i][cb], 0, 0, 0);
          }
        }
      }
      lds_barrier();
      t3_lds_write(ta[0], 0.f); t3_lds_write(ta[1], 0.f);
    }
    lds_barrier();
  }

  // ---- the two k-step halves meet (waves 4-7 park theirs in the Wt region, free by now), then waves 0-3 store the tile ----
  f32x4 *park = reinterpret_cast<f32x4 *>(S.W);
  if (kstep == 1) {
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
      for (int cb = 0; cb < 3; ++cb) park[((wq * NACC + i) * 3 + cb) * 64 + lane] = acc[i][cb];
  }
  lds_barrier();
  if (kstep == 0) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      float *dst = tile_dst(i);
      if (dst) {
#pragma unroll
        for (int cb = 0; cb < 3; ++cb)
          *reinterpret_cast<gemm_f32x4 *>(dst + 16 * cb) = acc[i][cb] + park[((wq * NACC + i) * 3 + cb) * 64 + lane];
      }
    }
  }
}

__global__ __launch_bounds__(kT3WideThreads) __attribute__((amdgpu_waves_per_eu(4, 4)))
void msda_bwd_d48_tile3_wide_kernel(const void *__restrict__ grad_out, const float *__restrict__ loc, const float *__restrict__ attn,
                                    CoreDims d, PatchPlan plan, T3Class cls, float *__restrict__ grad_value) {
  __shared__ Tile3LdsWide S;
  const int tiles = cls.tiles;
  const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
  int nm, ct;
  t3_order(d, plan, tiles, xcd, jb, false, nm, ct);
  if (nm >= d.N * d.M) return;
  const int m = nm % d.M, n = nm / d.M;
  int l = 0, t = 0;
  for (int i = 0; i < plan.L; ++i)
    if (cls.base[i] >= 0 && ct >= cls.base[i]) { l = i; t = ct - cls.base[i]; }
  tile3_body_wide(S, reinterpret_cast<const unsigned char *>(grad_out), loc, attn, d, plan, grad_value, n, m, l, t);
}

#ifndef T3_SMALL_WAVES
#define T3_SMALL_WAVES 5
#endif
template <int WPX>
__global__ __launch_bounds__(kPatchThreads) __attribute__((amdgpu_waves_per_eu(WPX == 256 ? 2 : T3_SMALL_WAVES, WPX == 256 ? 2 : T3_SMALL_WAVES)))
void msda_bwd_d48_tile3_kernel(const void *__restrict__ grad_out, const float *__restrict__ loc, const float *__restrict__ attn,
                               CoreDims d, PatchPlan plan, T3Class cls, float *__restrict__ grad_value) {
  __shared__ Tile3Lds<WPX> S;
  // XCD-major walk of the (n, m) pairs, as msda_bwd_d48_tile2_kernel; within an (n, m) the tiles of the COARSE levels go
  // first: they are reached by the most queries, i.e. they are the longest-running workgroups
  const int tiles = cls.tiles;
  const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
  int nm, ct;
  t3_order(d, plan, tiles, xcd, jb, WPX != 256, nm, ct);
  if (nm >= d.N * d.M) return;
  if (cls.ordered) ct = cls.order[tiles - 1 - ct];
  const int m = nm % d.M, n = nm / d.M;
  int l = 0, t = 0;
  for (int i = 0; i < plan.L; ++i)
    if (cls.base[i] >= 0 && ct >= cls.base[i]) { l = i; t = ct - cls.base[i]; }
  const unsigned char *go = reinterpret_cast<const unsigned char *>(grad_out);
  // one body per instance (round 5; the 64-pixel instance used to carry the 16-pixel body as well and spilled 2 VGPRs + 4 SGPRs
  // at its 96-register budget): 256 = 16 x 16 tiles, 64 = 8 x 8 tiles, 16 = 4 x 4 and smaller
  if constexpr (WPX == 256) tile3_body<16>(S, go, loc, attn, d, plan, grad_value, n, m, l, t);
  else if constexpr (WPX == 64) tile3_body<4>(S, go, loc, attn, d, plan, grad_value, n, m, l, t);
  else tile3_body<1>(S, go, loc, attn, d, plan, grad_value, n, m, l, t);
}

}  // namespace snipper
