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
//   Wt  4 planes x [pixels][8 floats], plane stride pixels * 32 + 16 bytes: lane (row r = lane & 15, k-group g = lane >> 4)
//       of an A fragment reads the 32 contiguous bytes of pixel 16 pb + r in plane g with two ds_read_b128, conflict-free;
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
constexpr int kT3PlaneMax = kTile2MaxPx * 32 + 16;  // bytes of one Wt plane of a 256-pixel tile
constexpr int kT3RowB = 96;                         // one bfloat16 grad_out head row

struct Tile3Lds {
  __attribute__((aligned(16))) unsigned char W[2 * 4 * kT3PlaneMax];  // 65 664 B: two k-steps x four planes
  __attribute__((aligned(16))) unsigned char G[kT3Hits * kT3RowB];    // 6 144 B
  float trash[kPatchThreads];                                         // where the taps this tile does not own go
  unsigned hits[kTile2HitList];                                       // 6 144 B
  int wsum[4];
  int total_hits;
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
// sums right, and hipcc may merge / reorder identical exec-masked bodies (it did: the four passes became one)
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
template <int NPB>
__device__ __forceinline__ void tile3_body(Tile3Lds &S, const unsigned char *__restrict__ grad_out,
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

  // ---- A. this thread's candidate block: its mark word (as msda_bwd_d48_tile2_kernel) ----
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
      c -= cnt;
    }
  }
  const int my_cnt = __popcll(mask);
  const int my_excl = block_incl_scan(my_cnt, S.wsum, tid) - my_cnt;
  if (tid == kPatchThreads - 1) S.total_hits = my_excl + my_cnt;
  // Wt starts as zeros and a column is cleared again by the lanes that wrote it; rows of G beyond a round's last hit are
  // written as zeros by the loader (0 x stale bits must not make a NaN)
  for (int i = tid; i < 2 * 4 * PLANE / 16; i += kPatchThreads) reinterpret_cast<u32x4 *>(S.W)[i] = u32x4{0u, 0u, 0u, 0u};
  for (int i = tid; i < kT3Hits * kT3RowB / 16; i += kPatchThreads) reinterpret_cast<u32x4 *>(S.G)[i] = u32x4{0u, 0u, 0u, 0u};
  lds_barrier();
  const int total_hits = S.total_hits;

  gemm_f32x4 acc[NACC][3];
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int cb = 0; cb < 3; ++cb) acc[i][cb] = gemm_f32x4{0.f, 0.f, 0.f, 0.f};

  // roles of a round: item = (hit h, point p), its four taps; G piece g = (hit g / 6, 16-byte part g % 6), 384 per round
  const int h = tid >> 2, p = tid & 3, h5 = h & 31;
  const unsigned w_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)S.W;
  const unsigned wcol = w_lds + (unsigned)(((h >> 5) * 4 + ((h5 >> 2) & 3)) * PLANE + ((h5 & 3) + ((h5 >> 4) << 2)) * 4);
  const unsigned trash = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float *)(S.trash + tid);
  const unsigned char *go_nm = grad_out + (row_base * d.M + m) * kT3RowB;
  const size_t q_stride = (size_t)d.M * kT3RowB;
  const float *loc_nm = loc + ((row_base * d.M + m) * LP + l * kPatchP + p) * 2;
  const float *attn_nm = attn + (row_base * d.M + m) * LP + l * kPatchP + p;
  const size_t s_stride = (size_t)d.M * LP;
  const int r16 = lane & 15, g4 = lane >> 4;
  const unsigned char *gfrag = S.G + (4 * g4 + ((lane >> 2) & 3)) * kT3RowB + 8 * (lane & 3);

  for (int pass0 = 0; pass0 < total_hits; pass0 += kTile2HitList) {
    const int pass1 = min(pass0 + kTile2HitList, total_hits), np = pass1 - pass0;
    // ---- expand the marks into the hit list (order: candidate, then bit -- fixed; see msda_bwd_d48_tile2_kernel) ----
    {
      unsigned long long todo = __ballot(my_cnt && my_excl < pass1 && my_excl + my_cnt > pass0);
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

    // this thread's item and G pieces of the first round (later rounds: in flight behind the previous round)
    unsigned n_hit = 0u;
    float2 n_xy = make_float2(-4.f, -4.f);
    float n_a = 0.f;
    u32x4 n_g[2];
    auto fetch = [&](int lo_, int nh_) {
      n_hit = 0u; n_xy = make_float2(-4.f, -4.f); n_a = 0.f;       // (a location outside every map decodes to "not near")
      if (h < nh_) {
        n_hit = S.hits[lo_ + h];
        const size_t so = (size_t)hit_query(lv, n_hit) * s_stride;
        n_xy = *reinterpret_cast<const float2 *>(loc_nm + 2 * so);
        n_a = attn_nm[so];
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int g = tid + i * kPatchThreads, gh = g / 6, gpart = g - gh * 6;
        n_g[i] = u32x4{0u, 0u, 0u, 0u};
        if (gh < nh_)       // (gh < 64 follows: nh_ <= 64)
          n_g[i] = *reinterpret_cast<const u32x4 *>(go_nm + (size_t)hit_query(lv, S.hits[lo_ + gh]) * q_stride + gpart * 16);
      }
    };
    fetch(0, min(kT3Hits, np));

    for (int lo = 0; lo < np; lo += kT3Hits) {
      const int nh = min(kT3Hits, np - lo);
      // ---- decode this thread's item: the four taps of sample (h, p); a tap this tile does not own adds 0 to `trash` ----
      unsigned ta[4] = {trash, trash, trash, trash};
      float tw[4] = {0.f, 0.f, 0.f, 0.f};
      {
        const float x = px_coord(n_xy.x, me.W), y = px_coord(n_xy.y, me.H);
        const bool inside = (y > -1.f) && (x > -1.f) && (y < (float)me.H) && (x < (float)me.W);
        if (inside) {
          int lq, qy, qx;
          hit_query(lv, n_hit, lq, qy, qx);
          float rwq, rhq;
          hit_ratios(lv, lq, rwq, rhq);
          if (near_anchor(x, y, anchor_from_ratio(qx, rwq), anchor_from_ratio(qy, rhq), plan.radius)) {
            const float yf = floorf(y), xf = floorf(x);
            const int y0 = (int)yf, x0 = (int)xf;
            const float lh = y - yf, lw = x - xf, hh = 1.f - lh, hw = 1.f - lw;
            const float w4[4] = {hh * hw * n_a, hh * lw * n_a, lh * hw * n_a, lh * lw * n_a};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const int ty = y0 + (k >> 1), tx = x0 + (k & 1);
              const bool mine = ty >= 0 && ty <= me.H - 1 && tx >= 0 && tx <= me.W - 1 && (ty >> me.shift) == tyi &&
                                (tx >> me.shift) == txi;
              if (mine) {
                ta[k] = wcol + (unsigned)(((ty - ty0) << me.shift) + (tx - tx0)) * 32u;
                tw[k] = w4[k];
              }
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
      //      four different pixels and different hits are different columns: no two lanes of an instruction meet ----
      //      (the phase numbers are made opaque: hipcc otherwise sees four mutually exclusive branches on p and is free to
      //      run them in ANY order -- it built a decision tree that ran point 3 first and point 0's plain store third)
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
      lds_barrier();      // Wt and G complete
      // ---- tile += Wt . G on the matrix pipe ----
      if (NPB >= 4 || wave == 0) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          if (s == 1 && nh <= 32) break;
          gemm_bf16x8 bf[3];
#pragma unroll
          for (int cb = 0; cb < 3; ++cb) bf[cb] = tile3_bfrag(gfrag + s * 32 * kT3RowB + 32 * cb);
#pragma unroll
          for (int i = 0; i < NACC; ++i) {
            const int pb = NPB >= 4 ? 4 * i + wave : 0;
            const unsigned char *wp = S.W + (s * 4 + g4) * PLANE + (16 * pb + r16) * 32;
            const f32x4 r0 = *reinterpret_cast<const f32x4 *>(wp), r1 = *reinterpret_cast<const f32x4 *>(wp + 16);
            gemm_bf16x8 ahi, alo;
            tile3_split(r0, r1, ahi, alo);
#pragma unroll
            for (int cb = 0; cb < 3; ++cb) acc[i][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ahi, bf[cb], acc[i][cb], 0, 0, 0);
#pragma unroll
            for (int cb = 0; cb < 3; ++cb) acc[i][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(alo, bf[cb], acc[i][cb], 0, 0, 0);
          }
        }
      }
      lds_barrier();      // fragment reads done: columns may be cleared, G overwritten
#pragma unroll
      for (int k = 0; k < 4; ++k) t3_lds_write(ta[k], 0.f);
    }
    lds_barrier();        // before the next pass overwrites the hit list
  }

  // ---- add the tile to grad_value: plain read-modify-write (the query-side kernel has finished; tiles are disjoint).
  //      Accumulator layout: lane (column c = lane & 15, k-group g) holds channel 16 cb + c of pixels 16 pb + 4 g + reg. ----
  if (NPB >= 4 || wave == 0) {
    const size_t img_base = ((size_t)n * d.S + me.start) * d.M;
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      const int pb = NPB >= 4 ? 4 * i + wave : 0;
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int pix = 16 * pb + 4 * g4 + reg;
        const int ty = ty0 + (pix >> me.shift), tx = tx0 + (pix & (edge - 1));
        if (pix < tpx && ty < me.H && tx < me.W) {
          float *dst = grad_value + (img_base + (size_t)(ty * me.W + tx) * d.M + m) * kD48 + r16;
#pragma unroll
          for (int cb = 0; cb < 3; ++cb) dst[16 * cb] += acc[i][cb][reg];
        }
      }
    }
  }
}

__global__ __launch_bounds__(kPatchThreads) __attribute__((amdgpu_waves_per_eu(2, 2))) void msda_bwd_d48_tile3_kernel(
    const void *__restrict__ grad_out, const float *__restrict__ loc, const float *__restrict__ attn, CoreDims d,
    PatchPlan plan, float *__restrict__ grad_value) {
  __shared__ Tile3Lds S;
  // XCD-major walk of the (n, m) pairs, as msda_bwd_d48_tile2_kernel; within an (n, m) the tiles of the COARSE levels go
  // first: they are reached by the most queries, i.e. they are the longest-running workgroups
  const int tiles = plan.total_tiles;
  const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
  const int nm = xcd * ((d.N * d.M + 7) >> 3) + jb / tiles;
  if (nm >= d.N * d.M) return;
  const int tile_id = tiles - 1 - jb % tiles;
  const int m = nm % d.M, n = nm / d.M;
  int l = 0;
  for (int i = 1; i < plan.L; ++i) l = (tile_id >= plan.lv[i].tile_base) ? i : l;
  const int t = tile_id - plan.lv[l].tile_base;
  const unsigned char *go = reinterpret_cast<const unsigned char *>(grad_out);
  const int shift = plan.lv[l].shift;
  if (shift >= 4) tile3_body<16>(S, go, loc, attn, d, plan, grad_value, n, m, l, t);
  else if (shift == 3) tile3_body<4>(S, go, loc, attn, d, plan, grad_value, n, m, l, t);
  else tile3_body<1>(S, go, loc, attn, d, plan, grad_value, n, m, l, t);
}

}  // namespace snipper
