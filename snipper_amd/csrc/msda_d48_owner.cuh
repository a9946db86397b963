// msda_d48_owner.cuh -- "owner-computes" grad_value for the encoder shape (D = 48, f32, P = 4, Lq == S).
//
// Why: the straightforward backward scatters every tap with a float atomic to HBM
// (the reference does exactly that, /root/reference/models/ops/src/cuda/ms_deform_im2col_cuda.cuh:125-152).
// On MI355X float atomics execute at the memory side at ~1.3 TB/s chip-wide whatever the locality
// (MI355X_MICROARCH.md "Global float atomics"); an encoder launch adds N*S*M*L*P*4*D*4 B = 728 MB per
// sample, i.e. ~0.5 ms against an 8.5 us HBM roofline for its 68 MB of algorithmic traffic.  Measured
// round 1: 503 us (N=1), 4.3 ms (N=8) = 1.6 % of the roofline, the largest kernel of the training step.
//
// Idea: in the encoder the queries ARE the pixels of the L feature maps and a query samples each level
// near its own position.  grad_value is cut into tiles (one level, one (batch, head), th x tw pixels);
// every tile has ONE owner workgroup that sums the taps landing in it and adds the tile to HBM with
// plain stores.  Two kernels:
//
//   1. msda_bwd_d48_bin_kernel    query-stationary (one 16-lane group per (n,q,m) row, as the atomic
//      kernel): computes grad_loc / grad_attn for every sample, and for the "near" samples of a
//      (row, level) marks the query in the byte map of every tile their taps touch (a plain byte
//      store at a slot fixed by (tile, query): no atomics, no capacity -- returning global atomics for
//      list slots were measured at 1.0 ms per launch).  A tap whose tile got the mark is OWNED; every
//      other tap (far sample, footprint spread over > 16 tiles) is added right here with the usual
//      HBM float atomic.
//   2. msda_bwd_d48_tile_kernel   tile-stationary: reads its byte map, walks the marked queries in chunks
//      of 64, stages their grad_out rows in LDS, re-decodes their sampling points, counting-sorts the
//      owned taps by pixel in LDS and lets each 16-lane group accumulate ITS pixels in registers
//      (3 channels per lane), so LDS is only read, never read-modify-written (measured: ds_add_f32
//      ~127 cycles per wave-instruction; plain LDS RMW is bound by the 64 B/clk/CU LDS write path).
//
// Exactness does not depend on locality.  "near" = inside the map and |pixel - anchor| <= R on both
// axes, anchor being a fixed function of the query INDEX (its pixel centre rescaled to the sampled
// level); both kernels evaluate it with the pinned arithmetic of msda_d48.cuh and derive a tap's tile
// with the same integer shifts, and kernel 2 visits exactly the queries kernel 1 marked.  So every tap is added exactly once for ANY input; locality only decides how many
// taps go the fast way.
#pragma once
#include "msda_d48.cuh"

namespace snipper {

constexpr int kOwnerMaxLevels = 4;          // L*P <= 16 so that one decode pass covers a row
constexpr int kOwnerP = 4;                  // the 4 points of a level form a DPP quad
constexpr int kOwnerBlock = 256;
constexpr int kOwnerScanBatch = 1024;       // byte-map cells examined per scan round (4 per thread)
constexpr int kOwnerMaxTilePx = 256;        // 16 x 16
constexpr int kOwnerMaxBBox = 16;           // tiles a (row, level) footprint may span and still be binned

struct OwnerLevel {
  int H, W, start;        // level geometry
  int shift;              // tile edge = 1 << shift (square tiles)
  int ntx, nty;           // tiles per row / column
  int tile_base;          // index of this level's first tile
};
// The "lists" are a dense byte map: for tile T of sampled level l and a query level lq, the queries whose
// near taps can reach T form a rectangle of the lq grid (anchor_range below, a pure function of T, l, lq
// and R).  Query (qy, qx) of that rectangle has the fixed slot
//   tile_bytes(T) + coff[l][lq] + (qy - qy0) * rw[l][lq] + (qx - qx0)
// with rw/rh host-side upper bounds of the rectangle's size.  The bin kernel stores a 1 there (a plain
// byte store: idempotent, no atomics, no capacity), the tile kernel reads its bytes back.
struct OwnerPlan {
  OwnerLevel lv[kOwnerMaxLevels];
  int rw[kOwnerMaxLevels][kOwnerMaxLevels];     // [l][lq] bound on the candidate rectangle's width
  int rh[kOwnerMaxLevels][kOwnerMaxLevels];
  int coff[kOwnerMaxLevels][kOwnerMaxLevels];   // [l][lq] byte offset of lq's rectangle inside a tile's bytes
  int tstride[kOwnerMaxLevels];                 // bytes per tile of level l
  long long lvl_base[kOwnerMaxLevels];          // byte offset of level l's first tile inside one (n, m) block
  long long bytes_per_nm;
  int L;
  int total_tiles;        // over all levels, per (n, m)
  float radius;
  int debug;              // timing ablations only: 1 = tile kernel skips accumulation, 2 = no binning
  unsigned char *bitmap;  // [N*M][bytes_per_nm]   (workspace, zeroed per call)
};

// conservative index range of the queries of a level with `n_lq` cells along an axis whose anchor,
// expressed in the sampled level (n_l cells), can fall in [lo_px, hi_px]
__device__ __forceinline__ void anchor_range(float lo_px, float hi_px, int n_l, int n_lq, int &i0, int &i1) {
  const float inv = __fdiv_rn((float)n_lq, (float)n_l);
  i0 = (int)floorf(__fmaf_rn(lo_px + 0.5f, inv, -0.5f)) - 1;
  i1 = (int)ceilf(__fmaf_rn(hi_px + 0.5f, inv, -0.5f)) + 1;
  i0 = i0 < 0 ? 0 : i0;
  i1 = i1 > n_lq - 1 ? n_lq - 1 : i1;
}
// candidate rectangle of query level lq for the tile with origin (ty0, tx0), edge `edge`, of level l
__device__ __forceinline__ void tile_candidates(const OwnerPlan &p, int l, int lq, int ty0, int tx0, int edge,
                                                int &qx0, int &qx1, int &qy0, int &qy1) {
  anchor_range((float)(tx0 - 1) - p.radius, (float)(tx0 + edge) + p.radius, p.lv[l].W, p.lv[lq].W, qx0, qx1);
  anchor_range((float)(ty0 - 1) - p.radius, (float)(ty0 + edge) + p.radius, p.lv[l].H, p.lv[lq].H, qy0, qy1);
}

// ------------------------------------------------------------------------------------------------
// Kernel 1: grad_loc / grad_attn for all samples + binning of near samples + far atomics.
// Lane mapping as the forward: lane i of a 16-lane group owns channels 3i..3i+2 (one dwordx3 per tap).
// ------------------------------------------------------------------------------------------------
struct BinRecord {   // 48 B per (row, sample)
  f32x4 q0;          // lh, lw, a, a*W
  u32x4 off;         // byte offsets of the taps in value / grad_value
  f32x4 q2;          // a*H, bits: mask of taps that still need an HBM atomic, -, -
};

__device__ __forceinline__ int quad_min(int v) {
  v = min(v, __shfl_xor(v, 1, 4));
  return min(v, __shfl_xor(v, 2, 4));
}
__device__ __forceinline__ int quad_max(int v) {
  v = max(v, __shfl_xor(v, 1, 4));
  return max(v, __shfl_xor(v, 2, 4));
}

__global__ __launch_bounds__(kD48Block) void msda_bwd_d48_bin_kernel(
    const float *__restrict__ grad_out, const float *__restrict__ value,
    const float *__restrict__ loc, const float *__restrict__ attn, CoreDims d, OwnerPlan plan,
    float *__restrict__ grad_value, float *__restrict__ grad_loc, float *__restrict__ grad_attn,
    int nblk_padded, int go_bf16) {
  constexpr int G = 16, kRows = kD48Block / G;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int LP = d.L * kOwnerP;     // runtime on purpose: see the unroll note below
  const int rec_stride = LP * (int)sizeof(BinRecord) + 16;
  const int grp = threadIdx.x / G, lane = threadIdx.x % G;
  const long long total_rows = (long long)d.N * d.Lq * d.M;
  const long long row = (long long)xcd_band_block(nblk_padded) * kRows + grp;
  const bool live = row < total_rows;

  unsigned char *my_recs = smem_raw + (size_t)grp * rec_stride;
  {
    // Every lane runs the decode (the quad shuffles need all four lanes of a level); lanes beyond L*P
    // and dead rows carry an empty footprint.
    const long long rrow = live ? row : 0;
    const int m = (int)(rrow % d.M);
    const int q = (int)((rrow / d.M) % d.Lq);
    const long long n = rrow / ((long long)d.M * d.Lq);
    const bool lane_live = live && lane < LP;
    const int s = lane_live ? lane : 0;
    const int l = s / kOwnerP;
    const OwnerLevel lvl = plan.lv[l];
    const int H = lvl.H, W = lvl.W;
    int lq = 0;
    for (int i = 1; i < plan.L; ++i) lq = (q >= plan.lv[i].start) ? i : lq;
    const int rq = q - plan.lv[lq].start;
    const int qy = rq / plan.lv[lq].W, qx = rq - qy * plan.lv[lq].W;

    const unsigned px_stride = (unsigned)d.M * kD48 * 4u;
    const unsigned base = (unsigned)(n * d.S) * px_stride + (unsigned)m * (kD48 * 4u);
    const long long li = rrow * LP + s;
    const float lx = loc[2 * li], ly = loc[2 * li + 1], a = attn[li];
    const float y = px_coord(ly, H), x = px_coord(lx, W);
    const bool inside = lane_live && (y > -1.f) && (x > -1.f) && (y < (float)H) && (x < (float)W);
    const bool cand = inside && near_anchor(x, y, anchor_coord(qx, W, plan.lv[lq].W),
                                            anchor_coord(qy, H, plan.lv[lq].H), plan.radius);
    const float yf = floorf(y), xf = floorf(x);
    const int y0 = (int)yf, x0 = (int)xf;
    const bool yok0 = inside && y0 >= 0, yok1 = inside && y0 + 1 <= H - 1;
    const bool xok0 = x0 >= 0, xok1 = x0 + 1 <= W - 1;
    // tile-coordinate bounding box of this sample's in-map taps (empty when not a candidate)
    const int sh = lvl.shift;
    const int bx0 = cand ? (max(x0, 0) >> sh) : 1 << 20, bx1 = cand ? (min(x0 + 1, W - 1) >> sh) : -1;
    const int by0 = cand ? (max(y0, 0) >> sh) : 1 << 20, by1 = cand ? (min(y0 + 1, H - 1) >> sh) : -1;
    const int minx = quad_min(bx0), maxx = quad_max(bx1), miny = quad_min(by0), maxy = quad_max(by1);
    const int nbx = maxx - minx + 1, nby = maxy - miny + 1;
    // quad leader: mark the query in the byte map of every tile of the box
    unsigned okmask = 0;
    if ((lane & 3) == 0 && nbx > 0 && nby > 0 && nbx * nby <= kOwnerMaxBBox) {
      unsigned char *bm = plan.bitmap + (n * d.M + m) * plan.bytes_per_nm + plan.lvl_base[l];
      const int edge = 1 << sh;
      for (int i = 0; i < nbx * nby; ++i) {
        const int ty = miny + i / nbx, tx = minx + i % nbx;
        int cx0, cx1, cy0, cy1;
        tile_candidates(plan, l, lq, ty << sh, tx << sh, edge, cx0, cx1, cy0, cy1);
        const int dx = qx - cx0, dy = qy - cy0;
        // always true for a near sample (the rectangle is conservative); checked so that a miss can only
        // cost speed (the taps then stay on the atomic path), never correctness
        if (dx >= 0 && dy >= 0 && qx <= cx1 && qy <= cy1 && dx < plan.rw[l][lq] && dy < plan.rh[l][lq]) {
          if (plan.debug != 2)
            bm[(long long)(ty * lvl.ntx + tx) * plan.tstride[l] + plan.coff[l][lq] + dy * plan.rw[l][lq] + dx] = 1;
          okmask |= 1u << i;
        }
      }
    }
    okmask = __shfl(okmask, lane & ~3, 16);
    // taps that are NOT owned by a tile keep their HBM atomic
    unsigned need = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int ty = y0 + (k >> 1), tx = x0 + (k & 1);
      const bool in_map = ((k >> 1) ? yok1 : yok0) && ((k & 1) ? xok1 : xok0);
      const int bi = ((ty >> sh) - miny) * nbx + ((tx >> sh) - minx);
      const bool owned = cand && in_map && nbx * nby <= kOwnerMaxBBox && ((okmask >> (bi & 31)) & 1u);
      need |= (in_map && !owned) ? (1u << k) : 0u;
    }
    if (lane_live) {
      const unsigned p00 = base + (unsigned)(lvl.start + y0 * W + x0) * px_stride;
      const float ai = inside ? a : 0.f;
      BinRecord r;
      r.q0.x = inside ? y - yf : 0.f; r.q0.y = inside ? x - xf : 0.f; r.q0.z = ai; r.q0.w = ai * (float)W;
      r.q2.x = ai * (float)H; r.q2.y = __uint_as_float(need); r.q2.z = 0.f; r.q2.w = 0.f;
      r.off.x = (yok0 && xok0) ? p00 : kOobOffset;
      r.off.y = (yok0 && xok1) ? p00 + px_stride : kOobOffset;
      r.off.z = (yok1 && xok0) ? p00 + (unsigned)W * px_stride : kOobOffset;
      r.off.w = (yok1 && xok1) ? p00 + (unsigned)(W + 1) * px_stride : kOobOffset;
      *reinterpret_cast<BinRecord *>(my_recs + s * sizeof(BinRecord)) = r;
    }
  }
  __syncthreads();
  if (!live) return;

  const unsigned value_bytes = (unsigned)((size_t)d.N * d.S * d.M * kD48 * 4u);
  const auto vsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(value), 0, (int)value_bytes, 0x00020000);
  const auto gsrc = __builtin_amdgcn_make_buffer_rsrc(grad_value, 0, (int)value_bytes, 0x00020000);
  const unsigned lane_off = (unsigned)lane * 12u;
  const size_t gi = (size_t)row * kD48 + lane * 3;
  const float g0 = ld_go(grad_out, gi, go_bf16), g1 = ld_go(grad_out, gi + 1, go_bf16), g2 = ld_go(grad_out, gi + 2, go_bf16);

  float keep_a = 0.f, keep_x = 0.f, keep_y = 0.f;
  // Pass 1 (branch-free, so the loads of several samples overlap): grad_attn / grad_loc of every sample.
  unsigned any_need = 0u;
#pragma unroll 4      // (a compile-time trip count makes the compiler unroll all 12 samples: 139 VGPRs, 3 waves/SIMD,
                      //  measured 20 % slower than this 63-VGPR form)
  for (int s = 0; s < LP; ++s) {
    const BinRecord r = *reinterpret_cast<const BinRecord *>(my_recs + s * sizeof(BinRecord));
    const float lh = r.q0.x, lw = r.q0.y;
    const float hh = 1.f - lh, hw = 1.f - lw;
    const float w[4] = {hh * hw, hh * lw, lh * hw, lh * lw};
    const unsigned off[4] = {r.off.x, r.off.y, r.off.z, r.off.w};
    any_need |= __float_as_uint(r.q2.y);
    float dot[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const u32x3 v = __builtin_amdgcn_raw_buffer_load_b96(vsrc, off[k] + lane_off, 0, 0);
      dot[k] = g0 * __uint_as_float(v.x) + g1 * __uint_as_float(v.y) + g2 * __uint_as_float(v.z);
    }
    float pa = w[0] * dot[0] + w[1] * dot[1] + w[2] * dot[2] + w[3] * dot[3];
    float px = hh * (dot[1] - dot[0]) + lh * (dot[3] - dot[2]);
    float py = hw * (dot[2] - dot[0]) + lw * (dot[3] - dot[1]);
    pa = row16_sum(pa);
    px = row16_sum(px) * r.q0.w;
    py = row16_sum(py) * r.q2.x;
    if (lane == s) { keep_a = pa; keep_x = px; keep_y = py; }
  }
  // Pass 2 (rare once the tiles own the near taps): the taps no tile took keep their HBM atomic.
  if (__builtin_amdgcn_ballot_w64(any_need != 0u) != 0ull) {
    // Re-deal the row so that lane i holds channels {i, i+16, i+32}: each atomic wave-instruction then adds 64
    // contiguous bytes per row (the shape the memory-side atomic units want), as in the atomic-only kernel.
    // Channel c lives in lane c/3, element c%3.
    float ga[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int c = lane + 16 * j, src = c / 3, e = c - 3 * src;
      const float s0 = __shfl(g0, src, 16), s1 = __shfl(g1, src, 16), s2 = __shfl(g2, src, 16);
      ga[j] = e == 0 ? s0 : (e == 1 ? s1 : s2);
    }
    for (int s = 0; s < LP; ++s) {
      const BinRecord r = *reinterpret_cast<const BinRecord *>(my_recs + s * sizeof(BinRecord));
      const unsigned need = __float_as_uint(r.q2.y);
      if (__builtin_amdgcn_ballot_w64(need != 0u) == 0ull) continue;
      const float lh = r.q0.x, lw = r.q0.y, a = r.q0.z;
      const float hh = 1.f - lh, hw = 1.f - lw;
      const float w[4] = {hh * hw, hh * lw, lh * hw, lh * lw};
      const unsigned off[4] = {r.off.x, r.off.y, r.off.z, r.off.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const unsigned go = ((need >> k) & 1u) ? off[k] + (unsigned)lane * 4u : kOobOffset;
        const float wa = w[k] * a;
        __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(wa * ga[0], gsrc, go, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(wa * ga[1], gsrc, go + 64u, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(wa * ga[2], gsrc, go + 128u, 0, 0);
      }
    }
  }
  if (lane < LP) {
    const long long li = row * LP + lane;
    grad_attn[li] = keep_a;
    *reinterpret_cast<float2 *>(grad_loc + 2 * li) = make_float2(keep_x, keep_y);
  }
}

// ------------------------------------------------------------------------------------------------
// Kernel 2: one workgroup per (n, m, tile): destination-sorted accumulation in registers.
// ------------------------------------------------------------------------------------------------
template <int kOwnerChunk> struct TileLds {   // kOwnerChunk = list entries (queries) per round
  float gbuf[kOwnerChunk * kD48];                    // grad_out rows of the chunk's queries   12 KiB
  int2 tap[kOwnerChunk * kOwnerP * 4];               // sorted taps: (chunk entry, weight bits)     8 KiB
  int cnt[kOwnerMaxTilePx];                          // taps per pixel in this chunk
  int off[kOwnerMaxTilePx];                          // exclusive prefix of cnt
  int ent[kOwnerChunk];                              // the chunk's query indices
  int wsum[kOwnerBlock / 64];
  int hits[kOwnerScanBatch];                         // queries marked in this tile's byte map      4 KiB
  int n_hits;
};

// 5 waves per SIMD (96 VGPRs, 4 dwords spilled outside the chunk loop): the kernel is a chain of dependent LDS reads and
// barriers, so residency is what hides them -- 123 VGPRs / 4 waves measured 617 us per launch in the step (rocprofv3), this 575.
template <int kOwnerChunk>
__global__ __launch_bounds__(kOwnerBlock) __attribute__((amdgpu_waves_per_eu(kOwnerChunk <= 64 ? 5 : 1, kOwnerChunk <= 64 ? 5 : 8))) void msda_bwd_d48_tile_kernel(
    const float *__restrict__ grad_out, const float *__restrict__ loc, const float *__restrict__ attn,
    CoreDims d, OwnerPlan plan, float *__restrict__ grad_value, int go_bf16) {
  __shared__ TileLds<kOwnerChunk> S;
  // ---- which tile am I? ----
  // XCD-major: workgroup ids go round-robin over the 8 XCDs; all tiles of one (n, m) are given to ONE XCD back to back,
  // so the grad_out rows / locations of that (n, m) (~2.4 MB), which every tile a query's samples touch stages again,
  // are fetched from HBM once and then hit that XCD's L2 (PMC before: 676 MB fetched per launch for 150 MB of inputs).
  const int tiles = plan.total_tiles;
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int nm = xcd + 8 * (j / tiles);
  if (nm >= d.N * d.M) return;                      // padding workgroups of the XCD-major grid
  const int tile_id = j % tiles;
  const int m = nm % d.M;
  const int n = nm / d.M;
  int l = 0;
  for (int i = 1; i < plan.L; ++i) l = (tile_id >= plan.lv[i].tile_base) ? i : l;   // bases ascend with the level
  const OwnerLevel me = plan.lv[l];
  const int t = tile_id - me.tile_base;
  const int edge = 1 << me.shift, tpx = edge * edge;
  const int ty0 = (t / me.ntx) << me.shift, tx0 = (t % me.ntx) << me.shift;
  const unsigned char *bm = plan.bitmap + ((long long)n * d.M + m) * plan.bytes_per_nm + plan.lvl_base[l] +
                            (long long)t * plan.tstride[l];

  const int tid = threadIdx.x, grp = tid >> 4, lane = tid & 15, wave = tid >> 6;
  const int LP = d.L * kOwnerP;
  const size_t row_base = (size_t)n * d.Lq;
  constexpr int kMaxPxPerGroup = kOwnerMaxTilePx / 16;
  float acc[kMaxPxPerGroup][3];
#pragma unroll
  for (int u = 0; u < kMaxPxPerGroup; ++u) acc[u][0] = acc[u][1] = acc[u][2] = 0.f;
  const int px_per_group = (tpx + 15) >> 4;

  if (tid == 0) S.n_hits = 0;
  __syncthreads();
  for (int lq = 0; lq < plan.L; ++lq) {
   int cx0, cx1, cy0, cy1;
   tile_candidates(plan, l, lq, ty0, tx0, edge, cx0, cx1, cy0, cy1);
   const int cw = min(cx1 - cx0 + 1, plan.rw[l][lq]), ch = min(cy1 - cy0 + 1, plan.rh[l][lq]);
   if (cw <= 0 || ch <= 0) continue;
   const unsigned char *bml = bm + plan.coff[l][lq];
   const int qbase = plan.lv[lq].start + cy0 * plan.lv[lq].W + cx0;
   for (int sb = 0; sb < cw * ch; sb += kOwnerScanBatch) {
    // ---- scan this tile's byte map: which queries left a mark? ----
#pragma unroll
    for (int u = 0; u < kOwnerScanBatch / kOwnerBlock; ++u) {
      const int c = sb + u * kOwnerBlock + tid;
      if (c < cw * ch) {
        const int cy = c / cw, cx = c - cy * cw;
        if (bml[cy * plan.rw[l][lq] + cx]) S.hits[atomicAdd(&S.n_hits, 1)] = qbase + cy * plan.lv[lq].W + cx;
      }
    }
    __syncthreads();
    const int total = plan.debug == 5 ? 0 : S.n_hits;
  for (int base = 0; base < total; base += kOwnerChunk) {
    const int nh = min(kOwnerChunk, total - base);
    if (tid < nh) S.ent[tid] = S.hits[base + tid];
    for (int i = tid; i < tpx; i += kOwnerBlock) S.cnt[i] = 0;
    __syncthreads();
    // ---- stage the grad_out rows of the chunk (16 B per lane, coalesced) ----
    if (go_bf16) {                                      // bf16 rows: 16 B = 8 channels per lane, widened into LDS
      for (int i = tid; i < (plan.debug == 6 ? 0 : nh * (kD48 / 8)); i += kOwnerBlock) {
        const int h = i / (kD48 / 8), c8 = i % (kD48 / 8);
        const u32x4 v = *reinterpret_cast<const u32x4 *>(reinterpret_cast<const uint16_t *>(grad_out) +
                                                         ((row_base + S.ent[h]) * d.M + m) * kD48 + c8 * 8);
        f32x4 lo, hi;
        lo.x = __uint_as_float(v.x << 16); lo.y = __uint_as_float(v.x & 0xffff0000u);
        lo.z = __uint_as_float(v.y << 16); lo.w = __uint_as_float(v.y & 0xffff0000u);
        hi.x = __uint_as_float(v.z << 16); hi.y = __uint_as_float(v.z & 0xffff0000u);
        hi.z = __uint_as_float(v.w << 16); hi.w = __uint_as_float(v.w & 0xffff0000u);
        *reinterpret_cast<f32x4 *>(S.gbuf + h * kD48 + c8 * 8) = lo;
        *reinterpret_cast<f32x4 *>(S.gbuf + h * kD48 + c8 * 8 + 4) = hi;
      }
    } else
    for (int i = tid; i < (plan.debug == 6 ? 0 : nh * (kD48 / 4)); i += kOwnerBlock) {
      const int h = i / (kD48 / 4), c4 = i % (kD48 / 4);
      const f32x4 v = *reinterpret_cast<const f32x4 *>(grad_out + ((row_base + S.ent[h]) * d.M + m) * kD48 + c4 * 4);
      *reinterpret_cast<f32x4 *>(S.gbuf + h * kD48 + c4 * 4) = v;
    }
    // ---- decode: thread = (entry h, point p); rank every owned tap within its pixel ----
    constexpr int kItems = kOwnerChunk * kOwnerP / kOwnerBlock;     // (entry, point) pairs per thread
    int t_pix[kItems][4], t_rank[kItems][4], t_h[kItems];
    float t_w[kItems][4];
#pragma unroll
    for (int it = 0; it < kItems; ++it) {
    const int h = (tid + it * kOwnerBlock) >> 2, p = tid & 3;
    t_h[it] = h;
#pragma unroll
    for (int k = 0; k < 4; ++k) { t_pix[it][k] = -1; t_rank[it][k] = 0; t_w[it][k] = 0.f; }
    if (h < nh && plan.debug != 7) {
      const int q = S.ent[h];
      int lq = 0;
      for (int i = 1; i < plan.L; ++i) lq = (q >= plan.lv[i].start) ? i : lq;
      const int rq = q - plan.lv[lq].start;
      const int qy = rq / plan.lv[lq].W, qx = rq - qy * plan.lv[lq].W;
      const size_t li = ((row_base + q) * d.M + m) * LP + l * kOwnerP + p;
      const float2 xy = *reinterpret_cast<const float2 *>(loc + 2 * li);
      const float x = px_coord(xy.x, me.W), y = px_coord(xy.y, me.H);
      const bool inside = (y > -1.f) && (x > -1.f) && (y < (float)me.H) && (x < (float)me.W);
      const bool cand = inside && near_anchor(x, y, anchor_coord(qx, me.W, plan.lv[lq].W),
                                              anchor_coord(qy, me.H, plan.lv[lq].H), plan.radius);
      if (cand) {
        const float a = attn[li];
        const float yf = floorf(y), xf = floorf(x);
        const int y0 = (int)yf, x0 = (int)xf;
        const float lh = y - yf, lw = x - xf, hh = 1.f - lh, hw = 1.f - lw;
        const float w4[4] = {hh * hw * a, hh * lw * a, lh * hw * a, lh * lw * a};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int ty = y0 + (k >> 1), tx = x0 + (k & 1);
          const bool in_map = ty >= 0 && ty <= me.H - 1 && tx >= 0 && tx <= me.W - 1;
          const bool mine = in_map && ((ty >> me.shift) << me.shift) == ty0 && ((tx >> me.shift) << me.shift) == tx0;
          if (mine) {
            t_pix[it][k] = ((ty - ty0) << me.shift) + (tx - tx0);
            t_w[it][k] = w4[k];
            t_rank[it][k] = plan.debug == 3 ? 0 : atomicAdd(&S.cnt[t_pix[it][k]], 1);
          }
        }
      }
    }
    }
    __syncthreads();
    // ---- exclusive prefix of cnt over the tile's pixels (one thread per pixel) ----
    {
      const int c = tid < tpx ? S.cnt[tid] : 0;
      int inc = c;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int up = __shfl_up(inc, o, 64);
        inc += ((tid & 63) >= o) ? up : 0;
      }
      if ((tid & 63) == 63) S.wsum[wave] = inc;
      __syncthreads();
      int before = 0;
      for (int w = 0; w < wave; ++w) before += S.wsum[w];
      if (tid < tpx) S.off[tid] = before + inc - c;
    }
    __syncthreads();
    // ---- scatter the taps into pixel order ----
#pragma unroll
    for (int it = 0; it < kItems; ++it) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (t_pix[it][k] >= 0) {
          const int pos = S.off[t_pix[it][k]] + t_rank[it][k];
          S.tap[pos] = make_int2(t_h[it], __float_as_int(t_w[it][k]));
        }
      }
    }
    __syncthreads();
    // ---- accumulate: group g owns pixels g, g+16, ...; lane i owns channels 3i..3i+2 ----
    // The phase is a chain of dependent LDS reads (list bounds -> tap record -> staged row -> FMA) with ~4 taps per
    // pixel and chunk, i.e. latency-bound: the tap loop handles two taps per trip (two row reads in flight, the next
    // pair of records prefetched).  A tap beyond the list end is clamped to the last valid one and gets weight 0.
    // (Fetching bounds and first taps of all 16 pixels up front was measured: 200 VGPRs, 2 waves/SIMD, 30 % slower.)
    if (plan.debug != 1) {
#pragma unroll
      for (int u = 0; u < kMaxPxPerGroup; ++u) {
        if (u < px_per_group) {
          const int pix = grp + 16 * u;
          const int b_u = pix < tpx ? S.off[pix] : 0, n_u = pix < tpx ? S.cnt[pix] : 0;
          const int last = n_u - 1;
          int2 a = S.tap[b_u], b = S.tap[b_u + (n_u > 1 ? 1 : 0)];
          for (int e = 0; e < n_u; e += 2) {
            const int2 na = S.tap[b_u + min(e + 2, last)], nb = S.tap[b_u + min(e + 3, last)];   // next pair
            const float wa = __int_as_float(a.y), wb = e + 1 < n_u ? __int_as_float(b.y) : 0.f;
            const float *ga = S.gbuf + a.x * kD48 + lane * 3, *gb = S.gbuf + b.x * kD48 + lane * 3;
            const float a0 = ga[0], a1 = ga[1], a2 = ga[2], b0 = gb[0], b1 = gb[1], b2 = gb[2];
            acc[u][0] = fmaf(wb, b0, fmaf(wa, a0, acc[u][0]));
            acc[u][1] = fmaf(wb, b1, fmaf(wa, a1, acc[u][1]));
            acc[u][2] = fmaf(wb, b2, fmaf(wa, a2, acc[u][2]));
            a = na; b = nb;
          }
        }
      }
    }
    __syncthreads();
  }
    if (tid == 0) S.n_hits = 0;
    __syncthreads();
   }
  }

  // ---- add the tile to grad_value: plain read-modify-write (kernel 1 has finished; tiles are disjoint) ----
  const size_t img_base = ((size_t)n * d.S + me.start) * d.M;
#pragma unroll
  for (int u = 0; u < kMaxPxPerGroup; ++u) {
    if (u < px_per_group) {
      const int pix = grp + 16 * u;
      const int ty = ty0 + (pix >> me.shift), tx = tx0 + (pix & (edge - 1));
      if (pix < tpx && ty < me.H && tx < me.W) {
        float *dst = grad_value + (img_base + (size_t)(ty * me.W + tx) * d.M + m) * kD48 + lane * 3;
        dst[0] += acc[u][0];
        dst[1] += acc[u][1];
        dst[2] += acc[u][2];
      }
    }
  }
}

}  // namespace snipper
