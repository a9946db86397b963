// msda_d48_owner.cuh -- "owner-computes" grad_value for the encoder shape (D = 48, f32, Lq == S).
//
// Why: the straightforward backward scatters every tap with a float atomic to HBM
// (the reference does exactly that, /root/reference/models/ops/src/cuda/ms_deform_im2col_cuda.cuh:125-152).
// On MI355X float atomics execute at the memory side at ~1.3 TB/s chip-wide whatever the locality
// (MI355X_MICROARCH.md "Global float atomics"); an encoder launch adds N*S*M*L*P*4*D*4 B = 728 MB per
// sample, i.e. ~0.5 ms against an 8.5 us HBM roofline for its 68 MB of algorithmic traffic.  Measured
// round 1: 503 us (N=1), 1.7 % of the roofline, the largest single kernel of the training step.
//
// Idea: in the encoder the queries ARE the pixels of the L feature maps, and a query samples each level
// near its own position (offsets of a few pixels).  So turn the scatter around: a workgroup OWNS one
// tile of one level of grad_value for one (batch, head), keeps it in LDS, walks the queries whose
// anchor position lies within the tile grown by a radius R, and accumulates the taps that land in its
// tile in LDS.  The tile is then written to HBM once with plain stores -- no HBM atomics and no
// pre-zeroing, since every pixel of grad_value has exactly one owner.
//
// No LDS atomics either (measured: ds_add_f32 costs ~127 cycles per wave-instruction in this access
// pattern, 4.7 of 6.6 ms at N=8).  Instead the block is three waves and wave w owns channels
// [16w, 16w+16) of the whole tile; a wave applies one sample per step with lane = (tap k, channel c):
// the four taps of a bilinear footprint are four different pixels, so the 64 lanes of a step touch 64
// different words and a plain LDS read-modify-write is race-free; steps of one wave execute in order,
// and waves never share a word.  The accumulation order is fixed, so the near part of grad_value is
// bitwise reproducible (the reference's atomics are not).
//
// Exactness does not depend on locality: a sample is "near" iff |pixel - anchor| <= R on both axes,
// where anchor is a fixed function of the query INDEX (its pixel centre rescaled to the sampled level).
// Near samples are accumulated here; far ones are added afterwards by the query-stationary kernel
// (msda_bwd_d48_f32_kernel<.., GRID=true>) with the usual HBM atomics.  Both kernels evaluate the same
// predicate with the same instructions (explicit fma/div intrinsics below), so every tap is added
// exactly once whatever the locations are; if the queries are not grid-like at all the result is still
// right, only slower.
#pragma once
#include "msda_d48.cuh"

namespace snipper {

constexpr int kOwnerMaxLevels = 8;
constexpr int kOwnerMaxPoints = 8;                            // P <= 8 on this path
constexpr int kOwnerBlock = 192;                              // 3 waves x 16 channels = D 48
constexpr int kOwnerScanPerThread = 4;                        // candidate queries a thread examines per batch
constexpr int kOwnerHitCap = kOwnerBlock * kOwnerScanPerThread;
constexpr int kOwnerHitChunk = kOwnerBlock;                   // hit queries decoded per round ...
constexpr int kOwnerMaxTilePixels = 256;                      // 256 px x 48 ch x 4 B = 48 KiB of LDS

struct OwnerLevel {
  int H, W, start;        // level geometry
  int th, tw;             // tile size in pixels
  int ntx, nty;           // tiles per row / column
  int tile_base;          // index of this level's first tile in the launch order
};
struct OwnerPlan {
  OwnerLevel lv[kOwnerMaxLevels];
  int L;
  int total_tiles;
  int max_tile_px;        // largest th*tw over the levels (sizes the LDS tile)
  float radius;
  int debug;              // 0 = normal; 1 = no tile updates; 2 = scan only (timing ablations, wrong results)
};
struct OwnerSample {       // one sampling point with at least one tap in the tile (32 B)
  f32x4 w;                 // bilinear weight x attention weight per tap
  int pix[4];              // pixel index inside the tile per tap, or -1
};
inline int owner_sample_cap(int P) { return kOwnerHitChunk * P; }   // ... so at most this many samples
inline size_t owner_lds_bytes(int max_tile_px, int P) {
  return (size_t)max_tile_px * kD48 * 4                         // the tile
         + (size_t)owner_sample_cap(P) * (sizeof(OwnerSample) + 4)   // samples + their query index
         + (size_t)kOwnerHitCap * 8                             // hit list: query index + packed (y, x)
         + 16;                                                  // counters
}

// conservative index range of the queries of a level with `n_lq` cells along an axis whose anchor,
// expressed in the sampled level (n_l cells), can fall in [lo_px, hi_px]
__device__ __forceinline__ void anchor_range(float lo_px, float hi_px, int n_l, int n_lq, int &i0, int &i1) {
  const float inv = (float)n_lq / (float)n_l;
  i0 = (int)floorf((lo_px + 0.5f) * inv - 0.5f) - 1;
  i1 = (int)ceilf((hi_px + 0.5f) * inv - 0.5f) + 1;
  i0 = i0 < 0 ? 0 : i0;
  i1 = i1 > n_lq - 1 ? n_lq - 1 : i1;
}

struct TileWindow { int ty0, tx0, th, tw, H, W; };

// Decode one sampling point for a tile: the four taps with their in-tile pixel index (or -1).
// Returns true when at least one tap belongs to the tile.  `near` uses the pinned arithmetic of
// msda_d48.cuh, so the far-only kernel drops exactly the taps accepted here.
__device__ __forceinline__ bool owner_decode(float lx, float ly, float a, float ax, float ay, float R,
                                             const TileWindow &t, OwnerSample &o) {
  const float x = px_coord(lx, t.W), y = px_coord(ly, t.H);
  const bool inside = (y > -1.f) && (x > -1.f) && (y < (float)t.H) && (x < (float)t.W);
  const bool take = inside && near_anchor(x, y, ax, ay, R);
  const float yf = floorf(y), xf = floorf(x);
  const int y0 = (int)yf, x0 = (int)xf;
  const float lh = y - yf, lw = x - xf, hh = 1.f - lh, hw = 1.f - lw;
  o.w.x = hh * hw * a; o.w.y = hh * lw * a; o.w.z = lh * hw * a; o.w.w = lh * lw * a;
  bool any = false;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int ty = y0 + (k >> 1), tx = x0 + (k & 1);
    const bool in_map = ty >= 0 && ty <= t.H - 1 && tx >= 0 && tx <= t.W - 1;
    const int py = ty - t.ty0, pxl = tx - t.tx0;
    const bool mine = take && in_map && py >= 0 && py < t.th && pxl >= 0 && pxl < t.tw;
    o.pix[k] = mine ? py * t.tw + pxl : -1;
    any |= mine;
  }
  return any;
}

__global__ __launch_bounds__(kOwnerBlock) void msda_bwd_d48_owner_kernel(
    const float *__restrict__ grad_out, const float *__restrict__ loc, const float *__restrict__ attn,
    CoreDims d, OwnerPlan plan, float *__restrict__ grad_value) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int P = d.P;
  const int samp_cap = kOwnerHitChunk * P;
  float *tile = reinterpret_cast<float *>(smem_raw);                                   // [th*tw][48]
  OwnerSample *samp = reinterpret_cast<OwnerSample *>(tile + plan.max_tile_px * kD48);  // [samp_cap]
  int *samp_q = reinterpret_cast<int *>(samp + samp_cap);                              // their query index
  int *hit_q = samp_q + samp_cap;                                                      // [kOwnerHitCap]
  int *hit_yx = hit_q + kOwnerHitCap;
  int *n_hit = hit_yx + kOwnerHitCap;
  int *n_samp = n_hit + 1;

  // ---- which tile am I?  launch order: coarse levels first (their tiles see more candidates) ----
  const int tiles = plan.total_tiles;
  int b = blockIdx.x;
  const int tile_id = b % tiles;  b /= tiles;
  const int m = b % d.M;
  const int n = b / d.M;
  int l = 0;   // the level whose [tile_base, tile_base + ntx*nty) holds tile_id (bases are not sorted by level)
  for (int i = 0; i < plan.L; ++i) {
    const int lo = plan.lv[i].tile_base;
    l = (tile_id >= lo && tile_id < lo + plan.lv[i].ntx * plan.lv[i].nty) ? i : l;
  }
  const OwnerLevel me = plan.lv[l];
  const int t = tile_id - me.tile_base;
  const TileWindow win{(t / me.ntx) * me.th, (t % me.ntx) * me.tw, me.th, me.tw, me.H, me.W};
  const int npx = me.th * me.tw;
  const float R = plan.radius;

  for (int i = threadIdx.x; i < npx * kD48; i += kOwnerBlock) tile[i] = 0.f;
  if (threadIdx.x == 0) { *n_hit = 0; *n_samp = 0; }
  __syncthreads();

  const int LP = d.L * P;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int tap_k = lane >> 4, chan = wave * 16 + (lane & 15);   // accumulate phase: lane = (tap, channel)
  const size_t row_base = (size_t)n * d.Lq;     // rows of this batch element: (row_base + q) * M + m

  for (int lq = 0; lq < plan.L; ++lq) {
    const int Hq = plan.lv[lq].H, Wq = plan.lv[lq].W, sq = plan.lv[lq].start;
    int qx0, qx1, qy0, qy1;
    anchor_range((float)(win.tx0 - 1) - R, (float)(win.tx0 + win.tw) + R, me.W, Wq, qx0, qx1);
    anchor_range((float)(win.ty0 - 1) - R, (float)(win.ty0 + win.th) + R, me.H, Hq, qy0, qy1);
    const int rw = qx1 - qx0 + 1, rh = qy1 - qy0 + 1;
    if (rw <= 0 || rh <= 0) continue;
    const int ncand = rw * rh;
    for (int base = 0; base < ncand; base += kOwnerHitCap) {
      // ---- A. scan: does candidate query c put any near tap into my tile?  All loads of the batch are
      //         issued before the first list append, so they overlap. ---------------------------------
      int cq[kOwnerScanPerThread], cyx[kOwnerScanPerThread];
      bool chit[kOwnerScanPerThread];
#pragma unroll
      for (int u = 0; u < kOwnerScanPerThread; ++u) {
        const int c = base + u * kOwnerBlock + (int)threadIdx.x;
        chit[u] = false;
        cq[u] = 0;
        cyx[u] = 0;
        if (c < ncand) {
          const int cy = c / rw;
          const int qy = qy0 + cy, qx = qx0 + (c - cy * rw);
          const int q = sq + qy * Wq + qx;
          const size_t li = ((row_base + q) * d.M + m) * LP + l * P;
          const float ax = anchor_coord(qx, me.W, Wq), ay = anchor_coord(qy, me.H, Hq);
          bool hit = false;
          for (int p = 0; p < P; ++p) {
            const float2 xy = *reinterpret_cast<const float2 *>(loc + 2 * (li + p));
            OwnerSample tmp;
            hit |= owner_decode(xy.x, xy.y, 1.f, ax, ay, R, win, tmp);
          }
          chit[u] = hit;
          cq[u] = q;
          cyx[u] = (qy << 16) | qx;
        }
      }
#pragma unroll
      for (int u = 0; u < kOwnerScanPerThread; ++u) {
        if (chit[u]) {
          const int slot = atomicAdd(n_hit, 1);
          hit_q[slot] = cq[u];
          hit_yx[slot] = cyx[u];
        }
      }
      __syncthreads();
      const int nh = plan.debug == 2 ? 0 : *n_hit;
      for (int hb = 0; hb < nh; hb += kOwnerHitChunk) {
        // ---- B. decode the (hit query, point) pairs of this chunk into the sample list -------------
        const int items = min(kOwnerHitChunk, nh - hb) * P;
        for (int it = threadIdx.x; it < items; it += kOwnerBlock) {
          const int e = hb + it / P, p = it % P;
          const int q = hit_q[e], yx = hit_yx[e];
          const size_t li = ((row_base + q) * d.M + m) * LP + l * P + p;
          const float2 xy = *reinterpret_cast<const float2 *>(loc + 2 * li);
          OwnerSample smp;
          if (owner_decode(xy.x, xy.y, attn[li], anchor_coord(yx & 0xffff, me.W, Wq),
                           anchor_coord(yx >> 16, me.H, Hq), R, win, smp)) {
            const int slot = atomicAdd(n_samp, 1);
            samp[slot] = smp;
            samp_q[slot] = q;
          }
        }
        __syncthreads();
        // ---- C. accumulate: every wave walks all samples for its 16 channels; lane = (tap, channel).
        //         Plain LDS read-modify-write: the 4 taps of a sample are 4 different pixels. ---------
        const int ns = *n_samp;
        if (ns > 0) {
          int pix = samp[0].pix[tap_k];
          float w = reinterpret_cast<const float *>(&samp[0].w)[tap_k];
          float g = grad_out[((row_base + samp_q[0]) * d.M + m) * kD48 + chan];
          for (int s = 0; s < ns; ++s) {
            const int sn = (s + 1 < ns) ? s + 1 : s;             // prefetch the next sample
            const int pix_n = samp[sn].pix[tap_k];
            const float w_n = reinterpret_cast<const float *>(&samp[sn].w)[tap_k];
            const float g_n = grad_out[((row_base + samp_q[sn]) * d.M + m) * kD48 + chan];
            if (pix >= 0 && plan.debug != 1) {
              float *dst = tile + pix * kD48 + chan;
              *dst = fmaf(w, g, *dst);
            }
            pix = pix_n; w = w_n; g = g_n;
          }
        }
        __syncthreads();
        if (threadIdx.x == 0) *n_samp = 0;
        // (the barrier that follows stage B's first append is the next __syncthreads below or above)
        __syncthreads();
      }
      if (threadIdx.x == 0) *n_hit = 0;
      __syncthreads();
    }
  }

  // ---- write the tile home: plain stores, 16 B per lane, 12 lanes per pixel ----------------------
  const size_t img_base = ((size_t)n * d.S + me.start) * d.M;
  for (int i = threadIdx.x; i < npx * (kD48 / 4); i += kOwnerBlock) {
    const int pix = i / (kD48 / 4), c4 = i % (kD48 / 4);
    const int ty = win.ty0 + pix / me.tw, tx = win.tx0 + pix % me.tw;
    if (ty < me.H && tx < me.W) {
      const f32x4 v = *reinterpret_cast<const f32x4 *>(tile + pix * kD48 + c4 * 4);
      float *dstp = grad_value + ((img_base + (size_t)(ty * me.W + tx) * d.M + m) * kD48) + c4 * 4;
      *reinterpret_cast<f32x4 *>(dstp) = v;
    }
  }
}

}  // namespace snipper
