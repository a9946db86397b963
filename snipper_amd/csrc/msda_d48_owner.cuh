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
// tile with LDS float atomics (ds_add_f32).  The tile is then written to HBM once with plain stores --
// no HBM atomics and no pre-zeroing, since every pixel of grad_value has exactly one owner.
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
constexpr int kOwnerBlock = 256;
constexpr int kOwnerScanPerThread = 2;                        // candidates a thread examines per batch
constexpr int kOwnerListCap = kOwnerBlock * kOwnerScanPerThread * 4;  // every candidate may yield 4 taps
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
};
inline size_t owner_lds_bytes(int max_tile_px) {
  return (size_t)max_tile_px * kD48 * 4 + (size_t)kOwnerListCap * 8 + 16;
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

__global__ __launch_bounds__(kOwnerBlock) void msda_bwd_d48_owner_kernel(
    const float *__restrict__ grad_out, const float *__restrict__ loc, const float *__restrict__ attn,
    CoreDims d, OwnerPlan plan, float *__restrict__ grad_value) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float *tile = reinterpret_cast<float *>(smem_raw);                                   // [th*tw][48]
  unsigned *list_key = reinterpret_cast<unsigned *>(tile + plan.max_tile_px * kD48);    // q << 8 | pixel
  float *list_w = reinterpret_cast<float *>(list_key + kOwnerListCap);
  int &list_n = *reinterpret_cast<int *>(list_w + kOwnerListCap);                      // hit counter

  // ---- which tile am I?  launch order: coarse levels first (their tiles see more candidates) ----
  const int tiles = plan.total_tiles;
  int b = blockIdx.x;
  const int tile_id = b % tiles;  b /= tiles;
  const int m = b % d.M;
  const int n = b / d.M;
  int l = 0;
  for (int i = 1; i < plan.L; ++i) l = (tile_id >= plan.lv[i].tile_base) ? i : l;
  const OwnerLevel me = plan.lv[l];
  const int t = tile_id - me.tile_base;
  const int ty0 = (t / me.ntx) * me.th, tx0 = (t % me.ntx) * me.tw;
  const int npx = me.th * me.tw;
  const float R = plan.radius;

  for (int i = threadIdx.x; i < npx * kD48; i += kOwnerBlock) tile[i] = 0.f;
  if (threadIdx.x == 0) list_n = 0;
  __syncthreads();

  const int LP = d.L * d.P;
  const int grp = threadIdx.x >> 4, lane = threadIdx.x & 15;
  const size_t row_base = (size_t)n * d.Lq;     // rows of this batch element: (row_base + q) * M + m

  for (int lq = 0; lq < plan.L; ++lq) {
    const int Hq = plan.lv[lq].H, Wq = plan.lv[lq].W, sq = plan.lv[lq].start;
    int qx0, qx1, qy0, qy1;
    anchor_range((float)(tx0 - 1) - R, (float)(tx0 + me.tw) + R, me.W, Wq, qx0, qx1);
    anchor_range((float)(ty0 - 1) - R, (float)(ty0 + me.th) + R, me.H, Hq, qy0, qy1);
    const int rw = qx1 - qx0 + 1, rh = qy1 - qy0 + 1;
    if (rw <= 0 || rh <= 0) continue;
    const int ncand = rw * rh * d.P;
    for (int base = 0; base < ncand; base += kOwnerBlock * kOwnerScanPerThread) {
      // ---- scan: one (query, point) candidate per thread and slot --------------------------------
#pragma unroll
      for (int u = 0; u < kOwnerScanPerThread; ++u) {
        const int c = base + u * kOwnerBlock + (int)threadIdx.x;
        if (c < ncand) {
          const int p = c % d.P, r = c / d.P;
          const int qy = qy0 + r / rw, qx = qx0 + r % rw;
          const int q = sq + qy * Wq + qx;
          const size_t li = ((row_base + q) * d.M + m) * LP + l * d.P + p;
          const float2 xy = *reinterpret_cast<const float2 *>(loc + 2 * li);
          const float x = px_coord(xy.x, me.W), y = px_coord(xy.y, me.H);
          const bool inside = (y > -1.f) && (x > -1.f) && (y < (float)me.H) && (x < (float)me.W);
          const float ax = anchor_coord(qx, me.W, Wq), ay = anchor_coord(qy, me.H, Hq);
          if (inside && near_anchor(x, y, ax, ay, R)) {
            const float yf = floorf(y), xf = floorf(x);
            const int y0 = (int)yf, x0 = (int)xf;
            const float lh = y - yf, lw = x - xf, hh = 1.f - lh, hw = 1.f - lw;
            const float a = attn[li];
            const float w4[4] = {hh * hw * a, hh * lw * a, lh * hw * a, lh * lw * a};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const int ty = y0 + (k >> 1), tx = x0 + (k & 1);
              const bool in_map = ty >= 0 && ty <= me.H - 1 && tx >= 0 && tx <= me.W - 1;
              const int py = ty - ty0, pxl = tx - tx0;
              if (in_map && py >= 0 && py < me.th && pxl >= 0 && pxl < me.tw) {
                const int slot = atomicAdd(&list_n, 1);
                list_key[slot] = ((unsigned)q << 8) | (unsigned)(py * me.tw + pxl);
                list_w[slot] = w4[k];
              }
            }
          }
        }
      }
      __syncthreads();
      // ---- accumulate: 16 lanes per hit, lane i adds channels i, i+16, i+32 ---------------------
      const int cnt = list_n;
      for (int e = grp; e < cnt; e += kOwnerBlock / 16) {
        const unsigned key = list_key[e];
        const float w = list_w[e];
        const float *g = grad_out + ((row_base + (key >> 8)) * d.M + m) * kD48 + lane;
        float *dst = tile + (key & 255u) * kD48 + lane;
        const float g0 = g[0], g1 = g[16], g2 = g[32];
        atomicAdd(dst, w * g0);
        atomicAdd(dst + 16, w * g1);
        atomicAdd(dst + 32, w * g2);
      }
      __syncthreads();
      if (threadIdx.x == 0) list_n = 0;
      __syncthreads();
    }
  }

  // ---- write the tile home: plain stores, 16 B per lane, 12 lanes per pixel ----------------------
  const size_t img_base = ((size_t)n * d.S + me.start) * d.M;
  for (int i = threadIdx.x; i < npx * (kD48 / 4); i += kOwnerBlock) {
    const int pix = i / (kD48 / 4), c4 = i % (kD48 / 4);
    const int ty = ty0 + pix / me.tw, tx = tx0 + pix % me.tw;
    if (ty < me.H && tx < me.W) {
      const f32x4 v = *reinterpret_cast<const f32x4 *>(tile + pix * kD48 + c4 * 4);
      float *dstp = grad_value + ((img_base + (size_t)(ty * me.W + tx) * d.M + m) * kD48) + c4 * 4;
      *reinterpret_cast<f32x4 *>(dstp) = v;
    }
  }
}

}  // namespace snipper
