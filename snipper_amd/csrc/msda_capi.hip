// msda_capi.hip -- extern "C" entry points of libsnipper_msda.so (see include/snipper_msda.h).
//
// Replaces the reference's two launchers, ms_deformable_im2col_cuda / ms_deformable_col2im_cuda
// (/root/reference/models/ops/src/cuda/ms_deform_im2col_cuda.cuh:923-954, :956-1327): validate,
// pick a kernel variant, launch on the caller's stream, report errors.  No allocation, no host
// synchronisation, no global mutable state besides a thread-local "last variant" string and the
// process-wide variant policy (tests / benchmarks only).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <string_view>

#include "../../include/snipper_msda.h"
#include "../../include/snipper_dense.h"
#include "../../include/snipper_layers.h"
#include "gemm_bf16.cuh"
#include "wgrad_bf16.cuh"
#include "wres_gemm_bf16.cuh"
#include "wgrad_ring_bf16.cuh"
#include "wgrad_wide_bf16.cuh"
#include "wgrad_conv_patch_bf16.cuh"
#include "misc_kernels.cuh"
#include "conv3x3_ring_bf16.cuh"
#include "conv3x3_patch_bf16.cuh"
#include "small_linear.cuh"
#include "small_attention.cuh"
#include "match_cost.cuh"
#include "ln_fused.cuh"
#include "small_ln.cuh"
#include "gn_tokens.cuh"
#include "pair_losses.cuh"
#include "heatmap_blur.cuh"
#include "heatmap_loss.cuh"
#include "msda_prologue.cuh"
#include "lsap.cuh"
#include "adamw_flat.cuh"
#include "msda_d48.cuh"
#include "msda_d48_patch.cuh"
#include "msda_d48_tilemm.cuh"
#include "msda_d48_sparse.cuh"
#include "msda_generic.cuh"

using namespace snipper;

namespace {

// name of the kernel variant the last core-op call OF THE CALLING THREAD dispatched to: a diagnostic for tests and profiles that
// never influences a result.  Thread-local: the library holds no process-wide mutable state (two callers -- the autograd
// engine's thread and the main thread -- used to race on one global; a caller that wants the variant of a call made on another
// thread reads it there, right after the call, as snipper_amd/MultiScaleDeformableAttention.py does)
thread_local const char *g_last_variant = "none";

constexpr int kWgradWgs = 512;          // workgroups of the split-reduction weight gradient (2 per CU; 384 .. 1024 measured)
constexpr int kLnBwdBlocks = 1536;      // 6 workgroups per CU x 256 CUs: one residency wave of the LayerNorm backward

// the library keeps no tuning state: every knob travels in the caller's snipper_msda_config (NULL = these defaults)
snipper_msda_config default_config() {
  snipper_msda_config c{};
  c.struct_bytes = (int32_t)sizeof(snipper_msda_config);
  c.policy = 0;
  // 24 px: measured at N = 8, bf16 value, reference offset bias grid + N(0, sigma) px (tools/sweep_radius.sh, profiles/
  // r03_radius_sweep.txt): backward 0.85 / 0.96 / 0.96 ms at sigma ~0 / 3 / 8 px, against 0.87 / 1.39 / 2.34 ms at round 2's
  // 6 px, which had been tuned on freshly initialised offsets only
  c.near_radius = 24.0f;
  c.tile_edge[0] = 0; c.tile_edge[1] = 0; c.tile_edge[2] = 0;     // 0 = the kernel's own best (patch_edge below)
  return c;
}
bool config_ok(const snipper_msda_config *cfg) {
  if (!cfg) return true;
  if (cfg->struct_bytes != (int32_t)sizeof(snipper_msda_config)) return false;
  if (cfg->policy < 0 || cfg->policy > 2) return false;
  if (!(cfg->near_radius >= 0.f && cfg->near_radius <= 64.f)) return false;
  for (int i = 0; i < 3; ++i) {
    const int e = cfg->tile_edge[i];
    if (e < 0 || e > 16 || (e & (e - 1))) return false;       // (0 = auto)
  }
  if (cfg->tile_kernel < 0 || cfg->tile_kernel > 2) return false;
  // the header's "must be 0": a caller that fills the struct by hand without zeroing it must not get through
  for (int i = 0; i < 3; ++i)
    if (cfg->reserved[i] != 0) return false;
  if (cfg->value_layout != 0 && cfg->value_layout != 1) return false;
  // debug_ablation selects timing ablations of the owner-computes backward (WRONG gradients): only a process that asked
  // for them in its environment BEFORE the library was loaded gets them (read once: no mutable state)
  static const bool allow_debug = [] { const char *e = getenv("SNIPPER_MSDA_ALLOW_DEBUG"); return e && e[0] == '1'; }();
  if (cfg->debug_ablation != 0 && !allow_debug) return false;
  return true;
}
snipper_msda_config resolve(const snipper_msda_config *cfg) { return cfg ? *cfg : default_config(); }

int check_dims(int N, int S, int M, int D, int L, int Lq, int P) {
  if (N <= 0 || S <= 0 || M <= 0 || D <= 0 || L <= 0 || Lq <= 0 || P <= 0) return SNIPPER_E_SHAPE;
  if ((long long)S * M * D >= (1LL << 31)) return SNIPPER_E_SHAPE;
  if ((long long)Lq * M * L * P * 2 >= (1LL << 31)) return SNIPPER_E_SHAPE;
  if ((long long)L * P > 1024) return SNIPPER_E_SHAPE;
  return SNIPPER_OK;
}

int launch_status() {
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? SNIPPER_OK : (int)e;
}

// smallest lane-group size covering D whose sample-point staging fits 64 KiB of LDS
template <typename CT> int pick_group(int D, int LP, size_t *lds) {
  const int cands[3] = {4, 16, 64};
  for (int i = 0; i < 3; ++i) {
    const int G = cands[i];
    if (G < D && G != 64) continue;
    const size_t need = (size_t)(kGenericBlock / G) * LP * sizeof(SamplePoint<CT>);
    if (need <= 64 * 1024) {
      *lds = need;
      return G;
    }
  }
  return 0;
}

int generic_grid(long long rows, int rows_per_block) {
  const long long want = (rows + rows_per_block - 1) / rows_per_block;
  const long long cap = 256LL * 16;  // 256 CUs x 16 resident blocks, grid-stride beyond
  return (int)(want < cap ? want : cap);
}

template <typename VT, typename CT>
int forward_generic(hipStream_t st, const VT *value, const int64_t *shapes, const int64_t *lsi,
                    const CT *loc, const CT *attn, CoreDims d, VT *out) {
  size_t lds = 0;
  const int G = pick_group<CT>(d.D, d.L * d.P, &lds);
  if (!G) return SNIPPER_E_SHAPE;
  const long long rows = (long long)d.N * d.Lq * d.M;
  const int grid = generic_grid(rows, kGenericBlock / G);
  g_last_variant = "generic";
  switch (G) {
    case 4: hipLaunchKernelGGL((msda_fwd_generic_kernel<VT, CT, 4>), dim3(grid), dim3(kGenericBlock), lds, st, value, shapes, lsi, loc, attn, d, out); break;
    case 16: hipLaunchKernelGGL((msda_fwd_generic_kernel<VT, CT, 16>), dim3(grid), dim3(kGenericBlock), lds, st, value, shapes, lsi, loc, attn, d, out); break;
    default: hipLaunchKernelGGL((msda_fwd_generic_kernel<VT, CT, 64>), dim3(grid), dim3(kGenericBlock), lds, st, value, shapes, lsi, loc, attn, d, out); break;
  }
  return launch_status();
}

template <typename VT, typename CT, typename GVT>
int backward_generic(hipStream_t st, const VT *grad_out, const VT *value, const int64_t *shapes,
                     const int64_t *lsi, const CT *loc, const CT *attn, CoreDims d,
                     GVT *grad_value, CT *grad_loc, CT *grad_attn) {
  size_t lds = 0;
  const int G = pick_group<CT>(d.D, d.L * d.P, &lds);
  if (!G) return SNIPPER_E_SHAPE;
  const long long rows = (long long)d.N * d.Lq * d.M;
  const int grid = generic_grid(rows, kGenericBlock / G);
  g_last_variant = "generic";
  switch (G) {
    case 4: hipLaunchKernelGGL((msda_bwd_generic_kernel<VT, CT, GVT, 4>), dim3(grid), dim3(kGenericBlock), lds, st, grad_out, value, shapes, lsi, loc, attn, d, grad_value, grad_loc, grad_attn); break;
    case 16: hipLaunchKernelGGL((msda_bwd_generic_kernel<VT, CT, GVT, 16>), dim3(grid), dim3(kGenericBlock), lds, st, grad_out, value, shapes, lsi, loc, attn, d, grad_value, grad_loc, grad_attn); break;
    default: hipLaunchKernelGGL((msda_bwd_generic_kernel<VT, CT, GVT, 64>), dim3(grid), dim3(kGenericBlock), lds, st, grad_out, value, shapes, lsi, loc, attn, d, grad_value, grad_loc, grad_attn); break;
  }
  return launch_status();
}

// The D=48 kernels address `value` through one raw-buffer descriptor with 32-bit byte offsets.
template <typename VT> bool d48_eligible(const CoreDims &d, int policy = 0) {
  if (policy == 1) return false;
  if ((d.D != kD48 && d.D != 24) || d.L > kMaxLevelsFast) return false;        // 48 = hidden 384, 24 = the reference's default 192
  if ((long long)d.N * d.S * d.M * d.D * (long long)sizeof(VT) >= (1LL << 31)) return false;
  return d.L * d.P <= 64;
}

template <typename VT, int DT>
int forward_d48_t(hipStream_t st, const VT *value, const int64_t *shapes, const int64_t *lsi,
                  const float *loc, const float *attn, CoreDims d, VT *out, int out_bf16) {
  constexpr int kRows = kD48Block / (D48Fwd<VT>::G * DT / kD48);
  const long long rows = (long long)d.N * d.Lq * d.M;
  const int LP = d.L * d.P;
  const int nblk = (int)((rows + kRows - 1) / kRows);
  const int nblk_padded = (nblk + 7) & ~7;
  const size_t lds = (size_t)kRows * (LP * sizeof(FwdRecord) + 16);
  if (LP == 12 && d.P == 4) {
    g_last_variant = DT == kD48 ? "d48_lp12" : "d24_lp12";
    hipLaunchKernelGGL((msda_fwd_d48_kernel<VT, 12, DT>), dim3(nblk_padded), dim3(kD48Block), lds, st, value, shapes, lsi, loc, attn, d, out, nblk_padded, out_bf16);
  } else {
    g_last_variant = DT == kD48 ? "d48" : "d24";
    hipLaunchKernelGGL((msda_fwd_d48_kernel<VT, 0, DT>), dim3(nblk_padded), dim3(kD48Block), lds, st, value, shapes, lsi, loc, attn, d, out, nblk_padded, out_bf16);
  }
  return launch_status();
}
template <typename VT>
int forward_d48(hipStream_t st, const VT *value, const int64_t *shapes, const int64_t *lsi,
                const float *loc, const float *attn, CoreDims d, VT *out, int out_bf16 = 0) {
  return d.D == kD48 ? forward_d48_t<VT, kD48>(st, value, shapes, lsi, loc, attn, d, out, out_bf16)
                     : forward_d48_t<VT, 24>(st, value, shapes, lsi, loc, attn, d, out, out_bf16);
}

template <int DT, typename VT = float>
int backward_d48_f32_t(hipStream_t st, const float *grad_out, const VT *value, const int64_t *shapes,
                       const int64_t *lsi, const float *loc, const float *attn, CoreDims d,
                       float *grad_value, float *grad_loc, float *grad_attn, int go_bf16) {
  constexpr int kRows = kD48Block / (DT / 3);
  const long long rows = (long long)d.N * d.Lq * d.M;
  const int LP = d.L * d.P;
  const int nblk = (int)((rows + kRows - 1) / kRows);
  const int nblk_padded = (nblk + 7) & ~7;
  const size_t lds = (size_t)kRows * (LP * sizeof(BwdRecord) + 16);
  if (LP == 12 && d.P == 4) {
    g_last_variant = DT == kD48 ? "d48_lp12" : "d24_lp12";
    hipLaunchKernelGGL((msda_bwd_d48_f32_kernel<12, DT, VT>), dim3(nblk_padded), dim3(kD48Block), lds, st, grad_out, value, shapes, lsi, loc, attn, d, grad_value, grad_loc, grad_attn, nblk_padded, go_bf16);
  } else {
    g_last_variant = DT == kD48 ? "d48" : "d24";
    hipLaunchKernelGGL((msda_bwd_d48_f32_kernel<0, DT, VT>), dim3(nblk_padded), dim3(kD48Block), lds, st, grad_out, value, shapes, lsi, loc, attn, d, grad_value, grad_loc, grad_attn, nblk_padded, go_bf16);
  }
  return launch_status();
}
int backward_d48_f32(hipStream_t st, const float *grad_out, const float *value, const int64_t *shapes,
                     const int64_t *lsi, const float *loc, const float *attn, CoreDims d,
                     float *grad_value, float *grad_loc, float *grad_attn, int go_bf16 = 0) {
  return d.D == kD48 ? backward_d48_f32_t<kD48>(st, grad_out, value, shapes, lsi, loc, attn, d, grad_value, grad_loc, grad_attn, go_bf16)
                     : backward_d48_f32_t<24>(st, grad_out, value, shapes, lsi, loc, attn, d, grad_value, grad_loc, grad_attn, go_bf16);
}

// ---- owner-computes backward for the encoder shape (msda_d48_patch.cuh): needs the level shapes on the HOST ----
// Returns false when the geometry does not fit the marks' bounds (then the plain D=48 kernels run).
// grad_value tile edge of a level class (0 big, 1 mid, 2 small).  tile_edge = 0 (the default) lets the grad_value-side kernel
// pick: 16 / 8 / 4 for the vector / LDS kernel (round 1-3 sweeps), 16 / 8 / 8 for the matrix-pipe kernel, whose cost is per
// hit and per workgroup, not per tap (whole backward at N = 8, sigma 0 / 3 px: 0.669 / 0.844 ms against 0.677 / 0.861)
// Round 5: 8 / 8 / 8 for the matrix-pipe kernel.  Once its workgroups walk the (n, m) pairs of an XCD TILE-major (t3_order in
// msda_d48_tilemm.cuh: a tile's hit rows are fetched once for all heads) the 8 x 8-tile instance at five workgroups per CU
// beats the 16 x 16-tile instance on the finest level as well: tile kernels 121 + 113 -> 188 us at sigma 0, 178 + 155 -> 261 us
// at 3 px (query side + 2 ... + 7 us for the extra mark words); 8 / 8 / 4, 8 / 4 / 4, 4 / 4 / 4: 196 / 206 / 245 us and a
// query side of 210 / 291 / 592 us (profiles/r05_tile_order_and_edges.txt).
int patch_edge(const snipper_msda_config &cfg, int cls, bool mfma_tiles) {
  if (cfg.tile_edge[cls] > 0) return cfg.tile_edge[cls];
  if (mfma_tiles) return 8;
  return cls == 0 ? 16 : (cls == 1 ? 8 : 4);
}
// which grad_value-side kernel a backward call with these row / value types takes (config.tile_kernel: 1 forces the vector one)
bool patch_uses_mfma(const CoreDims &d, const snipper_msda_config &cfg, int go_bf16) {
  const bool fits = (long long)d.Lq * d.M * d.L * kPatchP * 8 < (1LL << 31) && (long long)d.Lq * d.M * kT3RowB < (1LL << 31);
  return go_bf16 && fits && (cfg.tile_kernel == 2 || (cfg.tile_kernel == 0 && !cfg.debug_ablation));
}

bool make_patch_plan_r(const CoreDims &d, const int64_t *hs, const snipper_msda_config &cfg, bool mfma_tiles, float radius,
                       PatchPlan *out) {
  PatchPlan p{};
  p.L = d.L;
  p.radius = radius;
  p.debug = cfg.debug_ablation;
#ifdef TILE2_STAMPS       // diagnostic builds only: SNIPPER_TILE2_STAMPS=<device address, hex> of a 256 x 128 x 8-byte buffer
  static unsigned long long *const stamps = [] { const char *e = getenv("SNIPPER_TILE2_STAMPS"); return e ? (unsigned long long *)strtoull(e, nullptr, 16) : nullptr; }();
  p.stamps = stamps;
#endif
  int start = 0, tbase = 0, bbase = 0;
  for (int l = 0; l < d.L; ++l) {
    PatchLevel &v = p.lv[l];
    v.H = (int)hs[2 * l];
    v.W = (int)hs[2 * l + 1];
    v.start = start;
    start += v.H * v.W;
    v.nbx = (v.W + kPatchB - 1) / kPatchB;
    v.nby = (v.H + kPatchB - 1) / kPatchB;
    v.blk_base = bbase;
    bbase += v.nbx * v.nby;
    const int area = v.H * v.W;
    const int e = patch_edge(cfg, area > 4096 ? 0 : (area > 1024 ? 1 : 2), mfma_tiles);
    v.shift = e >= 16 ? 4 : (e >= 8 ? 3 : (e >= 4 ? 2 : (e >= 2 ? 1 : 0)));
    const int edge = 1 << v.shift;
    v.ntx = (v.W + edge - 1) / edge;
    v.nty = (v.H + edge - 1) / edge;
    v.tile_base = tbase;
    tbase += v.ntx * v.nty;
  }
  p.nblocks = bbase;
  p.total_tiles = tbase;
  for (int a = 0; a < d.L; ++a)
    for (int b = 0; b < d.L; ++b) {      // IEEE float division on the host == __fdiv_rn on the device
      p.rw[a][b] = (float)p.lv[a].W / (float)p.lv[b].W;
      p.rh[a][b] = (float)p.lv[a].H / (float)p.lv[b].H;
    }
  // marks geometry: bounds of the candidate block rectangles (patch_tile_cand: the span (edge + 1 + 2R) pixels of
  // level l rescaled to level lq, +2 cells of margin and +2 for rounding, in blocks of 8 queries, +2 for alignment)
  long long lvl_base = 0;
  for (int l = 0; l < d.L; ++l) {
    const int edge = 1 << p.lv[l].shift;
    int off = 0;
    for (int lq = 0; lq < d.L; ++lq) {
      const double span = (double)edge + 1.0 + 2.0 * p.radius;
      int qw = (int)(span * p.lv[lq].W / p.lv[l].W) + 5, qh = (int)(span * p.lv[lq].H / p.lv[l].H) + 5;
      int bw = qw / kPatchB + 2, bh = qh / kPatchB + 2;
      if (bw > p.lv[lq].nbx) bw = p.lv[lq].nbx;
      if (bh > p.lv[lq].nby) bh = p.lv[lq].nby;
      p.cbw[l][lq] = bw;
      p.cbh[l][lq] = bh;
      p.coff[l][lq] = off;
      off += bw * bh;
    }
    if (off > kPatchMaxCand) return false;
    p.tstride[l] = off;
    p.lvl_base[l] = lvl_base;
    lvl_base += (long long)p.lv[l].ntx * p.lv[l].nty * off;
  }
  p.words_per_nm = lvl_base;
  *out = p;
  return true;
}
// The plan at the configured near radius, or -- when a tile's candidate rectangles would not fit one thread each
// (kPatchMaxCand; maps well above 600 x 800: 720 x 1280, 1080 x 1920) -- at the largest smaller radius of a fixed ladder
// that fits.  Near + far is a partition of the taps at ANY radius (tests/test_owner_gpu.py), so the radius only decides how
// many taps go the fast way; the same ladder runs in the workspace query and in the launch, so both see the same plan.
bool make_patch_plan(const CoreDims &d, const int64_t *hs, const snipper_msda_config &cfg, bool mfma_tiles, PatchPlan *out) {
  if (make_patch_plan_r(d, hs, cfg, mfma_tiles, cfg.near_radius, out)) return true;
  const float ladder[] = {16.f, 12.f, 8.f, 6.f, 4.f, 3.f};
  for (float r : ladder)
    if (r < cfg.near_radius && make_patch_plan_r(d, hs, cfg, mfma_tiles, r, out)) return true;
  return false;
}

// workspace of the owner-computes backward: the marks, the far list's counter (zeroed with the marks), the far list (one
// 8-byte entry per sample at most: never zeroed, only its first far_count entries are read)
inline long long patch_zeroed_bytes(long long nm, const PatchPlan &plan) { return nm * plan.words_per_nm * 8 + 8; }
inline long long patch_workspace_bytes(long long nm, const CoreDims &d, const PatchPlan &plan) {
  return patch_zeroed_bytes(nm, plan) + (long long)d.N * d.Lq * d.M * d.L * kPatchP * 8;
}

template <typename VT>
int backward_d48_patch(hipStream_t st, const void *grad_out, const VT *value, const float *loc, const float *attn,
                       CoreDims d, PatchPlan plan, void *workspace, float *grad_value, float *grad_loc,
                       float *grad_attn, int go_bf16, bool mfma_tiles) {
  const long long nm = (long long)d.N * d.M;
  plan.marks = reinterpret_cast<unsigned long long *>(workspace);
  plan.far_count = reinterpret_cast<unsigned *>(plan.marks + nm * plan.words_per_nm);
  plan.far_list = reinterpret_cast<uint2 *>(plan.far_count + 2);
  if ((long long)d.N * d.Lq * d.M * d.L * kPatchP >= (1LL << 32)) return SNIPPER_E_SHAPE;      // (sample indices are 32-bit)
  hipError_t e = hipMemsetAsync(plan.marks, 0, (size_t)patch_zeroed_bytes(nm, plan), st);
  if (e != hipSuccess) return (int)e;
  // 1) query side: grad_loc / grad_attn, marks, the list of samples with taps no tile owns
  // 2) grad_value side: every tile stores what it owns (plain stores: grad_value needs no zeroing)
  // 3) the far list's taps: HBM float atomics on top of the stored tiles
  auto launch_far = [&]() {
    if (go_bf16) hipLaunchKernelGGL(msda_bwd_d48_far_kernel<true>, dim3(kFarBlocks), dim3(kPatchThreads), 0, st, grad_out, loc, attn, d, plan, grad_value);
    else hipLaunchKernelGGL(msda_bwd_d48_far_kernel<false>, dim3(kFarBlocks), dim3(kPatchThreads), 0, st, grad_out, loc, attn, d, plan, grad_value);
    return launch_status();
  };
  const long long nblk = nm * plan.nblocks;
  const long long nblk_padded = (nblk + 7) & ~7LL;
  // 2) grad_value side: every tile adds what it owns (XCD-major grid, see the kernel)
  const long long nblk_tiles = ((nm + 7) / 8) * 8 * plan.total_tiles;
  if (nblk_padded >= (1LL << 31) || nblk_tiles >= (1LL << 31)) return SNIPPER_E_SHAPE;
  if (go_bf16) {
    if (plan.L <= 3)
      hipLaunchKernelGGL((msda_bwd_d48_patchbin_kernel<VT, true, 3>), dim3((unsigned)nblk_padded), dim3(kPatchThreads), 0, st, grad_out,
                         value, loc, attn, d, plan, grad_value, grad_loc, grad_attn, (int)nblk_padded);
    else
      hipLaunchKernelGGL((msda_bwd_d48_patchbin_kernel<VT, true, 4>), dim3((unsigned)nblk_padded), dim3(kPatchThreads), 0, st, grad_out,
                         value, loc, attn, d, plan, grad_value, grad_loc, grad_attn, (int)nblk_padded);
    if (int rc = launch_status()) return rc;
    // grad_value side: dense per-tile scatter on the matrix pipe (msda_d48_tilemm.cuh) unless the caller asks for the
    // vector / LDS kernel (config.tile_kernel = 1), see patch_uses_mfma
    if (mfma_tiles) {
      // persistent: two workgroups per CU (its LDS footprint), a multiple of the 8 XCDs
      // (persistent workgroups -- two per CU walking a static list of tiles with the next tile's marks prefetched -- were
      //  measured: 373 / 544 / 664 us at sigma 0 / 3 / 8 px against 330 / 478 / 588 for one workgroup per tile: the
      //  dispatcher balances the very unequal tiles better than a static list, and the fixed cost is not launch latency but
      //  the read-modify-write of grad_value)
      // two instances by tile size (their LDS footprints differ 2.4x: two against five workgroups per CU); small tiles first:
      // they hold the longest-running workgroups (coarse levels)
      T3Class big{}, small{}, tiny{};      // 16 x 16 | 8 x 8 | 4 x 4 and smaller tiles: one kernel instance each
      for (int l = 0; l < kPatchMaxLevels; ++l) {
        big.base[l] = small.base[l] = tiny.base[l] = -1;
        if (l >= plan.L) continue;
        T3Class &c = plan.lv[l].shift >= 4 ? big : (plan.lv[l].shift == 3 ? small : tiny);
        c.base[l] = c.tiles;
        c.tiles += plan.lv[l].ntx * plan.lv[l].nty;
      }
      const long long nm8 = ((nm + 7) / 8) * 8;
      if (plan.debug & 128) {         // (debug bit 128: the walk region by region instead of level after level, for A/B runs --
                                      //  same results; measured SLOWER, profiles/r05_region_order_ab.txt)
        t3_region_order(plan, tiny);
        t3_region_order(plan, small);
      }
      if (tiny.tiles) {
        hipLaunchKernelGGL(msda_bwd_d48_tile3_kernel<16>, dim3((unsigned)(nm8 * tiny.tiles)), dim3(kPatchThreads), 0, st, grad_out,
                           loc, attn, d, plan, tiny, grad_value);
        if (int rc = launch_status()) return rc;
      }
      if (small.tiles) {
        hipLaunchKernelGGL(msda_bwd_d48_tile3_kernel<64>, dim3((unsigned)(nm8 * small.tiles)), dim3(kPatchThreads), 0, st, grad_out,
                           loc, attn, d, plan, small, grad_value);
        if (int rc = launch_status()) return rc;
      }
      if (big.tiles) {      // 16 x 16 tiles: 8-wave workgroups (tile_kernel = 2 with debug bit 32: the 4-wave form, for A/B runs)
        if (plan.debug & 32)
          hipLaunchKernelGGL(msda_bwd_d48_tile3_kernel<256>, dim3((unsigned)(nm8 * big.tiles)), dim3(kPatchThreads), 0, st, grad_out,
                             loc, attn, d, plan, big, grad_value);
        else
          hipLaunchKernelGGL(msda_bwd_d48_tile3_wide_kernel, dim3((unsigned)(nm8 * big.tiles)), dim3(kT3WideThreads), 0, st, grad_out,
                             loc, attn, d, plan, big, grad_value);
      }
      g_last_variant = plan.debug ? "d48_owner_mfma_debug" : "d48_owner_mfma";
      if (int rc = launch_status()) return rc;
      return launch_far();
    }
    hipLaunchKernelGGL((msda_bwd_d48_tile2_kernel<128, true>), dim3((unsigned)nblk_tiles), dim3(kPatchThreads), 0, st, grad_out,
                       loc, attn, d, plan, grad_value);
  } else {
    hipLaunchKernelGGL((msda_bwd_d48_patchbin_kernel<VT, false>), dim3((unsigned)nblk_padded), dim3(kPatchThreads), 0, st, grad_out,
                       value, loc, attn, d, plan, grad_value, grad_loc, grad_attn, (int)nblk_padded);
    if (int rc = launch_status()) return rc;
    hipLaunchKernelGGL((msda_bwd_d48_tile2_kernel<128, false>), dim3((unsigned)nblk_tiles), dim3(kPatchThreads), 0, st, grad_out,
                       loc, attn, d, plan, grad_value);
  }
  g_last_variant = plan.debug ? "d48_owner_debug" : "d48_owner";
  if (int rc = launch_status()) return rc;
  return launch_far();
}

bool owner_shape_ok(const CoreDims &d, const int64_t *hs, int policy) {
  if (policy != 0 || !hs || d.D != kD48 || !d48_eligible<float>(d, policy) || d.L > kPatchMaxLevels || d.P != kPatchP || d.Lq != d.S) return false;
  long long sum = 0;
  for (int l = 0; l < d.L; ++l) {
    if (hs[2 * l] <= 0 || hs[2 * l + 1] <= 0) return false;
    if (hs[2 * l] >= 32768 || hs[2 * l + 1] >= 32768) return false;     // hit queries are packed level<<30 | qy<<15 | qx
    sum += hs[2 * l] * hs[2 * l + 1];
  }
  if ((long long)d.N * d.Lq * d.M * d.L * kPatchP >= (1LL << 32)) return false;       // (sample indices are 32-bit)
  return sum == d.S && d.S < (1 << 24);
}

template <typename GVT> int zero_grad_value(hipStream_t st, GVT *gv, const CoreDims &d) {
  const hipError_t e = hipMemsetAsync(gv, 0, sizeof(GVT) * (size_t)d.N * d.S * d.M * d.D, st);
  return e == hipSuccess ? SNIPPER_OK : (int)e;
}

}  // namespace

extern "C" {

int snipper_msda_abi_version(void) { return SNIPPER_MSDA_ABI_VERSION; }

const char *snipper_msda_strerror(int code) {
  switch (code) {
    case SNIPPER_OK: return "ok";
    case SNIPPER_E_NULL: return "null pointer argument";
    case SNIPPER_E_SHAPE: return "size <= 0 or beyond the documented limits";
    case SNIPPER_E_UNSUPPORTED: return "variant not built";
    default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown snipper_msda error";
  }
}

const char *snipper_msda_last_variant(void) { return g_last_variant; }

void snipper_msda_config_init(snipper_msda_config *cfg) {
  if (cfg) *cfg = default_config();
}

// ---- the general entry points: dtype codes 0 = float32, 1 = bfloat16, 2 = float64 ----
int snipper_msda_forward_ex(void *stream, const snipper_msda_config *cfg, const int64_t *host_shapes, const void *value,
                            int value_dtype, const int64_t *shapes, const int64_t *level_start, const void *loc,
                            const void *attn, int N, int S, int M, int D, int L, int Lq, int P, void *out, int out_dtype) {
  if (!value || !shapes || !level_start || !loc || !attn || !out) return SNIPPER_E_NULL;
  if (int rc = check_dims(N, S, M, D, L, Lq, P)) return rc;
  if (!config_ok(cfg)) return SNIPPER_E_SHAPE;
  const snipper_msda_config c = resolve(cfg);
  const CoreDims d{N, S, M, D, L, Lq, P, c.value_layout};
  hipStream_t st = (hipStream_t)stream;
  // head-major value rows: the tuned D = 48 / 24 kernels only (the generic ones would silently read the other layout)
  if (d.head_major && (value_dtype == 2 || !d48_eligible<float>(d, c.policy))) return SNIPPER_E_UNSUPPORTED;
  if (value_dtype == 2) {
    if (out_dtype != 2) return SNIPPER_E_UNSUPPORTED;
    return forward_generic<double, double>(st, (const double *)value, shapes, level_start, (const double *)loc,
                                           (const double *)attn, d, (double *)out);
  }
  if (value_dtype == 1) {
    if (out_dtype != 1 && out_dtype != 0) return SNIPPER_E_UNSUPPORTED;
    if (d48_eligible<uint16_t>(d, c.policy))      // (out_dtype 0: float32 rows from a bf16 value -- the tuned kernels only)
      return forward_d48<uint16_t>(st, (const uint16_t *)value, shapes, level_start, (const float *)loc, (const float *)attn, d,
                                   (uint16_t *)out, out_dtype == 0);
    if (out_dtype != 1) return SNIPPER_E_UNSUPPORTED;
    return forward_generic<uint16_t, float>(st, (const uint16_t *)value, shapes, level_start, (const float *)loc,
                                            (const float *)attn, d, (uint16_t *)out);
  }
  if (value_dtype != 0 || (out_dtype != 0 && out_dtype != 1)) return SNIPPER_E_UNSUPPORTED;
  const float *v = (const float *)value, *lo = (const float *)loc, *at = (const float *)attn;
  (void)host_shapes;      // (no forward kernel needs the shapes on the host today; the argument is part of the ABI)
  if (d48_eligible<float>(d, c.policy)) return forward_d48<float>(st, v, shapes, level_start, lo, at, d, (float *)out, out_dtype == 1);
  if (out_dtype == 1) return SNIPPER_E_UNSUPPORTED;       // bf16 rows: D = 48 kernels only (the caller casts)
  return forward_generic<float, float>(st, v, shapes, level_start, lo, at, d, (float *)out);
}

long long snipper_msda_backward_ex_workspace_bytes(const snipper_msda_config *cfg, const int64_t *host_shapes,
                                                   int value_dtype, int N, int S, int M, int D, int L, int Lq, int P) {
  const CoreDims d{N, S, M, D, L, Lq, P};
  if (check_dims(N, S, M, D, L, Lq, P) != SNIPPER_OK || !config_ok(cfg) || (value_dtype != 0 && value_dtype != 1)) return 0;
  const snipper_msda_config c = resolve(cfg);
  if (!owner_shape_ok(d, host_shapes, c.policy)) return 0;
  // (the query does not know the grad_out row type: enough for either grad_value-side kernel's tiling)
  PatchPlan plan;
  long long need = 0;
  for (int mf = 0; mf < 2; ++mf) {
    if (!make_patch_plan(d, host_shapes, c, mf != 0, &plan)) return 0;
    need = std::max(need, patch_workspace_bytes((long long)N * M, d, plan));
  }
  return need;
}

int snipper_msda_backward_ex(void *stream, const snipper_msda_config *cfg, const int64_t *host_shapes, void *workspace,
                             long long workspace_bytes, const void *grad_out, int grad_out_dtype, const void *value,
                             int value_dtype, const int64_t *shapes, const int64_t *level_start, const void *loc,
                             const void *attn, int N, int S, int M, int D, int L, int Lq, int P, void *grad_value,
                             void *grad_loc, void *grad_attn) {
  if (!grad_out || !value || !shapes || !level_start || !loc || !attn || !grad_value || !grad_loc || !grad_attn)
    return SNIPPER_E_NULL;
  if (int rc = check_dims(N, S, M, D, L, Lq, P)) return rc;
  if (!config_ok(cfg)) return SNIPPER_E_SHAPE;
  const snipper_msda_config c = resolve(cfg);
  const CoreDims d{N, S, M, D, L, Lq, P, c.value_layout};
  hipStream_t st = (hipStream_t)stream;
  if (d.head_major && (value_dtype == 2 || !d48_eligible<float>(d, c.policy))) return SNIPPER_E_UNSUPPORTED;
  if (value_dtype == 2) {
    if (grad_out_dtype != 2) return SNIPPER_E_UNSUPPORTED;
    if (int rc = zero_grad_value(st, (double *)grad_value, d)) return rc;
    return backward_generic<double, double, double>(st, (const double *)grad_out, (const double *)value, shapes, level_start,
                                                    (const double *)loc, (const double *)attn, d, (double *)grad_value,
                                                    (double *)grad_loc, (double *)grad_attn);
  }
  if (value_dtype == 1) {        // bf16 value / grad_out, gradients accumulate in float32
    if (grad_out_dtype != 1) return SNIPPER_E_UNSUPPORTED;
    if (workspace && owner_shape_ok(d, host_shapes, c.policy)) {       // encoder shape: owner-computes (marks + sorted taps)
      PatchPlan plan;
      const bool mf = patch_uses_mfma(d, c, 1);
      if (make_patch_plan(d, host_shapes, c, mf, &plan) && workspace_bytes >= patch_workspace_bytes((long long)N * M, d, plan))
        return backward_d48_patch<uint16_t>(st, grad_out, (const uint16_t *)value, (const float *)loc, (const float *)attn, d,
                                            plan, workspace, (float *)grad_value, (float *)grad_loc, (float *)grad_attn, 1, mf);
    }
    if (d.head_major) return SNIPPER_E_UNSUPPORTED;       // (bf16 value without the owner path: the reference layout only)
    if (int rc = zero_grad_value(st, (float *)grad_value, d)) return rc;      // (the atomic kernels accumulate)
    if (d48_eligible<float>(d, c.policy) && (long long)N * S * M * D * 4 < (1LL << 31)) {      // (float32 grad_value offsets)
      // the decoder's shape (Lq = 60 queries) on a bf16 value: the tuned atomic kernel reading bf16 tap rows and bf16 rows
      const float *go = (const float *)grad_out;
      return d.D == kD48 ? backward_d48_f32_t<kD48, uint16_t>(st, go, (const uint16_t *)value, shapes, level_start, (const float *)loc,
                                                               (const float *)attn, d, (float *)grad_value, (float *)grad_loc, (float *)grad_attn, 1)
                         : backward_d48_f32_t<24, uint16_t>(st, go, (const uint16_t *)value, shapes, level_start, (const float *)loc,
                                                             (const float *)attn, d, (float *)grad_value, (float *)grad_loc, (float *)grad_attn, 1);
    }
    return backward_generic<uint16_t, float, float>(st, (const uint16_t *)grad_out, (const uint16_t *)value, shapes,
                                                    level_start, (const float *)loc, (const float *)attn, d,
                                                    (float *)grad_value, (float *)grad_loc, (float *)grad_attn);
  }
  if (value_dtype != 0 || (grad_out_dtype != 0 && grad_out_dtype != 1)) return SNIPPER_E_UNSUPPORTED;
  const float *v = (const float *)value, *lo = (const float *)loc, *at = (const float *)attn;
  float *gv = (float *)grad_value, *gl = (float *)grad_loc, *ga = (float *)grad_attn;
  const int go_bf16 = grad_out_dtype == 1;
  if (go_bf16 && !d48_eligible<float>(d, c.policy)) return SNIPPER_E_UNSUPPORTED;   // bf16 rows: D = 48 kernels only
  if (workspace && owner_shape_ok(d, host_shapes, c.policy)) {
    PatchPlan plan;
    const bool mf = patch_uses_mfma(d, c, go_bf16);
    if (make_patch_plan(d, host_shapes, c, mf, &plan) && workspace_bytes >= patch_workspace_bytes((long long)N * M, d, plan))
      return backward_d48_patch<float>(st, grad_out, v, lo, at, d, plan, workspace, gv, gl, ga, go_bf16, mf);
  }
  if (int rc = zero_grad_value(st, gv, d)) return rc;                         // (the atomic kernels accumulate)
  if (d48_eligible<float>(d, c.policy))
    return backward_d48_f32(st, (const float *)grad_out, v, shapes, level_start, lo, at, d, gv, gl, ga, go_bf16);
  return backward_generic<float, float, float>(st, (const float *)grad_out, v, shapes, level_start, lo, at, d, gv, gl, ga);
}

// grad_value of a bf16 value for FEW queries (the decoder's cross attention) without atomics, float32 buffer or cast:
// csrc/msda_d48_sparse.cuh.  grad_value [N][S][M][48] bf16 (fully written: zeroed here, touched pixels stored once),
// grad_loc / grad_attn float32 from the tuned atomic kernel run without its atomics.  SNIPPER_E_UNSUPPORTED when the shape is
// not this one (the caller then takes snipper_msda_backward_ex).
namespace {
int backward_sparse(void *stream, const void *grad_out, int go_f32, const uint16_t *value, const int64_t *shapes,
                    const int64_t *level_start, const float *loc, const float *attn, int N, int S, int M, int D,
                    int L, int Lq, int P, uint16_t *grad_value, float *grad_loc, float *grad_attn) {
  if (!grad_out || !value || !shapes || !level_start || !loc || !attn || !grad_value || !grad_loc || !grad_attn) return SNIPPER_E_NULL;
  if (int rc = check_dims(N, S, M, D, L, Lq, P)) return rc;
  const CoreDims d{N, S, M, D, L, Lq, P, 0};
  static const bool on = [] { const char *e = getenv("SNIPPER_MSDA_SPARSE"); return !(e && e[0] == '0'); }();
  if (!on || D != kSpD || Lq > kSpMaxLq || Lq * P * 4 > kSpTaps || L > kMaxLevelsFast || S >= (1 << 22) ||
      !d48_eligible<float>(d, 0) || (long long)N * S * M * D * 2 >= (1LL << 31) || ((uintptr_t)grad_value & 15) ||
      ((uintptr_t)grad_out & 15))
    return SNIPPER_E_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  if (const hipError_t e = hipMemsetAsync(grad_value, 0, (size_t)N * S * M * D * 2, st); e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(msda_bwd_d48_sparse_gv_kernel, dim3((unsigned)(N * M * L)), dim3(kSpTaps), 0, st, (const uint16_t *)grad_out, shapes,
                     level_start, loc, attn, d, grad_value, go_f32);
  if (int rc = launch_status()) return rc;
  const int rc = backward_d48_f32_t<kD48, uint16_t>(st, (const float *)grad_out, value, shapes, level_start, loc, attn, d, nullptr,
                                                    grad_loc, grad_attn, go_f32 ? 0 : 1);
  if (rc == SNIPPER_OK) g_last_variant = "d48_sparse";
  return rc;
}
}  // namespace

int snipper_msda_backward_sparse_bf16(void *stream, const uint16_t *grad_out, const uint16_t *value, const int64_t *shapes,
                                      const int64_t *level_start, const float *loc, const float *attn, int N, int S, int M, int D,
                                      int L, int Lq, int P, uint16_t *grad_value, float *grad_loc, float *grad_attn) {
  return backward_sparse(stream, grad_out, 0, value, shapes, level_start, loc, attn, N, S, M, D, L, Lq, P, grad_value, grad_loc, grad_attn);
}
// the same with FLOAT32 grad_out rows (a float32 consumer of the sampled rows -- the decoder's output projection -- hands its
// data gradient over as it is: no bf16 cast launch in front of this call)
int snipper_msda_backward_sparse_f32rows(void *stream, const float *grad_out, const uint16_t *value, const int64_t *shapes,
                                         const int64_t *level_start, const float *loc, const float *attn, int N, int S, int M, int D,
                                         int L, int Lq, int P, uint16_t *grad_value, float *grad_loc, float *grad_attn) {
  return backward_sparse(stream, grad_out, 1, value, shapes, level_start, loc, attn, N, S, M, D, L, Lq, P, grad_value, grad_loc, grad_attn);
}

// ---- the reference launchers' one-for-one replacements (default configuration, level shapes on the device only) ----
int snipper_msda_forward_f32(void *stream, const float *value, const int64_t *shapes,
                             const int64_t *level_start, const float *loc, const float *attn,
                             int N, int S, int M, int D, int L, int Lq, int P, float *out) {
  return snipper_msda_forward_ex(stream, nullptr, nullptr, value, 0, shapes, level_start, loc, attn, N, S, M, D, L, Lq, P, out, 0);
}
int snipper_msda_forward_f64(void *stream, const double *value, const int64_t *shapes,
                             const int64_t *level_start, const double *loc, const double *attn,
                             int N, int S, int M, int D, int L, int Lq, int P, double *out) {
  return snipper_msda_forward_ex(stream, nullptr, nullptr, value, 2, shapes, level_start, loc, attn, N, S, M, D, L, Lq, P, out, 2);
}
int snipper_msda_forward_bf16(void *stream, const uint16_t *value, const int64_t *shapes,
                              const int64_t *level_start, const float *loc, const float *attn,
                              int N, int S, int M, int D, int L, int Lq, int P, uint16_t *out) {
  return snipper_msda_forward_ex(stream, nullptr, nullptr, value, 1, shapes, level_start, loc, attn, N, S, M, D, L, Lq, P, out, 1);
}
int snipper_msda_backward_f32(void *stream, const float *grad_out, const float *value,
                              const int64_t *shapes, const int64_t *level_start,
                              const float *loc, const float *attn,
                              int N, int S, int M, int D, int L, int Lq, int P,
                              float *grad_value, float *grad_loc, float *grad_attn) {
  return snipper_msda_backward_ex(stream, nullptr, nullptr, nullptr, 0, grad_out, 0, value, 0, shapes, level_start, loc, attn,
                                  N, S, M, D, L, Lq, P, grad_value, grad_loc, grad_attn);
}
int snipper_msda_backward_f64(void *stream, const double *grad_out, const double *value,
                              const int64_t *shapes, const int64_t *level_start,
                              const double *loc, const double *attn,
                              int N, int S, int M, int D, int L, int Lq, int P,
                              double *grad_value, double *grad_loc, double *grad_attn) {
  return snipper_msda_backward_ex(stream, nullptr, nullptr, nullptr, 0, grad_out, 2, value, 2, shapes, level_start, loc, attn,
                                  N, S, M, D, L, Lq, P, grad_value, grad_loc, grad_attn);
}
int snipper_msda_backward_bf16(void *stream, const uint16_t *grad_out, const uint16_t *value,
                               const int64_t *shapes, const int64_t *level_start,
                               const float *loc, const float *attn,
                               int N, int S, int M, int D, int L, int Lq, int P,
                               float *grad_value, float *grad_loc, float *grad_attn) {
  return snipper_msda_backward_ex(stream, nullptr, nullptr, nullptr, 0, grad_out, 1, value, 1, shapes, level_start, loc, attn,
                                  N, S, M, D, L, Lq, P, grad_value, grad_loc, grad_attn);
}

namespace {
// compute units of the CURRENT device (one persistent workgroup each in the weight-stationary kernel): read once per device --
// a process may drive several GPUs (a cache keyed on nothing would hand the second one the first one's answer)
constexpr int kMaxDevices = 64;
int current_device_index() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
  return dev;
}
int device_cu_count() {
  static std::atomic<int> cache[kMaxDevices];          // zero-initialised: 0 = not read yet
  const int dev = current_device_index();
  if (dev < kMaxDevices) {
    const int got = cache[dev].load(std::memory_order_relaxed);
    if (got > 0) return got;
  }
  int cus = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
  if (dev < kMaxDevices) cache[dev].store(cus, std::memory_order_relaxed);
  return cus;
}
// SNIPPER_GEMM_WRES=0 in the environment at load time keeps every product on the tile kernels (A/B measurements)
int wres_debug() {       // SNIPPER_WRES_DEBUG: timing ablations of the weight-stationary / ring kernels (WRONG results), read once;
                         // like the owner-computes ablations they need SNIPPER_MSDA_ALLOW_DEBUG=1 in the same environment
  static const int v = [] {
    const char *e = getenv("SNIPPER_WRES_DEBUG"), *a = getenv("SNIPPER_MSDA_ALLOW_DEBUG");
    return (e && a && a[0] == '1') ? atoi(e) : 0;
  }();
  return v;
}
// diagnostic builds only (-DWRES_STAMPS): SNIPPER_WRES_STAMPS=<device address, hex> of a 2 x 320 x 8-byte buffer that workgroup 0
// fills with s_memtime stamps (tools/wres_stamps.py allocates it and passes the address through the environment)
unsigned long long *wres_stamp_buffer() {
  static unsigned long long *const p = [] { const char *e = getenv("SNIPPER_WRES_STAMPS"); return e ? (unsigned long long *)strtoull(e, nullptr, 16) : nullptr; }();
  return p;
}
bool wres_enabled() {
  static const bool on = [] { const char *e = getenv("SNIPPER_GEMM_WRES"); return !(e && e[0] == '0'); }();
  return on;
}
}  // namespace

int snipper_linear_wres_supported(int M, int N, int K) {
  return (K == 384 || K == 288) && N >= 16 && N % 8 == 0 && M >= 8192 && (long long)M * (N > K ? N : K) < (1LL << 30) ? 1 : 0;
}

int snipper_linear_wres_bf16(void *stream, const uint16_t *X, long long ldx, const uint16_t *W, long long ldw,
                             const float *bias, const uint16_t *A, long long lda, float gate_scale, uint16_t *Y,
                             long long ldy, int M, int N, int K, int relu, float dropout_p, uint64_t seed) {
  if (!X || !W || !Y) return SNIPPER_E_NULL;
  if (M <= 0 || N <= 0 || K <= 0 || ldx % 8 || ldw % 8 || ldy % 8 || ldx < K || ldw < K || ldy < N || (A && (lda % 8 || lda < N)) ||
      !(dropout_p >= 0.f && dropout_p < 1.f) || (dropout_p > 0.f && (long long)M * N >= (1LL << 32)))
    return SNIPPER_E_SHAPE;
  if (((uintptr_t)X | (uintptr_t)W | (uintptr_t)Y | (uintptr_t)A) & 15) return SNIPPER_E_SHAPE;
  if (!snipper_linear_wres_supported(M, N, K)) return SNIPPER_E_UNSUPPORTED;
  // 32-bit byte offsets inside a 32-row chunk and 31-bit descriptor sizes
  if (32 * ldx * 2 >= (1LL << 31) || 32 * ldy * 2 >= (1LL << 31) || (long long)N * ldw * 2 >= (1LL << 31) || (A && 32 * lda * 2 >= (1LL << 31)))
    return SNIPPER_E_SHAPE;
  const int ncb = (N + kWrCols - 1) / kWrCols;
  const int chunks = (M + kWrRows - 1) / kWrRows;
  int n_series = 8 * (device_cu_count() / (8 * ncb));
  if (n_series < 8) n_series = 8;
  if (n_series > ((chunks + 7) / 8) * 8) n_series = ((chunks + 7) / 8) * 8;
  const WresArgs g{X, ldx, W, ldw, bias, A, lda, gate_scale, Y, ldy, M, N, K, relu ? 1 : 0, dropout_p,
                   (uint32_t)seed, (uint32_t)(seed >> 32), n_series, wres_stamp_buffer(), wres_debug()};
  const dim3 grid((unsigned)(n_series * ncb)), block(kWrThreads);
  hipStream_t st = (hipStream_t)stream;
  const bool act = relu || dropout_p > 0.f;
#define WRES_LAUNCH(KS_, GATE_, ACT_) hipLaunchKernelGGL((wres_gemm_kernel<KS_, GATE_, ACT_>), grid, block, 0, st, g)
  if (K == 384) {
    if (A) { if (act) WRES_LAUNCH(12, true, true); else WRES_LAUNCH(12, true, false); }
    else { if (act) WRES_LAUNCH(12, false, true); else WRES_LAUNCH(12, false, false); }
  } else {
    if (A) { if (act) WRES_LAUNCH(9, true, true); else WRES_LAUNCH(9, true, false); }
    else { if (act) WRES_LAUNCH(9, false, true); else WRES_LAUNCH(9, false, false); }
  }
#undef WRES_LAUNCH
  return launch_status();
}

// ---- gradient clipping + AdamW on a flat buffer (csrc/adamw_flat.cuh) ----
int snipper_gradnorm_partials_f32(void *stream, const float *grad, long long n, float *partials, int nparts) {
  if (!grad || !partials) return SNIPPER_E_NULL;
  if (n <= 0 || n % 4 || nparts <= 0 || nparts > kAdamMaxParts || ((uintptr_t)grad & 15)) return SNIPPER_E_SHAPE;
  hipLaunchKernelGGL(gradnorm_partials_kernel, dim3((unsigned)nparts), dim3(kAdamThreads), 0, (hipStream_t)stream, grad, n / 4, partials);
  return launch_status();
}

int snipper_adamw_clip_f32(void *stream, float *param, const float *grad, float *exp_avg, float *exp_avg_sq, long long n,
                           const long long *seg_begin, const long long *seg_end, const float *seg_lr, const float *seg_wd, int nseg,
                           float beta1, float beta2, float eps, long long step, const float *partials, int nparts,
                           float max_norm, float *norm_out) {
  if (!param || !grad || !exp_avg || !exp_avg_sq || !seg_begin || !seg_end || !seg_lr || !seg_wd) return SNIPPER_E_NULL;
  if (n <= 0 || n % 4 || nseg <= 0 || nseg > kAdamMaxSeg || step <= 0 || !(beta1 >= 0.f && beta1 < 1.f) || !(beta2 >= 0.f && beta2 < 1.f) ||
      !(eps > 0.f) || (partials && (nparts <= 0 || nparts > kAdamMaxParts)))
    return SNIPPER_E_SHAPE;
  if (((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) return SNIPPER_E_SHAPE;
  AdamArgs a{};
  a.p = param; a.g = grad; a.m = exp_avg; a.v = exp_avg_sq; a.n = n; a.nseg = nseg;
  for (int k = 0; k < nseg; ++k) {
    if (seg_begin[k] < 0 || seg_end[k] > n || seg_begin[k] > seg_end[k] || seg_begin[k] % 4 || seg_end[k] % 4) return SNIPPER_E_SHAPE;
    a.seg[k] = AdamSeg{seg_begin[k], seg_end[k], seg_lr[k], seg_wd[k]};
  }
  a.beta1 = beta1; a.beta2 = beta2; a.eps = eps;
  a.inv_bc1 = (float)(1.0 / (1.0 - std::pow((double)beta1, (double)step)));
  a.inv_bc2_sqrt = (float)(1.0 / std::sqrt(1.0 - std::pow((double)beta2, (double)step)));
  a.partials = partials; a.nparts = partials ? nparts : 0; a.max_norm = max_norm; a.norm_out = norm_out;
  const long long want = (n / 4 + kAdamThreads - 1) / kAdamThreads;
  const unsigned grid = (unsigned)std::min<long long>(want, 8LL * device_cu_count());
  hipLaunchKernelGGL(adamw_clip_kernel, dim3(grid), dim3(kAdamThreads), 0, (hipStream_t)stream, a);
  return launch_status();
}

namespace {
// measurement aid: a float4 device-to-device copy, the form MI355X_MICROARCH.md quotes the achievable HBM rate for
__global__ __launch_bounds__(256) void hbm_copy_probe_kernel(const gemm_u32x4 *__restrict__ src, gemm_u32x4 *__restrict__ dst, long long n16) {
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) dst[i] = src[i];
}
}  // namespace

int snipper_hbm_copy_probe(void *stream, const void *src, void *dst, long long bytes) {
  if (!src || !dst) return SNIPPER_E_NULL;
  if (bytes <= 0 || bytes % 16 || (((uintptr_t)src | (uintptr_t)dst) & 15)) return SNIPPER_E_SHAPE;
  const long long n16 = bytes / 16;
  const long long want = (n16 + 255) / 256;
  const unsigned grid = (unsigned)(want < 8192 ? want : 8192);          // 32 workgroups per CU, grid-stride
  hipLaunchKernelGGL(hbm_copy_probe_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const gemm_u32x4 *)src, (gemm_u32x4 *)dst, n16);
  return launch_status();
}

int snipper_transpose_batch_bf16(void *stream, int count, const void *const *src, void *const *dst, const int *rows,
                                 const int *cols, const long long *ld_src, const long long *ld_dst) {
  if (count <= 0) return SNIPPER_OK;
  if (!src || !dst || !rows || !cols || !ld_src || !ld_dst) return SNIPPER_E_NULL;
  if (count > kTrMaxItems) return SNIPPER_E_SHAPE;
  TransposeBatch b{};
  int tiles = 0;
  for (int i = 0; i < count; ++i) {
    if (!src[i] || !dst[i]) return SNIPPER_E_NULL;
    if (rows[i] <= 0 || cols[i] <= 0 || ld_src[i] < cols[i] || ld_dst[i] < rows[i]) return SNIPPER_E_SHAPE;
    const int tr = (rows[i] + 63) / 64, tc = (cols[i] + 63) / 64;
    b.it[i] = TransposeItem{(const uint16_t *)src[i], (uint16_t *)dst[i], rows[i], cols[i], (int)ld_src[i], (int)ld_dst[i], tiles, tc};
    tiles += tr * tc;
  }
  b.count = count;
  hipLaunchKernelGGL(transpose_batch_bf16_kernel, dim3((unsigned)tiles), dim3(256), 0, (hipStream_t)stream, b);
  return launch_status();
}

int snipper_linear_bf16(void *stream, const uint16_t *X, long long ldx, const uint16_t *W,
                        const float *bias, const uint16_t *R, long long ldr, uint16_t *Y, long long ldy,
                        int M, int N, int K, int relu, float dropout_p, uint64_t seed) {
  if (!X || !W || !Y) return SNIPPER_E_NULL;
  if (!R && wres_enabled() && snipper_linear_wres_supported(M, N, K) && ldx % 8 == 0 && ldy % 8 == 0 &&
      !(((uintptr_t)X | (uintptr_t)W | (uintptr_t)Y) & 15))
    return snipper_linear_wres_bf16(stream, X, ldx, W, K, bias, nullptr, 0, 1.f, Y, ldy, M, N, K, relu, dropout_p, seed);
  if (M <= 0 || N <= 0 || K <= 0 || K % kGemmBK || N % 4 || ldx % 8 || ldy % 4 || ldx < K || ldy < N ||
      (R && (ldr % 4 || ldr < N)) || !(dropout_p >= 0.f && dropout_p < 1.f) ||
      (dropout_p > 0.f && (long long)M * N >= (1LL << 32)))
    return SNIPPER_E_SHAPE;
  const GemmArgs g{X, ldx, W, bias, R, ldr, Y, ldy, M, N, K, dropout_p, (uint32_t)seed, (uint32_t)(seed >> 32)};
  if (gemm_use_n64(M, N, K)) {
    const long long tiles_m = (M + 127) / 128, tiles_n = (N + 63) / 64;
    const dim3 grid64((unsigned)(tiles_n * 8 * ((tiles_m + 7) / 8)));
    if (relu) hipLaunchKernelGGL(linear_bf16_n64_kernel<true>, grid64, dim3(kGemmThreads), 0, (hipStream_t)stream, g);
    else hipLaunchKernelGGL(linear_bf16_n64_kernel<false>, grid64, dim3(kGemmThreads), 0, (hipStream_t)stream, g);
    return launch_status();
  }
  const dim3 grid(gemm_grid_size(M, N));
  if (relu)
    hipLaunchKernelGGL(linear_bf16_kernel<true>, grid, dim3(kGemmThreads), 0, (hipStream_t)stream, g);
  else
    hipLaunchKernelGGL(linear_bf16_kernel<false>, grid, dim3(kGemmThreads), 0, (hipStream_t)stream, g);
  return launch_status();
}

int snipper_linear_nn_bf16(void *stream, const uint16_t *X, long long ldx, const uint16_t *W, long long ldw,
                           const uint16_t *R, long long ldr, const uint16_t *A, long long lda, float gate_scale,
                           uint16_t *Y, long long ldy, int M, int N, int K) {
  if (!X || !W || !Y) return SNIPPER_E_NULL;
  if (M <= 0 || N <= 0 || K <= 0 || K % kGemmBK || N % 8 || ldx % 8 || ldw % 8 || ldy % 4 || ldx < K || ldw < N || ldy < N ||
      (R && (ldr % 4 || ldr < N)) || (A && (lda % 4 || lda < N)))
    return SNIPPER_E_SHAPE;
  if (((uintptr_t)X | (uintptr_t)W) & 15) return SNIPPER_E_SHAPE;
  const GemmNNArgs g{X, ldx, W, ldw, Y, ldy, M, N, K, R, ldr, A, lda, gate_scale};
  const dim3 grid(gemm_grid_size(M, N));
  hipLaunchKernelGGL(linear_bf16_nn_kernel<128>, grid, dim3(kGemmThreads), 0, (hipStream_t)stream, g);
  return launch_status();
}

namespace {
int small_attn_check(const void *q, long long q_ld, const void *k, long long k_ld, const void *v, long long v_ld,
                     const void *o, long long o_ld, const void *P, int bs, int H, int L, int hd, float p) {
  if (!q || !k || !v || !o || !P) return SNIPPER_E_NULL;
  if (bs <= 0 || H <= 0 || L <= 0 || L > kSaMaxL2 || (hd != 48 && hd != 32) || (q_ld | k_ld | v_ld | o_ld) % 4 ||
      !(p >= 0.f && p < 1.f) || (long long)bs * H * L * L >= (1LL << 32))
    return SNIPPER_E_SHAPE;
  if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o) & 15) return SNIPPER_E_SHAPE;
  return SNIPPER_OK;
}
}  // namespace

int snipper_heatmap_blur_f32(void *stream, const float *in, float *out, int n_images, int H, int W, int ksize,
                             const float *weights, float clamp_max) {
  if (!in || !out || !weights) return SNIPPER_E_NULL;
  if (n_images <= 0 || H <= 0 || W <= 0 || ksize <= 0 || ksize > kBlurMaxTaps || (ksize & 1) == 0 || ksize / 2 >= H ||
      ksize / 2 >= W || in == out)
    return SNIPPER_E_SHAPE;
  BlurArgs a{};
  a.in = in; a.out = out; a.n_images = n_images; a.H = H; a.W = W; a.k = ksize; a.clamp_max = clamp_max;
  for (int i = 0; i < ksize; ++i) a.w[i] = weights[i];              // (host array: the taps travel in the kernel argument)
  const long long total = (long long)n_images * H * W;
  hipLaunchKernelGGL(heatmap_blur_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  return launch_status();
}

int snipper_level_pos_bf16(void *stream, const float *const *pos, const int *hw, int levels, const float *level_embed, int bt,
                           int C, uint16_t *out) {
  if (!pos || !hw || !level_embed || !out) return SNIPPER_E_NULL;
  if (levels <= 0 || levels > kLpMaxLevels || bt <= 0 || C <= 0 || C % 8 || (((uintptr_t)level_embed | (uintptr_t)out) & 15))
    return SNIPPER_E_SHAPE;
  LevelPosArgs a{};
  a.level_embed = level_embed; a.out = out; a.levels = levels; a.bt = bt; a.C = C;
  int S = 0;
  for (int l = 0; l < levels; ++l) {
    if (!pos[l] || hw[l] <= 0 || ((uintptr_t)pos[l] & 15)) return SNIPPER_E_SHAPE;
    a.pos[l] = pos[l]; a.hw[l] = hw[l]; a.start[l] = S;
    S += hw[l];
  }
  a.S = S;
  const long long total = (long long)bt * S * (C / 8);
  if (total >= (1LL << 40)) return SNIPPER_E_SHAPE;
  hipLaunchKernelGGL(level_pos_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  return launch_status();
}

int snipper_cast_scale_table_bf16(void *stream, const void *items, const int *block_end, int n_items, int n_blocks) {
  if (!items || !block_end) return SNIPPER_E_NULL;
  if (n_items <= 0 || n_blocks <= 0 || (((uintptr_t)items | (uintptr_t)block_end) & 7) || sizeof(CastItem) != 40) return SNIPPER_E_SHAPE;
  hipLaunchKernelGGL(cast_scale_table_kernel, dim3((unsigned)n_blocks), dim3(256), 0, (hipStream_t)stream,
                     (const CastItem *)items, block_end, n_items);
  return launch_status();
}

int snipper_heatmap_scatter_f32(void *stream, const float *kpts, const long long *sample, int n_person, int Tk, int T, int K,
                                int levels, const int *h, const int *w, const long long *base, float *out) {
  if (!kpts || !sample || !h || !w || !base || !out) return SNIPPER_E_NULL;
  if (n_person < 0 || Tk <= 0 || T <= 0 || T > Tk || K <= 0 || levels <= 0 || levels > kHlMaxLevels) return SNIPPER_E_SHAPE;
  if (n_person == 0) return SNIPPER_OK;
  HeatmapScatterArgs a{};
  a.kpts = kpts; a.sample = sample; a.out = out; a.levels = levels; a.n_person = n_person; a.Tk = Tk; a.T = T; a.K = K;
  for (int l = 0; l < levels; ++l) {
    if (h[l] <= 0 || w[l] <= 0 || base[l] < 0) return SNIPPER_E_SHAPE;
    a.h[l] = h[l]; a.w[l] = w[l]; a.base[l] = base[l];
  }
  const long long total = (long long)levels * n_person * T * K;
  hipLaunchKernelGGL(heatmap_scatter_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  return launch_status();
}

namespace {
int heatmap_loss_args(HeatmapLossArgs &a, const float *mem, const float *const *tm, const int *hw, const int *start, int levels,
                      int bs, int T, int S, int C, int nhead, int K) {
  if (!mem || !tm || !hw || !start) return SNIPPER_E_NULL;
  if (levels <= 0 || levels > kHlMaxLevels || bs <= 0 || T <= 0 || S <= 0 || C <= 0 || nhead <= 0 || C % nhead || (C / nhead) % 4 ||
      K <= 0 || K > C / nhead || ((uintptr_t)mem & 15) || (long long)bs * T * S * C >= (1LL << 40))
    return SNIPPER_E_SHAPE;
  a = HeatmapLossArgs{};
  a.mem = mem; a.levels = levels; a.bs = bs; a.T = T; a.S = S; a.C = C; a.nhead = nhead; a.D = C / nhead; a.K = K;
  int pos = 0;
  for (int l = 0; l < levels; ++l) {
    if (!tm[l] || hw[l] <= 0 || start[l] != pos) return SNIPPER_E_SHAPE;        // levels tile S in order
    a.tm[l] = tm[l]; a.hw[l] = hw[l]; a.start[l] = start[l];
    pos += hw[l];
  }
  return pos == S ? SNIPPER_OK : SNIPPER_E_SHAPE;
}
}  // namespace

int snipper_heatmap_loss_forward_f32(void *stream, const float *mem, const float *const *tm, const int *hw, const int *start,
                                     int levels, int bs, int T, int S, int C, int nhead, int K, float *partial, int n_partial) {
  HeatmapLossArgs a;
  if (int rc = heatmap_loss_args(a, mem, tm, hw, start, levels, bs, T, S, C, nhead, K)) return rc;
  if (!partial || n_partial <= 0) return SNIPPER_E_NULL;
  a.partial = partial;
  hipLaunchKernelGGL(heatmap_loss_fwd_kernel, dim3((unsigned)n_partial), dim3(256), 0, (hipStream_t)stream, a);
  return launch_status();
}

int snipper_heatmap_loss_backward_f32(void *stream, const float *mem, const float *const *tm, const int *hw, const int *start,
                                      int levels, int bs, int T, int S, int C, int nhead, int K, const float *gscale, float *gmem) {
  HeatmapLossArgs a;
  if (int rc = heatmap_loss_args(a, mem, tm, hw, start, levels, bs, T, S, C, nhead, K)) return rc;
  if (!gscale || !gmem || ((uintptr_t)gmem & 15)) return SNIPPER_E_NULL;
  a.gscale = gscale; a.gmem = gmem;
  const long long total = (long long)bs * T * S * (C / 4);
  hipLaunchKernelGGL(heatmap_loss_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  return launch_status();
}

int snipper_small_attention_forward_f32(void *stream, const float *q, long long q_ld, long long q_bs, const float *k,
                                        long long k_ld, long long k_bs, const float *v, long long v_ld, long long v_bs,
                                        float *out, long long o_ld, long long o_bs, float *P, int bs, int H, int L, int hd,
                                        float scale, float dropout_p, uint64_t seed) {
  if (int rc = small_attn_check(q, q_ld, k, k_ld, v, v_ld, out, o_ld, P, bs, H, L, hd, dropout_p)) return rc;
  SmallAttnArgs a{};
  a.q = q; a.q_ld = q_ld; a.q_bs = q_bs; a.k = k; a.k_ld = k_ld; a.k_bs = k_bs; a.v = v; a.v_ld = v_ld; a.v_bs = v_bs;
  a.out = out; a.o_ld = o_ld; a.o_bs = o_bs; a.P = P; a.bs = bs; a.H = H; a.L = L; a.scale = scale; a.drop_p = dropout_p;
  a.seed_lo = (uint32_t)seed; a.seed_hi = (uint32_t)(seed >> 32);
  if (L > kSaMaxL) {          // 257 .. 384 queries: one row buffer staged twice, 32 rows per workgroup (csrc/small_attention.cuh)
    const dim3 grid2((unsigned)(bs * H * ((L + kSaRows2 - 1) / kSaRows2)));
    if (hd == 48) hipLaunchKernelGGL(small_attn_fwd_seq_kernel<48>, grid2, dim3(kSaThreads2), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(small_attn_fwd_seq_kernel<32>, grid2, dim3(kSaThreads2), 0, (hipStream_t)stream, a);
    return launch_status();
  }
  const dim3 grid((unsigned)(bs * H * ((L + kSaRows - 1) / kSaRows)));
  if (hd == 48) hipLaunchKernelGGL(small_attn_fwd_kernel<48>, grid, dim3(kSaThreads), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(small_attn_fwd_kernel<32>, grid, dim3(kSaThreads), 0, (hipStream_t)stream, a);
  return launch_status();
}

int snipper_small_attention_backward_f32(void *stream, const float *q, long long q_ld, long long q_bs, const float *k,
                                         long long k_ld, long long k_bs, const float *v, long long v_ld, long long v_bs,
                                         const float *out, long long o_ld, long long o_bs, const float *P,
                                         const float *dout, long long do_ld, long long do_bs, float *dq, long long dq_ld,
                                         long long dq_bs, float *dk, long long dk_ld, long long dk_bs, float *dv,
                                         long long dv_ld, long long dv_bs, int bs, int H, int L, int hd, float scale,
                                         float dropout_p, uint64_t seed) {
  if (int rc = small_attn_check(q, q_ld, k, k_ld, v, v_ld, out, o_ld, P, bs, H, L, hd, dropout_p)) return rc;
  if (!dout || !dq || !dk || !dv) return SNIPPER_E_NULL;
  if (do_ld % 4 || ((uintptr_t)dout & 15)) return SNIPPER_E_SHAPE;
  SmallAttnArgs a{};
  a.q = q; a.q_ld = q_ld; a.q_bs = q_bs; a.k = k; a.k_ld = k_ld; a.k_bs = k_bs; a.v = v; a.v_ld = v_ld; a.v_bs = v_bs;
  a.out = const_cast<float *>(out); a.o_ld = o_ld; a.o_bs = o_bs; a.P = const_cast<float *>(P);
  a.dout = dout; a.do_ld = do_ld; a.do_bs = do_bs; a.dq = dq; a.dq_ld = dq_ld; a.dq_bs = dq_bs;
  a.dk = dk; a.dk_ld = dk_ld; a.dk_bs = dk_bs; a.dv = dv; a.dv_ld = dv_ld; a.dv_bs = dv_bs;
  a.bs = bs; a.H = H; a.L = L; a.scale = scale; a.drop_p = dropout_p;
  a.seed_lo = (uint32_t)seed; a.seed_hi = (uint32_t)(seed >> 32);
  // 16 rows per workgroup while that is one residency round (one 140-KB workgroup per CU), 32 rows (512 threads) beyond: at the
  // decoder's L = 240 the 480 sixteen-row workgroups were two rounds (csrc/small_attention.cuh)
  if (L > kSaMaxL) {
    const dim3 grid2((unsigned)(bs * H * 2 * ((L + kSaRows2 - 1) / kSaRows2)));
    if (hd == 48) hipLaunchKernelGGL(small_attn_bwd_seq_kernel<48>, grid2, dim3(kSaThreads2), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(small_attn_bwd_seq_kernel<32>, grid2, dim3(kSaThreads2), 0, (hipStream_t)stream, a);
    return launch_status();
  }
  static const int forced_rows = [] { const char *e = getenv("SNIPPER_SMALL_ATTN_ROWS"); return e ? atoi(e) : 0; }();      // (A/B aid)
  const int wg16 = bs * H * 2 * ((L + 15) / 16);
  const int rows = forced_rows == 16 || forced_rows == 32 ? forced_rows : (wg16 > device_cu_count() ? 32 : 16);
  const dim3 grid((unsigned)(bs * H * 2 * ((L + rows - 1) / rows)));
  if (hd == 48 && rows == 32) hipLaunchKernelGGL((small_attn_bwd_kernel<48, 32>), grid, dim3(512), 0, (hipStream_t)stream, a);
  else if (hd == 48) hipLaunchKernelGGL((small_attn_bwd_kernel<48, 16>), grid, dim3(256), 0, (hipStream_t)stream, a);
  else if (rows == 32) hipLaunchKernelGGL((small_attn_bwd_kernel<32, 32>), grid, dim3(512), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL((small_attn_bwd_kernel<32, 16>), grid, dim3(256), 0, (hipStream_t)stream, a);
  return launch_status();
}

int snipper_small_gemm_batch_f32(void *stream, const snipper_small_gemm *problems, int count) {
  if (!problems) return SNIPPER_E_NULL;
  if (count <= 0 || count > kSgMaxProblems) return SNIPPER_E_SHAPE;
  SmallGemmBatch batch{};
  int tiles = 0;
  for (int i = 0; i < count; ++i) {
    const snipper_small_gemm &q = problems[i];
    if (!q.A || !q.B || (!q.out && !q.colsum)) return SNIPPER_E_NULL;
    const int a_cols = q.a_transposed ? q.I : q.R, b_cols = q.b_transposed ? q.R : q.J;
    if (q.I <= 0 || q.J <= 0 || q.R <= 0 || a_cols % 4 || b_cols % 4 || q.lda % 4 || q.ldb % 4 || q.lda < a_cols ||
        q.ldb < b_cols || (q.out && q.ldo < q.J))
      return SNIPPER_E_SHAPE;
    if (((uintptr_t)q.A | (uintptr_t)q.B) & 15) return SNIPPER_E_SHAPE;
    if (q.B2 && (q.b_transposed || q.r_split <= 0 || q.r_split >= q.R || q.r_split % kSgTile || q.ldb2 % 4 || q.ldb2 < q.J ||
                 ((uintptr_t)q.B2 & 15)))
      return SNIPPER_E_SHAPE;
    if (!(q.dropout_p >= 0.f && q.dropout_p < 1.f) || (q.gate && q.ldgate < q.J) ||
        (q.dropout_p > 0.f && (long long)q.I * q.J >= (1LL << 32)))
      return SNIPPER_E_SHAPE;
    tiles += ((q.I + kSgTile - 1) / kSgTile) * ((q.J + kSgTile - 1) / kSgTile);
    batch.p[i] = SmallGemmProblem{q.A, q.lda, q.B, q.ldb, q.out, q.ldo, q.bias, q.colsum, q.I, q.J, q.R,
                                  q.a_transposed ? 1 : 0, q.b_transposed ? 1 : 0, tiles,
                                  q.B2, q.ldb2, q.r_split, q.relu ? 1 : 0, q.dropout_p, (uint32_t)q.seed,
                                  (uint32_t)(q.seed >> 32), q.gate, q.ldgate, q.gate_scale};
  }
  batch.count = count;
  hipLaunchKernelGGL(small_gemm_batch_f32_kernel, dim3(tiles), dim3(kSgThreads), 0, (hipStream_t)stream, batch);
  return launch_status();
}

int snipper_small_linear_forward_f32(void *stream, const float *X, long long ldx, const float *W, long long ldw,
                                     const float *bias, int M, int N, int K, float *Y, long long ldy) {
  snipper_small_gemm q{};
  q.A = X; q.lda = ldx; q.B = W; q.ldb = ldw; q.b_transposed = 1; q.out = Y; q.ldo = ldy; q.bias = bias;     // X . W^T + b
  q.I = M; q.J = N; q.R = K;
  return snipper_small_gemm_batch_f32(stream, &q, 1);
}

int snipper_small_linear_backward_f32(void *stream, const float *G, long long ldg, const float *X, long long ldx,
                                      const float *W, long long ldw, int M, int N, int K, float *dX, long long lddx,
                                      float *dW, long long lddw, float *db) {
  if (!G || (dX && !W) || ((dW || db) && !X)) return SNIPPER_E_NULL;
  snipper_small_gemm q[2] = {};
  int n = 0;
  if (dX) {                                                                                                // G . W
    q[n].A = G; q[n].lda = ldg; q[n].B = W; q[n].ldb = ldw; q[n].out = dX; q[n].ldo = lddx;
    q[n].I = M; q[n].J = K; q[n].R = N; ++n;
  }
  if (dW || db) {                                                                                          // G^T . X, column sums
    q[n].A = G; q[n].lda = ldg; q[n].a_transposed = 1; q[n].B = X; q[n].ldb = ldx; q[n].out = dW; q[n].ldo = lddw;
    q[n].colsum = db; q[n].I = N; q[n].J = K; q[n].R = M; ++n;
  }
  if (n == 0) return SNIPPER_OK;
  return snipper_small_gemm_batch_f32(stream, q, n);
}

int snipper_relu_dropout_backward_bf16(void *stream, const uint16_t *grad_y, const uint16_t *y, uint16_t *grad_pre,
                                       long long n, float dropout_p) {
  if (!grad_y || !y || !grad_pre) return SNIPPER_E_NULL;
  if (n <= 0 || n % 8 || !(dropout_p >= 0.f && dropout_p < 1.f)) return SNIPPER_E_SHAPE;
  if (((uintptr_t)grad_y | (uintptr_t)y | (uintptr_t)grad_pre) & 15) return SNIPPER_E_SHAPE;
  const long long n8 = n / 8;
  hipLaunchKernelGGL(relu_dropout_bwd_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     grad_y, y, grad_pre, n8, 1.f / (1.f - dropout_p));
  return launch_status();
}

namespace {
struct WgradPlan { int tiles_n, tiles_k, S, rows, wk; };
// split the reduction axis so that the grid has about g_wgrad_wgs workgroups (2 per CU); a multiple of 8 row-ranges
// lets the kernel keep the output tiles of one range on one XCD.
// (Round 3: an LDS-DMA ring variant -- one 4-wave workgroup per CU, 96 KB in flight -- was built, parity-green and slower:
//  49 against 40 us at 79 000 x 384 x 384, memory not the limiter; profiles/r03_wgrad_ring_experiment.json.)
WgradPlan wgrad_plan(int M, int N, int Kc) {
  WgradPlan p;
  const int wgs = kWgradWgs;
  p.wk = 128;      // (64-column X tiles, 4 waves per SIMD: measured slower at 512 / 768 / 1024 workgroups, wgrad_bf16.cuh)
  p.tiles_n = (N + kWgTile - 1) / kWgTile;
  p.tiles_k = (Kc + p.wk - 1) / p.wk;
  const int tiles = p.tiles_n * p.tiles_k;
  int s0 = std::max(1, wgs / tiles);
  if (s0 >= 8) s0 = (s0 + 7) / 8 * 8;
  // the register-prefetch kernel runs three workgroups per CU: a grid between 2 and 3 per CU (24 tiles x 24 ranges = 576)
  // leaves a quarter of the CUs with a third workgroup and the rest waiting for them; 3 per CU is balanced
  // (79 000 x 1024 x 384: 112 -> 102 us, x 384 x 1024: 105 -> 93 us; tools/wgradbench.py with SNIPPER_WGRAD_S)
  if (tiles > 12 && tiles * s0 > wgs && tiles * s0 < wgs * 3 / 2) {
    const int s3 = (wgs * 3 / 2 / tiles) / 8 * 8;
    if (s3 > s0) s0 = s3;
  }
  static const int s_env = [] { const char *e = getenv("SNIPPER_WGRAD_S"); return e ? atoi(e) : 0; }();   // (measurement aid)
  if (s_env > 0) s0 = s_env;
  s0 = std::min(s0, std::max(1, M / kWgRows));
  p.rows = (M + s0 - 1) / s0;
  p.S = (M + p.rows - 1) / p.rows;
  return p;
}
}  // namespace

// ---- 128 x 384 output tiles, one 8-wave workgroup per CU (csrc/wgrad_wide_bf16.cuh): long reductions whose output tiles well
// in one of the two orientations (dW = G^T X, or dW^T = X^T G with the second pass storing the transpose)
struct WidePlan { bool use, swapped; int rows_dim, cols_dim, tiles_n, tiles_kw, S, rows; };
WidePlan wgrad_wide_plan(int M, int N, int Kc) {
  WidePlan w{};
  static const bool wide_on = [] { const char *e = getenv("SNIPPER_WGRAD_WIDE"); return !(e && e[0] == '0'); }();      // (A/B aid)
  if (!wide_on || M < 32768) return w;
  double best = 0.0, best_rows = 0.0;
  for (int sw = 0; sw < 2; ++sw) {
    const int r = sw ? Kc : N, c = sw ? N : Kc;
    const int tn = (r + kWwTileN - 1) / kWwTileN, tk = (c + kWwTileK - 1) / kWwTileK;
    const double eff = (double)r * c / ((double)tn * kWwTileN * tk * kWwTileK), row_util = (double)r / (tn * kWwTileN);
    if (tn * tk > 32) continue;
    // (a tie goes to the orientation whose padding sits on the wide side: padded columns cost no operand traffic)
    if (eff > best + 1e-9 || (eff > best - 1e-9 && row_util > best_rows + 1e-9)) {
      best = eff; best_rows = row_util;
      w.swapped = sw != 0; w.rows_dim = r; w.cols_dim = c; w.tiles_n = tn; w.tiles_kw = tk;
    }
  }
  if (best < 0.7) return w;
  const int tiles = w.tiles_n * w.tiles_kw;
  // Measured on MI355X (tools/run_wgrad_ablation.sh, kernel + second pass, us; profiles/r05_wgrad_wide.txt): 79 000 x 1024 x 384
  // 82 against 102 and x 384 x 1024 99 against 105 for the 128 x 128-tile kernels, but 79 000 x 384 x 384 a tie (33.5 + 10.4
  // against 35.3 + 8.6) and x 288 x 384 slower (50 against 43): with three wide tiles the grid is 240 workgroups of 192 KB of
  // partial sums each (47 MB to write and re-read against 33), and the memory system's floor for that shape (121 MB from HBM
  // + 121 MB from L2 = 25 us, reached with the MFMAs compiled out) leaves the 4-wave ring kernel little to lose.
  static const int min_tiles = [] { const char *e = getenv("SNIPPER_WGRAD_WIDE_MIN_TILES"); return e ? atoi(e) : 6; }();
  if (tiles < min_tiles) return w;
  static const int wgs_env = [] { const char *e = getenv("SNIPPER_WGRAD_WIDE_WGS"); return e ? atoi(e) : 256; }();     // (measurement aid)
  int s0 = std::max(8, wgs_env / tiles / 8 * 8);           // one workgroup per CU; a multiple of 8 row ranges (XCD placement)
  const int rows = ((M + s0 - 1) / s0 + kWwRows - 1) / kWwRows * kWwRows;
  w.rows = rows;
  w.S = (M + rows - 1) / rows;
  w.use = w.S >= 1 && rows >= 4 * kWwRows;
  return w;
}

size_t snipper_wgrad_workspace_bytes(int M, int N, int Kc) {
  if (M <= 0 || N <= 0 || Kc <= 0) return 0;
  const WgradPlan p = wgrad_plan(M, N, Kc);
  // partial tiles are whole 128 x 128 accumulator images (wgrad_bf16.cuh), then the [S][N] partial column sums
  size_t need = ((size_t)p.S * p.tiles_n * p.tiles_k * 16384 + (size_t)p.S * N) * sizeof(float);
  const WidePlan w = wgrad_wide_plan(M, N, Kc);
  if (w.use)
    need = std::max(need, ((size_t)w.S * w.tiles_n * 3 * w.tiles_kw * 16384 + (size_t)w.S * std::max(N, Kc)) * sizeof(float));
  return need;
}

int snipper_wgrad_bf16(void *stream, const uint16_t *G, long long ldg, const uint16_t *X, long long ldx,
                       int M, int N, int Kc, const float *scale, float *dW, long long lddw, float *db,
                       int accumulate, void *workspace, size_t workspace_bytes) {
  if (!G || !X || !dW || !workspace) return SNIPPER_E_NULL;
  if (M <= 0 || N <= 0 || Kc <= 0 || N % 8 || Kc % 8 || ldg % 8 || ldx % 8 || lddw % 4 || ldg < N || ldx < Kc || lddw < Kc)
    return SNIPPER_E_SHAPE;
  if (((uintptr_t)G | (uintptr_t)X | (uintptr_t)dW | (uintptr_t)workspace) & 15) return SNIPPER_E_SHAPE;
  if (workspace_bytes < snipper_wgrad_workspace_bytes(M, N, Kc)) return SNIPPER_E_SHAPE;
  const WidePlan w = wgrad_wide_plan(M, N, Kc);
  if (w.use) {
    float *P = (float *)workspace, *Pb = db ? P + (size_t)w.S * w.tiles_n * 3 * w.tiles_kw * 16384 : nullptr;
    const WgradWideArgs a{w.swapped ? X : G, w.swapped ? ldx : ldg, w.swapped ? G : X, w.swapped ? ldg : ldx, P, Pb, M,
                          w.rows_dim, w.cols_dim, w.S, w.rows, w.tiles_n, w.tiles_kw, db ? (w.swapped ? 2 : 1) : 0, wres_debug()};
    hipLaunchKernelGGL(wgrad_wide_kernel, dim3(w.tiles_n * w.tiles_kw * w.S), dim3(kWwThreads), 0, (hipStream_t)stream, a);
    const WgradReduceArgs r{P, Pb, dW, lddw, db, scale, w.rows_dim, w.cols_dim, w.S, accumulate, w.tiles_n, 3 * w.tiles_kw,
                            w.swapped ? 1 : 0, N};
    const long long quads = (long long)w.tiles_n * 3 * w.tiles_kw * 4096;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((quads + 63) / 64)), dim3(256), 0, (hipStream_t)stream, r);
    return launch_status();
  }
  const WgradPlan p = wgrad_plan(M, N, Kc);
  float *P = (float *)workspace, *Pb = db ? P + (size_t)p.S * p.tiles_n * p.tiles_k * 16384 : nullptr;
  static const bool ring_on = [] { const char *e = getenv("SNIPPER_WGRAD_RING"); return !(e && e[0] == '0'); }();
  // long reductions onto few output tiles: the LDS-DMA ring kernel (csrc/wgrad_ring_bf16.cuh).  Measured (kernel only, us,
  // ring / register prefetch): 79 000 x 384 x 384 34.9 / 38.9, x 288 x 384 the same; 79 000 x 1024 x 384 (24 tiles) 115 / 103
  static const int ring_tiles = [] { const char *e = getenv("SNIPPER_WGRAD_RING_TILES"); return e ? atoi(e) : 12; }();   // (measurement aid)
  if (ring_on && M >= 8192 && p.rows >= 4 * kWrgRows && p.tiles_n * p.tiles_k <= ring_tiles) {
    const WgradRingArgs a{G, ldg, X, ldx, P, Pb, M, N, Kc, p.S, p.rows, p.tiles_n, p.tiles_k, wres_debug()};
    hipLaunchKernelGGL(wgrad_ring_kernel, dim3(p.tiles_n * p.tiles_k * p.S), dim3(kWrgThreads), 0, (hipStream_t)stream, a);
  } else {
    const WgradArgs a{G, ldg, X, ldx, P, Pb, M, N, Kc, p.S, p.rows, p.tiles_n, p.tiles_k, 0, 0, 0, 0, 0, 0, 0};
    hipLaunchKernelGGL(wgrad_bf16_kernel<128>, dim3(p.tiles_n * p.tiles_k * p.S), dim3(kWgThreads), 0, (hipStream_t)stream, a);
  }
  const WgradReduceArgs r{P, Pb, dW, lddw, db, scale, N, Kc, p.S, accumulate, p.tiles_n, p.tiles_k};
  const long long quads = (long long)p.tiles_n * p.tiles_k * 4096;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((quads + 63) / 64)), dim3(256), 0, (hipStream_t)stream, r);
  return launch_status();
}

namespace {
// stride-1 3x3 weight gradients with the input patch resident in LDS (csrc/wgrad_conv_patch_bf16.cuh)
struct ConvPatchPlan { bool use; int cb, tiles_co, tiles_ci, S, span, lead, ring, tiles_k128, tiles128, lds; unsigned mag_row, mag_img; };
ConvPatchPlan wgrad_conv_patch_plan(int B, int H, int Wd, int Cin, int Cout, int stride) {
  ConvPatchPlan p{};
  static const bool on = [] { const char *e = getenv("SNIPPER_WGRAD_CONV_PATCH"); return !(e && e[0] == '0'); }();      // (A/B aid)
  static const int wgs_env = [] { const char *e = getenv("SNIPPER_WGRAD_CONV_WGS"); return e ? atoi(e) : 0; }();       // (measurement aid)
  if (!on || stride != 1 || Cin % 128 || B <= 0 || H <= 0 || Wd <= 0) return p;
  p.cb = 2;          // (4 co blocks per wave -- 144 accumulator registers, half the X fragment reads per MFMA -- measured slower:
                     //  60 / 50 / 55 us against 44 / 47 / 46 at layer2 / 3 / 4, profiles/r06_wgrad_conv_patch_ab.txt)
  if (Cout % (16 * p.cb)) return p;
  const long long Wp = Wd + 2, Hp = H + 2, Q = Hp * Wp, T = (long long)B * Q;
  p.lead = (Wd + 3 + 31) / 32 * 32;
  p.ring = 2 * p.lead + kWcChunk;
  p.lds = wgrad_conv_lds_bytes(p.ring, p.cb);
  if (Q < p.lead || (double)(B + 2) * Q * Wp >= 4294967296.0 || p.lds > 160 * 1024) return p;
  p.tiles_co = Cout / (16 * p.cb);
  p.tiles_ci = Cin / kWcCi;
  const int tiles = p.tiles_co * p.tiles_ci;
  const int wgs = wgs_env > 0 ? wgs_env : 2 * device_cu_count();       // two workgroups per CU (LDS and registers allow two)
  int s0 = std::max(1, wgs / tiles);
  if (s0 >= 8) s0 = s0 / 8 * 8;
  p.span = (int)(((T + s0 - 1) / s0 + kWcChunk - 1) / kWcChunk * kWcChunk);
  p.S = (int)((T + p.span - 1) / p.span);
  if (s0 >= 8) p.S = (p.S + 7) / 8 * 8;               // (ranges past the batch write zero partials)
  p.tiles_k128 = 9 * Cin / 128;
  p.tiles128 = (Cout + 127) / 128 * p.tiles_k128;
  p.mag_row = (unsigned)((4294967296ULL + (unsigned long long)Wp - 1) / (unsigned long long)Wp);
  p.mag_img = (unsigned)((4294967296ULL + (unsigned long long)Hp - 1) / (unsigned long long)Hp);
  p.use = true;
  return p;
}
int launch_wgrad_conv_patch(hipStream_t stream, const ConvPatchPlan &p, const WgradConvArgs &a) {
  static std::atomic<int> raised[kMaxDevices];       // more than 64 KB of dynamic LDS: a per-device function attribute
  const int dev = current_device_index();
  if (dev >= kMaxDevices || raised[dev].load(std::memory_order_acquire) == 0) {
    const hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void *>(&wgrad_conv_patch_kernel<2>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e2 != hipSuccess) return (int)e2;
    if (dev < kMaxDevices) raised[dev].store(1, std::memory_order_release);
  }
  const dim3 grid(p.tiles_co * p.tiles_ci * p.S);
  hipLaunchKernelGGL(wgrad_conv_patch_kernel<2>, grid, dim3(kWcThreads), p.lds, stream, a);
  return SNIPPER_OK;
}
}  // namespace

size_t snipper_wgrad_conv3x3_workspace_bytes(int B, int H, int Wd, int Cin, int Cout, int stride) {
  if (B <= 0 || H <= 0 || Wd <= 0 || Cin <= 0 || Cout <= 0 || (stride != 1 && stride != 2)) return 0;
  const int Ho = (H - 1) / stride + 1, Wo = (Wd - 1) / stride + 1;
  size_t need = snipper_wgrad_workspace_bytes(B * Ho * Wo, Cout, 9 * Cin);
  const ConvPatchPlan cp = wgrad_conv_patch_plan(B, H, Wd, Cin, Cout, stride);
  if (cp.use) need = std::max(need, (size_t)cp.S * cp.tiles128 * 16384 * sizeof(float));
  return need;
}

int snipper_wgrad_conv3x3_bf16(void *stream, const uint16_t *G, const uint16_t *X, int B, int H, int Wd, int Cin, int Cout,
                               int stride, const float *scale, float *dW, int accumulate, void *workspace,
                               size_t workspace_bytes) {
  if (!G || !X || !dW || !workspace) return SNIPPER_E_NULL;
  if (B <= 0 || H <= 0 || Wd <= 0 || Cin <= 0 || Cout <= 0 || Cin % kWgTile || Cout % 8 || (stride != 1 && stride != 2))
    return SNIPPER_E_SHAPE;
  if (((uintptr_t)G | (uintptr_t)X | (uintptr_t)dW | (uintptr_t)workspace) & 15) return SNIPPER_E_SHAPE;
  const int Ho = (H - 1) / stride + 1, Wo = (Wd - 1) / stride + 1;
  const long long M = (long long)B * Ho * Wo;
  if (M >= (1LL << 31) || (long long)B * H * Wd * Cin >= (1LL << 30)) return SNIPPER_E_SHAPE;      // (32-bit byte offsets)
  const int Kc = 9 * Cin;
  if (workspace_bytes < snipper_wgrad_conv3x3_workspace_bytes(B, H, Wd, Cin, Cout, stride)) return SNIPPER_E_SHAPE;
  float *P = (float *)workspace;
  const ConvPatchPlan cp = wgrad_conv_patch_plan(B, H, Wd, Cin, Cout, stride);
  if (cp.use) {
    const WgradConvArgs a{G, X, P, B, H, Wd, Cin, Cout, cp.S, cp.span, cp.tiles_co, cp.tiles_ci, cp.lead, cp.ring,
                          cp.tiles_k128, cp.tiles128, cp.mag_row, cp.mag_img, wres_debug()};
    const int rc = launch_wgrad_conv_patch((hipStream_t)stream, cp, a);
    if (rc != SNIPPER_OK) return rc;
    const WgradReduceArgs r{P, nullptr, dW, Kc, nullptr, scale, Cout, Kc, cp.S, accumulate, (Cout + 127) / 128, cp.tiles_k128};
    const long long quads = (long long)cp.tiles128 * 4096;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((quads + 63) / 64)), dim3(256), 0, (hipStream_t)stream, r);
    return launch_status();
  }
  const WgradPlan p = wgrad_plan((int)M, Cout, Kc);
  const WgradArgs a{G, Cout, X, Cin, P, nullptr, (int)M, Cout, Kc, p.S, p.rows, p.tiles_n, p.tiles_k, 1, H, Wd, Cin, Ho, Wo, stride};
  hipLaunchKernelGGL(wgrad_bf16_kernel<128>, dim3(p.tiles_n * p.tiles_k * p.S), dim3(kWgThreads), 0, (hipStream_t)stream, a);
  const WgradReduceArgs r{P, nullptr, dW, Kc, nullptr, scale, Cout, Kc, p.S, accumulate, p.tiles_n, p.tiles_k};
  const long long quads = (long long)p.tiles_n * p.tiles_k * 4096;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((quads + 63) / 64)), dim3(256), 0, (hipStream_t)stream, r);
  return launch_status();
}

namespace {
inline int ln_bwd_blocks(int rows) { return std::max(1, std::min((rows + 3) / 4, kLnBwdBlocks)); }
inline bool ln_dt_ok(int dt) { return dt == 0 || dt == 1; }
}  // namespace

int snipper_add_dropout_layernorm_forward_ex(void *stream, const void *x, int x_dt, const float *x_mean, const float *x_rstd,
                                             const float *x_gamma, const float *x_beta, const void *z, int z_dt,
                                             const void *pos, int pos_dt, const float *gamma, const float *beta,
                                             int rows, int C, float p, float eps, uint64_t seed,
                                             float *s_save, float *mean, float *rstd, uint8_t *keep,
                                             float *y32, uint16_t *y16, uint16_t *yq16) {
  if (!x || !gamma || !beta || (!y32 && !y16 && !yq16)) return SNIPPER_E_NULL;
  if ((yq16 && !pos) || ((mean == nullptr) != (rstd == nullptr))) return SNIPPER_E_NULL;
  const bool lazy = x_mean || x_rstd || x_gamma || x_beta;
  if (lazy && (!x_mean || !x_rstd || !x_gamma || !x_beta)) return SNIPPER_E_NULL;
  if (rows <= 0 || C <= 0 || C % 4 || C > kLnMaxIter * 256 || !(p >= 0.f && p < 1.f) ||
      (long long)rows * C >= (1LL << 32) || !ln_dt_ok(x_dt) || !ln_dt_ok(z_dt) || !ln_dt_ok(pos_dt) || (lazy && x_dt != 0))
    return SNIPPER_E_SHAPE;
  const LnFwdArgs a{x_mean, x_rstd, x_gamma, x_beta, x, x_dt, z, z_dt, pos, pos_dt, gamma, beta, s_save, mean, rstd,
                    (z && p > 0.f) ? keep : nullptr, y32, y16, yq16, rows, C, z ? p : 0.f, eps, (uint32_t)seed,
                    (uint32_t)(seed >> 32)};
  hipLaunchKernelGGL(ln_fused_fwd_kernel, dim3((rows + 3) / 4), dim3(kLnThreads), 0, (hipStream_t)stream, a);
  return launch_status();
}

int snipper_add_dropout_layernorm_forward(void *stream, const void *x, int x_dt, const void *z, int z_dt,
                                          const void *pos, int pos_dt, const float *gamma, const float *beta,
                                          int rows, int C, float p, float eps, uint64_t seed,
                                          float *s_save, float *mean, float *rstd, uint8_t *keep,
                                          float *y32, uint16_t *y16, uint16_t *yq16) {
  return snipper_add_dropout_layernorm_forward_ex(stream, x, x_dt, nullptr, nullptr, nullptr, nullptr, z, z_dt, pos, pos_dt, gamma,
                                                  beta, rows, C, p, eps, seed, s_save, mean, rstd, keep, y32, y16, yq16);
}

size_t snipper_add_dropout_layernorm_workspace_bytes(int rows, int C) {
  if (rows <= 0 || C <= 0) return 0;
  return (size_t)ln_bwd_blocks(rows) * 2 * C * sizeof(float);
}

int snipper_add_dropout_layernorm_backward(void *stream, const float *g32, const uint16_t *g16, const uint16_t *gq16,
                                           const float *s_save, const float *mean, const float *rstd,
                                           const float *gamma, const uint8_t *keep, int rows, int C, float p,
                                           void *dx, int dx_dt, void *dz, int dz_dt, float *dgamma, float *dbeta,
                                           void *workspace, size_t workspace_bytes) {
  if (!s_save || !mean || !rstd || !gamma || !dgamma || !dbeta || !workspace || (!g32 && !g16 && !gq16))
    return SNIPPER_E_NULL;
  if (rows <= 0 || C <= 0 || C % 4 || C > kLnMaxIter * 256 || !(p >= 0.f && p < 1.f) || !ln_dt_ok(dx_dt) || !ln_dt_ok(dz_dt))
    return SNIPPER_E_SHAPE;
  if (workspace_bytes < snipper_add_dropout_layernorm_workspace_bytes(rows, C)) return SNIPPER_E_SHAPE;
  const int blocks = ln_bwd_blocks(rows);
  const LnBwdArgs a{g32, g16, gq16, s_save, mean, rstd, gamma, p > 0.f ? keep : nullptr, dx, dx_dt, dz, dz_dt,
                    (float *)workspace, rows, C, p};
  switch ((C + 255) / 256) {
    case 1: hipLaunchKernelGGL(ln_fused_bwd_kernel<1>, dim3(blocks), dim3(kLnThreads), 0, (hipStream_t)stream, a); break;
    case 2: hipLaunchKernelGGL(ln_fused_bwd_kernel<2>, dim3(blocks), dim3(kLnThreads), 0, (hipStream_t)stream, a); break;
    case 3: hipLaunchKernelGGL(ln_fused_bwd_kernel<3>, dim3(blocks), dim3(kLnThreads), 0, (hipStream_t)stream, a); break;
    default: hipLaunchKernelGGL(ln_fused_bwd_kernel<4>, dim3(blocks), dim3(kLnThreads), 0, (hipStream_t)stream, a); break;
  }
  hipLaunchKernelGGL(ln_param_grad_kernel, dim3((2 * C + kLnPgCh - 1) / kLnPgCh), dim3(256), 0, (hipStream_t)stream,
                     (const float *)workspace, blocks, C, dgamma, dbeta);
  return launch_status();
}

// ---- decoder-size residual + dropout + LayerNorm chain (csrc/small_ln.cuh) ----
int snipper_small_ln_forward_f32(void *stream, const float *x, const float *z, const float *pos, const float *gamma, const float *beta,
                                 int rows, int C, float p, float eps, uint64_t seed, float *s_save, float *mean, float *rstd,
                                 uint8_t *keep, float *y, float *yq) {
  if (!x || !gamma || !beta || !y) return SNIPPER_E_NULL;
  if ((yq && !pos) || ((mean == nullptr) != (rstd == nullptr))) return SNIPPER_E_NULL;
  if (rows <= 0 || rows > kSmallLnMaxRows || C <= 0 || C % 4 || C > kLnMaxIter * 256 || !(p >= 0.f && p < 1.f)) return SNIPPER_E_SHAPE;
  if (((uintptr_t)x | (uintptr_t)z | (uintptr_t)pos | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)s_save | (uintptr_t)y | (uintptr_t)yq) & 15)
    return SNIPPER_E_SHAPE;
  const SmallLnFwdArgs a{x, z, pos, gamma, beta, s_save, mean, rstd, (z && p > 0.f) ? keep : nullptr, y, yq, rows, C, z ? p : 0.f, eps,
                         (uint32_t)seed, (uint32_t)(seed >> 32)};
  hipLaunchKernelGGL(small_ln_fwd_kernel, dim3((rows + 3) / 4), dim3(kLnThreads), 0, (hipStream_t)stream, a);
  return launch_status();
}

int snipper_small_ln_backward_f32(void *stream, const float *g0, const float *g1, const float *g2, const float *g3,
                                  const float *s_save, const float *mean, const float *rstd, const float *gamma, const uint8_t *keep,
                                  int rows, int C, float p, float *dx, float *dz, float *dgamma, float *dbeta) {
  if (!s_save || !mean || !rstd || !gamma || !dgamma || !dbeta || (!g0 && !g1 && !g2 && !g3)) return SNIPPER_E_NULL;
  if (rows <= 0 || rows > kSmallLnMaxRows || C <= 0 || C % 4 || C > kLnMaxIter * 256 || !(p >= 0.f && p < 1.f)) return SNIPPER_E_SHAPE;
  if (((uintptr_t)g0 | (uintptr_t)g1 | (uintptr_t)g2 | (uintptr_t)g3 | (uintptr_t)s_save | (uintptr_t)gamma | (uintptr_t)dx | (uintptr_t)dz) & 15)
    return SNIPPER_E_SHAPE;
  const int row_blocks = (rows + 3) / 4, col_blocks = (C + kSmallLnColsPerWg - 1) / kSmallLnColsPerWg;
  const SmallLnBwdArgs a{{g0, g1, g2, g3}, s_save, mean, rstd, gamma, p > 0.f ? keep : nullptr, dx, dz, dgamma, dbeta, rows, C,
                         row_blocks, p};
  hipLaunchKernelGGL(small_ln_bwd_kernel, dim3(row_blocks + col_blocks), dim3(kLnThreads), 0, (hipStream_t)stream, a);
  return launch_status();
}

int snipper_sum_f32(void *stream, const float *const *srcs, int n_src, float *out, long long numel) {
  if (!srcs || !out) return SNIPPER_E_NULL;
  if (n_src <= 0 || n_src > kSumF32MaxSrc || numel <= 0 || numel % 4 || ((uintptr_t)out % 16)) return SNIPPER_E_SHAPE;
  SumF32Srcs a{};
  a.n = n_src;
  for (int i = 0; i < n_src; ++i) {
    if (!srcs[i] || ((uintptr_t)srcs[i] % 16)) return srcs[i] ? SNIPPER_E_SHAPE : SNIPPER_E_NULL;
    a.p[i] = srcs[i];
  }
  const long long n4 = numel / 4;
  hipLaunchKernelGGL(sum_f32_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, out, n4);
  return launch_status();
}

int snipper_refine_reference_linear_f32(void *stream, const float *x, const float *W, const float *b, const float *ref,
                                        const float *valid_ratios, int rows, int C, int rows_per_batch, int L, float eps,
                                        float *new_ref, float *ref_in) {
  if (!x || !W || !ref || !valid_ratios || !new_ref || !ref_in) return SNIPPER_E_NULL;
  if (rows <= 0 || C <= 0 || C % 4 || rows_per_batch <= 0 || L <= 0 || (((uintptr_t)x | (uintptr_t)W) & 15)) return SNIPPER_E_SHAPE;
  hipLaunchKernelGGL(refine_reference_linear_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, W, b, ref, valid_ratios,
                     rows, C, rows_per_batch, L, eps, new_ref, ref_in);
  return launch_status();
}

namespace {
struct GnPlan { int rpp, nblk; };
inline GnPlan gn_plan(int hw, int C) {
  GnPlan p;
  p.rpp = std::max(1, 256 / (C / 4));
  p.nblk = std::max(1, std::min((hw + p.rpp * 8 - 1) / (p.rpp * 8), 128));
  return p;
}
inline bool gn_shape_ok(int n, int hw, int C, int G) {
  return n > 0 && hw > 0 && C > 0 && G > 0 && G <= kGnMaxGroups && C % G == 0 && (C / G) % 4 == 0 && C / 4 <= 256;
}
}  // namespace

size_t snipper_groupnorm_tokens_workspace_bytes(int n, int hw, int C, int G) {
  if (!gn_shape_ok(n, hw, C, G)) return 0;
  const GnPlan p = gn_plan(hw, C);
  return ((size_t)n * p.nblk * G * 2 + (size_t)n * p.nblk * 2 * C) * sizeof(float);
}

int snipper_groupnorm_tokens_forward(void *stream, const uint16_t *x, const float *gamma, const float *beta,
                                     int n, int hw, int C, int G, float eps,
                                     long long dst_rows_per_image, long long dst_row_offset,
                                     const void *pos, int pos_dt, float *y32, uint16_t *y16, uint16_t *yq16,
                                     float *stats, void *workspace, size_t workspace_bytes) {
  if (!x || !gamma || !beta || !stats || !workspace || (!y32 && !y16 && !yq16) || (yq16 && !pos)) return SNIPPER_E_NULL;
  if (!gn_shape_ok(n, hw, C, G) || !ln_dt_ok(pos_dt) || dst_rows_per_image < hw || dst_row_offset < 0 ||
      workspace_bytes < snipper_groupnorm_tokens_workspace_bytes(n, hw, C, G))
    return SNIPPER_E_SHAPE;
  const GnPlan p = gn_plan(hw, C);
  GnArgs a{};
  a.x = x; a.gamma = gamma; a.beta = beta; a.part = (float *)workspace; a.stats = stats;
  a.y32 = y32; a.y16 = y16; a.yq16 = yq16; a.pos = pos; a.pos_dt = pos_dt;
  a.dst_rows_per_image = dst_rows_per_image; a.dst_row_offset = dst_row_offset;
  a.n = n; a.hw = hw; a.C = C; a.G = G; a.nblk = p.nblk; a.rpp = p.rpp; a.eps = eps;
  const dim3 grid(p.nblk, n), block((C / 4) * p.rpp);
  hipLaunchKernelGGL(gn_stats_kernel, grid, block, 0, (hipStream_t)stream, a);
  hipLaunchKernelGGL(gn_apply_kernel, grid, block, 0, (hipStream_t)stream, a);
  return launch_status();
}

int snipper_groupnorm_tokens_backward(void *stream, const uint16_t *x, const float *gamma, const float *stats,
                                      const float *g32, const uint16_t *g16, const uint16_t *gq16,
                                      int n, int hw, int C, int G,
                                      long long dst_rows_per_image, long long dst_row_offset,
                                      uint16_t *dx, float *dgamma, float *dbeta, void *workspace, size_t workspace_bytes) {
  if (!x || !gamma || !stats || !dx || !dgamma || !dbeta || !workspace || (!g32 && !g16 && !gq16)) return SNIPPER_E_NULL;
  if (!gn_shape_ok(n, hw, C, G) || dst_rows_per_image < hw || dst_row_offset < 0 ||
      workspace_bytes < snipper_groupnorm_tokens_workspace_bytes(n, hw, C, G))
    return SNIPPER_E_SHAPE;
  const GnPlan p = gn_plan(hw, C);
  GnArgs a{};
  a.x = x; a.gamma = gamma; a.part = (float *)workspace; a.stats = const_cast<float *>(stats);
  a.part_param = (float *)workspace + (size_t)n * p.nblk * G * 2;
  a.g32 = g32; a.g16 = g16; a.gq16 = gq16; a.dx = dx;
  a.dst_rows_per_image = dst_rows_per_image; a.dst_row_offset = dst_row_offset;
  a.n = n; a.hw = hw; a.C = C; a.G = G; a.nblk = p.nblk; a.rpp = p.rpp;
  const dim3 grid(p.nblk, n), block((C / 4) * p.rpp);
  hipLaunchKernelGGL(gn_bwd_stats_kernel, grid, block, 0, (hipStream_t)stream, a);
  hipLaunchKernelGGL(gn_bwd_apply_kernel, grid, block, 0, (hipStream_t)stream, a);
  hipLaunchKernelGGL(ln_param_grad_kernel, dim3((2 * C + kLnPgCh - 1) / kLnPgCh), dim3(256), 0, (hipStream_t)stream,
                     (const float *)a.part_param, n * p.nblk, C, dgamma, dbeta);
  return launch_status();
}

size_t snipper_colsum_workspace_bytes(int n_images, int rows_per_seg, int C) {
  if (n_images <= 0 || rows_per_seg <= 0 || C <= 0 || C % 4 || C / 4 > 256) return 0;
  const GnPlan p = gn_plan(rows_per_seg, C);
  return (size_t)n_images * p.nblk * C * sizeof(float);
}

int snipper_stem_pool_bf16(void *stream, const uint16_t *y, const float *shift, int N, int H, int W, int C, uint16_t *out) {
  if (!y || !shift || !out) return SNIPPER_E_NULL;
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 8 || (((uintptr_t)y | (uintptr_t)out | (uintptr_t)shift) & 15)) return SNIPPER_E_SHAPE;
  const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;          // floor((H + 2 - 3) / 2) + 1
  const long long total = (long long)N * OH * OW * (C / 8);
  hipLaunchKernelGGL(stem_pool_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, y, shift,
                     N, H, W, C, OH, OW, out);
  return launch_status();
}

int snipper_colsum_segments_bf16(void *stream, const uint16_t *x, long long image_stride, int n_images,
                                 int rows_per_seg, int C, float *out, void *workspace, size_t workspace_bytes) {
  if (!x || !out || !workspace) return SNIPPER_E_NULL;
  if (n_images <= 0 || rows_per_seg <= 0 || C <= 0 || C % 4 || C / 4 > 256 || image_stride < (long long)rows_per_seg * C ||
      image_stride % 4 || ((uintptr_t)x % 8) || workspace_bytes < snipper_colsum_workspace_bytes(n_images, rows_per_seg, C))
    return SNIPPER_E_SHAPE;
  const GnPlan p = gn_plan(rows_per_seg, C);
  hipLaunchKernelGGL(colsum_partial_kernel, dim3(p.nblk, n_images), dim3((C / 4) * p.rpp), 0, (hipStream_t)stream, x,
                     image_stride, rows_per_seg, C, p.nblk, p.rpp, (float *)workspace);
  hipLaunchKernelGGL(colsum_final_kernel, dim3((C + 15) / 16), dim3(256), 0, (hipStream_t)stream,
                     (const float *)workspace, n_images * p.nblk, C, out);
  return launch_status();
}

int snipper_colsum_segments_multi_bf16(void *stream, const uint16_t *const *srcs, int n_src, long long elem_offset,
                                       long long image_stride, int n_images, int rows_per_seg, int C, float *out,
                                       void *workspace, size_t workspace_bytes) {
  if (!srcs || !out || !workspace) return SNIPPER_E_NULL;
  if (n_src <= 0 || n_src > 8 || n_images <= 0 || rows_per_seg <= 0 || C <= 0 || C % 4 || C / 4 > 256 || elem_offset < 0 ||
      elem_offset % 4 || image_stride < (long long)rows_per_seg * C || image_stride % 4 ||
      workspace_bytes < (size_t)n_src * snipper_colsum_workspace_bytes(n_images, rows_per_seg, C))
    return SNIPPER_E_SHAPE;
  ColsumSrcs a{};
  for (int i = 0; i < n_src; ++i) {
    if (!srcs[i] || ((uintptr_t)srcs[i] % 8)) return srcs[i] ? SNIPPER_E_SHAPE : SNIPPER_E_NULL;
    a.p[i] = srcs[i] + elem_offset;
  }
  const GnPlan p = gn_plan(rows_per_seg, C);
  hipLaunchKernelGGL(colsum_partial_multi_kernel, dim3(p.nblk, n_images), dim3((C / 4) * p.rpp), 0, (hipStream_t)stream, a,
                     n_src, image_stride, rows_per_seg, C, p.nblk, p.rpp, (float *)workspace);
  hipLaunchKernelGGL(colsum_final_kernel, dim3((C + 15) / 16), dim3(256), 0, (hipStream_t)stream,
                     (const float *)workspace, n_images * p.nblk, C, out);
  return launch_status();
}

int snipper_sum_bf16(void *stream, const uint16_t *const *srcs, int n_src, uint16_t *out, long long numel) {
  if (!srcs || !out) return SNIPPER_E_NULL;
  if (n_src <= 0 || n_src > kSumMaxSrc || numel <= 0 || numel % 8 || ((uintptr_t)out % 16)) return SNIPPER_E_SHAPE;
  SumSrcs a{};
  a.n = n_src;
  for (int i = 0; i < n_src; ++i) {
    if (!srcs[i] || ((uintptr_t)srcs[i] % 16)) return srcs[i] ? SNIPPER_E_SHAPE : SNIPPER_E_NULL;
    a.p[i] = srcs[i];
  }
  const long long n8 = numel / 8;
  hipLaunchKernelGGL(sum_bf16_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, out, n8);
  return launch_status();
}

int snipper_stem_pack_bf16(void *stream, const float *x, int N, int H, int W, uint16_t *out) {
  if (!x || !out) return SNIPPER_E_NULL;
  if (N <= 0 || H <= 0 || W <= 0 || W % 4 || ((uintptr_t)x % 16) || ((uintptr_t)out % 16) || (long long)N * H * W >= (1LL << 31))
    return SNIPPER_E_SHAPE;
  const long long plane = (long long)H * W, quads = (long long)N * plane / 4;
  hipLaunchKernelGGL(stem_pack_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, plane, quads, out);
  return launch_status();
}

namespace {
// 128 x 64 tiles when the 128 x 128 grid would leave the chip under-filled (or Cout has a half-empty last tile)
inline bool conv_use_n64(long long M, int Cout) {
  const long long tiles128 = ((M + 127) / 128) * ((Cout + 127) / 128);
  return tiles128 < 768 || (Cout % 128 != 0 && Cout % 128 <= 64);
}
inline unsigned conv_grid_n64(long long M, int Cout) {
  const long long tiles_m = (M + 127) / 128, tiles_n = (Cout + 63) / 64;
  return (unsigned)(tiles_n * 8 * ((tiles_m + 7) / 8));
}
// the LDS-DMA ring instance of the 128 x 64 kernel (csrc/conv3x3_ring_bf16.cuh)
// -- where it is faster: the small maps (layer3 / layer4 of the ResNet, M <= 32 768 output pixels: 42.5 against 46.0 us and
// 63 against 67 us, profiles/r04_conv_ring_ab.jsonl); on the large maps the register-prefetch kernel's four workgroups per
// CU win (40 against 53 us at 150 x 200).  SNIPPER_CONV_RING = 0 / 2: never / always (A/B runs).
inline bool conv_ring_on(long long M) {
  static const int mode = [] { const char *e = getenv("SNIPPER_CONV_RING"); return e ? atoi(e) : 1; }();
  return mode == 2 || (mode == 1 && M <= 32768);
}
}  // namespace

int snipper_conv3x3_bf16(void *stream, const uint16_t *X, const uint16_t *W, const float *bias, uint16_t *Y,
                         int B, int H, int Wd, int Cin, int Cout, int stride, int relu, const uint16_t *gate, int flip_taps) {
  if (!X || !W || !Y) return SNIPPER_E_NULL;
  if (gate && (((uintptr_t)gate & 15) || relu)) return SNIPPER_E_SHAPE;
  if (B <= 0 || H <= 0 || Wd <= 0 || Cin <= 0 || Cout <= 0 || Cin % kGemmBK || Cout % 4 || (stride != 1 && stride != 2))
    return SNIPPER_E_SHAPE;
  const int Ho = (H - 1) / stride + 1, Wo = (Wd - 1) / stride + 1;
  const long long M = (long long)B * Ho * Wo;
  if (M >= (1LL << 31) || (long long)B * H * Wd * Cin >= (1LL << 30)) return SNIPPER_E_SHAPE;   // (32-bit byte offsets)
  const Conv3x3Args g{X, W, bias, Y, B, H, Wd, Cin, Cout, Ho, Wo, stride, 0, 0, 0, 0, 0, gate, flip_taps ? 1 : 0};
  if (conv_use_n64(M, Cout)) {
    const dim3 grid64(conv_grid_n64(M, Cout));
    if (conv_ring_on(M)) {
      if (relu) hipLaunchKernelGGL(conv3x3_ring_kernel<true>, grid64, dim3(kGemmThreads), 0, (hipStream_t)stream, g);
      else hipLaunchKernelGGL(conv3x3_ring_kernel<false>, grid64, dim3(kGemmThreads), 0, (hipStream_t)stream, g);
      return launch_status();
    }
    if (relu) hipLaunchKernelGGL((conv3x3_bf16_kernel<true, 64>), grid64, dim3(kGemmThreads), 0, (hipStream_t)stream, g);
    else hipLaunchKernelGGL((conv3x3_bf16_kernel<false, 64>), grid64, dim3(kGemmThreads), 0, (hipStream_t)stream, g);
    return launch_status();
  }
  const dim3 grid(gemm_grid_size(M, Cout));
  if (relu)
    hipLaunchKernelGGL((conv3x3_bf16_kernel<true, 128>), grid, dim3(kGemmThreads), 0, (hipStream_t)stream, g);
  else
    hipLaunchKernelGGL((conv3x3_bf16_kernel<false, 128>), grid, dim3(kGemmThreads), 0, (hipStream_t)stream, g);
  return launch_status();
}

namespace {
// output tile of the patch-resident 3x3 kernel (csrc/conv3x3_patch_bf16.cuh): TH x TW <= 128 pixels whose input patch
// (TH + 2) x (TW + 2) fits the kernel's 192 LDS rows, with the fewest tiles per image (then the smallest patch)
bool conv_patch_tile(int H, int Wd, int &TH, int &TW) {
  long long best = -1;
  for (int tw = 4; tw <= 62 && tw <= std::max(Wd, 4); ++tw) {
    int th = std::min(128 / tw, H);
    while (th >= 1 && (th + 2) * (tw + 2) > kCpRows) --th;
    if (th < 1) continue;
    const long long tiles = (long long)((H + th - 1) / th) * ((Wd + tw - 1) / tw);
    const long long cost = tiles * 1024 + (th + 2) * (tw + 2);
    if (best < 0 || cost < best) { best = cost; TH = th; TW = tw; }
  }
  return best >= 0;
}
bool conv_patch_ok(int B, int H, int Wd, int Cin, int Cout) {
  return B > 0 && H > 0 && Wd >= 4 && Cin > 0 && Cout > 0 && Cin % 64 == 0 && Cout % 64 == 0 &&
         (long long)B * H * Wd * Cin < (1LL << 30) && (long long)B * H * Wd * Cout < (1LL << 30);
}
}  // namespace

int snipper_conv3x3_patch_supported(int B, int H, int Wd, int Cin, int Cout) { return conv_patch_ok(B, H, Wd, Cin, Cout) ? 1 : 0; }

namespace {
int conv_pack(void *stream, int n, const void *const *src, void *const *dst, const int *cout, const int *cin,
              const int *transposed, int taps) {
  if (n < 0 || (n > 0 && (!src || !dst || !cout || !cin || !transposed))) return SNIPPER_E_NULL;
  for (int lo = 0; lo < n; lo += kCpPackMax) {
    ConvPackBatch b{};
    b.count = std::min(n - lo, kCpPackMax);
    long long pieces = 0;
    for (int i = 0; i < b.count; ++i) {
      const int co = cout[lo + i], ci = cin[lo + i], tr = transposed[lo + i] ? 1 : 0;
      if (!src[lo + i] || !dst[lo + i]) return SNIPPER_E_NULL;
      if (co <= 0 || ci <= 0 || co % 64 || ci % 64 || (((uintptr_t)src[lo + i] | (uintptr_t)dst[lo + i]) & 15)) return SNIPPER_E_SHAPE;
      pieces += (long long)co * ci * taps / 8;
      if (pieces >= (1LL << 31)) return SNIPPER_E_SHAPE;
      b.it[i] = ConvPackItem{(const uint16_t *)src[lo + i], (uint16_t *)dst[lo + i], co, ci, tr, (int)pieces, taps};
    }
    hipLaunchKernelGGL(conv3x3_pack_kernel, dim3((unsigned)((pieces + 255) / 256)), dim3(256), 0, (hipStream_t)stream, b);
    if (int rc = launch_status()) return rc;
  }
  return SNIPPER_OK;
}
}  // namespace

int snipper_conv3x3_pack_bf16(void *stream, int n, const void *const *src, void *const *dst, const int *cout, const int *cin,
                              const int *transposed) {
  return conv_pack(stream, n, src, dst, cout, cin, transposed, 9);
}

int snipper_linear_pack_bf16(void *stream, int n, const void *const *src, void *const *dst, const int *N, const int *K,
                             const int *transposed) {
  return conv_pack(stream, n, src, dst, N, K, transposed, 1);
}

int snipper_linear_patch_supported(long long M, int N, int K) {
  return (M >= 256 && M < (1LL << 30) && N > 0 && K > 0 && N % 64 == 0 && K % 64 == 0 && M * K < (1LL << 30) && M * N < (1LL << 30)) ? 1 : 0;
}

int snipper_linear_patch_bf16(void *stream, const uint16_t *X, const uint16_t *Wp, const float *bias, const uint16_t *res,
                              uint16_t *Y, int M, int N, int K, int relu, const uint16_t *gate, int bn) {
  if (!X || !Wp || !Y) return SNIPPER_E_NULL;
  if (!snipper_linear_patch_supported(M, N, K) || (gate && relu) ||
      (((uintptr_t)X | (uintptr_t)Wp | (uintptr_t)Y | (uintptr_t)gate) & 15) || ((uintptr_t)res & 7) || (bn != 0 && bn != 64 && bn != 128))
    return SNIPPER_E_SHAPE;
  const ConvPatchArgs g{X, Wp, bias, Y, gate, 1, 1, 1, K, N, 1, 1, 1, 1, M, res};
  const long long tiles_m = ((long long)M + 127) / 128;
  const bool n128 = N % 128 == 0 && (bn == 128 || (bn == 0 && tiles_m * (N / 128) >= 512));
  const int b_n = n128 ? 128 : 64;
  const dim3 grid((unsigned)((N / b_n) * 8 * ((tiles_m + 7) / 8)));
  const unsigned lds = (unsigned)conv_patch_lds_bytes(K, b_n, 1);
  if (n128) {
    if (relu) hipLaunchKernelGGL((conv3x3_patch_kernel<true, 128, 1>), grid, dim3(kCpThreads), lds, (hipStream_t)stream, g);
    else hipLaunchKernelGGL((conv3x3_patch_kernel<false, 128, 1>), grid, dim3(kCpThreads), lds, (hipStream_t)stream, g);
  } else {
    if (relu) hipLaunchKernelGGL((conv3x3_patch_kernel<true, 64, 1>), grid, dim3(kCpThreads), lds, (hipStream_t)stream, g);
    else hipLaunchKernelGGL((conv3x3_patch_kernel<false, 64, 1>), grid, dim3(kCpThreads), lds, (hipStream_t)stream, g);
  }
  return launch_status();
}

int snipper_conv3x3_patch_bf16(void *stream, const uint16_t *X, const uint16_t *Wp, const float *bias, uint16_t *Y,
                               int B, int H, int Wd, int Cin, int Cout, int relu, const uint16_t *gate) {
  if (!X || !Wp || !Y) return SNIPPER_E_NULL;
  if (gate && (((uintptr_t)gate & 15) || relu)) return SNIPPER_E_SHAPE;
  if (!conv_patch_ok(B, H, Wd, Cin, Cout) || (((uintptr_t)X | (uintptr_t)Wp | (uintptr_t)Y) & 15)) return SNIPPER_E_SHAPE;
  int TH = 0, TW = 0;
  if (!conv_patch_tile(H, Wd, TH, TW)) return SNIPPER_E_SHAPE;
  const int nty = (H + TH - 1) / TH, ntx = (Wd + TW - 1) / TW;
  const ConvPatchArgs g{X, Wp, bias, Y, gate, B, H, Wd, Cin, Cout, TH, TW, nty, ntx, 0, nullptr};
  const long long tiles_m = (long long)B * nty * ntx;
  // 128 output channels per workgroup unless that leaves CUs without one (SNIPPER_CONV_PATCH_MIN128: workgroups below which
  // the 64-channel instance is taken)
  // (measured, tools/convbench.py: 256 workgroups of 128 channels at 38 x 50 x 256 take 23.2 us, 512 of 64 channels 21.2;
  //  480 / 960 at 75 x 100 x 128: 22.5 / 26.1; 128 / 256 at 19 x 25 x 512: 35.4 / 27.3)
  static const long long min128 = [] { const char *e = getenv("SNIPPER_CONV_PATCH_MIN128"); return e ? atoll(e) : 384LL; }();
  const bool n128 = Cout % 128 == 0 && tiles_m * (Cout / 128) >= min128;
  const int bn = n128 ? 128 : 64;
  const dim3 grid((unsigned)((Cout / bn) * 8 * ((tiles_m + 7) / 8)));
  const unsigned lds = (unsigned)conv_patch_lds_bytes(Cin, bn);
  if (n128) {
    if (relu) hipLaunchKernelGGL((conv3x3_patch_kernel<true, 128>), grid, dim3(kCpThreads), lds, (hipStream_t)stream, g);
    else hipLaunchKernelGGL((conv3x3_patch_kernel<false, 128>), grid, dim3(kCpThreads), lds, (hipStream_t)stream, g);
  } else {
    if (relu) hipLaunchKernelGGL((conv3x3_patch_kernel<true, 64>), grid, dim3(kCpThreads), lds, (hipStream_t)stream, g);
    else hipLaunchKernelGGL((conv3x3_patch_kernel<false, 64>), grid, dim3(kCpThreads), lds, (hipStream_t)stream, g);
  }
  return launch_status();
}

int snipper_linear_wide_supported(long long M, int N, int K) {
  return (M >= 8192 && N == kLwN && K >= 256 && K % 128 == 0 && M * K < (1LL << 30)) ? 1 : 0;
}

int snipper_linear_wide_bf16(void *stream, const uint16_t *X, const uint16_t *Wp, const float *bias, uint16_t *Y, int M, int N, int K) {
  if (!X || !Wp || !Y) return SNIPPER_E_NULL;
  if (!snipper_linear_wide_supported(M, N, K) || (((uintptr_t)X | (uintptr_t)Wp | (uintptr_t)Y) & 15) || (bias && ((uintptr_t)bias & 15)))
    return SNIPPER_E_SHAPE;
  const LinearWideArgs g{X, Wp, bias, Y, M, K};
  // 160-row tiles when they fill the rounds of one workgroup per CU better than 128-row tiles (SNIPPER_LINEAR_WIDE_MT = 4 / 5 forces)
  static const int forced = [] { const char *e = getenv("SNIPPER_LINEAR_WIDE_MT"); return e ? atoi(e) : 0; }();
  const int cus = device_cu_count();
  auto cost = [&](int rows) { const long long t = ((long long)M + rows - 1) / rows; return (double)((t + cus - 1) / cus) * rows; };
  const int mt = forced == 4 || forced == 5 ? forced : (cost(160) < cost(128) ? 5 : 4);
  // (more than 64 KB of dynamic LDS: the limit is a per-DEVICE function attribute -- raised once per device and kernel)
  {
    static std::atomic<int> raised[kMaxDevices];       // 0 = not yet, 1 = done
    const int dev = current_device_index();
    if (dev >= kMaxDevices || raised[dev].load(std::memory_order_acquire) == 0) {
      const hipError_t attr4 = hipFuncSetAttribute(reinterpret_cast<const void *>(&linear_wide_kernel<4>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, linear_wide_lds_bytes<4>());
      const hipError_t attr5 = hipFuncSetAttribute(reinterpret_cast<const void *>(&linear_wide_kernel<5>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, linear_wide_lds_bytes<5>());
      if (attr4 != hipSuccess || attr5 != hipSuccess) return (int)(attr4 != hipSuccess ? attr4 : attr5);
      if (dev < kMaxDevices) raised[dev].store(1, std::memory_order_release);
    }
  }
  if (mt == 5)
    hipLaunchKernelGGL(linear_wide_kernel<5>, dim3((unsigned)((M + 159) / 160)), dim3(kLwThreads), linear_wide_lds_bytes<5>(), (hipStream_t)stream, g);
  else
    hipLaunchKernelGGL(linear_wide_kernel<4>, dim3((unsigned)((M + 127) / 128)), dim3(kLwThreads), linear_wide_lds_bytes<4>(), (hipStream_t)stream, g);
  return launch_status();
}

int snipper_stem7x7_bf16(void *stream, const uint16_t *X4, const uint16_t *Wp, uint16_t *Y, int B, int H, int Wd) {
  if (!X4 || !Wp || !Y) return SNIPPER_E_NULL;
  if (B <= 0 || H <= 0 || Wd <= 0) return SNIPPER_E_SHAPE;
  if (((uintptr_t)X4 | (uintptr_t)Wp | (uintptr_t)Y) & 15) return SNIPPER_E_SHAPE;
  const int Ho = (H - 1) / 2 + 1, Wo = (Wd - 1) / 2 + 1;           // (H + 2*3 - 7) / 2 + 1
  const long long M = (long long)B * Ho * Wo;
  // the kernel addresses both arrays with 32-bit byte offsets (like the conv3x3 entry points): X4 is 8 B per pixel,
  // Y 128 B per output pixel
  if ((long long)B * H * Wd * 8 >= (1LL << 31) || M * 128 >= (1LL << 31)) return SNIPPER_E_SHAPE;
  const StemArgs g{X4, Wp, Y, B, H, Wd, Ho, Wo};
  hipLaunchKernelGGL(stem7x7_bf16_kernel, dim3((unsigned)((M + kGemmBM - 1) / kGemmBM)), dim3(kGemmThreads), 0,
                     (hipStream_t)stream, g);
  return launch_status();
}

int snipper_conv3x3_dgrad_s2_bf16(void *stream, const uint16_t *G, const uint16_t *Wt, uint16_t *dX,
                                  int B, int Hx, int Wx, int Cx, int Cg, const uint16_t *gate) {
  if (!G || !Wt || !dX) return SNIPPER_E_NULL;
  if (gate && ((uintptr_t)gate & 15)) return SNIPPER_E_SHAPE;
  if (B <= 0 || Hx <= 0 || Wx <= 0 || Cx <= 0 || Cg <= 0 || Cg % kGemmBK || Cx % 4) return SNIPPER_E_SHAPE;
  const int Hg = (Hx - 1) / 2 + 1, Wg = (Wx - 1) / 2 + 1;          // the stride-2 convolution's output size
  if ((long long)B * Hx * Wx >= (1LL << 31) || (long long)B * Hg * Wg * Cg >= (1LL << 30)) return SNIPPER_E_SHAPE;
  // all four parity classes as ONE launch when they take the same kernel instance (they do unless a size threshold falls
  // between two class sizes): four launches of 4-16 K-steps were mostly ramp and tail (round 6; SNIPPER_DGRAD2_MERGE=0: A/B)
  static const bool merge_on = [] { const char *e = getenv("SNIPPER_DGRAD2_MERGE"); return !(e && e[0] == '0'); }();
  if (merge_on) {
    Conv3x3Args g{G, Wt, nullptr, dX, B, Hg, Wg, Cg, Cx, 0, 0, 1, 2, 0, 0, Hx, Wx, gate, 0, {0, 0, 0, 0}};
    int kind = -1;                  // 0 ring n64, 1 register-prefetch n64, 2 register-prefetch n128
    bool same = true;
    unsigned end = 0;
    for (int c = 0; c < 4; ++c) {
      const int cy = c < 2 ? 1 : 0, cx = (c & 1) ? 0 : 1;
      const int Hc = (Hx - cy + 1) / 2, Wc = (Wx - cx + 1) / 2;
      const long long Mc = (long long)B * Hc * Wc;
      if (Hc > 0 && Wc > 0) {
        const int k = conv_use_n64(Mc, Cx) ? (conv_ring_on(Mc) ? 0 : 1) : 2;
        if (kind < 0) kind = k;
        same = same && k == kind;
        end += k == 2 ? gemm_grid_size(Mc, Cx) : conv_grid_n64(Mc, Cx);
      }
      g.cls_end[c] = (int)end;
    }
    if (same && kind >= 0 && end > 0) {
      if (kind == 0)
        hipLaunchKernelGGL(conv3x3_ring_kernel<false>, dim3(end), dim3(kGemmThreads), 0, (hipStream_t)stream, g);
      else if (kind == 1)
        hipLaunchKernelGGL((conv3x3_bf16_kernel<false, 64>), dim3(end), dim3(kGemmThreads), 0, (hipStream_t)stream, g);
      else
        hipLaunchKernelGGL((conv3x3_bf16_kernel<false, 128>), dim3(end), dim3(kGemmThreads), 0, (hipStream_t)stream, g);
      return launch_status();
    }
  }
  for (int cy = 0; cy < 2; ++cy)
    for (int cx = 0; cx < 2; ++cx) {
      const int Hc = (Hx - cy + 1) / 2, Wc = (Wx - cx + 1) / 2;    // input pixels (2a + cy, 2b + cx) of this class
      if (Hc <= 0 || Wc <= 0) continue;
      const Conv3x3Args g{G, Wt, nullptr, dX, B, Hg, Wg, Cg, Cx, Hc, Wc, 1, 1, cy, cx, Hx, Wx, gate, 0};
      const long long Mc = (long long)B * Hc * Wc;
      if (conv_use_n64(Mc, Cx) && conv_ring_on(Mc))
        hipLaunchKernelGGL(conv3x3_ring_kernel<false>, dim3(conv_grid_n64(Mc, Cx)), dim3(kGemmThreads), 0, (hipStream_t)stream, g);
      else if (conv_use_n64(Mc, Cx))
        hipLaunchKernelGGL((conv3x3_bf16_kernel<false, 64>), dim3(conv_grid_n64(Mc, Cx)), dim3(kGemmThreads), 0,
                           (hipStream_t)stream, g);
      else
        hipLaunchKernelGGL((conv3x3_bf16_kernel<false, 128>), dim3(gemm_grid_size(Mc, Cx)), dim3(kGemmThreads), 0,
                           (hipStream_t)stream, g);
    }
  return launch_status();
}

int snipper_temporal_mix(void *stream, const void *in, int in_dtype, const unsigned char *mask, int mask_on_input,
                         const float *mix, int N, int Ti, int To, long long S, int C, void *out, int out_dtype) {
  return snipper_temporal_mix_ex(stream, in, in_dtype, mask, mask_on_input, mix, N, Ti, To, S, C, out, out_dtype, 0, 0, 0);
}

int snipper_temporal_mix_ex(void *stream, const void *in, int in_dtype, const unsigned char *mask, int mask_on_input,
                            const float *mix, int N, int Ti, int To, long long S, int C, void *out, int out_dtype,
                            int head_dim, int in_head_major, int out_head_major) {
  if (!in || !out || !mix) return SNIPPER_E_NULL;
  if (N <= 0 || Ti <= 0 || To <= 0 || Ti > kMixMaxFrames || To > kMixMaxFrames || S <= 0 || C <= 0 || C % 4)
    return SNIPPER_E_SHAPE;
  const bool hm = in_head_major || out_head_major;
  if (hm && (head_dim <= 0 || head_dim % 8 || C % head_dim || C % 8 || Ti > 4 || To > 4 ||
             (((uintptr_t)in | (uintptr_t)out) & 15)))
    return SNIPPER_E_SHAPE;                 // (head-major: 8 channels per lane within one head row, <= 4 frames)
  MixMatrix m{};
  for (int a = 0; a < To; ++a)
    for (int b = 0; b < Ti; ++b) m.w[a][b] = mix[a * Ti + b];
  if ((long long)N * S * C / 4 >= (1LL << 31) || S >= (1LL << 31)) return SNIPPER_E_SHAPE;
  const bool few = Ti <= 4 && To <= 4;      // (more frames: 4 channels per lane, or the registers would not hold them)
  const bool wide = few && C % 8 == 0 && (((uintptr_t)in | (uintptr_t)out) & 15) == 0;
  const long long total = (long long)N * S * C / (wide ? 8 : 4);
  const unsigned grid = (unsigned)((total + 255) / 256);
  hipStream_t st = (hipStream_t)stream;
#define SNIPPER_MIX_FW(TI, TO, MI, F, W)                                                                          \
  hipLaunchKernelGGL((temporal_mix_kernel<TI, TO, MI, F, W>), dim3(grid), dim3(256), 0, st, (const TI *)in, mask, m, N, \
                     Ti, To, (int)S, C, (TO *)out)
#define SNIPPER_MIX_M(TI, TO, MI)                                                                                 \
  do {                                                                                                            \
    if (hm) hipLaunchKernelGGL((temporal_mix_kernel<TI, TO, MI, 4, 8, true>), dim3(grid), dim3(256), 0, st,       \
                               (const TI *)in, mask, m, N, Ti, To, (int)S, C, (TO *)out, head_dim,                \
                               in_head_major ? 1 : 0, out_head_major ? 1 : 0);                                    \
    else if (wide) SNIPPER_MIX_FW(TI, TO, MI, 4, 8);                                                              \
    else if (few) SNIPPER_MIX_FW(TI, TO, MI, 4, 4);                                                               \
    else SNIPPER_MIX_FW(TI, TO, MI, 8, 4);                                                                        \
  } while (0)
#define SNIPPER_MIX(TI, TO)                                                                                       \
  do {                                                                                                            \
    if (mask_on_input) SNIPPER_MIX_M(TI, TO, true);                                                               \
    else SNIPPER_MIX_M(TI, TO, false);                                                                            \
  } while (0)
  if (in_dtype == 0 && out_dtype == 0) SNIPPER_MIX(float, float);
  else if (in_dtype == 1 && out_dtype == 0) SNIPPER_MIX(uint16_t, float);
  else if (in_dtype == 0 && out_dtype == 1) SNIPPER_MIX(float, uint16_t);
  else if (in_dtype == 1 && out_dtype == 1) SNIPPER_MIX(uint16_t, uint16_t);
  else return SNIPPER_E_UNSUPPORTED;
#undef SNIPPER_MIX_FW
#undef SNIPPER_MIX_M
#undef SNIPPER_MIX
  return launch_status();
}

static int make_level_scale(const float *inv_w, const float *inv_h, int L, LevelScale *sc) {
  if (!inv_w || !inv_h) return SNIPPER_E_NULL;
  for (int l = 0; l < L; ++l) { sc->inv_w[l] = inv_w[l]; sc->inv_h[l] = inv_h[l]; }
  return SNIPPER_OK;
}

namespace {
// 4-element vector accesses need LP % 4 == 0 and slices that start and advance on 4-element boundaries
inline bool prologue_vec_ok(const void *a, long long a_ld, const void *b, long long b_ld, int LP, int dtype) {
  const uintptr_t align = dtype == 0 ? 16 : 8;
  return LP % 4 == 0 && a_ld % 4 == 0 && b_ld % 4 == 0 && ((uintptr_t)a % align) == 0 && ((uintptr_t)b % align) == 0;
}
}  // namespace

int snipper_msda_prologue_forward(void *stream, const void *off, long long off_ld, const void *logit, long long logit_ld,
                                  int dtype, const float *ref, const float *inv_w, const float *inv_h, long long rows,
                                  int M, int L, int P, float *loc, float *prob) {
  return snipper_msda_prologue_forward_ex(stream, off, off_ld, logit, logit_ld, dtype, nullptr, ref, inv_w, inv_h, rows, M, L, P,
                                          loc, prob);
}

int snipper_msda_prologue_forward_ex(void *stream, const void *off, long long off_ld, const void *logit, long long logit_ld,
                                     int dtype, const float *off_bias, const float *ref, const float *inv_w,
                                     const float *inv_h, long long rows, int M, int L, int P, float *loc, float *prob) {
  if (!off || !logit || !ref || !loc || !prob) return SNIPPER_E_NULL;
  if (rows <= 0 || M <= 0 || L <= 0 || P <= 0 || L > kPrologueMaxL || L * P > kPrologueMaxLP) return SNIPPER_E_SHAPE;
  if (off_ld < (long long)M * L * P * 2 || logit_ld < (long long)M * L * P) return SNIPPER_E_SHAPE;
  LevelScale sc{};
  if (int rc = make_level_scale(inv_w, inv_h, L, &sc)) return rc;
  const dim3 grid((unsigned)((rows + 255) / 256));
  const bool vec = prologue_vec_ok(off, off_ld, logit, logit_ld, L * P, dtype) && ((uintptr_t)loc % 16) == 0 &&
                   ((uintptr_t)prob % 16) == 0;
#define SNIPPER_PROLOGUE_FWD(T, V, CL, CP)                                                                      \
  hipLaunchKernelGGL((prologue_fwd_kernel<T, V, CL, CP>), grid, dim3(256), 0, (hipStream_t)stream, (const T *)off, \
                     off_ld, (const T *)logit, logit_ld, ref, sc, rows, M, L, P, loc, prob, off_bias)
#define SNIPPER_PROLOGUE_FWD_T(T)                                            \
  do {                                                                       \
    if (vec && L == 3 && P == 4) SNIPPER_PROLOGUE_FWD(T, true, 3, 4);        \
    else if (vec && L == 4 && P == 4) SNIPPER_PROLOGUE_FWD(T, true, 4, 4);   \
    else SNIPPER_PROLOGUE_FWD(T, false, 0, 0);                               \
  } while (0)
  if (dtype == 0) SNIPPER_PROLOGUE_FWD_T(float);
  else if (dtype == 1) SNIPPER_PROLOGUE_FWD_T(uint16_t);
  else return SNIPPER_E_UNSUPPORTED;
#undef SNIPPER_PROLOGUE_FWD_T
#undef SNIPPER_PROLOGUE_FWD
  return launch_status();
}

int snipper_msda_prologue_backward(void *stream, const float *grad_loc, const float *grad_prob, const float *prob,
                                   const float *inv_w, const float *inv_h, long long rows, int M, int L, int P,
                                   void *grad_off, long long grad_off_ld, void *grad_logit, long long grad_logit_ld,
                                   int dtype, float *grad_ref) {
  if (!grad_loc || !grad_prob || !prob || !grad_off || !grad_logit) return SNIPPER_E_NULL;
  if (rows <= 0 || M <= 0 || L <= 0 || P <= 0 || L > kPrologueMaxL || L * P > kPrologueMaxLP) return SNIPPER_E_SHAPE;
  if (grad_off_ld < (long long)M * L * P * 2 || grad_logit_ld < (long long)M * L * P) return SNIPPER_E_SHAPE;
  if (grad_ref && (M > 64 || (M & (M - 1)))) return SNIPPER_E_SHAPE;
  LevelScale sc{};
  if (int rc = make_level_scale(inv_w, inv_h, L, &sc)) return rc;
  const dim3 grid((unsigned)((rows + 255) / 256));
  const bool vec = prologue_vec_ok(grad_off, grad_off_ld, grad_logit, grad_logit_ld, L * P, dtype) &&
                   (((uintptr_t)grad_loc | (uintptr_t)grad_prob | (uintptr_t)prob) % 16) == 0;
#define SNIPPER_PROLOGUE_BWD(T, V, CL, CP)                                                                        \
  hipLaunchKernelGGL((prologue_bwd_kernel<T, V, CL, CP>), grid, dim3(256), 0, (hipStream_t)stream, grad_loc,       \
                     grad_prob, prob, sc, rows, M, L, P, (T *)grad_off, grad_off_ld, (T *)grad_logit, grad_logit_ld, \
                     grad_ref)
#define SNIPPER_PROLOGUE_BWD_T(T)                                            \
  do {                                                                       \
    if (vec && L == 3 && P == 4) SNIPPER_PROLOGUE_BWD(T, true, 3, 4);        \
    else if (vec && L == 4 && P == 4) SNIPPER_PROLOGUE_BWD(T, true, 4, 4);   \
    else SNIPPER_PROLOGUE_BWD(T, false, 0, 0);                               \
  } while (0)
  if (dtype == 0) SNIPPER_PROLOGUE_BWD_T(float);
  else if (dtype == 1) SNIPPER_PROLOGUE_BWD_T(uint16_t);
  else return SNIPPER_E_UNSUPPORTED;
#undef SNIPPER_PROLOGUE_BWD_T
#undef SNIPPER_PROLOGUE_BWD
  return launch_status();
}

// ---- the whole tied spatiotemporal module core in one call ("the fused entry") ------------------
// forward : temporal mean of the neighbouring value frames (+ padding mask) -> sampling locations / probabilities ->
//           core op with the T1 query frames folded into the batch.
// backward: core op backward -> adjoint of locations / softmax -> adjoint of the temporal mean.
// Pure compositions of the entry points above on one stream; the caller keeps vbar / loc / prob between the two.
namespace {
inline bool st_dims_ok(int N, int T1, int T2, int S, int M, int D, int L, int Lq, int P) {
  return N > 0 && T1 > 0 && T2 > 0 && T1 <= kMixMaxFrames && T2 <= kMixMaxFrames && S > 0 && M > 0 && D > 0 && L > 0 &&
         Lq > 0 && P > 0 && (long long)N * T1 < (1LL << 31);
}
inline void st_transpose_mix(const float *mix, int T1, int T2, float *out) {
  for (int a = 0; a < T1; ++a)
    for (int b = 0; b < T2; ++b) out[b * T1 + a] = mix[a * T2 + b];
}
}  // namespace

int snipper_st_msda_forward(void *stream, const void *value, int value_dtype, const unsigned char *mask, const float *mix,
                            const void *off, long long off_ld, const void *logit, long long logit_ld, int ql_dtype,
                            const float *ref, const float *inv_w, const float *inv_h,
                            const int64_t *shapes, const int64_t *level_start, const int64_t *host_shapes,
                            int N, int T1, int T2, int S, int M, int D, int L, int Lq, int P,
                            float *vbar, float *loc, float *prob, void *out, int out_bf16) {
  if (!value || !mix || !off || !logit || !ref || !shapes || !level_start || !vbar || !loc || !prob || !out)
    return SNIPPER_E_NULL;
  if (!st_dims_ok(N, T1, T2, S, M, D, L, Lq, P)) return SNIPPER_E_SHAPE;
  if (int rc = snipper_temporal_mix(stream, value, value_dtype, mask, 1, mix, N, T2, T1, S, M * D, vbar, 0)) return rc;
  const long long rows = (long long)N * T1 * Lq * M;
  if (int rc = snipper_msda_prologue_forward(stream, off, off_ld, logit, logit_ld, ql_dtype, ref, inv_w, inv_h, rows, M, L,
                                             P, loc, prob))
    return rc;
  return snipper_msda_forward_ex(stream, nullptr, host_shapes, vbar, 0, shapes, level_start, loc, prob, N * T1, S, M, D, L, Lq, P,
                                 out, out_bf16 ? 1 : 0);
}

size_t snipper_st_msda_backward_workspace_bytes(int N, int T1, int S, int M, int D, int L, int Lq, int P,
                                                const int64_t *host_shapes) {
  const size_t gv = (size_t)N * T1 * S * M * D * sizeof(float);
  const size_t gl = (size_t)N * T1 * Lq * M * L * P * 3 * sizeof(float);
  const long long owner = host_shapes ? snipper_msda_backward_ex_workspace_bytes(nullptr, host_shapes, 0, N * T1, S, M, D, L, Lq, P) : 0;
  return gv + gl + (size_t)((owner + 15) / 16 * 16);
}

int snipper_st_msda_backward(void *stream, const void *grad_out, int grad_out_bf16, const float *vbar, const float *loc,
                             const float *prob, const unsigned char *mask, const float *mix,
                             const float *inv_w, const float *inv_h, const int64_t *shapes, const int64_t *level_start,
                             const int64_t *host_shapes, int N, int T1, int T2, int S, int M, int D, int L, int Lq, int P,
                             void *workspace, size_t workspace_bytes,
                             void *grad_value, int value_dtype, void *grad_off, long long grad_off_ld,
                             void *grad_logit, long long grad_logit_ld, int ql_dtype, float *grad_ref) {
  if (!grad_out || !vbar || !loc || !prob || !mix || !shapes || !level_start || !workspace || !grad_value || !grad_off ||
      !grad_logit)
    return SNIPPER_E_NULL;
  if (!st_dims_ok(N, T1, T2, S, M, D, L, Lq, P)) return SNIPPER_E_SHAPE;
  if (workspace_bytes < snipper_st_msda_backward_workspace_bytes(N, T1, S, M, D, L, Lq, P, host_shapes)) return SNIPPER_E_SHAPE;
  const size_t gv_bytes = (size_t)N * T1 * S * M * D * sizeof(float);
  const size_t rows_lp = (size_t)N * T1 * Lq * M * L * P;
  float *g_vbar = (float *)workspace;
  float *g_loc = (float *)((char *)workspace + gv_bytes);
  float *g_prob = g_loc + 2 * rows_lp;
  void *owner_ws = (char *)workspace + gv_bytes + rows_lp * 3 * sizeof(float);
  const long long owner_bytes = host_shapes ? snipper_msda_backward_ex_workspace_bytes(nullptr, host_shapes, 0, N * T1, S, M, D, L, Lq, P) : 0;
  int rc = snipper_msda_backward_ex(stream, nullptr, owner_bytes > 0 ? host_shapes : nullptr, owner_bytes > 0 ? owner_ws : nullptr,
                                    owner_bytes, grad_out, grad_out_bf16 ? 1 : 0, vbar, 0, shapes, level_start, loc, prob,
                                    N * T1, S, M, D, L, Lq, P, g_vbar, g_loc, g_prob);
  if (rc) return rc;
  const long long rows = (long long)N * T1 * Lq * M;
  rc = snipper_msda_prologue_backward(stream, g_loc, g_prob, prob, inv_w, inv_h, rows, M, L, P, grad_off, grad_off_ld,
                                      grad_logit, grad_logit_ld, ql_dtype, grad_ref);
  if (rc) return rc;
  float mix_t[kMixMaxFrames * kMixMaxFrames];
  st_transpose_mix(mix, T1, T2, mix_t);
  return snipper_temporal_mix(stream, g_vbar, 0, mask, 0, mix_t, N, T1, T2, S, M * D, grad_value, value_dtype);
}

int snipper_pair_losses_forward(void *stream, const float *sk, const float *sd, const float *tk, const float *td,
                                const float *cont_w, const float *max_depth, int n_layers, int pairs, int T, int K,
                                float eps, float *out) {
  if (!sk || !sd || !tk || !td || !cont_w || !max_depth || !out) return SNIPPER_E_NULL;
  if (n_layers <= 0 || pairs < 0 || T <= 0 || K <= 0 || T * K > kPairMaxTK) return SNIPPER_E_SHAPE;
  if (pairs == 0) return SNIPPER_OK;
  PairLossArgs a{};
  a.sk = sk; a.sd = sd; a.tk = tk; a.td = td; a.cont_w = cont_w; a.max_depth = max_depth; a.out = out;
  a.P = n_layers * pairs; a.T = T; a.K = K; a.pairs_per_layer = pairs; a.eps = eps;
  hipLaunchKernelGGL(pair_losses_fwd_kernel, dim3((a.P + 3) / 4), dim3(256), 0, (hipStream_t)stream, a);
  return launch_status();
}

int snipper_pair_losses_backward(void *stream, const float *sk, const float *sd, const float *tk, const float *td,
                                 const float *cont_w, const float *max_depth, const float *grad_terms,
                                 int n_layers, int pairs, int T, int K, float eps, float *grad_sk, float *grad_sd) {
  if (!sk || !sd || !tk || !td || !cont_w || !max_depth || !grad_terms || !grad_sk || !grad_sd) return SNIPPER_E_NULL;
  if (n_layers <= 0 || pairs < 0 || T <= 0 || K <= 0 || T * K > kPairMaxTK) return SNIPPER_E_SHAPE;
  if (pairs == 0) return SNIPPER_OK;
  PairLossArgs a{};
  a.sk = sk; a.sd = sd; a.tk = tk; a.td = td; a.cont_w = cont_w; a.max_depth = max_depth; a.gw = grad_terms;
  a.dsk = grad_sk; a.dsd = grad_sd;
  a.P = n_layers * pairs; a.T = T; a.K = K; a.pairs_per_layer = pairs; a.eps = eps;
  hipLaunchKernelGGL(pair_losses_bwd_kernel, dim3((a.P + 3) / 4), dim3(256), 0, (hipStream_t)stream, a);
  return launch_status();
}

int snipper_match_cost_f32(void *stream, const float *kpts, long long kp_sl, long long kp_sq, int kp_sk, const float *depth,
                           long long d_sl, long long d_sq, int d_sk, const float *logits, long long lg_sl, long long lg_sq,
                           const float *tgt_kpts, const float *tgt_depth, const float *max_depth,
                           int L, int Q, int M, int T, int K, const float *weights7, float eps, float *out) {
  if (!kpts || !depth || !logits || !tgt_kpts || !tgt_depth || !max_depth || !weights7 || !out) return SNIPPER_E_NULL;
  if (L <= 0 || Q <= 0 || M <= 0 || T <= 0 || K < 2 || kp_sk < 3 || d_sk < 1) return SNIPPER_E_SHAPE;
  const MatchCostArgs a{kpts, kp_sl, kp_sq, kp_sk, depth, d_sl, d_sq, d_sk, logits, lg_sl, lg_sq, tgt_kpts, tgt_depth, max_depth,
                        L, Q, M, T, K, weights7[0], weights7[1], weights7[2], weights7[3], weights7[4], weights7[5],
                        weights7[6], eps, out};
  const long long total = (long long)L * Q * M;
  hipLaunchKernelGGL(match_cost_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  return launch_status();
}

int snipper_refine_reference_f32(void *stream, const float *delta, long long ld_delta, const float *ref,
                                 const float *valid_ratios, int rows, int rows_per_batch, int L, float eps,
                                 float *new_ref, float *ref_in) {
  if (!delta || !ref || !valid_ratios || !new_ref || !ref_in) return SNIPPER_E_NULL;
  if (rows <= 0 || rows_per_batch <= 0 || rows % rows_per_batch || L <= 0 || ld_delta < 2) return SNIPPER_E_SHAPE;
  hipLaunchKernelGGL(refine_reference_kernel, dim3((unsigned)((rows * 2 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     delta, ld_delta, ref, valid_ratios, rows, rows_per_batch, L, eps, new_ref, ref_in);
  return launch_status();
}

int snipper_lsap_f32(void *stream, const float *cost, int P, int n, int m, long long *out_src, long long *out_tgt) {
  if (!cost || !out_src || !out_tgt) return SNIPPER_E_NULL;
  if (P <= 0 || m <= 0 || n < m || n > kLsapMaxCols || m > kLsapMaxRows) return SNIPPER_E_SHAPE;
  hipLaunchKernelGGL(lsap_kernel, dim3(P), dim3(64), 0, (hipStream_t)stream, cost, n, m, out_src, out_tgt);
  return launch_status();
}

}  // extern "C"

// ================================================================================================================
// One native call per decoder layer and direction (include/snipper_layers.h).  Pure sequencing: every launch below is an
// entry point of this library with the arguments the Python host used to pass one by one (snipper_amd/deformable_transformer.py
// DeformableTransformerDecoderLayer.forward_chain), in the same order -- results are bit-identical to that sequence.
// ================================================================================================================
namespace {
struct DlArena {            // byte offsets into the forward's arena (all 256-byte aligned)
  size_t qk, v, att, P, mixed, s2, st2, keep2, t2, t2q, raw, sampled, attended, s1, st1, keep1, t3, h, y, s3, st3, keep3, total;
};
struct DlScratch {          // byte offsets into the backward's scratch
  size_t d_y, gh, d_t3a, d_t3b, d_att_out, d_sampled, g_loc, g_prob, g_raw, d_t2b, d_mixed, d_att, dqk, dv, total;
};
inline size_t dl_up(size_t x) { return (x + 255) & ~(size_t)255; }
inline bool dl_dims_ok(const snipper_decoder_layer_dims *d) {
  if (!d || d->struct_bytes != (int32_t)sizeof(snipper_decoder_layer_dims)) return false;
  if (d->bs <= 0 || d->tokens <= 0 || d->frames <= 0 || d->queries <= 0 || d->frames * d->queries != d->tokens) return false;
  if (d->C <= 0 || d->C % 4 || d->C > kLnMaxIter * 256 || d->heads <= 0 || d->C % d->heads) return false;
  const int hd = d->C / d->heads;
  if ((hd != 32 && hd != 48) || d->tokens > kSaMaxL2 || (long long)d->bs * d->tokens > kSmallLnMaxRows) return false;
  if (d->d_ffn <= 0 || d->d_ffn % 4 || d->levels <= 0 || d->levels > 8 || d->points <= 0 || d->levels * d->points > 16) return false;
  if ((d->heads & (d->heads - 1)) || d->heads > 64 || d->S <= 0) return false;
  if ((d->heads * d->levels * d->points * 2) % 32) return false;       // (the pair's stacked-weight product splits there)
  if (d->value_dtype != 0 && d->value_dtype != 1) return false;
  if (d->value_dtype == 1 &&
      (hd != kSpD || d->queries > kSpMaxLq || d->queries * d->points * 4 > kSpTaps || d->S >= (1 << 22) ||
       (long long)d->bs * d->frames * d->S * d->C * 2 >= (1LL << 31)))
    return false;                                                        // (bf16 value: the sort-by-pixel backward's shape)
  for (float p : {d->p_attn, d->p_norm2, d->p_norm1, d->p_ffn, d->p_norm3})
    if (!(p >= 0.f && p < 1.f)) return false;
  return true;
}
inline DlArena dl_arena(const snipper_decoder_layer_dims &d) {
  const size_t R = (size_t)d.bs * d.tokens, C = d.C, f = sizeof(float);
  const size_t nraw = (size_t)d.heads * d.levels * d.points * 3;
  DlArena a{};
  size_t o = 0;
  auto take = [&](size_t bytes) { const size_t at = o; o += dl_up(bytes); return at; };
  a.qk = take(R * 2 * C * f); a.v = take(R * C * f); a.att = take(R * C * f);
  a.P = take((size_t)d.bs * d.heads * d.tokens * d.tokens * f);
  a.mixed = take(R * C * f);
  a.s2 = take(R * C * f); a.st2 = take(2 * R * f); a.keep2 = take(R * C / 4);
  a.t2 = take(R * C * f); a.t2q = take(R * C * f);
  a.raw = take(R * nraw * f); a.sampled = take(R * C * f); a.attended = take(R * C * f);
  a.s1 = take(R * C * f); a.st1 = take(2 * R * f); a.keep1 = take(R * C / 4);
  a.t3 = take(R * C * f); a.h = take(R * (size_t)d.d_ffn * f); a.y = take(R * C * f);
  a.s3 = take(R * C * f); a.st3 = take(2 * R * f); a.keep3 = take(R * C / 4);
  a.total = o;
  return a;
}
inline DlScratch dl_scratch(const snipper_decoder_layer_dims &d) {
  const size_t R = (size_t)d.bs * d.tokens, C = d.C, f = sizeof(float);
  const size_t nlp = (size_t)d.heads * d.levels * d.points;
  DlScratch s{};
  size_t o = 0;
  auto take = [&](size_t bytes) { const size_t at = o; o += dl_up(bytes); return at; };
  s.d_y = take(R * C * f); s.gh = take(R * (size_t)d.d_ffn * f); s.d_t3a = take(R * C * f); s.d_t3b = take(R * C * f);
  s.d_att_out = take(R * C * f); s.d_sampled = take(R * C * f);
  s.g_loc = take(R * nlp * 2 * f); s.g_prob = take(R * nlp * f); s.g_raw = take(R * nlp * 3 * f);
  s.d_t2b = take(R * C * f); s.d_mixed = take(R * C * f); s.d_att = take(R * C * f);
  s.dqk = take(R * 2 * C * f); s.dv = take(R * C * f);
  s.total = o;
  return s;
}
inline snipper_small_gemm dl_prob(const float *A, long long lda, int a_tr, const float *B, long long ldb, int b_tr, float *out,
                                  long long ldo, const float *bias, float *colsum, int I, int J, int Rr) {
  snipper_small_gemm g{};
  g.A = A; g.lda = lda; g.a_transposed = a_tr; g.B = B; g.ldb = ldb; g.b_transposed = b_tr; g.out = out; g.ldo = ldo;
  g.bias = bias; g.colsum = colsum; g.I = I; g.J = J; g.R = Rr;
  return g;
}
}  // namespace

extern "C" {

int snipper_layers_abi_version(void) { return SNIPPER_LAYERS_ABI_VERSION; }
int snipper_decoder_layer_supported(const snipper_decoder_layer_dims *d) { return dl_dims_ok(d) ? 1 : 0; }
size_t snipper_decoder_layer_arena_bytes(const snipper_decoder_layer_dims *d) { return dl_dims_ok(d) ? dl_arena(*d).total : 0; }
size_t snipper_decoder_layer_scratch_bytes(const snipper_decoder_layer_dims *d) { return dl_dims_ok(d) ? dl_scratch(*d).total : 0; }

int snipper_decoder_layer_forward(void *stream, const snipper_decoder_layer_fwd *a) {
  if (!a) return SNIPPER_E_NULL;
  const snipper_decoder_layer_dims &d = a->d;
  if (!dl_dims_ok(&d)) return SNIPPER_E_SHAPE;
  const snipper_decoder_layer_params &w = a->w;
  if (!a->x_v || !a->x_res || !a->x_q || !a->pos_a || !a->value || !a->shapes || !a->level_start || !a->ref_in || !a->inv_w ||
      !a->inv_h || !a->out || !a->loc || !a->prob || !a->arena || (a->out_q && !a->pos_b))
    return SNIPPER_E_NULL;
  for (const float *p : {w.so_w, w.so_b, w.aw_w, w.aw_b, w.op_w, w.op_b, w.norm1_w, w.norm1_b, w.in_proj_w, w.in_proj_b, w.out_proj_w,
                         w.out_proj_b, w.norm2_w, w.norm2_b, w.lin1_w, w.lin1_b, w.lin2_w, w.lin2_b, w.norm3_w, w.norm3_b})
    if (!p) return SNIPPER_E_NULL;
  if (a->root_w && (!a->ref_points || !a->valid_ratios || !a->new_ref || !a->ref_in_next)) return SNIPPER_E_NULL;
  const DlArena L = dl_arena(d);
  if (a->arena_bytes < L.total || ((uintptr_t)a->arena & 255)) return SNIPPER_E_SHAPE;
  unsigned char *ar = (unsigned char *)a->arena;
  auto F = [&](size_t off) { return (float *)(ar + off); };
  const int R = d.bs * d.tokens, C = d.C, E = d.C, hd = d.C / d.heads, nlp = d.heads * d.levels * d.points;
  int rc;
  // 1. packed q | k and v projections of the self-attention (reference :282-287 through nn.MultiheadAttention's in_proj)
  {
    snipper_small_gemm pr[2] = {dl_prob(a->x_q, E, 0, w.in_proj_w, E, 1, F(L.qk), 2 * E, w.in_proj_b, nullptr, R, 2 * E, E),
                                dl_prob(a->x_v, E, 0, w.in_proj_w + (size_t)2 * E * E, E, 1, F(L.v), E, w.in_proj_b + 2 * E, nullptr, R, E, E)};
    if ((rc = snipper_small_gemm_batch_f32(stream, pr, 2))) return rc;
  }
  // 2. attention, 3. its output projection, 4. norm2(x + dropout(.)) and the position-added copy for the cross attention's query
  if ((rc = snipper_small_attention_forward_f32(stream, F(L.qk), 2 * E, (long long)d.tokens * 2 * E, F(L.qk) + E, 2 * E,
                                                (long long)d.tokens * 2 * E, F(L.v), E, (long long)d.tokens * E, F(L.att), E,
                                                (long long)d.tokens * E, F(L.P), d.bs, d.heads, d.tokens, hd, d.attn_scale,
                                                d.p_attn, d.seed_attn)))
    return rc;
  if ((rc = snipper_small_linear_forward_f32(stream, F(L.att), E, w.out_proj_w, E, w.out_proj_b, R, E, E, F(L.mixed), E))) return rc;
  if ((rc = snipper_small_ln_forward_f32(stream, a->x_res, F(L.mixed), a->pos_a, w.norm2_w, w.norm2_b, R, C, d.p_norm2, d.eps_norm2,
                                         d.seed_norm2, F(L.s2), F(L.st2), F(L.st2) + R, ar + L.keep2, F(L.t2), F(L.t2q))))
    return rc;
  // 5. offsets | logits of the cross attention as one product pair, 6. locations + softmax, 7. the core op, 8. output projection
  {
    float *raw = F(L.raw);
    snipper_small_gemm pr[2] = {dl_prob(F(L.t2q), E, 0, w.so_w, E, 1, raw, 3 * nlp, w.so_b, nullptr, R, 2 * nlp, E),
                                dl_prob(F(L.t2q), E, 0, w.aw_w, E, 1, raw + 2 * nlp, 3 * nlp, w.aw_b, nullptr, R, nlp, E)};
    if ((rc = snipper_small_gemm_batch_f32(stream, pr, 2))) return rc;
    if ((rc = snipper_msda_prologue_forward_ex(stream, raw, 3 * nlp, raw + 2 * nlp, 3 * nlp, 0, nullptr, a->ref_in, a->inv_w, a->inv_h,
                                               (long long)R * d.heads, d.heads, d.levels, d.points, a->loc, a->prob)))
      return rc;
  }
  if ((rc = snipper_msda_forward_ex(stream, nullptr, a->host_shapes, a->value, d.value_dtype, a->shapes, a->level_start, a->loc, a->prob,
                                    d.bs * d.frames, d.S, d.heads, hd, d.levels, d.queries, d.points, F(L.sampled), 0)))
    return rc;
  if ((rc = snipper_small_linear_forward_f32(stream, F(L.sampled), E, w.op_w, E, w.op_b, R, E, E, F(L.attended), E))) return rc;
  // 9. norm1, 10-11. feed-forward block, 12. norm3 (+ the position-added copy for the next layer's q / k input)
  if ((rc = snipper_small_ln_forward_f32(stream, F(L.t2), F(L.attended), nullptr, w.norm1_w, w.norm1_b, R, C, d.p_norm1, d.eps_norm1,
                                         d.seed_norm1, F(L.s1), F(L.st1), F(L.st1) + R, ar + L.keep1, F(L.t3), nullptr)))
    return rc;
  {
    snipper_small_gemm p1 = dl_prob(F(L.t3), E, 0, w.lin1_w, E, 1, F(L.h), d.d_ffn, w.lin1_b, nullptr, R, d.d_ffn, E);
    p1.relu = 1; p1.dropout_p = d.p_ffn; p1.seed = d.seed_ffn;
    if ((rc = snipper_small_gemm_batch_f32(stream, &p1, 1))) return rc;
    snipper_small_gemm p2 = dl_prob(F(L.h), d.d_ffn, 0, w.lin2_w, d.d_ffn, 1, F(L.y), E, w.lin2_b, nullptr, R, E, d.d_ffn);
    if ((rc = snipper_small_gemm_batch_f32(stream, &p2, 1))) return rc;
  }
  if ((rc = snipper_small_ln_forward_f32(stream, F(L.t3), F(L.y), a->out_q ? a->pos_b : nullptr, w.norm3_w, w.norm3_b, R, C, d.p_norm3,
                                         d.eps_norm3, d.seed_norm3, F(L.s3), F(L.st3), F(L.st3) + R, ar + L.keep3, a->out, a->out_q)))
    return rc;
  // 13. reference-point refinement for the next layer (reference :329-333), no gradient
  if (a->root_w)
    if ((rc = snipper_refine_reference_linear_f32(stream, a->out, a->root_w, a->root_b, a->ref_points, a->valid_ratios, R, C, d.tokens,
                                                  d.levels, 1e-5f, a->new_ref, a->ref_in_next)))
      return rc;
  return SNIPPER_OK;
}

int snipper_decoder_layer_backward(void *stream, const snipper_decoder_layer_bwd *a) {
  if (!a) return SNIPPER_E_NULL;
  const snipper_decoder_layer_dims &d = a->d;
  if (!dl_dims_ok(&d)) return SNIPPER_E_SHAPE;
  const snipper_decoder_layer_params &w = a->w, &g = a->dw;
  if (!a->x_v || !a->x_q || !a->value || !a->shapes || !a->level_start || !a->inv_w || !a->inv_h || !a->loc || !a->prob || !a->arena ||
      !a->scratch || !a->d_xv || !a->d_xres || !a->d_xq || !a->d_pos_a || !a->d_value || (!a->g[0] && !a->g[1] && !a->g[2] && !a->g[3]))
    return SNIPPER_E_NULL;
  for (const float *p : {w.so_w, w.aw_w, w.op_w, w.norm1_w, w.in_proj_w, w.out_proj_w, w.norm2_w, w.lin1_w, w.lin2_w, w.norm3_w, g.so_w, g.so_b,
                         g.aw_w, g.aw_b, g.op_w, g.op_b, g.norm1_w, g.norm1_b, g.in_proj_w, g.in_proj_b, g.out_proj_w, g.out_proj_b, g.norm2_w,
                         g.norm2_b, g.lin1_w, g.lin1_b, g.lin2_w, g.lin2_b, g.norm3_w, g.norm3_b})
    if (!p) return SNIPPER_E_NULL;
  const DlArena L = dl_arena(d);
  const DlScratch S = dl_scratch(d);
  if (a->scratch_bytes < S.total || ((uintptr_t)a->scratch & 255) || ((uintptr_t)a->arena & 255)) return SNIPPER_E_SHAPE;
  const unsigned char *ar = (const unsigned char *)a->arena;
  unsigned char *sc = (unsigned char *)a->scratch;
  auto F = [&](size_t off) { return (const float *)(ar + off); };
  auto W = [&](size_t off) { return (float *)(sc + off); };
  const int R = d.bs * d.tokens, C = d.C, E = d.C, hd = d.C / d.heads, nlp = d.heads * d.levels * d.points;
  int rc;
  // norm3
  if ((rc = snipper_small_ln_backward_f32(stream, a->g[0], a->g[1], a->g[2], a->g[3], F(L.s3), F(L.st3), F(L.st3) + R, w.norm3_w,
                                          ar + L.keep3, R, C, d.p_norm3, W(S.d_t3b), W(S.d_y), g.norm3_w, g.norm3_b)))
    return rc;
  // feed-forward block: [dH (gated by the hidden activation), dW2 + db2], then [dX, dW1 + db1]
  {
    snipper_small_gemm pr[2] = {dl_prob(W(S.d_y), E, 0, w.lin2_w, d.d_ffn, 0, W(S.gh), d.d_ffn, nullptr, nullptr, R, d.d_ffn, E),
                                dl_prob(W(S.d_y), E, 1, F(L.h), d.d_ffn, 0, g.lin2_w, d.d_ffn, nullptr, g.lin2_b, E, d.d_ffn, R)};
    pr[0].gate = F(L.h); pr[0].ldgate = d.d_ffn; pr[0].gate_scale = 1.f / (1.f - d.p_ffn);
    if ((rc = snipper_small_gemm_batch_f32(stream, pr, 2))) return rc;
    snipper_small_gemm p2[2] = {dl_prob(W(S.gh), d.d_ffn, 0, w.lin1_w, E, 0, W(S.d_t3a), E, nullptr, nullptr, R, E, d.d_ffn),
                                dl_prob(W(S.gh), d.d_ffn, 1, F(L.t3), E, 0, g.lin1_w, E, nullptr, g.lin1_b, d.d_ffn, E, R)};
    if ((rc = snipper_small_gemm_batch_f32(stream, p2, 2))) return rc;
  }
  // norm1 (its result fed the feed-forward block and norm3's residual)
  if ((rc = snipper_small_ln_backward_f32(stream, W(S.d_t3a), W(S.d_t3b), nullptr, nullptr, F(L.s1), F(L.st1), F(L.st1) + R, w.norm1_w,
                                          ar + L.keep1, R, C, d.p_norm1, W(S.d_t2b), W(S.d_att_out), g.norm1_w, g.norm1_b)))
    return rc;
  // cross attention: output projection, core op, locations + softmax, offset | logit pair
  if ((rc = snipper_small_linear_backward_f32(stream, W(S.d_att_out), E, F(L.sampled), E, w.op_w, E, R, E, E, W(S.d_sampled), E, g.op_w, E,
                                              g.op_b)))
    return rc;
  if (d.value_dtype == 1) {
    if ((rc = snipper_msda_backward_sparse_f32rows(stream, W(S.d_sampled), (const uint16_t *)a->value, a->shapes, a->level_start, a->loc,
                                                   a->prob, d.bs * d.frames, d.S, d.heads, hd, d.levels, d.queries, d.points,
                                                   (uint16_t *)a->d_value, W(S.g_loc), W(S.g_prob))))
      return rc;
  } else {
    if ((rc = snipper_msda_backward_ex(stream, nullptr, a->host_shapes, nullptr, 0, W(S.d_sampled), 0, a->value, 0, a->shapes,
                                       a->level_start, a->loc, a->prob, d.bs * d.frames, d.S, d.heads, hd, d.levels, d.queries, d.points,
                                       a->d_value, W(S.g_loc), W(S.g_prob))))
      return rc;
  }
  {
    float *graw = W(S.g_raw);
    if ((rc = snipper_msda_prologue_backward(stream, W(S.g_loc), W(S.g_prob), a->prob, a->inv_w, a->inv_h, (long long)R * d.heads, d.heads,
                                             d.levels, d.points, graw, 3 * nlp, graw + 2 * nlp, 3 * nlp, 0, a->d_ref_in)))
      return rc;
    snipper_small_gemm pr[3] = {dl_prob(graw, 3 * nlp, 0, w.so_w, E, 0, a->d_pos_a, E, nullptr, nullptr, R, E, 3 * nlp),
                                dl_prob(graw, 3 * nlp, 1, F(L.t2q), E, 0, g.so_w, E, nullptr, g.so_b, 2 * nlp, E, R),
                                dl_prob(graw + 2 * nlp, 3 * nlp, 1, F(L.t2q), E, 0, g.aw_w, E, nullptr, g.aw_b, nlp, E, R)};
    pr[0].B2 = w.aw_w; pr[0].ldb2 = E; pr[0].r_split = 2 * nlp;
    if ((rc = snipper_small_gemm_batch_f32(stream, pr, 3))) return rc;
  }
  // norm2 (its result fed norm1's residual; its position-added copy the pair above: d_pos_a is that copy's gradient)
  if ((rc = snipper_small_ln_backward_f32(stream, W(S.d_t2b), a->d_pos_a, nullptr, nullptr, F(L.s2), F(L.st2), F(L.st2) + R, w.norm2_w,
                                          ar + L.keep2, R, C, d.p_norm2, a->d_xres, W(S.d_mixed), g.norm2_w, g.norm2_b)))
    return rc;
  // self-attention: output projection, attention, packed input projections
  if ((rc = snipper_small_linear_backward_f32(stream, W(S.d_mixed), E, F(L.att), E, w.out_proj_w, E, R, E, E, W(S.d_att), E, g.out_proj_w, E,
                                              g.out_proj_b)))
    return rc;
  if ((rc = snipper_small_attention_backward_f32(stream, F(L.qk), 2 * E, (long long)d.tokens * 2 * E, F(L.qk) + E, 2 * E,
                                                 (long long)d.tokens * 2 * E, F(L.v), E, (long long)d.tokens * E, F(L.att), E,
                                                 (long long)d.tokens * E, F(L.P), W(S.d_att), E, (long long)d.tokens * E, W(S.dqk), 2 * E,
                                                 (long long)d.tokens * 2 * E, W(S.dqk) + E, 2 * E, (long long)d.tokens * 2 * E, W(S.dv), E,
                                                 (long long)d.tokens * E, d.bs, d.heads, d.tokens, hd, d.attn_scale, d.p_attn,
                                                 d.seed_attn)))
    return rc;
  {
    snipper_small_gemm pr[4] = {dl_prob(W(S.dqk), 2 * E, 0, w.in_proj_w, E, 0, a->d_xq, E, nullptr, nullptr, R, E, 2 * E),
                                dl_prob(W(S.dqk), 2 * E, 1, a->x_q, E, 0, g.in_proj_w, E, nullptr, g.in_proj_b, 2 * E, E, R),
                                dl_prob(W(S.dv), E, 0, w.in_proj_w + (size_t)2 * E * E, E, 0, a->d_xv, E, nullptr, nullptr, R, E, E),
                                dl_prob(W(S.dv), E, 1, a->x_v, E, 0, g.in_proj_w + (size_t)2 * E * E, E, nullptr, g.in_proj_b + 2 * E, E, E, R)};
    if ((rc = snipper_small_gemm_batch_f32(stream, pr, 4))) return rc;
  }
  return SNIPPER_OK;
}

}  // extern "C"

// ================================================================================================================
// Encoder layer composites (include/snipper_layers.h): the launches of DeformableTransformerEncoderLayer.forward_fused
// (snipper_amd/deformable_transformer.py) and of its autograd nodes' backwards, in their order, from one call each.
// ================================================================================================================
namespace {
struct ElArena { size_t value, raw, loc, prob, vbar, samp, att, s1, st1, keep1, y16, h, z, keep2, total; };
struct ElScratch { size_t dz2, gh, dxf, dz1, dsamp, gv, gl, ga, gval, graw, ws_ln, ws_wg, ws_msda, total; size_t ln_bytes, wg_bytes; long long msda_bytes; };
inline bool el_dims_ok(const snipper_encoder_layer_dims *d) {
  if (!d || d->struct_bytes != (int32_t)sizeof(snipper_encoder_layer_dims)) return false;
  if (d->bs <= 0 || d->frames <= 0 || d->frames > 4 || d->S <= 0 || d->C <= 0 || d->C % 64 || d->heads <= 0 || d->C % d->heads) return false;
  if (d->C / d->heads != kD48 || d->d_ffn <= 0 || d->d_ffn % 128 || d->levels <= 0 || d->levels > 4 || d->points != 4) return false;
  if ((d->heads & (d->heads - 1)) || d->heads > 64) return false;
  const long long R = (long long)d->bs * d->frames * d->S;
  if (R < 8192 || R * d->d_ffn >= (1LL << 32) || R * d->C >= (1LL << 31)) return false;
  if (!snipper_linear_wres_supported((int)R, d->C, d->C) || !snipper_linear_wide_supported(R, d->C, d->d_ffn)) return false;
  for (float p : {d->p_norm1, d->p_ffn, d->p_norm2})
    if (!(p >= 0.f && p < 1.f)) return false;
  return true;
}
inline ElArena el_arena(const snipper_encoder_layer_dims &d) {
  const size_t R = (size_t)d.bs * d.frames * d.S, C = d.C, nlp = (size_t)d.heads * d.levels * d.points;
  ElArena a{};
  size_t o = 0;
  auto take = [&](size_t bytes) { const size_t at = o; o += dl_up(bytes); return at; };
  a.value = take(R * C * 2); a.raw = take(R * 3 * nlp * 2);
  a.loc = take(R * nlp * 2 * 4); a.prob = take(R * nlp * 4);
  a.vbar = take(R * C * 2); a.samp = take(R * C * 2); a.att = take(R * C * 2);
  a.s1 = take(R * C * 4); a.st1 = take(2 * R * 4); a.keep1 = take(R * C / 4);
  a.y16 = take(R * C * 2); a.h = take(R * (size_t)d.d_ffn * 2); a.z = take(R * C * 2); a.keep2 = take(R * C / 4);
  a.total = o;
  return a;
}
inline ElScratch el_scratch(const snipper_encoder_layer_dims &d, const snipper_msda_config *cfg, const int64_t *host_shapes) {
  const size_t R = (size_t)d.bs * d.frames * d.S, C = d.C, nlp = (size_t)d.heads * d.levels * d.points;
  ElScratch s{};
  size_t o = 0;
  auto take = [&](size_t bytes) { const size_t at = o; o += dl_up(bytes); return at; };
  s.dz2 = take(R * C * 2); s.gh = take(R * (size_t)d.d_ffn * 2); s.dxf = take(R * C * 2); s.dz1 = take(R * C * 2);
  s.dsamp = take(R * C * 2); s.gv = take(R * C * 4); s.gl = take(R * nlp * 2 * 4); s.ga = take(R * nlp * 4);
  s.gval = take(R * C * 2); s.graw = take(R * 3 * nlp * 2);
  s.ln_bytes = snipper_add_dropout_layernorm_workspace_bytes((int)R, (int)C);
  s.wg_bytes = std::max(std::max(snipper_wgrad_workspace_bytes((int)R, (int)C, (int)C), snipper_wgrad_workspace_bytes((int)R, (int)(3 * nlp), (int)C)),
                        std::max(snipper_wgrad_workspace_bytes((int)R, (int)C, d.d_ffn), snipper_wgrad_workspace_bytes((int)R, d.d_ffn, (int)C)));
  s.msda_bytes = host_shapes ? snipper_msda_backward_ex_workspace_bytes(cfg, host_shapes, 1, d.bs * d.frames, d.S, d.heads, d.C / d.heads,
                                                                        d.levels, d.S, d.points) : 0;
  s.ws_ln = take(s.ln_bytes); s.ws_wg = take(s.wg_bytes); s.ws_msda = take((size_t)std::max(0LL, s.msda_bytes));
  s.total = o;
  return s;
}
}  // namespace

extern "C" {

int snipper_encoder_layer_supported(const snipper_encoder_layer_dims *d) { return el_dims_ok(d) ? 1 : 0; }
size_t snipper_encoder_layer_arena_bytes(const snipper_encoder_layer_dims *d) { return el_dims_ok(d) ? el_arena(*d).total : 0; }
size_t snipper_encoder_layer_scratch_bytes(const snipper_encoder_layer_dims *d, const snipper_msda_config *cfg, const int64_t *host_shapes) {
  return el_dims_ok(d) ? el_scratch(*d, cfg, host_shapes).total : 0;
}

int snipper_encoder_layer_forward(void *stream, const snipper_encoder_layer_fwd *a) {
  if (!a) return SNIPPER_E_NULL;
  const snipper_encoder_layer_dims &d = a->d;
  if (!el_dims_ok(&d)) return SNIPPER_E_SHAPE;
  const snipper_encoder_layer_weights &w = a->w;
  if (!a->x32 || !a->src16 || !a->q16 || !a->ref || !a->shapes || !a->level_start || !a->inv_w || !a->inv_h || !a->mix || !a->s2 ||
      !a->mean2 || !a->rstd2 || !a->y16 || !a->arena || (d.last ? !a->y32 : (!a->pos16 || !a->yq16)))
    return SNIPPER_E_NULL;
  if (!w.wv || !w.wm || !w.wo || !w.w1 || !w.w2_packed || !w.bv || !w.bm || !w.so_bias || !w.bo || !w.b1 || !w.b2 || !w.norm1_w || !w.norm1_b ||
      !w.norm2_w || !w.norm2_b)
    return SNIPPER_E_NULL;
  const ElArena L = el_arena(d);
  if (a->arena_bytes < L.total || ((uintptr_t)a->arena & 255)) return SNIPPER_E_SHAPE;
  unsigned char *ar = (unsigned char *)a->arena;
  auto H = [&](size_t off) { return (uint16_t *)(ar + off); };
  auto F = [&](size_t off) { return (float *)(ar + off); };
  const int R = d.bs * d.frames * d.S, C = d.C, nlp = d.heads * d.levels * d.points, hd = d.C / d.heads;
  int rc;
  // value projection; merged offset | logit projection (the offsets' bias stays out: added in float32 by the prologue)
  if ((rc = snipper_linear_bf16(stream, a->src16, C, w.wv, w.bv, nullptr, 0, H(L.value), C, R, C, C, 0, 0.f, 0))) return rc;
  if ((rc = snipper_linear_bf16(stream, a->q16, C, w.wm, w.bm, nullptr, 0, H(L.raw), 3 * nlp, R, 3 * nlp, C, 0, 0.f, 0))) return rc;
  if ((rc = snipper_msda_prologue_forward_ex(stream, H(L.raw), 3 * nlp, H(L.raw) + 2 * nlp, 3 * nlp, 1, w.so_bias, a->ref, a->inv_w, a->inv_h,
                                             (long long)R * d.heads, d.heads, d.levels, d.points, F(L.loc), F(L.prob))))
    return rc;
  // temporal mean of the neighbouring value frames (bf16, head-major when the configuration says so), then the core op on it
  if ((rc = snipper_temporal_mix_ex(stream, H(L.value), 1, nullptr, 1, a->mix, d.bs, d.frames, d.frames, d.S, C, H(L.vbar), 1, hd, 0, d.head_major)))
    return rc;
  if ((rc = snipper_msda_forward_ex(stream, a->cfg, a->host_shapes, H(L.vbar), 1, a->shapes, a->level_start, F(L.loc), F(L.prob),
                                    d.bs * d.frames, d.S, d.heads, hd, d.levels, d.S, d.points, H(L.samp), 1)))
    return rc;
  if ((rc = snipper_linear_bf16(stream, H(L.samp), C, w.wo, w.bo, nullptr, 0, H(L.att), C, R, C, C, 0, 0.f, 0))) return rc;
  // norm1 (lazy float32 result: only its saved sum is written), feed-forward block, norm2
  if ((rc = snipper_add_dropout_layernorm_forward_ex(stream, a->x32, 0, a->x_mean, a->x_rstd, a->x_gamma, a->x_beta, H(L.att), 1, nullptr, 0,
                                                     w.norm1_w, w.norm1_b, R, C, d.p_norm1, d.eps_norm1, d.seed_norm1, F(L.s1), F(L.st1),
                                                     F(L.st1) + R, ar + L.keep1, nullptr, H(L.y16), nullptr)))
    return rc;
  if ((rc = snipper_linear_bf16(stream, H(L.y16), C, w.w1, w.b1, nullptr, 0, H(L.h), d.d_ffn, R, d.d_ffn, C, 1, d.p_ffn, d.seed_ffn))) return rc;
  if ((rc = snipper_linear_wide_bf16(stream, H(L.h), w.w2_packed, w.b2, H(L.z), R, C, d.d_ffn))) return rc;
  if ((rc = snipper_add_dropout_layernorm_forward_ex(stream, F(L.s1), 0, F(L.st1), F(L.st1) + R, w.norm1_w, w.norm1_b, H(L.z), 1, a->pos16, 1,
                                                     w.norm2_w, w.norm2_b, R, C, d.p_norm2, d.eps_norm2, d.seed_norm2, a->s2, a->mean2, a->rstd2,
                                                     ar + L.keep2, a->y32, a->y16, d.last ? nullptr : a->yq16)))
    return rc;
  return SNIPPER_OK;
}

int snipper_encoder_layer_backward(void *stream, const snipper_encoder_layer_bwd *a) {
  if (!a) return SNIPPER_E_NULL;
  const snipper_encoder_layer_dims &d = a->d;
  if (!el_dims_ok(&d)) return SNIPPER_E_SHAPE;
  const snipper_encoder_layer_weights &w = a->w;
  const snipper_encoder_layer_grads &g = a->dw;
  if ((!a->g32 && !a->g16 && !a->gq16) || !a->src16 || !a->q16 || !a->s2 || !a->mean2 || !a->rstd2 || !a->shapes || !a->level_start ||
      !a->host_shapes || !a->inv_w || !a->inv_h || !a->mix_t || !a->arena || !a->scratch || !a->d_x32 || !a->d_src16 || !a->d_q16)
    return SNIPPER_E_NULL;
  if (!w.wv_t || !w.wm_t || !w.wo_t || !w.w2_t || !w.w1_tpacked || !w.norm1_w || !w.norm2_w) return SNIPPER_E_NULL;
  for (const float *p : {g.wv, g.bv, g.wm, g.bm, g.wo, g.bo, g.w1, g.b1, g.w2, g.b2, g.norm1_w, g.norm1_b, g.norm2_w, g.norm2_b})
    if (!p) return SNIPPER_E_NULL;
  const ElArena L = el_arena(d);
  const ElScratch S = el_scratch(d, a->cfg, a->host_shapes);
  if (a->scratch_bytes < S.total || ((uintptr_t)a->scratch & 255) || ((uintptr_t)a->arena & 255) || S.msda_bytes <= 0) return SNIPPER_E_SHAPE;
  const unsigned char *ar = (const unsigned char *)a->arena;
  unsigned char *sc = (unsigned char *)a->scratch;
  auto H = [&](size_t off) { return (const uint16_t *)(ar + off); };
  auto F = [&](size_t off) { return (const float *)(ar + off); };
  auto WH = [&](size_t off) { return (uint16_t *)(sc + off); };
  auto WF = [&](size_t off) { return (float *)(sc + off); };
  const int R = d.bs * d.frames * d.S, C = d.C, nlp = d.heads * d.levels * d.points, hd = d.C / d.heads;
  int rc;
  // norm2: d(saved sum of norm1) stays in d_x32 for the moment (norm1's float32 result was lazy: its consumer's dx IS its g32)
  if ((rc = snipper_add_dropout_layernorm_backward(stream, a->g32, a->g16, a->gq16, a->s2, a->mean2, a->rstd2, w.norm2_w, ar + L.keep2, R, C,
                                                   d.p_norm2, a->d_x32, 0, WH(S.dz2), 1, g.norm2_w, g.norm2_b, sc + S.ws_ln, S.ln_bytes)))
    return rc;
  // feed-forward block (dense._BigFFN.backward): dW2, gated hidden gradient, dW1, input gradient
  if ((rc = snipper_wgrad_bf16(stream, WH(S.dz2), C, H(L.h), d.d_ffn, R, C, d.d_ffn, nullptr, g.w2, d.d_ffn, g.b2, 0, sc + S.ws_wg, S.wg_bytes))) return rc;
  if ((rc = snipper_linear_wres_bf16(stream, WH(S.dz2), C, w.w2_t, C, nullptr, H(L.h), d.d_ffn, 1.f / (1.f - d.p_ffn), WH(S.gh), d.d_ffn, R,
                                     d.d_ffn, C, 0, 0.f, 0)))
    return rc;
  if ((rc = snipper_wgrad_bf16(stream, WH(S.gh), d.d_ffn, H(L.y16), C, R, d.d_ffn, C, nullptr, g.w1, C, g.b1, 0, sc + S.ws_wg, S.wg_bytes))) return rc;
  if ((rc = snipper_linear_wide_bf16(stream, WH(S.gh), w.w1_tpacked, nullptr, WH(S.dxf), R, C, d.d_ffn))) return rc;
  // norm1: gradients of its (lazy) float32 result = norm2's dx, of its bf16 result = the feed-forward block's dx
  if ((rc = snipper_add_dropout_layernorm_backward(stream, a->d_x32, WH(S.dxf), nullptr, F(L.s1), F(L.st1), F(L.st1) + R, w.norm1_w, ar + L.keep1, R,
                                                   C, d.p_norm1, a->d_x32, 0, WH(S.dz1), 1, g.norm1_w, g.norm1_b, sc + S.ws_ln, S.ln_bytes)))
    return rc;
  // output projection (dense._BigLinear.backward: weight gradient, then data gradient on the weight-stationary kernel)
  if ((rc = snipper_wgrad_bf16(stream, WH(S.dz1), C, H(L.samp), C, R, C, C, nullptr, g.wo, C, g.bo, 0, sc + S.ws_wg, S.wg_bytes))) return rc;
  if ((rc = snipper_linear_wres_bf16(stream, WH(S.dz1), C, w.wo_t, C, nullptr, nullptr, 0, 1.f, WH(S.dsamp), C, R, C, C, 0, 0.f, 0))) return rc;
  // core op (owner-computes backward), transposed temporal mix
  if ((rc = snipper_msda_backward_ex(stream, a->cfg, a->host_shapes, sc + S.ws_msda, S.msda_bytes, WH(S.dsamp), 1, H(L.vbar), 1, a->shapes,
                                     a->level_start, F(L.loc), F(L.prob), d.bs * d.frames, d.S, d.heads, hd, d.levels, d.S, d.points, WF(S.gv),
                                     WF(S.gl), WF(S.ga))))
    return rc;
  if ((rc = snipper_temporal_mix_ex(stream, WF(S.gv), 0, nullptr, 0, a->mix_t, d.bs, d.frames, d.frames, d.S, C, WH(S.gval), 1, hd, d.head_major, 0)))
    return rc;
  // locations / softmax adjoint, merged offset | logit projection (data gradient first: dense._BigLinearPair.backward)
  if ((rc = snipper_msda_prologue_backward(stream, WF(S.gl), WF(S.ga), F(L.prob), a->inv_w, a->inv_h, (long long)R * d.heads, d.heads, d.levels,
                                           d.points, WH(S.graw), 3 * nlp, WH(S.graw) + 2 * nlp, 3 * nlp, 1, nullptr)))
    return rc;
  if ((rc = snipper_linear_wres_bf16(stream, WH(S.graw), 3 * nlp, w.wm_t, 3 * nlp, nullptr, nullptr, 0, 1.f, a->d_q16, C, R, C, 3 * nlp, 0, 0.f, 0)))
    return rc;
  if ((rc = snipper_wgrad_bf16(stream, WH(S.graw), 3 * nlp, a->q16, C, R, 3 * nlp, C, nullptr, g.wm, C, g.bm, 0, sc + S.ws_wg, S.wg_bytes))) return rc;
  // value projection
  if ((rc = snipper_wgrad_bf16(stream, WH(S.gval), C, a->src16, C, R, C, C, nullptr, g.wv, C, g.bv, 0, sc + S.ws_wg, S.wg_bytes))) return rc;
  if ((rc = snipper_linear_wres_bf16(stream, WH(S.gval), C, w.wv_t, C, nullptr, nullptr, 0, 1.f, a->d_src16, C, R, C, C, 0, 0.f, 0))) return rc;
  return SNIPPER_OK;
}

}  // extern "C"
