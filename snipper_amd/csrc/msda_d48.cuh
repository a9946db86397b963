// msda_d48.cuh -- kernels specialised for Snipper's head width D = 48
// (hidden_dim 384 / 8 heads), gfx950 only.
//
// Why a special path: 48 channels is 3/4 of a wave.  The reference runs this shape with a
// 48-thread block per (n,q,m) and a serial thread-0 reduction
// (/root/reference/models/ops/src/cuda/ms_deform_im2col_cuda.cuh:513-616, dispatch :1272-1295);
// on wave64 that idles a quarter of every wave and re-decodes each sampling point 48 times.
//
// Forward mapping (one "row" = one (n,q,m) = one 48-vector of the output):
//   * G lanes per row, each lane owning 12 contiguous BYTES of the row: f32 -> G=16 lanes x 3
//     channels, bf16 -> G=8 lanes x 6 channels.  A wave therefore covers 4 (f32) or 8 (bf16)
//     rows with every lane busy, and one tap of one sample is ONE buffer_load_dwordx3 whose
//     lanes read the 192-/96-byte head row contiguously.
//   * the row's L*P sampling points are decoded once by the first L*P lanes of its group
//     (pixel mapping, in-range test, bilinear weights already multiplied by the attention
//     weight, byte offsets of the four taps) and staged in LDS; the gather loop then reads one
//     32-byte record per sample (same address across the group -> LDS broadcast).
//   * taps outside the map get byte offset 0x80000000: the raw-buffer bounds check returns 0
//     for them without touching memory, so the gather loop has no branches and an out-of-map
//     tap contributes exactly 0 whatever `value` holds (reference semantics, .cuh:57-80).
//   * blockIdx is remapped so that each XCD (own 4 MiB L2) walks a contiguous band of rows:
//     encoder queries are raster-ordered, so a band samples a compact region of `value`.
//
// Backward mapping (f32): 16 lanes per row, lane i owns channels {i, i+16, i+32} so that every
// float atomic wave-instruction adds 64 contiguous bytes per row (MI355X_MICROARCH "Global float
// atomics": contiguous segments, never one lane per row).  The three per-sample scalars
// (grad_attn, grad_loc.x, grad_loc.y) are reduced over the 16 lanes with DPP row rotations
// (no LDS, no barriers), kept by lane s of the group, and written once per row, coalesced.
#pragma once
#include "msda_common.cuh"
#include "msda_generic.cuh"  // CoreDims

namespace snipper {

typedef unsigned int u32x3 __attribute__((ext_vector_type(3)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kD48 = 48;
constexpr int kD48Block = 256;
constexpr unsigned kOobOffset = 0x80000000u;  // >= num_records of any descriptor we build
constexpr int kMaxLevelsFast = 16;

struct LevelTable {  // staged per block in LDS
  int H[kMaxLevelsFast], W[kMaxLevelsFast], start[kMaxLevelsFast];
};

struct FwdRecord {   // 32 B, one per (row, sample)
  f32x4 w;           // bilinear weight x attention weight, per tap
  u32x4 off;         // byte offset of each tap's head row in `value`, or kOobOffset
};

__device__ __forceinline__ int xcd_band_block(int nblk_padded) {
  // blocks are dealt round-robin over the 8 XCDs (observed, speed only): give XCD j the
  // j-th contiguous eighth of the logical blocks.
  const int b = blockIdx.x;
  return (b & 7) * (nblk_padded >> 3) + (b >> 3);
}

// (D = 24 -- the reference's default hidden_dim 192 / 8 heads, main.py:88, the same shm_reduce branch as D = 48 in the
//  reference, .cuh:1272-1295 -- takes the same kernels with half the lanes per row: DT below)
template <typename VT> struct D48Fwd;
template <> struct D48Fwd<float> {
  static constexpr int G = 16, CPL = 3;
  static __device__ __forceinline__ void fma_tap(float (&acc)[3], float w, u32x3 v) {
    acc[0] = fmaf(w, __uint_as_float(v.x), acc[0]);
    acc[1] = fmaf(w, __uint_as_float(v.y), acc[1]);
    acc[2] = fmaf(w, __uint_as_float(v.z), acc[2]);
  }
  static __device__ __forceinline__ u32x3 pack(const float (&acc)[3]) {
    u32x3 r;
    r.x = __float_as_uint(acc[0]); r.y = __float_as_uint(acc[1]); r.z = __float_as_uint(acc[2]);
    return r;
  }
};
template <> struct D48Fwd<uint16_t> {
  static constexpr int G = 8, CPL = 6;
  static __device__ __forceinline__ void fma_tap(float (&acc)[6], float w, u32x3 v) {
    acc[0] = fmaf(w, __uint_as_float(v.x << 16), acc[0]);
    acc[1] = fmaf(w, __uint_as_float(v.x & 0xffff0000u), acc[1]);
    acc[2] = fmaf(w, __uint_as_float(v.y << 16), acc[2]);
    acc[3] = fmaf(w, __uint_as_float(v.y & 0xffff0000u), acc[3]);
    acc[4] = fmaf(w, __uint_as_float(v.z << 16), acc[4]);
    acc[5] = fmaf(w, __uint_as_float(v.z & 0xffff0000u), acc[5]);
  }
  static __device__ __forceinline__ u32x3 pack(const float (&acc)[6]) {
    u32x3 r;
    r.x = (unsigned)f32_to_bf16_bits(acc[0]) | ((unsigned)f32_to_bf16_bits(acc[1]) << 16);
    r.y = (unsigned)f32_to_bf16_bits(acc[2]) | ((unsigned)f32_to_bf16_bits(acc[3]) << 16);
    r.z = (unsigned)f32_to_bf16_bits(acc[4]) | ((unsigned)f32_to_bf16_bits(acc[5]) << 16);
    return r;
  }
};

__device__ __forceinline__ void stage_levels(LevelTable &t, const int64_t *shapes,
                                             const int64_t *level_start, int L) {
  if ((int)threadIdx.x < L) {
    t.H[threadIdx.x] = (int)shapes[2 * threadIdx.x];
    t.W[threadIdx.x] = (int)shapes[2 * threadIdx.x + 1];
    t.start[threadIdx.x] = (int)level_start[threadIdx.x];
  }
}

// LP_T = L*P when known at compile time (12 for Snipper's L=3,P=4), 0 = runtime.  DT = channels per head: 48 or 24.
template <typename VT, int LP_T, int DT = kD48>
__global__ __launch_bounds__(kD48Block) void msda_fwd_d48_kernel(
    const VT *__restrict__ value, const int64_t *__restrict__ shapes,
    const int64_t *__restrict__ level_start, const float *__restrict__ loc,
    const float *__restrict__ attn, CoreDims d, VT *__restrict__ out, int nblk_padded, int out_bf16) {
  // out_bf16 = "rows of the other type": float kernel -- `out` holds bfloat16 rows (the consumer, the output projection,
  // rounds to bf16 anyway: no cast pass, half the store traffic, same results); bf16 kernel -- `out` holds float32 rows
  using TR = D48Fwd<VT>;
  constexpr int G = TR::G * DT / kD48, CPL = TR::CPL, kRows = kD48Block / G;
  __shared__ LevelTable lv;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int LP = LP_T ? LP_T : d.L * d.P;
  const int rec_stride = LP * (int)sizeof(FwdRecord) + 16;   // +16 B: rows on distinct banks
  const int grp = threadIdx.x / G, lane = threadIdx.x % G;
  const long long total_rows = (long long)d.N * d.Lq * d.M;
  const long long row = (long long)xcd_band_block(nblk_padded) * kRows + grp;
  const bool live = row < total_rows;

  stage_levels(lv, shapes, level_start, d.L);
  __syncthreads();

  unsigned char *my_recs = smem_raw + (size_t)grp * rec_stride;
  if (live) {
    const int m = (int)(row % d.M);
    const long long n = row / ((long long)d.M * d.Lq);
    const unsigned row_bytes = DT * sizeof(VT);
    const unsigned px_stride = msda_px_stride(d, row_bytes);           // next pixel, same head
    const unsigned base = msda_row_base(d, (unsigned)n, (unsigned)m, row_bytes);
    for (int s = lane; s < LP; s += G) {
      const int l = s / d.P;
      const int H = lv.H[l], W = lv.W[l];
      const long long li = row * LP + s;
      const float lx = loc[2 * li], ly = loc[2 * li + 1], a_in = attn[li];
      const float y = ly * (float)H - 0.5f, x = lx * (float)W - 0.5f;
      const bool inside = (y > -1.f) && (x > -1.f) && (y < (float)H) && (x < (float)W);
      const float yf = floorf(y), xf = floorf(x);
      const int y0 = (int)yf, x0 = (int)xf;
      // a skipped sample gets all-zero weights (so a non-finite location cannot leak NaNs)
      const float lh = inside ? y - yf : 0.f, lw = inside ? x - xf : 0.f;
      const float hh = 1.f - lh, hw = 1.f - lw;
      const float a = inside ? a_in : 0.f;
      const bool yok0 = inside && y0 >= 0, yok1 = inside && y0 + 1 <= H - 1;
      const bool xok0 = x0 >= 0, xok1 = x0 + 1 <= W - 1;
      const unsigned p00 = base + (unsigned)(lv.start[l] + y0 * W + x0) * px_stride;
      FwdRecord r;
      r.w.x = hh * hw * a; r.w.y = hh * lw * a; r.w.z = lh * hw * a; r.w.w = lh * lw * a;
      r.off.x = (yok0 && xok0) ? p00 : kOobOffset;
      r.off.y = (yok0 && xok1) ? p00 + px_stride : kOobOffset;
      r.off.z = (yok1 && xok0) ? p00 + (unsigned)W * px_stride : kOobOffset;
      r.off.w = (yok1 && xok1) ? p00 + (unsigned)(W + 1) * px_stride : kOobOffset;
      *reinterpret_cast<FwdRecord *>(my_recs + s * sizeof(FwdRecord)) = r;
    }
  }
  __syncthreads();
  if (!live) return;

  const unsigned value_bytes = (unsigned)((size_t)d.N * d.S * d.M * DT * sizeof(VT));
  const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<VT *>(value), 0, (int)value_bytes, 0x00020000);
  const unsigned lane_off = (unsigned)lane * 12u;
  float acc[CPL];
#pragma unroll
  for (int c = 0; c < CPL; ++c) acc[c] = 0.f;
#pragma unroll LP_T ? LP_T : 4
  for (int s = 0; s < LP; ++s) {
    const FwdRecord r = *reinterpret_cast<const FwdRecord *>(my_recs + s * sizeof(FwdRecord));
    const u32x3 v0 = __builtin_amdgcn_raw_buffer_load_b96(rsrc, r.off.x + lane_off, 0, 0);
    const u32x3 v1 = __builtin_amdgcn_raw_buffer_load_b96(rsrc, r.off.y + lane_off, 0, 0);
    const u32x3 v2 = __builtin_amdgcn_raw_buffer_load_b96(rsrc, r.off.z + lane_off, 0, 0);
    const u32x3 v3 = __builtin_amdgcn_raw_buffer_load_b96(rsrc, r.off.w + lane_off, 0, 0);
    TR::fma_tap(acc, r.w.x, v0);
    TR::fma_tap(acc, r.w.y, v1);
    TR::fma_tap(acc, r.w.z, v2);
    TR::fma_tap(acc, r.w.w, v3);
  }
  if constexpr (sizeof(VT) == 4) {
    if (out_bf16) {
      uint16_t *o16 = reinterpret_cast<uint16_t *>(out) + (size_t)row * DT + lane * 3;
      o16[0] = f32_to_bf16_bits(acc[0]); o16[1] = f32_to_bf16_bits(acc[1]); o16[2] = f32_to_bf16_bits(acc[2]);
      return;
    }
  } else {
    if (out_bf16) {        // bf16 kernel: the flag means the OTHER row type -- float32 rows (the decoder's cross attention: a bf16
      // value projection sampled for float32 queries; the float32 sums are stored as they are, no bf16 rounding + cast pass)
      float *o32 = reinterpret_cast<float *>(out) + (size_t)row * DT + lane * CPL;
#pragma unroll
      for (int c = 0; c < CPL; c += 2) *reinterpret_cast<float2 *>(o32 + c) = make_float2(acc[c], acc[c + 1]);
      return;
    }
  }
  const u32x3 packed = TR::pack(acc);
  unsigned char *o = reinterpret_cast<unsigned char *>(out) + (size_t)row * DT * sizeof(VT) + lane_off;
  *reinterpret_cast<u32x3 *>(o) = packed;
}

// one element of a grad_out row that is stored as float32 or (go_bf16) bfloat16
__device__ __forceinline__ float ld_go(const float *grad_out, size_t idx, int go_bf16) {
  return go_bf16 ? bf16_bits_to_f32(reinterpret_cast<const uint16_t *>(grad_out)[idx]) : grad_out[idx];
}

// ---- backward, f32 ------------------------------------------------------------------------
struct BwdRecord {   // 48 B
  f32x4 q0;          // lh, lw, a, a*W
  u32x4 off;         // byte offsets of the taps (value and grad_value share the layout)
  f32x4 q2;          // a*H, -, -, -
};

template <int CTRL> __device__ __forceinline__ float dpp_f32(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// all-reduce over the 16 lanes of a DPP row (row_ror 8,4,2,1)
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_f32<0x128>(v);
  v += dpp_f32<0x124>(v);
  v += dpp_f32<0x122>(v);
  v += dpp_f32<0x121>(v);
  return v;
}

// ---- shared geometry of the owner-computes backward (msda_d48_patch.cuh) -----------------------
// Both kernels of that path must classify a sample as near/far identically, so the arithmetic is
// pinned with explicit intrinsics (no compiler-chosen fma contraction).
__device__ __forceinline__ float px_coord(float l, int size) { return __fmaf_rn(l, (float)size, -0.5f); }
__device__ __forceinline__ float anchor_coord(int qi, int size_l, int size_lq) {
  return __fmaf_rn((float)qi + 0.5f, __fdiv_rn((float)size_l, (float)size_lq), -0.5f);
}
// the same with the ratio size_l / size_lq handed in: the host computes it with one IEEE float division (bit-identical to
// __fdiv_rn), so the kernels do no divisions for it
__device__ __forceinline__ float anchor_from_ratio(int qi, float ratio) { return __fmaf_rn((float)qi + 0.5f, ratio, -0.5f); }
__device__ __forceinline__ bool near_anchor(float x, float y, float ax, float ay, float R) {
  return (fabsf(x - ax) <= R) && (fabsf(y - ay) <= R);
}
// query index -> (level, y, x) for queries laid out level after level, row-major (the encoder's order)
__device__ __forceinline__ void query_to_grid(int q, const int *start, const int *W, int L, int &lq, int &qy, int &qx) {
  lq = 0;
  for (int l = 1; l < L; ++l) lq = (q >= start[l]) ? l : lq;
  const int r = q - start[lq];
  qy = r / W[lq];
  qx = r - qy * W[lq];
}

// all-reduce over the 8 lanes of an aligned group (quad xor 1, quad xor 2, mirror within the half row)
__device__ __forceinline__ float row8_sum_d(float v) {
  v += dpp_f32<0xB1>(v);
  v += dpp_f32<0x4E>(v);
  v += dpp_f32<0x141>(v);
  return v;
}

// VT = storage type of `value` (float, or uint16_t = bfloat16 bits: the decoder's cross attention samples the bf16 projection
// of the premixed memory, round 5); grad_value is float32 either way, so a tap's byte offset there is (4 / sizeof(VT)) x its
// offset in `value` (both are (n S M + pixel M + m) rows of DT elements, or head-major alike).
template <int LP_T, int DT = kD48, typename VT = float>
__global__ __launch_bounds__(kD48Block) void msda_bwd_d48_f32_kernel(
    const float *__restrict__ grad_out, const VT *__restrict__ value,
    const int64_t *__restrict__ shapes, const int64_t *__restrict__ level_start,
    const float *__restrict__ loc, const float *__restrict__ attn, CoreDims d,
    float *__restrict__ grad_value, float *__restrict__ grad_loc, float *__restrict__ grad_attn,
    int nblk_padded, int go_bf16) {
  constexpr int G = DT / 3, kRows = kD48Block / G;        // lane i owns channels {i, i + G, i + 2G}
  __shared__ LevelTable lv;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int LP = LP_T ? LP_T : d.L * d.P;
  const int rec_stride = LP * (int)sizeof(BwdRecord) + 16;
  const int grp = threadIdx.x / G, lane = threadIdx.x % G;
  const long long total_rows = (long long)d.N * d.Lq * d.M;
  const long long row = (long long)xcd_band_block(nblk_padded) * kRows + grp;
  const bool live = row < total_rows;

  stage_levels(lv, shapes, level_start, d.L);
  __syncthreads();

  unsigned char *my_recs = smem_raw + (size_t)grp * rec_stride;
  if (live) {
    const int m = (int)(row % d.M);
    const long long n = row / ((long long)d.M * d.Lq);
    const unsigned px_stride = msda_px_stride(d, DT * (unsigned)sizeof(VT));
    const unsigned base = msda_row_base(d, (unsigned)n, (unsigned)m, DT * (unsigned)sizeof(VT));
    for (int s = lane; s < LP; s += G) {
      const int l = s / d.P;
      const int H = lv.H[l], W = lv.W[l];
      const long long li = row * LP + s;
      const float lx = loc[2 * li], ly = loc[2 * li + 1], a = attn[li];
      const float y = px_coord(ly, H), x = px_coord(lx, W);
      const bool inside = (y > -1.f) && (x > -1.f) && (y < (float)H) && (x < (float)W);
      const float yf = floorf(y), xf = floorf(x);
      const int y0 = (int)yf, x0 = (int)xf;
      const bool yok0 = inside && y0 >= 0, yok1 = inside && y0 + 1 <= H - 1;
      const bool xok0 = x0 >= 0, xok1 = x0 + 1 <= W - 1;
      const unsigned p00 = base + (unsigned)(lv.start[l] + y0 * W + x0) * px_stride;
      const float ai = inside ? a : 0.f;
      BwdRecord r;
      r.q0.x = inside ? y - yf : 0.f; r.q0.y = inside ? x - xf : 0.f; r.q0.z = ai; r.q0.w = ai * (float)W;
      r.q2.x = ai * (float)H; r.q2.y = 0.f; r.q2.z = 0.f; r.q2.w = 0.f;
      r.off.x = (yok0 && xok0) ? p00 : kOobOffset;
      r.off.y = (yok0 && xok1) ? p00 + px_stride : kOobOffset;
      r.off.z = (yok1 && xok0) ? p00 + (unsigned)W * px_stride : kOobOffset;
      r.off.w = (yok1 && xok1) ? p00 + (unsigned)(W + 1) * px_stride : kOobOffset;
      *reinterpret_cast<BwdRecord *>(my_recs + s * sizeof(BwdRecord)) = r;
    }
  }
  __syncthreads();
  // Dead rows keep running (offsets all out of range would need records; instead they exit:
  // the DPP reduction stays inside a 16-lane row, which is either fully live or fully dead).
  if (!live) return;

  constexpr unsigned kVE = (unsigned)sizeof(VT), kGS = 4u / kVE;      // bytes per value element; grad_value offset = kGS x value offset
  const unsigned value_bytes = (unsigned)((size_t)d.N * d.S * d.M * DT * kVE);
  const auto vsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<VT *>(value), 0, (int)value_bytes, 0x00020000);
  const auto gsrc = __builtin_amdgcn_make_buffer_rsrc(grad_value, 0, grad_value ? (int)(value_bytes * kGS) : 0, 0x00020000);
  const unsigned lane_off = (unsigned)lane * kVE;
  auto ldv = [&](unsigned o) -> float {
    if constexpr (sizeof(VT) == 4) return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(vsrc, o, 0, 0));
    else return __uint_as_float((unsigned)__builtin_amdgcn_raw_buffer_load_b16(vsrc, o, 0, 0) << 16);
  };
  const size_t gi = (size_t)row * DT + lane;
  const float g0 = ld_go(grad_out, gi, go_bf16), g1 = ld_go(grad_out, gi + G, go_bf16), g2 = ld_go(grad_out, gi + 2 * G, go_bf16);

  float keep_a = 0.f, keep_x = 0.f, keep_y = 0.f;
#pragma unroll LP_T ? 2 : 1
  for (int s = 0; s < LP; ++s) {
    const BwdRecord r = *reinterpret_cast<const BwdRecord *>(my_recs + s * sizeof(BwdRecord));
    const float lh = r.q0.x, lw = r.q0.y, a = r.q0.z;
    const float hh = 1.f - lh, hw = 1.f - lw;
    const float w[4] = {hh * hw, hh * lw, lh * hw, lh * lw};
    const unsigned off[4] = {r.off.x, r.off.y, r.off.z, r.off.w};
    float dot[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const unsigned o = off[k] + lane_off;
      const float v0 = ldv(o), v1 = ldv(o + kVE * G), v2 = ldv(o + 2u * kVE * G);
      dot[k] = g0 * v0 + g1 * v1 + g2 * v2;
      const float wa = w[k] * a;
      // (an out-of-map tap: kOobOffset stays out of range in grad_value as well -- never doubled into range)
      const unsigned og = off[k] == kOobOffset ? kOobOffset : o * kGS;
      if (grad_value) {        // (nullptr: grad_value comes from csrc/msda_d48_sparse.cuh; this kernel then only makes grad_loc / grad_attn)
        __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(wa * g0, gsrc, og, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(wa * g1, gsrc, og + 4u * G, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(wa * g2, gsrc, og + 8u * G, 0, 0);
      }
    }
    float pa = w[0] * dot[0] + w[1] * dot[1] + w[2] * dot[2] + w[3] * dot[3];
    float px = hh * (dot[1] - dot[0]) + lh * (dot[3] - dot[2]);
    float py = hw * (dot[2] - dot[0]) + lw * (dot[3] - dot[1]);
    if constexpr (G == 16) {
      pa = row16_sum(pa); px = row16_sum(px); py = row16_sum(py);
    } else {
      pa = row8_sum_d(pa); px = row8_sum_d(px); py = row8_sum_d(py);
    }
    px *= r.q0.w;
    py *= r.q2.x;
    if (LP <= G) {
      if (lane == s) { keep_a = pa; keep_x = px; keep_y = py; }
    } else if (lane == 0) {
      const long long li = row * LP + s;
      grad_attn[li] = pa; grad_loc[2 * li] = px; grad_loc[2 * li + 1] = py;
    }
  }
  if (LP <= G && lane < LP) {
    const long long li = row * LP + lane;
    grad_attn[li] = keep_a;
    *reinterpret_cast<float2 *>(grad_loc + 2 * li) = make_float2(keep_x, keep_y);
  }
}

}  // namespace snipper
