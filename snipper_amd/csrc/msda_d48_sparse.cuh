// msda_d48_sparse.cuh -- grad_value of the core op for FEW queries on a large bf16 value (the decoder's cross attention),
// without float atomics, without a float32 accumulation buffer and without a cast pass (gfx950).
//
// reference: models/ops/src/cuda/ms_deform_im2col_cuda.cuh:87-159 (col2im: every tap adds weight * attention * grad_out to
// its pixel's grad_value row with atomicAdd).  The decoder samples 60 queries x 8 heads x 12 points per frame out of 9 875
// positions (models/deformable_transformer.py:290-295): 184 000 taps per launch touch at most 29 % of the 632 000
// (position, head) rows, but the atomic formulation paid for all of them three times -- a 121 MB float32 memset, the atomics,
// and a 121 -> 60 MB cast to the bf16 gradient its consumers (the value projection's weight / data gradient) read:
// 83 us per decoder layer.  Here one workgroup of 1 024 threads owns one (sample, head, level) and one thread one tap:
//   1. the <= 1 024 taps (queries x points x 4 corners) become 32-bit keys  pixel << 10 | tap  (out-of-map taps: ~0);
//   2. a bitonic sort brings the taps of a pixel together, in tap order: the key stays in its thread's register, the 45 stages
//      with a partner inside the wave are lane exchanges, the 10 with a partner in another wave go through LDS;
//   3. the runs of equal pixels are numbered (ballot + prefix over the 16 waves); thread (run, c) adds up the run's
//      weight * attention * grad_out rows for channels 4c .. 4c + 3 (float32; the grad_out rows of the (sample, head) staged in
//      LDS) and stores them ONCE, as bf16 -- 12 neighbouring threads write a pixel's 96 bytes.
// The work of step 3 is spread over (run, channel group) items, so a workgroup whose queries all sample the same few pixels (the
// decoder at initialisation) costs little more than one with 1 024 distinct pixels: the whole call (memset + this kernel + query
// kernel) 36 / 47 us against 51 / 326 us when 256 threads sorted in LDS and one thread summed a whole run; this kernel in the
// training step 36 -> 16 us (tools/sparsebench.py, profiles/r05_sparse_backward_bench.jsonl).
// grad_value is zeroed by the launcher as bf16 (60 MB); untouched rows stay zero.  Deterministic (sorted order), one rounding
// per element.  grad_loc / grad_attn still come from msda_bwd_d48_f32_kernel (launched without its atomics).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "msda_d48.cuh"

namespace snipper {

constexpr int kSpTaps = 1024, kSpMaxLq = 64, kSpD = 48, kSpC4 = kSpD / 4;

__global__ __launch_bounds__(kSpTaps) void msda_bwd_d48_sparse_gv_kernel(
    const uint16_t *__restrict__ grad_out,      // [N][Lq][M][48] bf16, or float32 (go_f32)
    const int64_t *__restrict__ shapes, const int64_t *__restrict__ level_start,
    const float *__restrict__ loc, const float *__restrict__ attn, CoreDims d,
    uint16_t *__restrict__ grad_value,          // [N][S][M][48] bf16, zeroed
    int go_f32) {
  __shared__ __attribute__((aligned(16))) float g[kSpMaxLq * kSpD];
  __shared__ unsigned keys[kSpTaps + 1];
  __shared__ float wa[kSpTaps];                 // weight * attention by TAP, then (ws) by sorted position
  __shared__ float ws[kSpTaps];
  __shared__ unsigned short qs[kSpTaps], run_start[kSpTaps + 1];
  __shared__ int wave_runs[kSpTaps / 64], n_valid;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l = blockIdx.x % d.L;
  const int nm = blockIdx.x / d.L;
  const int m = nm % d.M, n = nm / d.M;
  const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1], start = (int)level_start[l];
  const int ntap = d.Lq * d.P * 4, P4 = d.P * 4;

  // ---- grad_out rows of this (sample, head): 16-byte pieces -- 6 per bf16 row, 12 per float32 row (go_f32)
  if (go_f32) {
    if (tid < d.Lq * kSpC4) {
      const int q = tid / kSpC4, c4 = tid - q * kSpC4;
      reinterpret_cast<float4 *>(g)[tid] = *reinterpret_cast<const float4 *>(
          reinterpret_cast<const float *>(grad_out) + (((size_t)n * d.Lq + q) * d.M + m) * kSpD + c4 * 4);
    }
  } else if (tid < d.Lq * 6) {
    const int q = tid / 6, c8 = tid - q * 6;
    const uint4 raw = *reinterpret_cast<const uint4 *>(grad_out + (((size_t)n * d.Lq + q) * d.M + m) * kSpD + c8 * 8);
    float4 lo, hi;
    lo.x = __uint_as_float(raw.x << 16); lo.y = __uint_as_float(raw.x & 0xffff0000u);
    lo.z = __uint_as_float(raw.y << 16); lo.w = __uint_as_float(raw.y & 0xffff0000u);
    hi.x = __uint_as_float(raw.z << 16); hi.y = __uint_as_float(raw.z & 0xffff0000u);
    hi.z = __uint_as_float(raw.w << 16); hi.w = __uint_as_float(raw.w & 0xffff0000u);
    reinterpret_cast<float4 *>(g)[2 * tid] = lo;
    reinterpret_cast<float4 *>(g)[2 * tid + 1] = hi;
  }
  // ---- this thread's tap -> key (the arithmetic of msda_bwd_d48_f32_kernel: same in-map tests, same weights)
  unsigned key = 0xffffffffu;
  {
    float w = 0.f;
    if (tid < ntap) {
      const int corner = tid & 3, sp = tid >> 2, q = sp / d.P, p = sp - q * d.P;
      const long long li = ((((long long)n * d.Lq + q) * d.M + m) * d.L + l) * d.P + p;
      const float lx = loc[2 * li], ly = loc[2 * li + 1], a = attn[li];
      const float y = px_coord(ly, H), x = px_coord(lx, W);
      const bool inside = (y > -1.f) && (x > -1.f) && (y < (float)H) && (x < (float)W);
      const float yf = floorf(y), xf = floorf(x);
      const float lh = y - yf, lw = x - xf;
      const int yy = (int)yf + (corner >> 1), xx = (int)xf + (corner & 1);
      if (inside && yy >= 0 && yy <= H - 1 && xx >= 0 && xx <= W - 1) {
        w = ((corner >> 1) ? lh : 1.f - lh) * ((corner & 1) ? lw : 1.f - lw) * a;
        key = ((unsigned)(yy * W + xx) << 10) | (unsigned)tid;
      }
    }
    wa[tid] = w;
  }
  // ---- bitonic sort, ascending: element e = tid; a stage (k, j) leaves min(key, partner's) in the lower element of a pair of
  // an ascending block and max in the upper (descending blocks the other way round)
  for (unsigned k = 2; k <= kSpTaps; k <<= 1)
    for (unsigned j = k >> 1; j > 0; j >>= 1) {
      unsigned other;
      if (j < 64) {
        other = (unsigned)__shfl_xor((int)key, (int)j, 64);
      } else {
        __syncthreads();                          // (the previous LDS stage's reads are done)
        keys[tid] = key;
        __syncthreads();
        other = keys[tid ^ j];
      }
      const bool keep_min = (((unsigned)tid & k) == 0) == (((unsigned)tid & j) == 0);
      key = keep_min ? min(key, other) : max(key, other);
    }
  // ---- runs of equal pixels.  Sorted position tid: is it the first tap of its pixel?  Which run is that?
  __syncthreads();
  keys[tid + 1] = key;
  if (tid == 0) { keys[0] = 0xffffffffu; n_valid = 0; }
  __syncthreads();
  const bool valid = key != 0xffffffffu;
  const unsigned prev = keys[tid];              // (keys[0]: the invalid key -- its "pixel" 2^22 - 1 is no pixel of a level)
  const bool is_head = valid && (prev >> 10) != (key >> 10);
  const unsigned long long heads = __ballot(is_head);
  if (lane == 0) wave_runs[wave] = __popcll(heads);
  {
    const unsigned tp = key & 1023u;
    ws[tid] = valid ? wa[tp] : 0.f;
    qs[tid] = (unsigned short)(tp / (unsigned)P4);
  }
  if (valid && (tid == kSpTaps - 1)) n_valid = kSpTaps;
  __syncthreads();
  if (!valid && (prev != 0xffffffffu || tid == 0)) n_valid = tid;      // the first invalid position = the number of valid taps
  int run = __popcll(heads & ((1ull << lane) - 1ull)), runs = 0;
#pragma unroll
  for (int w = 0; w < kSpTaps / 64; ++w) {
    const int c = wave_runs[w];
    if (w < wave) run += c;
    runs += c;
  }
  if (is_head) run_start[run] = (unsigned short)tid;
  __syncthreads();
  if (tid == 0) run_start[runs] = (unsigned short)n_valid;
  __syncthreads();
  // ---- one store per touched pixel and 4 channels: item = run * 12 + c
  for (int item = tid; item < runs * kSpC4; item += kSpTaps) {
    const int r = item / kSpC4, c = item - r * kSpC4;
    const int b = run_start[r], e = run_start[r + 1];
    const unsigned pix = keys[b + 1] >> 10;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int j = b; j < e; ++j) {
      const float w = ws[j];
      const float4 v = reinterpret_cast<const float4 *>(g)[qs[j] * kSpC4 + c];
      acc.x = fmaf(w, v.x, acc.x); acc.y = fmaf(w, v.y, acc.y); acc.z = fmaf(w, v.z, acc.z); acc.w = fmaf(w, v.w, acc.w);
    }
    auto pk = [](float x, float y) { return (unsigned)f32_to_bf16_bits(x) | ((unsigned)f32_to_bf16_bits(y) << 16); };
    uint2 o;
    o.x = pk(acc.x, acc.y); o.y = pk(acc.z, acc.w);
    reinterpret_cast<uint2 *>(grad_value + (((size_t)n * d.S + start + pix) * d.M + m) * kSpD)[c] = o;
  }
}

}  // namespace snipper
