/*
 * snipper_msda.h -- C ABI of the MI355X (gfx950) multi-scale deformable
 * attention library, libsnipper_msda.so.
 *
 * This is the drop-in boundary for the native op of JimmyZou/Snipper
 * (reference paths relative to /root/reference).  The reference reaches its CUDA
 * kernels through two launchers with this exact argument order:
 *
 *   ms_deformable_im2col_cuda  models/ops/src/cuda/ms_deform_im2col_cuda.cuh:923-954
 *       called from ms_deform_attn_cuda_forward,  cuda/ms_deform_attn_cuda.cu:64-74
 *   ms_deformable_col2im_cuda  models/ops/src/cuda/ms_deform_im2col_cuda.cuh:956-1327
 *       called from ms_deform_attn_cuda_backward, cuda/ms_deform_attn_cuda.cu:134-147
 *
 * and exposes them to Python as MultiScaleDeformableAttention.ms_deform_attn_forward /
 * ms_deform_attn_backward (models/ops/src/vision.cpp:13-16, ms_deform_attn.h:20-62).
 * snipper_msda_forward_* / snipper_msda_backward_* replace those two launchers
 * one for one (same pointers, same sizes, same order; the stream comes first).
 *
 * Conventions
 *   - every data pointer is a DEVICE pointer to a contiguous row-major array;
 *     inputs are borrowed and never written;
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream);
 *     all work is enqueued on it, nothing synchronises with the host, no memory
 *     is allocated, so the calls are hipGraph-capturable;
 *   - return value: 0 on success, a negative SNIPPER_E_* code for an argument
 *     the library rejects, or a positive hipError_t from the launch.  Unlike the
 *     reference (which only printf()s launch errors, .cuh:948-952,1321-1325) an
 *     error is always reported to the caller;
 *   - layouts: value [N,S,M,D]; shapes [L,2] int64 (H,W); level_start [L] int64;
 *     loc [N,Lq,M,L,P,2] normalised (x,y); attn [N,Lq,M,L,P];
 *     out / grad_out [N,Lq,M,D];
 *   - sampling semantics: pixel = loc*size - 0.5, bilinear, taps outside the map
 *     contribute zero, a sample is skipped unless -1 < pixel < size on both axes
 *     (.cuh:285-291) -- identical to grid_sample(align_corners=False, zeros).
 *   - limits: S*M*D < 2^31 and Lq*M*L*P*2 < 2^31 per batch element, L*P <= 1024.
 *
 * Outputs
 *   forward : `out` is fully overwritten (no pre-zeroing needed).
 *   backward: `grad_loc` and `grad_attn` are fully overwritten; `grad_value` is
 *             ZEROED BY THE CALLEE (hipMemsetAsync on `stream`) and then
 *             accumulated into.
 */
#ifndef SNIPPER_MSDA_H_
#define SNIPPER_MSDA_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SNIPPER_MSDA_ABI_VERSION 1

enum {
  SNIPPER_OK = 0,
  SNIPPER_E_NULL = -1,      /* a required pointer is NULL                         */
  SNIPPER_E_SHAPE = -2,     /* a size is <= 0 or exceeds the documented limits    */
  SNIPPER_E_UNSUPPORTED = -3 /* dtype / variant not built                          */
};

/* ABI version of the loaded library (== SNIPPER_MSDA_ABI_VERSION it was built with). */
int snipper_msda_abi_version(void);
/* Human-readable text for a code returned by any entry point (static storage). */
const char *snipper_msda_strerror(int code);
/* Name of the kernel variant the last forward / backward call on this thread
 * dispatched to ("generic", "d48", ...); for tests and profiles. */
const char *snipper_msda_last_variant(void);
/* Kernel-variant policy for tests and benchmarks: 0 = auto (tuned kernels where eligible),
 * 1 = generic kernels only.  Process-wide; returns 0 or SNIPPER_E_UNSUPPORTED. */
int snipper_msda_set_policy(int policy);   /* 2 = tuned kernels but never the owner-computes backward */
/* Tuning knobs (tests / benchmarks): "owner_enable" (0/1, default 1: owner-computes backward when
 * a host copy of the shapes and a workspace are supplied), "near_radius" (pixels,
 * default 6), "owner_tile_edge_big" / "_mid" / "_small" (tile edge, a power of two <= 16, for levels of > 4096 /
 * > 1024 / fewer pixels; default 16 / 8 / 4), "owner_debug" (timing ablations; wrong results when != 0). */
int snipper_msda_set_param(const char *name, double value);

/* ---- core op: replaces ms_deformable_im2col_cuda (.cuh:923-954) ------------------ */
int snipper_msda_forward_f32(void *stream, const float *value, const int64_t *shapes,
                             const int64_t *level_start, const float *loc, const float *attn,
                             int N, int S, int M, int D, int L, int Lq, int P, float *out);
int snipper_msda_forward_f64(void *stream, const double *value, const int64_t *shapes,
                             const int64_t *level_start, const double *loc, const double *attn,
                             int N, int S, int M, int D, int L, int Lq, int P, double *out);
/* bf16 storage (uint16_t = raw bfloat16 bits) for value/out; loc and attn stay f32;
 * accumulation in f32.  New capability: the reference dispatches float/double only
 * (ms_deform_attn_cuda.cu:64). */
int snipper_msda_forward_bf16(void *stream, const uint16_t *value, const int64_t *shapes,
                              const int64_t *level_start, const float *loc, const float *attn,
                              int N, int S, int M, int D, int L, int Lq, int P, uint16_t *out);

/* ---- core op backward: replaces ms_deformable_col2im_cuda (.cuh:956-1327) -------- */
int snipper_msda_backward_f32(void *stream, const float *grad_out, const float *value,
                              const int64_t *shapes, const int64_t *level_start,
                              const float *loc, const float *attn,
                              int N, int S, int M, int D, int L, int Lq, int P,
                              float *grad_value, float *grad_loc, float *grad_attn);
/* Owner-computes backward (csrc/msda_d48_owner.cuh); "owner_enable" = 0 switches it off.
 * Same contract as snipper_msda_backward_f32 plus
 *   host_shapes : the SAME [L,2] (H,W) values as `shapes`, readable by the host (NULL = unknown);
 *   workspace   : device scratch of at least snipper_msda_backward_workspace_bytes(...) bytes
 *                 (the library never allocates); contents are overwritten.
 * With D == 48, P == 4, L <= 4 and Lq == S == sum(H*W) -- the encoder's self-attention, whose queries
 * are the pixels of the L maps in level-major raster order -- grad_value is summed per tile by owner
 * workgroups instead of per tap by HBM float atomics.  The result is the same function of the inputs
 * for ANY locations; only the speed depends on how local they are.  Whenever the shape, the knobs or
 * the workspace do not qualify, the call is exactly snipper_msda_backward_f32.
 * snipper_msda_backward_workspace_bytes returns 0 when the fast path would not be taken. */
/* bfloat16 rows at the op's two activation interfaces, float32 everything else (value, loc, attn, all gradients,
 * all arithmetic).  Under bf16 autocast the producer of grad_out (the data gradient of the output projection) and the
 * consumer of out (the output projection itself) hold bf16 anyway; taking / writing bf16 here is exact with respect
 * to that pipeline and saves two cast passes and half of the row traffic.  D == 48 kernels only: any other shape
 * returns SNIPPER_E_UNSUPPORTED (the caller casts and uses the float32 entry points).
 *   snipper_msda_forward_f32_bf16out    : as snipper_msda_forward_f32, out [N,Lq,M*D] bfloat16
 *   snipper_msda_backward_ws_f32_bf16in : as snipper_msda_backward_ws_f32, grad_out [N,Lq,M*D] bfloat16
 *                                         (host_shapes / workspace may be NULL / 0: atomic kernel) */
int snipper_msda_forward_f32_bf16out(void *stream, const float *value, const int64_t *shapes,
                                     const int64_t *level_start, const float *loc, const float *attn,
                                     int N, int S, int M, int D, int L, int Lq, int P, uint16_t *out);
int snipper_msda_backward_ws_f32_bf16in(void *stream, const uint16_t *grad_out, const float *value,
                                        const int64_t *shapes, const int64_t *level_start,
                                        const int64_t *host_shapes, void *workspace, long long workspace_bytes,
                                        const float *loc, const float *attn,
                                        int N, int S, int M, int D, int L, int Lq, int P,
                                        float *grad_value, float *grad_loc, float *grad_attn);

long long snipper_msda_backward_workspace_bytes(int N, int S, int M, int D, int L, int Lq, int P,
                                                const int64_t *host_shapes);
int snipper_msda_backward_ws_f32(void *stream, const float *grad_out, const float *value,
                                 const int64_t *shapes, const int64_t *level_start,
                                 const int64_t *host_shapes, void *workspace, long long workspace_bytes,
                                 const float *loc, const float *attn,
                                 int N, int S, int M, int D, int L, int Lq, int P,
                                 float *grad_value, float *grad_loc, float *grad_attn);
int snipper_msda_backward_f64(void *stream, const double *grad_out, const double *value,
                              const int64_t *shapes, const int64_t *level_start,
                              const double *loc, const double *attn,
                              int N, int S, int M, int D, int L, int Lq, int P,
                              double *grad_value, double *grad_loc, double *grad_attn);
/* bf16 value / grad_out; grad_value is accumulated in f32 (float*), grad_loc / grad_attn f32. */
int snipper_msda_backward_bf16(void *stream, const uint16_t *grad_out, const uint16_t *value,
                               const int64_t *shapes, const int64_t *level_start,
                               const float *loc, const float *attn,
                               int N, int S, int M, int D, int L, int Lq, int P,
                               float *grad_value, float *grad_loc, float *grad_attn);

#ifdef __cplusplus
}
#endif
#endif /* SNIPPER_MSDA_H_ */
