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
 * one for one (same pointers, same sizes, same order; the stream comes first);
 * snipper_msda_forward_ex / _backward_ex are the same calls with the dtype as an
 * argument plus two optional inputs (a host copy of the level shapes, a config).
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
 *   backward: `grad_value`, `grad_loc` and `grad_attn` are fully written by the
 *             callee: nobody has to zero anything beforehand, and nothing the
 *             buffers held before the call is read.  (How: the kernels that
 *             accumulate with float atomics zero `grad_value` themselves with a
 *             hipMemsetAsync on `stream` first; the owner-computes backward of the
 *             encoder shape stores every element plainly and adds the far taps on
 *             top -- tests/test_owner_gpu.py poisons the buffer with NaN.)
 */
#ifndef SNIPPER_MSDA_H_
#define SNIPPER_MSDA_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SNIPPER_MSDA_ABI_VERSION 2

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
/* Name of the kernel variant the last forward / backward call OF THE CALLING THREAD dispatched to ("generic", "d48_lp12",
 * "d48_owner", ...; "none" before the thread's first call): a diagnostic for tests and profiles, never an input of any
 * computation.  Thread-local storage: the library keeps no process-wide mutable state. */
const char *snipper_msda_last_variant(void);

/* The library keeps NO tuning state: everything that can change which kernels run travels in this struct, passed by the
 * caller with every call of the *_ex entry points (NULL = the defaults).  Safe under concurrent callers (DDP's reducer
 * thread, the autograd engine's thread). */
typedef struct snipper_msda_config {
  int32_t struct_bytes;   /* sizeof(snipper_msda_config), checked                                                    */
  int32_t policy;         /* 0 auto; 1 generic kernels only; 2 tuned D=48 kernels but never the encoder-shape ones     */
  float near_radius;      /* owner-computes backward: a sample within this many pixels of its anchor is "near" (24)    */
  int32_t tile_kernel;    /* owner-computes backward, grad_value side, bfloat16 grad_out rows: 0 / 2 = per-tile dense scatter
                           * on the matrix pipe (csrc/msda_d48_tilemm.cuh), 1 = the vector / LDS sorted-list kernel
                           * (csrc/msda_d48_patch.cuh; float32 grad_out rows always take it)                            */
  int32_t tile_edge[3];   /* grad_value tile edge (power of two <= 16) for levels of > 4096 / > 1024 / fewer pixels;
                           * 0 (default) = the grad_value-side kernel's own choice: 16 / 8 / 4 vector, 16 / 8 / 8 matrix pipe  */
  int32_t debug_ablation; /* must be 0 (!= 0 selects timing ablations of the owner-computes backward with WRONG results;
                           * refused unless SNIPPER_MSDA_ALLOW_DEBUG=1 was in the environment at load time, and then
                           * reported as variant "d48_owner_debug")                                                     */
  int32_t value_layout;   /* 0 = `value` / `grad_value` as the reference holds them, [N, S, M, D]; 1 = head-major
                           * [N, M, S, D] -- a head's rows of neighbouring pixels are contiguous (the x-neighbour taps of a
                           * sample are one 2 D-element run): fewer L1 segments and a smaller footprint for the gathers.
                           * For callers that own both the producer and the consumer of `value` (the tied module core:
                           * snipper_temporal_mix_ex writes / reads this layout).  Tuned D = 48 / 24 kernels only, else
                           * SNIPPER_E_UNSUPPORTED; `out`, `loc`, `attn` and their gradients are unaffected.             */
  int32_t reserved[3];    /* must be 0, checked                                                                         */
} snipper_msda_config;
void snipper_msda_config_init(snipper_msda_config *cfg);     /* fills in the defaults */

/* General entry points.  dtype codes: 0 = float32, 1 = bfloat16 (raw bits), 2 = float64.
 *   value_dtype 0: loc / attn float32; out_dtype 0, or 1 = bfloat16 ROWS written by the kernel (D == 48 only, else
 *                  SNIPPER_E_UNSUPPORTED: the caller casts) -- under bf16 autocast the consumer of `out` (the output
 *                  projection) and the producer of grad_out hold bf16 anyway; every other array and all arithmetic stay f32;
 *   value_dtype 1: loc / attn float32, out bfloat16, gradients float32;   value_dtype 2: everything float64.
 *   host_shapes  : the SAME [L,2] (H,W) values as `shapes`, readable by the HOST, or NULL when unknown.  With D == 48,
 *                  P == 4, L <= 4 and Lq == S == sum(H*W) -- the encoder's self-attention, whose queries are the pixels of
 *                  the L maps in level-major raster order -- they enable the owner-computes backward
 *                  (csrc/msda_d48_patch.cuh; float32 or bfloat16 value): grad_value summed per tile by owner workgroups
 *                  -- in a fixed order, so bit-reproducible -- instead of per tap by HBM float atomics.  The result is the same function of
 *                  the inputs for ANY locations; only the speed depends on how local they are.  (Every H and W must be
 *                  < 32768 for this path -- larger maps take the atomic kernels.)
 *   workspace    : backward only; device scratch of at least snipper_msda_backward_ex_workspace_bytes(...) bytes (0 when
 *                  the encoder-shape path would not be taken: then NULL is fine).  The library never allocates.  (The
 *                  scratch holds the tiles' mark words and a list with room for one 8-byte entry per sample -- the samples
 *                  with a tap no tile owns; only the marks and the list's counter are zeroed per call.) */
int snipper_msda_forward_ex(void *stream, const snipper_msda_config *cfg, const int64_t *host_shapes, const void *value,
                            int value_dtype, const int64_t *shapes, const int64_t *level_start, const void *loc,
                            const void *attn, int N, int S, int M, int D, int L, int Lq, int P, void *out, int out_dtype);
long long snipper_msda_backward_ex_workspace_bytes(const snipper_msda_config *cfg, const int64_t *host_shapes,
                                                   int value_dtype, int N, int S, int M, int D, int L, int Lq, int P);
int snipper_msda_backward_ex(void *stream, const snipper_msda_config *cfg, const int64_t *host_shapes, void *workspace,
                             long long workspace_bytes, const void *grad_out, int grad_out_dtype, const void *value,
                             int value_dtype, const int64_t *shapes, const int64_t *level_start, const void *loc,
                             const void *attn, int N, int S, int M, int D, int L, int Lq, int P, void *grad_value,
                             void *grad_loc, void *grad_attn);

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

/* grad_value of a bfloat16 `value` for FEW queries (Lq <= 64 and Lq * P * 4 <= 1024, D == 48: the decoder's cross attention)
 * WITHOUT float atomics: one workgroup per (sample, head, level) sorts its taps by pixel in LDS and stores every touched
 * pixel's row once (csrc/msda_d48_sparse.cuh; reference ms_deform_im2col_cuda.cuh:87-159).  grad_value [N][S][M][D] bfloat16 is
 * fully written by the callee (zeroed, then the touched rows), deterministic, one rounding per element; grad_loc / grad_attn
 * float32 as snipper_msda_backward_bf16.  Returns SNIPPER_E_UNSUPPORTED for any other shape: call snipper_msda_backward_ex. */
int snipper_msda_backward_sparse_bf16(void *stream, const uint16_t *grad_out, const uint16_t *value, const int64_t *shapes,
                                      const int64_t *level_start, const float *loc, const float *attn, int N, int S, int M, int D,
                                      int L, int Lq, int P, uint16_t *grad_value, float *grad_loc, float *grad_attn);
/* The same with FLOAT32 grad_out rows [N][Lq][M*D] (the consumer of the sampled rows -- the decoder's float32 output
 * projection -- hands its data gradient over as it is; no bf16 cast launch in front of the call). */
int snipper_msda_backward_sparse_f32rows(void *stream, const float *grad_out, const uint16_t *value, const int64_t *shapes,
                                         const int64_t *level_start, const float *loc, const float *attn, int N, int S, int M, int D,
                                         int L, int Lq, int P, uint16_t *grad_value, float *grad_loc, float *grad_attn);

#ifdef __cplusplus
}
#endif
#endif /* SNIPPER_MSDA_H_ */
