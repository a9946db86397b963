/*
 * snipper_dense.h -- C ABI of the dense (MFMA) kernels of libsnipper_msda.so that sit on the
 * deformable-attention path's callers.
 *
 * Reference call sites: the Linears of MSDeformAttn (models/ops/modules/ms_deform_attn.py:114,143-163,
 * 237), the encoder FFN (models/deformable_transformer.py:194-198) and, in NHWC, the 1x1 convolutions
 * of the ResNet bottleneck with the frozen BatchNorm (models/backbone.py:54-64) folded in.  In the
 * reference all of these are separate PyTorch ops (cuBLAS / cuDNN + elementwise passes).
 *
 * Same conventions as snipper_msda.h: device pointers, `stream` = hipStream_t as void*, 0 = success,
 * negative SNIPPER_E_* (snipper_msda.h) for rejected arguments, positive hipError_t otherwise.
 */
#ifndef SNIPPER_DENSE_H_
#define SNIPPER_DENSE_H_

#include <stddef.h>
#include <stdint.h>
#include "snipper_msda.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Y[M,N] = act( X[M,K] . W[N,K]^T + bias[N] + R[M,N] ),  act = ReLU if relu != 0 else identity.
 * X, W, R, Y: bfloat16 bits (uint16_t); bias: float32 or NULL; R: NULL for no residual.
 * ldx / ldr / ldy: row strides in ELEMENTS (>= K / N / N).  Requirements: K % 64 == 0, N % 4 == 0,
 * X/W rows 16-byte aligned (ldx % 8 == 0), Y/R rows 8-byte aligned (ldy, ldr % 4 == 0).
 * Accumulation in float32, one rounding to bf16 at the end.
 * dropout_p > 0 applies inverted dropout AFTER the activation in the same epilogue (the FFN's
 * ``dropout(relu(linear1(x)))``, models/deformable_transformer.py:196); kept values are scaled by 1 / (1 - p).
 * WHICH elements are kept is IMPLEMENTATION-DEFINED: a deterministic function of (seed, shape, device) that is not
 * part of the contract -- the tile kernel hashes (seed, m * N + n), the weight-stationary kernel this entry forwards to
 * for K in {288, 384}, M >= 8192 draws 16-bit uniforms per thread (snipper_linear_wres_bf16 below; p resolved to 2^-16,
 * the stream depends on the device's CU count).  A caller must recover the mask FROM THE OUTPUT (with relu != 0 a kept,
 * active element is > 0 -- what this package's backward does) and never regenerate it from the seed.  M * N < 2^32. */
int snipper_linear_bf16(void *stream, const uint16_t *X, long long ldx, const uint16_t *W,
                        const float *bias, const uint16_t *R, long long ldr, uint16_t *Y, long long ldy,
                        int M, int N, int K, int relu, float dropout_p, uint64_t seed);

/* Data gradient of the same layers without a transposed weight copy: Y[M,N] = gate(X[M,K] . W[K,N] + R[M,N]), W
 * row-major with the reduction index as its SLOW axis (a Linear's / 1x1 convolution's weight [out, in] taken as
 * [K, N]); X, W, R, A, Y bf16.  R = NULL or an addend (the gradient that reaches the same tensor over a skip
 * connection).  A = NULL or the activation this gradient flows back through: gate(v) = A > 0 ? v * gate_scale : 0,
 * the backward of ReLU (gate_scale 1) or of ReLU followed by dropout with rate p applied to A (gate_scale 1/(1-p)).
 * Requirements: K % 64 == 0, N % 8 == 0, ldx / ldw % 8 == 0, ldy / ldr / lda % 4 == 0, X / W 16-byte aligned. */
int snipper_linear_nn_bf16(void *stream, const uint16_t *X, long long ldx, const uint16_t *W, long long ldw,
                           const uint16_t *R, long long ldr, const uint16_t *A, long long lda, float gate_scale,
                           uint16_t *Y, long long ldy, int M, int N, int K);

/* Weight-stationary form of snipper_linear_bf16 for SHORT reductions on many rows (csrc/wres_gemm_bf16.cuh):
 *   Y[M,N] = gate( dropout( act( X[M,K] . W[N,K]^T + bias ) ) ),   K in {288, 384}, N % 8 == 0, M >= 8192.
 * W (row stride ldw) stays in registers for the whole launch, X is streamed through an LDS-DMA ring, one persistent
 * workgroup per CU; a data gradient dX = dY . Wl passes the TRANSPOSED weight (snipper_transpose_batch_bf16).  A = NULL or
 * the gate activation [M][N]: gate(v) = A > 0 ? v * gate_scale : 0 (see snipper_linear_nn_bf16).  Dropout: every lane draws
 * 16-bit uniforms from its own xorshift32 stream seeded by a hash of (seed, global thread id) (p is resolved to 2^-16): the
 * mask is a deterministic function of (seed, shape, device CU count), not of the element index alone, and differs from the
 * tile kernel's for the same seed.  ldx, ldw, ldy, lda % 8 == 0; X, W, Y, A 16-byte aligned.
 * snipper_linear_wres_supported: 1 when the shape is taken (snipper_linear_bf16 itself dispatches here when it is and
 * there is no residual), else 0 and the entry point returns SNIPPER_E_UNSUPPORTED. */
int snipper_linear_wres_supported(int M, int N, int K);
int snipper_linear_wres_bf16(void *stream, const uint16_t *X, long long ldx, const uint16_t *W, long long ldw,
                             const float *bias, const uint16_t *A, long long lda, float gate_scale, uint16_t *Y,
                             long long ldy, int M, int N, int K, int relu, float dropout_p, uint64_t seed);

/* dst_i[c][r] = src_i[r][c] (bf16 bits) for count <= 48 small matrices in one launch: src_i [rows_i][cols_i] with row
 * stride ld_src_i, dst_i [cols_i][rows_i] with row stride ld_dst_i.  The arrays are HOST arrays of count entries. */
int snipper_transpose_batch_bf16(void *stream, int count, const void *const *src, void *const *dst, const int *rows,
                                 const int *cols, const long long *ld_src, const long long *ld_dst);

/* Measurement aid (tools/copybench.py): dst[i] = src[i], 16 bytes per lane, grid-stride -- the copy whose rate is the
 * achievable-HBM figure the roofline fractions are read beside.  bytes % 16 == 0, 16-byte aligned pointers. */
int snipper_hbm_copy_probe(void *stream, const void *src, void *dst, long long bytes);

/* Backward of the (ReLU -> dropout) epilogue above from the layer's OUTPUT alone: a kept, active element has y > 0,
 * a dropped or inactive one y == 0, so grad_pre = y > 0 ? grad_y / (1 - p) : 0 (p = 0: plain ReLU backward).
 * bf16 bits, n % 8 == 0, 16-byte aligned. */
/* The decoder's dense self-attention (reference models/deformable_transformer.py:282-287: nn.MultiheadAttention over the
 * nq * T object queries of a sample) for L <= 384 queries (up to 256: K and V staged side by side; beyond: one row buffer
 * staged twice, 32 queries per workgroup) and heads of 48 or 32 channels, one launch each way:
 *   out[b,i,h,:] = sum_j dropout(softmax_j(scale * q[b,i,h,:] . k[b,j,h,:]))[j] * v[b,j,h,:]
 * Element (b, i, h, e) of q / k / v / out / their gradients is at base + b * X_bs + i * X_ld + h * hd + e (so q and k
 * may be the two halves of a packed projection output and the gradients may be written into the halves of its
 * gradient).  P [bs,H,L,L] receives the probabilities before dropout (the backward reads it and the saved `out`).
 * Dropout: counter-based hash of (seed, element index); the backward must be given the forward's seed.  float32
 * throughout; leading dimensions % 4 == 0, 16-byte aligned pointers. */
int snipper_small_attention_forward_f32(void *stream, const float *q, long long q_ld, long long q_bs, const float *k,
                                        long long k_ld, long long k_bs, const float *v, long long v_ld, long long v_bs,
                                        float *out, long long o_ld, long long o_bs, float *P, int bs, int H, int L, int hd,
                                        float scale, float dropout_p, uint64_t seed);
int snipper_small_attention_backward_f32(void *stream, const float *q, long long q_ld, long long q_bs, const float *k,
                                         long long k_ld, long long k_bs, const float *v, long long v_ld, long long v_bs,
                                         const float *out, long long o_ld, long long o_bs, const float *P,
                                         const float *dout, long long do_ld, long long do_bs, float *dq, long long dq_ld,
                                         long long dq_bs, float *dk, long long dk_ld, long long dk_bs, float *dv,
                                         long long dv_ld, long long dv_bs, int bs, int H, int L, int hd, float scale,
                                         float dropout_p, uint64_t seed);

/* Float32 products of decoder size (a few hundred rows: the Linears of reference models/deformable_transformer.py:244-343),
 * SEVERAL PER LAUNCH:  out[I,J] = opA(A)[I,R] . opB(B)[R,J] (+ bias[J]),  optionally colsum[I] = sum_r opA(A)[i,r].
 * A is stored [I][R] (a_transposed = 0) or [R][I] (1); B is stored [R][J] (0) or [J][R] (1); out / colsum may be NULL
 * (not both).  One launch is e.g. a Linear's forward (x . W^T + b: B transposed), its whole backward (dX = G . W and
 * dW = G^T . X with db as the column sums), or a packed projection pair.  float32 in, float32 accumulate
 * (v_mfma_f32_32x32x2_f32: bitwise an fmaf chain), fixed summation order.  Stored row lengths and leading dimensions
 * % 4 == 0, 16-byte aligned operands, count <= 6. */
typedef struct snipper_small_gemm {
  const float *A; long long lda; int a_transposed;
  const float *B; long long ldb; int b_transposed;
  float *out; long long ldo;
  const float *bias;
  float *colsum;
  int I, J, R;
  /* optional (zero = absent).  B2 / r_split (b_transposed == 0, r_split % 32 == 0): rows r >= r_split of opB come from
   * B2[r - r_split] (two stacked weights stored apart).  Epilogue: out = gate(dropout(relu(acc + bias))): relu != 0,
   * dropout_p with the counter-based hash of (seed, row * J + col), gate [I][J] (ldgate): result * gate_scale where
   * gate > 0, else 0 -- the backward of ReLU + dropout given the layer's output. */
  const float *B2; long long ldb2; int r_split;
  int relu; float dropout_p; unsigned long long seed;
  const float *gate; long long ldgate; float gate_scale;
} snipper_small_gemm;
int snipper_small_gemm_batch_f32(void *stream, const snipper_small_gemm *problems, int count);
/* convenience: y[M,N] = x[M,K] . W[N,K]^T + b[N] (b may be NULL) */
int snipper_small_linear_forward_f32(void *stream, const float *X, long long ldx, const float *W, long long ldw,
                                     const float *bias, int M, int N, int K, float *Y, long long ldy);
/* convenience: backward of y = x . W^T + b:  dX[M,K] = G[M,N] . W[N,K], dW[N,K] = G^T . X[M,K], db[N] (each optional) */
int snipper_small_linear_backward_f32(void *stream, const float *G, long long ldg, const float *X, long long ldx,
                                      const float *W, long long ldw, int M, int N, int K, float *dX, long long lddx,
                                      float *dW, long long lddw, float *db);
int snipper_relu_dropout_backward_bf16(void *stream, const uint16_t *grad_y, const uint16_t *y, uint16_t *grad_pre,
                                       long long n, float dropout_p);

/* Weight and bias gradient of the same layers (csrc/wgrad_bf16.cuh):
 *   dW[N,Kc] (+)= scale[n] * sum_m G[m,n] * X[m,kc]      db[N] (+)= sum_m G[m,n]
 * G [M,N] (leading dimension ldg) = gradient of the layer's output, X [M,Kc] (ldx) = its input, both bf16 bits;
 * dW (leading dimension lddw) and db float32; scale [N] float32 or NULL (the folded BatchNorm factor of a 1x1
 * convolution); db may be NULL; accumulate != 0 adds to dW / db instead of overwriting.  The reduction over M is
 * split over the chip into per-range partial sums in `workspace` (>= snipper_wgrad_workspace_bytes(M,N,Kc) bytes,
 * 16-byte aligned) and summed in a fixed order by a second kernel: results are deterministic.
 * Requirements: N, Kc, ldg, ldx % 8 == 0, lddw % 4 == 0, 16-byte aligned pointers. */
size_t snipper_wgrad_workspace_bytes(int M, int N, int Kc);
int snipper_wgrad_bf16(void *stream, const uint16_t *G, long long ldg, const uint16_t *X, long long ldx,
                       int M, int N, int Kc, const float *scale, float *dW, long long lddw, float *db,
                       int accumulate, void *workspace, size_t workspace_bytes);

/* Residual + dropout + LayerNorm, the tail of every transformer sub-layer (csrc/ln_fused.cuh; reference
 * models/deformable_transformer.py:200-216, 266-300: ``src = src + self.dropoutN(src2); src = self.normN(src)``):
 *   s = x + dropout_p(z);  y = LayerNorm(s) * gamma + beta
 * x, z, pos [rows, C] with dtype codes 0 = float32, 1 = bfloat16 bits; z may be NULL (plain LayerNorm), pos only
 * feeds yq16.  Outputs (any non-empty subset): y32 float32, y16 bfloat16, yq16 = bfloat16(y + pos).  For the
 * backward: s_save [rows, C] float32, mean / rstd [rows], keep [rows, C/4] bytes (low 4 bits = kept elements; only
 * written when z != NULL and p > 0).  seed selects the dropout mask (a counter-based hash of seed and element index).
 * C % 4 == 0, C <= 1024, rows * C < 2^32. */
int snipper_add_dropout_layernorm_forward(void *stream, const void *x, int x_dt, const void *z, int z_dt,
                                          const void *pos, int pos_dt, const float *gamma, const float *beta,
                                          int rows, int C, float p, float eps, uint64_t seed,
                                          float *s_save, float *mean, float *rstd, uint8_t *keep,
                                          float *y32, uint16_t *y16, uint16_t *yq16);
/* The same with a LAZY residual: when x_mean / x_rstd [rows] and x_gamma / x_beta [C] are given (all four or none), `x`
 * (float32) is the s_save of the PREVIOUS sub-layer's call and its LayerNorm output is recomputed on load,
 *   x := (x - x_mean[row]) * x_rstd[row] * x_gamma + x_beta,
 * so that call need not write y32 at all (a chain of sub-layers keeps only the s_save arrays its backward needs anyway).
 * The gradient of this call's x (dx of the backward below) is the previous call's g32, as before. */
int snipper_add_dropout_layernorm_forward_ex(void *stream, const void *x, int x_dt, const float *x_mean, const float *x_rstd,
                                             const float *x_gamma, const float *x_beta, const void *z, int z_dt,
                                             const void *pos, int pos_dt, const float *gamma, const float *beta,
                                             int rows, int C, float p, float eps, uint64_t seed,
                                             float *s_save, float *mean, float *rstd, uint8_t *keep,
                                             float *y32, uint16_t *y16, uint16_t *yq16);
/* Backward of the above.  g32 / g16 / gq16 = gradients of y32 / y16 / yq16 (any non-empty subset; they are summed).
 * dx (= dL/ds) and dz (masked and rescaled) may each be NULL; dgamma / dbeta [C] float32 are overwritten (summed in
 * a fixed order through `workspace`, >= snipper_add_dropout_layernorm_workspace_bytes(rows, C) bytes). */
size_t snipper_add_dropout_layernorm_workspace_bytes(int rows, int C);
int snipper_add_dropout_layernorm_backward(void *stream, const float *g32, const uint16_t *g16, const uint16_t *gq16,
                                           const float *s_save, const float *mean, const float *rstd,
                                           const float *gamma, const uint8_t *keep, int rows, int C, float p,
                                           void *dx, int dx_dt, void *dz, int dz_dt, float *dgamma, float *dbeta,
                                           void *workspace, size_t workspace_bytes);

/* GroupNorm of the input projections on token rows (csrc/gn_tokens.cuh; reference models/model.py:62-84 followed by
 * the flatten + concatenation of models/deformable_transformer.py:103-122).  x [n, hw, C] bfloat16 = the 1x1
 * projection's output in NHWC (one row per pixel); statistics per (image, group) over hw x C/G values.  The result
 * is written into rows (i * dst_rows_per_image + dst_row_offset + r) of [*, C] buffers -- the level's slice of the
 * concatenated [b, t, S, C] token arrays: y32 float32, y16 bfloat16, yq16 = bfloat16(y + pos) with pos (dtype code
 * 0 = f32 / 1 = bf16) addressed like the outputs; any non-empty subset.  stats [n, G, 2] (mean, rstd) is kept for
 * the backward, whose g32 / g16 / gq16 (gradients of the three outputs, same addressing, summed) give dx [n, hw, C]
 * bfloat16 and dgamma / dbeta [C] (overwritten, summed in a fixed order).
 * Requirements: C % G == 0, (C / G) % 4 == 0, C <= 1024, G <= 64; workspace >= ..._workspace_bytes(n, hw, C, G). */
size_t snipper_groupnorm_tokens_workspace_bytes(int n, int hw, int C, int G);
int snipper_groupnorm_tokens_forward(void *stream, const uint16_t *x, const float *gamma, const float *beta,
                                     int n, int hw, int C, int G, float eps,
                                     long long dst_rows_per_image, long long dst_row_offset,
                                     const void *pos, int pos_dt, float *y32, uint16_t *y16, uint16_t *yq16,
                                     float *stats, void *workspace, size_t workspace_bytes);
int snipper_groupnorm_tokens_backward(void *stream, const uint16_t *x, const float *gamma, const float *stats,
                                      const float *g32, const uint16_t *g16, const uint16_t *gq16,
                                      int n, int hw, int C, int G,
                                      long long dst_rows_per_image, long long dst_row_offset,
                                      uint16_t *dx, float *dgamma, float *dbeta, void *workspace, size_t workspace_bytes);

/* ---- ResNet stem tail (csrc/gn_tokens.cuh) ------------------------------------------------------------------
 * out[n, oh, ow, c] = relu(max over the 3x3 / stride 2 / pad 1 window of y[n, :, :, c] + shift[c]): the frozen-BN
 * shift (models/backbone.py:54-64), the ReLU and torchvision's MaxPool2d(3, 2, 1) after conv1 in one pass.
 * y [N, H, W, C] bf16 NHWC = conv1 with the BN scale folded into its weight and NO bias; shift [C] float32;
 * out [N, (H-1)/2+1, (W-1)/2+1, C] bf16 NHWC.  C % 8 == 0, 16-byte aligned pointers.  Forward only (the stem is frozen). */
int snipper_stem_pool_bf16(void *stream, const uint16_t *y, const float *shift, int N, int H, int W, int C, uint16_t *out);

/* Column sums of row segments: out[c] = sum_{i < n_images} sum_{r < rows_per_seg} x[i * image_stride + r * C + c],
 * x bfloat16 (image_stride in elements), out [C] float32, summed in a fixed order.  The gradient of a per-level
 * embedding broadcast over a level's tokens (reference models/deformable_transformer.py:118), taken directly from the
 * level's slice of the [b, t, S, C] gradient.  C % 4 == 0, C <= 1024. */
size_t snipper_colsum_workspace_bytes(int n_images, int rows_per_seg, int C);
int snipper_colsum_segments_bf16(void *stream, const uint16_t *x, long long image_stride, int n_images,
                                 int rows_per_seg, int C, float *out, void *workspace, size_t workspace_bytes);
/* The same over n_src (<= 8) source tensors of one shape, all read from element `elem_offset` on: the column sums of their
 * SUM without forming it (a parameter used n times collects n gradients: level_embed behind the position encoding's six
 * consumers, reference models/deformable_transformer.py:118-121).  workspace: n_src x snipper_colsum_workspace_bytes.  `srcs`
 * is a HOST array of device pointers. */
int snipper_colsum_segments_multi_bf16(void *stream, const uint16_t *const *srcs, int n_src, long long elem_offset,
                                       long long image_stride, int n_images, int rows_per_seg, int C, float *out,
                                       void *workspace, size_t workspace_bytes);

/* out = bf16(sum_i float(srcs[i])) over `numel` (% 8 == 0) elements, n_src <= 8 bf16 tensors (16-byte aligned; `srcs` is a HOST
 * array of device pointers): the gradient of a tensor with several consumers in one pass with float32 accumulation
 * (autograd's own accumulation: n_src - 1 add launches, each partial sum rounded to bf16). */
int snipper_sum_bf16(void *stream, const uint16_t *const *srcs, int n_src, uint16_t *out, long long numel);

/* out = sum of n_src (<= 16) float32 arrays of numel elements (numel % 4 == 0, 16-byte aligned) in one pass: the gradients of
 * the aliases of a float32 tensor with several consumers (query_pos feeds two additions per decoder layer, reference
 * models/deformable_transformer.py:252-254), which autograd would add pairwise, one launch per pair. */
int snipper_sum_f32(void *stream, const float *const *srcs, int n_src, float *out, long long numel);

/* Decoder-size residual + dropout + LayerNorm (csrc/small_ln.cuh; reference models/deformable_transformer.py:266-300), float32,
 * rows <= 16384, C % 4 == 0, C <= 1024, all arrays 16-byte aligned:
 *   forward : y = LayerNorm(x + dropout_p(z)) * gamma + beta and, with pos != NULL, yq = y + pos (the reference's
 *             with_pos_embed, :252-254, for the projection that consumes the result) in ONE launch; z / pos / yq / keep may be
 *             NULL; s_save, mean, rstd, keep are what the backward reads (all NULL: inference).  The dropout mask is the one
 *             snipper_add_dropout_layernorm_forward draws for the same (seed, element index).
 *   backward: ONE launch; the gradient of y is the SUM of g0 .. g3 (each may be NULL, not all): the outputs' consumers hand
 *             their gradients over separately (a gradient of yq counts as one of y).  dx = dL/dx, dz = dL/dz (masked, rescaled),
 *             dgamma / dbeta [C] are fully written (column workgroups of the same launch; deterministic). */
int snipper_small_ln_forward_f32(void *stream, const float *x, const float *z, const float *pos, const float *gamma, const float *beta,
                                 int rows, int C, float p, float eps, uint64_t seed, float *s_save, float *mean, float *rstd,
                                 uint8_t *keep, float *y, float *yq);
int snipper_small_ln_backward_f32(void *stream, const float *g0, const float *g1, const float *g2, const float *g3,
                                  const float *s_save, const float *mean, const float *rstd, const float *gamma, const uint8_t *keep,
                                  int rows, int C, float p, float *dx, float *dz, float *dgamma, float *dbeta);

/* Stem input: float32 images [N, 3, H, W] (planar; W % 4 == 0) -> bf16 [N, H, W, 4] with a zero fourth channel, the layout
 * snipper_stem7x7_bf16 reads. */
int snipper_stem_pack_bf16(void *stream, const float *x, int N, int H, int W, uint16_t *out);


/* 3x3 convolution, padding 1, stride 1 or 2, NHWC bf16, as an implicit GEMM on the same MFMA tiles:
 * Y[B,Ho,Wo,Cout] = act(conv(X[B,H,W,Cin], W[Cout,3,3,Cin]) + bias[Cout]), Ho = (H-1)/stride + 1 (same for Wo).
 * Requirements: Cin % 64 == 0, Cout % 4 == 0.  (ResNet bottleneck conv2 with the frozen BatchNorm folded in.) */
/* Data gradient of the STRIDE-2 case: dX[B,Hx,Wx,Cx] = sum_taps G[B,Hg,Wg,Cg] . Wt, Hg = (Hx-1)/2 + 1, with
 * Wt [Cx,3,3,Cg] = the convolution's weight with its channel roles swapped (NOT flipped).  Four launches of the same
 * kernel, one per parity class of the input pixel: a class uses 1, 2, 2 or 4 of the nine taps, so no MFMA multiplies a
 * structural zero (a zero-insertion formulation would waste 3/4 of them).  Cg % 64 == 0, Cx % 4 == 0; every pixel of
 * dX is written.
 * Weight gradient: dW[Cout,3,3,Cin] (+)= scale[co] * sum_{b,oy,ox} G[b,oy,ox,co] * X[b, oy*s+ky-1, ox*s+kx-1, ci] on the
 * split-reduction kernel of snipper_wgrad_bf16 (taps = extra output columns, the input pixel under a tap gathered by
 * the loader, zeros outside the image); float32 result in the channels_last layout of a convolution weight;
 * deterministic.  Cin % 128 == 0, Cout % 8 == 0, stride 1 or 2. */
/* The ResNet stem: Y[B,Ho,Wo,64] = conv7x7(stride 2, padding 3) of X4[B,H,W,4] (the 3-channel image padded to 4 bf16 channels,
 * NHWC) with Wp[64,256] = the weight (BN scale folded in) laid out as k = ky*32 + kx*4 + c, zero in the padding slots;
 * Ho = (H-1)/2 + 1.  Forward only (the stem is frozen, reference backbone.py:71-73); bias / ReLU / pooling follow in
 * snipper_stem_pool_bf16. */
int snipper_stem7x7_bf16(void *stream, const uint16_t *X4, const uint16_t *Wp, uint16_t *Y, int B, int H, int W);
int snipper_conv3x3_dgrad_s2_bf16(void *stream, const uint16_t *G, const uint16_t *Wt, uint16_t *dX,
                                  int B, int Hx, int Wx, int Cx, int Cg, const uint16_t *gate);
size_t snipper_wgrad_conv3x3_workspace_bytes(int B, int H, int W, int Cin, int Cout, int stride);
int snipper_wgrad_conv3x3_bf16(void *stream, const uint16_t *G, const uint16_t *X, int B, int H, int W, int Cin, int Cout,
                               int stride, const float *scale, float *dW, int accumulate, void *workspace,
                               size_t workspace_bytes);
/* `gate` (optional, the layout of the result, 16-byte aligned, not together with relu): results whose gate value is not > 0
 * are written as 0.  A data gradient is itself such a convolution; when the activation it differentiates came out of a
 * ReLU, passing that activation as the gate does the ReLU's backward in the store phase (no separate pass).
 * `flip_taps` != 0: tap (ky, kx) multiplies W[:, 2 - ky, 2 - kx, :] -- with the weight's channel axes swapped that is the
 * stride-1 data gradient (no flipped copy of the weight). */
int snipper_conv3x3_bf16(void *stream, const uint16_t *X, const uint16_t *W, const float *bias, uint16_t *Y,
                         int B, int H, int Wd, int Cin, int Cout, int stride, int relu, const uint16_t *gate, int flip_taps);
/* Stride-1 3x3 convolution with the input patch of a 2-D output tile resident in LDS and the weight streamed from a PACKED
 * copy (csrc/conv3x3_patch_bf16.cuh; same arithmetic as snipper_conv3x3_bf16 with stride 1 up to the order of the float32
 * accumulation).  Replaces the same cuDNN / torchvision convolutions as snipper_conv3x3_bf16 (reference models/backbone.py:67-111).
 *   snipper_conv3x3_pack_bf16: packs n weights in one launch.  src[i] = [cout[i]][3][3][cin[i]] bf16 (a channels_last conv
 *     weight), dst[i] = cout[i] * 9 * cin[i] bf16, 16-byte aligned, cout and cin multiples of 64.  transposed[i] == 0: the
 *     forward's weight.  != 0: the stride-1 DATA GRADIENT's weight (channel roles swapped, taps reversed): pass it to
 *     snipper_conv3x3_patch_bf16 with Cin = cout[i], Cout = cin[i].
 *   snipper_conv3x3_patch_supported: 1 when snipper_conv3x3_patch_bf16 takes the shape (Cin, Cout multiples of 64, W >= 4,
 *     32-bit byte offsets), else 0.
 *   snipper_conv3x3_patch_bf16: Y = act(conv(X, W, padding 1) + bias); X [B][H][W][Cin], Y [B][H][W][Cout] bf16; relu / gate as
 *     snipper_conv3x3_bf16. */
int snipper_conv3x3_pack_bf16(void *stream, int n, const void *const *src, void *const *dst, const int *cout, const int *cin,
                              const int *transposed);
int snipper_conv3x3_patch_supported(int B, int H, int Wd, int Cin, int Cout);
int snipper_conv3x3_patch_bf16(void *stream, const uint16_t *X, const uint16_t *Wp, const float *bias, uint16_t *Y,
                               int B, int H, int Wd, int Cin, int Cout, int relu, const uint16_t *gate);
/* The same machinery for a plain product on rows (the 1x1 convolutions of the ResNet body in NHWC, reference
 * models/backbone.py:67-111, and Linears): Y[M, N] = act(X[M, K] . W^T + bias + res), X / Y / res / gate contiguous rows,
 * N, K multiples of 64.
 *   snipper_linear_pack_bf16: src[i] = [N[i]][K[i]] bf16 row-major -> dst[i]; transposed[i] != 0 packs the data gradient's
 *     operand (pass it to snipper_linear_patch_bf16 with N and K swapped: dX[M, K] = dY[M, N] . W).
 *   snipper_linear_patch_bf16: relu / gate as above (not both); res (8-byte aligned) is added before the activation;
 *     bn = 0 (the launcher picks 128 or 64 output columns per workgroup), 64 or 128. */
int snipper_linear_pack_bf16(void *stream, int n, const void *const *src, void *const *dst, const int *N, const int *K,
                             const int *transposed);
int snipper_linear_patch_supported(long long M, int N, int K);
int snipper_linear_patch_bf16(void *stream, const uint16_t *X, const uint16_t *Wp, const float *bias, const uint16_t *res,
                              uint16_t *Y, int M, int N, int K, int relu, const uint16_t *gate, int bn);

/* Deep reductions into 384 columns on full-width tiles (csrc/conv3x3_patch_bf16.cuh, linear_wide_kernel): Y[M, 384] =
 * X[M, K] . W^T + bias, K a multiple of 128, M >= 8192, X / Y contiguous rows, Wp = snipper_linear_pack_bf16's pack of W
 * [384][K] (transposed == 0) or, for a data gradient dX[M, 384] = dY[M, K] . W with W [K][384], its transposed pack.  The
 * encoder feed-forward block's linear2 forward and linear1 data gradient (reference models/deformable_transformer.py:194-198). */
int snipper_linear_wide_supported(long long M, int N, int K);
int snipper_linear_wide_bf16(void *stream, const uint16_t *X, const uint16_t *Wp, const float *bias, uint16_t *Y, int M, int N, int K);

/* out[bt][S][C] (bf16) = cat over levels of pos[l][bt][hw[l]][C] (float32) + level_embed[l][C]: the encoder's position +
 * level embedding (reference models/deformable_transformer.py:118-121) written once in the dtype its kernels read. */
int snipper_level_pos_bf16(void *stream, const float *const *pos, const int *hw, int levels, const float *level_embed, int bt,
                           int C, uint16_t *out);

/* bf16 working copies of many float32 tensors in one launch: dst[e] = bf16(src[e] * scale[e / inner]) (scale == NULL: a plain
 * cast).  `items` is a DEVICE array of n_items records {const float *src; uint16_t *dst; const float *scale; int64 numel;
 * int32 inner; int32 pad} (40 bytes; numel % 8 == 0; inner % 8 == 0 where a scale is given; 16-byte aligned src / dst),
 * `block_end` a DEVICE array of the running count of 2 048-element blocks (block_end[n_items - 1] == n_blocks).  Replaces the
 * autocast casts of the weights (and the frozen-BatchNorm folding, reference models/backbone.py:27-64) once per step. */
int snipper_cast_scale_table_bf16(void *stream, const void *items, const int *block_end, int n_items, int n_blocks);

/* ---- heat-map targets and loss of the criterion (csrc/heatmap_loss.cuh; reference models/model.py:447-483) ----------------
 * snipper_heatmap_scatter_f32: one-hot joint maps of all levels in one launch.  kpts [n_person][Tk][K][3] float32 (x, y in
 *   [0, 1), visibility), sample [n_person] int64; level l's maps [bs][K][T][h[l]][w[l]] start at element base[l] of `out`,
 *   which the caller has zeroed; a visible joint whose pixel (trunc(x * w), trunc(y * h)) lies inside the map stores 1.0.
 * snipper_heatmap_loss_forward_f32: partial[i] = block sums of (tm - mem)^2 over the first K channels of each of the nhead
 *   heads of every position; mem [bs * T * S][C] float32 (the encoder memory), tm[l] the level's blurred targets
 *   [bs][K][T][hw[l]], start[l] the level's first position in S (levels tile S in order).  The loss is sum(partial) / nhead.
 * snipper_heatmap_loss_backward_f32: gmem [bs * T * S][C] = gscale[0] * 2 * (mem - tm) on those channels, 0 elsewhere
 *   (every element written); gscale is a DEVICE scalar. */
int snipper_heatmap_scatter_f32(void *stream, const float *kpts, const long long *sample, int n_person, int Tk, int T, int K,
                                int levels, const int *h, const int *w, const long long *base, float *out);
int snipper_heatmap_loss_forward_f32(void *stream, const float *mem, const float *const *tm, const int *hw, const int *start,
                                     int levels, int bs, int T, int S, int C, int nhead, int K, float *partial, int n_partial);
int snipper_heatmap_loss_backward_f32(void *stream, const float *mem, const float *const *tm, const int *hw, const int *start,
                                      int levels, int bs, int T, int S, int C, int nhead, int K, const float *gscale, float *gmem);

/* ---- element-wise fusions around the core op (csrc/msda_prologue.cuh) --------------------------------
 * dtype codes: 0 = float32, 1 = bfloat16 bits.
 *
 * snipper_temporal_mix: out[n,to,s,c] = sum_ti mix[to*Ti + ti] * in[n,ti,s,c]   (Ti, To <= 8, C % 4 == 0)
 *   mask [N, (mask_on_input ? Ti : To), S] bytes or NULL: a non-zero byte makes that frame position read as 0
 *   (mask_on_input, the forward: value.masked_fill of ms_deform_attn.py:116 folded in) or be written as 0
 *   (the backward).  `mix` is a HOST array of To*Ti floats (copied into the launch).
 *   Replaces masked_fill + the per-frame core-op loop's frame averaging + dtype casts. */
int snipper_temporal_mix(void *stream, const void *in, int in_dtype, const unsigned char *mask, int mask_on_input,
                         const float *mix, int N, int Ti, int To, long long S, int C, void *out, int out_dtype);
/* The same with either side in the head-major layout of snipper_msda_config.value_layout = 1: `in` [N, Ti, M, S, head_dim]
 * (in_head_major) and / or `out` [N, To, M, S, head_dim] (out_head_major), M = C / head_dim; head_dim % 8 == 0, C % 8 == 0,
 * Ti, To <= 4, 16-byte aligned arrays.  All flags 0 = snipper_temporal_mix. */
int snipper_temporal_mix_ex(void *stream, const void *in, int in_dtype, const unsigned char *mask, int mask_on_input,
                            const float *mix, int N, int Ti, int To, long long S, int C, void *out, int out_dtype,
                            int head_dim, int in_head_major, int out_head_major);

/* snipper_msda_prologue_forward: rows = N*T*Lq*M, one row = the L*P samples of a (query, head) (L*P <= 16, L <= 8)
 *   loc[row, l, p, :] = ref[row / M, l, :] + off[row, l, p, :] * (inv_w[l], inv_h[l])       (ms_deform_attn.py:164-165)
 *   prob[row, :]      = softmax(logit[row, :])                                               (:149, tied weights)
 *   off / logit: `dtype`, addressed per query with leading dimensions in elements -- row (q, m) reads
 *   off + q * off_ld + m * 2*L*P and logit + q * logit_ld + m * L*P (off_ld = M*L*P*2, logit_ld = M*L*P for separate
 *   contiguous arrays; both may be column slices of one merged projection output);
 *   ref, loc, prob: float32 contiguous;  inv_w / inv_h: HOST arrays of L floats.
 * snipper_msda_prologue_backward: the adjoint, grad_off / grad_logit with the same addressing; grad_ref may be NULL;
 * otherwise M must be a power of two <= 64. */
int snipper_msda_prologue_forward(void *stream, const void *off, long long off_ld, const void *logit, long long logit_ld,
                                  int dtype, const float *ref, const float *inv_w, const float *inv_h, long long rows,
                                  int M, int L, int P, float *loc, float *prob);
/* The same with the offsets' BIAS added inside, in float32: off_bias [M*L*P*2] float32 or NULL; `off` then holds W q only
 * (a projection launched without its bias).  Under bf16 autocast this keeps the bias grid of the reference initialisation
 * (ms_deform_attn.py:82-90, up to P pixels) out of the projection's bf16 output; the adjoint is unchanged (the bias
 * gradient is the column sum of grad_off either way). */
int snipper_msda_prologue_forward_ex(void *stream, const void *off, long long off_ld, const void *logit, long long logit_ld,
                                     int dtype, const float *off_bias, const float *ref, const float *inv_w,
                                     const float *inv_h, long long rows, int M, int L, int P, float *loc, float *prob);
int snipper_msda_prologue_backward(void *stream, const float *grad_loc, const float *grad_prob, const float *prob,
                                   const float *inv_w, const float *inv_h, long long rows, int M, int L, int P,
                                   void *grad_off, long long grad_off_ld, void *grad_logit, long long grad_logit_ld,
                                   int dtype, float *grad_ref);

/* ---- the tied spatiotemporal module core in one call ------------------------------------------------------
 * What MSDeformAttn.forward does between its value projection and its output projection when the per-frame
 * offset / weight Linears are one module (reference models/ops/modules/ms_deform_attn.py:130-233, see DESIGN.md
 * section 4): out[n, t1] = core_op(mean_{t2 in g(t1)} masked value[n, t2], ref + offsets, softmax(logits)).
 *   value [N,T2,S,M*D] (value_dtype 0 = f32 / 1 = bf16), mask [N,T2,S] bytes or NULL (non-zero = padding),
 *   mix: HOST [T1*T2] floats, mix[t1*T2 + t2] = 1/|g(t1)| for t2 in g(t1) else 0 (the neighbour table);
 *   off / logit (+ per-query leading dimensions, dtype ql_dtype) and ref [N*T1*Lq, L, 2] as for
 *   snipper_msda_prologue_forward; inv_w / inv_h HOST [L]; shapes / level_start device int64.
 *   Outputs: vbar [N*T1,S,M,D] f32, loc [N*T1,Lq,M,L,P,2] f32, prob [N*T1,Lq,M,L,P] f32 (kept by the caller for the
 *   backward) and out [N*T1,Lq,M*D] (f32, or bf16 rows when out_bf16 != 0 and D == 48).
 * Backward: grad_out (f32 or bf16 rows) -> grad_value [N,T2,S,M*D] (value_dtype), grad_off / grad_logit (ql_dtype,
 * same addressing as off / logit), grad_ref [N*T1*Lq,L,2] or NULL.  host_shapes (HOST [L,2], may be NULL) enables the
 * owner-computes backward (csrc/msda_d48_patch.cuh); workspace >= snipper_st_msda_backward_workspace_bytes(...). */
int snipper_st_msda_forward(void *stream, const void *value, int value_dtype, const unsigned char *mask, const float *mix,
                            const void *off, long long off_ld, const void *logit, long long logit_ld, int ql_dtype,
                            const float *ref, const float *inv_w, const float *inv_h,
                            const int64_t *shapes, const int64_t *level_start, const int64_t *host_shapes,
                            int N, int T1, int T2, int S, int M, int D, int L, int Lq, int P,
                            float *vbar, float *loc, float *prob, void *out, int out_bf16);
size_t snipper_st_msda_backward_workspace_bytes(int N, int T1, int S, int M, int D, int L, int Lq, int P,
                                                const int64_t *host_shapes);
int snipper_st_msda_backward(void *stream, const void *grad_out, int grad_out_bf16, const float *vbar, const float *loc,
                             const float *prob, const unsigned char *mask, const float *mix,
                             const float *inv_w, const float *inv_h, const int64_t *shapes, const int64_t *level_start,
                             const int64_t *host_shapes, int N, int T1, int T2, int S, int M, int D, int L, int Lq, int P,
                             void *workspace, size_t workspace_bytes,
                             void *grad_value, int value_dtype, void *grad_off, long long grad_off_ld,
                             void *grad_logit, long long grad_logit_ld, int ql_dtype, float *grad_ref);

/* ---- the keypoint / depth / continuity loss terms of SetCriterion (csrc/pair_losses.cuh) ---------------------------
 * Reference models/model.py:289-427 (loss_root, loss_joint, loss_joint_disp, loss_joint_cont) for all matched
 * (prediction, target) pairs of all decoder layers at once.  P = n_layers * pairs rows, layer-major:
 *   sk [P,T,K,3] predicted (x, y, vis), sd [P,T,K,1] predicted depth, tk [P,T,K,3] target (x, y, vis),
 *   td [P,T,K,2] target (depth, valid), cont_w [K], max_depth: DEVICE scalar, eps: the reference's 10e-6.
 *   out [P,9] per-pair terms in the order root, root_depth, root_vis, joint_disp, joint_depth_disp, joint, joint_depth,
 *   joint_vis, cont (the caller sums over the pairs of a layer and divides by the number of trajectories).
 * Backward: grad_terms [P, 9] = dL/d(out) -> grad_sk, grad_sd.
 * T * K <= 128, float32. */
/* Gaussian blur of the heat-map targets (reference models/model.py:447-483: torchvision gaussian_blur of the one-hot joint
 * maps): out[i] = separable ksize x ksize blur of min(in[i], clamp_max) with reflect padding, n_images maps of H x W
 * float32.  `weights` is a HOST array of ksize taps (odd, <= 31, ksize / 2 < min(H, W)); in != out. */
int snipper_heatmap_blur_f32(void *stream, const float *in, float *out, int n_images, int H, int W, int ksize,
                             const float *weights, float clamp_max);
int snipper_pair_losses_forward(void *stream, const float *sk, const float *sd, const float *tk, const float *td,
                                const float *cont_w, const float *max_depth, int n_layers, int pairs, int T, int K,
                                float eps, float *out);
int snipper_pair_losses_backward(void *stream, const float *sk, const float *sd, const float *tk, const float *td,
                                 const float *cont_w, const float *max_depth, const float *grad_terms,
                                 int n_layers, int pairs, int T, int K, float eps, float *grad_sk, float *grad_sd);

/* ---- the matcher's cost matrix (csrc/match_cost.cuh) ------------------------------------------------------
 * Replaces the ~60 broadcast / reduction launches per sample of models/matcher.py:60-127.  One entry per (layer l,
 * query q, target m): out[l][q][m] = w0 class + w1 root + w2 root_vis + w3 root_depth + w4 joint + w5 joint_vis +
 * w6 joint_depth (weights7 is a HOST array in that order; the terms as defined in the file header).
 * kpts [L][*][Q][T][K][3], depth [L][*][Q][T][K][1], logits [L][*][Q][T][2] float32 device memory, addressed through
 * the layer / query strides and the keypoint stride kp_sk / d_sk (elements; 3 / 1 when dense, 4 / 4 for slices of one
 * [..., K, 4] head output) so that a per-sample slice of the [L, bs, Q, ...] tensors needs no copy;
 * tgt_kpts [M][T][K][3], tgt_depth [M][T][K][2] contiguous; max_depth a device scalar; out [L][Q][M]. */
int snipper_match_cost_f32(void *stream, const float *kpts, long long kp_sl, long long kp_sq, int kp_sk, const float *depth,
                           long long d_sl, long long d_sq, int d_sk, const float *logits, long long lg_sl, long long lg_sq,
                           const float *tgt_kpts, const float *tgt_depth, const float *max_depth,
                           int L, int Q, int M, int T, int K, const float *weights7, float eps, float *out);

/* ---- decoder reference-point refinement (csrc/match_cost.cuh) ---------------------------------------------
 * models/deformable_transformer.py:329-333 + util/misc.py:481-485 + :319-321 in one launch:
 *   new_ref[row] = sigmoid(delta[row, 0:2] + inverse_sigmoid(ref[row]))     (clamps as the reference, eps = 1e-5)
 *   ref_in[row, l] = new_ref[row] * valid_ratios[row / rows_per_batch, l]
 * delta: float32 rows with leading dimension ld_delta (>= 2; e.g. the [.., 4] output of the root head); ref, new_ref
 * [rows, 2]; valid_ratios [rows / rows_per_batch, L, 2]; ref_in [rows, L, 2].  Forward only: the reference detaches. */
int snipper_refine_reference_f32(void *stream, const float *delta, long long ld_delta, const float *ref,
                                 const float *valid_ratios, int rows, int rows_per_batch, int L, float eps,
                                 float *new_ref, float *ref_in);
/* The same with the head evaluated in place: delta[row][c] = x[row] . W[c] + b[c] for c = 0, 1 -- the root head of the
 * reference's model is ONE Linear(C -> 4) (models/model.py:95) of which only the first two outputs reach the refinement
 * (models/deformable_transformer.py:329-333); x [rows][C], W [>= 2][C] row-major, b [>= 2] or NULL; C % 4 == 0. */
int snipper_refine_reference_linear_f32(void *stream, const float *x, const float *W, const float *b, const float *ref,
                                        const float *valid_ratios, int rows, int C, int rows_per_batch, int L, float eps,
                                        float *new_ref, float *ref_in);

/* ---- Hungarian matching on the device (csrc/lsap.cuh) ---------------------------------------------------
 * Replaces the host round trip of models/matcher.py:132 (`linear_sum_assignment(cost.cpu())`).
 * cost [P, n, m] float32 (P independent problems, n predictions, m targets, 1 <= m <= n <= 64), device memory.
 * out_src / out_tgt [P, m] int64: the matched (prediction, target) pairs of each problem, predictions ascending
 * (SciPy's order).  Float64 arithmetic inside. */
int snipper_lsap_f32(void *stream, const float *cost, int P, int n, int m, long long *out_src, long long *out_tgt);

/* ---- gradient clipping + AdamW on one flat float32 buffer (csrc/adamw_flat.cuh) -------------------------------
 * Replaces `torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm)` (engine.py:74) + `torch.optim.AdamW.step()` over the
 * reference's parameter groups (main.py:201-221) when parameters, gradients and both moments live in flat buffers of n
 * elements (n % 4 == 0, 16-byte aligned; snipper_amd/flat_params.py).  Two launches, deterministic:
 *   snipper_gradnorm_partials_f32   partials[nparts] (nparts <= 4096) = per-workgroup sums of grad^2
 *   snipper_adamw_clip_f32          coef = min(1, max_norm / (sqrt(sum partials) + 1e-6)) (max_norm <= 0 or partials == NULL:
 *                                   no clipping), then per element of every group k (elements [seg_begin[k], seg_end[k]),
 *                                   multiples of 4, lr / weight decay seg_lr[k] / seg_wd[k]; HOST arrays, nseg <= 8):
 *                                   p *= 1 - lr wd;  m += (coef g - m)(1 - beta1);  v = beta2 v + (1 - beta2)(coef g)^2;
 *                                   p -= lr / (1 - beta1^step) * m / (sqrt(v) / sqrt(1 - beta2^step) + eps)
 *                                   (torch.optim.AdamW, amsgrad = False, maximize = False; step counts from 1).  Elements
 *                                   outside every group are left alone; the gradient is not modified; norm_out (device, or
 *                                   NULL) receives the unclipped global norm. */
int snipper_gradnorm_partials_f32(void *stream, const float *grad, long long n, float *partials, int nparts);
int snipper_adamw_clip_f32(void *stream, float *param, const float *grad, float *exp_avg, float *exp_avg_sq, long long n,
                           const long long *seg_begin, const long long *seg_end, const float *seg_lr, const float *seg_wd, int nseg,
                           float beta1, float beta2, float eps, long long step, const float *partials, int nparts,
                           float max_norm, float *norm_out);

#ifdef __cplusplus
}
#endif
#endif /* SNIPPER_DENSE_H_ */
