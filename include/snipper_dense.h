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

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Y[M,N] = act( X[M,K] . W[N,K]^T + bias[N] + R[M,N] ),  act = ReLU if relu != 0 else identity.
 * X, W, R, Y: bfloat16 bits (uint16_t); bias: float32 or NULL; R: NULL for no residual.
 * ldx / ldr / ldy: row strides in ELEMENTS (>= K / N / N).  Requirements: K % 64 == 0, N % 4 == 0,
 * X/W rows 16-byte aligned (ldx % 8 == 0), Y/R rows 8-byte aligned (ldy, ldr % 4 == 0).
 * Accumulation in float32, one rounding to bf16 at the end. */
int snipper_linear_bf16(void *stream, const uint16_t *X, long long ldx, const uint16_t *W,
                        const float *bias, const uint16_t *R, long long ldr, uint16_t *Y, long long ldy,
                        int M, int N, int K, int relu);

#ifdef __cplusplus
}
#endif
#endif /* SNIPPER_DENSE_H_ */
