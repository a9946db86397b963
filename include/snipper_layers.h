/*
 * snipper_layers.h -- one native call per transformer layer and direction (C ABI of libsnipper_msda.so).
 *
 * Reference: DeformableTransformerDecoderLayer.forward, models/deformable_transformer.py:276-300 (self-attention over the
 * T * nq object queries, residual + LayerNorm, deformable cross attention into the encoder memory, residual + LayerNorm,
 * feed-forward block, residual + LayerNorm) and the reference-point refinement that follows it in
 * DeformableTransformerDecoder.forward (:329-333).  In the reference every one of those steps is a PyTorch module call; in this
 * package's Python host they were ~14 kernel wrappers forward and ~12 backward per layer, each an autograd node with its own
 * allocations and argument marshalling -- 0.5-0.6 ms of host time per layer and direction, which is what kept the host-issue
 * time of a training step (18.5 ms) within 3 ms of the GPU's.  The two entry points below sequence the SAME launches, in the same
 * order, with the same arguments, from one call: the caller (snipper_amd/decoder_native.py) allocates the outputs and ONE arena,
 * packs one argument block, and owns a single autograd node per layer.  Results are bit-identical to the per-module sequence.
 *
 * Conventions as snipper_msda.h: device pointers unless marked HOST, `stream` = hipStream_t as void*, returns 0 or the first
 * non-zero code of a constituent call, no allocation, no host synchronisation, no global state.
 *
 * Scope: float32 decoder rows (R = bs * tokens <= 16384), heads of 32 or 48 channels, tokens <= 384, tied offset / weight
 * Linears, the cross attention's VALUE already projected by the caller ([N = bs * frames, S, heads, C / heads] bfloat16 or
 * float32: the projection of the 79 000 memory rows is a full-size product that belongs with the encoder-size kernels).
 */
#ifndef SNIPPER_LAYERS_H_
#define SNIPPER_LAYERS_H_

#include <stddef.h>
#include <stdint.h>
#include "snipper_msda.h"
#include "snipper_dense.h"

#ifdef __cplusplus
extern "C" {
#endif

#define SNIPPER_LAYERS_ABI_VERSION 1

typedef struct snipper_decoder_layer_dims {
  int32_t struct_bytes;     /* sizeof(snipper_decoder_layer_dims), checked                                              */
  int32_t bs, tokens;       /* samples; object queries per sample = frames * queries_per_frame (self-attention length)  */
  int32_t frames, queries;  /* query frames T1 and queries per frame: the core op sees N = bs * frames, Lq = queries     */
  int32_t C, heads, d_ffn;  /* model width, attention heads (self- and cross-attention alike), feed-forward width       */
  int32_t levels, points;   /* L, P of the deformable cross attention                                                   */
  int32_t S;                /* memory positions per frame                                                               */
  int32_t value_dtype;      /* 0 float32, 1 bfloat16                                                                    */
  float p_attn, p_norm2, p_norm1, p_ffn, p_norm3;   /* dropout rates of this call (0 in eval mode)                      */
  float eps_norm2, eps_norm1, eps_norm3;
  float attn_scale;         /* softmax scale of the self-attention, (C / heads)^-0.5 as the caller rounds it              */
  uint64_t seed_attn, seed_norm2, seed_norm1, seed_ffn, seed_norm3;   /* the forward's seeds; the backward is given the same */
} snipper_decoder_layer_dims;

/* the layer's 20 parameter tensors (float32, contiguous), in state_dict order of the reference layer: cross_attn.
 * {sampling_offsets, attention_weights, output_proj} (value_proj is the caller's), norm1, self_attn.{in_proj, out_proj}, norm2,
 * linear1, linear2, norm3.  The same struct, non-const in spirit, names the 20 gradient outputs of the backward. */
typedef struct snipper_decoder_layer_params {
  float *so_w, *so_b, *aw_w, *aw_b, *op_w, *op_b, *norm1_w, *norm1_b, *in_proj_w, *in_proj_b, *out_proj_w, *out_proj_b,
      *norm2_w, *norm2_b, *lin1_w, *lin1_b, *lin2_w, *lin2_b, *norm3_w, *norm3_b;
} snipper_decoder_layer_params;

typedef struct snipper_decoder_layer_fwd {
  snipper_decoder_layer_dims d;
  snipper_decoder_layer_params w;             /* read only                                                                */
  /* the layer input as three handles, all [bs * tokens, C] float32: value input of the self-attention, residual of norm2,
   * input + query_pos (the q / k input); pos_a / pos_b = query_pos (added to norm2's and norm3's results for the projections
   * that follow; pos_b NULL on the last layer) */
  const float *x_v, *x_res, *x_q, *pos_a, *pos_b;
  const void *value;                          /* [bs * frames, S, heads, C / heads], value_dtype                          */
  const int64_t *shapes, *level_start;        /* [levels, 2] (H, W), [levels]                                             */
  const int64_t *host_shapes;                 /* HOST copy of shapes, or NULL                                             */
  const float *ref_in;                        /* [bs * tokens, levels, 2] reference points times valid ratios             */
  const float *inv_w, *inv_h;                 /* HOST [levels]: 1 / W_l, 1 / H_l                                          */
  /* reference-point refinement after the layer (root_w NULL: skipped): root head Linear(C -> >= 2) [rows 0, 1 used]       */
  const float *root_w, *root_b, *ref_points, *valid_ratios;
  float *new_ref, *ref_in_next;               /* [bs * tokens, 2], [bs * tokens, levels, 2]                               */
  /* outputs */
  float *out;                                 /* [bs * tokens, C] norm3's result                                          */
  float *out_q;                               /* out + pos_b, or NULL                                                     */
  float *loc, *prob;                          /* [bs * tokens * heads, levels, points, 2] / [.., levels, points]          */
  void *arena; size_t arena_bytes;            /* >= snipper_decoder_layer_arena_bytes(&d): saved + scratch tensors        */
} snipper_decoder_layer_fwd;

typedef struct snipper_decoder_layer_bwd {
  snipper_decoder_layer_dims d;
  snipper_decoder_layer_params w;             /* read only                                                                */
  snipper_decoder_layer_params dw;            /* the 20 gradients, fully written                                          */
  const float *g[4];                          /* gradients of `out` (and of out_q), summed; each may be NULL, not all     */
  const float *x_v, *x_q;                     /* the forward's inputs                                                     */
  const void *value; const int64_t *shapes, *level_start, *host_shapes;
  const float *inv_w, *inv_h;                 /* HOST                                                                     */
  const float *loc, *prob;                    /* the forward's outputs                                                    */
  const void *arena;                          /* the forward's arena, unchanged                                           */
  void *scratch; size_t scratch_bytes;        /* >= snipper_decoder_layer_scratch_bytes(&d)                               */
  /* outputs */
  float *d_xv, *d_xres, *d_xq;                /* [bs * tokens, C] each                                                    */
  float *d_pos_a;                             /* gradient of pos_a (= of norm2's position-added copy)                     */
  float *d_ref_in;                            /* gradient of ref_in [bs * tokens, levels, 2], or NULL (a refined, detached
                                                 reference: every layer but the first, reference :329-333)                */
  void *d_value;                              /* [bs * frames, S, heads, C / heads]: bfloat16 for a bfloat16 value (fully
                                                 written: zeroed + touched rows), float32 for a float32 value             */
} snipper_decoder_layer_bwd;

/* ---------------------------------------------------------------------------------------------------------------------------
 * Encoder layer under bf16 autocast (reference DeformableTransformerEncoderLayer.forward, models/deformable_transformer.py:
 * 200-216: deformable self-attention over the memory with tied per-frame Linears -- ms_deform_attn.py:99-243 --, residual +
 * LayerNorm, feed-forward block, residual + LayerNorm) as the launches of DeformableTransformerEncoderLayer.forward_fused in ONE
 * call per direction: 9 launches forward, ~25 backward on 79 000 token rows.  float32 residual stream with "lazy" LayerNorm
 * inputs (csrc/ln_fused.cuh), bf16 activations, bf16 weight shadows (snipper_amd/shadow.py) and their transposes / packs, the
 * bf16 temporal mean head-major (snipper_msda_config.value_layout = 1) when cfg says so; no padding mask (the training case:
 * every snippet is warped to the input size, datasets/transforms.py:137-144).  R = bs * frames * S rows. */
typedef struct snipper_encoder_layer_dims {
  int32_t struct_bytes;
  int32_t bs, frames, S;                 /* R = bs * frames * S token rows; the core op sees N = bs * frames, Lq = S              */
  int32_t C, heads, d_ffn, levels, points;
  int32_t last;                          /* 1: norm2's float32 result is written (y32) and no position-added copy is made         */
  int32_t head_major;                    /* layout of the temporal mean / its gradient (must match cfg->value_layout)             */
  int32_t reserved;
  float p_norm1, p_ffn, p_norm2, eps_norm1, eps_norm2;
  float reserved_f;
  uint64_t seed_norm1, seed_ffn, seed_norm2;
} snipper_encoder_layer_dims;

typedef struct snipper_encoder_layer_weights {
  /* bf16 shadows: value_proj [C,C], merged offsets|logits [3 heads L P, C], output_proj [C,C], linear1 [d_ffn,C], linear2 packed
   * for snipper_linear_wide_bf16; backward: the transposes W^T of value_proj / merged / output_proj / linear2 and the
   * transposed pack of linear1 */
  const uint16_t *wv, *wm, *wo, *w1, *w2_packed, *wv_t, *wm_t, *wo_t, *w2_t, *w1_tpacked;
  /* float32: biases (bm = [0; attention_weights.bias]: the offsets' bias so_bias is added in float32 by the prologue),
   * LayerNorm gains / shifts */
  const float *bv, *bm, *so_bias, *bo, *b1, *b2, *norm1_w, *norm1_b, *norm2_w, *norm2_b;
} snipper_encoder_layer_weights;

typedef struct snipper_encoder_layer_fwd {
  snipper_encoder_layer_dims d;
  snipper_encoder_layer_weights w;
  const snipper_msda_config *cfg;        /* HOST, or NULL                                                                         */
  const float *x32;                      /* [R, C] residual stream -- or, with x_mean != NULL, the producing LayerNorm's saved sum   */
  const float *x_mean, *x_rstd, *x_gamma, *x_beta;
  const uint16_t *src16, *q16, *pos16;   /* bf16 [R, C]: the stream, stream + pos, pos (pos16 NULL when d.last)                    */
  const float *ref;                      /* [R, levels, 2]                                                                         */
  const int64_t *shapes, *level_start, *host_shapes;
  const float *inv_w, *inv_h, *mix;      /* HOST: [levels], [levels], [frames * frames] temporal mix                               */
  /* outputs */
  float *s2, *mean2, *rstd2;             /* norm2's pre-norm sum and statistics (the lazy form of the float32 result)              */
  float *y32;                            /* norm2's float32 result, or NULL (lazy)                                                 */
  uint16_t *y16, *yq16;                  /* bf16 result, bf16(result + pos) (NULL when d.last)                                     */
  void *arena; size_t arena_bytes;       /* >= snipper_encoder_layer_arena_bytes(&d): what the backward reads + scratch           */
} snipper_encoder_layer_fwd;

typedef struct snipper_encoder_layer_grads {
  float *wv, *bv, *wm, *bm, *wo, *bo, *w1, *b1, *w2, *b2, *norm1_w, *norm1_b, *norm2_w, *norm2_b;
} snipper_encoder_layer_grads;            /* wm [3 heads L P, C], bm [3 heads L P]: the caller splits them into the two Linears'   */

typedef struct snipper_encoder_layer_bwd {
  snipper_encoder_layer_dims d;
  snipper_encoder_layer_weights w;
  snipper_encoder_layer_grads dw;
  const snipper_msda_config *cfg;
  const float *g32; const uint16_t *g16, *gq16;      /* gradients of the float32 / bf16 / position-added results, each may be NULL */
  const uint16_t *src16, *q16;           /* the forward's inputs                                                                    */
  const float *s2, *mean2, *rstd2;       /* the forward's outputs                                                                   */
  const int64_t *shapes, *level_start, *host_shapes;
  const float *inv_w, *inv_h, *mix_t;    /* HOST; mix_t = the transposed mix                                                        */
  const void *arena;                     /* the forward's arena                                                                     */
  void *scratch; size_t scratch_bytes;   /* >= snipper_encoder_layer_scratch_bytes(&d, cfg, host_shapes)                            */
  /* outputs */
  float *d_x32; uint16_t *d_src16, *d_q16;
} snipper_encoder_layer_bwd;

int snipper_encoder_layer_supported(const snipper_encoder_layer_dims *d);
size_t snipper_encoder_layer_arena_bytes(const snipper_encoder_layer_dims *d);
size_t snipper_encoder_layer_scratch_bytes(const snipper_encoder_layer_dims *d, const snipper_msda_config *cfg, const int64_t *host_shapes);
int snipper_encoder_layer_forward(void *stream, const snipper_encoder_layer_fwd *a);
int snipper_encoder_layer_backward(void *stream, const snipper_encoder_layer_bwd *a);

int snipper_layers_abi_version(void);
/* 1 when the composites take this layer shape, 0 otherwise (then the caller sequences the per-module calls itself). */
int snipper_decoder_layer_supported(const snipper_decoder_layer_dims *d);
size_t snipper_decoder_layer_arena_bytes(const snipper_decoder_layer_dims *d);
size_t snipper_decoder_layer_scratch_bytes(const snipper_decoder_layer_dims *d);
int snipper_decoder_layer_forward(void *stream, const snipper_decoder_layer_fwd *a);
int snipper_decoder_layer_backward(void *stream, const snipper_decoder_layer_bwd *a);

#ifdef __cplusplus
}
#endif
#endif /* SNIPPER_LAYERS_H_ */
