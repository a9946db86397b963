"""Autograd boundary of the deformable-attention core op.

Mirrors /root/reference/models/ops/functions/ms_deform_attn_func.py: the same two public names
with the same call signatures, so the reference's modules can import them unchanged.

* ``MSDeformAttnFunction``          autograd.Function over the gfx950 HIP kernels
                                    (reference :24-42 over its CUDA extension).
* ``ms_deform_attn_core_pytorch``   the reference's pure-PyTorch debug formulation (:45-65),
                                    kept because it is part of the public surface
                                    (``use_pytorch_deform=1``).  It is never used as a fallback:
                                    ``MSDeformAttnFunction`` raises if the HIP library is absent.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import MultiScaleDeformableAttention as MSDA


class MSDeformAttnFunction(Function):
    """apply(value, spatial_shapes, level_start_index, sampling_locations, attention_weights, im2col_step
             [, rows_bf16=False[, rows_f32=False]])

    ``rows_bf16`` (extension, float32 ``value`` only): the output is produced, and its gradient consumed, as bfloat16
    rows by the kernels themselves -- for callers whose neighbouring projections compute in bf16.  ``rows_f32`` (extension,
    bfloat16 ``value`` only): float32 output rows and a float32 gradient for them -- a float32 consumer beside a bf16 value
    (the decoder's cross attention under bf16 autocast), no cast launch either way."""

    @staticmethod
    def forward(ctx, value, value_spatial_shapes, value_level_start_index,
                sampling_locations, attention_weights, im2col_step, rows_bf16=False, rows_f32=False):
        ctx.im2col_step = im2col_step
        ctx.rows_bf16 = bool(rows_bf16) and value.dtype == torch.float32
        ctx.rows_f32 = bool(rows_f32) and value.dtype == torch.bfloat16
        # host copy of the level shapes, when our transformer attached one (spares a device sync)
        ctx.host_shapes = getattr(value_spatial_shapes, "_snipper_host", None)
        if value.dtype == torch.bfloat16:   # coordinates and weights stay fp32 beside bf16 values
            sampling_locations = sampling_locations.float()
            attention_weights = attention_weights.float()
        out = MSDA.ms_deform_attn_forward(value, value_spatial_shapes, value_level_start_index, sampling_locations,
                                          attention_weights, im2col_step, out_bf16=ctx.rows_bf16,
                                          host_shapes=ctx.host_shapes, out_f32=ctx.rows_f32)
        ctx.save_for_backward(value, value_spatial_shapes, value_level_start_index,
                              sampling_locations, attention_weights)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        value, shapes, lsi, loc, attn = ctx.saved_tensors
        grad_output = grad_output.contiguous()
        if grad_output.dtype != value.dtype and not (ctx.rows_bf16 and grad_output.dtype == torch.bfloat16) and \
                not (ctx.rows_f32 and grad_output.dtype == torch.float32):
            grad_output = grad_output.to(value.dtype)
        grad_value, grad_loc, grad_attn = MSDA.ms_deform_attn_backward(
            value, shapes, lsi, loc, attn, grad_output, ctx.im2col_step, host_shapes=ctx.host_shapes)
        return grad_value, None, None, grad_loc, grad_attn, None, None, None   # reference :42


def ms_deform_attn_core_pytorch(value, value_spatial_shapes, sampling_locations, attention_weights):
    """grid_sample formulation of the core op; any device, any float dtype, differentiable.

    value [N,S,M,D], sampling_locations [N,Lq,M,L,P,2] in [0,1], attention_weights [N,Lq,M,L,P]
    -> [N,Lq,M*D].
    """
    N, S, M, D = value.shape
    Lq, L, P = sampling_locations.shape[1], sampling_locations.shape[3], sampling_locations.shape[4]
    sizes = [int(h) * int(w) for h, w in value_spatial_shapes]
    grids = (sampling_locations * 2 - 1).permute(3, 0, 2, 1, 4, 5)      # [L,N,M,Lq,P,2]
    maps = value.permute(0, 2, 3, 1).split(sizes, dim=-1)               # L x [N,M,D,H*W]
    sampled = []
    for lvl, (hw, fmap) in enumerate(zip(value_spatial_shapes, maps)):
        H, W = int(hw[0]), int(hw[1])
        sampled.append(F.grid_sample(fmap.reshape(N * M, D, H, W), grids[lvl].reshape(N * M, Lq, P, 2),
                                     mode="bilinear", padding_mode="zeros", align_corners=False))
    sampled = torch.cat(sampled, dim=-1)                                 # [N*M, D, Lq, L*P]
    w = attention_weights.permute(0, 2, 1, 3, 4).reshape(N * M, 1, Lq, L * P)
    out = (sampled * w).sum(-1)                                          # [N*M, D, Lq]
    return out.view(N, M * D, Lq).transpose(1, 2).contiguous()
