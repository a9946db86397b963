"""ResNet-50 feature extractor with frozen BatchNorm, and the (t, y, x) sine position encoding.

API mirror of /root/reference/models/backbone.py (``FrozenBatchNorm2d``, ``BackboneBase``
semantics, ``Backbone``, ``Joiner``, ``build_backbone``) and models/position_encoding.py
(``PositionEmbeddingSine``, ``build_position_encoding``).

The reference does not contain the network itself: it instantiates ``torchvision.models.resnet50``
(backbone.py:105-107, un-vendored, version unpinned) and taps layer2/3/4 through
``IntermediateLayerGetter`` (:78-85).  torchvision is absent here, so the ResNet-50 (v1.5:
stride on the 3x3 convolution, as torchvision's) is restated in ``ResNet50Body`` with
torchvision's parameter names, so a reference checkpoint's ``backbone.0.body.*`` keys load
unchanged.  Parity status of this file: UNPINNED against torchvision (no reference test or
golden vector exists for it); it is checked layer by layer against ``F.conv2d`` compositions.

Round-1 status, for bf16 NHWC activations: every 1x1 convolution of the network (36 of its 53 convolutions, ~55 %
of its FLOPs) runs on this repository's bf16 MFMA kernel -- forward and data gradient with BN, residual and ReLU
fused in the epilogue (csrc/gemm_bf16.cuh), weight gradient through hipBLASLt; the 16 3x3 convolutions run their
forward and their stride-1 data gradient on the implicit-GEMM variant of the same kernel (stride-2 data gradient
and weight gradients through MIOpen); the 7x7 stem (3 input
channels, frozen) and the max-pool still run through PyTorch.
"""
from __future__ import annotations

import math
from typing import Dict, List

import torch
from ._autograd import Function as _Fn
import torch.nn.functional as F
from torch import nn

from .misc import NestedTensor, is_no_padding, no_padding_mask


class FrozenBatchNorm2d(nn.Module):
    """BatchNorm2d with fixed statistics and affine parameters (reference backbone.py:27-64):
    y = x * w * rsqrt(var + eps) + (b - mean * w * rsqrt(var + eps)); all four are buffers."""

    def __init__(self, n, eps=1e-5):
        super().__init__()
        self.register_buffer("weight", torch.ones(n))
        self.register_buffer("bias", torch.zeros(n))
        self.register_buffer("running_mean", torch.zeros(n))
        self.register_buffer("running_var", torch.ones(n))
        self.eps = eps

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        state_dict.pop(prefix + "num_batches_tracked", None)     # reference :43-51
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def scale_bias(self):
        """(scale, shift) of the affine map (at least float32), cached until one of the four buffers is modified or moved
        (they are constants during training; recomputing them cost ~5 tiny launches per convolution call)."""
        bufs = (self.weight, self.bias, self.running_mean, self.running_var)
        key = tuple((b._version, b.data_ptr(), b.dtype) for b in bufs)
        cached = getattr(self, "_affine_cache", None)
        if cached is None or cached[0] != key:
            with torch.no_grad():
                dt = torch.promote_types(self.weight.dtype, torch.float32)     # at least float32
                scale = self.weight.to(dt) * (self.running_var.to(dt) + self.eps).rsqrt()
                shift = self.bias.to(dt) - self.running_mean.to(dt) * scale
            cached = (key, scale, shift)
            self._affine_cache = cached
        return cached[1], cached[2]

    def forward(self, x):
        scale, bias = self.scale_bias()
        return x * scale.view(1, -1, 1, 1).to(x.dtype) + bias.view(1, -1, 1, 1).to(x.dtype)


# Fold the backward of the bottleneck ReLUs into the data-gradient kernels' store phase (tests switch it off to compare)
FOLD_RELU_BACKWARD = True


class _ReluGate(_Fn):
    """Identity whose backward zeroes the gradient where ``x`` (a ReLU output) is 0: the explicit form of the ReLU
    backward that the data-gradient kernels otherwise apply in their store phase (``gate_input`` below).  Only taken when
    a consumer cannot do that itself (shapes outside the kernels' requirements)."""

    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return torch.ops.aten.threshold_backward(g.contiguous(memory_format=torch.channels_last)
                                                 if x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last)
                                                 else g.contiguous(), x, 0)


def conv_frozen_bn(x, conv: nn.Conv2d, bn: FrozenBatchNorm2d, relu: bool, residual=None, gate_input: bool = False,
                   pregated: bool = False):
    """conv -> frozen BN (+ residual) (-> ReLU) with the BN folded into the convolution.

    The backward of a ReLU between two of these nodes is folded into the store phase of the data-gradient kernel that
    produces the gradient of the ReLU's output (its consumer knows the activation: it is its saved input):
    ``gate_input`` -- ``x`` came out of a ReLU whose own node does NOT apply the ReLU backward; this node gates the
    gradient it returns for ``x``.  ``pregated`` -- the gradient arriving for this node's output already is gated by its
    consumer (which was given ``gate_input``); this node skips its own ReLU backward.  The two flags are always set as a
    pair by the code that wires producer and consumer (Bottleneck.forward / ResNet50Body._run_layer).

    The BN is an affine map with constant coefficients (reference backbone.py:54-64), so
    ``bn(conv(x, w)) == conv(x, w * scale[:, None, None, None]) + shift``: one convolution with a bias
    instead of a convolution and two more passes over the activation.  The fold is recomputed from
    the live weight every call (a few MB), so gradients reach ``conv.weight`` exactly as before.
    """
    scale, shift = bn.scale_bias()
    assert relu or not pregated
    if _hip_pointwise_ok(x, conv, None):
        return _hip_pointwise(x, conv, scale.float(), shift.float(), relu, residual, gate_input, pregated)
    if residual is None and _hip_conv3x3_ok(x, conv):
        return _Conv3x3BN.apply(x, conv.weight, scale.float(), shift.float(), conv.stride[0], relu, gate_input, pregated)
    if gate_input and x.requires_grad and torch.is_grad_enabled():
        x = _ReluGate.apply(x)
    w = conv.weight * scale.view(-1, 1, 1, 1).to(conv.weight.dtype)
    y = F.conv2d(x, w, shift.to(w.dtype), conv.stride, conv.padding, conv.dilation, conv.groups)
    if residual is not None:
        y = y + residual
    if not relu:
        return y
    if pregated:                                  # the consumer gates the gradient: a ReLU without a backward of its own
        return _PregatedRelu.apply(y)
    return F.relu(y, inplace=True)


class _PregatedRelu(_Fn):
    """relu(y) whose gradient is passed through unchanged: the consumer of the result was told (``gate_input``) to zero
    the gradient where the result is 0, which is exactly this ReLU's backward."""

    @staticmethod
    def forward(ctx, y):
        return torch.relu(y)

    @staticmethod
    def backward(ctx, g):
        return g


def _hip_pointwise_ok(x, conv, w) -> bool:
    """1x1 convolution (stride 1, or stride s handled by sub-sampling first) of a bf16 NHWC activation: in NHWC that
    is exactly the dense kernel Y[M, Cout] = X[M, Cin] . W[Cout, Cin]^T (csrc/gemm_bf16.cuh)."""
    return (x.is_cuda and x.dtype == torch.bfloat16 and conv.kernel_size == (1, 1) and conv.padding == (0, 0) and
            conv.stride[0] == conv.stride[1] and conv.groups == 1 and conv.in_channels % 64 == 0 and
            conv.out_channels % 64 == 0 and x.is_contiguous(memory_format=torch.channels_last))


class _PointwiseConvBN(_Fn):
    """relu?(X . (W * scale)^T + shift (+ residual)) on [M, C] views of NHWC tensors, forward and backward.

    Forward and the data gradient run on this repository's MFMA kernel (one launch each, epilogue fused); the
    weight gradient (a [Cout, M] x [M, Cin] product, M = batch * H * W) runs on the split-reduction kernel
    (csrc/wgrad_bf16.cuh).  Besides the fused
    passes this avoids MIOpen's host-side cost per convolution call (~170 us measured: solver look-up), which was
    the largest single item of the step's CPU time."""

    @staticmethod
    def forward(ctx, x2, weight, scale, shift, res2, relu, fork=False, gate_input=False, pregated=False):
        """``fork``: also return ``x2`` itself (as a second output) for a skip connection; the gradient that comes
        back over that output is then added inside the data-gradient kernel instead of by a separate add.
        ``gate_input`` / ``pregated``: see conv_frozen_bn (with ``fork`` the gate covers the sum of both gradients of
        ``x2``, i.e. everything that flows back into the ReLU that produced it)."""
        from .dense import linear_bf16, linear_patch_bf16, linear_patch_supported
        cout, cin = weight.shape[0], weight.shape[1]
        from . import shadow
        w_eff = shadow.lookup(weight, scale)            # bf16(weight * scale), refreshed once per step for the model
        if w_eff is not None:
            w_eff = w_eff.view(cout, cin)
        else:
            w_eff = (weight.reshape(cout, cin).float() * scale[:, None]).to(torch.bfloat16)
        # the one-tap patch kernel where it measured faster (shadow.patch_forward_pays; always for the data gradient)
        pk = shadow.lookup_lpacked(weight, scale) if shadow.lookup(weight, scale) is not None else None
        ctx.packed_t = pk[1] if pk is not None else None
        if (pk is not None and pk[0] is not None and res2 is None and x2.is_contiguous() and
                linear_patch_supported(x2.shape[0], cout, cin)):
            y = linear_patch_bf16(x2, pk[0], cout, shift, None, relu, None, 64)
        else:
            y = linear_bf16(x2, w_eff, shift, res2, relu)
        ctx.relu, ctx.has_res = relu and not pregated, res2 is not None
        ctx.wshape, ctx.wdtype, ctx.wstride = weight.shape, weight.dtype, weight.stride()
        ctx.w_ref = weight
        ctx.save_for_backward(x2, w_eff, scale, y if ctx.relu else None)
        ctx.fork, ctx.gate_input = fork, gate_input
        if fork:
            return y, x2.view_as(x2)
        return y

    @staticmethod
    def backward(ctx, gy, gskip=None):
        from .dense import linear_bf16
        x2, w_eff, scale, y = ctx.saved_tensors
        g = gy.contiguous()
        if ctx.relu:
            g = torch.ops.aten.threshold_backward(g, y, 0)
        from .dense import _dgrad
        dw = None
        if ctx.needs_input_grad[1]:
            from .dense import wgrad_bf16
            from .dense import _grad_out_conv
            gv = _grad_out_conv(ctx.w_ref) if ctx.wdtype == torch.float32 else None      # the flat gradient buffer's slice, if any
            out2 = None if gv is None else gv.as_strided((ctx.wshape[0], ctx.wshape[1]), (ctx.wshape[1], 1))
            dw, _ = wgrad_bf16(g, x2, want_bias=False, scale=scale, out=out2)     # BN scale folded into the reduction kernel
            # same memory, the parameter's own strides (NHWC weights: DDP aliases its bucket only when they match)
            dw = gv if gv is not None else dw.to(ctx.wdtype).as_strided(ctx.wshape, ctx.wstride)
        dx = None
        if ctx.needs_input_grad[0]:
            skip = gskip if ctx.fork else None
            gate = x2 if ctx.gate_input else None
            from .dense import linear_patch_bf16, linear_patch_supported
            if (ctx.packed_t is not None and linear_patch_supported(g.shape[0], x2.shape[1], g.shape[1]) and
                    (skip is None or (skip.dtype == torch.bfloat16 and skip.is_contiguous())) and
                    (gate is None or gate.is_contiguous())):
                dx = linear_patch_bf16(g, ctx.packed_t, x2.shape[1], None, skip, False, gate, 64, kind="linear_nn")
            else:
                dx = _dgrad(g, w_eff, skip, gate)
        return dx, dw, None, None, (g if ctx.has_res else None), None, None, None, None


def _hip_conv3x3_ok(x, conv) -> bool:
    """3x3, padding 1, stride 1 or 2 (every bottleneck conv2) of a bf16 NHWC activation: implicit GEMM on the MFMA
    tiles (csrc/gemm_bf16.cuh, conv3x3_bf16_kernel)."""
    return (x.is_cuda and x.dtype == torch.bfloat16 and conv.kernel_size == (3, 3) and conv.padding == (1, 1) and
            conv.stride in ((1, 1), (2, 2)) and conv.dilation == (1, 1) and conv.groups == 1 and
            conv.in_channels % 64 == 0 and conv.out_channels % 4 == 0 and
            x.is_contiguous(memory_format=torch.channels_last))


class _Conv3x3BN(_Fn):
    """relu?(conv3x3(x, W * scale) + shift): forward on this repository's implicit-GEMM kernel (one launch, BN and
    ReLU in the epilogue); the stride-1 data gradient on the same kernel (taps reversed, channel roles swapped), the
    stride-2 data gradient as four parity-class launches of it, the weight gradient on the split-reduction kernel
    (csrc/wgrad_bf16.cuh, conv mode).  Nothing of a Snipper recipe's ResNet-50 goes through MIOpen."""

    @staticmethod
    def forward(ctx, x, weight, scale, shift, stride, relu, gate_input=False, pregated=False):
        from .dense import conv3x3_bf16, conv3x3_patch_bf16, conv3x3_patch_supported
        from . import shadow
        w_eff = shadow.lookup(weight, scale)
        if w_eff is None or not w_eff.is_contiguous(memory_format=torch.channels_last):
            w_eff = (weight.float() * scale.view(-1, 1, 1, 1)).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        # stride 1 with a packed shadow: the patch-resident kernel (csrc/conv3x3_patch_bf16.cuh), forward and data gradient
        pk = shadow.lookup_packed(weight, scale) if (stride == 1 and shadow.lookup(weight, scale) is w_eff) else None
        ctx.packed_t = None
        if pk is not None and conv3x3_patch_supported(x.shape[0], x.shape[2], x.shape[3], x.shape[1], w_eff.shape[0]):
            y = conv3x3_patch_bf16(x, pk[0], w_eff.shape[0], shift, relu)
            ctx.packed_t = pk[1]
        else:
            y = conv3x3_bf16(x, w_eff, shift, stride, relu)
        ctx.w_t = shadow.lookup_t(weight) if shadow.lookup(weight, scale) is w_eff else None      # (this step's, if kept)
        ctx.stride, ctx.relu, ctx.wdtype, ctx.gate_input = stride, relu and not pregated, weight.dtype, gate_input
        ctx.w_ref = weight
        ctx.save_for_backward(x, w_eff, scale, y if ctx.relu else None)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w_eff, scale, y = ctx.saved_tensors
        g = gy.contiguous(memory_format=torch.channels_last)
        if ctx.relu:
            g = torch.ops.aten.threshold_backward(g, y, 0)
        from .dense import conv3x3_bf16, conv3x3_dgrad_s2_bf16, wgrad_conv3x3_bf16
        cout, cin = w_eff.shape[0], w_eff.shape[1]
        own_dgrad = ctx.needs_input_grad[0] and cout % 64 == 0
        own_wgrad = ctx.needs_input_grad[1] and cin % 128 == 0 and cout % 8 == 0
        need = [ctx.needs_input_grad[0] and not own_dgrad, ctx.needs_input_grad[1] and not own_wgrad, False]
        dx, dw = None, None
        if need[0] or need[1]:                     # shapes outside the kernels' requirements (no Snipper recipe has any)
            dx, dw, _ = torch.ops.aten.convolution_backward(
                g, x, w_eff, None, [ctx.stride] * 2, [1, 1], [1, 1], False, [0, 0], 1, need)
            if dw is not None:
                dw = (dw.float() * scale.view(-1, 1, 1, 1)).to(ctx.wdtype)
            if dx is not None and ctx.gate_input:
                dx = torch.ops.aten.threshold_backward(dx, x, 0)
        if own_wgrad:
            # split-reduction kernel, BN scale folded into its second pass; float32, channels_last like the parameter
            from .dense import _grad_out_conv
            gv = _grad_out_conv(ctx.w_ref) if ctx.wdtype == torch.float32 else None
            dw = wgrad_conv3x3_bf16(g, x, ctx.stride, scale, out=gv).to(ctx.wdtype)
        gate = x if ctx.gate_input else None      # x came out of a ReLU that left its backward to this node
        if own_dgrad and ctx.stride == 1 and ctx.packed_t is not None:
            from .dense import conv3x3_patch_bf16
            dx = conv3x3_patch_bf16(g, ctx.packed_t, cin, None, False, gate, dgrad=True)
        elif own_dgrad and ctx.stride == 1:
            # stride 1: the data gradient is the same convolution with the taps reversed and the channel roles swapped
            # (the kernel reverses the taps itself; only the channel axes are swapped here)
            w_t = ctx.w_t if ctx.w_t is not None else w_eff.transpose(0, 1).contiguous(memory_format=torch.channels_last)
            dx = conv3x3_bf16(g, w_t, None, 1, False, gate, flip_taps=True)
        elif own_dgrad:
            # stride 2: four parity classes of the input pixel, each with its 1 / 2 / 2 / 4 taps (csrc/gemm_bf16.cuh)
            dx = conv3x3_dgrad_s2_bf16(g, ctx.w_t if ctx.w_t is not None else w_eff.transpose(0, 1), x.shape[-2:], gate)
        return dx, dw, None, None, None, None, None, None


class _Subsample(_Fn):
    """x[:, :, ::sh, ::sw] as a dense channels-last tensor (the input of a strided 1x1 convolution).  Autograd's own
    backward of the two slices is fill + copy twice into a contiguous (NCHW-strided) buffer, which then meets the other
    gradients of x in the generic strided add kernel (measured 100-190 us per downsample block); here it is one fill
    and one strided copy into a channels-last buffer, and the add that follows is the vectorised one."""

    @staticmethod
    def forward(ctx, x, sh, sw):
        ctx.shape, ctx.step = x.shape, (sh, sw)
        return x[:, :, ::sh, ::sw].contiguous(memory_format=torch.channels_last)

    @staticmethod
    def backward(ctx, g):
        sh, sw = ctx.step
        dx = torch.empty(ctx.shape, dtype=g.dtype, device=g.device, memory_format=torch.channels_last).zero_()
        dx[:, :, ::sh, ::sw].copy_(g)
        return dx, None, None


def _hip_pointwise(x, conv, scale, shift, relu, residual, gate_input=False, pregated=False):
    """conv1x1 + folded BN (+ residual) (+ ReLU) as ONE launch of the MFMA kernel."""
    if conv.stride != (1, 1):
        if gate_input and x.requires_grad and torch.is_grad_enabled():
            x, gate_input = _ReluGate.apply(x), False          # (the gate has x's shape, not the sub-sampled one)
        x = _Subsample.apply(x, conv.stride[0], conv.stride[1])
    b, c, h, wd = x.shape
    rows = x.permute(0, 2, 3, 1).reshape(b * h * wd, c)                      # a view: NHWC is row-major [M, Cin]
    res = None
    if residual is not None:
        res = residual.permute(0, 2, 3, 1).reshape(b * h * wd, -1)
        if res.dtype != torch.bfloat16:
            res = res.to(torch.bfloat16)
    y = _PointwiseConvBN.apply(rows, conv.weight, scale, shift.float(), res, relu, False, gate_input, pregated)
    return y.view(b, h, wd, -1).permute(0, 3, 1, 2)                          # logical NCHW, channels_last memory


def _hip_pointwise_fork(x, conv, bn, gate_input=False, pregated=False):
    """(relu(bn(conv1x1(x))), x) with both results coming out of ONE autograd node, so that the gradient of the skip
    connection is added inside that node's data-gradient kernel.  None when the MFMA path does not apply."""
    if not (conv.stride == (1, 1) and _hip_pointwise_ok(x, conv, None)):
        return None
    scale, shift = bn.scale_bias()
    b, c, h, wd = x.shape
    rows = x.permute(0, 2, 3, 1).reshape(b * h * wd, c)
    y, skip = _PointwiseConvBN.apply(rows, conv.weight, scale.float(), shift.float(), None, True, True, gate_input, pregated)
    return y.view(b, h, wd, -1).permute(0, 3, 1, 2), skip.view(b, h, wd, c).permute(0, 3, 1, 2)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, width, stride=1, downsample=False, dilation=1):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, width, 1, bias=False)
        self.bn1 = FrozenBatchNorm2d(width)
        self.conv2 = nn.Conv2d(width, width, 3, stride=stride, padding=dilation, dilation=dilation, bias=False)
        self.bn2 = FrozenBatchNorm2d(width)
        self.conv3 = nn.Conv2d(width, width * 4, 1, bias=False)
        self.bn3 = FrozenBatchNorm2d(width * 4)
        self.downsample = None
        if downsample:
            self.downsample = nn.Sequential(nn.Conv2d(inplanes, width * 4, 1, stride=stride, bias=False),
                                            FrozenBatchNorm2d(width * 4))

    def forward(self, x, in_gated: bool = False, out_pregated: bool = False):
        """``in_gated``: ``x`` is the previous block's output and that block left the backward of its final ReLU to this
        one; ``out_pregated``: the next block does the same for this block's final ReLU (ResNet50Body._run_layer sets
        both sides).  Inside the block the two inner ReLUs are always handled that way: conv2 gates the gradient of
        conv1's output, conv3 that of conv2's."""
        fold = torch.is_grad_enabled() and FOLD_RELU_BACKWARD
        fork = None
        if self.downsample is None and x.requires_grad and torch.is_grad_enabled():
            # identity skip: fold its gradient add away (and, if in_gated, the previous block's ReLU backward with it)
            fork = _hip_pointwise_fork(x, self.conv1, self.bn1, gate_input=in_gated, pregated=fold)
        if fork is not None:
            y, skip = fork
        else:
            if in_gated and x.requires_grad and torch.is_grad_enabled():
                x = _ReluGate.apply(x)                  # several consumers of x below: gate once, explicitly
            y = conv_frozen_bn(x, self.conv1, self.bn1, relu=True, pregated=fold)
            skip = x if self.downsample is None else conv_frozen_bn(x, self.downsample[0], self.downsample[1], relu=False)
        y = conv_frozen_bn(y, self.conv2, self.bn2, relu=True, gate_input=fold, pregated=fold)
        return conv_frozen_bn(y, self.conv3, self.bn3, relu=True, residual=skip, gate_input=fold, pregated=out_pregated)


class ResNet50Body(nn.Module):
    """conv1..layer4 of ResNet-50; forward returns {"0": layer2, "1": layer3, "2": layer4}
    (or {"0": layer4}) like IntermediateLayerGetter with the reference's return_layers."""

    def __init__(self, return_interm_layers=True, dilation=False):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = FrozenBatchNorm2d(64)
        inplanes = 64
        for i, (width, blocks, stride) in enumerate([(64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)]):
            # reference backbone.py:105-107, replace_stride_with_dilation=[False, False, dilation] (the DC5 variant):
            # layer4 keeps layer3's resolution -- its stride becomes 1, its first block keeps dilation 1 and the
            # following blocks' 3x3 convolutions take dilation 2 (torchvision's _make_layer rule).  The dilated 3x3
            # convolutions run through the library convolution (no Snipper recipe uses DC5); everything else is unchanged.
            dil = 1
            if dilation and i == 3:
                dil, stride = stride, 1
            layers = [Bottleneck(inplanes, width, stride, downsample=True)]
            inplanes = width * 4
            layers += [Bottleneck(inplanes, width, dilation=dil) for _ in range(blocks - 1)]
            setattr(self, f"layer{i + 1}", nn.Sequential(*layers))
        self.return_interm_layers = return_interm_layers
        self._stem_frozen = None      # decided at the first forward; reset it after changing layer1's requires_grad
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    def _stem(self, x):
        """relu(bn1(conv1(x))) -> MaxPool2d(3, 2, 1).  Frozen stem on bf16 NHWC: the convolution runs with the BN scale
        folded in and no bias, and shift + ReLU + pooling are one kernel (csrc/gn_tokens.cuh stem_pool_kernel)."""
        conv, bn = self.conv1, self.bn1
        if (x.is_cuda and torch.is_autocast_enabled('cuda') and torch.get_autocast_dtype('cuda') == torch.bfloat16 and
                not conv.weight.requires_grad and not (torch.is_grad_enabled() and x.requires_grad) and
                conv.out_channels % 8 == 0 and conv.bias is None):
            from . import _lib
            scale, shift = bn.scale_bias()
            with torch.no_grad():
                key = (conv.weight._version, id(scale), conv.weight.data_ptr())
                cached = getattr(self, "_stem_w", None)
                if cached is None or cached[0] != key:           # frozen: folded and cast once
                    w = (conv.weight * scale.view(-1, 1, 1, 1).to(conv.weight.dtype)).to(torch.bfloat16)
                    wp = None
                    if tuple(conv.weight.shape) == (64, 3, 7, 7) and conv.stride == (2, 2) and conv.padding == (3, 3):
                        # the stem kernel's K order: k = ky*32 + kx*4 + c, zero in the padding slots (csrc/gemm_bf16.cuh)
                        wp = torch.zeros(64, 8, 8, 4, dtype=torch.bfloat16, device=w.device)
                        wp[:, :7, :7, :3] = w.permute(0, 2, 3, 1)
                        wp = wp.view(64, 256)
                    cached = (key, w.contiguous(memory_format=torch.channels_last), wp)
                    object.__setattr__(self, "_stem_w", cached)
                y = None
                if cached[2] is not None and x.shape[1] == 3:
                    n, _, h, wd = x.shape
                    x4 = None
                    if x.dtype == torch.float32 and x.is_contiguous() and wd % 4 == 0 and x.data_ptr() % 16 == 0:
                        # planar float32 images (what the data loader delivers) -> bf16 [n, h, w, 4] in one pass
                        x4 = torch.empty((n, h, wd, 4), dtype=torch.bfloat16, device=x.device)
                        with _lib.device_guard(x.device):
                            rc = _lib.load().snipper_stem_pack_bf16(_lib.raw_stream(x.device), x.data_ptr(), n, h, wd, x4.data_ptr())
                        _lib.check(rc, "snipper_stem_pack_bf16")
                    if x4 is None:
                        x4 = torch.zeros((n, h, wd, 4), dtype=torch.bfloat16, device=x.device)     # channels padded 3 -> 4
                        x4[..., :3] = x.permute(0, 2, 3, 1)
                    y = torch.empty((n, 64, (h - 1) // 2 + 1, (wd - 1) // 2 + 1), dtype=torch.bfloat16, device=x.device,
                                    memory_format=torch.channels_last)
                    from . import dense as _dense
                    ho_, wo_ = (h - 1) // 2 + 1, (wd - 1) // 2 + 1
                    with _dense._timed("stem7x7", (n, h, wd, 3, 64, 2), 2 * n * ho_ * wo_ * 64 * 147,
                                       2 * (n * h * wd * 4 + 64 * 256 + n * ho_ * wo_ * 64), x.device), _lib.device_guard(x.device):
                        rc = _lib.load().snipper_stem7x7_bf16(_lib.raw_stream(x.device), x4.data_ptr(), cached[2].data_ptr(),
                                                              y.data_ptr(), n, h, wd)
                    _lib.check(rc, "snipper_stem7x7_bf16")
                if y is None:
                    y = F.conv2d(x.to(torch.bfloat16).contiguous(memory_format=torch.channels_last), cached[1], None, conv.stride, conv.padding, conv.dilation,
                                 conv.groups)
                if y.dtype == torch.bfloat16 and y.is_contiguous(memory_format=torch.channels_last):
                    n, c, h, wd = y.shape
                    out = torch.empty((n, c, (h - 1) // 2 + 1, (wd - 1) // 2 + 1), dtype=torch.bfloat16, device=y.device,
                                      memory_format=torch.channels_last)
                    sh = shift.float().contiguous()
                    with _lib.device_guard(y.device):
                        rc = _lib.load().snipper_stem_pool_bf16(_lib.raw_stream(y.device), y.data_ptr(), sh.data_ptr(),
                                                                n, h, wd, c, out.data_ptr())
                    _lib.check(rc, "snipper_stem_pool_bf16")
                    return out
        x = conv_frozen_bn(x.contiguous(memory_format=torch.channels_last), conv, bn, relu=True)
        return F.max_pool2d(x, 3, stride=2, padding=1)

    @staticmethod
    def _run_layer(layer, x):
        """The layer's blocks in sequence; between two blocks of one layer (identity skip on the consumer's side) the
        backward of the producer's final ReLU is left to the consumer's data-gradient kernel.  A layer's LAST output has
        other consumers (next layer, input projections) and keeps its own ReLU backward."""
        blocks = list(layer)
        fold = torch.is_grad_enabled() and FOLD_RELU_BACKWARD
        gated = False
        for i, blk in enumerate(blocks):
            nxt = blocks[i + 1] if i + 1 < len(blocks) else None
            pre = bool(fold and nxt is not None and isinstance(nxt, Bottleneck) and nxt.downsample is None)
            x = blk(x, gated, pre) if isinstance(blk, Bottleneck) else blk(x)
            gated = pre
        return x

    def forward(self, x) -> Dict[str, torch.Tensor]:
        # NHWC end to end: no layout shuffles around MIOpen.  (Planar float32 3-channel images are left as they are: the frozen
        # stem packs them into its own [n, h, w, 4] bf16 layout in one pass, snipper_stem_pack_bf16.)
        if not (x.is_cuda and x.dim() == 4 and x.shape[1] == 3 and x.dtype == torch.float32 and x.is_contiguous()):
            x = x.contiguous(memory_format=torch.channels_last)
        x = self._stem(x)
        if self._stem_frozen is None:                             # walked once (Backbone.__init__ freezes before use)
            self._stem_frozen = not any(p.requires_grad for p in self.layer1.parameters())
        if self._stem_frozen:
            with torch.no_grad():                                 # conv1 + layer1 are frozen: nothing to save
                c2 = self.layer1(x)
        else:
            c2 = self._run_layer(self.layer1, x)
        c3 = self._run_layer(self.layer2, c2)
        c4 = self._run_layer(self.layer3, c3)
        c5 = self._run_layer(self.layer4, c4)
        if self.return_interm_layers:
            return {"0": c3, "1": c4, "2": c5}
        return {"0": c5}


class _Segment(nn.Module):
    """One capturable piece of ``ResNet50Body.forward``: ``first`` = stem + layer1 (frozen, no autograd) + layer2, else one
    residual stage.  Holds the body's own sub-modules (shared parameters)."""

    def __init__(self, body: "ResNet50Body", which: str):
        super().__init__()
        self.which = which
        self.body_ref = [body]                    # (a list: not registered as a child again)
        self.layer = getattr(body, {"first": "layer2", "layer3": "layer3", "layer4": "layer4"}[which])

    def forward(self, x):
        body = self.body_ref[0]
        with torch.autocast("cuda", dtype=torch.bfloat16, cache_enabled=False):
            if self.which == "first":
                x = x.contiguous(memory_format=torch.channels_last)
                x = body._stem(x)
                with torch.no_grad():
                    x = body.layer1(x)
            return body._run_layer(self.layer, x)


def graphed_segments(body: "ResNet50Body", sample_images: torch.Tensor):
    """``body`` as three hipGraph-captured callables (forward AND backward each; ``torch.cuda.make_graphed_callables``):
    stem + layer1 + layer2 | layer3 | layer4.  Every shape in here is static (the images' size) and every kernel is this
    package's own (stream-ordered, no allocation outside torch's caching allocator, no host synchronisation), so the ~350
    launches of the region replay from three graph launches each way.  The cuts are the gradient all-reduce's stage
    boundaries (bench.grad_sync_stages): a stage's trigger parameters receive their gradients when its segment's backward
    has been replayed, i.e. in the same order as eagerly.  Returns ``f(images) -> {"0": c3, "1": c4, "2": c5}``.
    EXPERIMENT (round 4, tools/graph_backbone.py, tests/test_backbone.py): features bit-identical, host issue time of the
    region 4.1 -> 0.7 ms, its GPU time 5.2 -> 5.8 ms (replay costs ~1.7 us per node on the GPU side) -- a loss while the
    step is GPU-bound (25 ms of kernels against 19.6 ms of host issue), so nothing in the product path uses it; wired into
    bench.py's full step it also crashed the process (segmentation fault inside the captured backward), not root-caused.
    Requires bf16 autocast, training mode, frozen stem + layer1 (the reference's freeze rule) and all three taps.
    OPT-IN ONLY (``SNIPPER_EXPERIMENTAL_GRAPHS=1``): a known process-killing path must not be one import away."""
    import os
    if os.environ.get("SNIPPER_EXPERIMENTAL_GRAPHS") != "1":
        raise RuntimeError("graphed_segments is an experiment (inside the full training step its captured backward crashed "
                           "the process, not root-caused): set SNIPPER_EXPERIMENTAL_GRAPHS=1 to use it")
    assert body.return_interm_layers and sample_images.is_cuda
    segs = [_Segment(body, w) for w in ("first", "layer3", "layer4")]
    with torch.no_grad():
        c3 = segs[0](sample_images)
        c4 = segs[1](c3)
    samples = ((sample_images,), (c3.detach().clone().requires_grad_(True),), (c4.detach().clone().requires_grad_(True),))
    g = torch.cuda.make_graphed_callables(tuple(segs), samples)

    def fwd(images):
        c3 = g[0](images)
        c4 = g[1](c3)
        c5 = g[2](c4)
        return {"0": c3, "1": c4, "2": c5}
    return fwd


def backbone_parameter_is_trainable(name: str, train_backbone: bool) -> bool:
    """The reference's freeze rule (backbone.py:71-73): only layer2 / layer3 / layer4 parameters train, and only when the
    backbone has a learning rate; conv1 + layer1 are always frozen."""
    return bool(train_backbone) and any(k in name for k in ("layer2", "layer3", "layer4"))


class _LayerGetter(nn.ModuleDict):
    """What the reference takes from torchvision as ``IntermediateLayerGetter`` (backbone.py:19,85): the children of a
    network in registration order up to the last returned one; forward collects {new name: output} along the way."""

    def __init__(self, model: nn.Module, return_layers: Dict[str, str]):
        todo = dict(return_layers)
        layers = {}
        for name, module in model.named_children():
            layers[name] = module
            todo.pop(name, None)
            if not todo:
                break
        if todo:
            raise ValueError("return_layers are not present in model")
        super().__init__(layers)
        self.return_layers = dict(return_layers)

    def forward(self, x):
        out = {}
        for name, module in self.items():
            x = module(x)
            if name in self.return_layers:
                out[self.return_layers[name]] = x
        return out


class BackboneBase(nn.Module):
    """Reference ``BackboneBase`` (backbone.py:67-99) for ANY network with ResNet's child names: freeze rule, the
    layer2/3/4 (or layer4) taps, and the nearest-resized padding mask per tapped level."""

    def __init__(self, backbone: nn.Module, train_backbone: bool, return_interm_layers: bool, body: nn.Module = None):
        super().__init__()
        for pname, p in backbone.named_parameters():
            if not backbone_parameter_is_trainable(pname, train_backbone):
                p.requires_grad_(False)
        if return_interm_layers:
            return_layers = {"layer2": "0", "layer3": "1", "layer4": "2"}
            self.strides, self.num_channels = [8, 16, 32], [512, 1024, 2048]
        else:
            return_layers = {"layer4": "0"}
            self.strides, self.num_channels = [32], [2048]
        self.body = body if body is not None else _LayerGetter(backbone, return_layers)

    def forward(self, tensor_list: NestedTensor):
        feats = self.body(tensor_list.tensors)
        out: Dict[str, NestedTensor] = {}
        for name, x in feats.items():
            m = tensor_list.mask
            assert m is not None
            if is_no_padding(m):                       # nearest-resizing an all-False mask: all False, known here
                mask = no_padding_mask(m.shape[0], x.shape[-2], x.shape[-1], m.device)
            else:
                mask = F.interpolate(m[None].float(), size=x.shape[-2:]).to(torch.bool)[0]    # nearest (:93)
            out[name] = NestedTensor(x, mask)
        return out


class Backbone(BackboneBase):
    """ResNet backbone with frozen BatchNorm (reference ``Backbone``, :102-111).  The network is this package's own
    ResNet-50 (``ResNet50Body`` already returns the tapped levels, so it IS the body)."""

    def __init__(self, name: str = "resnet50", train_backbone: bool = True,
                 return_interm_layers: bool = True, dilation: bool = False):
        if name != "resnet50":
            raise ValueError("only resnet50 is restated here (the Snipper recipes use nothing else)")
        body = ResNet50Body(return_interm_layers, dilation)
        super().__init__(body, train_backbone, return_interm_layers, body=body)
        if dilation:
            self.strides[-1] = self.strides[-1] // 2


from .misc import BoundedCache  # noqa: E402
_POS_CACHE = BoundedCache(16)


class PositionEmbeddingSine(nn.Module):
    """Normalised sine embedding over (t, y, x): [b*t,h,w] mask -> [b, t, 3*num_pos_feats, h, w]
    (reference position_encoding.py:20-63)."""

    def __init__(self, num_pos_feats=64, num_frames=8, temperature=10000, normalize=False, scale=None):
        super().__init__()
        if scale is not None and normalize is False:
            raise ValueError("normalize should be True if scale is passed")
        self.num_pos_feats, self.frames = num_pos_feats, num_frames
        self.temperature, self.normalize = temperature, normalize
        self.scale = 2 * math.pi if scale is None else scale

    def forward(self, tensor_list: NestedTensor):
        return self.channel_last(tensor_list.mask).permute(0, 1, 4, 2, 3)   # (z, y, x) blocks -> [b,t,3F,h,w]

    def channel_last(self, mask):
        """[b*t, h, w] padding mask -> [b, t, h, w, 3F]: the embedding as it is computed, before the reference's
        permute to channel-first (position_encoding.py:62); the token-row path uses it as is."""
        n, h, w = mask.shape
        if is_no_padding(mask) and not torch.is_inference_mode_enabled():   # a constant of the shapes: built once
            key = (n, h, w, self.frames, self.num_pos_feats, self.temperature, self.normalize, self.scale, str(mask.device))
            got = _POS_CACHE.get(key)
            if got is None:
                got = _POS_CACHE[key] = self._channel_last(mask)
            return got
        return self._channel_last(mask)

    def _channel_last(self, mask):
        n, h, w = mask.shape
        live = ~mask.reshape(n // self.frames, self.frames, h, w)
        axes = []
        for dim in (1, 2, 3):                                   # t, y, x running counts of valid cells
            e = live.cumsum(dim, dtype=torch.float32)
            if self.normalize:
                last = e.select(dim, e.shape[dim] - 1).unsqueeze(dim)
                e = e / (last + 1e-6) * self.scale
            axes.append(e)
        k = torch.arange(self.num_pos_feats, dtype=torch.float32, device=mask.device)
        freq = self.temperature ** (2 * torch.div(k, 2, rounding_mode="floor") / self.num_pos_feats)
        parts = []
        for e in axes:
            ang = e[..., None] / freq                           # [b,t,h,w,F]
            parts.append(torch.stack((ang[..., 0::2].sin(), ang[..., 1::2].cos()), dim=5).flatten(4))
        return torch.cat(parts, dim=4)


class Joiner(nn.Sequential):
    def __init__(self, backbone, position_embedding):
        super().__init__(backbone, position_embedding)
        self.strides = backbone.strides
        self.num_channels = backbone.num_channels

    def forward(self, tensor_list: NestedTensor):
        xs = self[0](tensor_list)
        out: List[NestedTensor] = [x for _, x in sorted(xs.items())]
        pos = [self[1](x).to(x.tensors.dtype) for x in out]
        return out, pos


def build_position_encoding(args):
    if args.position_embedding not in ("v2", "sine"):
        raise ValueError(f"not supported {args.position_embedding}")
    return PositionEmbeddingSine(args.hidden_dim // 3, num_frames=args.num_frames, normalize=True)


def build_backbone(args):
    backbone = Backbone(args.backbone, args.lr_backbone > 0,
                        args.masks or (args.num_feature_levels > 1), args.dilation)
    return Joiner(backbone, build_position_encoding(args))
