"""bf16 working copies ("shadows") of the weights the hand-written kernels consume.

The parameters stay float32 (the optimizer's master copy); every kernel that multiplies by a weight wants it in bf16,
and the frozen-BatchNorm convolutions want it pre-multiplied by the BN scale (backbone.conv_frozen_bn).  Doing that
where the weight is used costs 2-3 tiny launches per layer and step -- ~190 for the 52 convolutions and ~50 for the
encoder's Linears.  ``WeightShadows.refresh()`` does it for the whole model with a handful of multi-tensor launches
at the start of a forward pass:

    conv + frozen BN :  shadow = bf16(weight * scale[:, None, None, None])      (same shape and strides as the weight)
    Linear           :  shadow = bf16(weight)
    merged Linears   :  shadow = bf16([W_a; W_b]) in one buffer, bias = [b_a; b_b] (the offset + logit projections)

A shadow is valid while the parameter's version counter is unchanged, so frozen layers are converted once.  An
optimizer's in-place update bumps the counter in the single-tensor and foreach implementations, but NOT in the fused
one (``torch.optim.AdamW(fused=True)`` leaves ``p._version`` alone: measured, PyTorch 2.10) -- a training loop with the
fused optimizer would keep multiplying by the weights of step 0 in every shadowed layer.  ``WeightShadows`` therefore
installs a global optimizer post-step hook (once) that bumps the version counter of every parameter the optimizer
owns; updates made through other tensors (flat_params.py) bump the counters themselves.  Consumers call
``lookup(param)`` and fall back to converting on the spot when there is no valid shadow -- results are identical
either way.
"""
from __future__ import annotations

import weakref
from typing import Dict, List, Optional, Tuple

import torch
from torch import nn

_entries: Dict[int, "_Entry"] = {}
_merged: Dict[Tuple[int, int], "_Merged"] = {}


class _Entry:
    __slots__ = ("ref", "dst", "scale", "scale_full", "version", "dst_t", "tap_pairs", "packed", "packed_t")

    def __init__(self, param, dst, scale=None):
        self.ref, self.dst, self.scale, self.version = weakref.ref(param), dst, scale, -1
        self.scale_full = None     # the BN scale expanded to the weight's shape AND strides (multi-tensor fast path)
        self.dst_t = None          # bf16 W^T [in, out], kept for the Linears whose data gradient wants it
        self.tap_pairs = None      # 3x3 weights: the nine (source, destination) tap views of the batched transpose, built once
        self.packed = None         # stride-1 3x3 weights in MFMA fragment order (csrc/conv3x3_patch_bf16.cuh), forward ...
        self.packed_t = None       # ... and data gradient (channel roles swapped, taps reversed; trainable weights only)


class _Merged:
    __slots__ = ("refs", "w", "b", "versions", "w_t", "b_second_only")

    def __init__(self, params, w, b):
        self.refs, self.w, self.b, self.versions = [weakref.ref(p) for p in params], w, b, None
        self.w_t = None
        self.b_second_only = torch.zeros_like(b)      # [0; b_b]: the merged projection launched WITHOUT the first Linear's bias


def wants_transpose(out_features: int, in_features: int) -> bool:
    """Linears whose data gradient dX = dY . W has a reduction (out_features) the weight-stationary kernel takes
    (csrc/wres_gemm_bf16.cuh: 288 or 384): their shadow also keeps W^T."""
    return out_features in (288, 384) and in_features % 8 == 0


def lookup(param: torch.Tensor, scale: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
    """The valid bf16 shadow of ``param`` or None.  ``scale``: the BatchNorm scale the caller is about to fold -- a shadow
    folded with a different scale tensor is not valid for it."""
    e = _entries.get(id(param))
    if (e is not None and e.ref() is param and e.version == param._version and e.dst.device == param.device and
            e.dst.data_ptr() != 0 and (scale is None or e.scale is None or e.scale is scale)):
        return e.dst
    return None


def lookup_t(param: torch.Tensor) -> Optional[torch.Tensor]:
    """The valid bf16 TRANSPOSED shadow of a weight, or None: W^T [in, out] of a Linear; [Cin, Cout, 3, 3] (channels_last)
    of a trainable 3x3 convolution (BatchNorm scale folded like the shadow itself)."""
    e = _entries.get(id(param))
    if (e is not None and e.ref() is param and e.version == param._version and e.dst_t is not None and
            e.dst_t.device == param.device):
        return e.dst_t
    return None


# the patch-resident 3x3 kernel for the stride-1 convolutions (SNIPPER_CONV_PATCH=0: the implicit-GEMM kernels, for A/B runs)
import os as _os
CONV_PATCH = _os.environ.get("SNIPPER_CONV_PATCH", "1") != "0"


def lookup_packed(param: torch.Tensor, scale: Optional[torch.Tensor] = None):
    """(packed, packed_t) of a stride-1 3x3 convolution weight whose shadow is valid -- the folded bf16 weight in the fragment
    order of csrc/conv3x3_patch_bf16.cuh for the forward and (trainable weights; else None) for the data gradient -- or None."""
    e = _entries.get(id(param))
    if (e is not None and e.packed is not None and e.ref() is param and e.version == param._version and
            e.packed.device == param.device and (scale is None or e.scale is None or e.scale is scale)):
        return e.packed, e.packed_t
    return None


# The one-tap form of the patch kernel for the 1x1 convolutions' / the feed-forward block's 1024-deep products.  In isolation it
# beats the tile kernels on most data gradients by 10-30 % and on the forward products with a reduction >= 256 by 5-20 %
# (tools/linpatchbench.py, profiles/r05_linear_patch_bench.jsonl).  Round 5 left it opt-in: the extra look-ups cost ~1 ms of host
# issue time when the step was host-bound (same-box 22.40 / 22.44 ms without, 23.5 / 25.8 ms with).  Round 6 took 6.6 ms off the
# host (one native call per layer and direction), the step is GPU-bound with a 9 ms margin, and the same switch now measures
# 21.10 / 21.12 -> 21.00 / 20.97 ms (profiles/r06_linear_patch_step_ab.txt): ON by default, SNIPPER_LINEAR_PATCH=0 = tile kernels.
LINEAR_PATCH = _os.environ.get("SNIPPER_LINEAR_PATCH", "1") != "0"
CAST_TABLE = _os.environ.get("SNIPPER_CAST_TABLE", "1") != "0"      # one-launch weight casts (0: PyTorch multi-tensor ops, A/B)
# Full-width (128 / 160 rows x 384 columns, 8 waves) tiles for the deep reductions into 384 columns: the feed-forward block's
# linear2 forward and linear1 data gradient (csrc/conv3x3_patch_bf16.cuh, linear_wide_kernel).  SNIPPER_LINEAR_WIDE=0: tile kernels.
LINEAR_WIDE = _os.environ.get("SNIPPER_LINEAR_WIDE", "1") != "0"


def lookup_lpacked(param: torch.Tensor, scale: Optional[torch.Tensor] = None):
    """(packed, packed_t) of a 1x1 convolution weight or a Linear weight whose shadow is valid -- the (folded) bf16 matrix
    [out, in] in the fragment order of csrc/conv3x3_patch_bf16.cuh (one tap) for the forward (None where the tile kernel is
    kept) and for the data gradient (None for frozen weights) -- or None."""
    e = _entries.get(id(param))
    if (e is not None and (e.packed is not None or e.packed_t is not None) and e.ref() is param and
            e.version == param._version and e.dst.device == param.device and
            (scale is None or e.scale is None or e.scale is scale)):
        return e.packed, e.packed_t
    return None


def patch_forward_pays(out_features: int, in_features: int) -> bool:
    """Forward products where the one-tap patch kernel beat the tile kernels (tools/linpatchbench.py, profiles/
    r05_linear_patch_bench.jsonl): a reduction of at least 256 channels into at least 128 columns, no residual."""
    return in_features >= 256 and out_features >= 128 and in_features % 64 == 0 and out_features % 64 == 0


def lookup_merged_t(lin_a: nn.Linear, lin_b: nn.Linear) -> Optional[torch.Tensor]:
    if lookup_merged(lin_a, lin_b) is None:
        return None
    return _merged[(id(lin_a.weight), id(lin_b.weight))].w_t


def invalidate(params=None) -> None:
    """Drop the shadows of ``params`` (an iterable of parameters), or all of them.  Validity is keyed on the parameter's
    version counter; writes that do not bump it -- ``p.data.copy_()``, an EMA swap through ``.data``, reference-style
    ``.data`` initialisers run after the first refresh -- must be followed by this call (or by ``torch.autograd.graph.
    increment_version``)."""
    if params is None:
        _entries.clear()
        _merged.clear()
        return
    ids = {id(p) for p in params}
    for k in [k for k in _entries if k in ids]:
        del _entries[k]
    for k in [k for k in _merged if k[0] in ids or k[1] in ids]:
        del _merged[k]


def lookup_merged(lin_a: nn.Linear, lin_b: nn.Linear, second_bias_only: bool = False):
    """(bf16 [W_a; W_b], float32 bias) of the merged projection, or None.  ``second_bias_only``: the bias is [0; b_b] -- the
    caller adds b_a itself, in float32 (the sampling offsets' bias grid: csrc/msda_prologue.cuh)."""
    m = _merged.get((id(lin_a.weight), id(lin_b.weight)))
    if m is None:
        return None
    ps = [lin_a.weight, lin_a.bias, lin_b.weight, lin_b.bias]
    if all(r() is p for r, p in zip(m.refs, ps)) and m.versions == [p._version for p in ps] and m.w.device == ps[0].device:
        return m.w, (m.b_second_only if second_bias_only else m.b)
    return None


_hook_installed = False


def _after_optimizer_step(optimizer, args, kwargs) -> None:
    ps = [p for g in optimizer.param_groups for p in g["params"] if isinstance(p, torch.Tensor)]
    if ps:
        torch.autograd.graph.increment_version(ps)


def install_optimizer_hook() -> None:
    """Every optimizer step from now on bumps its parameters' version counters (see the module docstring)."""
    global _hook_installed
    if not _hook_installed:
        from torch.optim.optimizer import register_optimizer_step_post_hook
        register_optimizer_step_post_hook(_after_optimizer_step)
        _hook_installed = True


class WeightShadows:
    def __init__(self, model: nn.Module):
        install_optimizer_hook()
        from .backbone import Bottleneck
        from .deformable_transformer import DeformableTransformerEncoderLayer
        from .ms_deform_attn import MSDeformAttn
        self.convs: List[Tuple[nn.Conv2d, nn.Module]] = []
        self.linears: List[nn.Linear] = []
        self.pairs: List[Tuple[nn.Linear, nn.Linear]] = []
        for mod in model.modules():
            if isinstance(mod, Bottleneck):
                self.convs += [(mod.conv1, mod.bn1), (mod.conv2, mod.bn2), (mod.conv3, mod.bn3)]
                if mod.downsample is not None:
                    self.convs.append((mod.downsample[0], mod.downsample[1]))
            elif isinstance(mod, MSDeformAttn):
                self.linears += [mod.value_proj, mod.output_proj]
                if mod.weights_are_tied():
                    self.pairs.append((mod.sampling_offsets[0], mod.attention_weights[0]))
            elif isinstance(mod, DeformableTransformerEncoderLayer):
                self.linears += [mod.linear1, mod.linear2]
        for proj in getattr(model, "input_proj", []):
            if isinstance(proj, nn.Sequential) and isinstance(proj[0], nn.Conv2d):
                self.linears.append(proj[0])                     # a 1x1 convolution's weight is used as [Cout, Cin]

    @torch.no_grad()
    def refresh(self) -> None:
        """Bring every stale shadow up to date (multi-tensor launches; nothing to do for unchanged parameters)."""
        mul_src, mul_scale, mul_dst, cp_src, cp_dst, tr, packs, lpacks = [], [], [], [], [], [], [], []
        for conv, bn in self.convs:
            w = conv.weight
            if not (w.is_cuda and w.dtype == torch.float32):
                continue
            e = _entries.get(id(w))
            if e is None or e.ref() is not w or e.dst.device != w.device:
                e = _entries[id(w)] = _Entry(w, torch.empty_like(w, dtype=torch.bfloat16), None)
            scale = bn.scale_bias()[0]
            patch = (CONV_PATCH and w.dim() == 4 and tuple(w.shape[2:]) == (3, 3) and tuple(conv.stride) == (1, 1) and
                     w.is_contiguous(memory_format=torch.channels_last) and w.shape[0] % 64 == 0 and w.shape[1] % 64 == 0)
            pw = (LINEAR_PATCH and w.dim() == 4 and tuple(w.shape[2:]) == (1, 1) and w.shape[0] % 64 == 0 and
                  w.shape[1] % 64 == 0 and e.dst.stride(0) == w.shape[1] and e.dst.stride(1) == 1)     # dense [Cout][Cin] memory
            if pw and e.packed is None and e.packed_t is None:
                # a 1x1 convolution = a product on NHWC rows: packed for the one-tap patch kernel where it pays (forward) and,
                # for trainable weights, for the data gradient
                if patch_forward_pays(w.shape[0], w.shape[1]):
                    e.packed = torch.empty(w.numel(), dtype=torch.bfloat16, device=w.device)
                if w.requires_grad:
                    e.packed_t = torch.empty(w.numel(), dtype=torch.bfloat16, device=w.device)
            if patch and e.packed is None:
                # a stride-1 3x3 convolution runs on the patch-resident kernel: its weight in fragment order, written by ONE
                # pack launch below for all of them (and no tap-transposed copy: that kernel's data gradient has its own pack)
                e.packed = torch.empty(w.numel(), dtype=torch.bfloat16, device=w.device)
                if w.requires_grad:
                    e.packed_t = torch.empty(w.numel(), dtype=torch.bfloat16, device=w.device)
            if (not patch and e.dst_t is None and w.dim() == 4 and tuple(w.shape[2:]) == (3, 3) and w.requires_grad and
                    w.is_contiguous(memory_format=torch.channels_last) and w.shape[0] % 8 == 0 and w.shape[1] % 8 == 0):
                # a trainable 3x3 convolution's data gradient multiplies by the weight with its channel roles swapped
                # ([Cin, Cout, 3, 3], channels_last; backbone._Conv3x3BN): kept beside the shadow, written by the batched
                # transpose below (nine [Cout, Cin] tap matrices per weight) instead of one strided copy per layer and step
                e.dst_t = torch.empty((w.shape[1], w.shape[0], 3, 3), dtype=torch.bfloat16,
                                      device=w.device).contiguous(memory_format=torch.channels_last)
            if e.version != w._version or e.scale is not scale:
                if e.dst_t is not None:
                    if e.tap_pairs is None:       # (18 slicing calls per weight: ~0.4 ms of host time per step when rebuilt every time)
                        src_t, dst_t = e.dst.permute(0, 2, 3, 1), e.dst_t.permute(0, 2, 3, 1)  # [Cout, 3, 3, Cin] / [Cin, 3, 3, Cout] views
                        e.tap_pairs = [(src_t[:, ky, kx, :], dst_t[:, ky, kx, :]) for ky in range(3) for kx in range(3)]
                    tr += e.tap_pairs
                if pw:
                    m2 = e.dst.as_strided((w.shape[0], w.shape[1]), (w.shape[1], 1))      # (a VIEW of the shadow: it is written below)
                    if e.packed is not None:
                        lpacks.append((m2, e.packed, False))
                    if e.packed_t is not None:
                        lpacks.append((m2, e.packed_t, True))
                elif e.packed is not None:
                    packs.append((e.dst, e.packed, False))
                    if e.packed_t is not None:
                        packs.append((e.dst, e.packed_t, True))
                if not CAST_TABLE and (e.scale is not scale or e.scale_full is None):
                    # (the multi-tensor fallback) frozen BatchNorm: built once.  A broadcast operand sends _foreach_mul down its
                    # one-kernel-per-tensor path (42 launches per step here); a full-size one with the weight's strides keeps
                    # it multi-tensor.
                    e.scale_full = torch.empty_like(w).copy_(scale.view(-1, 1, 1, 1).expand_as(w))
                mul_src.append(w)
                mul_scale.append(e.scale_full)
                mul_dst.append(e)
                e.scale = scale
        for lin in self.linears:
            w = lin.weight
            if not (w.is_cuda and w.dtype == torch.float32):
                continue
            e = _entries.get(id(w))
            if e is None or e.ref() is not w or e.dst.device != w.device:
                e = _entries[id(w)] = _Entry(w, torch.empty_like(w, dtype=torch.bfloat16))
                if w.dim() == 2 and wants_transpose(w.shape[0], w.shape[1]):
                    e.dst_t = torch.empty((w.shape[1], w.shape[0]), dtype=torch.bfloat16, device=w.device)
                if LINEAR_WIDE and w.dim() == 2 and e.packed is None and w.shape[0] == 384 and w.shape[1] >= 512 and w.shape[1] % 128 == 0:
                    e.packed = torch.empty(w.numel(), dtype=torch.bfloat16, device=w.device)        # forward: deep reduction -> 384
                if (LINEAR_WIDE and w.dim() == 2 and e.packed_t is None and w.shape[1] == 384 and w.shape[0] >= 512 and
                        w.shape[0] % 128 == 0 and w.requires_grad):
                    e.packed_t = torch.empty(w.numel(), dtype=torch.bfloat16, device=w.device)      # its data gradient likewise
                if LINEAR_PATCH and w.dim() == 2 and w.shape[0] % 64 == 0 and w.shape[1] % 64 == 0:
                    # the feed-forward block's 1024-deep products: linear2's forward, linear1's data gradient
                    if w.shape[1] >= 512 and w.shape[0] >= 128 and e.packed is None:
                        e.packed = torch.empty(w.numel(), dtype=torch.bfloat16, device=w.device)
                    if w.shape[0] >= 512 and w.shape[1] >= 128 and w.requires_grad and e.packed_t is None:
                        e.packed_t = torch.empty(w.numel(), dtype=torch.bfloat16, device=w.device)
            if e.version != w._version:
                cp_src.append(w)
                cp_dst.append(e)
                if e.dst_t is not None:
                    tr.append((e.dst, e.dst_t))
                if w.dim() == 2 and e.packed is not None:
                    lpacks.append((e.dst, e.packed, False))
                if w.dim() == 2 and e.packed_t is not None:
                    lpacks.append((e.dst, e.packed_t, True))
        fin = []
        for a, b in self.pairs:
            ps = [a.weight, a.bias, b.weight, b.bias]
            if not all(p is not None and p.is_cuda and p.dtype == torch.float32 for p in ps):
                continue
            key = (id(a.weight), id(b.weight))
            m = _merged.get(key)
            if m is None or any(r() is not p for r, p in zip(m.refs, ps)) or m.w.device != ps[0].device:
                na, nb, k = a.out_features, b.out_features, a.in_features
                m = _merged[key] = _Merged(ps, torch.empty((na + nb, k), dtype=torch.bfloat16, device=ps[0].device),
                                           torch.empty((na + nb,), dtype=torch.float32, device=ps[0].device))
                if wants_transpose(na + nb, k):
                    m.w_t = torch.empty((k, na + nb), dtype=torch.bfloat16, device=ps[0].device)
            vers = [p._version for p in ps]
            if m.versions != vers:
                na = a.out_features
                cp_src += [a.weight, b.weight, a.bias, b.bias, b.bias]
                cp_dst += [m.w[:na], m.w[na:], m.b[:na], m.b[na:], m.b_second_only[na:]]
                fin.append((m, vers))
                if m.w_t is not None:
                    tr.append((m.w, m.w_t))
        # ONE launch for every bf16 <- float32 weight copy of the step, the frozen-BN scale folded in per output channel
        # (csrc/misc_kernels.cuh, cast_scale_table_kernel; the pointer table is uploaded once): it replaces a _foreach_mul by
        # full-size scale copies into temporaries and three _foreach_copy_ casts (162 -> ~45 us per step).  Tensors the kernel
        # does not take (and the float32 <- float32 bias copies) keep the multi-tensor path below.
        if CAST_TABLE and (mul_src or cp_src):
            items = [(w, e.dst, e.scale) for w, e in zip(mul_src, mul_dst)]
            rest_src, rest_dst = [], []
            for d, w in zip(cp_dst, cp_src):
                t = d.dst if isinstance(d, _Entry) else d
                if t.dtype == torch.bfloat16 and w.dtype == torch.float32:
                    items.append((w, t, None))
                else:
                    rest_src.append(w)
                    rest_dst.append(d)
            from .dense import cast_scale_table_bf16
            if all(sc is None or (sc.dtype == torch.float32 and sc.dim() == 1) for _, _, sc in items) and cast_scale_table_bf16(items):
                for e, w in zip(mul_dst, mul_src):
                    e.version = w._version
                for d, w in zip(cp_dst, cp_src):
                    if isinstance(d, _Entry):
                        d.version = w._version
                mul_src, cp_src, cp_dst = [], rest_src, rest_dst
        if mul_src:
            tmp = torch._foreach_mul(mul_src, [e.scale_full if e.scale_full is not None else
                                               torch.empty_like(w).copy_(e.scale.view(-1, 1, 1, 1).expand_as(w))
                                               for e, w in zip(mul_dst, mul_src)])
            torch._foreach_copy_([e.dst for e in mul_dst], tmp)
            for e, w in zip(mul_dst, mul_src):
                e.version = w._version
        if cp_src:
            # one multi-tensor launch per (destination dtype, source dtype) pair: a list that mixes bf16 <- f32 weight
            # copies with f32 <- f32 bias copies is NOT safe -- PyTorch 2.10's _foreach_copy_ converted every source to
            # the FIRST destination's dtype and wrote bf16 bits into the float32 bias buffers of the merged projections
            # (found in round 3: the offset projection lost its bias under autocast)
            groups: Dict[Tuple[torch.dtype, torch.dtype], Tuple[list, list]] = {}
            for d, w in zip(cp_dst, cp_src):
                t = d.dst if isinstance(d, _Entry) else d
                gd, gs = groups.setdefault((t.dtype, w.dtype), ([], []))
                gd.append(t)
                gs.append(w)
            for gd, gs in groups.values():
                torch._foreach_copy_(gd, gs)
            for d, w in zip(cp_dst, cp_src):
                if isinstance(d, _Entry):
                    d.version = w._version
        if packs:                               # (after the folded bf16 copies they read have been written)
            from .dense import conv3x3_pack_bf16
            conv3x3_pack_bf16(packs)
        if lpacks:
            from .dense import linear_pack_bf16
            linear_pack_bf16(lpacks)
        if tr:                                  # W^T of the freshly written bf16 copies: one launch for all of them
            from .dense import transpose_batch_bf16
            transpose_batch_bf16(tr)
        for m, vers in fin:
            m.versions = vers
