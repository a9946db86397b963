"""Autograd wrappers of the fused element-wise kernels around the core op
(include/snipper_dense.h, csrc/msda_prologue.cuh).  CUDA tensors, float32 / bfloat16 only;
``MSDeformAttn`` falls back to the plain PyTorch formulation for anything else (e.g. float64)."""
from __future__ import annotations

import ctypes
from typing import List, Optional, Sequence

import torch
from torch.autograd import Function

from . import _lib

_DT = {torch.float32: 0, torch.bfloat16: 1}


def _stream(dev):
    return torch.cuda.current_stream(dev).cuda_stream


def _farray(vals: Sequence[float]):
    return (ctypes.c_float * len(vals))(*[float(v) for v in vals])


def supported(t: torch.Tensor) -> bool:
    return t.is_cuda and t.dtype in _DT


def _mix_launch(x, mask, mask_on_input, mix_rows: List[List[float]], out_dtype):
    N, Ti, S, C = x.shape
    To = len(mix_rows)
    out = torch.empty((N, To, S, C), dtype=out_dtype, device=x.device)
    flat = _farray([w for row in mix_rows for w in row])
    with torch.cuda.device(x.device):
        rc = _lib.load().snipper_temporal_mix(
            _stream(x.device), x.data_ptr(), _DT[x.dtype], mask.data_ptr() if mask is not None else None,
            int(mask_on_input), ctypes.cast(flat, ctypes.c_void_p), N, Ti, To, S, C, out.data_ptr(), _DT[out_dtype])
    _lib.check(rc, "snipper_temporal_mix")
    return out


class TemporalMix(Function):
    """out[n, t1] = sum_t2 mix[t1][t2] * (mask[n, t2] ? 0 : value[n, t2]), float32 out.

    value [N, T2, S, C] (f32 / bf16, contiguous), mask [N, T2, S] uint8 / bool or None,
    mix: host list [T1][T2].  Backward is the transposed mix with the mask applied to its output."""

    @staticmethod
    def forward(ctx, value, mask, mix):
        value = value.contiguous()
        m8 = None
        if mask is not None:
            m8 = mask.contiguous().view(torch.uint8) if mask.dtype == torch.bool else mask.contiguous()
        ctx.mix, ctx.in_dtype = mix, value.dtype
        ctx.save_for_backward(m8) if m8 is not None else None
        ctx.has_mask = m8 is not None
        return _mix_launch(value, m8, True, mix, torch.float32)

    @staticmethod
    def backward(ctx, grad_out):
        m8 = ctx.saved_tensors[0] if ctx.has_mask else None
        mix = ctx.mix
        mix_t = [[mix[a][b] for a in range(len(mix))] for b in range(len(mix[0]))]
        g = _mix_launch(grad_out.contiguous().float(), m8, False, mix_t, ctx.in_dtype)
        return g, None, None


class MSDAPrologue(Function):
    """(raw offsets, raw logits, reference points) -> (sampling locations, attention probabilities).

    off [..., M*L*P*2], logit [..., M*L*P] (f32 / bf16), ref [..., L, 2] float32 (leading dims = those of
    off), hw: host list of (H, W).  Returns loc [rows, L, P, 2] and prob [rows, L, P] as float32 with
    rows = prod(leading dims) * M."""

    @staticmethod
    def forward(ctx, off, logit, ref, hw, M, L, P):
        off, logit = off.contiguous(), logit.contiguous()
        ref = ref.contiguous().float()
        rows = off.numel() // (L * P * 2)
        loc = torch.empty((rows, L, P, 2), dtype=torch.float32, device=off.device)
        prob = torch.empty((rows, L, P), dtype=torch.float32, device=off.device)
        inv_w, inv_h = _farray([1.0 / w for h, w in hw]), _farray([1.0 / h for h, w in hw])
        with torch.cuda.device(off.device):
            rc = _lib.load().snipper_msda_prologue_forward(
                _stream(off.device), off.data_ptr(), logit.data_ptr(), _DT[off.dtype], ref.data_ptr(),
                ctypes.cast(inv_w, ctypes.c_void_p), ctypes.cast(inv_h, ctypes.c_void_p), rows, M, L, P,
                loc.data_ptr(), prob.data_ptr())
        _lib.check(rc, "snipper_msda_prologue_forward")
        ctx.save_for_backward(prob)
        ctx.meta = (hw, M, L, P, off.dtype, off.shape, logit.shape, ref.shape, ctx.needs_input_grad[2])
        return loc, prob

    @staticmethod
    def backward(ctx, grad_loc, grad_prob):
        (prob,) = ctx.saved_tensors
        hw, M, L, P, dtype, off_shape, logit_shape, ref_shape, need_ref = ctx.meta
        rows = prob.shape[0]
        grad_loc = grad_loc.contiguous().float() if grad_loc is not None else torch.zeros(rows, L, P, 2, device=prob.device)
        grad_prob = grad_prob.contiguous().float() if grad_prob is not None else torch.zeros_like(prob)
        g_off = torch.empty(off_shape, dtype=dtype, device=prob.device)
        g_logit = torch.empty(logit_shape, dtype=dtype, device=prob.device)
        g_ref = torch.empty(ref_shape, dtype=torch.float32, device=prob.device) if need_ref else None
        inv_w, inv_h = _farray([1.0 / w for h, w in hw]), _farray([1.0 / h for h, w in hw])
        with torch.cuda.device(prob.device):
            rc = _lib.load().snipper_msda_prologue_backward(
                _stream(prob.device), grad_loc.data_ptr(), grad_prob.data_ptr(), prob.data_ptr(),
                ctypes.cast(inv_w, ctypes.c_void_p), ctypes.cast(inv_h, ctypes.c_void_p), rows, M, L, P,
                g_off.data_ptr(), g_logit.data_ptr(), _DT[dtype], g_ref.data_ptr() if g_ref is not None else None)
        _lib.check(rc, "snipper_msda_prologue_backward")
        return g_off, g_logit, g_ref, None, None, None, None
