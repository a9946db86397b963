"""Drop-in for the reference's pybind module ``MultiScaleDeformableAttention``.

The reference builds a torch C++/CUDA extension of that name exposing two functions
(/root/reference/models/ops/src/vision.cpp:13-16, ms_deform_attn.h:20-62) and imports it as
``MSDA`` (models/ops/functions/ms_deform_attn_func.py:18-21).  This module offers the same two
functions with the same argument lists and error behaviour, implemented on the gfx950 HIP
library through its C ABI (include/snipper_msda.h).  ``snipper_amd.install()`` registers it
under the reference's module name.

Differences, all additive: bfloat16 ``value`` is accepted (loc/attn are then float32), launch
errors raise instead of being printed (.cuh:948-952), and ``im2col_step`` only has to satisfy
the reference's divisibility rule -- the HIP kernels take the whole batch in one launch.
"""
from __future__ import annotations

import ctypes
from typing import List

import torch

from . import _lib

_SUFFIX = {torch.float32: "f32", torch.float64: "f64", torch.bfloat16: "bf16"}


def _require(cond: bool, msg: str) -> None:
    if not cond:
        raise RuntimeError(msg)


def _check_common(named, value, spatial_shapes, level_start_index, im2col_step):
    # ms_deform_attn.h:38,60 -- the reference has no CPU implementation either
    _require(value.is_cuda, "Not implemented on the CPU")
    for name, t in named:   # ms_deform_attn_cuda.cu:28-38,93-105
        _require(t.is_contiguous(), f"{name} tensor has to be contiguous")
        _require(t.is_cuda, f"{name} must be a CUDA tensor")
        _require(t.device == value.device, f"{name} must be on the same device as value")
    _require(value.dtype in _SUFFIX, f"ms_deform_attn not implemented for '{value.dtype}'")
    _require(spatial_shapes.dtype == torch.int64 and level_start_index.dtype == torch.int64,
             "spatial_shapes and level_start_index must be int64")
    batch = value.size(0)
    step = min(batch, int(im2col_step))      # ms_deform_attn_cuda.cu:50-52
    _require(step > 0 and batch % step == 0, f"batch({batch}) must divide im2col_step({step})")


def _dims(value, spatial_shapes, sampling_loc, attn_weight):
    _require(value.dim() == 4 and sampling_loc.dim() == 6 and attn_weight.dim() == 5, "bad tensor ranks")
    N, S, M, D = value.shape
    L = spatial_shapes.size(0)
    Lq, P = sampling_loc.size(1), sampling_loc.size(4)
    _require(tuple(sampling_loc.shape) == (N, Lq, M, L, P, 2), "sampling_loc shape mismatch")
    _require(tuple(attn_weight.shape) == (N, Lq, M, L, P), "attn_weight shape mismatch")
    _require(level_start_index_ok(spatial_shapes), "spatial_shapes must be [L,2]")
    return N, S, M, D, L, Lq, P


def level_start_index_ok(spatial_shapes) -> bool:
    return spatial_shapes.dim() == 2 and spatial_shapes.size(1) == 2


def _coord_dtype(value):
    return torch.float32 if value.dtype == torch.bfloat16 else value.dtype


def _stream(device) -> int:
    return _lib.raw_stream(device)


# ---- optional launch timing (bench.py) --------------------------------------------------------
# When enabled, every C-ABI call is bracketed by two events recorded on the SAME stream the
# kernel is launched on; nothing synchronises until the caller reads the list.
_timing = None


def enable_launch_timing(on: bool = True) -> None:
    global _timing
    _timing = [] if on else None


def launch_timings():
    """[(kind, variant, dims dict, elapsed_ms)] for the calls since enable_launch_timing(True);
    synchronises the device."""
    if not _timing:
        return []
    torch.cuda.synchronize()
    return [(k, v, d, s.elapsed_time(e)) for (k, v, d, s, e) in _timing]


class _Timed:
    def __init__(self, kind, dims, device):
        self.kind, self.dims, self.device = kind, dims, device

    def __enter__(self):
        if _timing is not None:
            self.start = torch.cuda.Event(enable_timing=True)
            self.start.record(torch.cuda.current_stream(self.device))
        return self

    def __exit__(self, *exc):
        _lib.note_variant()
        if _timing is not None and exc[0] is None:
            end = torch.cuda.Event(enable_timing=True)
            end.record(torch.cuda.current_stream(self.device))
            _timing.append((self.kind, _lib.last_variant(), self.dims, self.start, end))
        return False


_DT = {torch.float32: 0, torch.bfloat16: 1, torch.float64: 2}


def _cfg_ptr(config):
    cfg = config if config is not None else _lib.active_config()
    return ctypes.byref(cfg) if cfg is not None else None


def _host_shapes_ptr(host_shapes, L):
    if host_shapes is None or len(host_shapes) != L:
        return None, None
    hs = (ctypes.c_int64 * (2 * L))(*[int(v) for hw in host_shapes for v in hw])
    return hs, ctypes.cast(hs, ctypes.c_void_p)


def ms_deform_attn_forward(value: torch.Tensor, spatial_shapes: torch.Tensor,
                           level_start_index: torch.Tensor, sampling_loc: torch.Tensor,
                           attn_weight: torch.Tensor, im2col_step: int, out_bf16: bool = False,
                           host_shapes=None, config=None, out_f32: bool = False) -> torch.Tensor:
    """-> Tensor[N, Lq, M*D]   (ms_deform_attn_cuda.cu:20-80)

    Extensions (all optional): ``out_bf16`` -- float32 ``value``, output rows written as bfloat16 by the kernel (exactly
    the float32 result rounded once; shapes without a bf16-row kernel get the float32 result cast here);
    ``out_f32`` -- bfloat16 ``value``, output rows written as float32 by the kernel (the float32 sums as they are, for a
    float32 consumer: the decoder's output projection; shapes without such a kernel get the bfloat16 result cast here);
    ``host_shapes`` -- the values of ``spatial_shapes`` as a host list [(H, W), ...], which lets the library run its
    encoder-shape kernels (include/snipper_msda.h) without a device-to-host copy; ``config`` -- a ``_lib.Config``."""
    _check_common([("value", value), ("spatial_shapes", spatial_shapes),
                   ("level_start_index", level_start_index), ("sampling_loc", sampling_loc),
                   ("attn_weight", attn_weight)], value, spatial_shapes, level_start_index, im2col_step)
    cd = _coord_dtype(value)
    _require(sampling_loc.dtype == cd and attn_weight.dtype == cd,
             f"sampling_loc/attn_weight must be {cd} for {value.dtype} value")
    N, S, M, D, L, Lq, P = _dims(value, spatial_shapes, sampling_loc, attn_weight)
    lib = _lib.load()
    rows16 = out_bf16 and value.dtype == torch.float32
    rows32 = out_f32 and value.dtype == torch.bfloat16
    dims = dict(N=N, S=S, M=M, D=D, L=L, Lq=Lq, P=P, esize=value.element_size(),
                row_esize=2 if rows16 else (4 if rows32 else value.element_size()))
    _keep, hs_p = _host_shapes_ptr(host_shapes, L)
    for out_dtype in ((torch.bfloat16, value.dtype) if rows16 else ((torch.float32, value.dtype) if rows32 else (value.dtype,))):
        out = torch.empty((N, Lq, M * D), dtype=out_dtype, device=value.device)
        with _lib.device_guard(value.device), _Timed("fwd", dims, value.device):
            rc = lib.snipper_msda_forward_ex(
                _stream(value.device), _cfg_ptr(config), hs_p, value.data_ptr(), _DT[value.dtype],
                spatial_shapes.data_ptr(), level_start_index.data_ptr(), sampling_loc.data_ptr(), attn_weight.data_ptr(),
                N, S, M, D, L, Lq, P, out.data_ptr(), _DT[out_dtype])
        if rc == _lib.E_UNSUPPORTED and out_dtype != value.dtype:
            continue                                      # no bf16-row kernel for this shape: float32, cast below
        _lib.check(rc, "ms_deform_attn_forward")
        break
    if rows32 and out.dtype != torch.float32:
        return out.float()
    return out.to(torch.bfloat16) if (rows16 and out.dtype != torch.bfloat16) else out


def ms_deform_attn_backward(value: torch.Tensor, spatial_shapes: torch.Tensor,
                            level_start_index: torch.Tensor, sampling_loc: torch.Tensor,
                            attn_weight: torch.Tensor, grad_output: torch.Tensor,
                            im2col_step: int, host_shapes=None, config=None,
                            grad_value_f32: bool = False) -> List[torch.Tensor]:
    """-> [grad_value, grad_sampling_loc, grad_attn_weight]   (ms_deform_attn_cuda.cu:83-153)

    For bfloat16 ``value`` the returned grad_value is bfloat16 (accumulated in float32).  ``grad_output`` may be bfloat16
    beside float32 ``value`` (bf16 rows, D == 48).  ``host_shapes`` / ``config``: as for the forward; with the host
    shapes the encoder shape (Lq == S) takes the owner-computes backward.  ``grad_value_f32``: return the float32
    accumulation buffer for bfloat16 ``value`` as it is (a caller that reduces it further spares two cast passes).
    """
    _check_common([("value", value), ("spatial_shapes", spatial_shapes),
                   ("level_start_index", level_start_index), ("sampling_loc", sampling_loc),
                   ("attn_weight", attn_weight), ("grad_output", grad_output)],
                  value, spatial_shapes, level_start_index, im2col_step)
    cd = _coord_dtype(value)
    _require(sampling_loc.dtype == cd and attn_weight.dtype == cd,
             f"sampling_loc/attn_weight must be {cd} for {value.dtype} value")
    go_bf16 = grad_output.dtype == torch.bfloat16 and value.dtype == torch.float32      # extension: bf16 rows in
    go_f32 = grad_output.dtype == torch.float32 and value.dtype == torch.bfloat16       # extension: float32 rows beside a bf16 value
    _require(grad_output.dtype == value.dtype or go_bf16 or go_f32, "grad_output dtype must match value")
    N, S, M, D, L, Lq, P = _dims(value, spatial_shapes, sampling_loc, attn_weight)
    _require(grad_output.numel() == N * Lq * M * D, "grad_output shape mismatch")
    lib = _lib.load()
    if (value.dtype == torch.bfloat16 and not grad_value_f32 and config is None and Lq != S and D == 48 and Lq <= 64):
        # few queries on a large bf16 value (the decoder's cross attention): grad_value without atomics, float32 buffer or
        # cast pass (csrc/msda_d48_sparse.cuh); any other shape answers E_UNSUPPORTED and takes the general entry below
        gv = torch.empty(value.shape, dtype=torch.bfloat16, device=value.device)
        gl, ga = torch.empty_like(sampling_loc), torch.empty_like(attn_weight)
        dims_s = dict(N=N, S=S, M=M, D=D, L=L, Lq=Lq, P=P, esize=2, row_esize=4 if go_f32 else 2)
        entry = lib.snipper_msda_backward_sparse_f32rows if go_f32 else lib.snipper_msda_backward_sparse_bf16
        with _lib.device_guard(value.device), _Timed("bwd", dims_s, value.device):
            rc = entry(
                _stream(value.device), grad_output.data_ptr(), value.data_ptr(), spatial_shapes.data_ptr(),
                level_start_index.data_ptr(), sampling_loc.data_ptr(), attn_weight.data_ptr(), N, S, M, D, L, Lq, P,
                gv.data_ptr(), gl.data_ptr(), ga.data_ptr())
        if rc == 0:
            return [gv, gl, ga]
        if rc != _lib.E_UNSUPPORTED:
            _lib.check(rc, "snipper_msda_backward_sparse")
    if go_f32:
        grad_output = grad_output.to(torch.bfloat16)         # (every other bf16-value kernel reads bf16 rows)
    acc_dtype = torch.float32 if value.dtype == torch.bfloat16 else value.dtype
    # grad_value is fully written by the callee, no pre-zeroing (include/snipper_msda.h, "Outputs")
    grad_value = torch.empty(value.shape, dtype=acc_dtype, device=value.device)
    grad_loc = torch.empty_like(sampling_loc)
    grad_attn = torch.empty_like(attn_weight)
    dims = dict(N=N, S=S, M=M, D=D, L=L, Lq=Lq, P=P, esize=value.element_size(),
                row_esize=grad_output.element_size())
    _keep, hs_p = _host_shapes_ptr(host_shapes, L)
    cfg_p = _cfg_ptr(config)
    with _lib.device_guard(value.device), _Timed("bwd", dims, value.device):
        ws_bytes = lib.snipper_msda_backward_ex_workspace_bytes(cfg_p, hs_p, _DT[value.dtype], N, S, M, D, L, Lq, P) \
            if hs_p is not None else 0
        workspace = torch.empty(ws_bytes, dtype=torch.uint8, device=value.device) if ws_bytes > 0 else None
        for _ in range(2):
            rc = lib.snipper_msda_backward_ex(
                _stream(value.device), cfg_p, hs_p, workspace.data_ptr() if workspace is not None else None, ws_bytes,
                grad_output.data_ptr(), _DT[grad_output.dtype], value.data_ptr(), _DT[value.dtype],
                spatial_shapes.data_ptr(), level_start_index.data_ptr(), sampling_loc.data_ptr(), attn_weight.data_ptr(),
                N, S, M, D, L, Lq, P, grad_value.data_ptr(), grad_loc.data_ptr(), grad_attn.data_ptr())
            if rc == _lib.E_UNSUPPORTED and go_bf16 and grad_output.dtype == torch.bfloat16:
                grad_output = grad_output.float()        # no bf16-row kernel for this shape: widen and go on
                continue
            break
    _lib.check(rc, "ms_deform_attn_backward")
    if grad_value.dtype != value.dtype and not grad_value_f32:
        grad_value = grad_value.to(value.dtype)
    return [grad_value, grad_loc, grad_attn]
