"""CPU oracle for Snipper's deformable-attention hot path.

TEST INFRASTRUCTURE ONLY.  May be imported by tests/, by
``__graft_entry__.smoke()`` and by the ``cpu_baseline`` leg of ``bench.py``;
never by anything under ``snipper_amd/`` (the product path fails loudly when
its HIP library is missing -- it does not fall back to this file).

Three independent restatements of the same function, so that they can pin each
other and the reference (citations relative to /root/reference):

* ``core_c``            scalar loops in C (oracle/msda_oracle.c), forward and
                        backward, restating the sampling kernel's semantics
                        (models/ops/src/cuda/ms_deform_im2col_cuda.cuh:33-159,237-299).
* ``core_gridsample``   the ``use_pytorch_deform=1`` formulation
                        (models/ops/functions/ms_deform_attn_func.py:45-65): one
                        ``F.grid_sample`` per level; differentiable through
                        torch autograd.  This is the path BASELINE.json names
                        as the CPU baseline.
* ``st_msdeform_attn``  the spatiotemporal module forward of
                        models/ops/modules/ms_deform_attn.py:99-243 written as
                        one function of explicit weights (per-(t1,t2) pair
                        formulation, joint softmax over L*P*|t2|).

Parity status: pinned.  tests/test_oracle.py checks all of them against golden
vectors produced by importing the reference itself (tests/golden/gen_golden.py).
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libmsda_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    """Compile oracle/msda_oracle.c with gcc (seconds)."""
    src = os.path.join(_HERE, "msda_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libmsda_oracle.so"])
    return _LIB_PATH


def _load():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.msda_oracle_max_threads.restype = ctypes.c_int
    return _lib


def max_threads() -> int:
    return int(_load().msda_oracle_max_threads())


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(ctypes.c_void_p)


def _prep(value, shapes, level_start, loc, attn):
    value = np.ascontiguousarray(value)
    dt = value.dtype
    assert dt in (np.float32, np.float64), dt
    loc = np.ascontiguousarray(loc, dtype=dt)
    attn = np.ascontiguousarray(attn, dtype=dt)
    shapes = np.ascontiguousarray(shapes, dtype=np.int64)
    level_start = np.ascontiguousarray(level_start, dtype=np.int64)
    N, S, M, D = value.shape
    _, Lq, M2, L, P, two = loc.shape
    assert M2 == M and two == 2 and shapes.shape == (L, 2) and attn.shape == (N, Lq, M, L, P)
    assert int((shapes[:, 0] * shapes[:, 1]).sum()) == S
    return value, shapes, level_start, loc, attn, (N, S, M, D, L, Lq, P)


def core_c_forward(value, shapes, level_start, loc, attn, threads: int = 1) -> np.ndarray:
    """out[N,Lq,M*D]; numpy in, numpy out; float32 or float64 (computed in that type)."""
    value, shapes, level_start, loc, attn, dims = _prep(value, shapes, level_start, loc, attn)
    N, S, M, D, L, Lq, P = dims
    out = np.empty((N, Lq, M * D), dtype=value.dtype)
    fn = getattr(_load(), "msda_oracle_forward_f64" if value.dtype == np.float64 else "msda_oracle_forward_f32")
    fn(_ptr(value), _ptr(shapes), _ptr(level_start), _ptr(loc), _ptr(attn),
       N, S, M, D, L, Lq, P, _ptr(out), int(threads))
    return out


def core_c_backward(value, shapes, level_start, loc, attn, grad_out, threads: int = 1):
    """(grad_value, grad_loc, grad_attn) as numpy arrays of the inputs' dtype."""
    value, shapes, level_start, loc, attn, dims = _prep(value, shapes, level_start, loc, attn)
    N, S, M, D, L, Lq, P = dims
    grad_out = np.ascontiguousarray(grad_out, dtype=value.dtype).reshape(N, Lq, M * D)
    gv = np.empty_like(value)
    gl = np.empty_like(loc)
    ga = np.empty_like(attn)
    fn = getattr(_load(), "msda_oracle_backward_f64" if value.dtype == np.float64 else "msda_oracle_backward_f32")
    fn(_ptr(value), _ptr(shapes), _ptr(level_start), _ptr(loc), _ptr(attn), _ptr(grad_out),
       N, S, M, D, L, Lq, P, _ptr(gv), _ptr(gl), _ptr(ga), int(threads))
    return gv, gl, ga


def level_start_index(shapes) -> np.ndarray:
    shapes = np.asarray(shapes, dtype=np.int64)
    sizes = shapes[:, 0] * shapes[:, 1]
    return np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)


def core_gridsample(value: torch.Tensor, shapes, loc: torch.Tensor, attn: torch.Tensor) -> torch.Tensor:
    """The use_pytorch_deform=1 formulation (ms_deform_attn_func.py:45-65), restated.

    value [N,S,M,D] (may be a strided slice), loc [N,Lq,M,L,P,2] in [0,1],
    attn [N,Lq,M,L,P]  ->  [N,Lq,M*D].  Differentiable.
    """
    N, S, M, D = value.shape
    _, Lq, _, L, P, _ = loc.shape
    hw = [(int(h), int(w)) for h, w in (shapes.tolist() if hasattr(shapes, "tolist") else shapes)]
    grid = loc * 2 - 1                                   # func.py:51
    per_level = []
    start = 0
    for l, (H, W) in enumerate(hw):
        # [N, H*W, M, D] -> [N*M, D, H, W]                 func.py:55
        v = value[:, start:start + H * W].permute(0, 2, 3, 1).reshape(N * M, D, H, W)
        start += H * W
        g = grid[:, :, :, l].permute(0, 2, 1, 3, 4).reshape(N * M, Lq, P, 2)   # func.py:57
        per_level.append(F.grid_sample(v, g, mode="bilinear", padding_mode="zeros",
                                       align_corners=False))                    # func.py:59-60
    sampled = torch.stack(per_level, dim=-2).reshape(N * M, D, Lq, L * P)
    w = attn.permute(0, 2, 1, 3, 4).reshape(N * M, 1, Lq, L * P)                # func.py:63
    out = (sampled * w).sum(-1).reshape(N, M * D, Lq)
    return out.transpose(1, 2).contiguous()


def temporal_neighbours(t1: int, n_frame: int, T2: int) -> List[int]:
    """Value frames a query frame attends to (ms_deform_attn.py:132-140,184-189)."""
    if t1 < n_frame:
        return [t for t in (t1 - 1, t1, t1 + 1) if 0 <= t < n_frame]
    return list(range(T2))


def st_msdeform_attn(query, reference_points, input_flatten, shapes, padding_mask,
                     value_w, value_b, off_w: Sequence, off_b: Sequence,
                     att_w: Sequence, att_b: Sequence, out_w, out_b,
                     n_heads: int, n_levels: int, n_points: int, n_frame: int,
                     core=core_gridsample):
    """Spatiotemporal MSDeformAttn.forward (ms_deform_attn.py:99-243), per-pair form.

    ``off_w[t] / off_b[t] / att_w[t] / att_b[t]`` are the Linear parameters indexed by
    value frame ``t`` (the reference ties them; this oracle does not assume it).
    Returns ``(output [N,T1,Lq,C], loc_list, weight_list)`` with the vis lists laid out
    as at :228-233.
    """
    N, T1, Lq, C = query.shape
    _, T2, S, _ = input_flatten.shape
    M, L, P = n_heads, n_levels, n_points
    hw = torch.as_tensor(shapes, dtype=torch.long)
    value = F.linear(input_flatten, value_w, value_b)                   # :114
    if padding_mask is not None:
        value = value.masked_fill(padding_mask, 0.0)                    # :116
    value = value.view(N, T2, S, M, C // M)
    normalizer = torch.stack([hw[:, 1], hw[:, 0]], -1).to(query.dtype)  # :126-127  (W,H)
    outs, locs, wts = [], [], []
    for t1 in range(T1):
        nb = temporal_neighbours(t1, n_frame, T2)
        q = query[:, t1]
        logits = torch.stack([F.linear(q, att_w[t2], att_b[t2]).view(N, Lq, M, L, P) for t2 in nb], -1)
        att = F.softmax(logits.flatten(-3), -1).view(N, Lq, M, L, P, len(nb))   # :147-150
        acc = 0
        loc_t1 = []
        for k, t2 in enumerate(nb):
            off = F.linear(q, off_w[t2], off_b[t2]).view(N, Lq, M, L, P, 2)
            off = off / normalizer[None, None, None, :, None, :]                # :164
            loc = reference_points[:, t1, :, None, :, None, :] + off             # :165
            loc_t1.append(loc)
            acc = acc + core(value[:, t2], hw, loc, att[..., k])                 # :172-181,225
        outs.append(acc)
        locs.append(torch.stack(loc_t1, dim=-2).detach())                       # :229-231
        wts.append(att.detach())                                                # :233
    out = F.linear(torch.stack(outs, dim=1), out_w, out_b)                      # :236-237
    return out, locs, wts
