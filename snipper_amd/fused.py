"""Autograd wrappers of the fused element-wise kernels around the core op
(include/snipper_dense.h, csrc/msda_prologue.cuh).  CUDA tensors, float32 / bfloat16 only;
``MSDeformAttn`` falls back to the plain PyTorch formulation for anything else (e.g. float64)."""
from __future__ import annotations

import ctypes
from typing import List, Optional, Sequence

import torch
from ._autograd import Function
from torch.autograd.function import once_differentiable

from . import _lib

_DT = {torch.float32: 0, torch.bfloat16: 1}


def _stream(dev):
    return _lib.raw_stream(dev)


def _farray(vals: Sequence[float]):
    return (ctypes.c_float * len(vals))(*[float(v) for v in vals])


def supported(t: torch.Tensor) -> bool:
    return t.is_cuda and t.dtype in _DT


def _mix_launch(x, mask, mask_on_input, mix_rows: List[List[float]], out_dtype, head_dim: int = 0,
                in_head_major: bool = False, out_head_major: bool = False):
    """``x`` [N, Ti, S, C] -> [N, To, S, C]; with ``in_head_major`` / ``out_head_major`` that side's MEMORY is
    [N, frames, C // head_dim, S, head_dim] (snipper_msda_config.value_layout = 1) under the same logical shape."""
    N, Ti, S, C = x.shape
    To = len(mix_rows)
    out = torch.empty((N, To, S, C), dtype=out_dtype, device=x.device)
    flat = _farray([w for row in mix_rows for w in row])
    with _lib.device_guard(x.device):
        rc = _lib.load().snipper_temporal_mix_ex(
            _stream(x.device), x.data_ptr(), _DT[x.dtype], mask.data_ptr() if mask is not None else None,
            int(mask_on_input), ctypes.cast(flat, ctypes.c_void_p), N, Ti, To, S, C, out.data_ptr(), _DT[out_dtype],
            int(head_dim), int(in_head_major), int(out_head_major))
    _lib.check(rc, "snipper_temporal_mix_ex")
    return out


def _head_major_config():
    """The caller's (test) Config, or the defaults, with value_layout = 1."""
    base = _lib.active_config()
    cfg = _lib.Config.defaults()
    if base is not None:
        ctypes.memmove(ctypes.byref(cfg), ctypes.byref(base), ctypes.sizeof(cfg))
    cfg.value_layout = 1
    return cfg


def _owner_backward_available(cfg, host_shapes, n, S, M, D, L, P) -> bool:
    """Does snipper_msda_backward_ex have an owner-computes plan (the only bf16-value backward that knows the head-major
    layout) for an encoder-shape call (Lq == S) of this geometry?  A host-side query, no launch."""
    key = (bytes(cfg), tuple(tuple(int(v) for v in hw) for hw in host_shapes), n, S, M, D, L, P)
    ok = _OWNER_OK.get(key)
    if ok is None:
        hs = (ctypes.c_int64 * (2 * L))(*[v for hw in key[1] for v in hw])
        ok = _lib.load().snipper_msda_backward_ex_workspace_bytes(
            ctypes.byref(cfg), ctypes.cast(hs, ctypes.c_void_p), 1, n, S, M, D, L, S, P) > 0
        if len(_OWNER_OK) < 64:
            _OWNER_OK[key] = ok
    return ok


_OWNER_OK = {}


import os as _os
# the encoder's bf16 temporal mean in the head-major layout (0 = the reference layout: A/B runs)
_HEAD_MAJOR = _os.environ.get("SNIPPER_VALUE_HEAD_MAJOR", "1") != "0"


class TemporalMix(Function):
    """out[n, t1] = sum_t2 mix[t1][t2] * (mask[n, t2] ? 0 : value[n, t2]), float32 out.

    value [N, T2, S, C] (f32 / bf16, contiguous), mask [N, T2, S] uint8 / bool or None,
    mix: host list [T1][T2]; ``out_dtype`` float32 (default) or bfloat16.  Backward is the transposed mix with the mask
    applied to its output."""

    @staticmethod
    def forward(ctx, value, mask, mix, out_dtype=torch.float32):
        value = value.contiguous()
        m8 = None
        if mask is not None:
            m8 = mask.contiguous().view(torch.uint8) if mask.dtype == torch.bool else mask.contiguous()
        ctx.mix, ctx.in_dtype = mix, value.dtype
        ctx.save_for_backward(m8) if m8 is not None else None
        ctx.has_mask = m8 is not None
        return _mix_launch(value, m8, True, mix, out_dtype)

    @staticmethod
    def backward(ctx, grad_out):
        m8 = ctx.saved_tensors[0] if ctx.has_mask else None
        mix = ctx.mix
        mix_t = [[mix[a][b] for a in range(len(mix))] for b in range(len(mix[0]))]
        go = grad_out.contiguous()
        if go.dtype not in _DT:
            go = go.float()
        g = _mix_launch(go, m8, False, mix_t, ctx.in_dtype)
        return g, None, None, None


class TiedSampler(Function):
    """The tied module core as ONE autograd node: core_op(temporal mean of the neighbouring, padding-masked value frames,
    loc, prob)  (reference models/ops/modules/ms_deform_attn.py:116, 130-226 with the per-frame Linears tied, DESIGN.md
    section 4).  Fusing TemporalMix with the core op keeps the mean -- ``vbar`` -- internal to the node, which is what allows
    it to be held in bfloat16 under bf16 autocast (``vbar_bf16``: the projected value is bf16 already; the tap rows of the
    gathers, which are bound by the bytes the texture path moves, halve: forward 0.28 -> 0.17 ms, backward 1.15 -> 0.97 ms per
    encoder launch at N = 8) while its float32 gradient goes straight into the transposed mix without a cast round trip.

    value [N, T2, S, C] (f32 / bf16), mask [N, T2, S] or None, mix: host list [T1][T2], loc [N*T1, Lq, M, L, P, 2] and
    prob [N*T1, Lq, M, L, P] float32  ->  [N*T1, Lq, C] (bf16 rows when ``rows_bf16`` or ``vbar_bf16``)."""

    @staticmethod
    def forward(ctx, value, mask, mix, loc, prob, shapes, lsi, M, im2col_step, rows_bf16, vbar_bf16):
        from . import MultiScaleDeformableAttention as MSDA
        value = value.contiguous()
        N, T2, S, C = value.shape
        m8 = None
        if mask is not None:
            m8 = mask.contiguous().view(torch.uint8) if mask.dtype == torch.bool else mask.contiguous()
        host_shapes = getattr(shapes, "_snipper_host", None)
        # The encoder's bf16 mean is internal to this node -- its producer (the mix kernel) and its consumers (the core
        # op's kernels) are this library's -- so it is kept HEAD-MAJOR, [n, head, position, 48]: the x-neighbour taps of a
        # sample are then one contiguous 192-byte run (forward 158 -> 146 us, query side of the backward 227 -> 207 us at
        # freshly initialised offsets, 296 -> 229 us at sigma = 3 px: profiles/r04_head_major_experiment.txt).
        D = C // M
        # (only where BOTH directions have a kernel for it: the tuned forward and the owner-computes backward -- D = 48, P = 4,
        #  L <= 4, Lq == S, default policy; anything else keeps the reference layout)
        base_cfg = _lib.active_config()
        hm = bool(_HEAD_MAJOR and vbar_bf16 and host_shapes is not None and D == 48 and loc.shape[-2] == 4 and
                  loc.shape[-3] <= 4 and loc.shape[1] == S and len(mix) <= 4 and T2 <= 4 and
                  (base_cfg is None or base_cfg.policy == 0))
        cfg = _head_major_config() if hm else None
        if hm and not _owner_backward_available(cfg, host_shapes, N * len(mix), S, M, D, loc.shape[-3], loc.shape[-2]):
            # the library has no owner-computes plan for this geometry (maps too large for the marks' bounds, or 2^32
            # samples): its backward would have to run kernels that only know the reference layout -- keep that layout
            hm, cfg = False, None
        vbar = _mix_launch(value, m8, True, mix, torch.bfloat16 if vbar_bf16 else torch.float32, D, False, hm)
        v4 = vbar.view(N * len(mix), S, M, D)             # (logical shape; the memory is head-major when `hm`)
        out = MSDA.ms_deform_attn_forward(v4, shapes, lsi, loc, prob, im2col_step,
                                          out_bf16=bool(rows_bf16) and not vbar_bf16, host_shapes=host_shapes, config=cfg)
        ctx.mix, ctx.in_dtype, ctx.dims, ctx.host_shapes, ctx.step = mix, value.dtype, (N, len(mix), S, C), host_shapes, im2col_step
        ctx.hm, ctx.head_dim = hm, D
        ctx.has_mask = m8 is not None
        ctx.save_for_backward(v4, shapes, lsi, loc, prob, *([m8] if m8 is not None else []))
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out):
        from . import MultiScaleDeformableAttention as MSDA
        v4, shapes, lsi, loc, prob = ctx.saved_tensors[:5]
        m8 = ctx.saved_tensors[5] if ctx.has_mask else None
        N, T1, S, C = ctx.dims
        go = grad_out.contiguous()
        if v4.dtype == torch.bfloat16 and go.dtype != torch.bfloat16:
            go = go.to(torch.bfloat16)
        elif v4.dtype == torch.float32 and go.dtype not in (torch.float32, torch.bfloat16):
            go = go.float()
        gv, gl, ga = MSDA.ms_deform_attn_backward(v4, shapes, lsi, loc, prob, go, ctx.step, host_shapes=ctx.host_shapes,
                                                  grad_value_f32=True, config=_head_major_config() if ctx.hm else None)
        mix = ctx.mix
        mix_t = [[mix[a][b] for a in range(len(mix))] for b in range(len(mix[0]))]
        g_value = _mix_launch(gv.view(N, T1, S, C), m8, False, mix_t, ctx.in_dtype, ctx.head_dim, ctx.hm, False)
        return g_value, None, None, gl, ga, None, None, None, None, None, None


def _query_rows(t: torch.Tensor, width: int):
    """``t`` [..., width] as (tensor, leading dimension in elements): a view whose rows are evenly strided and whose
    last dim is contiguous is taken as is (e.g. a column slice of a merged projection); anything else is copied."""
    t2 = t.reshape(-1, width) if t.dim() != 2 else t
    if t2.stride(1) != 1 or (t2.shape[0] > 1 and t2.stride(0) < width):
        t2 = t2.contiguous()
    return t2, (t2.stride(0) if t2.shape[0] > 1 else width)


class MSDAPrologue(Function):
    """(raw offsets, raw logits, reference points) -> (sampling locations, attention probabilities).

    off [..., M*L*P*2], logit [..., M*L*P] (f32 / bf16), ref [..., L, 2] float32 (leading dims = those of off), hw:
    host list of (H, W).  Returns loc [rows, L, P, 2] and prob [rows, L, P] as float32 with rows = prod(leading dims)
    * M.  With ``logit=None``, ``off`` is the output [..., M*L*P*3] of ONE merged projection (offset columns first,
    then logits): the kernels address both halves in place and the backward returns one dense gradient for it."""

    @staticmethod
    def forward(ctx, off, logit, ref, hw, M, L, P, off_bias=None):
        """``off_bias`` [M*L*P*2] float32 (no gradient through here) or None: added to the offsets inside the kernel, in
        float32 -- ``off`` then is W q alone (csrc/msda_prologue.cuh, snipper_msda_prologue_forward_ex)."""
        if logit is None:                       # merged input: [..., M*L*P*2 | M*L*P] in one tensor
            both, ld = _query_rows(off, M * L * P * 3)
            off2, logit2, off_ld, logit_ld = both[:, :M * L * P * 2], both[:, M * L * P * 2:], ld, ld
            off_shape, logit_shape = off.shape, None
        else:
            off_shape, logit_shape = off.shape, logit.shape
            off2, off_ld = _query_rows(off, M * L * P * 2)
            logit2, logit_ld = _query_rows(logit, M * L * P)
        ref = ref.contiguous().float()
        rows = off2.shape[0] * M
        loc = torch.empty((rows, L, P, 2), dtype=torch.float32, device=off.device)
        prob = torch.empty((rows, L, P), dtype=torch.float32, device=off.device)
        inv_w, inv_h = _farray([1.0 / w for h, w in hw]), _farray([1.0 / h for h, w in hw])
        ob = None
        if off_bias is not None:
            ob = off_bias.detach()
            ob = ob if (ob.dtype == torch.float32 and ob.is_contiguous()) else ob.float().contiguous()
            assert ob.numel() == M * L * P * 2 and ob.device == off.device
        with _lib.device_guard(off.device):
            rc = _lib.load().snipper_msda_prologue_forward_ex(
                _stream(off.device), off2.data_ptr(), off_ld, logit2.data_ptr(), logit_ld, _DT[off.dtype],
                ob.data_ptr() if ob is not None else None,
                ref.data_ptr(), ctypes.cast(inv_w, ctypes.c_void_p), ctypes.cast(inv_h, ctypes.c_void_p), rows, M, L, P,
                loc.data_ptr(), prob.data_ptr())
        _lib.check(rc, "snipper_msda_prologue_forward_ex")
        ctx.save_for_backward(prob)
        merged = logit is None
        ctx.meta = (hw, M, L, P, off.dtype, off_shape, logit_shape, ref.shape, ctx.needs_input_grad[2], merged)
        return loc, prob

    @staticmethod
    def backward(ctx, grad_loc, grad_prob):
        (prob,) = ctx.saved_tensors
        hw, M, L, P, dtype, off_shape, logit_shape, ref_shape, need_ref, merged = ctx.meta
        rows = prob.shape[0]
        nq, w_off, w_logit = rows // M, M * L * P * 2, M * L * P
        grad_loc = grad_loc.contiguous().float() if grad_loc is not None else torch.zeros(rows, L, P, 2, device=prob.device)
        grad_prob = grad_prob.contiguous().float() if grad_prob is not None else torch.zeros_like(prob)
        if merged:
            buf = torch.empty((nq, w_off + w_logit), dtype=dtype, device=prob.device)
            g_off, g_logit, ld_o, ld_l = buf[:, :w_off], buf[:, w_off:], w_off + w_logit, w_off + w_logit
        else:
            g_off = torch.empty((nq, w_off), dtype=dtype, device=prob.device)
            g_logit = torch.empty((nq, w_logit), dtype=dtype, device=prob.device)
            ld_o, ld_l = w_off, w_logit
        g_ref = torch.empty(ref_shape, dtype=torch.float32, device=prob.device) if need_ref else None
        inv_w, inv_h = _farray([1.0 / w for h, w in hw]), _farray([1.0 / h for h, w in hw])
        with _lib.device_guard(prob.device):
            rc = _lib.load().snipper_msda_prologue_backward(
                _stream(prob.device), grad_loc.data_ptr(), grad_prob.data_ptr(), prob.data_ptr(),
                ctypes.cast(inv_w, ctypes.c_void_p), ctypes.cast(inv_h, ctypes.c_void_p), rows, M, L, P,
                g_off.data_ptr(), ld_o, g_logit.data_ptr(), ld_l, _DT[dtype], g_ref.data_ptr() if g_ref is not None else None)
        _lib.check(rc, "snipper_msda_prologue_backward")
        if merged:
            return buf.view(off_shape), None, g_ref, None, None, None, None, None
        return g_off.view(off_shape), g_logit.view(logit_shape), g_ref, None, None, None, None, None


_dropout_calls = 0


def _next_seed() -> int:
    """A fresh 64-bit seed per call: torch's seed in the high half, a process-wide call counter in the low half
    (reproducible after ``torch.manual_seed`` for a fixed order of calls)."""
    global _dropout_calls
    if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
        # the seed is a kernel scalar: a captured graph would replay the SAME dropout mask every step
        raise RuntimeError("fused dropout (p > 0) cannot be captured into a hipGraph: its seed is a host-side scalar; "
                           "run the step eagerly or with dropout 0")
    _dropout_calls += 1
    return ((torch.initial_seed() & 0xFFFFFFFF) << 32) | (_dropout_calls & 0xFFFFFFFF)


class AddDropoutLayerNorm(Function):
    """y = LayerNorm(x + dropout_p(z)) in one kernel each way (csrc/ln_fused.cuh).

    x [..., C] float32 / bf16, z like x (float32 / bf16) or None, pos like x or None, gamma / beta [C] float32.
    ``want`` = (y32, y16, yq16) flags; returns those three (None where not wanted): y in float32, y in bf16 and
    bf16(y + pos).  ``seed`` None draws a fresh one.

    LAZY float32 output (``want[0] == "lazy"``): y32 is not written; the first result is the saved pre-norm sum, tagged
    (``_lazy_ln``) with this call's statistics and affine parameters, and stands for y32 ONLY as the ``x`` of the next
    ``add_dropout_layer_norm``, whose kernel recomputes this LayerNorm's output on load (csrc/ln_fused.cuh).  The autograd
    edge is the ordinary one: the next call's dx is this call's g32."""

    @staticmethod
    def forward(ctx, x, z, pos, gamma, beta, p, eps, want, seed, lazy_x=None):
        x = x.contiguous()
        z = z.contiguous() if z is not None else None
        pos = pos.contiguous() if pos is not None else None
        C = x.shape[-1]
        rows = x.numel() // C
        dev = x.device
        p = float(p) if z is not None else 0.0
        need_bwd = any(ctx.needs_input_grad[:5])
        s_save = torch.empty((rows, C), dtype=torch.float32, device=dev) if need_bwd else None
        stats = torch.empty((2, rows), dtype=torch.float32, device=dev) if need_bwd else None
        keep = torch.empty((rows, C // 4), dtype=torch.uint8, device=dev) if (need_bwd and p > 0) else None
        lazy_out = want[0] == "lazy"
        if lazy_out:
            assert need_bwd, "a lazy float32 output is the saved pre-norm sum: it exists only when a backward will run"
        y32 = torch.empty(x.shape, dtype=torch.float32, device=dev) if (want[0] and not lazy_out) else None
        y16 = torch.empty(x.shape, dtype=torch.bfloat16, device=dev) if want[1] else None
        yq = torch.empty(x.shape, dtype=torch.bfloat16, device=dev) if want[2] else None
        g32, b32 = gamma.float(), beta.float()
        ptr = lambda t: t.data_ptr() if t is not None else None
        lx = lazy_x if lazy_x is not None else (None, None, None)          # (stats [2, rows], gamma32, beta32) of x's producer
        if lazy_x is not None:
            assert x.dtype == torch.float32 and lx[0].shape == (2, rows)
        with _lib.device_guard(dev):
            rc = _lib.load().snipper_add_dropout_layernorm_forward_ex(
                _stream(dev), x.data_ptr(), _DT[x.dtype],
                lx[0][0].data_ptr() if lazy_x is not None else None, lx[0][1].data_ptr() if lazy_x is not None else None,
                ptr(lx[1]), ptr(lx[2]), ptr(z), _DT[z.dtype] if z is not None else 0,
                ptr(pos), _DT[pos.dtype] if pos is not None else 0, g32.data_ptr(), b32.data_ptr(), rows, C,
                p, float(eps), int(seed if seed is not None else (_next_seed() if p > 0 else 0)), ptr(s_save),
                stats[0].data_ptr() if stats is not None else None, stats[1].data_ptr() if stats is not None else None,
                ptr(keep), ptr(y32), ptr(y16), ptr(yq))
        _lib.check(rc, "snipper_add_dropout_layernorm_forward_ex")
        ctx.meta = (p, x.dtype, None if z is None else z.dtype, None if pos is None else pos.dtype, x.shape,
                    gamma.dtype, beta.dtype)
        ctx.save_for_backward(s_save, stats, keep, g32)
        if lazy_out:
            ctx.lazy_aux = (stats, g32, b32)         # add_dropout_layer_norm tags the result with it
            return s_save.view(x.shape), y16, yq
        return y32, y16, yq

    @staticmethod
    def backward(ctx, g32, g16, gq):
        s_save, stats, keep, gamma = ctx.saved_tensors
        p, x_dt, z_dt, pos_dt, shape, gamma_dt, beta_dt = ctx.meta
        rows, C = s_save.shape
        dev = s_save.device
        if g32 is None and g16 is None and gq is None:
            return (None,) * 10
        g32 = g32.contiguous().float() if g32 is not None else None
        g16 = g16.contiguous().to(torch.bfloat16) if g16 is not None else None
        gq = gq.contiguous().to(torch.bfloat16) if gq is not None else None
        dx = torch.empty(shape, dtype=x_dt, device=dev) if ctx.needs_input_grad[0] else None
        dz = torch.empty(shape, dtype=z_dt, device=dev) if (z_dt is not None and ctx.needs_input_grad[1]) else None
        dgb = torch.empty((2, C), dtype=torch.float32, device=dev)
        lib = _lib.load()
        nbytes = lib.snipper_add_dropout_layernorm_workspace_bytes(rows, C)
        ws = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
        ptr = lambda t: t.data_ptr() if t is not None else None
        with _lib.device_guard(dev):
            rc = lib.snipper_add_dropout_layernorm_backward(
                _stream(dev), ptr(g32), ptr(g16), ptr(gq), s_save.data_ptr(), stats[0].data_ptr(), stats[1].data_ptr(),
                gamma.data_ptr(), ptr(keep), rows, C, p, ptr(dx), _DT[x_dt] if dx is not None else 0,
                ptr(dz), _DT[z_dt] if dz is not None else 0, dgb[0].data_ptr(), dgb[1].data_ptr(), ws.data_ptr(), nbytes)
        _lib.check(rc, "snipper_add_dropout_layernorm_backward")
        dpos = gq if (pos_dt is not None and ctx.needs_input_grad[2]) else None
        dgamma = dgb[0].to(gamma_dt) if ctx.needs_input_grad[3] else None
        dbeta = dgb[1].to(beta_dt) if ctx.needs_input_grad[4] else None
        return dx, dz, dpos, dgamma, dbeta, None, None, None, None, None


def add_dropout_layer_norm(x, z, norm: torch.nn.LayerNorm, p: float, training: bool, pos=None,
                           want=(True, False, False), seed=None):
    """``norm(x + dropout(z, p, training))`` on the fused kernel -> (y32, y16, yq16) per ``want``.  The caller checks
    ``ln_fusable`` first.  ``want[0] == "lazy"``: the float32 result is not materialised (see AddDropoutLayerNorm); it is
    honoured only when a backward will run (otherwise there is no saved sum to stand for it) and falls back to True."""
    want = tuple(want)
    if want[0] == "lazy" and not (torch.is_grad_enabled() and (x.requires_grad or (z is not None and z.requires_grad) or
                                                               norm.weight.requires_grad)):
        want = (True,) + want[1:]
    lazy_x = getattr(x, "_lazy_ln", None)
    if lazy_x is None and x.is_cuda and x.data_ptr() in _LAZY_PTRS:
        # a view / detach / hook product of a lazy result: same memory (the UN-normalised pre-norm sum), tag lost
        raise RuntimeError("add_dropout_layer_norm: `x` aliases a lazy LayerNorm result but does not carry its statistics "
                           "(a lazy float32 output may only be passed on unchanged; see AddDropoutLayerNorm)")
    out = AddDropoutLayerNorm.apply(x, z, pos, norm.weight, norm.bias, p if training else 0.0, norm.eps, want, seed, lazy_x)
    if want[0] == "lazy":
        out[0]._lazy_ln = out[0].grad_fn.lazy_aux
        ptr = out[0].data_ptr()
        _LAZY_PTRS.add(ptr)
        weakref.finalize(out[0], _LAZY_PTRS.discard, ptr)      # (the tagged tensor object gone: nobody can pass it on)
    return out


import weakref
_LAZY_PTRS = set()        # data pointers of live lazy results (an untagged alias must not be read as y32)


def ln_fusable(x: torch.Tensor, norm: torch.nn.LayerNorm) -> bool:
    C = x.shape[-1]
    return (x.is_cuda and x.dtype in _DT and isinstance(norm, torch.nn.LayerNorm) and norm.elementwise_affine and
            norm.bias is not None and tuple(norm.normalized_shape) == (C,) and C % 4 == 0 and C <= 1024 and
            x.numel() < 2 ** 32)


class InputProjTokens(Function):
    """All input projections of the model (1x1 convolution + GroupNorm per backbone level, reference
    models/model.py:62-84) followed by the flatten + concatenation of models/deformable_transformer.py:103-122, written
    directly as token rows:

        feats[l] [n, Cin_l, h_l, w_l] bf16, channels_last   (n = b * T frames)
        -> src32 [b, T, S, C] float32, src16 (bf16 copy), q16 = bf16(src + pos)       S = sum_l h_l * w_l

    The convolution is ONE GEMM per level on the NHWC rows (dense.linear_bf16, bias fused), GroupNorm runs on those
    rows and writes the level's slice of the three outputs (csrc/gn_tokens.cuh): no layout copy, concatenation, cast
    or add kernel.  ``pos`` [b, T, S, C] (bf16 or float32) only feeds q16; ``want`` = (src16, q16) flags.
    apply(T, groups, eps, pos, want, *feats, *(weight, bias, gamma, beta per level))"""

    @staticmethod
    def forward(ctx, T, groups, eps, pos, want, *tensors):
        from .dense import linear_bf16
        L = len(tensors) // 5
        feats, params = tensors[:L], tensors[L:]
        n, dev = feats[0].shape[0], feats[0].device
        C = params[0].shape[0]
        hws = [(int(f.shape[2]), int(f.shape[3])) for f in feats]
        S = sum(h * w for h, w in hws)
        b = n // T
        src32 = torch.empty((b, T, S, C), dtype=torch.float32, device=dev)
        src16 = torch.empty((b, T, S, C), dtype=torch.bfloat16, device=dev) if want[0] else None
        q16 = torch.empty((b, T, S, C), dtype=torch.bfloat16, device=dev) if (want[1] and pos is not None) else None
        pos = pos.contiguous() if pos is not None else None
        lib = _lib.load()
        ptr = lambda t: t.data_ptr() if t is not None else None
        saved, off = [], 0
        for l, f in enumerate(feats):
            w, bias, gamma, beta = params[4 * l: 4 * l + 4]
            h, wd = hws[l]
            x2 = f.permute(0, 2, 3, 1).reshape(n * h * wd, f.shape[1])          # NHWC rows (a view)
            from . import shadow
            wb = shadow.lookup(w)
            wb = wb.view(C, -1) if wb is not None else w.reshape(C, -1).to(torch.bfloat16)
            y = linear_bf16(x2, wb, bias.float())                               # [n*hw, C] bf16
            stats = torch.empty((n, groups, 2), dtype=torch.float32, device=dev)
            nbytes = lib.snipper_groupnorm_tokens_workspace_bytes(n, h * wd, C, groups)
            ws = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
            g32, b32 = gamma.float(), beta.float()
            with _lib.device_guard(dev):
                rc = lib.snipper_groupnorm_tokens_forward(
                    _stream(dev), y.data_ptr(), g32.data_ptr(), b32.data_ptr(), n, h * wd, C, groups, float(eps),
                    S, off, ptr(pos) if q16 is not None else None, _DT[pos.dtype] if pos is not None else 0,
                    src32.data_ptr(), ptr(src16), ptr(q16), stats.data_ptr(), ws.data_ptr(), nbytes)
            _lib.check(rc, "snipper_groupnorm_tokens_forward")
            saved += [x2, wb, y, stats, g32]
            off += h * wd
        ctx.save_for_backward(*saved)
        ctx.meta = (L, T, groups, hws, S, C, [tuple(f.shape) for f in feats], [p.shape for p in params],
                    [p.dtype for p in params], pos is not None and q16 is not None)
        return src32, src16, q16

    @staticmethod
    def backward(ctx, g32, g16, gq):
        from .dense import linear_bf16, wgrad_bf16
        L, T, groups, hws, S, C, fshapes, pshapes, pdtypes, has_q = ctx.meta
        saved = ctx.saved_tensors
        dev = saved[0].device
        g32 = g32.contiguous().float() if g32 is not None else None
        g16 = g16.contiguous().to(torch.bfloat16) if g16 is not None else None
        gq = gq.contiguous().to(torch.bfloat16) if gq is not None else None
        lib = _lib.load()
        ptr = lambda t: t.data_ptr() if t is not None else None
        dfeats, dparams, off = [], [], 0
        for l in range(L):
            x2, wb, y, stats, gamma = saved[5 * l: 5 * l + 5]
            h, wd = hws[l]
            n = fshapes[l][0]
            dy = torch.empty_like(y)
            dgb = torch.empty((2, C), dtype=torch.float32, device=dev)
            nbytes = lib.snipper_groupnorm_tokens_workspace_bytes(n, h * wd, C, groups)
            ws = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
            with _lib.device_guard(dev):
                rc = lib.snipper_groupnorm_tokens_backward(
                    _stream(dev), y.data_ptr(), gamma.data_ptr(), stats.data_ptr(), ptr(g32), ptr(g16), ptr(gq),
                    n, h * wd, C, groups, S, off, dy.data_ptr(), dgb[0].data_ptr(), dgb[1].data_ptr(),
                    ws.data_ptr(), nbytes)
            _lib.check(rc, "snipper_groupnorm_tokens_backward")
            dfeat = None
            if ctx.needs_input_grad[5 + l]:
                from .dense import _dgrad
                dx2 = _dgrad(dy, wb)                                            # [n*hw, Cin]
                dfeat = dx2.view(n, h, wd, -1).permute(0, 3, 1, 2)              # logical NCHW, NHWC memory
            dW, db = wgrad_bf16(dy, x2, want_bias=True)
            shapes, dts = pshapes[4 * l: 4 * l + 4], pdtypes[4 * l: 4 * l + 4]
            dparams += [dW.view(shapes[0]).to(dts[0]), db.to(dts[1]), dgb[0].to(dts[2]), dgb[1].to(dts[3])]
            dfeats.append(dfeat)
            off += h * wd
        dpos = gq if (has_q and ctx.needs_input_grad[3]) else None
        return (None, None, None, dpos, None, *dfeats, *dparams)


class LevelPosTokens(Function):
    """pos16[b, t, S, C] = bf16(cat_l(pos_tokens[l] + level_embed[l])): the position encoding of every level plus its
    learned level embedding (reference models/deformable_transformer.py:118-121), written once in the dtype the
    encoder's kernels read.  Backward: level_embed's gradient is the column sum of each level's slice of the incoming
    bf16 gradient (csrc/gn_tokens.cuh colsum kernels) -- no float32 copy of the [b, t, S, C] gradient, no cat /
    slice / broadcast-reduction chain.  The sine encoding itself carries no gradient.

    ``n_uses`` > 1 returns that many ALIASES of the one buffer, one per consumer (the input projections' query and the
    query of every encoder layer but the last): each consumer's gradient then comes back on its own edge and the column
    sums are taken over all of them in one launch per level (the sum of column sums is the column sum of the sum) --
    autograd would otherwise add the n gradients of [b, t, S, C] pairwise first, n - 1 launches of 3 x 60 MB each.
    apply(level_embed [L, C], n_uses, *pos_tokens (L x [b, t, hw_l, C] float32))"""

    @staticmethod
    def forward(ctx, level_embed, n_uses, *pos_tokens):
        b, t, _, C = pos_tokens[0].shape
        sizes = [int(p.shape[2]) for p in pos_tokens]
        S = sum(sizes)
        out = torch.empty((b, t, S, C), dtype=torch.bfloat16, device=level_embed.device)
        if (level_embed.is_cuda and level_embed.dtype == torch.float32 and level_embed.is_contiguous() and C % 8 == 0 and
                len(pos_tokens) <= 4 and all(p.dtype == torch.float32 and p.is_contiguous() and p.shape[:2] == (b, t) and
                                             p.shape[3] == C for p in pos_tokens)):
            # one launch for all levels (csrc/misc_kernels.cuh, level_pos_kernel)
            nl = len(pos_tokens)
            ptrs = (ctypes.c_void_p * nl)(*[p.data_ptr() for p in pos_tokens])
            hws = (ctypes.c_int * nl)(*sizes)
            with _lib.device_guard(out.device):
                rc = _lib.load().snipper_level_pos_bf16(_stream(out.device), ptrs, hws, nl, level_embed.data_ptr(), b * t, C,
                                                        out.data_ptr())
            _lib.check(rc, "snipper_level_pos_bf16")
        else:
            off = 0
            for l, p in enumerate(pos_tokens):
                torch.add(p, level_embed[l].view(1, 1, 1, C), out=out[:, :, off:off + sizes[l]])
                off += sizes[l]
        ctx.sizes, ctx.shape, ctx.le_dtype = sizes, (b, t, S, C), level_embed.dtype
        ctx.n_levels = level_embed.shape[0]
        if n_uses <= 1:
            return out
        return tuple(out.view(b, t, S, C) for _ in range(n_uses))

    @staticmethod
    def backward(ctx, *gs):
        b, t, S, C = ctx.shape
        gs = [g for g in gs if g is not None]
        gs = [(g if g.dtype == torch.bfloat16 else g.to(torch.bfloat16)).contiguous() for g in gs]
        lib = _lib.load()
        dev = gs[0].device
        d = torch.zeros((ctx.n_levels, C), dtype=torch.float32, device=dev)
        for lo in range(0, len(gs), 8):
            part = gs[lo:lo + 8]
            srcs = (ctypes.c_void_p * len(part))(*[g.data_ptr() for g in part])
            dl = d if lo == 0 else torch.zeros_like(d)
            off = 0
            for l, hw in enumerate(ctx.sizes):
                nbytes = lib.snipper_colsum_workspace_bytes(b * t, hw, C) * len(part)
                ws = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
                with _lib.device_guard(dev):
                    rc = lib.snipper_colsum_segments_multi_bf16(_stream(dev), srcs, len(part), off * C, S * C, b * t, hw, C,
                                                                dl[l].data_ptr(), ws.data_ptr(), nbytes)
                _lib.check(rc, "snipper_colsum_segments_multi_bf16")
                off += hw
            if lo:
                d += dl
        return (d.to(ctx.le_dtype), None) + (None,) * len(ctx.sizes)


class FanOut(Function):
    """``n`` aliases of a tensor, one per consumer; backward = ONE pass that sums the consumers' gradients with float32
    accumulation (csrc/misc_kernels.cuh, csrc/small_ln.cuh) instead of autograd's n - 1 pairwise adds.  bf16: the encoder
    memory's twin feeds the value projection of every decoder layer (reference models/deformable_transformer.py:290-295): six
    [b, T, S, C] gradients per step.  float32 (round 6): ``query_pos`` is added in front of two projections per decoder layer
    (:252-254): twelve [b, T * queries, C] gradients per step."""

    @staticmethod
    def forward(ctx, x, n):
        ctx.n = n
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, *gs):
        gs = [g for g in gs if g is not None]
        if len(gs) == 1:
            return gs[0], None
        same = all(g.is_cuda and g.dtype == gs[0].dtype and g.shape == gs[0].shape for g in gs)
        if same and gs[0].dtype == torch.bfloat16 and gs[0].numel() % 8 == 0 and len(gs) <= 8:
            gs = [g.contiguous() for g in gs]
            out = torch.empty_like(gs[0])
            srcs = (ctypes.c_void_p * len(gs))(*[g.data_ptr() for g in gs])
            with _lib.device_guard(out.device):
                rc = _lib.load().snipper_sum_bf16(_stream(out.device), srcs, len(gs), out.data_ptr(), out.numel())
            _lib.check(rc, "snipper_sum_bf16")
            return out, None
        if same and gs[0].dtype == torch.float32 and gs[0].numel() % 4 == 0 and len(gs) <= 16:
            gs = [g.contiguous() for g in gs]
            if all(g.data_ptr() % 16 == 0 for g in gs):
                out = torch.empty_like(gs[0])
                srcs = (ctypes.c_void_p * len(gs))(*[g.data_ptr() for g in gs])
                with _lib.device_guard(out.device):
                    rc = _lib.load().snipper_sum_f32(_stream(out.device), srcs, len(gs), out.data_ptr(), out.numel())
                _lib.check(rc, "snipper_sum_f32")
                return out, None
        tot = gs[0]
        for g in gs[1:]:
            tot = tot + g
        return tot, None


class SmallLayerNorm(Function):
    """``LayerNorm(x + dropout_p(z))`` for decoder-size float32 rows as one launch each way (csrc/small_ln.cuh; reference
    models/deformable_transformer.py:266-300), with what keeps the decoder's chain free of element-wise launches:

    * ``n_alias`` aliases of the result, one per consumer, and -- with ``pos`` -- ``yq = y + pos`` (the reference's
      ``with_pos_embed``, :252-254) from the same launch;
    * the backward takes every consumer's gradient on its own edge and sums them in registers (up to four), and writes
      dgamma / dbeta itself (no second kernel).

    apply(x, z, pos, gamma, beta, p, eps, n_alias, want_q) -> (alias_1, ..., alias_n[, yq])"""

    @staticmethod
    def forward(ctx, x, z, pos, gamma, beta, p, eps, n_alias, want_q):
        x = x.contiguous()
        z = z.contiguous() if z is not None else None
        pos = pos.contiguous() if (pos is not None and want_q) else None
        C = x.shape[-1]
        rows = x.numel() // C
        dev = x.device
        p = float(p) if z is not None else 0.0
        need_bwd = any(ctx.needs_input_grad[:5])
        s_save = torch.empty((rows, C), dtype=torch.float32, device=dev) if need_bwd else None
        stats = torch.empty((2, rows), dtype=torch.float32, device=dev) if need_bwd else None
        keep = torch.empty((rows, C // 4), dtype=torch.uint8, device=dev) if (need_bwd and p > 0) else None
        y = torch.empty(x.shape, dtype=torch.float32, device=dev)
        yq = torch.empty(x.shape, dtype=torch.float32, device=dev) if pos is not None else None
        ptr = lambda t: t.data_ptr() if t is not None else None
        with _lib.device_guard(dev):
            rc = _lib.load().snipper_small_ln_forward_f32(
                _stream(dev), x.data_ptr(), ptr(z), ptr(pos), gamma.data_ptr(), beta.data_ptr(), rows, C, p, float(eps),
                _next_seed() if p > 0 else 0, ptr(s_save), stats[0].data_ptr() if stats is not None else None,
                stats[1].data_ptr() if stats is not None else None, ptr(keep), y.data_ptr(), ptr(yq))
        _lib.check(rc, "snipper_small_ln_forward_f32")
        ctx.meta = (p, x.shape, z is not None, pos is not None, n_alias)
        ctx.save_for_backward(s_save, stats, keep, gamma)
        ctx.set_materialize_grads(False)               # (an unused alias costs nothing: its gradient arrives as None)
        outs = tuple(y.view_as(y) for _ in range(n_alias))
        return outs + ((yq,) if yq is not None else ())

    @staticmethod
    def backward(ctx, *gs):
        s_save, stats, keep, gamma = ctx.saved_tensors
        p, shape, has_z, has_pos, n_alias = ctx.meta
        rows, C = s_save.shape
        dev = s_save.device
        gq = gs[n_alias] if has_pos else None
        live = [g.contiguous() for g in gs if g is not None]
        if not live:
            return (None,) * 9
        if any(g.dtype != torch.float32 for g in live):
            live = [g.float() for g in live]
        while len(live) > 4:                        # (more consumers than the kernel has sources: fold the extras first)
            live = live[:3] + [sum(live[3:])]
        dx = torch.empty(shape, dtype=torch.float32, device=dev) if ctx.needs_input_grad[0] else None
        dz = torch.empty(shape, dtype=torch.float32, device=dev) if (has_z and ctx.needs_input_grad[1]) else None
        dgb = torch.empty((2, C), dtype=torch.float32, device=dev)
        g4 = [g.data_ptr() for g in live] + [None] * (4 - len(live))
        ptr = lambda t: t.data_ptr() if t is not None else None
        with _lib.device_guard(dev):
            rc = _lib.load().snipper_small_ln_backward_f32(
                _stream(dev), g4[0], g4[1], g4[2], g4[3], s_save.data_ptr(), stats[0].data_ptr(), stats[1].data_ptr(),
                gamma.data_ptr(), ptr(keep), rows, C, p, ptr(dx), ptr(dz), dgb[0].data_ptr(), dgb[1].data_ptr())
        _lib.check(rc, "snipper_small_ln_backward_f32")
        dpos = gq if (has_pos and ctx.needs_input_grad[2]) else None
        return (dx, dz, dpos, dgb[0] if ctx.needs_input_grad[3] else None, dgb[1] if ctx.needs_input_grad[4] else None,
                None, None, None, None)


SMALL_LN_MAX_ROWS = 16384      # csrc/small_ln.cuh: kSmallLnMaxRows


def small_ln_ok(x: torch.Tensor, norm: torch.nn.LayerNorm) -> bool:
    """Float32 CUDA rows outside autocast, few enough for the decoder-size kernels, an affine float32 LayerNorm over C."""
    C = x.shape[-1]
    return (x.is_cuda and x.dtype == torch.float32 and not torch.is_autocast_enabled('cuda') and
            isinstance(norm, torch.nn.LayerNorm) and norm.elementwise_affine and norm.bias is not None and
            norm.weight.dtype == torch.float32 and norm.weight.is_contiguous() and norm.bias.is_contiguous() and
            tuple(norm.normalized_shape) == (C,) and C % 4 == 0 and C <= 1024 and 0 < x.numel() // C <= SMALL_LN_MAX_ROWS)


def small_layer_norm(x, z, norm: torch.nn.LayerNorm, p: float, training: bool, pos=None, n_alias: int = 1):
    """``norm(x + dropout(z, p, training))`` on the decoder-size kernel -> (alias_1, ..., alias_n[, y + pos]).  The caller checks
    ``small_ln_ok`` first."""
    return SmallLayerNorm.apply(x, z, pos, norm.weight, norm.bias, p if training else 0.0, norm.eps, n_alias, pos is not None)
