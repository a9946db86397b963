"""One native call -- and ONE autograd node -- per encoder layer and direction (include/snipper_layers.h).

reference: DeformableTransformerEncoderLayer.forward, models/deformable_transformer.py:200-216, with MSDeformAttn.forward,
models/ops/modules/ms_deform_attn.py:99-243, inside it.  ``DeformableTransformerEncoderLayer.forward_fused`` runs a layer under
bf16 autocast as 9 launches forward and ~25 backward through eight autograd nodes (value / merged offset-logit / output projections,
prologue, tied sampler, two fused LayerNorms, feed-forward block), each with its own module dispatch, shadow look-ups, allocations
and ctypes call: ~1.3 ms of host time per layer and step.  ``snipper_encoder_layer_forward`` / ``_backward`` issue the same
launches with the same arguments in the same order; this module allocates, looks the weight shadows up, packs one argument block
per direction and keeps the autograd contract of ``forward_fused`` (the lazy float32 LayerNorm results included).  Bit-identical
to the per-module path (tests/test_encoder_native_gpu.py).
"""
from __future__ import annotations

import ctypes
import struct
import weakref
from ctypes import c_float

import torch

from . import _lib
from ._autograd import Function

_DIMS_FMT = "12i6f3Q"
_W_FMT = "20P"
_FWD_FMT = "@" + _DIMS_FMT + _W_FMT + "P" + "5P" + "3P" + "P" + "3P" + "3P" + "3P" + "P" + "2P" + "P" + "N"
_BWD_FMT = "@" + _DIMS_FMT + _W_FMT + "14P" + "P" + "3P" + "2P" + "3P" + "3P" + "3P" + "P" + "P" + "N" + "3P"
DIMS_BYTES = struct.calcsize("@" + _DIMS_FMT)


class EncPlan:
    __slots__ = ("dims", "ps", "R", "nlp", "arena_bytes", "scratch_bytes", "cfg", "cfg_p", "hs", "hs_p", "inv_w", "inv_h", "inv_w_p",
                 "inv_h_p", "mix", "mix_t", "mix_p", "mix_t_p", "hm")


_PLANS: dict = {}


def make_plan(layer, bs, frames, S, hw, last: bool):
    """EncPlan for this layer and geometry, or None when the composites (or the head-major owner-computes backward they are built
    around) do not take it."""
    from . import fused
    from .ms_deform_attn import frame_neighbours
    att = layer.self_attn
    training = layer.training
    base_cfg = _lib.active_config()
    key = (id(layer), bs, frames, S, tuple(hw), last, training, layer.dropout1.p, layer.dropout2.p, layer.dropout3.p,
           None if base_cfg is None else bytes(base_cfg), fused._HEAD_MAJOR)
    plan = _PLANS.get(key)
    if plan is not None:
        return plan if plan is not False else None

    def refuse():
        if len(_PLANS) > 64:
            _PLANS.clear()
        _PLANS[key] = False
        return None

    C, M, L, P = att.d_model, att.n_heads, att.n_levels, att.n_points
    if not (att.weights_are_tied() and att.value_bf16 and not att.use_pytroch_deform and att.fused_elementwise and
            not att.attention_vis and C // M == 48 and P == 4 and L <= 4 and frames <= 4 and att.n_frame == frames and
            len(hw) == L and (base_cfg is None or base_cfg.policy == 0) and layer.linear2.in_features == layer.linear1.out_features):
        return refuse()
    hm = bool(fused._HEAD_MAJOR)
    cfg = fused._head_major_config() if hm else (base_cfg if base_cfg is not None else None)
    if hm and not fused._owner_backward_available(cfg, hw, bs * frames, S, M, C // M, L, P):
        return refuse()                               # (forward_fused then keeps the reference layout: leave it that path)
    ps = tuple(float(p) if training else 0.0 for p in (layer.dropout1.p, layer.dropout2.p, layer.dropout3.p))
    dims = (DIMS_BYTES, bs, frames, S, C, M, layer.linear1.out_features, L, P, 1 if last else 0, 1 if hm else 0, 0) + ps + \
           (float(layer.norm1.eps), float(layer.norm2.eps), 0.0)
    lib = _lib.load()
    probe = struct.pack("@" + _DIMS_FMT, *(dims + (0, 0, 0)))
    if not lib.snipper_encoder_layer_supported(probe):
        return refuse()
    plan = EncPlan()
    plan.dims, plan.ps, plan.hm = dims, ps, hm
    plan.R, plan.nlp = bs * frames * S, M * L * P
    plan.cfg = cfg
    plan.cfg_p = ctypes.addressof(cfg) if cfg is not None else 0
    plan.hs = (ctypes.c_int64 * (2 * L))(*[int(v) for p_ in hw for v in p_])
    plan.hs_p = ctypes.addressof(plan.hs)
    plan.inv_w = (c_float * L)(*[1.0 / w for h, w in hw])
    plan.inv_h = (c_float * L)(*[1.0 / h for h, w in hw])
    plan.inv_w_p, plan.inv_h_p = ctypes.addressof(plan.inv_w), ctypes.addressof(plan.inv_h)
    groups = [frame_neighbours(t1, att.n_frame, frames) for t1 in range(frames)]
    mix = [[(1.0 / len(g)) if t2 in g else 0.0 for t2 in range(frames)] for g in groups]
    plan.mix = (c_float * (frames * frames))(*[w for row in mix for w in row])
    plan.mix_t = (c_float * (frames * frames))(*[mix[a][b] for b in range(frames) for a in range(frames)])
    plan.mix_p, plan.mix_t_p = ctypes.addressof(plan.mix), ctypes.addressof(plan.mix_t)
    plan.arena_bytes = int(lib.snipper_encoder_layer_arena_bytes(probe))
    plan.scratch_bytes = int(lib.snipper_encoder_layer_scratch_bytes(probe, plan.cfg_p or None, plan.hs_p))
    if plan.arena_bytes <= 0 or plan.scratch_bytes <= 0:
        return refuse()
    if len(_PLANS) > 64:
        _PLANS.clear()
    _PLANS[key] = plan
    return plan


def layer_params(layer):
    """The 16 parameters in the order EncoderLayerFn takes them."""
    att = layer.self_attn
    so, aw = att.sampling_offsets[0], att.attention_weights[0]
    return (att.value_proj.weight, att.value_proj.bias, so.weight, so.bias, aw.weight, aw.bias, att.output_proj.weight,
            att.output_proj.bias, layer.norm1.weight, layer.norm1.bias, layer.linear1.weight, layer.linear1.bias,
            layer.linear2.weight, layer.linear2.bias, layer.norm2.weight, layer.norm2.bias)


def lookup_shadows(layer):
    """This step's bf16 weight shadows, transposes and packs the composites read (shadow.py), as the 20-pointer weight block's
    tensors -- or None when any of them is missing or stale (the per-module path then converts on the spot)."""
    from . import shadow
    att = layer.self_attn
    so, aw = att.sampling_offsets[0], att.attention_weights[0]
    wv, wo, w1, w2 = att.value_proj.weight, att.output_proj.weight, layer.linear1.weight, layer.linear2.weight
    m = shadow.lookup_merged(so, aw, second_bias_only=True)
    if m is None:
        return None
    wm16, bm = m
    wm_t = shadow.lookup_merged_t(so, aw)
    wv16, wo16, w116, w216 = shadow.lookup(wv), shadow.lookup(wo), shadow.lookup(w1), shadow.lookup(w2)
    wv_t, wo_t, w2_t = shadow.lookup_t(wv), shadow.lookup_t(wo), shadow.lookup_t(w2)
    pk1, pk2 = shadow.lookup_lpacked(w1), shadow.lookup_lpacked(w2)
    if (wm_t is None or wv16 is None or wo16 is None or w116 is None or w216 is None or wv_t is None or wo_t is None or w2_t is None or
            pk1 is None or pk2 is None or pk1[1] is None or pk2[0] is None or not shadow.LINEAR_WIDE):
        return None
    b16 = (wv16, wm16, wo16, w116, pk2[0], wv_t, wm_t, wo_t, w2_t, pk1[1])
    f32 = (att.value_proj.bias, bm, so.bias, att.output_proj.bias, layer.linear1.bias, layer.linear2.bias, layer.norm1.weight,
           layer.norm1.bias, layer.norm2.weight, layer.norm2.bias)
    if not all(t.dtype == torch.bfloat16 and t.is_contiguous() and t.data_ptr() % 16 == 0 for t in b16):
        return None
    if not all(t.dtype == torch.float32 and t.is_contiguous() and t.data_ptr() % 16 == 0 for t in f32):
        return None
    return b16 + f32


class EncoderLayerFn(Function):
    """apply(plan, shadows, x32, src16, q16, pos16 | None, ref, shapes, lsi, *16 parameters) -> (out0, y16, yq16 | None)
    out0 = norm2's float32 result (last layer) or its saved pre-norm sum, to be tagged lazy by the caller."""

    @staticmethod
    def forward(ctx, plan, shadows, x32, src16, q16, pos16, ref, shapes, lsi, *params):
        from .fused import _next_seed
        dev = x32.device
        R, C = plan.R, x32.shape[-1]
        last = bool(plan.dims[9])
        seeds = tuple(_next_seed() if p > 0 else 0 for p in plan.ps)            # (norm1, feed-forward, norm2: forward_fused's order)
        lazy = getattr(x32, "_lazy_ln", None)                                   # (stats [2, R], gamma32, beta32) of x32's producer
        f32, b16 = torch.float32, torch.bfloat16
        s2 = torch.empty(x32.shape, dtype=f32, device=dev)
        stats2 = torch.empty((2, R), dtype=f32, device=dev)
        y32 = torch.empty(x32.shape, dtype=f32, device=dev) if last else None
        y16 = torch.empty(x32.shape, dtype=b16, device=dev)
        yq16 = None if last else torch.empty(x32.shape, dtype=b16, device=dev)
        arena = torch.empty(plan.arena_bytes, dtype=torch.uint8, device=dev)
        ptr = lambda t: t.data_ptr() if t is not None else 0
        blob = struct.pack(
            _FWD_FMT, *plan.dims, *seeds, *[t.data_ptr() for t in shadows], plan.cfg_p,
            x32.data_ptr(), lazy[0][0].data_ptr() if lazy else 0, lazy[0][1].data_ptr() if lazy else 0, ptr(lazy[1]) if lazy else 0,
            ptr(lazy[2]) if lazy else 0,
            src16.data_ptr(), q16.data_ptr(), ptr(pos16), ref.data_ptr(), shapes.data_ptr(), lsi.data_ptr(), plan.hs_p,
            plan.inv_w_p, plan.inv_h_p, plan.mix_p, s2.data_ptr(), stats2[0].data_ptr(), stats2[1].data_ptr(), ptr(y32),
            y16.data_ptr(), ptr(yq16), arena.data_ptr(), plan.arena_bytes)
        with _lib.device_guard(dev):
            rc = _lib.load().snipper_encoder_layer_forward(_lib.raw_stream(dev), blob)
        _lib.check(rc, "snipper_encoder_layer_forward")
        _lib.note_variant()
        ctx.plan, ctx.seeds, ctx.shadows, ctx.n_params = plan, seeds, shadows, len(params)
        ctx.prefs = params
        ctx.pos_needs = pos16 is not None
        ctx.set_materialize_grads(False)               # (an unused output's gradient arrives as None, not as a zero-filled tensor)
        ctx.save_for_backward(src16, q16, s2, stats2, arena, shapes, lsi)
        ctx.stats2 = stats2
        return (y32 if last else s2), y16, yq16

    @staticmethod
    def backward(ctx, g0, g16, gq):
        from .dense import _grad_out
        plan = ctx.plan
        src16, q16, s2, stats2, arena, shapes, lsi = ctx.saved_tensors
        n_in = 9 + ctx.n_params
        if g0 is None and g16 is None and gq is None:
            return (None,) * n_in
        dev = s2.device
        R, C = plan.R, s2.shape[-1]
        f32, b16 = torch.float32, torch.bfloat16
        g0 = g0.contiguous().float() if g0 is not None else None
        g16 = g16.contiguous().to(b16) if g16 is not None else None
        gq = gq.contiguous().to(b16) if gq is not None else None
        wv, bv, so_w, so_b, aw_w, aw_b, wo, bo, n1w, n1b, w1, b1, w2, b2, n2w, n2b = ctx.prefs
        nlp, d_ffn = plan.nlp, plan.dims[6]

        def out_for(p, shape):
            v = _grad_out(p, shape)
            return v if v is not None else torch.empty(shape, dtype=f32, device=dev)

        dWv, dbv = out_for(wv, (C, C)), out_for(bv, (C,))
        dWm = torch.empty((3 * nlp, C), dtype=f32, device=dev)
        dbm = torch.empty((3 * nlp,), dtype=f32, device=dev)
        dWo, dbo = out_for(wo, (C, C)), out_for(bo, (C,))
        dW1, db1 = out_for(w1, (d_ffn, C)), out_for(b1, (d_ffn,))
        dW2, db2 = out_for(w2, (C, d_ffn)), out_for(b2, (C,))
        dn1 = torch.empty((2, C), dtype=f32, device=dev)
        dn2 = torch.empty((2, C), dtype=f32, device=dev)
        d_x32 = torch.empty(s2.shape, dtype=f32, device=dev)
        d_src16 = torch.empty(s2.shape, dtype=b16, device=dev)
        d_q16 = torch.empty(s2.shape, dtype=b16, device=dev)
        scratch = torch.empty(plan.scratch_bytes, dtype=torch.uint8, device=dev)
        ptr = lambda t: t.data_ptr() if t is not None else 0
        blob = struct.pack(
            _BWD_FMT, *plan.dims, *ctx.seeds, *[t.data_ptr() for t in ctx.shadows],
            dWv.data_ptr(), dbv.data_ptr(), dWm.data_ptr(), dbm.data_ptr(), dWo.data_ptr(), dbo.data_ptr(), dW1.data_ptr(), db1.data_ptr(),
            dW2.data_ptr(), db2.data_ptr(), dn1[0].data_ptr(), dn1[1].data_ptr(), dn2[0].data_ptr(), dn2[1].data_ptr(),
            plan.cfg_p, ptr(g0), ptr(g16), ptr(gq), src16.data_ptr(), q16.data_ptr(), s2.data_ptr(), stats2[0].data_ptr(),
            stats2[1].data_ptr(), shapes.data_ptr(), lsi.data_ptr(), plan.hs_p, plan.inv_w_p, plan.inv_h_p, plan.mix_t_p,
            arena.data_ptr(), scratch.data_ptr(), plan.scratch_bytes, d_x32.data_ptr(), d_src16.data_ptr(), d_q16.data_ptr())
        with _lib.device_guard(dev):
            rc = _lib.load().snipper_encoder_layer_backward(_lib.raw_stream(dev), blob)
        _lib.check(rc, "snipper_encoder_layer_backward")
        _lib.note_variant()
        na = 2 * nlp
        d_pos = gq if (ctx.pos_needs and ctx.needs_input_grad[5]) else None
        return (None, None, d_x32, d_src16, d_q16, d_pos, None, None, None,
                dWv, dbv, dWm[:na], dbm[:na], dWm[na:], dbm[na:], dWo, dbo, dn1[0], dn1[1], dW1, db1, dW2, db2, dn2[0], dn2[1])


def layer_forward(layer, plan, shadows, src32, src16, q16, pos16, ref, shapes, lsi, last: bool):
    """``DeformableTransformerEncoderLayer.forward_fused`` through EncoderLayerFn: (src32 or its lazy stand-in, src16, q16)."""
    from . import fused
    ref = ref.contiguous()
    out0, y16, yq16 = EncoderLayerFn.apply(plan, shadows, src32, src16.contiguous(), q16.contiguous(),
                                           None if last else pos16.contiguous(), ref, shapes, lsi, *layer_params(layer))
    if not last:
        # the float32 result is lazy: the saved pre-norm sum tagged with norm2's statistics and affine parameters (fused.
        # AddDropoutLayerNorm's protocol: the next layer's first LayerNorm recomputes the result on load)
        out0._lazy_ln = (out0.grad_fn.stats2, layer.norm2.weight, layer.norm2.bias)
        p = out0.data_ptr()
        fused._LAZY_PTRS.add(p)
        weakref.finalize(out0, fused._LAZY_PTRS.discard, p)
    return out0, y16, yq16


def timing_off() -> bool:
    """The per-launch event brackets of bench.py's roofline records (MSDA / dense ``enable_launch_timing``) live in the per-module
    wrappers: while they are on, the layers run through those wrappers (the same launches)."""
    from . import MultiScaleDeformableAttention as MSDA
    from . import dense
    return MSDA._timing is None and dense._timing is None
