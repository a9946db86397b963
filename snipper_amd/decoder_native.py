"""One native call -- and ONE autograd node -- per decoder layer and direction (include/snipper_layers.h).

reference: DeformableTransformerDecoderLayer.forward, models/deformable_transformer.py:276-300, plus the reference-point
refinement of DeformableTransformerDecoder.forward (:329-333).  ``DeformableTransformerDecoderLayer.forward_chain`` runs the layer
as ~14 launches forward and ~12 backward through eight autograd nodes, each with its own allocations, checks and ctypes call:
0.5-0.6 ms of Python per layer and direction, 6-7 ms of the 18.5 ms the host needs to issue a training step.  Here the host does
what only it can do -- allocate the outputs and one arena, pack one argument block -- and ``snipper_decoder_layer_forward`` /
``_backward`` issue the SAME launches with the same arguments in the same order (bit-identical results:
tests/test_decoder_native_gpu.py).  The cross attention's value projection (a full-size product over the encoder memory, on the
bf16 kernels with their per-step weight shadows) stays outside: its result is an input of the node, its gradient an output.
"""
from __future__ import annotations

import ctypes
import struct
from ctypes import c_float, c_int32, c_size_t, c_uint64, c_void_p

import torch

from . import _lib
from ._autograd import Function


class Dims(ctypes.Structure):
    """snipper_decoder_layer_dims"""
    _fields_ = ([(n, c_int32) for n in ("struct_bytes", "bs", "tokens", "frames", "queries", "C", "heads", "d_ffn", "levels", "points",
                                        "S", "value_dtype")] +
                [(n, c_float) for n in ("p_attn", "p_norm2", "p_norm1", "p_ffn", "p_norm3", "eps_norm2", "eps_norm1", "eps_norm3",
                                        "attn_scale")] +
                [(n, c_uint64) for n in ("seed_attn", "seed_norm2", "seed_norm1", "seed_ffn", "seed_norm3")])


PARAM_FIELDS = ("so_w", "so_b", "aw_w", "aw_b", "op_w", "op_b", "norm1_w", "norm1_b", "in_proj_w", "in_proj_b", "out_proj_w",
                "out_proj_b", "norm2_w", "norm2_b", "lin1_w", "lin1_b", "lin2_w", "lin2_b", "norm3_w", "norm3_b")


class Params(ctypes.Structure):
    """snipper_decoder_layer_params"""
    _fields_ = [(n, c_void_p) for n in PARAM_FIELDS]


class Fwd(ctypes.Structure):
    """snipper_decoder_layer_fwd"""
    _fields_ = ([("d", Dims), ("w", Params)] +
                [(n, c_void_p) for n in ("x_v", "x_res", "x_q", "pos_a", "pos_b", "value", "shapes", "level_start", "host_shapes",
                                         "ref_in", "inv_w", "inv_h", "root_w", "root_b", "ref_points", "valid_ratios", "new_ref",
                                         "ref_in_next", "out", "out_q", "loc", "prob", "arena")] +
                [("arena_bytes", c_size_t)])


class Bwd(ctypes.Structure):
    """snipper_decoder_layer_bwd"""
    _fields_ = ([("d", Dims), ("w", Params), ("dw", Params), ("g", c_void_p * 4)] +
                [(n, c_void_p) for n in ("x_v", "x_q", "value", "shapes", "level_start", "host_shapes", "inv_w", "inv_h", "loc", "prob",
                                         "arena", "scratch")] +
                [("scratch_bytes", c_size_t)] +
                [(n, c_void_p) for n in ("d_xv", "d_xres", "d_xq", "d_pos_a", "d_ref_in", "d_value")])


# the same layouts for struct.pack (filling ~70 ctypes fields one by one costs more host time than the call itself)
_DIMS_FMT = "12i9f5Q"
_FWD_FMT = "@" + _DIMS_FMT + "20P" + "23P" + "N"
_BWD_FMT = "@" + _DIMS_FMT + "20P" + "20P" + "4P" + "12P" + "N" + "6P"
assert struct.calcsize("@" + _DIMS_FMT) == ctypes.sizeof(Dims), (struct.calcsize("@" + _DIMS_FMT), ctypes.sizeof(Dims))
assert struct.calcsize(_FWD_FMT) == ctypes.sizeof(Fwd), (struct.calcsize(_FWD_FMT), ctypes.sizeof(Fwd))
assert struct.calcsize(_BWD_FMT) == ctypes.sizeof(Bwd), (struct.calcsize(_BWD_FMT), ctypes.sizeof(Bwd))

LAYERS_ABI_VERSION = 1
_checked = False


def _lib_typed():
    """The loaded library (typed by _lib.EXPORTS), with the composites' ABI version checked once."""
    global _checked
    lib = _lib.load()
    if not _checked:
        if lib.snipper_layers_abi_version() != LAYERS_ABI_VERSION:
            raise _lib.SnipperLibraryError("snipper_layers ABI mismatch")
        _checked = True
    return lib


class LayerPlan:
    """What is constant for a (layer, shape, mode): the dims without seeds, host arrays, the arena / scratch sizes."""
    __slots__ = ("dims", "arena_bytes", "scratch_bytes", "inv_w", "inv_h", "inv_w_p", "inv_h_p", "hs", "hs_p", "ps", "R", "nlp",
                 "groups")

    def pack_dims(self, seeds):
        return self.dims + tuple(seeds)


def layer_params(layer):
    """The layer's 20 parameter tensors in PARAM_FIELDS order (tied offset / weight Linears: entry 0 of each list)."""
    ca, mha = layer.cross_attn, layer.self_attn
    return (ca.sampling_offsets[0].weight, ca.sampling_offsets[0].bias, ca.attention_weights[0].weight, ca.attention_weights[0].bias,
            ca.output_proj.weight, ca.output_proj.bias, layer.norm1.weight, layer.norm1.bias, mha.in_proj_weight, mha.in_proj_bias,
            mha.out_proj.weight, mha.out_proj.bias, layer.norm2.weight, layer.norm2.bias, layer.linear1.weight, layer.linear1.bias,
            layer.linear2.weight, layer.linear2.bias, layer.norm3.weight, layer.norm3.bias)


_PLANS: dict = {}


def make_plan(layer, bs, frames, queries, C, S, hw, value_dtype, value_frames=None):
    """LayerPlan for this layer and geometry, or None when the native composites do not take the shape.  ``frames`` = query
    frames (T + F for the forecast model), ``value_frames`` = frames of the memory (T; default: ``frames``)."""
    training = layer.training
    value_frames = frames if value_frames is None else value_frames
    key = (id(layer), bs, frames, value_frames, queries, C, S, tuple(hw), value_dtype, training,
           layer.dropout1.p, layer.dropout2.p, layer.dropout3.p, layer.dropout4.p, layer.self_attn.dropout)
    plan = _PLANS.get(key)
    if plan is not None:
        return plan if plan is not False else None
    ca, mha = layer.cross_attn, layer.self_attn
    H = mha.num_heads
    ps = tuple(float(p) if training else 0.0 for p in (mha.dropout, layer.dropout2.p, layer.dropout1.p, layer.dropout3.p, layer.dropout4.p))
    hd = C // H
    dims = (ctypes.sizeof(Dims), bs, frames * queries, frames, queries, C, H, layer.linear1.out_features, ca.n_levels, ca.n_points, S,
            1 if value_dtype == torch.bfloat16 else 0) + ps + (float(layer.norm2.eps), float(layer.norm1.eps), float(layer.norm3.eps),
                                                                float(hd ** -0.5))
    plan = LayerPlan()
    plan.dims, plan.ps = dims, ps
    plan.R, plan.nlp = bs * frames * queries, H * ca.n_levels * ca.n_points
    lib = _lib_typed()
    probe = struct.pack("@" + _DIMS_FMT, *(dims + (0, 0, 0, 0, 0)))
    ok = (ca.n_heads == H and layer.linear2.in_features == layer.linear1.out_features and len(hw) == ca.n_levels and
          bool(lib.snipper_decoder_layer_supported(probe)))
    if not ok:
        if len(_PLANS) > 64:
            _PLANS.clear()
        _PLANS[key] = False
        return None
    plan.arena_bytes = int(lib.snipper_decoder_layer_arena_bytes(probe))
    plan.scratch_bytes = int(lib.snipper_decoder_layer_scratch_bytes(probe))
    L = len(hw)
    plan.inv_w = (c_float * L)(*[1.0 / w for h, w in hw])
    plan.inv_h = (c_float * L)(*[1.0 / h for h, w in hw])
    plan.hs = (ctypes.c_int64 * (2 * L))(*[int(v) for p_ in hw for v in p_])
    plan.inv_w_p, plan.inv_h_p, plan.hs_p = (ctypes.addressof(plan.inv_w), ctypes.addressof(plan.inv_h), ctypes.addressof(plan.hs))
    from .ms_deform_attn import frame_neighbours
    plan.groups = [frame_neighbours(t1, ca.n_frame, value_frames) for t1 in range(frames)]
    if len(_PLANS) > 64:
        _PLANS.clear()
    _PLANS[key] = plan
    return plan


class DecoderLayerFn(Function):
    """apply(plan, want_q, x_v, x_res, x_q, pos_a, pos_b, value, ref_in, ref_points, valid_ratios, shapes, lsi, root_w, root_b,
             *20 parameters)
       -> (n_v, n_res, out, n_q | None, new_ref | None, ref_in_next | None, loc, prob)
    n_v / n_res / out are three aliases of norm3's result (the next layer's value input and residual, and the decoder's
    per-layer output), n_q = result + pos_b.  Every differentiable input gets its gradient from ONE backward call."""

    @staticmethod
    def forward(ctx, plan, want_q, x_v, x_res, x_q, pos_a, pos_b, value, ref_in, ref_points, valid_ratios, shapes, lsi, root_w,
                root_b, *params):
        from .fused import _next_seed
        dev = x_v.device
        R, C = plan.R, x_v.shape[-1]
        ps = plan.ps
        seeds = tuple(_next_seed() if p > 0 else 0 for p in ps)      # (attention, norm2, norm1, feed-forward, norm3: the chain's order)
        f32 = torch.float32
        out = torch.empty(x_v.shape, dtype=f32, device=dev)
        out_q = torch.empty(x_v.shape, dtype=f32, device=dev) if want_q else None
        M, L, P = plan.dims[6], plan.dims[8], plan.dims[9]
        loc = torch.empty((R * M, L, P, 2), dtype=f32, device=dev)
        prob = torch.empty((R * M, L, P), dtype=f32, device=dev)
        arena = torch.empty(plan.arena_bytes, dtype=torch.uint8, device=dev)
        refine = root_w is not None
        new_ref = torch.empty((R, 2), dtype=f32, device=dev) if refine else None
        ref_next = torch.empty((R, L, 2), dtype=f32, device=dev) if refine else None
        ptr = lambda t: t.data_ptr() if t is not None else 0
        blob = struct.pack(
            _FWD_FMT, *plan.dims, *seeds, *[p.data_ptr() for p in params],
            x_v.data_ptr(), x_res.data_ptr(), x_q.data_ptr(), pos_a.data_ptr(), ptr(pos_b) if want_q else 0, value.data_ptr(),
            shapes.data_ptr(), lsi.data_ptr(), plan.hs_p, ref_in.data_ptr(), plan.inv_w_p, plan.inv_h_p, ptr(root_w), ptr(root_b),
            ptr(ref_points) if refine else 0, ptr(valid_ratios) if refine else 0, ptr(new_ref), ptr(ref_next), out.data_ptr(),
            ptr(out_q), loc.data_ptr(), prob.data_ptr(), arena.data_ptr(), plan.arena_bytes)
        with _lib.device_guard(dev):
            rc = _lib_typed().snipper_decoder_layer_forward(_lib.raw_stream(dev), blob)
        _lib.check(rc, "snipper_decoder_layer_forward")
        _lib.note_variant()
        ctx.plan, ctx.seeds, ctx.want_q, ctx.ref_shape = plan, seeds, want_q, ref_in.shape
        # (an output nobody differentiates -- the refined references, loc / prob, the last layer's unused aliases -- must reach the
        #  backward as None, not as a freshly zero-filled tensor: 26 fill launches per step otherwise)
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(x_v, x_q, value, shapes, lsi, loc, prob, arena, *params)
        outs = (out.view_as(out), out.view_as(out), out, out_q, new_ref, ref_next, loc, prob)
        ctx.mark_non_differentiable(*[t for t in (new_ref, ref_next, loc, prob) if t is not None])
        return outs

    @staticmethod
    def backward(ctx, g_v, g_res, g_out, g_q, *_unused):
        plan = ctx.plan
        x_v, x_q, value, shapes, lsi, loc, prob, arena = ctx.saved_tensors[:8]
        params = ctx.saved_tensors[8:]
        dev = x_v.device
        R, C = plan.R, x_v.shape[-1]
        live = [g.contiguous() if g.dtype == torch.float32 else g.float().contiguous() for g in (g_v, g_res, g_out, g_q) if g is not None]
        if not live:
            return (None,) * (15 + len(params))
        gp = [g.data_ptr() for g in live] + [0] * (4 - len(live))
        f32 = torch.float32
        d_xv = torch.empty(x_v.shape, dtype=f32, device=dev)
        d_xres = torch.empty(x_v.shape, dtype=f32, device=dev)
        d_xq = torch.empty(x_v.shape, dtype=f32, device=dev)
        d_pos_a = torch.empty(x_v.shape, dtype=f32, device=dev)
        need_ref = ctx.needs_input_grad[8]
        d_ref = torch.empty(ctx.ref_shape, dtype=f32, device=dev) if need_ref else None
        d_value = torch.empty(value.shape, dtype=value.dtype, device=dev)
        scratch = torch.empty(plan.scratch_bytes, dtype=torch.uint8, device=dev)
        grads = [torch.empty_like(p) for p in params]
        blob = struct.pack(
            _BWD_FMT, *plan.dims, *ctx.seeds, *[p.data_ptr() for p in params], *[g.data_ptr() for g in grads], *gp,
            x_v.data_ptr(), x_q.data_ptr(), value.data_ptr(), shapes.data_ptr(), lsi.data_ptr(), plan.hs_p, plan.inv_w_p, plan.inv_h_p,
            loc.data_ptr(), prob.data_ptr(), arena.data_ptr(), scratch.data_ptr(), plan.scratch_bytes,
            d_xv.data_ptr(), d_xres.data_ptr(), d_xq.data_ptr(), d_pos_a.data_ptr(), d_ref.data_ptr() if d_ref is not None else 0,
            d_value.data_ptr())
        with _lib.device_guard(dev):
            rc = _lib_typed().snipper_decoder_layer_backward(_lib.raw_stream(dev), blob)
        _lib.check(rc, "snipper_decoder_layer_backward")
        _lib.note_variant()
        d_pos_b = g_q if ctx.want_q else None
        return (None, None, d_xv, d_xres, d_xq, d_pos_a, d_pos_b, d_value, d_ref, None, None, None, None, None, None, *grads)
