"""Spatiotemporal deformable transformer (host side, PyTorch) over the gfx950 MSDeformAttn.

API mirror of /root/reference/models/deformable_transformer.py: the classes
``DeformableTransformer``, ``DeformableTransformerEncoderLayer`` / ``Encoder``,
``DeformableTransformerDecoderLayer`` / ``Decoder`` and the builder
``build_deforamble_transformer`` (sic) keep the reference's constructor arguments, forward
signatures, return values and state_dict keys (checked against a state_dict saved from the
reference: tests/test_transformer.py).  The implementation is this repository's own; the hot
call -- ``MSDeformAttn`` -- runs one HIP launch per module forward (ms_deform_attn.py).
"""
from __future__ import annotations

import copy
from typing import List, Optional

import torch
from ._autograd import Function as _Fn
import torch.nn.functional as F
from torch import nn

from .dense import big_ffn, big_linear
from .misc import is_no_padding
from .ms_deform_attn import MSDeformAttn


def inverse_sigmoid(x, eps=1e-5):
    """logit with clamping (reference util/misc.py:481-485)."""
    x = x.clamp(min=0, max=1)
    return torch.log(x.clamp(min=eps) / (1 - x).clamp(min=eps))


def _refine_reference(head, out, reference_points, valid_ratios):
    """(sigmoid(head(out)[..., :2] + inverse_sigmoid(ref)).detach(), that times the valid ratios) from one launch of
    csrc/match_cost.cuh's refine kernel after the head's own GEMMs -- the result carries no gradient (the reference
    detaches it, :333), so the head runs without a graph here.  None when the float32 CUDA layout does not apply."""
    if not (out.is_cuda and out.dtype == torch.float32 and reference_points.dtype == torch.float32 and
            reference_points.dim() == 4 and reference_points.shape[-1] == 2 and valid_ratios.dtype == torch.float32 and
            not torch.is_autocast_enabled('cuda')):
        return None
    from . import _lib
    with torch.no_grad():
        bs, t, lq = reference_points.shape[:3]
        lin = _single_linear(head)
        L = valid_ratios.shape[1]
        if (lin is not None and lin.out_features >= 2 and lin.weight.dtype == torch.float32 and lin.weight.is_contiguous() and
                out.is_contiguous() and out.shape[-1] % 4 == 0 and out.numel() // out.shape[-1] == bs * t * lq and
                out.data_ptr() % 16 == 0 and lin.weight.data_ptr() % 16 == 0 and
                (lin.bias is None or lin.bias.dtype == torch.float32)):
            # the head is ONE Linear (the reference's MLP(hidden, hidden, 4, 1), models/model.py:95) and only its first two
            # outputs are used here: two dot products per row inside the refinement launch (csrc/small_ln.cuh)
            ref = reference_points.contiguous()
            vr = valid_ratios.contiguous()
            new_ref = torch.empty_like(ref)
            ref_in = torch.empty((bs, t, lq, L, 2), dtype=torch.float32, device=out.device)
            with _lib.device_guard(out.device):
                rc = _lib.load().snipper_refine_reference_linear_f32(
                    _lib.raw_stream(out.device), out.data_ptr(), lin.weight.data_ptr(),
                    lin.bias.data_ptr() if lin.bias is not None else None, ref.data_ptr(), vr.data_ptr(), bs * t * lq,
                    out.shape[-1], t * lq, L, 1e-5, new_ref.data_ptr(), ref_in.data_ptr())
            _lib.check(rc, "snipper_refine_reference_linear_f32")
            return new_ref, ref_in
        tmp = head(out)
        if tmp.dtype != torch.float32 or tmp.shape[:3] != (bs, t, lq) or tmp.stride(-1) != 1 or not tmp.is_contiguous():
            return None
        ref = reference_points.contiguous()
        vr = valid_ratios.contiguous()
        new_ref = torch.empty_like(ref)
        ref_in = torch.empty((bs, t, lq, L, 2), dtype=torch.float32, device=out.device)
        with _lib.device_guard(out.device):
            rc = _lib.load().snipper_refine_reference_f32(_lib.raw_stream(out.device), tmp.data_ptr(), tmp.shape[-1],
                                                          ref.data_ptr(), vr.data_ptr(), bs * t * lq, t * lq, L, 1e-5,
                                                          new_ref.data_ptr(), ref_in.data_ptr())
        _lib.check(rc, "snipper_refine_reference_f32")
    return new_ref, ref_in


def _single_linear(head):
    """The nn.Linear a head consists of (an nn.Linear, or an MLP-like module whose ``layers`` hold exactly one), else None."""
    if isinstance(head, nn.Linear):
        return head
    layers = getattr(head, "layers", None)
    if layers is not None and len(layers) == 1 and isinstance(layers[0], nn.Linear):
        return layers[0]
    return None


from .misc import BoundedCache  # noqa: E402
_LEVEL_CACHE = BoundedCache(32)


def _level_tensors(hw, device):
    """(spatial_shapes [L,2], level_start_index [L]) int64 on `device`, cached per geometry: creating them
    from Python lists is a blocking host-to-device copy (the reference pays it every forward, :100-101).
    The host copy rides along as an attribute so that the modules never have to read it back."""
    key = (hw, str(device))
    got = _LEVEL_CACHE.get(key)
    if got is None:
        shapes = torch.as_tensor(hw, dtype=torch.long, device=device)
        shapes._snipper_host = [tuple(x) for x in hw]
        starts = [0]
        for h, w in hw[:-1]:
            starts.append(starts[-1] + h * w)
        got = (shapes, torch.as_tensor(starts, dtype=torch.long, device=device))
        _LEVEL_CACHE[key] = got
    return got


def _clones(module, n):
    return nn.ModuleList([copy.deepcopy(module) for _ in range(n)])


def _activation(name):
    try:
        return {"relu": F.relu, "gelu": F.gelu, "glu": F.glu}[name]
    except KeyError:
        raise RuntimeError(f"activation should be relu/gelu, not {name}.")


def _amp_bf16(x) -> bool:
    return x.is_cuda and torch.is_autocast_enabled('cuda') and torch.get_autocast_dtype('cuda') == torch.bfloat16


def _residual_norm(x, z, norm, drop):
    """norm(x + drop(z)): one launch of the fused kernel (csrc/ln_fused.cuh) when it applies."""
    from .fused import add_dropout_layer_norm, ln_fusable
    if ln_fusable(x, norm) and z.dtype in (torch.float32, torch.bfloat16) and z.shape == x.shape and \
            isinstance(drop, nn.Dropout):
        return add_dropout_layer_norm(x, z, norm, drop.p, drop.training)[0].to(torch.result_type(x, z))
    return norm(x + drop(z))


class _QKVProj(_Fn):
    """The packed input projection of nn.MultiheadAttention for "query is key, value differs" (the decoder's
    self-attention: q = k = tgt + pos, v = tgt), batch-first: [q | k] = x_qk @ W[:2E]^T + b[:2E], v = x_v @ W[2E:]^T +
    b[2E:].  Two GEMMs instead of three, and dW / db are assembled here -- the module's own path slices the packed
    weight three ways, which autograd pays back with a zero-fill, a copy and an add per slice."""

    @staticmethod
    def forward(ctx, x_qk, x_v, weight, bias):
        E = x_qk.shape[-1]
        a, b = x_qk.reshape(-1, E), x_v.reshape(-1, E)
        ctx.save_for_backward(a, b, weight)
        from .dense import _sg_dense, _sg_ok, small_gemm_batch
        a2, b2 = _sg_dense(a), _sg_dense(b)
        if _sg_ok(a2, b2, weight) and bias.dtype == torch.float32 and bias.is_contiguous() and a2.shape[0] <= 8192:
            # both projections in one launch of the small-GEMM kernel (csrc/small_linear.cuh)
            qk = torch.empty((a2.shape[0], 2 * E), dtype=torch.float32, device=a.device)
            v = torch.empty((b2.shape[0], E), dtype=torch.float32, device=a.device)
            small_gemm_batch([(a2, False, weight[:2 * E], True, qk, bias[:2 * E], None),
                              (b2, False, weight[2 * E:], True, v, bias[2 * E:], None)])
        else:
            qk = torch.addmm(bias[:2 * E], a, weight[:2 * E].t())
            v = torch.addmm(bias[2 * E:], b, weight[2 * E:].t())
        return qk.view(*x_qk.shape[:-1], 2 * E), v.view(*x_v.shape[:-1], E)

    @staticmethod
    def backward(ctx, gqk, gv):
        from .dense import _ones_row
        a, b, weight = ctx.saved_tensors
        E = a.shape[1]
        gqk2, gv2 = gqk.reshape(-1, 2 * E), gv.reshape(-1, E)
        from .dense import small_gemm_batch, small_linear_backward
        if gqk2.is_cuda and all(ctx.needs_input_grad):
            # the whole backward in one launch: two data gradients, the two row blocks of the packed dW and of db
            dW = torch.empty_like(weight)
            db = torch.empty((3 * E,), dtype=weight.dtype, device=weight.device)
            r1 = small_linear_backward(gqk2, a, weight[:2 * E], dw_out=dW[:2 * E], db_out=db[:2 * E], launch=False)
            r2 = small_linear_backward(gv2, b, weight[2 * E:], dw_out=dW[2 * E:], db_out=db[2 * E:], launch=False)
            if r1 is not None and r2 is not None:
                small_gemm_batch(r1[3] + r2[3])
                return r1[0].view(*gqk.shape[:-1], E), r2[0].view(*gv.shape[:-1], E), dW, db
        dxa = torch.mm(gqk2, weight[:2 * E]).view(*gqk.shape[:-1], E) if ctx.needs_input_grad[0] else None
        dxb = torch.mm(gv2, weight[2 * E:]).view(*gv.shape[:-1], E) if ctx.needs_input_grad[1] else None
        dW = db = None
        if ctx.needs_input_grad[2]:
            dW = torch.cat([torch.mm(gqk2.t(), a), torch.mm(gv2.t(), b)], 0)
        if ctx.needs_input_grad[3]:
            ones = _ones_row(gqk2.shape[0], gqk2.device, gqk2.dtype)
            db = torch.cat([torch.mm(ones, gqk2).view(-1), torch.mm(ones, gv2).view(-1)], 0)
        return dxa, dxb, dW, db


class _SmallAttention(_Fn):
    """softmax(q k^T / sqrt(hd)) with dropout, times v, for the decoder's few hundred queries: one launch forward, one
    backward (csrc/small_attention.cuh).  ``qk`` [bs, L, 2E] is the packed q | k projection output, read in place; its
    gradient comes back as one [bs, L, 2E] tensor (no concatenation)."""

    @staticmethod
    def forward(ctx, qk, v, H, p):
        from . import _lib
        bs, L, E2 = qk.shape
        E = E2 // 2
        hd = E // H
        out = torch.empty((bs, L, E), dtype=torch.float32, device=qk.device)
        P = torch.empty((bs, H, L, L), dtype=torch.float32, device=qk.device)
        seed = 0
        if p > 0.0:
            from .fused import _next_seed
            seed = _next_seed()
        scale = hd ** -0.5
        with _lib.device_guard(qk.device):
            rc = _lib.load().snipper_small_attention_forward_f32(
                _lib.raw_stream(qk.device), qk.data_ptr(), E2, L * E2, qk.data_ptr() + 4 * E, E2, L * E2,
                v.data_ptr(), E, L * E, out.data_ptr(), E, L * E, P.data_ptr(), bs, H, L, hd, scale, float(p), seed)
        _lib.check(rc, "snipper_small_attention_forward_f32")
        ctx.save_for_backward(qk, v, out, P)
        ctx.cfg = (H, float(p), seed, scale)
        return out

    @staticmethod
    def backward(ctx, gout):
        from . import _lib
        qk, v, out, P = ctx.saved_tensors
        H, p, seed, scale = ctx.cfg
        bs, L, E2 = qk.shape
        E = E2 // 2
        g = gout.contiguous()
        dqk = torch.empty_like(qk)
        dv = torch.empty_like(v)
        with _lib.device_guard(qk.device):
            rc = _lib.load().snipper_small_attention_backward_f32(
                _lib.raw_stream(qk.device), qk.data_ptr(), E2, L * E2, qk.data_ptr() + 4 * E, E2, L * E2,
                v.data_ptr(), E, L * E, out.data_ptr(), E, L * E, P.data_ptr(), g.data_ptr(), E, L * E,
                dqk.data_ptr(), E2, L * E2, dqk.data_ptr() + 4 * E, E2, L * E2, dv.data_ptr(), E, L * E,
                bs, H, L, E // H, scale, p, seed)
        _lib.check(rc, "snipper_small_attention_backward_f32")
        return dqk, dv, None, None


SMALL_ATTENTION_MAX_L = 384       # csrc/small_attention.cuh: kSaMaxL2 (up to 256: both row sets staged at once)


def _self_attention(mha: nn.MultiheadAttention, x_qk, x_v):
    """``mha(x_qk^T, x_qk^T, x_v^T, need_weights=False)[0]^T`` for batch-first inputs [bs, L, E] -- the decoder's dense
    self-attention (reference models/deformable_transformer.py:282-287) without the module's layout round trips: it
    wants [L, bs, E], so the reference transposes in and out and the module copies q, k, v and the result between the
    layouts (~10 copy kernels per call and direction).  Same parameters, same arithmetic (scaled dot-product attention
    with dropout on the probabilities, packed input projection, output projection)."""
    E = x_qk.shape[-1]
    ok = (x_qk.is_cuda and x_qk.dtype == torch.float32 and not torch.is_autocast_enabled('cuda') and
          mha._qkv_same_embed_dim and mha.in_proj_bias is not None and mha.bias_k is None and not mha.add_zero_attn and
          not mha.batch_first and mha.out_proj.bias is not None)
    if not ok:
        return mha(x_qk.transpose(0, 1), x_qk.transpose(0, 1), x_v.transpose(0, 1), need_weights=False)[0].transpose(0, 1)
    from .dense import _SmallLinear
    bs, L, _ = x_qk.shape
    H = mha.num_heads
    qk, v = _QKVProj.apply(x_qk, x_v, mha.in_proj_weight, mha.in_proj_bias)
    if L <= SMALL_ATTENTION_MAX_L and E // H in (32, 48) and qk.is_contiguous() and v.is_contiguous():
        att = _SmallAttention.apply(qk, v, H, mha.dropout if mha.training else 0.0)
        return _SmallLinear.apply(att, mha.out_proj.weight, mha.out_proj.bias)
    q = qk[..., :E].view(bs, L, H, E // H).transpose(1, 2)               # [bs, H, L, hd] views
    k = qk[..., E:].view(bs, L, H, E // H).transpose(1, 2)
    vh = v.view(bs, L, H, E // H).transpose(1, 2)
    att = F.scaled_dot_product_attention(q, k, vh, dropout_p=mha.dropout if mha.training else 0.0)
    att = att.transpose(1, 2).reshape(bs, L, E)
    return _SmallLinear.apply(att, mha.out_proj.weight, mha.out_proj.bias)


class _FFNMixin:
    """Linear -> act -> dropout -> Linear, residual, LayerNorm (shared by both layer types)."""

    def _ffn(self, x, drop_a, drop_b, norm):
        if self.activation is F.relu:
            y = big_ffn(x, self.linear1, self.linear2, drop_a)            # the whole block as one node when it applies
            if y is None:
                from .dense import small_ffn
                y = small_ffn(x, self.linear1, self.linear2, drop_a)      # decoder-size float32 rows
            if y is not None:
                return _residual_norm(x, y, norm, drop_b)
            h = big_linear(x, self.linear1, relu=True, dropout=drop_a)   # ReLU + dropout in the kernel's epilogue
        else:
            h = drop_a(self.activation(big_linear(x, self.linear1)))
        y = big_linear(h, self.linear2)
        return _residual_norm(x, y, norm, drop_b)


class DeformableTransformerEncoderLayer(nn.Module, _FFNMixin):
    def __init__(self, d_model=256, d_ffn=1024, dropout=0.1, activation="relu",
                 n_levels=4, n_heads=8, n_points=4, n_frame=4, use_pytroch_deform=False):
        super().__init__()
        self.self_attn = MSDeformAttn(d_model, n_levels, n_heads, n_points, n_frame, 'encoder', use_pytroch_deform)
        self.dropout1 = nn.Dropout(dropout)
        self.norm1 = nn.LayerNorm(d_model)
        self.linear1 = nn.Linear(d_model, d_ffn)
        self.activation = _activation(activation)
        self.dropout2 = nn.Dropout(dropout)
        self.linear2 = nn.Linear(d_ffn, d_model)
        self.dropout3 = nn.Dropout(dropout)
        self.norm2 = nn.LayerNorm(d_model)

    @staticmethod
    def with_pos_embed(tensor, pos):
        return tensor if pos is None else tensor + pos

    def forward_ffn(self, src):
        return self._ffn(src, self.dropout2, self.dropout3, self.norm2)

    def forward(self, src, pos, reference_points, spatial_shapes, level_start_index, padding_mask=None):
        attended = self.self_attn(self.with_pos_embed(src, pos), reference_points, src,
                                  spatial_shapes, level_start_index, padding_mask)
        src = _residual_norm(src, attended, self.norm1, self.dropout1)
        return self.forward_ffn(src)

    # -- the same layer with the bf16 companions of the residual stream carried along -----------------------
    fused_residual = True        # class-level switch (tests compare both formulations)
    native = __import__("os").environ.get("SNIPPER_ENC_NATIVE", "1") != "0"     # (A/B aid: 0 = one autograd node per module)

    def fused_ok(self, src) -> bool:
        from .fused import ln_fusable
        return (self.fused_residual and src.dtype == torch.float32 and _amp_bf16(src) and ln_fusable(src, self.norm1) and
                ln_fusable(src, self.norm2) and self.activation is F.relu)

    def forward_fused(self, src32, src16, q16, pos16, reference_points, spatial_shapes, level_start_index,
                      padding_mask, last: bool):
        """One encoder layer under bf16 autocast.  The residual stream stays float32 (``src32``); every consumer that
        computes in bf16 gets a bf16 view written by the kernel that produced the float32 one: ``src16`` feeds the
        value projection, ``q16`` = bf16(src + pos) the offset / weight projections.  Their gradients come back into
        the fused LayerNorm backward, which adds them to the residual gradient in registers -- no cast, add or
        gradient-accumulation kernels in between.  Returns (src32, src16, q16) of the next layer."""
        from .fused import add_dropout_layer_norm
        if (self.native and _LN_LAZY and torch.is_grad_enabled() and src32.requires_grad and src32.is_contiguous() and
                is_no_padding(padding_mask) and src32.dim() == 4 and src16.dtype == torch.bfloat16 and q16.dtype == torch.bfloat16 and
                (last or (torch.is_tensor(pos16) and pos16.dtype == torch.bfloat16))):
            # round 6: the whole layer from ONE native call per direction (include/snipper_layers.h, encoder_native.py)
            from . import encoder_native as EN
            hw = getattr(spatial_shapes, "_snipper_host", None)
            if hw is not None and EN.timing_off():
                bs_, t_, s_, _c = src32.shape
                plan = EN.make_plan(self, bs_, t_, s_, hw, last)
                shadows = EN.lookup_shadows(self) if plan is not None else None
                if shadows is not None:
                    return EN.layer_forward(self, plan, shadows, src32, src16, q16, pos16, reference_points, spatial_shapes,
                                            level_start_index, last)
        attended = self.self_attn(q16, reference_points, src16, spatial_shapes, level_start_index, padding_mask)
        # (the float32 result of norm1 feeds norm2's residual only, and -- unless this is the last layer -- norm2's feeds the
        #  next layer's norm1 only: neither is materialised, the consuming kernel recomputes it from the saved pre-norm sum)
        lazy = "lazy" if _LN_LAZY else True
        y32, y16, _ = add_dropout_layer_norm(src32, attended, self.norm1, self.dropout1.p, self.training,
                                             want=(lazy, True, False))
        z = big_ffn(y16, self.linear1, self.linear2, self.dropout2)
        if z is None:
            z = big_linear(big_linear(y16, self.linear1, relu=True, dropout=self.dropout2), self.linear2)
        return add_dropout_layer_norm(y32, z, self.norm2, self.dropout3.p, self.training, pos=pos16,
                                      want=(True if last else lazy, True, not last))


import os as _os
_DEC_PREMIX = _os.environ.get("SNIPPER_DEC_PREMIX", "1") != "0"    # (A/B aid: 0 = temporal mean per decoder layer, as round 4)
_LN_LAZY = _os.environ.get("SNIPPER_LN_LAZY", "1") != "0"      # (A/B aid: 0 = materialise the float32 residual stream)


class DeformableTransformerEncoder(nn.Module):
    def __init__(self, encoder_layer, num_layers):
        super().__init__()
        self.layers = _clones(encoder_layer, num_layers)
        self.num_layers = num_layers

    @staticmethod
    def get_reference_points(spatial_shapes, valid_ratios, device):
        """Pixel centres of every level, normalised by the valid extent, then re-expressed in each
        level's padded frame: [B, S, L, 2] (reference :220-232)."""
        hw = getattr(spatial_shapes, "_snipper_host", None) or spatial_shapes.tolist()
        if getattr(valid_ratios, "_snipper_ones", False) and not torch.is_inference_mode_enabled():   # a constant grid
            key = ("ref", valid_ratios.shape[0], tuple(tuple(x) for x in hw), str(device))
            got = _LEVEL_CACHE.get(key)
            if got is None:
                got = _LEVEL_CACHE[key] = DeformableTransformerEncoder._reference_points(hw, valid_ratios, device)
            return got
        return DeformableTransformerEncoder._reference_points(hw, valid_ratios, device)

    @staticmethod
    def _reference_points(hw, valid_ratios, device):
        per_level = []
        for lvl, (H, W) in enumerate(hw):
            ys = torch.arange(H, dtype=torch.float32, device=device) + 0.5
            xs = torch.arange(W, dtype=torch.float32, device=device) + 0.5
            gy = ys.view(1, H, 1) / (valid_ratios[:, lvl, 1].view(-1, 1, 1) * H)
            gx = xs.view(1, 1, W) / (valid_ratios[:, lvl, 0].view(-1, 1, 1) * W)
            per_level.append(torch.stack([gx.expand(-1, H, W), gy.expand(-1, H, W)], -1).flatten(1, 2))
        centres = torch.cat(per_level, 1)                       # [B, S, 2]
        return centres[:, :, None] * valid_ratios[:, None]      # [B, S, L, 2]

    def fused_ok(self, src) -> bool:
        return all(hasattr(l, "fused_ok") and l.fused_ok(src) for l in self.layers)

    def forward(self, src, spatial_shapes, level_start_index, valid_ratios, pos=None, padding_mask=None, n_frame=1,
                twins=None):
        """``twins`` = (src16, q16, pos16): the bf16 companions of ``src`` / ``src + pos`` / ``pos`` when the caller
        already has them (the token-row input projections write them); otherwise they are made here."""
        ref = self.get_reference_points(spatial_shapes, valid_ratios, device=src.device)
        ref = ref.unsqueeze(1).expand(-1, n_frame, -1, -1, -1)  # same grid for every frame
        if getattr(valid_ratios, "_snipper_ones", False) and not torch.is_inference_mode_enabled():
            # a constant of the shapes (no padding): keep the expanded grid materialised -- every layer's prologue kernel wants
            # it contiguous, and expanding + copying [b, T, S, L, 2] once per layer and step was six 6-us copies
            hw = getattr(spatial_shapes, "_snipper_host", None) or spatial_shapes.tolist()
            key = ("ref_frames", valid_ratios.shape[0], n_frame, tuple(tuple(x) for x in hw), str(src.device))
            got = _LEVEL_CACHE.get(key)
            if got is None:
                got = _LEVEL_CACHE[key] = ref.contiguous()
            ref = got
        out = src
        if pos is not None and self.fused_ok(src):
            if twins is not None:
                out16, q16, pos16 = twins
            else:
                pos16 = pos.to(torch.bfloat16)                  # gradients of the 6 uses accumulate in bf16
                out16, q16 = out.to(torch.bfloat16), (out + pos).to(torch.bfloat16)
            # (pos16 may be a list: one alias per layer that adds it, see DeformableTransformer.forward_from_features)
            pos_l = list(pos16) if isinstance(pos16, (list, tuple)) else None
            for i, layer in enumerate(self.layers):
                p16 = pos16 if pos_l is None else pos_l[min(i, len(pos_l) - 1)]
                out, out16, q16 = layer.forward_fused(out, out16, q16, p16, ref, spatial_shapes, level_start_index,
                                                      padding_mask, last=i + 1 == len(self.layers))
            out._snipper_bf16 = out16                           # big_linear picks the bf16 twin up (decoder memory)
            return out
        for layer in self.layers:
            out = layer(out, pos, ref, spatial_shapes, level_start_index, padding_mask)
        return out


class DeformableTransformerDecoderLayer(nn.Module, _FFNMixin):
    def __init__(self, d_model=256, d_ffn=1024, dropout=0.1, activation="relu",
                 n_levels=4, n_heads=8, n_points=4, n_frame=4, use_pytroch_deform=False):
        super().__init__()
        self.cross_attn = MSDeformAttn(d_model, n_levels, n_heads, n_points, n_frame,
                                       'decoder', use_pytroch_deform, True)
        self.dropout1 = nn.Dropout(dropout)
        self.norm1 = nn.LayerNorm(d_model)
        self.self_attn = nn.MultiheadAttention(d_model, n_heads, dropout=dropout)
        self.dropout2 = nn.Dropout(dropout)
        self.norm2 = nn.LayerNorm(d_model)
        self.linear1 = nn.Linear(d_model, d_ffn)
        self.activation = _activation(activation)
        self.dropout3 = nn.Dropout(dropout)
        self.linear2 = nn.Linear(d_ffn, d_model)
        self.dropout4 = nn.Dropout(dropout)
        self.norm3 = nn.LayerNorm(d_model)

    @staticmethod
    def with_pos_embed(tensor, pos):
        return tensor if pos is None else tensor + pos

    def forward_ffn(self, tgt):
        return self._ffn(tgt, self.dropout3, self.dropout4, self.norm3)

    # Under autocast the layer's own tensors ([bs, T * queries, C]: a few hundred rows) stay in float32: at that
    # size a GEMM is launch-bound in any precision, while every autocast cast (input, weight and bias of each
    # Linear, and their gradients back) is one more launch -- ~75 per layer and step.  Only the cross-attention,
    # whose value projection runs over the whole encoder memory, keeps the reduced precision.
    small_in_fp32 = True

    def forward(self, tgt, query_pos, reference_points, src, src_spatial_shapes, level_start_index,
                src_padding_mask=None, cross_amp_dtype=None):
        """``cross_amp_dtype``: set by a caller that has already left autocast (the decoder below) to the autocast
        dtype the cross-attention should still run in."""
        if cross_amp_dtype is None and self.small_in_fp32 and tgt.is_cuda and torch.is_autocast_enabled('cuda'):
            amp_dtype = torch.get_autocast_dtype('cuda')
            with torch.autocast("cuda", enabled=False):
                return self._forward(tgt.float(), query_pos.float(), reference_points, src, src_spatial_shapes,
                                     level_start_index, src_padding_mask, amp_dtype)
        return self._forward(tgt, query_pos, reference_points, src, src_spatial_shapes, level_start_index,
                             src_padding_mask, cross_amp_dtype)

    def _cross(self, amp_dtype, query, reference_points, src, *rest):
        # When the encoder memory carries the bf16 twin its last kernel wrote (big_linear picks it up for the value
        # projection, the only large GEMM here), the module can stay outside autocast: its query-side Linears see
        # 480 rows, where autocast only adds casts of inputs, weights, biases and their gradients (~18 per layer).
        if (amp_dtype is None or getattr(src, "_snipper_bf16", None) is not None or
                getattr(src, "_snipper_premixed", None) is not None):
            return self.cross_attn(query, reference_points, src, *rest)
        with torch.autocast("cuda", dtype=amp_dtype):
            return self.cross_attn(query, reference_points, src, *rest)

    def _forward(self, tgt, query_pos, reference_points, src, src_spatial_shapes, level_start_index,
                 src_padding_mask, amp_dtype):
        bs, t, lq, c = tgt.shape
        # dense self-attention over all (frame, query) tokens of a sample (reference :282-287)
        flat = tgt.reshape(bs, t * lq, c)
        mixed = _self_attention(self.self_attn, self.with_pos_embed(flat, query_pos.reshape(bs, t * lq, c)), flat)
        tgt = _residual_norm(flat, mixed, self.norm2, self.dropout2).view(bs, t, lq, c)
        # deformable cross-attention into the encoder memory (reference :290-295)
        attended, atten_data = self._cross(amp_dtype, self.with_pos_embed(tgt, query_pos.view(bs, t, lq, c)),
                                           reference_points, src, src_spatial_shapes,
                                           level_start_index, src_padding_mask)
        tgt = _residual_norm(tgt, attended, self.norm1, self.dropout1)
        return self.forward_ffn(tgt), atten_data


    # -- round 6: the layer as a chain without element-wise launches ------------------------------------------------------
    chain = __import__("os").environ.get("SNIPPER_DEC_CHAIN", "1") != "0"      # (A/B aid: 0 = the node-per-module form above)

    def chain_ok(self, tgt) -> bool:
        """Can ``forward_chain`` run?  Float32 CUDA rows outside autocast, the fused decoder-size kernels for every piece
        (csrc/small_ln.cuh, small_linear.cuh, small_attention.cuh), ReLU feed-forward."""
        from .fused import small_ln_ok
        mha = self.self_attn
        c = tgt.shape[-1]
        probe = tgt.reshape(-1, c)
        return (self.chain and tgt.is_cuda and tgt.dtype == torch.float32 and not torch.is_autocast_enabled('cuda') and
                torch.is_grad_enabled() and self.activation is F.relu and
                all(small_ln_ok(probe, n) for n in (self.norm1, self.norm2, self.norm3)) and
                all(isinstance(d, nn.Dropout) for d in (self.dropout1, self.dropout2, self.dropout3, self.dropout4)) and
                mha._qkv_same_embed_dim and mha.in_proj_bias is not None and mha.bias_k is None and not mha.add_zero_attn and
                not mha.batch_first and mha.out_proj.bias is not None and c // mha.num_heads in (32, 48) and
                mha.in_proj_weight.dtype == torch.float32 and 16 <= probe.shape[0])

    def forward_chain(self, state, pos_a, pos_b, reference_points, src, src_spatial_shapes, level_start_index,
                      src_padding_mask, amp_dtype, shape, last: bool):
        """One decoder layer (reference :276-300) on ``state = (x_v, x_res, x_q)``: three handles of the layer input -- the
        value input of the self-attention, the residual of norm2, and input + query_pos (the q / k input) -- all [bs, T * nq, C]
        float32.  ``pos_a`` / ``pos_b``: this layer's two aliases of query_pos (fused.FanOut).  Returns (next state or None,
        layer output [bs, T, nq, C], attention data).  Every residual + dropout + LayerNorm is ONE launch that also writes the
        position-added copy the next projection wants (fused.SmallLayerNorm); no ``with_pos_embed`` add, no gradient-accumulation
        add in the backward (every consumer of a LayerNorm output holds its own alias)."""
        from .dense import _SmallLinear, small_ffn
        from .fused import small_layer_norm
        bs, t, lq, c = shape
        x_v, x_res, x_q = state
        mha = self.self_attn
        if x_q.shape[1] > SMALL_ATTENTION_MAX_L:
            return None
        qk, v = _QKVProj.apply(x_q, x_v, mha.in_proj_weight, mha.in_proj_bias)
        att = _SmallAttention.apply(qk.contiguous(), v.contiguous(), mha.num_heads, mha.dropout if mha.training else 0.0)
        mixed = _SmallLinear.apply(att, mha.out_proj.weight, mha.out_proj.bias)
        t2_res, t2_q = small_layer_norm(x_res, mixed, self.norm2, self.dropout2.p, self.training, pos=pos_a, n_alias=1)
        attended, atten_data = self._cross(amp_dtype, t2_q.view(bs, t, lq, c), reference_points, src, src_spatial_shapes,
                                           level_start_index, src_padding_mask)
        t3_ffn, t3_res = small_layer_norm(t2_res, attended.reshape(bs, t * lq, c), self.norm1, self.dropout1.p, self.training,
                                          n_alias=2)
        y = small_ffn(t3_ffn, self.linear1, self.linear2, self.dropout3)
        if y is None:
            y = self.linear2(self.dropout3(F.relu(self.linear1(t3_ffn))))
        if last:
            (out,) = small_layer_norm(t3_res, y, self.norm3, self.dropout4.p, self.training, n_alias=1)
            return None, out.view(bs, t, lq, c), atten_data
        n_v, n_res, out, n_q = small_layer_norm(t3_res, y, self.norm3, self.dropout4.p, self.training, pos=pos_b, n_alias=3)
        return (n_v, n_res, n_q), out.view(bs, t, lq, c), atten_data


    # -- round 6: the same chain from ONE native call per direction (include/snipper_layers.h, decoder_native.py) ------------
    native = __import__("os").environ.get("SNIPPER_DEC_NATIVE", "1") != "0"    # (A/B aid: 0 = forward_chain's Python sequencing)

    def native_plan(self, shape, src, hw):
        """The decoder_native.LayerPlan for this layer when the native composites apply: the chain's conditions (checked by the
        caller), a premixed bf16 memory (DeformableTransformerDecoder._forward), tied Linears, contiguous float32 parameters."""
        pre = getattr(src, "_snipper_premixed", None)
        ca = self.cross_attn
        from .encoder_native import timing_off
        if not (self.native and timing_off() and pre is not None and hw is not None and pre.dtype == torch.bfloat16 and pre.dim() == 4 and
                ca.weights_are_tied() and not ca.use_pytroch_deform and ca.fused_elementwise):
            return None
        bs, t, lq, c = shape
        if pre.shape[0] != bs or pre.shape[1] != t or pre.shape[3] != c:
            return None
        from .decoder_native import layer_params, make_plan
        if not all(p.dtype == torch.float32 and p.is_contiguous() and p.data_ptr() % 16 == 0 for p in layer_params(self)):
            return None
        return make_plan(self, bs, t, lq, c, int(pre.shape[2]), hw, torch.bfloat16, int(src.shape[1]))

    def forward_native(self, plan, state, pos_a, pos_b, ref_in, ref_points, valid_ratios, src, shapes, lsi, root_lin, shape,
                       last: bool):
        """``forward_chain`` through decoder_native.DecoderLayerFn; also refines the reference points when ``root_lin`` (the
        root head as one nn.Linear) is given.  Returns (state, out, atten_data, (new_ref, ref_in_next) or None)."""
        from .decoder_native import DecoderLayerFn, layer_params
        bs, t, lq, c = shape
        x_v, x_res, x_q = state
        ca = self.cross_attn
        value = big_linear(src._snipper_premixed, ca.value_proj)           # [bs, t, S, c] bf16: the full-size product stays outside
        rw = rb = None
        if root_lin is not None:
            rw, rb = root_lin.weight.detach(), (root_lin.bias.detach() if root_lin.bias is not None else None)
        n_v, n_res, out, n_q, new_ref, ref_next, loc, prob = DecoderLayerFn.apply(
            plan, not last, x_v, x_res, x_q, pos_a, None if last else pos_b, value.contiguous(), ref_in.contiguous(),
            ref_points.contiguous() if rw is not None else None, valid_ratios.contiguous() if rw is not None else None,
            shapes, lsi, rw, rb, *layer_params(self))
        M, L, P = ca.n_heads, ca.n_levels, ca.n_points
        atten = ca._vis_lists(loc.view(bs, t, lq, M, L, P, 2), prob.view(bs, t, lq, M, L, P), plan.groups, bs, lq, M, L, P)
        refined = None
        if rw is not None:
            refined = (new_ref.view(bs, t, lq, 2), ref_next.view(bs, t, lq, L, 2))
        return (None if last else (n_v, n_res, n_q)), out.view(bs, t, lq, c), atten, refined


class DeformableTransformerDecoder(nn.Module):
    def __init__(self, decoder_layer, num_layers, return_intermediate=False):
        super().__init__()
        self.layers = _clones(decoder_layer, num_layers)
        self.num_layers = num_layers
        self.return_intermediate = return_intermediate
        # assigned by the model after construction (reference :310-311, model.py:103-104)
        self.root_embed = None
        self.class_embed = None

    def forward(self, query_obj, reference_points, src, src_spatial_shapes, src_level_start_index,
                src_valid_ratios, query_pos=None, src_padding_mask=None):
        if query_obj.is_cuda and torch.is_autocast_enabled('cuda') and \
                all(getattr(l, "small_in_fp32", False) for l in self.layers):
            # the decoder's own tensors stay float32 (see DeformableTransformerDecoderLayer.small_in_fp32)
            amp_dtype = torch.get_autocast_dtype('cuda')
            with torch.autocast("cuda", enabled=False):
                return self._forward(query_obj.float(), reference_points.float(), src, src_spatial_shapes,
                                     src_level_start_index, src_valid_ratios,
                                     None if query_pos is None else query_pos.float(), src_padding_mask, amp_dtype)
        return self._forward(query_obj, reference_points, src, src_spatial_shapes, src_level_start_index,
                             src_valid_ratios, query_pos, src_padding_mask, None)

    def _forward(self, query_obj, reference_points, src, src_spatial_shapes, src_level_start_index,
                 src_valid_ratios, query_pos, src_padding_mask, amp_dtype):
        out = query_obj
        inter, inter_ref, inter_att = [], [], []
        ref_in = None
        # the memory's bf16 twin feeds every layer's value projection: one alias per layer, so that the six gradients come
        # back on six edges and are summed in ONE pass (fused.FanOut) instead of five pairwise bf16 adds of [b, T, S, C]
        srcs = [src] * len(self.layers)
        twin = getattr(src, "_snipper_bf16", None)
        if (twin is not None and twin.is_cuda and twin.dtype == torch.bfloat16 and twin.requires_grad and
                torch.is_grad_enabled() and len(self.layers) > 1):
            from .fused import FanOut, TemporalMix
            from .ms_deform_attn import frame_neighbours
            # Round 5: the memory path ONCE per step instead of once per layer.  Every layer's cross attention samples the
            # temporal mean of the padding-masked, PROJECTED memory frames (ms_deform_attn.py:114-118, 130-226 with tied
            # Linears); mask fill, mean and projection are linear, so with no padding anywhere
            #     mix(W x + b) = W mix(x) + b        (the rows of the mix sum to 1)
            # and the mean of the MEMORY is the same for all layers: one mix launch here instead of one per layer (six 35-us
            # passes of 60 -> 121 MB), each layer projects the mean and samples its bf16 projection directly.  Taken when no
            # frame is padded (known on the host), there are no forecast frames (their query frames would add rows) and every
            # layer's Linears are tied; anything else keeps the per-layer evaluation.
            t_q, t_v = query_obj.shape[1], twin.shape[1]
            att0 = self.layers[0].cross_attn
            from .dense import BIG_LINEAR_MIN_ROWS
            # (round 6: forecast query frames -- t_q = T + F > t_v = T, reference ms_deform_attn.py:184-223 -- sample the mean of
            #  ALL value frames: their rows of the mix sum to 1 like everybody's, so the identity holds; the premixed memory then
            #  has t_q frames, 1.5 x the rows to project at T = 4 + 2, which still beats a temporal mix + float32 atomic backward
            #  per layer)
            premix = (_DEC_PREMIX and is_no_padding(src_padding_mask) and t_q >= t_v and t_q <= 8 and twin.dim() == 4 and
                      twin.numel() // twin.shape[-1] >= BIG_LINEAR_MIN_ROWS and         # (the bf16 GEMM path projects the mean)
                      all(getattr(l.cross_attn, "weights_are_tied", lambda: False)() and l.cross_attn.n_frame == t_v and
                          not l.cross_attn.use_pytroch_deform and l.cross_attn.d_model // l.cross_attn.n_heads == 48
                          for l in self.layers))
            fan_src = twin
            if premix:
                groups = [frame_neighbours(t1, att0.n_frame, t_v) for t1 in range(t_q)]
                mix = [[(1.0 / len(g)) if t2 in g else 0.0 for t2 in range(t_v)] for g in groups]
                fan_src = TemporalMix.apply(twin, None, mix, torch.bfloat16)
            srcs = []
            for alias in FanOut.apply(fan_src, len(self.layers)):
                s_l = src.view_as(src)
                if premix:
                    s_l._snipper_premixed = alias
                else:
                    s_l._snipper_bf16 = alias
                srcs.append(s_l)
        # round 6: the layers as ONE chain of fused launches (DeformableTransformerDecoderLayer.forward_chain) when every layer
        # can: the layer input travels as three handles, query_pos as one alias per use (summed by one launch in the backward)
        state = pos_aliases = None
        n_l = len(self.layers)
        if (query_pos is not None and out.dim() == 4 and query_pos.shape == out.shape and
                all(hasattr(l, "chain_ok") and l.chain_ok(out) for l in self.layers) and
                out.shape[1] * out.shape[2] <= SMALL_ATTENTION_MAX_L):
            from .fused import FanOut
            bs_, t_, lq_, c_ = out.shape
            flat = out.reshape(bs_, t_ * lq_, c_).contiguous()
            pos_flat = query_pos.reshape(bs_, t_ * lq_, c_).contiguous()
            pos_aliases = list(FanOut.apply(pos_flat, 2 * n_l)) if pos_flat.requires_grad else [pos_flat] * (2 * n_l)
            state = (flat, flat, flat + pos_aliases[0])
        for lid, layer in enumerate(self.layers):
            if ref_in is None:
                ref_in = reference_points[:, :, :, None, :] * src_valid_ratios[:, None, None, :, :]
            refined = None
            if state is not None:
                shape4 = tuple(out.shape)
                hw_host = getattr(src_spatial_shapes, "_snipper_host", None)
                plan = layer.native_plan(shape4, srcs[lid], hw_host) if hasattr(layer, "native_plan") else None
                if plan is not None:
                    root_lin = None
                    if (self.root_embed is not None and reference_points.dtype == torch.float32 and reference_points.dim() == 4 and
                            reference_points.shape[-1] == 2 and src_valid_ratios.dtype == torch.float32):
                        root_lin = _single_linear(self.root_embed[lid])
                        if root_lin is not None and not (root_lin.out_features >= 2 and root_lin.weight.dtype == torch.float32 and
                                                         root_lin.weight.is_contiguous() and root_lin.weight.data_ptr() % 16 == 0 and
                                                         (root_lin.bias is None or root_lin.bias.dtype == torch.float32)):
                            root_lin = None
                    state, out, atten_data, refined = layer.forward_native(
                        plan, state, pos_aliases[2 * lid + 1], pos_aliases[(2 * lid + 2) % (2 * n_l)], ref_in, reference_points,
                        src_valid_ratios, srcs[lid], src_spatial_shapes, src_level_start_index, root_lin, shape4,
                        last=lid + 1 == n_l)
                else:
                    state, out, atten_data = layer.forward_chain(
                        state, pos_aliases[2 * lid + 1], pos_aliases[(2 * lid + 2) % (2 * n_l)], ref_in, srcs[lid],
                        src_spatial_shapes, src_level_start_index, src_padding_mask, amp_dtype, shape4, last=lid + 1 == n_l)
            else:
                out, atten_data = layer(out, query_pos, ref_in, srcs[lid], src_spatial_shapes,
                                        src_level_start_index, src_padding_mask, cross_amp_dtype=amp_dtype)
            ref_in = None
            if refined is not None:           # (the native layer call has refined the reference points already)
                reference_points, ref_in = refined
            elif self.root_embed is not None:   # iterative refinement of the reference points (:329-333)
                fused = _refine_reference(self.root_embed[lid], out, reference_points, src_valid_ratios)
                if fused is not None:
                    reference_points, ref_in = fused
                else:
                    delta = self.root_embed[lid](out)[..., 0:2]
                    reference_points = (delta + inverse_sigmoid(reference_points)).sigmoid().detach()
            if self.return_intermediate:
                inter.append(out)
                inter_ref.append(reference_points)
                inter_att.append(atten_data)
        if self.return_intermediate:
            return torch.stack(inter), torch.stack(inter_ref), inter_att
        return out, reference_points, inter_att


class DeformableTransformer(nn.Module):
    def __init__(self, d_model=256, nhead=8, num_encoder_layers=6, num_decoder_layers=6,
                 dim_feedforward=1024, dropout=0.1, activation="relu", return_intermediate_dec=False,
                 num_feature_levels=4, dec_n_points=4, enc_n_points=4,
                 n_frame=4, n_future_frame=2, use_pytroch_deform=False, num_keypoints=15):
        super().__init__()
        self.d_model, self.nhead = d_model, nhead
        self.n_frame, self.n_future_frame = n_frame, n_future_frame
        self.num_keypoints = num_keypoints
        enc_layer = DeformableTransformerEncoderLayer(d_model, dim_feedforward, dropout, activation,
                                                      num_feature_levels, nhead, enc_n_points, n_frame,
                                                      use_pytroch_deform)
        self.encoder = DeformableTransformerEncoder(enc_layer, num_encoder_layers)
        dec_layer = DeformableTransformerDecoderLayer(d_model, dim_feedforward, dropout, activation,
                                                      num_feature_levels, nhead, dec_n_points, n_frame,
                                                      use_pytroch_deform)
        self.decoder = DeformableTransformerDecoder(dec_layer, num_decoder_layers, return_intermediate_dec)
        self.level_embed = nn.Parameter(torch.Tensor(num_feature_levels, d_model))
        self.temporal_embed = nn.Parameter(torch.Tensor(n_frame + n_future_frame, d_model))
        self.reference_points = nn.Linear(d_model, 2)
        self._reset_parameters()

    def _reset_parameters(self):
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)   # includes temporal_embed, as in the reference (:59-61)
        for m in self.modules():
            if isinstance(m, MSDeformAttn):
                m._reset_parameters()
        nn.init.xavier_uniform_(self.reference_points.weight, gain=1.0)
        nn.init.constant_(self.reference_points.bias, 0.)
        nn.init.normal_(self.level_embed)

    def get_valid_ratio(self, mask):
        """mask [bs, c, t, h, w] -> fraction of each axis that is not padding, (w, h) (reference :69-77)."""
        H, W = mask.shape[-2:]
        live_h = (~mask[:, 0, 0, :, 0]).sum(1).float() / H
        live_w = (~mask[:, 0, 0, 0, :]).sum(1).float() / W
        return torch.stack([live_w, live_h], -1)

    def forward(self, srcs, masks, pos_embeds, query_embed):
        """srcs / masks / pos_embeds: per level [bs, c, t, h, w]; query_embed [(T+F)*n_query, 2c].

        Returns (hs, heatmaps, init_reference, inter_references, inter_att_data) as the reference (:167).
        """
        hw = [tuple(s.shape[-2:]) for s in srcs]
        tokens = lambda x: x.flatten(3).permute(0, 2, 3, 1)              # [bs, t, h*w, c]
        src = torch.cat([tokens(s) for s in srcs], 2)
        # the padding mask is constant along channels (model.py:156-157 replicates it; get_valid_ratio reads
        # channel 0 only): keep one channel and expand, so the modules can see that and skip the C-fold work
        mask = torch.cat([tokens(m[:, :1]) for m in masks], 2).expand(-1, -1, -1, srcs[0].shape[1])
        pos = torch.cat([tokens(p) + self.level_embed[lvl].view(1, 1, 1, -1)
                         for lvl, p in enumerate(pos_embeds)], 2)
        sizes = [h * w for h, w in hw]
        spatial_shapes, level_start_index = _level_tensors(tuple((int(h), int(w)) for h, w in hw), src.device)
        valid_ratios = torch.stack([self.get_valid_ratio(m) for m in masks], 1)   # [bs, L, 2]

        memory = self.encoder(src, spatial_shapes, level_start_index, valid_ratios, pos, mask, self.n_frame)
        return self._decode(memory, hw, sizes, spatial_shapes, level_start_index, valid_ratios, mask, query_embed)

    def tokens_path_ok(self, feats, input_proj) -> bool:
        """Can the input projections be evaluated as token rows (fused.InputProjTokens)?  bf16 NHWC feature maps under
        bf16 autocast, one (1x1 Conv2d + GroupNorm) per level, shapes the kernels take, a fused encoder."""
        if not (feats and feats[0].is_cuda and _amp_bf16(feats[0]) and len(feats) == len(input_proj)):
            return False
        C = self.d_model
        for f, proj in zip(feats, input_proj):
            if not (isinstance(proj, nn.Sequential) and len(proj) == 2 and isinstance(proj[0], nn.Conv2d) and
                    isinstance(proj[1], nn.GroupNorm)):
                return False
            conv, gn = proj
            if not (conv.kernel_size == (1, 1) and conv.stride == (1, 1) and conv.padding == (0, 0) and
                    conv.groups == 1 and conv.bias is not None and conv.in_channels % 64 == 0 and
                    conv.out_channels == C and gn.affine and C % gn.num_groups == 0 and
                    (C // gn.num_groups) % 4 == 0 and gn.num_groups <= 64 and C <= 1024 and C % 64 == 0 and
                    gn.num_groups == input_proj[0][1].num_groups and gn.eps == input_proj[0][1].eps):
                return False
            if not (f.dtype == torch.bfloat16 and f.dim() == 4 and f.is_contiguous(memory_format=torch.channels_last)):
                return False
        probe = torch.empty(0, dtype=torch.float32, device=feats[0].device).view(0, C)
        return self.encoder.fused_ok(probe)

    def forward_from_features(self, feats, masks, pos_tokens, input_proj, query_embed):
        """The same computation as ``forward(srcs, masks, pos_embeds, query_embed)`` starting one step earlier:
        feats L x [b*T, Cin, h, w] (backbone maps, bf16 NHWC), masks L x [b*T, h, w] bool, pos_tokens L x
        [b, T, h*w, C] float32 (position encoding, token-major), input_proj the model's projection modules.
        Check ``tokens_path_ok`` first."""
        from .fused import InputProjTokens, LevelPosTokens
        T, c = self.n_frame, self.d_model
        b = feats[0].shape[0] // T
        hw = [tuple(int(v) for v in f.shape[-2:]) for f in feats]
        sizes = [h * w for h, w in hw]
        # the fused encoder only ever reads the position encoding in bf16 (query = bf16(src + pos))
        # one alias of the bf16 position encoding per consumer (the projections' query + every encoder layer but the last):
        # their gradients then reach level_embed as column sums, not through n - 1 full-size adds (fused.LevelPosTokens)
        n_uses = max(1, len(self.encoder.layers))
        pos16_all = LevelPosTokens.apply(self.level_embed[:len(pos_tokens)], n_uses, *pos_tokens)
        pos16_all = list(pos16_all) if isinstance(pos16_all, (tuple, list)) else [pos16_all]
        pos16 = pos16_all[0]
        pos = pos16
        if all(is_no_padding(m) for m in masks) and not torch.is_inference_mode_enabled():
            # no padding anywhere (known on the host): the flattened mask and the valid ratios are constants of the shapes
            key = ("no_pad", b, T, tuple(hw), str(feats[0].device))
            got = _LEVEL_CACHE.get(key)
            if got is None:
                flat = torch.zeros((b, T, sum(sizes), 1), dtype=torch.bool, device=feats[0].device)
                ones = torch.ones((b, len(hw), 2), dtype=torch.float32, device=feats[0].device)
                ones._snipper_ones = True
                got = _LEVEL_CACHE[key] = (flat, ones)
            mask, valid_ratios = got[0].expand(-1, -1, -1, c), got[1]
            mask._snipper_all_false = True                    # (known on the host: the decoder's premixed memory path asks)
        else:
            mask = torch.cat([m.reshape(b, T, -1) for m in masks], 2)[..., None].expand(-1, -1, -1, c)
            ratios = []
            for m, (h, w) in zip(masks, hw):                              # get_valid_ratio on frame 0 of each sample
                m0 = m.view(b, T, h, w)[:, 0]
                ratios.append(torch.stack([(~m0[:, 0, :]).sum(1).float() / w, (~m0[:, :, 0]).sum(1).float() / h], -1))
            valid_ratios = torch.stack(ratios, 1)                         # [bs, L, 2]
        spatial_shapes, level_start_index = _level_tensors(tuple(hw), feats[0].device)
        params = [t for proj in input_proj for t in (proj[0].weight, proj[0].bias, proj[1].weight, proj[1].bias)]
        gn = input_proj[0][1]
        src, src16, q16 = InputProjTokens.apply(T, gn.num_groups, gn.eps, pos16, (True, True), *feats, *params)
        memory = self.encoder(src, spatial_shapes, level_start_index, valid_ratios, pos, mask, self.n_frame,
                              twins=(src16, q16, pos16_all[1:] if len(pos16_all) > 1 else pos16))
        return self._decode(memory, hw, sizes, spatial_shapes, level_start_index, valid_ratios, mask, query_embed)

    def _decode(self, memory, hw, sizes, spatial_shapes, level_start_index, valid_ratios, mask, query_embed):
        bs, _, _, c = memory.shape
        heatmaps = HeatmapViews()   # first num_keypoints channels of every head, per level (views; reference :141-149)
        for chunk, (h, w) in zip(memory.split(sizes, dim=2), hw):
            grid = chunk.reshape(bs, self.n_frame, h, w, self.nhead, c // self.nhead)
            heatmaps.append(grid[..., 0:self.num_keypoints])
        # (what the views are views OF: the criterion's fused heat-map loss reads the memory itself and writes its whole
        #  gradient in one launch -- criterion.HeatmapLoss)
        heatmaps.source = (memory, [tuple(x) for x in hw], self.nhead, self.num_keypoints)

        t_all = self.n_frame + self.n_future_frame
        n_query = query_embed.shape[0] // t_all
        query_pos, query_obj = torch.split(query_embed, c, dim=-1)
        query_pos = query_pos.reshape(t_all, n_query, c).unsqueeze(0).expand(bs, -1, -1, -1)
        query_pos = query_pos + self.temporal_embed.view(1, t_all, 1, c)
        query_obj = query_obj.reshape(t_all, n_query, c).unsqueeze(0).expand(bs, -1, -1, -1)
        init_reference = self.reference_points(query_pos).sigmoid()      # [bs, t, n_query, 2]

        hs, inter_references, inter_att = self.decoder(query_obj, init_reference, memory, spatial_shapes,
                                                       level_start_index, valid_ratios, query_pos, mask)
        return hs, heatmaps, init_reference, inter_references, inter_att


class HeatmapViews(list):
    """The per-level heat-map views of the encoder memory (a plain list for every consumer) with a note of where they
    came from: ``source`` = (memory [bs, T, S, C], [(h, w)] per level, n_heads, n_keypoints)."""
    source = None


def build_deforamble_transformer(args):
    return DeformableTransformer(
        d_model=args.hidden_dim, nhead=args.nheads,
        num_encoder_layers=args.enc_layers, num_decoder_layers=args.dec_layers,
        dim_feedforward=args.dim_feedforward, dropout=args.dropout, activation="relu",
        return_intermediate_dec=True, num_feature_levels=args.num_feature_levels,
        dec_n_points=args.dec_n_points, enc_n_points=args.enc_n_points,
        n_frame=args.num_frames, n_future_frame=args.num_future_frames,
        use_pytroch_deform=args.use_pytorch_deform, num_keypoints=args.num_kpts)
