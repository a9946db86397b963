"""Spatiotemporal multi-scale deformable attention module (host side, PyTorch).

API mirror of /root/reference/models/ops/modules/ms_deform_attn.py (``MSDeformAttn``): same
constructor (including the ``use_pytroch_deform`` spelling), same parameters and state_dict keys
(the per-frame offset / weight Linears are ONE module listed ``n_frame`` times, :68-71), same
``forward`` signature and return value (a tensor, or ``(tensor, (loc_list, weight_list))`` when
``attention_vis``).

What is different is how the forward is evaluated.  The reference walks every (query frame t1,
value frame t2) pair in Python and launches one core op per pair (:130-226; 10 launches per
encoder layer at T=4).  Here:

* tied path (the Linears of all frames are the same object, which is how the reference always
  builds them): offsets and logits do not depend on t2, so the joint softmax over L*P*|t2| is
  softmax_{L*P}/|t2| and, the core op being linear in ``value``, the sum over t2 equals ONE core
  op on the temporal mean of the neighbouring value frames.  All T1 query frames are folded into
  the batch dimension: one kernel launch per module forward (SURVEY.md section 0 fact 4 verified
  this identity against the reference in fp64).
* general path (someone untied the Linears, e.g. by loading a checkpoint into separately
  constructed modules): the per-pair formulation, one core op per pair, as the reference.

The core op is the gfx950 HIP kernel behind ``MSDeformAttnFunction``; there is no fallback.
``use_pytroch_deform=True`` selects the reference's own pure-PyTorch debug formulation, exactly
as the reference's flag does.
"""
from __future__ import annotations

import math
from collections.abc import Sequence
from typing import List, Optional

import torch
import torch.nn.functional as F
from torch import nn

from .dense import big_linear, big_linear_merged, merged_bias_is_outside
from .ms_deform_attn_func import MSDeformAttnFunction, ms_deform_attn_core_pytorch


def frame_neighbours(t1: int, n_frame: int, n_value_frames: int) -> List[int]:
    """Value frames that query frame ``t1`` attends to (reference :132-140, :184-189)."""
    if t1 < n_frame:   # observed frame: itself and its temporal neighbours
        return [t for t in (t1 - 1, t1, t1 + 1) if 0 <= t < n_frame]
    return list(range(n_value_frames))   # future (forecast) frame: every value frame


from .misc import BoundedCache  # noqa: E402
_SCALE_CACHE = BoundedCache(16)


def _level_scale(hw, dtype, device):
    """[(W_l, H_l)] as a device tensor.  Cached: building it from a Python list is a blocking host-to-device
    copy, i.e. a hidden synchronisation point in every forward."""
    key = (tuple(map(tuple, hw)), dtype, str(device))
    t = _SCALE_CACHE.get(key)
    if t is None:
        t = torch.tensor([[w, h] for h, w in hw], dtype=dtype, device=device)
        _SCALE_CACHE[key] = t
    return t


class _LazyList(Sequence):
    """A list whose items are built when first looked at (the attention-visualisation lists of the reference,
    ms_deform_attn.py:228-233: one entry per query frame)."""

    def __init__(self, build, n):
        self._build, self._n, self._items = build, n, None

    def _get(self):
        if self._items is None:
            self._items, self._build = self._build(), None
        return self._items

    def __len__(self):
        return self._n

    def __getitem__(self, i):
        return self._get()[i]

    def __iter__(self):
        return iter(self._get())


class MSDeformAttn(nn.Module):
    def __init__(self, d_model=256, n_levels=4, n_heads=8, n_points=4, n_frame=4,
                 mode='encoder', use_pytroch_deform=False, attention_vis=False):
        super().__init__()
        if d_model % n_heads != 0:
            raise ValueError('d_model must be divisible by n_heads, but got {} and {}'.format(d_model, n_heads))
        assert mode in ('encoder', 'decoder')
        self.im2col_step = 64
        self.d_model, self.n_levels, self.n_heads, self.n_points = d_model, n_levels, n_heads, n_points
        self.n_frame = n_frame
        self.mode = mode
        self.use_pytroch_deform = use_pytroch_deform
        self.attention_vis = attention_vis
        self.core_in_fp32 = True    # see _core
        self.fused_elementwise = True   # HIP kernels for mask / temporal mean / locations / softmax
        # bf16 autocast, encoder (Lq == S): keep the temporal mean of the (bf16) projected value in bf16 for the sampling
        # kernels -- halves the tap-row bytes that bound them; coordinates, weights, sums and all gradients stay float32
        # (fused.TiedSampler).  False = float32 mean, as in round 1.
        self.value_bf16 = True

        shared_offsets = nn.Linear(d_model, n_heads * n_levels * n_points * 2)
        shared_weights = nn.Linear(d_model, n_heads * n_levels * n_points)
        # one Linear per value frame in name, one Linear in fact (reference :68-71)
        self.sampling_offsets = nn.ModuleList([shared_offsets] * n_frame)
        self.attention_weights = nn.ModuleList([shared_weights] * n_frame)
        self.value_proj = nn.Linear(d_model, d_model)
        self.output_proj = nn.Linear(d_model, d_model)
        self._reset_parameters()

    def _reset_parameters(self):
        """Reference :78-97: zero offset weights, offsets biased onto n_heads directions scaled by
        the point index, zero attention logits, Xavier projections."""
        M, L, P = self.n_heads, self.n_levels, self.n_points
        theta = torch.arange(M, dtype=torch.float32) * (2.0 * math.pi / M)
        direction = torch.stack([theta.cos(), theta.sin()], -1)
        direction = direction / direction.abs().max(-1, keepdim=True)[0]
        grid = direction.view(M, 1, 1, 2) * torch.arange(1, P + 1, dtype=torch.float32).view(1, 1, P, 1)
        grid = grid.expand(M, L, P, 2).reshape(-1)
        with torch.no_grad():
            for lin in self.sampling_offsets:
                lin.weight.zero_()
                lin.bias.copy_(grid)
            for lin in self.attention_weights:
                lin.weight.zero_()
                lin.bias.zero_()
            nn.init.xavier_uniform_(self.value_proj.weight)
            self.value_proj.bias.zero_()
            nn.init.xavier_uniform_(self.output_proj.weight)
            self.output_proj.bias.zero_()

    # ------------------------------------------------------------------------------------
    def weights_are_tied(self) -> bool:
        o0, a0 = self.sampling_offsets[0], self.attention_weights[0]
        return all(m is o0 for m in self.sampling_offsets) and all(m is a0 for m in self.attention_weights)

    def untie_frame_weights(self):
        """Give every value frame its own copy of the offset / weight Linears (what loading a checkpoint with genuinely
        different ``sampling_offsets.{t}`` entries needs; the reference's forward indexes them by frame, :144,167).  The
        forward then takes the general per-pair evaluation."""
        import copy
        self.sampling_offsets = nn.ModuleList([copy.deepcopy(self.sampling_offsets[0]) for _ in range(self.n_frame)])
        self.attention_weights = nn.ModuleList([copy.deepcopy(self.attention_weights[0]) for _ in range(self.n_frame)])
        return self

    def _core(self, value, shapes, lsi, loc, attn):
        if self.use_pytroch_deform:
            return ms_deform_attn_core_pytorch(value, shapes, loc, attn)
        if value.dtype == torch.bfloat16 and self.core_in_fp32:
            # under bf16 autocast the sampling itself stays in fp32: its f32 backward has the
            # owner-computes kernels (3x faster than the atomic one), which dwarfs the bf16 forward's gain
            value, loc, attn = value.float(), loc.float(), attn.float()
        # under bf16 autocast the output projection that follows (and its data gradient that comes back) compute in
        # bf16: let the kernels write / read bf16 rows instead of casting around them
        rows_bf16 = (value.is_cuda and value.dtype == torch.float32 and torch.is_autocast_enabled('cuda') and
                     torch.get_autocast_dtype('cuda') == torch.bfloat16)
        return MSDeformAttnFunction.apply(value.contiguous(), shapes, lsi, loc.contiguous(),
                                          attn.contiguous(), self.im2col_step, rows_bf16)

    def forward(self, query, reference_points, input_flatten, input_spatial_shapes,
                input_level_start_index, input_padding_mask=None):
        """
        query            [N, T1, Lq, C]
        reference_points [N, T1, Lq, n_levels, 2]   in [0,1], (x, y)
        input_flatten    [N, T2, S, C]              S = sum_l H_l*W_l
        input_spatial_shapes [n_levels, 2] (H, W), input_level_start_index [n_levels]
        input_padding_mask   [N, T2, S, C] bool (True = padding), or None
        ->  [N, T1, Lq, C]
        """
        N, T1, Lq, C = query.shape
        _, T2, S, _ = input_flatten.shape
        M, L, P = self.n_heads, self.n_levels, self.n_points
        shapes = input_spatial_shapes
        hw = getattr(shapes, "_snipper_host", None)     # host copy cached by our transformer
        if hw is None:
            hw = shapes.tolist()                         # reference :112 pays the same host sync
        assert sum(h * w for h, w in hw) == S

        pre = getattr(input_flatten, "_snipper_premixed", None)
        if (pre is not None and self.weights_are_tied() and not self.use_pytroch_deform and hw is not None and
                pre.shape == (N, T1, S, C) and self._fusable(query, reference_points, None)):
            # the caller has already taken the temporal mean of the (unpadded) memory frames, the same for every decoder layer
            # (DeformableTransformerDecoder._forward): project the mean and sample the projection as it is
            out, locs, wts = self._forward_premixed(query, reference_points, pre, shapes, input_level_start_index, hw, T2)
            out = big_linear(out, self.output_proj)
            return (out, (locs, wts)) if self.attention_vis else out

        value = big_linear(input_flatten, self.value_proj)
        scale = _level_scale(hw, query.dtype, query.device)                       # (W_l, H_l), cached
        groups = [frame_neighbours(t1, self.n_frame, T2) for t1 in range(T1)]

        if self.weights_are_tied():
            # (the padding mask is applied inside, fused with the temporal mean where possible)
            out, locs, wts = self._forward_tied(query, reference_points, value.view(N, T2, S, M, C // M), shapes,
                                                input_level_start_index, scale, groups, hw, input_padding_mask)
        else:
            if input_padding_mask is not None:
                value = value.masked_fill(input_padding_mask, 0.0)
            value = value.view(N, T2, S, M, C // M)
            out, locs, wts = self._forward_pairs(query, reference_points, value, shapes,
                                                 input_level_start_index, scale, groups)
        out = big_linear(out, self.output_proj)
        if self.attention_vis:
            return out, (locs, wts)
        return out

    # -- one launch for the whole module ---------------------------------------------------
    def _fusable(self, query, ref, mask) -> bool:
        """The fused element-wise kernels take CUDA f32 / bf16, L*P <= 16, a power-of-two head count and a
        padding mask that is constant along C (how the model builds it: model.py:156-157)."""
        from . import fused
        M, L, P = self.n_heads, self.n_levels, self.n_points
        return (self.fused_elementwise and not self.use_pytroch_deform and fused.supported(query) and
                L * P <= 16 and L <= 8 and M <= 64 and (M & (M - 1)) == 0 and
                (self.d_model // M) % 4 == 0 and
                (mask is None or mask.stride(-1) == 0 or mask.dim() == 3))

    def _forward_tied(self, query, ref, value, shapes, lsi, scale, groups, hw=None, mask=None):
        N, T1, Lq, C = query.shape
        T2, S = value.shape[1], value.shape[2]
        M, L, P = self.n_heads, self.n_levels, self.n_points
        fuse = hw is not None and self._fusable(query, ref, mask)
        # offsets and logits are two Linears of the same query: one merged projection when the fused prologue (which
        # reads both halves in place) follows
        # (on the bf16 GEMM path the offsets' bias stays OUT of the projection's bf16 output and is added in float32 by the
        #  prologue kernel: csrc/msda_prologue.cuh -- the bias grid is the large part of an offset)
        pair = [self.sampling_offsets[0], self.attention_weights[0]]
        bias_out = bool(fuse and merged_bias_is_outside(query, pair))
        raw = big_linear_merged(query, pair, first_bias_outside=bias_out) if fuse else None
        if raw is None:
            off_raw = big_linear(query, self.sampling_offsets[0])                   # [N,T1,Lq, M*L*P*2]
            logit_raw = big_linear(query, self.attention_weights[0])                # [N,T1,Lq, M*L*P]
        if fuse:
            from .fused import MSDAPrologue, TemporalMix
            if raw is not None:
                loc, prob = MSDAPrologue.apply(raw, None, ref.expand(N, T1, Lq, L, 2), hw, M, L, P,
                                               pair[0].bias if bias_out else None)
            else:
                loc, prob = MSDAPrologue.apply(off_raw, logit_raw, ref.expand(N, T1, Lq, L, 2), hw, M, L, P)
            loc, prob = loc.view(N, T1, Lq, M, L, P, 2), prob.view(N, T1, Lq, M, L, P)
        else:
            off = off_raw.view(N, T1, Lq, M, L, P, 2)
            loc = ref[:, :, :, None, :, None, :] + off / scale[None, None, None, None, :, None, :]
            prob = F.softmax(logit_raw.view(N, T1, Lq, M, L * P), -1).view(N, T1, Lq, M, L, P)

        # temporal mean of the neighbouring value frames, per query frame (and the padding mask)
        identity = all(g == [t] for t, g in enumerate(groups))                     # T=1: nothing to average
        if fuse and not (identity and mask is None and value.dtype == torch.float32):
            mix = [[(1.0 / len(g)) if t2 in g else 0.0 for t2 in range(T2)] for g in groups]
            m2 = None if mask is None else (mask if mask.dim() == 3 else mask[..., 0])
            amp16 = (value.is_cuda and torch.is_autocast_enabled('cuda') and
                     torch.get_autocast_dtype('cuda') == torch.bfloat16)
            if value.is_cuda and value.dtype in (torch.float32, torch.bfloat16) and C // M == 48:
                # mean + core op as one node (fused.TiedSampler); under bf16 autocast the encoder's mean stays bf16
                from .fused import TiedSampler
                vbar16 = bool(self.value_bf16 and amp16 and value.dtype == torch.bfloat16 and Lq == S and
                              not self.attention_vis)
                out = TiedSampler.apply(value.reshape(N, T2, S, C), m2, mix, loc.reshape(N * T1, Lq, M, L, P, 2).contiguous(),
                                        prob.reshape(N * T1, Lq, M, L, P).contiguous(), shapes, lsi, M, self.im2col_step,
                                        amp16, vbar16)
                return out.view(N, T1, Lq, C), *self._vis_lists(loc, prob, groups, N, Lq, M, L, P)
            vbar = TemporalMix.apply(value.reshape(N, T2, S, C), m2, mix).view(N, T1, S, M, C // M)
        else:
            if mask is not None:
                value = value.masked_fill(mask.view(N, T2, S, -1, 1) if mask.dim() == 3 else
                                          mask.view(N, T2, S, M, C // M), 0.0)
            if identity:
                vbar = value
            else:
                mixm = torch.zeros(T1, T2, dtype=value.dtype, device=value.device)
                for t1, g in enumerate(groups):
                    mixm[t1, g] = 1.0 / len(g)
                vbar = torch.einsum('ts,nsx->ntx', mixm, value.reshape(N, T2, -1)).view(N, T1, S, M, C // M)
        out = self._core(vbar.reshape(N * T1, S, M, C // M), shapes, lsi,
                         loc.reshape(N * T1, Lq, M, L, P, 2), prob.reshape(N * T1, Lq, M, L, P))
        out = out.view(N, T1, Lq, C)
        return out, *self._vis_lists(loc, prob, groups, N, Lq, M, L, P)

    def _forward_premixed(self, query, ref, xbar, shapes, lsi, hw, value_frames=None):
        """Tied module core on a memory whose temporal mean has been taken already: value = value_proj(xbar) IS the sampled
        tensor (mask fill, mean and projection commute when nothing is padded: the mix rows sum to 1)."""
        from .fused import MSDAPrologue
        N, T1, Lq, C = query.shape
        S = xbar.shape[2]
        M, L, P = self.n_heads, self.n_levels, self.n_points
        value = big_linear(xbar, self.value_proj)                                   # [N, T1, S, C], bf16 on the big-GEMM path
        pair = [self.sampling_offsets[0], self.attention_weights[0]]
        bias_out = merged_bias_is_outside(query, pair)
        raw = big_linear_merged(query, pair, first_bias_outside=bias_out)
        if raw is not None:
            loc, prob = MSDAPrologue.apply(raw, None, ref.expand(N, T1, Lq, L, 2), hw, M, L, P, pair[0].bias if bias_out else None)
        else:
            loc, prob = MSDAPrologue.apply(big_linear(query, self.sampling_offsets[0]), big_linear(query, self.attention_weights[0]),
                                           ref.expand(N, T1, Lq, L, 2), hw, M, L, P)
        loc, prob = loc.view(N, T1, Lq, M, L, P, 2), prob.view(N, T1, Lq, M, L, P)
        # (float32 queries beside the bf16 projection -- the decoder under bf16 autocast: the kernels write / read float32 rows)
        out = MSDeformAttnFunction.apply(value.reshape(N * T1, S, M, C // M), shapes, lsi,
                                         loc.reshape(N * T1, Lq, M, L, P, 2), prob.reshape(N * T1, Lq, M, L, P), self.im2col_step,
                                         False, query.dtype == torch.float32 and value.dtype == torch.bfloat16)
        if out.dtype != query.dtype:
            out = out.to(query.dtype)
        groups = [frame_neighbours(t1, self.n_frame, T1 if value_frames is None else value_frames) for t1 in range(T1)]
        return out.view(N, T1, Lq, C), *self._vis_lists(loc, prob, groups, N, Lq, M, L, P)

    def _vis_lists(self, loc, prob, groups, N, Lq, M, L, P):
        locs = wts = None
        if self.attention_vis:   # same lists as reference :228-233, built as views -- and only when somebody looks at them
            # (a training step never does: the decoder hands them through to the model's output tuple, model.py:221; building
            #  them eagerly was one division launch and ~30 view operations per decoder layer and step)
            sizes = [len(g) for g in groups]
            ld, pd = loc.detach(), prob.detach()

            def build_locs():
                return [ld[:, t1].unsqueeze(-2).expand(N, Lq, M, L, P, k, 2) for t1, k in enumerate(sizes)]

            def build_wts():
                scaled = {k: pd / k for k in sorted(set(sizes))}      # one division per distinct group size, not per frame
                return [scaled[k][:, t1].unsqueeze(-1).expand(N, Lq, M, L, P, k) for t1, k in enumerate(sizes)]

            locs, wts = _LazyList(build_locs, len(sizes)), _LazyList(build_wts, len(sizes))
        return locs, wts

    # -- the reference's per-pair evaluation, for untied Linears ------------------------------
    def _forward_pairs(self, query, ref, value, shapes, lsi, scale, groups):
        N, T1, Lq, C = query.shape
        M, L, P = self.n_heads, self.n_levels, self.n_points
        outs, locs, wts = [], [], []
        for t1, g in enumerate(groups):
            q = query[:, t1]
            logits = torch.stack([self.attention_weights[t2](q).view(N, Lq, M, L, P) for t2 in g], -1)
            prob = F.softmax(logits.flatten(-3), -1).view(N, Lq, M, L, P, len(g))
            acc, loc_t1 = None, []
            for k, t2 in enumerate(g):
                off = self.sampling_offsets[t2](q).view(N, Lq, M, L, P, 2)
                loc = ref[:, t1, :, None, :, None, :] + off / scale[None, None, None, :, None, :]
                loc_t1.append(loc)
                o = self._core(value[:, t2], shapes, lsi, loc, prob[..., k])
                acc = o if acc is None else acc + o
            outs.append(acc)
            if self.attention_vis:
                locs.append(torch.stack(loc_t1, dim=-2).detach())
                wts.append(prob.detach())
        return torch.stack(outs, dim=1), (locs or None), (wts or None)
