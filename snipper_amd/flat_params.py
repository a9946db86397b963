"""One flat float32 buffer per optimizer group: the reference's AdamW + clipping on 3 tensors instead of ~330.

The reference updates its parameters with ``torch.optim.AdamW`` over three groups (main.py:201-221) after
``clip_grad_norm_`` (engine.py:74).  With this model's ~330 parameter tensors the fused multi-tensor implementation
launches 10 AdamW kernels of <= 36 tensors each -- most of them biases and LayerNorm vectors, so a launch covers a few
dozen workgroups of a 256-CU GPU (measured: 0.50 ms for 1.2 GB of traffic = 2.4 TB/s) -- and the clipping another
4 norm + 4 scale launches (0.2 ms).  AdamW and the global-norm clipping are element-wise / a plain sum of squares, so
they do not care where tensor boundaries are: ``FlatParameters`` moves every trainable parameter of a group into ONE
flat buffer (each ``p.data`` becomes a view of its slice with the parameter's own shape and strides -- NHWC
convolution weights stay NHWC), exposes one leaf tensor per group to the optimizer, and keeps one flat gradient buffer
next to it.  The optimizer is still ``torch.optim.AdamW`` (fused), the clipping still ``clip_grad_norm_``: same
arithmetic per element, 3 tensors.

Gradients: autograd hands over a fresh tensor per parameter (``p.grad = None`` before backward, so no accumulation
kernel per parameter); ``pack()`` copies them into the flat buffer with multi-tensor copies.  With
``grad_sync.FlatGradSync`` (N > 1) the gradients already live in ITS flat buffer after ``sync()``; pass that buffer
as ``grad_flat`` (same parameter order) and ``pack()`` has nothing to do.

The kernels' bf16 weight copies (shadow.py) are valid while a parameter's version counter is unchanged; an update
through the flat leaf does not touch the views' counters, so ``after_step()`` bumps them.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch
from torch import nn


class FlatParameters:
    def __init__(self, groups: Sequence[Sequence[nn.Parameter]], grad_flat: Optional[torch.Tensor] = None):
        self.groups: List[List[nn.Parameter]] = [[p for p in g if p.requires_grad] for g in groups]
        self.params: List[nn.Parameter] = [p for g in self.groups for p in g]
        assert self.params, "no trainable parameters"
        assert len({id(p) for p in self.params}) == len(self.params), "a parameter may belong to one group only"
        dev = self.params[0].device
        assert all(p.device == dev and p.dtype == torch.float32 for p in self.params), "float32 parameters on one device"
        from .grad_sync import flat_offsets
        offsets, total = flat_offsets(self.params)            # the layout FlatGradSync uses for the same list
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        if grad_flat is None:
            grad_flat = torch.zeros(total, dtype=torch.float32, device=dev)
        assert grad_flat.shape == (total,) and grad_flat.dtype == torch.float32 and grad_flat.device == dev
        self.grad_flat = grad_flat
        self.grad_views: List[torch.Tensor] = []
        self.ranges = []
        k = 0
        with torch.no_grad():
            for g in self.groups:
                start = offsets[k] if g else (offsets[k] if k < len(offsets) else total)
                for p in g:
                    off = offsets[k]
                    view = self.flat.as_strided(p.shape, p.stride(), off)      # dense, non-overlapping: every parameter is
                    view.copy_(p.data)
                    p.data = view
                    self.grad_views.append(grad_flat.as_strided(p.shape, p.stride(), off))
                    k += 1
                end = offsets[k] if k < len(offsets) else total
                self.ranges.append((start, end))          # padding elements are zeros with zero gradients: they stay zero
        # one leaf per group; it shares the group's slice of the flat buffer
        self.leaves: List[nn.Parameter] = [nn.Parameter(self.flat[a:b]) for a, b in self.ranges if b > a]
        self._leaf_ranges = [(a, b) for a, b in self.ranges if b > a]
        self.bind_grads()

    def leaf_of_group(self, i: int) -> Optional[nn.Parameter]:
        a, b = self.ranges[i]
        if b <= a:
            return None
        return self.leaves[self._leaf_ranges.index((a, b))]

    def bind_grads(self) -> None:
        """(Re-)attach every leaf to its slice of the flat gradient buffer (``zero_grad(set_to_none=True)`` drops it)."""
        for leaf, (a, b) in zip(self.leaves, self._leaf_ranges):
            leaf.grad = self.grad_flat[a:b]

    def drop_param_grads(self) -> None:
        """``p.grad = None`` for every parameter: autograd then hands its gradient buffers over without an add."""
        for p in self.params:
            p.grad = None

    @torch.no_grad()
    def pack(self) -> None:
        """Bring the parameters' gradients into the flat buffer (nothing to copy for those that already live there)."""
        have_dst, have_src, missing = [], [], []
        for v, p in zip(self.grad_views, self.params):
            g = p.grad
            if g is None:
                missing.append(v)
            elif g is not v and g.data_ptr() != v.data_ptr():
                have_dst.append(v)
                have_src.append(g)
        if missing:
            torch._foreach_zero_(missing)
        if have_src:
            torch._foreach_copy_(have_dst, have_src)
        self.bind_grads()

    def after_step(self) -> None:
        """The optimizer wrote through the leaves: tell everything keyed on the parameters' version counters."""
        torch.autograd.graph.increment_version(self.params)
