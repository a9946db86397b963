"""Data-parallel gradient synchronisation for the training step: one flat buffer, a few large all-reduces.

The reference trains with ``torch.nn.parallel.DistributedDataParallel`` (main.py:193-196).  DDP's reducer hooks every
parameter's gradient accumulator; with the ~330 parameter tensors of this model that is ~15 ms of host work per step
(measured on MI355X: the step goes from GPU-bound, 49 ms, to host-bound, 57 ms).  The semantics the path needs are
just "after backward every rank holds the mean of the ranks' gradients", so this module does exactly that:

  * all parameters that require grad own a slice of ONE flat float32 buffer;
  * after ``loss.backward()`` (grads produced with ``zero_grad(set_to_none=True)``, so autograd hands its buffers
    over without an accumulation kernel per parameter), ``sync()`` packs the gradients into the flat buffer with
    multi-tensor copies, all-reduces it in a few large chunks (RCCL rings over xGMI are per-link bound: few, large
    messages), scales by 1/world and re-points every ``p.grad`` at its slice -- views with the parameter's own
    strides (NHWC convolution weights stay NHWC);
  * ``broadcast_parameters()`` makes rank 0's initial weights everybody's, as DDP's constructor does.

Parameters that took no part in the step (``grad is None``) contribute zeros, like DDP with static_graph.

Overlap: backward reaches the backbone last (~1/4 of the step's GPU time, >half of the gradient bytes still to come).
If ``early`` names the parameters whose gradients are complete before that (everything but the backbone) and
``trigger`` the ones whose gradients arrive last among them (the 1x1 input projections between backbone and
encoder), a post-accumulate hook on the trigger parameters packs and all-reduces the early slice while the backbone's
backward is still running; ``sync()`` then only has the backbone's slice left.  The early launch happens as soon as the
last trigger gradient is complete, on every rank alike (the same collectives in the same order); autograd's ordering
makes most early parameters complete by then (their nodes were created after the projections'); one that is completed
later (e.g. ``level_embed`` on the token-row path, whose gradient also collects the projections' own contribution) is
detected in ``sync()`` -- its gradient object differs from what was packed -- and reduced there.
"""
from __future__ import annotations

import os
from typing import Iterable, List

import torch
import torch.distributed as dist


FLAT_ALIGN = 64       # elements (256 B): every tensor's slice of a flat buffer starts on a cache-line boundary


def flat_offsets(params, align: int = FLAT_ALIGN):
    """Start offset (elements) of every tensor in a flat buffer and the buffer's size; slices are padded to ``align``
    (an unaligned weight pointer takes the slow path of vectorised kernels and of the BLAS library's heuristics)."""
    offs, off = [], 0
    for p in params:
        offs.append(off)
        off += (p.numel() + align - 1) // align * align
    return offs, off


def _dense_view(flat: torch.Tensor, offset: int, like: torch.Tensor) -> torch.Tensor:
    """A view of ``flat[offset : offset + like.numel()]`` with ``like``'s shape and strides (``like`` must be dense and
    non-overlapping, which every parameter is)."""
    return flat.as_strided(like.shape, like.stride(), offset)


class FlatGradSync:
    def __init__(self, params: Iterable[torch.nn.Parameter], group=None, chunks: int = 4,
                 early: Iterable[torch.nn.Parameter] = (), trigger: Iterable[torch.nn.Parameter] = ()):
        params = [p for p in params if p.requires_grad]
        early_ids = {id(p) for p in early if p.requires_grad}
        # flat layout: [early | rest]
        self.params: List[torch.nn.Parameter] = [p for p in params if id(p) in early_ids] + \
                                                [p for p in params if id(p) not in early_ids]
        self.n_early = sum(1 for p in params if id(p) in early_ids)
        assert self.params, "no trainable parameters"
        dev, dt = self.params[0].device, torch.float32
        assert all(p.device == dev and p.dtype == dt for p in self.params), "float32 parameters on one device expected"
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # SNIPPER_SYNC_FORCE=1: run hooks and collectives even in a 1-rank group (exercises the path on a single GPU)
        self.collective = self.world > 1 or (dist.is_initialized() and os.environ.get("SNIPPER_SYNC_FORCE") == "1")
        self.offsets, total = flat_offsets(self.params)
        self.flat = torch.zeros(total, dtype=dt, device=dev)
        self.views = [_dense_view(self.flat, off, p) for off, p in zip(self.offsets, self.params)]
        ends = self.offsets[1:] + [total]                      # (padded) end of every parameter's slice
        # chunk boundaries on parameter boundaries: contiguous slices of the flat buffer
        self.chunks, start, target = [], 0, (total + chunks - 1) // max(1, chunks)
        for end in ends:
            if end - start >= target:
                self.chunks.append((start, end))
                start = end
        if start < total:
            self.chunks.append((start, total))
        self.early_elems = self.offsets[self.n_early] if self.n_early < len(self.params) else total
        self._early_works, self._early_done, self._early_seen = [], False, []
        self._trigger = [p for p in trigger if p.requires_grad]
        self._pending = len(self._trigger)
        if self.n_early and self._trigger and self.collective:
            for p in self._trigger:
                p.register_post_accumulate_grad_hook(self._on_trigger)

    def _pack(self, lo: int, hi: int) -> None:
        have = [(v, p.grad) for v, p in zip(self.views[lo:hi], self.params[lo:hi])
                if p.grad is not None and p.grad is not v]
        missing = [v for v, p in zip(self.views[lo:hi], self.params[lo:hi]) if p.grad is None]
        if missing:
            torch._foreach_zero_(missing)
        if have:
            torch._foreach_copy_([v for v, _ in have], [g for _, g in have])

    def _reduce(self, lo_elem: int, hi_elem: int):
        works = []
        for a, b in self.chunks:
            a, b = max(a, lo_elem), min(b, hi_elem)
            if a < b:
                works.append(dist.all_reduce(self.flat[a:b], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        return works

    @torch.no_grad()
    def _on_trigger(self, _param) -> None:
        """Runs on the autograd thread when one trigger parameter's gradient is complete."""
        self._pending -= 1
        if self._pending > 0 or self._early_done:
            return
        # Taken unconditionally once the last trigger has fired, so that every rank issues the same collectives in the
        # same order whatever its batch looked like; an early parameter without a gradient contributes zeros.
        self._early_seen = [p.grad for p in self.params[:self.n_early]]      # what was packed (None = zeros)
        self._pack(0, self.n_early)
        self._early_works = self._reduce(0, self.early_elems)
        self._early_done = True

    @torch.no_grad()
    def broadcast_parameters(self, modules_or_tensors: Iterable[torch.Tensor], src: int = 0) -> None:
        if self.world == 1:
            return
        tensors = list(modules_or_tensors)
        for t in tensors:
            # (detach() shares the parameter's version counter, .data does not: shadow.py keys its bf16 copies on it)
            dist.broadcast(t.detach() if isinstance(t, torch.nn.Parameter) else t, src, group=self.group)

    @torch.no_grad()
    def sync(self) -> None:
        """Mean of the ranks' gradients into every ``p.grad`` (collective: every rank must call it)."""
        first = self.n_early if self._early_done else 0
        self._pack(first, len(self.params))
        late = []
        if self._early_done:
            # an "early" gradient that was completed (or replaced) after the early launch -- a property of the graph,
            # hence the same on every rank: its slice is packed and reduced now
            for i, p in enumerate(self.params[:self.n_early]):
                if p.grad is not None and p.grad is not self._early_seen[i] and p.grad is not self.views[i]:
                    self.views[i].copy_(p.grad)
                    late.append((self.offsets[i], self.offsets[i] + p.numel()))
            self._early_seen = []
        if self.collective:
            works = self._early_works + self._reduce(self.early_elems if self._early_done else 0, self.flat.numel())
            works += [dist.all_reduce(self.flat[a:b], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                      for a, b in late]
            for w in works:
                w.wait()
            self.flat.mul_(1.0 / self.world)
        for v, p in zip(self.views, self.params):
            p.grad = v
        self._early_works, self._early_done, self._pending = [], False, len(self._trigger)
