"""Data-parallel gradient synchronisation for the training step: one flat buffer, a few large all-reduces,
launched stage by stage while backward is still running.

The reference trains with ``torch.nn.parallel.DistributedDataParallel`` (main.py:193-196).  DDP's reducer hooks every
parameter's gradient accumulator; with the ~330 parameter tensors of this model that is ~15 ms of host work per step
(measured on MI355X: the step goes from GPU-bound, 49 ms, to host-bound, 57 ms).  The semantics the path needs are
just "after backward every rank holds the mean of the ranks' gradients", so this module does exactly that:

  * all parameters that require grad own a slice of ONE flat float32 buffer, in the order they are given;
  * after ``loss.backward()`` (grads produced with ``zero_grad(set_to_none=True)``, so autograd hands its buffers
    over without an accumulation kernel per parameter), ``sync()`` packs the gradients into the flat buffer with
    multi-tensor copies, all-reduces what is still outstanding, scales by 1/world and re-points every ``p.grad`` at its
    slice -- views with the parameter's own strides (NHWC convolution weights stay NHWC);
  * ``broadcast_parameters()`` makes rank 0's initial weights everybody's, as DDP's constructor does.

Parameters that took no part in the step (``grad is None``) contribute zeros, like DDP with static_graph.

Overlap.  ``stages`` lists contiguous runs of the parameter list in the order backward completes them, each with its
``trigger`` parameters -- the ones whose gradients arrive LAST within the stage (bench.py: everything but the backbone,
triggered by the 1x1 input projections; then layer4 / layer3 / layer2 of the ResNet, each triggered by its first
block).  A post-accumulate hook on the trigger parameters packs the stage's slice and launches its all-reduce as soon
as the stage's last trigger gradient is complete, while backward goes on with the next stage.  The launch is taken
unconditionally, on every rank alike (the same collectives in the same order whatever the batch looked like); a
parameter of the stage whose gradient is completed or replaced AFTER the launch (e.g. ``level_embed`` on the token-row
path, whose gradient also collects the projections' own contribution) is detected in ``sync()`` by object identity -- a
property of the graph, hence the same on every rank -- and its slice is reduced again there, strictly after the first
all-reduce of that slice has completed.

Contract: ONE backward per ``sync()``, and gradients reset with ``set_to_none=True`` (or ``FlatParameters.
drop_param_grads()``) in between.  A second backward before ``sync()`` or a gradient that still is last step's view of
the flat buffer would silently corrupt a slice that is being all-reduced, so both raise.

Deferred error.  A gradient that arrives late on ONE rank outside the agreed late set cannot raise there and then (the
other ranks would hang in their next collective): the step completes everywhere and every rank raises together from its
NEXT ``sync()`` / ``check_errors()``.  The optimizer step that followed the flagged ``sync()`` has therefore used a wrong
(stale-reduced) gradient for that slice: call ``check_errors()`` after the last step of a run and before saving a
checkpoint, and treat a raise as "the previous step's update is not to be trusted".
"""
from __future__ import annotations

import os
from typing import Iterable, List, Optional, Sequence, Tuple

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


class _DoneEvent:          # CPU tensors (gloo tests): the copy above is synchronous
    def synchronize(self) -> None:
        pass


class _Stage:
    __slots__ = ("lo", "hi", "lo_elem", "hi_elem", "triggers", "pending", "ready", "launched", "works", "seen")

    def __init__(self, lo, hi, lo_elem, hi_elem, triggers):
        self.lo, self.hi, self.lo_elem, self.hi_elem = lo, hi, lo_elem, hi_elem
        self.triggers = triggers
        self.pending, self.ready, self.launched, self.works, self.seen = len(triggers), False, False, [], []


class FlatGradSync:
    def __init__(self, params: Iterable[torch.nn.Parameter], group=None, chunks: int = 4,
                 early: Iterable[torch.nn.Parameter] = (), trigger: Iterable[torch.nn.Parameter] = (),
                 stages: Optional[Sequence[Tuple[Sequence[torch.nn.Parameter], Sequence[torch.nn.Parameter]]]] = None,
                 late: Optional[Iterable[torch.nn.Parameter]] = None):
        """``stages``: [(parameters, trigger parameters), ...] in the order backward completes them; every stage must
        be a contiguous run of ``params``.  ``early`` / ``trigger`` is the one-stage shorthand; for it alone the early
        parameters are moved to the front of the flat layout.  ``late``: parameters of a stage whose gradient is only
        complete after the stage's triggers have fired (re-reduced by sync() on every rank); None = agreed on at the first
        sync()."""
        params = [p for p in params if p.requires_grad]
        if stages is None:
            early_ids = {id(p) for p in early if p.requires_grad}
            params = [p for p in params if id(p) in early_ids] + [p for p in params if id(p) not in early_ids]
            stages = [([p for p in params if id(p) in early_ids], list(trigger))] if early_ids else []
        self.params: List[torch.nn.Parameter] = params
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
        self._view_ids = {id(v) for v in self.views}
        self.max_chunk = (total + max(1, chunks) - 1) // max(1, chunks)
        index = {id(p): i for i, p in enumerate(self.params)}
        self.stages: List[_Stage] = []
        covered = set()
        for sp, trig in stages:
            idx = sorted(index[id(p)] for p in sp if p.requires_grad)
            if not idx:
                continue
            assert idx == list(range(idx[0], idx[-1] + 1)), "a stage must be a contiguous run of the parameter list"
            assert not (covered & set(idx)), "stages overlap"
            covered |= set(idx)
            hi_elem = self.offsets[idx[-1] + 1] if idx[-1] + 1 < len(self.params) else total
            trig = [p for p in trig if p.requires_grad]
            assert all(id(p) in index for p in trig)
            self.stages.append(_Stage(idx[0], idx[-1] + 1, self.offsets[idx[0]], hi_elem, trig))
        # whatever no stage covers is reduced by sync(): consecutive uncovered parameters form ONE run (lo, hi), packed by
        # one multi-tensor copy and all-reduced in max_chunk-sized messages -- few, large messages, not one per parameter
        self.rest: List[Tuple[int, int]] = []
        for i in range(len(self.params)):
            if i in covered:
                continue
            if self.rest and self.rest[-1][1] == i:
                self.rest[-1] = (self.rest[-1][0], i + 1)
            else:
                self.rest.append((i, i + 1))
        self.trace: Optional[list] = None          # set to [] to record (stage index, event on the compute stream) per launch
        # "a gradient arrived late on SOME rank outside the agreed set": MAX-all-reduced with every sync() and read at the NEXT
        # one (through pinned memory and an event of its own: no wait on the step in flight), so that all ranks raise together
        self._err_dev = self._err_host = self._err_event = None
        self._err_text = ""
        self._index = index
        self.late_idx: Optional[List[int]] = None if late is None else sorted(index[id(p)] for p in late if id(p) in index)
        if not self.stages:
            self.late_idx = []                    # nothing is launched before sync(): nothing can arrive late
        self.n_early = self.stages[0].hi if self.stages and self.stages[0].lo == 0 else 0     # (kept for callers / tests)
        self.early_elems = self.stages[0].hi_elem if self.n_early else 0
        if self.collective:
            for st in self.stages:
                for p in st.triggers:
                    p.register_post_accumulate_grad_hook(lambda _p, st=st: self._on_trigger(st))

    def slice_is_free(self, p: torch.nn.Parameter) -> bool:
        """May a backward kernel write ``p``'s gradient straight into its slice of the flat buffer right now
        (flat_params.claim_grad_view with this buffer as the FlatParameters' ``grad_flat``)?  Not once the slice's stage is
        ready or launched -- an all-reduce may be reading and writing it: such a (late) gradient must arrive in a tensor of
        its own, so that ``sync()`` sees it and reduces it again -- and never for the agreed late set."""
        if self.late_idx is None:                  # (the late set is agreed on at the first sync(): no direct writes before)
            return False
        i = self._index.get(id(p))
        if i is None or i in self.late_idx:
            return False
        for st in self.stages:
            if st.lo <= i < st.hi:
                return not (st.ready or st.launched)
        return True

    # ------------------------------------------------------------------------------------------------------------
    def _pack(self, lo: int, hi: int) -> None:
        stale = [i for i in range(lo, hi) if self.params[i].grad is not None and id(self.params[i].grad) in self._view_ids]
        if stale:
            raise RuntimeError(
                "FlatGradSync: a parameter's .grad still is last step's view of the flat buffer (autograd accumulated into "
                "it in place). Reset gradients with set_to_none=True / FlatParameters.drop_param_grads() before backward.")
        # (a gradient that a backward kernel wrote straight into its slice -- flat_params.claim_grad_view with this buffer as
        #  the FlatParameters' grad_flat -- is already where it belongs)
        have = [(v, p.grad) for v, p in zip(self.views[lo:hi], self.params[lo:hi])
                if p.grad is not None and p.grad.data_ptr() != v.data_ptr()]
        missing = [v for v, p in zip(self.views[lo:hi], self.params[lo:hi]) if p.grad is None]
        if missing:
            torch._foreach_zero_(missing)
        if have:
            torch._foreach_copy_([v for v, _ in have], [g for _, g in have])

    def _reduce(self, lo_elem: int, hi_elem: int):
        works, a = [], lo_elem
        while a < hi_elem:                       # few, large messages: RCCL rings over xGMI are per-link bound
            b = min(hi_elem, a + self.max_chunk)
            works.append(dist.all_reduce(self.flat[a:b], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            a = b
        return works

    @torch.no_grad()
    def _on_trigger(self, st: _Stage) -> None:
        """Runs on the autograd thread when one trigger parameter's gradient is complete."""
        if st.launched or st.ready or st.pending <= 0:
            # (ready but not launched = waiting for an earlier stage: a second backward would re-mark it and the stage would
            #  later be packed from ACCUMULATED gradients -- ADVICE r03)
            raise RuntimeError("FlatGradSync: a second backward reached a stage whose triggers have already fired; "
                               "call sync() after every backward (no gradient accumulation across backwards)")
        st.pending -= 1
        if st.pending > 0:
            return
        # Collectives are issued strictly in STAGE ORDER, on every rank alike: a stage whose triggers have all fired is
        # launched only once every earlier stage has been (a rank on which a trigger parameter got no gradient this step
        # -- an unused branch of the loss for its batch -- never sees that stage's hook; it and the stages after it are
        # then issued by sync(), in the same order the other ranks used).  A parameter without a gradient contributes zeros.
        st.ready = True
        for s2 in self.stages:
            if s2.launched:
                continue
            if not s2.ready:
                break
            self._launch(s2)

    def _launch(self, st: _Stage) -> None:
        if self.trace is not None and self.flat.is_cuda:               # where in the GPU timeline the stage was launched
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            self.trace.append((self.stages.index(st), ev))
        st.seen = [p.grad for p in self.params[st.lo:st.hi]]          # what was packed (None = zeros)
        self._pack(st.lo, st.hi)
        st.works = self._reduce(st.lo_elem, st.hi_elem) if self.collective else []
        st.launched = True

    @torch.no_grad()
    def broadcast_parameters(self, modules_or_tensors: Iterable[torch.Tensor], src: int = 0) -> None:
        if self.world == 1:
            return
        tensors = list(modules_or_tensors)
        for t in tensors:
            # (detach() shares the parameter's version counter, .data does not: shadow.py keys its bf16 copies on it)
            dist.broadcast(t.detach() if isinstance(t, torch.nn.Parameter) else t, src, group=self.group)

    def check_errors(self) -> None:
        """Raise, on EVERY rank, if any rank saw an unexpected late gradient at the previous sync() (called by sync();
        call it once more after the last step of a run)."""
        if self._err_event is None:
            return
        self._err_event.synchronize()
        self._err_event = None
        if int(self._err_host[0]) != 0:
            text, self._err_text = self._err_text, ""            # (reported once: a later, unrelated error must not quote it)
            raise RuntimeError(
                "FlatGradSync: on at least one rank a gradient was completed after its stage's all-reduce had been launched "
                "and is not in the agreed set of late parameters (pass it in `late=`); that rank's slice was reduced from a "
                "stale value in the previous step, and the optimizer step that followed used it" + (": " + text if text else ""))

    @torch.no_grad()
    def sync(self) -> None:
        """Mean of the ranks' gradients into every ``p.grad`` (collective: every rank must call it)."""
        self.check_errors()
        works = []
        # 1) everything that was not launched from a hook: pack and reduce now (stage by stage, then the uncovered rest)
        for st in self.stages:
            if not st.launched:
                self._launch(st)
        for lo, hi in self.rest:
            self._pack(lo, hi)
            if self.collective:
                hi_elem = self.offsets[hi] if hi < len(self.params) else self.flat.numel()
                works += self._reduce(self.offsets[lo], hi_elem)
        # 2) wait for EVERY in-flight all-reduce before touching a slice again: the collective runs on the process
        #    group's own stream, so a late gradient copied into a slice that is still being reduced would race with it
        for st in self.stages:
            works += st.works
        for w in works:
            w.wait()
        # 3) gradients completed (or replaced) after their stage's launch: their slices hold a reduced stale value; put the
        #    complete local gradient there and reduce those slices again.  WHICH parameters get that second all-reduce
        #    must be the same on every rank (a rank whose batch left a trigger parameter unused launches its stages from
        #    sync() and sees nothing arrive late): the set is fixed -- given as ``late=`` or agreed on at the first sync()
        #    by one MAX all-reduce over per-parameter flags (the only host synchronisation this class ever makes) -- and
        #    from then on every rank re-packs and re-reduces exactly those slices, late locally or not.
        local_late = []
        for st in self.stages:
            for i, p in zip(range(st.lo, st.hi), self.params[st.lo:st.hi]):
                g = p.grad
                if g is not None and g is not st.seen[i - st.lo]:
                    if id(g) in self._view_ids:
                        raise RuntimeError("FlatGradSync: .grad is a stale view of the flat buffer (see _pack)")
                    local_late.append(i)
        if self.collective:
            if self.late_idx is None:
                flags = torch.zeros(len(self.params), dtype=torch.int32, device=self.flat.device)
                if local_late:
                    flags[torch.tensor(local_late, device=flags.device)] = 1
                dist.all_reduce(flags, op=dist.ReduceOp.MAX, group=self.group)
                self.late_idx = [int(i) for i in flags.nonzero().flatten().tolist()]
            # A locally late gradient OUTSIDE the agreed set: raising here would leave the other ranks blocked in their next
            # all-reduce until the process-group timeout.  This rank keeps the collective order, flags the step,
            # and every rank raises from its next sync() / check_errors().
            unexpected = sorted(set(local_late) - set(self.late_idx))
            err_work = None
            if self.stages and self._err_dev is None:
                self._err_dev = torch.zeros(1, dtype=torch.int32, device=self.flat.device)
                self._err_host = torch.zeros(1, dtype=torch.int32, pin_memory=self.flat.is_cuda)
            if self.stages:                       # (without stages nothing is launched before sync(): nothing can be late)
                self._err_dev.fill_(1 if unexpected else 0)
                if unexpected:
                    self._err_text = f"parameter #{unexpected[0]} on rank {dist.get_rank(self.group)}"
                err_work = dist.all_reduce(self._err_dev, op=dist.ReduceOp.MAX, group=self.group, async_op=True)
            runs, prev = [], None
            for i in self.late_idx:               # re-pack: the complete local gradient (zeros if there is none)
                g = self.params[i].grad
                if g is None:
                    self.views[i].zero_()
                else:
                    self.views[i].copy_(g)
                a, b = self.offsets[i], self.offsets[i] + self.params[i].numel()
                if prev is not None and i == prev + 1:          # neighbours in the flat layout: one message
                    runs[-1] = (runs[-1][0], b)
                else:
                    runs.append((a, b))
                prev = i
            for w in [dist.all_reduce(self.flat[a:b], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                      for a, b in runs]:
                w.wait()
            if err_work is not None:
                err_work.wait()
                self._err_host.copy_(self._err_dev, non_blocking=True)
                if self.flat.is_cuda:
                    self._err_event = torch.cuda.Event()
                    self._err_event.record()
                else:
                    self._err_event = _DoneEvent()
            if self.world > 1:                    # (a forced 1-rank group -- the single-GPU test of this path -- has nothing to average)
                self.flat.mul_(1.0 / self.world)
        else:
            for i in local_late:
                self.views[i].copy_(self.params[i].grad)
        for v, p in zip(self.views, self.params):
            p.grad = v
        for st in self.stages:
            st.pending, st.ready, st.launched, st.works, st.seen = len(st.triggers), False, False, [], []

    # (tests look at this)
    @property
    def _early_done(self) -> bool:
        return bool(self.stages) and self.stages[0].launched
