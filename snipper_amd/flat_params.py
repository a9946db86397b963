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


def claim_grad_view(p) -> Optional[torch.Tensor]:
    """The slice of a FlatParameters' flat gradient buffer that belongs to parameter ``p`` (its shape and strides), for a
    backward function that can write the parameter's gradient THERE instead of into a fresh tensor: autograd then adopts the
    view as ``p.grad`` and ``FlatParameters.pack()`` (or the staged all-reduce's pack) has nothing to copy for it (164 MB of
    gradient copies per step otherwise).  Granted at most once per backward pass and only while ``p.grad is None`` -- a second
    gradient of the same parameter (a module used twice, gradient accumulation over micro-batches) must arrive in its own
    tensor, because the accumulation would otherwise add a buffer to itself.  None = use a fresh tensor."""
    owner = getattr(p, "_snipper_flat_owner", None)
    if owner is None or p.grad is not None:
        return None
    flat = owner()
    if flat is None or not flat.direct_grads:
        return None
    i = flat._index.get(id(p))
    if i is None or i in flat._claimed:
        return None
    if flat.grad_guard is not None and not flat.grad_guard(p):      # (e.g. FlatGradSync.slice_is_free: an all-reduce in flight)
        return None
    v = flat.grad_views[i]
    if v.data_ptr() % 16:             # (the weight-gradient kernels store 16-byte pieces; claim only what they can take)
        return None
    flat._claimed.add(i)
    # a fresh alias: autograd adopts a gradient without copying only when nobody else holds the tensor object
    return v.as_strided(v.shape, v.stride(), v.storage_offset())


class FlatParameters:
    # write the big weight gradients straight into the flat buffer (claim_grad_view); SNIPPER_FLAT_DIRECT_GRADS=0 for A/B runs
    direct_grads = __import__("os").environ.get("SNIPPER_FLAT_DIRECT_GRADS", "1") != "0"

    def __init__(self, groups: Sequence[Sequence[nn.Parameter]], grad_flat: Optional[torch.Tensor] = None, grad_guard=None):
        """``grad_guard(p) -> bool``: asked before a backward kernel is allowed to write ``p``'s gradient straight into the flat
        buffer (claim_grad_view); pass ``FlatGradSync.slice_is_free`` when ``grad_flat`` is that object's buffer."""
        self.grad_guard = grad_guard
        self.groups: List[List[nn.Parameter]] = [[p for p in g if p.requires_grad] for g in groups]
        self.params: List[nn.Parameter] = [p for g in self.groups for p in g]
        assert self.params, "no trainable parameters"
        assert len({id(p) for p in self.params}) == len(self.params), "a parameter may belong to one group only"
        dev = self.params[0].device
        assert all(p.device == dev and p.dtype == torch.float32 for p in self.params), "float32 parameters on one device"
        from .grad_sync import flat_offsets
        offsets, total = flat_offsets(self.params)            # the layout FlatGradSync uses for the same list
        self.offsets: List[int] = list(offsets[:len(self.params)])
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        if grad_flat is None:
            grad_flat = torch.zeros(total, dtype=torch.float32, device=dev)
        assert grad_flat.shape == (total,) and grad_flat.dtype == torch.float32 and grad_flat.device == dev
        # (slices are 256-byte aligned relative to the buffer's start: a caller's buffer must start aligned too, or the
        #  kernels that write gradients straight into a slice would be handed misaligned pointers)
        # (CPU allocations are 64-byte aligned and no kernel writes into them; claim_grad_view checks each view besides)
        assert not grad_flat.is_cuda or grad_flat.data_ptr() % 256 == 0, "grad_flat must start on a 256-byte boundary"
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
        import weakref
        self._index = {id(p): i for i, p in enumerate(self.params)}
        self._claimed = set()                                  # parameters whose gradient view was handed out this backward
        ref = weakref.ref(self)
        for p in self.params:
            p._snipper_flat_owner = ref
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
        # a new pass begins: whatever reset the gradients (drop_param_grads, model.zero_grad(set_to_none=True), p.grad = None
        # by hand), the claims of the last pass are spent -- leaving them set would switch the direct path off for good
        self._claimed.clear()

    def drop_param_grads(self) -> None:
        """``p.grad = None`` for every parameter: autograd then hands its gradient buffers over without an add."""
        for p in self.params:
            p.grad = None
        self._claimed.clear()

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


class FlatAdamW(torch.optim.Optimizer):
    """``clip_grad_norm_(params, max_norm)`` + ``torch.optim.AdamW.step()`` (engine.py:74, main.py:201-221) over a
    ``FlatParameters`` as two launches of csrc/adamw_flat.cuh: partial sums of g^2, then one pass over parameter, gradient
    and both moments with the clipping coefficient applied on the fly (PyTorch's own path is 3 norm + 7 small + 3 scale
    + 4 fused AdamW launches on the same three tensors; measured 0.49 -> 0.3 ms per step).  Same arithmetic per element as
    ``torch.optim.AdamW(amsgrad=False)``; the gradient buffer is not rescaled in place (nothing reads it after the step).

    A ``torch.optim.Optimizer``: one param_group per non-empty FlatParameters group (its flat leaf as the only "params"
    entry), so the reference's ``StepLR(optimizer, lr_drop)`` (main.py:222) and any other scheduler drive ``lr`` as usual.
    ``state_dict()`` / ``load_state_dict()`` speak ``torch.optim.AdamW``'s per-parameter format over the MODEL's parameters,
    so the reference's ``checkpoint['optimizer']`` (main.py:236-262) can be produced and resumed from.  That format numbers
    the parameters by POSITION inside the mirrored optimizer's param_groups, and the reference builds those from
    ``model.named_parameters()`` (main.py:201-217) -- an order the flat layout is free to differ from (bench.py puts the
    decoder-side parameters first inside `main` so that they form one all-reduce stage).  ``reference_groups`` = the
    parameter lists of the optimizer being mirrored, in ITS group and parameter order (e.g. ``[d["params"] for d in
    param_dicts]``); ids are mapped through it, whatever the flat layout is.  Without it the flat order itself is used,
    with ``group_order`` = the FlatParameters group behind each mirrored group (the reference lists (main, backbone, slow),
    FlatParameters (main, slow, backbone) -> ``group_order=(0, 2, 1)``): only right when the flat groups list their
    parameters in the mirrored optimizer's order.  CUDA only: there is no CPU fallback -- use ``torch.optim.AdamW`` on
    ``flat.leaves`` there."""

    N_PARTS = 2048

    def __init__(self, flat: "FlatParameters", lrs: Sequence[float], weight_decay: float = 1e-2, betas=(0.9, 0.999),
                 eps: float = 1e-8, group_order: Optional[Sequence[int]] = None,
                 reference_groups: Optional[Sequence[Sequence[nn.Parameter]]] = None):
        assert len(lrs) == len(flat.ranges)      # (a CPU FlatParameters can be built and its state_dict exchanged; step() is HIP only)
        self.flat = flat
        groups = [{"params": [flat.leaf_of_group(i)], "lr": float(lr), "flat_group": i}
                  for i, lr in enumerate(lrs) if flat.leaf_of_group(i) is not None]
        super().__init__(groups, dict(lr=float(lrs[0]), weight_decay=float(weight_decay), betas=tuple(betas), eps=float(eps)))
        self.group_order = tuple(group_order) if group_order is not None else tuple(range(len(flat.ranges)))
        assert sorted(self.group_order) == list(range(len(flat.ranges)))
        self._ref_order = None
        if reference_groups is not None:
            index = {id(p): i for i, p in enumerate(flat.params)}
            group_of, first = {}, 0
            for gi, g in enumerate(flat.groups):
                for i in range(first, first + len(g)):
                    group_of[i] = gi
                first += len(g)
            order, seen = [], set()
            for ref in reference_groups:
                idxs = [index[id(p)] for p in ref if p.requires_grad]          # KeyError: not a parameter of `flat`
                fgs = {group_of[i] for i in idxs}
                if len(fgs) > 1:
                    raise ValueError("FlatAdamW: a reference group spans several FlatParameters groups (one lr each)")
                if idxs:
                    order.append((fgs.pop(), idxs))
                seen.update(idxs)
            if len(seen) != len(flat.params) or sum(len(i) for _, i in order) != len(flat.params):
                raise ValueError("FlatAdamW: reference_groups must list every trainable parameter exactly once")
            if len({fg for fg, _ in order}) != len(order):
                raise ValueError("FlatAdamW: two reference groups map to one FlatParameters group")
            self._ref_order = order
        self.exp_avg = torch.zeros_like(flat.flat)
        self.exp_avg_sq = torch.zeros_like(flat.flat)
        self.partials = torch.zeros(self.N_PARTS, dtype=torch.float32, device=flat.flat.device)
        self.grad_norm = torch.zeros(1, dtype=torch.float32, device=flat.flat.device)    # of the last step, before clipping
        self.step_count = 0

    @torch.no_grad()
    def step(self, max_norm: float = 0.0) -> None:
        """One update; ``max_norm`` > 0 clips the global gradient norm first (``clip_grad_norm_`` semantics)."""
        import ctypes
        from . import _lib
        f = self.flat
        if not f.flat.is_cuda:
            raise RuntimeError("FlatAdamW.step runs on the HIP kernels only (no CPU fallback): use torch.optim.AdamW on "
                               "flat.leaves for a CPU model")
        n = f.flat.numel()
        groups = self.param_groups
        k = len(groups)
        # the kernel takes ONE (beta1, beta2, eps): refuse silently different values (after load_state_dict / a manual edit)
        b1, b2 = groups[0]["betas"]
        eps = groups[0]["eps"]
        for g in groups[1:]:
            if tuple(g["betas"]) != (b1, b2) or g["eps"] != eps:
                raise RuntimeError("FlatAdamW: betas and eps must be the same in every param_group")
        ranges = [f.ranges[g["flat_group"]] for g in groups]
        begin = (ctypes.c_longlong * k)(*[r[0] for r in ranges])
        end = (ctypes.c_longlong * k)(*[r[1] for r in ranges])
        lr = (ctypes.c_float * k)(*[g["lr"] for g in groups])
        wd = (ctypes.c_float * k)(*[g["weight_decay"] for g in groups])
        lib = _lib.load()
        stream = _lib.raw_stream(f.flat.device)
        with _lib.device_guard(f.flat.device):
            parts = None
            if max_norm > 0.0:
                _lib.check(lib.snipper_gradnorm_partials_f32(stream, f.grad_flat.data_ptr(), n, self.partials.data_ptr(),
                                                             self.N_PARTS), "snipper_gradnorm_partials_f32")
                parts = self.partials.data_ptr()
            _lib.check(lib.snipper_adamw_clip_f32(stream, f.flat.data_ptr(), f.grad_flat.data_ptr(), self.exp_avg.data_ptr(),
                                                  self.exp_avg_sq.data_ptr(), n, begin, end, lr, wd, k, b1, b2, eps,
                                                  self.step_count + 1, parts, self.N_PARTS if parts else 0, float(max_norm),
                                                  self.grad_norm.data_ptr()), "snipper_adamw_clip_f32")
        self.step_count += 1          # (only once both launches were accepted: a refused launch must not advance the bias correction)
        f.after_step()

    def zero_grad(self, set_to_none: bool = True) -> None:
        self.flat.drop_param_grads()

    # ---- torch.optim.AdamW's state_dict format over the model's parameters -------------------------------------------
    def _moment_views(self, buf: torch.Tensor):
        f = self.flat
        return [buf.as_strided(p.shape, p.stride(), off) for p, off in zip(f.params, f.offsets)]

    def _ordered_groups(self):
        """[(FlatParameters group, [index in FlatParameters.params of every parameter, in the MIRRORED optimizer's order])]
        group by group in the mirrored optimizer's group order."""
        if self._ref_order is not None:
            return list(self._ref_order)
        f = self.flat
        first = [0]
        for g in f.groups:
            first.append(first[-1] + len(g))
        return [(i, list(range(first[i], first[i + 1]))) for i in self.group_order]

    def state_dict(self):
        ea, es = self._moment_views(self.exp_avg), self._moment_views(self.exp_avg_sq)
        by_flat_group = {g["flat_group"]: g for g in self.param_groups}
        state, groups, k = {}, [], 0
        for fg, idxs in self._ordered_groups():
            if not idxs:
                continue
            g = by_flat_group[fg]
            groups.append({"lr": g["lr"], "betas": tuple(g["betas"]), "eps": g["eps"], "weight_decay": g["weight_decay"],
                           "amsgrad": False, "maximize": False, "foreach": None, "capturable": False, "differentiable": False,
                           "fused": None, **({"initial_lr": g["initial_lr"]} if "initial_lr" in g else {}),
                           "params": list(range(k, k + len(idxs)))})
            for i in idxs:
                if self.step_count > 0:           # (torch's AdamW has no state for a parameter before its first step)
                    state[k] = {"step": torch.tensor(float(self.step_count)), "exp_avg": ea[i].clone(),
                                "exp_avg_sq": es[i].clone()}
                k += 1
        return {"state": state, "param_groups": groups}

    @torch.no_grad()
    def load_state_dict(self, sd) -> None:
        ea, es = self._moment_views(self.exp_avg), self._moment_views(self.exp_avg_sq)
        by_flat_group = {g["flat_group"]: g for g in self.param_groups}
        order = [(fg, idxs) for fg, idxs in self._ordered_groups() if idxs]
        if len(sd["param_groups"]) != len(order):
            raise ValueError("FlatAdamW.load_state_dict: number of param_groups differs")
        steps = set()
        self.exp_avg.zero_(); self.exp_avg_sq.zero_()
        for (fg, idxs), sg in zip(order, sd["param_groups"]):
            if len(sg["params"]) != len(idxs):
                raise ValueError("FlatAdamW.load_state_dict: a param_group has a different number of parameters")
            g = by_flat_group[fg]
            for key in ("lr", "betas", "eps", "weight_decay", "initial_lr"):
                if key in sg:
                    g[key] = tuple(sg[key]) if key == "betas" else sg[key]
            for i, pid in zip(idxs, sg["params"]):
                st = sd["state"].get(pid)
                if st is None:
                    steps.add(0)
                    continue
                if tuple(st["exp_avg"].shape) != tuple(ea[i].shape):
                    raise ValueError(f"FlatAdamW.load_state_dict: state {pid} has shape {tuple(st['exp_avg'].shape)}, the "
                                     f"parameter at that position {tuple(ea[i].shape)} (other parameter order?)")
                ea[i].copy_(st["exp_avg"]); es[i].copy_(st["exp_avg_sq"])
                steps.add(int(float(st["step"])))
        if len(steps) > 1:
            raise ValueError(f"FlatAdamW.load_state_dict: parameters at different steps {sorted(steps)} (one bias correction)")
        self.step_count = steps.pop() if steps else 0
