"""The few helpers of the reference's util/misc.py that sit on the hot path's boundary:
``NestedTensor`` (:333-352), ``nested_tensor_from_tensor_list`` (:310-330) and ``collate_fn`` (:295-298).
"""
from __future__ import annotations

from typing import List, Optional

import torch
from torch import Tensor


class NestedTensor(object):
    def __init__(self, tensors, mask: Optional[Tensor]):
        self.tensors = tensors
        self.mask = mask

    def to(self, device):
        return NestedTensor(self.tensors.to(device), None if self.mask is None else self.mask.to(device))

    def decompose(self):
        return self.tensors, self.mask

    def __repr__(self):
        return str(self.tensors)


class BoundedCache(dict):
    """A dict that keeps at most ``limit`` entries (least recently inserted / touched first out).  The geometry caches
    (masks, position encodings, level tensors, reference grids) are keyed by shapes and batch size: with variable input
    sizes -- multi-scale training, evaluation on differently sized videos -- an unbounded dict would pin device memory."""

    def __init__(self, limit: int = 16):
        super().__init__()
        self.limit = limit

    def get(self, key, default=None):
        if key in self:
            val = super().pop(key)
            super().__setitem__(key, val)          # touched: most recent
            return val
        return default

    def __setitem__(self, key, val):
        if key in self:
            super().pop(key)
        super().__setitem__(key, val)
        while len(self) > self.limit:
            super().pop(next(iter(self)))


_NO_PAD_MASKS = BoundedCache(16)


def no_padding_mask(n: int, h: int, w: int, device) -> Tensor:
    """An all-False [n, h, w] padding mask that says so on the host (attribute ``_snipper_all_false``): everything that
    is a function of the mask alone -- the resized masks, the position encoding, the valid ratios, the encoder's
    reference grid -- is then a constant of the shapes and is built once instead of ~150 small launches per step.
    Shared and read-only; a copy (``.to``, ``.clone``) drops the attribute and with it the shortcut."""
    if torch.is_inference_mode_enabled():            # inference tensors must not end up in a cache that training reads
        return torch.zeros((n, h, w), dtype=torch.bool, device=device)
    key = (n, h, w, str(device))
    m = _NO_PAD_MASKS.get(key)
    if m is None:
        m = torch.zeros((n, h, w), dtype=torch.bool, device=device)
        m._snipper_all_false = True
        _NO_PAD_MASKS[key] = m
    return m


def is_no_padding(mask) -> bool:
    return getattr(mask, "_snipper_all_false", False)


def nested_tensor_from_tensor_list(tensor_list: List[Tensor], split=True) -> NestedTensor:
    """Each snippet arrives as [T*3, H, W]; it is split into T images of [3, H, W], all images are
    zero-padded to the largest H, W in the batch, and the mask is True on padding."""
    if split:
        tensor_list = [img for snippet in tensor_list for img in snippet.split(3, dim=0)]
    if tensor_list[0].ndim != 3:
        raise ValueError('not supported')
    c = max(t.shape[0] for t in tensor_list)
    h = max(t.shape[1] for t in tensor_list)
    w = max(t.shape[2] for t in tensor_list)
    if all(tuple(t.shape) == (c, h, w) for t in tensor_list):     # the training case: nothing to pad
        batch = torch.stack(list(tensor_list))
        return NestedTensor(batch, no_padding_mask(len(tensor_list), h, w, batch.device))
    ref = tensor_list[0]
    batch = torch.zeros((len(tensor_list), c, h, w), dtype=ref.dtype, device=ref.device)
    mask = torch.ones((len(tensor_list), h, w), dtype=torch.bool, device=ref.device)
    for img, slot, m in zip(tensor_list, batch, mask):
        slot[: img.shape[0], : img.shape[1], : img.shape[2]].copy_(img)
        m[: img.shape[1], : img.shape[2]] = False
    return NestedTensor(batch, mask)


def collate_fn(batch):
    batch = list(zip(*batch))
    batch[0] = nested_tensor_from_tensor_list(batch[0])
    return tuple(batch)
