"""FlatParameters (snipper_amd/flat_params.py): torch.optim.AdamW + clip_grad_norm_ on one flat tensor per group give
the updates of the reference's per-parameter form (main.py:201-221, engine.py:74)."""
import copy

import torch
from torch import nn

from snipper_amd.flat_params import FlatParameters


def _model():
    torch.manual_seed(0)
    m = nn.Sequential(nn.Conv2d(3, 8, 3, padding=1), nn.ReLU(), nn.Conv2d(8, 4, 1), nn.Flatten(), nn.Linear(4 * 6 * 6, 5))
    return m.to(memory_format=torch.channels_last)          # NHWC weights: strides must survive the re-homing


def _groups(m):
    ps = list(m.parameters())
    return [ps[:2], ps[2:4], ps[4:]]


def test_flat_adamw_and_clipping_equal_the_per_parameter_form():
    ref, new = _model(), _model()
    lrs = [1e-2, 1e-3, 1e-3]
    opt_ref = torch.optim.AdamW([{"params": g, "lr": lr} for g, lr in zip(_groups(ref), lrs)], lr=1e-2, weight_decay=1e-2)
    strides = [p.stride() for p in new.parameters()]
    fp = FlatParameters(_groups(new))
    assert [p.stride() for p in new.parameters()] == strides
    assert all(torch.equal(a, b) for a, b in zip(ref.parameters(), new.parameters()))
    opt_new = torch.optim.AdamW([{"params": [fp.leaf_of_group(i)], "lr": lr} for i, lr in enumerate(lrs)], lr=1e-2,
                                weight_decay=1e-2)
    g = torch.Generator().manual_seed(1)
    for step in range(4):
        x = torch.randn(7, 3, 6, 6, generator=g).contiguous(memory_format=torch.channels_last)
        y = torch.randn(7, 5, generator=g)
        opt_ref.zero_grad(set_to_none=True)
        (ref(x) - y).pow(2).sum().backward()
        n_ref = torch.nn.utils.clip_grad_norm_(ref.parameters(), 0.1)
        opt_ref.step()

        versions = [p._version for p in new.parameters()]
        fp.drop_param_grads()
        (new(x) - y).pow(2).sum().backward()
        fp.pack()
        n_new = torch.nn.utils.clip_grad_norm_(fp.leaves, 0.1)
        opt_new.step()
        fp.after_step()
        assert all(p._version > v for p, v in zip(new.parameters(), versions))     # shadow.py keys on this
        torch.testing.assert_close(n_new, n_ref, rtol=1e-5, atol=0)
        for a, b in zip(ref.parameters(), new.parameters()):
            torch.testing.assert_close(b, a, rtol=1e-5, atol=1e-7)


def test_missing_gradients_count_as_zero_and_external_gradient_buffer():
    from snipper_amd.grad_sync import FLAT_ALIGN, flat_offsets
    m = _model()
    offsets, total = flat_offsets(list(m.parameters()))
    assert all(o % FLAT_ALIGN == 0 for o in offsets) and total % FLAT_ALIGN == 0
    ext = torch.zeros(total)                                 # (FlatGradSync's buffer: the padding between slices is 0)
    fp = FlatParameters(_groups(m), grad_flat=ext)
    assert all(p.data_ptr() % (4 * FLAT_ALIGN) == fp.flat.data_ptr() % (4 * FLAT_ALIGN) for p in m.parameters())
    for v in fp.grad_views:
        v.fill_(7.0)
    assert fp.grad_flat is ext and all(leaf.grad is not None for leaf in fp.leaves)
    ps = list(m.parameters())
    ps[0].grad = torch.ones_like(ps[0])
    ps[1].grad = fp.grad_views[1]                       # already lives in the flat buffer (FlatGradSync's case)
    fp.grad_views[1].fill_(3.0)
    fp.pack()
    assert torch.equal(fp.grad_views[0], torch.ones_like(ps[0])) and torch.all(fp.grad_views[1] == 3.0)
    assert all(torch.all(v == 0) for v in fp.grad_views[2:])
