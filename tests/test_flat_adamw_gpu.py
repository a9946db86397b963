"""FlatAdamW (snipper_amd/flat_params.py, csrc/adamw_flat.cuh), ``-m gpu``: global-norm clipping + AdamW on the flat
buffers against torch.optim.AdamW + clip_grad_norm_ on the same parameters (engine.py:74, main.py:201-221)."""
import pytest
import torch
from torch import nn

from snipper_amd.flat_params import FlatAdamW, FlatParameters

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _model(seed=0):
    torch.manual_seed(seed)
    m = nn.Sequential(nn.Conv2d(3, 24, 3, padding=1), nn.ReLU(), nn.Conv2d(24, 12, 1), nn.Flatten(), nn.Linear(12 * 9 * 9, 37),
                      nn.ReLU(), nn.Linear(37, 5))
    return m.to(DEV).to(memory_format=torch.channels_last)


def _groups(m):
    ps = list(m.parameters())
    return [ps[:2], ps[2:4], ps[4:]]


@pytest.mark.parametrize("max_norm", [0.1, 1e6, 0.0])
def test_flat_adamw_equals_torch_adamw_with_clipping(max_norm):
    """Clipping active (0.1, the reference's value), inactive (norm below the bound) and off; three groups with their
    own learning rates; 6 steps, so the bias corrections and both moments are exercised."""
    ref, new = _model(), _model()
    lrs = [1e-2, 1e-3, 3e-3]
    opt_ref = torch.optim.AdamW([{"params": g, "lr": lr} for g, lr in zip(_groups(ref), lrs)], lr=1e-2, weight_decay=1e-2,
                                fused=False, foreach=False)
    fp = FlatParameters(_groups(new))
    opt_new = FlatAdamW(fp, lrs, weight_decay=1e-2)
    g = torch.Generator().manual_seed(3)
    for step in range(6):
        x = torch.randn(5, 3, 9, 9, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
        y = torch.randn(5, 5, generator=g).to(DEV)
        opt_ref.zero_grad(set_to_none=True)
        (ref(x) - y).pow(2).sum().backward()
        raw = [p.grad.clone() for p in ref.parameters()]
        n_ref = torch.nn.utils.clip_grad_norm_(ref.parameters(), max_norm if max_norm > 0 else float("inf"))
        opt_ref.step()

        versions = [p._version for p in new.parameters()]
        # the SAME gradients for both optimizers: Adam turns a gradient of +-1e-9 into an update of +-lr, so the last-bit
        # noise between two backward passes over differently placed weights must not enter the comparison
        with torch.no_grad():
            for gv, r in zip(fp.grad_views, raw):
                gv.copy_(r)                                  # (the unclipped ones: clip_grad_norm_ rescaled ref's in place)
        grads_before = fp.grad_flat.clone()
        opt_new.step(max_norm)
        assert all(p._version > v for p, v in zip(new.parameters(), versions))       # shadow.py keys on this
        assert torch.equal(fp.grad_flat, grads_before)                                # the gradient is not rescaled in place
        if max_norm > 0:
            torch.testing.assert_close(opt_new.grad_norm[0], n_ref, rtol=2e-6, atol=0)
        for a, b in zip(ref.parameters(), new.parameters()):
            torch.testing.assert_close(b, a, rtol=2e-5, atol=2e-7)
    # the padding between the parameters' slices stays zero (it has zero gradients and zero moments)
    mask = torch.ones_like(fp.flat, dtype=torch.bool)
    from snipper_amd.grad_sync import flat_offsets
    offs, _ = flat_offsets(fp.params)
    for o, p in zip(offs, fp.params):
        mask[o:o + p.numel()] = False
    assert float(fp.flat[mask].abs().max()) == 0.0


def test_flat_adamw_is_deterministic_and_state_round_trips():
    a, b = _model(1), _model(1)
    fa, fb = FlatParameters(_groups(a)), FlatParameters(_groups(b))
    oa, ob = FlatAdamW(fa, [1e-3] * 3), FlatAdamW(fb, [1e-3] * 3)
    g = torch.Generator().manual_seed(5)
    for step in range(3):
        grad = torch.randn(fa.flat.numel(), generator=g).to(DEV)
        fa.grad_flat.copy_(grad); fb.grad_flat.copy_(grad)
        if step == 2:                                   # b continues from a's state through the state_dict
            ob.load_state_dict(oa.state_dict())
            fb.flat.copy_(fa.flat)
        oa.step(0.1); ob.step(0.1)
        if step != 1:
            assert torch.equal(fa.flat, fb.flat)
