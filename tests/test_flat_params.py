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


def _bench_module():
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    return b


def test_flat_adamw_state_dict_is_in_the_reference_optimizers_parameter_order():
    """``checkpoint['optimizer']`` numbers parameters by position inside the reference's
    param_groups, which main.py:201-217 builds in ``model.named_parameters()`` order; bench.py's flat layout lists the
    decoder-side parameters first.  With ``reference_groups`` the state dict FlatAdamW emits / consumes pairs every
    moment tensor with the parameter torch.optim.AdamW -- built the reference's way on the REAL model -- means."""
    from types import SimpleNamespace
    from snipper_amd.flat_params import FlatAdamW
    from snipper_amd.model import build_model
    b = _bench_module()
    a = SimpleNamespace(hidden_dim=192, enc_layers=2, dec_layers=2, frames=2, future_frames=0, use_pytorch_deform=0,
                        batch=1, height=96, width=128)
    torch.manual_seed(0)
    model = build_model(b.model_args(a))
    named = list(model.named_parameters())

    def match(n, kws):                       # main.py:190-196
        return any(k in n for k in kws)
    bb, slow = ["backbone.0"], ["reference_points", "sampling_offsets"]
    ref_groups = [[p for n, p in named if not match(n, bb) and not match(n, slow) and p.requires_grad],
                  [p for n, p in named if match(n, bb) and p.requires_grad],
                  [p for n, p in named if match(n, slow) and p.requires_grad]]
    assert [[id(p) for p in g] for g in b.reference_param_groups(named)] == [[id(p) for p in g] for g in ref_groups]
    opt_ref = torch.optim.AdamW([{"params": g, "lr": lr} for g, lr in zip(ref_groups, (1e-4, 1e-5, 1e-5))], lr=1e-4,
                                weight_decay=1e-4)
    main, backbone, slow_g = b.optimizer_groups(named)
    assert [id(p) for p in main] != [id(p) for p in ref_groups[0]], "the flat layout is expected to reorder `main`"
    assert {id(p) for p in main} == {id(p) for p in ref_groups[0]}
    fp = FlatParameters([main, slow_g, backbone])
    opt = FlatAdamW(fp, [1e-4, 1e-5, 1e-5], weight_decay=1e-4, reference_groups=ref_groups)
    # FlatAdamW -> torch.optim.AdamW: moment of flat parameter i holds the constant i + 1
    for i, (m1, m2) in enumerate(zip(opt._moment_views(opt.exp_avg), opt._moment_views(opt.exp_avg_sq))):
        m1.fill_(float(i + 1)); m2.fill_(float(i + 1) * 0.5)
    opt.step_count = 3
    sd = opt.state_dict()
    assert [len(g["params"]) for g in sd["param_groups"]] == [len(g) for g in ref_groups]
    assert [g["lr"] for g in sd["param_groups"]] == [1e-4, 1e-5, 1e-5]
    opt_ref.load_state_dict(sd)
    flat_index = {id(p): i for i, p in enumerate(fp.params)}
    for g in opt_ref.param_groups:
        for p in g["params"]:
            st = opt_ref.state[p]
            assert st["exp_avg"].shape == p.shape
            assert float(st["exp_avg"].flatten()[0]) == flat_index[id(p)] + 1.0
            assert float(st["exp_avg_sq"].flatten()[-1]) == (flat_index[id(p)] + 1.0) * 0.5
            assert float(st["step"]) == 3.0
    # torch.optim.AdamW -> FlatAdamW: tag every state by the parameter's position in named_parameters()
    pos = {id(p): k for k, (n, p) in enumerate(named)}
    for g in opt_ref.param_groups:
        for p in g["params"]:
            opt_ref.state[p]["exp_avg"].fill_(float(pos[id(p)]))
            opt_ref.state[p]["exp_avg_sq"].fill_(float(pos[id(p)]) + 0.25)
    opt2 = FlatAdamW(fp, [1e-4, 1e-5, 1e-5], weight_decay=1e-4, reference_groups=ref_groups)
    opt2.load_state_dict(opt_ref.state_dict())
    assert opt2.step_count == 3
    for p, m1, m2 in zip(fp.params, opt2._moment_views(opt2.exp_avg), opt2._moment_views(opt2.exp_avg_sq)):
        assert float(m1.flatten()[0]) == float(pos[id(p)]) and float(m2.flatten()[-1]) == float(pos[id(p)]) + 0.25
    # the flat order itself (no reference_groups) is NOT the reference's: a state dict loaded that way must not pass silently
    # where shapes differ
    opt3 = FlatAdamW(fp, [1e-4, 1e-5, 1e-5], weight_decay=1e-4, group_order=(0, 2, 1))
    import pytest
    with pytest.raises(ValueError):
        opt3.load_state_dict(opt_ref.state_dict())
    # every trainable parameter exactly once
    with pytest.raises(ValueError):
        FlatAdamW(fp, [1e-4, 1e-5, 1e-5], reference_groups=[ref_groups[0], ref_groups[1]])


class _LinearIntoView(torch.autograd.Function):
    """y = x W^T with the weight gradient written where flat_params.claim_grad_view says (what dense._BigLinear's backward does
    with its kernels) -- pure PyTorch, so that the CPU suite covers the claim / adopt / veto logic."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        ctx.w_param = w
        return x @ w.t()

    @staticmethod
    def backward(ctx, g):
        from snipper_amd.flat_params import claim_grad_view
        x, w = ctx.saved_tensors
        out = claim_grad_view(ctx.w_param)
        if out is None:
            out = torch.empty_like(w)
        torch.mm(g.t(), x, out=out)
        return g @ w, out


def test_claimed_gradient_views_are_adopted_once_per_backward_and_can_be_vetoed():
    """flat_params.claim_grad_view on the CPU (the GPU counterpart with the real kernels:
    tests/test_flat_adamw_gpu.py::test_weight_gradients_written_straight_into_the_flat_buffer)."""
    from snipper_amd.flat_params import FlatParameters
    torch.manual_seed(0)
    w1 = torch.nn.Parameter(torch.randn(6, 5))
    w2 = torch.nn.Parameter(torch.randn(5, 6))
    x = torch.randn(7, 5)

    def run(direct, guard=None):
        w1.grad = w2.grad = None
        flat = FlatParameters([[w1, w2]], grad_guard=guard)
        old = FlatParameters.direct_grads
        FlatParameters.direct_grads = direct
        try:
            flat.drop_param_grads()
            y = _LinearIntoView.apply(x, w1)                       # w1 once
            z = _LinearIntoView.apply(_LinearIntoView.apply(y, w2), w1)    # ... and a second time
            z.square().sum().backward()
            in_place = [p.grad.data_ptr() == v.data_ptr() for p, v in zip(flat.params, flat.grad_views)]
            flat.pack()
            return flat.grad_flat.clone(), in_place
        finally:
            FlatParameters.direct_grads = old

    ref, in_ref = run(False)
    assert not any(in_ref)
    got, in_got = run(True)
    # w2 (used once) lives in the flat buffer; w1's two gradients -- one in the view, one in a tensor of its own -- are summed
    # by the engine into a new tensor, which pack() copies
    assert in_got == [False, True]
    torch.testing.assert_close(got, ref)
    vetoed, in_v = run(True, guard=lambda p: p is not w2)
    assert in_v == [False, False]
    torch.testing.assert_close(vetoed, ref)
    # .grad kept between two backward passes (micro-batches): the second pass gets no view, the sum is right
    w1.grad = w2.grad = None
    flat = FlatParameters([[w2]])
    flat.drop_param_grads()
    for _ in range(2):
        _LinearIntoView.apply(torch.randn(3, 6, generator=torch.Generator().manual_seed(1)), w2).square().sum().backward()
    twice = w2.grad.clone()
    flat.drop_param_grads()
    _LinearIntoView.apply(torch.randn(3, 6, generator=torch.Generator().manual_seed(1)), w2).square().sum().backward()
    torch.testing.assert_close(twice, 2 * w2.grad)
