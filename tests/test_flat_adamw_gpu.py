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
        # gradients of the PARAMETERS (the padding between their slices has none: its moments stay zero and are not part
        # of the torch-format state_dict)
        for va, vb in zip(fa.grad_views, fb.grad_views):
            gr = torch.randn(va.shape, generator=g).to(DEV)
            va.copy_(gr); vb.copy_(gr)
        if step == 2:                                   # b continues from a's state through the state_dict
            sd = oa.state_dict()
            assert all(v["exp_avg"].data_ptr() != oa.exp_avg.data_ptr() for v in sd["state"].values())    # copies, not live views
            ob.load_state_dict(sd)
            fb.flat.copy_(fa.flat)
        oa.step(0.1); ob.step(0.1)
        if step != 1:
            assert torch.equal(fa.flat, fb.flat)


def test_flat_adamw_is_an_optimizer_schedulers_and_torch_state_dicts_work():
    """the reference wraps its optimizer in ``StepLR`` (main.py:222) and checkpoints ``optimizer.state_dict()``
    (main.py:236-262).  FlatAdamW is a torch.optim.Optimizer; its state_dict is torch.optim.AdamW's over the MODEL's
    parameters in the mirrored optimizer's group order, in both directions."""
    ref, new = _model(2), _model(2)
    lrs = [1e-2, 1e-3, 3e-3]                       # of the flat groups (g0, g1, g2)
    order = (0, 2, 1)                              # the mirrored optimizer lists them as (g0, g2, g1)
    rg = _groups(ref)
    opt_ref = torch.optim.AdamW([{"params": rg[i], "lr": lrs[i]} for i in order], lr=1e-2, weight_decay=1e-2, fused=False,
                                foreach=False)
    fp = FlatParameters(_groups(new))
    opt_new = FlatAdamW(fp, lrs, weight_decay=1e-2, group_order=order)
    assert isinstance(opt_new, torch.optim.Optimizer)
    sch_ref = torch.optim.lr_scheduler.StepLR(opt_ref, 2, gamma=0.5)
    sch_new = torch.optim.lr_scheduler.StepLR(opt_new, 2, gamma=0.5)
    g = torch.Generator().manual_seed(8)

    def one_step(copy_grads=True):
        x = torch.randn(5, 3, 9, 9, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
        y = torch.randn(5, 5, generator=g).to(DEV)
        opt_ref.zero_grad(set_to_none=True)
        (ref(x) - y).pow(2).sum().backward()
        with torch.no_grad():
            for gv, p in zip(fp.grad_views, ref.parameters()):
                gv.copy_(p.grad)
        opt_ref.step(); opt_new.step(0.0)
        sch_ref.step(); sch_new.step()

    for _ in range(3):
        one_step()
    assert [g_["lr"] for g_ in opt_new.param_groups] == [lrs[0] * 0.5, lrs[1] * 0.5, lrs[2] * 0.5]      # StepLR fired once
    for a, b in zip(ref.parameters(), new.parameters()):
        torch.testing.assert_close(b, a, rtol=2e-5, atol=2e-7)
    # FlatAdamW -> torch.optim.AdamW: same keys / group layout, loads, and the next step agrees
    sd_new, sd_ref = opt_new.state_dict(), opt_ref.state_dict()
    assert [len(g_["params"]) for g_ in sd_new["param_groups"]] == [len(g_["params"]) for g_ in sd_ref["param_groups"]]
    assert [g_["lr"] for g_ in sd_new["param_groups"]] == [g_["lr"] for g_ in sd_ref["param_groups"]]
    for k, st in sd_ref["state"].items():
        # (moments are sums of terms of both signs: compare relative to the tensor's size, not element by element)
        sc = float(st["exp_avg"].abs().max()) + 1e-30
        torch.testing.assert_close(sd_new["state"][k]["exp_avg"] / sc, st["exp_avg"] / sc, rtol=1e-4, atol=1e-5)
        sc2 = float(st["exp_avg_sq"].abs().max()) + 1e-30
        torch.testing.assert_close(sd_new["state"][k]["exp_avg_sq"] / sc2, st["exp_avg_sq"] / sc2, rtol=1e-4, atol=1e-5)
        assert float(sd_new["state"][k]["step"]) == float(st["step"])
    opt_ref.load_state_dict(sd_new)
    # torch.optim.AdamW -> FlatAdamW on a fresh optimizer
    new2 = _model(2)
    fp2 = FlatParameters(_groups(new2))
    with torch.no_grad():
        fp2.flat.copy_(fp.flat)
    opt2 = FlatAdamW(fp2, lrs, weight_decay=1e-2, group_order=order)
    opt2.load_state_dict(sd_ref)
    assert opt2.step_count == 3 and [g_["lr"] for g_ in opt2.param_groups] == [g_["lr"] for g_ in opt_new.param_groups]
    grad = torch.randn(fp.flat.numel(), generator=g).to(DEV)
    fp.grad_flat.copy_(grad); fp2.grad_flat.copy_(grad)
    opt_new.step(0.1); opt2.step(0.1)
    torch.testing.assert_close(fp2.flat, fp.flat, rtol=2e-5, atol=2e-7)
    # betas / eps that differ between groups are refused instead of silently ignored
    opt2.param_groups[1]["eps"] = 1e-6
    with pytest.raises(RuntimeError):
        opt2.step(0.0)


def test_weight_gradients_written_straight_into_the_flat_buffer():
    """flat_params.claim_grad_view: the big Linears' weight-gradient kernels write into the parameter's slice of the flat
    gradient buffer and autograd adopts that view -- same gradients as through fresh tensors + pack(); a parameter used TWICE
    in one backward gets the view for one use and a tensor of its own for the other (their sum is right); a parameter whose
    .grad is still set gets none (accumulation over micro-batches keeps working); a guard can veto it."""
    from snipper_amd.dense import big_linear
    torch.manual_seed(0)
    dev = "cuda:0"
    lin = nn.Linear(384, 384).to(dev)
    lin2 = nn.Linear(384, 1024).to(dev)
    x = torch.randn(8192, 384, device=dev)

    def grads(direct, guard=None, twice=True):
        for p in list(lin.parameters()) + list(lin2.parameters()):
            p.grad = None
        flat = FlatParameters([list(lin.parameters()) + list(lin2.parameters())], grad_guard=guard)
        old = FlatParameters.direct_grads
        FlatParameters.direct_grads = direct
        try:
            flat.drop_param_grads()
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = big_linear(x, lin)
                z = big_linear(y.float(), lin) if twice else y          # the same Linear a second time
                out = big_linear(z.float(), lin2)
            out.float().square().mean().backward()
            in_place = [p.grad.data_ptr() == v.data_ptr() for p, v in zip(flat.params, flat.grad_views)]
            flat.pack()
            return flat.grad_flat.clone(), in_place
        finally:
            FlatParameters.direct_grads = old

    ref, in_ref = grads(False)
    got, in_got = grads(True)
    assert not any(in_ref)
    assert in_got[2] and in_got[3]                       # lin2 (used once): weight and bias live in the flat buffer
    torch.testing.assert_close(got, ref, rtol=1e-5, atol=1e-6)        # (lin: one use in place, the other added to it)
    vetoed, in_v = grads(True, guard=lambda p: False)
    assert not any(in_v)
    torch.testing.assert_close(vetoed, ref, rtol=1e-5, atol=1e-6)
    # gradient accumulation: .grad is kept between two backward passes -> the second pass must not claim
    flat = FlatParameters([list(lin2.parameters())])
    flat.drop_param_grads()
    for _ in range(2):
        with torch.autocast("cuda", dtype=torch.bfloat16):
            big_linear(x, lin2).float().square().mean().backward()
    twice_grad = lin2.weight.grad.clone()
    flat.drop_param_grads()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        big_linear(x, lin2).float().square().mean().backward()
    torch.testing.assert_close(twice_grad, 2 * lin2.weight.grad, rtol=1e-5, atol=1e-7)
