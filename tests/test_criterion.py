"""SetCriterion + HungarianMatcher (snipper_amd/criterion.py) against golden vectors produced by the reference's own
classes (tests/golden/gen_golden.py::gen_g5; torchvision's gaussian_blur is a stand-in there, so the blur itself is
unpinned).  Runs on CPU."""
import os

import pytest
import torch

from snipper_amd.criterion import HungarianMatcher, SetCriterion, gaussian_blur


def _load(golden_dir):
    return torch.load(os.path.join(golden_dir, "g5_criterion.pt"))


def _build(b):
    matcher = HungarianMatcher(**b["matcher_costs"])
    return SetCriterion(matcher, ["is_human", "root", "joint", "joint_disp", "joint_cont", "heatmap"], 0.5, b["weight"])


def _outputs(b, grad=False, stacked=False):
    layers = [{k: v.clone().requires_grad_(grad) for k, v in o.items()} for o in b["layers"]]
    heat = [h.clone().requires_grad_(grad) for h in b["heatmaps"]]
    out = dict(layers[-1], heatmaps=heat, aux_outputs=layers[:-1])
    if stacked:   # the model's extra entry: all decoder layers at once
        out["all_layers"] = {"pred_logits": torch.stack([o["pred_logits"] for o in layers]),
                             "pred_kpts": torch.stack([torch.cat([o["pred_kpts2d"], o["pred_depth"]], -1) for o in layers])}
    return out, layers, heat


@pytest.mark.parametrize("stacked", [False, True])
def test_losses_and_matching_equal_reference(golden_dir, stacked):
    b = _load(golden_dir)
    crit = _build(b)
    out, layers, heat = _outputs(b, stacked=stacked)
    losses, indices = crit(out, b["targets"])
    assert set(losses) == set(b["losses"])
    for k, v in b["losses"].items():
        torch.testing.assert_close(losses[k], v.reshape(losses[k].shape), rtol=2e-5, atol=1e-6, msg=lambda m: f"{k}: {m}")
    for (a, c), (ra, rc) in zip(indices, b["indices"]):
        assert torch.equal(a, ra) and torch.equal(c, rc)
    total = crit.weighted_sum({k: v for k, v in losses.items()})
    want = sum(b["losses"][k] * b["weight"][k.rsplit("_", 1)[0] if k[-1].isdigit() else k] for k in b["losses"])
    # weighted_sum needs the aux weights too
    crit.weight_dict.update({f"{k}_{i}": w for k, w in b["weight"].items() for i in range(len(layers) - 1)})
    torch.testing.assert_close(crit.weighted_sum(losses), want.reshape(()), rtol=2e-5, atol=1e-5)


def test_gradients_equal_reference(golden_dir):
    b = _load(golden_dir)
    crit = _build(b)
    crit.weight_dict.update({f"{k}_{i}": w for k, w in b["weight"].items() for i in range(len(b["layers"]) - 1)})
    out, layers, heat = _outputs(b, grad=True)
    losses, _ = crit(out, b["targets"])
    leaves = [v for o in layers for v in o.values()] + heat
    grads = torch.autograd.grad(crit.weighted_sum(losses), leaves, allow_unused=True)
    for g, r in zip(grads, b["grads"]):
        if r is None:
            assert g is None or float(g.abs().max()) == 0
        else:
            torch.testing.assert_close(g, r, rtol=1e-4, atol=1e-6)


def test_single_layer_matcher_call_and_empty_targets(golden_dir):
    b = _load(golden_dir)
    matcher = HungarianMatcher(**b["matcher_costs"])
    idx = matcher(b["layers"][-1], b["targets"])                 # the reference's one-layer call signature
    for (a, c), (ra, rc) in zip(idx, b["indices"]):
        assert torch.equal(a, ra) and torch.equal(c, rc)
    # a sample without any person: no pairs, finite losses
    crit = _build(b)
    out, _, _ = _outputs(b)
    empty = {"kpts2d": torch.zeros(0, 3, 15, 3), "depth": torch.zeros(0, 3, 15, 2), "traj_ids": torch.zeros(0, dtype=torch.long),
             "max_depth": torch.tensor(15.0)}
    losses, indices = crit(out, [b["targets"][0], empty])
    assert indices[1][0].numel() == 0
    assert all(torch.isfinite(v).all() for v in losses.values())


def test_gaussian_blur_properties():
    img = torch.zeros(2, 9, 11)
    img[0, 4, 5] = 1.0
    out = gaussian_blur(img, 5)
    assert abs(float(out[0].sum()) - 1.0) < 1e-6                  # normalised kernel, mass preserved away from the border
    assert float(out[0, 4, 5]) == float(out[0].max())
    torch.testing.assert_close(out[0, 4, 4], out[0, 4, 6])
    assert torch.equal(gaussian_blur(img, 1), img)


def test_weighted_sum_fast_path_equals_reference_formula():
    """The matrix form of the weighted loss equals engine.py:56's entry-by-entry sum, value and gradients."""
    import importlib.util, os
    from types import SimpleNamespace
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    a = SimpleNamespace(hidden_dim=192, enc_layers=1, dec_layers=3, frames=2, future_frames=0, use_pytorch_deform=1,
                        batch=1, height=64, width=96)
    from snipper_amd.criterion import build_criterion
    crit = build_criterion(b.criterion_args(a))
    torch.manual_seed(0)
    n_dec, bs, nq, T, K = 3, 1, 60, 2, 15
    _, tgt = b.make_batches(a, "cpu", 1, seed=5)[0]
    logits = torch.randn(n_dec, bs, nq, T, 2, requires_grad=True)
    kpts = torch.rand(n_dec, bs, nq, T, K, 4, requires_grad=True)
    hm = [torch.rand(bs, T, 8, 12, 8, K, requires_grad=True)]
    def run(fast):
        out = {"pred_logits": logits[-1], "pred_kpts2d": kpts[-1, ..., :3], "pred_depth": kpts[-1, ..., 3:4],
               "heatmaps": hm, "all_layers": {"pred_logits": logits, "pred_kpts": kpts}}
        losses, _ = crit(out, tgt["targets"])
        if not fast:
            losses = dict(losses)                     # a different dict object: the generic formula
        total = crit.weighted_sum(losses)
        return total, torch.autograd.grad(total, (logits, kpts, hm[0]))
    t1, g1 = run(True)
    t2, g2 = run(False)
    torch.testing.assert_close(t1, t2, rtol=1e-5, atol=1e-5)
    for x, y in zip(g1, g2):
        torch.testing.assert_close(x, y, rtol=1e-4, atol=1e-6)
