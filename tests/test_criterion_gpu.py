"""GPU tests of the criterion: the fused pair-loss kernels against the PyTorch formulation in the same class, and the
reference's golden (g5) through the CUDA path."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _inputs(n_dec, ms, T, K, seed):
    g = torch.Generator().manual_seed(seed)
    sk = torch.rand(n_dec, ms, T, K, 3, generator=g).to(DEV).requires_grad_(True)
    sd = torch.rand(n_dec, ms, T, K, 1, generator=g).to(DEV).requires_grad_(True)
    tk = torch.rand(1, ms, T, K, 3, generator=g).expand(n_dec, -1, -1, -1, -1).clone()
    tk[..., 2] = (torch.rand(1, ms, T, K, generator=g) < 0.75).float()
    td = torch.rand(1, ms, T, K, 2, generator=g).expand(n_dec, -1, -1, -1, -1).clone()
    td[..., 1] = (torch.rand(1, ms, T, K, generator=g) < 0.7).float()
    return sk, sd, tk.to(DEV), td.to(DEV)


@pytest.mark.parametrize("n_dec,ms,T,K", [(6, 14, 4, 15), (1, 1, 1, 15), (3, 5, 6, 15), (2, 3, 2, 2)])
def test_pair_loss_kernels_match_pytorch_formulation(n_dec, ms, T, K):
    from snipper_amd.criterion import SetCriterion, _PAIR_TERMS
    crit = SetCriterion(None, ["is_human", "root", "joint", "joint_disp", "joint_cont", "heatmap"], 0.5, {},
                        torch.rand(1, 1, K, 1) + 0.5).to(DEV)
    sk, sd, tk, td = _inputs(n_dec, ms, T, K, seed=n_dec * 100 + ms)
    logits = torch.zeros(n_dec, 1, 1, T, 2, device=DEV)
    crit.losses = ["root", "joint", "joint_disp", "joint_cont"]
    md, nt = torch.tensor(15.0, device=DEV), torch.full((1,), 7.0, device=DEV)
    res = {}
    for fused in (True, False):
        crit.fused_pair_losses = fused
        out = crit._all_losses(logits, sk, sd, tk, td, None, None, md, nt)
        assert list(out) == list(_PAIR_TERMS)
        w = torch.linspace(0.5, 1.5, n_dec * len(out), device=DEV).view(len(out), n_dec)
        total = sum((out[k] * w[i]).sum() for i, k in enumerate(out))
        grads = torch.autograd.grad(total, (sk, sd))
        res[fused] = ({k: v.detach() for k, v in out.items()}, grads)
    for k in _PAIR_TERMS:
        torch.testing.assert_close(res[True][0][k], res[False][0][k], rtol=2e-5, atol=1e-6, msg=lambda m: f"{k}: {m}")
    for a, b in zip(res[True][1], res[False][1]):
        torch.testing.assert_close(a, b, rtol=2e-5, atol=1e-7)


def test_criterion_golden_on_gpu(golden_dir):
    """Golden g5 (the reference SetCriterion + HungarianMatcher) through the CUDA path: device assignment kernel,
    fused pair losses, sync-free heat-map targets."""
    from snipper_amd.criterion import HungarianMatcher, SetCriterion
    b = torch.load(os.path.join(golden_dir, "g5_criterion.pt"))
    dev = lambda t: t.to(DEV) if torch.is_tensor(t) else t
    matcher = HungarianMatcher(**b["matcher_costs"])
    crit = SetCriterion(matcher, ["is_human", "root", "joint", "joint_disp", "joint_cont", "heatmap"], 0.5, b["weight"]).to(DEV)
    layers = [{k: dev(v).requires_grad_(True) for k, v in o.items()} for o in b["layers"]]
    heat = [dev(h).requires_grad_(True) for h in b["heatmaps"]]
    outputs = dict(layers[-1], heatmaps=heat, aux_outputs=layers[:-1])
    targets = [{k: dev(v) for k, v in t.items()} for t in b["targets"]]
    losses, indices = crit(outputs, targets)
    assert set(losses) == set(b["losses"])
    for k, v in b["losses"].items():
        torch.testing.assert_close(losses[k].cpu(), v, rtol=2e-4, atol=1e-5, msg=lambda m: f"{k}: {m}")
    for (s, t), (sr, tr) in zip(indices, b["indices"]):
        assert torch.equal(s.cpu(), torch.as_tensor(sr)) and torch.equal(t.cpu(), torch.as_tensor(tr))
    weight = b["weight"]
    total = sum(losses[k] * weight[k.rsplit("_", 1)[0] if k[-1].isdigit() else k] for k in losses)
    leaves = [v for o in layers for v in o.values()] + heat
    grads = torch.autograd.grad(total, leaves, allow_unused=True)
    for g, gr in zip(grads, b["grads"]):
        if gr is None:
            assert g is None or float(g.abs().max()) == 0.0
        else:
            s = max(float(gr.abs().max()), 1e-6)
            torch.testing.assert_close(g.cpu() / s, gr / s, rtol=1e-3, atol=2e-5)


@pytest.mark.parametrize("n_dec,bs,nq,T,K,packed", [(6, 2, 60, 4, 15, True), (1, 1, 7, 1, 15, False), (3, 2, 20, 6, 15, True),
                                                     (2, 1, 5, 2, 2, False)])
def test_match_cost_kernel_equals_pytorch_formulation(n_dec, bs, nq, T, K, packed):
    """csrc/match_cost.cuh against the broadcast formulation of the same class (matcher.py:60-127): dense inputs and
    slices of one [..., K, 4] head output (keypoint stride 4), targets with invisible joints and whole invisible
    frames, uneven target counts; the assignments computed from both matrices agree."""
    from snipper_amd.criterion import HungarianMatcher
    g = torch.Generator().manual_seed(n_dec * 1000 + nq)
    matcher = HungarianMatcher(cost_is_human=1.0, cost_root=5.0, cost_root_vis=0.1, cost_joint=5.0, cost_joint_vis=0.1,
                               cost_joint_depth=5.0, cost_root_depth=5.0)
    logits = torch.randn(n_dec, bs, nq, T, 2, generator=g).to(DEV)
    if packed:
        head = torch.rand(n_dec, bs, nq, T, K, 4, generator=g).to(DEV)
        kpts2d, depth = head[..., 0:3], head[..., 3:4]
    else:
        kpts2d = torch.rand(n_dec, bs, nq, T, K, 3, generator=g).to(DEV)
        depth = torch.rand(n_dec, bs, nq, T, K, 1, generator=g).to(DEV)
    targets = []
    for i in range(bs):
        m = 3 + 4 * i
        tk = torch.rand(m, T, K, 3, generator=g)
        tk[..., 2] = (torch.rand(m, T, K, generator=g) < 0.7).float()
        tk[0, 0, :, 2] = 0.0                                  # a frame with nothing visible
        td = torch.rand(m, T, K, 2, generator=g)
        td[..., 1] = (torch.rand(m, T, K, generator=g) < 0.6).float()
        targets.append({"kpts2d": tk.to(DEV), "depth": td.to(DEV), "max_depth": torch.tensor(15.0, device=DEV)})
    res = {}
    for on in (True, False):
        matcher.device_cost = on
        res[on] = (matcher.cost_matrices(logits, kpts2d, depth, targets),
                   matcher.match_all_layers(logits, kpts2d, depth, targets))
    for a, b in zip(res[True][0], res[False][0]):
        assert a.shape == b.shape
        torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-6)
    for a, b in zip(res[True][1][:3], res[False][1][:3]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("h,w,k", [(75, 100, 9), (38, 50, 5), (68, 120, 11), (9, 7, 3), (40, 40, 31)])
def test_heatmap_blur_kernel_matches_the_convolution_formulation(h, w, k):
    """csrc/heatmap_blur.cuh (one launch: clamp + reflect padding + separable Gaussian) against the tensor formulation of
    criterion.gaussian_blur evaluated on the CPU (padding + 2-D convolution with the outer-product kernel)."""
    from snipper_amd.criterion import gaussian_blur
    g = torch.Generator().manual_seed(h * 31 + k)
    img = (torch.rand(2, 5, 3, h, w, generator=g) < 0.02).float() * torch.randint(1, 4, (2, 5, 3, h, w), generator=g).float()
    want = gaussian_blur(img.clamp(max=1.0), k)                       # CPU path
    got = gaussian_blur(img.to(DEV), k, clamp_max=1.0)
    torch.testing.assert_close(got.cpu(), want, rtol=1e-5, atol=1e-6)
    noclamp = gaussian_blur(img.to(DEV), k)
    torch.testing.assert_close(noclamp.cpu(), gaussian_blur(img, k), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("bs,T,hw,counts", [(2, 4, [(19, 25), (10, 13), (5, 7)], [3, 2]), (1, 3, [(12, 9), (6, 5)], [4]),
                                             (2, 4, [(75, 100), (38, 50), (19, 25)], [2, 0])])
def test_fused_heatmap_targets_and_loss_equal_the_tensor_formulation(bs, T, hw, counts):
    """csrc/heatmap_loss.cuh: the one-launch scatter of the joint maps and the one-node heat-map loss on the encoder memory
    (value and the WHOLE gradient of the memory) against the tensor formulation of models/model.py:447-483 on the same inputs
    (views of the memory as models/deformable_transformer.py:141-149 makes them; joints outside the map, invisible joints,
    several joints on one pixel, a sample without persons)."""
    from snipper_amd.criterion import SetCriterion, HeatmapLoss
    from snipper_amd.deformable_transformer import HeatmapViews
    torch.manual_seed(sum(counts) + T)
    nhead, K, C = 8, 15, 384
    S = sum(h * w for h, w in hw)
    crit = SetCriterion(None, ["heatmap"], 0.5, {}).to(DEV)
    targets = []
    for n in counts:
        k = torch.rand(n, T + 1, K, 3, device=DEV)
        k[..., 0:2] = k[..., 0:2] * 1.3 - 0.15                       # some joints outside the image
        k[..., 2] = (k[..., 2] > 0.3).float()                        # some invisible
        if n > 1:
            k[1, :, :3] = k[0, :, :3]                                # several joints on one pixel
        targets.append({"kpts2d": k})

    def views(mem):
        out = HeatmapViews()
        pos = 0
        for h, w in hw:
            grid = mem[:, :, pos:pos + h * w].reshape(bs, T, h, w, nhead, C // nhead)
            out.append(grid[..., 0:K])
            pos += h * w
        return out

    mem_a = torch.randn(bs, T, S, C, device=DEV, requires_grad=True)
    mem_b = mem_a.detach().clone().requires_grad_(True)
    va, vb = views(mem_a), views(mem_b)
    va.source = (mem_a, hw, nhead, K)
    la = crit.loss_heatmap({"heatmaps": va}, targets)
    assert la.grad_fn is not None and "HeatmapLoss" in type(la.grad_fn).__name__
    # the reference formulation: plain list (no source), and the tensor path of the targets (CPU copy of the keypoints)
    tm_ref = crit.heatmap_targets([{"kpts2d": t["kpts2d"].cpu()} for t in targets], [v.shape[1:4] for v in vb], torch.device("cpu"))
    tm_gpu = crit.heatmap_targets(targets, [v.shape[1:4] for v in vb], torch.device(DEV))
    for a, b in zip(tm_gpu, tm_ref):
        torch.testing.assert_close(a.cpu(), b, rtol=1e-5, atol=1e-6)
    lb = crit.loss_heatmap({"heatmaps": list(vb)}, targets)
    torch.testing.assert_close(la, lb, rtol=2e-5, atol=1e-4)
    w = torch.tensor(0.37, device=DEV)
    (la * w).backward()
    (lb * w).backward()
    torch.testing.assert_close(mem_a.grad, mem_b.grad, rtol=1e-5, atol=1e-6)
    assert torch.equal(la, crit.loss_heatmap({"heatmaps": va}, targets))          # deterministic
