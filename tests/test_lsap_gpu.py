"""Device Hungarian matching (csrc/lsap.cuh) against SciPy."""
import numpy as np
import pytest
import torch
from scipy.optimize import linear_sum_assignment

from snipper_amd import _lib

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("n,m", [(60, 8), (60, 1), (60, 21), (7, 7), (64, 64), (5, 3), (33, 32)])
def test_lsap_matches_scipy(n, m):
    rng = np.random.RandomState(n * 100 + m)
    P = 12
    cost = rng.standard_normal((P, n, m)).astype(np.float32) * rng.uniform(0.1, 10)
    c = torch.from_numpy(cost).to(DEV)
    src = torch.empty(P, m, dtype=torch.long, device=DEV)
    tgt = torch.empty(P, m, dtype=torch.long, device=DEV)
    rc = _lib.load().snipper_lsap_f32(torch.cuda.current_stream().cuda_stream, c.data_ptr(), P, n, m, src.data_ptr(), tgt.data_ptr())
    _lib.check(rc, "lsap")
    for p in range(P):
        r, cidx = linear_sum_assignment(cost[p].astype(np.float64))
        np.testing.assert_array_equal(src[p].cpu().numpy(), r)
        np.testing.assert_array_equal(tgt[p].cpu().numpy(), cidx)


def test_criterion_device_and_host_matching_agree(golden_dir):
    import os
    from snipper_amd.criterion import HungarianMatcher, SetCriterion
    b = torch.load(os.path.join(golden_dir, "g5_criterion.pt"))
    crit = SetCriterion(HungarianMatcher(**b["matcher_costs"]),
                        ["is_human", "root", "joint", "joint_disp", "joint_cont", "heatmap"], 0.5, b["weight"]).to(DEV)
    mv = lambda x: x.to(DEV) if isinstance(x, torch.Tensor) else x
    layers = [{k: mv(v) for k, v in o.items()} for o in b["layers"]]
    out = dict(layers[-1], heatmaps=[mv(h) for h in b["heatmaps"]], aux_outputs=layers[:-1])
    targets = [{k: mv(v) for k, v in t.items()} for t in b["targets"]]
    losses, indices = crit(out, targets)
    for k, v in b["losses"].items():
        torch.testing.assert_close(losses[k].cpu(), v.reshape(losses[k].shape), rtol=1e-4, atol=1e-5, msg=lambda m: f"{k}: {m}")
    for (a, c), (ra, rc) in zip(indices, b["indices"]):
        assert torch.equal(a.cpu(), ra) and torch.equal(c.cpu(), rc)
