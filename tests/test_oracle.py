"""Pin the CPU oracle (oracle/) against golden vectors produced by the reference itself.

The fixtures come from tests/golden/gen_golden.py, which imports /root/reference in the build
container; here only the stored tensors are read.  Bar: fp64 agreement to rounding.
"""
import os

import numpy as np
import pytest
import torch

from oracle import msda_oracle as O


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def _testpy_inputs(channels):
    """Same draw order as models/ops/test.py:33-36 (value, loc, attn from the global CPU RNG)."""
    N, M, Lq, L, P, S = 1, 2, 2, 2, 2, 30
    value = torch.rand(N, S, M, channels) * 0.01
    loc = torch.rand(N, Lq, M, L, P, 2)
    attn = torch.rand(N, Lq, M, L, P) + 1e-5
    attn /= attn.sum(-1, keepdim=True).sum(-2, keepdim=True)
    return value, loc, attn


def test_rng_stream_matches_fixture(golden_dir):
    """The test.py inputs are re-drawn (seed 3) rather than stored for the big channel counts;
    make sure this torch build still produces the stream the fixtures were made with."""
    g = _load(golden_dir, "g1_testpy.npz")
    torch.manual_seed(3)
    v, l, a = _testpy_inputs(2)
    np.testing.assert_array_equal(v.numpy(), g["fwd64_value"])
    v, l, a = _testpy_inputs(2)
    np.testing.assert_array_equal(l.numpy(), g["fwd32_loc"])
    for D in [30, 32, 64, 71, 1025, 2048, 3096]:
        v, l, a = _testpy_inputs(D)
        if D <= 71:
            np.testing.assert_array_equal(a.numpy(), g[f"gc{D}_attn"])
        else:
            assert float(v.double().sum()) == float(g[f"gc{D}_value_sum"])


def test_c_oracle_forward_testpy_cases(golden_dir):
    g = _load(golden_dir, "g1_testpy.npz")
    shapes = g["shapes"]
    lsi = O.level_start_index(shapes)
    out64 = O.core_c_forward(g["fwd64_value"].astype(np.float64), shapes, lsi, g["fwd64_loc"], g["fwd64_attn"])
    np.testing.assert_allclose(out64, g["fwd64_out"], rtol=1e-12, atol=1e-15)
    out32 = O.core_c_forward(g["fwd32_value"], shapes, lsi, g["fwd32_loc"], g["fwd32_attn"])
    assert out32.dtype == np.float32
    np.testing.assert_allclose(out32, g["fwd32_out"], rtol=1e-2, atol=1e-3)   # test.py:56 tolerance
    np.testing.assert_allclose(out32, g["fwd32_out"], rtol=2e-5, atol=1e-8)   # and much tighter


@pytest.mark.parametrize("D", [30, 32, 64, 71, 1025, 2048, 3096])
def test_c_oracle_gradients_testpy_channels(golden_dir, D):
    g = _load(golden_dir, "g1_testpy.npz")
    shapes = g["shapes"]
    lsi = O.level_start_index(shapes)
    torch.manual_seed(3)
    _testpy_inputs(2), _testpy_inputs(2)
    for d in [30, 32, 64, 71, 1025, 2048, 3096]:
        v, l, a = _testpy_inputs(d)
        if d == D:
            break
    v, l, a = v.double().numpy(), l.double().numpy(), a.double().numpy()
    out = O.core_c_forward(v, shapes, lsi, l, a)
    np.testing.assert_allclose(out, g[f"gc{D}_out"], rtol=1e-11, atol=1e-15)
    gv, gl, ga = O.core_c_backward(v, shapes, lsi, l, a, g[f"gc{D}_grad_out"])
    np.testing.assert_allclose(gl, g[f"gc{D}_grad_loc"], rtol=1e-9, atol=1e-13)
    np.testing.assert_allclose(ga, g[f"gc{D}_grad_attn"], rtol=1e-9, atol=1e-13)
    if D <= 71:
        np.testing.assert_allclose(gv, g[f"gc{D}_grad_value"], rtol=1e-10, atol=1e-14)
    else:
        np.testing.assert_allclose(gv[..., ::64], g[f"gc{D}_grad_value_s64"], rtol=1e-10, atol=1e-14)
        np.testing.assert_allclose(gv.sum(-1), g[f"gc{D}_grad_value_sum"], rtol=1e-9, atol=1e-12)


def test_c_oracle_d48_edges(golden_dir):
    """D=48 / L=3 / P=4 with locations in [-0.2, 1.2] and exact boundary hits."""
    g = _load(golden_dir, "g2_core_d48.npz")
    shapes = g["shapes"]
    lsi = O.level_start_index(shapes)
    v, l, a = (g[k].astype(np.float64) for k in ("value", "loc", "attn"))
    out = O.core_c_forward(v, shapes, lsi, l, a, threads=2)
    np.testing.assert_allclose(out, g["out"], rtol=1e-11, atol=1e-13)
    gv, gl, ga = O.core_c_backward(v, shapes, lsi, l, a, g["grad_out"].astype(np.float64), threads=2)
    np.testing.assert_allclose(gv, g["grad_value"], rtol=1e-5, atol=1e-6)     # stored as float32
    np.testing.assert_allclose(gl, g["grad_loc"], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(ga, g["grad_attn"], rtol=1e-10, atol=1e-12)
    out32 = O.core_c_forward(g["value"], shapes, lsi, g["loc"], g["attn"])
    np.testing.assert_allclose(out32, g["out32"], rtol=1e-4, atol=1e-5)


def test_oracle_threads_do_not_change_results(golden_dir):
    g = _load(golden_dir, "g2_core_d48.npz")
    shapes = g["shapes"]
    lsi = O.level_start_index(shapes)
    a1 = O.core_c_backward(g["value"], shapes, lsi, g["loc"], g["attn"], g["grad_out"], threads=1)
    a4 = O.core_c_backward(g["value"], shapes, lsi, g["loc"], g["attn"], g["grad_out"], threads=4)
    for x, y in zip(a1, a4):
        np.testing.assert_array_equal(x, y)


def test_gridsample_restatement(golden_dir):
    g = _load(golden_dir, "g2_core_d48.npz")
    v, l, a = (torch.from_numpy(g[k]).double().requires_grad_(True) for k in ("value", "loc", "attn"))
    out = O.core_gridsample(v, g["shapes"], l, a)
    np.testing.assert_allclose(out.detach().numpy(), g["out"], rtol=1e-12, atol=1e-14)
    gv, gl, ga = torch.autograd.grad(out, (v, l, a), torch.from_numpy(g["grad_out"]).double())
    np.testing.assert_allclose(gl.numpy(), g["grad_loc"], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(ga.numpy(), g["grad_attn"], rtol=1e-10, atol=1e-12)


def test_empty_and_degenerate_inputs():
    """Lq=1, one level of 1x1, a sample exactly on the pixel centre and samples far outside."""
    shapes = np.array([[1, 1]], dtype=np.int64)
    lsi = np.array([0], dtype=np.int64)
    value = np.array([3.0, -2.0], dtype=np.float64).reshape(1, 1, 1, 2)
    loc = np.array([[0.5, 0.5], [7.0, 7.0], [-3.0, 0.5]], dtype=np.float64).reshape(1, 1, 1, 1, 3, 2)
    attn = np.array([0.5, 0.25, 0.25]).reshape(1, 1, 1, 1, 3)
    out = O.core_c_forward(value, shapes, lsi, loc, attn)
    np.testing.assert_allclose(out.reshape(-1), [1.5, -1.0])
    gv, gl, ga = O.core_c_backward(value, shapes, lsi, loc, attn, np.ones((1, 1, 2)))
    np.testing.assert_allclose(gv.reshape(-1), [0.5, 0.5])
    np.testing.assert_allclose(ga.reshape(-1), [1.0, 0.0, 0.0])
    assert np.all(gl.reshape(3, 2)[1:] == 0)


@pytest.mark.parametrize("name", ["enc_t3", "dec_t3", "dec_t3f2"])
def test_module_oracle_matches_reference_module(golden_dir, name):
    """oracle.st_msdeform_attn (per-pair formulation) == reference MSDeformAttn.forward, fp64."""
    b = torch.load(os.path.join(golden_dir, f"g3_module_{name}.pt"))
    cfg, sd = b["cfg"], b["state_dict"]
    T = cfg["n_frame"]
    out, locs, wts = O.st_msdeform_attn(
        b["query"], b["ref"], b["src"], b["shapes"].tolist(), b["mask"],
        sd["value_proj.weight"], sd["value_proj.bias"],
        [sd[f"sampling_offsets.{t}.weight"] for t in range(T)], [sd[f"sampling_offsets.{t}.bias"] for t in range(T)],
        [sd[f"attention_weights.{t}.weight"] for t in range(T)], [sd[f"attention_weights.{t}.bias"] for t in range(T)],
        sd["output_proj.weight"], sd["output_proj.bias"],
        cfg["n_heads"], cfg["n_levels"], cfg["n_points"], T)
    torch.testing.assert_close(out, b["out"], rtol=1e-10, atol=1e-12)
    if "vis_loc" in b:
        for x, y in zip(locs, b["vis_loc"]):
            torch.testing.assert_close(x, y, rtol=1e-12, atol=1e-14)
        for x, y in zip(wts, b["vis_w"]):
            torch.testing.assert_close(x, y, rtol=1e-10, atol=1e-13)
