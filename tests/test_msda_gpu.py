"""GPU parity tests of the HIP core op, through the C ABI (ctypes) -- run with ``-m gpu``.

Layers of evidence:
  1. the counterpart of the reference's own test, models/ops/test.py (same shapes, seed, draw
     order, tolerances; gradcheck in double for its seven channel counts);
  2. golden vectors produced by the reference (tests/golden) -- forward and all three gradients;
  3. HIP vs the CPU oracle (oracle/) on seeded inputs for every kernel variant (generic, d48,
     bf16) incl. ragged sizes and edge cases;
  4. size-independent properties at BASELINE's full encoder size (linearity in value and in
     attn, a checksum identity for grad_value, zero output for out-of-map samples).
"""
import os

import numpy as np
import pytest
import torch
from torch.autograd import gradcheck

from oracle import msda_oracle as O
from snipper_amd import MultiScaleDeformableAttention as MSDA
from snipper_amd import _lib
from snipper_amd.ms_deform_attn_func import MSDeformAttnFunction, ms_deform_attn_core_pytorch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def lsi_of(shapes):
    return torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))


# ------------------------------------------------------------------ 1. models/ops/test.py ----
N_, M_, Lq_, L_, P_ = 1, 2, 2, 2, 2
SHAPES_T = [(6, 4), (3, 2)]


def _testpy_inputs(channels):
    S = sum(h * w for h, w in SHAPES_T)
    value = torch.rand(N_, S, M_, channels).to(DEV) * 0.01
    loc = torch.rand(N_, Lq_, M_, L_, P_, 2).to(DEV)
    attn = torch.rand(N_, Lq_, M_, L_, P_).to(DEV) + 1e-5
    attn /= attn.sum(-1, keepdim=True).sum(-2, keepdim=True)
    return value, loc, attn


def test_testpy_forward_double_float_and_gradcheck():
    """test.py:31-78,85-86 in its own order under torch.manual_seed(3)."""
    shapes = torch.as_tensor(SHAPES_T, dtype=torch.long, device=DEV)
    lsi = lsi_of(shapes)
    torch.manual_seed(3)
    v, l, a = _testpy_inputs(2)
    ref = ms_deform_attn_core_pytorch(v.double(), shapes, l.double(), a.double()).cpu()
    got = MSDeformAttnFunction.apply(v.double(), shapes, lsi, l.double(), a.double(), 2).cpu()
    assert torch.allclose(got, ref)                                   # test.py:40 (defaults)
    v, l, a = _testpy_inputs(2)
    ref = ms_deform_attn_core_pytorch(v, shapes, l, a).cpu()
    got = MSDeformAttnFunction.apply(v, shapes, lsi, l, a, 2).cpu()
    assert torch.allclose(got, ref, rtol=1e-2, atol=1e-3)             # test.py:56
    assert torch.allclose(got, ref, rtol=1e-4, atol=1e-7)
    for channels in [30, 32, 64, 71, 1025, 2048, 3096]:               # test.py:85-86
        v, l, a = _testpy_inputs(channels)
        v, l, a = v.double().requires_grad_(True), l.double().requires_grad_(True), a.double().requires_grad_(True)
        assert gradcheck(MSDeformAttnFunction.apply, (v, shapes, lsi, l, a, 2)), channels


# ------------------------------------------------------------------ 2. reference goldens -----
def test_reference_golden_testpy_vectors(golden_dir):
    g = np.load(os.path.join(golden_dir, "g1_testpy.npz"))
    shapes = torch.from_numpy(g["shapes"]).to(DEV)
    lsi = lsi_of(shapes)
    t = lambda k, dt: torch.from_numpy(g[k]).to(DEV).to(dt)
    out = MSDA.ms_deform_attn_forward(t("fwd64_value", torch.float64), shapes, lsi, t("fwd64_loc", torch.float64),
                                      t("fwd64_attn", torch.float64), 2)
    np.testing.assert_allclose(out.cpu().numpy(), g["fwd64_out"], rtol=1e-12, atol=1e-15)
    out = MSDA.ms_deform_attn_forward(t("fwd32_value", torch.float32), shapes, lsi, t("fwd32_loc", torch.float32),
                                      t("fwd32_attn", torch.float32), 2)
    np.testing.assert_allclose(out.cpu().numpy(), g["fwd32_out"], rtol=2e-5, atol=1e-8)
    for D in [30, 32, 64, 71]:
        v, l, a = (t(f"gc{D}_{k}", torch.float64) for k in ("value", "loc", "attn"))
        go = t(f"gc{D}_grad_out", torch.float64)
        out = MSDA.ms_deform_attn_forward(v, shapes, lsi, l, a, 2)
        np.testing.assert_allclose(out.cpu().numpy(), g[f"gc{D}_out"], rtol=1e-11, atol=1e-15)
        gv, gl, ga = MSDA.ms_deform_attn_backward(v, shapes, lsi, l, a, go, 2)
        np.testing.assert_allclose(gv.cpu().numpy(), g[f"gc{D}_grad_value"], rtol=1e-10, atol=1e-14)
        np.testing.assert_allclose(gl.cpu().numpy(), g[f"gc{D}_grad_loc"], rtol=1e-9, atol=1e-13)
        np.testing.assert_allclose(ga.cpu().numpy(), g[f"gc{D}_grad_attn"], rtol=1e-9, atol=1e-13)


@pytest.mark.parametrize("policy", [0, 1], ids=["auto_d48", "generic"])
def test_reference_golden_d48_edges(golden_dir, policy):
    g = np.load(os.path.join(golden_dir, "g2_core_d48.npz"))
    shapes = torch.from_numpy(g["shapes"]).to(DEV)
    lsi = lsi_of(shapes)
    _lib.set_policy(policy)
    try:
        for dt, rt, at in [(torch.float64, 1e-10, 1e-12), (torch.float32, 2e-4, 2e-5)]:
            v, l, a, go = (torch.from_numpy(g[k]).to(DEV).to(dt) for k in ("value", "loc", "attn", "grad_out"))
            out = MSDA.ms_deform_attn_forward(v, shapes, lsi, l, a, 64)
            if dt == torch.float32:
                assert _lib.last_variant() == ("d48_lp12" if policy == 0 else "generic")
            np.testing.assert_allclose(out.cpu().numpy(), g["out"], rtol=rt, atol=at)
            gv, gl, ga = MSDA.ms_deform_attn_backward(v, shapes, lsi, l, a, go, 64)
            np.testing.assert_allclose(gv.cpu().numpy(), g["grad_value"], rtol=max(rt, 1e-5), atol=max(at, 2e-6))
            np.testing.assert_allclose(gl.cpu().numpy(), g["grad_loc"], rtol=rt * 5, atol=at * 50)
            np.testing.assert_allclose(ga.cpu().numpy(), g["grad_attn"], rtol=rt * 5, atol=at * 10)
    finally:
        _lib.set_policy(0)


# ------------------------------------------------------------------ 3. HIP vs CPU oracle -----
def _case(N, shapes, M, D, Lq, P, seed, lo=-0.15, hi=1.15, dtype=np.float32):
    rng = np.random.RandomState(seed)
    shapes = np.asarray(shapes, dtype=np.int64)
    L = len(shapes)
    S = int((shapes[:, 0] * shapes[:, 1]).sum())
    value = rng.standard_normal((N, S, M, D)).astype(dtype)
    # locations on a half-offset 2^-12 lattice (never exactly on a pixel boundary): loc*size-0.5 is then exact in fp32 AND fp64 (with or without FMA
    # contraction), so kernel and oracle always agree on floor() / the in-range test and the
    # comparison cannot flake on a pixel-boundary flip (grad_loc is discontinuous there)
    loc = ((np.round(rng.uniform(lo, hi, (N, Lq, M, L, P, 2)) * 4096) + 0.5) / 4096).astype(dtype)
    attn = rng.uniform(0.0, 1.0, (N, Lq, M, L, P)).astype(dtype)
    attn /= attn.sum((-1, -2), keepdims=True)
    go = rng.standard_normal((N, Lq, M * D)).astype(dtype)
    return value, shapes, O.level_start_index(shapes), loc, attn, go


CASES = {
    # name: (N, shapes, M, D, Lq, P)        ragged on purpose: rows not a multiple of any tile
    "d48_snipper_small": (2, [(19, 25), (10, 13), (5, 7)], 8, 48, 77, 4),
    "d48_one_row": (1, [(3, 5)], 1, 48, 1, 1),
    "d48_lp_runtime": (3, [(7, 9), (4, 5)], 5, 48, 33, 3),
    "d48_many_points": (1, [(6, 6), (3, 3), (2, 2), (1, 1), (5, 2)], 2, 48, 9, 5),   # L*P=25 > 16
    "d24_hidden192": (2, [(9, 12), (5, 6), (3, 3)], 8, 24, 41, 4),
    "d24_lp_runtime": (1, [(7, 5), (3, 4)], 3, 24, 19, 3),
    "d2": (1, [(6, 4), (3, 2)], 2, 2, 2, 2),
    "d71": (2, [(5, 4), (2, 3)], 3, 71, 7, 2),
    "d130": (1, [(4, 4)], 2, 130, 5, 3),
}


@pytest.mark.parametrize("name", sorted(CASES))
@pytest.mark.parametrize("policy", [0, 1], ids=["auto", "generic"])
def test_hip_matches_oracle_f32(name, policy):
    v, shapes, lsi, loc, attn, go = _case(*CASES[name], seed=len(name) * 7919)
    ref_out = O.core_c_forward(v.astype(np.float64), shapes, lsi, loc.astype(np.float64), attn.astype(np.float64))
    ref_gv, ref_gl, ref_ga = O.core_c_backward(v.astype(np.float64), shapes, lsi, loc.astype(np.float64),
                                               attn.astype(np.float64), go.astype(np.float64))
    tv, tl, ta, tg = (torch.from_numpy(x).to(DEV) for x in (v, loc, attn, go))
    ts, ti = torch.from_numpy(shapes).to(DEV), torch.from_numpy(lsi).to(DEV)
    _lib.set_policy(policy)
    try:
        out = MSDA.ms_deform_attn_forward(tv, ts, ti, tl, ta, 64)
        variant = _lib.last_variant()
        gv, gl, ga = MSDA.ms_deform_attn_backward(tv, ts, ti, tl, ta, tg, 64)
    finally:
        _lib.set_policy(0)
    if policy == 0 and CASES[name][3] in (48, 24):          # 24 = hidden_dim 192, the reference's default (main.py:88)
        assert variant.startswith("d%d" % CASES[name][3]), variant
    else:
        assert variant == "generic"
    # fp32 kernel vs fp64 oracle: a handful of ulps of the largest term
    np.testing.assert_allclose(out.cpu().numpy(), ref_out, rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(gv.cpu().numpy(), ref_gv, rtol=1e-4, atol=3e-5)
    np.testing.assert_allclose(ga.cpu().numpy(), ref_ga, rtol=1e-4, atol=1e-4)
    scale = float(np.abs(ref_gl).max()) + 1e-6
    np.testing.assert_allclose(gl.cpu().numpy() / scale, ref_gl / scale, rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize("name", ["d48_snipper_small", "d71", "d2"])
def test_hip_matches_oracle_f64(name):
    v, shapes, lsi, loc, attn, go = _case(*CASES[name], seed=11, dtype=np.float64)
    ref_out = O.core_c_forward(v, shapes, lsi, loc, attn)
    ref = O.core_c_backward(v, shapes, lsi, loc, attn, go)
    tv, tl, ta, tg = (torch.from_numpy(x).to(DEV) for x in (v, loc, attn, go))
    ts, ti = torch.from_numpy(shapes).to(DEV), torch.from_numpy(lsi).to(DEV)
    out = MSDA.ms_deform_attn_forward(tv, ts, ti, tl, ta, 64)
    got = MSDA.ms_deform_attn_backward(tv, ts, ti, tl, ta, tg, 64)
    np.testing.assert_allclose(out.cpu().numpy(), ref_out, rtol=1e-12, atol=1e-13)
    for x, y in zip(got, ref):
        np.testing.assert_allclose(x.cpu().numpy(), y, rtol=1e-10, atol=1e-11)


@pytest.mark.parametrize("name", ["d48_snipper_small", "d48_lp_runtime", "d24_hidden192"])
def test_hip_bf16_against_fp32_oracle(name):
    """bf16 storage is a new capability (the reference dispatches float/double only).  Tolerance:
    inputs are rounded to bf16 first, so the only error left is f32 accumulation + one bf16
    rounding of the output: rtol 2^-8 on outputs; gradients are f32-accumulated."""
    v, shapes, lsi, loc, attn, go = _case(*CASES[name], seed=5)
    bf = lambda x: torch.from_numpy(x).to(torch.bfloat16)
    v_b, go_b = bf(v), bf(go)
    v_r, go_r = v_b.float().numpy().astype(np.float64), go_b.float().numpy().astype(np.float64)
    l64, a64 = loc.astype(np.float64), attn.astype(np.float64)
    ref_out = O.core_c_forward(v_r, shapes, lsi, l64, a64)
    ref_gv, ref_gl, ref_ga = O.core_c_backward(v_r, shapes, lsi, l64, a64, go_r)
    ts, ti = torch.from_numpy(shapes).to(DEV), torch.from_numpy(lsi).to(DEV)
    tl, ta = torch.from_numpy(loc).to(DEV), torch.from_numpy(attn).to(DEV)
    out = MSDA.ms_deform_attn_forward(v_b.to(DEV), ts, ti, tl, ta, 64)
    assert out.dtype == torch.bfloat16
    np.testing.assert_allclose(out.float().cpu().numpy(), ref_out, rtol=2 ** -7, atol=2e-2)
    gv, gl, ga = MSDA.ms_deform_attn_backward(v_b.to(DEV), ts, ti, tl, ta, go_b.to(DEV), 64)
    # round 5: a bf16 value off the encoder shape (the decoder's premixed cross attention) takes the tuned atomic kernel
    assert _lib.last_variant().startswith(("d48", "d24")), _lib.last_variant()
    np.testing.assert_allclose(gv.float().cpu().numpy(), ref_gv, rtol=2 ** -7, atol=2e-2)
    np.testing.assert_allclose(ga.cpu().numpy(), ref_ga, rtol=1e-3, atol=1e-3)
    scale = float(np.abs(ref_gl).max())
    np.testing.assert_allclose(gl.cpu().numpy() / scale, ref_gl / scale, rtol=1e-3, atol=1e-4)


def test_d24_bf16_rows_and_full_size_forward_backward():
    """hidden_dim = 192 (D = 24) on the tuned kernels at the 600x800 geometry: float32 with bf16 rows against the float32
    interface (same kernel, rows converted in place), and the bf16-value forward against the oracle."""
    shapes = [(75, 100), (38, 50), (19, 25)]
    v, sh, lsi, loc, attn, go = _case(1, shapes, 8, 24, 9875, 4, seed=3, lo=0.0, hi=1.0)
    tv, tl, ta, tg = (torch.from_numpy(x).to(DEV) for x in (v, loc, attn, go))
    ts, ti = torch.from_numpy(sh).to(DEV), torch.from_numpy(lsi).to(DEV)
    out = MSDA.ms_deform_attn_forward(tv, ts, ti, tl, ta, 64)
    assert _lib.last_variant() == "d24_lp12", _lib.last_variant()
    f64 = lambda a: a.astype(np.float64)
    ref_out = O.core_c_forward(f64(v), sh, lsi, f64(loc), f64(attn), threads=16)
    np.testing.assert_allclose(out.cpu().numpy(), ref_out, rtol=1e-4, atol=2e-5)
    out16 = MSDA.ms_deform_attn_forward(tv, ts, ti, tl, ta, 64, out_bf16=True)
    assert torch.equal(out16, out.to(torch.bfloat16))
    go16 = tg.bfloat16()
    a = MSDA.ms_deform_attn_backward(tv, ts, ti, tl, ta, go16, 64)
    assert _lib.last_variant() == "d24_lp12", _lib.last_variant()
    ref = O.core_c_backward(f64(v), sh, lsi, f64(loc), f64(attn), f64(go16.float().cpu().numpy()), threads=16)
    np.testing.assert_allclose(a[0].cpu().numpy(), ref[0], rtol=1e-4, atol=2e-4)
    s = float(np.abs(ref[1]).max())
    np.testing.assert_allclose(a[1].cpu().numpy() / s, ref[1] / s, rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(a[2].cpu().numpy(), ref[2], rtol=1e-4, atol=1e-4)


def test_edge_cases_all_outside_and_nonfinite():
    shapes = torch.tensor([[4, 5], [2, 3]], device=DEV)
    lsi = lsi_of(shapes)
    S = 26
    for D in (48, 7):
        v = torch.full((1, S, 2, D), float("nan"), device=DEV)       # value is never read for skipped samples
        loc = torch.full((1, 3, 2, 2, 2, 2), 5.0, device=DEV)
        loc[0, 0] = -4.0
        loc[0, 1, 0, 0, 0, 0] = float("nan")
        loc[0, 1, 0, 0, 1, 1] = float("inf")
        attn = torch.full((1, 3, 2, 2, 2), 0.25, device=DEV)
        out = MSDA.ms_deform_attn_forward(v, shapes, lsi, loc, attn, 64)
        assert torch.equal(out, torch.zeros_like(out))
        gv, gl, ga = MSDA.ms_deform_attn_backward(v, shapes, lsi, loc, attn, torch.ones_like(out), 64)
        assert float(gl.abs().sum()) == 0.0 and float(ga.abs().sum()) == 0.0
        assert float(gv.abs().sum()) == 0.0


def test_argument_errors_mirror_reference():
    shapes = torch.tensor([[2, 2]], device=DEV)
    lsi = torch.tensor([0], device=DEV)
    v = torch.zeros(3, 4, 1, 4, device=DEV)
    loc = torch.zeros(3, 1, 1, 1, 1, 2, device=DEV)
    attn = torch.zeros(3, 1, 1, 1, 1, device=DEV)
    with pytest.raises(RuntimeError, match="contiguous"):            # ms_deform_attn_cuda.cu:28
        MSDA.ms_deform_attn_forward(torch.zeros(3, 4, 1, 8, device=DEV)[..., ::2], shapes, lsi, loc, attn, 64)
    with pytest.raises(RuntimeError, match="must divide"):           # :52
        MSDA.ms_deform_attn_forward(v, shapes, lsi, loc, attn, 2)
    with pytest.raises(RuntimeError, match="CUDA tensor"):           # :34
        MSDA.ms_deform_attn_forward(v, shapes.cpu(), lsi, loc, attn, 64)
    with pytest.raises(RuntimeError, match="not implemented for"):
        MSDA.ms_deform_attn_forward(v.half(), shapes, lsi, loc.half(), attn.half(), 64)


# ------------------------------------------------------------------ 4. full-size properties ---
ENC_SHAPES = [(75, 100), (38, 50), (19, 25)]      # 600x800 input, strides 8/16/32


def _full_encoder_inputs(N=2, seed=0, local=True):
    g = torch.Generator(device="cpu").manual_seed(seed)
    shapes = torch.as_tensor(ENC_SHAPES, dtype=torch.long)
    S = int(shapes.prod(1).sum())
    M, D, L, P = 8, 48, 3, 4
    value = torch.randn(N, S, M, D, generator=g)
    if local:   # reference points on the pixel grid + offsets of a few pixels, as a trained model produces
        refs = []
        for h, w in ENC_SHAPES:
            ys, xs = torch.meshgrid(torch.arange(h) + 0.5, torch.arange(w) + 0.5, indexing="ij")
            refs.append(torch.stack([xs.reshape(-1) / w, ys.reshape(-1) / h], -1))
        ref = torch.cat(refs)[None, :, None, None, None, :]
        norm = torch.tensor([[w, h] for h, w in ENC_SHAPES], dtype=torch.float32)[None, None, None, :, None, :]
        loc = ref + torch.randn(N, S, M, L, P, 2, generator=g) * 3.0 / norm
    else:
        loc = torch.rand(N, S, M, L, P, 2, generator=g)
    loc = (torch.round(loc * 4096) + 0.5) / 4096     # exact pixel coordinates in fp32 and fp64 (see _case)
    attn = torch.softmax(torch.randn(N, S, M, L * P, generator=g), -1).view(N, S, M, L, P)
    return value.to(DEV), shapes.to(DEV), lsi_of(shapes).to(DEV), loc.to(DEV), attn.to(DEV)


@pytest.mark.parametrize("local", [True, False], ids=["local", "uniform"])
def test_full_size_properties(local):
    v, shapes, lsi, loc, attn = _full_encoder_inputs(local=local)
    f = lambda vv, aa=attn: MSDA.ms_deform_attn_forward(vv, shapes, lsi, loc, aa, 64)
    out = f(v)
    assert _lib.last_variant() == "d48_lp12"
    # (a) linearity in value and in attn
    v2 = torch.randn_like(v)
    torch.testing.assert_close(f(v * 0.5 + v2 * 2.0), out * 0.5 + f(v2) * 2.0, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(f(v, attn * 3.0), out * 3.0, rtol=1e-5, atol=1e-5)
    # (b) a constant value map is reproduced wherever the whole footprint is inside the map
    ones = torch.ones_like(v)
    inner = ((loc > 0.1) & (loc < 0.9)).all(-1).all(-1).all(-1)          # [N,S,M]
    o1 = f(ones).view(*inner.shape, -1)
    torch.testing.assert_close(o1[inner], torch.ones_like(o1[inner]), rtol=1e-5, atol=1e-5)
    # (c) a sub-sampled slice agrees with the CPU oracle
    idx = torch.arange(0, loc.shape[1], 97, device=DEV)
    sub_out = MSDA.ms_deform_attn_forward(v, shapes, lsi, loc[:, idx].contiguous(), attn[:, idx].contiguous(), 64)
    torch.testing.assert_close(sub_out, out[:, idx], rtol=0, atol=0)     # row-independent, bit-exact
    ref = O.core_c_forward(v.cpu().numpy(), shapes.cpu().numpy(), lsi.cpu().numpy(),
                           loc[:, idx].cpu().numpy(), attn[:, idx].cpu().numpy(), threads=4)
    np.testing.assert_allclose(sub_out.cpu().numpy(), ref, rtol=1e-4, atol=2e-5)
    # (d) backward: checksum identities.  sum(grad_value) == sum_rows g * (sum of in-map tap weights);
    #     <grad_attn, attn> == <out, g> ; and the sub-sampled slice matches the oracle.
    go = torch.randn_like(out)
    gv, gl, ga = MSDA.ms_deform_attn_backward(v, shapes, lsi, loc, attn, go, 64)
    lhs = (ga.double() * attn.double()).sum()
    rhs = (out.double() * go.double()).sum()
    assert abs(float(lhs - rhs)) <= 1e-5 * float(out.double().abs().mul(go.double().abs()).sum())
    gv1, _, _ = MSDA.ms_deform_attn_backward(ones, shapes, lsi, loc, attn, go, 64)
    per_row = (f(ones).double() * go.double()).sum()                     # = sum over taps of w*a*g
    assert abs(float(gv1.double().sum() - per_row)) <= 1e-5 * float(go.abs().sum())
    sgv, sgl, sga = MSDA.ms_deform_attn_backward(v, shapes, lsi, loc[:, idx].contiguous(), attn[:, idx].contiguous(),
                                                 go[:, idx].contiguous(), 64)
    torch.testing.assert_close(sgl, gl[:, idx], rtol=0, atol=0)
    torch.testing.assert_close(sga, ga[:, idx], rtol=0, atol=0)
    rgv, rgl, rga = O.core_c_backward(v.cpu().numpy(), shapes.cpu().numpy(), lsi.cpu().numpy(), loc[:, idx].cpu().numpy(),
                                      attn[:, idx].cpu().numpy(), go[:, idx].cpu().numpy(), threads=4)
    np.testing.assert_allclose(sgv.cpu().numpy(), rgv, rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(sga.cpu().numpy(), rga, rtol=1e-3, atol=1e-4)
    s = float(np.abs(rgl).max())
    np.testing.assert_allclose(sgl.cpu().numpy() / s, rgl / s, rtol=1e-3, atol=1e-4)


@pytest.mark.parametrize("Lq_is_S", [True, False])
def test_bf16_rows_interface_matches_f32(Lq_is_S):
    """bf16 rows at the op's activation interfaces (out written / grad_out read as bf16 by the kernels): the output is
    the float32 result rounded once, and the gradients equal the float32 path fed the same bf16-valued grad_out --
    for the owner-computes shape (Lq == S, host shapes known) and for the atomic kernel."""
    from snipper_amd import MultiScaleDeformableAttention as MSDA
    torch.manual_seed(5)
    shapes = [(19, 25), (10, 13), (5, 7)]
    S = sum(h * w for h, w in shapes)
    N, M, D, L, P = 2, 8, 48, 3, 4
    Lq = S if Lq_is_S else 77
    sh = torch.tensor(shapes, device="cuda:0")
    lsi = torch.cat((sh.new_zeros(1), sh.prod(1).cumsum(0)[:-1]))
    value = torch.randn(N, S, M, D, device="cuda:0")
    loc = torch.rand(N, Lq, M, L, P, 2, device="cuda:0")
    attn = torch.softmax(torch.randn(N, Lq, M, L * P, device="cuda:0"), -1).view(N, Lq, M, L, P)
    out32 = MSDA.ms_deform_attn_forward(value, sh, lsi, loc, attn, 64)
    out16 = MSDA.ms_deform_attn_forward(value, sh, lsi, loc, attn, 64, out_bf16=True)
    assert out16.dtype == torch.bfloat16 and torch.equal(out16, out32.to(torch.bfloat16))
    go16 = torch.randn(N, Lq, M * D, device="cuda:0").bfloat16()
    hs = shapes if Lq_is_S else None
    a = MSDA.ms_deform_attn_backward(value, sh, lsi, loc, attn, go16, 64, host_shapes=hs)
    b = MSDA.ms_deform_attn_backward(value, sh, lsi, loc, attn, go16.float(), 64, host_shapes=hs)
    for i, (x, y) in enumerate(zip(a, b)):
        scale = max(float(y.abs().max()), 1.0)
        # grad_value of the owner-computes shape: bf16 rows take the matrix-pipe tile kernel, whose weights are split into
        # bf16 hi + lo parts (relative error <= 2^-17 per weight), float32 rows the vector kernel
        tol = dict(rtol=4e-5, atol=8e-6) if (Lq_is_S and i == 0) else dict(rtol=1e-5, atol=2e-6)
        torch.testing.assert_close(x / scale, y / scale, **tol)


def test_bf16_rows_unsupported_shape_falls_back_to_cast():
    """D != 48 has no bf16-row kernel: the wrapper casts around the float32 entry points instead."""
    from snipper_amd import MultiScaleDeformableAttention as MSDA
    torch.manual_seed(6)
    shapes = [(6, 4), (3, 2)]
    S = sum(h * w for h, w in shapes)
    N, M, D, L, P, Lq = 1, 2, 16, 2, 2, 9
    sh = torch.tensor(shapes, device="cuda:0")
    lsi = torch.cat((sh.new_zeros(1), sh.prod(1).cumsum(0)[:-1]))
    value = torch.randn(N, S, M, D, device="cuda:0")
    loc = torch.rand(N, Lq, M, L, P, 2, device="cuda:0")
    attn = torch.softmax(torch.randn(N, Lq, M, L * P, device="cuda:0"), -1).view(N, Lq, M, L, P)
    out32 = MSDA.ms_deform_attn_forward(value, sh, lsi, loc, attn, 64)
    out16 = MSDA.ms_deform_attn_forward(value, sh, lsi, loc, attn, 64, out_bf16=True)
    assert torch.equal(out16, out32.to(torch.bfloat16))
    go16 = torch.randn(N, Lq, M * D, device="cuda:0").bfloat16()
    a = MSDA.ms_deform_attn_backward(value, sh, lsi, loc, attn, go16, 64)
    b = MSDA.ms_deform_attn_backward(value, sh, lsi, loc, attn, go16.float(), 64)
    for x, y in zip(a, b):
        torch.testing.assert_close(x, y, rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("N,Lq,mode", [(8, 60, "mixed"), (2, 64, "mixed"), (3, 7, "mixed"), (2, 60, "collapsed"),
                                       (2, 64, "one_pixel"), (2, 60, "outside")])
def test_sparse_backward_for_few_queries_on_bf16_value(N, Lq, mode):
    """csrc/msda_d48_sparse.cuh (the decoder's cross attention: few queries on the whole bf16 memory): grad_value without atomics
    -- against the C oracle on the bf16-rounded inputs and against the float32-accumulating atomic path (``grad_value_f32``)
    rounded once; samples outside the maps, on their borders, and many taps on the same pixels (every query of a head at the
    same reference point -- "collapsed": ALL queries at the same points, "one_pixel": every tap of a level in one pixel cell,
    "outside": no tap in any map); bit-reproducible."""
    shapes = np.array(ENC_SHAPES, dtype=np.int64)
    S = int(shapes.prod(1).sum())
    M, D, L, P = 8, 48, 3, 4
    g = torch.Generator(device="cpu").manual_seed(N * 100 + Lq)
    value = torch.randn(N, S, M, D, generator=g).bfloat16()
    loc = torch.rand(N, Lq, M, L, P, 2, generator=g) * 1.2 - 0.1              # some samples outside [0, 1]
    loc[:, : Lq // 2, 0] = loc[:, :1, 0]                                      # head 0: half of the queries share their taps
    if mode == "collapsed":
        loc = loc[:, :1].expand(-1, Lq, -1, -1, -1, -1).clone()
    elif mode == "one_pixel":
        loc = loc[:, :1, :, :, :1].expand(-1, Lq, -1, -1, P, -1).clamp(0.1, 0.9).clone()
    elif mode == "outside":
        loc = loc + 2.0
    loc = (torch.round(loc * 4096) + 0.5) / 4096
    attn = torch.softmax(torch.randn(N, Lq, M, L * P, generator=g), -1).view(N, Lq, M, L, P)
    go = torch.randn(N, Lq, M * D, generator=g).bfloat16()
    ts = torch.from_numpy(shapes).to(DEV)
    ti = lsi_of(torch.from_numpy(shapes)).to(DEV)
    args = (value.to(DEV), ts, ti, loc.to(DEV), attn.to(DEV), go.to(DEV), 64)
    gv, gl, ga = MSDA.ms_deform_attn_backward(*args)
    assert _lib.last_variant() == "d48_sparse", _lib.last_variant()
    assert gv.dtype == torch.bfloat16
    gv32, gl2, ga2 = MSDA.ms_deform_attn_backward(*args, grad_value_f32=True)
    assert _lib.last_variant() == "d48_lp12"
    assert torch.equal(gl, gl2) and torch.equal(ga, ga2)                      # (the same kernel makes them either way)
    # one bf16 rounding of sums that differ only in their float32 summation order
    diff = (gv.float() - gv32.bfloat16().float()).abs()
    assert float(diff.max()) <= 2 ** -7 * float(gv32.abs().max()) + 1e-6
    assert float((diff > 0).float().mean()) < 0.02
    if mode == "outside":
        assert not bool(gv.any())
    if N <= 3:
        ref_gv, _, _ = O.core_c_backward(value.float().numpy().astype(np.float64), shapes, ti.cpu().numpy(),
                                         loc.numpy().astype(np.float64), attn.numpy().astype(np.float64),
                                         go.float().numpy().astype(np.float64))
        np.testing.assert_allclose(gv.float().cpu().numpy(), ref_gv, rtol=2 ** -7, atol=2e-2)
    again = MSDA.ms_deform_attn_backward(*args)[0]
    assert torch.equal(gv, again)
