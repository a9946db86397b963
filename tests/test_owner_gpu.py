"""GPU tests of the owner-computes backward (csrc/msda_d48_patch.cuh), ``-m gpu``.

The path is taken when the host knows the level shapes, D == 48 and Lq == S (encoder).  Its result
must be the same function of the inputs for ANY locations: near samples are accumulated in LDS by
the tile owners, far ones by HBM atomics; the split must be a partition.
"""
import numpy as np
import pytest
import torch

from oracle import msda_oracle as O
from snipper_amd import MultiScaleDeformableAttention as MSDA
from snipper_amd import _lib

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(autouse=True)
def _owner_enabled():
    """The owner-computes path is the library default (csrc/msda_capi.hip); make sure a test that switched it off
    cannot leak into this file, and leave the default behind."""
    _lib.set_param("owner_enable", 1)
    yield
    _lib.set_param("owner_enable", 1)


def grid_case(N, shapes, M, P, seed, spread_px, frac_far=0.0, dtype=np.float32):
    """Encoder-like inputs: Lq == S, query q sits on pixel q; offsets ~ N(0, spread_px) pixels, a
    fraction `frac_far` of samples uniformly anywhere (incl. outside the map)."""
    rng = np.random.RandomState(seed)
    shapes = np.asarray(shapes, dtype=np.int64)
    L = len(shapes)
    S = int((shapes[:, 0] * shapes[:, 1]).sum())
    refs = []
    for h, w in shapes:
        ys, xs = np.meshgrid(np.arange(h) + 0.5, np.arange(w) + 0.5, indexing="ij")
        refs.append(np.stack([xs.reshape(-1) / w, ys.reshape(-1) / h], -1))
    ref = np.concatenate(refs)[None, :, None, None, None, :]
    norm = np.array([[w, h] for h, w in shapes], dtype=np.float64)[None, None, None, :, None, :]
    loc = ref + rng.standard_normal((N, S, M, L, P, 2)) * spread_px / norm
    far = rng.uniform(size=(N, S, M, L, P, 1)) < frac_far
    loc = np.where(far, rng.uniform(-0.2, 1.2, loc.shape), loc)
    loc = ((np.round(loc * 4096) + 0.5) / 4096).astype(dtype)     # exact pixel coordinates (see test_msda_gpu)
    value = rng.standard_normal((N, S, M, 48)).astype(dtype)
    attn = rng.uniform(0, 1, (N, S, M, L, P)).astype(dtype)
    attn /= attn.sum((-1, -2), keepdims=True)
    go = rng.standard_normal((N, S, M * 48)).astype(dtype)
    return value, shapes, O.level_start_index(shapes), loc, attn, go


def run_hip(v, shapes, lsi, loc, attn, go, host_shapes):
    t = lambda a: torch.from_numpy(a).to(DEV)
    out = MSDA.ms_deform_attn_backward(t(v), t(shapes), t(lsi), t(loc), t(attn), t(go), 64, host_shapes=host_shapes)
    return [x.cpu().numpy() for x in out], _lib.last_variant()


CASES = {
    "snipper_small_local": (2, [(19, 25), (10, 13), (5, 7)], 8, 4, 2.0, 0.0),
    "snipper_small_mixed": (2, [(19, 25), (10, 13), (5, 7)], 8, 4, 3.0, 0.2),
    "all_far": (1, [(19, 25), (10, 13), (5, 7)], 3, 4, 1.0, 1.0),
    "big_level_tiles16": (1, [(70, 67), (35, 34)], 2, 4, 4.0, 0.05),       # 4690 px -> 16x16 tiles, ragged edges
    "single_level": (3, [(9, 31)], 5, 4, 2.5, 0.1),
    "four_levels": (1, [(24, 20), (12, 10), (6, 5), (3, 3)], 4, 4, 2.0, 0.1),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_owner_backward_matches_oracle(name):
    N, shapes, M, P, spread, far = CASES[name]
    v, sh, lsi, loc, attn, go = grid_case(N, shapes, M, P, seed=len(name), spread_px=spread, frac_far=far)
    f64 = lambda a: a.astype(np.float64)
    ref = O.core_c_backward(f64(v), sh, lsi, f64(loc), f64(attn), f64(go), threads=4)
    (gv, gl, ga), variant = run_hip(v, sh, lsi, loc, attn, go, [tuple(x) for x in sh.tolist()])
    assert variant == "d48_owner", variant
    np.testing.assert_allclose(gv, ref[0], rtol=1e-4, atol=5e-5)
    s = float(np.abs(ref[1]).max())
    np.testing.assert_allclose(gl / s, ref[1] / s, rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(ga, ref[2], rtol=1e-4, atol=1e-4)
    # and against the atomics-only path of the same library (no host shapes)
    (gv2, gl2, ga2), variant2 = run_hip(v, sh, lsi, loc, attn, go, None)
    assert variant2.startswith("d48") and "owner" not in variant2
    np.testing.assert_allclose(gv, gv2, rtol=1e-4, atol=5e-5)
    np.testing.assert_allclose(gl / s, gl2 / s, rtol=1e-5, atol=1e-6)   # same math, other summation order
    np.testing.assert_allclose(ga, ga2, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("radius", [0.0, 0.75, 3.0, 40.0])
@pytest.mark.parametrize("edges", [(16, 8, 4), (8, 8, 8), (4, 4, 2), (16, 16, 16), (1, 1, 1)])
def test_partition_holds_for_any_radius_and_tiling(radius, edges):
    """Whatever the near radius / tile sizes, near + far must add up to the same gradient."""
    v, sh, lsi, loc, attn, go = grid_case(2, [(21, 26), (11, 13), (6, 7)], 4, 4, seed=5, spread_px=2.5, frac_far=0.1)
    f64 = lambda a: a.astype(np.float64)
    ref = O.core_c_backward(f64(v), sh, lsi, f64(loc), f64(attn), f64(go), threads=4)[0]
    try:
        _lib.set_param("near_radius", radius)
        for k, e in zip(("big", "mid", "small"), edges):
            _lib.set_param(f"owner_tile_edge_{k}", e)
        (gv, _, _), variant = run_hip(v, sh, lsi, loc, attn, go, [tuple(x) for x in sh.tolist()])
    finally:
        _lib.reset_config()
    assert variant == "d48_owner"
    np.testing.assert_allclose(gv, ref, rtol=1e-4, atol=5e-5)


def test_owner_not_taken_when_not_encoder_shape():
    v, sh, lsi, loc, attn, go = grid_case(1, [(9, 8), (4, 4)], 2, 4, seed=1, spread_px=1.0)
    Lq = 10
    (_, _, _), variant = run_hip(v, sh, lsi, loc[:, :Lq].copy(), attn[:, :Lq].copy(), go[:, :Lq].copy(),
                                 [tuple(x) for x in sh.tolist()])
    assert "owner" not in variant
    with pytest.raises(RuntimeError):
        _lib.set_param("no_such_knob", 1.0)


def test_full_size_encoder_backward_owner_vs_atomics():
    """BASELINE geometry (600x800 -> 9875 tokens, N=2): owner path == atomics path, and the gradient
    checksum identity holds:  sum(grad_value) == <out(value=1), grad_out>."""
    shapes = [(75, 100), (38, 50), (19, 25)]
    v, sh, lsi, loc, attn, go = grid_case(2, shapes, 8, 4, seed=3, spread_px=3.0, frac_far=0.01)
    (gv, gl, ga), variant = run_hip(v, sh, lsi, loc, attn, go, shapes)
    assert variant == "d48_owner"
    (gv2, gl2, ga2), _ = run_hip(v, sh, lsi, loc, attn, go, None)
    np.testing.assert_allclose(gv, gv2, rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(gl, gl2, rtol=1e-4, atol=1e-3)
    t = lambda a: torch.from_numpy(a).to(DEV)
    ones = torch.ones_like(t(v))
    out1 = MSDA.ms_deform_attn_forward(ones, t(sh), t(lsi), t(loc), t(attn), 64)
    (gv1, _, _), _ = run_hip(np.ones_like(v), sh, lsi, loc, attn, go, shapes)
    lhs, rhs = float(gv1.astype(np.float64).sum()), float((out1.double() * t(go).double()).sum())
    assert abs(lhs - rhs) <= 1e-5 * float(np.abs(go).sum())


GEOMETRIES = {
    "600x800": [(75, 100), (38, 50), (19, 25)],          # BASELINE configs[2]/[3]: S = 9875
    "540x960": [(68, 120), (34, 60), (17, 30)],          # the README's JTA / Panoptic recipe: S = 10710
}


@pytest.mark.parametrize("geom", sorted(GEOMETRIES))
@pytest.mark.parametrize("spread,far", [(2.0, 0.0), (3.0, 0.2), (8.0, 0.5)], ids=["local", "far20", "wide_far50"])
def test_full_size_owner_backward_directly_against_oracle(geom, spread, far):
    """The dominant kernels of the training step (owner-computes backward, D=48 forward) at FULL map size, N=1,
    compared DIRECTLY with the C oracle (float64), not through the atomic kernel."""
    shapes = GEOMETRIES[geom]
    v, sh, lsi, loc, attn, go = grid_case(1, shapes, 8, 4, seed=11, spread_px=spread, frac_far=far)
    f64 = lambda a: a.astype(np.float64)
    ref = O.core_c_backward(f64(v), sh, lsi, f64(loc), f64(attn), f64(go), threads=32)
    (gv, gl, ga), variant = run_hip(v, sh, lsi, loc, attn, go, shapes)
    assert variant == "d48_owner"
    np.testing.assert_allclose(gv, ref[0], rtol=1e-4, atol=2e-4)
    s = float(np.abs(ref[1]).max())
    np.testing.assert_allclose(gl / s, ref[1] / s, rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(ga, ref[2], rtol=1e-4, atol=1e-4)
    # forward of the same inputs
    t = lambda a: torch.from_numpy(a).to(DEV)
    ref_out = O.core_c_forward(f64(v), sh, lsi, f64(loc), f64(attn), threads=32)
    out = MSDA.ms_deform_attn_forward(t(v), t(sh), t(lsi), t(loc), t(attn), 64, host_shapes=shapes).cpu().numpy()
    assert _lib.last_variant() == "d48_lp12", _lib.last_variant()
    np.testing.assert_allclose(out, ref_out, rtol=1e-4, atol=2e-5)
    out16 = MSDA.ms_deform_attn_forward(t(v), t(sh), t(lsi), t(loc), t(attn), 64, out_bf16=True, host_shapes=shapes)
    assert out16.dtype == torch.bfloat16
    assert torch.equal(out16.cpu(), torch.from_numpy(out).to(torch.bfloat16))
    # the atomic kernel (no host shapes) at the same size
    (gv2, _, _), variant2 = run_hip(v, sh, lsi, loc, attn, go, None)
    assert "owner" not in variant2
    np.testing.assert_allclose(gv2, ref[0], rtol=1e-4, atol=2e-4)


def test_owner_backward_is_bit_reproducible_for_near_samples():
    """All-near inputs (no far taps, hence no HBM float atomics) must give identical bits from launch to launch, for
    float32 and for bfloat16 grad_out rows: no float sum of the owner-computes kernels depends on an arrival order."""
    shapes = GEOMETRIES["600x800"]
    v, sh, lsi, loc, attn, go = grid_case(2, shapes, 8, 4, seed=4, spread_px=1.5, frac_far=0.0)
    (a, _, _), variant = run_hip(v, sh, lsi, loc, attn, go, shapes)
    assert variant == "d48_owner"
    for _ in range(3):
        (b, _, _), _ = run_hip(v, sh, lsi, loc, attn, go, shapes)
        assert np.array_equal(a, b)
    t = lambda x: torch.from_numpy(x).to(DEV)
    outs = [MSDA.ms_deform_attn_backward(t(v), t(sh), t(lsi), t(loc), t(attn), t(go).to(torch.bfloat16), 64,
                                         host_shapes=shapes)[0] for _ in range(3)]
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    np.testing.assert_allclose(outs[0].cpu().numpy(), a, rtol=0.02, atol=0.05)      # bf16-rounded grad_out rows


@pytest.mark.parametrize("geom", ["small", "600x800"])
@pytest.mark.parametrize("spread,far", [(2.0, 0.0), (4.0, 0.3)], ids=["local", "far30"])
def test_bf16_value_owner_backward_against_oracle(geom, spread, far):
    """bfloat16 ``value`` (and grad_out rows) through the owner-computes kernels (v_dot2c_f32_bf16 dot products, float32
    everything else) against the C oracle in float64 on the SAME bf16-rounded value / grad_out."""
    shapes = GEOMETRIES[geom] if geom in GEOMETRIES else [(19, 25), (10, 13), (5, 7)]
    v, sh, lsi, loc, attn, go = grid_case(1 if geom != "small" else 2, shapes, 8, 4, seed=21, spread_px=spread, frac_far=far)
    t = lambda a: torch.from_numpy(a).to(DEV)
    v16, go16 = t(v).to(torch.bfloat16), t(go).to(torch.bfloat16)
    f64 = lambda a: a.astype(np.float64)
    ref_out = O.core_c_forward(f64(v16.float().cpu().numpy()), sh, lsi, f64(loc), f64(attn), threads=32)
    ref = O.core_c_backward(f64(v16.float().cpu().numpy()), sh, lsi, f64(loc), f64(attn), f64(go16.float().cpu().numpy()),
                            threads=32)
    out = MSDA.ms_deform_attn_forward(v16, t(sh), t(lsi), t(loc), t(attn), 64, host_shapes=shapes)
    assert out.dtype == torch.bfloat16
    np.testing.assert_allclose(out.float().cpu().numpy(), ref_out, rtol=2 ** -7, atol=2e-2)
    gv, gl, ga = MSDA.ms_deform_attn_backward(v16, t(sh), t(lsi), t(loc), t(attn), go16, 64, host_shapes=shapes,
                                              grad_value_f32=True)
    assert _lib.last_variant() == "d48_owner_mfma" and gv.dtype == torch.float32      # bf16 rows: msda_d48_tilemm.cuh
    np.testing.assert_allclose(gv.cpu().numpy(), ref[0], rtol=1e-4, atol=2e-4)
    s = float(np.abs(ref[1]).max())
    np.testing.assert_allclose(gl.cpu().numpy() / s, ref[1] / s, rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(ga.cpu().numpy(), ref[2], rtol=1e-4, atol=1e-4)
    gv2 = MSDA.ms_deform_attn_backward(v16, t(sh), t(lsi), t(loc), t(attn), go16, 64, host_shapes=shapes)[0]
    assert gv2.dtype == torch.bfloat16


# ---- the matrix-pipe tile kernel (csrc/msda_d48_tilemm.cuh): taken for bfloat16 grad_out rows ------------------------
def _bf16_rows_case(name):
    N, shapes, M, P, spread, far = CASES[name]
    v, sh, lsi, loc, attn, go = grid_case(N, shapes, M, P, seed=len(name) + 40, spread_px=spread, frac_far=far)
    go16 = torch.from_numpy(go).to(torch.bfloat16)
    return v, sh, lsi, loc, attn, go16, go16.float().numpy()


def _run_rows16(v, sh, lsi, loc, attn, go16, host_shapes):
    t = lambda a: torch.from_numpy(a).to(DEV)
    out = MSDA.ms_deform_attn_backward(t(v), t(sh), t(lsi), t(loc), t(attn), go16.to(DEV), 64, host_shapes=host_shapes)
    return [x.cpu().numpy() for x in out], _lib.last_variant()


@pytest.mark.parametrize("name", sorted(CASES))
def test_mfma_tile_kernel_matches_oracle(name):
    """float32 value, bfloat16 grad_out rows (the training step's mode): grad_value by the dense per-tile scatter on the
    matrix pipe, against the C oracle in float64 on the SAME bf16-rounded rows, at the float32 kernels' tolerance (the
    weights are split into bf16 hi + lo parts: float32-class), and against the vector tile kernel of the same library."""
    v, sh, lsi, loc, attn, go16, go_r = _bf16_rows_case(name)
    f64 = lambda a: a.astype(np.float64)
    ref = O.core_c_backward(f64(v), sh, lsi, f64(loc), f64(attn), f64(go_r), threads=4)
    hs = [tuple(x) for x in sh.tolist()]
    (gv, gl, ga), variant = _run_rows16(v, sh, lsi, loc, attn, go16, hs)
    assert variant == "d48_owner_mfma", variant
    np.testing.assert_allclose(gv, ref[0], rtol=1e-4, atol=5e-5)
    s = float(np.abs(ref[1]).max())
    np.testing.assert_allclose(gl / s, ref[1] / s, rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(ga, ref[2], rtol=1e-4, atol=1e-4)
    try:
        _lib.set_param("tile_kernel", 1)
        (gv1, gl1, ga1), variant1 = _run_rows16(v, sh, lsi, loc, attn, go16, hs)
    finally:
        _lib.reset_config()
    assert variant1 == "d48_owner", variant1
    np.testing.assert_allclose(gv, gv1, rtol=1e-4, atol=5e-5)
    assert np.array_equal(gl, gl1) and np.array_equal(ga, ga1)          # the query side is the same kernel


@pytest.mark.parametrize("radius", [0.0, 3.0, 40.0])
@pytest.mark.parametrize("edges", [(16, 8, 4), (8, 8, 8), (4, 4, 2), (16, 16, 16), (1, 1, 1)])
def test_mfma_tile_kernel_partition_holds_for_any_radius_and_tiling(radius, edges):
    v, sh, lsi, loc, attn, go = grid_case(2, [(21, 26), (11, 13), (6, 7)], 4, 4, seed=6, spread_px=2.5, frac_far=0.1)
    go16 = torch.from_numpy(go).to(torch.bfloat16)
    f64 = lambda a: a.astype(np.float64)
    ref = O.core_c_backward(f64(v), sh, lsi, f64(loc), f64(attn), f64(go16.float().numpy()), threads=4)[0]
    try:
        _lib.set_param("near_radius", radius)
        for k, e in zip(("big", "mid", "small"), edges):
            _lib.set_param(f"owner_tile_edge_{k}", e)
        (gv, _, _), variant = _run_rows16(v, sh, lsi, loc, attn, go16, [tuple(x) for x in sh.tolist()])
    finally:
        _lib.reset_config()
    assert variant == "d48_owner_mfma"
    np.testing.assert_allclose(gv, ref, rtol=1e-4, atol=5e-5)


@pytest.mark.parametrize("edges", [(16, 8, 4), (16, 16, 16)])
@pytest.mark.parametrize("spread,far", [(2.0, 0.0), (8.0, 0.3)], ids=["local", "wide_far30"])
def test_mfma_tile_kernel_full_size_against_oracle(edges, spread, far):
    """600x800 geometry, N = 1, directly against the C oracle; tiles whose hit list needs several passes (16 x 16 tiles on
    the coarse levels are reached by most queries)."""
    shapes = GEOMETRIES["600x800"]
    v, sh, lsi, loc, attn, go = grid_case(1, shapes, 8, 4, seed=12, spread_px=spread, frac_far=far)
    go16 = torch.from_numpy(go).to(torch.bfloat16)
    f64 = lambda a: a.astype(np.float64)
    ref = O.core_c_backward(f64(v), sh, lsi, f64(loc), f64(attn), f64(go16.float().numpy()), threads=32)[0]
    try:
        for k, e in zip(("big", "mid", "small"), edges):
            _lib.set_param(f"owner_tile_edge_{k}", e)
        (gv, _, _), variant = _run_rows16(v, sh, lsi, loc, attn, go16, shapes)
        (gvb, _, _), _ = _run_rows16(v, sh, lsi, loc, attn, go16, shapes)
    finally:
        _lib.reset_config()
    assert variant == "d48_owner_mfma"
    np.testing.assert_allclose(gv, ref, rtol=1e-4, atol=2e-4)
    if far == 0.0:
        assert np.array_equal(gv, gvb)          # no far taps, no atomics: bit-reproducible


def test_owner_backward_writes_every_element_of_grad_value():
    """The owner-computes path no longer zeroes grad_value: the tile kernels store every pixel of every (n, m) plainly and
    the far kernel adds on top.  Poison the memory the caching allocator is about to hand out (NaN) and compare with the
    oracle, for float32 rows (vector tile kernel) and bf16 rows (matrix-pipe kernels), with far taps present."""
    N, shapes, M, P = 2, [(37, 29), (19, 15), (10, 8)], 4, 4
    v, sh, lsi, loc, attn, go = grid_case(N, shapes, M, P, seed=11, spread_px=2.5, frac_far=0.1)
    f64 = lambda a: a.astype(np.float64)
    ref = O.core_c_backward(f64(v), sh, lsi, f64(loc), f64(attn), f64(go), threads=4)
    t = lambda a: torch.from_numpy(a).to(DEV)
    hs = [tuple(x) for x in sh.tolist()]
    for rows16 in (False, True):
        gt = t(go).to(torch.bfloat16) if rows16 else t(go)
        ref_gv = ref[0]
        if rows16:
            ref_gv = O.core_c_backward(f64(v), sh, lsi, f64(loc), f64(attn), f64(gt.float().cpu().numpy()), threads=4)[0]
        for _ in range(3):
            poison = torch.full((N * sum(h * w for h, w in shapes) * M * 48 + 4096,), float("nan"), device=DEV)
            del poison                     # back to the allocator: the next float32 block of this size is this one
            gv, _, _ = MSDA.ms_deform_attn_backward(t(v), t(sh), t(lsi), t(loc), t(attn), gt, 64, host_shapes=hs,
                                                    grad_value_f32=True)
            assert _lib.last_variant().startswith("d48_owner")
            assert torch.isfinite(gv).all()
            np.testing.assert_allclose(gv.cpu().numpy(), ref_gv, rtol=1e-4, atol=5e-5)


def test_head_major_value_layout_equals_the_reference_layout():
    """snipper_msda_config.value_layout = 1: value / grad_value as [N, M, S, D].  The same inputs in the two layouts must
    give the same forward rows and the same gradients (grad_value compared after permuting back; the owned taps bit for
    bit -- the layout only changes addresses, not the order of any sum -- everything within the usual tolerance because the
    far taps' atomics are unordered).  bf16 value with bf16 rows (matrix-pipe tile kernels) and float32 value (vector kernel)."""
    from snipper_amd.fused import _head_major_config
    N, shapes, M, P = 2, [(37, 29), (19, 15), (10, 8)], 4, 4
    v, sh, lsi, loc, attn, go = grid_case(N, shapes, M, P, seed=5, spread_px=2.0, frac_far=0.05)
    t = lambda a: torch.from_numpy(a).to(DEV)
    hs = [tuple(x) for x in sh.tolist()]
    cfg = _head_major_config()
    for dtype in (torch.bfloat16, torch.float32):
        val = t(v).to(dtype)
        got = t(go).to(dtype)
        val_hm = val.permute(0, 2, 1, 3).contiguous().view(val.shape)          # memory [N, M, S, D] under the logical shape
        out = MSDA.ms_deform_attn_forward(val, t(sh), t(lsi), t(loc), t(attn), 64, host_shapes=hs)
        out_hm = MSDA.ms_deform_attn_forward(val_hm, t(sh), t(lsi), t(loc), t(attn), 64, host_shapes=hs, config=cfg)
        assert torch.equal(out, out_hm)
        gv, gl, ga = MSDA.ms_deform_attn_backward(val, t(sh), t(lsi), t(loc), t(attn), got, 64, host_shapes=hs, grad_value_f32=True)
        assert _lib.last_variant().startswith("d48_owner")
        gv2, gl2, ga2 = MSDA.ms_deform_attn_backward(val_hm, t(sh), t(lsi), t(loc), t(attn), got, 64, host_shapes=hs,
                                                     grad_value_f32=True, config=cfg)
        assert _lib.last_variant().startswith("d48_owner")
        S = val.shape[1]
        gv2 = gv2.view(N, M, S, 48).permute(0, 2, 1, 3)
        assert torch.equal(gl, gl2) and torch.equal(ga, ga2)
        torch.testing.assert_close(gv2, gv.view(N, S, M, 48), rtol=1e-5, atol=1e-5)
    # shapes without a tuned kernel refuse the layout instead of reading the wrong one
    val71 = torch.randn(1, 9 * 31, 5, 71, device=DEV)
    sh1 = torch.tensor([[9, 31]], device=DEV)
    with pytest.raises(RuntimeError):
        MSDA.ms_deform_attn_forward(val71, sh1, torch.zeros(1, dtype=torch.int64, device=DEV),
                                    torch.rand(1, 7, 5, 1, 4, 2, device=DEV), torch.rand(1, 7, 5, 1, 4, device=DEV), 64, config=cfg)


def test_tile_walk_orders_give_identical_results():
    """The order in which the matrix-pipe tile kernel's workgroups take the tiles -- tile-major (default), pair-major (debug
    bit 64), region by region (debug bit 128: csrc/msda_d48_tilemm.cuh, t3_region_order) -- is a scheduling choice: every
    gradient must come out bit for bit the same (all taps near, so no unordered atomics).  Debug bits need
    SNIPPER_MSDA_ALLOW_DEBUG=1 at library load: a child process."""
    import os, subprocess, sys, textwrap
    code = textwrap.dedent("""
        import numpy as np, torch, sys
        sys.path.insert(0, "tests")
        from test_owner_gpu import grid_case, MSDA, _lib, DEV
        N, shapes, M, P = 2, [(75, 100), (38, 50), (19, 25)], 8, 4
        v, sh, lsi, loc, attn, go = grid_case(N, shapes, M, P, seed=11, spread_px=2.0, frac_far=0.0)
        t = lambda a: torch.from_numpy(a).to(DEV)
        hs = [tuple(x) for x in sh.tolist()]
        val, got = t(v).to(torch.bfloat16), t(go).to(torch.bfloat16)
        res = []
        for dbg in (0, 64, 128, 192):
            cfg = _lib.Config.defaults()
            cfg.debug_ablation, cfg.tile_kernel = dbg, 2        # (2 = the matrix-pipe tile kernel also with debug bits set)
            out = MSDA.ms_deform_attn_backward(val, t(sh), t(lsi), t(loc), t(attn), got, 64, host_shapes=hs,
                                               grad_value_f32=True, config=cfg)
            assert _lib.last_variant().startswith("d48_owner_mfma"), _lib.last_variant()
            res.append([o.clone() for o in out])
        for r in res[1:]:
            assert all(torch.equal(a, b) for a, b in zip(res[0], r))
        print("ok")
    """)
    env = dict(os.environ, SNIPPER_MSDA_ALLOW_DEBUG="1")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600,
                         cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert out.returncode == 0 and "ok" in out.stdout, (out.stdout[-500:], out.stderr[-2000:])
