#!/usr/bin/env python3
"""Development aid: the encoder-shape kernels (csrc/msda_d48_patch.cuh) against the C oracle on a sweep of geometries
and localities, printing the maximum error of every output (no asserts: a broken kernel should show WHERE it breaks)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from oracle import msda_oracle as O                              # noqa: E402
from snipper_amd import MultiScaleDeformableAttention as MSDA   # noqa: E402
from snipper_amd import _lib                                     # noqa: E402
from test_owner_gpu import grid_case                             # noqa: E402

DEV = "cuda:0"
CASES = [
    ("tiny", 1, [(9, 8), (4, 4)], 2, 1.0, 0.0),
    ("small_local", 2, [(19, 25), (10, 13), (5, 7)], 8, 2.0, 0.0),
    ("small_mixed", 2, [(19, 25), (10, 13), (5, 7)], 8, 3.0, 0.2),
    ("all_far", 1, [(19, 25), (10, 13), (5, 7)], 3, 1.0, 1.0),
    ("big16", 1, [(70, 67), (35, 34)], 2, 4.0, 0.05),
    ("single", 3, [(9, 31)], 5, 2.5, 0.1),
    ("four", 1, [(24, 20), (12, 10), (6, 5), (3, 3)], 4, 2.0, 0.1),
    ("full", 1, [(75, 100), (38, 50), (19, 25)], 8, 3.0, 0.02),
]
f64 = lambda a: a.astype(np.float64)
t = lambda a: torch.from_numpy(a).to(DEV)
for name, N, shapes, M, spread, far in CASES:
    v, sh, lsi, loc, attn, go = grid_case(N, shapes, M, 4, seed=len(name), spread_px=spread, frac_far=far)
    hs = [tuple(x) for x in sh.tolist()]
    ref_o = O.core_c_forward(f64(v), sh, lsi, f64(loc), f64(attn), threads=16)
    ref = O.core_c_backward(f64(v), sh, lsi, f64(loc), f64(attn), f64(go), threads=16)
    out = MSDA.ms_deform_attn_forward(t(v), t(sh), t(lsi), t(loc), t(attn), 64, host_shapes=hs).cpu().numpy()
    vf = _lib.last_variant()
    cfg = None
    gv, gl, ga = [x.cpu().numpy() for x in MSDA.ms_deform_attn_backward(t(v), t(sh), t(lsi), t(loc), t(attn), t(go), 64,
                                                                        host_shapes=hs, config=cfg)]
    vb = _lib.last_variant()
    gv2 = MSDA.ms_deform_attn_backward(t(v), t(sh), t(lsi), t(loc), t(attn), t(go), 64, host_shapes=hs,
                                       config=cfg)[0].cpu().numpy()
    s = float(np.abs(ref[1]).max())
    bad = np.argwhere(np.abs(gv - ref[0]) > 1e-3)
    print(f"{name:12s} fwd[{vf}] {np.abs(out - ref_o).max():.2e}  bwd[{vb}] gv {np.abs(gv - ref[0]).max():.2e} "
          f"gl {np.abs(gl - ref[1]).max() / s:.2e} ga {np.abs(ga - ref[2]).max():.2e}  reproducible {np.array_equal(gv, gv2)} "
          f"bad_gv {len(bad)} {bad[:3].tolist()}", flush=True)
