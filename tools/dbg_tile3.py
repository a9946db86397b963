#!/usr/bin/env python3
"""Debug aid: matrix-pipe tile kernel against the vector tile kernel on small encoder-shaped inputs."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snipper_amd import MultiScaleDeformableAttention as MSDA, _lib
from tests.test_owner_gpu import grid_case

DEV = "cuda:0"
shapes = [tuple(int(v) for v in s.split("x")) for s in (sys.argv[1:] or ["19x25", "10x13", "5x7"])]
for spread in (0.3, 2.0):
  for M in (1, 2):
    v, sh, lsi, loc, attn, go = grid_case(1, shapes, M, 4, seed=3, spread_px=spread, frac_far=0.0)
    t = lambda a: torch.from_numpy(a).to(DEV)
    go16 = t(go).to(torch.bfloat16)
    res = {}
    for tk in (1, 2):
        _lib.set_param("tile_kernel", tk)
        res[tk] = MSDA.ms_deform_attn_backward(t(v), t(sh), t(lsi), t(loc), t(attn), go16, 64, host_shapes=shapes)[0].cpu().numpy()
        print("variant", _lib.last_variant())
    _lib.reset_config()
    a, b = res[1][0], res[2][0]       # [S, M, 48]
    start = 0
    for (H, W) in shapes:
        da = np.abs(a[start:start + H * W] - b[start:start + H * W]).reshape(H, W, M, 48)
        ref = np.abs(a[start:start + H * W]).reshape(H, W, M, 48)
        bad = da.max(-1) > 1e-4 * (1 + ref.max(-1))
        print(f"spread {spread} M {M} level {H}x{W}: max diff {da.max():.3e}  bad pixels {int(bad.sum())} / {bad.size}")
        if bad.any() and H * W <= 500:
            for m in range(M):
                print(" head", m)
                for y in range(H):
                    print("  " + "".join("#" if bad[y, x, m] else "." for x in range(W)))
        start += H * W
