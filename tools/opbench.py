#!/usr/bin/env python3
"""Op-level timing of the deformable-attention core op on one MI355X (not the contract bench).

Encoder call (Lq=S=9875) and decoder call (Lq=60) at 600x800 geometry; HIP kernels (tuned and
generic) next to the PyTorch grid_sample formulation on the same GPU (the ">=3x" denominator of
BASELINE.json).  Prints one JSON line per measurement.
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if "--debug" in sys.argv:        # the wrong-result timing ablations are refused unless asked for before the library loads
    os.environ.setdefault("SNIPPER_MSDA_ALLOW_DEBUG", "1")
from snipper_amd import MultiScaleDeformableAttention as MSDA   # noqa: E402
from snipper_amd import _lib                                     # noqa: E402
from snipper_amd.ms_deform_attn_func import ms_deform_attn_core_pytorch  # noqa: E402

SHAPES = [(75, 100), (38, 50), (19, 25)]
M, D, L, P = 8, 48, 3, 4


def make(N, Lq, local, dtype, seed=0, dev="cuda:0", sigma=3.0, far=0.0, grid=False):
    g = torch.Generator().manual_seed(seed)
    S = sum(h * w for h, w in SHAPES)
    value = torch.randn(N, S, M, D, generator=g)
    if local and Lq == S:
        refs = []
        for h, w in SHAPES:
            ys, xs = torch.meshgrid(torch.arange(h) + 0.5, torch.arange(w) + 0.5, indexing="ij")
            refs.append(torch.stack([xs.reshape(-1) / w, ys.reshape(-1) / h], -1))
        ref = torch.cat(refs)[None, :, None, None, None, :]
        norm = torch.tensor([[w, h] for h, w in SHAPES], dtype=torch.float32)[None, None, None, :, None, :]
        off = torch.randn(N, Lq, M, L, P, 2, generator=g) * sigma
        if grid:          # the reference's offset bias: 8 directions (one per head) x point index 1..P (ms_deform_attn.py:82-90)
            import math
            th = torch.arange(M, dtype=torch.float32) * (2.0 * math.pi / M)
            d_ = torch.stack([th.cos(), th.sin()], -1)
            d_ = d_ / d_.abs().max(-1, keepdim=True)[0]
            off = off + (d_.view(1, 1, M, 1, 1, 2) * torch.arange(1, P + 1, dtype=torch.float32).view(1, 1, 1, 1, P, 1))
        loc = ref + off / norm
        if far > 0:
            pick = torch.rand(N, Lq, M, L, P, 1, generator=g) < far
            loc = torch.where(pick, torch.rand(N, Lq, M, L, P, 2, generator=g) * 1.4 - 0.2, loc)
    else:
        loc = torch.rand(N, Lq, M, L, P, 2, generator=g)
    attn = torch.softmax(torch.randn(N, Lq, M, L * P, generator=g), -1).view(N, Lq, M, L, P)
    go = torch.randn(N, Lq, M * D, generator=g)
    shapes = torch.tensor(SHAPES, dtype=torch.long, device=dev)
    lsi = torch.cat((shapes.new_zeros(1), shapes.prod(1).cumsum(0)[:-1]))
    return value.to(dev).to(dtype), shapes, lsi, loc.to(dev), attn.to(dev), go.to(dev).to(dtype)


def timeit(fn, iters, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
    ev[0].record()
    for i in range(iters):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(iters))
    return ts[len(ts) // 2], ts[0]


def alg_bytes(N, Lq, S, e, bwd):
    if not bwd:
        return e * (N * S * M * D + N * Lq * M * D) + 4 * 3 * N * Lq * M * L * P
    return e * (2 * N * S * M * D + N * Lq * M * D) + 4 * 6 * N * Lq * M * L * P


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--N", type=int, nargs="+", default=[1, 2, 8])
    ap.add_argument("--skip-torch", action="store_true")
    ap.add_argument("--owner", type=int, default=1, help="pass host shapes (owner-computes backward)")
    ap.add_argument("--radius", type=float, default=None)
    ap.add_argument("--debug", type=int, default=0, help="timing ablations of the encoder-shape kernels (wrong results)")
    ap.add_argument("--edges", type=int, nargs=3, default=None)
    ap.add_argument("--tile-kernel", type=int, default=None, help="owner-computes grad_value side for bf16 rows: 1 = vector / LDS kernel, 2 = matrix pipe (default)")
    ap.add_argument("--value-layout", type=int, default=0, help="1 = address value / grad_value head-major (timing: the buffers keep their shape)")
    ap.add_argument("--cases", nargs="+", default=["enc_local", "enc_uniform", "dec"])
    ap.add_argument("--dtypes", nargs="+", default=["float32", "bfloat16"])
    ap.add_argument("--sigma", type=float, nargs="+", default=[3.0], help="enc_local: std of the offsets in pixels")
    ap.add_argument("--far", type=float, nargs="+", default=[0.0], help="enc_local: fraction of samples placed anywhere")
    ap.add_argument("--grid", type=int, default=0, help="enc_local: 1 = add the reference's 8-direction x (1..P) offset bias grid")
    ap.add_argument("--rows-bf16", type=int, default=0, help="1 = bf16 out / grad_out rows beside f32 value (the step's mode)")
    args = ap.parse_args()
    S = sum(h * w for h, w in SHAPES)
    if args.radius is not None:
        _lib.set_param("near_radius", args.radius)
    if args.debug:
        _lib.set_param("debug", args.debug)
    if args.tile_kernel is not None:
        _lib.set_param("tile_kernel", args.tile_kernel)
    if args.value_layout:
        _lib.set_param("value_layout", args.value_layout)
    if args.edges:
        for k, e in zip(("big", "mid", "small"), args.edges):
            _lib.set_param(f"owner_tile_edge_{k}", e)
    for N in args.N:
        for name, Lq, local in [("enc_local", S, True), ("enc_uniform", S, False), ("dec", 60, False)]:
            if name not in args.cases:
                continue
            sweeps = [(sg, fr) for sg in args.sigma for fr in args.far] if name == "enc_local" else [(3.0, 0.0)]
            for dtype in [getattr(torch, x) for x in args.dtypes]:
              for sg, fr in sweeps:
                v, shapes, lsi, loc, attn, go = make(N, Lq, local, dtype, sigma=sg, far=fr, grid=bool(args.grid))
                rows16 = bool(args.rows_bf16) and dtype == torch.float32
                if rows16:
                    go = go.to(torch.bfloat16)
                for policy in ((0, 2, 1) if dtype == torch.float32 else (0,)):
                    if policy == 1 and rows16:
                        continue
                    _lib.set_policy(policy)
                    hs = SHAPES if (policy == 0 and args.owner) else None
                    f = lambda: MSDA.ms_deform_attn_forward(v, shapes, lsi, loc, attn, 64, out_bf16=rows16, host_shapes=hs)
                    b = lambda: MSDA.ms_deform_attn_backward(v, shapes, lsi, loc, attn, go, 64, host_shapes=hs)
                    f(); var_f = _lib.last_variant()
                    b(); var_b = _lib.last_variant()
                    tf, tf0 = timeit(f, args.iters)
                    tb, tb0 = timeit(b, args.iters)
                    e = v.element_size()
                    print(json.dumps({
                        "case": name, "N": N, "sigma_px": sg, "far": fr, "grid": args.grid, "radius": args.radius, "edges": args.edges, "tile_kernel": args.tile_kernel, "rows_bf16": int(rows16),
                        "dtype": str(dtype).split(".")[-1], "fwd_variant": var_f,
                        "bwd_variant": var_b, "fwd_ms": round(tf, 4), "fwd_min_ms": round(tf0, 4),
                        "bwd_ms": round(tb, 4), "bwd_min_ms": round(tb0, 4),
                        "fwd_alg_GBps": round(alg_bytes(N, Lq, S, e, False) / tf / 1e6, 1),
                        "bwd_alg_GBps": round(alg_bytes(N, Lq, S, e, True) / tb / 1e6, 1)}), flush=True)
                _lib.set_policy(0)
            if not args.skip_torch:
                v, shapes, lsi, loc, attn, go = make(N, Lq, local, torch.float32)
                hw = [tuple(x) for x in SHAPES]
                vv, ll, aa = v.clone().requires_grad_(True), loc.clone().requires_grad_(True), attn.clone().requires_grad_(True)
                f = lambda: ms_deform_attn_core_pytorch(vv, hw, ll, aa)
                def fb():
                    o = ms_deform_attn_core_pytorch(vv, hw, ll, aa)
                    torch.autograd.grad(o, (vv, ll, aa), go)
                with torch.no_grad():
                    tf, _ = timeit(f, max(3, args.iters // 4), warm=1)
                tfb, _ = timeit(fb, max(3, args.iters // 4), warm=1)
                print(json.dumps({"case": name, "N": N, "dtype": "float32", "fwd_variant": "torch_grid_sample",
                                  "fwd_ms": round(tf, 3), "fwd_bwd_ms": round(tfb, 3)}), flush=True)


if __name__ == "__main__":
    main()
