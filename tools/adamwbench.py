#!/usr/bin/env python3
"""Optimizer phase on the bench's flat layout (41 M float32 parameters in three groups): FlatAdamW (two launches of
csrc/adamw_flat.cuh) against clip_grad_norm_ + torch.optim.AdamW(fused) on the same three flat tensors."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snipper_amd.flat_params import FlatAdamW, FlatParameters
dev = "cuda:0"
sizes = [17_500_000, 400_000, 23_300_000]          # ~ main / slow / backbone of the bench model
groups = [[torch.nn.Parameter(torch.randn(n, device=dev) * 0.02)] for n in sizes]
fp = FlatParameters(groups)
own = FlatAdamW(fp, [1e-4, 1e-5, 1e-5], weight_decay=1e-4)
ref = torch.optim.AdamW([{"params": [fp.leaf_of_group(i)], "lr": lr} for i, lr in enumerate([1e-4, 1e-5, 1e-5])], lr=1e-4,
                        weight_decay=1e-4, fused=True)
fp.grad_flat.normal_()
fp.bind_grads()
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
def torch_step():
    torch.nn.utils.clip_grad_norm_(fp.leaves, 0.1)
    ref.step()
nbytes = fp.flat.numel() * 4
us_own, us_ref = t(lambda: own.step(0.1)), t(torch_step)
print(json.dumps({"elements": fp.flat.numel(), "own_us": round(us_own, 1), "torch_us": round(us_ref, 1),
                  "own_GBps": round(8 * nbytes / us_own / 1e3, 1), "note": "8 array passes: g for the norm, then p g m v read and p m v written"}))
