#!/usr/bin/env python3
"""Full-width tiles (snipper_linear_wide_bf16) against the tile kernels at the feed-forward block's 1024-deep products on 79 000
rows: linear2's forward (bias) and linear1's data gradient.  SNIPPER_LINEAR_WIDE_MT=4 / 5 forces 128- / 160-row tiles."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snipper_amd.dense import linear_bf16, linear_nn_bf16, linear_pack_bf16, linear_wide_bf16

dev = "cuda:0"
torch.manual_seed(0)


def timeit(f, n=20):
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) * 1e3 / n, 1)


for M, K in [(79000, 1024), (79000, 512), (158000, 1024), (39500, 1024)]:
    x = torch.randn(M, K, device=dev).bfloat16()
    w = (torch.randn(384, K, device=dev) / K ** 0.5).bfloat16()
    b = torch.randn(384, device=dev)
    wt = (torch.randn(K, 384, device=dev) / K ** 0.5).bfloat16()
    p1 = torch.empty(w.numel(), dtype=torch.bfloat16, device=dev)
    p2 = torch.empty(w.numel(), dtype=torch.bfloat16, device=dev)
    linear_pack_bf16([(w, p1, False), (wt, p2, True)])
    rec = {"case": f"{M}x{K}->384", "mt": os.environ.get("SNIPPER_LINEAR_WIDE_MT", "auto"),
           "fwd_wide_us": timeit(lambda: linear_wide_bf16(x, p1, b)), "fwd_tile_us": timeit(lambda: linear_bf16(x, w, b)),
           "dgrad_wide_us": timeit(lambda: linear_wide_bf16(x, p2, None)), "dgrad_tile_us": timeit(lambda: linear_nn_bf16(x, wt))}
    rec["fwd_wide_tflops"] = round(2.0 * M * K * 384 / rec["fwd_wide_us"] / 1e6, 1)
    rec["fwd_wide_GBps_compulsory"] = round(2 * (M * K + M * 384) / rec["fwd_wide_us"] / 1e3, 1)
    print(json.dumps(rec), flush=True)
