#!/usr/bin/env python3
"""The patch kernel with one tap (snipper_linear_patch_bf16: X slices through LDS, W from a fragment-order pack) against the
tile kernels (snipper_linear_bf16 / snipper_linear_nn_bf16) at the step's large products: the ResNet body's 1x1 convolutions
(8 frames of 600 x 800) forward (bias + ReLU, conv3 with a residual) and data gradient (ReLU gate), and the feed-forward
block's K = 1024 products.  Kernel time by events; max |difference| between the two kernels on the same operands."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snipper_amd.dense import linear_bf16, linear_nn_bf16, linear_pack_bf16, linear_patch_bf16

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


# (rows, K, N, mode): fwd = bias + ReLU, res = bias + residual + ReLU, plain = bias, dgrad = dY[M, N] . W[N, K] with a ReLU gate
CASES = [(240000, 64, 64, "fwd"), (240000, 64, 256, "res"), (240000, 256, 64, "fwd"), (60000, 256, 128, "fwd"),
         (60000, 128, 512, "res"), (60000, 512, 128, "fwd"), (15200, 512, 256, "fwd"), (15200, 256, 1024, "res"),
         (15200, 1024, 256, "fwd"), (3800, 1024, 512, "fwd"), (3800, 512, 2048, "res"), (3800, 2048, 512, "fwd"),
         (79000, 1024, 384, "plain"), (79000, 384, 1024, "plain"),
         (240000, 64, 256, "dgrad"), (240000, 256, 64, "dgrad"), (60000, 128, 512, "dgrad"), (60000, 512, 128, "dgrad"),
         (15200, 256, 1024, "dgrad"), (15200, 1024, 256, "dgrad"), (3800, 512, 2048, "dgrad"), (3800, 2048, 512, "dgrad"),
         (79000, 1024, 384, "dgrad_plain")]
for M, K, N, mode in CASES:
    rec = {"case": f"{M}x{K}->{N} {mode}"}
    if mode.startswith("dgrad"):
        # forward layer K -> N; the data gradient multiplies dY [M, N] by W [N, K]
        w = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16()
        gy = torch.randn(M, N, device=dev).bfloat16()
        gate = torch.randn(M, K, device=dev).relu().bfloat16() if mode == "dgrad" else None
        packed = torch.empty(w.numel(), dtype=torch.bfloat16, device=dev)
        linear_pack_bf16([(w, packed, True)])
        old = lambda: linear_nn_bf16(gy, w, None, gate)
        flops = 2.0 * M * N * K
        outs = {}
        for bn in (64, 128):
            if K % bn:
                continue
            new = lambda: linear_patch_bf16(gy, packed, K, None, None, False, gate, bn)
            rec[f"patch{bn}_us"] = timeit(new)
            outs[bn] = new()
        rec["tile_us"] = timeit(old)
        ref = old()
    else:
        x = torch.randn(M, K, device=dev).bfloat16()
        w = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16()
        b = torch.randn(N, device=dev)
        res = torch.randn(M, N, device=dev).bfloat16() if mode == "res" else None
        relu = mode in ("fwd", "res")
        packed = torch.empty(w.numel(), dtype=torch.bfloat16, device=dev)
        linear_pack_bf16([(w, packed, False)])
        old = lambda: linear_bf16(x, w, b, res, relu)
        flops = 2.0 * M * N * K
        outs = {}
        for bn in (64, 128):
            if N % bn:
                continue
            new = lambda: linear_patch_bf16(x, packed, N, b, res, relu, None, bn)
            rec[f"patch{bn}_us"] = timeit(new)
            outs[bn] = new()
        rec["tile_us"] = timeit(old)
        ref = old()
    rec["max_abs_diff"] = max(float((o.float() - ref.float()).abs().max()) for o in outs.values())
    rec["ref_abs_max"] = float(ref.float().abs().max())
    best = min(v for k, v in rec.items() if k.startswith("patch"))
    rec["best_patch_tflops"] = round(flops / best / 1e6, 1)
    print(json.dumps(rec), flush=True)
