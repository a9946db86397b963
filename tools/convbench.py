#!/usr/bin/env python3
"""3x3 convolution micro-benchmark at the ResNet-50 body's shapes (8 frames of 600 x 800): forward, stride-1 data gradient
(flipped taps) and the stride-2 data gradient, kernel time by events.  SNIPPER_CONV_RING=0 / 2 selects the register-prefetch /
the LDS-DMA ring kernel for every shape (A/B runs); the two must agree bit for bit (same tiles, same summation order)."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snipper_amd.dense import conv3x3_bf16, conv3x3_dgrad_s2_bf16

SHAPES = [(64, 64, 150, 200, 1), (128, 128, 75, 100, 1), (256, 256, 38, 50, 1), (512, 512, 19, 25, 1),
          (128, 128, 150, 200, 2), (256, 256, 75, 100, 2), (512, 512, 38, 50, 2)]
dev = "cuda:0"
torch.manual_seed(0)
for cin, cout, h, w, st in SHAPES:
    x = torch.randn(8, cin, h, w, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    wt = (torch.randn(cout, cin, 3, 3, device=dev) * 0.05).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    b = torch.randn(cout, device=dev)
    run = lambda: conv3x3_bf16(x, wt, b, st, True)
    y = run()
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    flops = 2.0 * y.shape[0] * y.shape[2] * y.shape[3] * cin * cout * 9
    rec = {"case": f"{cin}->{cout} {h}x{w} s{st}", "us": round(us, 1), "tflops": round(flops / us / 1e6, 1),
           "frac_mfma": round(flops / us / 1e6 / 2500, 3), "ring": os.environ.get("SNIPPER_CONV_RING", "1"),
           "checksum": float(y.float().abs().sum())}
    if st == 2:
        gy = torch.randn_like(y)
        r2 = lambda: conv3x3_dgrad_s2_bf16(gy, wt.transpose(0, 1), (h, w))
        d = r2()
        for _ in range(3):
            r2()
        e0.record()
        for _ in range(20):
            r2()
        e1.record()
        torch.cuda.synchronize()
        rec["dgrad_s2_us"] = round(e0.elapsed_time(e1) * 1e3 / 20, 1)
        rec["dgrad_checksum"] = float(d.float().abs().sum())
    if st == 1 and os.environ.get("SNIPPER_CONV_PATCH", "1") != "0":
        from snipper_amd.dense import conv3x3_pack_bf16, conv3x3_patch_bf16
        packed = torch.empty(wt.numel(), dtype=torch.bfloat16, device=dev)
        conv3x3_pack_bf16([(wt, packed, False)])
        rp = lambda: conv3x3_patch_bf16(x, packed, cout, b, True)
        yp = rp()
        for _ in range(3):
            rp()
        e0.record()
        for _ in range(20):
            rp()
        e1.record()
        torch.cuda.synchronize()
        pus = e0.elapsed_time(e1) * 1e3 / 20
        rec.update({"patch_us": round(pus, 1), "patch_frac_mfma": round(flops / pus / 1e6 / 2500, 3),
                    "patch_max_abs_diff": float((yp.float() - y.float()).abs().max())})
    print(json.dumps(rec), flush=True)
