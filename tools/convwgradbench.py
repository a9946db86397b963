#!/usr/bin/env python3
"""3x3-convolution weight gradients at the ResNet-50 body's shapes (8 frames of 600 x 800): kernel + second pass by events."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snipper_amd.dense import wgrad_conv3x3_bf16

dev = "cuda:0"
torch.manual_seed(0)
for cin, cout, h, w, st in [(128, 128, 75, 100, 1), (256, 256, 38, 50, 1), (512, 512, 19, 25, 1), (128, 128, 150, 200, 2),
                            (256, 256, 75, 100, 2), (512, 512, 38, 50, 2)]:
    x = torch.randn(8, cin, h, w, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    ho, wo = (h - 1) // st + 1, (w - 1) // st + 1
    g = torch.randn(8, cout, ho, wo, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    sc = torch.rand(cout, device=dev) + 0.5
    run = lambda: wgrad_conv3x3_bf16(g, x, st, sc)
    d = run()
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    flops = 2.0 * 8 * ho * wo * cin * cout * 9
    print(json.dumps({"case": f"{cin}->{cout} {h}x{w} s{st}", "us": round(us, 1), "frac_mfma": round(flops / us / 1e6 / 2500, 3),
                      "checksum": float(d.float().abs().sum())}), flush=True)
