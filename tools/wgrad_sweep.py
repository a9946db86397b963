#!/usr/bin/env python3
"""Weight-gradient entry at the ResNet body's 1x1 shapes (M = pixels of 8 frames) under the environment's dispatch knobs
(SNIPPER_WGRAD_RING_TILES, SNIPPER_WGRAD_S, SNIPPER_WGRAD_WIDE ...): us per call incl. the second pass."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snipper_amd.dense import wgrad_bf16
dev = 'cuda:0'
def t(fn, n=40):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
out = {}
for (M, N, Kc) in [(15200, 1024, 256), (15200, 256, 1024), (15200, 1024, 512), (60000, 512, 128), (60000, 128, 512), (60000, 512, 256),
                   (3800, 2048, 512), (3800, 512, 2048), (3800, 2048, 1024), (240000, 256, 64), (240000, 64, 256)]:
    g = torch.randn(M, N, device=dev).bfloat16(); x = torch.randn(M, Kc, device=dev).bfloat16()
    out[f"{M}x{N}x{Kc}"] = round(t(lambda: wgrad_bf16(g, x, bias=False) if False else wgrad_bf16(g, x)), 1)
print(json.dumps({"env": {k: v for k, v in os.environ.items() if k.startswith("SNIPPER_WGRAD")}, "us": out}), flush=True)
