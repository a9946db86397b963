#!/usr/bin/env python3
"""bf16 linear: HIP MFMA kernel vs torch (hipBLASLt) on the path's shapes."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snipper_amd.dense import linear_bf16

def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    e[0].record()
    for i in range(n):
        fn(); e[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(e[i].elapsed_time(e[i + 1]) for i in range(n))
    return ts[n // 2]

shapes = [(79000, 384, 384), (79000, 384, 192), (79000, 384, 96), (79000, 384, 1024), (79000, 1024, 384),
          (240000, 64, 256), (240000, 256, 64), (240000, 64, 64), (60000, 512, 128), (60000, 128, 512), (15200, 1024, 256),
          (15200, 256, 1024), (3800, 2048, 512), (3800, 512, 2048)]
for M, K, N in shapes:
    x = torch.randn(M, K, device="cuda").bfloat16(); w = torch.randn(N, K, device="cuda").bfloat16()
    b = torch.randn(N, device="cuda"); bb = b.bfloat16()
    a = t(lambda: linear_bf16(x, w, b, None, True))
    c = t(lambda: torch.relu_(torch.nn.functional.linear(x, w, bb)))
    d = t(lambda: torch.nn.functional.linear(x, w, bb))
    byt = 2 * (M * K + N * K + M * N)
    print(json.dumps({"M": M, "K": K, "N": N, "hip_bias_relu_ms": round(a, 4), "torch_linear_relu_ms": round(c, 4),
                      "torch_linear_ms": round(d, 4), "hip_TFLOPs": round(2 * M * N * K / a / 1e9, 1),
                      "hip_GBps": round(byt / a / 1e6, 1)}), flush=True)
