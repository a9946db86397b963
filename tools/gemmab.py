#!/usr/bin/env python3
"""Timing ablations of linear_bf16 (SNIPPER_GEMM_DEBUG: 1 no epilogue, 2 no MFMA / LDS reads, 4 no in-loop global loads)."""
import json, os, subprocess, sys
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
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
    out = {}
    for M, K, N in [(79000, 384, 384), (79000, 384, 1024), (79000, 1024, 384), (240000, 64, 256), (240000, 256, 64), (60000, 512, 128)]:
        x = torch.randn(M, K, device="cuda").bfloat16(); w = torch.randn(N, K, device="cuda").bfloat16()
        b = torch.randn(N, device="cuda")
        out[f"{M}x{K}x{N}"] = round(t(lambda: linear_bf16(x, w, b, None, True)) * 1e3, 1)
    print(json.dumps({"dbg": int(os.environ.get("SNIPPER_GEMM_DEBUG", "0")), "us": out}), flush=True)
else:
    for d in (sys.argv[1:] or ["0", "1", "2", "4", "3", "6", "7"]):
        subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, SNIPPER_GEMM_DEBUG=d))
