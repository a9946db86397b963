#!/usr/bin/env python3
"""Weight-gradient kernels (split reduction): time per call incl. the partial reduction and the bias gradient.
`SNIPPER_WGRAD_RING=0 python tools/wgradbench.py` times the register-prefetch kernel behind the same entry point."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snipper_amd.dense import wgrad_bf16
dev = 'cuda:0'
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
ring = os.environ.get("SNIPPER_WGRAD_RING", "1") != "0"
wide = os.environ.get("SNIPPER_WGRAD_WIDE", "1") != "0"
for (M, N, Kc) in [(79000, 384, 384), (79000, 1024, 384), (79000, 384, 1024), (79000, 288, 384), (79000, 192, 384), (158000, 384, 384),
                   (60000, 512, 256), (60000, 384, 512), (240000, 64, 256), (15200, 1024, 512)]:
    g = torch.randn(M, N, device=dev).bfloat16(); x = torch.randn(M, Kc, device=dev).bfloat16()
    us = t(lambda: wgrad_bf16(g, x))
    dW, db = wgrad_bf16(g, x)
    ref = g.double().t() @ x.double()
    err = float((dW.double() - ref).abs().max() / ref.abs().max())
    errb = float((db.double() - g.double().sum(0)).abs().max() / g.double().sum(0).abs().max())
    print(json.dumps({"M": M, "N": N, "Kc": Kc, "ring": ring, "wide": wide, "us_incl_reduce_and_bias": round(us, 1), "rel_err_dW": err, "rel_err_db": errb,
                      "GBps": round(2 * M * (N + Kc) / us / 1e3, 1)}), flush=True)
