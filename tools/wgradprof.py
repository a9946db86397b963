#!/usr/bin/env python3
"""a few launches of the weight-gradient kernels per shape (for rocprofv3 --kernel-trace)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snipper_amd.dense import wgrad_bf16
dev = "cuda:0"
shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(79000, 384, 384)]
for M, N, Kc in shapes:
    g = torch.randn(M, N, device=dev).bfloat16(); x = torch.randn(M, Kc, device=dev).bfloat16()
    if os.environ.get("SNIPPER_PROF_ZERO") == "1":          # (is a difference to the ablations data-dependent, i.e. clock / power?)
        g.zero_(); x.zero_()
    for _ in range(20): wgrad_bf16(g, x)
    torch.cuda.synchronize()
