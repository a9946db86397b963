#!/usr/bin/env python3
"""a few launches of the weight-stationary kernel per shape (for rocprofv3 --kernel-trace --stats)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snipper_amd.dense import linear_bf16
dev = "cuda:0"
shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(79000, 384, 384)]
for M, K, N in shapes:
    x = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16(); b = torch.randn(N, device=dev)
    for _ in range(20): linear_bf16(x, w, b)
    torch.cuda.synchronize()
