#!/usr/bin/env python3
"""Decoder-shaped core-op backward on a bf16 value through snipper_msda_backward_sparse_bf16 (memset + per-(sample, head, level)
sort kernel + atomic-free query kernel): time per call for uniformly random sampling points and for the decoder's situation at
initialisation (all queries of a head around the same few points).  (The per-phase figures in
profiles/r05_sparse_backward_bench.jsonl came from a diagnostic build that ended the kernel after its key phase / after the sort.)"""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snipper_amd import MultiScaleDeformableAttention as MSDA, _lib
dev = "cuda:0"
shapes = [(75, 100), (38, 50), (19, 25)]
S = sum(h * w for h, w in shapes)
N, M, D, L, P, Lq = 8, 8, 48, 3, 4, 60
g = torch.Generator().manual_seed(0)
sh = torch.tensor(shapes, device=dev)
lsi = torch.cat((sh.new_zeros(1), sh.prod(1).cumsum(0)[:-1]))
attn = torch.softmax(torch.randn(N, Lq, M, L * P, generator=g), -1).view(N, Lq, M, L, P).to(dev)
v = torch.randn(N, S, M, D, generator=g).to(dev).to(torch.bfloat16)
go = torch.randn(N, Lq, M * D, generator=g).to(dev).to(torch.bfloat16)
def t(fn, n=100):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
out = {}
for name, loc in (("uniform", torch.rand(N, Lq, M, L, P, 2, generator=g)),
                  ("clustered", (0.5 + 0.02 * torch.randn(N, 1, M, L, P, 2, generator=g) + 0.002 * torch.randn(N, Lq, M, L, P, 2, generator=g)).clamp(0, 1))):
    loc = loc.to(dev)
    out[name + "_us"] = round(t(lambda: MSDA.ms_deform_attn_backward(v, sh, lsi, loc, attn, go, 64)), 1)
    out["variant"] = _lib.last_variant()
print(json.dumps(out))
