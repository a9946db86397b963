#!/usr/bin/env python3
"""Decoder-shaped core op (Lq = 60 queries, N = 8, 600x800 levels): forward + backward time for a float32 and a bfloat16
`value` (the latter: tuned forward, generic backward) -- is a bf16 value worth it for the decoder's cross attention?"""
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
loc = torch.rand(N, Lq, M, L, P, 2, generator=g).to(dev)
attn = torch.softmax(torch.randn(N, Lq, M, L * P, generator=g), -1).view(N, Lq, M, L, P).to(dev)
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for dt in (torch.float32, torch.bfloat16):
    v = torch.randn(N, S, M, D, generator=g).to(dev).to(dt)
    go = torch.randn(N, Lq, M * D, generator=g).to(dev).to(dt)
    f = t(lambda: MSDA.ms_deform_attn_forward(v, sh, lsi, loc, attn, 64, host_shapes=shapes))
    fv = _lib.last_variant()
    b = t(lambda: MSDA.ms_deform_attn_backward(v, sh, lsi, loc, attn, go, 64, host_shapes=shapes, grad_value_f32=True))
    bv = _lib.last_variant()
    print(json.dumps({"value": str(dt), "fwd_us": round(f, 1), "fwd_variant": fv, "bwd_us_incl_memset": round(b, 1), "bwd_variant": bv}))
