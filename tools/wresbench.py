#!/usr/bin/env python3
"""Weight-stationary GEMM (csrc/wres_gemm_bf16.cuh): parity against float64 and timing against the tile kernels.
`SNIPPER_GEMM_WRES=0 python tools/wresbench.py` times the tile kernels behind the same entry points."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snipper_amd.dense import linear_bf16, linear_wres_bf16, linear_nn_bf16, transpose_batch_bf16

def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

dev = "cuda:0"
torch.manual_seed(0)
wres_on = os.environ.get("SNIPPER_GEMM_WRES", "1") != "0"
# ---- parity (float64 of the same bf16 operands)
if wres_on:
    for (M, K, N, relu, gate) in [(8192, 384, 384, False, False), (9001, 384, 288, True, False), (8200, 288, 384, False, False),
                                  (8192, 384, 1024, True, False), (8192 + 17, 384, 1024, False, True), (40000, 384, 96, False, False),
                                  (8195, 288, 384, False, True)]:
        x = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16()
        b = torch.randn(N, device=dev)
        a = torch.randn(M, N, device=dev).relu().bfloat16() if gate else None
        y = linear_wres_bf16(x, w, b, relu=relu, gate=a, gate_scale=1.25 if gate else 1.0)
        ref = x.double() @ w.double().t() + b.double()
        if relu: ref = ref.relu()
        if gate: ref = torch.where(a.double() > 0, ref * 1.25, torch.zeros_like(ref))
        err = (y.double() - ref).abs().max().item()
        tol = (ref.abs().max().item()) * 2 ** -7
        print(json.dumps({"parity": [M, K, N, relu, gate], "max_err": err, "tol": tol, "ok": err <= tol}), flush=True)
    # transposes
    srcs = [torch.randn(r, c, device=dev).bfloat16() for r, c in [(384, 384), (288, 384), (384, 1024), (70, 130)]]
    dsts = [torch.empty(s.shape[1], s.shape[0], device=dev, dtype=torch.bfloat16) for s in srcs]
    transpose_batch_bf16(list(zip(srcs, dsts)))
    print(json.dumps({"transpose_ok": all(torch.equal(d, s.t()) for s, d in zip(srcs, dsts))}), flush=True)
# ---- timing
for (M, K, N) in [(79000, 384, 384), (79000, 384, 288), (79000, 384, 1024), (79000, 288, 384), (158000, 384, 384), (39500, 384, 384)]:
    x = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16()
    b = torch.randn(N, device=dev)
    wt = w.t().contiguous()
    a = torch.randn(M, N, device=dev).relu().bfloat16()
    r = {"M": M, "K": K, "N": N, "wres": wres_on}
    r["nt_bias_us"] = round(t(lambda: linear_bf16(x, w, b)), 1)
    r["nt_bias_relu_drop_us"] = round(t(lambda: linear_bf16(x, w, b, None, True, 0.1, 1234)), 1)
    if K % 64 == 0:
        r["nn_tile_us"] = round(t(lambda: linear_nn_bf16(x, wt)), 1)
        r["nn_tile_gate_us"] = round(t(lambda: linear_nn_bf16(x, wt, None, a, 1.1)), 1)
    if wres_on:
        r["wres_gate_us"] = round(t(lambda: linear_wres_bf16(x, w, None, gate=a, gate_scale=1.1)), 1)
    byt = 2 * (M * K + N * K + M * N)
    r["nt_GBps"] = round(byt / r["nt_bias_us"] / 1e3, 1)
    print(json.dumps(r), flush=True)
