#!/usr/bin/env python3
"""What does hipGraph replay buy on this stack (ROCm 7.2, PyTorch 2.10)?  A chain of N short kernels (the step has ~1 000 per
iteration, ~700 of them in the static-shape backbone + encoder region) issued eagerly and replayed from a captured graph:
host time to ISSUE the chain (until the call returns) and time until the GPU has retired it.  VERDICT r03 #2."""
import json
import sys
import time

import torch

dev = "cuda:0"
res = []
for n_nodes, numel in ((700, 1 << 16), (700, 1 << 22), (2000, 1 << 16)):
    xs = [torch.zeros(numel, device=dev) for _ in range(8)]

    def chain():
        for i in range(n_nodes):
            xs[i & 7].add_(1.0)

    for _ in range(3):
        chain()
    torch.cuda.synchronize()

    def measure(fn, reps=20):
        issue, total = [], []
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            issue.append(t1 - t0); total.append(t2 - t0)
        issue.sort(); total.sort()
        return issue[len(issue) // 2] * 1e3, total[len(total) // 2] * 1e3

    e_issue, e_total = measure(chain)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        chain()
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        chain()
    g.replay(); torch.cuda.synchronize()
    r_issue, r_total = measure(g.replay)
    res.append({"nodes": n_nodes, "elements_per_kernel": numel, "eager_issue_ms": round(e_issue, 3), "eager_retire_ms": round(e_total, 3),
                "replay_issue_ms": round(r_issue, 3), "replay_retire_ms": round(r_total, 3),
                "eager_us_per_kernel": round(e_total / n_nodes * 1e3, 2), "replay_us_per_kernel": round(r_total / n_nodes * 1e3, 2)})
    print(json.dumps(res[-1]), flush=True)
