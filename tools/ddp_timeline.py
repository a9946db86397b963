#!/usr/bin/env python3
"""From a rocprofv3 kernel trace of `SNIPPER_FORCE_DDP=1 SNIPPER_SYNC_FORCE=1 bench.py` (one GPU, a 1-rank RCCL group):
where do the staged gradient all-reduces sit relative to the backward kernels?  Prints, for one steady-state step, every
RCCL kernel with its start / end relative to the step, the compute kernels that RUN WHILE it runs (other stream) and the
last backward kernel of the step -- the all-reduces of the stages launched from autograd hooks must start before it.

    python tools/ddp_timeline.py <kernel_trace.csv> [step_index]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
name = lambda r: r["Kernel_Name"]
short = lambda n: (n.split("snipper::")[1].split("(")[0] if "snipper::" in n else n.split("(")[0])[:70]
# a step starts at the stem convolution
starts = [i for i, r in enumerate(rows) if "stem7x7" in name(r)]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(starts) - 3
a, b = starts[k], starts[k + 1]
step = rows[a:b]
t0 = step[0]["s"]
us = lambda t: (t - t0) / 1e3
print(f"step {k}: {len(step)} kernels, {us(step[-1]['e']):.1f} us from the stem convolution to the last kernel")
nccl = [r for r in step if "nccl" in name(r).lower() or "rccl" in name(r).lower()]
bwd = [r for r in step if any(t in name(r) for t in ("wgrad_bf16_kernel", "linear_bf16_nn_kernel", "conv3x3"))]
last_bwd = max(bwd, key=lambda r: r["e"])
first_wgrad = min((r for r in step if "wgrad_bf16_kernel" in name(r)), key=lambda r: r["s"])
print(f"backward compute: first weight-gradient kernel at {us(first_wgrad['s']):.1f} us, last backward GEMM ends at {us(last_bwd['e']):.1f} us")
print(f"RCCL kernels in the step: {len(nccl)}")
for i, r in enumerate(nccl):
    over = [x for x in step if x is not r and x["s"] < r["e"] and x["e"] > r["s"] and "nccl" not in name(x).lower()]
    prev = [x for x in step if x["e"] <= r["s"] and "nccl" not in name(x).lower()]
    nxt = [x for x in step if x["s"] >= r["e"] and "nccl" not in name(x).lower()]
    where = "INSIDE backward" if r["s"] < last_bwd["e"] else "after backward"
    print(f"  [{i}] {short(name(r))}: {us(r['s']):9.1f} .. {us(r['e']):9.1f} us ({(r['e'] - r['s']) / 1e3:6.1f} us)  {where}")
    print(f"       overlapping compute kernels: {len(over)}" + (f", e.g. {short(name(over[0]))} ... {short(name(over[-1]))}" if over else ""))
    if prev:
        print(f"       previous compute kernel: {short(name(prev[-1]))} (ended {us(prev[-1]['e']):.1f} us)")
    if nxt:
        print(f"       next compute kernel:     {short(name(nxt[0]))} (started {us(nxt[0]['s']):.1f} us)")
