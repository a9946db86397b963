#!/usr/bin/env python3
"""Experiment: the static-shape ResNet-50 region of the step as three hipGraph-captured segments
(stem + layer1 + layer2 | layer3 | layer4 -- the boundaries of the gradient all-reduce stages, so that each stage's hook
still fires when its segment's backward has run), via torch.cuda.make_graphed_callables over this package's own kernels.
Prints host issue time and GPU time of forward + backward, eager against replayed, and the largest output / gradient
difference."""
import json
import os
import sys
import time
from types import SimpleNamespace

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("SNIPPER_EXPERIMENTAL_GRAPHS", "1")   # this tool IS the experiment
import bench                                            # noqa: E402
from snipper_amd.backbone import graphed_segments       # noqa: E402
from snipper_amd.model import build_model               # noqa: E402

dev = torch.device("cuda:0")
a = SimpleNamespace(hidden_dim=384, enc_layers=1, dec_layers=1, frames=4, future_frames=0, batch=2, height=600, width=800,
                    use_pytorch_deform=0)
torch.manual_seed(0)
model = build_model(bench.model_args(a)).to(dev).to(memory_format=torch.channels_last).train()
body = model.backbone[0].body
x = torch.rand(8, 3, 600, 800, device=dev)
gen = torch.Generator(device="cpu").manual_seed(1)


def run(fn, reps=12):
    """fn() issues one forward + backward; returns (issue ms, total ms), medians."""
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
    return round(issue[len(issue) // 2] * 1e3, 3), round(total[len(total) // 2] * 1e3, 3)


params = [p for p in body.parameters() if p.requires_grad]
gouts = None


def step(fwd):
    global gouts
    for p in params:
        p.grad = None
    with torch.autocast("cuda", dtype=torch.bfloat16):
        feats = fwd(x)
    feats = [feats[k] for k in sorted(feats)]
    if gouts is None:
        gouts = [torch.randn(f.shape, generator=gen).to(dev).to(f.dtype).contiguous(memory_format=torch.channels_last) * 1e-2 for f in feats]
    torch.autograd.backward(feats, gouts)
    return feats


from snipper_amd import model as _m   # the model refreshes the bf16 weight shadows at the top of its forward
from snipper_amd.shadow import WeightShadows
sh = WeightShadows(model)
sh.refresh()
eager = lambda: step(body)
MODE = os.environ.get("GB_MODE", "")            # "eager" / "graph": only that mode, 12 steps (for a kernel trace of each)
if MODE:
    fn = eager if MODE == "eager" else (lambda g=graphed_segments(body, x): step(g))
    for _ in range(12):
        fn()
    torch.cuda.synchronize()
    print(json.dumps({"mode": MODE, "timing": run(fn)}))
    sys.exit(0)
for _ in range(3):
    eager()
f_ref = [f.detach().float().clone() for f in eager()]
g_ref = [p.grad.detach().float().clone() for p in params]
t_eager = run(eager)

graphed = graphed_segments(body, x)
gfwd = lambda: step(graphed)
for _ in range(3):
    gfwd()
f_new = [f.detach().float().clone() for f in gfwd()]
g_new = [p.grad.detach().float().clone() for p in params]
t_graph = run(gfwd)
ferr = max(float((u - v).abs().max()) for u, v in zip(f_new, f_ref))
gerr = max(float((u - v).abs().max() / v.abs().max().clamp_min(1e-20)) for u, v in zip(g_new, g_ref))
print(json.dumps({"eager_issue_ms": t_eager[0], "eager_total_ms": t_eager[1], "graph_issue_ms": t_graph[0],
                  "graph_total_ms": t_graph[1], "max_abs_output_diff": ferr, "max_rel_grad_diff": gerr}))
