#!/usr/bin/env python3
"""Steady-state summary of a rocprofv3 --kernel-trace CSV: only dispatches in the last `--frac` of the
timeline are counted (the first steps carry MIOpen's find-mode benchmarking kernels)."""
import argparse, collections, csv, re, sys

ap = argparse.ArgumentParser()
ap.add_argument("trace")
ap.add_argument("--frac", type=float, default=0.4)
ap.add_argument("--top", type=int, default=40)
ap.add_argument("--csv", default=None)
ap.add_argument("--by", default="time", choices=["time", "count"])
a = ap.parse_args()
rows = list(csv.DictReader(open(a.trace)))
st = [int(r["Start_Timestamp"]) for r in rows]
en = [int(r["End_Timestamp"]) for r in rows]
t0, t1 = min(st), max(en)
cut = t1 - (t1 - t0) * a.frac
agg = collections.defaultdict(lambda: [0, 0])
busy = 0
for r, s, e in zip(rows, st, en):
    if s >= cut:
        k = r["Kernel_Name"]
        agg[k][0] += 1
        agg[k][1] += e - s
        busy += e - s
win = t1 - cut
print(f"window {win/1e6:.1f} ms, kernels busy {busy/1e6:.1f} ms ({100*busy/win:.1f}%), dispatches {sum(v[0] for v in agg.values())}")
items = sorted(agg.items(), key=lambda kv: -kv[1][1 if a.by == "time" else 0])
for k, (n, t) in items[:a.top]:
    short = re.sub(r"at::native::|\(anonymous namespace\)::|void ", "", k)[:110]
    print(f"{100*t/win:5.1f}% {n:6d} {t/n/1e3:9.1f}us  {short}")
if a.csv:
    w = csv.writer(open(a.csv, "w"))
    w.writerow(["kernel", "dispatches_in_window", "total_ns", "avg_ns", "pct_of_window", f"window_ns={win}"])
    for k, (n, t) in items[:100]:
        w.writerow([k, n, t, round(t / n, 1), round(100 * t / win, 3)])
