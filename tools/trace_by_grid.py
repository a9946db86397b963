#!/usr/bin/env python3
"""rocprofv3 kernel_trace.csv -> average duration per (kernel, grid size): tells apart the shapes one kernel was launched on."""
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"]
    short = name.split("snipper::")[1].split("(")[0] if "snipper::" in name else name[:50]
    acc[(short, r.get("Grid_Size_X", r.get("Grid_Size", "?")))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for (k, gsz), v in sorted(acc.items()):
    if pat in k:
        v = sorted(v)[: max(1, len(v) * 3 // 4)]          # drop the slowest quarter (first calls)
        print(f"{k:48s} grid {gsz:>8s} n {len(v):4d} avg_us {sum(v) / len(v) / 1e3:8.2f}")
