#!/usr/bin/env python3
"""Mean per-dispatch value of one PMC counter per kernel from a rocprofv3 --pmc counter_collection CSV."""
import collections, csv, sys
path, counter = sys.argv[1], sys.argv[2]
pat = sys.argv[3:] or ["snipper::"]
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(path)):
    if r.get("Counter_Name") != counter:
        continue
    k = r["Kernel_Name"]
    if not any(p in k for p in pat):
        continue
    agg[k][0] += 1
    agg[k][1] += float(r["Counter_Value"])
for k, (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{k[:90]},{counter},{n},{v / n:.1f}")
