#!/bin/bash
# Runs ON the GPU box: kernel traces of the ResNet-50 forward + backward eagerly and as three replayed hipGraph segments
# (tools/graph_backbone.py, GB_MODE) -- launches, summed kernel time and the kernels that only one of the two modes runs.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for m in eager graph; do
  GB_MODE=$m rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/gb_$m -- python3 $R/tools/graph_backbone.py > $R/gpurun_out/gb_$m.log 2>&1
  tail -1 $R/gpurun_out/gb_$m.log
done
python3 - "$R" <<'PY'
import csv, glob, sys
R = sys.argv[1]
tot = {}
for m in ("eager", "graph"):
    f = glob.glob(f"{R}/gpurun_out/gb_{m}/**/*kernel_stats.csv", recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    tot[m] = {r["Name"]: (int(r["Calls"]), float(r["TotalDurationNs"])) for r in rows}
    print(m, "launches", sum(c for c, _ in tot[m].values()), "kernel ms", round(sum(t for _, t in tot[m].values()) / 1e6, 2))
names = set(tot["eager"]) | set(tot["graph"])
diff = sorted(((tot["graph"].get(n, (0, 0))[1] - tot["eager"].get(n, (0, 0))[1], n) for n in names), reverse=True)
for d, n in diff[:12]:
    print(f"{d / 1e6:8.3f} ms more in graph mode  calls {tot['eager'].get(n, (0, 0))[0]:5d} -> {tot['graph'].get(n, (0, 0))[0]:5d}  {n[:110]}")
for d, n in diff[-5:]:
    print(f"{d / 1e6:8.3f} ms  calls {tot['eager'].get(n, (0, 0))[0]:5d} -> {tot['graph'].get(n, (0, 0))[0]:5d}  {n[:110]}")
PY
rm -rf $R/gpurun_out/gb_eager $R/gpurun_out/gb_graph
