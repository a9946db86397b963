#!/bin/bash
# Runs ON the GPU box: kernel durations (rocprofv3 kernel trace) of the weight-stationary GEMM with phases compiled out
# at run time (SNIPPER_WRES_DEBUG: 1 no W loads, 2 no MFMA, 4 no stores, 8 no X reads -- wrong results, timing only).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
SHAPES=${SHAPES:-"79000x384x384 8192x384x384 158000x384x384"}
for dbg in ${DBGS:-0 1 2 4 8 15 6}; do
  out=$R/gpurun_out/wres_dbg$dbg
  SNIPPER_MSDA_ALLOW_DEBUG=1 SNIPPER_WRES_DEBUG=$dbg rocprofv3 --kernel-trace --output-format csv -d $out -- python3 $R/tools/wresprof.py $SHAPES > $out.log 2>&1
  python3 - "$out" "$dbg" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "wres" in r["Kernel_Name"] or "linear_bf16" in r["Kernel_Name"]]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
meds = []
for i in range(0, len(d), 20):
    seg = sorted(d[i:i + 20])
    meds.append(round(seg[len(seg) // 2], 1))
print(f"debug={sys.argv[2]:>2s} median kernel us per shape: {meds}")
PY
  rm -rf $out
done
