#!/bin/bash
# Runs ON the GPU box: kernel durations of the weight-gradient kernels (rocprofv3 kernel trace, median of 20 launches),
# the ring kernel with phases switched off at run time (SNIPPER_WRES_DEBUG: 1 no memory reads, 2 no MFMA, 8 no DMA
# instructions, 16 no partial stores, 32 exit after the prologue -- wrong results, timing only).
# SNIPPER_WGRAD_RING=0 times the register-prefetch kernel, SNIPPER_WGRAD_WIDE=0 the 128 x 128-tile kernels instead of the 128 x 384 one
# (its switches: 1 no memory reads, 2 no MFMA, 4 no fragment reads, 16 no partial stores); SNIPPER_WGRAD_WIDE_WGS=<grid size>.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
SHAPES=${SHAPES:-"79000x384x384 79000x1024x384"}
for dbg in ${DBGS:-0 1 2 8 10}; do
  out=$R/gpurun_out/wgrad_dbg$dbg
  SNIPPER_MSDA_ALLOW_DEBUG=1 SNIPPER_WRES_DEBUG=$dbg rocprofv3 --kernel-trace --output-format csv -d $out -- python3 $R/tools/wgradprof.py $SHAPES > $out.log 2>&1
  python3 - "$out" "$dbg" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
for tag in ("wgrad_wide", "wgrad_ring", "wgrad_bf16_kernel", "wgrad_reduce"):
    rows = [r for r in csv.DictReader(open(f)) if tag in r["Kernel_Name"]]
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
    meds = []
    for i in range(0, len(d), 20):
        seg = sorted(d[i:i + 20])
        meds.append(round(seg[len(seg) // 2], 1))
    if meds:
        print(f"debug={sys.argv[2]:>2s} {tag:18s} median kernel us per shape: {meds}")
PY
  rm -rf $out
done
