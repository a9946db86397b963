#!/bin/bash
# Runs ON the GPU box: VALU / LDS / SALU instruction counts of the tile kernel with phases compiled out at run time
# (opbench --debug: 1 no row staging, 2 no accumulate, 4 no decode, 8 skeleton; wrong results, counting only).
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-t2ph}
mkdir -p $out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export SNIPPER_MSDA_ALLOW_DEBUG=1
for dbg in ${DBGS:-0 2 4 6 8}; do
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $out/p_$dbg -- python3 $R/tools/opbench.py --N 8 --cases enc_local --dtypes bfloat16 --skip-torch --sigma 0.01 --grid 1 --iters 3 --debug $dbg > $out/p_$dbg.log 2>&1
  f=$(ls $out/p_$dbg/*/*counter_collection.csv 2>/dev/null | head -1)
  [ -z "$f" ] && { echo "debug $dbg: no counters"; tail -3 $out/p_$dbg.log; continue; }
  echo "debug=$dbg"
  for c in SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES; do
    (cd $R && python3 tools/pmc_summary.py $f $c msda_bwd_d48_tile2 | cut -d, -f2-)
  done
  rm -rf $out/p_$dbg
done
