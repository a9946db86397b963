#!/bin/bash
# Runs ON the GPU box: vector-L1 / texture-addresser counters of the sampling kernels on the encoder shape (opbench, N = 8,
# bf16 value): is the gather bound by L1 hits, by L2 -> L1 fills or by the addresser?   usage: tools/pmc_gather_l1.sh <name>
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-l1}
mkdir -p $out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
pass() {
  name=$1; shift
  timeout 150 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $out/p_$name -- python3 $R/tools/opbench.py --N 8 --cases enc_local --dtypes bfloat16 --skip-torch --sigma ${SIGMA:-0.01} --grid 1 --iters 4 --value-layout ${VL:-0} > $out/p_$name.log 2>&1
}
pass a TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_READ_sum TCP_PENDING_STALL_CYCLES_sum
# (a pass with TA_* counters -- TA_TA_BUSY_sum, TA_BUFFER_READ_WAVEFRONTS_sum, ... -- aborted inside rocprofv3 on this image and
#  then sat in its finaliser until the box's time limit: every pass now runs under `timeout`, and the TA block is left out)
pass c TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum
cd $R
{
  echo "kernel,counter,dispatches,mean_value"
  for p in a c; do
    f=$(ls $out/p_$p/*/*counter_collection.csv 2>/dev/null | head -1)
    [ -z "$f" ] && { echo "pass $p: no counter file"; tail -3 $out/p_$p.log; continue; }
    for c in $(python3 -c "
import csv
print(' '.join(sorted({r['Counter_Name'] for r in csv.DictReader(open('$f'))})))"); do
      python3 tools/pmc_summary.py $f $c msda_bwd_d48_tile3 msda_bwd_d48_patchbin msda_fwd_d48
    done
  done
} > $out/pmc_gather_l1.csv
rm -rf $out/p_a $out/p_c
cat $out/pmc_gather_l1.csv
