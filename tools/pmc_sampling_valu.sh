#!/bin/bash
# Runs ON the GPU box: instruction mix / issue counters of the sampling kernels on the encoder shape (opbench, N = 8,
# bf16 value, reference offset grid).  Separate PMC passes, kernel trace only.   usage: tools/pmc_sampling_valu.sh <name>
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-valu}
mkdir -p $out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
pass() {
  name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $out/p_$name -- python3 $R/tools/opbench.py --N 8 --cases enc_local --dtypes bfloat16 --skip-torch --sigma 0.01 --grid 1 --iters 4 > $out/p_$name.log 2>&1
}
pass a SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES
pass b SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU
pass c GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_FLAT SQ_VALU_MFMA_BUSY_CYCLES SQ_IFETCH SQ_INSTS_VALU_CVT
cd $R
{
  echo "kernel,counter,dispatches,mean_value"
  for p in a b c; do
    f=$(ls $out/p_$p/*/*counter_collection.csv 2>/dev/null | head -1)
    [ -z "$f" ] && { echo "pass $p: no counter file"; tail -3 $out/p_$p.log; continue; }
    for c in $(python3 -c "
import csv,sys
print(' '.join(sorted({r['Counter_Name'] for r in csv.DictReader(open('$f'))})))"); do
      python3 tools/pmc_summary.py $f $c msda_bwd_d48_tile2 msda_bwd_d48_patchbin msda_fwd_d48
    done
  done
} > $out/pmc_sampling_valu.csv
rm -rf $out/p_a $out/p_b $out/p_c
cat $out/pmc_sampling_valu.csv
