#!/bin/bash
# Runs ON the GPU box: kernel-level times (rocprofv3 kernel trace) of the 3x3 weight-gradient kernels per shape, with ablations.
tag=${1:-r06n}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cd $R
timeout 600 python3 -m pytest tests/test_dense_gpu.py -x -q -k "conv3x3_weight_gradient" > $out/tests.txt 2>&1; echo "pytest rc=$?" >> $out/tests.txt
tail -3 $out/tests.txt
cd /tmp && export TMPDIR=/tmp
export SNIPPER_MSDA_ALLOW_DEBUG=1
run() {   # name, env assignments...
  name=$1; shift
  for kv in "$@"; do export "$kv"; done
  rocprofv3 --kernel-trace --output-format csv -d $out/$name -- python3 $R/tools/convwgradbench.py > $out/$name.log 2>&1
  for kv in "$@"; do unset "${kv%%=*}"; done
  t=$(ls $out/$name/*/*kernel_trace.csv | head -1)
  echo "== $name"; python3 $R/tools/trace_by_grid.py $t wgrad_conv
  rm -rf $out/$name
}
run patch
run nocompute SNIPPER_WRES_DEBUG=2

run noldsstore SNIPPER_WRES_DEBUG=8

run only_compute SNIPPER_WRES_DEBUG=28

