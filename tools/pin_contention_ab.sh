#!/bin/bash
# Runs ON the GPU box: the default bench step with eight busy-loop processes sitting on the FIRST eight CPUs of the GPU's NUMA
# node (what a neighbour that pins the same way would do), with the fixed block (SNIPPER_PIN_IDLE=0) and with the load-aware
# choice (bench.pick_cpus rule "numa+idle").  Prints ms per step and the CPUs each run was pinned to.
cd $GRAFT_REPO_ROOT
CPUS=$(python3 - <<'PY'
import bench
node = bench.gpu_numa_node(0)
cpus = bench._parse_cpulist(open(f"/sys/devices/system/node/node{node}/cpulist").read()) if node is not None else list(range(8))
print(" ".join(str(c) for c in cpus[:8]))
PY
)
echo "burners on CPUs: $CPUS"
run() { python3 bench.py --steps 40 --warmup 10 --no-extras --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); h=d['config']['host']; print('$1', d['ms_per_step'], h[h.find('process pinned'):])"; }
SNIPPER_PIN_IDLE=0 run "quiet, fixed block   "
PIDS=""
for c in $CPUS; do taskset -c $c python3 -c "
while True: pass" & PIDS="$PIDS $!"; done
sleep 1
SNIPPER_PIN_IDLE=0 run "burners, fixed block "
SNIPPER_PIN_IDLE=1 run "burners, numa+idle   "
kill $PIDS
wait 2>/dev/null
