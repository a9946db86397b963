#!/bin/bash
# A/B runs of bench.py under different environment settings: tools/ab.sh "VAR=a" "VAR=b" ... (40 timed steps each, twice)
for rep in 1 2; do
  for kv in "$@"; do
    ms=$(env $kv python bench.py --no-cpu-baseline --no-extras --steps 40 --warmup 5 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "$kv  $ms ms/step"
  done
done
