#!/bin/bash
# Runs ON the GPU box: encoder-shape backward (bf16 value + rows, N = 8, reference offset bias grid + N(0, sigma) px) against
# the near radius of the owner-computes split.   usage: tools/sweep_radius.sh "6 8 12 16" 
for r in ${1:-6 8 10 12}; do
  python tools/opbench.py --N 8 --cases enc_local --dtypes bfloat16 --skip-torch --sigma 0.01 3 8 --grid 1 --radius $r --iters 8 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except ValueError:
        print(l.rstrip()[-200:]); continue
    print('R', d['radius'], 'sigma', d['sigma_px'], 'fwd', d['fwd_ms'], 'bwd', d['bwd_ms'], d['bwd_variant'])"
done
