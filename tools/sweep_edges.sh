for e in "16 8 4" "16 16 4" "16 16 8" "16 8 8" "16 16 16" "8 8 4"; do
  python tools/opbench.py --N 8 --cases enc_local --dtypes bfloat16 --skip-torch --sigma 0.01 3 --grid 1 --radius 24 --iters 8 --edges $e 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except ValueError: continue
    print('edges', d['edges'], 'sigma', d['sigma_px'], 'bwd', d['bwd_ms'], d['bwd_variant'])"
done
