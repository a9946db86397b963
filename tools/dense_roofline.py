#!/usr/bin/env python3
"""Per-shape roofline table of every GEMM-shaped launch of the training step.

    SNIPPER_DENSE_TABLE=/tmp/dense.json python bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-locality-sweep
    python tools/dense_roofline.py /tmp/dense.json > profiles/r04_backbone_roofline.csv

bench.py brackets every linear / weight-stationary / NN data-gradient / weight-gradient / 3x3 / stem launch of two eager
steps with events on its stream and records the product's FLOPs (as defined, no padding) and COMPULSORY HBM bytes
(operands read once, result written once).  A shape is HBM-bound when its intensity is below peak_flops / peak_bytes =
2.5e15 / 8e12 = 312 FLOP/B, else MFMA-bound; `frac` is the achieved fraction of that bound's peak."""
import collections
import json
import sys

PEAK_F, PEAK_B = 2.5e15, 8.0e12
rows = json.load(open(sys.argv[1]))
agg = collections.OrderedDict()
for kind, shape, flops, nbytes, ms in rows:
    e = agg.setdefault((kind, tuple(shape)), [0, flops, nbytes, 0.0])
    e[0] += 1
    e[3] += ms


def where(kind, shape):
    if kind.startswith("conv3x3") or kind == "stem7x7":
        b, h, w, cin, cout, st = shape
        return f"backbone {h}x{w} {cin}->{cout} s{st}"
    m, n, k = shape
    if m in (79000,):
        return "encoder / projections (79 000 token rows)"
    if m >= 3000:
        return f"backbone 1x1 / input projection ({m} pixels)"
    return "other"


print("kernel,shape,where,launches_per_step,us_per_launch,ms_per_step,gflop,compulsory_MB,flop_per_byte,bound,achieved,unit,frac_of_peak")
tot_ms = tot_f = 0.0
for (kind, shape), (n, flops, nbytes, ms) in sorted(agg.items(), key=lambda kv: -kv[1][3]):
    t = ms / n * 1e-3
    inten = flops / nbytes
    if inten < PEAK_F / PEAK_B:
        bound, ach, unit, frac = "hbm", nbytes / t / 1e9, "GB/s", nbytes / t / PEAK_B
    else:
        bound, ach, unit, frac = "mfma", flops / t / 1e12, "TFLOP/s", flops / t / PEAK_F
    tot_ms += ms / 2
    tot_f += flops * n / 2
    print(f"{kind},{'x'.join(map(str, shape))},{where(kind, shape)},{n // 2},{ms / n * 1e3:.1f},{ms / 2:.3f},{flops / 1e9:.2f},"
          f"{nbytes / 1e6:.1f},{inten:.0f},{bound},{ach:.1f},{unit},{frac:.3f}")
print(f"# total: {tot_ms:.2f} ms per step, {tot_f / 1e12:.2f} TFLOP per step = {tot_f / (tot_ms * 1e-3) / 1e12:.0f} TFLOP/s = "
      f"{tot_f / (tot_ms * 1e-3) / PEAK_F:.3f} of the dense bf16 MFMA peak")
