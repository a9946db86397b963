#!/usr/bin/env python3
"""List the kernels of the last training step of a rocprofv3 --kernel-trace CSV in launch order (steps are cut at the
fused-optimizer kernel, as tools/gap_analysis.py does).  Development aid for finding runs of tiny launches."""
import csv, re, sys

rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
marks = [i for i, r in enumerate(rows) if "FusedAdam" in r[2] or "adamw_clip_kernel" in r[2]]      # (torch fused AdamW, or csrc/adamw_flat.cuh)
ends = [m for j, m in enumerate(marks) if j + 1 == len(marks) or rows[marks[j + 1]][0] - rows[m][1] > 5_000_000]
seg = rows[ends[-2] + 1: ends[-1] + 1]


def short(k):
    k = re.sub(r"at::native::|\(anonymous namespace\)::|void |snipper::", "", k)
    k = re.sub(r"vectorized_elementwise_kernel<\d+, ", "vec<", k)
    k = re.sub(r"elementwise_kernel_manual_unroll<128, \d+, gpu_kernel_impl(_nocast)?<", "elem<", k)
    return k[:110]


t0 = seg[0][0]
for i, (s, e, k) in enumerate(seg):
    print(f"{i:5d} +{(s - t0) / 1e6:7.3f} {(e - s) / 1e3:8.1f}us {short(k)}")
