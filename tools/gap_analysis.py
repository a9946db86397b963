#!/usr/bin/env python3
"""Where does a training step's wall time go on the GPU timeline?  Reads a rocprofv3 --kernel-trace CSV, cuts it
into steps at the fused-optimizer kernel, and for the last few steps reports busy time, the idle gaps between
consecutive dispatches (histogram), and the kernels that precede the longest gaps (= where the host is late)."""
import argparse, collections, csv, re

ap = argparse.ArgumentParser()
ap.add_argument("trace")
ap.add_argument("--delim", default="FusedAdam")
ap.add_argument("--steps", type=int, default=4)
ap.add_argument("--top", type=int, default=25)
ap.add_argument("--csv", default=None, help="write the last step's per-kernel totals here")
ap.add_argument("--region", type=float, nargs=2, default=None, help="ms range of the last step to list kernels for")
a = ap.parse_args()
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(a.trace))]
rows.sort()
marks = [i for i, r in enumerate(rows) if a.delim in r[2]]
# one optimizer step = a burst of delimiter kernels; keep the last index of each burst
ends = [m for j, m in enumerate(marks) if j + 1 == len(marks) or rows[marks[j + 1]][0] - rows[m][1] > 5_000_000]
print(f"{len(ends)} optimizer steps found")
short = lambda k: re.sub(r"at::native::|\(anonymous namespace\)::|void ", "", k)[:70]
for si in range(max(1, len(ends) - a.steps), len(ends)):
    seg = rows[ends[si - 1] + 1: ends[si] + 1]
    wall = seg[-1][1] - seg[0][0]
    busy = sum(e - s for s, e, _ in seg)
    gaps = [(seg[i + 1][0] - max(x[1] for x in seg[max(0, i - 3): i + 1]), i) for i in range(len(seg) - 1)]
    hist = collections.Counter()
    tot = collections.Counter()
    for g, _ in gaps:
        g = max(g, 0)
        b = "<2us" if g < 2000 else "2-5us" if g < 5000 else "5-10us" if g < 10000 else "10-20us" if g < 20000 else \
            "20-50us" if g < 50000 else "50-200us" if g < 200000 else ">200us"
        hist[b] += 1
        tot[b] += g
    print(f"step {si}: wall {wall/1e6:.2f} ms, busy {busy/1e6:.2f} ms, dispatches {len(seg)}")
    for b in ["<2us", "2-5us", "5-10us", "10-20us", "20-50us", "50-200us", ">200us"]:
        print(f"   gap {b:9s} n={hist[b]:5d} total {tot[b]/1e6:6.2f} ms")
    if si == len(ends) - 1:
        print("longest gaps (idle before the kernel on the right):")
        for g, i in sorted(gaps, reverse=True)[:a.top]:
            print(f"   {g/1e3:8.1f} us  after [{short(seg[i][2])}] -> [{short(seg[i+1][2])}]  at +{(seg[i][1]-seg[0][0])/1e6:.2f} ms")
        # coarse timeline: idle per 5 ms bucket
        print("idle per 5 ms of the step:")
        buckets = collections.Counter()
        for g, i in gaps:
            buckets[(seg[i][1] - seg[0][0]) // 5_000_000] += max(g, 0)
        print("   " + " ".join(f"{buckets[b]/1e6:.1f}" for b in range(int(wall // 5_000_000) + 1)))

        agg = collections.defaultdict(lambda: [0, 0])
        for s_, e_, k in seg:
            agg[k][0] += 1
            agg[k][1] += e_ - s_
        if a.csv:
            w = csv.writer(open(a.csv, "w"))
            w.writerow(["kernel", "dispatches_per_step", "total_ns_per_step", "avg_ns", f"step_wall_ns={wall}", f"step_busy_ns={busy}"])
            for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
                w.writerow([k, n, t, round(t / n, 1)])
        print("kernels of the last step by time:")
        for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
            print(f"   {t/1e6:6.2f} ms {n:5d} x {t/n/1e3:8.1f}us  {short(k)}")
        if a.region:
            lo, hi = a.region[0] * 1e6, a.region[1] * 1e6
            cnt = collections.Counter()
            dur = collections.Counter()
            seq = []
            for s_, e_, k in seg:
                if lo <= s_ - seg[0][0] < hi:
                    cnt[k] += 1
                    dur[k] += e_ - s_
                    seq.append(k)
            print(f"kernels in [{a.region[0]}, {a.region[1]}) ms: {sum(cnt.values())} dispatches, busy {sum(dur.values())/1e6:.2f} ms")
            for k, n in cnt.most_common(70):
                print(f"   {n:5d} {dur[k]/n/1e3:7.1f}us  {short(k)}")
