#!/usr/bin/env python3
"""Where does a workgroup of the owner-computes tile kernel (msda_bwd_d48_tile2_kernel) spend its cycles?  Needs the
diagnostic build:

    SNIPPER_MSDA_LIB=libsnipper_msda_stamps.so SNIPPER_HIPCC_EXTRA=-DTILE2_STAMPS python -m snipper_amd.build
    SNIPPER_MSDA_LIB=libsnipper_msda_stamps.so python tools/tile2_stamps.py [sigma_px]

Thread 0 of every 61st workgroup records s_memtime at the phase boundaries (up to 128 stamps).  Prints, over the sampled
workgroups, the median / mean cycles of every segment and its share of the workgroup's life time."""
import os, sys, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
buf = torch.zeros(256 * 128 + 256 * 64, dtype=torch.int64, device="cuda:0")     # tile kernel, then query-side kernel
os.environ["SNIPPER_TILE2_STAMPS"] = hex(buf.data_ptr())
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import opbench
from opbench import MSDA, SHAPES
if os.environ.get("SNIPPER_TILE_KERNEL"):
    from snipper_amd import _lib as _l
    _l.set_param("tile_kernel", int(os.environ["SNIPPER_TILE_KERNEL"]))
sigma = float(sys.argv[1]) if len(sys.argv) > 1 else 0.01
S = sum(h * w for h, w in SHAPES)
v_, shapes, lsi, loc, attn, go = opbench.make(8, S, True, torch.bfloat16, sigma=sigma, grid=True)
run = lambda: MSDA.ms_deform_attn_backward(v_, shapes, lsi, loc, attn, go, 64, host_shapes=SHAPES)
for _ in range(3):
    run()
torch.cuda.synchronize()
buf.zero_()
run()
torch.cuda.synchronize()
allv = buf.cpu()
v = allv[:256 * 128].view(256, 128).tolist()
pv = allv[256 * 128:].view(256, 64).tolist()
names = {0: "start", 1: "marks+scan done", 2: "hits expanded, first fetch issued", 3: "round start", 4: "row loads issued",
         5: "decode+rank done", 6: "barrier (ranks)", 7: "prefix done", 8: "scatter done", 9: "rows written",
         10: "barrier (sorted)", 11: "next fetch issued + counters zeroed", 12: "accumulate done", 13: "barrier (round end)",
         14: "rounds done", 15: "tile stored"}
seg, life, rounds, hits = {}, [], [], []
for wg in v:
    st = [((x >> 56) & 0xff, x & ((1 << 56) - 1)) for x in wg if x]
    hits += [t for a, t in st if a == 200]
    st = [(a, t) for a, t in st if a != 200]
    if len(st) < 2 or st[0][0] != 0 or st[-1][0] != 15:
        continue
    life.append(st[-1][1] - st[0][1])
    rounds.append(sum(1 for a, _ in st if a == 3))
    acc = {}
    for (a, ta), (b, tb) in zip(st[:-1], st[1:]):
        acc[(a, b)] = acc.get((a, b), 0) + (tb - ta)
    for k, d in acc.items():
        seg.setdefault(k, []).append(d)
if life:
    print(f"sigma {sigma} px: {len(life)} sampled workgroups, life time median {statistics.median(life):.0f} mean {statistics.mean(life):.0f} cycles, "
          f"rounds per workgroup mean {statistics.mean(rounds):.2f}, hits per tile mean {statistics.mean(hits):.0f} max {max(hits)}")
    tot = statistics.mean(life)
    for k, ds in sorted(seg.items(), key=lambda kv: -sum(kv[1])):
        share = sum(ds) / len(life) / tot
        print(f"  {names.get(k[0], k[0])!s:38s} -> {names.get(k[1], k[1])!s:38s}: per-workgroup total median {statistics.median(ds):8.0f}  mean {statistics.mean(ds):8.0f}  "
              f"share {100 * share:5.1f} %  (n {len(ds)})")
else:
    print("tile kernel: no stamps (the matrix-pipe tile kernel carries none; SNIPPER_TILE_KERNEL=1 selects the vector kernel)")

# ---- the query-side kernel (msda_bwd_d48_patchbin_kernel): wave 0 of every 97th workgroup
pnames = {0: "start", 1: "loads of all levels issued, slots of all levels computed", 2: "barrier (slots)",
          3: "decode + marks of all levels done, grad_out rows in LDS", 4: "barrier (records)", 5: "marks stored",
          6: "gathers + dots + reductions + stores of all levels done", 7: "level: far atomics done"}
seg, life = {}, []
for wg in pv:
    st = [((x >> 56) & 0xff, x & ((1 << 56) - 1)) for x in wg if x]
    if len(st) < 4 or st[0][0] != 0:
        continue
    life.append(st[-1][1] - st[0][1])
    acc = {}
    for (a, ta), (b, tb) in zip(st[:-1], st[1:]):
        acc[(a, b)] = acc.get((a, b), 0) + (tb - ta)
    for k, d in acc.items():
        seg.setdefault(k, []).append(d)
if life:
    tot = statistics.mean(life)
    print(f"query-side kernel: {len(life)} sampled workgroups, life time median {statistics.median(life):.0f} mean {tot:.0f} cycles")
    for k, ds in sorted(seg.items(), key=lambda kv: -sum(kv[1])):
        print(f"  {pnames.get(k[0], k[0])!s:38s} -> {pnames.get(k[1], k[1])!s:38s}: per-workgroup total median {statistics.median(ds):8.0f}  mean {statistics.mean(ds):8.0f}  "
              f"share {100 * sum(ds) / len(life) / tot:5.1f} %  (n {len(ds)})")
