#!/usr/bin/env python3
"""Where does an iteration of the weight-stationary GEMM spend its cycles?  Needs the diagnostic build:

    SNIPPER_MSDA_LIB=libsnipper_msda_stamps.so SNIPPER_HIPCC_EXTRA=-DWRES_STAMPS python -m snipper_amd.build
    SNIPPER_MSDA_LIB=libsnipper_msda_stamps.so python tools/wres_stamps.py

Workgroup 0, waves 0 (group A) and 4 (group B) record s_memtime at the phase boundaries of their first 40 iterations:
1 compute starts, 2 compute done, 3 store phase starts (after the barrier), 4 epilogue done, 5 flush done, 6 DMAs issued,
7 chunk wait done (then the barrier).  Prints the median cycles per segment."""
import os, sys, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
buf = torch.zeros(2 * 320, dtype=torch.int64, device="cuda:0")
os.environ["SNIPPER_WRES_STAMPS"] = hex(buf.data_ptr())
from snipper_amd.dense import linear_wres_bf16
M, K, N = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (79000, 384, 384)))
x = torch.randn(M, K, device="cuda:0").bfloat16(); w = (torch.randn(N, K, device="cuda:0") / K ** 0.5).bfloat16(); b = torch.randn(N, device="cuda:0")
for _ in range(5):
    linear_wres_bf16(x, w, b)
torch.cuda.synchronize()
v = buf.cpu().tolist()
names = {1: "compute start", 2: "compute done", 3: "store phase start", 4: "epilogue done", 5: "flush done", 6: "DMAs issued", 7: "chunk wait done"}
for grp in (0, 1):
    st = [(x_ >> 56, x_ & ((1 << 56) - 1)) for x_ in v[grp * 320:(grp + 1) * 320] if x_]
    seg = {}
    for (a, ta), (b_, tb) in zip(st[:-1], st[1:]):
        seg.setdefault((a, b_), []).append(tb - ta)
    print(f"group {'AB'[grp]}: {len(st)} stamps")
    for k, ds in seg.items():
        ds = ds[2:] if len(ds) > 6 else ds
        print(f"   {names.get(k[0], k[0]):18s} -> {names.get(k[1], k[1]):18s}: median {statistics.median(ds):7.0f} cycles  (min {min(ds)}, max {max(ds)}, n {len(ds)})")
