#!/usr/bin/env python3
"""Achievable HBM bandwidth of this box: a float4 device-to-device copy (read + write bytes / time), a read-only reduction
and a write-only fill at 1 GiB -- the ceiling the roofline fractions of DESIGN.md are read against (spec peak: 8 TB/s)."""
import json
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snipper_amd import _lib

dev = "cuda:0"
n = 1 << 28                       # 2^28 float32 = 1 GiB
src = torch.randn(n, device=dev)
dst = torch.empty_like(src)


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
    ev[0].record()
    for i in range(iters):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(iters))
    return ts[len(ts) // 2]


gb = n * 4 / 1e9
t_copy = timeit(lambda: dst.copy_(src))
lib = _lib.load()
t_own = timeit(lambda: _lib.check(lib.snipper_hbm_copy_probe(_lib.raw_stream(src.device), src.data_ptr(), dst.data_ptr(), n * 4), "copy probe"))
t_read = timeit(lambda: src.sum())
t_fill = timeit(lambda: dst.fill_(1.0))
print(json.dumps({"what": "HBM ceiling, 1 GiB float32 buffers, median of 20", "device": torch.cuda.get_device_name(0),
                  "copy_GBps_read_plus_write_own_float4_kernel": round(2 * gb / t_own * 1e3, 1),
                  "copy_GBps_read_plus_write_torch_copy": round(2 * gb / t_copy * 1e3, 1), "copy_ms_torch": round(t_copy, 4),
                  "read_GBps_sum_reduction": round(gb / t_read * 1e3, 1), "write_GBps_fill": round(gb / t_fill * 1e3, 1),
                  "spec_peak_GBps": 8000.0}))
