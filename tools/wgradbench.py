import torch, time, sys
sys.path.insert(0, '/root/repo')
from snipper_amd.dense import wgrad_bf16
from snipper_amd import _lib
dev = 'cuda:0'
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (M, N, Kc) in [(79000, 384, 384), (79000, 1024, 384), (79000, 384, 1024), (79000, 192, 384), (79000, 96, 384), (60000, 512, 256), (15200, 1024, 512), (3800, 2048, 1024)]:
    g = torch.randn(M, N, device=dev).bfloat16(); x = torch.randn(M, Kc, device=dev).bfloat16()
    base = t(lambda: torch.mm(g.t(), x))
    baseb = t(lambda: g.sum(0))
    res = []
    for wgs in (256, 512, 768, 1024, 2048):
        _lib.set_param("wgrad_wgs", wgs)
        res.append((wgs, round(t(lambda: wgrad_bf16(g, x)), 1)))
    print(f"M={M} N={N} Kc={Kc}: torch.mm {base:.1f} us + bias-sum {baseb:.1f} us | ours (incl. bias) {res}", flush=True)
