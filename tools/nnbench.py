"""Data-gradient GEMM: own NN kernel vs torch.mm (hipBLASLt) on the step's shapes."""
import sys, torch
sys.path.insert(0, '/root/repo')
from snipper_amd.dense import linear_nn_bf16, linear_bf16
dev = 'cuda:0'
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (M, K, N) in [(79000, 384, 384), (79000, 1024, 384), (79000, 384, 1024), (79000, 320, 384), (60000, 128, 512), (60000, 512, 256), (15200, 256, 1024), (15200, 1024, 512), (3800, 512, 2048), (3800, 2048, 1024)]:
    x = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(K, N, device=dev) / K ** 0.5).bfloat16()
    wt = w.t().contiguous()
    print(f"M={M} K={K} N={N}: nn kernel {t(lambda: linear_nn_bf16(x, w)):.1f} us | torch.mm {t(lambda: torch.mm(x, w)):.1f} us | nt kernel (pre-transposed W) {t(lambda: linear_bf16(x, wt)):.1f} us", flush=True)
