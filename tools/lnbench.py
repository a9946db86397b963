"""Micro-benchmark of the fused residual + dropout + LayerNorm kernels at the encoder's shape."""
import sys, torch
sys.path.insert(0, '/root/repo')
from snipper_amd import _lib
from snipper_amd.fused import AddDropoutLayerNorm
dev = 'cuda:0'
rows, C = 79000, 384
x = torch.randn(rows, C, device=dev, requires_grad=True)
z = torch.randn(rows, C, device=dev).bfloat16().requires_grad_(True)
pos = torch.randn(rows, C, device=dev).bfloat16()
gamma = torch.ones(C, device=dev, requires_grad=True); beta = torch.zeros(C, device=dev, requires_grad=True)
g32 = torch.randn(rows, C, device=dev); g16 = torch.randn(rows, C, device=dev).bfloat16(); gq = g16.clone()
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for blocks in (512, 1024, 2048, 4096, 8192):
    _lib.set_param("ln_bwd_blocks", blocks)
    def fb():
        y = AddDropoutLayerNorm.apply(x, z, pos, gamma, beta, 0.1, 1e-5, (True, True, True), 1)
        torch.autograd.grad(y, (x, z, gamma, beta), (g32, g16, gq))
    def f():
        with torch.no_grad():
            AddDropoutLayerNorm.apply(x, z, pos, gamma, beta, 0.1, 1e-5, (True, True, True), 1)
    tf = t(f); tfb = t(fb)
    print(f"bwd blocks {blocks}: fwd(no save) {tf:.1f} us, fwd+bwd {tfb:.1f} us", flush=True)
