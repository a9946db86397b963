import sys, torch
sys.path.insert(0, '/root/repo')
from snipper_amd.fused import add_dropout_layer_norm
dev='cuda:0'; rows, C = 79000, 384
x = torch.randn(rows, C, device=dev, requires_grad=True)
zs = [torch.randn(rows, C, device=dev).bfloat16().requires_grad_(True) for _ in range(3)]
pos = torch.randn(rows, C, device=dev).bfloat16()
norms = [torch.nn.LayerNorm(C).to(dev) for _ in range(3)]
def chain(mode):
    cur = x
    outs = []
    for k in range(3):
        last = k == 2
        cur, y16, yq = add_dropout_layer_norm(cur, zs[k], norms[k], 0.1, True, pos=pos, want=(True if (mode == 'plain' or last) else 'lazy', True, True), seed=7 + k)
        outs += [y16, yq]
    return cur, outs
for mode in ('plain', 'lazy', 'plain', 'lazy'):
    for _ in range(3): chain(mode)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): chain(mode)
    e1.record(); torch.cuda.synchronize()
    print(mode, 'forward chain of 3: %.1f us per LayerNorm' % (e0.elapsed_time(e1) / 60 * 1e3))
