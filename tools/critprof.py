import sys, time, torch, importlib.util
sys.path.insert(0, '/root/repo')
spec = importlib.util.spec_from_file_location("bench_mod", "/root/repo/bench.py"); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
from types import SimpleNamespace
from snipper_amd.criterion import build_criterion
a = SimpleNamespace(hidden_dim=384, enc_layers=6, dec_layers=6, frames=4, future_frames=0, use_pytorch_deform=0, batch=2, height=600, width=800)
dev = 'cuda:0'
crit = build_criterion(b.criterion_args(a)).to(dev)
_, tgt = b.make_batches(a, dev, 1, seed=1000)[0]
n_dec, bs, nq, T, K = 6, 2, 60, 4, 15
logits = torch.randn(n_dec, bs, nq, T, 2, device=dev, requires_grad=True)
kpts = torch.rand(n_dec, bs, nq, T, K, 4, device=dev, requires_grad=True)
hms = [torch.rand(bs, T, h, w, 8, K, device=dev, requires_grad=True) for h, w in [(75, 100), (38, 50), (19, 25)]]
def make_out():
    return {"pred_logits": logits[-1], "pred_kpts2d": kpts[-1, ..., :3], "pred_depth": kpts[-1, ..., 3:4], "heatmaps": hms,
            "all_layers": {"pred_logits": logits, "pred_kpts": kpts}}
def full():
    losses, _ = crit(make_out(), tgt["targets"]); total = crit.weighted_sum(losses); total.backward()
def tm(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    return (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3
print("full criterion fwd+bwd: issue %.2f ms, wall %.2f ms" % tm(full))
from snipper_amd.criterion import _stack_layers
def match_only():
    with torch.no_grad():
        l, k2, dp = _stack_layers(make_out()); crit.matcher.match_all_layers(l, k2, dp, tgt["targets"])
print("matcher only: issue %.2f ms, wall %.2f ms" % tm(match_only))
def heat_only():
    x = crit.loss_heatmap(make_out(), tgt["targets"]); x.backward()
print("heatmap loss fwd+bwd: issue %.2f ms, wall %.2f ms" % tm(heat_only))
