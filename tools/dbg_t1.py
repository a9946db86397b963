"""debug: bf16-autocast forward with grad enabled vs under no_grad -- per-module outputs by NAME."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace
import torch
from snipper_amd.model import build_model
import snipper_amd.backbone as bb
DEV = "cuda:0"
T = 2
args = dict(hidden_dim=384, nheads=8, enc_layers=1, dec_layers=1, dim_feedforward=1024, dropout=0.0,
            num_feature_levels=3, dec_n_points=4, enc_n_points=4, num_frames=T, num_future_frames=0, num_kpts=15,
            position_embedding="sine", backbone="resnet50", lr_backbone=1e-5, masks=False, dilation=False,
            num_queries=60, aux_loss=True, use_pytorch_deform=False)
torch.manual_seed(11)
m = build_model(SimpleNamespace(**args)).to(DEV).train()
rand = "--rand" in sys.argv
if rand:
    with torch.no_grad():
        for n, p in m.named_parameters():
            if "sampling_offsets" in n and n.endswith("weight"): p.normal_(0, 0.02)
            elif "attention_weights" in n: p.normal_(0, 0.3)
g = torch.Generator().manual_seed(6)
sn = [torch.rand(3 * T, 600, 800, generator=g).to(DEV)]
log = {}
def hook(name):
    def f(mod, a, o):
        t = o[0] if isinstance(o, (tuple, list)) else o
        if isinstance(t, torch.Tensor) and t.is_floating_point():
            log[name] = t.detach().float().clone()
    return f
for n, mod in m.named_modules():
    if isinstance(mod, (bb.Bottleneck,)) or n.endswith("self_attn") or n.endswith("cross_attn") or "encoder.layers" in n and n.count(".") == 3:
        mod.register_forward_hook(hook(n))
m.backbone[0].register_forward_hook(lambda mod, a, o: [log.__setitem__(f"feat{k}", v.tensors.detach().float().clone()) for k, v in o.items()] and None)
runs = []
for mode in ("grad", "nograd", "fp32"):
    log.clear()
    with torch.set_grad_enabled(mode != "nograd"), torch.autocast("cuda", dtype=torch.bfloat16, enabled=mode != "fp32"):
        out, _ = m(sn)
    runs.append(dict(log))
rel = lambda a, b: float((a - b).norm() / b.norm().clamp_min(1e-20))
for k in runs[0]:
    if k in runs[1] and k in runs[2] and runs[0][k].shape == runs[1][k].shape:
        print(f"{k:45s} grad-vs-nograd {rel(runs[0][k], runs[1][k]):.3e}   grad-vs-fp32 {rel(runs[0][k], runs[2][k]):.3e}   nograd-vs-fp32 {rel(runs[1][k], runs[2][k]):.3e}")
import snipper_amd.shadow as sh
names = {id(p): n for n, p in m.named_parameters()}
bad = 0
for k, e in sh._entries.items():
    w = e.ref()
    if w is None: continue
    src = w.detach().float()
    if e.scale is not None:
        src = src * e.scale.view(-1, 1, 1, 1)
    r = rel(e.dst.float(), src)
    if r > 1e-2:
        bad += 1
        print("BAD shadow", names.get(k), tuple(w.shape), w.stride(), e.dst.stride(), f"{r:.3e}")
for k, mm in sh._merged.items():
    ps = [r() for r in mm.refs]
    src = torch.cat([ps[0], ps[2]], 0).detach().float()
    print("merged", names.get(k[0]), f"{rel(mm.w.float(), src):.3e}", f"bias {rel(mm.b, torch.cat([ps[1], ps[3]]).detach()):.3e}")
print("entries", len(sh._entries), "bad", bad)
for k, mm in sh._merged.items():
    ps = [r() for r in mm.refs]
    want = torch.cat([ps[1], ps[3]]).detach()
    print("m.b[:12]", mm.b[:12].tolist())
    print("want[:12]", want[:12].tolist())
    print("m.b[192:200]", mm.b[192:200].tolist(), "want", want[192:200].tolist())
    print("versions", mm.versions, [p._version for p in ps])
    break
