"""Round 6: one native call per ENCODER layer and direction (include/snipper_layers.h, snipper_amd/encoder_native.py) against
DeformableTransformerEncoderLayer.forward_fused's node-per-module sequence (reference models/deformable_transformer.py:200-216,
models/ops/modules/ms_deform_attn.py:99-243): the same launches with the same arguments, dropout on, same seeds -- the model's
outputs must agree bit for bit, the gradients to the last bits (the owner-computes backward's far-tap atomics are unordered)."""
import importlib.util
import os
from types import SimpleNamespace

import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    return b


@pytest.mark.parametrize("frames,enc_layers", [(3, 2), (4, 1)])
def test_native_encoder_layer_equals_the_node_per_module_path(frames, enc_layers):
    from snipper_amd import fused
    from snipper_amd.deformable_transformer import DeformableTransformerEncoderLayer as Layer
    from snipper_amd.encoder_native import EncoderLayerFn
    from snipper_amd.model import build_model
    b = _bench()
    a = SimpleNamespace(hidden_dim=384, enc_layers=enc_layers, dec_layers=1, frames=frames, future_frames=0, use_pytorch_deform=0,
                        batch=2, height=192, width=256)          # 2 x frames x 1 008 rows >= 8 192 only for frames >= 5 ...
    a.height, a.width = 256, 352                                  # ... so a larger map: 2 x 3 x 1 848 = 11 088 rows
    torch.manual_seed(0)
    model = build_model(b.model_args(a)).to(DEV).to(memory_format=torch.channels_last)
    model.train()
    with torch.no_grad():
        for n, p in model.named_parameters():          # real offsets / logits instead of the zero initialisation
            if "sampling_offsets" in n and n.endswith("weight"):
                p.normal_(0, 0.02)
            elif "attention_weights" in n:
                p.normal_(0, 0.3)
    imgs, _ = b.make_batches(a, DEV, 1, seed=3)[0]
    calls = []
    real = EncoderLayerFn.forward
    res = {}
    seed0 = fused._dropout_calls
    try:
        EncoderLayerFn.forward = staticmethod(lambda *x: (calls.append(1), real(*x))[1])
        for native in (True, False, "again"):             # "again": the per-module path a second time -- its own run-to-run noise
            Layer.native = native is True
            fused._dropout_calls = seed0
            calls.clear()
            with torch.autocast("cuda", dtype=torch.bfloat16):
                out, _ = model(list(imgs))
            assert len(calls) == (enc_layers if native is True else 0), calls
            k = out["all_layers"]["pred_kpts"].float()
            w = torch.linspace(-1, 1, k.numel(), device=DEV).view_as(k)
            hm = sum(h.float().pow(2).sum() for h in out["heatmaps"])
            names = [n for n, p in model.named_parameters() if p.requires_grad]
            params = [p for p in model.parameters() if p.requires_grad]
            grads = torch.autograd.grad((k * w).sum() + out["pred_logits"].float().sum() + 1e-3 * hm, params, allow_unused=True)
            res[native] = (k.detach(), out["pred_logits"].float().detach(), dict(zip(names, grads)))
    finally:
        Layer.native = True
        EncoderLayerFn.forward = real
    assert torch.equal(res[True][0], res[False][0]) and torch.equal(res[True][1], res[False][1])
    assert torch.equal(res["again"][0], res[False][0])
    rel = lambda x, y: float((x.double() - y.double()).norm() / y.double().norm().clamp_min(1e-30))
    n_equal, worst, worst_self = 0, (0.0, ""), (0.0, "")
    for n, g1 in res[True][2].items():
        g0, g2 = res[False][2][n], res["again"][2][n]
        if g1 is None or g0 is None:
            assert g1 is None and g0 is None, n
            continue
        n_equal += int(torch.equal(g1, g0))
        worst = max(worst, (rel(g1, g0), n))
        worst_self = max(worst_self, (rel(g2, g0), n))
    print(f"native vs per-module: {n_equal}/{len(res[True][2])} gradients bit-equal, worst {worst}; per-module vs itself: worst {worst_self}")
    # the two forms launch the same kernels on the same data; what can differ is what differs between two runs of ONE form: the
    # owner-computes backward's far-tap atomics are unordered, one flipped bf16 rounding behind them moves a backbone weight
    # gradient by ~2.5e-3 in relative L2 (measured between two runs of the per-module form: 0 in some pairs of runs, 1.2e-4 ...
    # 2.6e-3 in others).  ONE self-comparison is a sample of that noise, not its bound (it was exactly 0 in a run where the native
    # form hit the other rounding: 2.4e-3), hence the floor; a wrong pointer or a missing term in the composite shows up as O(1).
    assert worst[0] <= max(1e-2, 4.0 * worst_self[0]), (worst, worst_self)
    assert n_equal >= len(res[True][2]) // 4, n_equal          # (what does not pass through the encoder's backward is bit-equal: 56-59 of 145)
