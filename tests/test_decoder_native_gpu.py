"""Round 6: one native call per decoder layer and direction (include/snipper_layers.h, snipper_amd/decoder_native.py) against the
Python-sequenced chain it replaces (DeformableTransformerDecoderLayer.forward_chain; reference
models/deformable_transformer.py:276-300, 329-333): the same launches with the same arguments, so outputs, refined reference
points and every gradient must agree BIT FOR BIT -- with dropout on, under the same seeds."""
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


@pytest.mark.parametrize("train", [True, False])
def test_native_decoder_layer_is_bit_identical_to_the_chain(train):
    from snipper_amd import fused
    from snipper_amd.deformable_transformer import DeformableTransformerDecoderLayer as Layer
    from snipper_amd.decoder_native import DecoderLayerFn
    from snipper_amd.model import build_model
    b = _bench()
    a = SimpleNamespace(hidden_dim=384, enc_layers=1, dec_layers=3, frames=3, future_frames=0, use_pytorch_deform=0,
                        batch=2, height=192, width=256)          # 6 048 memory rows: the premixed bf16 path (>= 4 096 rows)
    torch.manual_seed(0)
    model = build_model(b.model_args(a)).to(DEV).to(memory_format=torch.channels_last)
    model.train(train)
    with torch.no_grad():
        for n, p in model.named_parameters():          # real offsets / logits instead of the zero initialisation
            if "sampling_offsets" in n and n.endswith("weight"):
                p.normal_(0, 0.02)
            elif "attention_weights" in n:
                p.normal_(0, 0.3)
    imgs, _ = b.make_batches(a, DEV, 1, seed=3)[0]
    calls = []
    real = DecoderLayerFn.forward
    res = {}
    seed0 = fused._dropout_calls
    try:
        DecoderLayerFn.forward = staticmethod(lambda *x: (calls.append(1), real(*x))[1])
        for native in (True, False):
            Layer.native = native
            fused._dropout_calls = seed0                      # the same dropout seeds for both forms
            calls.clear()
            with torch.autocast("cuda", dtype=torch.bfloat16):
                out, _ = model(list(imgs))
            assert len(calls) == (3 if native else 0), calls
            k = out["all_layers"]["pred_kpts"].float()
            w = torch.linspace(-1, 1, k.numel(), device=DEV).view_as(k)
            names = [n for n, p in model.named_parameters() if p.requires_grad]
            params = [p for p in model.parameters() if p.requires_grad]
            grads = torch.autograd.grad((k * w).sum() + out["pred_logits"].float().sum(), params, allow_unused=True)
            res[native] = (k.detach(), out["pred_logits"].float().detach(), dict(zip(names, grads)))
    finally:
        Layer.native = True
        DecoderLayerFn.forward = real
    assert torch.equal(res[True][0], res[False][0]) and torch.equal(res[True][1], res[False][1])
    dec = [n for n in res[True][2] if ".decoder." in n or "query_embed" in n or "reference_points" in n or "temporal_embed" in n]
    assert len(dec) > 50
    for n in dec:                                             # everything the decoder's own launches produce: bit for bit
        g1, g0 = res[True][2][n], res[False][2][n]
        if g1 is None or g0 is None:
            assert g1 is None and g0 is None, n
            continue
        assert torch.equal(g1, g0), (n, float((g1 - g0).abs().max()))
    for n, g1 in res[True][2].items():                        # the rest goes through the encoder's backward (its far-tap atomics
        g0 = res[False][2][n]                                 # are unordered): equal to float32 summation noise
        if g1 is None or g0 is None:
            continue
        d = float((g1.double() - g0.double()).norm() / g0.double().norm().clamp_min(1e-30))
        assert d < 1e-4, (n, d)
