"""BASELINE.json configs[0]: T=1, enc2/dec4, hidden_dim 384, 60 queries on 600x800 frames through the
``use_pytorch_deform=1`` CPU path (plumbing, no GPU).

* the transformer of this package against the REFERENCE transformer imported from /root/reference (build container
  only: skipped where the tree is absent), same state_dict, same inputs, at the full 75x100 / 38x50 / 19x25 geometry;
* the whole model (ResNet-50 restatement + projections + transformer + heads) forward on one 600x800 frame.
Both are a few seconds of CPU work; bench.py's ``cpu_baseline`` times the same configuration on the GPU box's host.
"""
import os
import sys
import types
from types import SimpleNamespace

import pytest
import torch
import torch.nn.functional as F

from snipper_amd.deformable_transformer import DeformableTransformer

REF = "/root/reference"
HW = [(75, 100), (38, 50), (19, 25)]          # 600x800 at strides 8 / 16 / 32
CFG = dict(d_model=384, nhead=8, num_encoder_layers=2, num_decoder_layers=4, dim_feedforward=1024, dropout=0.1,
           activation="relu", return_intermediate_dec=True, num_feature_levels=3, dec_n_points=4, enc_n_points=4,
           n_frame=1, n_future_frame=0, use_pytroch_deform=True, num_keypoints=15)


def _inputs(T=1, bs=1, d=384, nq=60):
    g = torch.Generator().manual_seed(7)
    srcs = [torch.randn(bs, d, T, h, w, generator=g) for h, w in HW]
    masks = [torch.zeros(bs, d, T, h, w, dtype=torch.bool) for h, w in HW]
    pos = [torch.randn(bs, d, T, h, w, generator=g) for h, w in HW]
    query_embed = torch.randn(nq * T, 2 * d, generator=g)
    return srcs, masks, pos, query_embed


def _import_reference_transformer():
    tv = types.ModuleType("torchvision")
    tv.__version__ = "0.9.0"
    tv.ops = types.ModuleType("torchvision.ops")
    tv.ops.misc = types.ModuleType("torchvision.ops.misc")
    tv.ops.misc.interpolate = F.interpolate
    for k, v in (("torchvision", tv), ("torchvision.ops", tv.ops), ("torchvision.ops.misc", tv.ops.misc)):
        sys.modules.setdefault(k, v)
    sys.dont_write_bytecode = True
    # snipper_amd.install() (tests/test_capi.py) may have registered this package's mirrors under the reference's module
    # names: take them out for the import so that the class below really is the reference's, then put them back
    saved = {k: sys.modules.pop(k) for k in list(sys.modules) if k == "models" or k.startswith("models.")}
    sys.path.insert(0, REF)
    try:
        from models.deformable_transformer import DeformableTransformer as RefTransformer
    finally:
        sys.path.remove(REF)
        for k in [k for k in sys.modules if k == "models" or k.startswith("models.")]:
            del sys.modules[k]
        sys.modules.update(saved)
    assert RefTransformer.forward.__code__.co_filename.startswith(REF)
    return RefTransformer


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "models")), reason="reference tree only exists in the build container")
def test_config1_transformer_matches_the_imported_reference():
    Ref = _import_reference_transformer()
    torch.manual_seed(3)
    ref = Ref(**CFG).eval()
    with torch.no_grad():
        ref.temporal_embed.normal_()                     # deformable_transformer.py:52 leaves it uninitialised
        for n, p in ref.named_parameters():             # give the zero-initialised offset / logit Linears real values
            if "sampling_offsets" in n and n.endswith("weight"):
                p.normal_(0, 0.02)
            elif "attention_weights" in n:
                p.normal_(0, 0.3)
    ours = DeformableTransformer(**CFG).eval()
    ours.load_state_dict(ref.state_dict(), strict=True)
    assert sum(p.numel() for p in ours.parameters()) == 9_546_434          # SURVEY.md section 8b (T=1, enc2/dec4)
    srcs, masks, pos, qe = _inputs()
    with torch.no_grad():
        hs_r, heat_r, init_r, inter_r, _ = ref(srcs, masks, pos, qe)
        hs_o, heat_o, init_o, inter_o, _ = ours(srcs, masks, pos, qe)
    assert hs_o.shape == hs_r.shape == (4, 1, 1, 60, 384)
    torch.testing.assert_close(hs_o, hs_r, rtol=2e-4, atol=2e-4)
    torch.testing.assert_close(inter_o, inter_r, rtol=2e-4, atol=1e-5)
    torch.testing.assert_close(init_o, init_r, rtol=1e-5, atol=1e-6)
    for a, b in zip(heat_o, heat_r):
        torch.testing.assert_close(a, b, rtol=2e-4, atol=2e-4)


def test_config1_whole_model_forward_on_cpu():
    from snipper_amd.model import build_model
    args = SimpleNamespace(hidden_dim=384, nheads=8, enc_layers=2, dec_layers=4, dim_feedforward=1024, dropout=0.1,
                           num_feature_levels=3, dec_n_points=4, enc_n_points=4, num_frames=1, num_future_frames=0,
                           use_pytorch_deform=True, num_kpts=15, position_embedding="sine", backbone="resnet50",
                           lr_backbone=1e-5, masks=False, dilation=False, num_queries=60, aux_loss=True)
    torch.manual_seed(0)
    model = build_model(args).eval()
    img = torch.rand(1, 3, 600, 800)
    with torch.no_grad():
        out, (init_ref, inter_refs, _) = model(list(img))
    assert out["pred_logits"].shape == (1, 60, 1, 2)
    assert out["pred_kpts2d"].shape == (1, 60, 1, 15, 3) and out["pred_depth"].shape == (1, 60, 1, 15, 1)
    assert [tuple(h.shape[2:4]) for h in out["heatmaps"]] == HW
    assert len(out["aux_outputs"]) == 3
    for v in (out["pred_logits"], out["pred_kpts2d"], out["pred_depth"]):
        assert torch.isfinite(v).all()
