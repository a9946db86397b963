"""BASELINE configs[4] (T = 4 observed + 2 forecast frames, 60 queries per frame -> 360 decoder queries) on the GPU.

The forecast path (reference models/deformable_transformer.py:244-343 with n_future_frame > 0: the decoder's queries of
the future frames attend to the observed frames only) was pinned at fixture size by goldens g3 ``dec_t3f2`` / g6; here it
runs at the benchmark geometry (600x800 -> 75x100 / 38x50 / 19x25 maps, hidden 384, 8 heads of 48) in training mode:
the HIP path against this package's ``use_pytorch_deform=1`` formulation in float32 (outputs and gradients), and the
bench command itself with ``--future-frames 2``."""
import json
import os
import subprocess
import sys
from types import SimpleNamespace

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_forecast_configuration_hip_vs_pytorch_formulation_at_600x800():
    from snipper_amd import _lib
    from snipper_amd.model import build_model
    T, F = 4, 2
    args = dict(hidden_dim=384, nheads=8, enc_layers=1, dec_layers=2, dim_feedforward=1024, dropout=0.0,
                num_feature_levels=3, dec_n_points=4, enc_n_points=4, num_frames=T, num_future_frames=F, num_kpts=15,
                position_embedding="sine", backbone="resnet50", lr_backbone=1e-5, masks=False, dilation=False,
                num_queries=60, aux_loss=True)
    torch.manual_seed(7)
    hip = build_model(SimpleNamespace(use_pytorch_deform=False, **args)).to(DEV).train()
    ref = build_model(SimpleNamespace(use_pytorch_deform=True, **args)).to(DEV).train()
    with torch.no_grad():
        for n, p in hip.named_parameters():          # real offsets / logits instead of the zero initialisation
            if "sampling_offsets" in n and n.endswith("weight"):
                p.normal_(0, 0.02)
            elif "attention_weights" in n:
                p.normal_(0, 0.3)
    ref.load_state_dict(hip.state_dict(), strict=True)
    g = torch.Generator().manual_seed(9)
    snippets = [torch.rand(T * 3, 600, 800, generator=g).to(DEV)]
    res = []
    for m in (hip, ref):
        out, _ = m(snippets)
        if m is hip:
            variant = _lib.last_variant()
        assert out["pred_kpts2d"].shape == (1, 60, T + F, 15, 3)
        loss = sum((out[k].float() ** 2).mean() for k in ("pred_logits", "pred_kpts2d", "pred_depth"))
        loss = loss + sum((h.float() ** 2).mean() for h in out["heatmaps"])
        names = ["transformer.decoder.layers.1.cross_attn.sampling_offsets.0.weight",
                 "transformer.decoder.layers.0.self_attn.in_proj_weight",
                 "transformer.encoder.layers.0.self_attn.value_proj.weight",
                 "input_proj.0.0.weight"]
        pd = dict(m.named_parameters())
        grads = torch.autograd.grad(loss, [pd[n] for n in names])
        res.append((out, float(loss.detach()), grads))
    assert variant.startswith("d48"), variant
    (oh, lh, gh), (orf, lr, gr) = res
    for k in ("pred_logits", "pred_kpts2d", "pred_depth"):
        torch.testing.assert_close(oh[k], orf[k], rtol=2e-3, atol=3e-4, msg=lambda m: f"{k}: {m}")
    assert abs(lh - lr) <= 1e-4 * abs(lr)
    for a, b in zip(gh, gr):
        rel = float((a - b).norm() / b.norm().clamp_min(1e-20))
        assert rel <= 2e-3, rel


def test_bench_runs_the_forecast_configuration():
    """`bench.py --future-frames 2` (configs[4] on one GPU): a few steps must run and report a finite loss."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--future-frames", "2", "--steps", "3", "--warmup", "2",
           "--no-extras", "--no-cpu-baseline"]
    p = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads(p.stdout.strip().splitlines()[-1])
    assert "T=4+2" in line["config"]["workload"] and line["value"] > 0
    assert line["final_loss"] == line["final_loss"] and abs(line["final_loss"]) < 1e9      # finite
