"""BASELINE configs[1]: T = 1, enc2 / dec4, hidden 384, 60 queries, one MI355X -- the HIP MSDeformAttn path against the
``use_pytorch_deform=1`` formulation at model level (the op-level counterpart of models/ops/test.py is
tests/test_msda_gpu.py; the T = 1 module itself is pinned to the reference by goldens g3_module_{enc,dec}_t1_d48).

With one frame there are no temporal neighbours: the tied module's mix is the identity (``_forward_tied``'s ``identity``
branch in float32, ``TiedSampler`` with T2 = 1 under bf16 autocast), reference models/deformable_transformer.py:361-380 with
``num_frames=1``."""
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

ARGS = dict(hidden_dim=384, nheads=8, enc_layers=2, dec_layers=4, dim_feedforward=1024, dropout=0.0,
            num_feature_levels=3, dec_n_points=4, enc_n_points=4, num_frames=1, num_future_frames=0, num_kpts=15,
            position_embedding="sine", backbone="resnet50", lr_backbone=1e-5, masks=False, dilation=False,
            num_queries=60, aux_loss=True)
GRAD_NAMES = ["transformer.encoder.layers.0.self_attn.sampling_offsets.0.weight",
              "transformer.encoder.layers.1.self_attn.attention_weights.0.bias",
              "transformer.encoder.layers.0.self_attn.value_proj.weight",
              "transformer.decoder.layers.3.cross_attn.sampling_offsets.0.weight",
              "transformer.decoder.layers.0.cross_attn.value_proj.weight",
              "transformer.level_embed", "input_proj.0.0.weight"]


def _pair():
    from snipper_amd.model import build_model
    torch.manual_seed(11)
    hip = build_model(SimpleNamespace(use_pytorch_deform=False, **ARGS)).to(DEV).train()
    ref = build_model(SimpleNamespace(use_pytorch_deform=True, **ARGS)).to(DEV).train()
    with torch.no_grad():
        for n, p in hip.named_parameters():          # real offsets / logits instead of the zero initialisation
            if "sampling_offsets" in n and n.endswith("weight"):
                p.normal_(0, 0.02)
            elif "attention_weights" in n:
                p.normal_(0, 0.3)
    ref.load_state_dict(hip.state_dict(), strict=True)
    assert hip.transformer.encoder.num_layers == 2 and hip.transformer.decoder.num_layers == 4
    return hip, ref


def _loss(out):
    loss = sum((out[k].float() ** 2).mean() for k in ("pred_logits", "pred_kpts2d", "pred_depth"))
    return loss + sum((h.float() ** 2).mean() for h in out["heatmaps"])


def test_config1_hip_vs_pytorch_formulation_at_600x800():
    from snipper_amd import _lib
    hip, ref = _pair()
    g = torch.Generator().manual_seed(5)
    snippets = [torch.rand(3, 600, 800, generator=g).to(DEV) for _ in range(2)]      # two snippets of one frame
    res = []
    for m in (hip, ref):
        out, _ = m(snippets)
        assert out["pred_kpts2d"].shape == (2, 60, 1, 15, 3)
        fwd_variant = _lib.last_variant()
        loss = _loss(out)
        pd = dict(m.named_parameters())
        grads = torch.autograd.grad(loss, [pd[n] for n in GRAD_NAMES])
        res.append((out, float(loss.detach()), grads, fwd_variant, _lib.last_variant()))
    (oh, lh, gh, vf, vb), (orf, lr, gr, _, _) = res
    assert vf == "d48_lp12", vf                # last forward = the decoder's cross attention (D = 48, L*P = 12)
    assert vb == "d48_owner", vb               # last backward = the first encoder layer's owner-computes kernels
    for k in ("pred_logits", "pred_kpts2d", "pred_depth"):
        torch.testing.assert_close(oh[k], orf[k], rtol=2e-3, atol=3e-4, msg=lambda m: f"{k}: {m}")
    for a, b in zip(oh["heatmaps"], orf["heatmaps"]):
        torch.testing.assert_close(a, b, rtol=2e-3, atol=3e-4)
    assert abs(lh - lr) <= 1e-4 * abs(lr)
    for n, a, b in zip(GRAD_NAMES, gh, gr):
        rel = float((a - b).norm() / b.norm().clamp_min(1e-20))
        assert rel <= 2e-3, (n, rel)


def test_config1_bf16_autocast_path_runs_the_tied_sampler_with_one_frame():
    """The training precision of the bench (bf16 autocast: bf16 value, TiedSampler node with T2 = 1, fused residual
    path) against the float32 HIP evaluation of the same model: outputs and gradients to bf16 accuracy."""
    from snipper_amd import _lib
    hip, _ = _pair()
    g = torch.Generator().manual_seed(6)
    snippets = [torch.rand(3, 600, 800, generator=g).to(DEV)]
    res = []
    for amp in (True, False):
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
            out, _ = hip(snippets)
        loss = _loss(out)
        pd = dict(hip.named_parameters())
        grads = torch.autograd.grad(loss, [pd[n] for n in GRAD_NAMES])
        assert _lib.last_variant() == ("d48_owner_mfma" if amp else "d48_owner"), _lib.last_variant()   # bf16 grad_out rows: matrix-pipe tile kernel
        res.append((out, grads))
    (oa, ga), (of, gf) = res
    rel = lambda a, b: float((a.detach().float() - b.detach().float()).norm() / b.detach().float().norm().clamp_min(1e-20))
    errs = {k: rel(oa[k], of[k]) for k in ("pred_logits", "pred_kpts2d", "pred_depth")}
    gerrs = {n: rel(a, b) for n, a, b in zip(GRAD_NAMES, ga, gf)}
    print("[config1 bf16 vs fp32] outputs", {k: f"{v:.3e}" for k, v in errs.items()}, "grads", {".".join(k.split(".")[1:]): f"{v:.3e}" for k, v in gerrs.items()})
    # measured: outputs 3.0-4.5e-2 (bf16 backbone 0.9 % -> encoder 1.4 % -> four decoder layers); the shadow bug that this
    # test found (merged offset bias lost under autocast) gave 0.39-0.55
    for k, v in errs.items():
        assert v <= 8e-2, (k, v)
    # measured (round 5): the two sampling_offsets weights (first encoder layer 0.18, last decoder layer 0.21), every other
    # gradient 3.0e-2 .. 7.1e-2: bounds = ~2x the other ones, 1.4x the worst.  WHICH layer the 0.2 comes from:
    # none -- tools/dbg_attribution.py, profiles/r05_bf16_attribution.jsonl: switching the value projection, the output
    # projection or the offset / logit projection of the encoder back to float32 leaves it at 0.16-0.20, while in PURE float32 a
    # 2^-9 relative perturbation of the input images (one bf16 rounding) already moves these two gradients by 2.3-2.6 % and all
    # the others by 0.4-0.8 %: the gradient with respect to a sampling location is piecewise constant per pixel cell
    # (ms_deform_im2col_cuda.cuh:87-159), so ANY upstream perturbation is amplified 4-6x on the location side -- the bf16
    # step's general 3-7 % lands at ~0.2 there.
    for n, v in gerrs.items():
        assert v <= (0.30 if "sampling_offsets" in n else 0.14), (n, v)


def test_bench_runs_config1():
    """`bench.py --frames 1 --enc-layers 2 --dec-layers 4`: the training step of configs[1]'s model on one GPU."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--frames", "1", "--enc-layers", "2", "--dec-layers", "4",
           "--steps", "3", "--warmup", "2", "--no-extras", "--no-cpu-baseline"]
    p = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads(p.stdout.strip().splitlines()[-1])
    assert "T=1+0 enc2/dec4" in line["config"]["workload"] and line["value"] > 0
    assert line["final_loss"] == line["final_loss"] and abs(line["final_loss"]) < 1e9      # finite
