"""BASELINE configs[2] at its REAL size on one MI355X (``-m gpu``): T = 4 frames, batch 2, enc6 / dec6, hidden 384, 60 queries,
600x800 snippets -- the step `bench.py` times.

VERDICT r03 (weak #2): at this size the bench's step had only been asserted *finite*; kernels were compared at full size
against the oracle and whole models only at T = 1 / enc2 / dec4 or at 192x256.  Two tests:

  1. float32, HIP kernels against the reference's ``use_pytorch_deform=1`` formulation with the same state_dict: outputs and
     ten parameter gradients spread over backbone / input projections / encoder / decoder / heads (the counterpart of
     tests/test_config1_gpu.py at configs[2]'s depth and frame count; reference README.md:67-125 flags, main.py:183-221);
  2. the bench's own step (bf16 autocast, weight shadows, flat parameters, clipping + AdamW in csrc/adamw_flat.cuh, real
     SetCriterion + Hungarian matcher) for 4 steps against float32 ``use_pytorch_deform=1`` with per-parameter
     ``clip_grad_norm_`` + ``torch.optim.AdamW`` from the same weights on the same batches: loss per step and the norms of
     updated weights.
"""
import os
import sys
from types import SimpleNamespace

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

GRAD_NAMES = ["backbone.0.body.layer2.0.conv1.weight",
              "backbone.0.body.layer4.2.conv3.weight",
              "input_proj.0.0.weight", "input_proj.2.0.weight",
              "transformer.level_embed",
              "transformer.encoder.layers.0.self_attn.sampling_offsets.0.weight",
              "transformer.encoder.layers.3.linear1.weight",
              "transformer.encoder.layers.5.self_attn.value_proj.weight",
              "transformer.decoder.layers.0.cross_attn.value_proj.weight",
              "transformer.decoder.layers.5.cross_attn.sampling_offsets.0.weight",
              "joint_embed.0.3.layers.0.weight"]


def _bench_args(use_pytorch_deform):
    return SimpleNamespace(hidden_dim=384, enc_layers=6, dec_layers=6, frames=4, future_frames=0, batch=2, height=600,
                           width=800, use_pytorch_deform=int(use_pytorch_deform))


def _build(use_pytorch_deform, seed=42):
    import bench
    from snipper_amd.model import build_model
    margs = bench.model_args(_bench_args(use_pytorch_deform))
    margs.dropout = 0.0                                   # the arms must not differ by their random streams
    torch.manual_seed(seed)
    model = build_model(margs).to(DEV).to(memory_format=torch.channels_last)
    return model.train()


def _real_offsets(model):
    with torch.no_grad():                                 # real offsets / logits instead of the zero initialisation
        g = torch.Generator().manual_seed(3)
        for n, p in model.named_parameters():
            if "sampling_offsets" in n and n.endswith("weight"):
                p.copy_(torch.randn(p.shape, generator=g).to(DEV) * 0.02)
            elif "attention_weights" in n:
                p.copy_(torch.randn(p.shape, generator=g).to(DEV) * 0.3)


def test_config2_float32_hip_vs_pytorch_formulation_at_full_size():
    from snipper_amd import _lib
    hip, ref = _build(False), _build(True)
    _real_offsets(hip)
    ref.load_state_dict(hip.state_dict(), strict=True)
    assert hip.transformer.encoder.num_layers == 6 and hip.transformer.decoder.num_layers == 6
    g = torch.Generator().manual_seed(5)
    snippets = [torch.rand(12, 600, 800, generator=g).to(DEV) for _ in range(2)]            # two snippets of four frames
    res = []
    for m in (hip, ref):
        out, _ = m(snippets)
        assert out["pred_kpts2d"].shape == (2, 60, 4, 15, 3)
        loss = sum((out[k].float() ** 2).mean() for k in ("pred_logits", "pred_kpts2d", "pred_depth"))
        loss = loss + sum((h.float() ** 2).mean() for h in out["heatmaps"])
        pd = dict(m.named_parameters())
        grads = torch.autograd.grad(loss, [pd[n] for n in GRAD_NAMES])
        res.append(({k: v.detach() for k, v in out.items() if torch.is_tensor(v)}, float(loss.detach()), grads, _lib.last_variant()))
        del out, loss
        torch.cuda.empty_cache()
    (oh, lh, gh, vh), (orf, lr, gr, _) = res
    assert vh == "d48_owner", vh                    # last backward of the HIP model = the first encoder layer's owner-computes pair
    for k in ("pred_logits", "pred_kpts2d", "pred_depth"):
        torch.testing.assert_close(oh[k], orf[k], rtol=2e-3, atol=3e-4, msg=lambda m: f"{k}: {m}")
    assert abs(lh - lr) <= 1e-4 * abs(lr)
    errs = {}
    for n, a, b in zip(GRAD_NAMES, gh, gr):
        errs[n] = float((a - b).norm() / b.norm().clamp_min(1e-20))
    print("[config2 fp32 hip vs pytorch-deform] grad rel-L2", {".".join(k.split(".")[-4:]): f"{v:.2e}" for k, v in errs.items()})
    for n, v in errs.items():
        assert v <= 2e-3, (n, v)


STEPS = 4
NORM_NAMES = ["backbone.0.body.layer3.0.conv2.weight", "input_proj.1.0.weight",
              "transformer.encoder.layers.2.self_attn.output_proj.weight", "transformer.encoder.layers.5.linear2.weight",
              "transformer.decoder.layers.3.cross_attn.attention_weights.0.weight", "transformer.decoder.class_embed.0.weight",
              "joint_embed.0.7.layers.0.weight"]
ZERO_INIT = "transformer.decoder.layers.3.cross_attn.attention_weights.0.weight"     # its norm is made of the 4 Adam updates alone


def _train(arm):
    """arm "bench": exactly bench.py's default step; arm "ref": float32, use_pytorch_deform=1, per-parameter optimizer."""
    import bench
    from snipper_amd.criterion import build_criterion
    a = _bench_args(arm == "ref")
    model = _build(arm == "ref")
    amp = arm == "bench"
    named = list(model.named_parameters())
    flatp = own_opt = opt = None
    if arm == "bench":
        from snipper_amd.flat_params import FlatAdamW, FlatParameters
        g_main, g_backbone, g_slow = bench.optimizer_groups(named)
        flatp = FlatParameters([g_main, g_slow, g_backbone])
        own_opt = FlatAdamW(flatp, [1e-4, 1e-5, 1e-5], weight_decay=1e-4, reference_groups=bench.reference_param_groups(named))
    else:
        opt = bench.build_optimizer(named)
    criterion = build_criterion(bench.criterion_args(a)).to(DEV)
    batches = bench.make_batches(a, torch.device(DEV), 2, seed=1000)
    params = [p for p in model.parameters() if p.requires_grad]
    losses = []
    for i in range(STEPS):
        imgs, tgt = batches[i % 2]
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
            out, _ = model(list(imgs))
        ls, _ = criterion(out, tgt["targets"])
        loss = criterion.weighted_sum(ls)
        if flatp is not None:
            flatp.drop_param_grads()
        else:
            opt.zero_grad(set_to_none=True)
        loss.backward()
        if own_opt is not None:
            flatp.pack()
            own_opt.step(0.1)
        else:
            torch.nn.utils.clip_grad_norm_(params, 0.1)
            opt.step()
        losses.append(float(loss.detach()))
        del out, ls, loss
    pd = dict(model.named_parameters())
    norms = {n: float(pd[n].detach().float().norm()) for n in NORM_NAMES if n in pd}
    return losses, norms


def test_config2_bench_step_follows_the_float32_reference_trajectory():
    res = {}
    for arm in ("ref", "bench"):
        res[arm] = _train(arm)
        torch.cuda.empty_cache()
    for arm, (ls, _) in res.items():
        print(f"[config2 training parity] {arm:6s} " + " ".join(f"{v:.3f}" for v in ls))
    ref, got = res["ref"][0], res["bench"][0]
    assert all(v == v and abs(v) < 1e9 for v in ref + got)
    assert min(ref[1:]) < ref[0], "the float32 reference trajectory does not descend: the test would prove nothing"
    rel = [abs(v - r) / abs(r) for v, r in zip(got, ref)]
    print("[config2 training parity] loss rel diff per step", " ".join(f"{v:.2e}" for v in rel))
    for v in rel:
        assert v <= 2e-3, (got, ref)              # measured 1.1e-4 / 3.1e-5 / 7.5e-4 / 1.0e-3
    assert set(res["ref"][1]) == set(NORM_NAMES), sorted(set(NORM_NAMES) - set(res["ref"][1]))
    nrel = {n: abs(res["bench"][1][n] - res["ref"][1][n]) / res["ref"][1][n] for n in NORM_NAMES}
    print("[config2 training parity] weight-norm rel diff", {".".join(k.split(".")[-3:]): f"{v:.2e}" for k, v in nrel.items()})
    for n, v in nrel.items():                     # measured <= 5.3e-6; the zero-initialised one 1.4e-3
        assert v <= (8e-3 if n == ZERO_INIT else 2e-5), (n, v)      # (ZERO_INIT: 1.5e-3 in round 4, 4.5e-3 with the premixed decoder memory)
