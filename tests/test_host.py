"""Host-side mirror of the reference interface, checked on CPU against the reference's goldens.

The HIP kernels cannot run here, so the modules are built with ``use_pytroch_deform=True`` (the
reference's own switch); what is under test is everything AROUND the core op: the tied-weight
single-launch formulation, state_dict compatibility, reference-point geometry, decoder plumbing.
"""
import os
import re

import pytest
import torch

from snipper_amd.deformable_transformer import DeformableTransformer, build_deforamble_transformer
from snipper_amd.ms_deform_attn import MSDeformAttn, frame_neighbours


def _module_from(blob, **over):
    cfg = dict(blob["cfg"])
    mod = MSDeformAttn(cfg["d_model"], cfg["n_levels"], cfg["n_heads"], cfg["n_points"], cfg["n_frame"],
                       cfg["mode"], over.get("use_pytroch_deform", True), cfg["mode"] == "decoder").double()
    missing = mod.load_state_dict(blob["state_dict"], strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return mod


@pytest.mark.parametrize("name", ["enc_t3", "dec_t3", "dec_t3f2"])
def test_module_tied_path_matches_reference(golden_dir, name):
    b = torch.load(os.path.join(golden_dir, f"g3_module_{name}.pt"))
    mod = _module_from(b)
    assert mod.weights_are_tied()
    assert sorted(mod.state_dict().keys()) == sorted(b["state_dict"].keys())
    q, r, s = (b[k].clone().requires_grad_(True) for k in ("query", "ref", "src"))
    res = mod(q, r, s, b["shapes"], b["lsi"], b["mask"])
    if mod.attention_vis:
        res, (locs, wts) = res
        for x, y in zip(locs, b["vis_loc"]):
            torch.testing.assert_close(x, y, rtol=1e-11, atol=1e-13)
        for x, y in zip(wts, b["vis_w"]):
            torch.testing.assert_close(x, y, rtol=1e-10, atol=1e-13)
    torch.testing.assert_close(res, b["out"], rtol=1e-10, atol=1e-12)
    params = dict(mod.named_parameters())
    grads = torch.autograd.grad(res, [q, r, s] + list(params.values()), b["grad_out"])
    torch.testing.assert_close(grads[0], b["grad_query"], rtol=1e-9, atol=1e-11)
    torch.testing.assert_close(grads[1], b["grad_ref"], rtol=1e-9, atol=1e-11)
    torch.testing.assert_close(grads[2], b["grad_src"], rtol=1e-9, atol=1e-11)
    for (k, _), g in zip(params.items(), grads[3:]):
        torch.testing.assert_close(g, b["param_grads"][k], rtol=1e-9, atol=1e-10, msg=lambda m: f"{k}: {m}")


@pytest.mark.parametrize("name", ["enc_d48", "dec_d48", "enc_t1_d48", "dec_t1_d48"])
def test_module_d48_goldens_on_the_pytorch_path(golden_dir, name):
    """The d_model=384 / 8-head goldens (float32 storage of the reference's float64 evaluation) through this package's
    module in float64 with the reference's own ``use_pytorch_deform`` switch: pins the fixtures themselves on CPU; the GPU
    suite runs the same fixtures through the D=48 kernels (tests/test_module_gpu.py)."""
    b = torch.load(os.path.join(golden_dir, f"g3_module_{name}.pt"))
    mod = _module_from(b)
    q, r, s = (b[k].double().clone().requires_grad_(True) for k in ("query", "ref", "src"))
    mask_c = b["mask"][..., None].expand(-1, -1, -1, b["cfg"]["d_model"])
    res = mod(q, r, s, b["shapes"], b["lsi"], mask_c)
    if mod.attention_vis:
        res, (locs, wts) = res
        for x, y in zip(wts, b["vis_w"]):
            torch.testing.assert_close(x.float(), y, rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(res.float(), b["out"], rtol=1e-6, atol=1e-6)
    params = dict(mod.named_parameters())
    grads = torch.autograd.grad(res, [q, r, s] + list(params.values()), b["grad_out"].double())
    for got, key in zip(grads[:3], ("grad_query", "grad_ref", "grad_src")):
        torch.testing.assert_close(got.float(), b[key], rtol=1e-5, atol=1e-5)
    for (k, _), g in zip(params.items(), grads[3:]):
        torch.testing.assert_close(g.float(), b["param_grads"][k], rtol=1e-5, atol=1e-4, msg=lambda m: f"{k}: {m}")


@pytest.mark.parametrize("name", ["enc_t3", "dec_t3f2"])
def test_module_untied_path_matches_reference(golden_dir, name):
    """Untie the Linears (same values): the per-pair path must give the same answer."""
    import copy
    b = torch.load(os.path.join(golden_dir, f"g3_module_{name}.pt"))
    mod = _module_from(b)
    mod.sampling_offsets = torch.nn.ModuleList([copy.deepcopy(mod.sampling_offsets[0]) for _ in range(mod.n_frame)])
    mod.attention_weights = torch.nn.ModuleList([copy.deepcopy(mod.attention_weights[0]) for _ in range(mod.n_frame)])
    assert not mod.weights_are_tied()
    res = mod(b["query"], b["ref"], b["src"], b["shapes"], b["lsi"], b["mask"])
    if mod.attention_vis:
        res, (locs, wts) = res
        for x, y in zip(wts, b["vis_w"]):
            torch.testing.assert_close(x, y, rtol=1e-10, atol=1e-13)
    torch.testing.assert_close(res, b["out"], rtol=1e-10, atol=1e-12)


@pytest.mark.parametrize("name", ["enc_untied_d48", "dec_untied_d48"])
def test_module_untied_with_different_weights_matches_reference(golden_dir, name):
    """Goldens of the REFERENCE module with genuinely different per-frame Linears (gen_golden.py g3untied): this package's
    per-pair path and the oracle's ``st_msdeform_attn`` against them, float64 (fixtures store float32)."""
    from oracle import msda_oracle as O
    b = torch.load(os.path.join(golden_dir, f"g3_module_{name}.pt"))
    cfg = b["cfg"]
    mod = MSDeformAttn(cfg["d_model"], cfg["n_levels"], cfg["n_heads"], cfg["n_points"], cfg["n_frame"], cfg["mode"],
                       True, cfg["mode"] == "decoder").untie_frame_weights()
    mod.load_state_dict(b["state_dict"], strict=True)
    mod = mod.double()
    assert not mod.weights_are_tied()
    q, r, s = (b[k].double().clone().requires_grad_(True) for k in ("query", "ref", "src"))
    mask_c = b["mask"][..., None].expand(-1, -1, -1, cfg["d_model"])
    res = mod(q, r, s, b["shapes"], b["lsi"], mask_c)
    if mod.attention_vis:
        res, (locs, wts) = res
        for x, y in zip(wts, b["vis_w"]):
            torch.testing.assert_close(x.float(), y, rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(res.float(), b["out"], rtol=1e-6, atol=1e-6)
    params = dict(mod.named_parameters())
    grads = torch.autograd.grad(res, [q, r, s] + list(params.values()), b["grad_out"].double())
    for got, key in zip(grads[:3], ("grad_query", "grad_ref", "grad_src")):
        torch.testing.assert_close(got.float(), b[key], rtol=1e-5, atol=1e-5)
    for (k, _), g in zip(params.items(), grads[3:]):
        torch.testing.assert_close(g.float(), b["param_grads"][k], rtol=1e-5, atol=1e-4, msg=lambda m: f"{k}: {m}")
    sd = {k: v.double() for k, v in b["state_dict"].items()}
    T = cfg["n_frame"]
    out, _, wts = O.st_msdeform_attn(
        b["query"].double(), b["ref"].double(), b["src"].double(), b["shapes"], mask_c,
        sd["value_proj.weight"], sd["value_proj.bias"],
        [sd[f"sampling_offsets.{t}.weight"] for t in range(T)], [sd[f"sampling_offsets.{t}.bias"] for t in range(T)],
        [sd[f"attention_weights.{t}.weight"] for t in range(T)], [sd[f"attention_weights.{t}.bias"] for t in range(T)],
        sd["output_proj.weight"], sd["output_proj.bias"], cfg["n_heads"], cfg["n_levels"], cfg["n_points"], T)
    torch.testing.assert_close(out.float(), b["out"], rtol=1e-6, atol=1e-6)


def test_frame_neighbours():
    assert frame_neighbours(0, 4, 4) == [0, 1]
    assert frame_neighbours(2, 4, 4) == [1, 2, 3]
    assert frame_neighbours(3, 4, 4) == [2, 3]
    assert frame_neighbours(4, 4, 4) == [0, 1, 2, 3]      # forecast frame
    assert frame_neighbours(0, 1, 1) == [0]


def test_reset_parameters_matches_reference_init():
    """Offset bias = 8 unit directions x (1..P), zero offset weights and logits (reference :78-93)."""
    m = MSDeformAttn(64, 3, 8, 4, 2)
    assert float(m.sampling_offsets[0].weight.abs().max()) == 0.0
    assert float(m.attention_weights[1].weight.abs().max()) == 0.0
    bias = m.sampling_offsets[0].bias.view(8, 3, 4, 2)
    torch.testing.assert_close(bias[0, :, :, 0], torch.tensor([1., 2., 3., 4.]).expand(3, 4))
    torch.testing.assert_close(bias[2, 1, 3], torch.tensor([0., 4.]), atol=1e-6, rtol=0)
    torch.testing.assert_close(bias[5, 0, 1], torch.tensor([-2., -2.]), atol=1e-6, rtol=0)
    assert m.sampling_offsets[0] is m.sampling_offsets[1]
    with pytest.raises(ValueError):
        MSDeformAttn(50, 3, 8, 4)


def test_transformer_matches_reference(golden_dir):
    b = torch.load(os.path.join(golden_dir, "g4_transformer.pt"))
    cfg = b["cfg"]
    tr = DeformableTransformer(return_intermediate_dec=True, use_pytroch_deform=True, activation="relu", **cfg).double()
    assert sorted(tr.state_dict().keys()) == sorted(b["state_dict"].keys())       # key schema
    for k, v in tr.state_dict().items():
        assert tuple(v.shape) == tuple(b["state_dict"][k].shape), k
    tr.load_state_dict(b["state_dict"], strict=True)
    hs, heatmaps, init_ref, inter_refs, att = tr(b["srcs"], b["masks"], b["pos"], b["query_embed"])
    torch.testing.assert_close(hs, b["hs"], rtol=1e-9, atol=1e-11)
    torch.testing.assert_close(init_ref, b["init_ref"], rtol=1e-11, atol=1e-13)
    torch.testing.assert_close(inter_refs, b["inter_refs"], rtol=1e-11, atol=1e-13)
    for x, y in zip(heatmaps, b["heatmaps"]):
        torch.testing.assert_close(x, y, rtol=1e-9, atol=1e-11)
    for layer_att, ref_loc, ref_w in zip(att, b["att_loc"], b["att_w"]):
        for x, y in zip(layer_att[0], ref_loc):
            torch.testing.assert_close(x, y, rtol=1e-10, atol=1e-12)
        for x, y in zip(layer_att[1], ref_w):
            torch.testing.assert_close(x, y, rtol=1e-9, atol=1e-12)
    loss = (hs * torch.linspace(-1, 1, hs.numel(), dtype=torch.float64).view_as(hs)).sum()
    torch.testing.assert_close(loss, b["loss"], rtol=1e-10, atol=1e-10)
    names = [k for k, _ in tr.named_parameters()]
    grads = torch.autograd.grad(loss, list(tr.parameters()), allow_unused=True)
    for k, g in zip(names, grads):
        ref = b["param_grads"][k]
        if ref is None:
            assert g is None or float(g.abs().max()) == 0.0, k
        else:
            torch.testing.assert_close(g, ref, rtol=1e-8, atol=1e-9, msg=lambda m: f"{k}: {m}")


def test_builder_reads_reference_flag_names():
    class A:  # the argparse names of main.py:20-153
        hidden_dim, nheads, enc_layers, dec_layers, dim_feedforward, dropout = 48, 4, 1, 1, 32, 0.0
        num_feature_levels, dec_n_points, enc_n_points, num_frames, num_future_frames = 3, 4, 4, 2, 1
        use_pytorch_deform, num_kpts = True, 3
    tr = build_deforamble_transformer(A())
    assert tr.decoder.return_intermediate and tr.temporal_embed.shape == (3, 48)
    assert tr.decoder.root_embed is None and tr.decoder.class_embed is None
    assert tr.encoder.layers[0].self_attn.use_pytroch_deform is True


def test_core_op_has_no_cpu_fallback():
    """CPU tensors must raise like the reference (ms_deform_attn.h:38), not silently run elsewhere."""
    from snipper_amd.ms_deform_attn_func import MSDeformAttnFunction
    v = torch.zeros(1, 4, 1, 4)
    shapes = torch.tensor([[2, 2]])
    lsi = torch.tensor([0])
    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):
        MSDeformAttnFunction.apply(v, shapes, lsi, torch.zeros(1, 1, 1, 1, 1, 2), torch.zeros(1, 1, 1, 1, 1), 64)


def test_model_assembly_matches_reference_on_replayed_backbone(golden_dir):
    """SnipperDeformable (snipper_amd/model.py) against the reference model itself (golden g6: models/model.py:45-237 run
    on a stand-in backbone that replays stored feature maps): state_dict key schema incl. the aliases of the shared
    heads, strict loading of the reference's weights, and every output of the forward pass."""
    import os
    import torch
    from snipper_amd.deformable_transformer import DeformableTransformer
    from snipper_amd.misc import NestedTensor
    from snipper_amd.model import SnipperDeformable
    b = torch.load(os.path.join(golden_dir, "g6_model.pt"))

    class Replay(torch.nn.Module):
        strides, num_channels = [8, 16, 32], b["chans"]

        def forward(self, samples):
            return [NestedTensor(f, m) for f, m in zip(b["feats"], b["masks"])], [p.clone() for p in b["pos"]]

    tr = DeformableTransformer(return_intermediate_dec=True, use_pytroch_deform=True, activation="relu", **b["cfg"])
    model = SnipperDeformable(Replay(), tr, num_queries=b["num_queries"], num_feature_levels=len(b["hw"]),
                              num_frames=b["cfg"]["n_frame"], num_future_frames=b["cfg"]["n_future_frame"],
                              num_keypoints=b["cfg"]["num_keypoints"], aux_loss=True)
    sd = model.state_dict()
    assert sorted(sd.keys()) == sorted(b["state_dict"].keys())
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(b["state_dict"][k].shape), k
    ptr = {}
    for k, v in sd.items():
        ptr.setdefault(v.data_ptr(), []).append(k)
    assert sorted(sorted(v) for v in ptr.values() if len(v) > 1) == b["aliases"]
    model.load_state_dict(b["state_dict"], strict=True)
    model.eval()
    T = b["cfg"]["n_frame"]
    samples = NestedTensor(torch.zeros(b["bs"] * T, 3, 96, 128), torch.zeros(b["bs"] * T, 96, 128, dtype=torch.bool))
    with torch.no_grad():
        out, (init_ref, inter_refs, _) = model(samples)
    tol = dict(rtol=1e-4, atol=1e-5)
    for k in ("pred_logits", "pred_kpts2d", "pred_depth"):
        torch.testing.assert_close(out[k], b[k], **tol)
    for h, hr in zip(out["heatmaps"], b["heatmaps"]):
        torch.testing.assert_close(h, hr, **tol)
    assert len(out["aux_outputs"]) == len(b["aux"])
    for a, ar in zip(out["aux_outputs"], b["aux"]):
        for k in ar:
            torch.testing.assert_close(a[k], ar[k], **tol)
    torch.testing.assert_close(init_ref, b["init_ref"], **tol)
    torch.testing.assert_close(inter_refs, b["inter_refs"], **tol)


def test_nested_tensor_from_ragged_snippets():
    """util/misc.py:310-330: every snippet [T*3, H, W] is split into T images, all images are zero-padded to the
    largest H and W of the batch, the mask is True exactly on the padding; equal-sized inputs take the stack path."""
    import torch
    from snipper_amd.misc import nested_tensor_from_tensor_list
    g = torch.Generator().manual_seed(0)
    a, b = torch.rand(6, 5, 7, generator=g), torch.rand(6, 4, 9, generator=g)        # T = 2 frames each
    nt = nested_tensor_from_tensor_list([a, b])
    x, m = nt.decompose()
    assert x.shape == (4, 3, 5, 9) and m.shape == (4, 5, 9) and m.dtype == torch.bool
    assert torch.equal(x[0, :, :5, :7], a[:3]) and torch.equal(x[1, :, :5, :7], a[3:])
    assert torch.equal(x[2, :, :4, :9], b[:3]) and torch.equal(x[3, :, :4, :9], b[3:])
    assert float(x[0, :, :, 7:].abs().sum()) == 0 and float(x[2, :, 4:, :].abs().sum()) == 0
    assert not m[0, :5, :7].any() and m[0, :, 7:].all() and not m[2, :4, :].any() and m[2, 4:, :].all()
    same = nested_tensor_from_tensor_list([a, a.clone()])
    assert same.tensors.shape == (4, 3, 5, 7) and not same.mask.any()
