"""GPU tests of the host-side mirror running on the HIP kernels (``-m gpu``)."""
import os

import pytest
import torch

from snipper_amd import _lib
from snipper_amd.deformable_transformer import DeformableTransformer
from snipper_amd.ms_deform_attn import MSDeformAttn

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _to(x, dev=DEV, dtype=None):
    if isinstance(x, torch.Tensor):
        x = x.to(dev)
        return x.to(dtype) if (dtype is not None and x.is_floating_point()) else x
    if isinstance(x, (list, tuple)):
        return type(x)(_to(y, dev, dtype) for y in x)
    if isinstance(x, dict):
        return {k: _to(v, dev, dtype) for k, v in x.items()}
    return x


@pytest.mark.parametrize("name", ["enc_t3", "dec_t3", "dec_t3f2"])
@pytest.mark.parametrize("dtype,rtol,atol", [(torch.float64, 1e-9, 1e-11), (torch.float32, 2e-4, 2e-5)],
                         ids=["f64", "f32"])
def test_module_on_hip_matches_reference(golden_dir, name, dtype, rtol, atol):
    b = torch.load(os.path.join(golden_dir, f"g3_module_{name}.pt"))
    cfg = b["cfg"]
    mod = MSDeformAttn(cfg["d_model"], cfg["n_levels"], cfg["n_heads"], cfg["n_points"], cfg["n_frame"],
                       cfg["mode"], False, cfg["mode"] == "decoder")
    mod.load_state_dict(b["state_dict"], strict=True)
    mod = mod.to(DEV).to(dtype)
    q, r, s = (_to(b[k], dtype=dtype).clone().requires_grad_(True) for k in ("query", "ref", "src"))
    res = mod(q, r, s, _to(b["shapes"]), _to(b["lsi"]), _to(b["mask"]))
    assert _lib.last_variant() == "generic"          # D=12 here
    if mod.attention_vis:
        res, (locs, wts) = res
        for x, y in zip(wts, b["vis_w"]):
            torch.testing.assert_close(x.cpu().double(), y, rtol=rtol, atol=atol)
    torch.testing.assert_close(res.cpu().double(), b["out"], rtol=rtol, atol=atol)
    params = dict(mod.named_parameters())
    grads = torch.autograd.grad(res, [q, r, s] + list(params.values()), _to(b["grad_out"], dtype=dtype))
    scale = lambda ref: max(float(ref.abs().max()), 1.0)
    for got, key in zip(grads[:3], ("grad_query", "grad_ref", "grad_src")):
        torch.testing.assert_close(got.cpu().double() / scale(b[key]), b[key] / scale(b[key]), rtol=rtol * 5, atol=atol * 5)
    for (k, _), g in zip(params.items(), grads[3:]):
        ref = b["param_grads"][k]
        torch.testing.assert_close(g.cpu().double() / scale(ref), ref / scale(ref), rtol=rtol * 5, atol=atol * 5,
                                   msg=lambda m: f"{k}: {m}")


@pytest.mark.parametrize("name", ["enc_d48", "dec_d48", "enc_t1_d48", "dec_t1_d48"])
@pytest.mark.parametrize("mask_kind", ["expanded", "3d", "none"])
def test_module_d48_on_hip_matches_reference(golden_dir, name, mask_kind):
    """The reference module at Snipper's head geometry (d_model 384 / 8 heads -> D = 48) against the kernels the
    training step runs: tied single-launch path + fused prologue / temporal mean + the D=48 forward and, for the encoder
    (Lq == S), the owner-computes backward."""
    b = torch.load(os.path.join(golden_dir, f"g3_module_{name}.pt"))
    cfg = b["cfg"]
    enc = cfg["mode"] == "encoder"
    mod = MSDeformAttn(cfg["d_model"], cfg["n_levels"], cfg["n_heads"], cfg["n_points"], cfg["n_frame"],
                       cfg["mode"], False, not enc)
    mod.load_state_dict(b["state_dict"], strict=True)
    mod = mod.to(DEV)
    q, r, s = (_to(b[k]).clone().requires_grad_(True) for k in ("query", "ref", "src"))
    shapes = _to(b["shapes"])
    shapes._snipper_host = [tuple(x) for x in b["shapes"].tolist()]      # what our transformer attaches
    m3 = _to(b["mask"])
    mask = {"expanded": m3[..., None].expand(-1, -1, -1, cfg["d_model"]), "3d": m3, "none": None}[mask_kind]
    res = mod(q, r, s, shapes, _to(b["lsi"]), mask)
    assert _lib.last_variant() == "d48_lp12", _lib.last_variant()
    if mask_kind == "none":          # the golden was made with the mask: only check that the path runs and differs
        res0 = res[0] if mod.attention_vis else res
        assert torch.isfinite(res0).all()
        return
    if mod.attention_vis:
        res, (locs, wts) = res
        for x, y in zip(wts, b["vis_w"]):
            torch.testing.assert_close(x.cpu(), y, rtol=2e-4, atol=2e-6)
    torch.testing.assert_close(res.cpu(), b["out"], rtol=2e-4, atol=2e-4)
    params = dict(mod.named_parameters())
    grads = torch.autograd.grad(res, [q, r, s] + list(params.values()), _to(b["grad_out"]))
    assert _lib.last_variant() == ("d48_owner" if enc else "d48_lp12"), _lib.last_variant()
    scale = lambda ref: max(float(ref.abs().max()), 1.0)
    for got, key in zip(grads[:3], ("grad_query", "grad_ref", "grad_src")):
        torch.testing.assert_close(got.cpu() / scale(b[key]), b[key] / scale(b[key]), rtol=1e-3, atol=1e-4,
                                   msg=lambda m: f"{key}: {m}")
    for (k, _), g in zip(params.items(), grads[3:]):
        ref = b["param_grads"][k]
        torch.testing.assert_close(g.cpu() / scale(ref), ref / scale(ref), rtol=1e-3, atol=1e-4, msg=lambda m: f"{k}: {m}")


@pytest.mark.parametrize("name", ["enc_untied_d48", "dec_untied_d48"])
def test_module_untied_per_pair_path_on_hip_matches_reference(golden_dir, name):
    """Genuinely different per-frame Linears (goldens made by the reference module itself): the general per-pair path on
    the HIP core op -- one launch per (t1, t2) on the non-contiguous ``value[:, t2]`` slices, joint softmax over
    L*P*|t2| -- outputs, input and parameter gradients, vis lists."""
    b = torch.load(os.path.join(golden_dir, f"g3_module_{name}.pt"))
    cfg = b["cfg"]
    enc = cfg["mode"] == "encoder"
    mod = MSDeformAttn(cfg["d_model"], cfg["n_levels"], cfg["n_heads"], cfg["n_points"], cfg["n_frame"],
                       cfg["mode"], False, not enc).untie_frame_weights()
    mod.load_state_dict(b["state_dict"], strict=True)
    mod = mod.to(DEV)
    assert not mod.weights_are_tied()
    q, r, s = (_to(b[k]).clone().requires_grad_(True) for k in ("query", "ref", "src"))
    m3 = _to(b["mask"])
    res = mod(q, r, s, _to(b["shapes"]), _to(b["lsi"]), m3[..., None].expand(-1, -1, -1, cfg["d_model"]))
    assert _lib.last_variant() == "d48_lp12", _lib.last_variant()
    if mod.attention_vis:
        res, (locs, wts) = res
        for x, y in zip(wts, b["vis_w"]):
            torch.testing.assert_close(x.cpu(), y, rtol=2e-4, atol=2e-6)
    torch.testing.assert_close(res.cpu(), b["out"], rtol=2e-4, atol=2e-4)
    params = dict(mod.named_parameters())
    assert len(params) == 4 + 4 * cfg["n_frame"]
    grads = torch.autograd.grad(res, [q, r, s] + list(params.values()), _to(b["grad_out"]))
    scale = lambda ref: max(float(ref.abs().max()), 1.0)
    for got, key in zip(grads[:3], ("grad_query", "grad_ref", "grad_src")):
        torch.testing.assert_close(got.cpu() / scale(b[key]), b[key] / scale(b[key]), rtol=1e-3, atol=1e-4,
                                   msg=lambda m: f"{key}: {m}")
    for (k, _), g in zip(params.items(), grads[3:]):
        ref = b["param_grads"][k]
        torch.testing.assert_close(g.cpu() / scale(ref), ref / scale(ref), rtol=1e-3, atol=1e-4, msg=lambda m: f"{k}: {m}")


@pytest.mark.parametrize("dtype,rtol,atol", [(torch.float64, 1e-8, 1e-10), (torch.float32, 1e-3, 1e-4)],
                         ids=["f64", "f32"])
def test_transformer_on_hip_matches_reference(golden_dir, dtype, rtol, atol):
    b = torch.load(os.path.join(golden_dir, "g4_transformer.pt"))
    tr = DeformableTransformer(return_intermediate_dec=True, use_pytroch_deform=False, activation="relu", **b["cfg"])
    tr.load_state_dict(b["state_dict"], strict=True)
    tr = tr.to(DEV).to(dtype)
    hs, heatmaps, init_ref, inter_refs, att = tr(_to(b["srcs"], dtype=dtype), _to(b["masks"]),
                                                  _to(b["pos"], dtype=dtype), _to(b["query_embed"], dtype=dtype))
    torch.testing.assert_close(hs.cpu().double(), b["hs"], rtol=rtol, atol=atol)
    torch.testing.assert_close(inter_refs.cpu().double(), b["inter_refs"], rtol=rtol, atol=atol)
    loss = (hs.double() * torch.linspace(-1, 1, hs.numel(), dtype=torch.float64, device=DEV).view_as(hs)).sum()
    names = [k for k, _ in tr.named_parameters()]
    grads = torch.autograd.grad(loss, list(tr.parameters()), allow_unused=True)
    for k, g in zip(names, grads):
        ref = b["param_grads"][k]
        if ref is None:
            continue
        s = max(float(ref.abs().max()), 1.0)
        torch.testing.assert_close(g.cpu().double() / s, ref / s, rtol=rtol * 10, atol=atol * 10, msg=lambda m: f"{k}: {m}")


def test_snipper_geometry_forward_backward_runs_d48_kernels():
    """hidden 384 / 8 heads (D=48), 3 levels, T=2 at a reduced map size: the tuned kernels must be the
    ones that run, and the module must agree with its own pure-PyTorch formulation."""
    torch.manual_seed(0)
    shapes = [(19, 25), (10, 13), (5, 7)]
    S = sum(h * w for h, w in shapes)
    hip = MSDeformAttn(384, 3, 8, 4, 2, 'encoder', False).to(DEV)
    ref = MSDeformAttn(384, 3, 8, 4, 2, 'encoder', True).to(DEV)
    with torch.no_grad():
        for p in hip.parameters():
            if float(p.abs().max()) == 0:
                p.normal_(0, 0.05)
    ref.load_state_dict(hip.state_dict())
    sh = torch.tensor(shapes, device=DEV)
    lsi = torch.cat((sh.new_zeros(1), sh.prod(1).cumsum(0)[:-1]))
    src = torch.randn(2, 2, S, 384, device=DEV, requires_grad=True)
    refp = torch.rand(2, 2, S, 3, 2, device=DEV)
    go = torch.randn(2, 2, S, 384, device=DEV)
    out = hip(src, refp, src, sh, lsi, None)
    assert _lib.last_variant() == "d48_lp12"
    g_hip = torch.autograd.grad(out, [src] + list(hip.parameters()), go)
    assert _lib.last_variant() == "d48_lp12"
    out_ref = ref(src, refp, src, sh, lsi, None)
    g_ref = torch.autograd.grad(out_ref, [src] + list(ref.parameters()), go)
    torch.testing.assert_close(out, out_ref, rtol=1e-3, atol=1e-4)
    for a, c in zip(g_hip, g_ref):
        s = max(float(c.abs().max()), 1.0)
        torch.testing.assert_close(a / s, c / s, rtol=1e-3, atol=2e-4)


def test_transformer_bf16_fused_residual_path_matches_plain_path(golden_dir):
    """Under bf16 autocast the encoder carries bf16 twins of its float32 residual stream through the fused
    residual + LayerNorm kernels; the result must agree with the plain per-op formulation (same autocast, same
    weights, dropout off) to bf16 accuracy, outputs and parameter gradients, and with the float64 golden output."""
    from snipper_amd.deformable_transformer import DeformableTransformerEncoderLayer as EncLayer
    b = torch.load(os.path.join(golden_dir, "g4_transformer.pt"))
    tr = DeformableTransformer(return_intermediate_dec=True, use_pytroch_deform=False, activation="relu", **b["cfg"])
    tr.load_state_dict(b["state_dict"], strict=True)
    tr = tr.to(DEV).float().eval()
    args = (_to(b["srcs"], dtype=torch.float32), _to(b["masks"]), _to(b["pos"], dtype=torch.float32),
            _to(b["query_embed"], dtype=torch.float32))
    res = {}
    for fused in (True, False):
        EncLayer.fused_residual = fused
        try:
            with torch.autocast("cuda", dtype=torch.bfloat16):
                hs = tr(*args)[0]
            w = torch.linspace(-1, 1, hs.numel(), device=DEV).view_as(hs)
            grads = torch.autograd.grad((hs.float() * w).sum(), list(tr.parameters()), allow_unused=True)
        finally:
            EncLayer.fused_residual = True
        res[fused] = (hs.float(), grads)
    rel = lambda a, c: ((a.double() - c.double()).norm() / c.double().norm().clamp_min(1e-20)).item()
    gold = b["hs"].to(DEV)
    e_fused, e_plain = rel(res[True][0], gold), rel(res[False][0], gold)
    assert e_fused < 3e-2 and e_fused < 2.0 * e_plain + 1e-3, (e_fused, e_plain)
    names = [k for k, _ in tr.named_parameters()]
    for k, gf, gp in zip(names, res[True][1], res[False][1]):
        ref = b["param_grads"][k]
        if ref is None or gf is None:
            continue
        ref = ref.to(DEV)
        ef, ep = rel(gf, ref), rel(gp, ref)
        assert ef < 2.0 * ep + 3e-2, (k, ef, ep)


def test_model_token_row_path_matches_channel_first_path():
    """SnipperDeformable under bf16 autocast: projections + position encoding produced as token rows (the fused
    path) against the reference's channel-first route through the same modules -- outputs and parameter gradients."""
    import importlib.util
    from types import SimpleNamespace
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    a = SimpleNamespace(hidden_dim=192, enc_layers=2, dec_layers=2, frames=2, future_frames=0, use_pytorch_deform=0,
                        batch=1, height=96, width=128)
    from snipper_amd.model import build_model, SnipperDeformable
    torch.manual_seed(0)
    model = build_model(b.model_args(a)).to(DEV).to(memory_format=torch.channels_last).eval()
    imgs, _ = b.make_batches(a, DEV, 1, seed=3)[0]
    res = {}
    for fast in (True, False):
        SnipperDeformable.token_rows = fast
        try:
            with torch.autocast("cuda", dtype=torch.bfloat16):
                out, _ = model(list(imgs))
            k = out["all_layers"]["pred_kpts"].float()
            w = torch.linspace(-1, 1, k.numel(), device=DEV).view_as(k)
            params = [p for p in model.parameters() if p.requires_grad]
            grads = torch.autograd.grad((k * w).sum() + out["heatmaps"][0].float().sum(), params, allow_unused=True)
        finally:
            SnipperDeformable.token_rows = True
        res[fast] = (k, out["pred_logits"].float(), grads)
    rel = lambda x, y: ((x.double() - y.double()).norm() / y.double().norm().clamp_min(1e-20)).item()
    assert rel(res[True][0], res[False][0]) < 3e-2 and rel(res[True][1], res[False][1]) < 3e-2
    names = [n for n, p in model.named_parameters() if p.requires_grad]
    bad = []
    for n, gf, gs in zip(names, res[True][2], res[False][2]):
        if gf is None or gs is None:
            assert gf is None and gs is None, n
            continue
        if rel(gf, gs) > 0.15:
            bad.append((n, rel(gf, gs)))
    assert len(bad) <= len(names) // 20, bad[:10]


def test_decoder_self_attention_matches_nn_multihead_attention():
    """The batch-first evaluation of the decoder's self-attention against nn.MultiheadAttention itself (dropout off):
    output and all gradients, float32."""
    from snipper_amd.deformable_transformer import _self_attention
    torch.manual_seed(2)
    mha = torch.nn.MultiheadAttention(384, 8, dropout=0.1).to(DEV).eval()
    x_qk = torch.randn(2, 240, 384, device=DEV, requires_grad=True)
    x_v = torch.randn(2, 240, 384, device=DEV, requires_grad=True)
    gy = torch.randn(2, 240, 384, device=DEV)
    params = [mha.in_proj_weight, mha.in_proj_bias, mha.out_proj.weight, mha.out_proj.bias]
    y = _self_attention(mha, x_qk, x_v)
    g1 = torch.autograd.grad(y, [x_qk, x_v] + params, gy)
    yr = mha(x_qk.transpose(0, 1), x_qk.transpose(0, 1), x_v.transpose(0, 1), need_weights=False)[0].transpose(0, 1)
    g2 = torch.autograd.grad(yr, [x_qk, x_v] + params, gy)
    torch.testing.assert_close(y, yr, rtol=1e-4, atol=1e-5)
    for a, b in zip(g1, g2):
        torch.testing.assert_close(a, b, rtol=1e-3, atol=1e-4)


def test_model_on_hip_matches_reference_model(golden_dir):
    """Golden g6 (the reference SnipperDeformable on a replayed backbone) through the HIP kernels in float32."""
    from snipper_amd.misc import NestedTensor
    from snipper_amd.model import SnipperDeformable
    b = torch.load(os.path.join(golden_dir, "g6_model.pt"))

    class Replay(torch.nn.Module):
        strides, num_channels = [8, 16, 32], b["chans"]

        def forward(self, samples):
            return ([NestedTensor(f.to(DEV), m.to(DEV)) for f, m in zip(b["feats"], b["masks"])],
                    [p.to(DEV) for p in b["pos"]])

    tr = DeformableTransformer(return_intermediate_dec=True, use_pytroch_deform=False, activation="relu", **b["cfg"])
    model = SnipperDeformable(Replay(), tr, num_queries=b["num_queries"], num_feature_levels=len(b["hw"]),
                              num_frames=b["cfg"]["n_frame"], num_future_frames=b["cfg"]["n_future_frame"],
                              num_keypoints=b["cfg"]["num_keypoints"], aux_loss=True)
    model.load_state_dict(b["state_dict"], strict=True)
    model = model.to(DEV).eval()
    T = b["cfg"]["n_frame"]
    samples = NestedTensor(torch.zeros(b["bs"] * T, 3, 96, 128, device=DEV),
                           torch.zeros(b["bs"] * T, 96, 128, dtype=torch.bool, device=DEV))
    with torch.no_grad():
        out, (init_ref, inter_refs, _) = model(samples)
    tol = dict(rtol=1e-3, atol=1e-4)
    for k in ("pred_logits", "pred_kpts2d", "pred_depth"):
        torch.testing.assert_close(out[k].cpu(), b[k], **tol)
    for a, ar in zip(out["aux_outputs"], b["aux"]):
        for k in ar:
            torch.testing.assert_close(a[k].cpu(), ar[k], **tol)
    torch.testing.assert_close(inter_refs.cpu(), b["inter_refs"], **tol)


def test_model_token_row_path_with_padded_inputs():
    """Two snippets of different sizes (zero padding, masks True on it, valid ratios != 1): the token-row path against
    the channel-first route, bf16 autocast, eval mode."""
    import importlib.util
    from types import SimpleNamespace
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    a = SimpleNamespace(hidden_dim=192, enc_layers=1, dec_layers=1, frames=2, future_frames=0, use_pytorch_deform=0,
                        batch=2, height=96, width=128)
    from snipper_amd.model import build_model, SnipperDeformable
    torch.manual_seed(1)
    model = build_model(b.model_args(a)).to(DEV).to(memory_format=torch.channels_last).eval()
    g = torch.Generator().manual_seed(4)
    imgs = [torch.rand(6, 96, 128, generator=g).to(DEV), torch.rand(6, 80, 104, generator=g).to(DEV)]
    res = {}
    for fast in (True, False):
        SnipperDeformable.token_rows = fast
        try:
            with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
                out, _ = model(imgs)
        finally:
            SnipperDeformable.token_rows = True
        res[fast] = (out["pred_kpts2d"].float(), out["pred_logits"].float())
    rel = lambda x, y: ((x.double() - y.double()).norm() / y.double().norm().clamp_min(1e-20)).item()
    assert rel(res[True][0], res[False][0]) < 3e-2 and rel(res[True][1], res[False][1]) < 3e-2


def test_no_padding_constants_leave_the_outputs_unchanged():
    """Equal-sized snippets: the mask says "no padding" on the host and the position encoding, valid ratios, resized
    masks and reference grid come from the constant cache -- same outputs as with an unmarked copy of the mask."""
    import importlib.util
    from types import SimpleNamespace
    from snipper_amd.misc import NestedTensor, is_no_padding, nested_tensor_from_tensor_list
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    a = SimpleNamespace(hidden_dim=192, enc_layers=1, dec_layers=1, frames=2, future_frames=0, use_pytorch_deform=0,
                        batch=2, height=96, width=128)
    from snipper_amd.model import build_model
    torch.manual_seed(1)
    model = build_model(b.model_args(a)).to(DEV).to(memory_format=torch.channels_last).eval()
    g = torch.Generator().manual_seed(5)
    imgs = [torch.rand(6, 96, 128, generator=g).to(DEV) for _ in range(2)]
    marked = nested_tensor_from_tensor_list(imgs)
    assert is_no_padding(marked.mask)
    plain = NestedTensor(marked.tensors, marked.mask.clone())
    assert not is_no_padding(plain.mask)
    outs = []
    for samples in (marked, marked, plain):            # twice marked: the second call reads the cache
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            out, _ = model(samples)
        outs.append(out)
    for k in ("pred_logits", "pred_kpts2d", "pred_depth"):
        d01 = (outs[0][k].float() - outs[1][k].float()).abs().max().item()
        d02 = (outs[0][k].float() - outs[2][k].float()).abs().max().item()
        print(k, "marked/marked", d01, "marked/plain", d02)
        # not bit-equal even between two identical calls: the decoder's float32 GEMMs (hipBLASLt) and the bf16 encoder
        # do not fix the summation order; the shortcut must stay inside that run-to-run noise
        assert d02 <= max(4 * d01, 2e-3 * outs[0][k].float().abs().max().item())


def test_refine_reference_kernel_equals_the_reference_chain():
    """sigmoid(head(out)[..., :2] + inverse_sigmoid(ref)) and its product with the valid ratios from one launch
    (deformable_transformer.py:319-333 of the reference) against the PyTorch chain, incl. ref = 0, 1 and below eps."""
    from snipper_amd.deformable_transformer import _refine_reference, inverse_sigmoid
    g = torch.Generator().manual_seed(3)
    head = torch.nn.Sequential(torch.nn.Linear(32, 16), torch.nn.ReLU(), torch.nn.Linear(16, 4)).to(DEV)
    out = torch.randn(2, 3, 5, 32, generator=g).to(DEV)
    ref = torch.rand(2, 3, 5, 2, generator=g)
    ref[0, 0, 0] = torch.tensor([0.0, 1.0])
    ref[0, 0, 1] = torch.tensor([1e-7, 1 - 1e-7])
    ref = ref.to(DEV)
    vr = (0.5 + 0.5 * torch.rand(2, 4, 2, generator=g)).to(DEV)
    got = _refine_reference(head, out, ref, vr)
    assert got is not None
    new_ref, ref_in = got
    want = (head(out)[..., 0:2] + inverse_sigmoid(ref)).sigmoid().detach()
    torch.testing.assert_close(new_ref, want, rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(ref_in, want[:, :, :, None, :] * vr[:, None, None, :, :], rtol=1e-6, atol=1e-7)
    assert not new_ref.requires_grad and not ref_in.requires_grad


def test_inference_style_padded_snippets_at_540x960():
    """the README's JTA / Panoptic recipe feeds 540x960 frames (levels 68x120, 34x60, 17x30,
    S = 10 710) and inference.py-style snippets arrive as a list of differently sized images that
    nested_tensor_from_tensor_list pads (valid_ratios != 1, real padding masks).  The whole model under
    torch.inference_mode() on the HIP kernels must agree with its own ``use_pytorch_deform=1`` evaluation (float32)."""
    from types import SimpleNamespace
    from snipper_amd.model import build_model
    T = 2
    args = dict(hidden_dim=384, nheads=8, enc_layers=2, dec_layers=2, dim_feedforward=512, dropout=0.0,
                num_feature_levels=3, dec_n_points=4, enc_n_points=4, num_frames=T, num_future_frames=1, num_kpts=15,
                position_embedding="sine", backbone="resnet50", lr_backbone=1e-5, masks=False, dilation=False,
                num_queries=20, aux_loss=True)
    torch.manual_seed(3)
    hip = build_model(SimpleNamespace(use_pytorch_deform=False, **args)).to(DEV).eval()
    ref = build_model(SimpleNamespace(use_pytorch_deform=True, **args)).to(DEV).eval()
    with torch.no_grad():
        for n, p in hip.named_parameters():          # real offsets / logits instead of the zero initialisation
            if "sampling_offsets" in n and n.endswith("weight"):
                p.normal_(0, 0.02)
            elif "attention_weights" in n:
                p.normal_(0, 0.3)
    ref.load_state_dict(hip.state_dict(), strict=True)
    g = torch.Generator().manual_seed(5)
    snippets = [torch.rand(T * 3, 540, 960, generator=g).to(DEV), torch.rand(T * 3, 500, 900, generator=g).to(DEV)]
    with torch.inference_mode():
        out_h, (init_h, inter_h, _) = hip(snippets)
        variant = _lib.last_variant()
        out_r, (init_r, inter_r, _) = ref(snippets)
    assert variant.startswith("d48"), variant                    # D = 48 kernels (the decoder's call is the last one)
    assert out_h["pred_kpts2d"].shape == (2, 20, T + 1, 15, 3)
    assert [tuple(h.shape[2:4]) for h in out_h["heatmaps"]] == [(68, 120), (34, 60), (17, 30)]
    for k in ("pred_logits", "pred_kpts2d", "pred_depth"):
        torch.testing.assert_close(out_h[k], out_r[k], rtol=2e-3, atol=2e-4, msg=lambda m: f"{k}: {m}")
    torch.testing.assert_close(inter_h, inter_r, rtol=2e-3, atol=2e-4)
    for a, b in zip(out_h["heatmaps"], out_r["heatmaps"]):
        torch.testing.assert_close(a, b, rtol=2e-3, atol=5e-4)


@pytest.mark.parametrize("L,E,H,p", [(240, 384, 8, 0.1), (60, 384, 8, 0.0), (37, 256, 8, 0.25), (256, 384, 8, 0.1),
                                     (360, 384, 8, 0.1), (384, 384, 8, 0.0), (301, 256, 8, 0.2), (257, 384, 8, 0.1)])
def test_small_attention_kernel_with_dropout_against_composition(L, E, H, p, monkeypatch):
    """csrc/small_attention.cuh: forward and the one-launch backward against softmax(q k^T / sqrt(hd)) . v written out in
    PyTorch WITH THE KERNEL'S OWN dropout mask.  The mask is recovered by linearity: for a fixed seed the output is
    Pd . V, so probing with one-hot V blocks returns the dropped probabilities themselves."""
    import snipper_amd.fused as fused
    from snipper_amd.deformable_transformer import _SmallAttention
    monkeypatch.setattr(fused, "_next_seed", lambda: 0x1234567 + L)
    g = torch.Generator().manual_seed(L + H)
    bs, hd = 2, E // H
    qk = torch.randn(bs, L, 2 * E, generator=g).to(DEV).requires_grad_(True)
    v = torch.randn(bs, L, E, generator=g).to(DEV).requires_grad_(True)
    gy = torch.randn(bs, L, E, generator=g).to(DEV)
    out = _SmallAttention.apply(qk, v, H, p)
    # dropped probabilities by probing: V_probe[b, j, h*hd + e] = 1 if j == c * hd + e
    Pd = torch.zeros(bs, H, L, L, device=DEV)
    with torch.no_grad():
        for c in range((L + hd - 1) // hd):
            probe = torch.zeros(bs, L, H, hd, device=DEV)
            n = min(hd, L - c * hd)
            idx = torch.arange(n, device=DEV)
            probe[:, c * hd + idx, :, idx] = 1.0
            o = _SmallAttention.apply(qk.detach(), probe.view(bs, L, E), H, p).view(bs, L, H, hd)
            Pd[:, :, :, c * hd:c * hd + n] = o.permute(0, 2, 1, 3)[..., :n]
    q = qk[..., :E].view(bs, L, H, hd).transpose(1, 2)
    k = qk[..., E:].view(bs, L, H, hd).transpose(1, 2)
    P = torch.softmax(q @ k.transpose(-1, -2) * hd ** -0.5, -1)
    keep = (Pd > 0).float()
    if p == 0.0:
        assert bool(keep.all())
    else:
        frac = float(keep.mean())
        assert abs(frac - (1 - p)) < 0.01, frac
    ref = ((P * keep / (1 - p)) @ v.view(bs, L, H, hd).transpose(1, 2)).transpose(1, 2).reshape(bs, L, E)
    torch.testing.assert_close(Pd, (P * keep / (1 - p)).detach(), rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(out, ref, rtol=1e-4, atol=1e-5)
    g1 = torch.autograd.grad(out, (qk, v), gy)
    g2 = torch.autograd.grad(ref, (qk, v), gy)
    for a, b in zip(g1, g2):
        torch.testing.assert_close(a, b, rtol=1e-3, atol=2e-5 * float(b.abs().max()) + 1e-6)


@pytest.mark.parametrize("future", [0, 2])        # (2: forecast query frames sample the mean of ALL value frames, round 6)
def test_decoder_premixed_memory_path_equals_the_per_layer_path(future):
    """Round 5: with no padding the temporal mean of the memory is taken ONCE for all decoder layers and
    each layer samples value_proj(mean) -- mask fill, mean and projection commute (reference
    models/ops/modules/ms_deform_attn.py:114-118, 130-226 with tied Linears).  Same model, same inputs, bf16 autocast:
    outputs and parameter gradients against the per-layer evaluation."""
    import importlib.util
    from types import SimpleNamespace
    import snipper_amd.deformable_transformer as DT
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    a = SimpleNamespace(hidden_dim=384, enc_layers=1, dec_layers=3, frames=3, future_frames=future, use_pytorch_deform=0,
                        batch=2, height=192, width=256)          # 6 048 memory rows: the bf16 GEMM path (>= 4 096 rows)
    from snipper_amd.model import build_model
    torch.manual_seed(0)
    model = build_model(b.model_args(a)).to(DEV).to(memory_format=torch.channels_last).eval()
    with torch.no_grad():
        for n, p in model.named_parameters():          # real offsets / logits instead of the zero initialisation
            if "sampling_offsets" in n and n.endswith("weight"):
                p.normal_(0, 0.02)
            elif "attention_weights" in n:
                p.normal_(0, 0.3)
    imgs, _ = b.make_batches(a, DEV, 1, seed=3)[0]
    res = {}
    calls = []
    real = MSDeformAttn._forward_premixed
    MSDeformAttn._forward_premixed = lambda self, *a_, **k_: (calls.append(1), real(self, *a_, **k_))[1]
    # (round 6: with the premixed memory the decoder layers run as one native call each, decoder_native.DecoderLayerFn, which
    #  samples value_proj(mean) itself -- count those calls as well)
    from snipper_amd.decoder_native import DecoderLayerFn
    real_native = DecoderLayerFn.forward
    DecoderLayerFn.forward = staticmethod(lambda *x: (calls.append(1), real_native(*x))[1])
    for premix in (True, False):
        DT._DEC_PREMIX = premix
        calls.clear()
        try:
            with torch.autocast("cuda", dtype=torch.bfloat16):
                out, _ = model(list(imgs))
            assert len(calls) == (3 if premix else 0), calls        # taken by all three decoder layers, and only when on
            variant = _lib.last_variant()
            k = out["all_layers"]["pred_kpts"].float()
            w = torch.linspace(-1, 1, k.numel(), device=DEV).view_as(k)
            params = [p for p in model.parameters() if p.requires_grad]
            grads = torch.autograd.grad((k * w).sum(), params, allow_unused=True)
        finally:
            DT._DEC_PREMIX = True
        res[premix] = (k, out["pred_logits"].float(), grads, variant)
    MSDeformAttn._forward_premixed = real
    DecoderLayerFn.forward = real_native
    assert res[True][3].startswith("d48")
    rel = lambda x, y: ((x.double() - y.double()).norm() / y.double().norm().clamp_min(1e-20)).item()
    assert rel(res[True][0], res[False][0]) < 2e-2 and rel(res[True][1], res[False][1]) < 2e-2
    names = [n for n, p in model.named_parameters() if p.requires_grad]
    bad = []
    for n, g1, g0 in zip(names, res[True][2], res[False][2]):
        if g1 is None or g0 is None:
            assert g1 is None and g0 is None, n
            continue
        if rel(g1, g0) > 0.15:
            bad.append((n, rel(g1, g0)))
    assert len(bad) <= len(names) // 20, bad[:10]
