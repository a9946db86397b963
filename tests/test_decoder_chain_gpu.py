"""Round 6: the decoder layer as a chain of fused launches -- csrc/small_ln.cuh, the float32-row forms of the
core op beside a bf16 value, the root head inside the refinement launch, fused.FanOut for float32 -- each against the plain
PyTorch formulation, and the whole chain against the node-per-module decoder it replaces
(reference models/deformable_transformer.py:244-343)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


@pytest.mark.parametrize("rows,C,p,n_alias,with_pos", [(480, 384, 0.1, 3, True), (720, 384, 0.0, 1, True), (37, 256, 0.25, 2, False),
                                                       (480, 384, 0.1, 1, False), (5, 1024, 0.0, 4, True)])
def test_small_layer_norm_against_pytorch(rows, C, p, n_alias, with_pos):
    """y = LayerNorm(x + dropout(z)), yq = y + pos, and the backward with one gradient per consumer: against the PyTorch
    composition under the kernel's own mask (read back from the output: with p > 0 the mask is recovered by probing)."""
    from snipper_amd.fused import SmallLayerNorm
    g = torch.Generator().manual_seed(rows + C)
    x = torch.randn(2, rows // 2 if rows % 2 == 0 else rows, C, generator=g) if rows % 2 == 0 else torch.randn(1, rows, C, generator=g)
    z = torch.randn(x.shape, generator=g)
    pos = torch.randn(x.shape, generator=g)
    norm = torch.nn.LayerNorm(C)
    with torch.no_grad():
        norm.weight.copy_(1 + 0.3 * torch.randn(C, generator=g))
        norm.bias.copy_(0.2 * torch.randn(C, generator=g))
    norm = norm.to(DEV)
    x, z, pos = (t.to(DEV).requires_grad_(True) for t in (x, z, pos))
    seed_state = torch.initial_seed()
    from snipper_amd import fused
    calls0 = fused._dropout_calls
    outs = SmallLayerNorm.apply(x, z, pos if with_pos else None, norm.weight, norm.bias, p, norm.eps, n_alias, with_pos)
    assert len(outs) == n_alias + (1 if with_pos else 0)
    # the mask: run the same call (same seed) on z = ones, x = 0 with an identity-free probe -> kept elements are where the
    # pre-norm sum is non-zero; simpler: the kernel's saved keep-bits are internal, so recover the mask from linearity in z
    if p > 0:
        fused._dropout_calls = calls0                        # replay the seed
        probe_z = torch.ones_like(z)
        big = SmallLayerNorm.apply(torch.zeros_like(x), probe_z, None, torch.ones_like(norm.weight), torch.zeros_like(norm.bias),
                                   p, norm.eps, 1, False)[0]
        # rows of the probe: values are (k / (1 - p) - mean) * rstd with k in {0, 1}: kept elements are the larger value of the row
        mask = (big > big.mean(-1, keepdim=True)).float() if p > 0 else torch.ones_like(z)
        assert abs(float(mask.mean()) - (1 - p)) < 0.03
        assert torch.initial_seed() == seed_state
    else:
        mask = torch.ones_like(z)
    s = x + z * mask / (1 - p)
    want = torch.nn.functional.layer_norm(s, (C,), norm.weight, norm.bias, norm.eps)
    for o in outs[:n_alias]:
        torch.testing.assert_close(o, want, rtol=2e-5, atol=2e-5)
    if with_pos:
        torch.testing.assert_close(outs[-1], want + pos, rtol=2e-5, atol=2e-5)
    ws = [torch.randn(want.shape, generator=g).to(DEV) for _ in outs]
    got = torch.autograd.grad(sum((o * w).sum() for o, w in zip(outs, ws)), [x, z] + ([pos] if with_pos else []) +
                              [norm.weight, norm.bias])
    gy = sum(ws)
    ref = torch.autograd.grad((want * gy).sum() + ((pos * ws[-1]).sum() if with_pos else 0.0), [x, z] + ([pos] if with_pos else []) +
                              [norm.weight, norm.bias])
    for a, b, name in zip(got, ref, ["dx", "dz"] + (["dpos"] if with_pos else []) + ["dgamma", "dbeta"]):
        assert _rel(a, b) < 3e-5, (name, _rel(a, b))


def test_small_layer_norm_is_bit_reproducible():
    from snipper_amd.fused import SmallLayerNorm
    g = torch.Generator().manual_seed(1)
    x, z = torch.randn(2, 240, 384, generator=g).to(DEV).requires_grad_(True), torch.randn(2, 240, 384, generator=g).to(DEV).requires_grad_(True)
    gamma, beta = torch.randn(384, generator=g).to(DEV).requires_grad_(True), torch.randn(384, generator=g).to(DEV).requires_grad_(True)
    w = torch.randn(2, 240, 384, generator=g).to(DEV)
    res = []
    for _ in range(2):
        (y,) = SmallLayerNorm.apply(x, z, None, gamma, beta, 0.0, 1e-5, 1, False)
        res.append([y.detach().clone()] + [t.clone() for t in torch.autograd.grad((y * w).sum(), [x, z, gamma, beta])])
    for a, b in zip(*res):
        assert torch.equal(a, b)


def test_fan_out_float32_sums_the_aliases_gradients_in_one_pass():
    from snipper_amd.fused import FanOut
    g = torch.Generator().manual_seed(2)
    x = torch.randn(2, 240, 384, generator=g).to(DEV).requires_grad_(True)
    ws = [torch.randn(2, 240, 384, generator=g).to(DEV) for _ in range(12)]
    outs = FanOut.apply(x, 12)
    (gx,) = torch.autograd.grad(sum((o * w).sum() for o, w in zip(outs, ws)), [x])
    want = torch.stack(ws).double().sum(0)
    assert _rel(gx, want) < 1e-6


def test_refinement_with_the_root_head_in_the_same_launch():
    """reference models/deformable_transformer.py:329-333 with the model's root head (ONE Linear(C, 4), models/model.py:95)."""
    from snipper_amd.deformable_transformer import _refine_reference, inverse_sigmoid
    from snipper_amd.model import MLP
    g = torch.Generator().manual_seed(3)
    head = MLP(384, 384, 4, 1).to(DEV)
    out = torch.randn(2, 4, 60, 384, generator=g).to(DEV)
    ref = torch.rand(2, 4, 60, 2, generator=g)
    ref[0, 0, 0] = torch.tensor([0.0, 1.0])
    ref[0, 0, 1] = torch.tensor([1e-7, 1 - 1e-7])
    ref = ref.to(DEV)
    vr = (0.5 + 0.5 * torch.rand(2, 3, 2, generator=g)).to(DEV)
    new_ref, ref_in = _refine_reference(head, out, ref, vr)
    want = (head(out)[..., 0:2] + inverse_sigmoid(ref)).sigmoid().detach()
    torch.testing.assert_close(new_ref, want, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(ref_in, want[:, :, :, None, :] * vr[:, None, None, :, :], rtol=1e-5, atol=1e-6)
    assert not new_ref.requires_grad and not ref_in.requires_grad


def _core_inputs(N=8, Lq=60, seed=0):
    from oracle import msda_oracle as O
    rng = np.random.RandomState(seed)
    shapes = np.array([(38, 50), (19, 25), (10, 13)], dtype=np.int64)
    lsi = O.level_start_index(shapes)
    M, D, L, P = 8, 48, 3, 4
    S = int((shapes[:, 0] * shapes[:, 1]).sum())
    value = rng.standard_normal((N, S, M, D)).astype(np.float32)
    loc = rng.uniform(-0.1, 1.1, (N, Lq, M, L, P, 2)).astype(np.float32)
    attn = rng.uniform(0, 1, (N, Lq, M, L, P)).astype(np.float32)
    attn /= attn.sum((-1, -2), keepdims=True)
    go = rng.standard_normal((N, Lq, M * D)).astype(np.float32)
    return shapes, lsi, value, loc, attn, go


def test_float32_rows_beside_a_bf16_value_against_the_oracle():
    """The decoder's cross attention under bf16 autocast: bf16 value projection, float32 queries.  Forward rows written as
    float32 (the float32 sums, not their bf16 rounding); backward from float32 grad_out rows through the sort-by-pixel kernel --
    both against the C oracle on the bf16-rounded value."""
    from oracle import msda_oracle as O
    from snipper_amd import MultiScaleDeformableAttention as MSDA
    from snipper_amd import _lib
    shapes, lsi, value, loc, attn, go = _core_inputs()
    t = lambda a: torch.from_numpy(a).to(DEV)
    v16 = t(value).to(torch.bfloat16)
    vr = v16.float().cpu().numpy().astype(np.float64)
    f64 = lambda a: a.astype(np.float64)
    out = MSDA.ms_deform_attn_forward(v16, t(shapes), t(lsi), t(loc), t(attn), 64, out_f32=True)
    assert out.dtype == torch.float32 and _lib.last_variant() == "d48_lp12"
    ref = O.core_c_forward(vr, shapes, lsi, f64(loc), f64(attn))
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=1e-4, atol=2e-5)
    # the bf16-row form of the same launch is this result rounded once
    out16 = MSDA.ms_deform_attn_forward(v16, t(shapes), t(lsi), t(loc), t(attn), 64)
    assert torch.equal(out16, out.to(torch.bfloat16))
    gv, gl, ga = MSDA.ms_deform_attn_backward(v16, t(shapes), t(lsi), t(loc), t(attn), t(go), 64)
    assert _lib.last_variant() == "d48_sparse" and gv.dtype == torch.bfloat16
    rgv, rgl, rga = O.core_c_backward(vr, shapes, lsi, f64(loc), f64(attn), f64(go))
    np.testing.assert_allclose(gv.float().cpu().numpy(), rgv, rtol=2 ** -7, atol=2e-3)
    np.testing.assert_allclose(ga.cpu().numpy(), rga, rtol=1e-4, atol=1e-4)
    s = np.abs(rgl).max()
    np.testing.assert_allclose(gl.cpu().numpy() / s, rgl / s, rtol=1e-4, atol=2e-5)
    # and through the autograd node
    from snipper_amd.ms_deform_attn_func import MSDeformAttnFunction
    vq = v16.clone().requires_grad_(True)
    lq, aq = t(loc).requires_grad_(True), t(attn).requires_grad_(True)
    o = MSDeformAttnFunction.apply(vq, t(shapes), t(lsi), lq, aq, 64, False, True)
    assert o.dtype == torch.float32
    g1 = torch.autograd.grad(o, [vq, lq, aq], t(go))
    assert torch.equal(g1[0], gv) and torch.equal(g1[1], gl) and torch.equal(g1[2], ga)


@pytest.mark.parametrize("future,T,nq", [(0, 3, 20), (2, 3, 20), (2, 4, 60)])          # (the last: 360 object queries, BASELINE configs[4])
def test_decoder_chain_equals_the_node_per_module_decoder(future, T, nq):
    """The whole transformer (float32, d_model 384 so that every decoder-size kernel applies, dropout 0 so that both forms
    are deterministic) with the decoder as a chain and as one node per module: outputs, refined references and every
    parameter gradient."""
    from snipper_amd.deformable_transformer import DeformableTransformer, DeformableTransformerDecoderLayer as Layer
    torch.manual_seed(0)
    tr = DeformableTransformer(d_model=384, nhead=8, num_encoder_layers=1, num_decoder_layers=3, dim_feedforward=512, dropout=0.0,
                               return_intermediate_dec=True, num_feature_levels=3, dec_n_points=4, enc_n_points=4, n_frame=T,
                               n_future_frame=future, num_keypoints=15).to(DEV)
    from snipper_amd.model import MLP
    heads = torch.nn.ModuleList([MLP(384, 384, 4, 1).to(DEV)] * 3)
    tr.decoder.root_embed = heads
    with torch.no_grad():
        for n, p in tr.named_parameters():
            if "sampling_offsets" in n and n.endswith("weight"):
                p.normal_(0, 0.02)
            elif "attention_weights" in n:
                p.normal_(0, 0.3)
    g = torch.Generator().manual_seed(5)
    hw = [(12, 16), (6, 8), (3, 4)]
    bs = 2
    srcs = [torch.randn(bs, 384, T, h, w, generator=g).to(DEV) for h, w in hw]
    masks = [torch.zeros(bs, 384, T, h, w, dtype=torch.bool, device=DEV) for h, w in hw]
    pos = [torch.randn(bs, 384, T, h, w, generator=g).to(DEV) for h, w in hw]
    qe = torch.randn((T + future) * nq, 768, generator=g).to(DEV).requires_grad_(True)
    res = {}
    seen = []
    real = Layer.forward_chain
    Layer.forward_chain = lambda self, *a, **k: (seen.append(1), real(self, *a, **k))[1]
    try:
        for chain in (True, False):
            Layer.chain = chain
            seen.clear()
            hs, _, _, refs, _ = tr(srcs, masks, pos, qe)
            assert len(seen) == (3 if chain else 0)
            w = torch.linspace(-1, 1, hs.numel(), device=DEV).view_as(hs)
            params = [qe] + [p for p in tr.parameters() if p.requires_grad] + [p for p in heads.parameters()]
            grads = torch.autograd.grad((hs * w).sum(), params, allow_unused=True)
            res[chain] = (hs.detach(), refs.detach(), grads)
    finally:
        Layer.chain = True
        Layer.forward_chain = real
    assert _rel(res[True][0], res[False][0]) < 1e-5 and _rel(res[True][1], res[False][1]) < 1e-5
    names = ["query_embed"] + [n for n, p in tr.named_parameters() if p.requires_grad] + ["head.w", "head.b"]
    for n, a, b in zip(names, res[True][2], res[False][2]):
        if a is None or b is None:
            assert a is None and b is None, n
            continue
        assert _rel(a, b) < 2e-4, (n, _rel(a, b))
