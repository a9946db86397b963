"""GPU tests of the fused element-wise kernels (csrc/msda_prologue.cuh) against their PyTorch formulation."""
import pytest
import torch
import torch.nn.functional as F

from snipper_amd.fused import MSDAPrologue, TemporalMix
from snipper_amd.ms_deform_attn import MSDeformAttn

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("with_mask", [False, True])
def test_temporal_mix_forward_backward(dtype, with_mask):
    g = torch.Generator().manual_seed(0)
    N, T2, S, C = 2, 3, 37, 24
    mix = [[0.5, 0.5, 0.0], [1 / 3, 1 / 3, 1 / 3], [0.0, 0.5, 0.5], [1 / 3, 1 / 3, 1 / 3], [0.2, 0.3, 0.5]]
    v = torch.randn(N, T2, S, C, generator=g).to(dtype).to(DEV).requires_grad_(True)
    mask = (torch.rand(N, T2, S, generator=g) < 0.2).to(DEV) if with_mask else None
    out = TemporalMix.apply(v, mask, mix)
    assert out.dtype == torch.float32 and out.shape == (N, 5, S, C)
    vv = v.detach().double().requires_grad_(True)
    vm = vv.masked_fill(mask[..., None], 0.0) if with_mask else vv
    ref = torch.einsum("ts,nsx->ntx", torch.tensor(mix, dtype=torch.float64, device=DEV), vm.reshape(N, T2, -1)).view(N, 5, S, C)
    torch.testing.assert_close(out.double(), ref, rtol=1e-5, atol=1e-5)
    go = torch.randn(out.shape, generator=g).to(DEV)
    (gv,) = torch.autograd.grad(out, v, go)
    (gr,) = torch.autograd.grad(ref, vv, go.double())
    assert gv.dtype == dtype
    tol = dict(rtol=1e-5, atol=1e-5) if dtype == torch.float32 else dict(rtol=2 ** -7, atol=2e-2)
    torch.testing.assert_close(gv.double(), gr, **tol)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,L,P", [(8, 3, 4), (4, 2, 2), (1, 4, 4), (2, 1, 1)])
def test_prologue_forward_backward(dtype, M, L, P):
    g = torch.Generator().manual_seed(M * 10 + L)
    N, T, Lq = 2, 3, 29
    hw = [(19, 25), (10, 13), (5, 7), (3, 2)][:L]
    off = (torch.randn(N, T, Lq, M * L * P * 2, generator=g) * 3).to(dtype).to(DEV).requires_grad_(True)
    logit = torch.randn(N, T, Lq, M * L * P, generator=g).to(dtype).to(DEV).requires_grad_(True)
    ref = torch.rand(N, T, Lq, L, 2, generator=g).to(DEV).requires_grad_(True)
    loc, prob = MSDAPrologue.apply(off, logit, ref, hw, M, L, P)
    o64, l64, r64 = (t.detach().double().requires_grad_(True) for t in (off, logit, ref))
    scale = torch.tensor([[w, h] for h, w in hw], dtype=torch.float64, device=DEV)
    loc_ref = r64[:, :, :, None, :, None, :] + o64.view(N, T, Lq, M, L, P, 2) / scale[None, None, None, None, :, None, :]
    prob_ref = F.softmax(l64.view(N, T, Lq, M, L * P), -1)
    torch.testing.assert_close(loc.double().view_as(loc_ref), loc_ref, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(prob.double().view_as(prob_ref), prob_ref, rtol=1e-5, atol=1e-6)
    gl, gp = torch.randn(loc.shape, generator=g).to(DEV), torch.randn(prob.shape, generator=g).to(DEV)
    got = torch.autograd.grad([loc, prob], [off, logit, ref], [gl, gp])
    want = torch.autograd.grad([loc_ref, prob_ref], [o64, l64, r64], [gl.double().view_as(loc_ref), gp.double().view_as(prob_ref)])
    tol = dict(rtol=1e-4, atol=1e-5) if dtype == torch.float32 else dict(rtol=2 ** -7, atol=2e-2)
    for a, b in zip(got, want):
        torch.testing.assert_close(a.double(), b, **tol)


@pytest.mark.parametrize("mode,T1", [("encoder", 3), ("decoder", 5)])
@pytest.mark.parametrize("amp", [False, True])
def test_module_fused_equals_unfused(mode, T1, amp):
    torch.manual_seed(1)
    shapes = [(9, 12), (5, 6), (3, 3)]
    S = sum(h * w for h, w in shapes)
    C, M = 96, 8
    Lq = S if mode == "encoder" else 7
    mod = MSDeformAttn(C, 3, M, 4, 3, mode, False, mode == "decoder").to(DEV)
    with torch.no_grad():
        for p in mod.parameters():
            if float(p.abs().max()) == 0:
                p.normal_(0, 0.05)
    sh = torch.tensor(shapes, device=DEV)
    lsi = torch.cat((sh.new_zeros(1), sh.prod(1).cumsum(0)[:-1]))
    q = torch.randn(2, T1, Lq, C, device=DEV, requires_grad=True)
    src = torch.randn(2, 3, S, C, device=DEV, requires_grad=True)
    ref = torch.rand(2, T1, Lq, 3, 2, device=DEV, requires_grad=True)
    mask = torch.zeros(2, 3, S, 1, dtype=torch.bool, device=DEV)
    mask[1, :, -4:] = True
    mask = mask.expand(-1, -1, -1, C)
    go = torch.randn(2, T1, Lq, C, device=DEV)
    res = {}
    for fused in (True, False):
        mod.fused_elementwise = fused
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
            out = mod(q, ref, src, sh, lsi, mask)
        out = out[0] if isinstance(out, tuple) else out
        res[fused] = (out, torch.autograd.grad(out, [q, src, ref] + list(mod.parameters()), go.to(out.dtype)))
    tol = dict(rtol=2e-4, atol=2e-5) if not amp else dict(rtol=5e-2, atol=5e-2)
    torch.testing.assert_close(res[True][0].float(), res[False][0].float(), **tol)
    for a, b in zip(res[True][1], res[False][1]):
        s = max(float(b.float().abs().max()), 1.0)
        torch.testing.assert_close(a.float() / s, b.float() / s, **tol)
