"""Weight-stationary GEMM (csrc/wres_gemm_bf16.cuh) and the transposed weight shadows, ``-m gpu``.

Reference layers: the 384 -> 384 / 288 / 1024 Linears of MSDeformAttn and of the encoder FFN (models/ops/modules/
ms_deform_attn.py:60-66, models/deformable_transformer.py:180-198) and their data gradients; checked against float64 on the
same bf16 operands (bf16 result: half an ulp of the largest output) and against the tile kernels they replace."""
import pytest
import torch

from snipper_amd import dense, shadow
from snipper_amd.dense import (linear_bf16, linear_nn_bf16, linear_wres_bf16, transpose_batch_bf16, wres_supported)

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _ops(M, K, N, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(M, K, generator=g).to(DEV).bfloat16()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(DEV).bfloat16()
    b = torch.randn(N, generator=g).to(DEV)
    return x, w, b


@pytest.mark.parametrize("M,K,N,relu,gate", [
    (8192, 384, 384, False, False), (9001, 384, 288, True, False), (8200, 288, 384, False, False),
    (8192, 384, 1024, True, False), (8209, 384, 1024, False, True), (40000, 384, 96, False, False),
    (8195, 288, 384, False, True), (79000, 384, 384, False, False), (8192 + 31, 384, 400, True, True)])
def test_wres_matches_float64(M, K, N, relu, gate):
    """Ragged row counts (the last chunk is partial), column counts that leave waves / whole workgroup columns partly
    empty (288, 96, 400, 1024 = 384 + 384 + 256), both reduction lengths, the gate ring."""
    assert wres_supported(M, N, K)
    x, w, b = _ops(M, K, N, M + N)
    a = torch.randn(M, N, device=DEV).relu().bfloat16() if gate else None
    y = linear_wres_bf16(x, w, b, relu=relu, gate=a, gate_scale=1.25 if gate else 1.0)
    ref = x.double() @ w.double().t() + b.double()
    if relu:
        ref = ref.relu()
    if gate:
        ref = torch.where(a.double() > 0, ref * 1.25, torch.zeros_like(ref))
    err = (y.double() - ref).abs().max().item()
    assert err <= ref.abs().max().item() * 2 ** -8 * 1.01, err          # half an ulp of bf16 at the largest magnitude
    if gate:
        assert torch.equal(y == 0, ~(a > 0) | (y == 0))


def test_linear_bf16_dispatches_to_wres_and_agrees_with_the_tile_kernel():
    """snipper_linear_bf16 takes the weight-stationary kernel for K = 384 on many rows; the tile kernel (reached with a
    zero residual) gives the same bf16 numbers up to the summation order."""
    x, w, b = _ops(20000, 384, 384, 3)
    y = linear_bf16(x, w, b)
    y_tile = linear_bf16(x, w, b, residual=torch.zeros(20000, 384, device=DEV, dtype=torch.bfloat16))
    assert (y.float() - y_tile.float()).abs().max().item() <= 2 ** -7 * y_tile.float().abs().max().item()
    assert torch.equal(y, linear_wres_bf16(x, w, b))
    # strided input rows (a column slice of a wider matrix)
    wide = torch.randn(20000, 512, device=DEV).bfloat16()
    xs = wide[:, 64:448]
    assert torch.equal(linear_bf16(xs, w, b), linear_bf16(xs.contiguous(), w, b))


def test_dropout_epilogue_statistics_and_seed():
    x, w, b = _ops(16384, 384, 1024, 5)
    plain = linear_wres_bf16(x, w, b, relu=True).float()
    p = 0.1
    y1 = linear_wres_bf16(x, w, b, relu=True, dropout_p=p, seed=1234).float()
    y2 = linear_wres_bf16(x, w, b, relu=True, dropout_p=p, seed=1234).float()
    y3 = linear_wres_bf16(x, w, b, relu=True, dropout_p=p, seed=99).float()
    assert torch.equal(y1, y2) and not torch.equal(y1, y3)
    active = plain > 0
    kept = (y1 != 0) & active
    frac = kept.sum().item() / active.sum().item()
    assert abs(frac - (1 - p)) < 2e-3, frac
    torch.testing.assert_close(y1[kept], (plain / (1 - p))[kept], rtol=2 ** -7, atol=1e-6)
    assert float(y1[~active].abs().max()) == 0.0
    # the four 16-bit fields of a hash word pair are usable uniforms: per-column keep rates within 5 sigma of 1 - p
    n_act = active.float().sum(0)
    col = ((y1 != 0) & active).float().sum(0) / n_act.clamp_min(1)
    z = (col - (1 - p)).abs() / (p * (1 - p) / n_act.clamp_min(1)).sqrt()
    assert float(z[n_act > 500].max()) < 5.0, float(z[n_act > 500].max())


def test_transposed_shadows_and_dgrad_route():
    """The per-step weight refresh keeps W^T for the Linears whose data gradient has a 288 / 384-long reduction; _dgrad
    with it runs the weight-stationary kernel and agrees with the transposing tile kernel."""
    torch.manual_seed(0)
    lin = torch.nn.Linear(384, 384).to(DEV)
    lin2 = torch.nn.Linear(1024, 384).to(DEV)           # FFN linear2: W [384, 1024]; dH = dZ . W2 has reduction 384
    off, att = torch.nn.Linear(384, 192).to(DEV), torch.nn.Linear(384, 96).to(DEV)

    class Holder(torch.nn.Module):
        pass
    ws = shadow.WeightShadows(Holder())
    ws.linears += [lin, lin2]
    ws.pairs.append((off, att))
    ws.refresh()
    for l in (lin, lin2):
        wt = shadow.lookup_t(l.weight)
        assert wt is not None and torch.equal(wt, l.weight.detach().bfloat16().t())
    wmt = shadow.lookup_merged_t(off, att)
    wm, _ = shadow.lookup_merged(off, att)
    assert wmt is not None and torch.equal(wmt, wm.t()) and wmt.shape == (384, 288)
    with torch.no_grad():
        lin.weight.add_(1.0)
    assert shadow.lookup_t(lin.weight) is None                     # stale with the weight
    ws.refresh()
    assert torch.equal(shadow.lookup_t(lin.weight), lin.weight.detach().bfloat16().t())
    g = torch.randn(20000, 384, device=DEV).bfloat16()
    wb = shadow.lookup(lin2.weight)
    h = torch.randn(20000, 1024, device=DEV).relu().bfloat16()
    a = dense._dgrad(g, wb, None, h, wt=shadow.lookup_t(lin2.weight), gate_scale=1.1)
    b_ = linear_nn_bf16(g, wb, None, h, 1.1)
    assert (a.float() - b_.float()).abs().max().item() <= 2 ** -7 * b_.float().abs().max().item()
    assert torch.equal(a == 0, b_ == 0)
    g288 = torch.randn(20000, 288, device=DEV).bfloat16()
    c = dense._dgrad(g288, wm, wt=wmt)
    ref = g288.double() @ wm.double()
    assert (c.double() - ref).abs().max().item() <= ref.abs().max().item() * 2 ** -8 * 1.01


def test_transpose_batch():
    srcs = [torch.randn(r, c, device=DEV).bfloat16() for r, c in [(384, 384), (288, 384), (384, 1024), (70, 130), (1, 9)]]
    dsts = [torch.empty(s.shape[1], s.shape[0], device=DEV, dtype=torch.bfloat16) for s in srcs]
    transpose_batch_bf16(list(zip(srcs, dsts)))
    for s, d in zip(srcs, dsts):
        assert torch.equal(d, s.t())
