"""The one-call spatiotemporal entry points (snipper_st_msda_forward / _backward, include/snipper_dense.h) against the
composition the Python module makes from the individual entry points (fused.TemporalMix + fused.MSDAPrologue +
MSDeformAttnFunction and their autograd backward)."""
import ctypes

import pytest
import torch

from snipper_amd import _lib
from snipper_amd.fused import MSDAPrologue, TemporalMix
from snipper_amd.ms_deform_attn import frame_neighbours
from snipper_amd.ms_deform_attn_func import MSDeformAttnFunction

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _f(vals):
    return (ctypes.c_float * len(vals))(*[float(v) for v in vals])


@pytest.mark.parametrize("encoder", [True, False])
@pytest.mark.parametrize("vdt", [torch.float32, torch.bfloat16])
def test_st_entry_matches_composition(encoder, vdt):
    torch.manual_seed(17)
    hw = [(19, 25), (10, 13), (5, 7)]
    S = sum(h * w for h, w in hw)
    N, T, M, D, L, P = 2, 3, 8, 48, 3, 4
    C = M * D
    Lq = S if encoder else 40
    shapes = torch.tensor(hw, device=DEV)
    shapes._snipper_host = hw
    lsi = torch.cat((shapes.new_zeros(1), shapes.prod(1).cumsum(0)[:-1]))
    value = torch.randn(N, T, S, C, device=DEV).to(vdt).requires_grad_(True)
    mask = torch.rand(N, T, S, device=DEV) < 0.1
    off = (torch.randn(N, T, Lq, M * L * P * 2, device=DEV) * 2).requires_grad_(True)
    logit = torch.randn(N, T, Lq, M * L * P, device=DEV).requires_grad_(True)
    ref = torch.rand(N, T, Lq, L, 2, device=DEV).requires_grad_(True)
    groups = [frame_neighbours(t1, T, T) for t1 in range(T)]
    mix = [[(1.0 / len(g)) if t2 in g else 0.0 for t2 in range(T)] for g in groups]
    go = torch.randn(N * T, Lq, C, device=DEV)

    # composition (what MSDeformAttn._forward_tied does)
    vbar = TemporalMix.apply(value, mask, mix)
    loc, prob = MSDAPrologue.apply(off, logit, ref, hw, M, L, P)
    out = MSDeformAttnFunction.apply(vbar.view(N * T, S, M, D), shapes, lsi, loc.view(N * T, Lq, M, L, P, 2),
                                     prob.view(N * T, Lq, M, L, P), 64)
    g_value, g_off, g_logit, g_ref = torch.autograd.grad(out, (value, off, logit, ref), go)

    # one-call entry points
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    m8 = mask.contiguous().view(torch.uint8)
    inv_w, inv_h = _f([1.0 / w for h, w in hw]), _f([1.0 / h for h, w in hw])
    mixf = _f([w for row in mix for w in row])
    cv = lambda a: ctypes.cast(a, ctypes.c_void_p)
    hs = (ctypes.c_int64 * (2 * L))(*[int(v) for p in hw for v in p])
    vbar2 = torch.empty(N * T, S, M, D, device=DEV)
    loc2 = torch.empty(N * T, Lq, M, L, P, 2, device=DEV)
    prob2 = torch.empty(N * T, Lq, M, L, P, device=DEV)
    out2 = torch.empty(N * T, Lq, C, device=DEV)
    vd = 0 if vdt == torch.float32 else 1
    v_in, off_in, logit_in, ref_in = value.detach(), off.detach(), logit.detach(), ref.detach().contiguous()
    rc = lib.snipper_st_msda_forward(st, v_in.data_ptr(), vd, m8.data_ptr(), cv(mixf), off_in.data_ptr(), M * L * P * 2,
                                     logit_in.data_ptr(), M * L * P, 0, ref_in.data_ptr(), cv(inv_w), cv(inv_h),
                                     shapes.data_ptr(), lsi.data_ptr(), cv(hs), N, T, T, S, M, D, L, Lq, P,
                                     vbar2.data_ptr(), loc2.data_ptr(), prob2.data_ptr(), out2.data_ptr(), 0)
    _lib.check(rc, "snipper_st_msda_forward")
    _lib.note_variant()                 # (the C diagnostic is thread-local; this thread made the call)
    assert _lib.last_variant() == "d48_lp12", _lib.last_variant()
    torch.testing.assert_close(out2, out.detach(), rtol=0, atol=0)

    nbytes = lib.snipper_st_msda_backward_workspace_bytes(N, T, S, M, D, L, Lq, P, cv(hs))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
    gv2 = torch.empty_like(v_in)
    goff2, glogit2, gref2 = torch.empty_like(off_in), torch.empty_like(logit_in), torch.empty_like(ref_in)
    rc = lib.snipper_st_msda_backward(st, go.data_ptr(), 0, vbar2.data_ptr(), loc2.data_ptr(), prob2.data_ptr(),
                                      m8.data_ptr(), cv(mixf), cv(inv_w), cv(inv_h), shapes.data_ptr(), lsi.data_ptr(),
                                      cv(hs), N, T, T, S, M, D, L, Lq, P, ws.data_ptr(), nbytes,
                                      gv2.data_ptr(), vd, goff2.data_ptr(), M * L * P * 2, glogit2.data_ptr(), M * L * P,
                                      0, gref2.data_ptr())
    _lib.check(rc, "snipper_st_msda_backward")
    _lib.note_variant()
    # the encoder shape must run the owner-computes kernels (the ones the training step runs), the decoder shape the
    # atomic D=48 kernel
    assert _lib.last_variant() == ("d48_owner" if encoder else "d48_lp12"), _lib.last_variant()
    # (float atomics of the far taps are unordered: float32 sums differ in the last bits, a bf16 result by one ulp)
    rtol = 1e-5 if vdt == torch.float32 else 2 ** -7
    for a, b in ((gv2, g_value), (goff2, g_off), (glogit2, g_logit), (gref2, g_ref)):
        scale = max(float(b.float().abs().max()), 1.0)
        torch.testing.assert_close(a.float() / scale, b.float() / scale, rtol=rtol, atol=2e-6)
