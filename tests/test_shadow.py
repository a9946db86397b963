"""Weight shadows (snipper_amd/shadow.py) stay in step with the parameters under every optimizer implementation.

torch.optim.AdamW(fused=True) updates parameters without touching their version counters (PyTorch 2.10); the shadows
are keyed on those counters, so without the optimizer hook a fused training loop multiplies by the weights of step 0."""
import pytest
import torch
from torch import nn

from snipper_amd import shadow


@pytest.mark.parametrize("kw", [dict(fused=True), dict(foreach=True), dict(foreach=False, fused=False)])
def test_optimizer_step_bumps_version_counters_once_the_hook_is_installed(kw):
    shadow.install_optimizer_hook()
    p = nn.Parameter(torch.randn(4, 4))
    p.grad = torch.randn(4, 4)
    v0 = p._version
    torch.optim.AdamW([p], lr=0.1, **kw).step()
    assert p._version > v0


@pytest.mark.gpu
def test_shadows_follow_a_fused_optimizer_on_the_gpu():
    lin = nn.Linear(128, 64).cuda()

    class Holder(nn.Module):          # WeightShadows walks encoder layers / attention modules; drive it by hand here
        pass
    ws = shadow.WeightShadows(Holder())
    ws.linears.append(lin)
    ws.refresh()
    first = shadow.lookup(lin.weight)
    assert first is not None and torch.equal(first, lin.weight.detach().bfloat16())
    before = lin.weight.detach().clone()
    opt = torch.optim.AdamW(lin.parameters(), lr=0.5, fused=True)
    lin.weight.grad = torch.ones_like(lin.weight)
    lin.bias.grad = torch.ones_like(lin.bias)
    opt.step()
    assert not torch.equal(lin.weight.detach(), before)
    assert shadow.lookup(lin.weight) is None                 # stale: must not be served
    ws.refresh()
    again = shadow.lookup(lin.weight)
    assert again is not None and torch.equal(again, lin.weight.detach().bfloat16())


@pytest.mark.gpu
def test_refresh_with_weights_and_merged_biases_in_one_pass_keeps_every_dtype():
    """Round 3: one _foreach_copy_ over a list mixing bf16 <- f32 (weights) and f32 <- f32 (the merged projections' biases)
    wrote bf16 bits into the float32 bias buffers -- the offset projection ran without its bias under autocast.  Every
    shadow of a refresh that has Linears AND merged pairs must equal its source."""
    torch.manual_seed(0)
    lin = nn.Linear(128, 64).cuda()
    off, att = nn.Linear(128, 48).cuda(), nn.Linear(128, 24).cuda()
    with torch.no_grad():
        off.bias.copy_(torch.arange(48.0) - 20.0)

    class Holder(nn.Module):
        pass
    ws = shadow.WeightShadows(Holder())
    ws.linears.append(lin)
    ws.pairs.append((off, att))
    ws.refresh()
    assert torch.equal(shadow.lookup(lin.weight), lin.weight.detach().bfloat16())
    w, b = shadow.lookup_merged(off, att)
    assert w.dtype == torch.bfloat16 and b.dtype == torch.float32
    assert torch.equal(w, torch.cat([off.weight, att.weight]).detach().bfloat16())
    assert torch.equal(b, torch.cat([off.bias, att.bias]).detach())
