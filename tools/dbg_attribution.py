#!/usr/bin/env python3
"""where does the bf16 step's error in the FIRST encoder layer's sampling_offsets gradient come from?
BASELINE configs[1] (T = 1, enc2 / dec4, 600x800), bf16 autocast against the float32 HIP evaluation of the same model, with
single ingredients of the bf16 path switched back to float32:
  value_proj_f32  : the encoder's value projection (and with it the sampled tensor) in float32
  output_proj_f32 : the encoder's output projection in float32 (the core op's rows and their gradient then are float32)
  offsets_f32     : the merged offset + logit projection evaluated in float32 (F.linear outside autocast)
Prints relative L2 errors of the test's gradients per variant."""
import json, os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_config1_gpu as T
import snipper_amd.ms_deform_attn as MOD

DEV = "cuda:0"
hip, _ = T._pair()
g = torch.Generator().manual_seed(6)
snippets = [torch.rand(3, 600, 800, generator=g).to(DEV)]
rel = lambda a, b: float((a.detach().float() - b.detach().float()).norm() / b.detach().float().norm().clamp_min(1e-20))


def run(amp):
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
        out, _ = hip(snippets)
    loss = T._loss(out)
    pd = dict(hip.named_parameters())
    return torch.autograd.grad(loss, [pd[n] for n in T.GRAD_NAMES])


ref = run(False)
real_merged, real_outside = MOD.big_linear_merged, MOD.merged_bias_is_outside


def merged_f32(q, lins, first_bias_outside=False):
    with torch.autocast("cuda", enabled=False):
        twin = q
        return torch.cat([F.linear(twin.float(), l.weight, l.bias) for l in lins], -1)


real_big = MOD.big_linear
value_projs = {id(l.self_attn.value_proj) for l in hip.transformer.encoder.layers}
out_projs = {id(l.self_attn.output_proj) for l in hip.transformer.encoder.layers}


def big_linear_sel(which):
    def f(x, lin, *a, **k):
        if id(lin) in which:
            with torch.autocast("cuda", enabled=False):
                return F.linear(x.float(), lin.weight, lin.bias)
        return real_big(x, lin, *a, **k)
    return f


for name, vf32, of32 in (("bench path", False, False), ("value_proj_f32", "v", False), ("output_proj_f32", "o", False),
                         ("value+output_proj_f32", "vo", False), ("offsets_f32", False, True)):
    MOD.big_linear = real_big if not vf32 else big_linear_sel((value_projs if "v" in vf32 else set()) | (out_projs if "o" in vf32 else set()))
    MOD.big_linear_merged = merged_f32 if of32 else real_merged
    MOD.merged_bias_is_outside = (lambda q, lins: False) if of32 else real_outside
    got = run(True)
    print(json.dumps({"variant": name, **{".".join(n.split(".")[1:]): round(rel(a, b), 4) for n, a, b in zip(T.GRAD_NAMES, got, ref)}}))
MOD.big_linear_merged, MOD.merged_bias_is_outside, MOD.big_linear = real_merged, real_outside, real_big
# the function itself: float32 everywhere, the input images perturbed by a relative 2^-9 (one bf16 rounding of the pixels)
clean = snippets
for name, eps in (("float32, images rounded to bf16", None), ("float32, images * (1 + 1e-3 noise)", 1e-3)):
    snippets = [x.to(torch.bfloat16).float() if eps is None else x * (1 + eps * torch.randn_like(x)) for x in clean]
    got = run(False)
    print(json.dumps({"variant": name, **{".".join(n.split(".")[1:]): round(rel(a, b), 4) for n, a, b in zip(T.GRAD_NAMES, got, ref)}}))
snippets = clean
