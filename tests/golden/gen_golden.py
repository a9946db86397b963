#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE itself.

Runs only in the build container (needs /root/reference); the GPU box never
sees the reference, only the .npz/.pt files written here.  Nothing from the
reference is copied: it is imported (PYTHONDONTWRITEBYTECODE=1, read-only tree),
fed seeded inputs, and its outputs / autograd gradients are stored as tensors.

    python tests/golden/gen_golden.py            # rewrites every fixture

Fixtures:
  g1_testpy.npz        the cases of models/ops/test.py (seed 3, CPU RNG, call order of
                       :31-78,85-86): use_pytorch_deform=1 outputs in fp64/fp32 and, for the
                       seven gradcheck channel counts, autograd gradients in fp64.
  g2_core_d48.npz      D=48, M=8, L=3, P=4 core op with out-of-range locations; fp64.
  g3_module_*.pt       MSDeformAttn modules (encoder T=3, decoder T=3 and T=3+2) with randomised
                       tied Linears and a padding mask: outputs, input grads, param grads, vis lists.
  g6_model.pt          the reference SnipperDeformable on a replayed backbone: outputs + state_dict schema + aliases
  g5_criterion.pt      SetCriterion + HungarianMatcher (reference classes, torchvision / cv2 stubbed) on random
                       3-layer outputs: every loss, the matching, gradients of the weighted sum.
  g3_module_*_d48.pt   the same module at Snipper's real head geometry (d_model 384, 8 heads -> D = 48, the width the tuned
                       HIP kernels are specialised for), float32 storage of a float64 evaluation.
  g3_module_*_t1_d48.pt  the same at T = 1 (BASELINE configs[1]: no temporal neighbours, identity mix).
  g3_module_*_untied_d48.pt  per-frame Linears untied with different values (general per-pair path).
  g7_posenc.npz        PositionEmbeddingSine (models/position_encoding.py:20-63) on padded and unpadded masks.
  g4_transformer.pt    one DeformableTransformer forward (T=2+1, enc2/dec2) + its state_dict
                       (pins the key schema) + gradients of a scalar loss.
"""
import os
import sys
import types

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True

import numpy as np
import torch
import torch.nn.functional as F

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def import_reference():
    # util/misc.py:19-49 gates on torchvision.__version__ at import time; torchvision is absent here.
    tv = types.ModuleType("torchvision")
    tv.__version__ = "0.9.0"
    tv.ops = types.ModuleType("torchvision.ops")
    tv.ops.misc = types.ModuleType("torchvision.ops.misc")
    tv.ops.misc.interpolate = F.interpolate
    sys.modules.setdefault("torchvision", tv)
    sys.modules.setdefault("torchvision.ops", tv.ops)
    sys.modules.setdefault("torchvision.ops.misc", tv.ops.misc)
    sys.path.insert(0, REF)
    from models.ops.functions.ms_deform_attn_func import ms_deform_attn_core_pytorch
    from models.ops.modules import MSDeformAttn
    from models.deformable_transformer import DeformableTransformer
    return ms_deform_attn_core_pytorch, MSDeformAttn, DeformableTransformer


def lsi_of(shapes):
    return torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))


def testpy_inputs(channels, gen=None):
    """The draw sequence every check in models/ops/test.py uses (:33-36)."""
    N, M, Lq, L, P, S = 1, 2, 2, 2, 2, 30
    value = torch.rand(N, S, M, channels) * 0.01
    loc = torch.rand(N, Lq, M, L, P, 2)
    attn = torch.rand(N, Lq, M, L, P) + 1e-5
    attn /= attn.sum(-1, keepdim=True).sum(-2, keepdim=True)
    return value, loc, attn


def gen_g1(core):
    shapes = torch.as_tensor([(6, 4), (3, 2)], dtype=torch.long)
    torch.manual_seed(3)                                      # test.py:28
    out = {"shapes": shapes.numpy()}
    v, l, a = testpy_inputs(2)                                # check_forward_equal_with_pytorch_double
    out["fwd64_value"], out["fwd64_loc"], out["fwd64_attn"] = v.numpy(), l.numpy(), a.numpy()
    out["fwd64_out"] = core(v.double(), shapes, l.double(), a.double()).numpy()
    v, l, a = testpy_inputs(2)                                # check_forward_equal_with_pytorch_float
    out["fwd32_value"], out["fwd32_loc"], out["fwd32_attn"] = v.numpy(), l.numpy(), a.numpy()
    out["fwd32_out"] = core(v, shapes, l, a).numpy()
    for D in [30, 32, 64, 71, 1025, 2048, 3096]:              # test.py:85-86
        v, l, a = testpy_inputs(D)
        v64 = v.double().requires_grad_(True)
        l64 = l.double().requires_grad_(True)
        a64 = a.double().requires_grad_(True)
        o = core(v64, shapes, l64, a64)
        go = torch.from_numpy(np.random.RandomState(D).standard_normal(tuple(o.shape)))
        gv, gl, ga = torch.autograd.grad(o, (v64, l64, a64), go)
        out[f"gc{D}_out"] = o.detach().numpy()
        out[f"gc{D}_grad_out"] = go.numpy()
        out[f"gc{D}_grad_loc"] = gl.numpy()
        out[f"gc{D}_grad_attn"] = ga.numpy()
        if D <= 71:
            out[f"gc{D}_value"], out[f"gc{D}_loc"], out[f"gc{D}_attn"] = v.numpy(), l.numpy(), a.numpy()
            out[f"gc{D}_grad_value"] = gv.numpy()
        else:  # keep the file small: every 64th channel plus the channel sum pin grad_value
            out[f"gc{D}_grad_value_s64"] = gv[..., ::64].contiguous().numpy()
            out[f"gc{D}_grad_value_sum"] = gv.sum(-1).numpy()
            out[f"gc{D}_value_sum"] = v.double().sum().numpy()    # pins the RNG stream
    np.savez_compressed(os.path.join(OUT, "g1_testpy.npz"), **out)


def gen_g2(core):
    g = torch.Generator().manual_seed(1234)
    N, M, D, L, P, Lq = 2, 8, 48, 3, 4, 50
    shapes = torch.as_tensor([(12, 16), (6, 8), (3, 4)], dtype=torch.long)
    S = int(shapes.prod(1).sum())
    value = torch.randn(N, S, M, D, generator=g)
    loc = torch.rand(N, Lq, M, L, P, 2, generator=g) * 1.4 - 0.2          # exercises the >-1 / <H edges
    loc = (torch.round(loc * 4096) + 0.5) / 4096   # half-offset 2^-12 lattice: pixel coordinates exact in
    # fp32 and fp64 and never ON a pixel boundary (where the CUDA op's `> -1` rule and grid_sample's
    # gradient differ on a measure-zero set)
    # exact-boundary locations along x at level 0 (W=16, so the pixel coordinate is exact in fp32 too):
    # pixel x = -1 (skipped by the `> -1` rule), 0 (lw = 0), W (skipped), W-1 (right tap outside).
    # y stays on the lattice, except for the first one where y = -1 as well (with only x = -1 the
    # CUDA op's rule and grid_sample's gradient differ on this measure-zero set).
    ylat = (1234 + 0.5) / 4096
    loc[0, 0, 0, 0, 0] = torch.tensor([-0.5 / 16, -0.5 / 12])
    loc[0, 0, 0, 0, 1] = torch.tensor([0.5 / 16, ylat])
    loc[0, 0, 0, 0, 2] = torch.tensor([1.0 + 0.5 / 16, ylat])
    loc[0, 0, 0, 0, 3] = torch.tensor([1.0 - 0.5 / 16, ylat])
    attn = torch.softmax(torch.randn(N, Lq, M, L * P, generator=g), -1).view(N, Lq, M, L, P)
    grad_out = torch.randn(N, Lq, M * D, generator=g)
    v64, l64, a64 = (t.double().requires_grad_(True) for t in (value, loc, attn))
    o = core(v64, shapes, l64, a64)
    gv, gl, ga = torch.autograd.grad(o, (v64, l64, a64), grad_out.double())
    np.savez_compressed(
        os.path.join(OUT, "g2_core_d48.npz"), shapes=shapes.numpy(),
        value=value.numpy(), loc=loc.numpy(), attn=attn.numpy(), grad_out=grad_out.numpy(),
        out=o.detach().numpy(), grad_value=gv.numpy().astype(np.float32), grad_loc=gl.numpy(),
        grad_attn=ga.numpy(),
        out32=core(value, shapes, loc, attn).numpy())


def randomise_(module, gen, scale=0.3):
    """Give the zero-initialised offset/weight Linears (ms_deform_attn.py:78-97) real values."""
    with torch.no_grad():
        for name, p in module.named_parameters():
            if "sampling_offsets" in name and name.endswith("weight"):
                p.copy_(torch.randn(p.shape, generator=gen) * 0.05)
            elif "attention_weights" in name:
                p.copy_(torch.randn(p.shape, generator=gen) * scale)


def gen_g3(MSDeformAttn):
    d_model, M, L, P = 48, 4, 3, 4
    shapes = torch.as_tensor([(6, 8), (3, 4), (2, 2)], dtype=torch.long)
    S = int(shapes.prod(1).sum())
    lsi = lsi_of(shapes)
    cases = {"enc_t3": ("encoder", 3, 3, S), "dec_t3": ("decoder", 3, 3, 5), "dec_t3f2": ("decoder", 3, 5, 5)}
    for name, (mode, n_frame, T1, Lq) in cases.items():
        g = torch.Generator().manual_seed({"enc_t3": 101, "dec_t3": 102, "dec_t3f2": 103}[name])
        torch.manual_seed(11)
        mod = MSDeformAttn(d_model, L, M, P, n_frame, mode, True, mode == "decoder").double()
        randomise_(mod, g)
        N, T2 = 2, n_frame
        query = torch.randn(N, T1, Lq, d_model, generator=g).double().requires_grad_(True)
        ref = (torch.rand(N, T1, Lq, L, 2, generator=g).double() * 1.1 - 0.05).requires_grad_(True)
        src = torch.randn(N, T2, S, d_model, generator=g).double().requires_grad_(True)
        mask = torch.zeros(N, T2, S, dtype=torch.bool)
        mask[1, :, -3:] = True
        mask[0, :, 5] = True
        mask_c = mask[..., None].expand(-1, -1, -1, d_model).contiguous()   # model.py:156-157 C-expanded
        res = mod(query, ref, src, shapes, lsi, mask_c)
        vis = None
        if isinstance(res, tuple):
            res, vis = res
        go = torch.randn(res.shape, generator=g).double()
        params = dict(mod.named_parameters())   # de-duplicated (tied) parameters
        grads = torch.autograd.grad(res, [query, ref, src] + list(params.values()), go)
        blob = {
            "cfg": dict(d_model=d_model, n_levels=L, n_heads=M, n_points=P, n_frame=n_frame, mode=mode),
            "state_dict": {k: v.detach().clone() for k, v in mod.state_dict().items()},
            "shapes": shapes, "lsi": lsi, "query": query.detach(), "ref": ref.detach(), "src": src.detach(),
            "mask": mask_c, "out": res.detach(), "grad_out": go,
            "grad_query": grads[0], "grad_ref": grads[1], "grad_src": grads[2],
            "param_grads": {k: g_ for k, g_ in zip(params.keys(), grads[3:])},
        }
        if vis is not None:
            blob["vis_loc"] = [t.clone() for t in vis[0]]
            blob["vis_w"] = [t.clone() for t in vis[1]]
        torch.save(blob, os.path.join(OUT, f"g3_module_{name}.pt"))


def gen_g3_t1_d48(MSDeformAttn):
    """BASELINE configs[1]'s frame count (T = 1: one query frame, one value frame, no temporal neighbours) at the same head
    geometry: g3_module_enc_t1_d48 / g3_module_dec_t1_d48."""
    gen_g3_d48(MSDeformAttn, {"enc_t1_d48": ("encoder", 1, 1, None, 203), "dec_t1_d48": ("decoder", 1, 1, 6, 204)})


def gen_g3_untied_d48(MSDeformAttn):
    """The per-frame offset / weight Linears UNTIED with different values per value frame (the reference's forward indexes
    them by t2, ms_deform_attn.py:144,167,197,210, so it evaluates an untied module as written): pins the general
    per-pair path -- joint softmax over L*P*|t2|, one core call per (t1, t2), non-contiguous value[:, t2] slices."""
    gen_g3_d48(MSDeformAttn, {"enc_untied_d48": ("encoder", 3, 3, None, 205), "dec_untied_d48": ("decoder", 2, 3, 6, 206)},
               untie=True)


def gen_g3_d48(MSDeformAttn, cases=None, untie=False):
    """MSDeformAttn at d_model=384 / 8 heads (D=48): encoder (Lq == S, T=2) and decoder (Lq=6, T=2+1).  Everything is
    drawn in float32 and evaluated by the reference in float64; results are stored as float32 (the consumers are the
    float32 D=48 kernels, compared at 2e-4), which keeps the two files at a few MB."""
    d_model, M, L, P = 384, 8, 3, 4
    shapes = torch.as_tensor([(12, 16), (6, 8), (3, 4)], dtype=torch.long)
    S = int(shapes.prod(1).sum())
    lsi = lsi_of(shapes)
    if cases is None:
        cases = {"enc_d48": ("encoder", 2, 2, S, 201), "dec_d48": ("decoder", 2, 3, 6, 202)}
    for name, (mode, n_frame, T1, Lq, seed) in cases.items():
        Lq = S if Lq is None else Lq
        g = torch.Generator().manual_seed(seed)
        torch.manual_seed(13)
        mod = MSDeformAttn(d_model, L, M, P, n_frame, mode, True, mode == "decoder")
        if untie:
            import copy
            mod.sampling_offsets = torch.nn.ModuleList([copy.deepcopy(mod.sampling_offsets[0]) for _ in range(n_frame)])
            mod.attention_weights = torch.nn.ModuleList([copy.deepcopy(mod.attention_weights[0]) for _ in range(n_frame)])
        randomise_(mod, g, scale=0.5)       # (untied: every copy draws its own values)
        if untie:
            with torch.no_grad():
                for lin in mod.sampling_offsets:
                    lin.bias.add_(torch.randn(lin.bias.shape, generator=g) * 0.5)
            assert not torch.equal(mod.sampling_offsets[0].weight, mod.sampling_offsets[1].weight)
        sd32 = {k: v.detach().clone() for k, v in mod.state_dict().items()}
        mod = mod.double()
        N, T2 = 1, n_frame
        query32 = torch.randn(N, T1, Lq, d_model, generator=g)
        if mode == "encoder":     # encoder-like reference points: the query's own pixel centre on every level
            refs = []
            for h, w in shapes.tolist():
                ys, xs = torch.meshgrid(torch.arange(h) + 0.5, torch.arange(w) + 0.5, indexing="ij")
                refs.append(torch.stack([xs.reshape(-1) / w, ys.reshape(-1) / h], -1))
            ref32 = torch.cat(refs)[None, None, :, None, :].expand(N, T1, S, L, 2).contiguous()
        else:
            ref32 = torch.rand(N, T1, Lq, L, 2, generator=g) * 1.1 - 0.05
        src32 = torch.randn(N, T2, S, d_model, generator=g)
        query, ref, src = (t.double().requires_grad_(True) for t in (query32, ref32, src32))
        mask = torch.zeros(N, T2, S, dtype=torch.bool)
        mask[0, :, 7] = True
        mask[0, T2 - 1, -2:] = True
        mask_c = mask[..., None].expand(-1, -1, -1, d_model).contiguous()
        res = mod(query, ref, src, shapes, lsi, mask_c)
        vis = None
        if isinstance(res, tuple):
            res, vis = res
        go32 = torch.randn(res.shape, generator=g)
        params = dict(mod.named_parameters())
        grads = torch.autograd.grad(res, [query, ref, src] + list(params.values()), go32.double())
        blob = {
            "cfg": dict(d_model=d_model, n_levels=L, n_heads=M, n_points=P, n_frame=n_frame, mode=mode,
                        **({"untied": True} if untie else {})),
            "state_dict": sd32, "shapes": shapes, "lsi": lsi, "query": query32, "ref": ref32, "src": src32,
            "mask": mask, "out": res.detach().float(), "grad_out": go32,
            "grad_query": grads[0].float(), "grad_ref": grads[1].float(), "grad_src": grads[2].float(),
            "param_grads": {k: g_.float() for k, g_ in zip(params.keys(), grads[3:])},
        }
        if vis is not None:
            blob["vis_w"] = [t.float().clone() for t in vis[1]]
        torch.save(blob, os.path.join(OUT, f"g3_module_{name}.pt"))


def gen_g7():
    """PositionEmbeddingSine of the reference (models/position_encoding.py:20-63) with build_position_encoding's
    arguments (:93-99: hidden_dim // 3 features, normalize=True) on an unpadded and on a padded batch."""
    import_reference()
    from models.position_encoding import PositionEmbeddingSine
    from util.misc import NestedTensor
    out = {}
    for name, (feats, frames, b, h, w) in {"t4_f128": (128, 4, 2, 7, 9), "t2_f16": (16, 2, 3, 5, 6)}.items():
        pe = PositionEmbeddingSine(feats, num_frames=frames, normalize=True)
        clean = torch.zeros(b * frames, h, w, dtype=torch.bool)
        padded = clean.clone()
        padded[frames:, :, w - 2:] = True           # sample 1: two padded columns, one padded row
        padded[frames:, h - 1:, :] = True
        if b > 2:
            padded[2 * frames:, :, w - 1:] = True
        for tag, mask in (("clean", clean), ("padded", padded)):
            pos = pe(NestedTensor(torch.zeros(b * frames, 3, h, w), mask))
            out[f"{name}_{tag}_mask"] = mask.numpy()
            out[f"{name}_{tag}_pos"] = pos.numpy()
        out[f"{name}_cfg"] = np.array([feats, frames])
    np.savez_compressed(os.path.join(OUT, "g7_posenc.npz"), **out)


def gen_g4(DeformableTransformer):
    torch.manual_seed(5)
    d_model, nhead, L = 48, 4, 3
    T, Fu, nq = 2, 1, 4
    tr = DeformableTransformer(d_model=d_model, nhead=nhead, num_encoder_layers=2, num_decoder_layers=2,
                               dim_feedforward=32, dropout=0.0, activation="relu",
                               return_intermediate_dec=True, num_feature_levels=L, dec_n_points=4,
                               enc_n_points=4, n_frame=T, n_future_frame=Fu, use_pytroch_deform=True,
                               num_keypoints=3)
    g = torch.Generator().manual_seed(99)
    with torch.no_grad():
        tr.temporal_embed.copy_(torch.randn(tr.temporal_embed.shape, generator=g))   # :52 is uninitialised
    randomise_(tr, g)
    tr = tr.double()
    bs = 2
    hw = [(6, 8), (3, 4), (2, 2)]
    srcs = [torch.randn(bs, d_model, T, h, w, generator=g).double() for h, w in hw]
    masks = []
    for h, w in hw:   # right/bottom padding on sample 1 -> valid_ratios != 1
        m = torch.zeros(bs, d_model, T, h, w, dtype=torch.bool)
        m[1, :, :, :, w - max(1, w // 4):] = True
        m[1, :, :, h - max(1, h // 3):, :] = True
        masks.append(m)
    pos = [torch.randn(bs, d_model, T, h, w, generator=g).double() for h, w in hw]
    query_embed = torch.randn(nq * (T + Fu), 2 * d_model, generator=g).double()
    hs, heatmaps, init_ref, inter_refs, att = tr(srcs, masks, pos, query_embed)
    loss = (hs * torch.linspace(-1, 1, hs.numel(), dtype=torch.float64).view_as(hs)).sum()
    names = [k for k, _ in tr.named_parameters()]
    grads = torch.autograd.grad(loss, list(tr.parameters()), allow_unused=True)
    torch.save({
        "cfg": dict(d_model=d_model, nhead=nhead, num_encoder_layers=2, num_decoder_layers=2,
                    dim_feedforward=32, dropout=0.0, num_feature_levels=L, dec_n_points=4, enc_n_points=4,
                    n_frame=T, n_future_frame=Fu, num_keypoints=3),
        "state_dict": {k: v.detach().clone() for k, v in tr.state_dict().items()},
        "srcs": srcs, "masks": masks, "pos": pos, "query_embed": query_embed,
        "hs": hs.detach(), "heatmaps": [h.detach().clone() for h in heatmaps], "init_ref": init_ref.detach(),
        "inter_refs": inter_refs.detach(), "loss": loss.detach(),
        "att_loc": [[t.clone() for t in a[0]] for a in att], "att_w": [[t.clone() for t in a[1]] for a in att],
        "param_grads": {k: (g_ if g_ is not None else None) for k, g_ in zip(names, grads)},
    }, os.path.join(OUT, "g4_transformer.pt"))


def _blur_like_torchvision(img, kernel_size, sigma=None):
    """Stand-in for torchvision.transforms.functional.gaussian_blur (torchvision is not installed and not
    vendored by the reference): its documented algorithm -- sigma = 0.3*((k-1)*0.5-1)+0.8, reflect padding,
    separable Gaussian.  The blur itself is therefore NOT pinned by these fixtures; everything around it is."""
    k = kernel_size[0] if isinstance(kernel_size, (list, tuple)) else kernel_size
    if k <= 1:
        return img
    sig = 0.3 * ((k - 1) * 0.5 - 1) + 0.8
    x = torch.linspace(-(k - 1) * 0.5, (k - 1) * 0.5, k, dtype=img.dtype)
    k1 = torch.exp(-0.5 * (x / sig) ** 2)
    k1 = k1 / k1.sum()
    k2 = (k1[:, None] * k1[None, :])[None, None]
    shp = img.shape
    flat = F.pad(img.reshape(-1, 1, shp[-2], shp[-1]), [k // 2] * 4, mode="reflect")
    return F.conv2d(flat, k2).reshape(shp)


def import_reference_criterion():
    """models/model.py imports torchvision.transforms.functional, the backbone (torchvision.models) and
    datasets.hybrid_dataloader (cv2): none of them is needed by SetCriterion / HungarianMatcher themselves, so
    they are stubbed; ROOTJOINTCONT is read out of the reference's file."""
    import re
    import numpy as np
    import_reference()
    tv = sys.modules["torchvision"]
    tv.transforms = types.ModuleType("torchvision.transforms")
    tv.transforms.functional = types.ModuleType("torchvision.transforms.functional")
    tv.transforms.functional.gaussian_blur = _blur_like_torchvision
    tv.models = types.ModuleType("torchvision.models")
    tv.models._utils = types.ModuleType("torchvision.models._utils")
    tv.models._utils.IntermediateLayerGetter = object
    sys.modules["torchvision.transforms"] = tv.transforms
    sys.modules["torchvision.transforms.functional"] = tv.transforms.functional
    sys.modules["torchvision.models"] = tv.models
    sys.modules["torchvision.models._utils"] = tv.models._utils
    text = open(os.path.join(REF, "datasets", "hybrid_dataloader.py")).read()
    vals = re.search(r"ROOTJOINTCONT = np.array\(\[(.*?)\]\)", text).group(1)
    ds = types.ModuleType("datasets")
    ds.hybrid_dataloader = types.ModuleType("datasets.hybrid_dataloader")
    ds.hybrid_dataloader.ROOTJOINTCONT = np.array([float(v) for v in vals.split(",")])
    sys.modules["datasets"] = ds
    sys.modules["datasets.hybrid_dataloader"] = ds.hybrid_dataloader
    import models.matcher as ref_matcher
    import models.model as ref_model

    class _NoListEq(np.ndarray):         # matcher.py:134 compares the LSAP result with [] (breaks on numpy >= 1.25)
        def __eq__(self, other):
            return False if isinstance(other, list) else np.ndarray.__eq__(self, other)
    lsa = ref_matcher.linear_sum_assignment
    ref_matcher.linear_sum_assignment = lambda c: tuple(np.asarray(a).view(_NoListEq) for a in lsa(c))
    return ref_model, ref_matcher


def gen_g5():
    ref_model, ref_matcher = import_reference_criterion()
    g = torch.Generator().manual_seed(77)
    n_dec, bs, nq, T, F_, K = 3, 2, 7, 2, 1, 15
    Ta = T + F_
    matcher = ref_matcher.HungarianMatcher(cost_is_human=1, cost_root=5, cost_root_vis=0.1, cost_joint=5,
                                           cost_joint_vis=0.1, cost_joint_depth=5, cost_root_depth=5)
    weight = {"loss_is_human": 1, "loss_root": 1, "loss_root_vis": 0.1, "loss_root_depth": 1, "loss_joint_disp": 1,
              "loss_joint_depth_disp": 1, "loss_joint": 1, "loss_joint_vis": 1, "loss_joint_depth": 1, "loss_cont": 0.1,
              "loss_heatmap": 0.01}
    cont = torch.from_numpy(sys.modules["datasets.hybrid_dataloader"].ROOTJOINTCONT).float()[None, None, :, None]
    crit = ref_model.SetCriterion(matcher, ["is_human", "root", "joint", "joint_disp", "joint_cont", "heatmap"],
                                  0.5, weight, cont)
    r = lambda *s: torch.rand(*s, generator=g)
    layers = [{"pred_logits": torch.randn(bs, nq, Ta, 2, generator=g), "pred_kpts2d": r(bs, nq, Ta, K, 3),
               "pred_depth": r(bs, nq, Ta, K, 1)} for _ in range(n_dec)]
    hw = [(20, 30), (10, 15), (5, 8)]
    heat = [torch.randn(bs, T, h, w, 4, K, generator=g) * 0.1 for h, w in hw]
    outputs = dict(layers[-1], heatmaps=heat, aux_outputs=layers[:-1])
    targets = []
    for m in (3, 1):
        k2 = r(m, Ta, K, 3) * 1.2 - 0.1
        k2[..., 2] = (torch.rand(m, Ta, K, generator=g) < 0.8)
        d = r(m, Ta, K, 2)
        d[..., 1] = (torch.rand(m, Ta, K, generator=g) < 0.7)
        targets.append({"kpts2d": k2, "depth": d, "traj_ids": torch.arange(m), "max_depth": torch.tensor(15.0)})
    for o in layers:
        for v in o.values():
            v.requires_grad_(True)
    for h in heat:
        h.requires_grad_(True)
    losses, indices = crit(outputs, targets)
    total = sum(losses[k] * weight[k.rsplit("_", 1)[0] if k[-1].isdigit() else k] for k in losses)
    leaves = [v for o in layers for v in o.values()] + heat
    grads = torch.autograd.grad(total, leaves, allow_unused=True)
    torch.save({
        "layers": [{k: v.detach() for k, v in o.items()} for o in layers], "heatmaps": [h.detach() for h in heat],
        "targets": targets, "weight": weight, "losses": {k: v.detach() for k, v in losses.items()},
        "indices": [(a.clone(), b.clone()) for a, b in indices], "total": total.detach(),
        "grads": [None if x is None else x.detach() for x in grads],
        "matcher_costs": dict(cost_is_human=1, cost_root=5, cost_root_vis=0.1, cost_joint=5, cost_joint_vis=0.1,
                              cost_joint_depth=5, cost_root_depth=5),
    }, os.path.join(OUT, "g5_criterion.pt"))


def gen_g6(DeformableTransformer):
    """The reference SnipperDeformable (models/model.py:45-237) itself on top of a stand-in backbone that replays stored
    feature maps / masks / position encodings: pins the model assembly -- input projections, the [b*t,c,h,w] ->
    [b,c,t,h,w] reshapes, query embedding, the shared prediction heads incl. the reference-point offset of the root
    joint, aux outputs -- and the full state_dict key schema with its aliases (class_embed.N, root_embed.N,
    joint_embed.N.K and their transformer.decoder.* twins)."""
    ref_model, _ = import_reference_criterion()
    from util.misc import NestedTensor
    torch.manual_seed(11)
    g = torch.Generator().manual_seed(123)
    d_model, nhead, L, T, Fu, nq, K = 96, 4, 3, 2, 1, 5, 15
    hw, chans, bs = [(12, 16), (6, 8), (3, 4)], [16, 32, 64], 2

    feats = [torch.randn(bs * T, c, h, w, generator=g) for c, (h, w) in zip(chans, hw)]
    masks = []
    for h, w in hw:
        m = torch.zeros(bs * T, h, w, dtype=torch.bool)
        m[T:, :, w - max(1, w // 4):] = True            # sample 1 is padded on the right / bottom
        m[T:, h - max(1, h // 3):, :] = True
        masks.append(m)
    pos = [torch.randn(bs * T, d_model, h, w, generator=g) for h, w in hw]

    class Replay(torch.nn.Module):
        strides, num_channels = [8, 16, 32], chans

        def forward(self, samples):
            return [NestedTensor(f, m) for f, m in zip(feats, masks)], [p.clone() for p in pos]

    tr = DeformableTransformer(d_model=d_model, nhead=nhead, num_encoder_layers=1, num_decoder_layers=2,
                               dim_feedforward=64, dropout=0.0, activation="relu", return_intermediate_dec=True,
                               num_feature_levels=L, dec_n_points=4, enc_n_points=4, n_frame=T, n_future_frame=Fu,
                               use_pytroch_deform=True, num_keypoints=K)
    model = ref_model.SnipperDeformable(Replay(), tr, num_queries=nq, num_feature_levels=L, num_frames=T,
                                        num_future_frames=Fu, num_keypoints=K, aux_loss=True)
    with torch.no_grad():
        tr.temporal_embed.copy_(torch.randn(tr.temporal_embed.shape, generator=g))
    randomise_(model, g)
    model.eval()
    samples = NestedTensor(torch.zeros(bs * T, 3, 96, 128), torch.zeros(bs * T, 96, 128, dtype=torch.bool))
    out, (init_ref, inter_refs, _) = model(samples)
    sd = model.state_dict()
    ptr = {}
    for k, v in sd.items():
        ptr.setdefault(v.data_ptr(), []).append(k)
    torch.save({
        "cfg": dict(d_model=d_model, nhead=nhead, num_encoder_layers=1, num_decoder_layers=2, dim_feedforward=64,
                    dropout=0.0, num_feature_levels=L, dec_n_points=4, enc_n_points=4, n_frame=T, n_future_frame=Fu,
                    num_keypoints=K),
        "num_queries": nq, "chans": chans, "hw": hw, "bs": bs,
        "feats": feats, "masks": masks, "pos": pos,
        "state_dict": {k: v.detach().clone() for k, v in sd.items()},
        "aliases": sorted(sorted(v) for v in ptr.values() if len(v) > 1),
        "pred_logits": out["pred_logits"].detach(), "pred_kpts2d": out["pred_kpts2d"].detach(),
        "pred_depth": out["pred_depth"].detach(), "heatmaps": [h.detach().clone() for h in out["heatmaps"]],
        "aux": [{k: v.detach() for k, v in a.items()} for a in out["aux_outputs"]],
        "init_ref": init_ref.detach(), "inter_refs": inter_refs.detach(),
    }, os.path.join(OUT, "g6_model.pt"))


if __name__ == "__main__":
    core, MSDeformAttn, DeformableTransformer = import_reference()
    only = set(sys.argv[1:])          # e.g. `gen_golden.py g3d48 g7` regenerates just those
    todo = [("g1", lambda: gen_g1(core)), ("g2", lambda: gen_g2(core)), ("g3", lambda: gen_g3(MSDeformAttn)),
            ("g3d48", lambda: gen_g3_d48(MSDeformAttn)), ("g3t1", lambda: gen_g3_t1_d48(MSDeformAttn)),
            ("g3untied", lambda: gen_g3_untied_d48(MSDeformAttn)),
            ("g4", lambda: gen_g4(DeformableTransformer)),
            ("g7", gen_g7), ("g5", gen_g5), ("g6", lambda: gen_g6(DeformableTransformer))]
    for tag, fn in todo:
        if not only or tag in only:
            fn()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))
