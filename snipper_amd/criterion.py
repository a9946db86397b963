"""Training criterion of Snipper: Hungarian matching + the six loss families, every decoder layer at once.

API mirror of ``HungarianMatcher`` (/root/reference/models/matcher.py:10-141) and ``SetCriterion``
(/root/reference/models/model.py:240-545): same constructor arguments, same loss names (``loss_is_human``,
``loss_root``, ``loss_root_depth``, ``loss_root_vis``, ``loss_joint_disp``, ``loss_joint_depth_disp``,
``loss_joint``, ``loss_joint_depth``, ``loss_joint_vis``, ``loss_cont``, ``loss_heatmap`` and their ``_{i}``
auxiliary copies), same values (tests/test_criterion.py checks them against golden vectors produced by the
reference's own classes).

What is different is the evaluation order.  The reference matches and scores the last decoder layer, then
loops over the auxiliary layers, each with its own cost matrices, its own ``cost.cpu()`` round trip
(matcher.py:132) and ~100 small kernels; ``num_traj`` is read back with ``.item()`` (model.py:526).  Here all
``n_dec`` layers are stacked: one cost tensor [n_dec, n_query, m] per sample, ONE device-to-host copy for the
whole step, SciPy's LSAP on the host for the n_dec x batch small matrices, and every loss evaluated over the
layer axis in one go.  Each layer matches exactly min(n_query, m_i) pairs per sample, so the gathered tensors
are regular [n_dec, sum_i m_i, ...] arrays.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
from ._autograd import Function as _Fn
import torch.nn.functional as F
from scipy.optimize import linear_sum_assignment
from torch import nn

# temporal-continuity weight per joint (reference datasets/hybrid_dataloader.py:20)
ROOTJOINTCONT = (0, 0.2, 0.8, 0.8, 0.8, 0.2, 0.2, 0.1, 0.1, 0.8, 0.8, 0.2, 0.2, 0.1, 0.1)

_EPS = 10e-6   # the reference's "eps" (matcher.py:33, model.py:264)


def _stack_layers(outputs: dict) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """(logits [n_dec,bs,nq,T,2], kpts2d [n_dec,bs,nq,T,K,3], depth [n_dec,bs,nq,T,K,1]) with the MAIN output last
    (reference order: aux_outputs[0..n_dec-2] are decoder layers 0..n_dec-2, the main output is the last layer)."""
    al = outputs.get("all_layers")
    if al is not None:
        k = al["pred_kpts"]
        return al["pred_logits"], k[..., 0:3], k[..., 3:4]
    layers = list(outputs.get("aux_outputs", [])) + [outputs]
    return (torch.stack([o["pred_logits"] for o in layers]), torch.stack([o["pred_kpts2d"] for o in layers]),
            torch.stack([o["pred_depth"] for o in layers]))


class HungarianMatcher(nn.Module):
    def __init__(self, cost_is_human: float = 1, cost_root: float = 1, cost_root_vis: float = 1,
                 cost_joint: float = 1, cost_joint_vis: float = 1, cost_joint_depth: float = 1,
                 cost_root_depth: float = 1):
        super().__init__()
        self.cost_is_human, self.cost_root, self.cost_root_vis = cost_is_human, cost_root, cost_root_vis
        self.cost_joint, self.cost_joint_vis = cost_joint, cost_joint_vis
        self.cost_joint_depth, self.cost_root_depth = cost_joint_depth, cost_root_depth
        self.eps = _EPS
        self.device_lsap = True     # HIP assignment kernel for CUDA inputs (n_query <= 64); SciPy otherwise
        self.device_cost = True     # HIP cost-matrix kernel for CUDA inputs; the PyTorch formulation otherwise

    @torch.no_grad()
    def cost_matrices(self, logits, kpts2d, depth, targets) -> List[torch.Tensor]:
        """Per sample i: cost [n_dec, n_query, m_i] (the seven terms of matcher.py:86-130)."""
        if logits.is_cuda and self.device_cost and all(t["kpts2d"].shape[0] > 0 for t in targets):
            got = self._cost_matrices_device(logits, kpts2d, depth, targets)
            if got is not None:
                return got
        costs = []
        for i, tgt in enumerate(targets):
            tk = tgt["kpts2d"][None, None]                       # 1 x 1 x m x T x K x 3
            td = tgt["depth"][None, None]                        # 1 x 1 x m x T x K x 2
            max_depth = tgt["max_depth"]
            ok = kpts2d[:, i].unsqueeze(2)                       # n_dec x nq x 1 x T x K x 3
            od = depth[:, i].unsqueeze(2)                        # n_dec x nq x 1 x T x K x 1
            prob = logits[:, i].softmax(-1)[..., 1].unsqueeze(2)  # n_dec x nq x 1 x T

            t_root, t_joint, j_vis = tk[..., :1, :], tk[..., 1:, 0:2], tk[..., 1:, 2:3]
            td_root, td_root_ok = td[..., :1, 0:1], td[..., :1, 1:2]
            td_joint, td_joint_ok = td[..., 1:, 0:1], td[..., 1:, 1:2]
            o_root, o_joint_raw = ok[..., :1, :], ok[..., 1:, :]
            o_joint = o_joint_raw[..., 0:2] + o_root[..., 0:2]                       # root + displacement
            od_root = od[..., :1, :]
            od_joint = od_root + od[..., 1:, :] / max_depth

            vis = (j_vis.sum((-2, -1)) > 0).float()                                   # 1 x 1 x m x T
            c_class = -(prob * vis).sum(-1) / (vis.sum(-1) + self.eps)
            c_joint = (j_vis * (o_joint - t_joint)).abs().sum((-1, -2, -3)) / (j_vis.sum((-1, -2, -3)) + self.eps)
            c_joint_vis = (o_joint_raw[..., 2:3] - j_vis).pow(2).mean((-1, -2, -3))
            c_joint_depth = (td_joint_ok * (od_joint - td_joint)).abs().sum((-1, -2, -3)) / \
                (td_joint_ok.sum((-1, -2, -3)) + self.eps)
            r_vis = t_root[..., 2:3]
            c_root = (r_vis * (o_root[..., 0:2] - t_root[..., 0:2])).abs().sum((-1, -2, -3)) / \
                (r_vis.sum((-1, -2, -3)) + self.eps)
            c_root_vis = (o_root[..., 2:3] - r_vis).pow(2).mean((-1, -2, -3))
            c_root_depth = (td_root_ok * (od_root - td_root)).abs().sum((-1, -2, -3)) / \
                (td_root_ok.sum((-1, -2, -3)) + self.eps)
            costs.append(self.cost_is_human * c_class + self.cost_root * c_root + self.cost_root_vis * c_root_vis +
                         self.cost_root_depth * c_root_depth + self.cost_joint * c_joint +
                         self.cost_joint_vis * c_joint_vis + self.cost_joint_depth * c_joint_depth)
        return costs

    def _cost_matrices_device(self, logits, kpts2d, depth, targets):
        """The same matrices from csrc/match_cost.cuh: one launch per sample instead of ~60.  None when the layouts are
        not the ones the kernel addresses (then the formulation above runs)."""
        import ctypes
        from . import _lib
        n_dec, bs, nq, T, K = kpts2d.shape[:5]

        def rows(t, width):          # [n_dec, bs, nq, T, K, width] float32 with dense-compatible (T, K, width) axes
            if t.dtype != torch.float32 or t.stride(-1) != 1:
                t = t.float().contiguous()
            sk = t.stride(-2)
            if t.stride(-3) != K * sk or sk < width:
                t = t.contiguous()
                sk = t.stride(-2)
            return t, sk

        kp, kp_sk = rows(kpts2d, 3)
        dp, d_sk = rows(depth, 1)
        lg = logits if (logits.dtype == torch.float32 and logits[0, 0].is_contiguous()) else logits.float().contiguous()
        if kp.stride(-4) != T * K * kp_sk or dp.stride(-4) != T * K * d_sk:
            return None
        w7 = (ctypes.c_float * 7)(self.cost_is_human, self.cost_root, self.cost_root_vis, self.cost_root_depth,
                                  self.cost_joint, self.cost_joint_vis, self.cost_joint_depth)
        lib, dev, costs = _lib.load(), logits.device, []
        for i, tgt in enumerate(targets):
            tk, td, md = tgt["kpts2d"], tgt["depth"], tgt["max_depth"]
            if not (torch.is_tensor(md) and md.is_cuda):          # a host scalar would cost a blocking copy per call
                return None
            tk = tk.float().contiguous()
            td = td.float().contiguous()
            md = md.reshape(-1)[:1].float().contiguous()
            m = tk.shape[0]
            out = torch.empty((n_dec, nq, m), dtype=torch.float32, device=dev)
            ki, di, li = kp[:, i], dp[:, i], lg[:, i]
            with _lib.device_guard(dev):
                rc = lib.snipper_match_cost_f32(
                    _lib.raw_stream(dev), ki.data_ptr(), ki.stride(0), ki.stride(1), kp_sk, di.data_ptr(), di.stride(0),
                    di.stride(1), d_sk, li.data_ptr(), li.stride(0), li.stride(1), tk.data_ptr(), td.data_ptr(),
                    md.data_ptr(), n_dec, nq, m, T, K, ctypes.cast(w7, ctypes.c_void_p), float(self.eps), out.data_ptr())
            _lib.check(rc, "snipper_match_cost_f32")
            costs.append(out)
        return costs

    @torch.no_grad()
    def match_all_layers(self, logits, kpts2d, depth, targets):
        """-> (src [n_dec, Msum], batch [Msum], tgt [n_dec, Msum], offsets) as device long tensors, where the pairs
        of sample i occupy columns offsets[i]:offsets[i+1].  One device-to-host copy for everything."""
        costs = self.cost_matrices(logits.float(), kpts2d.float(), depth.float(), targets)
        n_dec, nq = logits.shape[0], logits.shape[2]
        if logits.is_cuda and self.device_lsap and nq <= 64 and all(c.shape[2] <= nq for c in costs):
            return self._match_on_device(costs, n_dec, nq)
        flat = torch.cat([c.reshape(-1) for c in costs]).cpu().numpy()      # the step's only matcher sync
        src_cols, tgt_cols, batch_cols, offsets, pos = [], [], [], [0], 0
        for i, c in enumerate(costs):
            m = c.shape[2]
            mat = flat[pos:pos + n_dec * nq * m].reshape(n_dec, nq, m)
            pos += n_dec * nq * m
            k = min(nq, m)
            s = np.zeros((n_dec, k), dtype=np.int64)
            t = np.zeros((n_dec, k), dtype=np.int64)
            for l in range(n_dec):
                if k:
                    s[l], t[l] = linear_sum_assignment(mat[l])                # rows ascending, as the reference
            src_cols.append(s)
            tgt_cols.append(t)
            batch_cols.append(np.full((k,), i, dtype=np.int64))
            offsets.append(offsets[-1] + k)
        dev = logits.device
        src = torch.from_numpy(np.concatenate(src_cols, 1)).to(dev, non_blocking=True)
        tgt = torch.from_numpy(np.concatenate(tgt_cols, 1)).to(dev, non_blocking=True)
        batch = torch.from_numpy(np.concatenate(batch_cols)).to(dev, non_blocking=True)
        return src, batch, tgt, offsets

    def _match_on_device(self, costs, n_dec, nq):
        """One wave per (layer, sample) problem (csrc/lsap.cuh): no device-to-host copy at all."""
        from . import _lib
        lib = _lib.load()
        dev = costs[0].device
        srcs, tgts, batch, offsets = [], [], [], [0]
        for i, c in enumerate(costs):
            m = c.shape[2]
            s = torch.empty((n_dec, m), dtype=torch.long, device=dev)
            t = torch.empty((n_dec, m), dtype=torch.long, device=dev)
            if m:
                c = c.contiguous()
                with _lib.device_guard(dev):
                    rc = lib.snipper_lsap_f32(_lib.raw_stream(dev), c.data_ptr(), n_dec, nq, m,
                                              s.data_ptr(), t.data_ptr())
                _lib.check(rc, "snipper_lsap_f32")
            srcs.append(s)
            tgts.append(t)
            batch.append(torch.full((m,), i, dtype=torch.long, device=dev))
            offsets.append(offsets[-1] + m)
        return torch.cat(srcs, 1), torch.cat(batch), torch.cat(tgts, 1), offsets

    @torch.no_grad()
    def forward(self, outputs, targets):
        """The reference's call: indices of ONE layer's outputs, a list of (index_i, index_j) per sample."""
        src, batch, tgt, offsets = self.match_all_layers(outputs["pred_logits"][None], outputs["pred_kpts2d"][None],
                                                         outputs["pred_depth"][None], targets)
        return [(src[0, a:b], tgt[0, a:b]) for a, b in zip(offsets[:-1], offsets[1:])]


_blur_kernels: Dict = {}


def gaussian_blur(img: torch.Tensor, kernel_size: int, clamp_max: Optional[float] = None) -> torch.Tensor:
    """torchvision.transforms.functional.gaussian_blur(img, [k, k]) restated (sigma = 0.3*((k-1)*0.5-1)+0.8,
    reflect padding, separable kernel).  torchvision is not vendored by the reference: parity unpinned.
    ``clamp_max``: blur ``img.clamp(max=clamp_max)`` (folded into the HIP kernel on the GPU).  The HIP kernel is a
    forward-only fast path (the heat-map TARGETS need no gradient): an input that autograd tracks takes the differentiable
    tensor formulation below."""
    if (img.is_cuda and img.dtype == torch.float32 and img.dim() >= 2 and 1 < kernel_size <= 31 and kernel_size % 2 == 1
            and kernel_size // 2 < min(img.shape[-2:]) and img.is_contiguous()
            and not (img.requires_grad and torch.is_grad_enabled())):
        # one launch of csrc/heatmap_blur.cuh instead of a padding kernel and two library convolutions
        import ctypes
        from . import _lib
        key = ("taps", kernel_size)
        taps = _blur_kernels.get(key)
        if taps is None:                            # the same float32 taps as the tensor formulation below
            sigma = 0.3 * ((kernel_size - 1) * 0.5 - 1) + 0.8
            half = (kernel_size - 1) * 0.5
            xs = torch.linspace(-half, half, kernel_size, dtype=torch.float32)
            k1 = torch.exp(-0.5 * (xs / sigma) ** 2)
            k1 = k1 / k1.sum()
            taps = _blur_kernels[key] = (ctypes.c_float * kernel_size)(*k1.tolist())
        out = torch.empty_like(img)
        h, w = img.shape[-2:]
        with _lib.device_guard(img.device):
            rc = _lib.load().snipper_heatmap_blur_f32(
                _lib.raw_stream(img.device), img.data_ptr(), out.data_ptr(), img.numel() // (h * w), h, w, kernel_size,
                ctypes.cast(taps, ctypes.c_void_p), float("inf") if clamp_max is None else float(clamp_max))
        _lib.check(rc, "snipper_heatmap_blur_f32")
        return out
    if clamp_max is not None:
        img = img.clamp(max=clamp_max)
    if kernel_size <= 1:
        return img
    key = (kernel_size, str(img.device), img.dtype)
    ks = _blur_kernels.get(key)
    if ks is None:                                  # constants of (size, device, dtype): built once
        sigma = 0.3 * ((kernel_size - 1) * 0.5 - 1) + 0.8
        half = (kernel_size - 1) * 0.5
        x = torch.linspace(-half, half, kernel_size, device=img.device, dtype=img.dtype)
        k1 = torch.exp(-0.5 * (x / sigma) ** 2)
        k1 = k1 / k1.sum()
        ks = _blur_kernels[key] = ((k1[:, None] * k1[None, :])[None, None], k1.view(1, 1, -1, 1).contiguous(),
                                   k1.view(1, 1, 1, -1).contiguous())
    k2, kv, kh = ks
    shape = img.shape
    flat = img.reshape(-1, 1, shape[-2], shape[-1])
    pad = kernel_size // 2
    flat = F.pad(flat, [pad, pad, pad, pad], mode="reflect")
    if img.is_cuda:
        # the kernel is the outer product k1 k1^T: two 1-D passes (2k taps instead of k^2; the k x k float32 convolution
        # was 88 us of MIOpen time on the finest level) -- equal to the 2-D form up to float32 rounding of the products
        return F.conv2d(F.conv2d(flat, kv), kh).reshape(shape)
    return F.conv2d(flat, k2).reshape(shape)


_PAIR_TERMS = ("loss_root", "loss_root_depth", "loss_root_vis", "loss_joint_disp", "loss_joint_depth_disp",
               "loss_joint", "loss_joint_depth", "loss_joint_vis", "loss_cont")


class PairLosses(_Fn):
    """The nine keypoint / depth / continuity terms of every matched pair of every decoder layer in one launch each way
    (csrc/pair_losses.cuh; the PyTorch formulation below it in ``_all_losses`` is the specification and the CPU path).
    sk / sd / tk / td as in ``_all_losses``; -> [n_dec, Msum, 9] per-pair terms in ``_PAIR_TERMS`` order."""

    @staticmethod
    def forward(ctx, sk, sd, tk, td, cont_w, max_depth, eps):
        from . import _lib
        sk, sd, tk, td = (t.contiguous().float() for t in (sk, sd, tk, td))
        n_dec, ms, T, K = sk.shape[:4]
        out = torch.zeros((n_dec, ms, len(_PAIR_TERMS)), dtype=torch.float32, device=sk.device)
        cw = cont_w.reshape(-1).contiguous().float()
        md = max_depth.reshape(-1)[:1].to(device=sk.device, dtype=torch.float32).contiguous()
        with _lib.device_guard(sk.device):
            rc = _lib.load().snipper_pair_losses_forward(
                _lib.raw_stream(sk.device), sk.data_ptr(), sd.data_ptr(), tk.data_ptr(),
                td.data_ptr(), cw.data_ptr(), md.data_ptr(), n_dec, ms, T, K, float(eps), out.data_ptr())
        _lib.check(rc, "snipper_pair_losses_forward")
        ctx.save_for_backward(sk, sd, tk, td, cw, md)
        ctx.eps = float(eps)
        return out

    @staticmethod
    def backward(ctx, g):
        from . import _lib
        sk, sd, tk, td, cw, md = ctx.saved_tensors
        n_dec, ms, T, K = sk.shape[:4]
        gl = g.contiguous().float()                               # [n_dec, Msum, 9]
        dsk, dsd = torch.zeros_like(sk), torch.zeros_like(sd)
        with _lib.device_guard(sk.device):
            rc = _lib.load().snipper_pair_losses_backward(
                _lib.raw_stream(sk.device), sk.data_ptr(), sd.data_ptr(), tk.data_ptr(),
                td.data_ptr(), cw.data_ptr(), md.data_ptr(), gl.data_ptr(), n_dec, ms, T, K, ctx.eps,
                dsk.data_ptr(), dsd.data_ptr())
        _lib.check(rc, "snipper_pair_losses_backward")
        return dsk, dsd, None, None, None, None, None


class HeatmapLoss(_Fn):
    """sum over levels of mse_loss(target, heat-map views, reduction="sum") / n_heads as ONE node on the encoder memory the
    views are views of (csrc/heatmap_loss.cuh): one forward launch over all levels (deterministic block sums), one backward
    launch that writes the whole gradient of the memory.  Reference: models/model.py:447-483."""

    @staticmethod
    def try_apply(heatmaps, tmaps):
        src = getattr(heatmaps, "source", None)
        if src is None or not tmaps:
            return None
        memory, hw, nhead, K = src
        if not (memory.is_cuda and memory.dtype == torch.float32 and memory.is_contiguous() and memory.dim() == 4 and
                len(hw) == len(tmaps) <= 4 and memory.shape[-1] % nhead == 0 and (memory.shape[-1] // nhead) % 4 == 0):
            return None
        bs, T, S, C = memory.shape
        raws = []
        for (h, w), tm, hm in zip(hw, tmaps, heatmaps):
            raw = tm.permute(0, 4, 1, 2, 3)                 # back to the blur's [bs, K, t, h, w]
            if not (raw.is_contiguous() and raw.dtype == torch.float32 and raw.shape == (bs, K, T, h, w) and
                    not raw.requires_grad and tuple(hm.shape) == (bs, T, h, w, nhead, K)):
                return None
            raws.append(raw)
        if sum(h * w for h, w in hw) != S:
            return None
        return HeatmapLoss.apply(memory, nhead, K, tuple(hw), *raws)

    @staticmethod
    def _call(fn_name, memory, nhead, K, hw, raws, *tail):
        import ctypes
        from . import _lib
        bs, T, S, C = memory.shape
        nl = len(hw)
        tm = (ctypes.c_void_p * nl)(*[r.data_ptr() for r in raws])
        px = (ctypes.c_int * nl)(*[h * w for h, w in hw])
        starts, pos = [], 0
        for h, w in hw:
            starts.append(pos)
            pos += h * w
        st = (ctypes.c_int * nl)(*starts)
        with _lib.device_guard(memory.device):
            rc = getattr(_lib.load(), fn_name)(_lib.raw_stream(memory.device), memory.data_ptr(), tm, px, st, nl, bs, T, S, C,
                                              nhead, K, *tail)
        _lib.check(rc, fn_name)

    @staticmethod
    def forward(ctx, memory, nhead, K, hw, *raws):
        n_partial = 1024
        partial = torch.empty(n_partial, dtype=torch.float32, device=memory.device)
        HeatmapLoss._call("snipper_heatmap_loss_forward_f32", memory, nhead, K, hw, raws, partial.data_ptr(), n_partial)
        ctx.save_for_backward(memory, *raws)
        ctx.geom = (nhead, K, hw)
        return partial.sum() / nhead

    @staticmethod
    def backward(ctx, g):
        memory, *raws = ctx.saved_tensors
        nhead, K, hw = ctx.geom
        gs = (g.float() / nhead).contiguous()               # device scalar: d loss / d (sum of squared differences)
        gmem = torch.empty_like(memory)
        HeatmapLoss._call("snipper_heatmap_loss_backward_f32", memory, nhead, K, hw, raws, gs.data_ptr(), gmem.data_ptr())
        return (gmem, None, None, None) + (None,) * len(raws)


class SetCriterion(nn.Module):
    def __init__(self, matcher, losses, eos_coef, weight_dict, cont_weights=None):
        super().__init__()
        self.matcher, self.losses, self.eos_coef, self.weight_dict = matcher, losses, eos_coef, weight_dict
        empty_weight = torch.ones(2)
        empty_weight[0] = eos_coef
        self.register_buffer("empty_weight", empty_weight)
        self.eps = _EPS
        if cont_weights is None:
            cont_weights = torch.tensor(ROOTJOINTCONT).float()[None, None, :, None]
        self.register_buffer("cont_weights", cont_weights)          # [1, 1, K, 1]
        self.fused_pair_losses = True                                # HIP kernel for the pair terms on CUDA (else PyTorch)

    # ---- the losses, over [n_dec, Msum, ...] gathered pairs ------------------------------------------
    def _all_losses(self, logits, sk, sd, tk, td, src, batch, max_depth, num_traj) -> Dict[str, torch.Tensor]:
        """sk / sd: matched predictions [n_dec, Msum, T, K, 3] / [.., 1]; tk / td: their targets [.., 3] / [.., 2].
        Every entry of the result is a [n_dec] tensor (one value per decoder layer)."""
        eps, out = self.eps, {}
        n_dec, bs, nq, T = logits.shape[:4]
        per = lambda x: x.sum(-1) / num_traj                                        # sum over pairs -> [n_dec]

        if "is_human" in self.losses:                                               # model.py:266-287
            tgt_vis = (tk[..., 2].sum(-1) > 0).long()                               # [n_dec, Msum, T]
            classes = torch.zeros(n_dec, bs, nq, T, dtype=torch.long, device=logits.device)
            lidx = torch.arange(n_dec, device=logits.device)[:, None].expand_as(src)
            classes[lidx, batch[None].expand_as(src), src] = tgt_vis
            ce = F.cross_entropy(logits.reshape(-1, 2).float(), classes.reshape(-1), self.empty_weight, reduction="none")
            out["loss_is_human"] = ce.view(n_dec, -1).mean(-1)

        if (self.fused_pair_losses and sk.is_cuda and sk.dtype == torch.float32 and sk.shape[2] * sk.shape[3] <= 128 and
                all(k in self.losses for k in ("root", "joint", "joint_disp", "joint_cont")) and torch.is_tensor(max_depth)):
            # all nine terms in one kernel; the sum over pairs and the division by the trajectory count stay here
            terms = PairLosses.apply(sk, sd, tk, td, self.cont_weights, max_depth, eps).sum(1) / num_traj   # [n_dec, 9]
            for i, name in enumerate(_PAIR_TERMS):
                out[name] = terms[:, i]
            # forward() stacks the entries as [n_names, n_dec]; this is the same tensor built with one cat instead of nine
            # selects + a stack (whose backward is a fill, a copy and an add per name)
            head = [out[k][None] for k in out if k not in _PAIR_TERMS]
            self._stacked_hint = (list(out), torch.cat(head + [terms.t()], 0))
            return out

        t_root_vis = tk[..., :1, 2:3]
        if "root" in self.losses:                                                   # model.py:289-324
            e = t_root_vis * (sk[..., :1, 0:2] - tk[..., :1, 0:2]).abs()
            out["loss_root"] = per((e.sum((-2, -3)) / (t_root_vis.sum((-2, -3)) + eps)).sum(-1))
            ok = td[..., :1, 1:2]
            e = ok * (td[..., :1, 0:1] - sd[..., :1, :]).abs()
            out["loss_root_depth"] = per((e.sum((-2, -3)) / (ok.sum((-2, -3)) + eps)).sum(-1))
            out["loss_root_vis"] = per((sk[..., :1, 2:3] - t_root_vis).pow(2).mean((-2, -3)).sum(-1))

        t_joint_vis = tk[..., 1:, 2:3]
        if "joint_disp" in self.losses:                                             # model.py:364-400
            vis = t_joint_vis * t_root_vis
            e = vis * (sk[..., 1:, 0:2] - (tk[..., 1:, 0:2] - tk[..., :1, 0:2])).abs()
            out["loss_joint_disp"] = per((e.sum((-2, -3)) / (vis.sum((-2, -3)) + eps)).sum(-1))
            ok = td[..., 1:, 1:2] * td[..., :1, 1:2]
            e = ok * (sd[..., 1:, :] - (td[..., 1:, 0:1] - td[..., :1, 0:1])).abs()
            out["loss_joint_depth_disp"] = per((e.sum((-2, -3)) / (ok.sum((-2, -3)) + eps)).sum(-1))

        if "joint" in self.losses:                                                  # model.py:326-362
            s_joint = sk[..., 1:, 0:2] + sk[..., :1, 0:2]
            e = t_joint_vis * (s_joint - tk[..., 1:, 0:2]).abs()
            out["loss_joint"] = per((e.sum((-2, -3)) / (t_joint_vis.sum((-2, -3)) + eps)).sum(-1))
            s_jd = sd[..., :1, :] + sd[..., 1:, :] / max_depth
            ok = td[..., 1:, 1:2]
            e = ok * (s_jd - td[..., 1:, 0:1]).abs()
            out["loss_joint_depth"] = per((e.sum((-2, -3)) / (ok.sum((-2, -3)) + eps)).sum(-1))
            out["loss_joint_vis"] = per((sk[..., 1:, 2:3] - t_joint_vis).pow(2).mean((-2, -3)).sum(-1))

        if "joint_cont" in self.losses:                                             # model.py:402-427
            vis = tk[..., 2:3]
            depth_abs = torch.cat([sd[..., :1, :], sd[..., :1, :] + sd[..., 1:, :] / max_depth], -2)
            kp = torch.cat([sk[..., 0:2], depth_abs], -1)                           # [.., T, K, 3]
            kp = torch.cat([kp[..., :1, :], kp[..., 1:, :] - kp[..., :1, :].detach()], -2)
            cont_vis = vis[:, :, 1:] * vis[:, :, :-1]
            e = self.cont_weights * cont_vis * (kp[:, :, 1:] - kp[:, :, :-1]).pow(2)
            out["loss_cont"] = per((e.sum((-2, -3)) / (cont_vis.sum((-2, -3)) + eps)).sum(-1))
        return out

    def heatmap_targets(self, targets, spatial, device) -> List[torch.Tensor]:
        """Gaussian-blurred one-hot joint maps per level, [bs, t, h, w, K] (model.py:447-483)."""
        maps = []
        bs = len(targets)
        counts = [int(tgt["kpts2d"].shape[0]) for tgt in targets]
        skey = (tuple(counts), str(device))
        if getattr(self, "_sample_key", None) != skey:               # (a constant of the persons per sample: built once)
            self._sample = torch.cat([torch.full((n,), i, dtype=torch.long, device=device) for i, n in enumerate(counts)])
            self._sample_key = skey
        sample = self._sample
        t_all = max(t for t, _, _ in spatial)
        k_all = torch.cat([tgt["kpts2d"][:, :t_all] for tgt in targets], 0)         # [Nsum, t, K, 3]
        K = k_all.shape[2]
        ksizes = [max(h // 10 + h // 10 % 2 - 1, w // 10 + w // 10 % 2 - 1) for _, h, w in spatial]

        def pixel_indices():
            """Pixel coordinates and validity for ALL levels in one set of launches: [levels, Nsum, t, K] (the PyTorch paths
            only: the fused scatter kernel below does this arithmetic itself)."""
            whs = torch.tensor([[w, h] for _, h, w in spatial], dtype=k_all.dtype).to(device, non_blocking=True) \
                if getattr(self, "_wh_key", None) != (tuple(spatial), str(device), k_all.dtype) else self._wh
            self._wh_key, self._wh = (tuple(spatial), str(device), k_all.dtype), whs
            xy = (k_all[None, ..., 0:2] * whs[:, None, None, None, :]).long()
            lim = whs.long()[:, None, None, None, :]
            ok_all = (k_all[None, ..., 2] > 0) & ((xy >= 0) & (xy < lim)).all(-1)
            xy = torch.minimum(xy.clamp(min=0), lim - 1)
            ti = torch.arange(t_all, device=device)[None, :, None]
            ki = torch.arange(K, device=device)[None, None, :]
            return xy, ok_all, ti, ki

        # One scatter-add over the flattened [bs, K, t, h, w] maps.  No boolean-mask indexing (it would read the count
        # back to the host) and no multi-index index_put_ (its accumulate path range-checks every index tensor with
        # separate reductions and sorts: ~50 launches per call): invalid joints add 0 at a clamped position, valid ones
        # add 1, several on one pixel still give 1 (the clamp is folded into the blur).
        if all(t == t_all for t, _, _ in spatial):
            # all levels in ONE buffer and one set of launches: level l's maps start at base[l]
            sizes = [bs * K * t_all * h * w for _, h, w in spatial]
            if (k_all.is_cuda and k_all.dtype == torch.float32 and len(spatial) <= 4 and k_all.shape[-1] == 3 and
                    not (k_all.requires_grad and torch.is_grad_enabled())):
                # ONE launch for the whole index arithmetic below (csrc/heatmap_loss.cuh, heatmap_scatter_kernel): a visible
                # joint inside the map stores 1.0 -- the blur clamps at 1, so it never sees the count of joints on a pixel
                import ctypes
                from . import _lib
                k_all = k_all.contiguous()
                hm_all = torch.zeros(sum(sizes), device=device)
                nl = len(spatial)
                hh = (ctypes.c_int * nl)(*[h for _, h, _ in spatial])
                ww = (ctypes.c_int * nl)(*[w for _, _, w in spatial])
                base = (ctypes.c_longlong * nl)(*[sum(sizes[:i]) for i in range(nl)])
                with _lib.device_guard(k_all.device):
                    rc = _lib.load().snipper_heatmap_scatter_f32(
                        _lib.raw_stream(k_all.device), k_all.data_ptr(), sample.data_ptr(), k_all.shape[0], k_all.shape[1], t_all,
                        K, nl, hh, ww, base, hm_all.data_ptr())
                _lib.check(rc, "snipper_heatmap_scatter_f32")
                off = 0
                for (t, h, w), n, ksize in zip(spatial, sizes, ksizes):
                    hm = hm_all[off:off + n].view(bs, K, t, h, w)
                    off += n
                    maps.append(gaussian_blur(hm, ksize, clamp_max=1.0).permute(0, 2, 3, 4, 1))   # [bs, t, h, w, K]
                return maps
            xy, ok_all, ti, ki = pixel_indices()
            key = ("hm_geom", tuple(spatial), bs, K, str(device))
            geom = getattr(self, "_hm_geom", None)
            if geom is None or geom[0] != key:
                hs = torch.tensor([h for _, h, _ in spatial], dtype=torch.long)
                ws = torch.tensor([w for _, _, w in spatial], dtype=torch.long)
                base = torch.tensor([sum(sizes[:i]) for i in range(len(sizes))], dtype=torch.long)
                geom = self._hm_geom = (key, hs.to(device)[:, None, None, None], ws.to(device)[:, None, None, None],
                                        base.to(device)[:, None, None, None])
            _, hs, ws, base = geom
            plane = ((sample[:, None, None] * K + ki) * t_all + ti)[None]                     # [1, Nsum, t, K]
            lin = (plane * hs + xy[..., 1]) * ws + xy[..., 0] + base                          # [levels, Nsum, t, K]
            hm_all = torch.zeros(sum(sizes), device=device)
            hm_all.index_add_(0, lin.reshape(-1), ok_all.reshape(-1).to(hm_all.dtype))
            off = 0
            for (t, h, w), n, ksize in zip(spatial, sizes, ksizes):
                hm = hm_all[off:off + n].view(bs, K, t, h, w)
                off += n
                maps.append(gaussian_blur(hm, ksize, clamp_max=1.0).permute(0, 2, 3, 4, 1))   # [bs, t, h, w, K]
            return maps
        xy, ok_all, ti, ki = pixel_indices()
        for lvl, (t, h, w) in enumerate(spatial):
            x, y, ok = xy[lvl, :, :t, :, 0], xy[lvl, :, :t, :, 1], ok_all[lvl, :, :t]
            lin = (((sample[:, None, None] * K + ki) * t + ti[:, :t]) * h + y) * w + x
            hm = torch.zeros(bs * K * t * h * w, device=device)
            hm.index_add_(0, lin.reshape(-1), ok.reshape(-1).to(hm.dtype))
            hm = hm.view(bs, K, t, h, w)
            maps.append(gaussian_blur(hm, ksizes[lvl], clamp_max=1.0).permute(0, 2, 3, 4, 1))   # [bs, t, h, w, K]
        return maps

    def loss_heatmap(self, outputs, targets):
        heatmaps = outputs["heatmaps"]                                               # [(bs, t, h, w, nhead, K')]
        tmaps = self.heatmap_targets(targets, [hm.shape[1:4] for hm in heatmaps], heatmaps[0].device)
        fused = HeatmapLoss.try_apply(heatmaps, tmaps)
        if fused is not None:
            return fused
        total = 0
        for hm, tm in zip(heatmaps, tmaps):
            nhead = hm.shape[4]
            total = total + F.mse_loss(tm.unsqueeze(4).expand(-1, -1, -1, -1, nhead, -1), hm.float(), reduction="sum") / nhead
        return total

    def forward(self, outputs, targets):
        """-> (losses dict with the reference's keys, indices of the last layer as the reference returns them)."""
        logits, kpts2d, depth = _stack_layers(outputs)
        n_dec = logits.shape[0]
        src, batch, tgt, offsets = self.matcher.match_all_layers(logits, kpts2d, depth, targets)

        # normaliser = persons per rank (model.py:521-526); built with a fill kernel -- a tensor made from a Python
        # list is a blocking host-to-device copy, and the reference's .item() a device-to-host one
        num_traj = torch.full((1,), float(sum(len(t["traj_ids"]) for t in targets)), device=logits.device)
        world = 1
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            torch.distributed.all_reduce(num_traj)
            world = torch.distributed.get_world_size()
        num_traj = torch.clamp(num_traj / world, min=1)

        # matched predictions, gathered with ONE flat index: the backward of index_select is a single index_add
        # kernel, that of three-tensor advanced indexing the sort-based accumulate path (~40 launches per gather)
        bs, nq = logits.shape[1], logits.shape[2]
        lidx = torch.arange(n_dec, device=logits.device)[:, None]
        lin = ((lidx * bs + batch[None]) * nq + src).reshape(-1)                    # [n_dec * Msum]
        al = outputs.get("all_layers")
        if al is not None:                                       # (x, y, vis, depth) live in one tensor: one gather
            kd = al["pred_kpts"].reshape(n_dec * bs * nq, *al["pred_kpts"].shape[3:]).index_select(0, lin).float()
            kd = kd.view(n_dec, -1, *kd.shape[1:])
            sk, sd = kd[..., 0:3], kd[..., 3:4]                  # [n_dec, Msum, T, K, 3] / [.., 1]
        else:
            pick = lambda t: t.reshape(n_dec * bs * nq, *t.shape[3:]).index_select(0, lin).float().view(
                n_dec, -1, *t.shape[3:])
            sk, sd = pick(kpts2d), pick(depth)
        tk = torch.cat([t["kpts2d"][tgt[:, a:b]] for t, a, b in zip(targets, offsets[:-1], offsets[1:])], 1)
        td = torch.cat([t["depth"][tgt[:, a:b]] for t, a, b in zip(targets, offsets[:-1], offsets[1:])], 1)
        self._stacked_hint = None
        per_layer = self._all_losses(logits, sk, sd, tk, td, src, batch, targets[0]["max_depth"], num_traj)

        names = list(per_layer)
        hint, self._stacked_hint = getattr(self, "_stacked_hint", None), None
        if hint is not None and hint[0] == names:
            stacked = hint[1]
        else:
            stacked = torch.stack([per_layer[n] for n in names])     # [n_names, n_dec]
        losses = {}
        for j, name in enumerate(names):                          # entries are (differentiable) views of `stacked`
            losses[name] = stacked[j, -1]                        # main output = last decoder layer
            for i in range(n_dec - 1):
                losses[f"{name}_{i}"] = stacked[j, i]
        heat = None
        if "heatmap" in self.losses:
            heat = losses["loss_heatmap"] = self.loss_heatmap(outputs, targets)
        self._fast = (losses, names, stacked, heat)              # lets weighted_sum() skip ~130 scalar kernels
        indices = [(src[-1, a:b], tgt[-1, a:b]) for a, b in zip(offsets[:-1], offsets[1:])]
        return losses, indices

    def _weight_matrix(self, names, n_dec, device) -> torch.Tensor:
        """W[j, i] = weight of loss `names[j]` at decoder layer i (last row index = the main output), 0 if unweighted."""
        key = (tuple(names), n_dec, str(device))
        cached = getattr(self, "_wm_cache", None)
        if cached is None or cached[0] != key:
            rows = [[self.weight_dict.get(f"{n}_{i}", 0.0) for i in range(n_dec - 1)] + [self.weight_dict.get(n, 0.0)]
                    for n in names]
            cached = (key, torch.tensor(rows, dtype=torch.float32, device=device))
            self._wm_cache = cached
        return cached[1]

    def weighted_sum(self, losses: Dict[str, torch.Tensor]) -> torch.Tensor:
        """engine.py:56: sum(loss_dict[k] * weight_dict[k] for k in loss_dict if k in weight_dict).  For the dictionary
        this criterion has just returned, the same sum is one product with a cached weight matrix instead of a
        multiply and an add per entry (66 entries with 6 decoder layers, each also a backward kernel)."""
        fast = getattr(self, "_fast", None)
        if fast is not None and fast[0] is losses:
            _, names, stacked, heat = fast
            total = (stacked * self._weight_matrix(names, stacked.shape[1], stacked.device)).sum()
            if heat is not None and "loss_heatmap" in self.weight_dict:
                total = total + heat * self.weight_dict["loss_heatmap"]
            return total
        return sum(losses[k] * self.weight_dict[k] for k in losses if k in self.weight_dict)


def build_matcher(args):
    if args.max_depth == -1:
        args.set_cost_root_depth = 0
        args.set_cost_joint_depth = 0
    return HungarianMatcher(cost_is_human=args.set_cost_is_human, cost_root=args.set_cost_root,
                            cost_root_vis=args.set_cost_root_vis, cost_root_depth=args.set_cost_root_depth,
                            cost_joint=args.set_cost_joint, cost_joint_vis=args.set_cost_joint_vis,
                            cost_joint_depth=args.set_cost_joint_depth)


def build_criterion(args, matcher=None):
    """The criterion half of the reference's build_model (models/model.py:635-676)."""
    matcher = matcher or build_matcher(args)
    losses = ["is_human", "root", "joint", "joint_disp", "joint_cont", "heatmap"]
    if args.max_depth == -1:
        args.root_depth_loss_coef = args.joint_disp_depth_loss_coef = args.joint_depth_loss_coef = 0
    wd = {"loss_is_human": args.is_human_loss_coef, "loss_root": args.root_loss_coef,
          "loss_root_vis": args.root_vis_loss_coef, "loss_root_depth": args.root_depth_loss_coef,
          "loss_joint_disp": args.joint_disp_loss_coef, "loss_joint_depth_disp": args.joint_disp_depth_loss_coef,
          "loss_joint": args.joint_loss_coef, "loss_joint_vis": args.joint_vis_loss_coef,
          "loss_joint_depth": args.joint_depth_loss_coef, "loss_cont": args.cont_loss_coef,
          "loss_heatmap": args.heatmap_loss_coef}
    if args.aux_loss:
        for i in range(args.dec_layers - 1):
            wd.update({f"{k}_{i}": v for k, v in list(wd.items()) if not k[-1].isdigit()})
    return SetCriterion(matcher, losses, args.eos_coef, wd)
