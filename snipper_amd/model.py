"""Model assembly around the hot path: backbone -> 1x1 projections -> spatiotemporal deformable
transformer -> prediction heads.

API mirror of ``SnipperDeformable`` in /root/reference/models/model.py:45-237 (constructor
arguments, parameter names -- ``input_proj.N.{0,1}``, ``query_embed``, ``class_embed.N``,
``root_embed.N.layers.0``, ``joint_embed.N.K.layers.0`` with the same tying across decoder layers --
and the output dictionary); pinned against the reference class itself by golden g6 (tests/test_host.py).  The
criterion / Hungarian matcher (model.py:240-545, matcher.py) live in ``snipper_amd/criterion.py``.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F
from torch import nn

from .backbone import build_backbone
from .deformable_transformer import build_deforamble_transformer, inverse_sigmoid
from .misc import NestedTensor, nested_tensor_from_tensor_list


class MLP(nn.Module):
    def __init__(self, input_dim, hidden_dim, output_dim, num_layers):
        super().__init__()
        self.num_layers = num_layers
        dims = [input_dim] + [hidden_dim] * (num_layers - 1) + [output_dim]
        self.layers = nn.ModuleList(nn.Linear(a, b) for a, b in zip(dims[:-1], dims[1:]))

    def forward(self, x):
        for i, layer in enumerate(self.layers):
            x = layer(x) if i == self.num_layers - 1 else F.relu(layer(x))
        return x


class SnipperDeformable(nn.Module):
    def __init__(self, backbone, transformer, num_queries, num_feature_levels,
                 num_frames, num_future_frames, num_keypoints, aux_loss=True):
        super().__init__()
        self.num_queries, self.transformer, self.backbone = num_queries, transformer, backbone
        self.aux_loss = aux_loss
        self.num_frames, self.num_future_frames = num_frames, num_future_frames
        self.num_keypoints, self.num_feature_levels = num_keypoints, num_feature_levels
        d = transformer.d_model

        def proj(cin, k=1, **kw):
            return nn.Sequential(nn.Conv2d(cin, d, kernel_size=k, **kw), nn.GroupNorm(32, d))

        if num_feature_levels > 1:
            projs = [proj(c) for c in backbone.num_channels]
            cin = backbone.num_channels[-1]
            for _ in range(num_feature_levels - len(backbone.num_channels)):   # extra strided levels (:73-78)
                projs.append(proj(cin, 3, stride=2, padding=1))
                cin = d
        else:
            projs = [proj(backbone.num_channels[0])]
        self.input_proj = nn.ModuleList(projs)

        self.query_embed = nn.Embedding(num_queries * (num_frames + num_future_frames), d * 2)
        n_dec = transformer.decoder.num_layers
        cls, root = nn.Linear(d, 2), MLP(d, d, 4, 1)
        joints = nn.ModuleList([MLP(d, d, 4, 1) for _ in range(num_keypoints - 1)])
        # one set of heads shared by every decoder layer (reference :99-101)
        self.class_embed = nn.ModuleList([cls] * n_dec)
        self.root_embed = nn.ModuleList([root] * n_dec)
        self.joint_embed = nn.ModuleList([joints] * n_dec)
        self.transformer.decoder.root_embed = self.root_embed
        self.transformer.decoder.class_embed = self.class_embed

    def _joints(self, l, h):
        """The K-1 single-layer joint heads (reference :193-195 runs them one by one and concatenates)
        evaluated as ONE matmul over their stacked weights: same parameters, same result."""
        heads = self.joint_embed[l]
        if all(hd.num_layers == 1 for hd in heads):
            w = torch.cat([hd.layers[0].weight for hd in heads], 0)               # [(K-1)*4, d]
            b = torch.cat([hd.layers[0].bias for hd in heads], 0)
            return F.linear(h, w, b).view(*h.shape[:3], len(heads), 4)
        return torch.cat([hd(h).reshape(*h.shape[:3], 1, 4) for hd in heads], dim=3)

    def forward(self, samples):
        if not isinstance(samples, NestedTensor):
            samples = nested_tensor_from_tensor_list(samples)
        if samples.tensors.is_cuda and torch.is_grad_enabled() and torch.is_autocast_enabled('cuda'):
            # bf16 working copies of the weights (BN scale folded in for the backbone): a few multi-tensor launches
            # here instead of 2-3 tiny ones at every layer (snipper_amd/shadow.py)
            from .shadow import WeightShadows
            if getattr(self, "_shadows", None) is None:
                object.__setattr__(self, "_shadows", WeightShadows(self))
            self._shadows.refresh()
        fast = self._forward_tokens(samples)
        if fast is not None:
            hs, heatmaps, init_reference, inter_references, inter_att = fast
            return self._heads_entry(hs, heatmaps, init_reference, inter_references, inter_att)
        features, pos = self.backbone(samples)
        srcs, masks = [], []
        for lvl, feat in enumerate(features):
            src, mask = feat.decompose()
            srcs.append(self.input_proj[lvl](src))
            masks.append(mask)
        for lvl in range(len(srcs), self.num_feature_levels):       # levels beyond the backbone's (:135-147)
            src = self.input_proj[lvl](features[-1].tensors if lvl == len(features) else srcs[-1])
            mask = F.interpolate(samples.mask[None].float(), size=src.shape[-2:]).to(torch.bool)[0]
            pos.append(self.backbone[1](NestedTensor(src, mask)).to(src.dtype))
            srcs.append(src)
            masks.append(mask)

        T = self.num_frames
        for lvl in range(self.num_feature_levels):                   # [b*t,c,h,w] -> [b,c,t,h,w] (:150-159)
            n, c, h, w = srcs[lvl].shape
            srcs[lvl] = srcs[lvl].reshape(n // T, T, c, h, w).transpose(1, 2)
            masks[lvl] = masks[lvl].reshape(n // T, T, 1, h, w).expand(-1, -1, c, -1, -1).transpose(1, 2)
            pos[lvl] = pos[lvl].reshape(n // T, T, c, h, w).transpose(1, 2)

        hs, heatmaps, init_reference, inter_references, inter_att = \
            self.transformer(srcs, masks, pos, self.query_embed.weight)
        return self._heads_entry(hs, heatmaps, init_reference, inter_references, inter_att)

    token_rows = True        # class-level switch (tests compare both formulations)

    def _forward_tokens(self, samples):
        """Backbone maps -> transformer without the channel-first detour (models/model.py:128-159 builds [b,c,t,h,w]
        tensors that the transformer immediately flattens back to token rows): the projections and the position
        encoding are produced token-major.  Returns None when the conditions of the fused path do not hold."""
        joiner = self.backbone
        if not (isinstance(joiner, nn.Sequential) and len(joiner) >= 2):
            return None
        if not (self.token_rows and samples.tensors.is_cuda and hasattr(self.transformer, "forward_from_features") and
                hasattr(joiner[1], "channel_last") and self.num_feature_levels == len(joiner.num_channels)):
            return None
        xs = joiner[0](samples)
        feats = [x for _, x in sorted(xs.items())]
        maps = [f.tensors for f in feats]
        if not self.transformer.tokens_path_ok(maps, self.input_proj):
            return None
        T = self.num_frames
        pos_tokens = []
        for f in feats:
            p = joiner[1].channel_last(f.mask)                        # [b, t, h, w, C] float32
            pos_tokens.append(p.flatten(2, 3))
        return self.transformer.forward_from_features(maps, [f.mask for f in feats], pos_tokens, self.input_proj,
                                                      self.query_embed.weight)

    def _heads_entry(self, hs, heatmaps, init_reference, inter_references, inter_att):
        n_dec, bs, t, _, c = hs.shape
        if hs.is_cuda and torch.is_autocast_enabled('cuda'):
            # the heads see [n_dec, bs, T, queries, C] (a few thousand rows): float32, no autocast casts (the same
            # reasoning as DeformableTransformerDecoderLayer.small_in_fp32)
            with torch.autocast("cuda", enabled=False):
                return self._heads(hs.float(), heatmaps, init_reference, inter_references, inter_att)
        return self._heads(hs, heatmaps, init_reference, inter_references, inter_att)

    def _heads(self, hs, heatmaps, init_reference, inter_references, inter_att):
        n_dec, bs, t, _, c = hs.shape
        tied = all(m is self.class_embed[0] for m in self.class_embed) and \
            all(m is self.root_embed[0] for m in self.root_embed) and \
            all(m is self.joint_embed[0] for m in self.joint_embed)
        if tied:
            # the heads are one set of modules shared by every decoder layer (reference :99-101), so the per-layer
            # loop of reference :172-201 is one batched evaluation over the stacked decoder outputs
            classes = self.class_embed[0](hs).transpose(2, 3)                              # [n_dec, bs, nq, t, 2]
            anchors = inverse_sigmoid(torch.cat([init_reference[None], inter_references[:-1]], 0))
            root = self.root_embed[0](hs).view(n_dec, bs, t, self.num_queries, 1, 4)
            root = torch.cat([root[..., :2] + anchors[:, :, :, :, None, :], root[..., 2:]], -1).sigmoid()
            joints = self._joints(0, hs.flatten(0, 1)).view(n_dec, bs, t, self.num_queries, -1, 4)
            kpts = torch.cat([root, joints], dim=4).transpose(2, 3)                        # [n_dec, bs, nq, t, K, 4]
        else:
            classes, kpts = [], []
            for l in range(n_dec):
                classes.append(self.class_embed[l](hs[l]).transpose(1, 2))             # [bs, nq, t, 2]
                anchor = inverse_sigmoid(init_reference if l == 0 else inter_references[l - 1])
                root = self.root_embed[l](hs[l]).view(bs, t, self.num_queries, 1, 4)
                root = torch.cat([root[..., :2] + anchor[:, :, :, None, :], root[..., 2:]], -1).sigmoid()
                kpts.append(torch.cat([root, self._joints(l, hs[l])], dim=3).transpose(1, 2))   # [bs, nq, t, K, 4]
            classes, kpts = torch.stack(classes), torch.stack(kpts)
        out = {'pred_logits': classes[-1], 'pred_kpts2d': kpts[-1, ..., 0:3],
               'pred_depth': kpts[-1, ..., 3:4], 'heatmaps': heatmaps,
               # every decoder layer at once (not in the reference's dict; lets a loss avoid a per-layer loop)
               'all_layers': {'pred_logits': classes, 'pred_kpts': kpts}}
        if self.aux_loss:
            out['aux_outputs'] = [{'pred_logits': classes[i], 'pred_kpts2d': kpts[i, ..., 0:3],
                                   'pred_depth': kpts[i, ..., 3:4]} for i in range(n_dec - 1)]
        return out, (init_reference, inter_references, inter_att)


def build_model(args):
    """Model only (reference build_model :618-632 also builds the criterion and post-processor)."""
    return SnipperDeformable(build_backbone(args), build_deforamble_transformer(args), args.num_queries,
                             args.num_feature_levels, args.num_frames, args.num_future_frames,
                             args.num_kpts, args.aux_loss)
