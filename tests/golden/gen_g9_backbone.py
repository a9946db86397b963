#!/usr/bin/env python3
"""Golden g9: the importable half of the reference's models/backbone.py -- ``FrozenBatchNorm2d`` (:27-64, incl. its
``_load_from_state_dict``), ``BackboneBase`` (:67-99: freeze rule, ``return_layers``, the nearest mask resize of
``forward``) and ``Joiner`` (:114-131) -- run by the REFERENCE classes themselves in the build container.

torchvision is neither vendored nor installed, so the two names backbone.py imports from it are stood in for:
``torchvision.models`` (only touched by ``Backbone.__init__``, which is not used here) and ``IntermediateLayerGetter``,
replaced by a module that walks the children of the given network in order and collects the outputs named in
``return_layers`` -- torchvision's documented behaviour.  The network handed to ``BackboneBase`` is a small stand-in with
ResNet's child names (conv1, bn1, relu, maxpool, layer1..4) built from plain convolutions and the reference's
FrozenBatchNorm2d, so the arithmetic of the frozen BN sits inside the golden feature maps.

    python tests/golden/gen_g9_backbone.py

Stores tensors only (inputs, state_dict, outputs); no reference source travels."""
import os
import sys
import types
from collections import OrderedDict

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


class LayerGetter(nn.ModuleDict):
    """Stand-in for torchvision.models._utils.IntermediateLayerGetter: children in registration order up to the last
    returned one; forward feeds x through them and collects {new_name: output}."""

    def __init__(self, model, return_layers):
        todo = dict(return_layers)
        layers = OrderedDict()
        for name, module in model.named_children():
            layers[name] = module
            todo.pop(name, None)
            if not todo:
                break
        super().__init__(layers)
        self.return_layers = dict(return_layers)

    def forward(self, x):
        out = OrderedDict()
        for name, module in self.items():
            x = module(x)
            if name in self.return_layers:
                out[self.return_layers[name]] = x
        return out


def import_reference_backbone():
    tv = types.ModuleType("torchvision")
    tv.__version__ = "0.9.0"
    tv.ops = types.ModuleType("torchvision.ops")
    tv.ops.misc = types.ModuleType("torchvision.ops.misc")
    tv.ops.misc.interpolate = F.interpolate
    tv.models = types.ModuleType("torchvision.models")
    tv.models._utils = types.ModuleType("torchvision.models._utils")
    tv.models._utils.IntermediateLayerGetter = LayerGetter
    for name, mod in (("torchvision", tv), ("torchvision.ops", tv.ops), ("torchvision.ops.misc", tv.ops.misc),
                      ("torchvision.models", tv.models), ("torchvision.models._utils", tv.models._utils)):
        sys.modules[name] = mod
    sys.path.insert(0, REF)
    import models.backbone as ref_backbone
    from models.position_encoding import PositionEmbeddingSine
    from util.misc import NestedTensor
    return ref_backbone, PositionEmbeddingSine, NestedTensor


def tiny_resnet(FrozenBN, g):
    """ResNet's top-level child names; strides 2 (conv1), 2 (maxpool), 1, 2, 2, 2 -> layer2/3/4 at 8/16/32."""
    def block(cin, cout, stride):
        return nn.Sequential(OrderedDict([("conv", nn.Conv2d(cin, cout, 3, stride, 1, bias=False)),
                                          ("bn", FrozenBN(cout)), ("relu", nn.ReLU())]))
    net = nn.Sequential(OrderedDict([
        ("conv1", nn.Conv2d(3, 4, 7, 2, 3, bias=False)), ("bn1", FrozenBN(4)), ("relu", nn.ReLU()),
        ("maxpool", nn.MaxPool2d(3, 2, 1)), ("layer1", block(4, 6, 1)), ("layer2", block(6, 8, 2)),
        ("layer3", block(8, 10, 2)), ("layer4", block(10, 12, 2)), ("avgpool", nn.AdaptiveAvgPool2d(1))]))
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, FrozenBN):
                m.weight.copy_(torch.rand(m.weight.shape, generator=g) + 0.5)
                m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.2)
                m.running_mean.copy_(torch.randn(m.bias.shape, generator=g) * 0.2)
                m.running_var.copy_(torch.rand(m.bias.shape, generator=g) + 0.3)
            elif isinstance(m, nn.Conv2d):
                m.weight.copy_(torch.randn(m.weight.shape, generator=g) * (2.0 / m.weight[0].numel()) ** 0.5)
    return net


def main():
    ref, PosSine, NestedTensor = import_reference_backbone()
    g = torch.Generator().manual_seed(909)
    blob = {}

    # -- FrozenBatchNorm2d alone: float32 and float64, and the num_batches_tracked key dropped on load (:43-51)
    for tag, dt in (("f32", torch.float32), ("f64", torch.float64)):
        bn = ref.FrozenBatchNorm2d(5, eps=1e-5)
        sd = {"weight": torch.rand(5, generator=g) + 0.5, "bias": torch.randn(5, generator=g),
              "running_mean": torch.randn(5, generator=g), "running_var": torch.rand(5, generator=g) * 2 + 0.01,
              "num_batches_tracked": torch.tensor(7)}
        bn.load_state_dict(dict(sd), strict=True)          # strict: the extra key must have been removed
        bn = bn.to(dt)
        x = torch.randn(2, 5, 3, 4, generator=g).to(dt)
        blob[f"bn_{tag}"] = {"state_dict": {k: v.clone() for k, v in sd.items()}, "x": x, "y": bn(x),
                             "keys_after": sorted(bn.state_dict().keys())}

    # -- BackboneBase + Joiner on the stand-in network, with padded masks
    T, bs, H, W = 2, 2, 64, 96
    for tag, (train_backbone, interm) in {"interm": (True, True), "last_frozen": (False, False)}.items():
        net = tiny_resnet(ref.FrozenBatchNorm2d, g)
        base = ref.BackboneBase(net, train_backbone, interm)
        pe = PosSine(8, num_frames=T, normalize=True)
        joiner = ref.Joiner(base, pe)
        imgs = torch.rand(bs * T, 3, H, W, generator=g)
        mask = torch.zeros(bs * T, H, W, dtype=torch.bool)
        mask[T:, :, W - 21:] = True                         # sample 1: 21 padded columns, 13 padded rows (not multiples
        mask[T:, H - 13:, :] = True                         # of any stride: exercises the nearest rule of :93)
        imgs = imgs.masked_fill(mask[:, None], 0.0)
        out, pos = joiner(NestedTensor(imgs, mask))
        blob[f"base_{tag}"] = {
            "train_backbone": train_backbone, "return_interm_layers": interm,
            "body_state_dict": {k: v.detach().clone() for k, v in base.body.state_dict().items()},
            "requires_grad": {k: bool(p.requires_grad) for k, p in base.named_parameters()},
            "strides": list(joiner.strides), "num_channels": list(joiner.num_channels),
            "imgs": imgs, "mask": mask, "num_frames": T, "pos_feats": 8,
            "features": [o.tensors.detach().clone() for o in out], "masks": [o.mask.clone() for o in out],
            "pos": [p.detach().clone() for p in pos],
        }
    # -- Joiner casts the position encoding to the features' dtype (:129)
    net = tiny_resnet(ref.FrozenBatchNorm2d, g).double()
    joiner = ref.Joiner(ref.BackboneBase(net, True, True), PosSine(8, num_frames=T, normalize=True))
    out, pos = joiner(NestedTensor(torch.rand(T, 3, 32, 32, generator=g).double(), torch.zeros(T, 32, 32, dtype=torch.bool)))
    blob["joiner_pos_dtype_f64_features"] = str(pos[0].dtype)
    assert pos[0].dtype == torch.float64
    torch.save(blob, os.path.join(OUT, "g9_backbone.pt"))
    print("g9_backbone.pt", os.path.getsize(os.path.join(OUT, "g9_backbone.pt")))


if __name__ == "__main__":
    main()
