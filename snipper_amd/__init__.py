"""snipper_amd -- MI355X (gfx950) native implementation of Snipper's deformable-attention hot path.

Scope: the multi-scale deformable attention core op (forward + backward) as
hand-written HIP kernels behind a C ABI (include/snipper_msda.h, csrc/), and the host-side mirror
of the reference's interface for that path:

    MultiScaleDeformableAttention   ms_deform_attn_forward / ms_deform_attn_backward
    ms_deform_attn_func             MSDeformAttnFunction, ms_deform_attn_core_pytorch
    ms_deform_attn                  MSDeformAttn (spatiotemporal module)
    deformable_transformer          DeformableTransformer*, build_deforamble_transformer

Nothing here imports ``oracle/`` (test infrastructure) and nothing falls back to a CPU path.
"""
import sys as _sys

__version__ = "0.1.0"


def install(also_reference_paths: bool = True) -> None:
    """Register the drop-in modules under the names the reference imports.

    After ``snipper_amd.install()``, ``import MultiScaleDeformableAttention as MSDA``
    (reference models/ops/functions/ms_deform_attn_func.py:18-21) binds to the HIP library, and
    -- with ``also_reference_paths`` -- ``models.ops.functions`` / ``models.ops.modules`` /
    ``models.deformable_transformer`` resolve to this package's mirrors when the reference tree
    itself is not importable.
    """
    from . import MultiScaleDeformableAttention as _msda
    _sys.modules["MultiScaleDeformableAttention"] = _msda
    if also_reference_paths:
        from . import deformable_transformer as _dt
        from . import ms_deform_attn as _mod
        from . import ms_deform_attn_func as _fn
        _sys.modules.setdefault("models.ops.functions.ms_deform_attn_func", _fn)
        _sys.modules.setdefault("models.ops.modules.ms_deform_attn", _mod)
        _sys.modules.setdefault("models.deformable_transformer", _dt)
