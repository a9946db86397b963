"""Top-level alias so that the unmodified reference line ``import MultiScaleDeformableAttention as MSDA``
(/root/reference/models/ops/functions/ms_deform_attn_func.py:18-21) binds to the gfx950 HIP library whenever this
repository's root is on ``sys.path`` (or the file is installed beside the package) -- no ``snipper_amd.install()`` call,
no edit of the reference.  The implementation is ``snipper_amd/MultiScaleDeformableAttention.py``."""
from snipper_amd.MultiScaleDeformableAttention import ms_deform_attn_backward, ms_deform_attn_forward  # noqa: F401

__all__ = ["ms_deform_attn_forward", "ms_deform_attn_backward"]
