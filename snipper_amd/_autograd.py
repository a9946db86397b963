"""``torch.autograd.Function`` for this package's own nodes, minus the functorch bookkeeping of ``Function.apply``.

``torch.autograd.Function.apply`` (torch/autograd/function.py) binds default arguments for ``setup_context`` and unwraps dead
functorch wrappers before it reaches the C++ ``apply`` -- 1.6 us per call on the host that issues a training step through
~180 such nodes.  None of the nodes here defines ``setup_context`` or is used under a functorch transform (vmap / jvp), so
they go to the C++ entry directly.  ``MSDeformAttnFunction`` (the reference's public node, reference
models/ops/functions/ms_deform_attn_func.py:24) keeps the stock ``apply``."""
import torch


class Function(torch.autograd.Function):
    @classmethod
    def apply(cls, *args):
        return super(torch.autograd.Function, cls).apply(*args)
