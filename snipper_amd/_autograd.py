"""``torch.autograd.Function`` for this package's own nodes, minus the functorch bookkeeping of ``Function.apply``.

``torch.autograd.Function.apply`` (torch/autograd/function.py) binds default arguments for ``setup_context`` and unwraps dead
functorch wrappers before it reaches the C++ ``apply`` -- 1.6 us per call on the host that issues a training step through
~180 such nodes.  None of the nodes here defines ``setup_context``, so outside a functorch transform they go to the C++ entry
directly; INSIDE one (``torch.func.grad`` / ``vmap`` / ``jvp`` over a model that contains them) the call is handed to the stock
``apply``, which raises PyTorch's own readable error for a Function without ``setup_context`` instead of an internal assert.
Positional arguments only (the stock ``apply`` accepts keywords through ``setup_context`` binding, which these nodes do not
have): a keyword raises ``TypeError`` here.  ``MSDeformAttnFunction`` (the reference's public node, reference
models/ops/functions/ms_deform_attn_func.py:24) keeps the stock ``apply``."""
import torch

_functorch_active = torch._C._are_functorch_transforms_active


class Function(torch.autograd.Function):
    @classmethod
    def apply(cls, *args, **kwargs):
        if kwargs:
            raise TypeError(f"{cls.__name__}.apply() takes positional arguments only (got {sorted(kwargs)})")
        if _functorch_active():
            return torch.autograd.Function.apply.__func__(cls, *args)
        return super(torch.autograd.Function, cls).apply(*args)
