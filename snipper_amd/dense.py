"""Host binding of the bf16 MFMA dense kernel (include/snipper_dense.h, csrc/gemm_bf16.cuh)."""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib


def linear_bf16(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None,
                residual: Optional[torch.Tensor] = None, relu: bool = False) -> torch.Tensor:
    """act(x @ weight.T + bias + residual) on the HIP kernel.

    x [..., K] bf16 (last dim contiguous, rows evenly strided), weight [N, K] bf16 contiguous,
    bias [N] float32 or None, residual like the output (bf16) or None  ->  [..., N] bf16.
    """
    assert x.is_cuda and x.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16
    K = x.shape[-1]
    N = weight.shape[0]
    x2 = x.reshape(-1, K)
    if x2.stride(1) != 1:
        x2 = x2.contiguous()
    M = x2.shape[0]
    weight = weight.contiguous()
    out = torch.empty((M, N), dtype=torch.bfloat16, device=x.device)
    r2 = None
    if residual is not None:
        r2 = residual.reshape(-1, N)
        if r2.stride(1) != 1 or r2.dtype != torch.bfloat16:
            r2 = r2.to(torch.bfloat16).contiguous()
    if bias is not None and bias.dtype != torch.float32:
        bias = bias.float()
    lib = _lib.load()
    with torch.cuda.device(x.device):
        rc = lib.snipper_linear_bf16(
            torch.cuda.current_stream(x.device).cuda_stream, x2.data_ptr(), x2.stride(0), weight.data_ptr(),
            bias.data_ptr() if bias is not None else None, r2.data_ptr() if r2 is not None else None,
            r2.stride(0) if r2 is not None else 0, out.data_ptr(), out.stride(0), M, N, K, int(relu))
    _lib.check(rc, "snipper_linear_bf16")
    return out.view(*x.shape[:-1], N)
