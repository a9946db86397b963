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


def conv3x3_bf16(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, stride: int = 1,
                 relu: bool = False) -> torch.Tensor:
    """act(conv2d(x, weight, padding=1, stride=stride) + bias) on the implicit-GEMM HIP kernel.

    x [B, Cin, H, W] bf16 in channels_last memory, weight [Cout, Cin, 3, 3] bf16 in channels_last memory (i.e.
    [Cout][3][3][Cin] contiguous), bias [Cout] float32 or None  ->  [B, Cout, Ho, Wo] bf16, channels_last."""
    assert x.is_cuda and x.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16
    assert x.is_contiguous(memory_format=torch.channels_last)
    if not weight.is_contiguous(memory_format=torch.channels_last):
        weight = weight.contiguous(memory_format=torch.channels_last)
    B, Cin, H, W = x.shape
    Cout = weight.shape[0]
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    out = torch.empty((B, Cout, Ho, Wo), dtype=torch.bfloat16, device=x.device, memory_format=torch.channels_last)
    if bias is not None and bias.dtype != torch.float32:
        bias = bias.float()
    with torch.cuda.device(x.device):
        rc = _lib.load().snipper_conv3x3_bf16(
            torch.cuda.current_stream(x.device).cuda_stream, x.data_ptr(), weight.data_ptr(),
            bias.data_ptr() if bias is not None else None, out.data_ptr(), B, H, W, Cin, Cout, int(stride), int(relu))
    _lib.check(rc, "snipper_conv3x3_bf16")
    return out
