"""Host binding of the bf16 MFMA dense kernel (include/snipper_dense.h, csrc/gemm_bf16.cuh)."""
from __future__ import annotations

from typing import Optional

import torch
from ._autograd import Function as _Fn

from . import _lib

# ---- optional launch timing (bench.py `roofline_dense`, tools/dense_roofline.py) -------------------------------------
# When enabled, every GEMM-shaped C-ABI call below is bracketed by two events on the stream it is launched on, together
# with its FLOPs and compulsory HBM bytes (operands read once, result written once); nothing synchronises until the list
# is read.  Off (the default) the bracket is one shared no-op object.
_timing = None


def enable_launch_timing(on: bool = True) -> None:
    global _timing
    _timing = [] if on else None


def launch_timings():
    """[(kind, shape tuple, flops, bytes, elapsed_ms)] for the calls since enable_launch_timing(True); synchronises."""
    if not _timing:
        return []
    torch.cuda.synchronize()
    return [(k, sh, f, b, s.elapsed_time(e)) for (k, sh, f, b, s, e) in _timing]


class _Timed:
    __slots__ = ("kind", "shape", "flops", "bytes", "device", "start")

    def __init__(self, kind, shape, flops, nbytes, device):
        self.kind, self.shape, self.flops, self.bytes, self.device = kind, shape, flops, nbytes, device

    def __enter__(self):
        self.start = torch.cuda.Event(enable_timing=True)
        self.start.record(torch.cuda.current_stream(self.device))
        return self

    def __exit__(self, *exc):
        if _timing is not None and exc[0] is None:
            end = torch.cuda.Event(enable_timing=True)
            end.record(torch.cuda.current_stream(self.device))
            _timing.append((self.kind, self.shape, self.flops, self.bytes, self.start, end))
        return False


def _timed(kind, shape, flops, nbytes, device):
    return _lib._NO_GUARD if _timing is None else _Timed(kind, shape, flops, nbytes, device)


def linear_bf16(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None,
                residual: Optional[torch.Tensor] = None, relu: bool = False, dropout_p: float = 0.0,
                seed: int = 0) -> torch.Tensor:
    """dropout(act(x @ weight.T + bias + residual)) on the HIP kernel (dropout_p = 0: no dropout).

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
    with _timed("linear", (M, N, K), 2 * M * N * K, 2 * (M * K + N * K + M * N * (2 if r2 is not None else 1)), x.device), \
            _lib.device_guard(x.device):
        rc = lib.snipper_linear_bf16(
            _lib.raw_stream(x.device), x2.data_ptr(), x2.stride(0), weight.data_ptr(),
            bias.data_ptr() if bias is not None else None, r2.data_ptr() if r2 is not None else None,
            r2.stride(0) if r2 is not None else 0, out.data_ptr(), out.stride(0), M, N, K, int(relu),
            float(dropout_p), int(seed))
    _lib.check(rc, "snipper_linear_bf16")
    return out.view(*x.shape[:-1], N)


def wres_supported(M: int, N: int, K: int) -> bool:
    """Shape taken by the weight-stationary kernel (csrc/wres_gemm_bf16.cuh): K in {288, 384}, many rows."""
    return bool(_lib.load().snipper_linear_wres_supported(int(M), int(N), int(K)))


def linear_wres_bf16(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, relu: bool = False,
                     dropout_p: float = 0.0, seed: int = 0, gate: Optional[torch.Tensor] = None,
                     gate_scale: float = 1.0) -> torch.Tensor:
    """gate(dropout(act(x @ weight.T + bias))) on the weight-stationary kernel: x [M, K] bf16 (K = 288 or 384, rows
    16-byte aligned), weight [N, K] bf16 (row stride % 8 == 0) -- a data gradient dX = dY @ W passes W^T here --,
    bias [N] float32 or None, gate [M, N] bf16 or None (result = gate > 0 ? result * gate_scale : 0)  ->  [M, N] bf16."""
    assert x.is_cuda and x.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16 and x.dim() == 2 and weight.dim() == 2
    M, K = x.shape
    N = weight.shape[0]
    if x.stride(1) != 1 or x.stride(0) % 8 or x.data_ptr() % 16:
        x = x.contiguous()
    if weight.stride(1) != 1 or weight.stride(0) % 8 or weight.data_ptr() % 16:
        weight = weight.contiguous()
    if gate is not None:
        assert gate.dtype == torch.bfloat16 and gate.shape == (M, N)
        if gate.stride(1) != 1 or gate.stride(0) % 8 or gate.data_ptr() % 16:
            gate = gate.contiguous()
    if bias is not None and bias.dtype != torch.float32:
        bias = bias.float()
    out = torch.empty((M, N), dtype=torch.bfloat16, device=x.device)
    with _timed("linear_wres", (M, N, K), 2 * M * N * K, 2 * (M * K + N * K + M * N * (2 if gate is not None else 1)), x.device), \
            _lib.device_guard(x.device):
        rc = _lib.load().snipper_linear_wres_bf16(
            _lib.raw_stream(x.device), x.data_ptr(), x.stride(0), weight.data_ptr(), weight.stride(0),
            bias.data_ptr() if bias is not None else None, gate.data_ptr() if gate is not None else None,
            gate.stride(0) if gate is not None else 0, float(gate_scale), out.data_ptr(), out.stride(0), M, N, K,
            int(relu), float(dropout_p), int(seed))
    _lib.check(rc, "snipper_linear_wres_bf16")
    return out


_CAST_TABLES: dict = {}


def _dense_layout(t: torch.Tensor) -> bool:
    return t.is_contiguous() or (t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last))


def cast_scale_table_bf16(items) -> bool:
    """dst = bf16(src * scale[out channel]) for a list of (src float32, dst bf16 with src's layout, scale [src.shape[0]] float32 or
    None) in ONE launch (csrc/misc_kernels.cuh, cast_scale_table_kernel).  The table of pointers is uploaded once per distinct
    list (the per-step refresh of the weight shadows passes the same tensors every step).  Returns False -- nothing launched --
    when a tensor does not fit the kernel (the caller then takes its PyTorch path)."""
    import numpy as np
    items = list(items)
    if not items:
        return True
    key = tuple((s_.data_ptr(), d_.data_ptr(), 0 if sc is None else sc.data_ptr(), tuple(s_.shape), s_.stride()) for s_, d_, sc in items)
    hit = _CAST_TABLES.get(key)
    if hit is None:
        dev = items[0][0].device
        rec = np.zeros(len(items), dtype=np.dtype([("src", "<u8"), ("dst", "<u8"), ("scale", "<u8"), ("numel", "<i8"),
                                                   ("inner", "<i4"), ("pad", "<i4")]))
        ends, blocks = [], 0
        for i, (s_, d_, sc) in enumerate(items):
            n = s_.numel()
            inner = n // max(1, s_.shape[0])
            ok = (s_.is_cuda and s_.device == dev and d_.device == dev and s_.dtype == torch.float32 and d_.dtype == torch.bfloat16 and
                  s_.shape == d_.shape and s_.stride() == d_.stride() and _dense_layout(s_) and n % 8 == 0 and n > 0 and
                  s_.data_ptr() % 16 == 0 and d_.data_ptr() % 16 == 0 and
                  (sc is None or (sc.dtype == torch.float32 and sc.is_contiguous() and sc.numel() == s_.shape[0] and
                                  sc.device == dev and inner % 8 == 0)))
            if not ok:
                return False
            rec[i] = (s_.data_ptr(), d_.data_ptr(), 0 if sc is None else sc.data_ptr(), n, inner, 0)
            blocks += (n + 2047) // 2048
            ends.append(blocks)
        table = torch.from_numpy(rec.view(np.uint8).copy()).to(dev)
        block_end = torch.tensor(ends, dtype=torch.int32).to(dev)
        if len(_CAST_TABLES) > 8:
            _CAST_TABLES.clear()
        hit = _CAST_TABLES[key] = (table, block_end, len(items), blocks)
    table, block_end, n_items, blocks = hit
    with _lib.device_guard(table.device):
        rc = _lib.load().snipper_cast_scale_table_bf16(_lib.raw_stream(table.device), table.data_ptr(), block_end.data_ptr(),
                                                       n_items, blocks)
    _lib.check(rc, "snipper_cast_scale_table_bf16")
    return True


_TRANSPOSE_ARGS: dict = {}      # tuple of the tensors' ids -> (weak references, prepared argument arrays per launch)


def transpose_batch_bf16(pairs) -> None:
    """dst[c][r] = src[r][c] for a list of (src [rows, cols], dst [cols, rows]) bf16 matrices (unit inner strides) in ONE
    launch per 48 matrices (csrc/wres_gemm_bf16.cuh, transpose_batch_bf16_kernel).  The argument arrays of a list that
    recurs (the per-step refresh of the weight shadows: the same ~150 views every step) are built once."""
    import ctypes
    import weakref
    pairs = list(pairs)
    if not pairs:
        return
    lib = _lib.load()
    key = tuple(id(t) for pr in pairs for t in pr)
    hit = _TRANSPOSE_ARGS.get(key)
    if hit is not None and all(r() is t for r, t in zip(hit[0], (t for pr in pairs for t in pr))) and \
            all(t.data_ptr() == ptr for ptr, t in zip(hit[2], (t for pr in pairs for t in pr))):
        launches = hit[1]
    else:
        launches = []
        for lo in range(0, len(pairs), 48):
            part = pairs[lo:lo + 48]
            n = len(part)
            for s_, d_ in part:
                assert s_.dtype == torch.bfloat16 and d_.dtype == torch.bfloat16 and s_.stride(1) == 1 and d_.stride(1) == 1
                assert d_.shape == (s_.shape[1], s_.shape[0]) and s_.is_cuda and d_.device == s_.device
            launches.append((n, (ctypes.c_void_p * n)(*[s_.data_ptr() for s_, _ in part]),
                             (ctypes.c_void_p * n)(*[d_.data_ptr() for _, d_ in part]),
                             (ctypes.c_int * n)(*[s_.shape[0] for s_, _ in part]),
                             (ctypes.c_int * n)(*[s_.shape[1] for s_, _ in part]),
                             (ctypes.c_longlong * n)(*[s_.stride(0) for s_, _ in part]),
                             (ctypes.c_longlong * n)(*[d_.stride(0) for _, d_ in part])))
        if len(_TRANSPOSE_ARGS) > 16:
            _TRANSPOSE_ARGS.clear()
        flat = [t for pr in pairs for t in pr]
        _TRANSPOSE_ARGS[key] = ([weakref.ref(t) for t in flat], launches, [t.data_ptr() for t in flat])
    dev = pairs[0][0].device
    with _lib.device_guard(dev):
        for n, src, dst, rows, cols, lds, ldd in launches:
            rc = lib.snipper_transpose_batch_bf16(_lib.raw_stream(dev), n, src, dst, rows, cols, lds, ldd)
            _lib.check(rc, "snipper_transpose_batch_bf16")


def _gate_ptr(gate: Optional[torch.Tensor], like: torch.Tensor):
    """Data pointer of a ReLU gate laid out exactly like ``like`` (bf16, channels_last), or None."""
    if gate is None:
        return None, None
    if gate.dtype != torch.bfloat16 or not gate.is_contiguous(memory_format=torch.channels_last):
        gate = gate.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    assert gate.shape == like.shape
    return gate.data_ptr(), gate          # (the tensor is returned so that it outlives the launch call)


def conv3x3_bf16(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, stride: int = 1,
                 relu: bool = False, gate: Optional[torch.Tensor] = None, flip_taps: bool = False) -> torch.Tensor:
    """act(conv2d(x, weight, padding=1, stride=stride) + bias) on the implicit-GEMM HIP kernel.  ``gate`` (shape of the
    result): results whose gate is not > 0 are written as 0 (a ReLU backward fused into a data gradient's store phase).
    ``flip_taps``: the kernel reads the weight with reversed taps (``weight.flip(2, 3)`` without the copy).

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
    gp, _keep = _gate_ptr(gate, out)
    with _timed("conv3x3" + ("_dgrad" if flip_taps else ""), (B, H, W, Cin, Cout, stride), 2 * B * Ho * Wo * Cout * 9 * Cin,
                2 * (B * H * W * Cin + 9 * Cin * Cout + B * Ho * Wo * Cout * (2 if gate is not None else 1)), x.device), \
            _lib.device_guard(x.device):
        rc = _lib.load().snipper_conv3x3_bf16(
            _lib.raw_stream(x.device), x.data_ptr(), weight.data_ptr(),
            bias.data_ptr() if bias is not None else None, out.data_ptr(), B, H, W, Cin, Cout, int(stride), int(relu), gp,
            int(flip_taps))
    _lib.check(rc, "snipper_conv3x3_bf16")
    return out


_PACK_ARGS: dict = {}


def conv3x3_pack_bf16(items) -> None:
    """Pack 3x3 convolution weights for ``conv3x3_patch_bf16`` in one launch (csrc/conv3x3_patch_bf16.cuh).  ``items``: a list
    of (src, dst, transposed) -- src [Cout, Cin, 3, 3] bf16 in channels_last memory, dst a flat bf16 tensor of the same number
    of elements; transposed = True packs the stride-1 data gradient's weight (channel roles swapped, taps reversed)."""
    import ctypes
    items = list(items)
    if not items:
        return
    key = tuple((s_.data_ptr(), d_.data_ptr(), bool(t_), tuple(s_.shape)) for s_, d_, t_ in items)
    args = _PACK_ARGS.get(key)
    if args is None:
        n = len(items)
        for s_, d_, _ in items:
            assert s_.is_cuda and s_.dtype == torch.bfloat16 and d_.dtype == torch.bfloat16 and s_.dim() == 4
            assert tuple(s_.shape[2:]) == (3, 3) and s_.is_contiguous(memory_format=torch.channels_last)
            assert d_.is_contiguous() and d_.numel() == s_.numel() and d_.device == s_.device
        args = (n, (ctypes.c_void_p * n)(*[s_.data_ptr() for s_, _, _ in items]),
                (ctypes.c_void_p * n)(*[d_.data_ptr() for _, d_, _ in items]),
                (ctypes.c_int * n)(*[s_.shape[0] for s_, _, _ in items]), (ctypes.c_int * n)(*[s_.shape[1] for s_, _, _ in items]),
                (ctypes.c_int * n)(*[int(bool(t_)) for _, _, t_ in items]))
        if len(_PACK_ARGS) > 8:
            _PACK_ARGS.clear()
        _PACK_ARGS[key] = args
    dev = items[0][0].device
    with _lib.device_guard(dev):
        rc = _lib.load().snipper_conv3x3_pack_bf16(_lib.raw_stream(dev), *args)
    _lib.check(rc, "snipper_conv3x3_pack_bf16")


_LPACK_ARGS: dict = {}


def linear_pack_bf16(items) -> None:
    """Pack [N, K] bf16 matrices (contiguous rows) for ``linear_patch_bf16`` in one launch.  ``items``: (src, dst, transposed);
    transposed = True packs the operand of the data gradient dX[M, K] = dY[M, N] . W (call linear_patch_bf16 with N = K)."""
    import ctypes
    items = list(items)
    if not items:
        return
    key = tuple((s_.data_ptr(), d_.data_ptr(), bool(t_), tuple(s_.shape)) for s_, d_, t_ in items)
    args = _LPACK_ARGS.get(key)
    if args is None:
        n = len(items)
        for s_, d_, _ in items:
            assert s_.is_cuda and s_.dtype == torch.bfloat16 and d_.dtype == torch.bfloat16 and s_.dim() == 2 and s_.is_contiguous()
            assert d_.is_contiguous() and d_.numel() == s_.numel() and d_.device == s_.device
        args = (n, (ctypes.c_void_p * n)(*[s_.data_ptr() for s_, _, _ in items]),
                (ctypes.c_void_p * n)(*[d_.data_ptr() for _, d_, _ in items]),
                (ctypes.c_int * n)(*[s_.shape[0] for s_, _, _ in items]), (ctypes.c_int * n)(*[s_.shape[1] for s_, _, _ in items]),
                (ctypes.c_int * n)(*[int(bool(t_)) for _, _, t_ in items]))
        if len(_LPACK_ARGS) > 8:
            _LPACK_ARGS.clear()
        _LPACK_ARGS[key] = args
    dev = items[0][0].device
    with _lib.device_guard(dev):
        rc = _lib.load().snipper_linear_pack_bf16(_lib.raw_stream(dev), *args)
    _lib.check(rc, "snipper_linear_pack_bf16")


def linear_patch_supported(M: int, N: int, K: int) -> bool:
    return bool(_lib.load().snipper_linear_patch_supported(int(M), int(N), int(K)))


def linear_patch_bf16(x: torch.Tensor, packed: torch.Tensor, N: int, bias: Optional[torch.Tensor] = None,
                      residual: Optional[torch.Tensor] = None, relu: bool = False, gate: Optional[torch.Tensor] = None,
                      bn: int = 0, kind: str = "linear") -> torch.Tensor:
    """act(x @ W^T + bias + residual) (or gate > 0 ? . : 0) for a weight packed by ``linear_pack_bf16``: the X tile is
    streamed through LDS in 64-channel slices, W goes straight into MFMA operand registers (csrc/conv3x3_patch_bf16.cuh with
    one tap).  x [M, K] bf16 contiguous  ->  [M, N] bf16."""
    assert x.is_cuda and x.dtype == torch.bfloat16 and packed.dtype == torch.bfloat16 and x.dim() == 2 and x.is_contiguous()
    M, K = x.shape
    assert packed.numel() == N * K
    out = torch.empty((M, N), dtype=torch.bfloat16, device=x.device)
    if bias is not None and bias.dtype != torch.float32:
        bias = bias.float()
    if residual is not None:
        assert residual.dtype == torch.bfloat16 and residual.shape == (M, N) and residual.is_contiguous()
    if gate is not None:
        assert gate.dtype == torch.bfloat16 and gate.shape == (M, N) and gate.is_contiguous()
    extra = (1 if residual is not None else 0) + (1 if gate is not None else 0)
    with _timed(kind, (M, N, K), 2 * M * N * K, 2 * (M * K + N * K + M * N * (1 + extra)), x.device), _lib.device_guard(x.device):
        rc = _lib.load().snipper_linear_patch_bf16(
            _lib.raw_stream(x.device), x.data_ptr(), packed.data_ptr(), bias.data_ptr() if bias is not None else None,
            residual.data_ptr() if residual is not None else None, out.data_ptr(), M, N, K, int(relu),
            gate.data_ptr() if gate is not None else None, int(bn))
    _lib.check(rc, "snipper_linear_patch_bf16")
    return out


def linear_wide_supported(M: int, N: int, K: int) -> bool:
    return bool(_lib.load().snipper_linear_wide_supported(int(M), int(N), int(K)))


def linear_wide_bf16(x: torch.Tensor, packed: torch.Tensor, bias: Optional[torch.Tensor] = None, kind: str = "linear") -> torch.Tensor:
    """x [M, K] @ W^T + bias -> [M, 384] bf16 on full-width tiles (csrc/conv3x3_patch_bf16.cuh, linear_wide_kernel): K a multiple of
    128, ``packed`` = linear_pack_bf16's pack of W [384, K] (or the transposed pack of W [K, 384] for a data gradient)."""
    assert x.is_cuda and x.dtype == torch.bfloat16 and packed.dtype == torch.bfloat16 and x.dim() == 2 and x.is_contiguous()
    M, K = x.shape
    assert packed.numel() == 384 * K
    if bias is not None and bias.dtype != torch.float32:
        bias = bias.float()
    if bias is not None and (bias.data_ptr() % 16 or not bias.is_contiguous()):
        bias = bias.contiguous().clone()       # (the kernel reads it in 16-byte pieces; snipper_linear_wide_supported sees no pointers)
    out = torch.empty((M, 384), dtype=torch.bfloat16, device=x.device)
    with _timed(kind, (M, 384, K), 2 * M * 384 * K, 2 * (M * K + 384 * K + M * 384), x.device), _lib.device_guard(x.device):
        rc = _lib.load().snipper_linear_wide_bf16(_lib.raw_stream(x.device), x.data_ptr(), packed.data_ptr(),
                                                  bias.data_ptr() if bias is not None else None, out.data_ptr(), M, 384, K)
    _lib.check(rc, "snipper_linear_wide_bf16")
    return out


def conv3x3_patch_supported(B: int, H: int, W: int, Cin: int, Cout: int) -> bool:
    return bool(_lib.load().snipper_conv3x3_patch_supported(int(B), int(H), int(W), int(Cin), int(Cout)))


def conv3x3_patch_bf16(x: torch.Tensor, packed: torch.Tensor, cout: int, bias: Optional[torch.Tensor] = None,
                       relu: bool = False, gate: Optional[torch.Tensor] = None, dgrad: bool = False) -> torch.Tensor:
    """act(conv2d(x, w, padding=1) + bias) for a weight packed by ``conv3x3_pack_bf16`` (stride 1): the input patch of a 2-D
    output tile stays in LDS for all nine taps, the weight streams into registers (csrc/conv3x3_patch_bf16.cuh).
    x [B, Cin, H, W] bf16 channels_last  ->  [B, cout, H, W] bf16 channels_last.  ``dgrad`` only labels the timing record."""
    assert x.is_cuda and x.dtype == torch.bfloat16 and packed.dtype == torch.bfloat16
    assert x.is_contiguous(memory_format=torch.channels_last)
    B, Cin, H, W = x.shape
    assert packed.numel() == cout * 9 * Cin
    out = torch.empty((B, cout, H, W), dtype=torch.bfloat16, device=x.device, memory_format=torch.channels_last)
    if bias is not None and bias.dtype != torch.float32:
        bias = bias.float()
    gp, _keep = _gate_ptr(gate, out)
    with _timed("conv3x3" + ("_dgrad" if dgrad else ""), (B, H, W, Cin, cout, 1), 2 * B * H * W * cout * 9 * Cin,
                2 * (B * H * W * Cin + 9 * Cin * cout + B * H * W * cout * (2 if gate is not None else 1)), x.device), \
            _lib.device_guard(x.device):
        rc = _lib.load().snipper_conv3x3_patch_bf16(
            _lib.raw_stream(x.device), x.data_ptr(), packed.data_ptr(), bias.data_ptr() if bias is not None else None,
            out.data_ptr(), B, H, W, Cin, int(cout), int(relu), gp)
    _lib.check(rc, "snipper_conv3x3_patch_bf16")
    return out


def conv3x3_dgrad_s2_bf16(g: torch.Tensor, weight_t: torch.Tensor, in_hw, gate: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Data gradient of a stride-2, padding-1 3x3 convolution on the implicit-GEMM kernel (four parity-class launches).

    g [B, Cout, Ho, Wo] bf16 channels_last = dL/dy; weight_t [Cin, Cout, 3, 3] bf16 channels_last = the weight with its
    channel roles swapped (``w.transpose(0, 1)``, not flipped); in_hw = (H, W) of the convolution's input
    ->  dL/dx [B, Cin, H, W] bf16 channels_last."""
    assert g.is_cuda and g.dtype == torch.bfloat16 and weight_t.dtype == torch.bfloat16
    g = g.contiguous(memory_format=torch.channels_last)
    weight_t = weight_t.contiguous(memory_format=torch.channels_last)
    B, Cg, Hg, Wg = g.shape
    Cx = weight_t.shape[0]
    H, W = in_hw
    assert weight_t.shape[1] == Cg and (H - 1) // 2 + 1 == Hg and (W - 1) // 2 + 1 == Wg
    dx = torch.empty((B, Cx, H, W), dtype=torch.bfloat16, device=g.device, memory_format=torch.channels_last)
    gp, _keep = _gate_ptr(gate, dx)
    with _timed("conv3x3_dgrad_s2", (B, H, W, Cx, Cg, 2), 2 * B * Hg * Wg * Cg * 9 * Cx,
                2 * (B * Hg * Wg * Cg + 9 * Cx * Cg + B * H * W * Cx * (2 if gate is not None else 1)), g.device), \
            _lib.device_guard(g.device):
        rc = _lib.load().snipper_conv3x3_dgrad_s2_bf16(_lib.raw_stream(g.device), g.data_ptr(), weight_t.data_ptr(),
                                                       dx.data_ptr(), B, H, W, Cx, Cg, gp)
    _lib.check(rc, "snipper_conv3x3_dgrad_s2_bf16")
    return dx


def wgrad_conv3x3_bf16(g: torch.Tensor, x: torch.Tensor, stride: int, scale: Optional[torch.Tensor] = None,
                       out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Weight gradient of a padding-1 3x3 convolution (stride 1 or 2) on the split-reduction MFMA kernel.

    g [B, Cout, Ho, Wo] = dL/dy and x [B, Cin, H, W] = the input, both bf16 channels_last; ``scale`` [Cout] float32
    multiplies the rows (folded BatchNorm)  ->  dW [Cout, Cin, 3, 3] float32 in channels_last memory; deterministic."""
    assert g.is_cuda and g.dtype == torch.bfloat16 and x.dtype == torch.bfloat16
    g = g.contiguous(memory_format=torch.channels_last)
    x = x.contiguous(memory_format=torch.channels_last)
    B, Cin, H, W = x.shape
    Cout = g.shape[1]
    lib = _lib.load()
    nbytes = lib.snipper_wgrad_conv3x3_workspace_bytes(B, H, W, Cin, Cout, int(stride))
    if scale is not None and scale.dtype != torch.float32:
        scale = scale.float()
    ws = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=g.device)
    if out is not None:                     # a [Cout, Cin, 3, 3] float32 tensor in channels_last memory (e.g. the parameter's
        assert (out.dtype == torch.float32 and tuple(out.shape) == (Cout, Cin, 3, 3) and       # slice of a flat gradient buffer)
                out.is_contiguous(memory_format=torch.channels_last) and out.data_ptr() % 16 == 0)
        dw = out.permute(0, 2, 3, 1)        # its memory as [Cout, 3, 3, Cin]
    else:
        dw = torch.empty((Cout, 3, 3, Cin), dtype=torch.float32, device=g.device)
    Ho_, Wo_ = g.shape[2], g.shape[3]
    with _timed("conv3x3_wgrad", (B, H, W, Cin, Cout, int(stride)), 2 * B * Ho_ * Wo_ * Cout * 9 * Cin,
                2 * (B * H * W * Cin + B * Ho_ * Wo_ * Cout) + 4 * 9 * Cin * Cout, g.device), _lib.device_guard(g.device):
        rc = lib.snipper_wgrad_conv3x3_bf16(_lib.raw_stream(g.device), g.data_ptr(), x.data_ptr(), B, H, W, Cin, Cout,
                                            int(stride), scale.data_ptr() if scale is not None else None, dw.data_ptr(),
                                            0, ws.data_ptr(), nbytes)
    _lib.check(rc, "snipper_wgrad_conv3x3_bf16")
    return out if out is not None else dw.permute(0, 3, 1, 2)            # [Cout, Cin, 3, 3] logical, channels_last memory


def linear_nn_bf16(x: torch.Tensor, w: torch.Tensor, residual: Optional[torch.Tensor] = None,
                   gate: Optional[torch.Tensor] = None, gate_scale: float = 1.0) -> torch.Tensor:
    """gate(x [M, K] @ w [K, N] + residual [M, N]) (all bf16, row-major; w = a Linear's weight [out, in] used for the
    data gradient) on the HIP kernel that reads the weight tile through the transposing LDS read -- no transposed
    copy.  ``gate`` [M, N]: the activation the gradient flows back through; result = gate > 0 ? result * gate_scale : 0
    (ReLU, or ReLU + dropout, backward fused into the epilogue)."""
    assert x.is_cuda and x.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and x.dim() == 2 and w.dim() == 2
    assert x.shape[1] == w.shape[0]
    if x.stride(1) != 1 or x.stride(0) % 8 or x.data_ptr() % 16:
        x = x.contiguous()
    if w.stride(1) != 1 or w.stride(0) % 8 or w.data_ptr() % 16:
        w = w.contiguous()
    M, K = x.shape
    N = w.shape[1]
    out = torch.empty((M, N), dtype=torch.bfloat16, device=x.device)
    r2 = None
    if residual is not None:
        r2 = residual.reshape(M, N)
        if r2.dtype != torch.bfloat16 or r2.stride(1) != 1 or r2.stride(0) % 4:
            r2 = r2.to(torch.bfloat16).contiguous()
    a2 = None
    if gate is not None:
        a2 = gate.reshape(M, N)
        if a2.dtype != torch.bfloat16 or a2.stride(1) != 1 or a2.stride(0) % 4:
            a2 = a2.to(torch.bfloat16).contiguous()
    with _timed("linear_nn", (M, N, K), 2 * M * N * K,
                2 * (M * K + N * K + M * N * (1 + (r2 is not None) + (a2 is not None))), x.device), _lib.device_guard(x.device):
        rc = _lib.load().snipper_linear_nn_bf16(
            _lib.raw_stream(x.device), x.data_ptr(), x.stride(0), w.data_ptr(), w.stride(0),
            r2.data_ptr() if r2 is not None else None, r2.stride(0) if r2 is not None else 0,
            a2.data_ptr() if a2 is not None else None, a2.stride(0) if a2 is not None else 0, float(gate_scale),
            out.data_ptr(), out.stride(0), M, N, K)
    _lib.check(rc, "snipper_linear_nn_bf16")
    return out


import os as _os
# Reductions from this length up would go to hipBLASLt.  In isolation (tools/nnbench.py) the library is 15-25 % faster
# from K = 512 up; inside the training step the A/B (tools/ab.sh, 40 steps, twice) showed no gain, so the own kernel
# (which also carries the skip-connection addend) stays the default.
_DGRAD_LIB_K = int(_os.environ.get("SNIPPER_DGRAD_LIB_K", "1000000"))


def _dgrad(g: torch.Tensor, w: torch.Tensor, residual: Optional[torch.Tensor] = None,
           gate: Optional[torch.Tensor] = None, wt: Optional[torch.Tensor] = None, gate_scale: float = 1.0) -> torch.Tensor:
    """gate(g [M, out] @ w [out, in] (+ residual)): own kernel when the shape allows (out % 64 == 0, in % 8 == 0),
    hipBLASLt otherwise.  ``gate`` [M, in]: the ReLU output the gradient flows back into (result zeroed where it is 0).
    ``wt`` = w^T [in, out] (the per-step transposed shadow, shadow.lookup_t): with it a short reduction (out = 288 / 384)
    over many rows goes to the weight-stationary kernel."""
    if (wt is not None and residual is None and g.dtype == torch.bfloat16 and g.dim() == 2 and
            wres_supported(g.shape[0], wt.shape[0], wt.shape[1]) and (gate is None or gate.dtype == torch.bfloat16)):
        return linear_wres_bf16(g, wt, None, gate=None if gate is None else gate.reshape(g.shape[0], wt.shape[0]),
                                gate_scale=gate_scale)
    if gate_scale != 1.0:
        assert gate is not None and w.shape[0] % 64 == 0 and w.shape[1] % 8 == 0 and g.shape[0] >= 256
        return linear_nn_bf16(g, w, residual, gate, gate_scale)
    if w.shape[0] % 64 == 0 and w.shape[1] % 8 == 0 and g.shape[0] >= 256 and (
            w.shape[0] < _DGRAD_LIB_K or residual is not None or gate is not None):
        return linear_nn_bf16(g, w, residual, gate)
    y = torch.mm(g, w)
    if residual is not None:
        y = y + residual.reshape(y.shape)
    return y if gate is None else torch.ops.aten.threshold_backward(y, gate.reshape(y.shape), 0)


def wgrad_bf16(g: torch.Tensor, x: torch.Tensor, want_bias: bool = True, scale: Optional[torch.Tensor] = None,
               out: Optional[torch.Tensor] = None, out_bias: Optional[torch.Tensor] = None, accumulate: bool = False):
    """Weight / bias gradient of ``y = x @ W^T + b`` on the split-reduction MFMA kernel (csrc/wgrad_bf16.cuh).

    g [M, N] bf16 = dL/dy, x [M, Kc] bf16 (both row-major, unit inner stride)  ->  (dW [N, Kc] float32,
    db [N] float32 or None).  ``scale`` [N] float32 multiplies the rows of dW (folded BatchNorm).  ``out`` /
    ``out_bias`` receive the result (``accumulate`` adds to them); deterministic."""
    assert g.is_cuda and g.dtype == torch.bfloat16 and x.dtype == torch.bfloat16 and g.dim() == 2 and x.dim() == 2
    assert g.shape[0] == x.shape[0]
    if g.stride(1) != 1 or g.stride(0) % 8 or g.data_ptr() % 16:
        g = g.contiguous()
    if x.stride(1) != 1 or x.stride(0) % 8 or x.data_ptr() % 16:
        x = x.contiguous()
    M, N = g.shape
    Kc = x.shape[1]
    lib = _lib.load()
    if scale is not None and scale.dtype != torch.float32:
        scale = scale.float()
    dW = out if out is not None else torch.empty((N, Kc), dtype=torch.float32, device=g.device)
    assert dW.dtype == torch.float32 and dW.stride(1) == 1 and dW.shape == (N, Kc)
    db = None
    if want_bias:
        db = out_bias if out_bias is not None else torch.empty((N,), dtype=torch.float32, device=g.device)
        assert db.dtype == torch.float32 and db.is_contiguous()
    nbytes = lib.snipper_wgrad_workspace_bytes(M, N, Kc)
    ws = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=g.device)
    with _timed("wgrad", (M, N, Kc), 2 * M * N * Kc, 2 * (M * N + M * Kc) + 4 * N * Kc, g.device), _lib.device_guard(g.device):
        rc = lib.snipper_wgrad_bf16(
            _lib.raw_stream(g.device), g.data_ptr(), g.stride(0), x.data_ptr(), x.stride(0),
            M, N, Kc, scale.data_ptr() if scale is not None else None, dW.data_ptr(), dW.stride(0),
            db.data_ptr() if db is not None else None, int(accumulate), ws.data_ptr(), nbytes)
    _lib.check(rc, "snipper_wgrad_bf16")
    return dW, db


def _grad_out(param, shape) -> Optional[torch.Tensor]:
    """Where a weight-gradient kernel may write ``param``'s gradient directly: its slice of a FlatParameters' flat gradient
    buffer (flat_params.claim_grad_view: once per backward pass, float32, contiguous rows of the expected shape) or None."""
    if param is None or not isinstance(param, torch.nn.Parameter) or param.dtype != torch.float32:
        return None
    from .flat_params import claim_grad_view
    if getattr(param, "_snipper_flat_owner", None) is None or tuple(param.shape) != tuple(shape) or not param.is_contiguous():
        return None
    return claim_grad_view(param)


def _grad_out_conv(param) -> Optional[torch.Tensor]:
    """``_grad_out`` for a convolution weight [Cout, Cin, kh, kw] whose memory is [Cout][kh][kw][Cin] (channels_last, or any
    layout when kh = kw = 1): the parameter's gradient view, or None."""
    if param is None or not isinstance(param, torch.nn.Parameter) or param.dtype != torch.float32 or param.dim() != 4:
        return None
    if getattr(param, "_snipper_flat_owner", None) is None:
        return None
    one = tuple(param.shape[2:]) == (1, 1)
    if not (param.is_contiguous(memory_format=torch.channels_last) or (one and param.is_contiguous())):
        return None
    from .flat_params import claim_grad_view
    v = claim_grad_view(param)
    return v if (v is not None and v.data_ptr() % 16 == 0) else None


class _BigLinear(_Fn):
    """act(x @ W^T + b) for activations with tens of thousands of rows (the encoder's 79 000 tokens), in bf16 on this
    repository's kernels: forward on ``linear_bf16`` (bias / ReLU in the epilogue), weight + bias gradient on the
    split-reduction ``wgrad_bf16`` (float32 results straight into the parameter's dtype: no cast kernels), data
    gradient through hipBLASLt.  Reference: every nn.Linear of models/ops/modules/ms_deform_attn.py:60-66 and of the
    encoder layer's FFN (models/deformable_transformer.py:180-198)."""

    @staticmethod
    def forward(ctx, x, weight, bias, relu, dropout_p=0.0):
        """``dropout_p`` > 0 (only together with ``relu``): inverted dropout in the same epilogue; the backward needs
        no mask -- a kept, active element of the output is > 0, everything else is 0."""
        n_out, k_in = weight.shape
        x2 = x.reshape(-1, k_in)
        xb = x2 if x2.dtype == torch.bfloat16 else x2.to(torch.bfloat16)
        from . import shadow
        wb = weight if weight.dtype == torch.bfloat16 else shadow.lookup(weight)
        if wb is None:
            wb = weight.to(torch.bfloat16)
        assert dropout_p == 0.0 or relu, "epilogue dropout is implemented for the ReLU layer only"
        seed = 0
        if dropout_p > 0.0:
            from .fused import _next_seed
            seed = _next_seed()
        y = linear_bf16(xb, wb, None if bias is None else bias.float(), None, relu, dropout_p, seed)
        ctx.drop_p = float(dropout_p)
        ctx.relu, ctx.has_bias = relu, bias is not None
        ctx.x_shape, ctx.w_dtype = x.shape, weight.dtype
        ctx.b_dtype = None if bias is None else bias.dtype
        ctx.wt = shadow.lookup_t(weight) if weight.dtype != torch.bfloat16 else None     # W^T of THIS step's weight, if kept
        ctx.w_ref, ctx.b_ref = weight, bias                  # (parameters: for flat_params.claim_grad_view in the backward)
        ctx.save_for_backward(xb, wb, y if relu else None)
        return y.view(*x.shape[:-1], n_out)

    @staticmethod
    def backward(ctx, gy):
        xb, wb, y = ctx.saved_tensors
        g = gy.reshape(-1, wb.shape[0])
        if g.dtype != torch.bfloat16:
            g = g.to(torch.bfloat16)
        g = g.contiguous()
        if ctx.relu:
            g = _relu_dropout_backward(g, y, ctx.drop_p)
        dW = db = None
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            want_b = ctx.has_bias and ctx.needs_input_grad[2]
            dW, db = wgrad_bf16(g, xb, want_bias=want_b, out=_grad_out(ctx.w_ref, (wb.shape[0], xb.shape[1])),
                                out_bias=_grad_out(ctx.b_ref, (wb.shape[0],)) if want_b else None)
            if dW.dtype != ctx.w_dtype:
                dW = dW.to(ctx.w_dtype)
            if db is not None and db.dtype != ctx.b_dtype:
                db = db.to(ctx.b_dtype)
        dx = _dgrad(g, wb, wt=ctx.wt).view(ctx.x_shape) if ctx.needs_input_grad[0] else None
        return dx, dW, db, None, None


class _BigFFN(_Fn):
    """linear2(dropout(relu(linear1(x)))) -- the feed-forward block of a transformer layer on 79 000 token rows
    (reference models/deformable_transformer.py:194-198) as one autograd node: the hidden activation is written once
    (ReLU and dropout in linear1's epilogue) and, in the backward, the gradient with respect to it gets its
    ReLU / dropout gate inside the data-gradient kernel of linear2 -- no separate pass over the [rows, d_ffn] matrix in
    either direction."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, dropout_p):
        from . import shadow
        k_in = w1.shape[1]
        x2 = x.reshape(-1, k_in)
        xb = x2 if x2.dtype == torch.bfloat16 else x2.to(torch.bfloat16)
        w1b = shadow.lookup(w1)
        w1b = w1b if w1b is not None else w1.to(torch.bfloat16)
        w2b = shadow.lookup(w2)
        w2b = w2b if w2b is not None else w2.to(torch.bfloat16)
        seed = 0
        if dropout_p > 0.0:
            from .fused import _next_seed
            seed = _next_seed()
        h = linear_bf16(xb, w1b, b1.float(), None, True, dropout_p, seed)
        # the 1024-deep products on the one-tap patch kernel when this step's packs are there (shadow.lookup_lpacked)
        pk2, pk1 = shadow.lookup_lpacked(w2), shadow.lookup_lpacked(w1)
        if pk2 is not None and pk2[0] is not None and shadow.lookup(w2) is w2b and shadow.LINEAR_WIDE and \
                linear_wide_supported(h.shape[0], w2.shape[0], w2.shape[1]):
            z = linear_wide_bf16(h, pk2[0], b2)                   # full-width tiles: the hidden activation crosses L2 once
        elif pk2 is not None and pk2[0] is not None and shadow.lookup(w2) is w2b and shadow.LINEAR_PATCH and \
                linear_patch_supported(h.shape[0], w2.shape[0], w2.shape[1]):
            z = linear_patch_bf16(h, pk2[0], w2.shape[0], b2, None, False, None, 64)
        else:
            z = linear_bf16(h, w2b, b2.float())
        ctx.w1tp = pk1[1] if (pk1 is not None and shadow.lookup(w1) is w1b) else None
        ctx.p, ctx.x_shape = float(dropout_p), x.shape
        ctx.dts = (w1.dtype, b1.dtype, w2.dtype, b2.dtype)
        ctx.w2t = shadow.lookup_t(w2)                        # W2^T [d_ffn, d_model]: dH = dZ . W2 on the weight-stationary kernel
        ctx.prefs = (w1, b1, w2, b2)                         # (parameters: for flat_params.claim_grad_view in the backward)
        ctx.save_for_backward(xb, w1b, w2b, h)
        return z.view(*x.shape[:-1], w2.shape[0])

    @staticmethod
    def backward(ctx, gz):
        xb, w1b, w2b, h = ctx.saved_tensors
        g = gz.reshape(-1, w2b.shape[0])
        if g.dtype != torch.bfloat16:
            g = g.to(torch.bfloat16)
        g = g.contiguous()
        w1, b1, w2, b2 = ctx.prefs
        dW2, db2 = wgrad_bf16(g, h, out=_grad_out(w2, w2b.shape), out_bias=_grad_out(b2, (w2b.shape[0],)))
        gh = _dgrad(g, w2b, None, h, wt=ctx.w2t, gate_scale=1.0 / (1.0 - ctx.p))    # gradient w.r.t. linear1's pre-activation
        dW1, db1 = wgrad_bf16(gh, xb, out=_grad_out(w1, w1b.shape), out_bias=_grad_out(b1, (w1b.shape[0],)))
        dx = None
        if ctx.needs_input_grad[0]:
            from . import shadow
            if ctx.w1tp is not None and gh.is_contiguous() and shadow.LINEAR_WIDE and \
                    linear_wide_supported(gh.shape[0], w1b.shape[1], w1b.shape[0]):
                dx = linear_wide_bf16(gh, ctx.w1tp, None, kind="linear_nn").view(ctx.x_shape)
            elif ctx.w1tp is not None and gh.is_contiguous() and shadow.LINEAR_PATCH and \
                    linear_patch_supported(gh.shape[0], w1b.shape[1], w1b.shape[0]):
                dx = linear_patch_bf16(gh, ctx.w1tp, w1b.shape[1], None, None, False, None, 64, kind="linear_nn").view(ctx.x_shape)
            else:
                dx = _dgrad(gh, w1b).view(ctx.x_shape)
        outs = [dW1, db1, dW2, db2]
        outs = [o if o.dtype == dt else o.to(dt) for o, dt in zip(outs, ctx.dts)]
        return (dx, *outs, None)


def big_ffn(x: torch.Tensor, lin1: torch.nn.Linear, lin2: torch.nn.Linear, dropout: Optional[torch.nn.Dropout]
            ) -> Optional[torch.Tensor]:
    """``lin2(dropout(relu(lin1(x))))`` as one node on the hand-written kernels, or None when the conditions of
    ``big_linear`` do not hold for both layers (the caller then composes it from ``big_linear`` calls)."""
    twin = getattr(x, "_snipper_bf16", None)
    if twin is not None and twin.shape == x.shape and twin.device == x.device:
        x = twin
    rows = x.numel() // max(1, x.shape[-1])
    in_bf16 = x.dtype == torch.bfloat16 or (torch.is_autocast_enabled('cuda') and
                                             torch.get_autocast_dtype('cuda') == torch.bfloat16)
    ok = (x.is_cuda and in_bf16 and rows >= BIG_LINEAR_MIN_ROWS and x.dtype in (torch.bfloat16, torch.float32) and
          lin1.in_features % 64 == 0 and lin1.out_features % 64 == 0 and lin2.in_features == lin1.out_features and
          lin2.out_features % 8 == 0 and lin1.bias is not None and lin2.bias is not None and
          rows * lin1.out_features < 2 ** 32)
    if not ok:
        return None
    p = dropout.p if (dropout is not None and dropout.training) else 0.0
    return _BigFFN.apply(x, lin1.weight, lin1.bias, lin2.weight, lin2.bias, p)


def _relu_dropout_backward(g: torch.Tensor, y: torch.Tensor, p: float) -> torch.Tensor:
    """y > 0 ? g / (1 - p) : 0 in one pass (bf16, contiguous)."""
    if p == 0.0 or g.numel() % 8:
        out = torch.ops.aten.threshold_backward(g, y, 0)
        return out if p == 0.0 else out * (1.0 / (1.0 - p))
    out = torch.empty_like(g)
    with _lib.device_guard(g.device):
        rc = _lib.load().snipper_relu_dropout_backward_bf16(
            _lib.raw_stream(g.device), g.data_ptr(), y.data_ptr(), out.data_ptr(), g.numel(), p)
    _lib.check(rc, "snipper_relu_dropout_backward_bf16")
    return out


class _SmallLinear(_Fn):
    """x @ W^T + b for a few hundred float32 rows (the decoder): identical arithmetic to F.linear, but the bias
    gradient is a [1, rows] x [rows, N] product instead of a column reduction -- PyTorch's reduce kernel needs ~25 us
    for a 480 x 384 column sum (one workgroup per few columns), a GEMM launch ~9 us; 48 of them per step."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        x2 = x.reshape(-1, x.shape[-1])
        ctx.save_for_backward(x2, weight)
        ctx.x_shape = x.shape
        y = small_linear_forward(x2, weight, bias)
        if y is None:
            y = torch.addmm(bias, x2, weight.t())
        return y.view(*x.shape[:-1], weight.shape[0])

    @staticmethod
    def backward(ctx, gy):
        x2, weight = ctx.saved_tensors
        g = gy.reshape(-1, weight.shape[0])
        fused = small_linear_backward(g, x2, weight, *ctx.needs_input_grad)
        if fused is not None:
            dx, dW, db = fused
            return (dx.view(ctx.x_shape) if dx is not None else None), dW, db
        dx = torch.mm(g, weight).view(ctx.x_shape) if ctx.needs_input_grad[0] else None
        dW = torch.mm(g.t(), x2) if ctx.needs_input_grad[1] else None
        db = None
        if ctx.needs_input_grad[2]:
            ones = _ones_row(g.shape[0], g.device, g.dtype)
            db = torch.mm(ones, g).view(-1)
        return dx, dW, db


def _sg_ok(*ts) -> bool:
    """float32 CUDA matrices the small-GEMM kernel can read in place (unit inner stride, rows a multiple of 16 bytes)."""
    return all(t.is_cuda and t.dtype == torch.float32 and t.dim() == 2 and t.stride(1) == 1 and t.stride(0) % 4 == 0 and
               t.shape[1] % 4 == 0 and t.data_ptr() % 16 == 0 for t in ts)


_SG_FMT = "@PqiPqiPqPPiiiPqiifQPqf0P"      # include/snipper_dense.h: snipper_small_gemm (native alignment, padded to 8)


def small_gemm_batch(problems) -> None:
    """One launch of csrc/small_linear.cuh for a list of float32 products (<= 6).  Each problem is a tuple
    ``(A, a_transposed, B, b_transposed, out, bias, colsum)``:  out[I, J] = opA(A) . opB(B) (+ bias), colsum[I] =
    row sums of opA(A); ``out`` (2-D, unit inner stride: may be a row block of a larger matrix), ``bias`` and ``colsum``
    may be None (not both out and colsum).  Operands must satisfy ``_sg_ok``.  An optional 8th element is a dict of
    extras: ``B2`` (the rows of opB from ``r_split = B.shape[0]`` on, stored apart), ``relu``, ``drop_p`` + ``seed``,
    ``gate`` + ``gate_scale`` (include/snipper_dense.h: snipper_small_gemm).
    (The argument array is packed with ``struct`` -- setting ~20 ctypes fields per problem cost more host time than the
    launch itself.)"""
    import struct
    parts = []
    dev = None
    for prob in problems:
        A, a_tr, B, b_tr, out, bias, colsum = prob[:7]
        extra = prob[7] if len(prob) > 7 else None
        dev = A.device
        b2p, ldb2, r_split, relu, drop_p, seed, gatep, ldgate, gate_scale = 0, 0, 0, 0, 0.0, 0, 0, 0, 0.0
        b_rows = B.shape[0]
        if extra:
            B2 = extra.get("B2")
            if B2 is not None:
                assert not b_tr and B2.shape[1] == B.shape[1]
                b2p, ldb2, r_split = B2.data_ptr(), B2.stride(0), B.shape[0]
                b_rows += B2.shape[0]
            relu = int(bool(extra.get("relu", False)))
            drop_p, seed = float(extra.get("drop_p", 0.0)), int(extra.get("seed", 0))
            gate = extra.get("gate")
            if gate is not None:
                gatep, ldgate, gate_scale = gate.data_ptr(), gate.stride(0), float(extra.get("gate_scale", 1.0))
        I, R = (A.shape[1], A.shape[0]) if a_tr else (A.shape[0], A.shape[1])
        J = B.shape[0] if b_tr else B.shape[1]
        assert (B.shape[1] if b_tr else b_rows) == R
        outp, ldo = 0, 0
        if out is not None:
            assert out.dtype == torch.float32 and out.shape == (I, J) and out.stride(1) == 1
            outp, ldo = out.data_ptr(), out.stride(0)
        parts.append(struct.pack(_SG_FMT, A.data_ptr(), A.stride(0), int(bool(a_tr)), B.data_ptr(), B.stride(0),
                                 int(bool(b_tr)), outp, ldo, bias.data_ptr() if bias is not None else 0,
                                 colsum.data_ptr() if colsum is not None else 0, I, J, R, b2p, ldb2, r_split, relu,
                                 drop_p, seed, gatep, ldgate, gate_scale))
    buf = b"".join(parts)
    with _lib.device_guard(dev):
        rc = _lib.load().snipper_small_gemm_batch_f32(_lib.raw_stream(dev), buf, len(problems))
    _lib.check(rc, "snipper_small_gemm_batch_f32")


def _sg_dense(t: torch.Tensor) -> torch.Tensor:
    return t if (t.stride(1) == 1 and t.stride(0) % 4 == 0 and t.data_ptr() % 16 == 0) else t.contiguous()


def small_linear_forward(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor]):
    """x [M, K] @ w [N, K]^T + b in one launch of the small-GEMM kernel, or None when it does not apply."""
    if not (x.is_cuda and x.dtype == torch.float32 and w.dtype == torch.float32 and x.dim() == 2 and
            x.shape[1] % 4 == 0 and x.shape[0] <= 8192 and (b is None or (b.dtype == torch.float32 and b.is_contiguous()))):
        return None
    x, w = _sg_dense(x), _sg_dense(w)
    if not _sg_ok(x, w):
        return None
    y = torch.empty((x.shape[0], w.shape[0]), dtype=torch.float32, device=x.device)
    with _lib.device_guard(x.device):
        rc = _lib.load().snipper_small_linear_forward_f32(
            _lib.raw_stream(x.device), x.data_ptr(), x.stride(0), w.data_ptr(), w.stride(0),
            b.data_ptr() if b is not None else None, x.shape[0], w.shape[0], x.shape[1], y.data_ptr(), y.stride(0))
    _lib.check(rc, "snipper_small_linear_forward_f32")
    return y


def small_linear_backward(g: torch.Tensor, x: torch.Tensor, w: torch.Tensor, need_dx=True, need_dw=True, need_db=True,
                          dw_out: Optional[torch.Tensor] = None, db_out: Optional[torch.Tensor] = None, launch=True):
    """(dX, dW, db) of ``y = x @ w.T + b`` from g = dL/dy in ONE launch (csrc/small_linear.cuh; float32, deterministic),
    or None when the shapes are outside the kernel's requirements (the caller then takes the three library GEMMs).
    ``dw_out`` / ``db_out``: where to write dW / db (e.g. a row block of a packed gradient).  ``launch=False`` returns
    ``(dX, dW, db, problems)`` without launching, for batching several layers into one launch."""
    M, N = g.shape
    K = x.shape[1]
    if not (g.is_cuda and g.dtype == torch.float32 and x.dtype == torch.float32 and w.dtype == torch.float32 and
            N % 4 == 0 and K % 4 == 0 and M <= 8192 and w.shape == (N, K) and x.shape[0] == M):
        return None
    g, x, w = _sg_dense(g), _sg_dense(x), _sg_dense(w)
    if not _sg_ok(g, x, w):
        return None
    dx = torch.empty((M, K), dtype=torch.float32, device=g.device) if need_dx else None
    dw = (dw_out if dw_out is not None else torch.empty((N, K), dtype=torch.float32, device=g.device)) if need_dw else None
    db = (db_out if db_out is not None else torch.empty((N,), dtype=torch.float32, device=g.device)) if need_db else None
    problems = []
    if need_dx:
        problems.append((g, False, w, False, dx, None, None))                  # G . W
    if need_dw or need_db:
        problems.append((g, True, x, False, dw, None, db))                     # G^T . X, column sums of G
    if not launch:
        return dx, dw, db, problems
    if problems:                       # (plain-argument entry point: cheaper on the host than building the problem list)
        with _lib.device_guard(g.device):
            rc = _lib.load().snipper_small_linear_backward_f32(
                _lib.raw_stream(g.device), g.data_ptr(), g.stride(0), x.data_ptr(), x.stride(0), w.data_ptr(), w.stride(0),
                M, N, K, dx.data_ptr() if dx is not None else None, K,
                dw.data_ptr() if dw is not None else None, dw.stride(0) if dw is not None else K,
                db.data_ptr() if db is not None else None)
        _lib.check(rc, "snipper_small_linear_backward_f32")
    return dx, dw, db


class _SmallFFN(_Fn):
    """linear2(dropout(relu(linear1(x)))) for a few hundred float32 rows (the decoder's feed-forward block, reference
    models/deformable_transformer.py:266-275) on the small-GEMM kernel: ReLU + dropout in linear1's epilogue; in the
    backward the gate of the hidden gradient in the data-gradient product of linear2, and each layer's (dX | dH, dW, db)
    triple as one launch -- 4 launches instead of 8, no mask tensor (a kept, active element of h is > 0)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, p):
        x2 = _sg_dense(x.reshape(-1, x.shape[-1]))
        M = x2.shape[0]
        h = torch.empty((M, w1.shape[0]), dtype=torch.float32, device=x.device)
        y = torch.empty((M, w2.shape[0]), dtype=torch.float32, device=x.device)
        seed = 0
        if p > 0.0:
            from .fused import _next_seed
            seed = _next_seed()
        small_gemm_batch([(x2, False, w1, True, h, b1, None, {"relu": True, "drop_p": p, "seed": seed})])
        small_gemm_batch([(h, False, w2, True, y, b2, None)])
        ctx.save_for_backward(x2, w1, w2, h)
        ctx.p, ctx.x_shape = float(p), x.shape
        return y.view(*x.shape[:-1], w2.shape[0])

    @staticmethod
    def backward(ctx, gy):
        x2, w1, w2, h = ctx.saved_tensors
        g = _sg_dense(gy.reshape(-1, w2.shape[0]))
        M = g.shape[0]
        dev = g.device
        gh = torch.empty_like(h)
        dw2, db2 = torch.empty_like(w2), torch.empty((w2.shape[0],), dtype=torch.float32, device=dev)
        small_gemm_batch([(g, False, w2, False, gh, None, None, {"gate": h, "gate_scale": 1.0 / (1.0 - ctx.p)}),
                          (g, True, h, False, dw2, None, db2)])
        dx = torch.empty_like(x2) if ctx.needs_input_grad[0] else None
        dw1, db1 = torch.empty_like(w1), torch.empty((w1.shape[0],), dtype=torch.float32, device=dev)
        probs = [(gh, True, x2, False, dw1, None, db1)]
        if dx is not None:
            probs.insert(0, (gh, False, w1, False, dx, None, None))
        small_gemm_batch(probs)
        return (dx.view(ctx.x_shape) if dx is not None else None), dw1, db1, dw2, db2, None


def small_ffn(x: torch.Tensor, lin1: torch.nn.Linear, lin2: torch.nn.Linear, dropout: Optional[torch.nn.Dropout]
              ) -> Optional[torch.Tensor]:
    """``lin2(dropout(relu(lin1(x))))`` as one node on the small-GEMM kernel for decoder-size float32 inputs, or None."""
    rows = x.numel() // max(1, x.shape[-1])
    ok = (x.is_cuda and x.dtype == torch.float32 and not torch.is_autocast_enabled('cuda') and torch.is_grad_enabled() and
          16 <= rows < BIG_LINEAR_MIN_ROWS and lin1.bias is not None and lin2.bias is not None and
          lin2.in_features == lin1.out_features and _sg_ok(lin1.weight, lin2.weight) and
          lin1.bias.is_contiguous() and lin2.bias.is_contiguous() and lin1.out_features % 4 == 0 and
          lin2.out_features % 4 == 0 and rows * lin1.out_features < 2 ** 32)
    if not ok:
        return None
    p = dropout.p if (dropout is not None and dropout.training) else 0.0
    return _SmallFFN.apply(x, lin1.weight, lin1.bias, lin2.weight, lin2.bias, float(p))


class _SmallLinearPair(_Fn):
    """cat([x @ Wa^T + ba, x @ Wb^T + bb], -1) for decoder-size float32 rows (the cross attention's offset and weight
    projections): one launch forward (two column blocks of one output), one launch backward (the shared input's gradient
    as ONE product over the stacked weights [Wa; Wb] read in place, both weight gradients, both bias gradients)."""

    @staticmethod
    def forward(ctx, x, wa, ba, wb, bb):
        x2 = _sg_dense(x.reshape(-1, x.shape[-1]))
        na, nb = wa.shape[0], wb.shape[0]
        y = torch.empty((x2.shape[0], na + nb), dtype=torch.float32, device=x.device)
        small_gemm_batch([(x2, False, wa, True, y[:, :na], ba, None), (x2, False, wb, True, y[:, na:], bb, None)])
        ctx.save_for_backward(x2, wa, wb)
        ctx.x_shape = x.shape
        return y.view(*x.shape[:-1], na + nb)

    @staticmethod
    def backward(ctx, gy):
        x2, wa, wb = ctx.saved_tensors
        na, nb = wa.shape[0], wb.shape[0]
        g = _sg_dense(gy.reshape(-1, na + nb))
        dev = g.device
        dwa, dwb = torch.empty_like(wa), torch.empty_like(wb)
        dba = torch.empty((na,), dtype=torch.float32, device=dev)
        dbb = torch.empty((nb,), dtype=torch.float32, device=dev)
        probs = [(g[:, :na], True, x2, False, dwa, None, dba), (g[:, na:], True, x2, False, dwb, None, dbb)]
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x2)
            probs.insert(0, (g, False, wa, False, dx, None, None, {"B2": wb}))
        small_gemm_batch(probs)
        return (dx.view(ctx.x_shape) if dx is not None else None), dwa, dba, dwb, dbb


def small_linear_pair(x: torch.Tensor, lin_a: torch.nn.Linear, lin_b: torch.nn.Linear) -> Optional[torch.Tensor]:
    rows = x.numel() // max(1, x.shape[-1])
    ok = (x.is_cuda and x.dtype == torch.float32 and not torch.is_autocast_enabled('cuda') and torch.is_grad_enabled() and
          16 <= rows < BIG_LINEAR_MIN_ROWS and lin_a.bias is not None and lin_b.bias is not None and
          lin_a.in_features == lin_b.in_features and _sg_ok(lin_a.weight, lin_b.weight) and
          lin_a.out_features % 32 == 0 and lin_b.out_features % 4 == 0 and
          lin_a.bias.is_contiguous() and lin_b.bias.is_contiguous())
    if not ok:
        return None
    return _SmallLinearPair.apply(x, lin_a.weight, lin_a.bias, lin_b.weight, lin_b.bias)


_ones_cache = {}


def _ones_row(n: int, device, dtype) -> torch.Tensor:
    key = (n, str(device), dtype)
    t = _ones_cache.get(key)
    if t is None:
        if len(_ones_cache) > 64:
            _ones_cache.clear()
        t = _ones_cache[key] = torch.ones((1, n), device=device, dtype=dtype)
    return t


BIG_LINEAR_MIN_ROWS = 4096


def big_linear(x: torch.Tensor, lin: torch.nn.Linear, relu: bool = False, dropout: Optional[torch.nn.Dropout] = None
               ) -> torch.Tensor:
    """``dropout(relu(lin(x)))`` (each optional; dropout only with relu), routed to the hand-written kernels when
    ``x`` is a CUDA tensor with at least BIG_LINEAR_MIN_ROWS rows that is computed in bf16 (bf16 input, or autocast to
    bf16); plain PyTorch otherwise, with identical semantics (the random stream differs: a counter-based hash instead
    of Philox)."""
    x_in = x
    twin = getattr(x, "_snipper_bf16", None)      # a bf16 copy written by the kernel that produced x (same graph)
    if twin is not None and twin.shape == x.shape and twin.device == x.device:
        x = twin
    rows = x.numel() // max(1, x.shape[-1])
    in_bf16 = x.dtype == torch.bfloat16 or (torch.is_autocast_enabled('cuda') and
                                             torch.get_autocast_dtype('cuda') == torch.bfloat16)
    if (x.is_cuda and in_bf16 and rows >= BIG_LINEAR_MIN_ROWS and lin.in_features % 64 == 0 and
            lin.out_features % 8 == 0 and x.dtype in (torch.bfloat16, torch.float32)):
        p = dropout.p if (dropout is not None and dropout.training and relu) else 0.0
        if x.numel() // x.shape[-1] * lin.out_features >= 2 ** 32:
            p = 0.0
        y = _BigLinear.apply(x, lin.weight, lin.bias, relu, p)
        return dropout(y) if (dropout is not None and p == 0.0) else y
    x = x_in                                      # the twin is only for the bf16 kernels
    if (x.is_cuda and x.dtype == torch.float32 and lin.weight.dtype == torch.float32 and lin.bias is not None and
            not torch.is_autocast_enabled('cuda') and 16 <= rows < BIG_LINEAR_MIN_ROWS and torch.is_grad_enabled()):
        y = _SmallLinear.apply(x, lin.weight, lin.bias)
    else:
        y = lin(x)
    y = torch.relu(y) if relu else y
    return dropout(y) if dropout is not None else y


def big_linear_merged(x: torch.Tensor, lins, first_bias_outside: bool = False) -> Optional[torch.Tensor]:
    """``cat([lin(x) for lin in lins], -1)`` as ONE projection (x is read once, one data-gradient GEMM, one
    weight-gradient launch) when the conditions of ``big_linear`` hold; ``None`` otherwise (the caller then evaluates
    the Linears one by one).  ``first_bias_outside`` (two Linears, bf16 path): the first Linear's columns come WITHOUT its
    bias -- the caller adds it in float32 (``merged_bias_is_outside`` says whether that happened); its gradient still
    comes from this node (the column sums of the output's gradient)."""
    twin = getattr(x, "_snipper_bf16", None)
    if twin is not None and twin.shape == x.shape and twin.device == x.device:
        x = twin
    rows = x.numel() // max(1, x.shape[-1])
    in_bf16 = x.dtype == torch.bfloat16 or (torch.is_autocast_enabled('cuda') and
                                             torch.get_autocast_dtype('cuda') == torch.bfloat16)
    k = lins[0].in_features
    if len(lins) == 2 and rows < BIG_LINEAR_MIN_ROWS:
        return small_linear_pair(x, lins[0], lins[1])       # decoder-size float32 rows (None when it does not apply)
    if not (x.is_cuda and in_bf16 and rows >= BIG_LINEAR_MIN_ROWS and k % 64 == 0 and
            x.dtype in (torch.bfloat16, torch.float32) and
            all(l.in_features == k and l.out_features % 8 == 0 and l.bias is not None for l in lins)):
        return None
    if len(lins) == 2:
        return _BigLinearPair.apply(x, lins[0].weight, lins[0].bias, lins[1].weight, lins[1].bias, lins[0], lins[1],
                                    bool(first_bias_outside))
    weight = torch.cat([l.weight for l in lins], 0)
    bias = torch.cat([l.bias for l in lins], 0)
    return _BigLinear.apply(x, weight, bias, False, 0.0)


class _BigLinearPair(_Fn):
    """cat([x @ Wa^T + ba, x @ Wb^T + bb], -1) as one projection (see big_linear_merged); the merged bf16 weight and
    bias come from the per-step shadows when they are valid, and the gradients go back to the four parameters as
    slices of one weight-gradient launch."""

    @staticmethod
    def forward(ctx, x, wa, ba, wb_, bb, lin_a, lin_b, first_bias_outside=False):
        from . import shadow
        k_in = wa.shape[1]
        x2 = x.reshape(-1, k_in)
        xb = x2 if x2.dtype == torch.bfloat16 else x2.to(torch.bfloat16)
        m = shadow.lookup_merged(lin_a, lin_b, second_bias_only=first_bias_outside)
        ctx.wt = None
        if m is not None:
            w16, bias = m
            ctx.wt = shadow.lookup_merged_t(lin_a, lin_b)
        else:
            w16 = torch.cat([wa, wb_], 0).to(torch.bfloat16)
            bias = torch.cat([torch.zeros_like(ba) if first_bias_outside else ba, bb], 0).float()
        y = linear_bf16(xb, w16, bias)
        ctx.na, ctx.x_shape = wa.shape[0], x.shape
        ctx.dts = (wa.dtype, ba.dtype, wb_.dtype, bb.dtype)
        ctx.save_for_backward(xb, w16)
        return y.view(*x.shape[:-1], w16.shape[0])

    @staticmethod
    def backward(ctx, gy):
        xb, w16 = ctx.saved_tensors
        g = gy.reshape(-1, w16.shape[0])
        if g.dtype != torch.bfloat16:
            g = g.to(torch.bfloat16)
        g = g.contiguous()
        dx = _dgrad(g, w16, wt=ctx.wt).view(ctx.x_shape) if ctx.needs_input_grad[0] else None
        dW, db = wgrad_bf16(g, xb, want_bias=True)
        na = ctx.na
        outs = [dW[:na], db[:na], dW[na:], db[na:]]
        outs = [o if o.dtype == dt else o.to(dt) for o, dt in zip(outs, ctx.dts)]
        return (dx, *outs, None, None, None)


def merged_bias_is_outside(x: torch.Tensor, lins) -> bool:
    """Does ``big_linear_merged(x, lins, first_bias_outside=True)`` take the path that leaves the first bias out (two Linears
    on the bf16 GEMM path)?  The decoder-size float32 pair adds both biases itself (no rounding to protect there)."""
    twin = getattr(x, "_snipper_bf16", None)
    if twin is not None and twin.shape == x.shape and twin.device == x.device:
        x = twin
    rows = x.numel() // max(1, x.shape[-1])
    return len(lins) == 2 and rows >= BIG_LINEAR_MIN_ROWS
