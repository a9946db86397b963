"""ctypes binding of libsnipper_msda.so (C ABI: include/snipper_msda.h).

The product path has no CPU or PyTorch fallback: if the HIP library cannot be loaded the
import of the op raises, loudly.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_int, c_longlong, c_size_t, c_void_p

from . import build as _build

_lib = None

_FWD_ARGS = [c_void_p] * 6 + [c_int] * 7 + [c_void_p]          # stream, value, shapes, lsi, loc, attn, dims, out
_BWD_ARGS = [c_void_p] * 7 + [c_int] * 7 + [c_void_p] * 3      # stream, grad_out, value, ..., dims, 3 grads

EXPORTS = {
    "snipper_msda_abi_version": ([], c_int),
    "snipper_msda_strerror": ([c_int], c_char_p),
    "snipper_msda_last_variant": ([], c_char_p),
    "snipper_msda_config_init": ([c_void_p], None),
    "snipper_msda_forward_ex": ([c_void_p] * 4 + [c_int] + [c_void_p] * 4 + [c_int] * 7 + [c_void_p, c_int], c_int),
    "snipper_msda_backward_ex_workspace_bytes": ([c_void_p, c_void_p] + [c_int] * 8, c_longlong),
    "snipper_msda_backward_ex": ([c_void_p] * 4 + [c_longlong, c_void_p, c_int, c_void_p, c_int] + [c_void_p] * 4 +
                                 [c_int] * 7 + [c_void_p] * 3, c_int),
    "snipper_msda_forward_f32": (_FWD_ARGS, c_int),
    "snipper_msda_forward_f64": (_FWD_ARGS, c_int),
    "snipper_msda_forward_bf16": (_FWD_ARGS, c_int),
    "snipper_msda_backward_f32": (_BWD_ARGS, c_int),
    "snipper_msda_backward_f64": (_BWD_ARGS, c_int),
    "snipper_msda_backward_bf16": (_BWD_ARGS, c_int),
    "snipper_temporal_mix": ([c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int,
                              ctypes.c_longlong, c_int, c_void_p, c_int], c_int),
    "snipper_temporal_mix_ex": ([c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int,
                                 ctypes.c_longlong, c_int, c_void_p, c_int, c_int, c_int, c_int], c_int),
    "snipper_msda_prologue_forward": ([c_void_p, c_void_p, c_longlong, c_void_p, c_longlong, c_int, c_void_p, c_void_p,
                                       c_void_p, c_longlong, c_int, c_int, c_int, c_void_p, c_void_p], c_int),
    "snipper_msda_prologue_forward_ex": ([c_void_p, c_void_p, c_longlong, c_void_p, c_longlong, c_int, c_void_p, c_void_p, c_void_p,
                                          c_void_p, c_longlong, c_int, c_int, c_int, c_void_p, c_void_p], c_int),
    "snipper_msda_prologue_backward": ([c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_longlong,
                                        c_int, c_int, c_int, c_void_p, c_longlong, c_void_p, c_longlong, c_int,
                                        c_void_p], c_int),
    "snipper_wgrad_workspace_bytes": ([c_int] * 3, c_size_t),
    "snipper_wgrad_bf16": ([c_void_p, c_void_p, c_longlong, c_void_p, c_longlong, c_int, c_int, c_int, c_void_p,
                            c_void_p, c_longlong, c_void_p, c_int, c_void_p, c_size_t], c_int),
    "snipper_add_dropout_layernorm_forward": ([c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p,
                                               c_void_p, c_int, c_int, ctypes.c_float, ctypes.c_float,
                                               ctypes.c_uint64] + [c_void_p] * 7, c_int),
    "snipper_add_dropout_layernorm_forward_ex": ([c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                                  c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, ctypes.c_float,
                                                  ctypes.c_float, ctypes.c_uint64] + [c_void_p] * 7, c_int),
    "snipper_add_dropout_layernorm_workspace_bytes": ([c_int, c_int], c_size_t),
    "snipper_add_dropout_layernorm_backward": ([c_void_p] * 9 + [c_int, c_int, ctypes.c_float, c_void_p, c_int,
                                                c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_size_t], c_int),
    "snipper_groupnorm_tokens_workspace_bytes": ([c_int] * 4, c_size_t),
    "snipper_groupnorm_tokens_forward": ([c_void_p] * 4 + [c_int] * 4 + [ctypes.c_float, c_longlong, c_longlong,
                                          c_void_p, c_int] + [c_void_p] * 5 + [c_size_t], c_int),
    "snipper_groupnorm_tokens_backward": ([c_void_p] * 7 + [c_int] * 4 + [c_longlong, c_longlong] + [c_void_p] * 4 +
                                           [c_size_t], c_int),
    "snipper_colsum_workspace_bytes": ([c_int] * 3, c_size_t),
    "snipper_colsum_segments_bf16": ([c_void_p, c_void_p, c_longlong, c_int, c_int, c_int, c_void_p, c_void_p,
                                      c_size_t], c_int),
    "snipper_colsum_segments_multi_bf16": ([c_void_p, c_void_p, c_int, c_longlong, c_longlong, c_int, c_int, c_int, c_void_p, c_void_p,
                                            c_size_t], c_int),
    "snipper_sum_bf16": ([c_void_p, c_void_p, c_int, c_void_p, c_longlong], c_int),
    "snipper_stem_pack_bf16": ([c_void_p, c_void_p, c_int, c_int, c_int, c_void_p], c_int),
    "snipper_conv3x3_bf16": ([c_void_p] * 5 + [c_int] * 7 + [c_void_p, c_int], c_int),
    "snipper_stem7x7_bf16": ([c_void_p] * 4 + [c_int] * 3, c_int),
    "snipper_conv3x3_dgrad_s2_bf16": ([c_void_p] * 4 + [c_int] * 5 + [c_void_p], c_int),
    "snipper_wgrad_conv3x3_workspace_bytes": ([c_int] * 6, c_size_t),
    "snipper_wgrad_conv3x3_bf16": ([c_void_p] * 3 + [c_int] * 6 + [c_void_p, c_void_p, c_int, c_void_p, c_size_t], c_int),
    "snipper_st_msda_forward": ([c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_longlong, c_void_p, c_longlong,
                                 c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p] + [c_int] * 9 +
                                [c_void_p] * 4 + [c_int], c_int),
    "snipper_st_msda_backward_workspace_bytes": ([c_int] * 8 + [c_void_p], c_size_t),
    "snipper_st_msda_backward": ([c_void_p, c_void_p, c_int] + [c_void_p] * 10 + [c_int] * 9 + [c_void_p, c_size_t,
                                 c_void_p, c_int, c_void_p, c_longlong, c_void_p, c_longlong, c_int, c_void_p], c_int),
    "snipper_pair_losses_forward": ([c_void_p] * 7 + [c_int] * 4 + [ctypes.c_float, c_void_p], c_int),
    "snipper_pair_losses_backward": ([c_void_p] * 8 + [c_int] * 4 + [ctypes.c_float, c_void_p, c_void_p], c_int),
    "snipper_match_cost_f32": ([c_void_p, c_void_p, c_longlong, c_longlong, c_int, c_void_p, c_longlong, c_longlong, c_int, c_void_p,
                                c_longlong, c_longlong, c_void_p, c_void_p, c_void_p] + [c_int] * 5 + [c_void_p, ctypes.c_float,
                                                                                                     c_void_p], c_int),
    "snipper_refine_reference_f32": ([c_void_p, c_void_p, c_longlong, c_void_p, c_void_p, c_int, c_int, c_int, ctypes.c_float,
                                      c_void_p, c_void_p], c_int),
    "snipper_stem_pool_bf16": ([c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p], c_int),
    "snipper_lsap_f32": ([c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p], c_int),
    "snipper_linear_bf16": ([c_void_p, c_void_p, ctypes.c_longlong, c_void_p, c_void_p, c_void_p, ctypes.c_longlong,
                             c_void_p, ctypes.c_longlong, c_int, c_int, c_int, c_int, ctypes.c_float,
                             ctypes.c_uint64], c_int),
    "snipper_linear_wres_supported": ([c_int, c_int, c_int], c_int),
    "snipper_linear_wres_bf16": ([c_void_p, c_void_p, c_longlong, c_void_p, c_longlong, c_void_p, c_void_p, c_longlong,
                                  ctypes.c_float, c_void_p, c_longlong, c_int, c_int, c_int, c_int, ctypes.c_float,
                                  ctypes.c_uint64], c_int),
    "snipper_hbm_copy_probe": ([c_void_p, c_void_p, c_void_p, c_longlong], c_int),
    "snipper_gradnorm_partials_f32": ([c_void_p, c_void_p, c_longlong, c_void_p, c_int], c_int),
    "snipper_adamw_clip_f32": ([c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_longlong, c_void_p, c_void_p, c_void_p,
                                c_void_p, c_int, ctypes.c_float, ctypes.c_float, ctypes.c_float, c_longlong, c_void_p, c_int,
                                ctypes.c_float, c_void_p], c_int),
    "snipper_transpose_batch_bf16": ([c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p], c_int),
    "snipper_linear_nn_bf16": ([c_void_p, c_void_p, c_longlong, c_void_p, c_longlong, c_void_p, c_longlong, c_void_p,
                                c_longlong, ctypes.c_float, c_void_p, c_longlong, c_int, c_int, c_int], c_int),
    "snipper_small_attention_forward_f32": ([c_void_p] + [c_void_p, c_longlong, c_longlong] * 4 + [c_void_p] + [c_int] * 4 +
                                             [ctypes.c_float, ctypes.c_float, ctypes.c_uint64], c_int),
    "snipper_small_attention_backward_f32": ([c_void_p] + [c_void_p, c_longlong, c_longlong] * 4 + [c_void_p] +
                                              [c_void_p, c_longlong, c_longlong] * 4 + [c_int] * 4 +
                                              [ctypes.c_float, ctypes.c_float, ctypes.c_uint64], c_int),
    "snipper_heatmap_blur_f32": ([c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, ctypes.c_float], c_int),
    "snipper_small_gemm_batch_f32": ([c_void_p, c_void_p, c_int], c_int),
    "snipper_small_linear_forward_f32": ([c_void_p, c_void_p, c_longlong, c_void_p, c_longlong, c_void_p, c_int, c_int, c_int,
                                          c_void_p, c_longlong], c_int),
    "snipper_small_linear_backward_f32": ([c_void_p, c_void_p, c_longlong, c_void_p, c_longlong, c_void_p, c_longlong,
                                           c_int, c_int, c_int, c_void_p, c_longlong, c_void_p, c_longlong, c_void_p], c_int),
    "snipper_conv3x3_pack_bf16": ([c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p], c_int),
    "snipper_conv3x3_patch_supported": ([c_int] * 5, c_int),
    "snipper_conv3x3_patch_bf16": ([c_void_p, c_void_p, c_void_p, c_void_p, c_void_p] + [c_int] * 6 + [c_void_p], c_int),
    "snipper_linear_pack_bf16": ([c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p], c_int),
    "snipper_linear_patch_supported": ([c_longlong, c_int, c_int], c_int),
    "snipper_linear_patch_bf16": ([c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                   c_void_p, c_int], c_int),
    "snipper_linear_wide_supported": ([c_longlong, c_int, c_int], c_int),
    "snipper_linear_wide_bf16": ([c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int], c_int),
    "snipper_level_pos_bf16": ([c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p], c_int),
    "snipper_cast_scale_table_bf16": ([c_void_p, c_void_p, c_void_p, c_int, c_int], c_int),
    "snipper_heatmap_scatter_f32": ([c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                     c_void_p], c_int),
    "snipper_heatmap_loss_forward_f32": ([c_void_p, c_void_p, c_void_p, c_void_p, c_void_p] + [c_int] * 7 + [c_void_p, c_int], c_int),
    "snipper_heatmap_loss_backward_f32": ([c_void_p, c_void_p, c_void_p, c_void_p, c_void_p] + [c_int] * 7 + [c_void_p, c_void_p], c_int),
    "snipper_msda_backward_sparse_bf16": ([c_void_p] * 7 + [c_int] * 7 + [c_void_p] * 3, c_int),
    "snipper_msda_backward_sparse_f32rows": ([c_void_p] * 7 + [c_int] * 7 + [c_void_p] * 3, c_int),
    "snipper_sum_f32": ([c_void_p, c_void_p, c_int, c_void_p, c_longlong], c_int),
    "snipper_small_ln_forward_f32": ([c_void_p] * 6 + [c_int, c_int, ctypes.c_float, ctypes.c_float, ctypes.c_uint64] +
                                     [c_void_p] * 6, c_int),
    "snipper_small_ln_backward_f32": ([c_void_p] * 10 + [c_int, c_int, ctypes.c_float] + [c_void_p] * 4, c_int),
    "snipper_refine_reference_linear_f32": ([c_void_p] * 6 + [c_int] * 4 + [ctypes.c_float, c_void_p, c_void_p], c_int),
    "snipper_layers_abi_version": ([], c_int),
    "snipper_decoder_layer_supported": ([c_char_p], c_int),
    "snipper_decoder_layer_arena_bytes": ([c_char_p], c_size_t),
    "snipper_decoder_layer_scratch_bytes": ([c_char_p], c_size_t),
    "snipper_decoder_layer_forward": ([c_void_p, c_char_p], c_int),
    "snipper_decoder_layer_backward": ([c_void_p, c_char_p], c_int),
    "snipper_encoder_layer_supported": ([c_char_p], c_int),
    "snipper_encoder_layer_arena_bytes": ([c_char_p], c_size_t),
    "snipper_encoder_layer_scratch_bytes": ([c_char_p, c_void_p, c_void_p], c_size_t),
    "snipper_encoder_layer_forward": ([c_void_p, c_char_p], c_int),
    "snipper_encoder_layer_backward": ([c_void_p, c_char_p], c_int),
    "snipper_relu_dropout_backward_bf16": ([c_void_p, c_void_p, c_void_p, c_void_p, c_longlong, ctypes.c_float], c_int),
}

ABI_VERSION = 2


class SnipperLibraryError(RuntimeError):
    pass


def lib_path() -> str:
    return _build.LIB_PATH


def load():
    """Load (building first if the .so is absent and hipcc is present) and type the library."""
    global _lib
    if _lib is not None:
        return _lib
    path = _build.LIB_PATH
    if not os.path.exists(path):
        try:
            _build.build_hip()
        except Exception as e:  # no silent fallback: the op is unusable without its HIP library
            raise SnipperLibraryError(
                f"libsnipper_msda.so is missing at {path} and could not be built ({e}). "
                "Run `python -m snipper_amd.build` (needs hipcc, gfx950).") from e
    try:
        lib = ctypes.CDLL(path)
    except OSError as e:
        raise SnipperLibraryError(f"cannot load {path}: {e}") from e
    for name, (argtypes, restype) in EXPORTS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise SnipperLibraryError(f"{path} does not export {name}") from e
        fn.argtypes = argtypes
        fn.restype = restype
    got = lib.snipper_msda_abi_version()
    if got != ABI_VERSION:
        raise SnipperLibraryError(f"ABI mismatch: library {got}, binding {ABI_VERSION}")
    _lib = lib
    return lib


E_UNSUPPORTED = -3      # include/snipper_msda.h


class _NoGuard:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


_NO_GUARD = _NoGuard()


def device_guard(device):
    """``torch.cuda.device(device)`` when that would change the current device, a no-op context otherwise (the usual
    case: the guard object, its index parsing and two device exchanges cost ~6 us per kernel wrapper call)."""
    import torch
    idx = device.index
    if idx is None or idx == torch.cuda.current_device():
        return _NO_GUARD
    return torch.cuda.device(device)


def raw_stream(device) -> int:
    """The current stream of ``device`` as a raw handle, without building a ``torch.cuda.Stream`` object."""
    import torch
    idx = device.index
    return torch._C._cuda_getCurrentRawStream(torch.cuda.current_device() if idx is None else idx)


def check(code: int, what: str) -> None:
    if code != 0:
        msg = load().snipper_msda_strerror(code).decode()
        raise RuntimeError(f"{what} failed: {msg} (code {code})")


_last_variant = "none"


def note_variant() -> None:
    """Record, on the Python side, the variant of the core-op call THIS thread has just made (the C diagnostic is thread-local;
    the wrappers call this right after every core-op entry point, so a test on the main thread can ask about a backward that
    ran on the autograd engine's thread)."""
    global _last_variant
    _last_variant = load().snipper_msda_last_variant().decode()


def last_variant() -> str:
    """Kernel variant of the most recent core-op call made through this package's wrappers, on whichever thread."""
    return _last_variant


class SmallGemm(ctypes.Structure):
    """include/snipper_dense.h: snipper_small_gemm (one product of a snipper_small_gemm_batch_f32 launch)."""
    _fields_ = [("A", c_void_p), ("lda", c_longlong), ("a_transposed", c_int),
                ("B", c_void_p), ("ldb", c_longlong), ("b_transposed", c_int),
                ("out", c_void_p), ("ldo", c_longlong), ("bias", c_void_p), ("colsum", c_void_p),
                ("I", c_int), ("J", c_int), ("R", c_int),
                ("B2", c_void_p), ("ldb2", c_longlong), ("r_split", c_int),
                ("relu", c_int), ("dropout_p", ctypes.c_float), ("seed", ctypes.c_ulonglong),
                ("gate", c_void_p), ("ldgate", c_longlong), ("gate_scale", ctypes.c_float)]


class Config(ctypes.Structure):
    """``snipper_msda_config`` (include/snipper_msda.h).  The library itself keeps no tuning state; the wrappers in
    MultiScaleDeformableAttention.py pass the caller's Config -- or ``None`` (the library defaults) -- with every call."""
    _fields_ = [("struct_bytes", ctypes.c_int32), ("policy", ctypes.c_int32), ("near_radius", ctypes.c_float),
                ("tile_kernel", ctypes.c_int32), ("tile_edge", ctypes.c_int32 * 3), ("debug_ablation", ctypes.c_int32),
                ("value_layout", ctypes.c_int32), ("reserved", ctypes.c_int32 * 3)]

    @classmethod
    def defaults(cls) -> "Config":
        c = cls()
        load().snipper_msda_config_init(ctypes.byref(c))
        return c


# Test / benchmark hook (Python side only): a Config the wrappers use when the caller passes none.  ``None`` = library
# defaults, which is what every product path runs with.
_test_config = None

_KNOBS = {"near_radius": ("near_radius", float),
          "owner_tile_edge_big": (("tile_edge", 0), int), "owner_tile_edge_mid": (("tile_edge", 1), int),
          "owner_tile_edge_small": (("tile_edge", 2), int), "debug": ("debug_ablation", int), "tile_kernel": ("tile_kernel", int), "value_layout": ("value_layout", int)}


def active_config():
    return _test_config


def reset_config() -> None:
    global _test_config
    _test_config = None


def _ensure_test_config() -> Config:
    global _test_config
    if _test_config is None:
        _test_config = Config.defaults()
    return _test_config


def set_policy(policy: int) -> None:
    """0 auto, 1 generic kernels only, 2 tuned D=48 kernels but never the encoder-shape ones (tests / benchmarks)."""
    if int(policy) not in (0, 1, 2):
        raise RuntimeError(f"snipper_msda policy {policy} not supported")
    _ensure_test_config().policy = int(policy)


def set_param(name: str, value: float) -> None:
    """Tests / benchmarks: change one field of the Python-side test Config ("owner_enable" 0 = policy 2)."""
    c = _ensure_test_config()
    if name == "owner_enable":
        c.policy = 0 if value else 2
        return
    if name not in _KNOBS:
        raise RuntimeError(f"snipper_msda_set_param({name}) failed: no such knob")
    field, typ = _KNOBS[name]
    if isinstance(field, tuple):
        getattr(c, field[0])[field[1]] = typ(value)
    else:
        setattr(c, field, typ(value))
