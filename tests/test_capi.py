"""The C-ABI library builds, loads and exports every symbol include/*.h declares (no compute)."""
import ctypes
import os
import re

import snipper_amd
from snipper_amd import _lib, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = set()
    inc = os.path.join(ROOT, "include")
    for f in os.listdir(inc):
        text = open(os.path.join(inc, f)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"\b(snipper_\w+)\s*\(", text))
    return names


def test_header_symbols_are_exported():
    path = build.build_hip()
    lib = ctypes.CDLL(path)
    syms = declared_symbols()
    assert len(syms) >= 10
    for s in sorted(syms):
        assert hasattr(lib, s), f"{s} declared in include/ but not exported by {path}"
    assert syms == set(_lib.EXPORTS), "ctypes binding and header disagree"


def test_library_loads_and_reports():
    lib = _lib.load()
    assert lib.snipper_msda_abi_version() == _lib.ABI_VERSION
    assert lib.snipper_msda_strerror(0) == b"ok"
    assert b"null" in lib.snipper_msda_strerror(-1)
    # argument validation happens before any HIP call, so it is testable without a GPU
    assert lib.snipper_msda_forward_f32(None, None, None, None, None, None, 1, 1, 1, 1, 1, 1, 1, None) == -1
    one = ctypes.c_void_p(8)
    assert lib.snipper_msda_forward_f32(None, one, one, one, one, one, 0, 1, 1, 1, 1, 1, 1, one) == -2
    assert lib.snipper_msda_forward_f64(None, one, one, one, one, one, 1, 1 << 20, 64, 64, 1, 1, 1, one) == -2
    cfg = _lib.Config.defaults()
    assert cfg.struct_bytes == ctypes.sizeof(_lib.Config) and cfg.policy == 0 and cfg.near_radius == 24.0
    assert list(cfg.tile_edge) == [0, 0, 0]          # 0 = the grad_value-side kernel's own choice
    cfg.policy = 7      # a config the library rejects is an error, not a silent default
    assert lib.snipper_msda_forward_ex(None, ctypes.byref(cfg), None, one, 0, one, one, one, one, 1, 1, 1, 1, 1, 1, 1, one, 0) == -2


def test_install_registers_reference_module_name():
    import sys
    snipper_amd.install()
    import MultiScaleDeformableAttention as MSDA
    assert hasattr(MSDA, "ms_deform_attn_forward") and hasattr(MSDA, "ms_deform_attn_backward")
    assert sys.modules["MultiScaleDeformableAttention"] is MSDA


def test_reference_import_line_binds_without_install():
    """``import MultiScaleDeformableAttention as MSDA`` (reference ms_deform_attn_func.py:18-21) resolves through the
    repository-root alias in a fresh interpreter that never calls ``snipper_amd.install()``; and, where the reference
    tree is present (build container only), the reference's own unmodified function file ends up with that module."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import MultiScaleDeformableAttention as MSDA\n"
            "assert MSDA.ms_deform_attn_forward.__module__ == 'snipper_amd.MultiScaleDeformableAttention'\n"
            "import os\n"
            "if os.path.isdir('/root/reference/models/ops/functions'):\n"
            "    sys.dont_write_bytecode = True\n"
            "    sys.path.insert(0, '/root/reference/models/ops')\n"
            "    from functions import ms_deform_attn_func as F\n"
            "    assert F.MSDA is MSDA\n"
            "print('ok')\n") % ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd="/tmp",
                         env=dict(os.environ, PYTHONDONTWRITEBYTECODE="1"))
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stderr[-2000:]


def test_config_reserved_fields_and_owner_map_bounds_are_checked():
    """the header's "must be 0" is enforced (a hand-filled, un-zeroed struct must not reach the kernels; the
    wrong-result timing ablations behind debug_ablation need SNIPPER_MSDA_ALLOW_DEBUG=1 at load time), and maps with H or
    W >= 32768 (the hit records pack qy / qx in 15 bits each) do not take the owner-computes path.  Host-only calls."""
    lib = _lib.load()
    import numpy as np
    hs = np.array([[75, 100], [38, 50], [19, 25]], dtype=np.int64)
    S = int((hs[:, 0] * hs[:, 1]).sum())
    hp = hs.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))
    ws = lambda cfg, hp=hp, S=S, L=3: lib.snipper_msda_backward_ex_workspace_bytes(
        None if cfg is None else ctypes.byref(cfg), hp, 0, 2, S, 8, 48, L, S, 4)
    assert ws(None) > 0 and ws(_lib.Config.defaults()) == ws(None)
    for field, idx in (("tile_kernel", None), ("value_layout", None), ("reserved", 0), ("reserved", 2)):
        cfg = _lib.Config.defaults()
        if idx is None:
            setattr(cfg, field, 3)          # (tile_kernel: 0 / 2 matrix pipe, 1 vector kernel; value_layout 0 / 1; anything else is refused)
        else:
            getattr(cfg, field)[idx] = 1
        assert ws(cfg) == 0, (field, idx)
    cfg = _lib.Config.defaults()
    cfg.debug_ablation = 1
    if os.environ.get("SNIPPER_MSDA_ALLOW_DEBUG") != "1":
        assert ws(cfg) == 0
    wide = np.array([[1, 40000]], dtype=np.int64)
    assert ws(None, wide.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), 40000, 1) == 0
    ok = np.array([[1, 30000]], dtype=np.int64)
    assert ws(None, ok.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), 30000, 1) > 0


# ---- argument lists: header prototype <-> ctypes tuple, parameter by parameter --------------------------
def _prototypes():
    """{name: (return class, [parameter classes])} parsed from include/*.h.  Classes: 'p' pointer, 'q' 64-bit integer,
    'i' 32-bit integer, 'f' float, 'd' double, 'v' void (return only)."""
    def klass(decl: str) -> str:
        decl = decl.strip()
        if "*" in decl or "[" in decl:
            return "p"
        words = [w for w in re.split(r"\s+", decl) if w not in ("const", "unsigned", "signed", "volatile")]
        # drop the parameter name (the last identifier) when a type word precedes it
        types = words[:-1] if len(words) > 1 else words
        t = " ".join(types)
        if t in ("long long", "int64_t", "uint64_t", "size_t", "long long int"):
            return "q"
        if t in ("int", "int32_t", "uint32_t"):
            return "i"
        if t == "float":
            return "f"
        if t == "double":
            return "d"
        if t == "void":
            return "v"
        raise AssertionError(f"unclassified C type in a prototype: {decl!r}")

    protos = {}
    inc = os.path.join(ROOT, "include")
    for f in sorted(os.listdir(inc)):
        text = open(os.path.join(inc, f)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        text = re.sub(r"//[^\n]*", "", text)
        text = re.sub(r"^\s*#[^\n]*", "", text, flags=re.M)             # preprocessor lines
        for m in re.finditer(r"([A-Za-z_][\w \t\*]*?)\b(snipper_\w+)\s*\(([^()]*)\)\s*;", text):
            ret, name, params = m.group(1).strip(), m.group(2), m.group(3).strip()
            rclass = "p" if "*" in ret else klass(ret + " x")
            plist = [] if params in ("", "void") else [klass(q) for q in params.split(",")]
            protos[name] = (rclass, plist)
    return protos


def _ctypes_class(t) -> str:
    if t is None:
        return "v"
    if t in (ctypes.c_void_p, ctypes.c_char_p) or isinstance(t, type(ctypes.POINTER(ctypes.c_int))):
        return "p"
    if t in (ctypes.c_longlong, ctypes.c_ulonglong, ctypes.c_int64, ctypes.c_uint64, ctypes.c_size_t, ctypes.c_ssize_t):
        return "q"
    if t in (ctypes.c_int, ctypes.c_uint, ctypes.c_int32, ctypes.c_uint32):
        return "i"
    if t is ctypes.c_float:
        return "f"
    if t is ctypes.c_double:
        return "d"
    raise AssertionError(f"unclassified ctypes type {t}")


def test_ctypes_argument_lists_match_the_header_prototypes():
    """Every entry of ``_lib.EXPORTS`` has the parameter COUNT and, parameter by parameter, the WIDTH CLASS (pointer / 64-bit
    integer / 32-bit integer / float) of its prototype in include/*.h, and the same return class: a drifted ``c_int`` /
    ``c_longlong`` would otherwise be silent until it corrupts a launch."""
    protos = _prototypes()
    assert set(protos) == set(_lib.EXPORTS), sorted(set(protos) ^ set(_lib.EXPORTS))
    bad = []
    for name, (argtypes, restype) in sorted(_lib.EXPORTS.items()):
        rclass, plist = protos[name]
        got = [_ctypes_class(t) for t in argtypes]
        if got != plist:
            first = next((i for i, (a, b) in enumerate(zip(got, plist)) if a != b), min(len(got), len(plist)))
            bad.append(f"{name}: header {''.join(plist)} ({len(plist)}) vs ctypes {''.join(got)} ({len(got)}), first difference at "
                       f"parameter {first}")
        if _ctypes_class(restype) != rclass:
            bad.append(f"{name}: return class header {rclass} vs ctypes {_ctypes_class(restype)}")
    assert not bad, "\n".join(bad)
