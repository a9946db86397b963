"""Build the gfx950 HIP library in-tree (snipper_amd/libsnipper_msda.so).

hipcc cross-compiles without a GPU, so this runs in the build container; the .so is
git-ignored but travels to the GPU box with the repo snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
# (diagnostic builds: SNIPPER_MSDA_LIB = another file name in this directory, SNIPPER_HIPCC_EXTRA = extra compiler flags,
#  e.g. -DWRES_STAMPS for the in-kernel cycle stamps of tools/wres_stamps.py; the product never sets them)
LIB_PATH = os.path.join(PKG_DIR, os.path.basename(os.environ.get("SNIPPER_MSDA_LIB", "libsnipper_msda.so")))
SOURCES = ["msda_capi.hip"]
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
               "-Wall", "-Wno-cuda-compat", "-fno-gpu-rdc"] + os.environ.get("SNIPPER_HIPCC_EXTRA", "").split()


HASH_PATH = LIB_PATH + ".srchash"


def source_hash() -> str:
    """SHA-256 over every source the library is built from (csrc/*.hip, *.cuh, include/*.h) and the compiler flags: the
    library is rebuilt whenever this differs from the hash recorded beside it (file times do not survive a checkout or
    a snapshot copy)."""
    import hashlib
    h = hashlib.sha256(" ".join(HIPCC_FLAGS).encode())
    for root in (CSRC, os.path.join(os.path.dirname(PKG_DIR), "include")):
        for f in sorted(os.listdir(root)):
            if f.endswith((".hip", ".cuh", ".h")):
                h.update(f.encode())
                with open(os.path.join(root, f), "rb") as fh:
                    h.update(fh.read())
    return h.hexdigest()


def is_current() -> bool:
    try:
        return os.path.exists(LIB_PATH) and open(HASH_PATH).read().strip() == source_hash()
    except OSError:
        return False


def hipcc_path():
    return shutil.which("hipcc") or ("/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else None)


def build_hip(force: bool = False, verbose: bool = False) -> str:
    """Compile csrc/*.hip into LIB_PATH if it is missing or was built from different sources (source_hash)."""
    if not force and is_current():
        return LIB_PATH
    # the timing-ablation builds (-DWRES_STAMPS, -DTILE2_STAMPS: in-kernel stamps, slower kernels) must never become the
    # product library: they need a library name of their own (SNIPPER_MSDA_LIB)
    if any(f.startswith("-D") and "STAMPS" in f for f in HIPCC_FLAGS) and os.path.basename(LIB_PATH) == "libsnipper_msda.so":
        raise RuntimeError("diagnostic build flags (…_STAMPS) need SNIPPER_MSDA_LIB=<another file name>")
    hipcc = hipcc_path()
    if hipcc is None:
        if os.path.exists(LIB_PATH):         # a box without the compiler uses the library that travelled with the tree
            if not is_current() and os.environ.get("SNIPPER_ALLOW_STALE_LIB") != "1":
                # kernels' argument structs / workspace layouts / exports change with the sources: a stale library fails
                # with an opaque missing symbol at best and runs old kernels against new host code at worst
                raise RuntimeError(
                    f"{LIB_PATH} was built from other sources than the tree's (source hash mismatch) and hipcc is not "
                    "available to rebuild it; set SNIPPER_ALLOW_STALE_LIB=1 to load it anyway")
            return LIB_PATH
        raise RuntimeError("hipcc not found: cannot build libsnipper_msda.so (gfx950)")
    cmd = [hipcc] + HIPCC_FLAGS + ["-o", LIB_PATH + ".tmp"] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    os.replace(LIB_PATH + ".tmp", LIB_PATH)
    with open(HASH_PATH, "w") as fh:
        fh.write(source_hash() + "\n")
    return LIB_PATH


if __name__ == "__main__":
    print(build_hip(force=True, verbose=True))
