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
LIB_PATH = os.path.join(PKG_DIR, "libsnipper_msda.so")
SOURCES = ["msda_capi.hip"]
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
               "-Wall", "-Wno-cuda-compat", "-fno-gpu-rdc"]


def _newest_source_mtime() -> float:
    mt = 0.0
    for root in (CSRC, os.path.join(os.path.dirname(PKG_DIR), "include")):
        for f in os.listdir(root):
            if f.endswith((".hip", ".cuh", ".h")):
                mt = max(mt, os.path.getmtime(os.path.join(root, f)))
    return mt


def hipcc_path():
    return shutil.which("hipcc") or ("/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else None)


def build_hip(force: bool = False, verbose: bool = False) -> str:
    """Compile csrc/*.hip into LIB_PATH if it is missing or older than its sources."""
    if not force and os.path.exists(LIB_PATH) and os.path.getmtime(LIB_PATH) >= _newest_source_mtime():
        return LIB_PATH
    hipcc = hipcc_path()
    if hipcc is None:
        raise RuntimeError("hipcc not found: cannot build libsnipper_msda.so (gfx950)")
    cmd = [hipcc] + HIPCC_FLAGS + ["-o", LIB_PATH + ".tmp"] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    os.replace(LIB_PATH + ".tmp", LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    print(build_hip(force=True, verbose=True))
