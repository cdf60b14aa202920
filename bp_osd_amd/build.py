"""Build libbposd_mi355x.so in-tree with hipcc for gfx950 (no cmake, no JIT cache)."""
from __future__ import annotations

import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB = os.path.join(_HERE, "libbposd_mi355x.so")
SOURCES = ["bposd_capi.hip"]
HEADERS = ["portable_math.h", "local_layout.h", "class_layout.h", "bp_class_kernel.hip.h", "bp_anydeg_kernel.hip.h", "bp_own_kernel.hip.h", "own_layout.h", "osd_wave_kernel.hip.h", "bp_kernel.hip.h", "bp_local_kernel.hip.h", "bp_serial_kernel.hip.h", "bp_large_kernel.hip.h", "osd_kernel.hip.h", "osd_large_kernel.hip.h", os.path.join("..", "..", "include", "bposd_mi355x.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-pthread"]


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    for f in SOURCES + HEADERS:
        p = os.path.join(CSRC, f)
        if os.path.exists(p) and os.path.getmtime(p) > t:
            return True
    return False


def build_library(force: bool = False, verbose: bool = False) -> str:
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not force and not _stale():
        return LIB
    if not os.path.exists(hipcc):
        if os.path.exists(LIB):
            return LIB  # prebuilt .so travelled with the snapshot
        raise RuntimeError("hipcc not found and libbposd_mi355x.so has not been built")
    cmd = [hipcc] + FLAGS + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    return LIB
