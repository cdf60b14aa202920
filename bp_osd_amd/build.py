"""Build libbposd_mi355x.so in-tree with hipcc for gfx950 (no cmake, no JIT cache).

One translation unit per kernel family (csrc/launch_*.hip) next to the C-ABI (csrc/bposd_capi.hip); the units compile in
parallel and only the stale ones are rebuilt (objects under csrc/_obj/, git-ignored).

Diagnostic / A-B builds (BPOSD_EXTRA_FLAGS="-DBPOSD_OSD_DIAG ...") never touch the product library: their objects go to
csrc/_obj_<hash of the flags>/ and the library to bp_osd_amd/_variants/libbposd_mi355x_<hash>.so (or $BPOSD_LIB_OUT); load
one with BPOSD_LIB=<path>.  A file lock serialises concurrent builders (two torchrun ranks on a stale tree)."""
from __future__ import annotations

import fcntl
import hashlib
import os
import re
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(_HERE, "libbposd_mi355x.so")
SOURCES = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
PUBLIC_HEADER = os.path.join(_HERE, "..", "include", "bposd_mi355x.h")
CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-fvisibility=hidden", "-pthread"]
EXTRA = os.environ.get("BPOSD_EXTRA_FLAGS", "").split()
if EXTRA:  # a build with extra flags is a different library: objects and output of its own
    _TAG = hashlib.sha1(" ".join(EXTRA).encode()).hexdigest()[:10]
    OBJ = os.path.join(CSRC, "_obj_" + _TAG)
    LIB = os.environ.get("BPOSD_LIB_OUT") or os.path.join(_HERE, "_variants", f"libbposd_mi355x_{_TAG}.so")


def _deps(path: str, seen=None) -> set:
    """The file and every header of csrc/ it includes, transitively."""
    seen = set() if seen is None else seen
    if path in seen or not os.path.exists(path):
        return seen
    seen.add(path)
    with open(path) as f:
        for inc in re.findall(r'^\s*#include\s+"([^"]+)"', f.read(), flags=re.M):
            _deps(os.path.normpath(os.path.join(os.path.dirname(path), inc)), seen)
    return seen


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force: bool = False, verbose: bool = False, jobs: int | None = None) -> str:
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    objs = [os.path.join(OBJ, s[:-4] + ".o") for s in SOURCES]
    alldeps = set().union(*[_deps(s) for s in srcs]) | {__file__}
    if not force and not _stale(LIB, alldeps):
        return LIB  # (also the case on the GPU box: the built library travels with the snapshot, the objects do not)
    if not os.path.exists(hipcc):
        if os.path.exists(LIB):
            return LIB  # prebuilt .so travelled with the snapshot
        raise RuntimeError("hipcc not found and libbposd_mi355x.so has not been built")
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    with open(os.path.join(OBJ, ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)  # one builder at a time per object directory
        if not force and not _stale(LIB, alldeps):
            return LIB  # another process built it while this one waited
        return _build_locked(hipcc, srcs, objs, force, verbose, jobs)


def _build_locked(hipcc, srcs, objs, force, verbose, jobs) -> str:
    todo = [(s, o) for s, o in zip(srcs, objs) if force or _stale(o, _deps(s) | {__file__})]

    def compile_one(so):
        cmd = [hipcc] + CFLAGS + EXTRA + ["-c", so[0], "-o", so[1]]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd, cwd=CSRC)

    with ThreadPoolExecutor(max_workers=jobs or min(8, os.cpu_count() or 1)) as ex:
        list(ex.map(compile_one, todo))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread"] + objs + ["-o", LIB]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd, cwd=CSRC)
    return LIB
