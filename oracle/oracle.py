"""ctypes binding of oracle/_build/libbposd_oracle.so (TEST INFRASTRUCTURE ONLY).

The C restatement follows SURVEY.md Appendix A; option names mirror the reference's
ctor kwargs (/root/reference/README.md:178-187,
/root/reference/src/bposd/css_decode_sim.py:444-463).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np
import scipy.sparse as sp

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libbposd_oracle.so")
_lib = None


class _Cfg(C.Structure):
    _fields_ = [
        ("bp_method", C.c_int32),
        ("ms_scaling_factor", C.c_double),
        ("max_iter", C.c_int32),
        ("osd_method", C.c_int32),
        ("osd_order", C.c_int32),
        ("sort_tie_policy", C.c_int32),
        ("weight_fn", C.c_int32),
        ("ps_clip", C.c_double),
        ("schedule", C.c_int32),
        ("ps_math", C.c_int32),
        ("osd_e_bit_order", C.c_int32),
    ]


def build_oracle(force: bool = False) -> str:
    src = os.path.join(_HERE, "bposd_oracle.c")
    deps = [src, os.path.join(_HERE, "bposd_oracle.h"), os.path.join(_HERE, "..", "bp_osd_amd", "csrc", "portable_math.h")]
    stale = (not os.path.exists(_SO)) or any(os.path.exists(f) and os.path.getmtime(f) > os.path.getmtime(_SO) for f in deps)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s"], stdout=subprocess.DEVNULL)
    return _SO


def _load():
    global _lib
    if _lib is None:
        build_oracle()
        _lib = C.CDLL(_SO)
        vp = C.c_void_p
        _lib.oracle_create.argtypes = [C.POINTER(_Cfg), vp, vp, C.c_int32, C.c_int32, vp, C.POINTER(vp)]
        _lib.oracle_create.restype = C.c_int
        _lib.oracle_destroy.argtypes = [vp]
        _lib.oracle_update_channel_probs.argtypes = [vp, vp]
        _lib.oracle_rank.argtypes = [vp]
        _lib.oracle_num_candidates.argtypes = [vp]
        _lib.oracle_decode_batch.argtypes = [vp, vp, C.c_int64, vp, vp, vp, vp, vp, vp]
        _lib.oracle_decode_batch.restype = C.c_int
        _lib.oracle_decode_batch_diag.argtypes = [vp, vp, C.c_int64, vp, vp, vp, vp, vp, vp, vp, vp, vp]
        _lib.oracle_decode_batch_diag.restype = C.c_int
        _lib.oracle_osd.argtypes = [vp, vp, vp, vp, vp, vp, vp]
        _lib.oracle_osd.restype = C.c_int
    return _lib


# Threads decode_batch spreads a batch over (each on a handle of its own; the C code holds per-handle scratch).  1 by
# default -- bench.py's one-core CPU baseline times this class as it is; tests/conftest.py raises it so that the parity
# suites, whose run time is this oracle's, use the host's cores.
DEFAULT_THREADS = 1

_BP = {"product_sum": 0, "prod_sum": 0, "ps": 0, "0": 0, "minimum_sum": 1, "min_sum": 1, "ms": 1, "1": 1}
_OSD = {"osd_off": 0, "off": 0, "osd_0": 1, "osd0": 1, "0": 1, "osd_e": 2, "e": 2, "exhaustive": 2,
        "osd_cs": 3, "cs": 3, "combination_sweep": 3}


def portable_math(which, x):
    """tanh / log / expm1 of bp_osd_amd/csrc/portable_math.h (compiled into the oracle library) on an array."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.empty_like(x)
    lib = _load()
    lib.oracle_portable_math.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_int64]
    lib.oracle_portable_math.restype = None
    lib.oracle_portable_math({"tanh": 0, "log": 1, "expm1": 2}[which], x.ctypes.data, y.ctypes.data, x.size)
    return y


def portable_tanh_half(x):
    """tanh(x / 2) as the product-sum kernels evaluate it (portable_math.h: pm_tanh_half, one division)."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.empty_like(x)
    lib = _load()
    lib.oracle_portable_tanh_half.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    lib.oracle_portable_tanh_half.restype = None
    lib.oracle_portable_tanh_half(x.ctypes.data, y.ctypes.data, x.size)
    return y


def portable_log_quot(a, b):
    """log(a / b) as the product-sum kernels evaluate it (portable_math.h: pm_log_quot, one division)."""
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    y = np.empty_like(a)
    lib = _load()
    lib.oracle_portable_log_quot.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
    lib.oracle_portable_log_quot.restype = None
    lib.oracle_portable_log_quot(a.ctypes.data, b.ctypes.data, y.ctypes.data, a.size)
    return y


class OracleDecoder:
    def __init__(self, pcm, error_rate=None, channel_probs=None, max_iter=0, bp_method="ms",
                 ms_scaling_factor=1.0, osd_method="osd0", osd_order=0, sort_tie_policy=0, weight_fn=0,
                 ps_clip=0.0, ps_math=0, schedule="parallel", osd_e_bit_order=0):
        lib = _load()
        h = sp.csr_matrix(pcm).astype(np.uint8)
        h.eliminate_zeros()
        h.sort_indices()
        self.m, self.n = h.shape
        if channel_probs is None or (len(channel_probs) == 1 and channel_probs[0] is None):
            probs = np.full(self.n, float(error_rate), dtype=np.float64)
        else:
            probs = np.ascontiguousarray(channel_probs, dtype=np.float64)
        self._indptr = np.ascontiguousarray(h.indptr, dtype=np.int32)
        self._indices = np.ascontiguousarray(h.indices, dtype=np.int32)
        cfg = _Cfg(_BP[str(bp_method).lower()], float(ms_scaling_factor), int(max_iter),
                   _OSD[str(osd_method).lower()], int(osd_order), int(sort_tie_policy), int(weight_fn), float(ps_clip),
                   {"parallel": 0, "serial": 1}[str(schedule).lower()], int(ps_math), int(osd_e_bit_order))
        self._cfg = cfg
        self._probs = probs.copy()
        self._extra = []  # further handles of the same decoder, one per additional thread of decode_batch
        self._h = C.c_void_p()
        rc = lib.oracle_create(C.byref(cfg), self._indptr.ctypes.data, self._indices.ctypes.data,
                               self.m, self.n, probs.ctypes.data, C.byref(self._h))
        if rc != 0:
            raise ValueError(f"oracle_create failed ({rc})")
        self.rank = lib.oracle_rank(self._h)
        self.num_candidates = lib.oracle_num_candidates(self._h)

    def __del__(self):
        for h in getattr(self, "_extra", []):
            _load().oracle_destroy(h)
        self._extra = []
        if getattr(self, "_h", None) is not None and self._h.value:
            _load().oracle_destroy(self._h)
            self._h = None

    def update_channel_probs(self, probs):
        probs = np.ascontiguousarray(probs, dtype=np.float64)
        assert probs.shape == (self.n,)
        self._probs = probs.copy()
        for h in [self._h] + self._extra:
            _load().oracle_update_channel_probs(h, probs.ctypes.data)

    def _handles(self, k):
        while len(self._extra) < k - 1:
            h = C.c_void_p()
            rc = _load().oracle_create(C.byref(self._cfg), self._indptr.ctypes.data, self._indices.ctypes.data,
                                       self.m, self.n, self._probs.ctypes.data, C.byref(h))
            if rc != 0:
                raise ValueError(f"oracle_create failed ({rc})")
            self._extra.append(h)
        return [self._h] + self._extra[:k - 1]

    def decode_batch(self, syndromes, want_llr=True, want_diag=False):
        s = np.ascontiguousarray(np.asarray(syndromes) & 1, dtype=np.uint8)
        if s.ndim == 1:
            s = s[None, :]
        B = s.shape[0]
        assert s.shape[1] == self.m
        out = {
            "osdw": np.zeros((B, self.n), np.uint8),
            "osd0": np.zeros((B, self.n), np.uint8),
            "bp": np.zeros((B, self.n), np.uint8),
            "converged": np.zeros(B, np.uint8),
            "iters": np.zeros(B, np.int32),
            "llr": np.zeros((B, self.n), np.float64) if want_llr else None,
        }
        if want_diag:  # per-shot BP saturation record (product-sum): first non-finite iteration, final inf / NaN
            out["first_nonfinite_iter"] = np.zeros(B, np.int32)
            out["final_has_inf"] = np.zeros(B, np.uint8)
            out["final_has_nan"] = np.zeros(B, np.uint8)
            rc = _load().oracle_decode_batch_diag(
                self._h, s.ctypes.data, B, out["osdw"].ctypes.data, out["osd0"].ctypes.data,
                out["bp"].ctypes.data, out["converged"].ctypes.data, out["iters"].ctypes.data,
                out["llr"].ctypes.data if want_llr else None, out["first_nonfinite_iter"].ctypes.data,
                out["final_has_inf"].ctypes.data, out["final_has_nan"].ctypes.data)
        elif DEFAULT_THREADS > 1 and B >= 4 * DEFAULT_THREADS:
            # contiguous slices, one handle and one thread each (ctypes releases the GIL for the call); per shot the
            # computation is the same function on the same inputs, so the result does not depend on the split
            from concurrent.futures import ThreadPoolExecutor

            k = int(DEFAULT_THREADS)
            hs = self._handles(k)
            cuts = [B * j // k for j in range(k + 1)]

            def run(j):
                lo, hi = cuts[j], cuts[j + 1]
                return _load().oracle_decode_batch(
                    hs[j], s[lo:hi].ctypes.data, hi - lo, out["osdw"][lo:hi].ctypes.data, out["osd0"][lo:hi].ctypes.data,
                    out["bp"][lo:hi].ctypes.data, out["converged"][lo:hi].ctypes.data, out["iters"][lo:hi].ctypes.data,
                    out["llr"][lo:hi].ctypes.data if want_llr else None)

            with ThreadPoolExecutor(max_workers=k) as ex:
                rc = max(ex.map(run, range(k)), key=abs)
        else:
            rc = _load().oracle_decode_batch(
                self._h, s.ctypes.data, B, out["osdw"].ctypes.data, out["osd0"].ctypes.data,
                out["bp"].ctypes.data, out["converged"].ctypes.data, out["iters"].ctypes.data,
                out["llr"].ctypes.data if want_llr else None)
        if rc != 0:
            raise RuntimeError(f"oracle_decode_batch failed ({rc})")
        return out

    def decode(self, syndrome):
        r = self.decode_batch(syndrome)
        return {k: (v[0] if v is not None else None) for k, v in r.items()}

    def osd(self, syndrome, llr):
        s = np.ascontiguousarray(np.asarray(syndrome) & 1, dtype=np.uint8)
        l = np.ascontiguousarray(llr, dtype=np.float64)
        osdw = np.zeros(self.n, np.uint8)
        osd0 = np.zeros(self.n, np.uint8)
        order = np.zeros(self.n, np.int32)
        piv = np.zeros(self.n, np.int32)
        rc = _load().oracle_osd(self._h, s.ctypes.data, l.ctypes.data, osdw.ctypes.data, osd0.ctypes.data,
                                order.ctypes.data, piv.ctypes.data)
        if rc != 0:
            raise RuntimeError(f"oracle_osd failed ({rc})")
        return {"osdw": osdw, "osd0": osd0, "order": order, "pivot_flag": piv}
