"""Host-side mirror of the reference's decoder object, on top of the MI355X C-ABI.

The reference does not implement its decoder: it re-exports ``ldpc.bposd_decoder``
(/root/reference/src/bposd/__init__.py:1) and its Monte-Carlo harness instantiates
``ldpc.BpOsdDecoder`` (/root/reference/src/bposd/css_decode_sim.py:6,444-463).  The two
classes below keep those names, ctor kwargs, methods and result attributes
(README.md:178-202; css_decode_sim.py:174-258,294-339) so that user code switches by
changing one import line, while every decode runs in the HIP kernels
behind ``libbposd_mi355x.so``.  There is no CPU path in this module.

Build-native extension beyond the reference surface: :meth:`BpOsdDecoder.decode_batch`
(B syndromes per call -- the unit the GPU path is designed around) and
:meth:`BpOsdDecoder.decode_batch_device` (device pointers, asynchronous).
"""
from __future__ import annotations

import ctypes as C

import sys

import numpy as np
import scipy.sparse as sp

from . import _lib

__all__ = ["BpOsdDecoder", "bposd_decoder"]

_BP_METHODS = {
    "product_sum": 0, "prod_sum": 0, "ps": 0, "0": 0,
    "minimum_sum": 1, "min_sum": 1, "ms": 1, "1": 1,
}
_BP_NAMES = {0: "product_sum", 1: "minimum_sum"}
_OSD_METHODS = {
    "osd_off": 0, "off": 0,
    "osd_0": 1, "osd0": 1, "0": 1, "osd_zero": 1,
    "osd_e": 2, "osde": 2, "e": 2, "exhaustive": 2,
    "osd_cs": 3, "osdcs": 3, "cs": 3, "combination_sweep": 3,
}
_OSD_NAMES = {0: "osd_off", 1: "osd_0", 2: "osd_e", 3: "osd_cs"}


def _parse(table, value, what):
    key = str(value).strip().lower()
    if key not in table:
        raise ValueError(f"{what}='{value}' is invalid. Valid options: {sorted(set(table))}")
    return table[key]


class BpOsdDecoder:
    """MI355X BP+OSD decoder with the call surface of ``ldpc.BpOsdDecoder``.

    Parameters follow the reference call sites (css_decode_sim.py:444-463, README.md:178-187):

    pcm : scipy.sparse matrix or ndarray, shape (m, n)
    error_rate : float, optional -- bit error probability broadcast to all n bits
    error_channel / channel_probs : sequence of n floats, optional -- overrides error_rate
    max_iter : int -- 0 means the block length n
    bp_method : "product_sum" | "minimum_sum" (aliases "ps", "ms", 0, 1)
    ms_scaling_factor : float -- 0 selects the variable factor 1 - 2^-iteration
    osd_method : "osd_0" | "osd_e" | "osd_cs" (aliases "osd0", "exhaustive", "combination_sweep")
    osd_order : int
    device : int -- HIP device ordinal (build-native; default 0)
    ps_clip : float -- build-native, product-sum only: 0 (default) keeps the reference formula, whose check-to-bit
        messages reach +-inf / NaN once tanh rounds to 1; C > 0 clamps them to [-C, C] (DESIGN.md "Product-sum")
    ps_math_form : int -- build-native, product-sum only: evaluation order of the check update.  0 (default) the reference's
        operation order, tanh(b2c / 2) then log((1 + x) / (1 - x)) of the rounded quotient (four divisions per edge; the
        form closest to the platform libm the reference calls); 1 two divisions per edge (1.4 x the throughput)
    """

    def __init__(self, pcm, error_rate=None, error_channel=None, max_iter=0, bp_method="minimum_sum",
                 ms_scaling_factor=1.0, schedule="parallel", omp_thread_count=1, osd_method="osd_0",
                 osd_order=0, input_vector_type="syndrome", channel_probs=None, device=0,
                 sort_tie_policy=0, weight_fn=0, ps_clip=0.0, osd_e_bit_order=0, ps_math_form=0, **kwargs):
        if kwargs:
            raise TypeError(f"unexpected keyword arguments: {sorted(kwargs)}")
        sched = str(schedule).lower()
        if sched not in ("parallel", "serial"):
            raise ValueError(f"schedule='{schedule}' is invalid. Valid options: ['parallel', 'serial']")
        self._schedule = sched
        ivt = {"-1": "auto", "0": "syndrome", "1": "received_vector"}.get(str(input_vector_type).lower(), str(input_vector_type).lower())
        if ivt not in ("syndrome", "received_vector", "auto"):
            raise ValueError(f"input_vector_type='{input_vector_type}' is invalid. Valid options: ['syndrome', 'received_vector', 'auto']")
        self._input_vector_type = ivt
        self.omp_thread_count = int(omp_thread_count)  # accepted for API compatibility; the batch is the unit of parallelism here

        if sp.issparse(pcm):
            h = sp.csr_matrix(pcm)
        else:
            arr = np.asarray(pcm)
            if arr.ndim != 2:
                raise ValueError("The parity check matrix must be a 2-D array or scipy.sparse matrix")
            h = sp.csr_matrix(arr)
        h = h.astype(np.int64)
        h.data %= 2
        h.eliminate_zeros()
        h.sum_duplicates()
        h.sort_indices()
        self.m, self.n = h.shape
        if self.m == 0 or self.n == 0:
            raise ValueError("The parity check matrix is empty")
        self._indptr = np.ascontiguousarray(h.indptr, dtype=np.int32)
        self._indices = np.ascontiguousarray(h.indices, dtype=np.int32)
        self._pcm = h.astype(np.int32)  # host copy: syndromes of received vectors (input_vector_type="received_vector")

        probs = error_channel if error_channel is not None else channel_probs
        if probs is not None and len(probs) == 1 and probs[0] is None:  # channel_probs=[None] (README.md:181)
            probs = None
        if probs is not None:
            probs = np.ascontiguousarray(probs, dtype=np.float64)
            if probs.shape != (self.n,):
                raise ValueError(f"The error channel vector must have length {self.n}, not {probs.shape}")
        else:
            if error_rate is None:
                raise ValueError("Please specify the error channel: either `error_rate` (float) or "
                                 "`channel_probs` / `error_channel` (list of floats of length n)")
            probs = np.full(self.n, float(error_rate), dtype=np.float64)
        if np.any(~((probs >= 0) & (probs <= 1))):
            raise ValueError("channel probabilities must lie in [0, 1]")
        self._probs = probs

        self._bp_method = bp_method if isinstance(bp_method, (int, np.integer)) and bp_method in (0, 1) \
            else _parse(_BP_METHODS, bp_method, "bp_method")
        self._osd_method = _parse(_OSD_METHODS, osd_method, "osd_method")
        max_iter = int(max_iter)
        osd_order = int(osd_order)
        if osd_order < 0:  # legacy "-1 = default"
            osd_order = 0
        if max_iter < 0:
            raise ValueError("max_iter must be >= 0")
        self._ms = float(ms_scaling_factor)
        self._osd_order = osd_order

        lib = _lib.load()
        self._lib = lib
        cfg = _lib.BposdConfig()
        cfg.device = int(device)
        self.device = int(device)
        cfg.bp_method = int(self._bp_method)
        cfg.ms_scaling_factor = self._ms
        cfg.max_iter = max_iter
        cfg.osd_method = int(self._osd_method)
        cfg.osd_order = osd_order
        cfg.sort_tie_policy = int(sort_tie_policy)
        cfg.osd_e_bit_order = int(osd_e_bit_order)
        cfg.weight_fn = int(weight_fn)
        cfg.schedule = 1 if sched == "serial" else 0
        if int(ps_math_form) not in (0, 1):
            raise ValueError("ps_math_form must be 0 (the reference's operation order) or 1 (two divisions per edge)")
        cfg.ps_math_form = int(ps_math_form)
        self.ps_math_form = int(ps_math_form)
        cfg.ps_clip = float(ps_clip)
        if not (cfg.ps_clip >= 0.0 and np.isfinite(cfg.ps_clip)):
            raise ValueError("ps_clip must be 0 (no clipping) or a finite positive bound")
        self._h = C.c_void_p()
        rc = lib.bposd_create(C.byref(cfg), self._indptr.ctypes.data, self._indices.ctypes.data,
                              self.m, self.n, self._probs.ctypes.data, C.byref(self._h))
        if rc != 0:
            self._h = None
            _lib.check(lib, None, rc)
        rank, ncand, mi, nnz = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        lib.bposd_info(self._h, C.byref(rank), C.byref(ncand), C.byref(mi), C.byref(nnz))
        self.rank, self.num_candidates, self._max_iter, self.nnz = rank.value, ncand.value, mi.value, nnz.value

        # one-syndrome decode(): buffers + pointers made on first use, raw results of the last call, converted attributes
        self._one = None
        self._single = None
        self._single_cache = {}
        self._llr_probs = None
        self._timing_override = None
        # batch results of the last decode_batch call
        self.batch_converge = None
        self.batch_osdw = None
        self.batch_iter = None
        self.batch_osd0 = None
        self.batch_bp = None
        self.batch_llr = None

    # ------------------------------------------------------------------ lifetime
    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            try:
                self._lib.bposd_destroy(h)
            except Exception:
                pass
            self._h = None

    # ------------------------------------------------------------------ decode
    @staticmethod
    def pack_rows(rows):
        """uint8 0/1 rows [B, k] -> uint64 [B, ceil(k/64)]: bit (i & 63) of word (i >> 6) = entry i (the C-ABI's packed form)."""
        a = np.ascontiguousarray(rows, dtype=np.uint8)
        by = np.packbits(a, axis=1, bitorder="little")
        w = (a.shape[1] + 63) // 64
        out = np.zeros((a.shape[0], 8 * w), np.uint8)
        out[:, :by.shape[1]] = by
        return out.view("<u8")

    @staticmethod
    def unpack_rows(words, k):
        """The inverse of :meth:`pack_rows`: uint64 [B, ceil(k/64)] -> uint8 [B, k]."""
        w = np.ascontiguousarray(words, dtype="<u8")
        return np.unpackbits(w.view(np.uint8), axis=1, bitorder="little")[:, :k]

    def _decode_batch_packed(self, syndromes, want_osd0, want_bp):
        """``decode_batch(..., packed=True)``: bit-packed rows across the host-device boundary (``bposd_decode_batch_packed``)."""
        s = np.asarray(syndromes)
        wm, wn = (self.m + 63) // 64, (self.n + 63) // 64
        if s.ndim == 2 and s.dtype == np.uint64 and s.shape[1] == wm:
            sw = np.ascontiguousarray(s)
        elif s.ndim == 2 and s.shape[1] == self.m:
            sw = self.pack_rows((s.astype(np.int64) & 1).astype(np.uint8) if s.dtype != np.uint8 else (s & 1))
        else:
            raise ValueError(f"packed decode takes syndromes of shape (B, {self.m}) or uint64 words of shape (B, {wm}). Not {s.shape}.")
        B = sw.shape[0]
        self.batch_osd0 = self.batch_bp = self.batch_llr = self.batch_osdw = None
        osdw = np.empty((B, wn), np.uint64)
        osd0 = np.empty((B, wn), np.uint64) if want_osd0 else None
        bp = np.empty((B, wn), np.uint64) if want_bp else None
        conv = np.empty(B, np.uint8)
        iters = np.empty(B, np.int32)
        self.decode_batch_packed_into(sw, osdw, osd0, bp, conv, iters)
        self.batch_converge = conv.astype(bool)
        self.batch_iter = iters
        self.batch_osd0, self.batch_bp, self.batch_osdw = osd0, bp, osdw
        return osdw

    def decode_batch_packed_into(self, syndrome_words, osdw_words, osd0_words=None, bp_words=None, converged=None, iters=None, wait=True):
        """Host-pointer decode of bit-packed rows into caller-owned C-contiguous arrays: ``uint64 [B, ceil(m/64)]`` in,
        ``uint64 [B, ceil(n/64)]`` out (page-locked arrays from :meth:`pinned_empty` make the copies asynchronous).
        ``wait=False``: the call is only enqueued (``bposd_decode_batch_packed_async``) and the lane to
        :meth:`synchronize` on is returned (None for an empty batch); the arrays must stay untouched until then (the decoder
        keeps them referenced).  One handle serves one caller thread at a time: the per-call mode flags live on
        the handle."""
        self._timing_override = None
        wm, wn = (self.m + 63) // 64, (self.n + 63) // 64
        s = syndrome_words
        if s.dtype != np.uint64 or s.ndim != 2 or s.shape[1] != wm or not s.flags.c_contiguous:
            raise ValueError(f"syndrome_words must be a C-contiguous uint64 array of shape (B, {wm})")
        B = s.shape[0]
        for name, a, dt, shp in (("osdw_words", osdw_words, np.uint64, (B, wn)), ("osd0_words", osd0_words, np.uint64, (B, wn)),
                                 ("bp_words", bp_words, np.uint64, (B, wn)), ("converged", converged, np.uint8, (B,)),
                                 ("iters", iters, np.int32, (B,))):
            if a is None:
                if name == "osdw_words":
                    raise ValueError("osdw_words is required")
                continue
            if a.dtype != dt or a.shape != shp or not a.flags.c_contiguous:
                raise ValueError(f"{name} must be a C-contiguous {np.dtype(dt).name} array of shape {shp}")
        ptr = lambda a: a.ctypes.data if a is not None else None
        fn = self._lib.bposd_decode_batch_packed if wait else self._lib.bposd_decode_batch_packed_async
        if B == 0:
            return osdw_words if wait else None  # nothing was enqueued: no lane to wait for
        rc = fn(self._h, s.ctypes.data, B, ptr(osdw_words), ptr(osd0_words), ptr(bp_words), ptr(converged), ptr(iters))
        _lib.check(self._lib, self._h, rc)
        if wait:
            return osdw_words
        lane = self.last_lane
        self._hold(lane, s, osdw_words, osd0_words, bp_words, converged, iters)
        return lane

    def decode_batch(self, syndromes, want_osd0=True, want_bp=True, want_llr=False, prior_select=None,
                     alt_channel_probs=None, packed=False):
        """Decode B syndromes (array [B, m], any integer dtype).  Returns the OSD-W (or BP, when BP
        converged) corrections as uint8 [B, n]; per-row ``batch_converge``, ``batch_iter`` and, if
        requested, ``batch_osd0`` / ``batch_bp`` / ``batch_llr`` are left on the object.

        ``packed=True``: rows cross the host-device boundary bit-packed (one eighth of the bytes).  The syndromes may be
        given as uint64 words [B, ceil(m/64)] (:meth:`pack_rows`); the result and ``batch_osd0`` / ``batch_bp`` are
        uint64 [B, ceil(n/64)] (:meth:`unpack_rows`).  Not combinable with ``want_llr`` / ``prior_select``.

        ``prior_select`` (uint8 [B, n]) with ``alt_channel_probs`` (n floats) gives every shot its own
        two-valued channel: bit i of shot b uses ``alt_channel_probs[i]`` where ``prior_select[b, i]`` is
        set and the decoder's ``channel_probs[i]`` elsewhere -- the batched form of the per-shot
        ``update_channel_probs`` of css_decode_sim.py:207-248."""
        self._timing_override = None
        if packed:
            if want_llr or prior_select is not None:
                raise ValueError("packed=True offers the integer outputs only (no want_llr, no prior_select)")
            return self._decode_batch_packed(syndromes, want_osd0, want_bp)
        s = np.asarray(syndromes)
        received = None
        if s.ndim == 2 and self._is_received(s.shape[1]):
            # received-vector input (ldpc's input_vector_type="received_vector"): decode the syndromes H r and hand back
            # r + correction; the batch_* arrays and the attributes keep the error estimates
            if s.shape[1] != self.n:
                raise ValueError(f"The received vectors must have shape (B, {self.n}). Not {s.shape}.")
            received = s if s.dtype == np.uint8 else None
            if received is None or (received > 1).any():
                received = np.empty(s.shape, np.uint8)
                for lo in range(0, len(s), 8192):  # (row chunks: no full-size int64 temporaries)
                    received[lo:lo + 8192] = (s[lo:lo + 8192].astype(np.int64) & 1).astype(np.uint8)
            received = np.ascontiguousarray(received)
            syn = np.empty((len(s), self.m), np.uint8)
            for lo in range(0, len(s), 8192):  # syndromes H r in row chunks of 8192: a sparse product per chunk
                syn[lo:lo + 8192] = (np.asarray(self._pcm @ received[lo:lo + 8192].T.astype(np.int32)) & 1).T
            s = syn
        if s.ndim != 2 or s.shape[1] != self.m:
            raise ValueError(f"The syndromes must have shape (B, {self.m}). Not {s.shape}.")
        # uint8 input goes to the device as it is (the kernels look at bit 0 only); other dtypes are reduced mod 2
        s8 = np.ascontiguousarray(s.astype(np.int64) & 1, dtype=np.uint8) if s.dtype != np.uint8 \
            else np.ascontiguousarray(s)
        B = s8.shape[0]
        # drop last call's arrays first: their buffers can then be reused (unless the caller kept them)
        self.batch_osd0 = self.batch_bp = self.batch_llr = self.batch_osdw = None
        osdw = self._out_array((B, self.n), np.uint8)
        osd0 = self._out_array((B, self.n), np.uint8) if want_osd0 else None
        bp = self._out_array((B, self.n), np.uint8) if want_bp else None
        conv = np.empty(B, np.uint8)
        iters = np.empty(B, np.int32)
        llr = self._out_array((B, self.n), np.float64) if want_llr else None
        ptr = lambda a: a.ctypes.data if a is not None else None
        if prior_select is not None:
            sel = np.ascontiguousarray(np.asarray(prior_select) != 0, dtype=np.uint8)
            if sel.shape != (B, self.n):
                raise ValueError(f"prior_select must have shape ({B}, {self.n}), not {sel.shape}")
            if alt_channel_probs is None:
                raise ValueError("alt_channel_probs is required with prior_select")
            alt = np.ascontiguousarray(alt_channel_probs, dtype=np.float64)
            if alt.shape != (self.n,):
                raise ValueError(f"alt_channel_probs must have length {self.n}")
            if B == 0:
                rc = 0
            else:
                rc = self._lib.bposd_decode_batch_select(self._h, s8.ctypes.data, B, sel.ctypes.data, alt.ctypes.data,
                                                         ptr(osdw), ptr(osd0), ptr(bp), ptr(conv), ptr(iters), ptr(llr))
        else:
            rc = self._lib.bposd_decode_batch(self._h, s8.ctypes.data, B, ptr(osdw), ptr(osd0), ptr(bp),
                                              ptr(conv), ptr(iters), ptr(llr))
        _lib.check(self._lib, self._h, rc)
        self.batch_converge = conv.astype(bool)
        self.batch_iter = iters
        self.batch_osd0, self.batch_bp, self.batch_llr = osd0, bp, llr
        self.batch_osdw = osdw
        return osdw if received is None else (osdw ^ received)

    def _is_received(self, length):
        """Does an input of this length hold received vectors (n bits) rather than syndromes (m bits)?"""
        if self._input_vector_type == "received_vector":
            return True
        return self._input_vector_type == "auto" and length == self.n and self.n != self.m

    # ------------------------------------------------------------------ recycled output buffers
    _POOL_MIN_BYTES = 1 << 20   # below this a fresh numpy array costs nothing
    _POOL_MAX_BYTES = 4 << 30   # host memory kept for reuse per decoder

    def _out_array(self, shape, dtype):
        """Output array for decode_batch.  A fresh 250 MB numpy array costs ~25 ms of page faults while the
        device-to-host copy lands in it (measured, tools/pin_probe.py), so large result buffers are recycled:
        a buffer returns to this decoder's pool as soon as no array refers to it any more (reference count),
        i.e. results the caller keeps are never overwritten.  Page-locked buffers were measured too and gain
        nothing over warm pageable memory on this platform."""
        count = int(np.prod(shape))
        nbytes = count * np.dtype(dtype).itemsize
        if nbytes < self._POOL_MIN_BYTES:
            return np.empty(shape, dtype)
        pool = self.__dict__.setdefault("_out_pool", [])
        best = -1
        for k in range(len(pool)):
            # references to a free buffer: the pool list and getrefcount's argument
            if pool[k].nbytes >= nbytes and sys.getrefcount(pool[k]) == 2:
                if best < 0 or pool[k].nbytes < pool[best].nbytes:
                    best = k
        if best < 0:
            pool.append(np.empty(nbytes, np.uint8))
            best = len(pool) - 1
            k = 0
            while sum(b.nbytes for b in pool) > self._POOL_MAX_BYTES and k < len(pool) - 1:
                if sys.getrefcount(pool[k]) == 2:
                    del pool[k]
                    best -= 1
                else:
                    k += 1
        return pool[best][:nbytes].view(dtype).reshape(shape)

    def decode(self, syndrome):
        """Decode one syndrome; returns the correction with the syndrome's dtype
        (README.md:197; css_decode_sim.py:174-202).  Result attributes are updated.

        The one-syndrome call has its own thin path: page-sized buffers and their pointers are made once per decoder, the
        library runs the call without copy commands or events, and the result attributes are converted when read."""
        self._timing_override = None
        s = np.asarray(syndrome)
        if s.ndim == 1 and self._is_received(len(s)):
            if len(s) != self.n:
                raise ValueError(f"The received vector must have length {self.n}. Not {len(s)}.")
            out = self.decode_batch(s[None, :], want_osd0=True, want_bp=True, want_llr=True)
            self._single = (self.batch_osdw, self.batch_osd0, self.batch_bp, self.batch_converge, self.batch_iter, self.batch_llr)
            self._single_cache = {}
            dtype = s.dtype if np.issubdtype(s.dtype, np.number) else int
            return out[0].astype(dtype)
        if s.ndim != 1 or len(s) != self.m:
            raise ValueError(f"The syndrome must have length {self.m}. Not {len(s) if s.ndim else 0}.")
        one = self._one
        if one is None:
            arrs = (np.empty((1, self.m), np.uint8), np.empty((1, self.n), np.uint8), np.empty((1, self.n), np.uint8),
                    np.empty((1, self.n), np.uint8), np.empty(1, np.uint8), np.empty(1, np.int32),
                    np.empty((1, self.n), np.float64))
            one = self._one = (arrs, tuple(C.c_void_p(a.ctypes.data) for a in arrs))
        (s8, osdw, osd0, bp, conv, iters, llr), ptrs = one
        # uint8 input goes to the device as it is (the kernels look at bit 0 only); other dtypes are reduced mod 2
        if s.dtype == np.uint8:
            s8[0] = s
        else:
            s8[0] = s.astype(np.int64) & 1
        # (no LLR pointer: with it the BP kernel stores every bit's posterior in every iteration, +30 % on this call;
        # ``log_prob_ratios`` repeats the -- deterministic -- decode with the pointer when it is read)
        rc = self._lib.bposd_decode_batch(self._h, ptrs[0], 1, ptrs[1], ptrs[2], ptrs[3], ptrs[4], ptrs[5], None)
        if rc:
            _lib.check(self._lib, self._h, rc)
        self._single = (osdw, osd0, bp, conv, iters, None)
        self._single_cache = {}
        self._llr_probs = None  # the channel this call was decoded with, kept only if update_channel_probs replaces it
        dtype = s.dtype if np.issubdtype(s.dtype, np.number) else int
        return osdw[0].astype(dtype)

    def _attr(self, which):
        """Result attribute of the last ``decode()``: converted from the call's raw buffers on first access (the buffers
        are reused by the next ``decode()``, the converted arrays are the caller's)."""
        c = self._single_cache
        if which not in c:
            src = self._single
            if src is None:
                c[which] = {0: np.zeros(self.n, dtype=int), 1: np.zeros(self.n, dtype=int), 2: np.zeros(self.n, dtype=int),
                            3: False, 4: 0, 5: np.zeros(self.n, dtype=np.float64)}[which]
            elif which <= 2:
                c[which] = src[which][0].astype(int)
            elif which == 3:
                c[which] = bool(src[3][0])
            elif which == 4:
                c[which] = int(src[4][0])
            elif src[5] is not None:
                c[which] = src[5][0].copy()
            else:
                c[which] = self._recompute_llr()
        return c[which]

    def _recompute_llr(self):
        """Posterior LLRs of the last ``decode()``: the call is repeated with the LLR output switched on (same syndrome,
        same channel -- also when update_channel_probs has been called since -- hence the same bits)."""
        (s8, _, _, _, _, _, llr), ptrs = self._one
        later = None
        # a property read must not disturb what last_timing() reports about the decode: keep its record
        try:
            self._timing_override = self.last_timing()
        except (ValueError, RuntimeError):
            self._timing_override = None
        if self._llr_probs is not None:
            later = self._probs
            _lib.check(self._lib, self._h, self._lib.bposd_update_channel_probs(self._h, self._llr_probs.ctypes.data))
        try:
            # BP only (the LLRs do not depend on the OSD stage -- on a large code that is a ~100 ms elimination)
            rc = self._lib.bposd_posterior_llr(self._h, ptrs[0], 1, ptrs[6], None, None, None)
            _lib.check(self._lib, self._h, rc)
        finally:
            if later is not None:
                try:
                    _lib.check(self._lib, self._h, self._lib.bposd_update_channel_probs(self._h, later.ctypes.data))
                except Exception:
                    self._probs = self._llr_probs  # the handle still holds the decode's channel: say so
                    raise
        return llr[0].copy()

    def decode_batch_device(self, d_syndromes, B, d_osdw, d_osd0=None, d_bp=None, d_converged=None,
                            d_iters=None, d_llr=None, d_prior_select=None, alt_channel_probs=None):
        """Asynchronous decode on device pointers (ints, e.g. ``tensor.data_ptr()``) that live on this
        decoder's device; call :meth:`synchronize` before reading the outputs.  ``d_prior_select`` (device uint8
        [B, n]) with ``alt_channel_probs`` (host array of n floats) is the per-shot two-valued channel of
        :meth:`decode_batch`."""
        self._timing_override = None
        if d_prior_select is not None:
            if alt_channel_probs is None:
                raise ValueError("alt_channel_probs is required with d_prior_select")
            alt = np.ascontiguousarray(alt_channel_probs, dtype=np.float64)
            if alt.shape != (self.n,):
                raise ValueError(f"alt_channel_probs must have length {self.n}")
            rc = self._lib.bposd_decode_batch_select_device(self._h, d_syndromes, int(B), d_prior_select, alt.ctypes.data,
                                                            d_osdw, d_osd0, d_bp, d_converged, d_iters, d_llr)
        else:
            rc = self._lib.bposd_decode_batch_device(self._h, d_syndromes, int(B), d_osdw, d_osd0, d_bp,
                                                     d_converged, d_iters, d_llr)
        _lib.check(self._lib, self._h, rc)

    def decode_batch_device_packed(self, d_syndrome_words, B, d_osdw_words, d_osd0_words=None, d_bp_words=None, d_converged=None, d_iters=None):
        """Asynchronous decode of ``B`` bit-packed syndromes (device uint64 [B, ceil(m/64)]) into bit-packed rows (device uint64
        [B, ceil(n/64)]) -- ``bposd_decode_batch_device_packed``; raises ValueError where the kernels take byte rows only (then
        :meth:`decode_batch_device` + :meth:`pack_rows_device`)."""
        self._timing_override = None
        rc = self._lib.bposd_decode_batch_device_packed(self._h, d_syndrome_words, int(B), d_osdw_words, d_osd0_words, d_bp_words,
                                                        d_converged, d_iters)
        _lib.check(self._lib, self._h, rc)

    def pack_rows_device(self, d_bytes, B, n, d_words, lane=None):
        """Bit-pack B device rows of n 0/1 bytes into ceil(n/64) uint64 words each (asynchronous).  Queued on the lane of
        the most recent device-pointer decode -- call it right after the decode whose rows it packs -- or on ``lane``
        (``last_lane`` read right after that decode) when other decode calls have been enqueued in between."""
        if lane is None:
            rc = self._lib.bposd_pack_rows_device(self._h, d_bytes, int(B), int(n), d_words)
        else:
            rc = self._lib.bposd_pack_rows_device_lane(self._h, int(lane), d_bytes, int(B), int(n), d_words)
        _lib.check(self._lib, self._h, rc)

    def synchronize(self, lane=None):
        """Wait for everything queued on this decoder, or (``lane``) for the calls queued on one lane only."""
        if lane is None:
            _lib.check(self._lib, self._h, self._lib.bposd_synchronize(self._h))
            self._inflight = {}
        else:
            _lib.check(self._lib, self._h, self._lib.bposd_synchronize_lane(self._h, int(lane)))
            getattr(self, "_inflight", {}).pop(int(lane), None)

    def _hold(self, lane, *arrays):
        """An asynchronous host call hands raw pointers to enqueued copies: the arrays (page-locked ones are freed by a
        finalizer when dropped) stay referenced here until the lane has been synchronised."""
        if not hasattr(self, "_inflight"):
            self._inflight = {}
        self._inflight.setdefault(int(lane), []).extend(a for a in arrays if a is not None)

    @property
    def num_lanes(self):
        """Streams the handle alternates between: consecutive ``decode_batch_device`` calls overlap on the device."""
        return int(self._lib.bposd_num_lanes(self._h))

    @property
    def last_lane(self):
        """Lane the last ``decode_batch_device`` call was queued on."""
        return int(self._lib.bposd_last_lane(self._h))

    def lane_timing(self, lane):
        """:meth:`last_timing` for the last call queued on ``lane`` (waits for that lane only)."""
        a, b = C.c_double(), C.c_double()
        it, no = C.c_int64(), C.c_int64()
        rc = self._lib.bposd_lane_timing(self._h, int(lane), C.byref(a), C.byref(b), C.byref(it), C.byref(no))
        _lib.check(self._lib, self._h, rc)
        return {"bp_ms": a.value, "osd_ms": b.value, "bp_iterations": it.value, "osd_invocations": no.value}

    def pinned_empty(self, shape, dtype=np.uint8):
        """numpy array in page-locked host memory of the library's HIP runtime (``bposd_host_alloc``): passing such
        arrays to :meth:`decode_batch_into` makes its chunked uploads / downloads asynchronous."""
        import weakref

        count = int(np.prod(shape))
        nbytes = max(1, count * np.dtype(dtype).itemsize)
        ptr = self._lib.bposd_host_alloc(nbytes)
        if not ptr:
            raise MemoryError(f"bposd_host_alloc({nbytes}) failed")
        buf = (C.c_uint8 * nbytes).from_address(ptr)
        arr = np.frombuffer(buf, dtype=np.uint8, count=nbytes)[:count * np.dtype(dtype).itemsize].view(dtype).reshape(shape)
        weakref.finalize(buf, self._lib.bposd_host_free, ptr)  # freed when the last view of the buffer is gone
        return arr

    def decode_batch_into(self, syndromes, osdw, osd0=None, bp=None, converged=None, iters=None, llr=None, wait=True):
        """Host-pointer decode into caller-owned C-contiguous arrays (``uint8 [B, m]`` in; ``uint8 [B, n]``, ``uint8 [B]``,
        ``int32 [B]``, ``float64 [B, n]`` out; any output but ``osdw`` may be None) -- ``bposd_decode_batch`` with no
        allocation or conversion on the way.  ``wait=False``: ``bposd_decode_batch_async`` -- the call is only enqueued and
        the lane to :meth:`synchronize` on is returned; the arrays must stay untouched until then."""
        self._timing_override = None
        s = syndromes
        if s.dtype != np.uint8 or s.ndim != 2 or s.shape[1] != self.m or not s.flags.c_contiguous:
            raise ValueError(f"syndromes must be a C-contiguous uint8 array of shape (B, {self.m})")
        B = s.shape[0]
        for name, a, dt, shp in (("osdw", osdw, np.uint8, (B, self.n)), ("osd0", osd0, np.uint8, (B, self.n)),
                                 ("bp", bp, np.uint8, (B, self.n)), ("converged", converged, np.uint8, (B,)),
                                 ("iters", iters, np.int32, (B,)), ("llr", llr, np.float64, (B, self.n))):
            if a is None:
                if name == "osdw":
                    raise ValueError("osdw is required")
                continue
            if a.dtype != dt or a.shape != shp or not a.flags.c_contiguous:
                raise ValueError(f"{name} must be a C-contiguous {np.dtype(dt).name} array of shape {shp}")
        ptr = lambda a: a.ctypes.data if a is not None else None
        fn = self._lib.bposd_decode_batch if wait else self._lib.bposd_decode_batch_async
        if B == 0:
            return osdw if wait else None  # nothing was enqueued: no lane to wait for
        rc = fn(self._h, s.ctypes.data, B, ptr(osdw), ptr(osd0), ptr(bp), ptr(converged), ptr(iters), ptr(llr))
        _lib.check(self._lib, self._h, rc)
        if wait:
            return osdw
        lane = self.last_lane
        self._hold(lane, s, osdw, osd0, bp, converged, iters, llr)
        return lane

    def last_timing(self):
        """dict(bp_ms, osd_ms, bp_iterations, osd_invocations) of the last decode call (HIP events).  Small host-pointer
        calls (up to 1 MB of staging, e.g. one ``decode()``) run without events: their two times read 0.0, the counters
        are exact."""
        if getattr(self, "_timing_override", None) is not None:
            return dict(self._timing_override)
        a, b = C.c_double(), C.c_double()
        it, no = C.c_int64(), C.c_int64()
        rc = self._lib.bposd_last_timing(self._h, C.byref(a), C.byref(b), C.byref(it), C.byref(no))
        _lib.check(self._lib, self._h, rc)
        return {"bp_ms": a.value, "osd_ms": b.value, "bp_iterations": it.value, "osd_invocations": no.value}

    def layout_info(self):
        """dict(natural, chosen, ideal): simulated LDS cycles of one bit pass for the bit orders considered."""
        a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
        _lib.check(self._lib, self._h, self._lib.bposd_layout_info(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return {"natural": a.value, "chosen": b.value, "ideal": c.value}

    BP_KERNEL_NAMES = {0: "bp_kernel", 1: "bp_local_kernel", 2: "bp_class_kernel", 3: "bp_large_kernel", 4: "bp_serial_kernel", 5: "bp_anydeg_kernel"}

    def bp_kernel_info(self):
        """Which BP kernel the last decode call launched, with the bank-conflict model of its bit pass (modelled LDS
        cycles per workgroup and their conflict-free floors; zeros where the kernel has no such model)."""
        k = C.c_int32()
        mdl = np.zeros(4, dtype=np.int64)
        _lib.check(self._lib, self._h, self._lib.bposd_bp_kernel_info(self._h, C.byref(k), mdl.ctypes.data))
        return {"kernel": self.BP_KERNEL_NAMES.get(k.value, "none"), "read_cycles": int(mdl[0]), "read_floor": int(mdl[1]),
                "write_cycles": int(mdl[2]), "write_floor": int(mdl[3])}

    OSD_KERNEL_NAMES = {1: "osd_kernel", 2: "osd_wave_kernel", 3: "osd_large_kernel", 4: "osd_mw_kernel"}

    def set_osd_variant(self, variant: int):
        """Tuning / test knob: 0 auto; 1 one workgroup per elimination; 2 one wave per elimination where it applies."""
        _lib.check(self._lib, self._h, self._lib.bposd_set_osd_variant(self._h, int(variant)))

    def last_osd_kernel(self):
        """Name of the OSD kernel the last decode call launched ("none" before the first one)."""
        return self.OSD_KERNEL_NAMES.get(self._lib.bposd_last_osd_kernel(self._h), "none")

    def set_bp_variant(self, variant: int):
        """Tuning / test knob: 0 auto; 1, 2, 4 LDS kernel shapes; 16 .. 26 local-edge kernel; 32 class kernel; 63 HBM-resident min-sum with whole check records in the workspace; 64 any-degree kernel (slow; cross-checks) -- see the C header."""
        _lib.check(self._lib, self._h, self._lib.bposd_set_bp_variant(self._h, int(variant)))

    # ------------------------------------------------------------------ mutators / attributes
    def update_channel_probs(self, channel_probs):
        """css_decode_sim.py:229,248."""
        p = np.ascontiguousarray(channel_probs, dtype=np.float64)
        if p.shape != (self.n,):
            raise ValueError(f"The error channel vector must have length {self.n}, not {p.shape}")
        rc = self._lib.bposd_update_channel_probs(self._h, p.ctypes.data)
        _lib.check(self._lib, self._h, rc)
        if self._single is not None and self._single[5] is None and 5 not in self._single_cache and self._llr_probs is None:
            self._llr_probs = self._probs  # the last decode()'s LLRs have not been read yet: they belong to this channel
        self._probs = p.copy()

    @property
    def osdw_decoding(self):
        return self._attr(0)

    @property
    def osd0_decoding(self):
        return self._attr(1)

    @property
    def bp_decoding(self):
        return self._attr(2)

    @property
    def decoding(self):
        return self._attr(0)

    @property
    def converge(self):
        return self._attr(3)

    @property
    def iter(self):
        return self._attr(4)

    @property
    def log_prob_ratios(self):
        return self._attr(5)

    @property
    def channel_probs(self):
        return self._probs.copy()

    error_channel = channel_probs

    @property
    def max_iter(self):
        return self._max_iter

    @property
    def schedule(self):
        return self._schedule

    @property
    def input_vector_type(self):
        return self._input_vector_type

    @property
    def bp_method(self):
        return _BP_NAMES[self._bp_method]

    @property
    def osd_method(self):
        return _OSD_NAMES[self._osd_method]

    @property
    def osd_order(self):
        return self._osd_order

    @property
    def ms_scaling_factor(self):
        return self._ms

    @property
    def check_count(self):
        return self.m

    @property
    def bit_count(self):
        return self.n


class bposd_decoder(BpOsdDecoder):
    """Legacy-name constructor, kwargs as at /root/reference/README.md:178-187
    (``from ldpc import bposd_decoder``, re-exported by /root/reference/src/bposd/__init__.py:1)."""

    def __init__(self, parity_check_matrix, error_rate=None, max_iter=0, bp_method=0, ms_scaling_factor=1.0,
                 channel_probs=[None], input_vector_type=-1, osd_order=-1, osd_method=0, **kwargs):
        if isinstance(osd_method, (int, np.integer)):
            osd_method = {0: "osd_0", 1: "osd_e", 2: "osd_cs"}.get(int(osd_method), osd_method)
        if isinstance(input_vector_type, (int, np.integer)):  # ldpc v1: -1 auto (by length), 0 syndrome, 1 received vector
            if int(input_vector_type) not in (-1, 0, 1):
                raise ValueError("input_vector_type must be -1 (auto), 0 (syndrome) or 1 (received vector)")
            input_vector_type = {-1: "auto", 0: "syndrome", 1: "received_vector"}[int(input_vector_type)]
        super().__init__(parity_check_matrix, error_rate=error_rate, max_iter=max_iter, bp_method=bp_method,
                         ms_scaling_factor=ms_scaling_factor, channel_probs=channel_probs,
                         osd_method=osd_method, osd_order=osd_order, input_vector_type=input_vector_type,
                         **kwargs)
