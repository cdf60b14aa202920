"""Host-side tests that need no GPU: the C-ABI library loads and exports every declared symbol,
option parsing mirrors the reference's kwargs, and the product path fails loudly without a device."""
import ctypes
import os
import re

import numpy as np
import pytest

import bp_osd_amd
from bp_osd_amd import _lib, bposd_decoder, BpOsdDecoder
from bp_osd_amd.build import build_library

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    build_library()
    return _lib.load()


def test_library_exports_every_header_symbol(lib):
    hdr = open(os.path.join(ROOT, "include", "bposd_mi355x.h")).read()
    code = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)  # strip comments
    declared = set(re.findall(r"\b(bposd_[a-z_0-9]+)\s*\(", code))
    assert declared == set(_lib.EXPORTED_SYMBOLS), declared ^ set(_lib.EXPORTED_SYMBOLS)
    for sym in declared:
        assert getattr(lib, sym) is not None
    assert b"gfx950" in lib.bposd_version()


def test_config_struct_layout_matches_header():
    # int32, int32, double, int32 x6, double, int32[2]  -> 8-byte aligned doubles at offsets 8 and 40
    assert _lib.BposdConfig.ms_scaling_factor.offset == 8
    assert _lib.BposdConfig.ps_clip.offset == 40
    assert ctypes.sizeof(_lib.BposdConfig) == 56


def _has_gpu(lib):
    return lib.bposd_device_count() > 0


def test_no_cpu_fallback_without_device(lib, surface13):
    if _has_gpu(lib):
        pytest.skip("a GPU is visible here")
    with pytest.raises(RuntimeError, match="no CPU path"):
        bposd_decoder(surface13.hz, error_rate=0.05)


def test_product_path_never_imports_oracle():
    """The shipped package must not reference the oracle (it is test infrastructure)."""
    pkg = os.path.dirname(bp_osd_amd.__file__)
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "bposd_oracle" not in txt, f


@pytest.mark.parametrize("kwargs,exc", [
    (dict(error_rate=0.05, bp_method="banana"), ValueError),
    (dict(error_rate=0.05, osd_method="osd_x"), ValueError),
    (dict(), ValueError),  # neither error_rate nor channel_probs
    (dict(channel_probs=[0.1, 0.2]), ValueError),  # wrong length
    (dict(error_rate=0.05, schedule="layered"), ValueError),
    (dict(error_rate=0.05, input_vector_type="codeword"), ValueError),
    (dict(error_rate=0.05, max_iter=-1), ValueError),
    (dict(error_rate=1.5), ValueError),
    (dict(error_rate=0.05, bogus=1), TypeError),
])
def test_ctor_argument_errors_raise_before_touching_the_device(surface13, kwargs, exc):
    with pytest.raises(exc):
        BpOsdDecoder(surface13.hz, **kwargs)


def test_create_rejects_bad_csr_via_c_abi(lib):
    cfg = _lib.BposdConfig()
    cfg.bp_method = 1
    cfg.ms_scaling_factor = 1.0
    indptr = np.array([0, 2], dtype=np.int32)
    indices = np.array([1, 1], dtype=np.int32)  # not strictly ascending
    probs = np.full(3, 0.1)
    h = ctypes.c_void_p()
    rc = lib.bposd_create(ctypes.byref(cfg), indptr.ctypes.data, indices.ctypes.data, 1, 3, probs.ctypes.data,
                          ctypes.byref(h))
    assert rc == _lib.BPOSD_ERR_INVALID and not h.value
    assert b"ascending" in lib.bposd_last_error(None)
    cfg.bp_method = 7
    indices[:] = [0, 1]
    rc = lib.bposd_create(ctypes.byref(cfg), indptr.ctypes.data, indices.ctypes.data, 1, 3, probs.ctypes.data,
                          ctypes.byref(h))
    assert rc == _lib.BPOSD_ERR_INVALID


def test_output_buffer_pool_never_hands_out_live_memory():
    """decode_batch recycles large result buffers; a buffer is reused only when no array refers to it."""
    import types

    from bp_osd_amd.decoder import BpOsdDecoder

    fake = types.SimpleNamespace(_POOL_MIN_BYTES=1 << 10, _POOL_MAX_BYTES=3 << 12)
    alloc = lambda shape, dt=np.uint8: BpOsdDecoder._out_array(fake, shape, dt)
    small = alloc((4, 8))
    assert "_out_pool" not in fake.__dict__ and small.shape == (4, 8)          # below the threshold: plain array
    a = alloc((64, 64))
    a[:] = 7
    b = alloc((64, 64))
    assert not np.shares_memory(a, b) and len(fake._out_pool) == 2               # a is alive -> new buffer
    addr_a = a.ctypes.data
    view = a[3:5]                                                                # a view keeps the buffer busy
    del a
    c = alloc((64, 64))
    assert c.ctypes.data != addr_a and (view == 7).all() and len(fake._out_pool) == 3
    del view
    e = alloc((64, 64))
    assert e.ctypes.data == addr_a and len(fake._out_pool) == 3                  # the freed buffer is reused (from the pool)
    d = alloc((32, 64), np.float64)                                              # 16 KB > 4 KB buffers: new one
    assert d.dtype == np.float64 and d.shape == (32, 64)
    assert sum(x.nbytes for x in fake._out_pool) <= max(3 << 12, d.nbytes + 2 * 4096) + 4096
