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


@pytest.mark.parametrize("header,symbols", [("bposd_mi355x.h", "EXPORTED_SYMBOLS"), ("bposd_mi355x_debug.h", "DEBUG_SYMBOLS")])
def test_library_exports_every_header_symbol(lib, header, symbols):
    hdr = open(os.path.join(ROOT, "include", header)).read()
    code = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)  # strip comments
    declared = set(re.findall(r"\b(bposd_[a-z_0-9]+)\s*\(", code))
    assert declared == set(getattr(_lib, symbols)), declared ^ set(getattr(_lib, symbols))
    for sym in declared:
        assert getattr(lib, sym) is not None
    assert b"gfx950" in lib.bposd_version()


def test_library_exports_nothing_else(lib):
    """Built with -fvisibility=hidden: the dynamic symbol table holds the two headers' functions (and HIP kernel handles)."""
    import subprocess

    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    funcs = {ln.split()[-1] for ln in out.splitlines() if len(ln.split()) == 3 and ln.split()[1] == "T"}
    assert funcs == set(_lib.EXPORTED_SYMBOLS) | set(_lib.DEBUG_SYMBOLS), funcs ^ (set(_lib.EXPORTED_SYMBOLS) | set(_lib.DEBUG_SYMBOLS))


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


def _csr32(H):
    import scipy.sparse as sp

    H = sp.csr_matrix(H)
    H.sort_indices()
    return np.ascontiguousarray(H.indptr, dtype=np.int32), np.ascontiguousarray(H.indices, dtype=np.int32), H


@pytest.mark.parametrize("seed_file", ["mkmn_16_4_6.txt", "mkmn_20_5_8.txt", "mkmn_24_6_10.txt", "surface:9", "surface:17"])
def test_class_kernel_tables_are_consistent(lib, seed_file):
    """Host-side tables of bp_class_kernel (no GPU needed): every check sits at exactly one position of the check waves,
    every bit in exactly one (thread, slot), every edge has its own LDS slot k * MP + position of its check with k its
    rank inside the check, padding lanes point at their thread's dummy slot, a group's degree is its bits' degree, a
    wave's checks all have the wave's degree (surface codes: degrees 3 and 4), and the bank model of the chosen layout
    is not worse than twice its floor."""
    import os

    from bp_osd_amd.codes import hgp, rep_code

    if seed_file.startswith("surface:"):
        seed = rep_code(int(seed_file.split(":")[1]))
    else:
        seed = np.loadtxt(os.path.join(os.path.dirname(__file__), "golden", seed_file)).astype(np.uint8)
    ip, ix, H = _csr32(hgp(seed, compute_logicals=False).hz)
    m, n = H.shape
    MPmax, VPT, DVHI = 1024, 2, 4
    pos_chk = np.full(MPmax, -7, np.int32)
    pos_bit = np.full(VPT * MPmax, -7, np.int32)
    bit_slot = np.full(DVHI * VPT * MPmax, -7, np.int32)
    grp_deg = np.full(VPT * MPmax // 64, -7, np.int32)
    grp_cdeg = np.full(MPmax // 64, -7, np.int32)
    info = np.zeros(11, np.int64)
    rc = lib.bposd_debug_class_layout(ip.ctypes.data, ix.ctypes.data, m, n, pos_chk.ctypes.data, pos_bit.ctypes.data,
                                      bit_slot.ctypes.data, grp_deg.ctypes.data, grp_cdeg.ctypes.data, info.ctypes.data)
    assert rc == 0
    DC, DVLO, DVHI_, VPT_, MP, NT = (int(x) for x in info[:6])
    DCLO = int(info[10])
    want = (3, 4, 1, 2, 2) if seed_file.startswith("surface:") else (7, 7, 3, 4, 2)
    assert (DCLO, DC, DVLO, DVHI_, VPT_) == want and MP >= m and NT % 64 == 0 and NT <= MP
    pc = pos_chk[:MP]
    assert sorted(pc[pc >= 0].tolist()) == list(range(m))
    cd = grp_cdeg[:MP // 64]
    row_deg = np.diff(H.indptr)
    for w in range(MP // 64):  # a wave's checks all have the wave's degree; waves without checks say 0
        here = pc[w * 64:(w + 1) * 64]
        assert (cd[w] == 0 and (here < 0).all()) or (cd[w] > 0 and (here >= 0).any() and (row_deg[here[here >= 0]] == cd[w]).all())
    assert (cd[NT // 64:] == 0).all()
    pos_of = np.empty(m, int)
    pos_of[pc[pc >= 0]] = np.flatnonzero(pc >= 0)
    pb = pos_bit[:VPT * MP].reshape(VPT, MP)
    assert sorted(pb[pb >= 0].tolist()) == list(range(n)) and (pb[:, NT:] < 0).all()
    bs = bit_slot[:DVHI_ * VPT * MP].reshape(DVHI_, VPT, MP)
    gd = grp_deg[:VPT * MP // 64].reshape(VPT, MP // 64)
    Hc = H.tocsc()
    seen = set()
    for r in range(VPT):
        for t in range(NT):
            i, deg_g = pb[r, t], gd[r, t // 64]
            if i < 0:
                assert (bs[:, r, t] == DC * MP + t).all()  # the thread's dummy slot
                continue
            checks = Hc.indices[Hc.indptr[i]:Hc.indptr[i + 1]]
            assert len(checks) == deg_g
            for d, c in enumerate(sorted(checks)):
                k = int(np.searchsorted(H.indices[H.indptr[c]:H.indptr[c + 1]], i))
                assert bs[d, r, t] == k * MP + pos_of[c]
                seen.add(int(bs[d, r, t]))
    assert len(seen) == H.nnz
    assert info[6] <= 2.1 * info[7] and info[8] <= 1.3 * info[9]


def test_local_edge_layout_model_regression(lib):
    """Host-side layout search of bp_local_kernel on H1922 (no GPU needed): a perfect ownership exists, at most one mixed
    group, and the modelled LDS cycles of the bit pass stay near what the search reached when it was tuned (218 read /
    408 write cycles against floors of 128 / 384)."""
    from bp_osd_amd.codes import h1922

    ip, ix, H = _csr32(h1922(compute_logicals=False).hz)
    out = np.zeros(16, np.int64)
    assert lib.bposd_debug_local_layout(ip.ctypes.data, ix.ctypes.data, H.shape[0], H.shape[1], out.ctypes.data) == 0
    assert out[1] == 128 and out[15] == 384 and out[4] == 1024
    assert out[0] <= 240 and out[14] <= 430 and out[3] <= 2
    assert out[5:14].sum() == H.shape[0]


def test_bench_batches_are_prefix_stable():
    """bench.py reuses the headline's seeded batch for the 65536-syndrome configurations: numpy's generator is consumed row by
    row, so the first B rows of a longer batch of the same seed ARE the batch of B rows (errors and syndromes)."""
    import bench
    from bp_osd_amd.codes import surface13

    H = surface13().hz
    bench._BATCH_CACHE.clear()
    e_small, s_small = bench.make_batch(H, 0.1, 40, seed=5, chunk=16)
    e_big, s_big = bench.make_batch(H, 0.1, 100, seed=5, chunk=16, cache_key="s13")
    assert (e_big[:40] == e_small).all() and (s_big[:40] == s_small).all()
    e_hit, s_hit = bench.make_batch(H, 0.1, 40, seed=5, chunk=16, cache_key="s13")  # served from the cached longer batch
    assert e_hit.base is not None and (e_hit == e_small).all() and (s_hit == s_small).all()
    assert ((H @ e_small.T.astype(int)) % 2 == s_small.T).all()


def test_panel_phase_by_six_column_sub_blocks_model():
    """osd_kernel's panel phase (claims by 6-bit value, the 64 values solved lane by lane, one table entry per row) replayed on
    random panels against column-by-column Gauss-Jordan: same pivot columns, reduced form, combination masks."""
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "panel_subblock_model", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "panel_subblock_model.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    for seed in range(300):
        mod.check(seed)
