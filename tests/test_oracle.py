"""CPU tests of the oracle (no GPU): known-answer, two-restatement agreement, invariants.

The reference holds no test of the decode path ("parity unpinned"); the only
reference-authored datum is the worked example at /root/reference/README.md:178-216.
"""
import itertools

import numpy as np
import pytest

from bp_osd_amd.codes import hamming_code, rep_code, hgp, gf2_rank
from oracle import OracleDecoder
from tests import ref_numpy

BP = {"ms": "ms", "ps": "ps"}


def test_readme_known_answer(surface13):
    """README.md:178-216: error on qubits {5,12} of the [[13,1,3]] surface code -> osdw_decoding = e_8."""
    hz = surface13.hz
    assert hz.shape == (6, 13)
    dec = OracleDecoder(hz, error_rate=0.05, channel_probs=[None], max_iter=13, bp_method="ms",
                        ms_scaling_factor=0, osd_method="osd_cs", osd_order=7)
    err = np.zeros(13, dtype=int)
    err[[5, 12]] = 1
    syn = hz @ err % 2
    assert list(syn) == [0, 0, 0, 0, 0, 1]
    r = dec.decode(syn)
    expect = np.zeros(13, dtype=np.uint8)
    expect[8] = 1
    assert (r["osdw"] == expect).all()
    residual = (r["osdw"] + err) % 2
    assert not (surface13.lz @ residual % 2).any()  # "Logical Error: No"


def test_candidate_counts(surface13):
    """a10: osd_cs order w -> k' + C(w,2); osd_e -> 2^w - 1; k' = n - rank."""
    d = OracleDecoder(surface13.hz, error_rate=0.05, osd_method="osd_cs", osd_order=7)
    assert d.rank == 6 and d.num_candidates == 7 + 21
    d = OracleDecoder(surface13.hz, error_rate=0.05, osd_method="osd_e", osd_order=5)
    assert d.num_candidates == 31
    with pytest.raises(ValueError):
        OracleDecoder(surface13.hz, error_rate=0.05, osd_method="osd_e", osd_order=8)


@pytest.mark.parametrize("bp_method", ["ms", "ps"])
@pytest.mark.parametrize("osd", [("osd0", 0), ("osd_e", 7), ("osd_cs", 7), ("osd_cs", 3)])
@pytest.mark.parametrize("ms", [0.0, 0.625])
def test_two_restatements_agree_s13_exhaustive(surface13, bp_method, osd, ms):
    """All 2^6 syndromes of S13: C oracle == numpy restatement bit for bit, LLR doubles included."""
    if bp_method == "ps" and ms != 0.0:
        pytest.skip("scaling factor unused by product-sum")
    H = surface13.hz.toarray()
    q = 0.05
    # max_iter=3 keeps most syndromes non-converged so that OSD is exercised
    for max_iter in (3, 13):
        dec = OracleDecoder(H, error_rate=q, max_iter=max_iter, bp_method=bp_method, ms_scaling_factor=ms,
                            osd_method=osd[0], osd_order=osd[1])
        syns = np.array(list(itertools.product([0, 1], repeat=6)), dtype=np.uint8)
        r = dec.decode_batch(syns)
        for b, s in enumerate(syns):
            ref = ref_numpy.bposd_decode(H, s, [q] * 13, max_iter, bp_method, ms, osd[0], osd[1])
            assert bool(r["converged"][b]) == ref["converged"], (b, s)
            assert r["iters"][b] == ref["iters"]
            assert (r["bp"][b] == ref["bp"]).all()
            assert (r["llr"][b].view(np.uint64) == ref["llr"].view(np.uint64)).all(), (b, r["llr"][b], ref["llr"])
            assert (r["osd0"][b] == ref["osd0"]).all(), (b, s)
            assert (r["osdw"][b] == ref["osdw"]).all(), (b, s)


def test_two_restatements_agree_nonuniform_probs(surface13):
    """Non-uniform channel (a1, a12) incl. update_channel_probs: weights are sum log(1/p_i)."""
    H = surface13.hz.toarray()
    rng = np.random.default_rng(7)
    probs = rng.uniform(0.01, 0.3, size=13)
    dec = OracleDecoder(H, channel_probs=probs, max_iter=2, bp_method="ms", ms_scaling_factor=0.8,
                        osd_method="osd_cs", osd_order=5)
    syns = np.array(list(itertools.product([0, 1], repeat=6)), dtype=np.uint8)
    for probs_now in (probs, rng.uniform(0.01, 0.3, size=13)):
        dec.update_channel_probs(probs_now)
        r = dec.decode_batch(syns)
        for b, s in enumerate(syns):
            ref = ref_numpy.bposd_decode(H, s, list(probs_now), 2, "ms", 0.8, "osd_cs", 5)
            assert (r["osdw"][b] == ref["osdw"]).all()
            assert (r["osd0"][b] == ref["osd0"]).all()
            assert (r["llr"][b].view(np.uint64) == ref["llr"].view(np.uint64)).all()


def test_two_restatements_agree_hgp400(hgp400):
    """Mid-size code from the reference's own seed matrix, elevated noise to force OSD."""
    H = hgp400.hx.toarray()
    q = 0.09
    rng = np.random.default_rng(3)
    for bp_method, osd in (("ms", ("osd_cs", 6)), ("ps", ("osd_e", 5))):
        dec = OracleDecoder(H, error_rate=q, max_iter=6, bp_method=bp_method, ms_scaling_factor=0,
                            osd_method=osd[0], osd_order=osd[1])
        err = (rng.random((6, 400)) < q).astype(np.uint8)
        syn = (err @ H.T % 2).astype(np.uint8)
        r = dec.decode_batch(syn)
        assert (~r["converged"].astype(bool)).any()
        for b in range(len(syn)):
            ref = ref_numpy.bposd_decode(H, syn[b], [q] * 400, 6, bp_method, 0, osd[0], osd[1])
            assert bool(r["converged"][b]) == ref["converged"]
            assert (r["llr"][b].view(np.uint64) == ref["llr"].view(np.uint64)).all()
            assert (r["osd0"][b] == ref["osd0"]).all()
            assert (r["osdw"][b] == ref["osdw"]).all()


@pytest.mark.parametrize("code", ["s13", "hamming"])
def test_osd_e_full_order_is_minimum_weight(surface13, code):
    """Appendix B item 3: OSD-E with order k' searches the whole coset -> true minimum weight."""
    H = surface13.hz.toarray() if code == "s13" else hamming_code(3)
    m, n = H.shape
    kp = n - gf2_rank(H)
    dec = OracleDecoder(H, error_rate=0.1, max_iter=1, bp_method="ms", ms_scaling_factor=1.0,
                        osd_method="osd_e", osd_order=kp)
    allx = np.array(list(itertools.product([0, 1], repeat=n)), dtype=np.uint8)
    allsyn = allx @ H.T % 2
    for s in itertools.product([0, 1], repeat=m):
        s = np.array(s, dtype=np.uint8)
        if not s.any():
            continue
        r = dec.decode(s)
        coset = allx[(allsyn == s).all(axis=1)]
        if len(coset) == 0:
            continue
        wmin = coset.sum(axis=1).min()
        if not r["converged"]:
            assert (H @ r["osdw"] % 2 == s).all()
            assert r["osdw"].sum() == wmin


def _check_invariants(H, syn, r, rank, order_check=None):
    Hd = H.toarray() if hasattr(H, "toarray") else H
    B = len(syn)
    assert ((r["osd0"] @ Hd.T) % 2 == syn).all()
    assert ((r["osdw"] @ Hd.T) % 2 == syn).all()
    conv = r["converged"].astype(bool)
    assert ((r["bp"][conv] @ Hd.T) % 2 == syn[conv]).all()
    assert (r["osd0"][conv] == r["bp"][conv]).all() and (r["osdw"][conv] == r["bp"][conv]).all()
    assert (r["osdw"].sum(axis=1) <= r["osd0"].sum(axis=1)).all()
    # supp(osd0) lies inside an independent column set: the support columns are independent
    for b in np.nonzero(~conv)[0][:8]:
        supp = np.nonzero(r["osd0"][b])[0]
        assert gf2_rank(Hd[:, supp]) == len(supp) <= rank


def test_invariants_h1922(h1922):
    """Appendix B item 2 on the benchmark code, with max_iter small enough to force OSD."""
    H = h1922.hz
    q = 0.06
    rng = np.random.default_rng(11)
    err = (rng.random((24, 1922)) < q).astype(np.uint8)
    syn = (H @ err.T % 2).T.astype(np.uint8)
    res = {}
    for osd in (("osd0", 0), ("osd_cs", 7), ("osd_cs", 20), ("osd_e", 8)):
        dec = OracleDecoder(H, error_rate=q, max_iter=12, bp_method="ms", ms_scaling_factor=0,
                            osd_method=osd[0], osd_order=osd[1])
        assert dec.rank == 936
        r = dec.decode_batch(syn)
        assert (~r["converged"].astype(bool)).sum() >= 4
        _check_invariants(H, syn, r, dec.rank)
        res[osd] = r
    # osd0 identical whatever the OSD-W method; weights ordered osd0 >= cs7 >= cs20; osd_e(8) <= pairs in first 8
    w = {k: v["osdw"].sum(axis=1) for k, v in res.items()}
    assert (res[("osd0", 0)]["osd0"] == res[("osd_cs", 7)]["osd0"]).all()
    assert (w[("osd_cs", 7)] <= w[("osd0", 0)]).all()
    assert (w[("osd_cs", 20)] <= w[("osd_cs", 7)]).all()


def test_sort_tie_policy_switch(surface13):
    """Appendix A.4: tie order among equal LLRs is a switch; policy 0 = stable ascending index."""
    H = surface13.hz.toarray()
    dec0 = OracleDecoder(H, error_rate=0.05, max_iter=1, osd_method="osd0", sort_tie_policy=0)
    dec1 = OracleDecoder(H, error_rate=0.05, max_iter=1, osd_method="osd0", sort_tie_policy=1)
    llr = np.ones(13)
    s = np.array([1, 0, 0, 0, 0, 0], dtype=np.uint8)
    o0 = dec0.osd(s, llr)
    o1 = dec1.osd(s, llr)
    assert list(o0["order"]) == list(range(13))
    assert list(o1["order"]) == list(range(12, -1, -1))
    assert (H @ o0["osd0"] % 2 == s).all() and (H @ o1["osd0"] % 2 == s).all()


def test_osd_e_bit_order_switch(surface13):
    """Appendix A.4: which T position bit b of OSD-E pattern i stands for is a switch (0 = position b, 1 = w - 1 - b).  It
    only decides ties between equally light patterns (first enumerated wins): both settings agree with the independent
    numpy restatement, the weights never differ, and on the uniform channel some winner does."""
    H = surface13.hz.toarray()
    probs = np.full(13, 0.05)
    kw = dict(error_rate=0.05, max_iter=2, bp_method="ms", ms_scaling_factor=0, osd_method="osd_e", osd_order=5)
    d0, d1 = OracleDecoder(H, **kw), OracleDecoder(H, osd_e_bit_order=1, **kw)
    differ = 0
    for bits in itertools.product([0, 1], repeat=6):
        s = np.array(bits, dtype=np.uint8)
        r0, r1 = d0.decode(s), d1.decode(s)
        for r, order in ((r0, 0), (r1, 1)):
            ref = ref_numpy.bposd_decode(H, s, list(probs), 2, "ms", 0, "osd_e", 5, e_bit_order=order)
            assert (r["osd0"] == ref["osd0"]).all() and (r["osdw"] == ref["osdw"]).all(), (bits, order)
        assert (r0["osd0"] == r1["osd0"]).all() and r0["osdw"].sum() == r1["osdw"].sum()
        differ += int((r0["osdw"] != r1["osdw"]).any())
    assert differ > 0


def test_zero_syndrome_shortcut(surface13):
    dec = OracleDecoder(surface13.hz, error_rate=0.05, osd_method="osd_cs", osd_order=3)
    r = dec.decode(np.zeros(6, dtype=np.uint8))
    assert r["converged"] == 1 and r["iters"] == 0 and not r["osdw"].any() and not r["osd0"].any()


def test_hgp_layout_matches_reference_fixture(hgp400):
    """hgp(mkmn_16_4_6) == examples/codes/hgp_codes/hgp_(4,7)-[[400,16,6]]_{hx,hz}.txt."""
    import os

    fx = np.load(os.path.join(os.path.dirname(__file__), "golden", "hgp_400_16_6_reference_fixture.npz"))
    m, n = fx["shape"]
    hx = np.unpackbits(fx["hx"], axis=1)[:, :n]
    hz = np.unpackbits(fx["hz"], axis=1)[:, :n]
    assert (hgp400.hx.toarray() == hx).all() and (hgp400.hz.toarray() == hz).all()
    assert (hgp400.N, hgp400.K) == (400, 16) and hgp400.test()


def test_code_parameters(surface13, h1922):
    """tests/test_hgp.py:16-18 (N=13,K=1); SURVEY §7: [[1922,50]] from the 31x31 circulant."""
    assert (surface13.N, surface13.K) == (13, 1) and surface13.test()
    assert (h1922.N, h1922.K) == (1922, 50) and h1922.hz.shape == (961, 1922) and h1922.hz.nnz == 5766
    assert h1922.test()
    steane = hgp(rep_code(2))  # tiny sanity: hgp of a 1x2 matrix
    assert steane.N == 5 and steane.test()


@pytest.mark.parametrize("seed_file,N,K", [("mkmn_16_4_6.txt", 400, 16), ("mkmn_20_5_8.txt", 625, 25), ("mkmn_24_6_10.txt", 900, 36)])
def test_logicals_span_the_reference_logical_space(seed_file, N, K):
    """The reference ships lx / lz for its three example codes (examples/codes/hgp_codes/*_{lx,lz}.txt, packed into
    tests/golden/hgp_reference_logicals_fixture.npz).  Our hgp() of the same seed must give the same code: the
    reference's logicals commute with our stabilisers, are independent of them, and span the same logical space as
    ours (css.py:75-95 fixes the space, not the basis)."""
    import os

    from bp_osd_amd.codes import gf2_rank, hgp

    here = os.path.join(os.path.dirname(__file__), "golden")
    code = hgp(np.loadtxt(os.path.join(here, seed_file)).astype(np.uint8))
    fx = np.load(os.path.join(here, "hgp_reference_logicals_fixture.npz"))
    lx = np.unpackbits(fx[f"lx_{N}"], axis=1)[:, :N]
    lz = np.unpackbits(fx[f"lz_{N}"], axis=1)[:, :N]
    assert (code.N, code.K) == (N, K) and lx.shape == (K, N) and lz.shape == (K, N)
    hx, hz = code.hx.toarray().astype(np.int64), code.hz.toarray().astype(np.int64)
    assert not ((hz @ lx.T) % 2).any() and not ((hx @ lz.T) % 2).any()
    rx, rz = gf2_rank(hx), gf2_rank(hz)
    assert gf2_rank(np.vstack([hx, lx])) == rx + K and gf2_rank(np.vstack([hz, lz])) == rz + K
    assert gf2_rank(np.vstack([hx, lx, code.lx])) == rx + K and gf2_rank(np.vstack([hz, lz, code.lz])) == rz + K
    assert gf2_rank((lx.astype(np.int64) @ code.lz.T.astype(np.int64)) % 2) == K  # their X logicals pair with our Z logicals
