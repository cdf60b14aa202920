"""Code construction pinned by what the reference itself records (CPU only):
README.ipynb cell 4 (Steane logicals), tests/test_css.py:13-15 and tests/test_hgp.py:16-18 (N, K, D)."""
import itertools

import numpy as np
import pytest

from bp_osd_amd.codes import CssCode, gf2_rank, hamming_code, hgp, rep_code, h1922


def _distance(code):
    """Brute force: lightest operator of ker(h_commute) that is not a product of stabilisers (tiny codes only)."""
    best = code.N
    for h_commute, logicals in ((code.hx.toarray(), code.lx), (code.hz.toarray(), code.lz)):
        for w in range(1, best):
            for supp in itertools.combinations(range(code.N), w):
                v = np.zeros(code.N, dtype=np.int64)
                v[list(supp)] = 1
                if not (h_commute @ v % 2).any() and (logicals.astype(np.int64) @ v % 2).any():
                    best = min(best, w)
                    break
            if best <= w:
                break
    return best


def test_steane_code_matches_the_reference_notebook():
    """/root/reference/README.ipynb cells 2-4: hx = hz = hamming_code(3), Lx = Lz = [[1 1 1 0 0 0 0]];
    /root/reference/tests/test_css.py:13-15: N = 7, K = 1, D = 3."""
    h = hamming_code(3)
    assert (h == np.array([[0, 0, 0, 1, 1, 1, 1], [0, 1, 1, 0, 0, 1, 1], [1, 0, 1, 0, 1, 0, 1]])).all()
    code = CssCode(h, h)
    assert (code.N, code.K) == (7, 1)
    assert (code.lx == np.array([[1, 1, 1, 0, 0, 0, 0]])).all()
    assert (code.lz == np.array([[1, 1, 1, 0, 0, 0, 0]])).all()
    assert code.test()
    assert _distance(code) == 3


def test_surface_code_parameters():
    """/root/reference/tests/test_hgp.py:16-18 and README.ipynb cell 10: hgp(rep_code(3), rep_code(3)) = [[13,1,3]]."""
    code = hgp(rep_code(3), rep_code(3))
    assert (code.N, code.K) == (13, 1)
    assert code.test()
    assert _distance(code) == 3


def test_invalid_css_code_is_rejected():
    """README.ipynb cell 8: hx = hz = rep_code(7) does not commute."""
    code = CssCode(rep_code(7), rep_code(7))
    assert code.K == 7 - 6 - 6
    assert not code.test()


@pytest.mark.parametrize("name", ["s13", "ham_rep", "hgp400", "h1922"])
def test_closed_form_logicals_span_the_same_logical_space(name):
    """HgpCode.closed_form_logicals (used for the 29524-qubit code, where the generic nullspace route is out of reach)
    against the reference's generic construction (css.py:75-95) on codes where both run."""
    import os

    seed = np.loadtxt(os.path.join(os.path.dirname(__file__), "golden", "mkmn_16_4_6.txt")).astype(np.uint8)
    code = {"s13": lambda: hgp(rep_code(3)), "ham_rep": lambda: hgp(hamming_code(3), rep_code(4)),
            "hgp400": lambda: hgp(seed), "h1922": h1922}[name]()
    lx, lz = code.closed_form_logicals()
    hx, hz = code.hx.toarray().astype(np.int64), code.hz.toarray().astype(np.int64)
    assert lx.shape == (code.K, code.N) and lz.shape == (code.K, code.N)
    assert not (hx @ lz.T % 2).any() and not (hz @ lx.T % 2).any()
    rz, rx = gf2_rank(hz), gf2_rank(hx)
    assert gf2_rank(np.vstack([hz, lz])) == rz + code.K == gf2_rank(np.vstack([hz, lz, code.lz]))
    assert gf2_rank(np.vstack([hx, lx])) == rx + code.K == gf2_rank(np.vstack([hx, lx, code.lx]))
    assert gf2_rank(lx.astype(np.int64) @ lz.T % 2) == code.K


def test_closed_form_logicals_of_the_large_code():
    from bp_osd_amd.codes import l29k

    code = l29k(compute_logicals="closed_form")
    assert (code.N, code.K) == (29524, 484) and code.lz.shape == (484, 29524)
    assert not ((code.hx.astype(np.int32) @ code.lz.T.astype(np.int32)) % 2).any()
    assert not ((code.hz.astype(np.int32) @ code.lx.T.astype(np.int32)) % 2).any()
