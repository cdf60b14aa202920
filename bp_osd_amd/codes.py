"""Code construction helpers for the BP+OSD hot path (CPU, setup-time only).

These are the minimum the decode path and its benchmark need to *have inputs*:
parity-check matrices in the reference's hypergraph-product layout and a valid
logical-operator basis for the logical-error-rate half of the metric.  They are
not accelerated (one-off setup work).

Reference behaviour followed (read as text, nothing imported):
  * HGP layout  hx=[h1 (x) I_n2 | I_m1 (x) h2^T], hz=[I_n1 (x) h2 | h1^T (x) I_m2]
    -- /root/reference/src/bposd/hgp.py:48-54
  * N, K from the seed ranks -- /root/reference/src/bposd/hgp.py:41-44
  * logicals: rows of ker(hx) not in rowspace(hz) (and vice versa)
    -- /root/reference/src/bposd/css.py:75-95
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp

__all__ = [
    "rep_code",
    "ring_code",
    "hamming_code",
    "circulant",
    "gf2_rank",
    "gf2_row_echelon",
    "gf2_nullspace",
    "gf2_pivot_rows",
    "hgp",
    "HgpCode",
    "CssCode",
    "h1922",
    "surface13",
    "regular_ldpc_seed",
    "l29k",
]


# ----------------------------------------------------------------------------
# classical seed codes
# ----------------------------------------------------------------------------
def rep_code(d: int) -> np.ndarray:
    """(d-1) x d repetition-code parity-check matrix, rows (i, i+1).

    Same matrix the reference obtains from ``ldpc.codes.rep_code(d)``
    (/root/reference/README.md:146-150)."""
    h = np.zeros((d - 1, d), dtype=np.uint8)
    for i in range(d - 1):
        h[i, i] = 1
        h[i, i + 1] = 1
    return h


def ring_code(d: int) -> np.ndarray:
    """d x d cyclic repetition code (adds the wrap-around check)."""
    h = np.zeros((d, d), dtype=np.uint8)
    for i in range(d):
        h[i, i] = 1
        h[i, (i + 1) % d] = 1
    return h


def hamming_code(r: int) -> np.ndarray:
    """r x (2^r - 1) Hamming parity-check matrix; column j is binary(j+1), MSB in row 0."""
    n = 2**r - 1
    h = np.zeros((r, n), dtype=np.uint8)
    for j in range(n):
        for i in range(r):
            h[i, j] = ((j + 1) >> (r - 1 - i)) & 1
    return h


def circulant(n: int, shifts) -> np.ndarray:
    """n x n circulant with ones at columns (i + s) mod n in row i for s in shifts."""
    h = np.zeros((n, n), dtype=np.uint8)
    for i in range(n):
        for s in shifts:
            h[i, (i + s) % n] ^= 1
    return h


def regular_ldpc_seed(m: int, n: int, col_w: int, row_w: int, seed: int = 0, tries: int = 200) -> np.ndarray:
    """Seeded (col_w,row_w)-regular m x n matrix by the configuration model,
    rejecting double edges (best effort on 4-cycles).  Used for the large
    L29k configuration."""
    assert m * row_w == n * col_w
    rng = np.random.default_rng(seed)
    for _ in range(tries):
        sockets_c = np.repeat(np.arange(m), row_w)
        sockets_v = np.repeat(np.arange(n), col_w)
        rng.shuffle(sockets_v)
        h = np.zeros((m, n), dtype=np.uint8)
        ok = True
        for c, v in zip(sockets_c, sockets_v):
            if h[c, v]:
                ok = False
                break
            h[c, v] = 1
        if ok:
            return h
    # fall back: repair double edges by swapping
    h = np.zeros((m, n), dtype=np.int64)
    sockets_c = np.repeat(np.arange(m), row_w)
    sockets_v = np.repeat(np.arange(n), col_w)
    rng.shuffle(sockets_v)
    for _ in range(100000):
        h[:] = 0
        np.add.at(h, (sockets_c, sockets_v), 1)
        bad = np.argwhere(h[sockets_c, sockets_v] > 1).ravel()
        if bad.size == 0:
            return h.astype(np.uint8)
        i = bad[0]
        j = rng.integers(0, sockets_v.size)
        sockets_v[i], sockets_v[j] = sockets_v[j], sockets_v[i]
    raise RuntimeError("could not build a simple regular graph")


# ----------------------------------------------------------------------------
# dense GF(2) helpers (packed into python ints per row: fast enough for setup)
# ----------------------------------------------------------------------------
def _rows_to_ints(a: np.ndarray) -> list:
    a = np.asarray(a, dtype=np.uint8) & 1
    out = []
    for row in a:
        bits = np.packbits(row, bitorder="little")  # bit j of the int == a[j]
        out.append(int.from_bytes(bits.tobytes(), "little"))
    return out


def _dense(a) -> np.ndarray:
    if sp.issparse(a):
        a = a.toarray()
    return (np.asarray(a).astype(np.int64) & 1).astype(np.uint8)


def gf2_row_echelon(a, full: bool = False):
    """Row echelon form over GF(2).

    Returns (R, rank, T, pivot_cols) with R = T @ a (mod 2).  Column order is
    natural (left to right)."""
    a = _dense(a)
    m, n = a.shape
    rows = _rows_to_ints(a)
    trans = [1 << i for i in range(m)]
    rank = 0
    pivots = []
    for j in range(n):
        if rank == m:
            break
        bit = 1 << j
        p = -1
        for r in range(rank, m):
            if rows[r] & bit:
                p = r
                break
        if p < 0:
            continue
        rows[rank], rows[p] = rows[p], rows[rank]
        trans[rank], trans[p] = trans[p], trans[rank]
        lo = 0 if full else rank + 1
        for r in range(lo, m):
            if r != rank and rows[r] & bit:
                rows[r] ^= rows[rank]
                trans[r] ^= trans[rank]
        pivots.append(j)
        rank += 1
    R = np.zeros((m, n), dtype=np.uint8)
    T = np.zeros((m, m), dtype=np.uint8)
    for r in range(m):
        v = rows[r]
        while v:
            low = v & -v
            R[r, low.bit_length() - 1] = 1
            v ^= low
        v = trans[r]
        while v:
            low = v & -v
            T[r, low.bit_length() - 1] = 1
            v ^= low
    return R, rank, T, pivots


def gf2_rank(a) -> int:
    a = _dense(a)
    if a.size == 0:
        return 0
    return gf2_row_echelon(a)[1]


def gf2_nullspace(a) -> np.ndarray:
    """Basis (rows) of {x : a x = 0 mod 2}."""
    a = _dense(a)
    m, n = a.shape
    # row-reduce a^T augmented with identity: rows of T giving zero rows of R span ker(a)
    R, rank, T, _ = gf2_row_echelon(a.T)
    return T[rank:, :].copy()


def gf2_pivot_rows(a) -> list:
    """Indices of a maximal set of linearly independent rows, greedy top to bottom."""
    a = _dense(a)
    _, _, _, piv = gf2_row_echelon(a.T)
    return piv


# ----------------------------------------------------------------------------
# hypergraph product
# ----------------------------------------------------------------------------
class HgpCode:
    """Container mirroring the attributes the reference's ``hgp`` object exposes
    that the decode path reads: hx, hz, lx, lz, N, K
    (/root/reference/src/bposd/hgp.py:26-57, css.py:75-95)."""

    def __init__(self, h1, h2=None, compute_logicals=True):
        """compute_logicals: True = the reference's generic route (nullspace + pivot rows, css.py:75-95; dense GF(2)
        work on the N-column matrices), "closed_form" = the product structure's own basis (closed_form_logicals below;
        only the seed matrices are row-reduced, so it also serves the 29524-qubit code), False = none."""
        h1 = _dense(h1)
        h2 = h1.copy() if h2 is None else _dense(h2)
        self.h1, self.h2 = h1, h2
        m1, n1 = h1.shape
        m2, n2 = h2.shape
        r1, r2 = gf2_rank(h1), gf2_rank(h2)
        k1, k1t = n1 - r1, m1 - r1
        k2, k2t = n2 - r2, m2 - r2
        self.N = n1 * n2 + m1 * m2
        self.K = k1 * k2 + k1t * k2t
        s1, s2 = sp.csr_matrix(h1), sp.csr_matrix(h2)
        I = lambda k: sp.identity(k, format="csr", dtype=np.uint8)
        hx1 = sp.kron(s1, I(n2), format="csr")
        hx2 = sp.kron(I(m1), s2.T, format="csr")
        self.hx = sp.hstack([hx1, hx2], format="csr").astype(np.uint8)
        hz1 = sp.kron(I(n1), s2, format="csr")
        hz2 = sp.kron(s1.T, I(m2), format="csr")
        self.hz = sp.hstack([hz1, hz2], format="csr").astype(np.uint8)
        self.hx.sort_indices()
        self.hz.sort_indices()
        self.lx = self.lz = None
        if isinstance(compute_logicals, str):
            if compute_logicals != "closed_form":
                raise ValueError("compute_logicals must be True, False or 'closed_form'")
            self.lx, self.lz = self.closed_form_logicals()
        elif compute_logicals:
            self.lx = self._logicals(self.hz, self.hx)
            self.lz = self._logicals(self.hx, self.hz)
            assert self.lx.shape[0] == self.K and self.lz.shape[0] == self.K

    def closed_form_logicals(self):
        """(lx, lz) from the seed codes alone (Tillich-Zemor).  With hx = [h1 (x) I | I (x) h2^T] and
        hz = [I (x) h2 | h1^T (x) I] (hgp.py:48-54), a Z-type operator (x (x) y | 0) commutes with every X check iff
        h1 x = 0, and (x (x) h2^T u | 0) is a product of Z checks, so y only matters modulo rowspace(h2): take x over a
        basis of ker(h1) and y = e_j over the free (non-pivot) columns of h2.  Likewise (0 | a (x) b) with h2^T b = 0 and
        a = e_i over the free columns of h1^T.  That is k1 k2 + k1t k2t = K operators; lx follows by exchanging the roles
        of the two seeds.  The logical error test (lz . residual != 0 for some row, css_decode_sim.py:257-272) does not
        depend on which basis of the logical space is used."""
        h1, h2 = self.h1, self.h2
        m1, n1 = h1.shape
        m2, n2 = h2.shape

        def free_cols(a):
            piv = set(gf2_row_echelon(a)[3])
            return [j for j in range(a.shape[1]) if j not in piv]

        def unit(k, i):
            e = np.zeros(k, dtype=np.uint8)
            e[i] = 1
            return e

        def block(first, second):  # rows (u (x) v | 0) for (u, v) in first, (0 | a (x) b) for (a, b) in second
            rows = [np.concatenate([np.kron(u, v), np.zeros(m1 * m2, dtype=np.uint8)]) for u, v in first]
            rows += [np.concatenate([np.zeros(n1 * n2, dtype=np.uint8), np.kron(a, b)]) for a, b in second]
            return np.array(rows, dtype=np.uint8).reshape(len(rows), self.N)

        ker1, ker2 = gf2_nullspace(h1), gf2_nullspace(h2)        # ker(h1) in F^n1, ker(h2) in F^n2
        ker1t, ker2t = gf2_nullspace(h1.T), gf2_nullspace(h2.T)  # ker(h1^T) in F^m1, ker(h2^T) in F^m2
        lz = block([(x, unit(n2, j)) for x in ker1 for j in free_cols(h2)],
                   [(unit(m1, i), b) for i in free_cols(h1.T) for b in ker2t])
        lx = block([(unit(n1, i), y) for i in free_cols(h1) for y in ker2],
                   [(a, unit(m2, j)) for a in ker1t for j in free_cols(h2.T)])
        assert lx.shape[0] == self.K and lz.shape[0] == self.K
        return lx, lz

    @staticmethod
    def _logicals(h_commute, h_stab) -> np.ndarray:
        """ker(h_commute) modulo rowspace(h_stab) -- css.py:75-95."""
        ker = gf2_nullspace(h_commute)
        hs = _dense(h_stab)
        stack = np.vstack([hs, ker])
        rank_s = gf2_rank(hs)
        piv = gf2_pivot_rows(stack)
        sel = piv[rank_s:]
        return stack[sel, :].astype(np.uint8)

    def test(self) -> bool:
        """Validity checks of css.py:122-191 (commutation, logicals in kernels,
        logicals pair up with full rank)."""
        hx, hz = _dense(self.hx), _dense(self.hz)
        if ((hx.astype(np.int64) @ hz.T.astype(np.int64)) % 2).any():
            return False
        if self.lx is None:
            return True
        lx, lz = self.lx.astype(np.int64), self.lz.astype(np.int64)
        if ((hz.astype(np.int64) @ lx.T) % 2).any():
            return False
        if ((hx.astype(np.int64) @ lz.T) % 2).any():
            return False
        return gf2_rank((lx @ lz.T) % 2) == self.K


class CssCode:
    """CSS code from a pair (hx, hz): the attributes the reference's ``css_code`` exposes to the harness
    (/root/reference/src/bposd/css.py:7-95; used at css_decode_sim.py:376-380): hx, hz, lx, lz, N, K."""

    def __init__(self, hx, hz):
        self.hx = sp.csr_matrix(_dense(hx)).astype(np.uint8)
        self.hz = sp.csr_matrix(_dense(hz)).astype(np.uint8)
        self.hx.sort_indices()
        self.hz.sort_indices()
        self.N = self.hx.shape[1]
        if self.N != self.hz.shape[1]:
            raise ValueError("Code block length (N) inconsistent!")
        self.K = self.N - gf2_rank(self.hx) - gf2_rank(self.hz)  # css.py:46-51
        self.lx = HgpCode._logicals(self.hz, self.hx)  # css.py:75-95: ker(hz) modulo rowspace(hx)
        self.lz = HgpCode._logicals(self.hx, self.hz)

    test = HgpCode.test


def hgp(h1, h2=None, compute_logicals=True) -> HgpCode:
    return HgpCode(h1, h2, compute_logicals)


def surface13() -> HgpCode:
    """[[13,1,3]] surface code = hgp(rep_code(3), rep_code(3)) (README.md:146-150)."""
    return hgp(rep_code(3))


def h1922(compute_logicals=True) -> HgpCode:
    """[[1922,50]] HGP of the 31x31 circulant 1 + x^2 + x^5 (SURVEY.md §7 'config
    ambiguities', §8d): hx, hz are 961 x 1922, row weight 6, column weight 3."""
    return hgp(circulant(31, (0, 2, 5)), compute_logicals=compute_logicals)


def l29k(seed: int = 0, compute_logicals=False) -> HgpCode:
    """The large configuration of SURVEY.md §8d: HGP of a seeded (5,6)-regular 110 x 132 matrix --
    hx, hz are 14520 x 29524, check weight 11, bit weight 5 or 6, 159 720 non-zeros.  The reference's generic logicals
    (dense nullspace of a 14520 x 29524 matrix) are out of reach of the setup-time GF(2) helpers above; pass
    compute_logicals="closed_form" for the product-structure basis (484 x 29524), which is all the LER needs."""
    return hgp(regular_ldpc_seed(110, 132, 5, 6, seed=seed), compute_logicals=compute_logicals)
