"""Second, deliberately different CPU restatement of the BP+OSD path (pure Python / numpy,
dense matrices, no code shared with oracle/bposd_oracle.c).  TEST INFRASTRUCTURE ONLY.

Used to cross-check the C oracle bit-for-bit on small codes.
Follows the same behavioural spec (reference call sites
/root/reference/README.md:178-202, /root/reference/src/bposd/css_decode_sim.py:444-463).
"""
from __future__ import annotations

import itertools
import math

import numpy as np

DBL_MAX = np.finfo(np.float64).max


def bp_decode(H, syndrome, probs, max_iter, bp_method, ms_scaling_factor):
    """Flooding BP.  Returns (decoding, converged, iterations, llr)."""
    H = np.asarray(H, dtype=np.uint8)
    m, n = H.shape
    rows = [list(np.nonzero(H[c])[0]) for c in range(m)]
    cols = [list(np.nonzero(H[:, i])[0]) for i in range(n)]
    llr0 = [math.log((1 - probs[i]) / probs[i]) for i in range(n)]
    b2c = {(c, i): llr0[i] for c in range(m) for i in rows[c]}
    c2b = {}
    max_iter = max_iter if max_iter > 0 else n
    llr = list(llr0)
    dec = [0] * n
    conv = False
    it_done = 0
    for it in range(1, max_iter + 1):
        if bp_method == "ps":
            for c in range(m):
                temp = 1.0
                fwd = []
                for i in rows[c]:
                    fwd.append(temp)
                    temp *= math.tanh(b2c[(c, i)] / 2)
                temp = 1.0
                for k in range(len(rows[c]) - 1, -1, -1):
                    i = rows[c][k]
                    x = fwd[k] * temp
                    sign = -1 if syndrome[c] else 1
                    try:
                        val = math.log((1 + x) / (1 - x))
                    except (ZeroDivisionError, ValueError):
                        val = math.inf if x > 0 else -math.inf
                    c2b[(c, i)] = sign * val
                    temp *= math.tanh(b2c[(c, i)] / 2)
        else:
            alpha = (1.0 - 2.0 ** (-it)) if ms_scaling_factor == 0 else ms_scaling_factor
            for c in range(m):
                total = int(syndrome[c])
                temp = DBL_MAX
                fwd = []
                for i in rows[c]:
                    v = b2c[(c, i)]
                    if v <= 0:
                        total += 1
                    fwd.append(temp)
                    if abs(v) < temp:
                        temp = abs(v)
                temp = DBL_MAX
                for k in range(len(rows[c]) - 1, -1, -1):
                    i = rows[c][k]
                    v = b2c[(c, i)]
                    sgn = total + (1 if v <= 0 else 0)
                    mag = fwd[k]
                    if temp < mag:
                        mag = temp
                    factor = (1 if sgn % 2 == 0 else -1) * alpha
                    c2b[(c, i)] = mag * factor
                    if abs(v) < temp:
                        temp = abs(v)
        cand = [0] * m
        for i in range(n):
            temp = llr0[i]
            for c in cols[i]:
                b2c[(c, i)] = temp
                temp += c2b[(c, i)]
            llr[i] = temp
            if temp <= 0:
                dec[i] = 1
                for c in cols[i]:
                    cand[c] ^= 1
            else:
                dec[i] = 0
        it_done = it
        if all(int(cand[c]) == int(syndrome[c]) for c in range(m)):
            conv = True
            break
        for i in range(n):
            temp = 0.0
            for c in reversed(cols[i]):
                b2c[(c, i)] += temp
                temp += c2b[(c, i)]
    return np.array(dec, dtype=np.uint8), conv, it_done, np.array(llr, dtype=np.float64)


def _gf2_solve_unique(A, t):
    """Solve A x = t over GF(2) for A with full column rank (dense elimination on ints)."""
    m, r = A.shape
    rows = [int("".join(str(int(b)) for b in A[i][::-1]), 2) | (int(t[i]) << r) if r else (int(t[i]) << r)
            for i in range(m)]
    piv_of_col = {}
    used = set()
    for j in range(r):
        bit = 1 << j
        p = next((i for i in range(m) if i not in used and rows[i] & bit), None)
        assert p is not None, "matrix does not have full column rank"
        used.add(p)
        piv_of_col[j] = p
        for i in range(m):
            if i != p and rows[i] & bit:
                rows[i] ^= rows[p]
    x = np.zeros(r, dtype=np.uint8)
    for j in range(r):
        x[j] = (rows[piv_of_col[j]] >> r) & 1
    return x


def osd_decode(H, syndrome, llr, probs, osd_method, osd_order, tie_policy=0, weight_fn=0, e_bit_order=0):
    """Returns (osd0, osdw, order, pivot_positions)."""
    H = np.asarray(H, dtype=np.uint8)
    m, n = H.shape
    idx = np.arange(n)
    if tie_policy == 1:
        idx = idx[::-1]
    order = idx[np.argsort(np.asarray(llr)[idx], kind="stable")]
    Hp = H[:, order]
    # greedy independent columns, left to right, by incremental basis reduction of column vectors
    basis = {}  # leading row -> int vector
    piv = []
    for j in range(n):
        v = int("".join(str(int(b)) for b in Hp[:, j][::-1]), 2)
        while v:
            lead = v.bit_length() - 1
            if lead in basis:
                v ^= basis[lead]
            else:
                basis[lead] = v
                piv.append(j)
                break
    nonpiv = [j for j in range(n) if j not in set(piv)]
    A = Hp[:, piv]

    def weight(x):
        w = 0.0
        for i in range(n):
            if x[i]:
                w += 1.0 if weight_fn == 1 else math.log(1 / probs[i])
        return w

    def solution(tsel):
        t = np.array(syndrome, dtype=np.uint8).copy()
        for k in tsel:
            t ^= Hp[:, nonpiv[k]]
        xs = _gf2_solve_unique(A, t)
        xp = np.zeros(n, dtype=np.uint8)
        xp[piv] = xs
        for k in tsel:
            xp[nonpiv[k]] = 1
        x = np.zeros(n, dtype=np.uint8)
        x[order] = xp
        return x

    osd0 = solution([])
    best, best_w = osd0, weight(osd0)
    kp = len(nonpiv)
    cands = []
    if osd_method == "osd_e" and osd_order > 0:
        for pat in range(1, 2 ** osd_order):
            cands.append([(osd_order - 1 - b if e_bit_order else b) for b in range(osd_order) if (pat >> b) & 1])
    elif osd_method == "osd_cs" and osd_order > 0:
        cands += [[k] for k in range(kp)]
        cands += [[a, b] for a, b in itertools.combinations(range(osd_order), 2)]
    for tsel in cands:
        x = solution(tsel)
        w = weight(x)
        if w < best_w:
            best, best_w = x, w
    return osd0, best, order, piv


def bposd_decode(H, syndrome, probs, max_iter, bp_method, ms_scaling_factor, osd_method, osd_order,
                 tie_policy=0, weight_fn=0, e_bit_order=0):
    H = np.asarray(H, dtype=np.uint8)
    n = H.shape[1]
    if not np.any(syndrome):
        z = np.zeros(n, dtype=np.uint8)
        llr0 = np.array([math.log((1 - probs[i]) / probs[i]) for i in range(n)])
        return dict(osdw=z, osd0=z.copy(), bp=z.copy(), converged=True, iters=0, llr=llr0)
    dec, conv, its, llr = bp_decode(H, syndrome, probs, max_iter, bp_method, ms_scaling_factor)
    if conv:
        return dict(osdw=dec, osd0=dec.copy(), bp=dec.copy(), converged=True, iters=its, llr=llr)
    osd0, osdw, _, _ = osd_decode(H, syndrome, llr, probs, osd_method, osd_order, tie_policy, weight_fn, e_bit_order)
    return dict(osdw=osdw, osd0=osd0, bp=dec, converged=False, iters=its, llr=llr)
