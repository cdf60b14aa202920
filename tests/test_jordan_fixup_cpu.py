"""The algebra behind the compact Gauss-Jordan panel phase of the HBM-resident OSD kernel: a pivot row of
an earlier panel need not take part in the panel's sequential pivot steps -- with the panel's pivot rows in their final,
mutually reduced state F_q it is brought up to date in one step,

    u  ^=  XOR of F_q over the pivot columns in which u has a one,      mask(u) = XOR of (mask(F_q) | own bit q),

and that equals what the sequential row updates produce (word and combination mask).  Pure numpy, bit matrices as arrays."""
import numpy as np


def _sequential_panel(rows, used):
    """Gauss-Jordan on a 64-column panel word the way the all-rows form does it: every row with a one in the pivot column
    adds the pivot row (pivot = first unused row with a one in the lowest candidate column); masks record the start-state
    pivot rows that went into each row.  Returns final words, masks, pivot list [(row, col)]."""
    rows = rows.copy()
    m = rows.shape[0]
    t = np.zeros((m, 64), dtype=np.uint8)
    used = used.copy()
    piv = []
    while True:
        cand = [(int(np.argmax(rows[r])), r) for r in range(m) if not used[r] and rows[r].any()]
        if not cand or len(piv) == 64:
            break
        col, r = min(cand)
        pw, pt = rows[r].copy(), t[r].copy()
        tq = pt.copy()
        tq[len(piv)] ^= 1
        for i in range(m):
            if i != r and rows[i, col]:
                rows[i] ^= pw
                t[i] ^= tq
        used[r] = True
        piv.append((r, col))
    return rows, t, piv


def test_jordan_fixup_equals_sequential_updates():
    rng = np.random.default_rng(12)
    for trial in range(30):
        m = int(rng.integers(8, 60))
        dens = float(rng.choice([0.05, 0.2, 0.5]))
        rows = (rng.random((m, 64)) < dens).astype(np.uint8)
        used = rng.random(m) < 0.4                      # pivot rows of earlier panels: any content, never candidates
        want_rows, want_t, piv = _sequential_panel(rows, used)
        # compact form: the pivot search sees the unused rows only ...
        idx = np.where(~used)[0]
        sub_rows, sub_t, sub_piv = _sequential_panel(rows[idx], np.zeros(len(idx), dtype=bool))
        assert [(int(idx[r]), c) for r, c in sub_piv] == piv
        got_rows, got_t = rows.copy(), np.zeros((m, 64), dtype=np.uint8)
        got_rows[idx], got_t[idx] = sub_rows, sub_t
        # ... and the earlier pivot rows are fixed up from the final pivot rows F_q
        for u in np.where(used)[0]:
            for q, (r, c) in enumerate(sub_piv):
                if rows[u, c]:
                    got_rows[u] ^= sub_rows[r]
                    got_t[u] ^= sub_t[r]
                    got_t[u, q] ^= 1
        assert (got_rows == want_rows).all() and (got_t == want_t).all()
        # no row keeps a one in a pivot column except the pivot row in its own column
        for q, (r, c) in enumerate(piv):
            col = want_rows[:, c].copy()
            assert col[r] == 1
            col[r] = 0
            assert not col.any()
