"""Lane-level model of osd_kernel's panel phase by 6-column sub-blocks (round 5).

The kernel finds the pivots of a 64-column panel six columns at a time: every unused row with a non-zero 6-bit value v
claims value v (first claimant publishes its panel word and combination mask), then two waves solve the 64 possible
values (lane v = value v) and write, per value, the XOR of the sub-block's <= 6 pivot rows (as they stood at the start of
the sub-block) that a row holding that value has to absorb; every row XORs the entry of its own value.  This script replays that on random panels and checks it
against the invariants of the one-pivot-per-barrier form it replaces:
  * the pivot columns are the greedy independent set of the column order;
  * afterwards every pivot column holds exactly one 1 over ALL rows (its pivot row), unused rows are zero on the panel;
  * row_final = row_start ^ XOR_{q in t} pivot_q(start of panel), for every row.
Run: python tools/panel_subblock_model.py  (CPU only, a few seconds).
"""
import random


def greedy_pivot_columns(rows, unused, ncols, rank_left):
    """Reference: column-by-column Gauss-Jordan on copies; returns the list of pivot columns."""
    rows = list(rows)
    unused = list(unused)
    piv = []
    for c in range(ncols):
        if len(piv) >= rank_left:
            break
        cand = [i for i in range(len(rows)) if unused[i] and (rows[i] >> c) & 1]
        if not cand:
            continue
        p = cand[-1]
        unused[p] = False
        piv.append(c)
        for i in range(len(rows)):
            if i != p and (rows[i] >> c) & 1:
                rows[i] ^= rows[p]
    return piv


def panel_by_subblocks(rows, unused, nbc, rank_left, rng):
    """rows: panel words (ints < 2^64); unused: flags.  Returns (rows, t, pinfo, npiv) like the kernel's registers."""
    nrows = len(rows)
    rows = list(rows)
    t = [0] * nrows
    pinfo = [-1] * nrows
    unused = list(unused)
    npiv = 0
    c0 = 0
    while c0 < nbc and npiv < rank_left:
        wsb = min(6, nbc - c0)
        vmask = (1 << wsb) - 1
        # A: claims (first claimant of a value wins; arrival order is arbitrary)
        b = [(rows[i] >> c0) & vmask for i in range(nrows)]
        order = list(range(nrows))
        rng.shuffle(order)
        claimed = {}
        won = [False] * nrows
        pbuf = {}
        for i in order:
            if unused[i] and b[i]:
                if b[i] not in claimed:
                    claimed[b[i]] = i
                    won[i] = True
                    pbuf[b[i]] = (rows[i], t[i])
        # C: solve on 64 value lanes (X) + the pivot rows' own entries by value lane (Y); tags are masks over the
        # sub-block's COLUMNS (bit j = has absorbed the pivot row of column c0 + j, as it stood at the sub-block's start)
        r = list(range(64))
        tg = [0] * 64
        av = [v in claimed for v in range(64)]
        r2 = [0] * 64
        tg2 = [0] * 64
        col2 = [-1] * 64
        lj = [0] * 6
        pivmask = 0
        room = rank_left - npiv
        for j in range(wsb):
            if bin(pivmask).count("1") >= room:
                break
            cand = [v for v in range(64) if av[v] and (r[v] >> j) & 1]
            if not cand:
                continue
            l = cand[0]
            p, pt = r[l], tg[l]
            for v in range(64):
                if (r[v] >> j) & 1:
                    r[v] ^= p
                    tg[v] ^= pt ^ (1 << j)
                if (r2[v] >> j) & 1:
                    r2[v] ^= p
                    tg2[v] ^= pt ^ (1 << j)
            assert col2[l] < 0
            r2[l], tg2[l], col2[l] = p, pt, j
            lj[j] = l
            pivmask |= 1 << j
        nps = bin(pivmask).count("1")
        # E: every row looks up its value
        new_rows = list(rows)
        for i in range(nrows):
            tag = tg[b[i]]
            is_piv = won[i] and col2[b[i]] >= 0
            if is_piv:
                tag = tg2[b[i]]
                j = col2[b[i]]
                pinfo[i] = ((c0 + j) << 6) | (npiv + bin(pivmask & ((1 << j) - 1)).count("1"))
                unused[i] = False
            for j in range(6):
                if (tag >> j) & 1:
                    assert (pivmask >> j) & 1
                    pw, ptt = pbuf[lj[j]]
                    new_rows[i] ^= pw
                    t[i] ^= ptt ^ (1 << (npiv + bin(pivmask & ((1 << j) - 1)).count("1")))
        rows = new_rows
        npiv += nps
        c0 += 6
    return rows, t, pinfo, npiv, unused


def check(seed):
    rng = random.Random(seed)
    nrows = rng.choice([5, 40, 130, 961])
    nbc = rng.choice([64, 64, 63, 37, 6, 5, 1])
    dens = rng.choice([0.02, 0.1, 0.5])
    start = []
    unused = []
    for i in range(nrows):
        w = 0
        for c in range(64):
            if rng.random() < dens:
                w |= 1 << c
        u = rng.random() < 0.7
        start.append(w)
        unused.append(u)
    rank_left = rng.choice([1000, 1000, 3, 17, 64])
    want = greedy_pivot_columns(start, unused, nbc, rank_left)
    rows, t, pinfo, npiv, unused_after = panel_by_subblocks(start, unused, nbc, rank_left, rng)
    got = sorted((pi >> 6, pi & 63, i) for i, pi in enumerate(pinfo) if pi >= 0)
    assert [g[0] for g in got] == want, (seed, want, got)
    assert [g[1] for g in got] == list(range(npiv)), (seed, got)
    prow = {q: i for (_, q, i) in got}
    for c, q, i in got:
        col = [(rows[x] >> c) & 1 for x in range(nrows)]
        assert sum(col) == 1 and col[i] == 1, (seed, c)
    lastc = want[-1] if (want and npiv >= rank_left) else nbc - 1
    for i in range(nrows):
        acc = start[i]
        for q in range(npiv):
            if (t[i] >> q) & 1:
                acc ^= start[prow[q]]
        assert acc == rows[i], (seed, i)
        if unused_after[i]:
            assert rows[i] & ((1 << (lastc + 1)) - 1) == 0, (seed, i)


if __name__ == "__main__":
    for seed in range(2000):
        check(seed)
    print("panel by sub-blocks: 2000 random panels agree with column-by-column Gauss-Jordan")
