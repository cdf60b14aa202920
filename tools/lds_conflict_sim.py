"""Simulate gfx950 LDS bank conflicts of the BP kernel's message accesses for a given layout.

Model (MI355X_MICROARCH.md §LDS): ds_read_b64 is serviced in two 32-lane groups, 64 banks of 4 B,
bank = (addr/4) % 64; ds_write_b64 in four 16-lane groups, bank = (addr/4) % 32.  Cycles per group =
max number of distinct 8-byte words mapped onto one bank (identical addresses broadcast)."""
import numpy as np


def group_cycles(addrs, nbanks):
    """addrs: byte addresses of 8-byte accesses by the lanes of one group."""
    words = {}
    for a in addrs:
        for half in (0, 4):
            b = ((a + half) // 4) % nbanks
            words.setdefault(b, set()).add((a + half) // 4)
    return max(len(v) for v in words.values()) if words else 0


def instr_cycles(addrs, kind):
    addrs = list(addrs)
    if kind == "read":
        groups = [addrs[0:32], addrs[32:64]]
        nb = 64
    else:
        groups = [addrs[i:i + 16] for i in range(0, 64, 16)]
        nb = 32
    return sum(group_cycles([a for a in g if a is not None], nb) for g in groups)


def simulate(H, MP, NT, CPT, VPT, chk_slot, bit_slot):
    """chk_slot[c] = LDS slot of check c; bit_slot[i] = position (thread = pos % NT, round = pos // NT)."""
    import scipy.sparse as sp

    H = sp.csr_matrix(H)
    m, n = H.shape
    # edge positions
    pos = {}
    for c in range(m):
        for k, i in enumerate(H.indices[H.indptr[c]:H.indptr[c + 1]]):
            pos[(c, i)] = (k * MP + chk_slot[c]) * 8
    Hc = H.tocsc()
    ideal = actual = 0
    inv_bit = {bit_slot[i]: i for i in range(n)}
    dv = int(np.diff(Hc.indptr).max())
    for r in range(VPT):
        for w0 in range(0, NT, 64):
            for d in range(dv):
                addrs = []
                for lane in range(64):
                    p = r * NT + w0 + lane
                    i = inv_bit.get(p)
                    if i is None:
                        addrs.append(None)
                        continue
                    rows = Hc.indices[Hc.indptr[i]:Hc.indptr[i + 1]]
                    addrs.append(pos[(rows[d], i)] if d < len(rows) else None)
                if all(a is None for a in addrs):
                    continue
                actual += instr_cycles(addrs, "read") + instr_cycles(addrs, "write")
                ideal += 2 + 4
    return ideal, actual


if __name__ == "__main__":
    import sys
    sys.path.insert(0, ".")
    from bp_osd_amd.codes import h1922

    H = h1922(compute_logicals=False).hz
    m, n = H.shape
    for name, cs, bs in (
        ("natural", np.arange(m), np.arange(n)),
        ("pad31->32", (np.arange(m) // 31) * 32 + np.arange(m) % 31, (np.arange(n) // 31) * 32 + np.arange(n) % 31),
    ):
        for NT, CPT, VPT in ((512, 2, 4), (1024, 1, 2), (256, 4, 8)):
            ideal, actual = simulate(H, 1024, NT, CPT, VPT, cs, bs)
            print(f"{name:10s} NT={NT:4d}: bit-pass LDS cycles ideal {ideal} simulated {actual}  (+{100*(actual/ideal-1):.0f}%)")


def breakdown(H, MP, NT, VPT, chk_slot, bit_slot):
    import scipy.sparse as sp
    H = sp.csr_matrix(H); m, n = H.shape
    pos = {}
    for c in range(m):
        for k, i in enumerate(H.indices[H.indptr[c]:H.indptr[c + 1]]):
            pos[(c, i)] = (k * MP + chk_slot[c]) * 8
    Hc = H.tocsc(); inv_bit = {bit_slot[i]: i for i in range(n)}
    rd = wr = 0; nin = 0
    for r in range(VPT):
        for w0 in range(0, NT, 64):
            for d in range(3):
                addrs = []
                for lane in range(64):
                    i = inv_bit.get(r * NT + w0 + lane)
                    if i is None: addrs.append(None); continue
                    rows = Hc.indices[Hc.indptr[i]:Hc.indptr[i + 1]]
                    addrs.append(pos[(rows[d], i)])
                if all(a is None for a in addrs): continue
                rd += instr_cycles(addrs, "read"); wr += instr_cycles(addrs, "write"); nin += 1
    return nin, rd, wr
