// own_layout.h -- host-side tables for bp_own_kernel.hip.h: every check OWNS two of its bits (a perfect b-matching,
// Kuhn's augmenting paths with capacity two on the checks); the thread that runs a check also runs its two owned bits, so
// the messages on the two owned edges never leave that thread's registers.  The bits no check owns (n - 2m of them) are
// sorted by degree into 64-lane groups as in class_layout.h.  Check positions and the slot roles of the owned bits are
// annealed on the measured two-rule LDS cycle model (tools/microbench/lds_scatter_probe.hip):
//   ds_read_b64  = sum over the two half-waves of the largest number of lanes on one 8-byte column (slot mod 32)
//   ds_write_b64 = max(6, sum over the four quarter-waves of the largest number of lanes on one column (slot mod 16))
// Pure C++ (no HIP).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <vector>

namespace own_layout {

struct Tables {
    int NT = 0, MP = 0, NTMAX = 0, DC = 0;
    std::vector<int> pos_chk;    // [NTMAX]          check at a position (thread), -1 = none
    std::vector<int> own_bit;    // [2 * NTMAX]      owned bit of (slot r, thread), -1 = padding position
    std::vector<int> own_rd;     // [6 * NTMAX]      entry (3 r + j) * NTMAX + t: LDS slot read for the j-th OTHER edge of that bit
                                 //                  (ascending check index); the zero slot for the absent third edge of a degree-3 bit
    std::vector<int> own_wr;     // [2 * NTMAX]      slot written for j = 2 (a private trash slot for a degree-3 bit)
    std::vector<int> own_dl;     // [2 * NTMAX]      rank of the owner among the bit's checks (0 .. 3): where the register message
                                 //                  enters the bit's sum
    std::vector<int> x_bit;      // [NTMAX]          unowned bit of a thread, -1 = none
    std::vector<int> x_slot;     // [4 * NTMAX]      entry d * NTMAX + t: LDS slot of its d-th edge; the thread's dummy slot where none
    std::vector<int> x_deg;      // [NTMAX / 64]     degree of a wave's unowned group, 0 = none
    int zero_slot = 0, priv0 = 0;  // LDS map (doubles): (DC - 2) * MP message planes | 2 (zero slot) | 3 * NTMAX private slots
    long read_cycles = 0, write_cycles = 0, read_floor = 0, write_floor = 0;  // modelled, one bit pass
};

inline size_t lds_doubles(int DC, int MP, int NTMAX) { return (size_t)(DC - 2) * MP + 2 + (size_t)3 * NTMAX; }

// rp / ci: CSR of the pcm; every check must have DC entries, every bit 3 or 4.  Returns false when the code does not fit.
inline bool build(const std::vector<int>& rp, const std::vector<int>& ci, int m, int n, int DC, int MP, int NTMAX, int iters, Tables& T) {
    if (m > MP || NTMAX % 64 != 0 || n < 2 * m) return false;
    const int E = rp[m];
    std::vector<int> cptr(n + 1, 0), fill(n, 0);
    for (int e = 0; e < E; ++e) cptr[ci[e] + 1]++;
    for (int i = 0; i < n; ++i) cptr[i + 1] += cptr[i];
    std::vector<int> col(E);  // checks of a bit, ascending
    for (int c = 0; c < m; ++c) {
        if (rp[c + 1] - rp[c] != DC) return false;
        for (int e = rp[c]; e < rp[c + 1]; ++e) col[cptr[ci[e]] + fill[ci[e]]++] = c;
    }
    for (int i = 0; i < n; ++i) {
        const int d = cptr[i + 1] - cptr[i];
        if (d < 3 || d > 4) return false;
    }
    // ---- perfect b-matching: owner[i] = check that owns bit i (-1 = none), every check owns exactly two
    std::vector<int> owner(n, -1), nown(m, 0), seen(n, 0);
    int stamp = 0;
    // try to give check c one more bit: DFS over alternating paths (a bit owned by c2 may move to another of c2's neighbours)
    struct Dfs {
        const std::vector<int>&rp, &ci;
        std::vector<int>&owner, &seen;
        int& stamp;
        bool run(int c) {
            for (int e = rp[c]; e < rp[c + 1]; ++e) {
                const int i = ci[e];
                if (seen[i] == stamp || owner[i] == c) continue;
                seen[i] = stamp;
                if (owner[i] < 0) { owner[i] = c; return true; }
                const int c2 = owner[i];
                owner[i] = c;  // tentatively take it; c2 needs a replacement
                if (run(c2)) return true;
                owner[i] = c2;
            }
            return false;
        }
    } dfs{rp, ci, owner, seen, stamp};
    for (int round = 0; round < 2; ++round)
        for (int c = 0; c < m; ++c) {
            ++stamp;
            if (!dfs.run(c)) return false;
            nown[c]++;
        }
    // ---- unowned bits -> degree groups -> waves
    const int NW = NTMAX / 64;
    const int cw = (m + 63) / 64;
    std::vector<std::vector<int>> xcls(5);
    for (int i = 0; i < n; ++i)
        if (owner[i] < 0) xcls[cptr[i + 1] - cptr[i]].push_back(i);
    struct Group { int deg; std::vector<int> bits; };
    std::vector<Group> xg;
    for (int d = 4; d >= 3; --d)
        for (size_t q = 0; q < xcls[d].size(); q += 64) {
            Group g{d, std::vector<int>(64, -1)};
            for (size_t l = 0; l < 64 && q + l < xcls[d].size(); ++l) g.bits[l] = xcls[d][q + l];
            xg.push_back(g);
        }
    if ((int)xg.size() > NW) return false;
    // unowned groups go to waves of their own while the table stride has room, else they share the last check waves
    const int nw = std::min(NW, cw + (int)xg.size());
    std::vector<int> xwave(xg.size());
    for (size_t q = 0; q < xg.size(); ++q) xwave[q] = nw - 1 - (int)q;
    T.NT = nw * 64; T.MP = MP; T.NTMAX = NTMAX; T.DC = DC;
    T.zero_slot = (DC - 2) * MP;
    T.priv0 = (DC - 2) * MP + 2;

    // ---- state of the search: position of every check, the two owned bits of every check in slot order
    std::vector<int> pos_of(m), chk_at(cw * 64, -1), own(2 * (size_t)m, -1);
    for (int c = 0; c < m; ++c) { pos_of[c] = c; chk_at[c] = c; }
    {
        std::vector<int> k(m, 0);
        for (int i = 0; i < n; ++i)
            if (owner[i] >= 0) own[2 * (size_t)owner[i] + k[owner[i]]++] = i;
    }
    // LDS edge number of (check c, bit i): the owned bits are edges 0 and 1 (registers), the others 2 .. DC - 1 in
    // ascending bit order -- the min-sum check update does not depend on the order of a check's edges.  kq[q]: that number
    // for the q-th entry of the column-major list (-1 for an owned edge); it does not change during the search.
    std::vector<int> kq(E, -1);
    for (int c = 0; c < m; ++c) {
        int k = 2;
        for (int e = rp[c]; e < rp[c + 1]; ++e) {
            const int b = ci[e];
            if (owner[b] == c) continue;
            for (int q = cptr[b]; q < cptr[b + 1]; ++q)
                if (col[q] == c) kq[q] = k;
            ++k;
        }
    }
    auto slot_q = [&](int q) { return (kq[q] - 2) * MP + pos_of[col[q]]; };
    // the (up to 64) slots of one wave-level access: owned slot r, other edge j of wave w; unowned group q, edge d
    auto own_access = [&](int w, int r, int j, int* slots, bool wr) {
        for (int l = 0; l < 64; ++l) {
            const int t = w * 64 + l;
            const int c = t < (int)chk_at.size() ? chk_at[t] : -1;
            if (c < 0) { slots[l] = T.priv0 + r * NTMAX + t; continue; }  // padding position: its private toy slot
            const int i = own[2 * (size_t)c + r];
            int jj = 0, s = -1;
            for (int q = cptr[i]; q < cptr[i + 1]; ++q) {
                if (col[q] == c) continue;
                if (jj++ == j) { s = slot_q(q); break; }
            }
            if (s < 0) s = wr ? T.priv0 + r * NTMAX + t : T.zero_slot;  // absent third edge of a degree-3 bit
            slots[l] = s;
        }
    };
    auto x_access = [&](int q, int d, int* slots) {
        for (int l = 0; l < 64; ++l) {
            const int i = xg[q].bits[l];
            slots[l] = i < 0 ? T.priv0 + 2 * NTMAX + xwave[q] * 64 + l : slot_q(cptr[i] + d);
        }
    };
    auto cycles = [&](const int* slots, long& rd, long& wr, long& pairs) {
        int c32[2][32] = {{0}}, c16[4][16] = {{0}}, mx32[2] = {0, 0}, mx16[4] = {0, 0, 0, 0};
        // lanes on the SAME slot are one broadcast, not a conflict (the zero slot of the degree-3 bits)
        bool zseen[2] = {false, false};
        for (int l = 0; l < 64; ++l) {
            const bool dup = slots[l] == T.zero_slot && zseen[l >> 5];
            if (slots[l] == T.zero_slot) zseen[l >> 5] = true;
            if (!dup) {
                int& a = c32[l >> 5][slots[l] & 31];
                pairs += a;
                if (++a > mx32[l >> 5]) mx32[l >> 5] = a;
            }
            int& b = c16[l >> 4][slots[l] & 15];
            pairs += b;
            if (++b > mx16[l >> 4]) mx16[l >> 4] = b;
        }
        rd = mx32[0] + mx32[1];
        wr = std::max(6, mx16[0] + mx16[1] + mx16[2] + mx16[3]);
    };
    auto total = [&](long* rd_out, long* wr_out) {
        long cost = 0, rds = 0, wrs = 0;
        int slots[64];
        for (int w = 0; w < cw; ++w)
            for (int r = 0; r < 2; ++r)
                for (int j = 0; j < 3; ++j) {
                    long rd, wr, pr = 0, dummy = 0;
                    own_access(w, r, j, slots, false);
                    cycles(slots, rd, wr, pr);
                    rds += rd;
                    own_access(w, r, j, slots, true);
                    long rd2;
                    cycles(slots, rd2, wr, dummy);
                    wrs += wr;
                    cost += 64 * (rd + wr) + pr;
                }
        for (size_t q = 0; q < xg.size(); ++q)
            for (int d = 0; d < xg[q].deg; ++d) {
                long rd, wr, pr = 0;
                x_access((int)q, d, slots);
                cycles(slots, rd, wr, pr);
                rds += rd; wrs += wr;
                cost += 64 * (rd + wr) + pr;
            }
        if (rd_out) *rd_out = rds;
        if (wr_out) *wr_out = wrs;
        return cost;
    };
    // ---- simulated annealing: swap the positions of two checks (or move one to a free position), swap a check's slot roles
    unsigned long long rs = 0x9E3779B97F4A7C15ull;
    auto rnd = [&](int mod) {
        rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17;
        return (int)((rs >> 11) % (unsigned long long)mod);
    };
    auto rnd01 = [&]() {
        rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17;
        return (double)(rs >> 11) * (1.0 / 9007199254740992.0);
    };
    long cur = total(nullptr, nullptr), best = cur;
    std::vector<int> b_pos = pos_of, b_at = chk_at, b_own = own;
    const double T0 = 8.0, T1 = 0.5;
    const double cool = iters > 0 ? std::pow(T1 / T0, 1.0 / (double)iters) : 1.0;
    double temp = T0;
    for (int it = 0; it < iters; ++it, temp *= cool) {
        if (rnd(4) == 0) {
            const int c = rnd(m);
            std::swap(own[2 * (size_t)c], own[2 * (size_t)c + 1]);
            const long nc = total(nullptr, nullptr);
            if (nc > cur && rnd01() >= std::exp(-(double)(nc - cur) / temp)) std::swap(own[2 * (size_t)c], own[2 * (size_t)c + 1]);
            else cur = nc;
        } else {
            const int p1 = pos_of[rnd(m)], p2 = rnd(cw * 64);
            if (p1 == p2) continue;
            const int c1 = chk_at[p1], c2 = chk_at[p2];
            chk_at[p1] = c2; chk_at[p2] = c1;
            pos_of[c1] = p2;
            if (c2 >= 0) pos_of[c2] = p1;
            const long nc = total(nullptr, nullptr);
            if (nc > cur && rnd01() >= std::exp(-(double)(nc - cur) / temp)) {
                chk_at[p1] = c1; chk_at[p2] = c2;
                pos_of[c1] = p1;
                if (c2 >= 0) pos_of[c2] = p2;
            } else cur = nc;
        }
        if (cur < best) { best = cur; b_pos = pos_of; b_at = chk_at; b_own = own; }
    }
    pos_of = b_pos; chk_at = b_at; own = b_own;

    // ---- tables
    T.pos_chk.assign(NTMAX, -1);
    for (size_t p = 0; p < chk_at.size(); ++p) T.pos_chk[p] = chk_at[p];
    T.own_bit.assign((size_t)2 * NTMAX, -1);
    T.own_rd.assign((size_t)6 * NTMAX, 0);
    T.own_wr.assign((size_t)2 * NTMAX, 0);
    T.own_dl.assign((size_t)2 * NTMAX, 0);
    int slots[64];
    for (int w = 0; w < NW; ++w)
        for (int r = 0; r < 2; ++r) {
            for (int j = 0; j < 3; ++j) {
                if (w < cw) own_access(w, r, j, slots, false);
                for (int l = 0; l < 64; ++l) T.own_rd[((size_t)3 * r + j) * NTMAX + w * 64 + l] = w < cw ? slots[l] : T.priv0 + r * NTMAX + w * 64 + l;
            }
            if (w < cw) own_access(w, r, 2, slots, true);
            for (int l = 0; l < 64; ++l) {
                const int t = w * 64 + l;
                T.own_wr[(size_t)r * NTMAX + t] = w < cw ? slots[l] : T.priv0 + r * NTMAX + t;
                const int c = t < (int)chk_at.size() ? chk_at[t] : -1;
                if (c < 0) continue;
                const int i = own[2 * (size_t)c + r];
                T.own_bit[(size_t)r * NTMAX + t] = i;
                int dl = 0;
                for (int q = cptr[i]; q < cptr[i + 1] && col[q] != c; ++q) ++dl;
                T.own_dl[(size_t)r * NTMAX + t] = dl;
            }
        }
    T.x_bit.assign(NTMAX, -1);
    T.x_slot.assign((size_t)4 * NTMAX, 0);
    T.x_deg.assign(NW, 0);
    for (int t = 0; t < NTMAX; ++t)
        for (int d = 0; d < 4; ++d) T.x_slot[(size_t)d * NTMAX + t] = T.priv0 + 2 * NTMAX + t;
    for (size_t q = 0; q < xg.size(); ++q) {
        const int w = xwave[q];
        T.x_deg[w] = xg[q].deg;
        for (int l = 0; l < 64; ++l) {
            const int i = xg[q].bits[l];
            if (i < 0) continue;
            T.x_bit[w * 64 + l] = i;
            for (int d = 0; d < xg[q].deg; ++d) T.x_slot[(size_t)d * NTMAX + w * 64 + l] = slot_q(cptr[i] + d);
        }
    }
    total(&T.read_cycles, &T.write_cycles);
    T.read_floor = 2L * (6 * cw);
    T.write_floor = 6L * (6 * cw);
    for (auto& g : xg) { T.read_floor += 2L * g.deg; T.write_floor += 6L * g.deg; }
    return true;
}

}  // namespace own_layout
