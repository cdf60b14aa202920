// class_layout.h -- host-side tables for bp_class_kernel.hip.h: bits sorted by degree into 64-lane groups, groups dealt
// to (wave, slot) pairs so that the waves of a workgroup carry similar work, and the lane order inside the groups
// searched for few LDS bank conflicts under the measured banking rules of gfx950 (tools/microbench/lds_scatter_probe.hip):
//   ds_read_b64  = sum over the two half-waves of the largest number of lanes on one 8-byte column (slot mod 32)
//   ds_write_b64 = max(6, sum over the four quarter-waves of the largest number of lanes on one column (slot mod 16))
// Pure C++ (no HIP): included by bposd_capi.hip and by tools/layout_probe.cpp.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <vector>

namespace class_layout {

struct Tables {
    int NT = 0;                  // threads per workgroup (multiple of 64)
    int VPT = 0, DVHI = 0, NTMAX = 0, MP = 0;
    std::vector<int> pos_chk;    // [NTMAX]  check at a position (thread), -1 = none
    std::vector<int> pos_bit;    // [VPT * NTMAX]
    std::vector<int> bit_slot;   // [DVHI * VPT * NTMAX]
    std::vector<int> grp_deg;    // [VPT * NTMAX / 64]
    std::vector<int> grp_cdeg;   // [NTMAX / 64]  check degree of a wave's positions, 0 = the wave holds no checks
    long read_cycles = 0, write_cycles = 0;  // modelled LDS cycles of one bit pass
    long read_floor = 0, write_floor = 0;
};

// rp / ci: CSR of the pcm; DCLO .. DC: check degrees, DVLO .. DVHI: bit degrees the kernel instance handles.  Checks are
// sorted by degree into waves (a wave's 64 positions hold checks of ONE degree), bits by degree into 64-lane groups.
// Returns false when the code does not fit (a degree outside the ranges, too many groups for VPT slots of NTMAX threads).
inline bool build(const std::vector<int>& rp, const std::vector<int>& ci, int m, int n, int DCLO, int DC, int DVLO, int DVHI, int VPT,
                  int MP, int NTMAX, int iters, Tables& T) {
    if (m > MP || NTMAX % 64 != 0) return false;
    // CSC view with the LDS slot (k * MP + c) of every edge, ascending check index inside a column
    std::vector<int> cptr(n + 1, 0), fill(n, 0);
    const int E = rp[m];
    for (int e = 0; e < E; ++e) cptr[ci[e] + 1]++;
    for (int i = 0; i < n; ++i) cptr[i + 1] += cptr[i];
    std::vector<int> ek(E), ec(E);  // per CSC entry: edge number k inside its check, and the check
    for (int c = 0; c < m; ++c) {
        if (rp[c + 1] - rp[c] < DCLO || rp[c + 1] - rp[c] > DC) return false;
        for (int e = rp[c]; e < rp[c + 1]; ++e) {
            const int i = ci[e];
            ek[cptr[i] + fill[i]] = e - rp[c];
            ec[cptr[i] + fill[i]++] = c;
        }
    }
    // checks may sit at any position of the waves that hold checks (the check pass is linear in the position whatever
    // the order; the syndrome bit and the mismatch bitmap go through pos_chk): LDS slot of an edge = k * MP + position
    // -- inside the waves of the check's degree class (heaviest class first)
    std::vector<int> reg_lo(DC + 2, 0), reg_hi(DC + 2, 0), wave_cdeg;
    std::vector<int> pos_of(m);
    {
        int at = 0;
        for (int d = DC; d >= DCLO; --d) {
            reg_lo[d] = at;
            for (int c = 0; c < m; ++c)
                if (rp[c + 1] - rp[c] == d) pos_of[c] = at++;
            at = (at + 63) & ~63;
            reg_hi[d] = at;
            for (int w = reg_lo[d] / 64; w < reg_hi[d] / 64; ++w) wave_cdeg.push_back(d);
        }
    }
    const int cw0 = (int)wave_cdeg.size();
    if (cw0 * 64 > MP) return false;
    std::vector<int> chk_at(cw0 * 64, -1);
    for (int c = 0; c < m; ++c) chk_at[pos_of[c]] = c;
    auto eslot_of = [&](int e) { return ek[e] * MP + pos_of[ec[e]]; };
    // degree classes -> groups of 64 lanes
    std::vector<std::vector<int>> cls(DVHI + 1);
    for (int i = 0; i < n; ++i) {
        const int d = cptr[i + 1] - cptr[i];
        if (d < DVLO || d > DVHI) return false;
        cls[d].push_back(i);
    }
    struct Group { int deg; std::vector<int> bits; };  // bits: 64 entries, -1 = padding
    std::vector<Group> groups;
    for (int d = DVHI; d >= DVLO; --d)
        for (size_t q = 0; q < cls[d].size(); q += 64) {
            Group g{d, std::vector<int>(64, -1)};
            for (size_t l = 0; l < 64 && q + l < cls[d].size(); ++l) g.bits[l] = cls[d][q + l];
            groups.push_back(g);
        }
    const int cw = cw0;  // waves that hold checks
    int nw = std::max(cw, ((int)groups.size() + VPT - 1) / VPT);
    if (nw * 64 > NTMAX) return false;
    // deal the groups to waves: heaviest first, to the wave with the least work that still has a free slot
    // (work in LDS cycles: a check wave 8 * DC, a bit group 8 * degree)
    std::vector<long> load(nw, 0);
    std::vector<int> used(nw, 0);
    for (int w = 0; w < cw; ++w) load[w] = 8L * wave_cdeg[w];
    std::vector<int> g_wave(groups.size()), g_slot(groups.size());
    for (size_t q = 0; q < groups.size(); ++q) {
        int best = -1;
        for (int w = 0; w < nw; ++w)
            if (used[w] < VPT && (best < 0 || load[w] < load[best])) best = w;
        g_wave[q] = best;
        g_slot[q] = used[best]++;
        load[best] += 8L * groups[q].deg;
    }
    T.NT = nw * 64; T.VPT = VPT; T.DVHI = DVHI; T.NTMAX = NTMAX; T.MP = MP;

    // ---- lane order: simulated annealing over swaps of two lanes of groups of the same degree
    auto column = [&](const Group& g, int q, int l, int d) {
        const int i = g.bits[l];
        if (i < 0) return (g_wave[q] * 64 + l);  // dummy slot DC * MP + tid: column = tid
        return pos_of[ec[cptr[i] + d]];  // (k * MP does not change the column: MP is a multiple of 32)
    };
    auto group_cost = [&](int q, long& rd, long& wr, long& pairs) {
        const Group& g = groups[q];
        rd = wr = pairs = 0;
        for (int d = 0; d < g.deg; ++d) {
            int c32[2][32] = {{0}}, c16[4][16] = {{0}}, mx32[2] = {0, 0}, mx16[4] = {0, 0, 0, 0};
            for (int l = 0; l < 64; ++l) {
                const int col = column(g, q, l, d);
                int& a = c32[l >> 5][col & 31];
                pairs += a;
                if (++a > mx32[l >> 5]) mx32[l >> 5] = a;
                int& b = c16[l >> 4][col & 15];
                pairs += b;
                if (++b > mx16[l >> 4]) mx16[l >> 4] = b;
            }
            rd += mx32[0] + mx32[1];
            wr += std::max(6, mx16[0] + mx16[1] + mx16[2] + mx16[3]);
        }
    };
    auto eval = [&](int q) {
        long rd, wr, pr;
        group_cost(q, rd, wr, pr);
        return 64 * (rd + wr) + pr;
    };
    std::vector<long> gcost(groups.size());
    for (size_t q = 0; q < groups.size(); ++q) gcost[q] = eval((int)q);
    std::vector<std::vector<int>> by_deg(DVHI + 1);
    for (size_t q = 0; q < groups.size(); ++q) by_deg[groups[q].deg].push_back((int)q);
    unsigned long long rs = 0x9E3779B97F4A7C15ull;
    auto rnd = [&](int mod) {
        rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17;
        return (int)((rs >> 11) % (unsigned long long)mod);
    };
    auto rnd01 = [&]() {
        rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17;
        return (double)(rs >> 11) * (1.0 / 9007199254740992.0);
    };
    std::vector<Group> best_groups = groups;
    std::vector<int> best_pos_of = pos_of, best_chk_at = chk_at;
    std::vector<int> grp_of_bit(n, -1);
    for (size_t q = 0; q < groups.size(); ++q)
        for (int l = 0; l < 64; ++l)
            if (groups[q].bits[l] >= 0) grp_of_bit[groups[q].bits[l]] = (int)q;
    std::vector<int> touched;
    std::vector<long> fresh;
    long cur = 0, best_cost = 0;
    double T0 = 8.0, T1 = 0.5;
    if (const char* e = getenv("BPOSD_LAYOUT_T0")) T0 = atof(e);
    if (const char* e = getenv("BPOSD_LAYOUT_T1")) T1 = atof(e);
    const double cool = iters > 0 ? std::pow(T1 / T0, 1.0 / (double)iters) : 1.0;
    double temp = T0;
    for (int it = 0; it < iters && !groups.empty(); ++it, temp *= cool) {
        if (rnd(3) == 0) {  // swap the positions of two checks (or move one to an empty position)
            const int p1 = pos_of[rnd(m)];
            const int dcl = wave_cdeg[p1 >> 6], p2 = reg_lo[dcl] + rnd(reg_hi[dcl] - reg_lo[dcl]);  // stays in its degree class
            if (p1 == p2) continue;
            const int c1 = chk_at[p1], c2 = chk_at[p2];
            touched.clear();
            for (int c : {c1, c2})
                if (c >= 0)
                    for (int e = rp[c]; e < rp[c + 1]; ++e) touched.push_back(grp_of_bit[ci[e]]);
            std::sort(touched.begin(), touched.end());
            touched.erase(std::unique(touched.begin(), touched.end()), touched.end());
            chk_at[p1] = c2; chk_at[p2] = c1;
            pos_of[c1] = p2;
            if (c2 >= 0) pos_of[c2] = p1;
            long d = 0;
            fresh.clear();
            for (int q : touched) { fresh.push_back(eval(q)); d += fresh.back() - gcost[q]; }
            if (d > 0 && rnd01() >= std::exp(-(double)d / temp)) {
                chk_at[p1] = c1; chk_at[p2] = c2;
                pos_of[c1] = p1;
                if (c2 >= 0) pos_of[c2] = p2;
                continue;
            }
            for (size_t k = 0; k < touched.size(); ++k) gcost[touched[k]] = fresh[k];
            cur += d;
            if (cur < best_cost) { best_cost = cur; best_groups = groups; best_pos_of = pos_of; best_chk_at = chk_at; }
            continue;
        }
        const int q1 = rnd((int)groups.size());
        const std::vector<int>& peers = by_deg[groups[q1].deg];
        const int q2 = rnd(4) != 0 ? q1 : peers[rnd((int)peers.size())];
        const int l1 = rnd(64), l2 = rnd(64);
        if (q1 == q2 && l1 == l2) continue;
        if (groups[q1].bits[l1] < 0 && groups[q2].bits[l2] < 0) continue;
        std::swap(groups[q1].bits[l1], groups[q2].bits[l2]);
        const long c1 = eval(q1), c2 = q2 != q1 ? eval(q2) : 0;
        const long d = c1 - gcost[q1] + (q2 != q1 ? c2 - gcost[q2] : 0);
        if (d > 0 && rnd01() >= std::exp(-(double)d / temp)) {
            std::swap(groups[q1].bits[l1], groups[q2].bits[l2]);
            continue;
        }
        gcost[q1] = c1;
        if (q2 != q1) gcost[q2] = c2;
        if (groups[q1].bits[l1] >= 0) grp_of_bit[groups[q1].bits[l1]] = q1;
        if (groups[q2].bits[l2] >= 0) grp_of_bit[groups[q2].bits[l2]] = q2;
        cur += d;
        if (cur < best_cost) { best_cost = cur; best_groups = groups; best_pos_of = pos_of; best_chk_at = chk_at; }
    }
    groups = best_groups;
    pos_of = best_pos_of;
    chk_at = best_chk_at;

    // ---- tables
    const int NW = NTMAX / 64;
    T.pos_chk.assign(NTMAX, -1);
    for (size_t p = 0; p < chk_at.size(); ++p) T.pos_chk[p] = chk_at[p];
    T.pos_bit.assign((size_t)VPT * NTMAX, -1);
    T.bit_slot.assign((size_t)DVHI * VPT * NTMAX, 0);
    T.grp_deg.assign((size_t)VPT * NW, 0);
    T.grp_cdeg.assign((size_t)NW, 0);
    for (int w = 0; w < cw; ++w) T.grp_cdeg[w] = wave_cdeg[w];
    for (int r = 0; r < VPT; ++r)
        for (int t = 0; t < NTMAX; ++t)
            for (int d = 0; d < DVHI; ++d) T.bit_slot[((size_t)d * VPT + r) * NTMAX + t] = DC * MP + t;  // the thread's dummy slot
    T.read_cycles = T.write_cycles = T.read_floor = T.write_floor = 0;
    for (size_t q = 0; q < groups.size(); ++q) {
        const int w = g_wave[q], r = g_slot[q];
        T.grp_deg[(size_t)r * NW + w] = groups[q].deg;
        for (int l = 0; l < 64; ++l) {
            const int i = groups[q].bits[l];
            if (i < 0) continue;
            const int t = w * 64 + l;
            T.pos_bit[(size_t)r * NTMAX + t] = i;
            for (int d = 0; d < groups[q].deg; ++d) T.bit_slot[((size_t)d * VPT + r) * NTMAX + t] = eslot_of(cptr[i] + d);
        }
        long rd, wr, pr;
        group_cost((int)q, rd, wr, pr);
        T.read_cycles += rd;
        T.write_cycles += wr;
        T.read_floor += 2L * groups[q].deg;
        T.write_floor += 6L * groups[q].deg;
    }
    return true;
}

}  // namespace class_layout
