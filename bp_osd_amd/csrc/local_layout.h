// local_layout.h -- host-side layout search for the local-edge BP kernel (bp_local_kernel.hip.h): which check owns which
// two of its six bits, and which position (thread slot) every check gets.  Pure C++ (no HIP): included by
// bposd_capi.hip and by tools/layout_probe.cpp, which runs the search on the CPU alone.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <thread>
#include <vector>

// ------------------------------------------------------------------ local-edge BP kernel: tables + launch
// Every check owns two of its six bits (perfect b-matching, Kuhn's augmenting paths with capacity 2); checks are
// grouped into 64-position groups whose slot-b bits share the position dl of the owner among the bit's checks.
namespace local_layout {

// An assignment (every check owns two bits) plus a position for every check.
struct Layout {
    std::vector<int> owner;    // [n]  owning check of a bit
    std::vector<int> load;     // [2m] load[2c + b] = b-th owned bit of check c
    std::vector<int> pos_of;   // [m]  position of a check
    std::vector<int> pos_chk;  // [MP] check at a position, -1 = empty
    int nfull = 0;             // positions [0, nfull) are class-uniform groups that the search must keep uniform
    double cost = 1e30;        // read cycles + write cycles beyond their floor + 5 * mixed (group, slot) pairs (measured exchange rate)
    long long passes = 0;      // modelled ds_read_b64 cycles of the bit pass (ideal: 4 * MP / 32)
    long long wcycles = 0;     // modelled ds_write_b64 cycles of the bit pass (floor: 6 per wave-level store)
    int mixed = 0;             // (group, slot) pairs whose lanes do not share one dl
};

struct Graph {
    int m, n, MP;
    std::vector<int> cols;  // [3n] checks of a bit, ascending
    int rank_of(int i, int c) const { return cols[3 * (size_t)i] == c ? 0 : (cols[3 * (size_t)i + 1] == c ? 1 : 2); }
    // the two non-local checks of bit i owned by c, in cyclic order after the owner (roles X, Y of the kernel)
    void others(int i, int c, int* o) const {
        const int d = rank_of(i, c);
        o[0] = cols[3 * (size_t)i + (d + 1) % 3];
        o[1] = cols[3 * (size_t)i + (d + 2) % 3];
    }
};

// LDS cost of the bit pass under the measured banking rules of gfx950 (tools/microbench/lds_scatter_probe.hip,
// MI355X_MICROARCH.md §LDS).  One wave-level access = the 64 positions of a group reading / writing the message of slot
// bs = 2 * (owned bit) + (role X / Y) at  k * MP + position of the other check:
//   ds_read_b64  is served in two half-waves of 32 lanes on 64 banks: cycles = sum over halves of the largest number of
//                lanes on one 8-byte column (position mod 32);
//   ds_write_b64 is served in four quarter-waves of 16 lanes on 32 banks and moves 3 source dwords at 2 cycles each:
//                cycles = max(6, sum over quarters of the largest number of lanes on one column (position mod 16)).
// A padding position accesses its own four slots (bp_local_kernel.hip.h), i.e. column = its own position.
struct LdsCost {
    int read_cycles = 0, write_cycles = 0, mixed = 0;
};

struct BitPassModel {
    const Graph& g;
    Layout& L;
    std::vector<int> tgt;                 // [4m] other check of (check, bs)
    std::vector<std::vector<int>> users;  // [m]  (4 * check + bs) entries that point at a check
    BitPassModel(const Graph& g_, Layout& L_) : g(g_), L(L_), tgt(4 * (size_t)g_.m), users(g_.m) {
        for (int c = 0; c < g.m; ++c)
            for (int b = 0; b < 2; ++b) {
                int o[2];
                g.others(L.load[2 * c + b], c, o);
                for (int s = 0; s < 2; ++s) {
                    tgt[4 * (size_t)c + 2 * b + s] = o[s];
                    users[o[s]].push_back(4 * c + 2 * b + s);
                }
            }
    }
    int column(int p, int bs) const {
        const int c = L.pos_chk[p];
        return c < 0 ? p : L.pos_of[tgt[4 * (size_t)c + bs]];
    }
    // cycles of the read and the write of (group, bs), and the colliding lane pairs (a smooth tie-breaker for the search)
    void group_cost(int gq, int bs, int& rd, int& wr, int& pairs) const {
        int c32[2][32] = {{0}}, c16[4][16] = {{0}};
        rd = 0; pairs = 0;
        int wsum = 0;
        int mx32[2] = {0, 0}, mx16[4] = {0, 0, 0, 0};
        for (int l = 0; l < 64; ++l) {
            const int col = column(64 * gq + l, bs);
            int& a = c32[l >> 5][col & 31];
            pairs += a;
            if (++a > mx32[l >> 5]) mx32[l >> 5] = a;
            int& w = c16[l >> 4][col & 15];
            pairs += w;
            if (++w > mx16[l >> 4]) mx16[l >> 4] = w;
        }
        rd = mx32[0] + mx32[1];
        wsum = mx16[0] + mx16[1] + mx16[2] + mx16[3];
        wr = wsum < 6 ? 6 : wsum;
    }
    LdsCost total() const {
        LdsCost t;
        for (int gq = 0; gq < g.MP / 64; ++gq)
            for (int bs = 0; bs < 4; ++bs) {
                int rd, wr, pr;
                group_cost(gq, bs, rd, wr, pr);
                t.read_cycles += rd;
                t.write_cycles += wr;
            }
        for (int gq = 0; gq < g.MP / 64; ++gq)
            for (int b = 0; b < 2; ++b) {
                int code = -1;
                for (int p = 64 * gq; p < 64 * gq + 64; ++p) {
                    const int c = L.pos_chk[p];
                    if (c < 0) continue;
                    const int d = g.rank_of(L.load[2 * c + b], c);
                    code = (code < 0 || code == d) ? d : 3;
                }
                t.mixed += code == 3;
            }
        return t;
    }
};

inline LdsCost lds_cost(const Graph& g, Layout& L) { return BitPassModel(g, L).total(); }

// position search: simulated annealing over swaps of two positions (inside the class-uniform region only swaps that keep
// every group uniform), objective = modelled LDS cycles of the bit pass, 64 * cycles + colliding lane pairs
inline void search(const Graph& g, Layout& L, bool constrained, int iters) {
    const int m = g.m, MP = g.MP;
    auto key = [&](int c) { return g.rank_of(L.load[2 * c], c) * 3 + g.rank_of(L.load[2 * c + 1], c); };
    BitPassModel M(g, L);
    const int NG = (MP / 64) * 4;
    std::vector<long long> gcost(NG);  // current cost of every (group, bs)
    auto eval = [&](int e) {
        int rd, wr, pr;
        M.group_cost(e >> 2, e & 3, rd, wr, pr);
        return 64LL * (rd + wr) + pr;
    };
    for (int e = 0; e < NG; ++e) gcost[e] = eval(e);
    std::vector<int> touched;
    std::vector<long long> fresh;
    auto collect = [&](int c) {
        for (int u : M.users[c]) touched.push_back((L.pos_of[u >> 2] >> 6) * 4 + (u & 3));
    };
    unsigned long long rs = 0x9E3779B97F4A7C15ull;
    auto rnd = [&](int mod) {
        rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17;
        return (int)((rs >> 11) % (unsigned long long)mod);
    };
    auto rnd01 = [&]() {
        rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17;
        return (double)(rs >> 11) * (1.0 / 9007199254740992.0);
    };
    // geometric cooling (a cycle = 64 cost units: from "a cycle uphill now and then" to "colliding pairs only"); the best
    // layout seen is kept
    double T0 = 8.0, T1 = 0.5;
    if (const char* e = getenv("BPOSD_LAYOUT_T0")) T0 = atof(e);
    if (const char* e = getenv("BPOSD_LAYOUT_T1")) T1 = atof(e);
    std::vector<int> best_pos_of = L.pos_of, best_pos_chk = L.pos_chk;
    long long cur = 0, best_cost = 0;
    const double cool = iters > 0 ? std::pow(T1 / T0, 1.0 / (double)iters) : 1.0;
    double T = T0;
    for (int it = 0; it < iters; ++it, T *= cool) {
        const int c1 = rnd(m);
        const int p1 = L.pos_of[c1];
        int p2;
        if (!constrained) p2 = rnd(MP);
        else if (p1 < L.nfull) {
            if (rnd(4) != 0) p2 = (p1 & ~63) + rnd(64);  // same group
            else {
                p2 = rnd(L.nfull);                         // another uniform position: must be the same class
                if (L.pos_chk[p2] < 0 || key(L.pos_chk[p2]) != key(c1)) continue;
            }
        } else p2 = L.nfull + rnd(MP - L.nfull);            // mixed region (with its empty positions)
        const int c2 = L.pos_chk[p2];
        if (c2 == c1) continue;
        touched.clear();
        collect(c1);
        if (c2 >= 0) collect(c2);
        for (int bs = 0; bs < 4; ++bs) { touched.push_back((p1 >> 6) * 4 + bs); touched.push_back((p2 >> 6) * 4 + bs); }
        std::sort(touched.begin(), touched.end());
        touched.erase(std::unique(touched.begin(), touched.end()), touched.end());
        L.pos_of[c1] = p2; L.pos_chk[p2] = c1; L.pos_chk[p1] = c2;
        if (c2 >= 0) L.pos_of[c2] = p1;
        long long d = 0;
        fresh.clear();
        for (int e : touched) {
            fresh.push_back(eval(e));
            d += fresh.back() - gcost[e];
        }
        if (d > 0 && rnd01() >= std::exp(-(double)d / T)) {
            L.pos_of[c1] = p1; L.pos_chk[p1] = c1; L.pos_chk[p2] = c2;
            if (c2 >= 0) L.pos_of[c2] = p2;
            continue;
        }
        for (size_t q = 0; q < touched.size(); ++q) gcost[touched[q]] = fresh[q];
        cur += d;
        if (cur < best_cost) { best_cost = cur; best_pos_of = L.pos_of; best_pos_chk = L.pos_chk; }
    }
    L.pos_of = best_pos_of;
    L.pos_chk = best_pos_chk;
    const LdsCost t = M.total();
    L.passes = t.read_cycles;
    L.wcycles = t.write_cycles;
    L.mixed = t.mixed;
    L.cost = (double)(t.read_cycles + t.write_cycles - 6 * 4 * (MP / 64)) + 5.0 * t.mixed;
}

// class-sorted start layout: full groups of one class first (uniform), the leftovers behind them
void class_sorted(const Graph& g, Layout& L) {
    const int m = g.m;
    std::vector<std::vector<int>> cls(9);
    for (int c = 0; c < m; ++c) {
        if (g.rank_of(L.load[2 * c], c) > g.rank_of(L.load[2 * c + 1], c)) std::swap(L.load[2 * c], L.load[2 * c + 1]);
        cls[g.rank_of(L.load[2 * c], c) * 3 + g.rank_of(L.load[2 * c + 1], c)].push_back(c);
    }
    // full groups of one class first; then, while the 64-position groups suffice, the largest leftover classes get a
    // partly filled group of their own (still uniform, and its empty positions give the search slack); what remains
    // shares the mixed group(s) at the end
    L.pos_chk.assign(g.MP, -1);
    L.pos_of.assign(m, -1);
    int p = 0;
    std::vector<std::vector<int>> left(9);
    for (int k = 0; k < 9; ++k) {
        const size_t full = cls[k].size() / 64 * 64;
        for (size_t q = 0; q < full; ++q) { L.pos_chk[p] = cls[k][q]; L.pos_of[cls[k][q]] = p; ++p; }
        left[k].assign(cls[k].begin() + full, cls[k].end());
    }
    const int G = g.MP / 64;
    for (;;) {
        int nleft = 0, big = -1;
        for (int k = 0; k < 9; ++k) {
            nleft += (int)left[k].size();
            if (!left[k].empty() && (big < 0 || left[k].size() > left[big].size())) big = k;
        }
        const int used = p / 64;
        // peel the largest leftover class off only if the others still fit behind it
        if (big < 0 || nleft <= 64 || used + 1 + (nleft - (int)left[big].size() + 63) / 64 > G) break;
        for (int c : left[big]) { L.pos_chk[p] = c; L.pos_of[c] = p; ++p; }
        left[big].clear();
        p = (p + 63) / 64 * 64;
    }
    L.nfull = p;
    for (int k = 0; k < 9; ++k)
        for (int c : left[k]) { L.pos_chk[p] = c; L.pos_of[c] = p; ++p; }
}


// Class rebalancing.  A check's class is the pair (rank of the check among its first owned bit's three checks, the same
// for the second), sorted: six classes.  A 64-position group runs select-free code only if all its checks share one
// class, so the number of groups a layout needs is  sum over classes of ceil(size / 64);  whatever does not fit ends up
// in "mixed" groups, and the waves of mixed groups are the ones every barrier waits for.  The assignment (which check
// owns which two of its six bits) has plenty of freedom: this search moves bits between checks along short alternating
// cycles (bit i: c -> c', a bit of c' moves on, ... until a bit arrives at c) and keeps a move when the group count does
// not grow, until the classes pack into the available groups.
inline int groups_needed(const int cnt[9]) {
    int g = 0;
    for (int k = 0; k < 9; ++k) g += (cnt[k] + 63) / 64;
    return g;
}

inline bool rebalance_classes(const Graph& g, Layout& L, int G, int max_moves, bool sideways) {
    const int m = g.m;
    auto key = [&](int c) {
        int a = g.rank_of(L.load[2 * c], c), b = g.rank_of(L.load[2 * c + 1], c);
        if (a > b) std::swap(a, b);
        return a * 3 + b;
    };
    int cnt[9] = {0};
    for (int c = 0; c < m; ++c) cnt[key(c)]++;
    int F = groups_needed(cnt);
    unsigned long long rs = 0x2545F4914F6CDD1Dull;
    auto rnd = [&](int mod) {
        rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17;
        return (int)((rs >> 11) % (unsigned long long)mod);
    };
    auto replace_bit = [&](int c, int oldb, int newb) {
        if (L.load[2 * c] == oldb) L.load[2 * c] = newb; else L.load[2 * c + 1] = newb;
    };
    // secondary objective (ties of F): fewer checks outside the two largest classes -> keeps the search moving
    auto slack = [&]() {
        int s2 = 0;
        for (int k = 0; k < 9; ++k) s2 += cnt[k] % 64 == 0 ? 0 : 64 - cnt[k] % 64;
        return s2;
    };
    int S = slack();
    std::vector<int> path_chk, path_bit;
    for (int mv = 0; mv < max_moves && F > G; ++mv) {
        // alternating cycle c0 -b0-> c1 -b1-> ... -> c0 : bit b_t leaves c_t for c_{t+1}
        const int c0 = rnd(m);
        path_chk.assign(1, c0);
        path_bit.clear();
        bool closed = false;
        int cur = c0;
        for (int depth = 0; depth < 6 && !closed; ++depth) {
            const int b = L.load[2 * cur + rnd(2)];
            if (std::find(path_bit.begin(), path_bit.end(), b) != path_bit.end()) break;
            // b goes to one of its other checks; prefer closing the cycle
            int o[2];
            g.others(b, cur, o);
            int nxt = -1;
            if (depth > 0 && (o[0] == c0 || o[1] == c0)) nxt = c0;
            else nxt = o[rnd(2)];
            if (nxt != c0 && std::find(path_chk.begin(), path_chk.end(), nxt) != path_chk.end()) break;
            path_bit.push_back(b);
            if (nxt == c0) closed = true;
            else { path_chk.push_back(nxt); cur = nxt; }
        }
        if (!closed || path_bit.size() < 2) continue;
        const size_t len = path_bit.size();  // checks path_chk[0..len-1], bit t moves path_chk[t] -> path_chk[(t+1) % len]
        int before[8], after_[8];
        for (size_t t = 0; t < len; ++t) before[t] = key(path_chk[t]);
        for (size_t t = 0; t < len; ++t) {
            const int from = path_chk[t], to = path_chk[(t + 1) % len];
            // `to` receives path_bit[t] in place of the bit it gives away (path_bit[(t+1) % len])
            replace_bit(to, path_bit[(t + 1) % len], -2 - (int)t);  // placeholder keeps slots distinct
            (void)from;
        }
        for (size_t t = 0; t < len; ++t) {
            const int to = path_chk[(t + 1) % len];
            replace_bit(to, -2 - (int)t, path_bit[t]);
            L.owner[path_bit[t]] = to;
        }
        for (size_t t = 0; t < len; ++t) { after_[t] = key(path_chk[t]); cnt[before[t]]--; }
        for (size_t t = 0; t < len; ++t) cnt[after_[t]]++;
        const int F2 = groups_needed(cnt), S2 = slack();
        if (F2 < F || (F2 == F && (S2 < S || (sideways && (S2 == S || rnd(8) == 0))))) { F = F2; S = S2; continue; }
        // undo
        for (size_t t = 0; t < len; ++t) { cnt[after_[t]]--; }
        for (size_t t = 0; t < len; ++t) { cnt[before[t]]++; }
        for (size_t t = 0; t < len; ++t) {
            const int to = path_chk[(t + 1) % len];
            replace_bit(to, path_bit[t], -2 - (int)t);
        }
        for (size_t t = 0; t < len; ++t) {
            const int to = path_chk[(t + 1) % len];
            replace_bit(to, -2 - (int)t, path_bit[(t + 1) % len]);
            L.owner[path_bit[(t + 1) % len]] = to;
        }
    }
    if (getenv("BPOSD_DEBUG_OCC")) {
        fprintf(stderr, "[bposd] class rebalancing: groups needed %d (available %d), classes", F, G);
        for (int k = 0; k < 9; ++k) fprintf(stderr, " %d", cnt[k]);
        fprintf(stderr, "\n");
    }
    return F <= G;
}

}  // namespace local_layout

// Every check owns two of its six bits; positions are chosen so that (a) as many 64-position groups as possible
// share the owner's rank per slot (select-free code) and (b) the bit pass's LDS accesses collide as little as
// possible.  Candidates: Kuhn's augmenting paths and, for two-block codes (hypergraph products), the nine "a bit
// prefers its rank-d check" block matchings; each is class-sorted (full groups of one class first), the cheapest start
// layout (simulated passes + 5 * mixed (group, slot) pairs) is then searched under the uniformity constraint.
// (A row layout of the circulant grid is conflict-free -- 128 passes for 128 accesses -- but the wrap-around makes
// every group mixed, and measured on the GPU a mixed pair costs as much as five extra passes: 36.3 ms against 29.8.)
// Host-only part: ownership assignment + positions for a (3,6)-regular code with n = 2m (rp / ci: CSR of the pcm).
// Returns false when no perfect assignment exists.
inline bool local_layout_host(const std::vector<int>& rp, const std::vector<int>& ci, int m, int n, int MP,
                              local_layout::Graph& g, local_layout::Layout& best) {
    using namespace local_layout;
    g.m = m; g.n = n; g.MP = MP;
    g.cols.assign(3 * (size_t)n, 0);
    {
        std::vector<int> fill(n, 0);
        for (int c = 0; c < m; ++c)
            for (int e = rp[c]; e < rp[c + 1]; ++e) {
                const int i = ci[e];
                g.cols[3 * (size_t)i + fill[i]++] = c;
            }
    }
    const std::vector<int>& cols = g.cols;
    std::vector<Layout> cands;
    // ---- (1) generic perfect assignment
    {
        Layout L;
        L.owner.assign(n, -1);
        L.load.assign(2 * (size_t)m, -1);
        auto cnt = [&](int c) { return (L.load[2 * c] >= 0) + (L.load[2 * c + 1] >= 0); };
        auto put = [&](int c, int i) { (L.load[2 * c] < 0 ? L.load[2 * c] : L.load[2 * c + 1]) = i; L.owner[i] = c; };
        auto drop = [&](int c, int i) { if (L.load[2 * c] == i) L.load[2 * c] = -1; else L.load[2 * c + 1] = -1; };
        std::vector<int> seen(m, -1);
        std::function<bool(int, int)> place = [&](int i, int stamp) -> bool {
            for (int d = 0; d < 3; ++d) {
                const int c = cols[3 * (size_t)i + d];
                if (seen[c] == stamp) continue;
                seen[c] = stamp;
                if (cnt(c) < 2) { put(c, i); return true; }
                for (int s2 = 0; s2 < 2; ++s2) {
                    const int j = L.load[2 * c + s2];
                    drop(c, j);
                    L.owner[j] = -1;
                    if (place(j, stamp)) { put(c, i); return true; }
                    L.load[2 * c + s2] = j;
                    L.owner[j] = c;
                }
            }
            return false;
        };
        for (int i = 0; i < n; ++i)
            if (!place(i, i)) return false;  // no perfect assignment: the LDS kernel is used
        for (int c = 0; c < m; ++c)
            if (cnt(c) != 2) return false;
        cands.push_back(L);
    }
    bool two_block = true;
    for (int c = 0; c < m && two_block; ++c) {
        int lo = 0;
        for (int e = rp[c]; e < rp[c + 1]; ++e) lo += ci[e] < n / 2;
        two_block = (lo == 3);
    }
    // ---- (2) rank-preference block matchings
    if (two_block) {
        std::vector<int> own2(n), slot(2 * (size_t)m), seen2(m);
        for (int p1 = 0; p1 < 3; ++p1)
            for (int p2 = 0; p2 < 3; ++p2) {
                std::fill(own2.begin(), own2.end(), -1);
                std::fill(slot.begin(), slot.end(), -1);
                std::fill(seen2.begin(), seen2.end(), -1);
                bool pref_only = true;
                std::function<bool(int, int, int)> aug = [&](int i, int stamp, int blk) -> bool {
                    const int pr = blk == 0 ? p1 : p2;
                    for (int dd = 0; dd < (pref_only ? 1 : 3); ++dd) {
                        const int c = cols[3 * (size_t)i + (pr + dd) % 3];
                        if (seen2[c] == stamp) continue;
                        seen2[c] = stamp;
                        const int j = slot[2 * c + blk];
                        if (j < 0 || aug(j, stamp, blk)) { slot[2 * c + blk] = i; own2[i] = c; return true; }
                    }
                    return false;
                };
                int stamp = 0;
                for (int i = 0; i < n; ++i) (void)aug(i, ++stamp, i < n / 2 ? 0 : 1);
                pref_only = false;
                bool ok = true;
                for (int i = 0; i < n && ok; ++i)
                    if (own2[i] < 0) ok = aug(i, ++stamp, i < n / 2 ? 0 : 1);
                for (int c = 0; c < m && ok; ++c) ok = slot[2 * c] >= 0 && slot[2 * c + 1] >= 0;
                if (!ok) continue;
                Layout L;
                L.owner = own2;
                L.load = slot;
                cands.push_back(L);
            }
    }
    // every candidate also in a class-rebalanced version (all groups uniform, if the classes can be made to pack)
    {
        const size_t n0 = cands.size();
        for (size_t k = 0; k < n0; ++k)
            for (int sideways = 0; sideways < 2; ++sideways) {  // strictly improving moves first: they disturb the structure least
                Layout R = cands[k];
                if (rebalance_classes(g, R, MP / 64, 100000, sideways != 0)) { cands.push_back(R); break; }
            }
    }
    // class-sorted start for every candidate; rank them by the cost of the start layout -- simulated passes + MIXW per
    // mixed (group, slot) pair -- and finish the best three with the full position search
    {
        double mixw = 5.0;  // exchange rate passes <-> mixed pairs (measured, DESIGN.md "BP kernels"); BPOSD_MIXW overrides
        if (const char* e = getenv("BPOSD_MIXW")) mixw = atof(e);
        auto total = [&](const Layout& L) { return L.cost - 5.0 * L.mixed + mixw * L.mixed; };
        std::vector<std::pair<double, size_t>> rank;
        for (size_t k = 0; k < cands.size(); ++k) {
            class_sorted(g, cands[k]);
            search(g, cands[k], true, 0);  // cost of the start layout
            rank.push_back({total(cands[k]), k});
        }
        std::sort(rank.begin(), rank.end());
        const size_t nfinish = std::min(rank.size(), getenv("BPOSD_LAYOUT_ALL") ? rank.size() : (size_t)3);
        // (0.4 M steps: 218 + 408 modelled cycles in 1.1 s; 1 M: 210 + 401 in 1.3 s; 2 M: 208 + 402 in 2.6 s -- and no measurable difference
        // in launch time between the three, same-box A/B of round 4: 26.1 +- 0.15 ms each)
        const int iters = getenv("BPOSD_LAYOUT_ITERS") ? atoi(getenv("BPOSD_LAYOUT_ITERS")) : 400000;
        std::vector<Layout> fin(nfinish);
        std::vector<std::thread> th;
        for (size_t q = 0; q < nfinish; ++q) {  // the searches are independent: one host thread each
            fin[q] = cands[rank[q].second];
            th.emplace_back([&g, &fin, q, iters]() { search(g, fin[q], true, iters); });
        }
        for (auto& t : th) t.join();
        for (size_t q = 0; q < nfinish; ++q) {
            if (getenv("BPOSD_DEBUG_OCC"))
                fprintf(stderr, "[bposd] candidate %zu: %lld read + %lld write cycles, %d mixed pairs\n", rank[q].second, fin[q].passes,
                        fin[q].wcycles, fin[q].mixed);
            if (q == 0 || total(fin[q]) < total(best)) best = fin[q];
        }
    }
    return true;
}

