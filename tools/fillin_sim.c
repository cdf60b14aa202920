// fillin_sim.c -- CPU model of the fill-in of the HBM-resident OSD elimination (osd_large_kernel, Gaussian mode) under
// different choices of the pivot ROW.  The column order is prescribed by the reliability sort; which
// of a column's candidate rows becomes the pivot row changes no output, only how fast the matrix fills in.  Test / analysis
// tool: nothing in the product path uses it.
//
//   fillin_sim <matrix.bin> <policy> [pure=1]     (matrix.bin: int32 m, n, then for each row int32 count + sorted column positions)
//
// Policies: 0 lowest row index, 1 exact remaining weight, 2 additive estimate (est[r] += est[p], saturating),
//           3 number of absorbed pivots, 4 popcount of the panel word only, 5 estimate, ties by panel-word popcount,
//           6 what the kernel keeps: sum over panels of popcount(combination mask of the panel), ties by row index
// Prints per-run totals: row additions, changed row-words (non-zero words of the pivot row to the right of the panel, per
// addition), listed rows per panel (sum / max), non-zero words of the pivot rows at the moment they are chosen.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 2;
    const int policy = atoi(argv[2]);
    const int pure = argc > 3 ? atoi(argv[3]) : 1;  // 1: plain Gaussian elimination (a pivot row is final when chosen, round 5); 0: rounds 2-4 (pivot rows of the open groups keep absorbing)
    int32_t m, n;
    if (fread(&m, 4, 1, f) != 1 || fread(&n, 4, 1, f) != 1) return 2;
    const int W = (n + 63) / 64;
    uint64_t* M = calloc((size_t)m * W, 8);
    for (int r = 0; r < m; ++r) {
        int32_t c;
        if (fread(&c, 4, 1, f) != 1) return 2;
        for (int i = 0; i < c; ++i) {
            int32_t j;
            if (fread(&j, 4, 1, f) != 1) return 2;
            M[(size_t)r * W + (j >> 6)] ^= 1ull << (j & 63);
        }
    }
    fclose(f);
    uint8_t* used = calloc(m, 1);
    uint32_t* est = malloc(4 * (size_t)m);
    uint32_t* nab = calloc(m, 4);
    uint32_t* cms = calloc(m, 4);   // policy 6: popcounts of the closed panels' masks
    uint64_t* pmk = calloc(m, 8);   // combination mask within the current panel
    for (int r = 0; r < m; ++r) est[r] = 1;
    int* cand = malloc(4 * (size_t)m);
    long long adds = 0, words = 0, listed = 0, pivwords = 0, listmax = 0;
    long long adds_q[4] = {0, 0, 0, 0}, words_q[4] = {0, 0, 0, 0};
    int rank = 0;
    // the kernel's lazy groups: a pivot row stays ACTIVE (takes row additions, is no candidate) until the apply pass that
    // closes its group (every 4th group with pivots); masks of the 4 open groups per row, as the apply pass sees them
    uint64_t* gm = calloc((size_t)m * 4, 8);
    int ngo = 0;
    long long hist_row[8] = {0}, hist_grp[8] = {0}, nact_tot = 0, bits_tot = 0, passes = 0, wavemax_tot = 0, waves_tot = 0;
    for (int w = 0; w < W && rank < m; ++w) {
        // rows with a non-zero panel word (unused ones): the list of E2c
        long long nl = 0;
        for (int r = 0; r < m; ++r)
            if (!used[r] && M[(size_t)r * W + w]) ++nl;
        listed += nl;
        int q = 0;
        if (nl > listmax) listmax = nl;
        for (int b = 0; b < 64 && w * 64 + b < n && rank < m; ++b) {
            int nc = 0;
            for (int r = 0; r < m; ++r)
                if (used[r] != 1 && ((M[(size_t)r * W + w] >> b) & 1ull)) cand[nc++] = r;
            { int any = 0; for (int i = 0; i < nc; ++i) any |= !used[cand[i]]; if (!any) continue; }
            int p = -1;
            for (int i = 0; i < nc && p < 0; ++i) if (!used[cand[i]]) p = cand[i];
            if (policy != 0) {
                unsigned long long best = ~0ull;
                for (int i = 0; i < nc; ++i) {
                    const int r = cand[i];
                    if (used[r]) continue;
                    unsigned long long key;
                    const uint64_t hi = b == 63 ? 0ull : (M[(size_t)r * W + w] >> (b + 1));
                    if (policy == 1) {
                        unsigned long long wt = __builtin_popcountll(hi);
                        for (int x = w + 1; x < W; ++x) wt += __builtin_popcountll(M[(size_t)r * W + x]);
                        key = wt;
                    } else if (policy == 2) key = est[r];
                    else if (policy == 3) key = nab[r];
                    else if (policy == 4) key = __builtin_popcountll(hi);
                    else if (policy == 6) key = cms[r] + (unsigned)__builtin_popcountll(pmk[r]);
                    else key = ((unsigned long long)est[r] << 8) | (unsigned)__builtin_popcountll(hi);
                    if (key < best) { best = key; p = r; }
                }
            }
            used[p] = pure ? 1 : 2;
            ++rank;
            const uint64_t* pr = M + (size_t)p * W;
            int pw = 0;
            for (int x = w + 1; x < W; ++x) pw += pr[x] != 0;
            pivwords += pw;
            const int qd = (int)((long long)rank * 4 / (m + 1));
            for (int i = 0; i < nc; ++i) {
                const int r = cand[i];
                if (r == p) continue;
                uint64_t* rr = M + (size_t)r * W;
                for (int x = w; x < W; ++x) rr[x] ^= pr[x];
                ++adds;
                words += pw;
                ++adds_q[qd];
                words_q[qd] += pw;
                ++nab[r];
                pmk[r] ^= pmk[p] ^ (1ull << q);
                const unsigned long long e = (unsigned long long)est[r] + est[p];
                est[r] = e > 0xffffffu ? 0xffffffu : (uint32_t)e;
            }
            ++q;
        }
        if (q > 0) {
            for (int r = 0; r < m; ++r) { gm[(size_t)r * 4 + ngo] = used[r] != 1 ? pmk[r] : 0ull; }
            ++ngo;
        }
        for (int r = 0; r < m; ++r) { if (!used[r]) cms[r] += (unsigned)__builtin_popcountll(pmk[r]); pmk[r] = 0; }
        if (ngo == 4 || (w == W - 1 && ngo > 0) || (rank >= m && ngo > 0)) {
            // the apply pass: listed rows = not frozen, some mask non-zero; in ascending row order, 64 per wave
            ++passes;
            int inwave = 0, wmax = 0;
            for (int r = 0; r < m; ++r) {
                if (used[r] == 1) continue;
                int tot = 0;
                for (int g = 0; g < ngo; ++g) {
                    const int pc = __builtin_popcountll(gm[(size_t)r * 4 + g]);
                    if (pc) { int bin = pc <= 4 ? pc - 1 : (pc <= 8 ? 4 : (pc <= 16 ? 5 : (pc <= 32 ? 6 : 7))); ++hist_grp[bin]; }
                    tot += pc;
                }
                if (!tot) continue;
                ++nact_tot; bits_tot += tot;
                { int bin = tot <= 4 ? tot - 1 : (tot <= 8 ? 4 : (tot <= 16 ? 5 : (tot <= 32 ? 6 : 7))); ++hist_row[bin]; }
                if (tot > wmax) wmax = tot;
                if (++inwave == 64) { wavemax_tot += wmax; ++waves_tot; inwave = 0; wmax = 0; }
            }
            if (inwave) { wavemax_tot += wmax; ++waves_tot; }
            for (int r = 0; r < m; ++r) { if (used[r] == 2) used[r] = 1; for (int g = 0; g < 4; ++g) gm[(size_t)r * 4 + g] = 0; }
            ngo = 0;
        }
    }
    printf("policy %d rank %d adds %lld row_words %lld listed %lld listmax %lld pivot_row_words %lld | by quarter of the rank: adds %lld %lld %lld %lld words %lld %lld %lld %lld\n",
           policy, rank, adds, words, listed, listmax, pivwords, adds_q[0], adds_q[1], adds_q[2], adds_q[3], words_q[0], words_q[1], words_q[2],
           words_q[3]);
    printf("  apply passes %lld listed rows %lld mask bits %lld (%.2f per listed row); per wave of 64 listed rows: mean of the max %.1f; rows by total bits 1 2 3 4 5-8 9-16 17-32 33+: %lld %lld %lld %lld %lld %lld %lld %lld; (row, group) by bits: %lld %lld %lld %lld %lld %lld %lld %lld\n",
           passes, nact_tot, bits_tot, (double)bits_tot / (nact_tot ? nact_tot : 1), (double)wavemax_tot / (waves_tot ? waves_tot : 1), hist_row[0], hist_row[1], hist_row[2], hist_row[3], hist_row[4], hist_row[5],
           hist_row[6], hist_row[7], hist_grp[0], hist_grp[1], hist_grp[2], hist_grp[3], hist_grp[4], hist_grp[5], hist_grp[6], hist_grp[7]);
    return 0;
}
