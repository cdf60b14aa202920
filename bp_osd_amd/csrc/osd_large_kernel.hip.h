// osd_large_kernel.hip.h -- OSD-0 / OSD-E / OSD-CS for codes beyond the register-resident kernel
// (m > 1024 or n > 2047; BASELINE configs[4]: 14520 x 29524 -> 53.7 MB of packed matrix per syndrome).
//
// Same algorithm and selection rule as osd_kernel.hip.h (rows a8-a11 of SURVEY.md §8), re-mapped for the
// HBM-bound regime.  One 1024-thread workgroup post-processes one syndrome at a time; thread t owns rows
// t, t + 1024, ... (RPT of them).  The permuted packed matrix lives in a per-workgroup global workspace
// M[word][row] (word-major: the rows of one 64-column word are contiguous -> every access by "my rows"
// is coalesced).
//
// Elimination = right-looking blocked Gauss-Jordan with LAZY trailing updates.  A "group" is the set of
// (<= 64) pivots found in one 64-column word; its effect on a row r is  row_r ^= XOR_{q in t_g(r)} P_g[q]
// with t_g(r) a 64-bit mask and P_g[q] the pivot rows as they were when the group started.  Instead of
// sweeping the whole remaining matrix after every word (2 * (W - w) * m * 8 bytes per word: 28 GB per
// syndrome at 14520 x 29524), up to OSDL_K groups stay OPEN: their masks (TmO) and pivot rows (PRO) are
// parked in HBM, and
//   E1  the next word is brought up to date on the fly (16-entry "four Russians" tables, one per nibble
//       of the mask, built from PRO; conflict-free in LDS because one table is 128 contiguous bytes);
//   E2  the panel phase runs in registers exactly as in the small kernel: one barrier per pivot, the
//       lowest proposed column wins;
//   E3  the new group's pivot rows are materialised for all later words (each wave owns a word and a
//       private table, no block barriers);
//   AP  when OSDL_K groups are open, ONE pass over the remaining matrix applies all of them
//       (tables for OSDL_CW words at a time) -> HBM traffic / OSDL_K (+ the masks re-read per chunk).
// OSD-0 / OSD-E need only wspan + 1 reduced columns (the first wspan non-pivot columns and the syndrome), so for them
// the elimination is GAUSSIAN with a back-substitution at the end instead of Gauss-Jordan ("gauss" below): a pivot row
// is FROZEN by the first apply pass after it became a pivot (that pass still brings it up to date with every open group,
// so the stored row is a complete linear combination); E1, the panel phase and all later apply passes leave frozen rows
// alone.  The apply pass walks a compacted list of the rows that are not frozen, so its work shrinks with the number of
// unused rows -- on average half the row updates of Gauss-Jordan -- and the wspan + 1 columns are then solved against
// the (partly reduced) pivot rows from the last pivot word to the first.
// OSD-CS weighs all k' single candidates, i.e. needs every non-pivot column reduced: it keeps Gauss-Jordan.
// E2c (round 2): when few rows have a non-zero panel word -- the usual case on sparse codes -- those words go to an LDS list
// and ONE wave runs the pivot loop on it without a barrier per pivot (osdl_e2_compact_wave below; up to 1024 rows); lists of
// 1025-2048 rows are searched by all sixteen waves, two entries per lane, one barrier per pivot (round 4).  Gauss-Jordan lists
// the unused rows only and brings the earlier pivot rows up to date afterwards in one step per row (Jordan fix-up).  The
// all-rows E2 above remains for longer lists.  The apply pass (AP) walks only the rows that some open group touches.
// Round 4 (Gaussian mode): every group KEEPS the rows E3 materialises for it (PRO holds all groups), and the
// back-substitution reads them from there, coalesced, instead of gathering rows of M (pmask: which pivots of its own group
// a pivot row absorbed).
// Earlier words never change: a row that becomes a pivot later has zeros in every earlier non-pivot
// column, so the reduced columns the sweep reads are final as soon as their word is stored.
// Sort: bitonic network over a global key array (n up to 32767).  Sweep: per-wave ballots over the
// finished words re-read from M; singles' weights accumulate in a global int array.
// Non-uniform channel (P.cost != null: channel_probs vector, update_channel_probs, per-shot two-valued channel): candidate
// weights are fp64 sums of log(1/p_i) over the candidate's set bits accumulated in ASCENDING ORIGINAL BIT INDEX -- the
// reference's order -- one candidate per lane, exactly as in osd_kernel.hip.h; see the "fp64 weights" block of the sweep.
// Limits: m <= 16384, n <= 32767, osd_e order <= 16, osd_cs order <= 64 (any channel).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "osd_kernel.hip.h"

namespace bposd {

constexpr int OSDL_NT = 1024;
constexpr int OSDL_NW = OSDL_NT / 64;  // waves
constexpr int OSDL_CW = 8;             // chunk width (words) of the apply pass
#ifndef OSDL_K_OPEN
#define OSDL_K_OPEN 4
#endif
constexpr int OSDL_K = OSDL_K_OPEN;    // open (lazily applied) pivot groups
constexpr int OSDL_G5 = 13;            // 5-bit fields of a 64-bit pivot mask in the apply pass (12 x 5 + 4)
#ifndef OSDL_E3_X
#define OSDL_E3_X 0   // timing experiments (wrong results): 1 no store of the materialised rows, 2 no reads of the older groups' rows
#endif
#ifndef OSDL_E3D
#define OSDL_E3D 2   // words in flight per wave in E3 beyond the one being processed.  Round 3 measured 3 and 4 SLOWER (42 -> 57 / 60 M
                     // cycles per elimination) -- with FLAT loads, whose waits also waited for every prefetch (see osdl_e3_materialise);
                     // with global loads: 1 / 2 / 4 words give 8.09 / 8.16 / 8.07 k syndromes/s on l29k_ms_e15 (same box)
#endif
constexpr size_t OSDL_E3_LDS_OFF = 57344;   // bytes from the start of the phase union: behind the E1 tables and the panel lists (54920 B)
constexpr size_t OSDL_PCZ_LDS_OFF = OSDL_E3_LDS_OFF + (size_t)OSDL_NW * (OSDL_K - 1) * 256 * 8;   // [64] chunk maps of the new pivot rows
#ifndef OSDL_BACKSUB_V2
#define OSDL_BACKSUB_V2 1   // back-substitution with everything requested a step ahead and the pivots of a word solved for all right-hand sides at once (round 5)
#endif
#ifndef OSDL_BS_XL
#define OSDL_BS_XL 28
#endif
#ifndef OSDL_OVERLAP_E3
#define OSDL_OVERLAP_E3 1   // E3 of a group on waves 1-15 beside the NEXT panel's one-wave pivot search (round 5)
#endif
constexpr int OSDL_MAXSPAN = 16;       // max osd_e order, and max osd_cs order with fp64 (non-uniform channel) weights
constexpr int OSDL_MAXSPAN_CS = 64;    // max osd_cs order with integer weights (uniform channel): as on the small path; the
                                       // reduced columns of the first w non-pivots then live in a global workspace

struct OsdLargeParams {
    int m, n, W;  // W = ceil((n + 1) / 64)
    int rank;     // pivots to find (min(m, n) when the true rank is unknown)
    int osd_method, osd_order, tie_policy;
    int e_msb_first;  // osd_e enumeration order (osd_kernel.hip.h: osd_e_index)
    int nsort;    // power of two >= n
    int mrl;      // padded row count = 1024 * RPT
    const uint8_t* __restrict__ synd;
    const int* __restrict__ rp;
    const int* __restrict__ ci;
    const double* __restrict__ llr_ws;
    const int* __restrict__ osd_list;
    int* __restrict__ counters;
    uint8_t* __restrict__ out_osd0;
    uint8_t* __restrict__ out_osdw;
    // nullable: the same two rows again at [list slot][n] -- the host-pointer API downloads the bulk outputs right after BP
    // and patches the few OSD rows from these compact copies afterwards
    uint8_t* __restrict__ cmp_osd0;
    uint8_t* __restrict__ cmp_osdw;
    // per-workgroup workspaces (index blockIdx.x)
    unsigned long long* __restrict__ mat;   // [grid][W * mrl]
    unsigned long long* __restrict__ keys;  // [grid][nsort]
    int* __restrict__ kidx;                 // [grid][nsort]   order[j] after the sort
    int* __restrict__ inv;                  // [grid][n]
    int* __restrict__ pivrow;               // [grid][64 * W]  sorted position -> pivot row, -1 non-pivot
    int* __restrict__ rowpos;               // [grid][mrl]     pivot position of a row, -1 if unused
    int* __restrict__ wt;                   // [grid][64 * W]  weights of the single candidates
    unsigned long long* __restrict__ tmo;   // [grid][OSDL_K * mrl]     masks of the open groups
    unsigned long long* __restrict__ pro;   // pivot rows of the pivot groups at their start state, [word][q] per group.  Gauss-Jordan
                                            // (OSD-CS): [grid][OSDL_K * W * 64], the open groups only.  Gaussian (OSD-0 / OSD-E):
                                            // [grid][osd_large_pro_rows(W) * 64] -- EVERY group keeps its rows (the group of word w: the
                                            // words w + 1 .. W - 1), the back-substitution reads them from here
    size_t pro_stride;                      // elements per workgroup of `pro`
    unsigned long long* __restrict__ pmask; // [grid][64 * W] nullable (Gaussian only): by sorted column position, the pivot row's
                                            // combination mask within its own group, own bit included
    int* __restrict__ alist;                // [grid][mrl]     rows the apply pass still updates (compacted, ascending)
    uint8_t* __restrict__ xout;             // [grid][n]
    unsigned long long* __restrict__ cnz;   // [grid][mrl]  per row: bit c set when the STORED words 8c .. 8c+7 may be non-zero (a superset; E3 skips the rest)
    unsigned long long* __restrict__ gcnz;  // [grid][OSDL_K * 64]  the same for the start-state pivot rows of the open groups
    long long* __restrict__ dbg;            // nullable: phase clocks of list slot 0 (s_memtime ticks)
    int* __restrict__ rank_out;             // nullable: [0] = pivots found for list slot 0 (ctor-time rank probe)
    // fp64 candidate weights (non-uniform channel); all null / unused when the channel is uniform
    const double* __restrict__ cost;        // [n] log(1/p_i), nullable
    const uint8_t* __restrict__ sel;        // [B, n] nullable: per-syndrome choice between cost and cost_alt
    const double* __restrict__ cost_alt;    // [n]
    double* __restrict__ costs_ws;          // [grid][n]   this syndrome's per-bit costs
    unsigned long long* __restrict__ colvec_ws;  // [grid][OSDL_MAXSPAN_CS][RPT * OSDL_NW] reduced columns when osd_cs order > 16
    double* __restrict__ wd_ws;             // [grid][wdn] weights of the single candidates / of the osd_e patterns
    unsigned short* __restrict__ am_ws;     // [grid][mrl] per row: its entries in the first <= 16 non-pivot columns
    // osd_cs with fp64 weights and a pair span beyond 16 (the reference harness's defaults on a large code: osd_cs with
    // channel_update = "x->z", css_decode_sim.py:73-80,207-248): the same per row / per bit in 64-bit words, in HBM
    unsigned long long* __restrict__ am64_ws;  // [grid][mrl] nullable
    unsigned long long* __restrict__ cm64_ws;  // [grid][n]   nullable: per original bit, its entries in the first <= 64 non-pivot columns
    int wdn;                                // max(64 * W, 2^16)
    int packed_io;                          // 1: synd is [B][ceil(m/64)] words, out_* / cmp_* are rows of ceil(n/64) 64-bit words (osd_kernel.hip.h)
};

constexpr int OSDL_MAXPAIRS = OSDL_MAXSPAN * (OSDL_MAXSPAN - 1) / 2;

// Gaussian mode: word rows (of 64 pivots) of `pro` per workgroup -- group of word w holds the words w + 1 .. W - 1 at row
// osd_large_pro_base(W, w) + x
__host__ __device__ inline long long osd_large_pro_base(int W, int w) { return (long long)w * (2 * W - w - 1) / 2 - (w + 1); }
__host__ __device__ inline size_t osd_large_pro_rows(int W) { return (size_t)W * (W - 1) / 2 + 1; }

// n_fp: block length when fp64 candidate weights are needed (adds the per-bit info words of that path), else 0
__host__ __device__ inline size_t osd_large_union_bytes(int W, int RPT, int n_fp = 0) {
    size_t e3 = OSDL_PCZ_LDS_OFF + 64 * 8;                  // wave-private tables of E3 behind the panel lists (so that E3 of one group can run
                                                            // beside the next panel's one-wave pivot search), the new pivot rows' chunk maps behind them
    size_t ap = (size_t)OSDL_K * OSDL_G5 * OSDL_CW * 32 * 8;  // apply-pass tables (5-bit; covers E1's OSDL_K * 256 entries)
    size_t sw = (size_t)OSDL_MAXSPAN * RPT * OSDL_NW * 8 + (size_t)RPT * OSDL_NW * 8 + (size_t)W * 8 + 64 * 4;
    // region R behind them: the wspan + 1 solution vectors of the back-substitution, later the per-bit info words of the
    // fp64-weight path
    const size_t zb = (size_t)(OSDL_MAXSPAN + 1) * (W + 2) * 8;  // the vectors + two result words each
    const size_t fb = n_fp > 0 ? ((size_t)n_fp * 4 + 7) / 8 * 8 + (size_t)OSDL_MAXPAIRS * 8 : 0;
    sw += zb > fb ? zb : fb;
    size_t b = e3 > ap ? e3 : ap;
    return b > sw ? b : sw;
}

__host__ __device__ inline size_t osd_large_lds_bytes(int W, int RPT, int n_fp = 0) {
    size_t b = 0;
    b += (size_t)2 * OSDL_NW * 2 * 8;            // pbuf
    b += (size_t)2 * OSDL_NW * 4;                // pcol
    b += (size_t)OSDL_K * 64 * 4 + 2 * OSDL_K * 4;   // grow, gnp, gbo
    b += 2 * 8 + 16 * 4;                         // best64, misc
    b += osd_large_union_bytes(W, RPT, n_fp);
    return b + 64;
}

// Row-word access as  uniform base (SGPR pair) + opaque 32-bit byte offset.  Without the opaque step the
// compiler hoists one 64-bit address per owned row and per base out of every loop and spills them.
__device__ __forceinline__ unsigned int osdl_opaque(unsigned int v) {
    asm volatile("" : "+v"(v));
    return v;
}
#ifndef BPOSD_LDS_MERGED
typedef const volatile __attribute__((address_space(3))) unsigned long long* osdl_lds_ptr;
#else
typedef const unsigned long long* osdl_lds_ptr;
#endif
#define OSDL_AT(type, base, byteoff) (*(type*)((char*)(base) + (size_t)(unsigned int)(byteoff)))
// matrix words of the apply pass: streamed once per pass (60 MB per workgroup, no reuse any cache could catch) -- non-temporal,
// so that what IS re-read (the rows' combination masks, once per 8-word chunk) stays in the L2 / Infinity Cache
#ifndef OSDL_NT_ROWS
#define OSDL_NT_ROWS 0  // measured (tools/ab_libs_l29k.sh): 179.8 vs 180.5 ms per 252 eliminations -- no effect, off
#endif
#if OSDL_NT_ROWS
#define OSDL_ROW_LD(base, byteoff) __builtin_nontemporal_load((const unsigned long long*)((const char*)(base) + (size_t)(unsigned int)(byteoff)))
#define OSDL_ROW_ST(base, byteoff, val) __builtin_nontemporal_store((unsigned long long)(val), (unsigned long long*)((char*)(base) + (size_t)(unsigned int)(byteoff)))
#else
#define OSDL_ROW_LD(base, byteoff) OSDL_AT(unsigned long long, base, byteoff)
#define OSDL_ROW_ST(base, byteoff, val) (OSDL_AT(unsigned long long, base, byteoff) = (val))
#endif

// ---- E2c: the panel phase on a compacted list (Gaussian mode: every unfrozen row with a non-zero panel word; Gauss-Jordan:
// the unused ones, see the Jordan fix-up at the call site).  A row whose panel word is zero after E1 stays zero for
// the whole panel, and with reliability-sorted columns of a sparse code few rows are non-zero (14520 x 29524 code, OSD-E:
// <= 256 in three panels out of four, never more than 1024).  The non-zero words sit in an LDS list; ONE wave runs the
// pivot loop on CR list entries per lane: wave minimum on the DPP path, pivot word and mask broadcast by v_readlane -- no
// barrier and no LDS round trip per pivot (~400 / ~900 cycles per pivot for CR = 4 / 16 against ~10000 for the all-rows
// form, which remains for denser panels).  A separate, non-inlined function: its registers are
// allocated on their own, the caller holds no 16-row window when it runs.
// LDS (offsets from lpw): words [CAP] u64, masks [CAP] u64, ids [CAP] u32 (row | used << 31), pivots [128] u32
// (list position of pivot q, then its column), new-pivot bits [1024] u16 (bit k of entry t: row t + 1024 k).
typedef volatile __attribute__((address_space(3))) unsigned long long* osdl_lds_w64;
typedef volatile __attribute__((address_space(3))) unsigned int* osdl_lds_w32;
constexpr int OSDL_E2C_CAP = 2048;   // longest panel list (rows); beyond it the all-rows form runs (4096: measured, no gain)
#ifndef OSDL_MW_MIN
#define OSDL_MW_MIN 1024  // lists longer than this are searched by all sixteen waves (512 / 256: measured 6 % slower in round 4; 512 again in round 5, when
                          // the sixteen-entries-per-lane instance of the one-wave loop had begun to spill 63 dwords under the lightest-row keys:
                          // l29k_ms_e15 10.39 k against 10.26 k syndromes/s, OSD alone 96.8 against 102 ms -- the spilling instance still wins)
#endif

// Which ROW becomes the pivot of a column is free (the pivot SET, hence every output, does not depend on it -- SURVEY.md
// Appendix A.4: upstream takes the lowest-weight row "only" to limit fill-in).  Round 5: among the candidates of a column the
// row that has absorbed the FEWEST pivot rows so far wins -- key (absorbed << 14) | row, unique, so the choice no longer
// depends on the order in which LDS atomics listed the rows.  "Absorbed" = sum over the closed panels of popcount(the row's
// combination mask in that panel), kept for unused rows in rowpos[] as -1 - count (a used row holds its pivot position there,
// >= 0), plus popcount(mask so far) inside the panel.  CPU model of L29k eliminations (tools/fillin_sim.c, policy 6 against 0):
// row additions 348 k -> 205 k and 190 k -> 174 k, changed row-words 22.3 M -> 7.7 M and 9.9 M -> 5.4 M, longest panel list
// 1199 -> 669 -- within 15 % of choosing by the exact remaining row weight (policy 1), which the lazy elimination cannot know.
#ifndef OSDL_PIVOT_LIGHT
#define OSDL_PIVOT_LIGHT 1
#endif
#ifndef OSDL_E2C_SKIP
#define OSDL_E2C_SKIP OSDL_PIVOT_LIGHT  // one-wave pivot search: register slots without a one in the column are skipped (round 5)
#endif
constexpr unsigned int OSDL_CNT_MAX = 16383u - 64u;  // stored counts saturate here: count + popcount(mask) stays below 2^14

// GAUSS (OSD-0 / OSD-E, round 5): plain Gaussian elimination -- a pivot row takes no further row additions once it is chosen
// (its word and mask go to the list at once and its registers are cleared, so it neither matches nor proposes again).  Rounds
// 2-4 kept reducing the pivot rows of the open groups (Jordan steps inside the panel and the window); those rows carried 70 % of
// all combination-mask bits (tools/fillin_sim.c: 17-64 bits per pivot row and pass against 1-2 for an unused row) and every
// one of them was walked by the apply pass, for nothing: in Gaussian mode nobody reads a pivot row's words to the right of its
// panel from M again (the back-substitution reads PRO).  What the Jordan steps bought -- the <= 64 pivots of a word being
// independent in the back-substitution -- is now a 64-step loop of one wave per word there.
template <int CR, bool GAUSS>
__device__ __attribute__((noinline)) void osdl_e2_compact_wave(unsigned int lpw_addr, unsigned int misc_addr, int nnz,
                                                              unsigned long long vmask, int rank, int nrank, int done_in, int* rowpos_) {
    constexpr int CAP = OSDL_E2C_CAP;
    const int lane = threadIdx.x & 63;
    osdl_lds_w64 Lpw = (osdl_lds_w64)(size_t)lpw_addr;
    osdl_lds_w64 Lt = Lpw + CAP;
    osdl_lds_w32 Lid = (osdl_lds_w32)(Lt + CAP);
    osdl_lds_w32 Lpiv = Lid + CAP;
    osdl_lds_w32 misc = (osdl_lds_w32)(size_t)misc_addr;
    typedef __attribute__((address_space(1))) int g_i32;
    g_i32* rowpos = (g_i32*)rowpos_;
    unsigned long long cp[CR], ct[CR];
    unsigned int cu = 0u;  // bit s: my s-th entry is a pivot row
#if OSDL_PIVOT_LIGHT
    unsigned int key0[CR];  // (pivot rows absorbed in earlier panels << 14) | row
#pragma unroll
    for (int s2 = 0; s2 < CR; ++s2) {
        const int pos = s2 * 64 + lane;
        key0[s2] = pos < nnz ? (Lid[pos] & 0x3fffu) : 0u;
    }
    {
        int rp[CR];  // the CR requests are in flight together
#pragma unroll
        for (int s2 = 0; s2 < CR; ++s2) rp[s2] = rowpos[key0[s2]];
#pragma unroll
        for (int s2 = 0; s2 < CR; ++s2) key0[s2] |= (rp[s2] < 0 ? (unsigned int)(-1 - rp[s2]) : 0u) << 14;
    }
#endif
#pragma unroll
    for (int s2 = 0; s2 < CR; ++s2) {
        const int pos = s2 * 64 + lane;
        cp[s2] = pos < nnz ? Lpw[pos] : 0ull;
        ct[s2] = 0ull;
        if (pos < nnz && (Lid[pos] >> 31)) cu |= 1u << s2;
    }
    int cnp = 0, cnr = nrank;
    bool cdone = done_in != 0;
#pragma clang loop unroll(disable)
    for (;;) {
        if (cnr >= rank) { cdone = true; break; }
        unsigned long long cand = 0ull;
#pragma unroll
        for (int s2 = 0; s2 < CR; ++s2) cand |= (!GAUSS && ((cu >> s2) & 1u)) ? 0ull : cp[s2];
        cand &= vmask;
        const unsigned int lb = osd_ffs64_or_64(cand);
        const unsigned int wk = osd_wave_min_u32((lb << 6) | (unsigned int)lane);  // wave-uniform
        const int col = (int)(wk >> 6);
        if (col >= 64) break;
        // every lane picks "its" entry with a one in the column (only the winning lane's pick is read)
        int kb = 0;
        unsigned long long a = 0ull, c = 0ull;
#if OSDL_E2C_SKIP
        // Round 5: a column of the panel has a handful of ones among the list's rows, so most of the CR register slots hold no
        // row with a one there in ANY lane: a slot is tested by one wave-uniform branch (4 instructions) and only the slots that
        // pass run the key comparison (12) and, below, the row addition (7) -- as first written every slot ran both (~20) per pivot.
        const bool colhi = col >= 32;                       // uniform
        const unsigned int cbit = 1u << (col & 31);
        unsigned int slots = 0u;                            // uniform: bit s2 -- some lane's entry s2 has a one in the column
        unsigned int bk = ~0u;  // the lane's lightest candidate: fewest absorbed pivot rows, then lowest row
#pragma unroll
        for (int s2 = CR - 1; s2 >= 0; --s2) {
            const unsigned int half = colhi ? (unsigned int)(cp[s2] >> 32) : (unsigned int)cp[s2];
            const bool one = (half & cbit) != 0u;
            if (__ballot(one) != 0ull) {  // uniform
                slots |= 1u << s2;
                const bool hit = one && (GAUSS || ((cu >> s2) & 1u) == 0u);
                const unsigned int key = key0[s2] + ((unsigned int)__popcll(ct[s2]) << 14);
                const bool better = hit && key < bk;
                bk = better ? key : bk;
                kb = better ? s2 : kb;
                a = better ? cp[s2] : a;
                c = better ? ct[s2] : c;
            }
        }
        const unsigned int wmin = osd_wave_min_u32(bk);  // (some lane has a hit: the column was proposed)
        const int first = (int)__builtin_ctzll(__ballot(bk == wmin));
        const unsigned long long pw_p =
            ((unsigned long long)(unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)(a >> 32), first) << 32) |
            (unsigned long long)(unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)a, first);
        const unsigned long long t_p =
            ((unsigned long long)(unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)(c >> 32), first) << 32) |
            (unsigned long long)(unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)c, first);
        const unsigned long long tq = t_p ^ (1ull << cnp);
#pragma unroll
        for (int s2 = 0; s2 < CR; ++s2) {
            if ((slots >> s2) & 1u) {  // uniform
                const unsigned int half = colhi ? (unsigned int)(cp[s2] >> 32) : (unsigned int)cp[s2];
                const unsigned int mm = (half & cbit) != 0u ? ~0u : 0u;
                const unsigned int plo = __builtin_amdgcn_bitop3_b32((unsigned int)cp[s2], (unsigned int)pw_p, mm, 0x78);  // a ^ (b & c)
                const unsigned int phi = __builtin_amdgcn_bitop3_b32((unsigned int)(cp[s2] >> 32), (unsigned int)(pw_p >> 32), mm, 0x78);
                const unsigned int tlo = __builtin_amdgcn_bitop3_b32((unsigned int)ct[s2], (unsigned int)tq, mm, 0x78);
                const unsigned int thi = __builtin_amdgcn_bitop3_b32((unsigned int)(ct[s2] >> 32), (unsigned int)(tq >> 32), mm, 0x78);
                cp[s2] = ((unsigned long long)phi << 32) | plo;
                ct[s2] = ((unsigned long long)thi << 32) | tlo;
            }
        }
#else
#if OSDL_PIVOT_LIGHT
        unsigned int bk = ~0u;  // the lane's lightest candidate: fewest absorbed pivot rows, then lowest row
#pragma unroll
        for (int s2 = CR - 1; s2 >= 0; --s2) {
            const bool hit = (((cp[s2] >> col) & 1ull) != 0ull) && (GAUSS || ((cu >> s2) & 1u) == 0u);
            const unsigned int key = key0[s2] + ((unsigned int)__popcll(ct[s2]) << 14);
            const bool better = hit && key < bk;
            bk = better ? key : bk;
            kb = better ? s2 : kb;
            a = better ? cp[s2] : a;
            c = better ? ct[s2] : c;
        }
        const unsigned int wmin = osd_wave_min_u32(bk);  // (some lane has a hit: the column was proposed)
        const int first = (int)__builtin_ctzll(__ballot(bk == wmin));
#else
        const int first = (int)(wk & 63u);
#pragma unroll
        for (int s2 = CR - 1; s2 >= 0; --s2) {
            const bool hit = (((cp[s2] >> col) & 1ull) != 0ull) && (GAUSS || ((cu >> s2) & 1u) == 0u);
            kb = hit ? s2 : kb;
            a = hit ? cp[s2] : a;
            c = hit ? ct[s2] : c;
        }
#endif
        const unsigned long long pw_p =
            ((unsigned long long)(unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)(a >> 32), first) << 32) |
            (unsigned long long)(unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)a, first);
        const unsigned long long t_p =
            ((unsigned long long)(unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)(c >> 32), first) << 32) |
            (unsigned long long)(unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)c, first);
        const unsigned long long tq = t_p ^ (1ull << cnp);
#pragma unroll
        for (int s2 = 0; s2 < CR; ++s2) {
            const unsigned long long mm = (unsigned long long)((long long)(cp[s2] << (63 - col)) >> 63);
            cp[s2] ^= pw_p & mm;
            ct[s2] ^= tq & mm;
        }
#endif
        if (lane == first) {
            if (GAUSS) {  // the pivot row is final: to the list, out of the registers
                Lpw[kb * 64 + lane] = pw_p;
                Lt[kb * 64 + lane] = t_p;
            }
#pragma unroll
            for (int s2 = 0; s2 < CR; ++s2)
                if (s2 == kb) { cp[s2] = GAUSS ? 0ull : pw_p; ct[s2] = GAUSS ? 0ull : t_p; }  // (Jordan: the pivot row itself is put back)
            cu |= 1u << kb;
            Lpiv[cnp] = (unsigned int)(kb * 64 + lane);
            Lpiv[64 + cnp] = (unsigned int)col;
        }
        ++cnp;
        ++cnr;
    }
#pragma unroll
    for (int s2 = 0; s2 < CR; ++s2) {
        const int pos = s2 * 64 + lane;
        if (pos < nnz && !(GAUSS && ((cu >> s2) & 1u))) {
            Lpw[pos] = cp[s2];
            Lt[pos] = ct[s2];
#if OSDL_PIVOT_LIGHT
            // the pivot rows this (still unused) row absorbed in the panel join its count
            if (ct[s2] != 0ull && ((cu >> s2) & 1u) == 0u) {
                const unsigned int cnt = (key0[s2] >> 14) + (unsigned int)__popcll(ct[s2]);
                rowpos[key0[s2] & 0x3fffu] = -1 - (int)(cnt < OSDL_CNT_MAX ? cnt : OSDL_CNT_MAX);
            }
#endif
        }
    }
    if (lane == 0) { misc[4] = (unsigned int)cnp; misc[5] = (unsigned int)cnr; misc[6] = cdone ? 1u : 0u; }
}

// a8: bitonic network on (key, index), blocks of 8192 elements staged through LDS (96 KB): every exchange whose partners are
// less than 8192 apart runs on LDS (114 of the 120 passes at NS = 32768); the three passes that cross blocks stay in the
// workspace, eight exchanges per thread requested together.  All in the workspace it was 120 passes of sixteen exchanges per
// thread, each waiting for its own loads: 7 M cycles per elimination, now 2.3 M.  A function of its own (named address
// spaces, registers of its own): inlined, the same code left the sort faster and the apply pass's row walk 50 % slower -- the
// kernel sits at its 128-VGPR cap and every change of its body moves the allocation of the hot loop.
__device__ __attribute__((noinline)) void osdl_sort(unsigned long long* keys_, int* kidx_, unsigned int lds_addr, int NS) {
    typedef __attribute__((address_space(1))) unsigned long long g_u64;
    typedef __attribute__((address_space(1))) int g_i32;
    typedef __attribute__((address_space(3))) unsigned long long l_u64;
    typedef __attribute__((address_space(3))) int l_i32;
    g_u64* keys = (g_u64*)keys_;
    g_i32* kidx = (g_i32*)kidx_;
    const int tid = threadIdx.x;
    constexpr int NT = OSDL_NT;
    const int BS = NS < 8192 ? NS : 8192;
    l_u64* lk = (l_u64*)(size_t)lds_addr;                 // [BS]
    l_i32* li = (l_i32*)(size_t)(lds_addr + 8192u * 8u);  // [BS]
    auto lds_passes = [&](int base, int k, int jstart) {
        for (int j = jstart; j > 0; j >>= 1) {
            for (int t = tid; t < (BS >> 1); t += NT) {
                const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int hi = lo | j;
                const bool up = (((base + lo) & k) == 0);
                const unsigned long long ka = lk[lo], kb = lk[hi];
                const int ia = li[lo], ib = li[hi];
                const bool a_gt_b = (ka > kb) || (ka == kb && ia > ib);
                if (a_gt_b == up) {
                    lk[lo] = kb; lk[hi] = ka;
                    li[lo] = ib; li[hi] = ia;
                }
            }
            __syncthreads();
        }
    };
    auto block_in = [&](int base) {
        for (int i = tid; i < BS; i += NT) { lk[i] = keys[base + i]; li[i] = kidx[base + i]; }
        __syncthreads();
    };
    auto block_out = [&](int base) {
        for (int i = tid; i < BS; i += NT) { keys[base + i] = lk[i]; kidx[base + i] = li[i]; }
        __syncthreads();
    };
    for (int base = 0; base < NS; base += BS) {
        block_in(base);
        for (int k = 2; k <= BS; k <<= 1) lds_passes(base, k, k >> 1);
        block_out(base);
    }
    for (int k = BS << 1; k <= NS; k <<= 1) {
        for (int j = k >> 1; j >= BS; j >>= 1) {
            constexpr int SB = 8;  // (NS / 2 is a multiple of 8 * 1024 here)
            for (int t0 = tid; t0 < (NS >> 1); t0 += NT * SB) {
                unsigned long long ka[SB], kb[SB];
                int ia[SB], ib[SB];
#pragma unroll
                for (int i = 0; i < SB; ++i) {
                    const int t = t0 + i * NT;
                    const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                    const int hi = lo | j;
                    ka[i] = keys[lo]; kb[i] = keys[hi];
                    ia[i] = kidx[lo]; ib[i] = kidx[hi];
                }
#pragma unroll
                for (int i = 0; i < SB; ++i) {
                    const int t = t0 + i * NT;
                    const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                    const int hi = lo | j;
                    const bool up = ((lo & k) == 0);
                    const bool a_gt_b = (ka[i] > kb[i]) || (ka[i] == kb[i] && ia[i] > ib[i]);
                    if (a_gt_b == up) {
                        keys[lo] = kb[i]; keys[hi] = ka[i];
                        kidx[lo] = ib[i]; kidx[hi] = ia[i];
                    }
                }
            }
            __syncthreads();
        }
        for (int base = 0; base < NS; base += BS) {
            block_in(base);
            lds_passes(base, k, BS >> 1);
            block_out(base);
        }
    }
}

// Build "my rows" of the permuted matrix (the matrix is zero on entry).  A row's column positions are fetched together (its
// edges' bit indices, then their sorted positions: two round trips), the bits that share a word are combined in registers and
// every word is written once.  As first written each edge was a chain of three dependent round trips (bit index, position,
// read-modify-write of the word), 33 per row, 528 per thread.  A function of its own (see osdl_sort).
template <int RPT>
__device__ __attribute__((noinline)) void osdl_build_rows(unsigned long long* M_, const int* rp_, const int* ci_, const int* inv_, const uint8_t* synd_,
                                                          long long s, int m, int W, int MRL, int packed, unsigned long long* cnz_) {
    typedef __attribute__((address_space(1))) unsigned long long g_u64;
    typedef __attribute__((address_space(1))) int g_i32;
    typedef __attribute__((address_space(1))) uint8_t g_u8;
    g_u64* M = (g_u64*)M_;
    const g_i32* rp = (const g_i32*)rp_;
    const g_i32* ci = (const g_i32*)ci_;
    const g_i32* inv = (const g_i32*)inv_;
    const g_u8* synd = (const g_u8*)synd_;
    const int tid = threadIdx.x;
    constexpr int NT = OSDL_NT;
#pragma clang loop unroll(disable)
    for (int k = 0; k < RPT; ++k) {
        const int r = tid + k * NT;
        if (r < m) {
            constexpr int CH = 12;  // edges per batch (the degree of a check of the codes this path serves; more: further batches)
            const int e0 = rp[r], e1 = rp[r + 1];
            unsigned long long cz = 0ull;  // the row's chunk map: bit (word >> 3)
            const bool sb = packed ? ((((const __attribute__((address_space(1))) unsigned long long*)synd)[(size_t)s * (size_t)((m + 63) >> 6) + (r >> 6)] >> (r & 63)) & 1ull) != 0ull
                                   : (synd[(size_t)s * m + r] & 1) != 0;
            for (int eb = e0; eb < e1 || eb == e0; eb += CH) {
                int jj[CH + 1];
#pragma unroll
                for (int i = 0; i < CH; ++i) jj[i] = (eb + i < e1) ? ci[eb + i] : -1;
#pragma unroll
                for (int i = 0; i < CH; ++i) jj[i] = (jj[i] >= 0) ? inv[jj[i]] : -1;
                jj[CH] = (eb == e0 && sb) ? (W - 1) * 64 + 63 : -1;  // the syndrome column, with the first batch
#pragma unroll
                for (int i = 0; i <= CH; ++i)
                    if (jj[i] >= 0) cz |= 1ull << (jj[i] >> 9);
#pragma unroll
                for (int i = 0; i <= CH; ++i) {
                    if (jj[i] < 0) continue;
                    unsigned long long word = 0ull;
                    bool first = true;  // is i the first entry of its word in this batch?
#pragma unroll
                    for (int i2 = 0; i2 <= CH; ++i2) {
                        const bool same = jj[i2] >= 0 && (jj[i2] >> 6) == (jj[i] >> 6);
                        if (same) word |= 1ull << (jj[i2] & 63);
                        if (same && i2 < i) first = false;
                    }
                    if (first) {
                        g_u64* dst = M + (size_t)(jj[i] >> 6) * MRL + r;  // only the owner touches row r
                        if (eb == e0) *dst = word;    // (the matrix is zero: no read)
                        else *dst |= word;            // a later batch may meet a word of an earlier one
                    }
                }
                if (e1 == e0) break;
            }
            ((__attribute__((address_space(1))) unsigned long long*)cnz_)[r] = cz;
        } else if (r < MRL) {
            ((__attribute__((address_space(1))) unsigned long long*)cnz_)[r] = 0ull;
        }
    }
}

// E3 as a function of its own: the pivot rows of the new group (group index ng, npiv rows listed in grow) at their start
// state for every word to the right of w, written to PRO.  Not inlined, so that its registers are allocated on their own
// (the kernel sits at its 128-VGPR cap: inside it, more than one word in flight per wave went to scratch and ran slower).
__device__ __attribute__((noinline)) void osdl_e3_materialise(unsigned long long* U, const int* grow_, const int* gnp_, const unsigned long long* TmO_,
                                                              unsigned long long* PRO_, const int* gbo_, const unsigned long long* M_, int MRL, int W, int w,
                                                              int ng, int npiv, const unsigned long long* cnz_, unsigned long long* pcz_,
                                                              int x_first, int x_end, int wave0, int nwv) {
    // words x_first .. x_end - 1, shared by the nwv waves wave0 .. wave0 + nwv - 1 (the others return at once): all sixteen for the
    // whole range, or wave 0 for the one word the next panel needs and waves 1-15 for the rest while wave 0 searches that panel's pivots
    const int lane = threadIdx.x & 63;
    const int wave = (int)(threadIdx.x >> 6) - wave0;
    if (wave < 0 || wave >= nwv) return;
    // The pointers arrive generic.  Left so, every access below is a FLAT instruction, which counts on the vector-memory AND the
    // LDS counter: the wait before a stage's first table write then waits for every row word requested for the LATER stages too,
    // and the OSDL_E3D-deep prefetch is no prefetch (E3 34 -> 31 M cycles per elimination with the spaces named).  Named address spaces
    // give global_load / ds_read with their own counters.
    typedef __attribute__((address_space(1))) unsigned long long g_u64;
    typedef __attribute__((address_space(3))) int l_i32;
    const g_u64* TmO = (const g_u64*)TmO_;
    g_u64* PRO = (g_u64*)PRO_;
    const g_u64* M = (const g_u64*)M_;
    const l_i32* grow = (const l_i32*)grow_;
    const l_i32* gnp = (const l_i32*)gnp_;
    const l_i32* gbo = (const l_i32*)gbo_;

                        typedef volatile __attribute__((address_space(3))) unsigned long long* lds_rw;
                        lds_rw tw = (lds_rw)(U + OSDL_E3_LDS_OFF / 8 + (size_t)wave * (OSDL_K - 1) * 256);
                        const int row = (lane < npiv) ? grow[ng * 64 + lane] : 0;
                        // Round 5: the row's chunk map (bit c: its STORED words 8c .. 8c+7 may be non-zero; kept conservatively by the
                        // build and the apply passes).  The lightest-row choice makes pivot rows of rows that absorbed little, so most of
                        // their stored words are zero -- and this phase is bound by its gather of 64 scattered row words per word.  A
                        // lane whose map clears the chunk reads the start of the matrix instead (one line for all such lanes).
                        const unsigned long long mycz = (lane < npiv) ? ((const g_u64*)cnz_)[row] : 0ull;
                        unsigned long long pmap = 0ull;  // chunks in which my pivot row's START state is non-zero
                        unsigned long long mrow[OSDL_K - 1];
    #pragma unroll
                        for (int g = 0; g < OSDL_K - 1; ++g) mrow[g] = (g < ng && lane < npiv) ? TmO[(size_t)g * MRL + row] : 0ull;
                        // lane L < 16 * ng builds the 16-entry table of (group L >> 4, nibble L & 15) by a Gray-code walk
                        // from its 4 pivot-row words; the inputs of the next word are requested before this word is
                        // processed (the loop is otherwise bound by the latency of its own loads)
                        const int tg = lane >> 4, tgrp = lane & 15;
                        const bool builder = tg < ng;
                        const int tnp = builder ? gnp[tg] - 4 * tgrp : 0;
                        const int q0 = 4 * tgrp;
                        // OSDL_E3D words in flight per wave: the loop is bound by the latency of its own gathers (a row word
                        // per lane from M, four pivot-row words from PRO), ~10 k cycles each under full load
                        unsigned long long prq[OSDL_E3D][4], mvq[OSDL_E3D];
                        auto issue = [&](int d, int xx) {
                            if (xx < x_end) {
                                const g_u64* src = PRO + ((long long)gbo[builder ? tg : 0] + xx) * 64 + q0;
    #pragma unroll
#if OSDL_E3_X != 2 && OSDL_E3_X != 3
                                for (int kk = 0; kk < 4; ++kk) prq[d][kk] = src[kk];
#else
                                for (int kk = 0; kk < 4; ++kk) prq[d][kk] = (unsigned long long)(size_t)src + kk;
#endif
                                unsigned long long ma = (unsigned long long)(((mycz >> (xx >> 3)) & 1ull) ? M + (size_t)xx * MRL + row : M);
                                asm volatile("" : "+v"(ma));  // (an opaque address: the request count stays a constant, nothing waits early)
#if OSDL_E3_X != 3
                                mvq[d] = *(const g_u64*)ma;
#else
                                mvq[d] = ma;
#endif
                            }
                        };
                        auto process = [&](int d, int x) {
                            unsigned long long pr[4];
    #pragma unroll
                            for (int kk = 0; kk < 4; ++kk) pr[kk] = (builder && kk < tnp) ? prq[d][kk] : 0ull;
                            unsigned long long v = (lane < npiv && ((mycz >> (x >> 3)) & 1ull)) ? mvq[d] : 0ull;
                            issue(d, x + nwv * OSDL_E3D);  // this stage's registers are free again
                            if (builder) {
                                unsigned int idx = osdl_opaque(((unsigned int)lane >> 1) & 15u);  // conflict-free start entries
                                unsigned long long tv = 0ull;
    #pragma unroll
                                for (int kk = 0; kk < 4; ++kk)
                                    if ((idx >> kk) & 1u) tv ^= pr[kk];
                                lds_rw tp = tw + lane * 16;
                                tp[idx] = tv;
    #pragma unroll
                                for (int i = 1; i < 16; ++i) {
                                    const int bit = (i & 1) ? 0 : ((i & 2) ? 1 : ((i & 4) ? 2 : 3));  // ctz(i)
                                    idx ^= 1u << bit;
                                    tv ^= pr[bit];
                                    tp[idx] = tv;
                                }
                            }
                            __builtin_amdgcn_wave_barrier();
    #pragma unroll
                            for (int g = 0; g < OSDL_K - 1; ++g) {
                                if (g < ng) {
                                    const int ngrp = (gnp[g] + 3) >> 2;
                                    for (int grp = 0; grp < ngrp; ++grp)
                                        v ^= tw[(g * 16 + grp) * 16 + (int)((mrow[g] >> (4 * grp)) & 15ull)];
                                }
                            }
#if OSDL_E3_X != 1 && OSDL_E3_X != 3
                            PRO[((long long)gbo[ng] + x) * 64 + lane] = v;
#else
                            if (v == 0x123456789abcdefull) PRO[((long long)gbo[ng] + x) * 64 + lane] = v;
#endif
                            pmap |= (v != 0ull) ? (1ull << (x >> 3)) : 0ull;
                            __builtin_amdgcn_wave_barrier();
                        };
                        const int xs = x_first + wave;
    #pragma unroll
                        for (int d = 0; d < OSDL_E3D; ++d) issue(d, xs + d * nwv);
                        for (int x = xs; x < x_end; x += nwv * OSDL_E3D) {
    #pragma unroll
                            for (int d = 0; d < OSDL_E3D; ++d)
                                if (x + d * nwv < x_end) process(d, x + d * nwv);
                        }
                        // the sixteen waves' shares of the new pivot rows' maps meet in LDS (zeroed by the caller before its barrier)
                        if (pmap) __hip_atomic_fetch_or((unsigned long long*)(pcz_ + lane), pmap, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// The back-substitution of the Gaussian mode as a function of its own (registers allocated on their own: inside the kernel, at
// its 128-VGPR cap, the values kept across a step went to scratch, and a scratch reload waits for every row word requested
// ahead).  zv: [17][W] the right-hand sides / solutions by sorted column position (LDS), resw: [2][17] (LDS).  Returns xt.
// Round 5 (second form of this loop).  Three things bound the first form (25-30 k cycles per pivot word, 14 M per L29k
// elimination under load):
//  - every right-hand side was carried through every word, although a unit right-hand side e_t is zero to the right of t
//    and so is its solution (a pivot row has no entry to the left of its pivot column): beyond the word of the LAST
//    search column (xt; the search columns are the first non-pivot columns in sorted order, words 23-30 of 462 on L29k)
//    only the syndrome's vector is non-zero -- one look-up and two bit operations per word instead of 17 and 34;
//  - a step was a chain of dependent round trips to memory (the pivot positions, two batches of row words, the own
//    word and mask of each pivot row), none of which depends on the solution: now each is requested a step ahead;
//  - the <= 64 pivots of a word were solved by a 12-instruction dependent step each, one right-hand side per lane.
//    Now lane l holds pivot row l's own word and, bit c, its bit of right-hand side c: solving pivot column j is one
//    v_readlane (column j's solution bits, all right-hand sides at once) and one v_bitop3 on every lane.
// One barrier per step: the sums meet in resw[step parity], and the only wave that reads the word just solved in the
// next step is wave 0, which solved it.
__device__ __attribute__((noinline)) int osdl_backsub(unsigned long long* zv_, unsigned long long* resw_, const int* tpos_, const int* pivrow_,
                                                       const unsigned long long* M_, const unsigned long long* pmask_, const unsigned long long* PRO_,
                                                       int W, int MRL, int wlast, int ntc_g) {
    typedef __attribute__((address_space(1))) unsigned long long g_u64;
    typedef __attribute__((address_space(1))) int g_i32;
    typedef volatile __attribute__((address_space(3))) unsigned long long l_u64;
    typedef __attribute__((address_space(3))) int l_i32;
    auto uni = [](const void* p_) {
        const unsigned long long a = (unsigned long long)p_;
        return ((unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)(a >> 32)) << 32) |
               (unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)a);
    };
    auto lds = [](const void* p_) { return (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)(size_t)(const __attribute__((address_space(3))) char*)p_); };
    l_u64* zv = (l_u64*)(size_t)lds(zv_);
    l_u64* resw = (l_u64*)(size_t)lds(resw_);
    const l_i32* tpos = (const l_i32*)(size_t)lds(tpos_);
    const g_i32* pivrow = (const g_i32*)uni(pivrow_);
    const g_u64* M = (const g_u64*)uni(M_);
    const g_u64* pmask = (const g_u64*)uni(pmask_);
    const g_u64* PRO = (const g_u64*)uni(PRO_);
    W = __builtin_amdgcn_readfirstlane(W);
    MRL = __builtin_amdgcn_readfirstlane(MRL);
    wlast = __builtin_amdgcn_readfirstlane(wlast);
    ntc_g = __builtin_amdgcn_readfirstlane(ntc_g);
    constexpr int NR = OSDL_MAXSPAN + 1;
    constexpr int SYN = OSDL_MAXSPAN;
    constexpr int XL = OSDL_BS_XL;  // row words a wave requests ahead of a step (L29k: <= 28 per wave beyond xt)
    const int tid = threadIdx.x, lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int xt = ntc_g > 0 ? (tpos[ntc_g - 1] >> 6) : -1;
    if (tid < NR) resw[NR + tid] = 0ull;  // resw: [2][NR] by step parity
    __syncthreads();
    l_u64* zsyn = zv + (size_t)SYN * W;
    unsigned long long vx[XL];
#pragma unroll
    for (int i = 0; i < XL; ++i) vx[i] = 0ull;
    // my first word of step w_ in which the syndrome's vector alone is non-zero
    auto first_alone = [&](int w_) { return (w_ > xt ? w_ : xt) + 1 + wave; };
    auto request = [&](int w_) {
        const g_u64* gq = PRO + osd_large_pro_base(W, w_) * 64 + lane;
        const int xb = first_alone(w_);
#pragma unroll
        for (int i4 = 0; i4 < XL / 4; ++i4) {
            if (xb + 64 * i4 < W) {  // uniform
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int x = xb + OSDL_NW * (4 * i4 + i);
                    vx[4 * i4 + i] = gq[(size_t)(x < W ? x : W - 1) * 64];
                }
            }
        }
    };
    int prow0 = pivrow[wlast * 64 + lane];                        // lane q: the pivot at sorted position 64 w + q, if any
    int prow1 = wlast > 0 ? pivrow[(wlast - 1) * 64 + lane] : -1;  // ... of the step after
    unsigned long long vb0 = 0ull, S0 = 0ull;                     // (wave 0) pivot row's own word (final in M) and its mask
    if (wave == 0 && prow0 >= 0) {
        vb0 = M[(size_t)wlast * MRL + prow0];
        S0 = pmask[wlast * 64 + lane];
    }
    request(wlast);
    // One step (pivot word w).  FULL: the step may have words up to xt to the right of it, where every right-hand side is carried
    // (17 accumulators and 17 solution words in registers: that code spills, and a spill reload anywhere in a loop makes every wait
    // behind it a wait for the row words requested ahead -- so the steps w >= xt, all but ~25 of 462, run in a loop without it).
    auto step = [&](const int w, auto FULL) {
        constexpr bool full = decltype(FULL)::value;
        const int prow = prow0;
        const unsigned long long pv = __ballot(prow >= 0);  // (the same in every wave)
        const int np = __popcll(pv);  // = the group's pivots: slots 0 .. np - 1 of its rows in PRO
        l_u64* rw = resw + (w & 1) * NR;
        unsigned long long asyn = 0ull;  // the syndrome's sum of this step, my words
        if (pv != 0ull) {  // uniform
            const g_u64* gp = PRO + osd_large_pro_base(W, w) * 64 + lane;
            const int xb = first_alone(w);
            unsigned int alo = 0u, ahi = 0u;
#pragma unroll
            for (int i = 0; i < XL; ++i) {
                const int x = xb + OSDL_NW * i;
                if (x < W) {  // uniform
                    const unsigned long long zs = zsyn[x];
                    alo = __builtin_amdgcn_bitop3_b32(alo, (unsigned int)vx[i], (unsigned int)zs, 0x78);  // a ^ (b & c)
                    ahi = __builtin_amdgcn_bitop3_b32(ahi, (unsigned int)(vx[i] >> 32), (unsigned int)(zs >> 32), 0x78);
                }
            }
            for (int x = xb + OSDL_NW * XL; x < W; x += OSDL_NW) {  // (wider matrices than the requests cover)
                const unsigned long long v = gp[(size_t)x * 64], zs = zsyn[x];
                alo ^= (unsigned int)v & (unsigned int)zs;
                ahi ^= (unsigned int)(v >> 32) & (unsigned int)(zs >> 32);
            }
            asyn = ((unsigned long long)ahi << 32) | alo;
        }
        // Everything requested a step ago has been waited for by now -- and the compiler is told so here: its wait counters run in
        // order and it cannot count requests behind uniform branches, so a wait for any of these AFTER the requests below would
        // be a wait for all of those too (as first written: wave 0 waited for its row words before solving the word, every step).
#pragma unroll
        for (int i = 0; i < XL; ++i) asm volatile("" : "+v"(vx[i]));
        asm volatile("" : "+v"(vb0), "+v"(S0), "+v"(prow0), "+v"(prow1));
        const int prow2 = w >= 2 ? pivrow[(w - 2) * 64 + lane] : -1;
        if (w > 0) request(w - 1);  // (vx is free again) the next step's words arrive while this one is solved
        if (full && pv != 0ull && w < xt) {  // uniform
            // the words up to the last search column's: every right-hand side, in two halves (seventeen sums and seventeen solution
            // words at once took this function to 128 VGPRs -- and a callee that clobbers the registers the KERNEL keeps its spilled
            // SGPRs in doubled the kernel's own scratch reloads, 536 -> 1164 in the ISA, l29k_ms_e15 11.0 k -> 10.6 k syndromes/s)
            const g_u64* gp = PRO + osd_large_pro_base(W, w) * 64 + lane;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                constexpr int HN = 9;
                unsigned long long a9[HN];
#pragma unroll
                for (int i = 0; i < HN; ++i) a9[i] = 0ull;
                for (int x = w + 1 + wave; x <= xt; x += OSDL_NW) {
                    const unsigned long long v = gp[(size_t)x * 64];
                    unsigned long long zz[HN];
#pragma unroll
                    for (int i = 0; i < HN; ++i) zz[i] = (h * HN + i < NR) ? zv[(size_t)(h * HN + i) * W + x] : 0ull;
                    const unsigned int vlo = (unsigned int)v, vhi = (unsigned int)(v >> 32);
#pragma unroll
                    for (int i = 0; i < HN; ++i) {
                        const unsigned int lo = __builtin_amdgcn_bitop3_b32((unsigned int)a9[i], vlo, (unsigned int)zz[i], 0x78);
                        const unsigned int hi = __builtin_amdgcn_bitop3_b32((unsigned int)(a9[i] >> 32), vhi, (unsigned int)(zz[i] >> 32), 0x78);
                        a9[i] = ((unsigned long long)hi << 32) | lo;
                    }
                }
#pragma unroll
                for (int i = 0; i < HN; ++i) {
                    const int c = h * HN + i;
                    if (c == SYN) asyn ^= a9[i];
                    else if (c < ntc_g) {  // uniform
                        const unsigned long long bits = __ballot(lane < np && (__popcll(a9[i]) & 1));
                        if (lane == 0 && bits) __hip_atomic_fetch_xor((unsigned long long*)(rw + c), bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
            }
        }
        if (pv != 0ull) {
            const unsigned long long bits = __ballot(lane < np && (__popcll(asyn) & 1));
            if (lane == 0 && bits) __hip_atomic_fetch_xor((unsigned long long*)(rw + SYN), bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        unsigned long long vb1 = 0ull, S1 = 0ull;
        if (wave == 0 && prow1 >= 0) {
            vb1 = M[(size_t)(w - 1) * MRL + prow1];
            S1 = pmask[(w - 1) * 64 + lane];
        }
        __syncthreads();
        if (wave == 0 && pv != 0ull) {
            // bit c of `part`: right-hand side c's sum for my pivot row over everything but this word's pivot columns --
            // parity(S & A_c) from the words to the right and the row's own word against the right-hand side's own bits
            unsigned int part = 0u;
            if (w <= xt) {  // uniform
#pragma unroll
                for (int c = 0; c < OSDL_MAXSPAN; ++c) {
                    if (c < ntc_g) {
                        const unsigned long long r = rw[c], z0 = zv[(size_t)c * W + w];
                        part |= (unsigned int)((__popcll(S0 & r) + __popcll(vb0 & z0)) & 1) << c;
                    }
                }
            }
            {
                const unsigned long long r = rw[SYN], z0 = zsyn[w];
                part |= (unsigned int)((__popcll(S0 & r) + __popcll(vb0 & z0)) & 1) << SYN;
            }
            __builtin_amdgcn_wave_barrier();
            if (lane < NR) rw[lane] = 0ull;
            // pivot columns from the last to the first: column j's row is complete when its turn comes, and its bits are
            // added to the rows that have an entry there (rows l < j; the own bit and the non-pivot columns are masked out)
            const unsigned long long vbm = vb0 & pv & ~(1ull << lane);
            const int mlo = (int)(unsigned int)vbm, mhi = (int)(unsigned int)(vbm >> 32);
#pragma unroll
            for (int j = 63; j >= 0; --j) {
                const unsigned int m = (unsigned int)__builtin_amdgcn_readlane((int)part, j);
                const unsigned int t = (unsigned int)__builtin_amdgcn_sbfe(j < 32 ? mlo : mhi, (unsigned int)(j & 31), 1u);
                part = __builtin_amdgcn_bitop3_b32(part, t, m, 0x78);
            }
#pragma unroll
            for (int c = 0; c < NR; ++c) {
                if (c == SYN || (c < ntc_g && w <= xt)) {  // uniform
                    const unsigned long long bits = __ballot(prow >= 0 && ((part >> c) & 1u));
                    if (lane == 0 && bits) zv[(size_t)c * W + w] |= bits;  // (no right-hand side has a bit of its own at a pivot column)
                }
            }
        }
        prow0 = prow1;
        prow1 = prow2;
        vb0 = vb1;
        S0 = S1;
    };
    int w = wlast;
#pragma clang loop unroll(disable)
    for (; w >= 0 && w >= xt; --w) step(w, std::false_type{});
#pragma clang loop unroll(disable)
    for (; w >= 0; --w) step(w, std::true_type{});
    return xt;
}

// E1c as a function of its own (round 5): every thread brings the panel word of its RPT rows up to date with the open groups and
// appends the non-zero ones to the LDS list.  In the kernel body this ran in batches of four rows, each batch waiting for its own
// loads (four dependent round trips per panel, ~6 k cycles each with 250 eliminations in flight), and looked every row's four
// masks up in the nibble tables -- although fewer than one row in fifty has a non-zero mask in an open group (`anymask`).  Here
// the words of ALL my rows and the masks of my first TWO rows that have any (a thread seldom owns more) are requested together
// and waited for once; only those rows go through the tables; further such rows take a loop of their own.  Lanes without a
// masked row read their own panel word again in the masks' place (the line has just been requested: no traffic of its own),
// so the number of requests is a constant and nothing waits early.  Returns the bit set of my rows whose stored word is
// non-zero but whose up-to-date word is zero (they are cleared in M by the caller).
template <int RPT>
__device__ __attribute__((noinline)) unsigned int osdl_e1c_list(const unsigned long long* Mw_, const unsigned long long* TmO_, const unsigned long long* U_,
                                                                unsigned long long* Lpw_, unsigned int* Lid_, int* cnt_, const int* gnp_, int MRL, int ng,
                                                                unsigned int skipmask, unsigned int anymask, unsigned int usedmask) {
    typedef __attribute__((address_space(1))) unsigned long long g_u64;
    typedef volatile __attribute__((address_space(3))) unsigned long long l_u64;
    typedef volatile __attribute__((address_space(3))) unsigned int l_u32;
    typedef __attribute__((address_space(3))) int l_i32;
    auto uni = [](const void* p_) {
        const unsigned long long a = (unsigned long long)p_;
        return ((unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)(a >> 32)) << 32) |
               (unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)a);
    };
    auto lds = [](const void* p_) { return (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)(size_t)(const __attribute__((address_space(3))) char*)p_); };
    const g_u64* Mw = (const g_u64*)uni(Mw_);
    const g_u64* TmO = (const g_u64*)uni(TmO_);
    l_u64* U = (l_u64*)(size_t)lds(U_);
    l_u64* Lpw = (l_u64*)(size_t)lds(Lpw_);
    l_u32* Lid = (l_u32*)(size_t)lds(Lid_);
    l_i32* cnt = (l_i32*)(size_t)lds(cnt_);
    const l_i32* gnp = (const l_i32*)(size_t)lds(gnp_);
    MRL = __builtin_amdgcn_readfirstlane(MRL);
    ng = __builtin_amdgcn_readfirstlane(ng);
    constexpr int NT = OSDL_NT;
    constexpr int CAP = OSDL_E2C_CAP;
    const int tid = threadIdx.x, lane = threadIdx.x & 63;
    const unsigned int need = anymask & ~skipmask;  // my rows with a non-zero mask in some open group
    const int s0 = need ? (int)__builtin_ctz(need) : -1;
    const unsigned int need1 = need & (need - 1u);
    const int s1 = need1 ? (int)__builtin_ctz(need1) : -1;
    unsigned int rest = need1 & (need1 - 1u);
    unsigned long long old[RPT], m0[OSDL_K], m1[OSDL_K];
#pragma unroll
    for (int k = 0; k < RPT; ++k) old[k] = Mw[tid + k * NT];
#pragma unroll
    for (int g = 0; g < OSDL_K; ++g) {
        // (the address goes through an opaque register: otherwise the compiler sees that the stand-in IS old[0], waits for it and puts
        // the real load under a branch -- two round trips again)
        unsigned long long a0 = (unsigned long long)((g < ng && s0 >= 0) ? TmO + (size_t)g * MRL + tid + s0 * NT : Mw + tid);
        unsigned long long a1 = (unsigned long long)((g < ng && s1 >= 0) ? TmO + (size_t)g * MRL + tid + s1 * NT : Mw + tid);
        asm volatile("" : "+v"(a0), "+v"(a1));
        m0[g] = *(const g_u64*)a0;
        m1[g] = *(const g_u64*)a1;
    }
    // the nibble tables: U[(g * 16 + grp) * 16 + nibble]
    auto delta = [&](const unsigned long long (&mm)[OSDL_K]) {
        unsigned long long d = 0ull;
#pragma unroll
        for (int g = 0; g < OSDL_K; ++g) {
            if (g < ng) {
                const int ngrp = (gnp[g] + 3) >> 2;
                for (int grp = 0; grp < ngrp; ++grp) d ^= U[(g * 16 + grp) * 16 + (int)((mm[g] >> (4 * grp)) & 15ull)];
            }
        }
        return d;
    };
    unsigned long long d0 = 0ull, d1 = 0ull;
    if (s0 >= 0) d0 = delta(m0);
    if (s1 >= 0) d1 = delta(m1);
    unsigned long long v[RPT];
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        old[k] = ((skipmask >> k) & 1u) ? 0ull : old[k];
        v[k] = old[k] ^ (k == s0 ? d0 : 0ull) ^ (k == s1 ? d1 : 0ull);
    }
    // a thread with more than two masked rows (rare): one row at a time
    while (__ballot(rest != 0u) != 0ull) {
        if (rest != 0u) {
            const int sk = (int)__builtin_ctz(rest);
            rest &= rest - 1u;
            unsigned long long mm[OSDL_K];
#pragma unroll
            for (int g = 0; g < OSDL_K; ++g) mm[g] = g < ng ? TmO[(size_t)g * MRL + tid + sk * NT] : 0ull;
            const unsigned long long d = delta(mm);
#pragma unroll
            for (int k = 0; k < RPT; ++k) v[k] ^= (k == sk) ? d : 0ull;
        }
    }
    unsigned int zmask = 0u;
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        // list positions: ONE LDS atomic per wave and row slot (the wave's entries are consecutive)
        const bool nz = v[k] != 0ull;
        const unsigned long long bal = __ballot(nz);
        int base = 0;
        if (bal) {  // uniform
            if (lane == (int)__builtin_ctzll(bal)) base = __hip_atomic_fetch_add((int*)cnt, __popcll(bal), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            base = __builtin_amdgcn_readlane(base, (int)__builtin_ctzll(bal));
        }
        if (nz) {
            const int pos = base + __popcll(bal & ((1ull << lane) - 1ull));
            if (pos < CAP) {
                Lpw[pos] = v[k];
                Lid[pos] = (unsigned int)(tid + k * NT) | (((usedmask >> k) & 1u) << 31);
            }
        } else if (old[k] != 0ull) {
            zmask |= 1u << k;
        }
    }
    return zmask;
}

// AP, sparse form (round 5).  The table walk of the apply pass costs the same whatever the rows' combination masks hold: per
// listed row and word one look-up per 5-bit field of every open group (46 of 52 on L29k, measured), although with plain
// Gaussian elimination a pass lists ~300 rows with 4-5 mask bits each (tools/fillin_sim.c) -- 96 % of the look-ups return entry
// 0 because SOME lane of the wave needs that field.  Here the pass works on the SET BITS of the masks: every bit is one
// entry (row, group, pivot) of a work list in LDS, the pivot rows of the open groups are staged raw (16 KB per 8-word chunk,
// double-buffered, requested three chunks ahead: no Gray-code table build), an entry reads the pivot row's eight words and
// sends each non-zero one to the matrix as a returnless atomic XOR.  Chosen per pass when the bits fit the list
// (OSDL_SPARSE_LIST; the caller keeps the count as the groups are formed); denser passes keep the table walk, whose cost per
// row does not grow with the bits.  Measured on l29k_ms_e15 (same box, tools/ab_libs_l29k.sh): 8.65 k -> 10.4 k syndromes/s
// together with the two changes it builds on (lightest pivot row; pivot rows final when chosen), OSD alone 118 -> 99 ms per
// 252 eliminations, median elimination 203 M -> 133 M cycles; what a pass still costs (~230 k cycles) is the rate at which a
// CU's vector-memory path takes scattered 64-bit atomics (~60 k per pass, ~4 cycles each).
// A function of its own: registers allocated on their own (the kernel body sits at its 128-VGPR cap), nothing of the caller's
// per-thread state is needed -- the row list, the masks, PRO and M are all in memory.
#ifndef OSDL_SPARSE_LIST
#define OSDL_SPARSE_LIST 6144  // passes with at most this many mask bits take the sparse form (eight entries per walking thread)
#endif
// Returns 1 when the pass was done, 0 -- nothing touched -- when the rows' mask bits do not fit the list after all (the caller's
// count is an upper bound; this is the belt to its braces).
__device__ __attribute__((noinline)) int osdl_apply_sparse(unsigned long long* U_, const int* alist_, const unsigned long long* TmO_,
                                                            const unsigned long long* PRO_, unsigned long long* M_, const int* gnp_, const int* gbo_,
                                                            int MRL, int W, int xlo, int ng, int nact, unsigned long long* cnz_, const unsigned long long* gcnz_) {
    typedef __attribute__((address_space(1))) unsigned long long g_u64;
    typedef __attribute__((address_space(1))) int g_i32;
    typedef __attribute__((address_space(1))) char g_char;
    typedef __attribute__((address_space(3))) unsigned long long l_u64;
    typedef __attribute__((address_space(3))) int l_i32;
    typedef __attribute__((address_space(3))) unsigned int l_u32;
    // The arguments of a non-inlined function arrive in VECTOR registers: left so, every address below is 64-bit vector
    // arithmetic on values the compiler must take for divergent (128 VGPRs and spills inside the loop in the first build).  Moved
    // to scalar registers once, the bases are SGPR pairs and the accesses take the base + 32-bit offset form.
    auto uni = [](const void* p_) {
        const unsigned long long a = (unsigned long long)p_;
        return ((unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)(a >> 32)) << 32) |
               (unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)a);
    };
    MRL = __builtin_amdgcn_readfirstlane(MRL); W = __builtin_amdgcn_readfirstlane(W); xlo = __builtin_amdgcn_readfirstlane(xlo);
    ng = __builtin_amdgcn_readfirstlane(ng); nact = __builtin_amdgcn_readfirstlane(nact);
    const g_i32* alist = (const g_i32*)uni(alist_);
    const g_u64* TmO = (const g_u64*)uni(TmO_);
    const g_u64* PRO = (const g_u64*)uni(PRO_);
    g_u64* M = (g_u64*)uni(M_);
    g_u64* cnz = (g_u64*)uni(cnz_);
    const g_u64* gcnz = (const g_u64*)uni(gcnz_);
    const l_i32* gnp = (const l_i32*)(size_t)(unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)(size_t)(const l_i32*)gnp_);
    const l_i32* gbo = (const l_i32*)(size_t)(unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)(size_t)(const l_i32*)gbo_);
    l_u64* PRW = (l_u64*)(size_t)(unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)(size_t)(l_u64*)U_);  // [2][OSDL_K][OSDL_CW][64]
    constexpr int NT = OSDL_NT;
    constexpr int BUF = OSDL_K * 64 * OSDL_CW;
    constexpr int NSW = 4;                       // staging waves
    constexpr int NST = NSW * 64;                // staging threads
    constexpr int NWT = NT - NST;                // walking threads
    constexpr int EPT = BUF / NST;               // staged words per staging thread and chunk
    constexpr int EPW = OSDL_SPARSE_LIST / NWT;  // list entries per walking thread
    static_assert(BUF % NST == 0 && OSDL_CW == 8, "staging assumes 8-word chunks and a whole number of words per thread");
    static_assert(OSDL_SPARSE_LIST % NWT == 0, "whole entries per walking thread");
    const int tid = threadIdx.x;
    const int nch = (W - xlo + OSDL_CW - 1) / OSDL_CW;
    l_u32* ELIST = (l_u32*)(PRW + 2 * BUF);      // [OSDL_SPARSE_LIST] the pass's (row, group, pivot) entries
    l_i32* ecount = (l_i32*)(ELIST + OSDL_SPARSE_LIST);
    // ---- The work list of the pass: one entry per SET BIT of a listed row's masks -- (row, group, pivot), i.e. "row absorbs that
    // pivot row" -- packed as  row * 8 | (group * 512 + pivot) << 17.  A lane walks ENTRIES, not rows: a row's mask over the
    // start states of a group's pivot rows has 1-2 bits as a rule but 30-100 for one row in twelve (a pivot row that absorbed
    // many others before it was chosen hands its whole combination on), and a wave that walks rows runs as long as its
    // heaviest row (first build: 25 k cycles per chunk, as slow as the table walk).  Entries are all alike.  Their changes of
    // one matrix word need no combining: XOR atomics commute.
    if (tid == 0) *ecount = 0;
    __syncthreads();
    for (int i = tid; i < nact; i += NT) {
        const int row = alist[i];
        unsigned long long mk[OSDL_K];
        int nb = 0;
#pragma unroll
        for (int g = 0; g < OSDL_K; ++g) {
            mk[g] = g < ng ? TmO[(size_t)g * MRL + row] : 0ull;
            nb += __popcll(mk[g]);
        }
        if (nb) {
            int pos = atomicAdd((int*)ecount, nb);
            unsigned long long cz = 0ull;  // chunks the absorbed pivot rows may fill in this row (E3's map of stored non-zero chunks)
#pragma unroll
            for (int g = 0; g < OSDL_K; ++g) {
                unsigned long long m = mk[g];
                while (m) {
                    const int q = (int)__builtin_ctzll(m);
                    m &= m - 1ull;
                    if (pos < OSDL_SPARSE_LIST) ELIST[pos] = ((unsigned int)row << 3) | ((unsigned int)(g * (64 * OSDL_CW) + q) << 17);
                    cz |= gcnz[g * 64 + q];
                    ++pos;
                }
            }
            cnz[row] |= cz;  // (the row is this thread's for the pass; if the list overflows the caller's table walk sets the whole map)
        }
    }
    __syncthreads();
    const int nent = __builtin_amdgcn_readfirstlane(*ecount);
    if (nent > OSDL_SPARSE_LIST) return 0;  // (uniform) the caller takes the table form
    if (nent == 0) return 1;
    // ---- The pass is a chain of short steps (a few hundred cycles of LDS work per chunk), and anything a step WAITS for from
    // HBM costs it a round trip (~6 k cycles with 250 eliminations in flight).  Two things keep waits off the steps' path:
    // (i) a row's words are not loaded at all -- the change goes out as a returnless 64-bit atomic XOR per changed word
    // (performed in the L2); the workgroup's later plain loads of M come after a barrier and an invalidation of this CU's L1.
    // (ii) The waves are SPECIALISED: four of them only stage (the words of chunk c + 3 are requested in step c and wait in
    // registers), the other twelve only walk and never wait for memory.  In one instruction stream the two do not mix: the
    // vector-memory counter of gfx9 counts loads and atomics together and in order, and the number of atomics a step issues is
    // not a compile-time constant, so every wait for a staging word also waited for the previous step's atomics to come back
    // from the L2 (one round trip per step again).  Steps are separated by a bare s_barrier (LDS counter drained by hand, no
    // memory fence: the atomics need not be complete before the END of the pass).
#define OSDL_STEP_BARRIER()                                   \
    do {                                                      \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    \
        __builtin_amdgcn_s_barrier();                         \
        asm volatile("" ::: "memory");                        \
    } while (0)
    if (tid < NST) {
        // ================= staging waves.  Element e = tid + i * 256 of a chunk's block: group e / 512, word (e >> 6) & 7, pivot
        // e & 63 (a wave reads 512 contiguous bytes of PRO); LDS layout [group][word][pivot] = e
        auto load_chunk = [&](int c, unsigned long long (&val)[EPT]) {
#pragma unroll
            for (int i = 0; i < EPT; ++i) {
                const int e = tid + i * NST;
                const int g = e >> 9, xx = (e >> 6) & 7, q = e & 63;
                const int x = xlo + c * OSDL_CW + xx;
                const bool ok = c < nch && g < ng && x < W;
                val[i] = PRO[((long long)gbo[ok ? g : 0] + (ok ? x : xlo)) * 64 + q];  // (clamped, unconditional: no branch, no wait)
            }
        };
        auto store_chunk = [&](int c, const unsigned long long (&val)[EPT]) {
#pragma unroll
            for (int i = 0; i < EPT; ++i) {
                const int e = tid + i * NST;
                const int g = e >> 9, xx = (e >> 6) & 7, q = e & 63;
                const bool ok = g < ng && xlo + c * OSDL_CW + xx < W && q < gnp[g < ng ? g : 0];
                PRW[(c & 1) * BUF + e] = ok ? val[i] : 0ull;  // [group][word][pivot]: e itself
            }
        };
        // R[c mod 3]: the words of chunk c.  The step is unrolled three times so that the ring is indexed by constants (a rotation
        // by register moves would wait for the newest request).
        unsigned long long R[3][EPT];
        load_chunk(0, R[0]);
        load_chunk(1, R[1]);
        load_chunk(2, R[2]);
        store_chunk(0, R[0]);
        load_chunk(3, R[0]);
        auto step = [&](int c, auto IC) {
            constexpr int i = decltype(IC)::value;  // c mod 3
            OSDL_STEP_BARRIER();  // block c is complete; every walker has left chunk c - 1, which read the other block
            store_chunk(c + 1, R[(i + 1) % 3]);   // (requested three steps ago)
            load_chunk(c + 4, R[(i + 1) % 3]);
        };
        // (the loop is left where the chunks end, not skipped through: a path "step 0, skip, skip, step 0" would make the compiler
        // wait in step 0 for the request step 0 has just made)
#pragma clang loop unroll(disable)
        for (int c = 0;;) {
            step(c, std::integral_constant<int, 0>{});
            if (++c >= nch) break;
            step(c, std::integral_constant<int, 1>{});
            if (++c >= nch) break;
            step(c, std::integral_constant<int, 2>{});
            if (++c >= nch) break;
        }
    } else {
        // ================= walking waves: thread wt takes entries wt, wt + 768, ... (in registers for the whole pass)
        const int wt = tid - NST;
        const int nj = (nent + NWT - 1) / NWT;
        unsigned int ent[EPW];
#pragma unroll
        for (int j = 0; j < EPW; ++j) ent[j] = (wt + j * NWT < nent) ? ELIST[wt + j * NWT] : 0xffffffffu;
#pragma clang loop unroll(disable)
        for (int c = 0; c < nch; ++c) {
            OSDL_STEP_BARRIER();
            const int x0 = xlo + c * OSDL_CW;
            const int cw = (W - x0) < OSDL_CW ? (W - x0) : OSDL_CW;
            const l_u64* blk = PRW + (c & 1) * BUF;
#pragma unroll
            for (int j = 0; j < EPW; ++j) {
                if (j >= nj) break;  // uniform
                if (ent[j] != 0xffffffffu) {
                    // [group][word][pivot]: the lanes of one read differ in (group, pivot), bank pair = pivot mod 32 -- two lanes
                    // collide when their pivots are 32 apart.  ([group][pivot][8 words] with 128-bit reads put all 64 lanes on 16
                    // banks: 16-way conflicts.)  Volatile: single ds_read_b64 with immediate offsets, not the half-rate
                    // ds_read2st64_b64 pairs (same finding as in the table walk).
                    const volatile l_u64* src = (const volatile l_u64*)(blk + (ent[j] >> 17));
                    const unsigned int ro = ent[j] & 0x1ffffu;
                    unsigned long long v[OSDL_CW];
#pragma unroll
                    for (int xx = 0; xx < OSDL_CW; ++xx) v[xx] = src[xx * 64];
#pragma unroll
                    for (int xx = 0; xx < OSDL_CW; ++xx)
                        if (xx < cw && v[xx] != 0ull)
                            __hip_atomic_fetch_xor((g_u64*)((g_char*)(M + (size_t)(x0 + xx) * MRL) + ro), v[xx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
    }
#undef OSDL_STEP_BARRIER
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // my atomics have been performed ...
    __syncthreads();                                   // ... and so have everyone's
    // The atomics were performed in the L2; this CU's vector L1 may still hold lines of M from before.  Agent-scope acquire =
    // invalidate the L1.
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    return 1;
}

// (The apply pass as a non-inlined function of its own was measured too: 90 -> 104-113 M cycles per elimination -- its list build and
// its row loop live on state of the caller (row masks, frozen / used bits) that then crosses the call in memory.  It stays inlined.)
// (The back-substitution loop as a non-inlined function of its own: 13 -> 29 M cycles per elimination; it stays inlined.  E3 is the one
// phase that gains from an allocation of its own: 42 -> 25.6 M.)
template <int RPT>
__global__ __launch_bounds__(OSDL_NT) void osd_large_kernel(const OsdLargeParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int m = P.m, n = P.n, W = P.W, NS = P.nsort, MRL = P.mrl;
    constexpr int NT = OSDL_NT;
    int tid = threadIdx.x;  // re-laundered per phase (OSDL_FRESH_TID) so that per-row addresses and predicates
                            // derived from it are recomputed where they are used instead of being hoisted + spilled
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    constexpr int NCV = RPT * OSDL_NW;
#define OSDL_FRESH_TID() asm volatile("" : "+v"(tid))

    unsigned char* p = smem;
    unsigned long long* pbuf = (unsigned long long*)p; p += (size_t)2 * OSDL_NW * 2 * 8;
    unsigned long long* best64 = (unsigned long long*)p; p += 2 * 8;
    unsigned long long* U = (unsigned long long*)p; p += osd_large_union_bytes(W, RPT, P.cost ? n : 0);  // phase-dependent (see header)
    unsigned int* pcol = (unsigned int*)p; p += (size_t)2 * OSDL_NW * 4;
    int* grow = (int*)p; p += OSDL_K * 64 * 4;
    int* gnp = (int*)p; p += OSDL_K * 4;
    int* gbo = (int*)p; p += OSDL_K * 4;  // word-row base of each open group in PRO: its word x is at PRO + (gbo[g] + x) * 64
    int* misc = (int*)p;
    // sweep-phase view of U
    unsigned long long* colvec = U;
    unsigned long long* yvec = colvec + (size_t)OSDL_MAXSPAN * NCV;
    unsigned long long* npmask = yvec + NCV;
    int* tpos = (int*)(npmask + W);
    unsigned long long* R = (unsigned long long*)(tpos + 64);  // back-substitution vectors, then fp64-weight info words

    unsigned long long* M = P.mat + (size_t)blockIdx.x * W * MRL;
    unsigned long long* keys = P.keys + (size_t)blockIdx.x * NS;
    int* kidx = P.kidx + (size_t)blockIdx.x * NS;
    int* inv = P.inv + (size_t)blockIdx.x * n;
    int* pivrow = P.pivrow + (size_t)blockIdx.x * 64 * W;
    int* rowpos = P.rowpos + (size_t)blockIdx.x * MRL;
    int* wt = P.wt + (size_t)blockIdx.x * 64 * W;
    uint8_t* xout = P.xout + (size_t)blockIdx.x * n;
    unsigned long long* TmO = P.tmo + (size_t)blockIdx.x * OSDL_K * MRL;
    unsigned long long* PRO = P.pro + (size_t)blockIdx.x * P.pro_stride;
    unsigned long long* pmask = P.pmask ? P.pmask + (size_t)blockIdx.x * 64 * W : nullptr;
    int* alist = P.alist + (size_t)blockIdx.x * MRL;
    unsigned long long* cnz = P.cnz + (size_t)blockIdx.x * MRL;
    unsigned long long* gcnz = P.gcnz + (size_t)blockIdx.x * OSDL_K * 64;
    unsigned long long* pcz = U + OSDL_PCZ_LDS_OFF / 8;  // [64] (behind E3's tables)

    for (;;) {
        OSDL_FRESH_TID();
        if (tid == 0) misc[0] = atomicAdd(&P.counters[2], 1);
        __syncthreads();
        const int slot_id = misc[0];
        const int nlist = P.counters[1];
        if (slot_id >= nlist) break;
        const long long s = P.osd_list[slot_id];
        const double* llr = P.llr_ws + (size_t)slot_id * n;

#ifdef BPOSD_OSD_DIAG
        long long tk[28] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // 12: sparse apply passes, 24: their ticks  // 17..20 (apply pass): own table build, wait for the builders, wait for the row walkers, list build  // 13..16: back-substitution, column vectors, candidate sweep, write-out  // sort, build, E1, E2, E3, AP, sweep, words, groups, applies,
                                                               // apply look-ups per thread, apply row-words per thread
        long long t0 = (long long)__builtin_amdgcn_s_memtime();
#define OSDL_TICK(i)                                                     \
    do {                                                                 \
        const long long t1_ = (long long)__builtin_amdgcn_s_memtime();   \
        tk[i] += t1_ - t0;                                               \
        t0 = t1_;                                                        \
    } while (0)
#define OSDL_COUNT(i) tk[i] += 1
#define OSDL_ADD(i, v) tk[i] += (v)
#else
#define OSDL_TICK(i) do { } while (0)
#define OSDL_COUNT(i) do { } while (0)
#define OSDL_ADD(i, v) do { } while (0)
#endif
        // ------------------------------------------------------------------ a8: sort (global bitonic)
        for (int i = tid; i < NS; i += NT) {
            if (i < n) {
                keys[i] = llr_sort_key(llr[i]);
                kidx[i] = (P.tie_policy == 1) ? n - 1 - i : i;
            } else {
                keys[i] = ~0ull;
                kidx[i] = i;
            }
        }
        for (int i = tid; i < 64 * W; i += NT) { pivrow[i] = -1; wt[i] = 1; }
        for (int i = tid; i < MRL; i += NT) rowpos[i] = -1;
        for (int i = tid; i < n; i += NT) xout[i] = 0;
        unsigned short* am = P.cost ? P.am_ws + (size_t)blockIdx.x * MRL : nullptr;
        if (am)
            for (int i = tid; i < MRL; i += NT) am[i] = 0;
        unsigned long long* am64 = (P.cost && P.am64_ws) ? P.am64_ws + (size_t)blockIdx.x * MRL : nullptr;
        if (am64)
            for (int i = tid; i < MRL; i += NT) am64[i] = 0ull;
        __syncthreads();
        osdl_sort(keys, kidx, (unsigned int)(size_t)(osdl_lds_w64)U, NS);
        if (P.tie_policy == 1) {
            for (int i = tid; i < n; i += NT) kidx[i] = n - 1 - kidx[i];
            __syncthreads();
        }
        for (int j = tid; j < n; j += NT) inv[kidx[j]] = j;
        OSDL_TICK(0);
        OSDL_FRESH_TID();
        // ------------------------------------------- build my rows (zero, then set the <= DC bits)
        for (int x = 0; x < W; ++x) {
            const unsigned int ro = osdl_opaque((unsigned int)tid * 8u);
#pragma unroll
            for (int k = 0; k < RPT; ++k) OSDL_AT(unsigned long long, M + (size_t)x * MRL, ro + k * NT * 8) = 0ull;
        }
        __syncthreads();
        osdl_build_rows<RPT>(M, P.rp, P.ci, inv, P.synd, s, m, W, (int)MRL, P.packed_io, cnz);
        if (tid < 2 * OSDL_NW) pcol[tid] = 64u;
        if (tid == 0) misc[8] = 0;  // mask bits of the open groups
        __syncthreads();

        OSDL_TICK(1);
        // ------------------------------------------------------- a9: blocked Gauss-Jordan, lazy groups
        unsigned int usedmask = 0u;  // bit k: my k-th row is a pivot row
        const bool gauss = P.osd_method != 3;  // OSD-0 / OSD-E: Gaussian elimination + back-substitution (see header)
        unsigned int anymask = 0u;  // bit k: my k-th row has a non-zero combination mask in some OPEN group (kept as the groups are
                                    // formed, so that neither E1c nor the apply pass's list build reads masks that are zero)
        unsigned int frozenmask = 0u;  // bit k: my k-th row takes no more updates (padding row, or a pivot row that an
                                       // apply pass has brought up to date -- gauss mode only)
#pragma unroll
        for (int k = 0; k < RPT; ++k)
            if ((int)threadIdx.x + k * NT >= m) frozenmask |= 1u << k;
        const unsigned int padmask = frozenmask;
        int nrank = 0;
        int par = 0;
        int ng = 0;                  // open groups
        int wlast = W - 1;           // last word whose panel phase ran
        bool done = false;
        // E3 of the newest group, deferred: only its first word (which the next panel's E1 needs) has been materialised; the rest runs on
        // waves 1-15 while wave 0 searches the next panel's pivots (or at once, where that search takes all waves / an apply pass is due)
        bool e3p = false;
        int e3p_g = 0, e3p_w = 0, e3p_npiv = 0;
        auto e3_rest_all_waves = [&]() {  // (uniform) the deferred part now, on all sixteen waves
            osdl_e3_materialise(U, grow, gnp, TmO, PRO, gbo, M, (int)MRL, W, e3p_w, e3p_g, e3p_npiv, cnz, pcz, e3p_w + 2, W, 0, OSDL_NW);
            __syncthreads();
            if (threadIdx.x < 64) gcnz[e3p_g * 64 + threadIdx.x] = pcz[threadIdx.x];
            e3p = false;
        };

        // AP: apply the ng open groups to words [xlo, W) of every row
        auto apply_open = [&](int xlo) {
            // ---- the rows this pass updates, compacted in ascending row order: everything that is not frozen AND has a
            // non-zero combination mask in some open group.  While the matrix is sparse most rows are untouched by the
            // <= 256 pivots of a pass (a row meets a given 64-column word with probability ~1 %): they are neither
            // read nor written, and their masks are read here once instead of once per chunk.
            __syncthreads();  // U is free (the tables of the previous phase are no longer read)
            int* cnt = (int*)U;  // [RPT * NW + 1]
            // (anymask is kept as the groups are formed: the list build used to read every row's masks in all open groups here)
            const unsigned int touched = anymask & ~frozenmask;  // bit k: my k-th row takes part in this pass
#pragma unroll
            for (int k = 0; k < RPT; ++k) {
                const unsigned long long bal = __ballot(((touched >> k) & 1u) != 0u);
                if (lane == 0) cnt[k * OSDL_NW + wave] = __popcll(bal);
            }
            __syncthreads();
            if (threadIdx.x == 0) {
                int run = 0;
                for (int i = 0; i < RPT * OSDL_NW; ++i) {
                    const int c = cnt[i];
                    cnt[i] = run;
                    run += c;
                }
                cnt[RPT * OSDL_NW] = run;
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < RPT; ++k) {
                const bool act = ((touched >> k) & 1u) != 0u;
                const unsigned long long bal = __ballot(act);
                if (act) alist[cnt[k * OSDL_NW + wave] + __popcll(bal & ((1ull << lane) - 1ull))] = (int)threadIdx.x + k * NT;
            }
            const int nact = cnt[RPT * OSDL_NW];
            const int nk = (nact + NT - 1) / NT;  // list entries per thread (the last round may be partial)
            const int passbits = misc[8];       // mask bits of the open groups (kept as the groups are formed)
            __syncthreads();
            OSDL_TICK(20);
            // sparse masks (the rule in Gaussian mode): walk the set bits, no tables -- osdl_apply_sparse
            const bool sparse_pass = OSDL_SPARSE_LIST > 0 && nact > 0 && xlo < W && passbits <= OSDL_SPARSE_LIST;
            bool sparse_done = false;
            if (sparse_pass) {
                OSDL_TICK(5);
                sparse_done = osdl_apply_sparse(U, alist, TmO, PRO, M, gnp, gbo, (int)MRL, W, xlo, ng, nact, cnz, gcnz) != 0;
                OSDL_TICK(24);
                if (sparse_done) { OSDL_COUNT(12); OSDL_ADD(25, nact); OSDL_ADD(26, passbits); }
            }
            if (!sparse_done)  // the table walk keeps no account of what it fills in: every chunk of a listed row may be non-zero now
                for (int i = threadIdx.x; i < nact; i += NT) cnz[alist[i]] = ~0ull;
            for (int x0 = xlo; x0 < W && nact > 0 && !sparse_done; x0 += OSDL_CW) {
                OSDL_FRESH_TID();
                const int cw = (W - x0) < OSDL_CW ? (W - x0) : OSDL_CW;
                OSDL_TICK(5);
                __syncthreads();  // the previous tables are no longer read
                OSDL_TICK(19);
#ifdef BPOSD_OSD_DIAG
                for (int g = 0; g < ng; ++g) OSDL_ADD(10, (long long)nk * cw * ((gnp[g] + 4) / 5 > 8 ? 13 : ((gnp[g] + 4) / 5 > 4 ? 8 : 4)));
                OSDL_ADD(11, (long long)nk * cw);
#endif
                // 32-entry tables, one per 5 pivots (a 256-byte table is as fast to look up as a 128-byte one --
                // tools/microbench/lds_probe -- and needs 13 instead of 16 look-ups per 64 pivots).  One thread
                // builds one table: 5 pivot-row words in registers, then a Gray-code walk (one XOR + one LDS
                // store per entry).  The walk starts at entry (lane & 31) so that the 32 lanes of a store
                // instruction hit 32 different entries = all 64 banks once.
                // Two threads per table (416 tables of 32 entries, 832 of the 1024 threads): thread 2t + h builds the 16 entries whose
                // top index bit is h, by a 4-bit Gray-code walk.  A 16-lane store group holds 16 different low index nibbles
                // (lane & 15 ^ the walk's code) of 8 tables 256 bytes apart: every bank pair once.
                static_assert(2 * OSDL_K * OSDL_G5 * OSDL_CW <= OSDL_NT, "one pass builds every table");
                if (tid < 2 * ng * OSDL_G5 * OSDL_CW) {
                    const int tt = tid >> 1, h = tid & 1;
                    const int xx = tt & (OSDL_CW - 1), gg = tt >> 3;
                    static_assert(OSDL_CW == 8, "table index decode assumes 8-word chunks");
                    const int g = gg / OSDL_G5, grp = gg - g * OSDL_G5;
                    const int np = gnp[g] - 5 * grp;
                    unsigned long long pr[5];
                    // unconditional (clamped) loads so that the five requests are in flight together
                    const unsigned long long* src = PRO + ((long long)gbo[g] + (xx < cw ? x0 + xx : W - 1)) * 64;
                    unsigned long long val[5];
#pragma unroll
                    for (int kk = 0; kk < 5; ++kk) {
                        const int q = 5 * grp + kk;
                        val[kk] = src[q < 63 ? q : 63];
                    }
                    // (all five values are "used" here: without this the compiler moves each load under the branch of its select and
                    // waits for it there -- five round trips in a row.  With 254 eliminations in flight one round trip costs ~6.6 k
                    // cycles, 17 M of an elimination's ~220 M; bringing the words into LDS a chunk ahead by LDS-direct loads
                    // removed that wait and made the row walks slower by as much or more -- measured twice, not kept.)
                    asm volatile("" : "+v"(val[0]), "+v"(val[1]), "+v"(val[2]), "+v"(val[3]), "+v"(val[4]));
#ifdef BPOSD_OSD_DIAG
                    if (tid == 0) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); OSDL_TICK(21); }
#endif
#pragma unroll
                    for (int kk = 0; kk < 5; ++kk) pr[kk] = (xx < cw && kk < np) ? val[kk] : 0ull;
                    unsigned int idx = osdl_opaque(((unsigned int)lane & 15u) | ((unsigned int)h << 4));  // (laundered: else the indices are hoisted + spilled)
                    unsigned long long v = h ? pr[4] : 0ull;
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk)
                        if ((idx >> kk) & 1u) v ^= pr[kk];
                    unsigned long long* tabp = U + (size_t)tt * 32;
                    tabp[idx] = v;
#pragma unroll
                    for (int i = 1; i < 16; ++i) {
                        const int bit = (i & 1) ? 0 : ((i & 2) ? 1 : ((i & 4) ? 2 : 3));  // ctz(i)
                        idx ^= 1u << bit;
                        v ^= pr[bit];
                        tabp[idx] = v;
                    }
                }
                OSDL_TICK(17);
                __syncthreads();
                OSDL_TICK(18);
                // Round 4: a row's words are touched only where they CHANGE.  The look-ups first form the change (delta) of the
                // row's eight words from the open groups' tables -- that needs the combination masks only --, then exactly the
                // words with a non-zero delta are loaded, XORed and stored (per lane and word).  While the matrix is sparse most
                // (row, word) visits are no-ops -- a listed row selects one or two pivot rows, each non-zero in a few of its
                // 462 words --; before, every listed row's eight words were read and written in every chunk, and with 252
                // eliminations in flight next to the following step's BP kernel the STEP is bound by the memory system's
                // 3.6-4.2 TB/s (BP 255 GB + OSD 486 GB per 178 ms), i.e. by these bytes.  The masks of row k + 1 are still
                // requested before the look-ups of row k start.
                unsigned long long mkn[OSDL_K];
                // list entry of round k: position tid + k * NT; lanes beyond the list's end work on its first row and
                // do not store
                int rown = alist[tid < nact ? tid : 0];
                int rown2 = alist[tid + NT < nact ? tid + NT : 0];
                {
                    const unsigned int ro = osdl_opaque((unsigned int)rown * 8u);
#pragma unroll
                    for (int g = 0; g < OSDL_K; ++g)
                        mkn[g] = (g < ng) ? OSDL_AT(unsigned long long, TmO + (size_t)g * MRL, ro) : 0ull;
                }
#pragma clang loop unroll(disable)
                for (int k = 0; k < nk; ++k) {
                    const bool live = tid + k * NT < nact;
                    const unsigned int ro = osdl_opaque((unsigned int)rown * 8u);
                    unsigned long long v[OSDL_CW], mks[OSDL_K];  // v: the delta of the row's words
#pragma unroll
                    for (int xx = 0; xx < OSDL_CW; ++xx) v[xx] = 0ull;
#pragma unroll
                    for (int g = 0; g < OSDL_K; ++g) mks[g] = mkn[g];
                    if (k + 1 < nk) {
                        rown = rown2;
                        const int p2 = tid + (k + 2) * NT;
                        rown2 = alist[p2 < nact ? p2 : 0];
                        const unsigned int rn = osdl_opaque((unsigned int)rown * 8u);
#pragma unroll
                        for (int g = 0; g < OSDL_K; ++g)
                            mkn[g] = (g < ng) ? OSDL_AT(unsigned long long, TmO + (size_t)g * MRL, rn) : 0ull;
                    }
#pragma unroll
                    for (int g = 0; g < OSDL_K; ++g) {
                        if (g >= ng) break;  // uniform
                        const unsigned long long mk = mks[g];
                        const int ngrp = (gnp[g] + 4) / 5;  // uniform: live 5-bit fields
                        // Half-wave skew: lanes 32-63 walk the fields in the rotated order (grp + 6) mod 13, so the two
                        // halves of a wave read different tables in the same instruction (lds_probe: 3.45 -> 2.95
                        // cycles per wave-level read).  Only when all 13 fields are live.
                        const bool skew = (ngrp == OSDL_G5) && ((lane >> 5) & 1);
                        constexpr int TS = OSDL_CW * 32;  // table stride (elements) between consecutive fields
                        const int up = skew ? 6 * TS : 0, down = skew ? 7 * TS : 0;
                        // volatile LDS pointer: keeps the look-ups as single ds_read_b64 (hipcc would pair them
                        // into ds_read2_b64, which issues at half rate on gfx950 -- same finding as bp_kernel.hip.h;
                        // measured here: 855 -> 677 ms per 254 L29k eliminations)
                        osdl_lds_ptr tb = (osdl_lds_ptr)(U + (size_t)g * OSDL_G5 * TS);
                        // Zero fields need no look-up: a whole group, or a block of 4-5 fields, that is zero in all 64 lanes is
                        // skipped by a wave-uniform test (~8 instructions per block against ~100 for its look-ups).
                        if (__ballot(mk != 0ull) == 0ull) continue;
#pragma unroll
                        for (int qb = 0; qb < 3; ++qb) {
                            if (qb * 4 < ngrp) {  // tables of the fields beyond the group's pivots are zero
                                {
                                    // fields of this block: grp in the block for the straight lanes, (grp + 6) mod 13 for the skewed ones
                                    constexpr unsigned long long LO[3] = {0x00000000000FFFFFull, 0x000000FFFFF00000ull, 0xFFFFFF0000000000ull};
                                    constexpr unsigned long long HI[3] = {0x0003FFFFC0000000ull, 0xFFFC00000000001Full, 0x000000003FFFFFE0ull};
                                    const unsigned long long bm = skew ? HI[qb] : LO[qb];
                                    if (__ballot((mk & bm) != 0ull) == 0ull) continue;
                                }
#pragma unroll
                                for (int grp = qb * 4; grp < (qb == 2 ? OSDL_G5 : qb * 4 + 4); ++grp) {
                                    const int grp2 = grp < 7 ? grp + 6 : grp - 7;  // the skewed lanes' field
                                    const unsigned int f1 = (unsigned int)(mk >> (5 * grp)) & 31u;
                                    const unsigned int f2 = (unsigned int)(mk >> (5 * grp2)) & 31u;
                                    const unsigned int nib = skew ? f2 : f1;
                                    osdl_lds_ptr e = tb + grp * TS + (int)nib + (grp < 7 ? up : -down);
#pragma unroll
                                    for (int xx = 0; xx < OSDL_CW; ++xx) v[xx] ^= e[xx * 32];
                                }
                            }
                        }
                    }
                    // the words that change: loads in flight together, then XOR and store
                    unsigned long long cur[OSDL_CW];
#pragma unroll
                    for (int xx = 0; xx < OSDL_CW; ++xx) {
                        cur[xx] = 0ull;
                        if (xx < cw && live && v[xx] != 0ull) cur[xx] = OSDL_ROW_LD(M + (size_t)(x0 + xx) * MRL, ro);
                    }
                    // The new words are formed for all eight positions in one straight-line block (the positions that do not change
                    // hold 0 ^ 0) and pinned there: with the XOR inside each store's own branch the compiler waits vmcnt(0) before
                    // EVERY store for "its" load -- and on gfx9 that counter counts stores too, so each store waited for the previous
                    // one to complete: up to eight write round trips in a row per row and chunk.
#pragma unroll
                    for (int xx = 0; xx < OSDL_CW; ++xx) cur[xx] ^= v[xx];
                    static_assert(OSDL_CW == 8, "the statement below names every word of a chunk");
                    asm volatile("" : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3]), "+v"(cur[4]), "+v"(cur[5]), "+v"(cur[6]), "+v"(cur[7]));
#pragma unroll
                    for (int xx = 0; xx < OSDL_CW; ++xx)
                        if (xx < cw && live && v[xx] != 0ull) OSDL_ROW_ST(M + (size_t)(x0 + xx) * MRL, ro, cur[xx]);
                }
            }
            if (threadIdx.x == 0) misc[8] = 0;
            __syncthreads();
            ng = 0;
            anymask = 0u;  // no group is open: no masks
            if (gauss) frozenmask = padmask | usedmask;  // no group is open now: every pivot row found so far is complete
        };

#pragma clang loop unroll(disable)
        for (int w = 0; w < W && !done; ++w) {
            wlast = w;
            OSDL_FRESH_TID();
            int npiv = 0;
            const int nb = n - w * 64;
            const unsigned long long vmask = nb >= 64 ? ~0ull : (nb <= 0 ? 0ull : ((1ull << nb) - 1ull));
            if (nb <= 0) done = true;
            // ---------------- E1 tables: XOR combinations of the open groups' pivot rows in word w (16 entries per nibble)
            if (ng > 0) {
                for (int e = tid; e < ng * 256; e += NT) {
                    const int idx = e & 15, grp = (e >> 4) & 15, g = e >> 8;
                    unsigned long long v = 0ull;
                    const int np = gnp[g] - 4 * grp;
                    if (np > 0) {
                        const unsigned long long* pr = PRO + ((long long)gbo[g] + w) * 64 + 4 * grp;  // (grp <= 15: all four words exist)
                        unsigned long long val[4];
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk) val[kk] = pr[kk];
                        asm volatile("" : "+v"(val[0]), "+v"(val[1]), "+v"(val[2]), "+v"(val[3]));  // four loads in flight, not four round trips
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk)
                            if (((idx >> kk) & 1) && kk < np) v ^= val[kk];
                    }
                    U[e] = v;
                }
            }
            // ---------------- E1c + E2c: rows in batches, non-zero panel words into the LDS list.  Gauss-Jordan (OSD-CS) lists
            // the UNUSED rows only -- the pivot search needs no others -- and brings its pivot rows of earlier panels
            // up to date afterwards in one step per row ("Jordan fix-up" below).
            bool compact = false;
            const bool jordan = !gauss;
            {
                constexpr int CAP = OSDL_E2C_CAP;
                unsigned long long* Lpw = U + OSDL_K * 256;          // behind the E1 tables
                unsigned long long* Lt = Lpw + CAP;
                unsigned int* Lid = (unsigned int*)(Lt + CAP);
                unsigned int* Lpiv = Lid + CAP;                      // [128]
                unsigned short* Lnew = (unsigned short*)(Lpiv + 128);  // [1024] bit k: row tid + 1024 k became a pivot row
                unsigned long long* PFW = (unsigned long long*)(Lnew + 1024);  // [64] by column: final panel word of the pivot there
                unsigned long long* TF = PFW + 64;                             // [64] by column: its mask, own bit included
                unsigned int* pcm = (unsigned int*)(TF + 64);                  // [2] pivot columns of this panel
                unsigned short* Lany = (unsigned short*)(pcm + 2);             // [1024] bit k: row tid + 1024 k got a non-zero mask in the new group
                if (tid == 0) { misc[7] = 0; pcm[0] = 0u; pcm[1] = 0u; }
                Lnew[threadIdx.x] = 0;
                Lany[threadIdx.x] = 0;
                __syncthreads();  // tables, counter
                const unsigned int skipmask = frozenmask | (jordan ? usedmask : 0u);  // rows that are not listed
                const unsigned int usedbefore = usedmask;
                // rows whose stored word is non-zero but whose up-to-date word is zero
                const unsigned int zmask = osdl_e1c_list<RPT>(M + (size_t)w * MRL, TmO, U, Lpw, Lid, &misc[7], gnp, (int)MRL, ng, skipmask, anymask, usedmask);
                __syncthreads();
                OSDL_TICK(22);  // (E1c: panel words brought up to date and listed)
                const int nnz = misc[7];
                if (e3p && nnz > OSDL_MW_MIN) e3_rest_all_waves();  // the pivot search will take every wave (or the all-rows form runs)
                if (nnz <= CAP) {  // uniform
                    compact = true;
                    if (nnz > OSDL_MW_MIN) {
                        // ---- long lists (1025 .. OSDL_E2C_CAP entries): ALL sixteen waves, CAP / 1024 entries per lane in registers, one barrier
                        // per pivot (every wave publishes its lowest candidate column and the entry that has it, double-buffered by
                        // parity; the lowest column, lowest wave wins).  ~60 instructions and a barrier per pivot against ~450 from
                        // one wave walking the list in LDS (rounds 2-3's form: 90 M cycles in the slowest elimination of an L29k batch).
                        constexpr int MWR = OSDL_E2C_CAP / OSDL_NT;  // entries per lane
                        unsigned long long cp[MWR], ct[MWR];
                        unsigned int cu = 0u;
#if OSDL_PIVOT_LIGHT
                        unsigned int key0[MWR];  // (pivot rows absorbed in earlier panels << 14) | row, see osdl_e2_compact_wave
                        unsigned int* pkey = (unsigned int*)(Lany + 1024);  // [2][NW] the waves' lightest candidates, behind the lists
#endif
#pragma unroll
                        for (int s2 = 0; s2 < MWR; ++s2) {
                            const int pos = s2 * NT + (int)threadIdx.x;
                            cp[s2] = pos < nnz ? Lpw[pos] : 0ull;
                            ct[s2] = 0ull;
                            if (pos < nnz && (Lid[pos] >> 31)) cu |= 1u << s2;
#if OSDL_PIVOT_LIGHT
                            key0[s2] = pos < nnz ? (Lid[pos] & 0x3fffu) : 0u;
                            const int rpv = rowpos[key0[s2]];
                            key0[s2] |= (rpv < 0 ? (unsigned int)(-1 - rpv) : 0u) << 14;
#endif
                        }
                        int cnp = 0, cnr = nrank;
                        bool cdone = done;
#pragma clang loop unroll(disable)
                        for (;;) {
                            if (cnr >= P.rank) { cdone = true; break; }
                            unsigned long long cand = 0ull;
#pragma unroll
                            for (int s2 = 0; s2 < MWR; ++s2) cand |= ((cu >> s2) & 1u) ? 0ull : cp[s2];
                            cand &= vmask;
                            const unsigned int lb = osd_ffs64_or_64(cand);
                            const unsigned int wk = osd_wave_min_u32((lb << 6) | (unsigned int)lane);  // wave-uniform
                            const int colw = (int)(wk >> 6);
                            int kb = 0;
                            unsigned long long a = 0ull, c = 0ull;
#if OSDL_PIVOT_LIGHT
                            unsigned int bk = ~0u;
#pragma unroll
                            for (int s2 = MWR - 1; s2 >= 0; --s2) {
                                const bool hit = colw < 64 && (((cp[s2] >> (colw & 63)) & 1ull) != 0ull) && (((cu >> s2) & 1u) == 0u);
                                const unsigned int key = key0[s2] + ((unsigned int)__popcll(ct[s2]) << 14);
                                const bool better = hit && key < bk;
                                bk = better ? key : bk;
                                kb = better ? s2 : kb;
                                a = better ? cp[s2] : a;
                                c = better ? ct[s2] : c;
                            }
                            const unsigned int wmin = osd_wave_min_u32(bk);
                            const int firstl = colw < 64 ? (int)__builtin_ctzll(__ballot(bk == wmin)) : 0;
                            if (lane == firstl) pkey[par * OSDL_NW + wave] = wmin;
#else
                            const int firstl = (int)(wk & 63u);
#pragma unroll
                            for (int s2 = MWR - 1; s2 >= 0; --s2) {
                                const bool hit = (((cp[s2] >> (colw & 63)) & 1ull) != 0ull) && (((cu >> s2) & 1u) == 0u);
                                kb = hit ? s2 : kb;
                                a = hit ? cp[s2] : a;
                                c = hit ? ct[s2] : c;
                            }
#endif
                            if (lane == firstl) {
                                pcol[par * OSDL_NW + wave] = (unsigned int)(colw < 64 ? colw : 64);
                                pbuf[(size_t)(par * OSDL_NW + wave) * 2 + 0] = a;
                                pbuf[(size_t)(par * OSDL_NW + wave) * 2 + 1] = c;
                            }
                            __syncthreads();
                            int mincol = 64, wv = 0;
#if OSDL_PIVOT_LIGHT
                            unsigned int minkey = ~0u;  // lowest column, then the lightest row among the waves that propose it
#pragma unroll
                            for (int q = OSDL_NW - 1; q >= 0; --q) {
                                const int pc = (int)pcol[par * OSDL_NW + q];
                                const unsigned int pk = pkey[par * OSDL_NW + q];
                                if (pc < mincol || (pc == mincol && pk <= minkey)) { mincol = pc; minkey = pk; wv = q; }
                            }
#else
#pragma unroll
                            for (int q = OSDL_NW - 1; q >= 0; --q) {
                                const int pc = (int)pcol[par * OSDL_NW + q];
                                if (pc <= mincol) { mincol = pc; wv = q; }
                            }
#endif
                            if (mincol >= 64) { par ^= 1; break; }
                            const unsigned long long pw_p = pbuf[(size_t)(par * OSDL_NW + wv) * 2 + 0];
                            const unsigned long long t_p = pbuf[(size_t)(par * OSDL_NW + wv) * 2 + 1];
                            const unsigned long long tq = t_p ^ (1ull << cnp);
#pragma unroll
                            for (int s2 = 0; s2 < MWR; ++s2) {
                                const unsigned long long mm = (unsigned long long)((long long)(cp[s2] << (63 - mincol)) >> 63);
                                cp[s2] ^= pw_p & mm;
                                ct[s2] ^= tq & mm;
                            }
                            if (wave == wv && lane == firstl) {  // Jordan: the pivot row itself is put back; Gaussian: it is final (see osdl_e2_compact_wave)
                                if (gauss) {
                                    Lpw[kb * NT + (int)threadIdx.x] = pw_p;
                                    Lt[kb * NT + (int)threadIdx.x] = t_p;
                                }
#pragma unroll
                                for (int s2 = 0; s2 < MWR; ++s2)
                                    if (s2 == kb) { cp[s2] = gauss ? 0ull : pw_p; ct[s2] = gauss ? 0ull : t_p; }
                                cu |= 1u << kb;
                                Lpiv[cnp] = (unsigned int)(kb * NT + (int)threadIdx.x);
                                Lpiv[64 + cnp] = (unsigned int)mincol;
                            }
                            ++cnp;
                            ++cnr;
                            par ^= 1;
                        }
#pragma unroll
                        for (int s2 = 0; s2 < MWR; ++s2) {
                            const int pos = s2 * NT + (int)threadIdx.x;
                            if (pos < nnz && !(gauss && ((cu >> s2) & 1u))) {
                                Lpw[pos] = cp[s2];
                                Lt[pos] = ct[s2];
#if OSDL_PIVOT_LIGHT
                                if (ct[s2] != 0ull && ((cu >> s2) & 1u) == 0u) {
                                    const unsigned int cnt = (key0[s2] >> 14) + (unsigned int)__popcll(ct[s2]);
                                    rowpos[key0[s2] & 0x3fffu] = -1 - (int)(cnt < OSDL_CNT_MAX ? cnt : OSDL_CNT_MAX);
                                }
#endif
                            }
                        }
                        if (tid == 0) { misc[4] = cnp; misc[5] = cnr; misc[6] = cdone ? 1 : 0; }
                    } else if (wave == 0) {
                        const unsigned int la = (unsigned int)(size_t)(osdl_lds_w64)Lpw, ma = (unsigned int)(size_t)(osdl_lds_w32)misc;
                        if (gauss) {
                            if (nnz <= 128) osdl_e2_compact_wave<2, true>(la, ma, nnz, vmask, P.rank, nrank, done ? 1 : 0, rowpos);
                            else if (nnz <= 256) osdl_e2_compact_wave<4, true>(la, ma, nnz, vmask, P.rank, nrank, done ? 1 : 0, rowpos);
                            else if (nnz <= 512 || OSDL_MW_MIN < 1024) osdl_e2_compact_wave<8, true>(la, ma, nnz, vmask, P.rank, nrank, done ? 1 : 0, rowpos);
                            else osdl_e2_compact_wave<16, true>(la, ma, nnz, vmask, P.rank, nrank, done ? 1 : 0, rowpos);
                        } else {
                            if (nnz <= 128) osdl_e2_compact_wave<2, false>(la, ma, nnz, vmask, P.rank, nrank, done ? 1 : 0, rowpos);
                            else if (nnz <= 256) osdl_e2_compact_wave<4, false>(la, ma, nnz, vmask, P.rank, nrank, done ? 1 : 0, rowpos);
                            else if (nnz <= 512 || OSDL_MW_MIN < 1024) osdl_e2_compact_wave<8, false>(la, ma, nnz, vmask, P.rank, nrank, done ? 1 : 0, rowpos);
                            else osdl_e2_compact_wave<16, false>(la, ma, nnz, vmask, P.rank, nrank, done ? 1 : 0, rowpos);
                        }
                        OSDL_TICK(2);  // (wave 0's pivot search alone; the rest of tick 23 is the wait for the other waves' share of E3)
                    } else if (e3p) {
                        // waves 1-15: the rest of the previous group's E3 beside wave 0's pivot search (its tables sit behind the lists)
                        osdl_e3_materialise(U, grow, gnp, TmO, PRO, gbo, M, (int)MRL, W, e3p_w, e3p_g, e3p_npiv, cnz, pcz, e3p_w + 2, W, 1, OSDL_NW - 1);
                    }
                    __syncthreads();
                    if (e3p) {  // (uniform) that group is complete now
                        if (threadIdx.x < 64) gcnz[e3p_g * 64 + threadIdx.x] = pcz[threadIdx.x];
                        e3p = false;
                    }
                    OSDL_TICK(23);  // (E2c: the pivot search on the list, one wave)
                    npiv = misc[4];
                    nrank = misc[5];
                    done = misc[6] != 0;
                    // word w is final: the list rows, and zero for the rows that E1 has just cleared
                    for (int i = threadIdx.x; i < nnz; i += NT) M[(size_t)w * MRL + (Lid[i] & 0x7fffffffu)] = Lpw[i];
                    {
                        const unsigned int ro = osdl_opaque((unsigned int)tid * 8u);
#pragma clang loop unroll(disable)
                        for (int k = 0; k < RPT; ++k)
                            if ((zmask >> k) & 1u) OSDL_AT(unsigned long long, M + (size_t)w * MRL, ro + k * NT * 8) = 0ull;
                    }
                    if (npiv > 0) {
                        // the new group's masks: zero for every row that is not frozen, then the list rows' on top
                        {
                            const unsigned int ro = osdl_opaque((unsigned int)tid * 8u);
#pragma clang loop unroll(disable)
                            for (int k = 0; k < RPT; ++k)
                                if (((frozenmask >> k) & 1u) == 0u) OSDL_AT(unsigned long long, TmO + (size_t)ng * MRL, ro + k * NT * 8) = 0ull;
                        }
                        if ((int)threadIdx.x < npiv) {
                            const int q = (int)threadIdx.x;
                            const int r = (int)(Lid[Lpiv[q]] & 0x7fffffffu);
                            const int j = w * 64 + (int)Lpiv[64 + q];
                            pivrow[j] = r;
                            rowpos[r] = j;
                            grow[ng * 64 + q] = r;
                            atomicOr((unsigned int*)(Lnew + ((r & (NT - 1)) & ~1)), (1u << (r / NT)) << (16 * (r & 1)));
                            if (jordan) {
                                const int col = (int)Lpiv[64 + q];
                                PFW[col] = Lpw[Lpiv[q]];
                                TF[col] = Lt[Lpiv[q]] ^ (1ull << q);
                                atomicOr(&pcm[col >> 5], 1u << (col & 31));
                            }
                        }
                        {
                            int mybits = 0;
                            for (int i = threadIdx.x; i < nnz; i += NT)
                                if (Lt[i] != 0ull) {
                                    const int r = (int)(Lid[i] & 0x7fffffffu);
                                    atomicOr((unsigned int*)(Lany + ((r & (NT - 1)) & ~1)), (1u << (r / NT)) << (16 * (r & 1)));
                                    mybits += __popcll(Lt[i]);
                                }
                            // (Gaussian: the new pivot rows are final and will not be listed -- their masks do not count)
                            if (gauss && (int)threadIdx.x < npiv) mybits -= __popcll(Lt[Lpiv[threadIdx.x]]);
                            if (mybits) atomicAdd(&misc[8], mybits);
                        }
                        __syncthreads();  // zero masks before the list rows' masks; new-pivot bits; fix-up tables
                        for (int i = threadIdx.x; i < nnz; i += NT) TmO[(size_t)ng * MRL + (Lid[i] & 0x7fffffffu)] = Lt[i];
                        usedmask |= (unsigned int)Lnew[threadIdx.x];
                        anymask |= (unsigned int)Lany[threadIdx.x];
                        if (gauss) frozenmask |= (unsigned int)Lnew[threadIdx.x];  // Gaussian: a pivot row takes no more updates
                    }
                    if (jordan && usedbefore != 0u) {
                        // ---- Jordan fix-up: a pivot row u of an earlier panel, brought up to date by E1, must lose its ones in
                        // this panel's pivot columns.  With the pivot rows of the panel in their final (mutually reduced)
                        // state F_q that is one step: u ^= XOR F_q over the pivot columns in which u has a one, and its
                        // combination mask is the XOR of the F_q's masks (own bit included) -- the result the sequential
                        // updates reach, since it is the only element of u + span(pivots) without ones in pivot columns.
                        const unsigned long long pcmask = ((unsigned long long)pcm[1] << 32) | pcm[0];
                        const unsigned int ro = osdl_opaque((unsigned int)tid * 8u);
                        constexpr int HB = RPT < 8 ? RPT : 8;
#pragma unroll
                        for (int k0 = 0; k0 < RPT; k0 += HB) {
                            if (((usedbefore >> k0) & ((1u << HB) - 1u)) == 0u) continue;
                            unsigned long long old[HB], v[HB];
#pragma unroll
                            for (int i = 0; i < HB; ++i) {
                                old[i] = ((usedbefore >> (k0 + i)) & 1u) ? OSDL_AT(unsigned long long, M + (size_t)w * MRL, ro + (k0 + i) * NT * 8) : 0ull;
                                v[i] = old[i];
                            }
#pragma unroll
                            for (int g = 0; g < OSDL_K; ++g) {
                                if (g < ng) {
                                    const int ngrp = (gnp[g] + 3) >> 2;
                                    unsigned long long mk[HB];
#pragma unroll
                                    for (int i = 0; i < HB; ++i)
                                        mk[i] = ((usedbefore >> (k0 + i)) & 1u) ? OSDL_AT(unsigned long long, TmO + (size_t)g * MRL, ro + (k0 + i) * NT * 8) : 0ull;
#pragma unroll
                                    for (int i = 0; i < HB; ++i)
                                        for (int grp = 0; grp < ngrp; ++grp) v[i] ^= U[(g * 16 + grp) * 16 + (int)((mk[i] >> (4 * grp)) & 15ull)];
                                }
                            }
#pragma unroll
                            for (int i = 0; i < HB; ++i) {
                                if (((usedbefore >> (k0 + i)) & 1u) == 0u) continue;
                                unsigned long long tm = 0ull;
                                unsigned long long sbits = npiv > 0 ? (v[i] & pcmask) : 0ull;
                                while (sbits) {
                                    const int c = __ffsll((long long)sbits) - 1;
                                    sbits &= sbits - 1;
                                    v[i] ^= PFW[c];
                                    tm ^= TF[c];
                                }
                                if (v[i] != old[i]) OSDL_AT(unsigned long long, M + (size_t)w * MRL, ro + (k0 + i) * NT * 8) = v[i];
                                if (tm != 0ull) {
                                    OSDL_AT(unsigned long long, TmO + (size_t)ng * MRL, ro + (k0 + i) * NT * 8) = tm;
                                    anymask |= 1u << (k0 + i);
                                    atomicAdd(&misc[8], __popcll(tm));
                                }
                            }
                        }
                    }
                }
                __syncthreads();  // the lists are free again (U is reused by E3)
            }
            if (!compact) {  // ======== the all-rows form: a 16-row register window per thread
            unsigned long long pw[RPT], t[RPT];
            {
                const unsigned int ro = osdl_opaque((unsigned int)tid * 8u);
#pragma unroll
                for (int k = 0; k < RPT; ++k) {
                    pw[k] = ((frozenmask >> k) & 1u) ? 0ull : OSDL_AT(unsigned long long, M + (size_t)w * MRL, ro + k * NT * 8);
                    t[k] = 0ull;
                }
                // ---------------- E1: bring word w up to date with the open groups
                if (ng > 0) {
#pragma unroll
                    for (int g = 0; g < OSDL_K; ++g) {
                        if (g < ng) {
                            const int ngrp = (gnp[g] + 3) >> 2;
#pragma unroll
                            for (int k = 0; k < RPT; ++k) {
                                // (a frozen row's mask slots are stale: treat them as empty)
                                const unsigned long long mk = ((frozenmask >> k) & 1u) ? 0ull : OSDL_AT(unsigned long long, TmO + (size_t)g * MRL, ro + k * NT * 8);
                                for (int grp = 0; grp < ngrp; ++grp)
                                    pw[k] ^= U[(g * 16 + grp) * 16 + (int)((mk >> (4 * grp)) & 15ull)];
                            }
                        }
                    }
                }
            }
            OSDL_TICK(2);
            OSDL_FRESH_TID();
            // ---------------- E2: panel phase in registers (all rows)
#pragma clang loop unroll(disable)
            for (;;) {
                if (nrank >= P.rank) { done = true; break; }
                // candidate columns of this lane = OR of its unused rows' panel words; lowest one is proposed
                unsigned long long cand = 0ull;
#pragma unroll
                for (int k = 0; k < RPT; ++k) {
                    const unsigned int um = (unsigned int)__builtin_amdgcn_sbfe((int)usedmask, k, 1);  // -1 if used
                    cand |= pw[k] & ~(((unsigned long long)um << 32) | um);
                }
                cand &= vmask;
                const int lb = cand ? (__ffsll((long long)cand) - 1) : 64;
                unsigned long long act = ~0ull;
                int col = 0;
#pragma unroll
                for (int bitp = 6; bitp >= 0; --bitp) {
                    const unsigned long long z = __ballot(((lb >> bitp) & 1) == 0) & act;
                    if (z) act = z; else col |= (1 << bitp);
                }
                // the proposing lane's row: any unused row of its with a one in column lb (the pivot set does not
                // depend on which row is taken)
                int kb = 0;
#pragma unroll
                for (int k = 0; k < RPT; ++k) {
                    const bool hit = (((pw[k] >> (lb & 63)) & 1ull) != 0ull) && (((usedmask >> k) & 1u) == 0u);
                    kb = hit ? k : kb;
                }
                const int first = __ffsll((long long)act) - 1;
                if (lane == first) {
                    unsigned long long a = 0ull, c = 0ull;
#pragma unroll
                    for (int k = 0; k < RPT; ++k)
                        if (k == kb) { a = pw[k]; c = t[k]; }
                    pcol[par * OSDL_NW + wave] = (unsigned int)col;
                    pbuf[(size_t)(par * OSDL_NW + wave) * 2 + 0] = a;
                    pbuf[(size_t)(par * OSDL_NW + wave) * 2 + 1] = c;
                }
                __syncthreads();
                int mincol = 64, wv = 0;
#pragma unroll
                for (int q = OSDL_NW - 1; q >= 0; --q) {
                    const int pc = (int)pcol[par * OSDL_NW + q];
                    if (pc <= mincol) { mincol = pc; wv = q; }
                }
                if (mincol >= 64) { par ^= 1; break; }
                const unsigned long long pw_p = pbuf[(size_t)(par * OSDL_NW + wv) * 2 + 0];
                const unsigned long long t_p = pbuf[(size_t)(par * OSDL_NW + wv) * 2 + 1];
                const unsigned long long tq = t_p ^ (1ull << npiv);
                const int j = w * 64 + mincol;
                // every row with a one in the pivot column adds the pivot row (branch-free: mask = -bit); the pivot
                // row itself is put back afterwards by the one lane that owns it
#pragma unroll
                for (int k = 0; k < RPT; ++k) {
                    unsigned long long mm = (unsigned long long)((long long)(pw[k] << (63 - mincol)) >> 63);
                    if (gauss) {  // (uniform) Gaussian: the pivot rows of this panel are final
                        const unsigned int um = (unsigned int)__builtin_amdgcn_sbfe((int)usedmask, k, 1);  // -1 if used
                        mm &= ~(((unsigned long long)um << 32) | um);
                    }
                    pw[k] ^= pw_p & mm;
                    t[k] ^= tq & mm;
                }
                if (wave == wv && lane == first) {
#pragma unroll
                    for (int k = 0; k < RPT; ++k)
                        if (k == kb) { pw[k] = pw_p; t[k] = t_p; }
                    usedmask |= 1u << kb;
                    const int r = tid + kb * NT;
                    pivrow[j] = r;
                    rowpos[r] = j;
                    grow[ng * 64 + npiv] = r;
                }
                ++npiv;
                ++nrank;
                par ^= 1;
            }
            OSDL_FRESH_TID();
            // word w is final; the new group's combination masks
            {
                const unsigned int ro = osdl_opaque((unsigned int)tid * 8u);
#pragma unroll
                for (int k = 0; k < RPT; ++k)
                    if (((frozenmask >> k) & 1u) == 0u) OSDL_AT(unsigned long long, M + (size_t)w * MRL, ro + k * NT * 8) = pw[k];
                if (npiv > 0) {
#pragma unroll
                    for (int k = 0; k < RPT; ++k)
                        if (((frozenmask >> k) & 1u) == 0u) {
                            OSDL_AT(unsigned long long, TmO + (size_t)ng * MRL, ro + k * NT * 8) = t[k];
                            if (t[k] != 0ull) anymask |= 1u << k;
                        }
                    {
                        int mybits = 0;
#pragma unroll
                        for (int k = 0; k < RPT; ++k)
                            if (((frozenmask >> k) & 1u) == 0u) mybits += __popcll(t[k]);
                        if (mybits) atomicAdd(&misc[8], mybits);
                    }
#if OSDL_PIVOT_LIGHT
                    // (this form takes whichever candidate row comes first, but keeps the rows' absorbed-pivot counts for the
                    // compact panels that follow: rowpos = -1 - count for an unused row)
#pragma unroll
                    for (int k = 0; k < RPT; ++k)
                        if (((frozenmask >> k) & 1u) == 0u && ((usedmask >> k) & 1u) == 0u && t[k] != 0ull) {
                            const int rp = OSDL_AT(int, rowpos, (ro >> 1) + k * NT * 4) - __popcll(t[k]);
                            OSDL_AT(int, rowpos, (ro >> 1) + k * NT * 4) = rp > -1 - (int)OSDL_CNT_MAX ? rp : -1 - (int)OSDL_CNT_MAX;
                        }
#endif
                }
            }
            if (gauss) frozenmask |= usedmask;  // Gaussian: a pivot row takes no more updates
            }  // ======== end of the all-rows form
            OSDL_TICK(3);
            OSDL_COUNT(7);
            if (npiv > 0) {
                OSDL_COUNT(8);
                if (tid == 0) {
                    gnp[ng] = npiv;
                    gbo[ng] = gauss ? (int)osd_large_pro_base(W, w) : ng * W;
                }
                if (threadIdx.x < 64) pcz[threadIdx.x] = 0ull;
                __syncthreads();  // grow / gnp / TmO of the new group are visible
                if (pmask && (int)threadIdx.x < npiv) {
                    // what the back-substitution needs of this group besides its rows: pivot row q ends as the XOR of the START
                    // states of the group's pivots in its mask (the ones it absorbed inside the panel) and its own
                    const int r = grow[ng * 64 + (int)threadIdx.x];
                    pmask[rowpos[r]] = TmO[(size_t)ng * MRL + r] ^ (1ull << threadIdx.x);
                }
                // ------------ E3: pivot rows of the new group at its start state, for every later word.
                // One wave per word, wave-private tables (a wave's LDS operations complete in order).
                // (Round 5, measured and not kept: (i) in Gaussian mode the new pivot rows are final, so the older open groups can be applied
                // to THEM in M by a sparse pass over just these <= 64 rows, which leaves a plain copy of E3 -- exact, and slower:
                // E3 30 M -> 45 M cycles per elimination, l29k_ms_e15 10.4 k -> 9.6 k syndromes/s.  A sparse pass costs ~4.5 k cycles
                // per 8-word chunk whatever it carries -- barrier, staging three chunks ahead -- times 29 chunks, per group.)
                // (ii) E3 on the set bits of the new pivot rows' masks -- one entry list per group, per word the older groups' pivot
                // words staged in wave-private LDS and XORed into per-slot accumulators by ds_xor_b64, four words in flight: a
                // fifth of the instructions of the table form, exact, and no faster (E3 30 -> 32-36 M cycles): the phase is
                // bound by its gather of 64 scattered row words per word (one cache line each in the word-major matrix, ~4-5
                // cycles per lane through the vector-memory path), which both forms share.
                if (w + 1 < W) {
                    // Round 5: where no apply pass follows at once, only word w + 1 -- which the next panel's E1 tables need -- is
                    // materialised here (wave 0); the rest of the group runs on waves 1-15 beside that panel's one-wave pivot search.
                    const bool defer = OSDL_OVERLAP_E3 && ng + 1 < OSDL_K && w + 2 < W;
                    if (defer) {
                        osdl_e3_materialise(U, grow, gnp, TmO, PRO, gbo, M, (int)MRL, W, w, ng, npiv, cnz, pcz, w + 1, w + 2, 0, 1);
                        e3p = true; e3p_g = ng; e3p_w = w; e3p_npiv = npiv;
                    } else {
                        osdl_e3_materialise(U, grow, gnp, TmO, PRO, gbo, M, (int)MRL, W, w, ng, npiv, cnz, pcz, w + 1, W, 0, OSDL_NW);
                    }
                }
                ++ng;
                __syncthreads();  // PRO of the new group is visible
                if (!e3p && threadIdx.x < 64) gcnz[(ng - 1) * 64 + threadIdx.x] = pcz[threadIdx.x];  // the new pivot rows' chunk maps, for the apply pass
                OSDL_TICK(4);
                if (ng == OSDL_K) {
                    apply_open(w + 1);
                    OSDL_COUNT(9);
                    OSDL_TICK(5);
                }
            }
            __syncthreads();
        }
        if (e3p) e3_rest_all_waves();
        // pending groups (rank reached before the last word): the sweep reads the syndrome column and the
        // non-pivot columns to the right of the last panel in their final state
        if (ng > 0 && wlast + 1 < W) apply_open(wlast + 1);
        OSDL_TICK(5);
        __syncthreads();
        if (P.rank_out && slot_id == 0 && tid == 0) P.rank_out[0] = nrank;

        OSDL_FRESH_TID();
        // --------------------------------------------------------------- OSD-0 solution
        if (tid == 0) { best64[0] = ~0ull; best64[1] = ~0ull; }
        for (int w = wave; w < W; w += OSDL_NW) {
            const int j = w * 64 + lane;
            const unsigned long long np = __ballot(j < n && pivrow[j] < 0);
            if (lane == 0) npmask[w] = np;
        }
        __syncthreads();
        bool y[RPT];
        int ntc_g = 0;  // gauss mode: reduced non-pivot columns available in colvec
        if (gauss) {
            // ---- back-substitution.  Right-hand sides: the first wspan non-pivot columns (unit vectors e_t) and the
            // syndrome column (index MAXSPAN); z_c holds, per sorted column position, the solution found so far plus the
            // right-hand side's own bit.  A pivot row p (pivot column c_p) is e_{c_p} + entries to the right of c_p (it is
            // zero in every earlier column and in the other pivot columns of its own panel), so x_c[c_p] = <row_p, z_c>.
            // Pivot words are solved from the last to the first; the <= 64 pivots of a word are independent.
            const int wspan_g = (P.osd_method >= 2 && P.osd_order > 0) ? (P.osd_order < OSDL_MAXSPAN ? P.osd_order : OSDL_MAXSPAN) : 0;
            if (tid == 0) {
                int a = 0;
                for (int w = 0; w < W && a < wspan_g; ++w) {
                    unsigned long long npm = npmask[w];
                    while (npm && a < wspan_g) {
                        tpos[a++] = w * 64 + (__ffsll((long long)npm) - 1);
                        npm &= npm - 1;
                    }
                }
                misc[1] = a;
            }
            constexpr int NR = OSDL_MAXSPAN + 1;
            unsigned long long* zv = R;            // [NR][W]
            unsigned long long* resw = zv + (size_t)NR * W;  // [NR]
            for (int i = tid; i < NR * W; i += NT) zv[i] = 0ull;
            if (tid < NR) resw[tid] = 0ull;
            __syncthreads();
            ntc_g = misc[1];
            if (tid < ntc_g) zv[(size_t)tid * W + (tpos[tid] >> 6)] = 1ull << (tpos[tid] & 63);
            if (tid == 0) zv[(size_t)OSDL_MAXSPAN * W + (W - 1)] = 1ull << 63;
            __syncthreads();
            // The rows come from PRO, where every group has kept its pivot rows at their START state for the words to the right
            // of its panel, [word][q] -- 64 lanes read 512 contiguous bytes -- and not from M, where a pivot row is one row
            // among 16384 (rounds 1-3: 64 distinct cache lines per wave-level load, 3.3 M of them on this loop's critical
            // path).  Pivot row p ended as XOR of start_l over l in S_p (pmask: the pivots of its own group it absorbed, and
            // itself), so <row_p, z> over the words right of the panel = parity(popcount(S_p & A)) with A_l = <start_l, z>;
            // the panel's own word is final in M and is read from there (one gather per step).
#if !OSDL_BACKSUB_V2
            unsigned long long* resb = resw + NR;  // [NR] the own-word part, by column lane (behind resw: R has room for NR + 1 vectors' tails)
            if (tid < NR) resb[tid] = 0ull;
            __syncthreads();
            // Round 5: a unit right-hand side e_t is zero to the right of t, and so is its solution (a pivot row has no entry to the
            // left of its pivot column): beyond the word of the LAST search column only the syndrome's vector is non-zero, and a word there
            // takes one look-up and two bit operations instead of seventeen and thirty-four.  The search columns are the first non-pivot
            // columns in sorted order, a few words into the matrix; the loop below was bound by those look-ups (each a 512-byte LDS
            // broadcast: ~16 k of a step's ~25-30 k cycles, 14 M cycles per L29k elimination).  xt = -1 without search columns (OSD-0).
            const int xt = ntc_g > 0 ? (tpos[ntc_g - 1] >> 6) : -1;
            OSDL_ADD(27, xt);
#pragma clang loop unroll(disable)
            for (int w = wlast; w >= 0; --w) {
                const int prow = pivrow[w * 64 + lane];  // lane q: the pivot at sorted position 64 w + q, if any
                const unsigned long long pv = __ballot(prow >= 0);
                if (pv == 0ull) continue;  // (uniform: no pivot in this word)
                const int np = __popcll(pv);  // = the group's pivots: slots 0 .. np - 1 of its rows in PRO
                unsigned long long acc[NR];
#pragma unroll
                for (int c = 0; c < NR; ++c) acc[c] = 0ull;
                const unsigned long long* gp = PRO + osd_large_pro_base(W, w) * 64 + lane;
                // the 16 waves split the words to the right; a wave keeps XB of its words in flight (the loads are coalesced now,
                // so that is XB requests, not XB x 64), and the NR solution words of a position are read from LDS together
                // before they are used: as first written the loop waited for every load and every LDS read on its own
                // (47 k cycles per step, 20 M per elimination)
                constexpr int XB = 8;
                for (int x0 = w + 1 + wave; x0 < W; x0 += OSDL_NW * XB) {
                    unsigned long long vx[XB];
#pragma unroll
                    for (int i = 0; i < XB; ++i) {
                        const int x = x0 + i * OSDL_NW;
                        vx[i] = gp[(size_t)(x < W ? x : W - 1) * 64];
                    }
#pragma unroll
                    for (int i = 0; i < XB; ++i) {
                        const int x = x0 + i * OSDL_NW;
                        if (x > xt && x < W) {  // uniform: the syndrome's vector alone
                            const unsigned long long zs = zv[(size_t)OSDL_MAXSPAN * W + x];
                            const unsigned int lo = __builtin_amdgcn_bitop3_b32((unsigned int)acc[OSDL_MAXSPAN], (unsigned int)vx[i], (unsigned int)zs, 0x78);
                            const unsigned int hi = __builtin_amdgcn_bitop3_b32((unsigned int)(acc[OSDL_MAXSPAN] >> 32), (unsigned int)(vx[i] >> 32), (unsigned int)(zs >> 32), 0x78);
                            acc[OSDL_MAXSPAN] = ((unsigned long long)hi << 32) | lo;
                        } else if (x < W) {  // uniform
                            unsigned long long zz[NR];
#pragma unroll
                            for (int c = 0; c < NR; ++c) zz[c] = zv[(size_t)c * W + x];
                            static_assert(NR == 17, "the statement below names every element");
                            asm volatile("" : "+v"(zz[0]), "+v"(zz[1]), "+v"(zz[2]), "+v"(zz[3]), "+v"(zz[4]), "+v"(zz[5]), "+v"(zz[6]), "+v"(zz[7]),
                                         "+v"(zz[8]), "+v"(zz[9]), "+v"(zz[10]), "+v"(zz[11]), "+v"(zz[12]), "+v"(zz[13]), "+v"(zz[14]), "+v"(zz[15]),
                                         "+v"(zz[16]));
                            const unsigned int vlo = (unsigned int)vx[i], vhi = (unsigned int)(vx[i] >> 32);
#pragma unroll
                            for (int c = 0; c < NR; ++c) {  // acc ^= v & z: one v_bitop3 per half (0x78 = a ^ (b & c))
                                const unsigned int lo = __builtin_amdgcn_bitop3_b32((unsigned int)acc[c], vlo, (unsigned int)zz[c], 0x78);
                                const unsigned int hi = __builtin_amdgcn_bitop3_b32((unsigned int)(acc[c] >> 32), vhi, (unsigned int)(zz[c] >> 32), 0x78);
                                acc[c] = ((unsigned long long)hi << 32) | lo;
                            }
                        }
                    }
                }
#pragma unroll
                for (int c = 0; c < NR; ++c) {
                    if (c < ntc_g || c == OSDL_MAXSPAN) {  // uniform
                        const unsigned long long bits = __ballot(lane < np && (__popcll(acc[c]) & 1));
                        if (lane == 0 && bits) atomicXor(&resw[c], bits);
                    }
                }
                // Round 5: the pivot rows of a word are no longer reduced against each other (plain Gaussian elimination, see
                // osdl_e2_compact_wave), so pivot row q has entries at the columns of the word's LATER pivots and the word is solved
                // from its last pivot column to its first: x[j] = parity(S_j & A) ^ <own word of row j, z[w]>, z[w] |= x[j] << j.
                // One wave, right-hand side c on lane c, the row's own word broadcast by v_readlane: <= 64 short dependent steps
                // per word (~1.5 M cycles per elimination) for the Jordan steps the panel phase and the apply passes no longer do.
                unsigned long long vb = 0ull, S = 0ull;
                if (wave == 0 && prow >= 0) {  // (requested before the barrier)
                    vb = M[(size_t)w * MRL + prow];
                    S = pmask[w * 64 + lane];
                }
                __syncthreads();
                if (wave == 0) {
#pragma unroll
                    for (int c = 0; c < NR; ++c) {
                        if (c < ntc_g || c == OSDL_MAXSPAN) {  // bit j: parity(S_j & A_c), the part from the words to the right
                            const unsigned long long pa = __ballot(prow >= 0 && (__popcll(S & resw[c]) & 1) != 0);
                            if (lane == 0) resb[c] = pa;
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                    const bool mine = lane < NR && (lane < ntc_g || lane == OSDL_MAXSPAN);
                    unsigned long long z = mine ? zv[(size_t)lane * W + w] : 0ull;
                    const unsigned long long pa = mine ? resb[lane] : 0ull;
                    unsigned long long pvm = pv;  // wave-uniform
                    while (pvm) {
                        const int j = 63 - __builtin_clzll(pvm);
                        pvm &= ~(1ull << j);
                        const unsigned long long vbj =
                            ((unsigned long long)(unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)(vb >> 32), j) << 32) |
                            (unsigned long long)(unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)vb, j);
                        const unsigned long long bit = ((pa >> j) ^ (unsigned long long)__popcll(vbj & z)) & 1ull;
                        z |= bit << j;  // (no right-hand side has a bit of its own at a pivot column)
                    }
                    if (mine) {
                        zv[(size_t)lane * W + w] = z;
                        resw[lane] = 0ull;
                    }
                }
                __syncthreads();
            }
#else
            {
                const int xt_ = osdl_backsub(zv, resw, tpos, pivrow, M, pmask, PRO, W, (int)MRL, wlast, ntc_g);
                OSDL_ADD(27, xt_);
                (void)xt_;
            }
            __syncthreads();
#endif
            OSDL_TICK(13);
            // reduced columns and reduced syndrome as bit vectors over the pivot ROWS (the layout the sweep below uses)
#pragma unroll
            for (int k = 0; k < RPT; ++k) {
                const bool usedk = (usedmask >> k) & 1u;
                const int r = tid + k * NT;
                const int j = usedk ? rowpos[r] : 0;
                for (int a = 0; a < ntc_g; ++a) {
                    const bool bit = usedk && ((zv[(size_t)a * W + (j >> 6)] >> (j & 63)) & 1ull);
                    const unsigned long long cb = __ballot(bit);
                    if (lane == 0) colvec[a * NCV + k * OSDL_NW + wave] = cb;
                    if (am && bit) am[r] |= (unsigned short)(1u << a);
                }
                y[k] = usedk && ((zv[(size_t)OSDL_MAXSPAN * W + (j >> 6)] >> (j & 63)) & 1ull);
            }
        } else {
            const unsigned int ro_y = osdl_opaque((unsigned int)tid * 8u);
#pragma unroll
            for (int k = 0; k < RPT; ++k)
                y[k] = ((OSDL_AT(unsigned long long, M + (size_t)(W - 1) * MRL, ro_y + k * NT * 8) >> 63) & 1ull) != 0ull;
        }
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            const bool usedk = (usedmask >> k) & 1u;
            const unsigned long long yb = __ballot(usedk && y[k]);
            if (lane == 0) yvec[k * OSDL_NW + wave] = yb;
            if (usedk && y[k]) xout[kidx[rowpos[tid + k * NT]]] = 1;
        }
        __syncthreads();
        osd_store_row(P.out_osd0, P.packed_io, (size_t)s, n, xout, tid, NT);
        osd_store_row(P.cmp_osd0, P.packed_io, (size_t)slot_id, n, xout, tid, NT);
        OSDL_TICK(14);
        int w0 = 0;
        for (int q = 0; q < NCV; ++q) w0 += __popcll(yvec[q]);

        int sel_a = -1, sel_b = -1;
        if (P.osd_method >= 2 && P.osd_order > 0) {
            // pair span of osd_cs with integer weights: up to 64 columns (/root/reference/examples/qldpc_decode_example.py:16
            // passes 42); beyond 16 the reduced columns go to a global workspace instead of LDS
            const int span_cap = (P.osd_method == 3 && P.colvec_ws && (!P.cost || am64)) ? OSDL_MAXSPAN_CS : OSDL_MAXSPAN;
            const int wspan = P.osd_order < span_cap ? P.osd_order : span_cap;
            unsigned long long* colv = wspan > OSDL_MAXSPAN ? P.colvec_ws + (size_t)blockIdx.x * OSDL_MAXSPAN_CS * NCV : colvec;
            int tcount = gauss ? ntc_g : 0;  // gauss mode: colvec / tpos / am come from the back-substitution
#pragma clang loop unroll(disable)
            for (int w = gauss ? W : 0; w < W; ++w) {
                unsigned long long npm = npmask[w];
                npm = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(npm >> 32)) << 32) |
                      (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)npm);
                if (!npm) continue;
                if (P.osd_method != 3 && tcount >= wspan) break;
                OSDL_FRESH_TID();  // osd_e only needs the first wspan columns
                unsigned long long rw[RPT];
                const unsigned int ro = osdl_opaque((unsigned int)tid * 8u);
#pragma unroll
                for (int k = 0; k < RPT; ++k) rw[k] = OSDL_AT(unsigned long long, M + (size_t)w * MRL, ro + k * NT * 8);
                int acc = 0;
                while (npm) {
                    const int b = __ffsll((long long)npm) - 1;
                    npm &= npm - 1;
                    const unsigned long long bmask = 1ull << b;
                    int cnt = 0;
#pragma unroll
                    for (int k = 0; k < RPT; ++k) {
                        const bool usedk = (usedmask >> k) & 1u;
                        const bool bitv = (rw[k] & bmask) != 0ull;
                        if (P.osd_method == 3) cnt += __popcll(__ballot(usedk && (bitv != y[k])));
                        if (tcount < wspan) {
                            const unsigned long long cb = __ballot(usedk && bitv);
                            if (lane == 0) colv[tcount * NCV + k * OSDL_NW + wave] = cb;
                            if (am64 && usedk && bitv) am64[tid + k * NT] |= 1ull << tcount;  // only the row's owner
                            else if (am && usedk && bitv && tcount < 16) am[tid + k * NT] |= (unsigned short)(1u << tcount);
                        }
                    }
                    if (lane == b) acc += cnt;
                    if (tid == 0 && tcount < 64) tpos[tcount] = w * 64 + b;
                    ++tcount;
                }
                if (P.osd_method == 3 && acc) atomicAdd(&wt[w * 64 + lane], acc);
            }
            __syncthreads();
            if (P.cost) {
                // ================= fp64 weights (non-uniform channel): sum of log(1/p_i) over the candidate's set bits in
                // ascending ORIGINAL bit index, one candidate per lane (bit-identical to the CPU's sums, hence the same
                // strict-< winner).  Per original bit i one LDS word: bits 0-15 its entries in the first wspan non-pivot
                // columns (for a pivot bit: of its pivot row; for one of those columns itself: its own bit), bits 16-29
                // its pivot row, bit 30 "is a pivot", bit 31 its OSD-0 value.
                const int ntc = tcount < wspan ? tcount : wspan;
                unsigned int* info = (unsigned int*)R;  // (the back-substitution vectors are dead by now)
                double* pairw = (double*)(info + ((n + 1) & ~1));
                double* costs = P.costs_ws + (size_t)blockIdx.x * n;
                double* wd = P.wd_ws + (size_t)blockIdx.x * P.wdn;
                unsigned long long* cm64 = am64 ? P.cm64_ws + (size_t)blockIdx.x * n : nullptr;
                if (cm64) pairw = wd + 32768;  // up to 2016 pair weights: behind the singles' (64 W <= 32768 entries of wd's 65536)
                OSDL_FRESH_TID();
                for (int i = tid; i < n; i += NT) {
                    double ci = P.cost[i];
                    if (P.sel) {  // per-syndrome two-valued channel (css_decode_sim.py:207-248)
                        const double ca = P.cost_alt[i];
                        if (P.sel[(size_t)s * n + i] != 0) ci = ca;
                    }
                    costs[i] = ci;
                    const int pos = inv[i];
                    const int pr = pivrow[pos];
                    unsigned int e = (unsigned int)xout[i] << 31;
                    if (cm64) {  // wide pair span: the column entries live in a 64-bit word per bit in HBM
                        unsigned long long cmv = 0ull;
                        if (pr >= 0) { e |= (1u << 30) | ((unsigned int)pr << 16); cmv = am64[pr]; }
                        else
                            for (int a = 0; a < ntc; ++a)
                                if (tpos[a] == pos) cmv |= 1ull << a;
                        cm64[i] = cmv;
                    } else if (pr >= 0) e |= (1u << 30) | ((unsigned int)pr << 16) | am[pr];
                    else
                        for (int a = 0; a < ntc; ++a)
                            if (tpos[a] == pos) e |= 1u << a;
                    info[i] = e;
                }
                int* besti = misc + 2;
                if (tid == 0) { besti[0] = 0x7fffffff; besti[1] = 0x7fffffff; }
                __syncthreads();
                double w0d = 0.0;  // weight of OSD-0 (every lane, uniform reads)
                for (int i = 0; i < n; ++i)
                    if (info[i] >> 31) w0d += costs[i];
                if (P.osd_method == 3) {
                    // ---- singles: lane c of the wave that owns word w evaluates the column at sorted position 64 w + c
                    for (int w = wave; w < W; w += OSDL_NW) {
                        const int j = w * 64 + lane;
                        const int own = j < n ? kidx[j] : -1;  // the candidate's own (non-pivot) bit
                        const unsigned long long* Mw = M + (size_t)w * MRL;
                        double acc = 0.0;
#pragma unroll 4
                        for (int i = 0; i < n; ++i) {
                            const unsigned int e = info[i];
                            bool xi;
                            if (e & (1u << 30)) {  // uniform: pivot bit, its row's word w is the same for the whole wave
                                const unsigned long long word = Mw[(e >> 16) & 0x3fffu];
                                xi = (((unsigned int)(word >> lane) ^ (e >> 31)) & 1u) != 0u;
                            } else {
                                xi = (i == own);
                            }
                            if (xi) acc += costs[i];
                        }
                        if (j < 64 * W) wd[j] = acc;
                        if (j < n && pivrow[j] < 0) atomicMin(&best64[0], (unsigned long long)__double_as_longlong(acc));
                    }
                    // ---- pairs (a < b < wspan)
                    const int npairs = ntc * (ntc - 1) / 2;
                    for (int pidx = tid; pidx < npairs; pidx += NT) {
                        int a = 0, rem = pidx;
                        while (rem >= ntc - 1 - a) { rem -= ntc - 1 - a; ++a; }
                        double acc = 0.0;
                        if (cm64) {
                            const unsigned long long pat = (1ull << a) | (1ull << (a + 1 + rem));
                            for (int i = 0; i < n; ++i)
                                if (((info[i] >> 31) ^ (unsigned int)__popcll(cm64[i] & pat)) & 1u) acc += costs[i];
                        } else {
                            const unsigned int pat = (1u << a) | (1u << (a + 1 + rem));
                            for (int i = 0; i < n; ++i) {
                                const unsigned int e = info[i];
                                if (((e >> 31) ^ (unsigned int)__popc(e & pat)) & 1u) acc += costs[i];
                            }
                        }
                        pairw[pidx] = acc;
                        atomicMin(&best64[1], (unsigned long long)__double_as_longlong(acc));
                    }
                    __syncthreads();
                    const unsigned long long ms = best64[0], mp = best64[1];
                    for (int j = tid; j < n; j += NT)
                        if (pivrow[j] < 0 && (unsigned long long)__double_as_longlong(wd[j]) == ms) atomicMin(&besti[0], j);
                    for (int pidx = tid; pidx < npairs; pidx += NT)
                        if ((unsigned long long)__double_as_longlong(pairw[pidx]) == mp) atomicMin(&besti[1], pidx);
                    __syncthreads();
                    double best = w0d;
                    if (ms != ~0ull && __longlong_as_double((long long)ms) < best) {
                        best = __longlong_as_double((long long)ms);
                        sel_a = besti[0];
                        sel_b = -1;
                    }
                    if (mp != ~0ull && __longlong_as_double((long long)mp) < best) {
                        best = __longlong_as_double((long long)mp);
                        int pidx = besti[1], a = 0, rem = pidx;
                        while (rem >= ntc - 1 - a) { rem -= ntc - 1 - a; ++a; }
                        sel_a = tpos[a];
                        sel_b = tpos[a + 1 + rem];
                    }
                } else {
                    // ---- osd_e: patterns 1 .. 2^w - 1 on the first w non-pivot columns
                    const unsigned int npat = (1u << ntc) - 1u;
                    for (unsigned int pat = tid + 1; pat <= npat; pat += NT) {
                        double acc = 0.0;
                        for (int i = 0; i < n; ++i) {
                            const unsigned int e = info[i];
                            if (((e >> 31) ^ (unsigned int)__popc(e & pat)) & 1u) acc += costs[i];
                        }
                        wd[pat] = acc;
                        atomicMin(&best64[0], (unsigned long long)__double_as_longlong(acc));
                    }
                    __syncthreads();
                    const unsigned long long ms = best64[0];
                    for (unsigned int pat = tid + 1; pat <= npat; pat += NT)
                        if ((unsigned long long)__double_as_longlong(wd[pat]) == ms) atomicMin(&besti[0], (int)osd_e_index(pat, ntc, P.e_msb_first));
                    __syncthreads();
                    if (ms != ~0ull && __longlong_as_double((long long)ms) < w0d) {
                        sel_a = -2;
                        sel_b = (int)osd_e_index((unsigned int)besti[0], ntc, P.e_msb_first);
                    }
                }
            } else if (P.osd_method == 3) {
                for (int j = tid; j < n; j += NT)
                    if (pivrow[j] < 0) atomicMin(&best64[0], ((unsigned long long)wt[j] << 32) | (unsigned)j);
                const int npairs = wspan * (wspan - 1) / 2;
                for (int pidx = tid; pidx < npairs; pidx += NT) {
                    int a = 0, rem = pidx;
                    while (rem >= wspan - 1 - a) { rem -= wspan - 1 - a; ++a; }
                    const int bq = a + 1 + rem;
                    int wgt = 2;
                    for (int q = 0; q < NCV; ++q) wgt += __popcll(yvec[q] ^ colv[a * NCV + q] ^ colv[bq * NCV + q]);
                    atomicMin(&best64[1], ((unsigned long long)wgt << 32) | (unsigned)pidx);
                }
                __syncthreads();
                int bestw = w0;
                const unsigned long long k1 = best64[0], k2 = best64[1];
                if (k1 != ~0ull && (int)(k1 >> 32) < bestw) { bestw = (int)(k1 >> 32); sel_a = (int)(k1 & 0xffffffffu); sel_b = -1; }
                if (k2 != ~0ull && (int)(k2 >> 32) < bestw) {
                    bestw = (int)(k2 >> 32);
                    int pidx = (int)(k2 & 0xffffffffu), a = 0, rem = pidx;
                    while (rem >= wspan - 1 - a) { rem -= wspan - 1 - a; ++a; }
                    sel_a = tpos[a];
                    sel_b = tpos[a + 1 + rem];
                }
            } else {
                const unsigned int npat = (1u << wspan) - 1u;
                if (wspan >= 4) {
                    // Gray-code walk: a thread takes aligned runs of 16 consecutive Gray indices g (pattern = g ^ (g >> 1));
                    // inside a run the pattern changes by ONE column per step -- column ctz(j) at step j, the same for
                    // every run -- so a candidate costs one LDS word per row word instead of one per set bit (the plain
                    // loop read ~8.5 words per row word; 14.8 M of an elimination's ~200 M cycles at order 15).  The
                    // weights are the same integers, the winner is the same (lowest enumeration index among the lightest);
                    // pattern 0 (OSD-0 itself) takes part with weight w0 and can never win the strict comparison.
                    const unsigned int nruns = (npat + 1u) >> 4;
                    unsigned long long mykey = ~0ull;
                    for (unsigned int run = tid; run < nruns; run += NT) {
                        const unsigned int g0 = run << 4;
                        const unsigned int p0 = g0 ^ (g0 >> 1);
                        int wg[16];
#pragma unroll
                        for (int j = 0; j < 16; ++j) wg[j] = 0;
                        for (int q = 0; q < NCV; ++q) {
                            unsigned long long v = yvec[q];
                            unsigned int pp = p0;
                            while (pp) {
                                const int bq = __ffs((int)pp) - 1;
                                pp &= pp - 1;
                                v ^= colvec[bq * NCV + q];
                            }
                            wg[0] += __popcll(v);
#pragma unroll
                            for (int j = 1; j < 16; ++j) {
                                v ^= colvec[(__builtin_ctz(j)) * NCV + q];
                                wg[j] += __popcll(v);
                            }
                        }
#pragma unroll
                        for (int j = 0; j < 16; ++j) {
                            const unsigned int g = g0 + j, pat = g ^ (g >> 1);
                            const int wgt = wg[j] + __popc(pat);
                            const unsigned long long key = ((unsigned long long)wgt << 32) | osd_e_index(pat, wspan, P.e_msb_first);
                            mykey = key < mykey ? key : mykey;
                        }
                    }
                    if (mykey != ~0ull) atomicMin(&best64[0], mykey);
                } else
                for (unsigned int pat = tid + 1; pat <= npat; pat += NT) {
                    int wgt = __popc(pat);
                    for (int q = 0; q < NCV; ++q) {
                        unsigned long long v = yvec[q];
                        unsigned int pp = pat;
                        while (pp) {
                            const int bq = __ffs((int)pp) - 1;
                            pp &= pp - 1;
                            v ^= colvec[bq * NCV + q];
                        }
                        wgt += __popcll(v);
                    }
                    atomicMin(&best64[0], ((unsigned long long)wgt << 32) | osd_e_index(pat, wspan, P.e_msb_first));
                }
                __syncthreads();
                const unsigned long long k1 = best64[0];
                if (k1 != ~0ull && (int)(k1 >> 32) < w0) { sel_a = -2; sel_b = (int)osd_e_index((unsigned int)(k1 & 0xffffffffu), wspan, P.e_msb_first); }
            }
        }

        OSDL_FRESH_TID();
        OSDL_TICK(15);
        // ------------------------------------------------- write the OSD-W solution
        if (sel_a == -1) {
            osd_store_row(P.out_osdw, P.packed_io, (size_t)s, n, xout, tid, NT);
            osd_store_row(P.cmp_osdw, P.packed_io, (size_t)slot_id, n, xout, tid, NT);
        } else {
            __syncthreads();
            for (int i = tid; i < n; i += NT) xout[i] = 0;
            __syncthreads();
#pragma unroll
            for (int k = 0; k < RPT; ++k) {
                bool xs = y[k];
                const size_t rr = (size_t)tid + (size_t)k * NT;
                const unsigned int ro = osdl_opaque((unsigned int)tid * 8u) + k * NT * 8;
                if (sel_a >= 0) {
                    xs ^= ((OSDL_AT(unsigned long long, M + (size_t)(sel_a >> 6) * MRL, ro) >> (sel_a & 63)) & 1ull) != 0ull;
                    if (sel_b >= 0) xs ^= ((OSDL_AT(unsigned long long, M + (size_t)(sel_b >> 6) * MRL, ro) >> (sel_b & 63)) & 1ull) != 0ull;
                } else {
                    unsigned int pp = (unsigned int)sel_b;
                    while (pp) {
                        const int bq = __ffs((int)pp) - 1;
                        pp &= pp - 1;
                        const int pos = tpos[bq];
                        if (gauss) xs ^= ((colvec[bq * NCV + k * OSDL_NW + wave] >> lane) & 1ull) != 0ull;  // M is not reduced
                        else xs ^= ((OSDL_AT(unsigned long long, M + (size_t)(pos >> 6) * MRL, ro) >> (pos & 63)) & 1ull) != 0ull;
                    }
                }
                if (((usedmask >> k) & 1u) && xs) xout[kidx[rowpos[rr]]] = 1;
            }
            if (tid == 0) {
                if (sel_a >= 0) {
                    xout[kidx[sel_a]] = 1;
                    if (sel_b >= 0) xout[kidx[sel_b]] = 1;
                } else {
                    unsigned int pp = (unsigned int)sel_b;
                    while (pp) {
                        const int bq = __ffs((int)pp) - 1;
                        pp &= pp - 1;
                        xout[kidx[tpos[bq]]] = 1;
                    }
                }
            }
            __syncthreads();
            osd_store_row(P.out_osdw, P.packed_io, (size_t)s, n, xout, tid, NT);
            osd_store_row(P.cmp_osdw, P.packed_io, (size_t)slot_id, n, xout, tid, NT);
        }
        __syncthreads();
        OSDL_TICK(16);
#ifdef BPOSD_OSD_DIAG
        tk[6] = tk[13] + tk[14] + tk[15] + tk[16];
        if (P.dbg && slot_id == 0 && tid == 0)
            for (int i = 0; i < 28; ++i) P.dbg[i] = tk[i];
        if (P.dbg && slot_id < 500 && tid == 0) {  // every elimination of the launch: total ticks, then the stamps 0..12 + 17..20
            long long tot = 0;
            for (int i = 0; i < 7; ++i) tot += tk[i];
            tot += tk[17] + tk[18] + tk[19] + tk[20] + tk[21] + tk[22] + tk[23] + tk[24];
            long long* d = P.dbg + 32 + slot_id * 16;
            d[0] = tot;
            for (int i = 0; i < 11; ++i) d[1 + i] = tk[i];
            d[12] = tk[17] + tk[21]; d[13] = tk[22]; d[14] = tk[23]; d[15] = tk[24];  // (15: sparse apply passes; was the builders' loads)
        }
#endif
#undef OSDL_TICK
#undef OSDL_COUNT
#undef OSDL_ADD
#undef OSDL_FRESH_TID
    }
}

}  // namespace bposd
