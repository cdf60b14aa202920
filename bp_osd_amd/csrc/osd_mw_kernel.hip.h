// osd_mw_kernel.hip.h -- OSD-0 / OSD-E / OSD-CS for MID-SIZE codes: NWV waves per non-converged syndrome, rows in registers.
//
// Between the one-wave kernel (osd_wave_kernel.hip.h, up to 320 checks: beyond that its rows no longer fit one wave's
// registers -- the 7 x 15 instance held 256 VGPRs + 51 AGPRs at one wave per SIMD and lost) and the codes osd_kernel.hip.h
// was built around (H1922: 961 checks, a whole CU per elimination, a blocked elimination with four-Russians tables) sit the
// reference's own [[900,36,10]] code (432 checks), surface codes of distance 19 ... 25 (342 ... 600 checks) and their like.
// osd_kernel gives such an elimination four waves that meet at a barrier per pivot AND at 218 VGPRs keeps only two
// workgroups per CU.  Here a workgroup of NWV waves owns the elimination; a lane owns RPL rows (row = 64 (wave RPL + q) +
// lane) as W 64-bit words in registers, in reliability-sorted column order with the syndrome as the last bit -- the one-wave
// kernel's plain Gauss-Jordan, column by column:
//   * every wave looks for an unused row with the column set among its own rows (one ballot per row slot) and PUBLISHES its
//     candidate row speculatively in its own LDS slot, with a flag; ONE workgroup barrier (NWV waves) per column; the
//     lowest wave with a candidate wins (any unused row with the column set is a valid pivot: the reduced system does not
//     depend on the choice); everybody reads the winner's row from LDS (a broadcast read) and XORs it into the rows that
//     have the column set, under the execution mask of those rows.  The slots are double-buffered by column parity.
//   * sort, OSD-0, the candidate sweep and the output stage are the one-wave kernel's, with workgroup barriers: the single
//     candidates' weights are summed over the waves through LDS atomics, pairs / OSD-E patterns run one candidate per thread
//     over the first w reduced columns kept in LDS.
// Several workgroups per CU (registers: RPL W 2 + ~50 VGPRs).  Integer weights only (uniform channel); a non-uniform /
// per-shot channel and osd_e orders above 12 stay on osd_kernel.hip.h.  Identical results to it and to the oracle
// (tests/test_gpu_parity.py::test_osd_wave_kernel_equals_workgroup_kernel_and_oracle).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "osd_wave_kernel.hip.h"

namespace bposd {

constexpr int OSDM_MAXSPAN = 64;  // osd_cs pair span / osd_e order kept as reduced columns in LDS

__host__ __device__ constexpr size_t osdm_lds_bytes(int ns, int nwv, int rpl, int w) {
    // keys u64 [ns] | xbuf u64 [2][nwv][w] | colvec u64 [MAXSPAN][nwv rpl] | yv, y0 u64 [nwv rpl] each | best u64 [2] | npmask u64 [w]
    // | wsum u32 [ns] | xflag u32 [2][nwv] | misc u32 [8] | part u32 [nwv] | kidx, inv, pivrow, tpos u16 [ns] each
    return (size_t)ns * 8 + (size_t)2 * nwv * w * 8 + (size_t)OSDM_MAXSPAN * nwv * rpl * 8 + (size_t)2 * nwv * rpl * 8 + 16 + (size_t)w * 8 +
           (size_t)ns * 4 + (size_t)2 * nwv * 4 + 32 + (size_t)nwv * 4 + (size_t)ns * 2 * 4 + 64;
}

// NWV: waves per elimination (= per workgroup);  RPL: rows per lane (m <= 64 NWV RPL);  W: 64-bit words per row (n + 1 <= 64 W)
// MINW: waves per SIMD the register allocation must admit
// PACKED: the packed-I/O form (OsdParams::packed_io) as a compile-time switch
template <int NWV, int RPL, int W, int MINW, bool PACKED = false>
__global__ __launch_bounds__(64 * NWV, MINW) void osd_mw_kernel(const OsdParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NT = 64 * NWV;
    constexpr int NSL = NWV * RPL;  // row slots of 64 rows
    const int m = P.m, n = P.n;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NS = osdw_nsort(n);
    unsigned char* p = smem;
    unsigned long long* keys = (unsigned long long*)p; p += (size_t)NS * 8;
    unsigned long long* xbuf = (unsigned long long*)p; p += (size_t)2 * NWV * W * 8;   // per column parity and wave: the candidate row
    unsigned long long* colvec = (unsigned long long*)p; p += (size_t)OSDM_MAXSPAN * NSL * 8;
    unsigned long long* yv = (unsigned long long*)p; p += (size_t)NSL * 8;      // the solution being written, by row slot
    unsigned long long* y0 = (unsigned long long*)p; p += (size_t)NSL * 8;      // OSD-0's reduced syndrome, by row slot
    unsigned long long* best = (unsigned long long*)p; p += 16;
    unsigned long long* npmask = (unsigned long long*)p; p += (size_t)W * 8;
    unsigned int* wsum = (unsigned int*)p; p += (size_t)NS * 4;                 // weight of the a-th single candidate
    unsigned int* xflag = (unsigned int*)p; p += (size_t)2 * NWV * 4;
    unsigned int* misc = (unsigned int*)p; p += 32;
    unsigned int* part = (unsigned int*)p; p += (size_t)NWV * 4;
    unsigned short* kidx = (unsigned short*)p; p += (size_t)NS * 2;   // sorted position -> original bit
    unsigned short* inv = (unsigned short*)p; p += (size_t)NS * 2;    // original bit -> sorted position
    short* pivrow = (short*)p; p += (size_t)NS * 2;                   // sorted position -> pivot row (64 * slot + lane), -1 = none
    unsigned short* tpos = (unsigned short*)p;                        // a-th non-pivot column (sorted position)

    for (;;) {
        if (tid == 0) misc[0] = (unsigned int)atomicAdd(&P.counters[2], 1);
        __syncthreads();
        const int slot_id = (int)misc[0];
        const int nlist = P.counters[1];
        if (slot_id >= nlist) break;  // workgroup-uniform
        const long long s = P.osd_list[slot_id];
        const double* llr = P.llr_ws + (size_t)slot_id * n;

        // ------------------------------------------------------------------ a8: sort (same order as osd_kernel's)
        for (int i = tid; i < NS; i += NT) {
            if (i < n) {
                keys[i] = llr_sort_key(llr[i]);
                kidx[i] = (unsigned short)(P.tie_policy == 1 ? n - 1 - i : i);
            } else {
                keys[i] = ~0ull;
                kidx[i] = (unsigned short)i;
            }
            pivrow[i] = -1;
            wsum[i] = 1u;
        }
        __syncthreads();
        for (int k = 2; k <= NS; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int t = tid; t < (NS >> 1); t += NT) {
                    const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                    const int hi = lo | j;
                    const bool up = ((lo & k) == 0);
                    const unsigned long long ka = keys[lo], kb = keys[hi];
                    const unsigned short ia = kidx[lo], ib = kidx[hi];
                    const bool a_gt_b = (ka > kb) || (ka == kb && ia > ib);
                    if (a_gt_b == up) {
                        keys[lo] = kb; keys[hi] = ka;
                        kidx[lo] = ib; kidx[hi] = ia;
                    }
                }
                __syncthreads();
            }
        }
        if (P.tie_policy == 1) {
            for (int i = tid; i < n; i += NT) kidx[i] = (unsigned short)(n - 1 - kidx[i]);
            __syncthreads();
        }
        for (int j = tid; j < n; j += NT) inv[kidx[j]] = (unsigned short)j;
        __syncthreads();

        // ------------------------------------------- my rows in sorted column order, syndrome = bit 63 of the last word
        unsigned long long row[RPL][W];
#pragma unroll
        for (int q = 0; q < RPL; ++q) {
#pragma unroll
            for (int w = 0; w < W; ++w) row[q][w] = 0ull;
            const int r = 64 * (wave * RPL + q) + lane;
            if (r < m) {
                const int e0 = P.rp[r], e1 = P.rp[r + 1];
                for (int e = e0; e < e1; ++e) {
                    const int j = inv[P.ci[e]];
                    const unsigned long long bit = 1ull << (j & 63);
#pragma unroll
                    for (int w = 0; w < W; ++w) row[q][w] |= (w == (j >> 6)) ? bit : 0ull;
                }
                if (osd_synd_bit(P.synd, PACKED ? 1 : 0, s, m, r)) row[q][W - 1] |= 1ull << 63;
            }
        }

        // ------------------------------------------------------------------ a9: Gauss-Jordan in sorted column order
        unsigned int usedm = 0u;  // bit q: my row of slot q is a pivot row
        int nrank = 0;            // (the same in every wave: it follows the shared flags)
        int par = 0;
#pragma unroll
        for (int w = 0; w < W; ++w) {
            const int nb = (n - 64 * w) < 64 ? (n - 64 * w) : 64;  // columns of this word (the syndrome bit is not one)
            for (int b = 0; b < nb && nrank < P.rank; ++b) {
                const unsigned long long mask = 1ull << b;
                // my wave's candidate: an unused row of mine with the column set
                bool found = false;
                int psrc = 0, pslot = 0;
#pragma unroll
                for (int q = 0; q < RPL; ++q) {
                    if (!found) {
                        const unsigned long long bal = __ballot((row[q][w] & mask) != 0ull && ((usedm >> q) & 1u) == 0u);
                        if (bal) {  // wave-uniform
                            found = true;
                            psrc = __ffsll((long long)bal) - 1;
                            pslot = q;
                        }
                    }
                }
                unsigned long long* xb = xbuf + (size_t)(par * NWV + wave) * W;
                if (found) {  // published speculatively: whether it is THE pivot row is known after the barrier
#pragma unroll
                    for (int q = 0; q < RPL; ++q) {
                        if (q == pslot) {  // wave-uniform
                            if (lane == psrc) {
#pragma unroll
                                for (int x = w; x < W; ++x) xb[x] = row[q][x];  // (words left of the panel: see below)
                            }
                        }
                    }
                }
                if (lane == 0) xflag[par * NWV + wave] = found ? 1u : 0u;
                __syncthreads();
                int win = -1;
#pragma unroll
                for (int v = NWV - 1; v >= 0; --v)
                    if (xflag[par * NWV + v]) win = v;
                win = __builtin_amdgcn_readfirstlane(win);
                const unsigned long long* pw = xbuf + (size_t)(par * NWV + (win < 0 ? 0 : win)) * W;
                par ^= 1;
                if (win < 0) continue;  // no unused row has this column set: a non-pivot column
                ++nrank;
                // only the words from the panel on: an unused row is zero in every column already passed (a column is declared
                // non-pivot when NO unused row has it set, and unused rows only ever absorb rows that were unused then), so the
                // pivot row leaves every earlier word as it is -- those words are final
                unsigned long long piv[W];
#pragma unroll
                for (int x = w; x < W; ++x) piv[x] = pw[x];  // broadcast reads
                const bool mine = (win == wave);
                if (mine) {
                    if (lane == psrc) usedm |= 1u << pslot;
                    if (lane == 0) pivrow[64 * w + b] = (short)(64 * (wave * RPL + pslot) + psrc);
                }
                // under the execution mask of the rows that have the column set (the pivot row itself excepted)
#pragma unroll
                for (int q = 0; q < RPL; ++q) {
                    const bool hit = (row[q][w] & mask) != 0ull && !(mine && q == pslot && lane == psrc);
                    if (hit) {
#pragma unroll
                        for (int x = w; x < W; ++x) row[q][x] ^= piv[x];
                    }
                }
            }
        }
        __syncthreads();

        // ------------------------------------------------------------------ non-pivot columns, reduced syndrome
        if (wave == 0) {
            int kpw = 0;
            for (int w = 0; w < W; ++w) {
                const int j = 64 * w + lane;
                const unsigned long long np = __ballot(j < n && pivrow[j] < 0);
                if (lane == 0) npmask[w] = np;
                if (j < n && pivrow[j] < 0) tpos[kpw + __popcll(np & ((1ull << lane) - 1ull))] = (unsigned short)j;
                kpw += __popcll(np);
            }
            if (lane == 0) misc[1] = (unsigned int)kpw;
        }
        unsigned long long y[RPL];  // wave-uniform: reduced syndrome over the pivot rows of my slot q
        {
            int pw0 = 0;
#pragma unroll
            for (int q = 0; q < RPL; ++q) {
                y[q] = __ballot(((usedm >> q) & 1u) != 0u && (row[q][W - 1] >> 63) != 0ull);
                pw0 += __popcll(y[q]);
                if (lane == 0) y0[wave * RPL + q] = y[q];
            }
            if (lane == 0) part[wave] = (unsigned int)pw0;
        }
        __syncthreads();
        const int kp = (int)misc[1];
        int w0 = 0;
#pragma unroll
        for (int v = 0; v < NWV; ++v) w0 += (int)part[v];

        // writes the solution "pivot bits from yy (my wave's slots), plus the columns of `flips` switched on" in original bit order
        auto write_solution = [&](const unsigned long long* yy, unsigned long long fa, int fpos_a, int fpos_b, uint8_t* out, uint8_t* cmp) {
            // fa: bit a set = T position a (< 64) switched on; fpos_a / fpos_b: sorted positions switched on (singles / pairs), -1 = none
            __syncthreads();  // yv / the bitmap below are free
#pragma unroll
            for (int q = 0; q < RPL; ++q)
                if (lane == q) yv[wave * RPL + q] = yy[q];
            const int wpn = (n + 63) >> 6;
            unsigned int* bits = (unsigned int*)keys;  // packed form: the row meets in an LDS bitmap (the sort keys are dead)
            if (PACKED)
                for (int w = tid; w < 2 * wpn; w += NT) bits[w] = 0u;
            __syncthreads();
            for (int j = tid; j < n; j += NT) {
                const int pr = pivrow[j];
                uint8_t bit;
                if (pr >= 0) bit = (uint8_t)((yv[pr >> 6] >> (pr & 63)) & 1ull);
                else bit = (uint8_t)((j == fpos_a || j == fpos_b) ? 1 : 0);
                if (pr < 0 && fa) {  // osd_e pattern: is this non-pivot column one of the pattern's?
                    for (unsigned long long pp = fa; pp; pp &= pp - 1)
                        if ((int)tpos[__ffsll((long long)pp) - 1] == j) bit = 1;
                }
                const int i = kidx[j];
                if (PACKED) {
                    if (bit) atomicOr(&bits[i >> 5], 1u << (i & 31));
                } else {
                    if (out) out[(size_t)s * n + i] = bit;
                    if (cmp) cmp[(size_t)slot_id * n + i] = bit;
                }
            }
            if (PACKED) {
                __syncthreads();
                for (int w = tid; w < wpn; w += NT) {
                    const unsigned long long v = (unsigned long long)bits[2 * w] | ((unsigned long long)bits[2 * w + 1] << 32);
                    if (out) ((unsigned long long*)out)[(size_t)s * wpn + w] = v;
                    if (cmp) ((unsigned long long*)cmp)[(size_t)slot_id * wpn + w] = v;
                }
            }
            __syncthreads();
        };
        // OSD-0
        if (P.out_osd0 || P.cmp_osd0) write_solution(y, 0ull, -1, -1, P.out_osd0, P.cmp_osd0);

        // ------------------------------------------------------------------ a10 / a11: candidates (integer weights)
        // reduced column (over my wave's pivot rows) of sorted position j as RPL ballot words; j is workgroup-uniform
        auto column_of = [&](int j, unsigned long long* cb) {
            const int jw = j >> 6, jb = j & 63;
#pragma unroll
            for (int q = 0; q < RPL; ++q) cb[q] = 0ull;
#pragma unroll
            for (int w = 0; w < W; ++w) {
                if (w == jw) {  // uniform
#pragma unroll
                    for (int q = 0; q < RPL; ++q) cb[q] = __ballot(((usedm >> q) & 1u) != 0u && ((row[q][w] >> jb) & 1ull) != 0ull);
                }
            }
        };
        int sel_a = -1, sel_b = -1;            // sorted positions switched on by the winner
        unsigned long long sel_pat = 0ull;     // or an osd_e pattern over T positions
        unsigned long long ybest[RPL];
#pragma unroll
        for (int q = 0; q < RPL; ++q) ybest[q] = y[q];
        if (P.osd_method >= 2 && P.osd_order > 0) {
            const int wspan = P.osd_order < OSDM_MAXSPAN ? P.osd_order : OSDM_MAXSPAN;
            int bestw = w0;
            // every wave walks the non-pivot columns in enumeration order: its share of each single candidate's weight goes to
            // wsum (osd_cs), the first wspan reduced columns to colvec (both methods)
            {
                int a = 0;
                for (int w = 0; w < W; ++w) {
                    unsigned long long npm = npmask[w];
                    npm = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(npm >> 32)) << 32) |
                          (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)npm);
                    if (P.osd_method != 3 && a >= wspan) break;
                    while (npm) {
                        const int b = __ffsll((long long)npm) - 1;
                        npm &= npm - 1;
                        if (P.osd_method != 3 && a >= wspan) break;
                        unsigned long long cb[RPL];
                        column_of(64 * w + b, cb);
                        if (a < wspan && lane == 0) {
#pragma unroll
                            for (int q = 0; q < RPL; ++q) colvec[a * NSL + wave * RPL + q] = cb[q];
                        }
                        if (P.osd_method == 3) {
                            int wgt = 0;
#pragma unroll
                            for (int q = 0; q < RPL; ++q) wgt += __popcll(y[q] ^ cb[q]);
                            if (lane == 0) atomicAdd(&wsum[a], (unsigned int)wgt);
                        }
                        ++a;
                    }
                }
            }
            if (tid == 0) best[0] = ~0ull;
            __syncthreads();
            const int ntc = kp < wspan ? kp : wspan;
            if (P.osd_method == 3) {
                // singles: the lightest, the first of equals (enumeration index = T position)
                {
                    unsigned long long mykey = ~0ull;
                    for (int a = tid; a < kp; a += NT) {
                        const unsigned long long key = ((unsigned long long)wsum[a] << 32) | (unsigned)a;
                        mykey = key < mykey ? key : mykey;
                    }
                    if (mykey != ~0ull) atomicMin(&best[0], mykey);
                }
                __syncthreads();
                const unsigned long long k1 = best[0];
                int single_a = -1;
                if (k1 != ~0ull && (int)(k1 >> 32) < bestw) {  // strictly lighter than OSD-0
                    bestw = (int)(k1 >> 32);
                    single_a = (int)(k1 & 0xffffffffu);
                }
                __syncthreads();
                if (tid == 0) best[0] = ~0ull;
                __syncthreads();
                // pairs (a < b < wspan), a outer, b inner: one per thread and round
                const int npairs = ntc * (ntc - 1) / 2;
                unsigned long long mykey = ~0ull;
                for (int pidx = tid; pidx < npairs; pidx += NT) {
                    int pa = 0, rem = pidx;
                    while (rem >= ntc - 1 - pa) { rem -= ntc - 1 - pa; ++pa; }
                    const int pb = pa + 1 + rem;
                    int wgt = 2;
#pragma clang loop unroll(disable)
                    for (int k = 0; k < NSL; ++k) wgt += __popcll(y0[k] ^ colvec[pa * NSL + k] ^ colvec[pb * NSL + k]);
                    const unsigned long long key = ((unsigned long long)wgt << 32) | (unsigned)pidx;
                    mykey = key < mykey ? key : mykey;
                }
                if (mykey != ~0ull) atomicMin(&best[0], mykey);
                __syncthreads();
                const unsigned long long k2 = best[0];
                if (k2 != ~0ull && (int)(k2 >> 32) < bestw) {  // strictly lighter than the best single
                    bestw = (int)(k2 >> 32);
                    int pidx = (int)(k2 & 0xffffffffu), pa = 0, rem = pidx;
                    while (rem >= ntc - 1 - pa) { rem -= ntc - 1 - pa; ++pa; }
                    const int pb = pa + 1 + rem;
                    sel_a = tpos[pa];
                    sel_b = tpos[pb];
#pragma unroll
                    for (int q = 0; q < RPL; ++q) ybest[q] = y[q] ^ colvec[pa * NSL + wave * RPL + q] ^ colvec[pb * NSL + wave * RPL + q];
                } else if (single_a >= 0) {
                    sel_a = tpos[single_a];
                    sel_b = -1;
                    unsigned long long cb[RPL];
                    column_of(sel_a, cb);  // (beyond the first wspan columns nothing is kept: from the registers again)
#pragma unroll
                    for (int q = 0; q < RPL; ++q) ybest[q] = y[q] ^ cb[q];
                }
            } else {
                // osd_e: patterns 1 .. 2^w - 1 over the first w non-pivot columns (enumeration index: osd_e_index)
                const unsigned int npat = (1u << ntc) - 1u;
                unsigned long long mykey = ~0ull;
                for (unsigned int pat = tid + 1; pat <= npat; pat += NT) {
                    int wgt = __popc(pat);
#pragma clang loop unroll(disable)
                    for (int k = 0; k < NSL; ++k) {
                        unsigned long long v = y0[k];
                        for (unsigned int pp = pat; pp; pp &= pp - 1) v ^= colvec[(__ffs((int)pp) - 1) * NSL + k];
                        wgt += __popcll(v);
                    }
                    const unsigned long long key = ((unsigned long long)wgt << 32) | osd_e_index(pat, ntc, P.e_msb_first);
                    mykey = key < mykey ? key : mykey;
                }
                if (mykey != ~0ull) atomicMin(&best[0], mykey);
                __syncthreads();
                const unsigned long long k1 = best[0];
                if (k1 != ~0ull && (int)(k1 >> 32) < bestw) {
                    const unsigned int pat = osd_e_index((unsigned int)(k1 & 0xffffffffu), ntc, P.e_msb_first);
                    sel_pat = pat;
                    for (unsigned int pp = pat; pp; pp &= pp - 1) {
                        const int bq = __ffs((int)pp) - 1;
#pragma unroll
                        for (int q = 0; q < RPL; ++q) ybest[q] ^= colvec[bq * NSL + wave * RPL + q];
                    }
                }
            }
        }
        write_solution(ybest, sel_pat, sel_a, sel_b, P.out_osdw, P.cmp_osdw);
    }
}

}  // namespace bposd
