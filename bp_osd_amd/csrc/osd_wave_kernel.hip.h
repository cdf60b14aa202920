// osd_wave_kernel.hip.h -- OSD-0 / OSD-E / OSD-CS for SMALL codes: one WAVE per non-converged syndrome.
//
// osd_kernel.hip.h gives a syndrome one workgroup and pays a workgroup barrier per pivot; for the codes the reference
// itself ships and runs (/root/reference/examples/qldpc_decode_example.py: [[400,16,6]], 192 checks; [[625,25,8]] 300;
// [[900,36,10]] 432) that is two to seven waves waiting on each other 200-400 times per elimination, and at the
// reference's own operating point (error rate 0.05) a quarter of all shots need OSD: 32 k eliminations per 131072-batch
// took 6.9 ms beside 15 ms of BP.  Here a lane owns RPL rows (row r = lane + 64 s), each as W 64-bit words in registers in
// reliability-sorted column order with the syndrome as the last bit; everything is wave-local:
//   a8   sort        bitonic network on (key, index) in the wave's LDS slice (same keys, same tie rule as osd_kernel)
//   a9   elimination Gauss-Jordan, column by column in sorted order: the pivot row is found with one ballot per row slot,
//                    broadcast with v_readlane, and XORed into every row that has the column set -- no barrier, no LDS
//   a9   OSD-0       x[pivot column] = reduced syndrome bit of its pivot row
//   a10/11 sweep     integer weights (uniform channel): singles = 1 + popcount(y ^ column) from ballots, all in scalar
//                    registers in enumeration order; pairs / osd_e patterns one candidate per lane from the first w
//                    reduced columns kept in LDS, (weight, enumeration index) minimum through an LDS atomic
// Waves are persistent on the OSD list's atomic queue.  Non-uniform / per-shot channels (fp64 weights), osd_e orders
// above 12 and codes beyond 448 checks / 959 bits stay on osd_kernel.hip.h.  Results are identical to it and to the
// oracle (tests/test_gpu_parity.py: every small-code OSD test runs on this kernel where it applies).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "osd_kernel.hip.h"

namespace bposd {

constexpr int OSDW_WAVES = 4;          // waves per workgroup (independent of each other)
constexpr int OSDW_MAXSPAN = 64;       // osd_cs pair span / osd_e order kept as reduced columns in LDS
constexpr int OSDW_MAX_E = 12;         // osd_e orders beyond this go to osd_kernel

__host__ __device__ constexpr int osdw_nsort(int n) {
    int p = 64;
    while (p < n) p <<= 1;
    return p;
}
__host__ __device__ constexpr size_t osdw_lds_per_wave(int ns, int rpl, int w) {
    // keys u64 [ns] | kidx, inv, pivrow, tpos u16 [ns] each | colvec u64 [MAXSPAN][rpl] | yv u64 [rpl] | best u64 [2] | npmask u64 [w]
    return (size_t)ns * 8 + (size_t)ns * 2 * 4 + (size_t)OSDW_MAXSPAN * rpl * 8 + (size_t)rpl * 8 + 16 + (size_t)w * 8;
}

__device__ __forceinline__ void osdw_sync() {  // wave-local: LDS written by other lanes of this wave is visible afterwards
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ unsigned long long osdw_readlane64(unsigned long long v, int src) {
    const unsigned int lo = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)v, src);
    const unsigned int hi = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)(v >> 32), src);
    return ((unsigned long long)hi << 32) | lo;
}

// RPL: rows per lane (m <= 64 * RPL);  W: 64-bit words per row (n + 1 <= 64 * W)
// PACKED: the packed-I/O form (OsdParams::packed_io) as a compile-time switch (as a run-time switch it cost the byte form 4 %)
template <int RPL, int W, bool PACKED = false>
__global__ __launch_bounds__(64 * OSDW_WAVES) void osd_wave_kernel(const OsdParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int m = P.m, n = P.n;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int NS = osdw_nsort(n);
    unsigned char* p = smem + (size_t)wave * osdw_lds_per_wave(NS, RPL, W);
    unsigned long long* keys = (unsigned long long*)p; p += (size_t)NS * 8;
    unsigned long long* colvec = (unsigned long long*)p; p += (size_t)OSDW_MAXSPAN * RPL * 8;
    unsigned long long* yv = (unsigned long long*)p; p += (size_t)RPL * 8;
    unsigned long long* best = (unsigned long long*)p; p += 16;
    unsigned long long* npmask = (unsigned long long*)p; p += (size_t)W * 8;
    unsigned short* kidx = (unsigned short*)p; p += (size_t)NS * 2;   // sorted position -> original bit
    unsigned short* inv = (unsigned short*)p; p += (size_t)NS * 2;    // original bit -> sorted position
    short* pivrow = (short*)p; p += (size_t)NS * 2;                   // sorted position -> pivot row (64 * slot + lane), -1 = none
    unsigned short* tpos = (unsigned short*)p;                        // a-th non-pivot column (sorted position)

    for (;;) {
        int slot_id = 0;
        if (lane == 0) slot_id = atomicAdd(&P.counters[2], 1);
        slot_id = __builtin_amdgcn_readfirstlane(slot_id);
        const int nlist = P.counters[1];
        if (slot_id >= nlist) break;  // wave-uniform
        const long long s = P.osd_list[slot_id];
        const double* llr = P.llr_ws + (size_t)slot_id * n;

        // ------------------------------------------------------------------ a8: sort (same order as osd_kernel's)
        for (int i = lane; i < NS; i += 64) {
            if (i < n) {
                keys[i] = llr_sort_key(llr[i]);
                kidx[i] = (unsigned short)(P.tie_policy == 1 ? n - 1 - i : i);
            } else {
                keys[i] = ~0ull;
                kidx[i] = (unsigned short)i;
            }
            pivrow[i] = -1;
        }
        osdw_sync();
        for (int k = 2; k <= NS; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int t = lane; t < (NS >> 1); t += 64) {
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
                osdw_sync();
            }
        }
        if (P.tie_policy == 1) {
            for (int i = lane; i < n; i += 64) kidx[i] = (unsigned short)(n - 1 - kidx[i]);
            osdw_sync();
        }
        for (int j = lane; j < n; j += 64) inv[kidx[j]] = (unsigned short)j;
        osdw_sync();

        // ------------------------------------------- my rows in sorted column order, syndrome = bit 63 of the last word
        unsigned long long row[RPL][W];
#pragma unroll
        for (int q = 0; q < RPL; ++q) {
#pragma unroll
            for (int w = 0; w < W; ++w) row[q][w] = 0ull;
            const int r = lane + 64 * q;
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
        int nrank = 0;
#pragma unroll
        for (int w = 0; w < W; ++w) {
            const int nb = (n - 64 * w) < 64 ? (n - 64 * w) : 64;  // columns of this word (the syndrome bit is not one)
            for (int b = 0; b < nb && nrank < P.rank; ++b) {
                const unsigned long long mask = 1ull << b;
                bool found = false;
                unsigned long long piv[W];
#pragma unroll
                for (int x = 0; x < W; ++x) piv[x] = 0ull;  // (defined on every path: no loop-carried copies)
                int psrc = 0, pslot = 0;
#pragma unroll
                for (int q = 0; q < RPL; ++q) {
                    if (!found) {
                        const unsigned long long bal = __ballot((row[q][w] & mask) != 0ull && ((usedm >> q) & 1u) == 0u);
                        if (bal) {  // wave-uniform
                            found = true;
                            psrc = __ffsll((long long)bal) - 1;
                            pslot = q;
#pragma unroll
                            for (int x = w; x < W; ++x) piv[x] = osdw_readlane64(row[q][x], psrc);  // (words left of the current one: see the update)
                            if (lane == psrc) usedm |= 1u << q;
                        }
                    }
                }
                if (!found) continue;
                if (lane == 0) pivrow[64 * w + b] = (short)(64 * pslot + psrc);
                ++nrank;
                // under the execution mask of the rows that have the column set: one v_xor with the (scalar) pivot word per
                // register, against a select + xor per register in the branch-free form (half the vector instructions).  Only the
                // words from the current one on (round 4): an unused row is zero in every column already passed -- a column is
                // declared non-pivot when NO unused row has it set, and unused rows only ever absorb rows that were unused then --
                // so the pivot row leaves every earlier word as it is
#pragma unroll
                for (int q = 0; q < RPL; ++q) {
                    const bool hit = (row[q][w] & mask) != 0ull && !(q == pslot && lane == psrc);
                    if (hit) {
#pragma unroll
                        for (int x = w; x < W; ++x) row[q][x] ^= piv[x];
                    }
                }
            }
        }
        osdw_sync();

        // ------------------------------------------------------------------ non-pivot columns, reduced syndrome
        int kp = 0;
        for (int w = 0; w < W; ++w) {
            const int j = 64 * w + lane;
            const unsigned long long np = __ballot(j < n && pivrow[j] < 0);
            if (lane == 0) npmask[w] = np;
            if (j < n && pivrow[j] < 0) tpos[kp + __popcll(np & ((1ull << lane) - 1ull))] = (unsigned short)j;
            kp += __popcll(np);
        }
        unsigned long long y[RPL];  // wave-uniform: reduced syndrome over the pivot rows of slot q
        int w0 = 0;
#pragma unroll
        for (int q = 0; q < RPL; ++q) {
            y[q] = __ballot(((usedm >> q) & 1u) != 0u && (row[q][W - 1] >> 63) != 0ull);
            w0 += __popcll(y[q]);
        }
        osdw_sync();

        // writes the solution "pivot bits from yy, plus the columns of `flips` switched on" in original bit order
        auto write_solution = [&](const unsigned long long* yy, unsigned long long fa, int fpos_a, int fpos_b, uint8_t* out, uint8_t* cmp) {
            // fa: bit a set = T position a (< 64) switched on; fpos_a / fpos_b: sorted positions switched on (singles / pairs), -1 = none
#pragma unroll
            for (int q = 0; q < RPL; ++q)
                if (lane == q) yv[q] = yy[q];
            osdw_sync();
            const int wpn = (n + 63) >> 6;
            unsigned int* bits = (unsigned int*)keys;  // packed form: the row meets in an LDS bitmap (the sort keys are dead)
            if (PACKED) {
                for (int w = lane; w < 2 * wpn; w += 64) bits[w] = 0u;
                osdw_sync();
            }
            for (int j = lane; j < n; j += 64) {
                const int pr = pivrow[j];
                uint8_t bit;
                if (pr >= 0) bit = (uint8_t)((yv[pr >> 6] >> (pr & 63)) & 1ull);
                else bit = (uint8_t)((j == fpos_a || j == fpos_b) ? 1 : 0);
                const int i = kidx[j];
                if (PACKED) {
                    if (bit) atomicOr(&bits[i >> 5], 1u << (i & 31));
                } else {
                    if (out) out[(size_t)s * n + i] = bit;
                    if (cmp) cmp[(size_t)slot_id * n + i] = bit;
                }
            }
            if (fa) {  // osd_e pattern: a few more ones
                unsigned long long pp = fa;
                while (pp) {
                    const int a = __ffsll((long long)pp) - 1;
                    pp &= pp - 1;
                    if (lane == 0) {
                        const int i = kidx[tpos[a]];
                        if (PACKED) {
                            atomicOr(&bits[i >> 5], 1u << (i & 31));
                        } else {
                            if (out) out[(size_t)s * n + i] = 1;
                            if (cmp) cmp[(size_t)slot_id * n + i] = 1;
                        }
                    }
                }
            }
            if (PACKED) {
                osdw_sync();
                for (int w = lane; w < wpn; w += 64) {
                    const unsigned long long v = (unsigned long long)bits[2 * w] | ((unsigned long long)bits[2 * w + 1] << 32);
                    if (out) ((unsigned long long*)out)[(size_t)s * wpn + w] = v;
                    if (cmp) ((unsigned long long*)cmp)[(size_t)slot_id * wpn + w] = v;
                }
            }
            osdw_sync();
        };
        // OSD-0
        if (P.out_osd0 || P.cmp_osd0) write_solution(y, 0ull, -1, -1, P.out_osd0, P.cmp_osd0);

        // ------------------------------------------------------------------ a10 / a11: candidates (integer weights)
        // reduced column (over the pivot rows) of sorted position j as RPL ballot words; j is wave-uniform
        auto column_of = [&](int j, unsigned long long* cb) {
            const int jw = j >> 6, jb = j & 63;
#pragma unroll
            for (int q = 0; q < RPL; ++q) cb[q] = 0ull;
#pragma unroll
            for (int w = 0; w < W; ++w) {
                if (w == jw) {  // wave-uniform
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
            const int wspan = P.osd_order < OSDW_MAXSPAN ? P.osd_order : OSDW_MAXSPAN;
            int bestw = w0;
            // singles (osd_cs) and the first wspan reduced columns (both methods), in enumeration order
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
                        for (int q = 0; q < RPL; ++q) colvec[a * RPL + q] = cb[q];
                    }
                    if (P.osd_method == 3) {
                        int wgt = 1;
#pragma unroll
                        for (int q = 0; q < RPL; ++q) wgt += __popcll(y[q] ^ cb[q]);
                        if (wgt < bestw) {  // strictly lighter: the first of equals wins
                            bestw = wgt;
                            sel_a = 64 * w + b;
                            sel_b = -1;
#pragma unroll
                            for (int q = 0; q < RPL; ++q) ybest[q] = y[q] ^ cb[q];
                        }
                    }
                    ++a;
                }
            }
            if (lane == 0) best[0] = ~0ull;
            osdw_sync();
            const int ntc = kp < wspan ? kp : wspan;
            if (P.osd_method == 3) {
                // pairs (a < b < wspan), a outer, b inner: one per lane and round
                const int npairs = ntc * (ntc - 1) / 2;
                unsigned long long mykey = ~0ull;
                for (int pidx = lane; pidx < npairs; pidx += 64) {
                    int pa = 0, rem = pidx;
                    while (rem >= ntc - 1 - pa) { rem -= ntc - 1 - pa; ++pa; }
                    const int pb = pa + 1 + rem;
                    int wgt = 2;
#pragma unroll
                    for (int q = 0; q < RPL; ++q) wgt += __popcll(y[q] ^ colvec[pa * RPL + q] ^ colvec[pb * RPL + q]);
                    const unsigned long long key = ((unsigned long long)wgt << 32) | (unsigned)pidx;
                    mykey = key < mykey ? key : mykey;
                }
                if (mykey != ~0ull) atomicMin(&best[0], mykey);
                osdw_sync();
                const unsigned long long k2 = best[0];
                if (k2 != ~0ull && (int)(k2 >> 32) < bestw) {
                    bestw = (int)(k2 >> 32);
                    int pidx = (int)(k2 & 0xffffffffu), pa = 0, rem = pidx;
                    while (rem >= ntc - 1 - pa) { rem -= ntc - 1 - pa; ++pa; }
                    const int pb = pa + 1 + rem;
                    sel_a = tpos[pa];
                    sel_b = tpos[pb];
#pragma unroll
                    for (int q = 0; q < RPL; ++q) ybest[q] = y[q] ^ colvec[pa * RPL + q] ^ colvec[pb * RPL + q];
                }
            } else {
                // osd_e: patterns 1 .. 2^w - 1 over the first w non-pivot columns (enumeration index: osd_e_index)
                const unsigned int npat = (1u << ntc) - 1u;
                unsigned long long mykey = ~0ull;
                for (unsigned int pat = lane + 1; pat <= npat; pat += 64) {
                    int wgt = __popc(pat);
#pragma unroll
                    for (int q = 0; q < RPL; ++q) {
                        unsigned long long v = y[q];
                        unsigned int pp = pat;
                        while (pp) {
                            const int bq = __ffs((int)pp) - 1;
                            pp &= pp - 1;
                            v ^= colvec[bq * RPL + q];
                        }
                        wgt += __popcll(v);
                    }
                    const unsigned long long key = ((unsigned long long)wgt << 32) | osd_e_index(pat, ntc, P.e_msb_first);
                    mykey = key < mykey ? key : mykey;
                }
                if (mykey != ~0ull) atomicMin(&best[0], mykey);
                osdw_sync();
                const unsigned long long k1 = best[0];
                if (k1 != ~0ull && (int)(k1 >> 32) < bestw) {
                    const unsigned int pat = osd_e_index((unsigned int)(k1 & 0xffffffffu), ntc, P.e_msb_first);
                    sel_pat = pat;
                    unsigned int pp = pat;
                    while (pp) {
                        const int bq = __ffs((int)pp) - 1;
                        pp &= pp - 1;
#pragma unroll
                        for (int q = 0; q < RPL; ++q) ybest[q] ^= colvec[bq * RPL + q];
                    }
                }
            }
        }
        write_solution(ybest, sel_pat, sel_a, sel_b, P.out_osdw, P.cmp_osdw);
    }
}

}  // namespace bposd
