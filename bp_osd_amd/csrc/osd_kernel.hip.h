// osd_kernel.hip.h -- batched ordered-statistics decoding (OSD-0 / OSD-E / OSD-CS) for gfx950.
//
// Restates rows a8-a11 of SURVEY.md §8 (the OSD half of `.decode(syndrome)`, results
// read as `.osd0_decoding` / `.osdw_decoding`: /root/reference/README.md:202,
// /root/reference/src/bposd/css_decode_sim.py:257-258,294-295).
//
// One workgroup post-processes one non-converged syndrome at a time (persistent, pulls
// from the list the BP kernel appended to).  Thread r owns row r of the parity-check
// matrix as W 64-bit words HELD IN REGISTERS -- the columns are physically permuted
// into reliability order first, so "process columns in sorted order" becomes a sweep
// over bit positions with compile-time word indices.  The syndrome rides along as the
// last bit of the last word.
//
//   1. a8  reliability sort: bitonic network in LDS on (order-preserving u64 image of the
//          LLR, bit index) -- a strict total order, so the result equals a stable sort.
//   2. a9  Gauss-Jordan sweep, one barrier per column: every wave ballots its candidate
//          rows, speculatively publishes its first candidate row to LDS; after the
//          barrier everybody knows the winning wave, and every row with a 1 in the
//          column XORs the pivot row in (words >= current word only: a not-yet-used row
//          has no support left of the sweep position).  Stops after `rank` pivots.
//          Pivot columns = the greedy independent set in sorted order, exactly the set
//          any row-pivoting strategy finds; OSD-0 = the syndrome bit of each pivot row.
//   3. a10/a11  OSD-W: in reduced form the solution for a candidate that switches on
//          non-pivot columns {t} is  x_S = y ^ XOR_t A_t,  so its weight is a popcount:
//          singles via per-wave ballots + LDS counters, pairs / exhaustive patterns via
//          transposed column bit-vectors.  Selection is the lexicographic minimum of
//          (weight, enumeration index) with OSD-0 first, i.e. "replace only if strictly
//          lighter, first found wins".
//
// Integer / bitwise throughout; weights are Hamming weights, which order candidates
// exactly like sum log(1/p) for uniform channel probabilities (host checks this).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bposd {

struct OsdParams {
    int m, n;
    int rank;
    int osd_method;  // 1 osd0, 2 osd_e, 3 osd_cs
    int osd_order;
    int tie_policy;
    int nsort;  // power of two >= n
    const uint8_t* __restrict__ synd;  // [B, m]
    const int* __restrict__ rp;        // CSR indptr [m+1]
    const int* __restrict__ ci;        // CSR indices [E]
    const double* __restrict__ llr_ws; // [cap, n]
    const int* __restrict__ osd_list;  // [cap]
    int* __restrict__ counters;        // [1] = number of list entries, [2] = OSD work queue
    uint8_t* __restrict__ out_osd0;    // [B, n] nullable
    uint8_t* __restrict__ out_osdw;    // [B, n]
};

__device__ __forceinline__ unsigned long long llr_sort_key(double x) {
    // order-preserving map double -> u64 (x + 0.0 folds -0.0 into +0.0: they compare equal)
    const unsigned long long u = (unsigned long long)__double_as_longlong(x + 0.0);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}

// LDS carve-up (bytes), all offsets 8-byte aligned
struct OsdLds {
    unsigned long long* keys;     // [nsort]
    unsigned short* kidx;         // [nsort]  -> order[j] after the sort
    unsigned short* inv;          // [n] original column -> sorted position
    short* pivrow;                // [nsort] sorted position -> pivot row, -1 if non-pivot
    int* wt;                      // [nsort] weight of single candidate at sorted position
    unsigned long long* rowbuf;   // [2][16][W]
    unsigned long long* colvec;   // [64][16]
    unsigned long long* yvec;     // [16]
    unsigned int* slot;           // [2][16]
    unsigned char* xout;          // [n]
    unsigned long long* best64;   // [2]
    int* misc;                    // [8]
};

__host__ __device__ inline size_t osd_lds_bytes(int n, int nsort, int W) {
    size_t b = 0;
    b += (size_t)nsort * 8;                 // keys
    b += (size_t)nsort * 2;                 // kidx
    b += ((size_t)n * 2 + 7) & ~(size_t)7;  // inv
    b += (size_t)nsort * 2;                 // pivrow
    b += (size_t)nsort * 4;                 // wt
    b += (size_t)2 * 16 * W * 8;            // rowbuf
    b += (size_t)64 * 16 * 8;               // colvec
    b += 16 * 8;                            // yvec
    b += 2 * 16 * 4;                        // slot
    b += ((size_t)n + 7) & ~(size_t)7;      // xout
    b += 2 * 8;                             // best64
    b += 8 * 4;                             // misc
    return b + 64;
}

template <int W>
__global__ __launch_bounds__(1024) void osd_kernel(const OsdParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int m = P.m, n = P.n, NS = P.nsort;
    const int NT = blockDim.x;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int nwaves = NT >> 6;
    constexpr int SB = 64 * W - 1;  // bit position of the syndrome column

    OsdLds L;
    {
        unsigned char* p = smem;
        L.keys = (unsigned long long*)p; p += (size_t)NS * 8;
        L.rowbuf = (unsigned long long*)p; p += (size_t)2 * 16 * W * 8;
        L.colvec = (unsigned long long*)p; p += (size_t)64 * 16 * 8;
        L.yvec = (unsigned long long*)p; p += 16 * 8;
        L.best64 = (unsigned long long*)p; p += 2 * 8;
        L.wt = (int*)p; p += (size_t)NS * 4;
        L.slot = (unsigned int*)p; p += 2 * 16 * 4;
        L.misc = (int*)p; p += 8 * 4;
        L.kidx = (unsigned short*)p; p += (size_t)NS * 2;
        L.pivrow = (short*)p; p += (size_t)NS * 2;
        L.inv = (unsigned short*)p; p += ((size_t)n * 2 + 7) & ~(size_t)7;
        L.xout = p;
    }

    for (;;) {
        if (tid == 0) L.misc[0] = atomicAdd(&P.counters[2], 1);
        __syncthreads();
        const int slot_id = L.misc[0];
        const int nlist = P.counters[1];
        if (slot_id >= nlist) break;  // uniform
        const long long s = P.osd_list[slot_id];
        const double* llr = P.llr_ws + (size_t)slot_id * n;

        // ------------------------------------------------------------------ a8: sort
        for (int i = tid; i < NS; i += NT) {
            if (i < n) {
                L.keys[i] = llr_sort_key(llr[i]);
                L.kidx[i] = (unsigned short)(P.tie_policy == 1 ? n - 1 - i : i);
            } else {
                L.keys[i] = ~0ull;
                L.kidx[i] = (unsigned short)i;  // >= n: pads sort last
            }
            L.pivrow[i] = -1;
            L.wt[i] = 1;
        }
        __syncthreads();
        for (int k = 2; k <= NS; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int t = tid; t < (NS >> 1); t += NT) {
                    // pair (lo, hi = lo ^ j) with bit j of lo clear
                    const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                    const int hi = lo | j;
                    const bool up = ((lo & k) == 0);
                    const unsigned long long ka = L.keys[lo], kb = L.keys[hi];
                    const unsigned short ia = L.kidx[lo], ib = L.kidx[hi];
                    const bool a_gt_b = (ka > kb) || (ka == kb && ia > ib);
                    if (a_gt_b == up) {
                        L.keys[lo] = kb; L.keys[hi] = ka;
                        L.kidx[lo] = ib; L.kidx[hi] = ia;
                    }
                }
                __syncthreads();
            }
        }
        if (P.tie_policy == 1) {
            for (int i = tid; i < n; i += NT) L.kidx[i] = (unsigned short)(n - 1 - L.kidx[i]);
            __syncthreads();
        }
        for (int j = tid; j < n; j += NT) L.inv[L.kidx[j]] = (unsigned short)j;
        __syncthreads();

        // ------------------------------------------- build my row in sorted column order
        unsigned long long row[W];
#pragma unroll
        for (int w = 0; w < W; ++w) row[w] = 0ull;
        if (tid < m) {
            const int e0 = P.rp[tid], e1 = P.rp[tid + 1];
            for (int e = e0; e < e1; ++e) {
                const int j = L.inv[P.ci[e]];
                const int jw = j >> 6;
                const unsigned long long bit = 1ull << (j & 63);
#pragma unroll
                for (int w = 0; w < W; ++w) row[w] |= (jw == w) ? bit : 0ull;
            }
            if (P.synd[(size_t)s * m + tid] & 1) row[W - 1] |= 1ull << 63;
        }

        // ------------------------------------------------------- a9: Gauss-Jordan sweep
        bool used = false;
        int mypos = -1;
        int nrank = 0;
        int par = 0;   // rowbuf double buffer
        int cur = 0;   // rotating candidate-wave mask word (3 deep, see below)
        bool done = false;
        if (tid < 3) L.slot[tid] = 0u;
        __syncthreads();
#pragma unroll
        for (int w = 0; w < W; ++w) {
            if (!done) {
                for (int b = 0; b < 64; ++b) {
                    const int j = w * 64 + b;
                    if (j >= n || nrank >= P.rank) { done = true; break; }
                    const bool bit = (row[w] >> b) & 1ull;
                    const bool cand = bit && !used;
                    const unsigned long long bal = __ballot(cand);
                    const int first = bal ? (__ffsll((long long)bal) - 1) : -1;
                    if (cand && lane == first) {
                        // speculative publish: my wave's first candidate row, words >= w
                        atomicOr(&L.slot[cur], 1u << wave);
                        unsigned long long* dst = L.rowbuf + (size_t)(par * 16 + wave) * W;
#pragma unroll
                        for (int x = w; x < W; ++x) dst[x] = row[x];
                    }
                    __syncthreads();
                    const unsigned int mask = L.slot[cur];
                    // Mask word (cur+2)%3 was last read before this barrier and is next written
                    // after the following one: safe to clear now.
                    const int nxt2 = (cur >= 1) ? cur - 1 : 2;
                    if (tid == 0) L.slot[nxt2] = 0u;
                    if (mask) {
                        const int wv = __ffs((int)mask) - 1;
                        if (cand && wave == wv && lane == first) {
                            used = true;
                            mypos = j;
                            L.pivrow[j] = (short)tid;
                        } else if (bit) {
                            const unsigned long long* src = L.rowbuf + (size_t)(par * 16 + wv) * W;
#pragma unroll
                            for (int x = w; x < W; ++x) row[x] ^= src[x];
                        }
                        ++nrank;
                    }
                    par ^= 1;
                    cur = (cur == 2) ? 0 : cur + 1;
                }
            }
        }
        const int y = (int)((row[W - 1] >> 63) & 1ull);

        // --------------------------------------------------------------- OSD-0 solution
        for (int i = tid; i < n; i += NT) L.xout[i] = 0;
        {
            const unsigned long long yb = __ballot(used && y);
            if (lane == 0) L.yvec[wave] = yb;
        }
        if (tid == 0) { L.best64[0] = ~0ull; L.best64[1] = ~0ull; }
        __syncthreads();  // also makes every pivrow[] write visible
        if (used && y) L.xout[L.kidx[mypos]] = 1;
        __syncthreads();
        if (P.out_osd0)
            for (int i = tid; i < n; i += NT) P.out_osd0[(size_t)s * n + i] = L.xout[i];

        int w0 = 0;
        for (int q = 0; q < nwaves; ++q) w0 += __popcll(L.yvec[q]);

        int sel_a = -1, sel_b = -1;  // sorted positions switched on by the winning candidate
        if (P.osd_method >= 2 && P.osd_order > 0) {
            // ---------------- a10/a11: singles weights + transposed columns of the first w non-pivots
            const int wspan = P.osd_order < 64 ? P.osd_order : 64;
            int tcount = 0;  // running T-index (uniform)
#pragma unroll
            for (int w = 0; w < W; ++w) {
                for (int b = 0; b < 64; ++b) {
                    const int j = w * 64 + b;
                    if (j >= n) break;
                    if (L.pivrow[j] >= 0) continue;  // uniform
                    const int bit = (int)((row[w] >> b) & 1ull);
                    if (P.osd_method == 3) {
                        const unsigned long long d = __ballot(used && (bit ^ y));
                        if (lane == 0 && d) atomicAdd(&L.wt[j], __popcll(d));
                    }
                    if (tcount < wspan) {
                        const unsigned long long cb = __ballot(used && bit);
                        if (lane == 0) L.colvec[tcount * 16 + wave] = cb;
                    }
                    if (tid == 0 && tcount < 64) ((unsigned short*)L.keys)[tcount] = (unsigned short)j;
                    ++tcount;
                }
            }
            __syncthreads();
            const unsigned short* tpos = (const unsigned short*)L.keys;  // T-index -> sorted position (first 64)
            if (P.osd_method == 3) {
                // singles: all k' non-pivot positions, enumeration order == position order
                for (int j = tid; j < n; j += NT) {
                    if (L.pivrow[j] < 0) {
                        const unsigned long long key = ((unsigned long long)L.wt[j] << 32) | (unsigned)j;
                        atomicMin(&L.best64[0], key);
                    }
                }
                // pairs (a < b < w), a outer, b inner
                const int npairs = wspan * (wspan - 1) / 2;
                for (int pidx = tid; pidx < npairs; pidx += NT) {
                    // decode pair index: rows of lengths wspan-1, wspan-2, ...
                    int a = 0, rem = pidx;
                    while (rem >= wspan - 1 - a) { rem -= wspan - 1 - a; ++a; }
                    const int bq = a + 1 + rem;
                    int wgt = 2;
                    for (int q = 0; q < nwaves; ++q)
                        wgt += __popcll(L.yvec[q] ^ L.colvec[a * 16 + q] ^ L.colvec[bq * 16 + q]);
                    const unsigned long long key = ((unsigned long long)wgt << 32) | (unsigned)pidx;
                    atomicMin(&L.best64[1], key);
                }
                __syncthreads();
                int bestw = w0;
                const unsigned long long k1 = L.best64[0], k2 = L.best64[1];
                if (k1 != ~0ull && (int)(k1 >> 32) < bestw) {
                    bestw = (int)(k1 >> 32);
                    sel_a = (int)(k1 & 0xffffffffu);
                    sel_b = -1;
                }
                if (k2 != ~0ull && (int)(k2 >> 32) < bestw) {
                    bestw = (int)(k2 >> 32);
                    int pidx = (int)(k2 & 0xffffffffu);
                    int a = 0, rem = pidx;
                    while (rem >= wspan - 1 - a) { rem -= wspan - 1 - a; ++a; }
                    sel_a = tpos[a];
                    sel_b = tpos[a + 1 + rem];
                }
            } else {
                // osd_e: patterns 1 .. 2^w - 1 on the first w non-pivots, LSB -> T position 0
                const unsigned int npat = (1u << wspan) - 1u;
                for (unsigned int pat = tid + 1; pat <= npat; pat += NT) {
                    int wgt = __popc(pat);
                    for (int q = 0; q < nwaves; ++q) {
                        unsigned long long v = L.yvec[q];
                        unsigned int pp = pat;
                        while (pp) {
                            const int bq = __ffs((int)pp) - 1;
                            pp &= pp - 1;
                            v ^= L.colvec[bq * 16 + q];
                        }
                        wgt += __popcll(v);
                    }
                    const unsigned long long key = ((unsigned long long)wgt << 32) | pat;
                    atomicMin(&L.best64[0], key);
                }
                __syncthreads();
                const unsigned long long k1 = L.best64[0];
                if (k1 != ~0ull && (int)(k1 >> 32) < w0) {
                    // winner is a multi-column pattern: fold it into sel via misc
                    sel_a = -2;
                    sel_b = (int)(k1 & 0xffffffffu);
                }
            }
        }

        // ------------------------------------------------- write the OSD-W solution
        if (sel_a == -1) {
            // OSD-0 stays the best
            for (int i = tid; i < n; i += NT) P.out_osdw[(size_t)s * n + i] = L.xout[i];
        } else {
            __syncthreads();
            for (int i = tid; i < n; i += NT) L.xout[i] = 0;
            __syncthreads();
            int xs = y;
            if (sel_a >= 0) {
                // one or two switched-on columns at sorted positions sel_a, sel_b
                int ba = 0, bb = 0;
#pragma unroll
                for (int w = 0; w < W; ++w) {
                    if ((sel_a >> 6) == w) ba = (int)((row[w] >> (sel_a & 63)) & 1ull);
                    if (sel_b >= 0 && (sel_b >> 6) == w) bb = (int)((row[w] >> (sel_b & 63)) & 1ull);
                }
                xs ^= ba ^ bb;
                if (tid == 0) {
                    L.xout[L.kidx[sel_a]] = 1;
                    if (sel_b >= 0) L.xout[L.kidx[sel_b]] = 1;
                }
            } else {
                const unsigned short* tpos = (const unsigned short*)L.keys;
                unsigned int pp = (unsigned int)sel_b;
                while (pp) {
                    const int bq = __ffs((int)pp) - 1;
                    pp &= pp - 1;
                    const int pos = tpos[bq];
                    int bv = 0;
#pragma unroll
                    for (int w = 0; w < W; ++w)
                        if ((pos >> 6) == w) bv = (int)((row[w] >> (pos & 63)) & 1ull);
                    xs ^= bv;
                    if (tid == 0) L.xout[L.kidx[pos]] = 1;
                }
            }
            if (used && xs) L.xout[L.kidx[mypos]] = 1;
            __syncthreads();
            for (int i = tid; i < n; i += NT) P.out_osdw[(size_t)s * n + i] = L.xout[i];
        }
        __syncthreads();
    }
}

}  // namespace bposd
