// osd_large_kernel.hip.h -- OSD-0 / OSD-E / OSD-CS for codes beyond the register-resident kernel
// (m > 1024 or n > 2047; BASELINE configs[4]: 14520 x 29524 -> 53.7 MB of packed matrix per syndrome).
//
// Same algorithm and selection rule as osd_kernel.hip.h (rows a8-a11 of SURVEY.md §8), re-mapped for the
// HBM-bound regime.  One 1024-thread workgroup post-processes one syndrome at a time; thread t owns rows
// t, t + 1024, ... (RPT of them).  The permuted packed matrix lives in a per-workgroup global workspace
// M[word][row] (word-major: the rows of one 64-column word are contiguous -> every access by "my rows"
// is coalesced).  Per 64-column panel:
//   (i)  panel phase in registers: the thread's panel words and 64-bit combination masks; one barrier
//        per pivot, lowest proposed column wins (identical to the small kernel);
//   (ii) trailing phase streamed in chunks of 16 words: the <= 64 pivot rows' words of the chunk are
//        published to LDS, 4-bit "four Russians" tables are built, every row read-modify-writes its
//        chunk words in HBM.  Traffic per syndrome ~ sum over panels of the remaining matrix (r + w).
// Sort: bitonic network over a global key array (n up to 32767).  Sweep: per-wave ballots over the
// finished words re-read from M; singles' weights accumulate in a global int array.
// Limits: m <= 16384, n <= 32767, osd_cs / osd_e order <= 16, uniform channel (Hamming weights).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "osd_kernel.hip.h"

namespace bposd {

constexpr int OSDL_NT = 1024;
constexpr int OSDL_NW = OSDL_NT / 64;  // waves
constexpr int OSDL_CW = 16;            // chunk width (words) of the trailing update
constexpr int OSDL_MAXSPAN = 16;       // max osd order

struct OsdLargeParams {
    int m, n, W;  // W = ceil((n + 1) / 64)
    int rank;     // pivots to find (min(m, n) when the true rank is unknown)
    int osd_method, osd_order, tie_policy;
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
    // per-workgroup workspaces (index blockIdx.x)
    unsigned long long* __restrict__ mat;   // [grid][W * mrl]
    unsigned long long* __restrict__ keys;  // [grid][nsort]
    int* __restrict__ kidx;                 // [grid][nsort]   order[j] after the sort
    int* __restrict__ inv;                  // [grid][n]
    int* __restrict__ pivrow;               // [grid][64 * W]  sorted position -> pivot row, -1 non-pivot
    int* __restrict__ rowpos;               // [grid][mrl]     pivot position of a row, -1 if unused
    int* __restrict__ wt;                   // [grid][64 * W]  weights of the single candidates
    uint8_t* __restrict__ xout;             // [grid][n]
    int* __restrict__ rank_out;             // nullable: [0] = pivots found for list slot 0 (ctor-time rank probe)
};

__host__ __device__ inline size_t osd_large_lds_bytes(int W, int RPT) {
    size_t b = 0;
    b += (size_t)2 * OSDL_NW * 2 * 8;            // pbuf
    b += (size_t)2 * OSDL_NW * 4;                // pcol
    b += (size_t)64 * 4;                         // qrow
    b += (size_t)64 * OSDL_CW * 8;               // prow chunk
    b += (size_t)16 * OSDL_CW * 16 * 8;          // tab
    b += (size_t)OSDL_MAXSPAN * RPT * OSDL_NW * 8;  // colvec
    b += (size_t)RPT * OSDL_NW * 8;              // yvec
    b += (size_t)W * 8;                          // npmask
    b += (size_t)64 * 4;                         // tpos
    b += 2 * 8 + 16 * 4;                         // best64, misc
    return b + 64;
}

template <int RPT>
__global__ __launch_bounds__(OSDL_NT) void osd_large_kernel(const OsdLargeParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int m = P.m, n = P.n, W = P.W, NS = P.nsort, MRL = P.mrl;
    constexpr int NT = OSDL_NT;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    constexpr int NCV = RPT * OSDL_NW;

    unsigned char* p = smem;
    unsigned long long* pbuf = (unsigned long long*)p; p += (size_t)2 * OSDL_NW * 2 * 8;
    unsigned long long* prow = (unsigned long long*)p; p += (size_t)64 * OSDL_CW * 8;
    unsigned long long* tab = (unsigned long long*)p; p += (size_t)16 * OSDL_CW * 16 * 8;
    unsigned long long* colvec = (unsigned long long*)p; p += (size_t)OSDL_MAXSPAN * NCV * 8;
    unsigned long long* yvec = (unsigned long long*)p; p += (size_t)NCV * 8;
    unsigned long long* npmask = (unsigned long long*)p; p += (size_t)W * 8;
    unsigned long long* best64 = (unsigned long long*)p; p += 2 * 8;
    unsigned int* pcol = (unsigned int*)p; p += (size_t)2 * OSDL_NW * 4;
    int* qrow = (int*)p; p += 64 * 4;
    int* tpos = (int*)p; p += 64 * 4;
    int* misc = (int*)p;

    unsigned long long* M = P.mat + (size_t)blockIdx.x * W * MRL;
    unsigned long long* keys = P.keys + (size_t)blockIdx.x * NS;
    int* kidx = P.kidx + (size_t)blockIdx.x * NS;
    int* inv = P.inv + (size_t)blockIdx.x * n;
    int* pivrow = P.pivrow + (size_t)blockIdx.x * 64 * W;
    int* rowpos = P.rowpos + (size_t)blockIdx.x * MRL;
    int* wt = P.wt + (size_t)blockIdx.x * 64 * W;
    uint8_t* xout = P.xout + (size_t)blockIdx.x * n;

    for (;;) {
        if (tid == 0) misc[0] = atomicAdd(&P.counters[2], 1);
        __syncthreads();
        const int slot_id = misc[0];
        const int nlist = P.counters[1];
        if (slot_id >= nlist) break;
        const long long s = P.osd_list[slot_id];
        const double* llr = P.llr_ws + (size_t)slot_id * n;

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
        __syncthreads();
        for (int k = 2; k <= NS; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int t = tid; t < (NS >> 1); t += NT) {
                    const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                    const int hi = lo | j;
                    const bool up = ((lo & k) == 0);
                    const unsigned long long ka = keys[lo], kb = keys[hi];
                    const int ia = kidx[lo], ib = kidx[hi];
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
            for (int i = tid; i < n; i += NT) kidx[i] = n - 1 - kidx[i];
            __syncthreads();
        }
        for (int j = tid; j < n; j += NT) inv[kidx[j]] = j;
        // ------------------------------------------- build my rows (zero, then set the <= DC bits)
        for (int x = 0; x < W; ++x)
#pragma unroll
            for (int k = 0; k < RPT; ++k) M[(size_t)x * MRL + tid + k * NT] = 0ull;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            const int r = tid + k * NT;
            if (r < m) {
                const int e0 = P.rp[r], e1 = P.rp[r + 1];
                for (int e = e0; e < e1; ++e) {
                    const int j = inv[P.ci[e]];
                    M[(size_t)(j >> 6) * MRL + r] |= 1ull << (j & 63);  // only the owner touches row r
                }
                if (P.synd[(size_t)s * m + r] & 1) M[(size_t)(W - 1) * MRL + r] |= 1ull << 63;
            }
        }
        if (tid < 2 * OSDL_NW) pcol[tid] = 64u;
        __syncthreads();

        // ------------------------------------------------------- a9: blocked Gauss-Jordan
        unsigned int usedmask = 0u;  // bit k: my k-th row is a pivot row
        int nrank = 0;
        int par = 0;
        bool done = false;
#pragma clang loop unroll(disable)
        for (int w = 0; w < W && !done; ++w) {
            unsigned long long pw[RPT], t[RPT];
#pragma unroll
            for (int k = 0; k < RPT; ++k) { pw[k] = M[(size_t)w * MRL + tid + k * NT]; t[k] = 0ull; }
            int npiv = 0;
            const int nb = n - w * 64;
            const unsigned long long vmask = nb >= 64 ? ~0ull : (nb <= 0 ? 0ull : ((1ull << nb) - 1ull));
            if (nb <= 0) done = true;
#pragma clang loop unroll(disable)
            for (;;) {
                if (nrank >= P.rank) { done = true; break; }
                int lb = 64, kb = 0;
#pragma unroll
                for (int k = 0; k < RPT; ++k) {
                    const unsigned long long cw = ((usedmask >> k) & 1u) ? 0ull : (pw[k] & vmask);
                    const int l = cw ? (__ffsll((long long)cw) - 1) : 64;
                    if (l < lb) { lb = l; kb = k; }
                }
                unsigned long long act = ~0ull;
                int col = 0;
#pragma unroll
                for (int bitp = 6; bitp >= 0; --bitp) {
                    const unsigned long long z = __ballot(((lb >> bitp) & 1) == 0) & act;
                    if (z) act = z; else col |= (1 << bitp);
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
                const unsigned long long bmask = 1ull << mincol;
                const unsigned long long qbit = 1ull << npiv;
                const int j = w * 64 + mincol;
#pragma unroll
                for (int k = 0; k < RPT; ++k) {
                    const bool is_pivot = (wave == wv) && (lane == first) && (k == kb) && (col == mincol);
                    if (is_pivot) {
                        usedmask |= 1u << k;
                        const int r = tid + k * NT;
                        pivrow[j] = r;
                        rowpos[r] = j;
                        qrow[npiv] = r;
                    } else if (pw[k] & bmask) {
                        pw[k] ^= pw_p;
                        t[k] ^= t_p ^ qbit;
                    }
                }
                ++npiv;
                ++nrank;
                par ^= 1;
            }
            // word w is final
#pragma unroll
            for (int k = 0; k < RPT; ++k) M[(size_t)w * MRL + tid + k * NT] = pw[k];
            // ---------------- trailing phase, chunks of OSDL_CW words
            if (npiv > 0) {
                const int ngroups = (npiv + 3) >> 2;
                for (int x0 = w + 1; x0 < W; x0 += OSDL_CW) {
                    const int cw = (W - x0) < OSDL_CW ? (W - x0) : OSDL_CW;
                    __syncthreads();  // previous chunk's tables are no longer read; qrow[] is published
                    for (int idx = tid; idx < npiv * cw; idx += NT) {
                        const int q = idx / cw, xx = idx - q * cw;
                        prow[q * OSDL_CW + xx] = M[(size_t)(x0 + xx) * MRL + qrow[q]];
                    }
                    __syncthreads();
                    for (int e = tid; e < ngroups * 16 * cw; e += NT) {
                        const int g = e / (16 * cw);
                        const int rem = e - g * (16 * cw);
                        const int xx = rem >> 4, idx = rem & 15;
                        unsigned long long v = 0ull;
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk)
                            if (((idx >> kk) & 1) && (4 * g + kk) < npiv) v ^= prow[(4 * g + kk) * OSDL_CW + xx];
                        tab[(g * OSDL_CW + xx) * 16 + idx] = v;
                    }
                    __syncthreads();
#pragma unroll
                    for (int k = 0; k < RPT; ++k) {
                        if (t[k] != 0ull) {
                            for (int xx = 0; xx < cw; ++xx) {
                                unsigned long long v = 0ull;
                                for (int g = 0; g < ngroups; ++g)
                                    v ^= tab[(g * OSDL_CW + xx) * 16 + (int)((t[k] >> (4 * g)) & 15ull)];
                                M[(size_t)(x0 + xx) * MRL + tid + k * NT] ^= v;
                            }
                        }
                    }
                }
            }
            __syncthreads();  // all row updates of this panel are visible before the next panel loads
        }
        __syncthreads();
        if (P.rank_out && slot_id == 0 && tid == 0) P.rank_out[0] = nrank;

        // --------------------------------------------------------------- OSD-0 solution
        bool y[RPT];
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            y[k] = ((M[(size_t)(W - 1) * MRL + tid + k * NT] >> 63) & 1ull) != 0ull;
            const bool usedk = (usedmask >> k) & 1u;
            const unsigned long long yb = __ballot(usedk && y[k]);
            if (lane == 0) yvec[k * OSDL_NW + wave] = yb;
            if (usedk && y[k]) xout[kidx[rowpos[tid + k * NT]]] = 1;
        }
        if (tid == 0) { best64[0] = ~0ull; best64[1] = ~0ull; }
        for (int w = wave; w < W; w += OSDL_NW) {
            const int j = w * 64 + lane;
            const unsigned long long np = __ballot(j < n && pivrow[j] < 0);
            if (lane == 0) npmask[w] = np;
        }
        __syncthreads();
        if (P.out_osd0)
            for (int i = tid; i < n; i += NT) P.out_osd0[(size_t)s * n + i] = xout[i];
        int w0 = 0;
        for (int q = 0; q < NCV; ++q) w0 += __popcll(yvec[q]);

        int sel_a = -1, sel_b = -1;
        if (P.osd_method >= 2 && P.osd_order > 0) {
            const int wspan = P.osd_order < OSDL_MAXSPAN ? P.osd_order : OSDL_MAXSPAN;
            int tcount = 0;
#pragma clang loop unroll(disable)
            for (int w = 0; w < W; ++w) {
                unsigned long long npm = npmask[w];
                npm = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(npm >> 32)) << 32) |
                      (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)npm);
                if (!npm) continue;
                if (P.osd_method != 3 && tcount >= wspan) break;  // osd_e only needs the first wspan columns
                unsigned long long rw[RPT];
#pragma unroll
                for (int k = 0; k < RPT; ++k) rw[k] = M[(size_t)w * MRL + tid + k * NT];
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
                            if (lane == 0) colvec[tcount * NCV + k * OSDL_NW + wave] = cb;
                        }
                    }
                    if (lane == b) acc += cnt;
                    if (tid == 0 && tcount < 64) tpos[tcount] = w * 64 + b;
                    ++tcount;
                }
                if (P.osd_method == 3 && acc) atomicAdd(&wt[w * 64 + lane], acc);
            }
            __syncthreads();
            if (P.osd_method == 3) {
                for (int j = tid; j < n; j += NT)
                    if (pivrow[j] < 0) atomicMin(&best64[0], ((unsigned long long)wt[j] << 32) | (unsigned)j);
                const int npairs = wspan * (wspan - 1) / 2;
                for (int pidx = tid; pidx < npairs; pidx += NT) {
                    int a = 0, rem = pidx;
                    while (rem >= wspan - 1 - a) { rem -= wspan - 1 - a; ++a; }
                    const int bq = a + 1 + rem;
                    int wgt = 2;
                    for (int q = 0; q < NCV; ++q) wgt += __popcll(yvec[q] ^ colvec[a * NCV + q] ^ colvec[bq * NCV + q]);
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
                    atomicMin(&best64[0], ((unsigned long long)wgt << 32) | pat);
                }
                __syncthreads();
                const unsigned long long k1 = best64[0];
                if (k1 != ~0ull && (int)(k1 >> 32) < w0) { sel_a = -2; sel_b = (int)(k1 & 0xffffffffu); }
            }
        }

        // ------------------------------------------------- write the OSD-W solution
        if (sel_a == -1) {
            for (int i = tid; i < n; i += NT) P.out_osdw[(size_t)s * n + i] = xout[i];
        } else {
            __syncthreads();
            for (int i = tid; i < n; i += NT) xout[i] = 0;
            __syncthreads();
#pragma unroll
            for (int k = 0; k < RPT; ++k) {
                bool xs = y[k];
                const size_t rr = (size_t)tid + (size_t)k * NT;
                if (sel_a >= 0) {
                    xs ^= ((M[(size_t)(sel_a >> 6) * MRL + rr] >> (sel_a & 63)) & 1ull) != 0ull;
                    if (sel_b >= 0) xs ^= ((M[(size_t)(sel_b >> 6) * MRL + rr] >> (sel_b & 63)) & 1ull) != 0ull;
                } else {
                    unsigned int pp = (unsigned int)sel_b;
                    while (pp) {
                        const int bq = __ffs((int)pp) - 1;
                        pp &= pp - 1;
                        const int pos = tpos[bq];
                        xs ^= ((M[(size_t)(pos >> 6) * MRL + rr] >> (pos & 63)) & 1ull) != 0ull;
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
            for (int i = tid; i < n; i += NT) P.out_osdw[(size_t)s * n + i] = xout[i];
        }
        __syncthreads();
    }
}

}  // namespace bposd
