// osd_kernel.hip.h -- batched ordered-statistics decoding (OSD-0 / OSD-E / OSD-CS) for gfx950.
//
// Restates rows a8-a11 of SURVEY.md §8 (the OSD half of `.decode(syndrome)`, results
// read as `.osd0_decoding` / `.osdw_decoding`: /root/reference/README.md:202,
// /root/reference/src/bposd/css_decode_sim.py:257-258,294-295).
//
// One workgroup post-processes one non-converged syndrome at a time (persistent, pulls
// from the list the BP kernel appended to).  A thread owns RPT rows of the parity-check
// matrix (row = tid + k * blockDim.x), each as W 64-bit words HELD IN REGISTERS -- the
// columns are physically permuted into reliability order first, so "process columns in
// sorted order" becomes a sweep over bit positions.  The register array always holds the
// words from the current 64-column panel onwards (word 0 = current panel): when a panel is
// finished its word is final, is flushed to a per-thread spill area in global memory
// (only ever re-read by the thread that wrote it) and the array shifts down by one, so
// every register index is a compile-time constant while the panel loop stays a real loop.
// The syndrome rides along as the last bit of the last word.  Few, fat waves (8 for H1922,
// two per SIMD) keep the per-column bookkeeping -- which every wave executes -- cheap.
//
//   1. a8  reliability sort: bitonic network in LDS on (order-preserving u64 image of the
//          LLR, bit index) -- a strict total order, so the result equals a stable sort.
//   2. a9  blocked Gauss-Jordan, panels of 64 columns (one register word):
//          (i) panel phase, two barriers per SIX columns (round 5; one per pivot before): which pivot rows a row
//          absorbs over six columns depends only on the row's six bits there, so every unused row with a non-zero
//          value v stores (sub-block number, row) into v's claim word -- a plain LDS store, the last one to land is
//          the row the solvers see -- and its panel word and combination mask into its own slot of rowbuf; two waves
//          -- alone on their SIMDs while the others wait -- solve the 64 values after the first barrier (lane v =
//          value v: <= 6 steps of ballot, ffs, readlane, masked XOR; the claimed value whose reduced form has the
//          lowest column set supplies the pivot row -- any unused row with a 1 in the column is a valid pivot, the
//          row choice never reaches the outputs) and write one (panel word, mask) XOR per value into a table; after
//          the second barrier every row XORs the table entry of its value into its panel word and its own 64-bit
//          combination mask t (row = row_at_panel_start ^ XOR_{q in t} pivot_q_at_panel_start);
//          tools/panel_subblock_model.py replays the scheme lane by lane against column-by-column elimination;
//          (ii) trailing phase, once per panel: the <= 64 pivot rows publish their
//          trailing words, 4-bit "four Russians" tables of their XOR combinations are
//          built in LDS, and every row applies its mask with one table lookup per 4
//          pivots and word.  A not-yet-used row has no support left of the sweep position,
//          so only words >= the panel word are ever touched.  Stops after `rank` pivots.
//          Pivot columns = the greedy independent set in sorted order, exactly the set
//          any row-pivoting strategy finds; OSD-0 = the syndrome bit of each pivot row.
//   3. a10/a11  OSD-W: in reduced form the solution for a candidate that switches on
//          non-pivot columns {t} is  x_S = y ^ XOR_t A_t,  so its weight is a popcount:
//          singles via per-wave ballots accumulated per lane, pairs / exhaustive patterns
//          via transposed column bit-vectors.  Selection is the lexicographic minimum of
//          (weight, enumeration index) with OSD-0 first, i.e. "replace only if strictly
//          lighter, first found wins".
//
// Integer / bitwise throughout for uniform channel probabilities: weights are Hamming weights,
// which order candidates exactly like sum log(1/p) then (host checks this).  With non-uniform
// probabilities (P.cost != null) candidate weights are fp64 sums of log(1/p_i) over the set bits
// of the candidate accumulated in ascending bit index, the reference's order, one candidate per
// lane: bit-identical to the CPU sum, hence the same strict-< winner.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace bposd {

constexpr int OSD_RPT = 2;      // rows per thread
constexpr int OSD_MAXW = 8;     // max waves per workgroup (512 threads)
#ifndef OSD_CHUNK
#define OSD_CHUNK 8
#endif
constexpr int OSD_MAXCV = 16;   // max ballot words per column vector (RPT * waves)

struct OsdParams {
    int m, n;
    int rank;
    int osd_method;  // 1 osd0, 2 osd_e, 3 osd_cs
    int osd_order;
    int tie_policy;
    int e_msb_first;  // osd_e enumeration: 0 = bit b of pattern i is T position b (LSB first), 1 = T position w - 1 - b
    const uint8_t* __restrict__ synd;  // [B, m]
    const int* __restrict__ rp;        // CSR indptr [m+1]
    const int* __restrict__ ci;        // CSR indices [E]
    const double* __restrict__ llr_ws; // [cap, n]
    const int* __restrict__ osd_list;  // [cap]
    int* __restrict__ counters;        // [1] = number of list entries, [2] = OSD work queue
    uint8_t* __restrict__ out_osd0;    // [B, n] nullable
    uint8_t* __restrict__ out_osdw;
    // nullable: the same two rows again at [list slot][n] -- the host-pointer API downloads the bulk outputs right after BP
    // and patches the few OSD rows from these compact copies afterwards
    uint8_t* __restrict__ cmp_osd0;
    uint8_t* __restrict__ cmp_osdw;    // [B, n]
    const double* __restrict__ cost;   // nullable: log(1/p_i) per bit -> fp64 weights summed in bit order
                                       // (ldpc v2 weight function with non-uniform channel_probs)
    const uint8_t* __restrict__ sel;   // [B, n] nullable: per-syndrome choice between cost and cost_alt
    const double* __restrict__ cost_alt;
    unsigned long long* __restrict__ rows_ws;  // [gridDim.x][W][blockDim.x * RPT] finished row words
    long long* __restrict__ dbg;       // nullable: 8 phase timestamps (s_memtime) of list slot 0
    int packed_io;  // 1: synd is [B][ceil(m/64)] and out_osd0 / out_osdw / cmp_* are rows of ceil(n/64) little-endian 64-bit words
};

__device__ __forceinline__ bool osd_synd_bit(const uint8_t* synd, int packed, long long s, int m, int r) {
    if (packed) return (((const unsigned long long*)synd)[(size_t)s * (size_t)((m + 63) >> 6) + (r >> 6)] >> (r & 63)) & 1ull;
    return (synd[(size_t)s * m + r] & 1) != 0;
}

// one result row from its 0/1 bytes in LDS: n bytes, or ceil(n/64) words in the packed form
__device__ __forceinline__ void osd_store_row(uint8_t* out, int packed, size_t row, int n, const uint8_t* x, int tid, int nthreads) {
    if (!out) return;
    if (packed) {
        const int wpn = (n + 63) >> 6;
        for (int w = tid; w < wpn; w += nthreads) {
            unsigned long long v = 0ull;
            for (int b = 0; b < 64 && 64 * w + b < n; ++b) v |= (unsigned long long)(x[64 * w + b] & 1) << b;
            ((unsigned long long*)out)[row * wpn + w] = v;
        }
    } else {
        for (int i = tid; i < n; i += nthreads) out[row * n + i] = x[i];
    }
}

// Diagnostics (phase timestamps + a dump of the sweep tables for list slot 0) are compiled in only with
// -DBPOSD_OSD_DIAG: their address arithmetic otherwise costs registers in the hot loops.
#ifdef BPOSD_OSD_DIAG
#define OSD_STAMP(k)                                                                              \
    do {                                                                                          \
        if (P.dbg && tid == 0 && slot_id == 0) P.dbg[k] = (long long)__builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define OSD_STAMP(k) do { } while (0)
#endif

__device__ __forceinline__ unsigned long long llr_sort_key(double x) {
    // order-preserving map double -> u64 (x + 0.0 folds -0.0 into +0.0: they compare equal).  A NaN (product-sum
    // without clipping) compares "equal" to everything in the reference's comparator; all NaNs share the key of the
    // padding entries, i.e. they sort last and keep their index order -- for an all-NaN vector (what a non-converged
    // product-sum run ends in) that is the identity order the reference's stable sort returns.
    if (x != x) return ~0ull;
    const unsigned long long u = (unsigned long long)__double_as_longlong(x + 0.0);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}

// Minimum of a 32-bit value over the 64 lanes of a wave, returned wave-uniform: an xor / rotate butterfly inside each row
// of 16 lanes on the DPP path (every source lane is valid, so no identity value is involved), then the four row results
// through SGPRs.
__device__ __forceinline__ unsigned int osd_wave_min_u32(unsigned int v) {
    // (s_nop 1: a DPP source written by the preceding VALU instruction needs two wait states)
    asm volatile("s_nop 1\n\tv_min_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(v));
    asm volatile("s_nop 1\n\tv_min_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "+v"(v));
    asm volatile("s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf" : "+v"(v));
    asm volatile("s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(v));
    const unsigned int r0 = (unsigned int)__builtin_amdgcn_readlane((int)v, 0);
    const unsigned int r1 = (unsigned int)__builtin_amdgcn_readlane((int)v, 16);
    const unsigned int r2 = (unsigned int)__builtin_amdgcn_readlane((int)v, 32);
    const unsigned int r3 = (unsigned int)__builtin_amdgcn_readlane((int)v, 48);
    const unsigned int a = r0 < r1 ? r0 : r1, b = r2 < r3 ? r2 : r3;
    return a < b ? a : b;
}

// Lowest set bit of a 64-bit word, 64 if there is none: v_ffbl_b32 returns ~0 for a zero input, the saturating add keeps
// that above 64, one v_min3 finishes.
__device__ __forceinline__ unsigned int osd_ffs64_or_64(unsigned long long x) {
    unsigned int lo, hi;
    asm("v_ffbl_b32 %0, %1" : "=v"(lo) : "v"((unsigned int)x));
    asm("v_ffbl_b32 %0, %1" : "=v"(hi) : "v"((unsigned int)(x >> 32)));
    asm("v_add_u32_e64 %0, %1, 32 clamp" : "=v"(hi) : "v"(hi));
    return min(min(lo, hi), 64u);
}

// A 64-bit value every lane holds alike, moved into SGPRs so that loops over its bits are scalar (readfirstlane returns
// int: go through unsigned before widening, a set bit 31 must not sign-extend into the high word).
__device__ __forceinline__ unsigned long long osd_uniform64(unsigned long long x) {
    return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(x >> 32)) << 32) |
           (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)x);
}

// Bit-sliced addition across lanes: c[0 .. NB-1] are the bits of 32 independent counters (one per bit position); the
// partner lane's counters arrive through DPP control CTRL (every source lane valid) and the sum has NB + 1 bits.
template <int CTRL, int NB>
__device__ __forceinline__ void osd_bs_add(unsigned int (&c)[6]) {
    unsigned int carry = 0u;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const unsigned int a = c[i];
        const unsigned int b = (unsigned int)__builtin_amdgcn_update_dpp((int)a, (int)a, CTRL, 0xF, 0xF, false);
        if (i == 0) {
            c[0] = a ^ b;
            carry = a & b;
        } else {
            c[i] = a ^ b ^ carry;
            carry = (a & b) | (carry & (a ^ b));
        }
    }
    c[NB] = carry;
}

// LDS carve-up, all offsets 8-byte aligned
struct OsdLds {
    unsigned long long* keys;     // [nsort]; after the sort: first 128 B reused as T-index -> position
    unsigned long long* pbuf;     // [2][64][2]  per 6-bit value v: tables D and D' of the current sub-block
    unsigned int* pcol;           // [2][64]     per value: (sub-block << 10 | row) of its last claim; pivot info
    unsigned long long* prow;     // [64][W rounded up to even]   trailing words of this panel's pivots
    unsigned long long* tab;      // [16][16][W | 1] XOR combinations of 4 pivots: [group][combination][word]
    unsigned long long* colvec;   // [64][OSD_MAXCV]
    unsigned long long* yvec;     // [OSD_MAXCV]
    unsigned long long* npmask;   // [W] non-pivot positions per word
    unsigned long long* best64;   // [2]
    int* wt;                      // [32 * W] 16-bit weight of the single candidate at each sorted position, two per word
    int* misc;                    // [8]
    unsigned short* kidx;         // [nsort]  -> order[j] after the sort
    short* pivrow;                // [nsort] sorted position -> pivot row, -1 if non-pivot
    unsigned short* inv;          // [n] original column -> sorted position
    unsigned char* xout;          // [n]
};

__host__ __device__ constexpr size_t osd_align8(size_t x) { return (x + 7) & ~(size_t)7; }

// sort size: the power of two >= 64 * W (n <= 64 W - 1).  Compile-time, so that every LDS offset below is
// an instruction immediate instead of a live register.
__host__ __device__ constexpr int osd_nsort(int W) {
    int ns = 2;
    while (ns < 64 * W) ns <<= 1;
    return ns;
}

// Row stride (in words) of the four-Russians tables [16 groups][16 combinations][stride]: the smallest odd number >= W.
__host__ __device__ constexpr int osd_tab_stride(int W) { return W | 1; }

// Elimination-phase buffers (prow, tab, colvec) and the fp64-weight tables of the non-uniform-channel
// path are never live at the same time: they share one union region placed last in the carve-up.
__host__ __device__ constexpr size_t osd_union_bytes(int nsort, int W, int mr) {
    const size_t elim = (size_t)64 * ((W + 1) & ~1) * 8 + (size_t)16 * 16 * osd_tab_stride(W) * 8 + (size_t)64 * OSD_MAXCV * 8;
    const size_t fpw = (size_t)mr * 8        // am: per-row entries in the first <= 64 non-pivot columns
                     + (size_t)nsort * 8     // costs
                     + (size_t)nsort * 8     // Mi
                     + (size_t)64 * W * 8    // wd: single-candidate weights
                     + (size_t)2016 * 8      // pair weights (order <= 64)
                     + osd_align8((size_t)nsort * 2);  // info
    return elim > fpw ? elim : fpw;
}

// rowbuf[mr] (16 bytes per row: the claimants' panel words and masks during the panel phase) lies over the sort keys, which are
// dead between the sort and the sweep; a shape with more rows than nsort / 2 (many short rows) gets a region of its own at the end.
__host__ __device__ constexpr size_t osd_rowbuf_extra(int nsort, int mr) {
    return (size_t)mr * 16 > (size_t)nsort * 8 ? (size_t)mr * 16 + 16 : 0;
}

__host__ __device__ constexpr size_t osd_lds_bytes(int W, int mr) {
    const int nsort = osd_nsort(W);
    size_t b = 0;
    b += (size_t)nsort * 8;                     // keys
    b += (size_t)2 * 64 * 2 * 8;                // pbuf
    b += (size_t)2 * 64 * 4;                    // pcol
    b += (size_t)OSD_MAXCV * 8;                 // yvec
    b += (size_t)W * 8;                         // npmask
    b += 2 * 8;                                 // best64
    b += (size_t)64 * W * 4;                    // wt
    b += 8 * 4;                                 // misc
    b += (size_t)nsort * 2;                     // kidx
    b += (size_t)nsort * 2;                     // pivrow
    b += (size_t)nsort * 2;                     // inv
    b += (size_t)nsort;                         // xout
    b = (b + 15) & ~(size_t)15;                 // the union region is 16-byte aligned (16-byte pivot-row writes)
    b += osd_union_bytes(nsort, W, mr);         // prow | tab | colvec  /  fp64-weight tables
    b += osd_rowbuf_extra(nsort, mr);           // rowbuf, where the (dead) sort keys cannot hold it
    return b + 64;
}

// Trailing update of one thread's rows with the four-Russians tables of a panel: for every group of 4 pivots one look-up
// row (chosen by 4 bits of the row's combination mask) is XORed into the first NCH chunks of OSD_CHUNK trailing words.
// One straight-line variant per chunk count: a chunk's loads are issued together and waited for one by one (loading a
// chunk ahead was measured and is no faster: the pass is bound by LDS / VALU throughput, not latency).  Words past the
// live window only ever receive table words that are never read.
// osd_e enumerates patterns i = 1 .. 2^w - 1 and keeps the FIRST lightest one (strict <), so the tie between equally
// light patterns depends on which T position bit b of i stands for.  `mask` has bit p set when T position p is switched
// on; the enumeration index of that pattern is mask itself when bit b <-> position b (LSB first: the restatement's
// reading of upstream) and the w-bit reversal of mask when bit b <-> position w - 1 - b.  The
// reversal is an involution, so the same function maps a winning index back to its column mask.
__device__ __forceinline__ unsigned int osd_e_index(unsigned int mask, int w, int msb_first) {
    return msb_first ? (__brev(mask) >> (32 - w)) : mask;
}

template <int W, int NCH>
__device__ __forceinline__ void osd_apply_tables(unsigned long long (&row)[OSD_RPT][W], const unsigned long long (&t)[OSD_RPT],
                                                 const unsigned long long* tab, int ngroups) {
    static_assert((W - 1 + OSD_CHUNK - 1) / OSD_CHUNK <= 4, "osd_apply_tables is instantiated for up to 4 chunks");
    constexpr int WS = osd_tab_stride(W);
    constexpr int NC = NCH < (W - 1 + OSD_CHUNK - 1) / OSD_CHUNK ? NCH : (W - 1 + OSD_CHUNK - 1) / OSD_CHUNK;
#pragma clang loop unroll(disable)
    for (int g = 0; g < ngroups; ++g) {
#pragma unroll
        for (int k = 0; k < OSD_RPT; ++k) {
            const int idx = (int)((t[k] >> (4 * g)) & 15ull);
            // the row's LDS address as one opaque register: every look-up below is then an immediate offset from it
            typedef const volatile __attribute__((address_space(3))) unsigned long long* lds_cptr;
            unsigned int addr = (unsigned int)(size_t)(lds_cptr)(tab + (size_t)(g * 16 + idx) * WS);
            asm volatile("" : "+v"(addr));
            const lds_cptr tg = (lds_cptr)(size_t)addr;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int x0 = 1 + c * OSD_CHUNK;
                unsigned long long buf[OSD_CHUNK];
#pragma unroll
                for (int i = 0; i < OSD_CHUNK; ++i)
                    if (x0 + i < W) buf[i] = tg[x0 + i];
#pragma unroll
                for (int i = 0; i < OSD_CHUNK; ++i)
                    if (x0 + i < W) row[k][x0 + i] ^= buf[i];
                // keep only one chunk's table loads in flight (else all 2 x 31 loads are hoisted and the 2 x 32-word
                // register window spills)
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
}

// PACKED: the packed-I/O form (OsdParams::packed_io) as a compile-time switch -- the byte form's register allocation (256 VGPRs
// at W = 31) does not pay for it
template <int W, bool PACKED = false>
__global__ __launch_bounds__(64 * OSD_MAXW) void osd_kernel(const OsdParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int RPT = OSD_RPT;
    const int m = P.m, n = P.n;
    constexpr int NS = osd_nsort(W);
    constexpr int WS = osd_tab_stride(W);
    constexpr int PS = (W + 1) & ~1;  // even row stride of prow: pivot rows are published in 16-byte pairs
    const int NT = blockDim.x;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int nwaves = NT >> 6;
    const int ncv = nwaves * RPT;  // ballot words per column vector; word index = k * nwaves + wave

    OsdLds L;
    {
        unsigned char* p = smem;
        L.keys = (unsigned long long*)p; p += (size_t)NS * 8;
        L.pbuf = (unsigned long long*)p; p += (size_t)2 * 64 * 2 * 8;
        L.pcol = (unsigned int*)p; p += (size_t)2 * 64 * 4;
        L.yvec = (unsigned long long*)p; p += (size_t)OSD_MAXCV * 8;
        L.npmask = (unsigned long long*)p; p += (size_t)W * 8;
        L.best64 = (unsigned long long*)p; p += 2 * 8;
        L.wt = (int*)p; p += (size_t)64 * W * 4;
        L.misc = (int*)p; p += 8 * 4;
        L.kidx = (unsigned short*)p; p += (size_t)NS * 2;
        L.pivrow = (short*)p; p += (size_t)NS * 2;
        L.inv = (unsigned short*)p; p += (size_t)NS * 2;
        L.xout = p; p += (size_t)NS;
        // union region (last): elimination buffers ...
        p = smem + (((size_t)(p - smem) + 15) & ~(size_t)15);
        L.prow = (unsigned long long*)p;
        L.tab = L.prow + (size_t)64 * PS;
        L.colvec = L.tab + (size_t)16 * 16 * WS;
    }
    // the claimants' (panel word, mask) pairs of the current sub-block, one slot per row
    ulonglong2* const rowbuf = osd_rowbuf_extra(NS, NT * RPT)
                                   ? reinterpret_cast<ulonglong2*>(smem + ((((size_t)((unsigned char*)L.prow - smem) + osd_union_bytes(NS, W, NT * RPT)) + 15) & ~(size_t)15))
                                   : reinterpret_cast<ulonglong2*>(L.keys);
    for (;;) {
        if (tid == 0) L.misc[0] = atomicAdd(&P.counters[2], 1);
        __syncthreads();
        const int slot_id = L.misc[0];
        const int nlist = P.counters[1];
        if (slot_id >= nlist) break;  // uniform
        const long long s = P.osd_list[slot_id];
        const double* llr = P.llr_ws + (size_t)slot_id * n;

        OSD_STAMP(0);
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
        }
        for (int i = tid; i < 32 * W; i += NT) L.wt[i] = 0x00010001;  // 16-bit single-candidate weights, two per word: 1 each
        for (int i = tid; i < 2 * 64; i += NT) L.pcol[i] = 0u;          // no value claimed yet: the panel phase's sub-blocks count from 1
        unsigned int sbc = 1u;  // number of the current six-column sub-block of this elimination (uniform; <= 11 * W)
        __syncthreads();
        for (int k = 2; k <= NS; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int t = tid; t < (NS >> 1); t += NT) {
                    // pair (lo, hi = lo | j) with bit j of lo clear
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

        OSD_STAMP(1);
        // ------------------------------------------- build my rows in sorted column order
        unsigned long long row[RPT][W];
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
#pragma unroll
            for (int w = 0; w < W; ++w) row[k][w] = 0ull;
            const int r = tid + k * NT;
            if (r < m) {
                const int e0 = P.rp[r], e1 = P.rp[r + 1];
                for (int e = e0; e < e1; ++e) {
                    const int j = L.inv[P.ci[e]];
                    const int jw = j >> 6;
                    const unsigned long long bit = 1ull << (j & 63);
#pragma unroll
                    for (int w = 0; w < W; ++w) row[k][w] |= (jw == w) ? bit : 0ull;
                }
                if (osd_synd_bit(P.synd, PACKED ? 1 : 0, s, m, r)) row[k][W - 1] |= 1ull << 63;
            }
        }

        OSD_STAMP(2);
        // ------------------------------------------------------- a9: blocked Gauss-Jordan
        const int MR = NT * RPT;  // padded row count of the spill layout
        unsigned long long* ws = P.rows_ws + (size_t)blockIdx.x * W * MR;  // [W][MR], word-major: coalesced
        // per-row pivot state during the elimination: pinfo = -1 while the row is unused, else (sorted position << 6) |
        // pivot index in its panel
        int pinfo[RPT];
#pragma unroll
        for (int k = 0; k < RPT; ++k) pinfo[k] = -1;
        int nrank = 0;
        bool done = false;
#ifdef BPOSD_OSD_DIAG
        long long diag_panel = 0, diag_trail = 0, diag_t0 = 0, diag_pub = 0, diag_build = 0, diag_claim = 0, diag_solve = 0, diag_absorb = 0, diag_t1 = 0, diag_steps = 0;
#define OSD_TICK() ((long long)__builtin_amdgcn_s_memtime())
#endif
#pragma clang loop unroll(disable)
        for (int w = 0; w < W; ++w) {
            // Once `done` (rank reached / past column n) the remaining iterations only flush word 0 and
            // shift: every word leaves through the same single store, no tail flush with W addresses.
            const int nvalid = W - 1 - w;  // live trailing words: row[k][1 .. nvalid]
            if (!done) {
#ifdef BPOSD_OSD_DIAG
            diag_t0 = OSD_TICK();
#endif
            // ---------------- (i) panel phase on the current word row[k][0]: two barriers per SIX columns.
            // What a row has to absorb while six columns are eliminated depends only on its six bits there (its
            // "value" v): all rows with the same v receive the same combination of the sub-block's <= 6 pivot rows
            // as they stood at its start.  (A) every unused row with v != 0 stores its number into v's claim word and
            // its (panel word, mask) into its own slot of rowbuf.  (C) behind the first barrier waves 0 and 1 -- alone on
            // their SIMDs while the others wait -- solve the 64 values, lane v = value v: for column j the lowest claimed
            // value whose reduced form has bit j set supplies the pivot row, the row whose claim landed last (none: the
            // column is non-pivot -- unused rows are zero in every column already passed, and every unused row's value
            // is a claimed one), every value with bit j set absorbs it; a second register follows the pivot rows
            // themselves.  Wave 0 then writes, per value, the XOR of the (word, mask) pairs its tag names (table D),
            // wave 1 the same for the pivot row taken from each value (table D').  (E) behind the second
            // barrier a row XORs ONE table entry, picked by its value (the claimant of a taken value: from D').
            unsigned long long t[RPT];
#pragma unroll
            for (int k = 0; k < RPT; ++k) t[k] = 0ull;
            int npiv = 0;  // pivots found in this panel (uniform)
            const int nb = n - w * 64;            // valid columns in this panel; the syndrome bit (bit 63 of the last
            const int nbc = nb < 64 ? nb : 64;    // word) lies beyond them and never takes part in a value
            if (nb <= 0 || nrank >= P.rank) done = true;
            {
                ulonglong2* const pub = reinterpret_cast<ulonglong2*>(L.pbuf);  // [64] table D, [64] table D'
                unsigned int* const claim = L.pcol;                              // [64] claims, [64] per-value pivot info
                const int nsolve = nwaves < 2 ? nwaves : 2;
                const int wave_u = __builtin_amdgcn_readfirstlane(wave);  // (the compiler cannot see that tid >> 6 is wave-uniform)
#pragma clang loop unroll(disable)
                for (int c0 = 0; c0 < nbc && !done; c0 += 6, ++sbc) {
                    const int wsb = nbc - c0 < 6 ? nbc - c0 : 6;
                    const unsigned int vmask = (1u << wsb) - 1u;
#ifdef BPOSD_OSD_DIAG
                    diag_t1 = OSD_TICK();
#endif
                    // (A) claims: a plain store of (sub-block number, row) per claimed value -- whichever claimant's store lands
                    // last is the row the solvers see -- and every claimant's (panel word, mask) into its own slot of rowbuf:
                    // nothing to wait for before the barrier (an atomic with return, the first form of this, cost a round trip)
                    unsigned int bv[RPT];
#pragma unroll
                    for (int k = 0; k < RPT; ++k) {
                        bv[k] = (unsigned int)(row[k][0] >> c0) & vmask;
                        if (pinfo[k] < 0 && bv[k] != 0u) {  // rows beyond m are zero: they never claim
                            claim[bv[k]] = (sbc << 10) | (unsigned int)(tid + k * NT);
                            rowbuf[tid + k * NT] = make_ulonglong2(row[k][0], t[k]);
                        }
                    }
                    __syncthreads();
#ifdef BPOSD_OSD_DIAG
                    { const long long t2 = OSD_TICK(); diag_claim += t2 - diag_t1; diag_t1 = t2; }
#endif
                    if (wave_u < nsolve) {
                        // (C) X, lane v: reduced value (bits 0-5) and tag (bits 8-13) of the rows that hold value v; Y, lane
                        // v: the same for the pivot row taken from value v, with its column (bits 16-18) and bit 20 set, 0 if
                        // none was.  Tag bit j = "has absorbed the pivot row of the sub-block's column j as it stood at the
                        // sub-block's start".  A value has no bit beyond
                        // the valid columns, so j >= wsb never finds a candidate.  (The same on the scalar unit -- value sets
                        // as 64-bit masks, s_bitcmp1 / s_cselect / s_xor -- was built and measured no faster: a lone wave
                        // issues about one instruction per eight cycles whatever the unit, and this form has fewer.)
                        const unsigned int cw = claim[lane];  // last claim of value `lane`: (sub-block number << 10) | row
                        const unsigned long long avm0 = __ballot((cw >> 10) == sbc) & ~1ull;  // claimed values
                        unsigned int X = (unsigned int)lane, Y = 0u;
                        unsigned int pivmask = 0u;  // columns of the sub-block that are pivot columns (uniform)
                        ulonglong2 pp[6];  // (panel word, mask) of column j's pivot row as it stood at the sub-block's start: requested the
                                           // moment the row is known, so that the table pass below finds them loaded
#pragma unroll
                        for (int j = 0; j < 6; ++j) pp[j] = make_ulonglong2(0ull, 0ull);
                        // LIMITED: fewer than six pivots are left to find (the elimination's last sub-block): count them down
                        auto solve = [&](auto limited) {
                            unsigned long long avm = avm0;
                            int room = P.rank - nrank;  // pivots still to find (> 0 here)
#pragma unroll
                            for (int j = 0; j < 6; ++j) {
                                unsigned int mh = (unsigned int)((int)(X << (31 - j)) >> 31);  // all ones where bit j of the reduced value is set
                                asm volatile("" : "+v"(mh));
                                const unsigned long long cand = __ballot(mh != 0u) & avm;
                                if (cand) {  // uniform
                                    const int l = __ffsll((long long)cand) - 1;
                                    const unsigned int ppk = (unsigned int)__builtin_amdgcn_readlane((int)X, l) & 0x3f3fu;
                                    const unsigned int pn = ppk ^ (0x100u << j);
                                    X = __builtin_amdgcn_bitop3_b32(pn, X, mh, 0x6c);  // X ^= pn & mh
                                    unsigned int my = (unsigned int)((int)(Y << (31 - j)) >> 31);
                                    asm volatile("" : "+v"(my));
                                    Y = __builtin_amdgcn_bitop3_b32(pn, Y, my, 0x6c);
                                    // (a value is taken at most once: Y was 0 in lane l)
                                    unsigned int ml = lane == l ? ~0u : 0u;
                                    asm volatile("" : "+v"(ml));
                                    Y |= (ppk | ((unsigned int)j << 16) | (1u << 20)) & ml;
                                    pp[j] = rowbuf[__builtin_amdgcn_readlane((int)cw, l) & 1023];
                                    pivmask |= 1u << j;
                                    if (decltype(limited)::value && --room == 0) avm = 0ull;  // rank reached: no further pivots
                                }
                            }
                        };
#ifdef BPOSD_OSD_DIAG
                        const long long ts0 = OSD_TICK();
#endif
                        if (P.rank - nrank >= 6) solve(std::false_type{});
                        else solve(std::true_type{});
#ifdef BPOSD_OSD_DIAG
                        diag_steps += OSD_TICK() - ts0;
#endif
                        if (wave_u == 0) {
                            // per value: if a pivot row was taken from it, the sub-block's number, that row and its column; else 0
                            claim[64 + lane] = ((Y >> 20) & 1u) * ((sbc << 13) | (((Y >> 16) & 7u) << 10) | (cw & 1023u));
                            if (lane == 0) L.misc[1] = (int)pivmask;
                        }
                        if (pivmask) {  // uniform
                            for (int which = wave_u; which < 2; which += nsolve) {  // 0: table D from X's tags, 1: table D' from Y's
                                const unsigned int tg = ((which ? Y : X) >> 8) & 63u;
                                unsigned int alo = 0u, ahi = 0u, blo = 0u, bhi = 0u;
#pragma unroll
                                for (int j = 0; j < 6; ++j) {
                                    unsigned int msk = (unsigned int)((int)(tg << (31 - j)) >> 31);
                                    asm volatile("" : "+v"(msk));
                                    alo = __builtin_amdgcn_bitop3_b32((unsigned int)pp[j].x, alo, msk, 0x6c);  // alo ^= word & msk
                                    ahi = __builtin_amdgcn_bitop3_b32((unsigned int)(pp[j].x >> 32), ahi, msk, 0x6c);
                                    blo = __builtin_amdgcn_bitop3_b32((unsigned int)pp[j].y, blo, msk, 0x6c);
                                    bhi = __builtin_amdgcn_bitop3_b32((unsigned int)(pp[j].y >> 32), bhi, msk, 0x6c);
                                }
                                // the absorbed pivots themselves: tag bit j stands for the panel's pivot number npiv + (pivot columns below j)
                                unsigned int tgc = tg;
                                if (pivmask != vmask) {  // uniform, rare (a non-pivot column inside the sub-block): squeeze the tag's bits together
                                    tgc = 0u;
                                    int cnt = 0;
#pragma unroll
                                    for (int j = 0; j < 6; ++j)
                                        if ((pivmask >> j) & 1u) { tgc |= ((tg >> j) & 1u) << cnt; ++cnt; }
                                }
                                const unsigned long long own = (unsigned long long)tgc << npiv;
                                pub[64 * which + lane] = make_ulonglong2(((unsigned long long)ahi << 32) | alo, (((unsigned long long)bhi << 32) | blo) ^ own);
                            }
                        }
                    }
                    __syncthreads();
#ifdef BPOSD_OSD_DIAG
                    { const long long t2 = OSD_TICK(); diag_solve += t2 - diag_t1; diag_t1 = t2; }
#endif
                    const unsigned int pivmask_all = (unsigned int)__builtin_amdgcn_readfirstlane(L.misc[1]);
                    if (pivmask_all) {  // uniform
                        // (E) one table entry per row
#pragma unroll
                        for (int k = 0; k < RPT; ++k) {
                            const unsigned int inf = claim[64 + bv[k]];
                            ulonglong2 d = pub[bv[k]];
                            if (inf == ((sbc << 13) | (inf & 0x1c00u) | (unsigned int)(tid + k * NT))) {  // the row the solvers took for column c0 + jc
                                const unsigned int jc = (inf >> 10) & 7u;
                                d = pub[64 + bv[k]];
                                pinfo[k] = ((w * 64 + c0 + (int)jc) << 6) | (npiv + __popc(pivmask_all & ((1u << jc) - 1u)));
                            }
                            row[k][0] ^= d.x;
                            t[k] ^= d.y;
                        }
                        const int nps = __popc(pivmask_all);
                        npiv += nps;
                        nrank += nps;
                        if (nrank >= P.rank) done = true;
                    }
#ifdef BPOSD_OSD_DIAG
                    diag_absorb += OSD_TICK() - diag_t1;
#endif
                }
            }
#ifdef BPOSD_OSD_DIAG
            { const long long t1 = OSD_TICK(); diag_panel += t1 - diag_t0; diag_t0 = t1; }
#endif
            // ---------------- (ii) trailing phase on row[k][1 .. nvalid], in chunks of 8 words
            if (nvalid > 0 && npiv > 0) {
#pragma unroll
                for (int k = 0; k < RPT; ++k) {
                    if (pinfo[k] >= 0 && (pinfo[k] >> 12) == w) {  // became a pivot row in this panel
                        const int myq = pinfo[k] & 63;
#pragma unroll
                        for (int x = 0; x < W; x += 2)  // word pairs (word 0 rides along unused): 16-byte writes
                            if (x <= nvalid)
                                *reinterpret_cast<ulonglong2*>(&L.prow[myq * PS + x]) = make_ulonglong2(row[k][x], x + 1 < W ? row[k][x + 1] : 0ull);
                    }
                }
                __syncthreads();
#ifdef BPOSD_OSD_DIAG
                { const long long t1 = OSD_TICK(); diag_pub += t1 - diag_t0; diag_t0 = t1; }
#endif
                // Tables of the XOR combinations of 4 pivots, laid out [group][combination][word] with an odd row
                // stride: a thread owns (group, word), reads the four pivot words once and writes all 16 combinations
                // (lanes run along the words: conflict-free writes); the look-ups below hit <= 16 different rows at
                // the same word offset, which the odd stride spreads over different banks.
                const int ngroups = (npiv + 3) >> 2;
                for (int e = tid; e < ngroups * 32; e += NT) {
                    const int g = e >> 5;
                    const int x = 1 + (e & 31);
                    if (x <= nvalid) {
                        unsigned long long pv[4];
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk) pv[kk] = (4 * g + kk) < npiv ? L.prow[(4 * g + kk) * PS + x] : 0ull;
                        unsigned long long* tg = L.tab + (size_t)g * 16 * WS + x;
                        unsigned long long c[16];
                        c[0] = 0ull;
#pragma unroll
                        for (int idx = 1; idx < 16; ++idx) {
                            const int low = idx & (-idx);  // lowest set bit: 1, 2, 4, 8
                            c[idx] = c[idx ^ low] ^ pv[low == 1 ? 0 : low == 2 ? 1 : low == 4 ? 2 : 3];
                        }
#pragma unroll
                        for (int idx = 0; idx < 16; ++idx) tg[idx * WS] = c[idx];
                    }
                }
                __syncthreads();
#ifdef BPOSD_OSD_DIAG
                { const long long t1 = OSD_TICK(); diag_build += t1 - diag_t0; diag_t0 = t1; }
#endif
                // one straight-line variant per number of live chunks (uniform), so that no branch cuts the pipeline
                switch ((nvalid + OSD_CHUNK - 1) / OSD_CHUNK) {
                    case 1: osd_apply_tables<W, 1>(row, t, L.tab, ngroups); break;
                    case 2: osd_apply_tables<W, 2>(row, t, L.tab, ngroups); break;
                    case 3: osd_apply_tables<W, 3>(row, t, L.tab, ngroups); break;
                    default: osd_apply_tables<W, 4>(row, t, L.tab, ngroups); break;
                }
            }
#ifdef BPOSD_OSD_DIAG
            diag_trail += OSD_TICK() - diag_t0;
#endif
            }  // if (!done)
            // ---------------- word w is final: flush it and shift the register window down
#pragma unroll
            for (int k = 0; k < RPT; ++k) {
                ws[(unsigned)(w * MR + tid + k * NT)] = row[k][0];
#pragma unroll
                for (int x = 0; x + 1 < W; ++x) row[k][x] = row[k][x + 1];
                row[k][W - 1] = 0ull;
            }
        }
        OSD_STAMP(3);
#ifdef BPOSD_OSD_DIAG
        if (P.dbg && tid == 0 && slot_id == 0) { P.dbg[1190] = diag_panel; P.dbg[1191] = diag_trail; P.dbg[1192] = nrank; P.dbg[1193] = diag_pub; P.dbg[1194] = diag_build; P.dbg[1195] = diag_claim; P.dbg[1196] = diag_solve; P.dbg[1197] = diag_absorb; P.dbg[1189] = diag_steps; }
#endif
        bool used[RPT];
        int mypos[RPT];
#pragma unroll
        for (int k = 0; k < RPT; ++k) { used[k] = pinfo[k] >= 0; mypos[k] = pinfo[k] >> 6; }
        bool y[RPT];
#pragma unroll
        for (int k = 0; k < RPT; ++k) y[k] = ((ws[(unsigned)((W - 1) * MR + tid + k * NT)] >> 63) & 1ull) != 0ull;

        // --------------------------------------------------------------- OSD-0 solution
        for (int i = tid; i < n; i += NT) L.xout[i] = 0;
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            const unsigned long long yb = __ballot(used[k] && y[k]);
            if (lane == 0) L.yvec[k * nwaves + wave] = yb;
        }
        if (tid == 0) { L.best64[0] = ~0ull; L.best64[1] = ~0ull; }
#pragma unroll
        for (int k = 0; k < RPT; ++k)
            if (used[k]) L.pivrow[mypos[k]] = (short)(tid + k * NT);
        __syncthreads();  // also makes every pivrow[] write visible
#pragma unroll
        for (int k = 0; k < RPT; ++k)
            if (used[k] && y[k]) L.xout[L.kidx[mypos[k]]] = 1;
        // non-pivot position masks, one 64-bit word per panel
        for (int w = wave; w < W; w += nwaves) {
            const int j = w * 64 + lane;
            const unsigned long long np = __ballot(j < n && L.pivrow[j] < 0);
            if (lane == 0) L.npmask[w] = np;
        }
        __syncthreads();
        osd_store_row(P.out_osd0, PACKED ? 1 : 0, (size_t)s, n, L.xout, tid, NT);
        osd_store_row(P.cmp_osd0, PACKED ? 1 : 0, (size_t)slot_id, n, L.xout, tid, NT);

        int w0 = 0;
        for (int q = 0; q < ncv; ++q) w0 += __popcll(L.yvec[q]);

        OSD_STAMP(4);
        int sel_a = -1, sel_b = -1;  // sorted positions switched on by the winning candidate
        if (P.osd_method >= 2 && P.osd_order > 0) {
            // ---------------- a10/a11: singles weights + transposed columns of the first w non-pivots
            const int wspan = P.osd_order < 64 ? P.osd_order : 64;
            unsigned short* tpos = (unsigned short*)L.keys;  // T-index -> sorted position (first 64)
            int tcount = 0;  // running T-index (uniform)
            unsigned long long amask[RPT];  // bit a = my row's entry in the a-th non-pivot column (a < wspan)
#pragma unroll
            for (int k = 0; k < RPT; ++k) amask[k] = 0ull;
            if (P.osd_method == 3 && !P.cost) {
                // ---- singles: the weight of "switch column j on" is 1 + #{pivot rows r : A[r][j] != y[r]}.  All 64
                // columns of a word are counted at once with bit-sliced counters: a thread adds its two rows, four DPP
                // steps add up the 16 lanes of a row (2 -> 6 counter bits), every lane unpacks four columns and the
                // 4 rows x <= 8 waves meet in LDS (two columns per 32-bit atomic).  Pivot columns are counted too and
                // never read.
                unsigned short* wt16 = reinterpret_cast<unsigned short*>(L.wt);
                const int l16 = lane & 15;
                const unsigned int sh = (unsigned int)(l16 & 7) * 4u;
#pragma clang loop unroll(disable)
                for (int w = 0; w < W; ++w) {
                    unsigned long long npm = L.npmask[w];
                    if (!osd_uniform64(npm)) continue;  // uniform
                    unsigned long long v[RPT];
#pragma unroll
                    for (int k = 0; k < RPT; ++k) {
                        const unsigned long long rw = ws[(unsigned)(w * MR + tid + k * NT)];  // my own words
                        v[k] = used[k] ? (y[k] ? ~rw : rw) : 0ull;
                    }
                    unsigned int clo[6], chi[6];
                    clo[0] = (unsigned int)(v[0] ^ v[1]); chi[0] = (unsigned int)((v[0] ^ v[1]) >> 32);
                    clo[1] = (unsigned int)(v[0] & v[1]); chi[1] = (unsigned int)((v[0] & v[1]) >> 32);
                    osd_bs_add<0xB1, 2>(clo);  osd_bs_add<0xB1, 2>(chi);    // quad_perm [1,0,3,2]
                    osd_bs_add<0x4E, 3>(clo);  osd_bs_add<0x4E, 3>(chi);    // quad_perm [2,3,0,1]
                    osd_bs_add<0x124, 4>(clo); osd_bs_add<0x124, 4>(chi);   // row_ror:4
                    osd_bs_add<0x128, 5>(clo); osd_bs_add<0x128, 5>(chi);   // row_ror:8
                    unsigned int packed = 0u;  // byte c = count of column 4 * l16 + c over this row of 16 lanes
#pragma unroll
                    for (int i = 0; i < 6; ++i) {
                        const unsigned int half = (l16 & 8) ? chi[i] : clo[i];
                        const unsigned int nib = (half >> sh) & 15u;
                        packed += ((nib * 0x204081u) & 0x01010101u) << i;
                    }
                    unsigned int* dst = reinterpret_cast<unsigned int*>(wt16 + w * 64 + 4 * l16);
                    atomicAdd(dst, (packed & 0xffu) | ((packed & 0xff00u) << 8));
                    atomicAdd(dst + 1, ((packed >> 16) & 0xffu) | ((packed >> 24) << 16));
                }
            }
            // ---- the first wspan non-pivot columns ("T columns"), transposed: one ballot word per wave and row slot
#pragma clang loop unroll(disable)
            for (int w = 0; w < W && tcount < wspan; ++w) {
                unsigned long long npm = osd_uniform64(L.npmask[w]);
                if (!npm) continue;  // uniform
                unsigned long long rw[RPT];
#pragma unroll
                for (int k = 0; k < RPT; ++k) rw[k] = ws[(unsigned)(w * MR + tid + k * NT)];  // my own words
                while (npm && tcount < wspan) {
                    const int b = __ffsll((long long)npm) - 1;
                    npm &= npm - 1;
                    const unsigned long long bmask = 1ull << b;
#pragma unroll
                    for (int k = 0; k < RPT; ++k) {
                        const bool bitv = (rw[k] & bmask) != 0ull;
                        const unsigned long long cb = __ballot(used[k] && bitv);
                        if (lane == 0) L.colvec[tcount * OSD_MAXCV + k * nwaves + wave] = cb;
                        if (bitv) amask[k] |= 1ull << tcount;
                    }
                    if (tid == 0) tpos[tcount] = (unsigned short)(w * 64 + b);
                    ++tcount;
                }
            }
            __syncthreads();
#ifdef BPOSD_OSD_DIAG
            if (P.dbg && slot_id == 0) {  // diagnostics: dump yvec, tpos, colvec behind the 8 stamps
                for (int i = tid; i < OSD_MAXCV; i += NT) P.dbg[8 + i] = (long long)L.yvec[i];
                for (int i = tid; i < 64; i += NT) P.dbg[8 + OSD_MAXCV + i] = (long long)tpos[i];
                for (int i = tid; i < 64 * OSD_MAXCV; i += NT) P.dbg[8 + OSD_MAXCV + 64 + i] = (long long)L.colvec[i];
                for (int i = tid; i < W; i += NT) P.dbg[8 + OSD_MAXCV + 64 + 64 * OSD_MAXCV + i] = (long long)L.npmask[i];
            }
#endif
            if (P.cost) {
                // ================= fp64 log-weights (non-uniform channel), candidates one per lane
                // scratch tables live in the (now idle) four-Russians table region
                // ... reused here (colvec / prow / tab are dead once the sweep above is done)
                unsigned long long* am = L.prow;                              // [MR] amask per row
                double* costs = (double*)(am + MR);                           // [n]
                unsigned long long* Mi = (unsigned long long*)(costs + NS);   // [n] per ORIGINAL bit: its entries
                                                                              //     in the first wspan T-columns
                double* wd = (double*)(Mi + NS);                              // [64 * W] single-candidate weights
                double* pairw = wd + 64 * W;                                  // [<= 2016] pair weights
                short* info = (short*)(pairw + 2016);                         // [n] pivot row of bit i, -1 if non-pivot
                __syncthreads();  // every wave is done with colvec before it is overwritten
#pragma unroll
                for (int k = 0; k < RPT; ++k) am[tid + k * NT] = amask[k];
                for (int i = tid; i < n; i += NT) {
                    double ci = P.cost[i];
                    if (P.sel) {  // per-syndrome two-valued channel (css_decode_sim.py:207-248)
                        const unsigned int pick = P.sel[(size_t)s * n + i];
                        const double ca = P.cost_alt[i];
                        if (pick != 0u) ci = ca;
                    }
                    costs[i] = ci;
                    info[i] = L.pivrow[L.inv[i]];
                }
                __syncthreads();
#ifdef BPOSD_OSD_DIAG
                if (P.dbg && slot_id == 0) {  // diagnostics: the per-bit costs this syndrome is weighed with
                    for (int i = tid; i < 512 && i < n; i += NT) P.dbg[1200 + i] = __double_as_longlong(costs[i]);
                    if (tid == 0) { P.dbg[1199] = s; P.dbg[1198] = (long long)(P.sel != nullptr); }
                    for (int i = tid; i < 64; i += NT) { P.dbg[1800 + i] = P.sel ? (long long)P.sel[(size_t)s * n + i] : -1; P.dbg[1870 + i] = __double_as_longlong(P.cost[i]); P.dbg[1940 + i] = __double_as_longlong(P.cost_alt[i]); }
                }
#endif
                for (int i = tid; i < n; i += NT) {
                    const int pr = info[i];
                    unsigned long long mi = 0ull;
                    if (pr >= 0) {
                        mi = am[pr];
                    } else {
                        const int pos = L.inv[i];
                        for (int a = 0; a < wspan && a < tcount; ++a)
                            if (tpos[a] == pos) mi = 1ull << a;
                    }
                    Mi[i] = mi;
                }
                __syncthreads();
                // weight of OSD-0 (every lane redundantly: uniform broadcast reads)
                double w0d = 0.0;
                for (int i = 0; i < n; ++i)
                    if (L.xout[i]) w0d += costs[i];
                unsigned long long* bestw = L.best64;  // [0] singles / patterns, [1] pairs (weight bit patterns)
                int* besti = L.misc + 2;                // [0], [1] first index attaining them
                if (tid == 0) { besti[0] = 0x7fffffff; besti[1] = 0x7fffffff; }
                if (P.osd_method == 3) {
                    // ---- singles: lane c of the wave that owns word w evaluates column 64 w + c
                    for (int w = wave; w < W; w += nwaves) {
                        const int j = w * 64 + lane;
                        double acc = 0.0;
                        for (int i = 0; i < n; ++i) {
                            const int pr = info[i];
                            unsigned long long word;
                            if (pr >= 0) {
                                word = ws[(unsigned)(w * MR + pr)];
                            } else {
                                const int pos = L.inv[i];
                                word = ((pos >> 6) == w) ? (1ull << (pos & 63)) : 0ull;
                            }
                            const int xi = (int)L.xout[i] ^ (int)((word >> lane) & 1ull);
                            if (xi) acc += costs[i];
                        }
                        wd[j] = acc;
                        if (j < n && L.pivrow[j] < 0) atomicMin(&bestw[0], (unsigned long long)__double_as_longlong(acc));
                    }
                    // ---- pairs (a < b < wspan)
                    const int npairs = wspan * (wspan - 1) / 2;
                    for (int pidx = tid; pidx < npairs; pidx += NT) {
                        int a = 0, rem = pidx;
                        while (rem >= wspan - 1 - a) { rem -= wspan - 1 - a; ++a; }
                        const unsigned long long pat = (1ull << a) | (1ull << (a + 1 + rem));
                        double acc = 0.0;
                        for (int i = 0; i < n; ++i) {
                            const int xi = (int)L.xout[i] ^ (__popcll(Mi[i] & pat) & 1);
                            if (xi) acc += costs[i];
                        }
                        atomicMin(&bestw[1], (unsigned long long)__double_as_longlong(acc));
                        // remember my value for pass 2 in LDS: pair weights array after the singles'
                        pairw[pidx] = acc;
                    }
                    __syncthreads();
                    // pass 2: first index attaining each minimum
                    const unsigned long long ms = bestw[0], mp = bestw[1];
                    for (int j = tid; j < n; j += NT)
                        if (L.pivrow[j] < 0 && (unsigned long long)__double_as_longlong(wd[j]) == ms) atomicMin(&besti[0], j);
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
                        int pidx = besti[1];
                        int a = 0, rem = pidx;
                        while (rem >= wspan - 1 - a) { rem -= wspan - 1 - a; ++a; }
                        sel_a = tpos[a];
                        sel_b = tpos[a + 1 + rem];
                    }
                } else {
                    // ---- osd_e: patterns 1 .. 2^w - 1
                    const unsigned int npat = (1u << wspan) - 1u;
                    for (int pass = 0; pass < 2; ++pass) {
                        const unsigned long long target = bestw[0];
                        for (unsigned int pat = tid + 1; pat <= npat; pat += NT) {
                            double acc = 0.0;
                            for (int i = 0; i < n; ++i) {
                                const int xi = (int)L.xout[i] ^ (__popcll(Mi[i] & (unsigned long long)pat) & 1);
                                if (xi) acc += costs[i];
                            }
                            const unsigned long long bits = (unsigned long long)__double_as_longlong(acc);
                            if (pass == 0) atomicMin(&bestw[0], bits);
                            else if (bits == target) atomicMin(&besti[0], (int)osd_e_index(pat, wspan, P.e_msb_first));
                        }
                        __syncthreads();
                    }
                    const unsigned long long ms = bestw[0];
                    if (ms != ~0ull && __longlong_as_double((long long)ms) < w0d) {
                        sel_a = -2;
                        sel_b = (int)osd_e_index((unsigned int)besti[0], wspan, P.e_msb_first);  // index -> column mask
                    }
                }
            } else if (P.osd_method == 3) {
                // singles: all k' non-pivot positions, enumeration order == position order
                for (int j = tid; j < n; j += NT) {
                    if (L.pivrow[j] < 0) {
                        const unsigned long long key = ((unsigned long long)reinterpret_cast<const unsigned short*>(L.wt)[j] << 32) | (unsigned)j;
                        atomicMin(&L.best64[0], key);
                    }
                }
                // pairs (a < b < w), a outer, b inner
                const int npairs = wspan * (wspan - 1) / 2;
                for (int pidx = tid; pidx < npairs; pidx += NT) {
                    int a = 0, rem = pidx;
                    while (rem >= wspan - 1 - a) { rem -= wspan - 1 - a; ++a; }
                    const int bq = a + 1 + rem;
                    int wgt = 2;
                    for (int q = 0; q < ncv; ++q)
                        wgt += __popcll(L.yvec[q] ^ L.colvec[a * OSD_MAXCV + q] ^ L.colvec[bq * OSD_MAXCV + q]);
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
                    for (int q = 0; q < ncv; ++q) {
                        unsigned long long v = L.yvec[q];
                        unsigned int pp = pat;
                        while (pp) {
                            const int bq = __ffs((int)pp) - 1;
                            pp &= pp - 1;
                            v ^= L.colvec[bq * OSD_MAXCV + q];
                        }
                        wgt += __popcll(v);
                    }
                    const unsigned long long key = ((unsigned long long)wgt << 32) | osd_e_index(pat, wspan, P.e_msb_first);
                    atomicMin(&L.best64[0], key);
                }
                __syncthreads();
                const unsigned long long k1 = L.best64[0];
                if (k1 != ~0ull && (int)(k1 >> 32) < w0) {
                    sel_a = -2;  // winner is a multi-column pattern
                    sel_b = (int)osd_e_index((unsigned int)(k1 & 0xffffffffu), wspan, P.e_msb_first);
                }
            }
        }

        OSD_STAMP(5);
        // ------------------------------------------------- write the OSD-W solution
        if (sel_a == -1) {
            // OSD-0 stays the best
            osd_store_row(P.out_osdw, PACKED ? 1 : 0, (size_t)s, n, L.xout, tid, NT);
            osd_store_row(P.cmp_osdw, PACKED ? 1 : 0, (size_t)slot_id, n, L.xout, tid, NT);
        } else {
            __syncthreads();
            for (int i = tid; i < n; i += NT) L.xout[i] = 0;
            __syncthreads();
            bool xs[RPT];
#pragma unroll
            for (int k = 0; k < RPT; ++k) xs[k] = y[k];
            if (sel_a >= 0) {
                // one or two switched-on columns at sorted positions sel_a, sel_b
#pragma unroll
                for (int k = 0; k < RPT; ++k) {
                    xs[k] ^= ((ws[(unsigned)((sel_a >> 6) * MR + tid + k * NT)] >> (sel_a & 63)) & 1ull) != 0ull;
                    if (sel_b >= 0) xs[k] ^= ((ws[(unsigned)((sel_b >> 6) * MR + tid + k * NT)] >> (sel_b & 63)) & 1ull) != 0ull;
                }
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
#pragma unroll
                    for (int k = 0; k < RPT; ++k)
                        xs[k] ^= ((ws[(unsigned)((pos >> 6) * MR + tid + k * NT)] >> (pos & 63)) & 1ull) != 0ull;
                    if (tid == 0) L.xout[L.kidx[pos]] = 1;
                }
            }
#pragma unroll
            for (int k = 0; k < RPT; ++k)
                if (used[k] && xs[k]) L.xout[L.kidx[mypos[k]]] = 1;
            __syncthreads();
            osd_store_row(P.out_osdw, PACKED ? 1 : 0, (size_t)s, n, L.xout, tid, NT);
            osd_store_row(P.cmp_osdw, PACKED ? 1 : 0, (size_t)slot_id, n, L.xout, tid, NT);
        }
        OSD_STAMP(6);
        __syncthreads();
    }
}

}  // namespace bposd
