// bp_own_kernel.hip.h -- min-sum BP for codes with ONE check degree and bit degrees 3 / 4 in which every check OWNS two
// of its bits (n >= 2m and a perfect b-matching exists: the reference's three example codes,
// /root/reference/examples/qldpc_decode_example.py:5-23, and every hypergraph product of regular seeds of that shape).
// What bp_local_kernel.hip.h does for (3,6)-regular codes, without its wave-uniform position classes: the thread that runs
// a check also runs the two bits the check owns, so the messages on the two owned edges stay in that thread's registers --
// DC - 2 instead of DC LDS messages per check in the check pass, one LDS message less per owned bit in the bit pass.
//   * A check's edges may be visited in any order in the min-sum update (minima and sign parity do not depend on it), so
//     the owned edges are edges 0 and 1 of every check at compile time.  (Product-sum multiplies in edge order: it stays on
//     bp_class_kernel.hip.h.)
//   * The bit update is order-dependent: the register message enters the bit's fp64 sums at the rank dl the owner has among
//     the bit's checks, chosen by three lane masks kept in scalar registers (selects, no branches).  A degree-3 bit is
//     summed as a degree-4 bit whose last message is +0.0 (exact: see bit_update_deg in bp_class_kernel.hip.h -- a prefix
//     is never -0.0); a (wave, slot) group whose bits all have degree 3 skips the fourth message altogether.
//   * The bits no check owns (n - 2m) sit in degree groups as in the class kernel.  Positions without a check run a closed
//     toy graph with positive messages (their planes and private slots start at a positive prior): nothing flips there.
// Scheme around it (persistent workgroups on an atomic queue, incremental mismatch bitmap, speculative check pass, two
// barriers per iteration, cold arguments read at use) as in bp_class_kernel.hip.h.  fp64, -ffp-contract=off.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bp_class_kernel.hip.h"

namespace bposd {

struct BpOwnParams {
    int m, n;
    long long B;
    int max_iter;
    double ms_scaling;
    int osd_enabled;
    int zero_slot, priv0;                 // LDS map in doubles: (DC - 2) * MP planes | zero slot (2) | 3 * NTMAX private slots
    const uint8_t* __restrict__ synd;     // [B, m]
    const double* __restrict__ llr0;      // [n]
    const uint8_t* __restrict__ sel;      // [B, n] nullable
    const double* __restrict__ llr0_alt;  // [n]
    const int* __restrict__ pos_chk;      // [NTMAX]
    const int* __restrict__ own_bit;      // [2 * NTMAX]
    const int* __restrict__ own_rd;       // [6 * NTMAX]  entry (3 r + j) * NTMAX + tid
    const int* __restrict__ own_wr;       // [2 * NTMAX]
    const int* __restrict__ own_dl;       // [2 * NTMAX]
    const int* __restrict__ x_bit;        // [NTMAX]
    const int* __restrict__ x_slot;       // [4 * NTMAX]
    const int* __restrict__ x_deg;        // [NTMAX / 64]
    uint8_t* __restrict__ out_bp;
    uint8_t* __restrict__ out_osd0;
    uint8_t* __restrict__ out_osdw;
    uint8_t* __restrict__ out_conv;
    int* __restrict__ out_iters;
    double* __restrict__ out_llr;
    double* __restrict__ llr_ws;
    double* __restrict__ llr_tmp;         // [gridDim.x][n]
    int* __restrict__ osd_list;
    int* __restrict__ counters;
    unsigned long long* __restrict__ iter_total;
    int* __restrict__ tail_flag;
};

__host__ __device__ inline size_t bp_own_lds_bytes(int DC, int mp, int ntmax) {
    return ((size_t)(DC - 2) * mp + 2 + (size_t)3 * ntmax) * 8 + (size_t)(mp / 32 + 2) * 4 + 8 * 4;
}

typedef const __attribute__((address_space(4))) BpOwnParams* bpo_args_ptr;
__device__ __forceinline__ bpo_args_ptr bpo_args() {
    bpo_args_ptr a = (bpo_args_ptr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(a));
    return a;
}

__device__ __forceinline__ double bpo_sel(bool on, double a, double b) { return on ? a : b; }

// DC: degree of every check;  MPT: LDS stride (power of two >= m) = table stride NTMAX;  UPRIOR: uniform channel
template <int DC, int MPT, int MINW, bool UPRIOR>
__global__ __launch_bounds__(MPT, MINW) void bp_own_kernel(const BpOwnParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int MP = MPT, NTMAX = MPT, NW = NTMAX / 64, DL = DC - 2;  // DL: LDS messages per check
    const int m = P.m, n = P.n;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    typedef __attribute__((address_space(3))) unsigned char* lds_bytes;
    const unsigned int msg_base = (unsigned int)(uintptr_t)(lds_bytes)smem;
    double* msg_plain = reinterpret_cast<double*>(smem);
    unsigned int* diffw = reinterpret_cast<unsigned int*>(msg_plain + (size_t)DL * MP + 2 + (size_t)3 * NTMAX);
    int* sh = reinterpret_cast<int*>(diffw + (MP / 32 + 2));
    const unsigned int diffw_base = (unsigned int)(uintptr_t)(lds_bytes)diffw;
    const unsigned int real_end = msg_base + 8u * (unsigned int)(DL * MP);  // addresses below this are message planes
#define BPO_AT(a) ((msg_ptr)(uintptr_t)(a))
#define BPO_TAB(tab, row) bpc_table_load(bpo_args()->tab, (unsigned int)tid * 4u, (unsigned int)((row) * NTMAX * 4))
#define BPO_LLRT (bpo_args()->llr_tmp + (size_t)blockIdx.x * n)

    // ---- per-thread tables
    unsigned int ard[2][3], awr[2], xaddr[4];
    bool m0[2], m1[2], m2[2];  // per lane: the owner is the bit's first / second / third check (else: fourth)
    bool d3[2];                // wave-uniform: no bit of this (wave, slot) has a fourth message
#pragma unroll
    for (int r = 0; r < 2; ++r) {
#pragma unroll
        for (int j = 0; j < 3; ++j) ard[r][j] = msg_base + 8u * (unsigned int)bpo_args()->own_rd[(3 * r + j) * NTMAX + tid];
        awr[r] = msg_base + 8u * (unsigned int)bpo_args()->own_wr[r * NTMAX + tid];
        const int dl = bpo_args()->own_dl[r * NTMAX + tid];
        m0[r] = dl == 0;
        m1[r] = dl == 1;
        m2[r] = dl == 2;
        d3[r] = __ballot(ard[r][2] < real_end) == 0ull;
    }
#pragma unroll
    for (int d = 0; d < 4; ++d) xaddr[d] = msg_base + 8u * (unsigned int)bpo_args()->x_slot[d * NTMAX + tid];
    const int xdeg = __builtin_amdgcn_readfirstlane(bpo_args()->x_deg[wave]);
    const bool has_chk = ((wave << 6) < ((m + 63) & ~63));  // wave-uniform: this wave's positions hold checks
    double l0[UPRIOR ? 1 : 3];
    if (UPRIOR) {
        const double v = bpo_args()->llr0[0];
        l0[0] = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
    }
#define BPO_L0(r) l0[UPRIOR ? 0 : (r)]
    const int want_llr_s = __builtin_amdgcn_readfirstlane(bpo_args()->out_llr != nullptr ? 1 : 0);
    auto want_llr = [&]() -> bool {
        int w = __builtin_amdgcn_readfirstlane(want_llr_s);
        asm volatile("" : "+s"(w));
        return w != 0;
    };
    if (tid == 0) {
        msg_plain[P.zero_slot] = 0.0;  // read as the absent fourth message of degree-3 bits; never written again
        sh[0] = 0;
        sh[1] = 0;
    }
    unsigned long long it_acc = 0ull;

    for (;;) {
        if (tid == 0) sh[2] = atomicAdd(&bpo_args()->counters[0], 1);
        __syncthreads();
        const long long s = __builtin_amdgcn_readfirstlane(sh[2]);
        if (s >= P.B) {
            if (s == P.B && tid == 0 && bpo_args()->tail_flag) *(volatile int*)bpo_args()->tail_flag = 1;
            break;
        }
        // ---- syndrome bit of my check; the mismatch bitmap (indexed by position) starts as the syndrome
        const int cpos = BPO_TAB(pos_chk, 0);
        const bool sbit = (cpos >= 0) ? ((bpo_args()->synd[(size_t)s * m + cpos] & 1) != 0) : false;
        {
            const unsigned long long bal = __ballot(sbit);
            if (lane == 0 && (wave << 6) < MP) {
                diffw[(wave << 6) >> 5] = (unsigned int)bal;
                diffw[((wave << 6) >> 5) + 1] = (unsigned int)(bal >> 32);
                if (bal) sh[0] = 1;
            }
        }
        // ---- a3: every edge's bit -> check message starts at the prior of its bit
        double loc[2];
        unsigned int decmask = 0u;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int i = r < 2 ? BPO_TAB(own_bit, r) : BPO_TAB(x_bit, 0);
            double lp;
            if (UPRIOR) {
                lp = l0[0];
                asm volatile("" : "+s"(lp));
            } else {
                lp = 1.0;  // padding: any positive number
                if (i >= 0) lp = (bpo_args()->sel && bpo_args()->sel[(size_t)s * n + i]) ? bpo_args()->llr0_alt[i] : bpo_args()->llr0[i];
                l0[r] = lp;
            }
            if (want_llr() && i >= 0) BPO_LLRT[i] = lp;
            if (r < 2) {
                if (!has_chk) continue;
                loc[r] = lp;
                *BPO_AT(ard[r][0]) = lp;
                *BPO_AT(ard[r][1]) = lp;
                if (awr[r] < real_end || cpos < 0) *BPO_AT(awr[r]) = lp;  // a real third edge, or the padding position's toy slot
                if (r == 0 && cpos < 0) {  // padding position: its own planes feed its check with positive numbers
#pragma unroll
                    for (int k = 0; k < DL; ++k) *BPO_AT(msg_base + 8u * (unsigned int)(k * MP + tid)) = lp;
                }
            } else {
#pragma unroll
                for (int d = 0; d < 4; ++d)
                    if (d < 3 || d < xdeg) *BPO_AT(xaddr[d]) = lp;
            }
        }
        if (!has_chk) loc[0] = loc[1] = 1.0;
        __syncthreads();

        int it_done = 0;
        bool conv = (sh[0] == 0);
        if (!conv) {
#pragma clang loop unroll(disable)
            for (int it = 1;; ++it) {
                const int fi = it & 1;
                if (has_chk) {
                    const unsigned long long mis = *(const volatile __attribute__((address_space(3))) unsigned long long*)(uintptr_t)(
                        diffw_base + (unsigned int)((wave << 6) >> 3));
                    if (lane == 0 && mis) sh[fi] = 1;
                }
                if (it > P.max_iter) {
                    __syncthreads();
                    conv = (sh[fi] == 0);
                    it_done = P.max_iter;
                    break;
                }
                // =================== check -> bit pass (a5), speculative for it >= 2 ===========
                if (has_chk) {
                    const unsigned long long alpha_u = alpha_bits_for_iteration(P.ms_scaling, it);
                    const int alpha_lo = (int)(unsigned int)alpha_u, alpha_hi = (int)(unsigned int)(alpha_u >> 32), nalpha_hi = alpha_hi ^ (int)0x80000000;
                    msg_ptr mc = BPO_AT(msg_base + 8u * (unsigned int)tid);
                    double v[DC];
                    v[0] = loc[0];
                    v[1] = loc[1];
#pragma unroll
                    for (int k = 0; k < DL; ++k) v[k + 2] = mc[k * MP];
                    double pre[DC], suf[DC];
                    pre[0] = __DBL_MAX__;
#pragma unroll
                    for (int k = 1; k < DC; ++k) pre[k] = min_abs(pre[k - 1], v[k - 1]);
                    suf[DC - 1] = __DBL_MAX__;
#pragma unroll
                    for (int k = DC - 2; k >= 0; --k) suf[k] = min_abs(suf[k + 1], v[k + 1]);
                    bool neg[DC];
                    bool par = sbit;
#pragma unroll
                    for (int k = 0; k < DC; ++k) {
                        neg[k] = (v[k] <= 0.0);  // a zero counts as negative, as in the reference
                        par ^= neg[k];
                    }
#pragma unroll
                    for (int k = 0; k < DC; ++k) {
                        const double mag = (k == 0) ? suf[0] : (k == DC - 1 ? pre[DC - 1] : min_pos(pre[k], suf[k]));
                        const double sa = __hiloint2double((par ^ neg[k]) ? nalpha_hi : alpha_hi, alpha_lo);
                        const double o = mag * sa;
                        if (k < 2) loc[k] = o;
                        else mc[(k - 2) * MP] = o;
                    }
                }
                __syncthreads();
                if (sh[fi] == 0) {
                    conv = true;
                    it_done = it - 1;
                    break;
                }
                if (tid == 0) sh[fi ^ 1] = 0;
                const bool keep_llr = (it == P.max_iter) || want_llr();  // uniform
                // ============ bit pass: posterior, decision, bit -> check (a6 / a7) ============
                if (has_chk) {
#pragma unroll
                    for (int r = 0; r < 2; ++r) {
                        const double L0 = *BPO_AT(ard[r][0]);
                        const double L1 = *BPO_AT(ard[r][1]);
                        double L2 = 0.0;
                        if (!d3[r]) L2 = *BPO_AT(ard[r][2]);  // wave-uniform
                        const double R = loc[r];
                        const bool m01 = m0[r] || m1[r], m012 = m01 || m2[r];
                        double c[4], out[4], t;
                        c[0] = bpo_sel(m0[r], R, L0);
                        c[1] = bpo_sel(m1[r], R, bpo_sel(m0[r], L0, L1));
                        c[2] = bpo_sel(m2[r], R, bpo_sel(m01, L1, L2));
                        c[3] = bpo_sel(m012, L2, R);
                        bit_update_deg<4>(BPO_L0(r), c, t, out);
                        loc[r] = bpo_sel(m0[r], out[0], bpo_sel(m1[r], out[1], bpo_sel(m2[r], out[2], out[3])));
                        *BPO_AT(ard[r][0]) = bpo_sel(m0[r], out[1], out[0]);
                        *BPO_AT(ard[r][1]) = bpo_sel(m01, out[2], out[1]);
                        if (!d3[r]) *BPO_AT(awr[r]) = bpo_sel(m012, out[3], out[2]);
                        if (keep_llr) {
                            const int bi = BPO_TAB(own_bit, r);
                            if (bi >= 0) BPO_LLRT[bi] = t;
                        }
                        const unsigned int dnew = (t <= 0.0) ? 1u : 0u;
                        if (dnew != ((decmask >> r) & 1u)) {  // (a padding position never gets here: its messages stay positive)
                            decmask ^= 1u << r;
                            atomicXor(&diffw[tid >> 5], 1u << (tid & 31));  // the owner's own check
#pragma unroll
                            for (int j = 0; j < 3; ++j) {
                                unsigned int pa = j < 2 ? ard[r][j] : awr[r];
                                asm volatile("" : "+v"(pa));
                                if (pa < real_end) {  // (the third edge of a degree-3 bit is a private slot beyond the planes)
                                    const int c2 = (int)((pa - msg_base) >> 3) & (MP - 1);
                                    atomicXor(&diffw[c2 >> 5], 1u << (c2 & 31));
                                }
                            }
                        }
                    }
                }
                if (xdeg != 0) {
                    bit_pass_arm<3, 4, MP>(xdeg, xaddr, BPO_L0(2), 2, decmask, msg_base, diffw, keep_llr, [&](double t) {
                        const int bi = BPO_TAB(x_bit, 0);
                        if (bi >= 0) BPO_LLRT[bi] = t;
                    });
                }
                __syncthreads();
            }
        }

        // ---- results
        const bool to_osd = (!conv) && bpo_args()->osd_enabled;
        if (tid == 0) {
            if (to_osd) {
                const int slot = atomicAdd(&bpo_args()->counters[1], 1);
                bpo_args()->osd_list[slot] = (int)s;
                sh[3] = slot;
            }
            if (bpo_args()->out_conv) bpo_args()->out_conv[s] = conv ? 1 : 0;
            if (bpo_args()->out_iters) bpo_args()->out_iters[s] = it_done;
        }
        it_acc += (unsigned long long)it_done;
        __syncthreads();
        const int slot = to_osd ? sh[3] : 0;
        if (tid == 0) {
            // the two convergence flags start the next syndrome at 0.  Only HERE: every wave has read its `conv` from them
            // before the barrier above (zeroing them in the block before it let a late wave read 0 -- "converged" -- for a
            // syndrome the others sent to OSD: its 64 bits' LLRs never reached the OSD workspace; found as a run-to-run
            // wobble of the logical error rate, 3 shots in 131072), and the next syndrome sets them after the barrier below.
            int zero = 0;
            asm volatile("" : "+v"(zero));
            sh[0] = zero;
            sh[1] = zero;
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int i = r < 2 ? BPO_TAB(own_bit, r) : BPO_TAB(x_bit, 0);
            if (i >= 0) {
                const size_t o = (size_t)s * n + i;
                const uint8_t b = (uint8_t)((decmask >> r) & 1u);
                if (bpo_args()->out_bp) bpo_args()->out_bp[o] = b;
                if (!to_osd) {
                    bpo_args()->out_osdw[o] = b;
                    if (bpo_args()->out_osd0) bpo_args()->out_osd0[o] = b;
                } else {
                    bpo_args()->llr_ws[(size_t)slot * n + i] = BPO_LLRT[i];
                }
                if (want_llr()) bpo_args()->out_llr[o] = BPO_LLRT[i];
            }
        }
        __syncthreads();
    }
    if (tid == 0 && it_acc) atomicAdd(bpo_args()->iter_total, it_acc);
#undef BPO_AT
#undef BPO_TAB
#undef BPO_LLRT
#undef BPO_L0
}

}  // namespace bposd
