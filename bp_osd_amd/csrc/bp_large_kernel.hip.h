// bp_large_kernel.hip.h -- belief propagation for codes whose messages do not fit one CU's LDS
// (BASELINE configs[4]: 14520 x 29524, E = 159720 -> 1.28 MB of fp64 messages per syndrome).
//
// Same algorithm, arithmetic order and convergence bookkeeping as bp_kernel.hip.h (rows a3-a7 of
// SURVEY.md §8), re-mapped for the HBM-bound regime:
//   * one 512-thread workgroup per syndrome (persistent, atomic queue); a thread walks checks
//     c = tid, tid + 512, ... in the check pass and bits i = tid, tid + 512, ... in the bit pass;
//   * the E messages live in a per-workgroup slice of a global workspace in the same check-major
//     structure-of-arrays layout  msg[k * MP + c]  -> the check pass streams fully coalesced
//     (consecutive lanes = consecutive checks), the bit pass gathers / scatters 8-byte words whose
//     cache lines are shared between neighbouring bits (hypergraph-product locality) and between
//     the d = 0..DV-1 sweeps, so L1/L2 absorb the re-use and HBM sees ~ (4E + 2n) * 8 bytes per
//     iteration -- the algorithmic figure SURVEY §8(d) prices this regime with;
//   * graph tables (degrees, edge slots) are re-read from global memory every iteration (+E * 4 B);
//   * hard decisions (bytes) and the mismatch bitmap stay in LDS; the final LLRs are written to the
//     workspace only in the last iteration (or every iteration when the caller wants LLRs).
//   * min-sum (METHOD 1; round 4): a check's deg check->bit messages take two magnitudes only -- the smallest |bit->check|
//     for every edge but the one that attains it, the second smallest for that one -- so the check pass writes ONE 32-byte
//     record per check {alpha * min1, alpha * min2, index of the minimum, sign flips} instead of deg 8-byte messages, and
//     the bit pass rebuilds each incoming message from its check's record (flip_sign(k == kmin ? a2 : a1, flip_k): the same
//     fp64 product the per-edge form computes -- min over a set is exact and order-free, the product is formed once).
//     HBM per iteration: (2E + 2n) * 8 + 2 * 32 m instead of (4E + 2n) * 8 -- 3.95 instead of 5.58 MB on 14520 x 29524.
//     METHOD 2 (where it fits the CU's LDS: 14520 x 29524 needs 151 KB) keeps a1 (8 bytes) and the flags (2 bytes up to check
//     degree 12) of every check in LDS, the hard decisions as bits (a wave's 64 decisions are one ballot), and writes only a2
//     to the workspace; the bit pass reads a1 and the flags from LDS and fetches a2 for the one edge in deg that holds the
//     minimum -- its only gather.
//     Product-sum (METHOD 0) keeps per-edge messages both ways -- the form min-sum had in rounds 1-3 too.
// Visibility: messages written by one wave and read by another of the SAME workgroup go through
// global memory between two __syncthreads() (workgroup-scope release/acquire; the waves share the
// CU's L1).  No inter-workgroup communication exists.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "portable_math.h"  // bit-reproducible tanh / log of the product-sum update

#include "bp_kernel.hip.h"

namespace bposd {

struct BpLargeParams {
    int m, n;
    long long B;
    int max_iter;
    double ms_scaling;
    double ps_clip;  // product-sum: 0 = none, C > 0 = check->bit messages clamped to [-C, C]
    int ps_form;     // product-sum: 0 the reference's operation order (four divisions per edge), 1 two divisions (portable_math.h)
    int packed_io;   // 1: synd is [B][ceil(m/64)] and out_bp / out_osd0 / out_osdw are rows of ceil(n/64) little-endian 64-bit words
    int osd_enabled;
    int mp;                              // check stride of the message layout (m rounded up to 64)
    const uint8_t* __restrict__ synd;    // [B, m]
    const double* __restrict__ llr0;     // [n]
    const uint8_t* __restrict__ sel;     // [B, n] nullable
    const double* __restrict__ llr0_alt; // [n]
    const int* __restrict__ chk_deg;     // [m]
    const int* __restrict__ var_deg;     // [n]
    const int* __restrict__ var_pos;     // [DV * n], entry d*n+i = k*mp + c
    const int* __restrict__ var_ck;      // [DV * n], entry d*n+i = c*16 + k   (min-sum)
    double* __restrict__ msg_ws;         // [gridDim.x][DC * mp]
    double* __restrict__ llr_tmp;        // [gridDim.x][n]  LLRs of the current syndrome
    uint8_t* __restrict__ out_bp;
    uint8_t* __restrict__ out_osd0;
    uint8_t* __restrict__ out_osdw;
    uint8_t* __restrict__ out_conv;
    int* __restrict__ out_iters;
    double* __restrict__ out_llr;
    double* __restrict__ llr_ws;         // [cap, n]
    int* __restrict__ osd_list;
    int* __restrict__ counters;
    unsigned long long* __restrict__ iter_total;
    int* __restrict__ tail_flag;  // nullable, host-visible: set to 1 by the workgroup that finds the queue empty (the tail begins)
};

// per_check_in_lds (METHOD 2): a1 of every check (8 bytes) and its flags (2 bytes up to check degree 12, else 4) in LDS, the hard
// decisions as one BIT per bit instead of one byte
__host__ __device__ inline size_t bp_large_lds_bytes(int m, int n, bool per_check_in_lds = false, int dc = 12) {
    const size_t tail = (size_t)((m + 31) / 32 + 2) * 4 + 8 * 4;
    if (!per_check_in_lds) return (size_t)((n + 15) & ~15) + tail;
    const size_t meta = ((size_t)m * (dc <= 12 ? 2 : 4) + 7) & ~(size_t)7;
    return (size_t)((m + 1) & ~1) * 8 + meta + (size_t)((n + 63) / 64) * 8 + tail;
}

constexpr int bp_large_threads(int method) { return method == 2 ? 1024 : 512; }  // METHOD 2: 80 VGPRs, one workgroup per CU (LDS): 16 waves

template <int DC, int DV, int METHOD>
__global__ __launch_bounds__(bp_large_threads(METHOD)) void bp_large_kernel(const BpLargeParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int m = P.m, n = P.n, MP = P.mp;
    const int NT = blockDim.x;
    const int tid = threadIdx.x;
    const int lane = tid & 63;

    constexpr bool REC = (METHOD == 1);                  // min-sum: one 32-byte record per check in the workspace
    constexpr bool RLDS = (METHOD == 2);                 // min-sum: a1 of every check in LDS, a2 and the flags in the workspace
    constexpr int SLOTS = REC ? DC + 4 : (RLDS ? DC + 1 : DC);  // message planes + the per-check data
    using meta_t = typename std::conditional<(DC <= 12), unsigned short, unsigned int>::type;  // kmin | flips << 4
    double* a1lds = reinterpret_cast<double*>(smem);     // [m] (RLDS)
    meta_t* metal = reinterpret_cast<meta_t*>(smem + (size_t)((m + 1) & ~1) * 8);       // [m] (RLDS)
    unsigned long long* decw = reinterpret_cast<unsigned long long*>(smem + (size_t)((m + 1) & ~1) * 8 + (((size_t)m * sizeof(meta_t) + 7) & ~(size_t)7));  // [n / 64] (RLDS)
    unsigned char* dec = smem;                                                        // [n] hard decisions (not RLDS)
    unsigned int* diffw = RLDS ? reinterpret_cast<unsigned int*>(decw + (n + 63) / 64)
                               : reinterpret_cast<unsigned int*>(dec + ((n + 15) & ~15));  // [m/32] mismatch bitmap
    int* sh = reinterpret_cast<int*>(diffw + ((m + 31) / 32 + 2));                    // flags / ids
    double* msg = P.msg_ws + (size_t)blockIdx.x * SLOTS * MP;
    double* rec = msg + (size_t)DC * MP;                 // REC: [MP][4]: a1, a2, (kmin | flips << 8), unused   (32-byte aligned: MP % 64 == 0)
    double* a2g = rec;                                   // RLDS: [MP] a2
    double* llrt = P.llr_tmp + (size_t)blockIdx.x * n;

#ifdef BPOSD_BPLARGE_DIAG  // phase clocks of the first workgroups, printed at exit (tools/bp_large_probe.py with a -D build)
    unsigned long long dg_chk = 0, dg_bit = 0, dg_all = __builtin_readcyclecounter(), dg_t = 0;
    int dg_its = 0;
#endif
    for (;;) {
        if (tid == 0) {
            sh[0] = 0;
            sh[1] = 0;
            sh[2] = atomicAdd(&P.counters[0], 1);
        }
        __syncthreads();
        const long long s = sh[2];
        if (s >= P.B) {
            // the chunk loop of the host-pointer API launches the next chunk's kernels when this one's tail begins
            if (s == P.B && tid == 0 && P.tail_flag) *(volatile int*)P.tail_flag = 1;
            break;
        }
        const uint8_t* syn = P.synd + (size_t)s * m;
        const unsigned long long* synw = (const unsigned long long*)P.synd + (size_t)s * (size_t)((m + 63) >> 6);  // (packed_io)

        // mismatch bitmap = syndrome; messages = priors; decisions = 0
        for (int c0 = tid - lane; c0 < m; c0 += NT) {
            const int c = c0 + lane;
            const bool sb = (c < m) && (P.packed_io ? ((synw[c >> 6] >> (c & 63)) & 1ull) != 0ull : (syn[c] & 1) != 0);
            const unsigned long long bal = __ballot(sb);
            if (lane == 0) {
                diffw[c0 >> 5] = (unsigned int)bal;
                diffw[(c0 >> 5) + 1] = (unsigned int)(bal >> 32);
                if (bal) sh[0] = 1;
            }
        }
        for (int i = tid; i < n; i += NT) {
            double l0 = P.llr0[i];
            if (P.sel && P.sel[(size_t)s * n + i]) l0 = P.llr0_alt[i];
            const int deg = P.var_deg[i];
            if (REC || RLDS) {
                for (int d = 0; d < deg; ++d) {
                    const int ck = P.var_ck[(size_t)d * n + i];
                    msg[(size_t)(ck & 15) * MP + (ck >> 4)] = l0;
                }
            } else {
                for (int d = 0; d < deg; ++d) msg[P.var_pos[(size_t)d * n + i]] = l0;
            }
            if (!RLDS) dec[i] = 0;
            llrt[i] = l0;
        }
        if (RLDS)
            for (int w = tid; w < (n + 63) / 64; w += NT) decw[w] = 0ull;
        __syncthreads();

        int it_done = 0;
        bool conv = (sh[0] == 0);
        if (!conv) {
#pragma clang loop unroll(disable)
            for (int it = 1;; ++it) {
                const int fi = it & 1;
                {
                    bool mis = false;
                    for (int c = tid; c < m; c += NT) mis |= ((diffw[c >> 5] >> (c & 31)) & 1u) != 0;
                    const unsigned long long anym = __ballot(mis);
                    if (lane == 0 && anym) sh[fi] = 1;
                }
                if (it > P.max_iter) {
                    __syncthreads();
                    conv = (sh[fi] == 0);
                    it_done = P.max_iter;
                    break;
                }
                // ---------------- check pass
#ifdef BPOSD_BPLARGE_DIAG
                dg_t = __builtin_readcyclecounter();
#endif
                const double alpha = alpha_for_iteration(P.ms_scaling, it);
#pragma clang loop unroll(disable)
                for (int c = tid; c < m; c += NT) {
                    const int deg = P.chk_deg[c];
                    // (packed_io: the row is ceil(m / 64) words -- the byte address does not exist there.  Round 5's first build of the packed
                    // path read it all the same: wrong parities for every row but the first few, and reads up to B * m bytes past a buffer of
                    // B * m / 8; no test ran THIS kernel on packed rows until the three-lane default moved the allocations and the read faulted.)
                    const bool sbit = P.packed_io ? ((synw[c >> 6] >> (c & 63)) & 1ull) != 0ull : (syn[c] & 1) != 0;
                    double* mc = msg + c;
                    // absent edges (k >= deg) enter as +DBL_MAX (neutral for the minima and the sign parity; tanh = 1): every
                    // array element is defined on every path (partially defined arrays turn into loop-carried registers)
                    double v[DC];
#pragma unroll
                    for (int k = 0; k < DC; ++k) {
                        v[k] = __DBL_MAX__;
                        if (k < deg) v[k] = mc[(size_t)k * MP];
                    }
                    if (REC || RLDS) {
                        // two smallest magnitudes, where the smallest sits, the signs: one record per check
                        unsigned int negm = 0u;
                        bool par = sbit;
                        double m1 = __DBL_MAX__, m2 = __DBL_MAX__;
                        int kmin = 0;
#pragma unroll
                        for (int k = 0; k < DC; ++k) {
                            const bool ng = (v[k] <= 0.0);  // (absent edges: +DBL_MAX, never negative, never below a present value)
                            negm |= ng ? (1u << k) : 0u;
                            par ^= ng;
                            const double a = min_abs(__DBL_MAX__, v[k]);  // |v[k]|
                            const bool lt = a < m1;
                            m2 = lt ? m1 : min_pos(m2, a);
                            kmin = lt ? k : kmin;
                            m1 = lt ? a : m1;
                        }
                        const unsigned int flips = (par ? ~negm : negm) & 0xffffu;
                        if (RLDS) {
                            a1lds[c] = m1 * alpha;
                            a2g[c] = m2 * alpha;
                            metal[c] = (meta_t)((unsigned int)kmin | (flips << 4));
                        } else {
                            double2 r01;
                            r01.x = m1 * alpha;
                            r01.y = m2 * alpha;
                            double2* rp = reinterpret_cast<double2*>(rec + (size_t)c * 4);
                            rp[0] = r01;
                            reinterpret_cast<unsigned int*>(rp + 1)[0] = (unsigned int)kmin | (flips << 8);
                        }
                    } else {
                        double pre[DC], th[DC];
                        double t = 1.0;
#pragma unroll
                        for (int k = 0; k < DC; ++k) {
                            pre[k] = t;
                            th[k] = 1.0;
                            if (k < deg) {
                                th[k] = pm_ps_tanh_half(v[k], P.ps_form);
                                t *= th[k];
                            }
                        }
                        t = 1.0;
                        const double sg = sbit ? -1.0 : 1.0;
#pragma unroll
                        for (int k = DC - 1; k >= 0; --k) {
                            if (k < deg) {
                                const double x = pre[k] * t;
                                double o = sg * pm_ps_log_ratio(x, P.ps_form);
                                if (P.ps_clip > 0.0) {
                                    if (o > P.ps_clip) o = P.ps_clip;
                                    if (o < -P.ps_clip) o = -P.ps_clip;
                                }
                                mc[(size_t)k * MP] = o;
                                t *= th[k];
                            }
                        }
                    }
                }
                __syncthreads();
#ifdef BPOSD_BPLARGE_DIAG
                dg_chk += __builtin_readcyclecounter() - dg_t;
                dg_t = __builtin_readcyclecounter();
                ++dg_its;
#endif
                if (sh[fi] == 0) {
                    conv = true;
                    it_done = it - 1;
                    break;
                }
                if (tid == 0) sh[fi ^ 1] = 0;
                // ---------------- bit pass
                const bool keep_llr = (it == P.max_iter) || (P.out_llr != nullptr);
#pragma clang loop unroll(disable)
                for (int i = tid; i < n; i += NT) {
                    const int deg = P.var_deg[i];
                    double l0 = P.llr0[i];
                    if (P.sel && P.sel[(size_t)s * n + i]) l0 = P.llr0_alt[i];
                    int pos[DV];
                    double cm[DV], pre[DV];
                    if (RLDS) {
                        int ck[DV];
#pragma unroll
                        for (int d = 0; d < DV; ++d) {
                            ck[d] = 0;
                            if (d < deg) ck[d] = P.var_ck[(size_t)d * n + i];
                        }
#pragma unroll
                        for (int d = 0; d < DV; ++d) {
                            const int k = ck[d] & 15, c = ck[d] >> 4;
                            pos[d] = k * MP + c;
                            double mag = 0.0;
                            unsigned int meta = 0u;
                            if (d < deg) {
                                meta = (unsigned int)metal[c];
                                mag = a1lds[c];
                                if ((int)(meta & 15u) == k) mag = a2g[c];  // the edge that holds the check's minimum: 1 in deg
                            }
                            cm[d] = (d < deg) ? flip_sign(mag, ((meta >> (4 + k)) & 1u) != 0u) : 0.0;
                        }
                    } else if (REC) {
                        int ck[DV];
#pragma unroll
                        for (int d = 0; d < DV; ++d) {
                            ck[d] = 0;
                            if (d < deg) ck[d] = P.var_ck[(size_t)d * n + i];
                        }
                        double2 r01[DV];
                        unsigned int meta[DV];
#pragma unroll
                        for (int d = 0; d < DV; ++d) {
                            r01[d].x = 0.0; r01[d].y = 0.0; meta[d] = 0u;
                            if (d < deg) {
                                const double2* rp = reinterpret_cast<const double2*>(rec + (size_t)(ck[d] >> 4) * 4);
                                r01[d] = rp[0];
                                meta[d] = reinterpret_cast<const unsigned int*>(rp + 1)[0];
                            }
                        }
#pragma unroll
                        for (int d = 0; d < DV; ++d) {
                            const int k = ck[d] & 15;
                            pos[d] = k * MP + (ck[d] >> 4);
                            const double mag = ((int)(meta[d] & 15u) == k) ? r01[d].y : r01[d].x;
                            cm[d] = (d < deg) ? flip_sign(mag, ((meta[d] >> (8 + k)) & 1u) != 0u) : 0.0;
                        }
                    } else {
#pragma unroll
                        for (int d = 0; d < DV; ++d) {
                            pos[d] = 0;
                            cm[d] = 0.0;
                            if (d < deg) pos[d] = P.var_pos[(size_t)d * n + i];
                        }
#pragma unroll
                        for (int d = 0; d < DV; ++d)
                            if (d < deg) cm[d] = msg[pos[d]];
                    }
                    double t = l0;
#pragma unroll
                    for (int d = 0; d < DV; ++d) {
                        pre[d] = t;
                        if (d < deg) t += cm[d];
                    }
                    if (keep_llr) llrt[i] = t;
                    const int dnew = (t <= 0.0) ? 1 : 0;
                    double suf = 0.0;
#pragma unroll
                    for (int d = DV - 1; d >= 0; --d)
                        if (d < deg) {
                            msg[pos[d]] = pre[d] + suf;
                            suf += cm[d];
                        }
                    int dold;
                    if (RLDS) {
                        // one bit per decision: the wave's 64 bits are one word (i - lane is a multiple of 64), read by every lane
                        // before lane 0 replaces it with the ballot of the new decisions
                        dold = (int)((decw[i >> 6] >> (i & 63)) & 1ull);
                        const unsigned long long bal = __ballot(dnew != 0);
                        if (lane == 0) decw[i >> 6] = bal;
                    } else {
                        dold = (int)dec[i];
                        if (dnew != dold) dec[i] = (unsigned char)dnew;
                    }
                    if (dnew != dold) {
#pragma unroll
                        for (int d = 0; d < DV; ++d)
                            if (d < deg) {
                                const int c = pos[d] % MP;
                                atomicXor(&diffw[c >> 5], 1u << (c & 31));
                            }
                    }
                }
                __syncthreads();
#ifdef BPOSD_BPLARGE_DIAG
                dg_bit += __builtin_readcyclecounter() - dg_t;
#endif
            }
        }

        const bool to_osd = (!conv) && P.osd_enabled;
        if (tid == 0) {
            if (to_osd) {
                const int slot = atomicAdd(&P.counters[1], 1);
                P.osd_list[slot] = (int)s;
                sh[3] = slot;
            }
            if (P.out_conv) P.out_conv[s] = conv ? 1 : 0;
            if (P.out_iters) P.out_iters[s] = it_done;
            if (it_done) atomicAdd(P.iter_total, (unsigned long long)it_done);
        }
        __syncthreads();
        const int slot = to_osd ? sh[3] : 0;
        if (P.packed_io) {
            // bit-packed result rows: bit (i & 63) of word (i >> 6) = entry i (the hard decisions are 64-bit words already in the
            // LDS form; bits beyond n are cleared)
            const int wpn = (n + 63) >> 6;
            for (int w = tid; w < wpn; w += NT) {
                unsigned long long v = 0ull;
                if (RLDS) {
                    v = decw[w];
                    if (64 * w + 64 > n) v &= (1ull << (n - 64 * w)) - 1ull;
                } else {
                    for (int b = 0; b < 64 && 64 * w + b < n; ++b) v |= (unsigned long long)(dec[64 * w + b] & 1) << b;
                }
                const size_t o = (size_t)s * wpn + w;
                if (P.out_bp) ((unsigned long long*)P.out_bp)[o] = v;
                if (!to_osd) {
                    ((unsigned long long*)P.out_osdw)[o] = v;
                    if (P.out_osd0) ((unsigned long long*)P.out_osd0)[o] = v;
                }
            }
            if (to_osd)
                for (int i = tid; i < n; i += NT) P.llr_ws[(size_t)slot * n + i] = llrt[i];
        } else
        for (int i = tid; i < n; i += NT) {
            const size_t o = (size_t)s * n + i;
            const uint8_t b = RLDS ? (uint8_t)((decw[i >> 6] >> (i & 63)) & 1ull) : dec[i];
            if (P.out_bp) P.out_bp[o] = b;
            if (!to_osd) {
                P.out_osdw[o] = b;
                if (P.out_osd0) P.out_osd0[o] = b;
            } else {
                P.llr_ws[(size_t)slot * n + i] = llrt[i];
            }
            if (P.out_llr) P.out_llr[o] = llrt[i];
        }
        __syncthreads();
    }
#ifdef BPOSD_BPLARGE_DIAG
    if (tid == 0 && (blockIdx.x & 63) == 0)
        printf("bp_large wg %d: %d check passes, check pass %.1f Mclk, bit pass %.1f Mclk, all %.1f Mclk\n", (int)blockIdx.x, dg_its, dg_chk * 1e-6,
               dg_bit * 1e-6, (__builtin_readcyclecounter() - dg_all) * 1e-6);
#endif
}

}  // namespace bposd
