// bp_anydeg_kernel.hip.h -- flooding-schedule BP for parity-check matrices of ANY row / column weight.
//
// The tuned kernels are compiled for check degrees <= 16 and bit degrees <= 8 (what sparse quantum codes have); the
// reference's decoder (ldpc's, /root/reference/src/bposd/__init__.py:1) has no such limit, and classical LDPC / BCH-like
// matrices handed to `bposd_decoder` exceed it.  This kernel serves those: same algorithm, arithmetic order and
// convergence bookkeeping as bp_kernel.hip.h / bp_large_kernel.hip.h (rows a3-a7 of SURVEY.md §8), degrees are run-time
// loop bounds.  One 256-thread workgroup per syndrome (persistent, atomic queue); messages live in a per-workgroup slice
// of a global workspace indexed by CSR edge id; a thread walks checks c = tid, tid + 256, ... and bits likewise.
//   check pass, two sweeps over the check's edges: forward stores the prefix (running minimum of |b2c|, or running
//     product of tanh(b2c / 2)) of every edge in a second array and accumulates the sign parity; backward combines it with
//     the running suffix and overwrites the edge's message in place.  The candidate syndrome of the previous bit pass'
//     decisions is accumulated in the same forward sweep (the flooding kernels keep an incremental bitmap instead).
//   bit pass, two sweeps over the bit's edges in ascending check order: forward stores the prefix sums (prior included),
//     backward adds the suffix sums -- prefix(d) + suffix(d) as the reference forms them.
// Built for completeness and exactness (tests/test_gpu_parity.py compares LLR bits with the oracle), not tuned.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bp_kernel.hip.h"
#include "portable_math.h"

namespace bposd {

constexpr int BPA_NT = 256;

struct BpAnyParams {
    int m, n, E;
    long long B;
    int max_iter;
    int bp_method;  // 0 product-sum, 1 min-sum
    double ms_scaling;
    double ps_clip;
    int ps_form;  // product-sum evaluation order (portable_math.h: pm_ps_tanh_half)
    int osd_enabled;
    const uint8_t* __restrict__ synd;     // [B, m]
    const double* __restrict__ llr0;      // [n]
    const uint8_t* __restrict__ sel;      // [B, n] nullable
    const double* __restrict__ llr0_alt;  // [n]
    const int* __restrict__ rp;           // CSR indptr [m + 1]
    const int* __restrict__ ci;           // CSR indices [E]
    const int* __restrict__ cp;           // CSC indptr [n + 1]
    const int* __restrict__ ce;           // [E] CSR edge ids of a column, ascending row
    double* __restrict__ msg_ws;          // [gridDim.x][3 * E]: messages | prefixes | tanh values (product-sum)
    double* __restrict__ llr_tmp;         // [gridDim.x][n]
    uint8_t* __restrict__ out_bp;
    uint8_t* __restrict__ out_osd0;
    uint8_t* __restrict__ out_osdw;
    uint8_t* __restrict__ out_conv;
    int* __restrict__ out_iters;
    double* __restrict__ out_llr;
    double* __restrict__ llr_ws;
    int* __restrict__ osd_list;
    int* __restrict__ counters;
    unsigned long long* __restrict__ iter_total;
    int* __restrict__ tail_flag;  // nullable, host-visible: set to 1 by the workgroup that finds the queue empty
};

__host__ __device__ inline size_t bp_anydeg_lds_bytes(int n) { return (size_t)((n + 15) & ~15) + 8 * 4; }

__global__ __launch_bounds__(BPA_NT) void bp_anydeg_kernel(const BpAnyParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int m = P.m, n = P.n;
    const int tid = threadIdx.x;
    unsigned char* dec = smem;
    int* sh = reinterpret_cast<int*>(smem + ((n + 15) & ~15));
    double* msg = P.msg_ws + (size_t)blockIdx.x * 3 * P.E;
    double* pre = msg + P.E;
    double* thv = pre + P.E;
    double* llrt = P.llr_tmp + (size_t)blockIdx.x * n;

    for (;;) {
        if (tid == 0) sh[2] = atomicAdd(&P.counters[0], 1);
        __syncthreads();
        const long long s = sh[2];
        if (s >= P.B) {
            if (s == P.B && tid == 0 && P.tail_flag) *(volatile int*)P.tail_flag = 1;
            break;
        }
        const uint8_t* syn = P.synd + (size_t)s * m;
        bool nz = false;
        for (int c = tid; c < m; c += BPA_NT) nz |= (syn[c] & 1) != 0;
        // a3: every edge's bit->check message starts at the prior; decisions = 0
        for (int i = tid; i < n; i += BPA_NT) {
            double l0 = P.llr0[i];
            if (P.sel && P.sel[(size_t)s * n + i]) l0 = P.llr0_alt[i];
            for (int k = P.cp[i]; k < P.cp[i + 1]; ++k) msg[P.ce[k]] = l0;
            dec[i] = 0;
            llrt[i] = l0;
        }
        const bool zero = !__syncthreads_or(nz);  // all-zero syndrome: zeros, converge = true, BP not run

        int it_done = 0;
        bool conv = zero;
        if (!zero) {
#pragma clang loop unroll(disable)
            for (int it = 1;; ++it) {
                const bool last = it > P.max_iter;  // only the convergence test of the last bit pass is left
                const double alpha = alpha_for_iteration(P.ms_scaling, it);
                // ---------------- check pass (a4 / a5), speculative; candidate syndrome of the previous decisions
                bool mis = false;
                for (int c = tid; c < m; c += BPA_NT) {
                    const int e0 = P.rp[c], e1 = P.rp[c + 1];
                    const bool sbit = (syn[c] & 1) != 0;
                    unsigned int dpar = sbit ? 1u : 0u;
                    if (P.bp_method == 1) {
                        double t = __DBL_MAX__;
                        bool par = sbit;
                        for (int e = e0; e < e1; ++e) {
                            dpar ^= dec[P.ci[e]];
                            if (last) continue;
                            const double v = msg[e];
                            pre[e] = t;
                            t = min_abs(t, v);
                            par ^= (v <= 0.0);  // a zero counts as negative, as in the reference
                        }
                        if (!last) {
                            double suf = __DBL_MAX__;
                            for (int e = e1 - 1; e >= e0; --e) {
                                const double v = msg[e];
                                const double mag = min_pos(pre[e], suf);
                                msg[e] = flip_sign(mag * alpha, par ^ (v <= 0.0));
                                suf = min_abs(suf, v);
                            }
                        }
                    } else {
                        double t = 1.0;
                        for (int e = e0; e < e1; ++e) {
                            dpar ^= dec[P.ci[e]];
                            if (last) continue;
                            pre[e] = t;
                            const double th = pm_ps_tanh_half(msg[e], P.ps_form);
                            thv[e] = th;
                            t *= th;
                        }
                        if (!last) {
                            t = 1.0;
                            const double sg = sbit ? -1.0 : 1.0;
                            for (int e = e1 - 1; e >= e0; --e) {
                                const double x = pre[e] * t;
                                double o = sg * pm_ps_log_ratio(x, P.ps_form);
                                if (P.ps_clip > 0.0) {  // the comparisons are false for NaN, as on the CPU
                                    if (o > P.ps_clip) o = P.ps_clip;
                                    if (o < -P.ps_clip) o = -P.ps_clip;
                                }
                                msg[e] = o;
                                t *= thv[e];
                            }
                        }
                    }
                    mis |= dpar != 0u;
                }
                const bool any_mis = __syncthreads_or(mis) != 0;  // (also publishes the check pass' messages)
                if (last) {
                    conv = !any_mis;
                    it_done = P.max_iter;
                    break;
                }
                if (!any_mis) {
                    conv = true;
                    it_done = it - 1;
                    break;
                }
                // ---------------- bit pass: posterior, decision, bit -> check (a6 / a7)
                for (int i = tid; i < n; i += BPA_NT) {
                    double l0 = P.llr0[i];
                    if (P.sel && P.sel[(size_t)s * n + i]) l0 = P.llr0_alt[i];
                    const int k0 = P.cp[i], k1 = P.cp[i + 1];
                    double t = l0;
                    for (int k = k0; k < k1; ++k) {
                        const int e = P.ce[k];
                        pre[e] = t;  // ((l0 + c[0]) + ...) + c[d-1]
                        t += msg[e];
                    }
                    llrt[i] = t;
                    dec[i] = (t <= 0.0) ? 1 : 0;
                    double suf = 0.0;  // ((0.0 + c[D-1]) + ...) + c[d+1]
                    for (int k = k1 - 1; k >= k0; --k) {
                        const int e = P.ce[k];
                        const double cm = msg[e];
                        msg[e] = pre[e] + suf;
                        suf += cm;
                    }
                }
                __syncthreads();
            }
        }

        // ---- results (as the other flooding kernels)
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
        for (int i = tid; i < n; i += BPA_NT) {
            const size_t o = (size_t)s * n + i;
            const uint8_t b = dec[i];
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
}

}  // namespace bposd
