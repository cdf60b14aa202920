// bp_serial_kernel.hip.h -- belief propagation with the SERIAL schedule (`schedule="serial"` of
// the ldpc v2 decoder; the reference itself never passes it -- css_decode_sim.py:444-463 -- so this path is built for
// API completeness and exactness, not tuned like the flooding kernels).
//
// Serial schedule: inside an iteration the bits are visited in ascending index and every bit sees the messages the bits
// before it have just written.  Bit j depends on an earlier bit i only if the two share a check, so the host sorts the
// bits into LEVELS (level(j) = 1 + max level of the earlier bits that share a check with j): bits of one level touch
// disjoint sets of checks, run in parallel, and give bit for bit what the sequential sweep gives.  One workgroup decodes
// one syndrome at a time (persistent, atomic queue); the E bit->check messages live in a per-workgroup slice of a global
// workspace indexed by CSR edge id (levels are separated by __syncthreads(): workgroup-scope release / acquire, the
// waves share the CU's L1); hard decisions sit in LDS for the convergence test after every sweep.
//
// Per bit (the CPU restatement under tests' checker directory walks the same steps): LLR := prior; for its checks top to
// bottom: message := f(current bit->check messages of the check's OTHER edges); bit->check := LLR (prefix); LLR +=
// message.  Decision = (LLR <= 0).  Then bottom to top: bit->check += suffix sum.  fp64, no contraction.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bp_kernel.hip.h"
#include "portable_math.h"

namespace bposd {

constexpr int BPS_NT = 256;
constexpr int BPS_MAXDV = 8;

struct BpSerialParams {
    int m, n, E;
    long long B;
    int max_iter;
    int bp_method;  // 0 product-sum, 1 min-sum
    double ms_scaling;
    double ps_clip;
    int ps_form;  // product-sum evaluation order (portable_math.h: pm_ps_tanh_half)
    int osd_enabled;
    int nlevels;
    const uint8_t* __restrict__ synd;     // [B, m]
    const double* __restrict__ llr0;      // [n]
    const uint8_t* __restrict__ sel;      // [B, n] nullable
    const double* __restrict__ llr0_alt;  // [n]
    const int* __restrict__ rp;           // CSR indptr [m + 1]
    const int* __restrict__ ci;           // CSR indices [E]
    const int* __restrict__ cp;           // CSC indptr [n + 1]
    const int* __restrict__ ce;           // [E] CSR edge ids of a column, ascending row
    const int* __restrict__ erow;         // [E] row of a CSR edge
    const int* __restrict__ lvl_ptr;      // [nlevels + 1]
    const int* __restrict__ lvl_bits;     // [n] bits by level, ascending inside a level
    double* __restrict__ msg_ws;          // [gridDim.x][E] bit->check messages
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
};

__host__ __device__ inline size_t bp_serial_lds_bytes(int n) { return (size_t)((n + 15) & ~15) + 8 * 4; }

__global__ __launch_bounds__(BPS_NT) void bp_serial_kernel(const BpSerialParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int m = P.m, n = P.n;
    const int tid = threadIdx.x;
    unsigned char* dec = smem;
    int* sh = reinterpret_cast<int*>(smem + ((n + 15) & ~15));
    double* b2c = P.msg_ws + (size_t)blockIdx.x * P.E;
    double* llrt = P.llr_tmp + (size_t)blockIdx.x * n;

    for (;;) {
        if (tid == 0) sh[2] = atomicAdd(&P.counters[0], 1);
        __syncthreads();
        const long long s = sh[2];
        if (s >= P.B) break;
        const uint8_t* syn = P.synd + (size_t)s * m;
        bool nz = false;
        for (int c = tid; c < m; c += BPS_NT) nz |= (syn[c] & 1) != 0;
        for (int i = tid; i < n; i += BPS_NT) {
            double l0 = P.llr0[i];
            if (P.sel && P.sel[(size_t)s * n + i]) l0 = P.llr0_alt[i];
            for (int k = P.cp[i]; k < P.cp[i + 1]; ++k) b2c[P.ce[k]] = l0;
            dec[i] = 0;
            llrt[i] = l0;
        }
        const bool zero = !__syncthreads_or(nz);  // all-zero syndrome: zeros, converge = true, BP not run (A.2)

        int it_done = 0;
        bool conv = zero;
        if (!zero) {
#pragma clang loop unroll(disable)
            for (int it = 1; it <= P.max_iter; ++it) {
                const double alpha = alpha_for_iteration(P.ms_scaling, it);
                for (int lv = 0; lv < P.nlevels; ++lv) {
                    const int lo = P.lvl_ptr[lv], hi = P.lvl_ptr[lv + 1];
                    for (int q = lo + tid; q < hi; q += BPS_NT) {
                        const int i = P.lvl_bits[q];
                        double l0 = P.llr0[i];
                        if (P.sel && P.sel[(size_t)s * n + i]) l0 = P.llr0_alt[i];
                        double llr = l0;
                        double c2b[BPS_MAXDV];
                        const int k0 = P.cp[i], deg = P.cp[i + 1] - k0;
                        for (int d = 0; d < deg; ++d) {
                            const int e = P.ce[k0 + d], c = P.erow[e];
                            const int g0 = P.rp[c], g1 = P.rp[c + 1];
                            double msg;
                            if (P.bp_method == 0) {
                                double prod = 1.0;
                                for (int g = g0; g < g1; ++g)
                                    if (g != e) prod *= pm_ps_tanh_half(b2c[g], P.ps_form);
                                msg = ((syn[c] & 1) ? -1.0 : 1.0) * pm_ps_log_ratio(prod, P.ps_form);
                                if (P.ps_clip > 0.0) {
                                    if (msg > P.ps_clip) msg = P.ps_clip;
                                    if (msg < -P.ps_clip) msg = -P.ps_clip;
                                }
                            } else {
                                int sgn = syn[c] & 1;
                                double temp = __DBL_MAX__;
                                for (int g = g0; g < g1; ++g) {
                                    if (g == e) continue;
                                    const double v = b2c[g];
                                    const double a = fabs(v);
                                    if (a < temp) temp = a;
                                    if (v <= 0.0) sgn += 1;
                                }
                                const double message_sign = (sgn & 1) ? -1.0 : 1.0;
                                msg = alpha * message_sign * temp;
                            }
                            c2b[d] = msg;
                            b2c[e] = llr;  // prefix from the top of the column (prior included)
                            llr += msg;
                        }
                        llrt[i] = llr;
                        dec[i] = (llr <= 0.0) ? 1 : 0;
                        double temp = 0.0;  // suffix from the bottom of the column
                        for (int d = deg - 1; d >= 0; --d) {
                            const int e = P.ce[k0 + d];
                            b2c[e] += temp;
                            temp += c2b[d];
                        }
                    }
                    __syncthreads();
                }
                // candidate syndrome of this sweep's decisions against the input syndrome
                bool mis = false;
                for (int c = tid; c < m; c += BPS_NT) {
                    unsigned int par = 0u;
                    for (int g = P.rp[c]; g < P.rp[c + 1]; ++g) par ^= dec[P.ci[g]];
                    mis |= par != (unsigned int)(syn[c] & 1);
                }
                it_done = it;
                if (!__syncthreads_or(mis)) {
                    conv = true;
                    break;
                }
            }
        }

        // ---- results (as the flooding kernels)
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
        for (int i = tid; i < n; i += BPS_NT) {
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
