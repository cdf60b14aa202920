// bp_kernel.hip.h -- batched belief propagation (flooding schedule, fp64) for gfx950.
//
// Restates rows a3-a7 of SURVEY.md §8 (the BP half of `.decode(syndrome)`,
// /root/reference/README.md:197, /root/reference/src/bposd/css_decode_sim.py:174-202)
// as one persistent HIP kernel:
//
//   * one workgroup decodes one syndrome at a time and pulls the next one from a
//     global atomic queue (iteration counts are wildly bimodal: 5 ... max_iter);
//   * the E edge messages of the Tanner graph live in LDS for the whole decode, in a
//     check-major structure-of-arrays layout  msg[k * m + c]  (k-th edge of check c),
//     and are updated IN PLACE: the check pass overwrites bit->check messages with
//     check->bit messages, the bit pass overwrites them back (every edge belongs to
//     exactly one check and one bit, so no thread reads what another one writes
//     inside a pass).  HBM traffic is m bytes in, a few n bytes out per syndrome;
//   * per-thread graph tables (edge positions of "my" bits) and the running LLRs are
//     held in registers: a thread owns the same checks / bits for every syndrome;
//   * convergence (H * decoding == syndrome) is tracked incrementally: a bit whose
//     hard decision flips toggles its checks' mismatch bits with LDS atomics and
//     updates one mismatch counter, so the test per iteration is one LDS read.
//
// Arithmetic is fp64 and mirrors the association order of the reference algorithm
// (prefix sums from the top of a column, suffix sums from the bottom, one multiply
// by the pre-formed (sign * alpha) factor) so that the LLRs -- which fix the OSD
// column order -- are bit-identical to a CPU run.  Build with -ffp-contract=off.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bposd {

struct BpParams {
    int m, n;
    long long B;
    int max_iter;
    double ms_scaling;  // 0 => 1 - 2^-it
    int osd_enabled;    // 0: osd off, results = bp decoding even when not converged
    int dc_rt;          // number of edge slots per check in the LDS layout (== template DC)
    const uint8_t* __restrict__ synd;   // [B, m]
    const double* __restrict__ llr0;    // [n]
    const int* __restrict__ chk_deg;    // [m]
    const int* __restrict__ var_deg;    // [n]
    const int* __restrict__ var_pos;    // [DVmax * n], entry d*n+i = k*m + c
    const int* __restrict__ var_row;    // [DVmax * n], entry d*n+i = c
    uint8_t* __restrict__ out_bp;       // [B, n] nullable
    uint8_t* __restrict__ out_osd0;     // [B, n] nullable
    uint8_t* __restrict__ out_osdw;     // [B, n]
    uint8_t* __restrict__ out_conv;     // [B] nullable
    int* __restrict__ out_iters;        // [B] nullable
    double* __restrict__ out_llr;       // [B, n] nullable
    double* __restrict__ llr_ws;        // [cap, n]: LLRs of non-converged syndromes, by list slot
    int* __restrict__ osd_list;         // [cap]: syndrome index of each slot
    int* __restrict__ counters;         // [0] work queue, [1] osd count
    unsigned long long* __restrict__ iter_total;  // sum of iterations executed
};

__device__ __forceinline__ double alpha_for_iteration(double ms_scaling, int it) {
    // a4: alpha = ms_scaling_factor, or 1 - 2^-it when the factor is 0 (README.md:184).
    if (ms_scaling != 0.0) return ms_scaling;
    if (it > 1074) return 1.0;
    return 1.0 - __builtin_ldexp(1.0, -it);
}

// DC / DV: edge slots per check / per bit (compile-time maxima)
// CPT / VPT: checks / bits owned by one thread;  blockDim.x * CPT >= m, blockDim.x * VPT >= n
// REG: every check has exactly DC edges and every bit exactly DV (no predication)
// METHOD: 0 product-sum, 1 min-sum
// MINW: minimum waves per SIMD the register allocation must allow (occupancy target)
template <int DC, int DV, int CPT, int VPT, int MAXNT, int MINW, bool REG, int METHOD>
__global__ __launch_bounds__(MAXNT, MINW) void bp_kernel(const BpParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int m = P.m, n = P.n;
    const int NT = blockDim.x;
    const int tid = threadIdx.x;
    const int lane = tid & 63;

    double* msg = reinterpret_cast<double*>(smem);
    const int dwords = (m + 31) >> 5;
    unsigned int* diffw = reinterpret_cast<unsigned int*>(msg + (size_t)DC * m);
    int* sh = reinterpret_cast<int*>(diffw + ((dwords + 1) & ~1));  // [0] mismatch [1] syndrome id [2] slot

    // ---- per-thread graph tables (same for every syndrome this workgroup decodes)
    int vpos[VPT][DV], vrow[VPT][DV], vdeg[VPT];
    double l0[VPT];
#pragma unroll
    for (int r = 0; r < VPT; ++r) {
        const int i = tid + r * NT;
        vdeg[r] = 0;
        l0[r] = 0.0;
#pragma unroll
        for (int d = 0; d < DV; ++d) { vpos[r][d] = 0; vrow[r][d] = 0; }
        if (i < n) {
            vdeg[r] = REG ? DV : P.var_deg[i];
            l0[r] = P.llr0[i];
#pragma unroll
            for (int d = 0; d < DV; ++d) {
                if (d < vdeg[r]) {
                    vpos[r][d] = P.var_pos[(size_t)d * n + i];
                    vrow[r][d] = P.var_row[(size_t)d * n + i];
                }
            }
        }
    }
    int cdeg[CPT];
#pragma unroll
    for (int r = 0; r < CPT; ++r) {
        const int c = tid + r * NT;
        cdeg[r] = (c < m) ? (REG ? DC : P.chk_deg[c]) : 0;
    }

    for (;;) {
        // ---- pull the next syndrome
        if (tid == 0) {
            sh[0] = 0;
            sh[1] = atomicAdd(&P.counters[0], 1);
        }
        __syncthreads();
        const long long s = sh[1];
        if (s >= P.B) break;  // uniform: every wave reaches this with the same value

        // ---- syndrome bits of my checks; mismatch bits start as the syndrome itself
        int sbit[CPT];
#pragma unroll
        for (int r = 0; r < CPT; ++r) {
            const int c = tid + r * NT;
            sbit[r] = (c < m) ? (P.synd[(size_t)s * m + c] & 1) : 0;
            const unsigned long long bal = __ballot(sbit[r]);
            if (lane == 0) {
                const int w0 = (c >> 5);  // c is a multiple of 64 for lane 0
                if (w0 < dwords) diffw[w0] = (unsigned int)bal;
                if (w0 + 1 < dwords) diffw[w0 + 1] = (unsigned int)(bal >> 32);
                const int pc = __popcll(bal);
                if (pc) atomicAdd(&sh[0], pc);
            }
        }
        // ---- a3: every edge's bit->check message starts at the prior
        double llr[VPT];
        int dec[VPT];
#pragma unroll
        for (int r = 0; r < VPT; ++r) {
            llr[r] = l0[r];
            dec[r] = 0;
#pragma unroll
            for (int d = 0; d < DV; ++d)
                if (d < vdeg[r]) msg[vpos[r][d]] = l0[r];
        }
        __syncthreads();

        int it_done = 0;
        bool conv = (sh[0] == 0);  // all-zero syndrome: zeros, converge = true, BP not run (A.2)
        if (!conv) {
            for (int it = 1; it <= P.max_iter; ++it) {
                // =================== check -> bit pass (a4 / a5) ===================
                const double alpha = alpha_for_iteration(P.ms_scaling, it);
#pragma unroll
                for (int r = 0; r < CPT; ++r) {
                    const int c = tid + r * NT;
                    const int deg = cdeg[r];
                    if (deg > 0) {
                        double v[DC];
#pragma unroll
                        for (int k = 0; k < DC; ++k)
                            if (REG || k < deg) v[k] = msg[k * m + c];
                        if (METHOD == 1) {
                            double pre[DC];
                            int neg[DC];
                            int par = sbit[r];
                            double t = __DBL_MAX__;
#pragma unroll
                            for (int k = 0; k < DC; ++k) {
                                if (REG || k < deg) {
                                    neg[k] = (v[k] <= 0.0) ? 1 : 0;
                                    par += neg[k];
                                    pre[k] = t;
                                    const double a = __builtin_fabs(v[k]);
                                    if (a < t) t = a;
                                }
                            }
                            t = __DBL_MAX__;
#pragma unroll
                            for (int k = DC - 1; k >= 0; --k) {
                                if (REG || k < deg) {
                                    double mag = pre[k];
                                    if (t < mag) mag = t;
                                    const double f = ((par + neg[k]) & 1) ? -alpha : alpha;
                                    msg[k * m + c] = mag * f;
                                    const double a = __builtin_fabs(v[k]);
                                    if (a < t) t = a;
                                }
                            }
                        } else {
                            double pre[DC], th[DC];
                            double t = 1.0;
#pragma unroll
                            for (int k = 0; k < DC; ++k) {
                                if (REG || k < deg) {
                                    pre[k] = t;
                                    th[k] = tanh(v[k] / 2);
                                    t *= th[k];
                                }
                            }
                            t = 1.0;
                            const double sg = sbit[r] ? -1.0 : 1.0;
#pragma unroll
                            for (int k = DC - 1; k >= 0; --k) {
                                if (REG || k < deg) {
                                    const double x = pre[k] * t;
                                    msg[k * m + c] = sg * log((1 + x) / (1 - x));
                                    t *= th[k];
                                }
                            }
                        }
                    }
                }
                __syncthreads();
                // ============ bit pass: posterior, decision, bit -> check (a6 / a7) ============
                int delta = 0;
#pragma unroll
                for (int r = 0; r < VPT; ++r) {
                    const int deg = vdeg[r];
                    if (tid + r * NT < n) {
                        double cm[DV], pre[DV];
                        double t = l0[r];
#pragma unroll
                        for (int d = 0; d < DV; ++d)
                            if (REG || d < deg) cm[d] = msg[vpos[r][d]];
#pragma unroll
                        for (int d = 0; d < DV; ++d) {
                            if (REG || d < deg) {
                                pre[d] = t;  // prefix from the top of the column (prior included)
                                t += cm[d];
                            }
                        }
                        llr[r] = t;
                        const int dnew = (t <= 0.0) ? 1 : 0;
                        double suf = 0.0;  // suffix from the bottom of the column
#pragma unroll
                        for (int d = DV - 1; d >= 0; --d) {
                            if (REG || d < deg) {
                                msg[vpos[r][d]] = pre[d] + suf;
                                suf += cm[d];
                            }
                        }
                        if (dnew != dec[r]) {
                            dec[r] = dnew;
#pragma unroll
                            for (int d = 0; d < DV; ++d) {
                                if (REG || d < deg) {
                                    const int c = vrow[r][d];
                                    const unsigned int bit = 1u << (c & 31);
                                    const unsigned int old = atomicXor(&diffw[c >> 5], bit);
                                    delta += (old & bit) ? -1 : 1;
                                }
                            }
                        }
                    }
                }
                if (delta != 0) atomicAdd(&sh[0], delta);
                __syncthreads();
                it_done = it;
                if (sh[0] == 0) { conv = true; break; }
            }
        }

        // ---- results
        const bool to_osd = (!conv) && P.osd_enabled;
        if (tid == 0) {
            if (to_osd) {
                const int slot = atomicAdd(&P.counters[1], 1);
                P.osd_list[slot] = (int)s;
                sh[2] = slot;
            }
            if (P.out_conv) P.out_conv[s] = conv ? 1 : 0;
            if (P.out_iters) P.out_iters[s] = it_done;
            if (it_done) atomicAdd(P.iter_total, (unsigned long long)it_done);
        }
        __syncthreads();  // publishes sh[2]; also fences this syndrome's reads of sh[0]
        const int slot = to_osd ? sh[2] : 0;
#pragma unroll
        for (int r = 0; r < VPT; ++r) {
            const int i = tid + r * NT;
            if (i < n) {
                const size_t o = (size_t)s * n + i;
                const uint8_t b = (uint8_t)dec[r];
                if (P.out_bp) P.out_bp[o] = b;
                if (!to_osd) {
                    P.out_osdw[o] = b;
                    if (P.out_osd0) P.out_osd0[o] = b;
                } else {
                    P.llr_ws[(size_t)slot * n + i] = llr[r];
                }
                if (P.out_llr) P.out_llr[o] = llr[r];
            }
        }
        // the next iteration's first barrier orders these reads before sh[] is rewritten:
        // tid 0 writes sh[0..1] only after it has itself passed the barrier above.
        __syncthreads();
    }
}

}  // namespace bposd
