// bp_kernel.hip.h -- batched belief propagation (flooding schedule, fp64) for gfx950.
//
// Restates rows a3-a7 of SURVEY.md §8 (the BP half of `.decode(syndrome)`,
// /root/reference/README.md:197, /root/reference/src/bposd/css_decode_sim.py:174-202)
// as one persistent HIP kernel:
//
//   * one workgroup decodes one syndrome at a time and pulls the next one from a
//     global atomic queue (iteration counts are wildly bimodal: 5 ... max_iter);
//   * the E edge messages of the Tanner graph live in LDS for the whole decode, in a
//     check-major structure-of-arrays layout  msg[k * MP + c]  (k-th edge of check c,
//     MP = checks padded to threads x CPT), and are updated IN PLACE: the check pass
//     overwrites bit->check messages with check->bit messages, the bit pass overwrites
//     them back (every edge belongs to exactly one check and one bit, so no thread reads
//     what another one writes inside a pass).  Consecutive lanes own consecutive checks
//     => conflict-free ds_read_b64 / ds_write_b64 with immediate offsets in the check
//     pass.  HBM traffic is m bytes in, a few n bytes out per syndrome;
//   * per-thread graph tables (LDS slots of "my" bits' edges, priors) and the running
//     LLRs are held in registers: a thread owns the same checks / bits for every syndrome;
//     padded checks / bits point at a dummy LDS slot so both passes are branch-free;
//   * convergence (H * decoding == syndrome) is tracked incrementally: a bit whose hard
//     decision flips toggles its checks' bits in an LDS mismatch bitmap (fire-and-forget
//     ds_xor); at the top of the next iteration every check thread looks at its own bit
//     and a wave with any mismatch raises an LDS flag that is read after the barrier the
//     check pass needs anyway.  The check pass of the iteration that discovers
//     convergence is speculative work (one pass per syndrome).
//
// Arithmetic is fp64 and mirrors the association order of the reference algorithm
// (prefix sums from the top of a column, suffix sums from the bottom, one multiply
// by the pre-formed (sign * alpha) factor) so that the LLRs -- which fix the OSD
// column order -- are bit-identical to a CPU run.  Build with -ffp-contract=off.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "portable_math.h"  // bit-reproducible tanh / log of the product-sum update

namespace bposd {

// LDS message array type: volatile LDS-address-space accesses stop hipcc from fusing pairs into
// ds_read2st64_b64 / ds_write2st64_b64 (half the LDS rate of two ds_read_b64 on gfx950,
// MI355X_MICROARCH.md §LDS; measured here: BP kernel 38.8 -> 35.2 ms).  -DBPOSD_LDS_MERGED restores
// the fused form for A/B runs.
#ifndef BPOSD_LDS_MERGED
typedef volatile __attribute__((address_space(3))) double* msg_ptr;
#else
typedef double* msg_ptr;
#endif

struct BpParams {
    int m, n;
    long long B;
    int max_iter;
    double ms_scaling;  // 0 => 1 - 2^-it
    double ps_clip;     // product-sum: 0 = no clipping (upstream), C > 0 = check->bit messages clamped to [-C, C]
    int osd_enabled;    // 0: osd off, results = bp decoding even when not converged
    int mp;             // padded check count = blockDim.x * CPT (stride of the LDS layout)
    const uint8_t* __restrict__ synd;   // [B, m]
    const double* __restrict__ llr0;    // [n]
    const uint8_t* __restrict__ sel;    // [B, n] nullable: per-syndrome choice between llr0 and llr0_alt
    const double* __restrict__ llr0_alt;  // [n] priors of the alternative channel (used where sel != 0)
    const int* __restrict__ chk_deg;    // [m]
    int np;             // bit positions = blockDim.x * VPT; position p is owned by thread p % blockDim.x
    const int* __restrict__ pos_bit;    // [np] bit handled at position p, -1 = padding.  The order is chosen by
                                        // the host so that the bit pass's LDS gathers are bank-conflict free
    const int* __restrict__ var_deg;    // [np]
    const int* __restrict__ var_pos;    // [DVmax * np], entry d*np+p = k*mp + c
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
    int* __restrict__ tail_flag;  // nullable, host-visible: set to 1 by the workgroup that finds the queue empty (the tail begins)
    int packed_io;  // 1: synd is [B][ceil(m/64)] and out_bp / out_osd0 / out_osdw are [B][ceil(n/64)] little-endian 64-bit words
                    // (bposd_decode_batch_packed*: no unpack / pack kernels around the call)
};

// entry c of row s of the syndromes: a 0/1 byte, or bit (c & 63) of word (c >> 6) in the packed form
__device__ __forceinline__ bool bp_synd_bit(const uint8_t* synd, int packed, long long s, int m, int c) {
    if (packed) return (((const unsigned long long*)synd)[(size_t)s * (size_t)((m + 63) >> 6) + (c >> 6)] >> (c & 63)) & 1ull;
    return (synd[(size_t)s * m + c] & 1) != 0;
}

// Packed result rows of one syndrome: the workgroup's hard decisions (thread-owned bits, any layout) meet in an LDS bitmap
// (`bits`: 2 * ceil(n/64) dead words, e.g. the start of the message array once the iterations are over), then ceil(n/64)
// threads store the 64-bit words.  bit_of(r) = index of the thread's r-th bit or -1, dec_of(r) = its decision.
template <int NBITS, class BitOf, class DecOf>
__device__ __forceinline__ void bp_store_packed_rows(unsigned int* bits, int tid, int nthreads, int n, long long s, bool to_osd,
                                                     unsigned long long* out_bp, unsigned long long* out_osd0,
                                                     unsigned long long* out_osdw, BitOf bit_of, DecOf dec_of) {
    const int wpn = (n + 63) >> 6;
    for (int w = tid; w < 2 * wpn; w += nthreads) bits[w] = 0u;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < NBITS; ++r) {
        const int i = bit_of(r);
        if (i >= 0 && dec_of(r)) atomicOr(&bits[i >> 5], 1u << (i & 31));
    }
    __syncthreads();
    for (int w = tid; w < wpn; w += nthreads) {
        const unsigned long long v = (unsigned long long)bits[2 * w] | ((unsigned long long)bits[2 * w + 1] << 32);
        const size_t o = (size_t)s * wpn + w;
        if (out_bp) out_bp[o] = v;
        if (!to_osd) {
            out_osdw[o] = v;
            if (out_osd0) out_osd0[o] = v;
        }
    }
    __syncthreads();  // the bitmap's words are message slots again
}

__host__ __device__ inline size_t bp_lds_bytes(int DC, int mp) {
    // messages (+1 dummy slot, padded to 16 B) + mismatch bitmap + 8 control words
    return ((size_t)DC * mp + 2) * 8 + (size_t)(mp / 32 + 2) * 4 + 8 * 4;
}

__device__ __forceinline__ double alpha_for_iteration(double ms_scaling, int it) {
    // a4: alpha = ms_scaling_factor, or 1 - 2^-it when the factor is 0 (README.md:184).
    if (ms_scaling != 0.0) return ms_scaling;
    if (it > 1074) return 1.0;
    return 1.0 - __builtin_ldexp(1.0, -it);
}

// The same value as an integer bit pattern, formed with shifts only (wave-uniform inputs: scalar instructions, where the
// floating-point form costs ~10 vector instructions per iteration).  1 - 2^-it for 1 <= it <= 53 is exactly representable:
// biased exponent 0x3FE, the top it - 1 mantissa bits set; from it = 54 on the difference rounds to 1.0 (ties to even),
// which is also what the subtraction above returns.
__device__ __forceinline__ unsigned long long alpha_bits_for_iteration(double ms_scaling, int it) {
    if (ms_scaling != 0.0) return (unsigned long long)__double_as_longlong(ms_scaling);
    if (it >= 54) return 0x3FF0000000000000ull;
    const int sh = 53 - it;  // 0 .. 52
    return 0x3FE0000000000000ull | ((0x000FFFFFFFFFFFFFull >> sh) << sh);
}

// r = min(t, |v|) exactly as `a = fabs(v); if (a < t) t = a;` for every non-NaN t:
// v_min_f64 returns the non-NaN operand, and t never is NaN (it starts at DBL_MAX).
__device__ __forceinline__ double min_abs(double t, double v) {
    double r;
    asm("v_min_f64 %0, %1, |%2|" : "=v"(r) : "v"(t), "v"(v));
    return r;
}
__device__ __forceinline__ double min_pos(double a, double b) {
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// x with its sign flipped when `flip` (exact: IEEE multiplication is sign-symmetric, so
// mag * (-alpha) == -(mag * alpha) bit for bit)
__device__ __forceinline__ double flip_sign(double x, bool flip) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(x);
    return __longlong_as_double((long long)(u ^ (flip ? 0x8000000000000000ull : 0ull)));
}

// DC / DV: edge slots per check / per bit (compile-time maxima)
// CPT / VPT: checks / bits owned by one thread;  blockDim.x * CPT >= m, blockDim.x * VPT >= n
// REG: every check has exactly DC edges and every bit exactly DV (no predication)
// METHOD: 0 product-sum (two divisions per edge), 2 product-sum in the reference's operation order (portable_math.h), 1 min-sum
// MINW: minimum waves per SIMD the register allocation must allow (occupancy target)
// MPT: compile-time check stride (0 = take P.mp); always a power of two and == blockDim.x * CPT,
//      so LDS offsets k * MP * 8 become instruction immediates
template <int DC, int DV, int CPT, int VPT, int MAXNT, int MINW, bool REG, int METHOD, int MPT>
__global__ __launch_bounds__(MAXNT, MINW) void bp_kernel(const BpParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int m = P.m, n = P.n;
    const int MP = MPT > 0 ? MPT : P.mp;  // power of two, == blockDim.x * CPT
    const int NT = MPT > 0 ? MPT / CPT : (int)blockDim.x;
    const int tid = threadIdx.x;
    const int lane = tid & 63;

    double* msg_plain = reinterpret_cast<double*>(smem);
    msg_ptr msg = (msg_ptr)msg_plain;
    const int dummy = DC * MP;  // slot nobody's result depends on
    unsigned int* diffw = reinterpret_cast<unsigned int*>(msg_plain + (size_t)DC * MP + 2);
    int* sh = reinterpret_cast<int*>(diffw + (MP / 32 + 2));  // [0],[1] mismatch flags, [2] syndrome id, [3] slot

    // ---- per-thread graph tables (same for every syndrome this workgroup decodes)
    int vaddr[VPT][DV], vdeg[VPT];
    bool vvalid[VPT];
    double l0[VPT];
    const int NP = P.np;
#pragma unroll
    for (int r = 0; r < VPT; ++r) {
        const int p = tid + r * NT;
        const int bit = P.pos_bit[p];
        vvalid[r] = bit >= 0;
        vdeg[r] = vvalid[r] ? (REG ? DV : P.var_deg[p]) : 0;
        l0[r] = vvalid[r] ? P.llr0[bit] : 1.0;
#pragma unroll
        for (int d = 0; d < DV; ++d) vaddr[r][d] = (d < vdeg[r]) ? P.var_pos[(size_t)d * NP + p] : dummy;
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
            sh[1] = 0;
            sh[2] = atomicAdd(&P.counters[0], 1);
        }
        __syncthreads();
        const long long s = sh[2];
        if (s >= P.B) {
            // the chunk loop of the host-pointer API launches the next chunk's kernels when this one's tail begins
            if (s == P.B && tid == 0 && P.tail_flag) *(volatile int*)P.tail_flag = 1;
            break;
        }  // uniform: every wave reaches this with the same value

        // ---- syndrome bits of my checks; the mismatch bitmap starts as the syndrome itself
        bool sbit[CPT];
#pragma unroll
        for (int r = 0; r < CPT; ++r) {
            const int c = tid + r * NT;
            sbit[r] = (c < m) ? bp_synd_bit(P.synd, P.packed_io, s, m, c) : false;
            const unsigned long long bal = __ballot(sbit[r]);
            if (lane == 0) {
                const int w0 = (c >> 5);  // c is a multiple of 64 for lane 0
                diffw[w0] = (unsigned int)bal;
                diffw[w0 + 1] = (unsigned int)(bal >> 32);
                if (bal) sh[0] = 1;
            }
        }
        // ---- per-syndrome two-valued channel (css_decode_sim.py:207-248): pick this shot's priors
        if (P.sel) {
#pragma unroll
            for (int r = 0; r < VPT; ++r) {
                const int i = P.pos_bit[tid + r * NT];
                if (i >= 0) l0[r] = P.sel[(size_t)s * n + i] ? P.llr0_alt[i] : P.llr0[i];
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
                if (REG || d < vdeg[r]) msg[vaddr[r][d]] = l0[r];
        }
        __syncthreads();

        int it_done = 0;
        bool conv = (sh[0] == 0);  // all-zero syndrome: zeros, converge = true, BP not run (A.2)
        if (!conv) {
#pragma clang loop unroll(disable)
            for (int it = 1;; ++it) {
                const int fi = it & 1;
                // ---- convergence test of iteration it-1: does any of my checks still mismatch?
                // (at it = 1 the bitmap is the non-zero syndrome itself, so the flag is raised)
                {
                    bool mis = false;
#pragma unroll
                    for (int r = 0; r < CPT; ++r) {
                        const int c = tid + r * NT;
                        mis |= ((diffw[c >> 5] >> (c & 31)) & 1u) != 0;
                    }
                    const unsigned long long anym = __ballot(mis);
                    if (lane == 0 && anym) sh[fi] = 1;
                }
                if (it > P.max_iter) {  // only the test is left
                    __syncthreads();
                    conv = (sh[fi] == 0);
                    it_done = P.max_iter;
                    break;
                }
                // =================== check -> bit pass (a4 / a5), speculative for it >= 2 ===========
                const double alpha = alpha_for_iteration(P.ms_scaling, it);
#pragma unroll
                for (int r = 0; r < CPT; ++r) {
                    const int c = tid + r * NT;
                    const int deg = cdeg[r];
                    if (REG || deg > 0) {
                        msg_ptr mc = msg + c;
                        // An absent edge (k >= deg) enters as +DBL_MAX: neutral for the running minima and for the sign parity
                        // (min-sum), tanh = 1 exactly (product-sum); its result is never stored.  Every element of the
                        // arrays below is defined on every path -- partially defined arrays became loop-carried registers
                        // (one per element) and were the source of this kernel's scratch.
                        double v[DC];
#pragma unroll
                        for (int k = 0; k < DC; ++k) {
                            v[k] = __DBL_MAX__;
                            if (REG || k < deg) v[k] = mc[k * MP];
                        }
                        if (METHOD == 1) {
                            // parity of (syndrome bit + #non-positive inputs); zero counts as negative
                            bool neg[DC];
                            bool par = sbit[r];
#pragma unroll
                            for (int k = 0; k < DC; ++k) {
                                neg[k] = (v[k] <= 0.0);  // (+DBL_MAX: false)
                                par ^= neg[k];
                            }
                            // forward / backward running minima of |b2c|, both started at DBL_MAX
                            double pre[DC], suf[DC];
                            pre[0] = __DBL_MAX__;
#pragma unroll
                            for (int k = 1; k < DC; ++k) pre[k] = min_abs(pre[k - 1], v[k - 1]);  // (min with +DBL_MAX: unchanged)
                            suf[DC - 1] = __DBL_MAX__;
#pragma unroll
                            for (int k = DC - 2; k >= 0; --k) suf[k] = min_abs(suf[k + 1], v[k + 1]);
#pragma unroll
                            for (int k = 0; k < DC; ++k) {
                                if (REG || k < deg) {
                                    const double mag = (k == 0) ? suf[0] : (k == DC - 1 ? pre[DC - 1] : min_pos(pre[k], suf[k]));
                                    mc[k * MP] = flip_sign(mag * alpha, par ^ neg[k]);
                                }
                            }
                        } else {
                            double pre[DC], th[DC];
                            double t = 1.0;
#pragma unroll
                            for (int k = 0; k < DC; ++k) {
                                pre[k] = t;
                                th[k] = 1.0;
                                if (REG || k < deg) {
                                    th[k] = pm_ps_tanh_half(v[k], METHOD == 0);
                                    t *= th[k];
                                }
                            }
                            t = 1.0;
                            const double sg = sbit[r] ? -1.0 : 1.0;
#pragma unroll
                            for (int k = DC - 1; k >= 0; --k) {
                                if (REG || k < deg) {
                                    const double x = pre[k] * t;
                                    double o = sg * pm_ps_log_ratio(x, METHOD == 0);
                                    if (P.ps_clip > 0.0) {  // uniform; the comparisons are false for NaN, as on the CPU
                                        if (o > P.ps_clip) o = P.ps_clip;
                                        if (o < -P.ps_clip) o = -P.ps_clip;
                                    }
                                    mc[k * MP] = o;
                                    t *= th[k];
                                }
                            }
                        }
                    }
                }
                __syncthreads();
                if (sh[fi] == 0) {  // iteration it-1 had already converged
                    conv = true;
                    it_done = it - 1;
                    break;
                }
                if (tid == 0) sh[fi ^ 1] = 0;  // next written after the barrier below, last read before the one above
                // ============ bit pass: posterior, decision, bit -> check (a6 / a7) ============
                double cm[VPT][DV];
#pragma unroll
                for (int r = 0; r < VPT; ++r)
#pragma unroll
                    for (int d = 0; d < DV; ++d) cm[r][d] = msg[vaddr[r][d]];
#pragma unroll
                for (int r = 0; r < VPT; ++r) {
                    const int deg = vdeg[r];
                    double pre[DV];
                    double t = l0[r];
#pragma unroll
                    for (int d = 0; d < DV; ++d) {
                        pre[d] = t;  // prefix from the top of the column (prior included); defined for absent edges too
                        if (REG || d < deg) t += cm[r][d];
                    }
                    llr[r] = t;
                    const int dnew = (t <= 0.0) ? 1 : 0;
                    double suf = 0.0;  // suffix from the bottom of the column
#pragma unroll
                    for (int d = DV - 1; d >= 0; --d) {
                        if (REG || d < deg) {
                            msg[vaddr[r][d]] = pre[d] + suf;
                            suf += cm[r][d];
                        }
                    }
                    if (vvalid[r] && dnew != dec[r]) {
                        dec[r] = dnew;
#pragma unroll
                        for (int d = 0; d < DV; ++d) {
                            if (REG || d < deg) {
                                int pos = vaddr[r][d];
                                // opaque copy: keeps the (rare-path) bitmap address / mask arithmetic inside
                                // this branch instead of being hoisted out of the iteration loop and spilled
                                asm volatile("" : "+v"(pos));
                                const int c = pos & (MP - 1);
                                atomicXor(&diffw[c >> 5], 1u << (c & 31));
                            }
                        }
                    }
                }
                __syncthreads();
            }
        }

        // ---- results
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
        __syncthreads();  // publishes sh[3]; also fences this syndrome's reads of sh[0..1]
        const int slot = to_osd ? sh[3] : 0;
        if (P.packed_io)
            bp_store_packed_rows<VPT>((unsigned int*)smem, tid, NT, n, s, to_osd, (unsigned long long*)P.out_bp, (unsigned long long*)P.out_osd0,
                                      (unsigned long long*)P.out_osdw, [&](int r) { return P.pos_bit[tid + r * NT]; }, [&](int r) { return dec[r] != 0; });
#pragma unroll
        for (int r = 0; r < VPT; ++r) {
            const int i = P.pos_bit[tid + r * NT];  // reloaded here (not kept in registers across the BP loop)
            if (i >= 0) {
                const size_t o = (size_t)s * n + i;
                const uint8_t b = (uint8_t)dec[r];
                if (!P.packed_io) {
                    if (P.out_bp) P.out_bp[o] = b;
                    if (!to_osd) {
                        P.out_osdw[o] = b;
                        if (P.out_osd0) P.out_osd0[o] = b;
                    }
                }
                if (to_osd) P.llr_ws[(size_t)slot * n + i] = llr[r];
                if (P.out_llr) P.out_llr[o] = llr[r];
            }
        }
        // tid 0 rewrites sh[] only after it has itself passed the barrier above, and every
        // thread has read sh[3] by the time it reaches the next loop-top barrier.
        __syncthreads();
    }
}

}  // namespace bposd
