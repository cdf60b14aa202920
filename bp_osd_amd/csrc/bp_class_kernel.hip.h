// bp_class_kernel.hip.h -- BP with all messages in LDS for codes whose checks and bits fall into a few degree classes:
// every hypergraph product of regular seeds, i.e. all three example codes the reference ships ([[400,16,6]],
// [[625,25,8]], [[900,36,10]]: check degree 7, bit degrees 3 and 4; /root/reference/examples/qldpc_decode_example.py:5-23),
// toric codes (4; 2), and products with open boundaries such as the surface codes (check degrees 3..4, bit degrees 1..2;
// the reference's README example is the distance-3 one).  Rows a3-a7 of SURVEY.md §8; the scheme (persistent workgroups
// on an atomic queue, in-place check-major messages  msg[k * MP + c], incremental mismatch bitmap, speculative check
// pass, two barriers per iteration) is bp_kernel.hip.h's, with what the local-edge kernel taught:
//
//   * NO per-lane degree predicates.  The host sorts checks by degree into waves and bits by degree into 64-lane
//     groups; a wave's checks have ONE degree (DCLO .. DC) and group (slot r, wave w) of bits has ONE degree
//     (DVLO .. DVHI); the kernel switches on those wave-uniform numbers into straight-line code (0 = nothing here:
//     skipped).  A padding lane of a group points all its
//     edges at a private dummy slot that starts at a positive prior: its messages stay positive for ever (sums of
//     positive numbers), its decision never flips, nobody else reads the slot.  Padding check lanes (positions without
//     a check) compute on their own never-referenced slots.  Checks sit at host-chosen positions (pos_chk).
//   * sign on the multiplier (one v_cndmask of alpha's high word per edge, `mag * (+-alpha)` is the one multiply),
//     alpha = 1 - 2^-it formed by scalar shifts, the suffix recurrence without its identity additions (see bit_update_deg),
//     LDS byte addresses in one register per edge (hot path and decision-flip path), the uniform prior in a scalar pair,
//     one broadcast read per wave for the mismatch test, cold kernel arguments read where they are used;
//   * threads per workgroup = 64 * waves actually needed (not a power of two), with the LDS stride MP a compile-time
//     power of two >= m, so offsets stay instruction immediates.
//
// Arithmetic and association order are the reference's, fp64, -ffp-contract=off: the LLR bits
// equal the CPU restatement's (tests/test_gpu_parity.py compares them).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bp_kernel.hip.h"

namespace bposd {

struct BpClassParams {
    int m, n;
    long long B;
    int max_iter;
    double ms_scaling;
    double ps_clip;
    int osd_enabled;
    const uint8_t* __restrict__ synd;     // [B, m]
    const double* __restrict__ llr0;      // [n]
    const uint8_t* __restrict__ sel;      // [B, n] nullable
    const double* __restrict__ llr0_alt;  // [n]
    const int* __restrict__ pos_chk;      // [CPT * NTMAX]         entry j * NTMAX + tid: check at that position, -1 = none (the host
                                          //                       orders checks for few bank conflicts in the bit pass)
    const int* __restrict__ pos_bit;      // [VPT * NTMAX]         entry r * NTMAX + tid: bit in slot r of thread tid, -1 = padding
    const int* __restrict__ bit_slot;     // [DVHI * VPT * NTMAX]  entry (d * VPT + r) * NTMAX + tid: LDS slot of the d-th edge
                                          //                       (ascending check index); the thread's dummy slot where there is none
    const int* __restrict__ grp_deg;      // [VPT * NTMAX / 64]    entry r * (NTMAX / 64) + wave: degree of that 64-lane group, 0 = empty
    const int* __restrict__ grp_cdeg;     // [CPT * NTMAX / 64]    entry j * (NTMAX / 64) + wave: degree of the checks at that wave's
                                          //                       positions, 0 = no checks there
    uint8_t* __restrict__ out_bp;
    uint8_t* __restrict__ out_osd0;
    uint8_t* __restrict__ out_osdw;
    uint8_t* __restrict__ out_conv;
    int* __restrict__ out_iters;
    double* __restrict__ out_llr;
    double* __restrict__ llr_ws;
    double* __restrict__ llr_tmp;         // [gridDim.x][n] LLRs of the syndrome a workgroup is on, written only when they will
                                          // be read: in the last iteration, or every iteration if out_llr is set
    int* __restrict__ osd_list;
    int* __restrict__ counters;
    unsigned long long* __restrict__ iter_total;
    int* __restrict__ tail_flag;  // nullable, host-visible: set to 1 by the workgroup that finds the queue empty (the tail begins)
    int packed_io;  // 1: packed syndromes in, packed result rows out (bp_kernel.hip.h: BpParams::packed_io)
    int queue_batch, queue_shift; // a workgroup takes  clamp((syndromes left) >> queue_shift, 1, queue_batch)  syndromes from the queue per
                                  // atomic -- big batches first, single syndromes at the end (2^queue_shift ~ 2 x the grid).  With codes
                                  // of a few hundred bits the chip retires a syndrome every ~25 ns, which is what one same-address
                                  // atomic per syndrome costs.
};

__host__ __device__ inline size_t bp_class_lds_bytes(int DC, int mp, int ntmax) {
    // messages + one dummy slot per thread + mismatch bitmap + control words
    return ((size_t)DC * mp + ntmax + 2) * 8 + (size_t)(mp / 32 + 2) * 4 + 8 * 4;
}

// CONTRACT: bp_class_kernel takes exactly ONE explicit argument, the BpClassParams struct BY VALUE -- it then sits at offset 0 of the
// kernarg segment and bpc_args() may read it there.  A second kernel parameter, or the struct passed by pointer, would make these
// reads return garbage without a diagnostic; a -DBPOSD_DEBUG build traps on the first workgroup if the two views disagree.
static_assert(__is_trivially_copyable(BpClassParams) && alignof(BpClassParams) <= 8, "BpClassParams is copied into the kernarg segment as it is");
typedef const __attribute__((address_space(4))) BpClassParams* bpc_args_ptr;
__device__ __forceinline__ bpc_args_ptr bpc_args() {  // cold arguments: read from the kernarg segment at the point of use
    bpc_args_ptr a = (bpc_args_ptr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(a));
    return a;
}
__device__ __forceinline__ int bpc_table_load(const int* base, unsigned int lane_off, unsigned int const_off) {
    asm volatile("" : "+v"(lane_off));  // address formed here, not hoisted into a 64-bit register pair per table row
    return *(const int*)((const char*)base + (lane_off + const_off));
}

// Bit update for a bit of degree D: posterior, and the D outgoing messages  out[d] = prefix(d) + suffix(d)  with
//   prefix(d) = ((l0 + c[0]) + ...) + c[d-1]   (from the top of the column, prior included)
//   suffix(d) = ((0.0 + c[D-1]) + ...) + c[d+1] (from the bottom)
// as the reference accumulates them.  The additions with the literal 0.0 are not executed:  x + 0.0 == x  unless x is
// -0.0, and 0.0 + c == c unless c is -0.0, in which case the two suffixes differ only in the sign of a zero that is then
// added to a prefix -- and a prefix is never -0.0 (a sum is -0.0 only if both operands are; the prior log((1-p)/p) never
// is), so  prefix + (+0.0) == prefix + (-0.0)  bit for bit.  tests/test_gpu_parity.py compares LLR bits.
template <int D>
__device__ __forceinline__ void bit_update_deg(double l0, const double* c, double& llr, double* out) {
    double pre[D + 1];
    pre[0] = l0;
#pragma unroll
    for (int d = 0; d < D; ++d) pre[d + 1] = pre[d] + c[d];
    llr = pre[D];
    out[D - 1] = pre[D - 1];
    if (D >= 2) {
        double suf = c[D - 1];
        out[D - 2] = pre[D - 2] + suf;
#pragma unroll
        for (int d = D - 3; d >= 0; --d) {
            suf = suf + c[d + 1];
            out[d] = pre[d] + suf;
        }
    }
}

// The bit pass of one owned bit with D edges (D compile-time): messages in, posterior, messages out, decision test.
template <int D, int MP, class KeepLlr>
__device__ __forceinline__ void bit_pass_deg(const unsigned int* eaddr, double l0, int r, unsigned int& decmask, unsigned int msg_base,
                                             unsigned int* diffw, bool keep_llr, KeepLlr&& store_llr) {
    double c[D], out[D], t;
#pragma unroll
    for (int d = 0; d < D; ++d) c[d] = *((msg_ptr)(uintptr_t)eaddr[d]);
    bit_update_deg<D>(l0, c, t, out);
#pragma unroll
    for (int d = 0; d < D; ++d) *((msg_ptr)(uintptr_t)eaddr[d]) = out[d];
    if (keep_llr) store_llr(t);
    const unsigned int dnew = (t <= 0.0) ? 1u : 0u;
    if (dnew != ((decmask >> r) & 1u)) {  // (a padding bit never gets here: its messages stay positive)
        decmask ^= 1u << r;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            unsigned int pa = eaddr[d];
            asm volatile("" : "+v"(pa));  // keep the rare path's address arithmetic in the branch
            const int c2 = (int)((pa - msg_base) >> 3) & (MP - 1);
            atomicXor(&diffw[c2 >> 5], 1u << (c2 & 31));
        }
    }
}
// run-time (wave-uniform) degree -> compile-time arm
template <int DVLO, int DVHI, int MP, class KeepLlr>
__device__ __forceinline__ void bit_pass_arm(int D, const unsigned int* eaddr, double l0, int r, unsigned int& decmask, unsigned int msg_base,
                                             unsigned int* diffw, bool keep_llr, KeepLlr&& store_llr) {
    if constexpr (DVLO >= DVHI) {
        bit_pass_deg<DVHI, MP>(eaddr, l0, r, decmask, msg_base, diffw, keep_llr, store_llr);
    } else {
        if (D == DVLO) bit_pass_deg<DVLO, MP>(eaddr, l0, r, decmask, msg_base, diffw, keep_llr, store_llr);
        else bit_pass_arm<DVLO + 1, DVHI, MP>(D, eaddr, l0, r, decmask, msg_base, diffw, keep_llr, store_llr);
    }
}

// The check pass of one check with D edges (D compile-time) at LDS position mc: a4 (product-sum) / a5 (min-sum).
template <int D, int MP, int METHOD>
__device__ __forceinline__ void check_pass_deg(msg_ptr mc, bool sbit, int alpha_lo, int alpha_hi, int nalpha_hi, double ps_clip) {
    double v[D];
#pragma unroll
    for (int k = 0; k < D; ++k) v[k] = mc[k * MP];
    if (METHOD == 1) {
        double pre[D], suf[D];
        pre[0] = __DBL_MAX__;
#pragma unroll
        for (int k = 1; k < D; ++k) pre[k] = min_abs(pre[k - 1], v[k - 1]);
        suf[D - 1] = __DBL_MAX__;
#pragma unroll
        for (int k = D - 2; k >= 0; --k) suf[k] = min_abs(suf[k + 1], v[k + 1]);
        bool neg[D];
        bool par = sbit;
#pragma unroll
        for (int k = 0; k < D; ++k) {
            neg[k] = (v[k] <= 0.0);  // a zero counts as negative, as in the reference
            par ^= neg[k];
        }
#pragma unroll
        for (int k = 0; k < D; ++k) {
            const double mag = (k == 0) ? suf[0] : (k == D - 1 ? pre[D - 1] : min_pos(pre[k], suf[k]));
            const double sa = __hiloint2double((par ^ neg[k]) ? nalpha_hi : alpha_hi, alpha_lo);
            mc[k * MP] = mag * sa;
        }
    } else {
        double pre[D], th[D];
        double t = 1.0;
#pragma unroll
        for (int k = 0; k < D; ++k) {
            pre[k] = t;
            th[k] = pm_ps_tanh_half(v[k], METHOD == 0);
            t *= th[k];
        }
        t = 1.0;
        const double sg = sbit ? -1.0 : 1.0;
#pragma unroll
        for (int k = D - 1; k >= 0; --k) {
            const double x = pre[k] * t;
            double o = sg * pm_ps_log_ratio(x, METHOD == 0);
            if (ps_clip > 0.0) {  // uniform; the comparisons are false for NaN, as on the CPU
                if (o > ps_clip) o = ps_clip;
                if (o < -ps_clip) o = -ps_clip;
            }
            mc[k * MP] = o;
            t *= th[k];
        }
    }
}
// run-time (wave-uniform) check degree -> compile-time arm
template <int DLO, int DHI, int MP, int METHOD>
__device__ __forceinline__ void check_pass_arm(int D, msg_ptr mc, bool sbit, int alpha_lo, int alpha_hi, int nalpha_hi, double ps_clip) {
    if constexpr (DLO >= DHI) {
        check_pass_deg<DHI, MP, METHOD>(mc, sbit, alpha_lo, alpha_hi, nalpha_hi, ps_clip);
    } else {
        if (D == DLO) check_pass_deg<DLO, MP, METHOD>(mc, sbit, alpha_lo, alpha_hi, nalpha_hi, ps_clip);
        else check_pass_arm<DLO + 1, DHI, MP, METHOD>(D, mc, sbit, alpha_lo, alpha_hi, nalpha_hi, ps_clip);
    }
}

// DCLO .. DC: check degrees that occur (DC also sizes the message array);  DVLO .. DVHI: bit degrees that occur;  CPT / VPT: check / bit slots per thread;
// MPT: LDS stride (power of two >= m);  NTMAX: table stride = largest workgroup this instantiation is launched with;
// METHOD: 0 product-sum (two divisions per edge), 2 product-sum in the reference's operation order (portable_math.h), 1 min-sum;  UPRIOR: uniform channel and no per-shot channel (prior in a scalar pair)
template <int DCLO, int DC, int DVLO, int DVHI, int CPT, int VPT, int MPT, int NTMAX, int MINW, int METHOD, bool UPRIOR>
__global__ __launch_bounds__(NTMAX, MINW) void bp_class_kernel(const BpClassParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int m = P.m, n = P.n;
#ifdef BPOSD_DEBUG
    if (blockIdx.x == 0 && threadIdx.x == 0 && (bpc_args()->m != P.m || bpc_args()->counters != P.counters)) __builtin_trap();
#endif
    constexpr int MP = MPT;
    constexpr int NW = NTMAX / 64;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NT = (int)blockDim.x;  // multiple of 64

    typedef __attribute__((address_space(3))) unsigned char* lds_bytes;
    const unsigned int msg_base = (unsigned int)(uintptr_t)(lds_bytes)smem;
    double* msg_plain = reinterpret_cast<double*>(smem);
    unsigned int* diffw = reinterpret_cast<unsigned int*>(msg_plain + (size_t)DC * MP + NTMAX + 2);
    int* sh = reinterpret_cast<int*>(diffw + (MP / 32 + 2));
    const unsigned int diffw_base = (unsigned int)(uintptr_t)(lds_bytes)diffw;
#define BPC_AT(a) ((msg_ptr)(uintptr_t)(a))
#define BPC_BIT(r) bpc_table_load(bpc_args()->pos_bit, (unsigned int)tid * 4u, (unsigned int)((r) * NTMAX * 4))

    // ---- per-thread tables: byte address of every edge of my bits, degree of my groups (wave-uniform)
    unsigned int eaddr[VPT][DVHI];
    int gdeg[VPT];
    double l0[UPRIOR ? 1 : VPT];
    if (UPRIOR) {
        const double v = bpc_args()->llr0[0];
        l0[0] = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
    }
#define BPC_L0(r) l0[UPRIOR ? 0 : (r)]
#pragma unroll
    for (int r = 0; r < VPT; ++r) {
        gdeg[r] = __builtin_amdgcn_readfirstlane(bpc_args()->grp_deg[r * NW + wave]);
        if (!UPRIOR) {
            const int bit = BPC_BIT(r);
            l0[r] = bit >= 0 ? bpc_args()->llr0[bit] : 1.0;  // (UPRIOR: the host has checked that the prior is > 0)
        }
#pragma unroll
        for (int d = 0; d < DVHI; ++d) eaddr[r][d] = msg_base + 8u * (unsigned int)bpc_args()->bit_slot[(d * VPT + r) * NTMAX + tid];
    }
    // wave-uniform: degree of the checks at this wave's positions of group j (0 = none)
    int cdeg[CPT];
#pragma unroll
    for (int j = 0; j < CPT; ++j) cdeg[j] = __builtin_amdgcn_readfirstlane(bpc_args()->grp_cdeg[j * NW + wave]);
    const int want_llr_s = __builtin_amdgcn_readfirstlane(bpc_args()->out_llr != nullptr ? 1 : 0);
    auto want_llr = [&]() -> bool {
        int w = __builtin_amdgcn_readfirstlane(want_llr_s);
        asm volatile("" : "+s"(w));
        return w != 0;
    };

    unsigned long long it_acc = 0ull;  // iterations this workgroup ran (uniform), added to iter_total once at the end
    long long s_seen = 0;  // queue position at my previous fetch (uniform): estimates what is left
    if (tid == 0) {
        sh[0] = 0;
        sh[1] = 0;
    }
    for (;;) {
        int qb = (int)((P.B - s_seen) >> bpc_args()->queue_shift);
        qb = qb < 1 ? 1 : (qb > bpc_args()->queue_batch ? bpc_args()->queue_batch : qb);
        qb = __builtin_amdgcn_readfirstlane(qb);
        if (tid == 0) sh[2] = atomicAdd(&bpc_args()->counters[0], qb);
        __syncthreads();
        const long long s0 = __builtin_amdgcn_readfirstlane(sh[2]);
        // the chunk loop of the host-pointer API launches the next chunk's kernels when this one's tail begins: the flag
        // is set by whoever takes the last syndrome or first finds the queue empty (setting it twice is harmless)
        if (s0 <= P.B && s0 + qb >= P.B && tid == 0 && bpc_args()->tail_flag) *(volatile int*)bpc_args()->tail_flag = 1;
        if (s0 >= P.B) break;
        s_seen = s0;
        const long long s1 = (s0 + qb < P.B) ? s0 + qb : P.B;
#pragma clang loop unroll(disable)
        for (long long s = s0; s < s1; ++s) {

        // ---- syndrome bits of my checks; the mismatch bitmap (indexed by check) starts as the syndrome
        bool sbit[CPT];
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
            const int c = bpc_table_load(bpc_args()->pos_chk, (unsigned int)tid * 4u, (unsigned int)(j * NTMAX * 4));
            sbit[j] = (c >= 0) ? bp_synd_bit(bpc_args()->synd, bpc_args()->packed_io, s, m, c) : false;
            const unsigned long long bal = __ballot(sbit[j]);
            if (lane == 0 && ((wave << 6) + j * NT) < MP) {
                const int w0 = ((wave << 6) + j * NT) >> 5;
                diffw[w0] = (unsigned int)bal;
                diffw[w0 + 1] = (unsigned int)(bal >> 32);
                if (bal) sh[0] = 1;
            }
        }
        if (!UPRIOR && bpc_args()->sel) {
#pragma unroll
            for (int r = 0; r < VPT; ++r) {
                const int i = BPC_BIT(r);
                if (i >= 0) l0[r] = bpc_args()->sel[(size_t)s * n + i] ? bpc_args()->llr0_alt[i] : bpc_args()->llr0[i];
            }
        }
        // ---- a3: every edge's bit->check message starts at the prior
#define BPC_LLRT (bpc_args()->llr_tmp + (size_t)blockIdx.x * n)
        unsigned int decmask = 0u;
#pragma unroll
        for (int r = 0; r < VPT; ++r) {
            double lp = BPC_L0(r);
            if (UPRIOR) asm volatile("" : "+s"(lp));
            if (want_llr()) {  // a syndrome that needs no iteration reports the priors
                const int i = BPC_BIT(r);
                if (i >= 0) BPC_LLRT[i] = lp;
            }
#pragma unroll
            for (int d = 0; d < DVHI; ++d)
                if (d < DVLO || d < gdeg[r]) *BPC_AT(eaddr[r][d]) = lp;
        }
        __syncthreads();

        int it_done = 0;
        bool conv = (sh[0] == 0);
        if (!conv) {
#pragma clang loop unroll(disable)
            for (int it = 1;; ++it) {
                const int fi = it & 1;
                {
                    unsigned long long mis = 0ull;
#pragma unroll
                    for (int j = 0; j < CPT; ++j)
                        if (cdeg[j] != 0)
                            mis |= *(const volatile __attribute__((address_space(3))) unsigned long long*)(uintptr_t)(
                                diffw_base + (unsigned int)(((wave << 6) + j * NT) >> 3));
                    if (lane == 0 && mis) sh[fi] = 1;
                }
                if (it > P.max_iter) {
                    __syncthreads();
                    conv = (sh[fi] == 0);
                    it_done = P.max_iter;
                    break;
                }
                // =================== check -> bit pass (a4 / a5), speculative for it >= 2 ===========
                const unsigned long long alpha_u = alpha_bits_for_iteration(P.ms_scaling, it);
                const int alpha_lo = (int)(unsigned int)alpha_u, alpha_hi = (int)(unsigned int)(alpha_u >> 32), nalpha_hi = alpha_hi ^ (int)0x80000000;
#pragma unroll
                for (int j = 0; j < CPT; ++j) {
                    if (cdeg[j] == 0) continue;  // wave-uniform
                    check_pass_arm<DCLO, DC, MP, METHOD>(cdeg[j], BPC_AT(msg_base + 8u * (unsigned int)(tid + j * NT)), sbit[j], alpha_lo, alpha_hi,
                                                         nalpha_hi, P.ps_clip);
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
                // one self-contained arm per (slot, degree): loads, sums, stores and the decision test of a D-edge bit with D a
                // compile-time constant; the wave-uniform degree of the group picks the arm (no partially defined arrays: those
                // became loop-carried registers, 24 of them in a first version)
#pragma unroll
                for (int r = 0; r < VPT; ++r) {
                    if (gdeg[r] == 0) continue;  // wave-uniform: no bits in this group
                    bool hit = false;
#pragma unroll
                    for (int D = DVLO; D <= DVHI; ++D) {
                        if (!hit && (D == DVHI || gdeg[r] == D)) {
                            hit = true;
                            bit_pass_arm<DVLO, DVHI, MP>(D, eaddr[r], BPC_L0(r), r, decmask, msg_base, diffw, keep_llr,
                                                         [&](double t) {
                                                             const int bi = BPC_BIT(r);
                                                             if (bi >= 0) BPC_LLRT[bi] = t;
                                                         });
                        }
                    }
                }
                __syncthreads();
            }
        }

        // ---- results
        const bool to_osd = (!conv) && bpc_args()->osd_enabled;
        if (tid == 0) {
            if (to_osd) {
                const int slot = atomicAdd(&bpc_args()->counters[1], 1);
                bpc_args()->osd_list[slot] = (int)s;
                sh[3] = slot;
            }
            if (bpc_args()->out_conv) bpc_args()->out_conv[s] = conv ? 1 : 0;
            if (bpc_args()->out_iters) bpc_args()->out_iters[s] = it_done;
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
        const int packed = __builtin_amdgcn_readfirstlane(bpc_args()->packed_io);
        if (packed)  // result rows as 64-bit words through an LDS bitmap (the messages are dead)
            bp_store_packed_rows<VPT>((unsigned int*)smem, tid, (int)blockDim.x, n, s, to_osd, (unsigned long long*)bpc_args()->out_bp,
                                      (unsigned long long*)bpc_args()->out_osd0, (unsigned long long*)bpc_args()->out_osdw,
                                      [&](int r) { return BPC_BIT(r); }, [&](int r) { return ((decmask >> r) & 1u) != 0u; });
#pragma unroll
        for (int r = 0; r < VPT; ++r) {
            const int i = BPC_BIT(r);
            if (i >= 0) {
                const size_t o = (size_t)s * n + i;
                const uint8_t b = (uint8_t)((decmask >> r) & 1u);
                if (!packed) {
                    if (bpc_args()->out_bp) bpc_args()->out_bp[o] = b;
                    if (!to_osd) {
                        bpc_args()->out_osdw[o] = b;
                        if (bpc_args()->out_osd0) bpc_args()->out_osd0[o] = b;
                    }
                }
                if (to_osd) bpc_args()->llr_ws[(size_t)slot * n + i] = BPC_LLRT[i];
                if (want_llr()) bpc_args()->out_llr[o] = BPC_LLRT[i];
            }
        }
        __syncthreads();
        }  // syndromes of this batch
    }
    if (tid == 0 && it_acc) atomicAdd(bpc_args()->iter_total, it_acc);
#undef BPC_AT
#undef BPC_BIT
#undef BPC_L0
#undef BPC_LLRT
}

}  // namespace bposd
