// bp_local_kernel.hip.h -- min-sum BP for (3,6)-regular codes with one third of the messages kept in registers.
//
// bp_kernel.hip.h moves every message through LDS twice per iteration (check -> bit, bit -> check) and is
// bound by LDS bandwidth (ds_write_b64 6.1 cycles, ds_read_b64 2.95 cycles per wave -- tools/microbench).
// For a code with check degree 6, bit degree 3 and n = 2m the host finds an assignment "every check OWNS two
// of its six neighbouring bits, every bit is owned by exactly one check" (a perfect b-matching; it exists for
// every (3,6)-biregular graph).  The thread that runs a check also runs its two owned bits, so the message on
// the edge (check, owned bit) never leaves that thread's registers: 2 of a check's 6 edges and 1 of a bit's 3
// edges are LOCAL; LDS carries 4m instead of 6m messages (32 instead of 48 bytes per check and direction).
//
// Exactness.  The min-sum check update is order-independent (exact minima of the other five magnitudes, sign
// parity), so a check may list its two local edges last.  The bit update is NOT (fp64 sums: prefix from the
// top of the column including the prior, suffix from the bottom -- SURVEY.md Appendix A.3), so the local
// message has to enter the sums at the position dl in {0,1,2} its check has among the bit's three checks in
// ascending order.  The host therefore sorts checks into 64-position groups (a wave = one group per owned
// check) whose slot-b bits share one dl wherever it can; the kernel switches on the wave-uniform code
// (0, 1, 2: straight-line code, instruction for instruction the arithmetic of bp_kernel; 3: a mixed group, the
// operands are routed by per-lane selects).  The position count MP is a compile-time power of two, as in
// bp_kernel, so LDS offsets are instruction immediates.
//
// Everything else -- persistent workgroups on an atomic queue, in-place messages, incremental convergence
// bitmap with a speculative check pass, outputs, OSD hand-off -- is the scheme of bp_kernel.hip.h (rows a3-a7).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bp_kernel.hip.h"

#ifndef BPOSD_BPL_SIGN_ON_MAG
#define BPOSD_BPL_SIGN_ON_MAG 0
#endif
namespace bposd {

struct BpLocalParams {
    int m, n;
    long long B;
    int max_iter;
    double ms_scaling;
    int osd_enabled;
    int mp;                               // positions = blockDim.x * CPT (multiple of 64)
    const uint8_t* __restrict__ synd;     // [B, m]
    const double* __restrict__ llr0;      // [n]
    const uint8_t* __restrict__ sel;      // [B, n] nullable
    const double* __restrict__ llr0_alt;  // [n]
    const int* __restrict__ pos_chk;      // [mp]      check at position p, -1 = padding
    const int* __restrict__ pos_bit;      // [2 * mp]  entry b*mp + p: b-th owned bit of position p, -1 = padding
    const int* __restrict__ pos_alo;      // [2 * mp]  LDS slot (k * mp + p') of the bit's edge to the check after its owner (cyclically)
    const int* __restrict__ pos_ahi;      // [2 * mp]  ... to the check after that one
    const int* __restrict__ pos_dl;       // [2 * mp]  position dl of the owner among the bit's three checks
    const int* __restrict__ grp_dl;       // [2 * mp / 64]  entry b*(mp/64) + g: dl if the whole group shares it, else 3
    uint8_t* __restrict__ out_bp;
    uint8_t* __restrict__ out_osd0;
    uint8_t* __restrict__ out_osdw;
    uint8_t* __restrict__ out_conv;
    int* __restrict__ out_iters;
    double* __restrict__ out_llr;
    double* __restrict__ llr_ws;
    double* __restrict__ llr_tmp;          // [gridDim.x][n] LLRs of the syndrome a workgroup is on, written only when they
                                           // will be read: in the last iteration, or every iteration if out_llr is set
    int* __restrict__ osd_list;
    int* __restrict__ counters;
    unsigned long long* __restrict__ iter_total;
    int* __restrict__ tail_flag;  // nullable, host-visible: set to 1 by the workgroup that finds the queue empty (the tail begins)
    int packed_io;  // 1: packed syndromes in, packed result rows out (bp_kernel.hip.h: BpParams::packed_io)
};

__host__ __device__ inline size_t bp_local_lds_bytes(int mp) {
    // 4 LDS edges per position (+ dummy slot) + mismatch bitmap + control words
    return ((size_t)4 * mp + 2) * 8 + (size_t)(mp / 32 + 2) * 4 + 8 * 4;
}

// A bit's three checks in ascending index order have ranks 0, 1, 2; its owner has rank DL.  X is the message of the
// check that follows the owner cyclically (rank DL+1 mod 3), Y the one after (rank DL+2 mod 3): the roles are tied to
// the owner, not to the index order, which keeps the LDS access pattern of structured codes regular across the
// wrap-around of their circulants.  The sums run in rank order.
// The reference starts the suffix sums at 0.0 ("temp = 0; b2c += temp; temp += c2b", bottom edge first).  Two of its
// additions are identities and are not executed here: pre2 + 0.0 (== pre2 unless pre2 is -0.0) and 0.0 + c2 (== c2
// unless c2 is -0.0, in which case the +0.0 it would give differs from c2 only in the sign of a zero that is then added
// to pre1 / to c1 + pre0).  A prefix is -0.0 only if the prior is, and a prior log((1 - p) / p) never is (log(1) = +0.0);
// with non-negative-zero prefixes x + (+0.0) == x + (-0.0) bit for bit.  tests/test_gpu_parity.py compares LLR bits.
template <int DL>
__device__ __forceinline__ void bit_update(double l0, double R, double X, double Y, double& llr, double& oR, double& oX,
                                           double& oY) {
    const double c0 = DL == 0 ? R : (DL == 1 ? Y : X);
    const double c1 = DL == 0 ? X : (DL == 1 ? R : Y);
    const double c2 = DL == 0 ? Y : (DL == 1 ? X : R);
    const double pre0 = l0;  // prefix from the top of the column (prior included)
    const double pre1 = pre0 + c0;
    const double pre2 = pre1 + c1;
    llr = pre2 + c2;
    const double o2 = pre2;            // + suffix 0.0
    const double o1 = pre1 + c2;       // + suffix (0.0 + c2)
    const double o0 = pre0 + (c2 + c1);
    oR = DL == 0 ? o0 : (DL == 1 ? o1 : o2);
    oX = DL == 0 ? o1 : (DL == 1 ? o2 : o0);
    oY = DL == 0 ? o2 : (DL == 1 ? o0 : o1);
}

// mixed group: the same sums with the operands routed per lane
__device__ __forceinline__ void bit_update_mixed(int dlv, double l0, double R, double X, double Y, double& llr, double& oR,
                                                 double& oX, double& oY) {
    const bool d0 = dlv == 0, d1 = dlv == 1;
    const double c0 = d0 ? R : (d1 ? Y : X);
    const double c1 = d0 ? X : (d1 ? R : Y);
    const double c2 = d0 ? Y : (d1 ? X : R);
    const double pre0 = l0;
    const double pre1 = pre0 + c0;
    const double pre2 = pre1 + c1;
    llr = pre2 + c2;
    const double o2 = pre2;
    const double o1 = pre1 + c2;
    const double o0 = pre0 + (c2 + c1);
    oR = d0 ? o0 : (d1 ? o1 : o2);
    oX = d0 ? o1 : (d1 ? o2 : o0);
    oY = d0 ? o2 : (d1 ? o0 : o1);
}

// table entry at a uniform base + per-lane byte offset; the offset is made opaque at the point of use so that the
// address is formed there (hoisted out of the syndrome loop it would occupy a 64-bit register pair per table row)
__device__ __forceinline__ int bpl_table_load(const int* base, unsigned int lane_off, unsigned int const_off) {
    asm volatile("" : "+v"(lane_off));
    return *(const int*)((const char*)base + (lane_off + const_off));
}

// Kernel arguments that are read once per syndrome or less often (tables, output pointers, the queue) are fetched from the
// kernarg segment where they are used -- s_load from the scalar cache -- instead of being held in ~50 SGPRs for the
// lifetime of the kernel: the 64-VGPR build had run out of SGPRs and of lanes in its SGPR-spill register.
// CONTRACT: bp_local_kernel takes exactly ONE explicit argument, the BpLocalParams struct BY VALUE -- it then sits at offset 0 of the
// kernarg segment and bpl_args() may read it there.  A second kernel parameter, or the struct passed by pointer, would make these
// reads return garbage without a diagnostic; a -DBPOSD_DEBUG build traps on the first workgroup if the two views disagree.
static_assert(__is_trivially_copyable(BpLocalParams) && alignof(BpLocalParams) <= 8, "BpLocalParams is copied into the kernarg segment as it is");
typedef const __attribute__((address_space(4))) BpLocalParams* bpl_args_ptr;
__device__ __forceinline__ bpl_args_ptr bpl_args() {
    bpl_args_ptr a = (bpl_args_ptr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(a));
    return a;
}

// CPT: checks per thread (each with its two owned bits); MPT: positions (power of two) = blockDim.x * CPT;
// EARLY: the check pass requests the LDS messages of all its checks before it computes the first one (the LDS accesses
// are volatile, i.e. issued in program order: without this the read latency is exposed once per check)
// UPRIOR: every bit has the same prior (uniform channel, no per-shot channel): it lives in a scalar register pair
// PACKED: the packed-I/O form (BpLocalParams::packed_io) as a compile-time switch: the byte form is then instruction for
// instruction what it was before the packed form existed (as a run-time switch it cost the headline launch 1.2 %, same-box A/B)
template <int CPT, int MPT, int MINW, bool EARLY, bool UPRIOR, bool PACKED = false>
__global__ __launch_bounds__(MPT / CPT, MINW) void bp_local_kernel(const BpLocalParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int m = P.m, n = P.n;
#ifdef BPOSD_DEBUG
    if (blockIdx.x == 0 && threadIdx.x == 0 && (bpl_args()->m != P.m || bpl_args()->counters != P.counters)) __builtin_trap();
#endif
    constexpr int NT = MPT / CPT;
    constexpr int MP = MPT;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int NB = 2 * CPT;  // owned bits per thread

    double* msg_plain = reinterpret_cast<double*>(smem);
    msg_ptr msg = (msg_ptr)msg_plain;
    unsigned int* diffw = reinterpret_cast<unsigned int*>(msg_plain + (size_t)4 * MP + 2);
    int* sh = reinterpret_cast<int*>(diffw + (MP / 32 + 2));
    const unsigned int diffw_base = (unsigned int)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)diffw;

    // ---- per-thread graph tables
    // Padding positions (no check, no bits) need no predicate anywhere in the iteration loop: the host wires the two
    // "bits" of a padding position to that position's own four LDS slots, i.e. a closed toy graph of one check and two
    // bits, three edges each, with a zero syndrome bit and a positive prior.  All its messages stay positive for ever
    // (they grow; sums of positive numbers never produce a NaN), its bits never change their decision and its check
    // never mismatches.  Only the stores that leave the workgroup (LLRs, results) test pos_bit >= 0.
    // alo / ahi: LDS BYTE addresses of the two non-local edges of an owned bit (one register each for the hot path AND
    // the rare decision-flip path, which recovers the position from the address; a slot index kept next to its byte
    // address cost the default build 84 B/lane of scratch and a scratch reload on the loop's critical path)
    typedef __attribute__((address_space(3))) unsigned char* lds_bytes;
    const unsigned int msg_base = (unsigned int)(uintptr_t)(lds_bytes)smem;
    unsigned int alo[NB], ahi[NB];
    int dl[NB];
    unsigned int dlpack = 0u;  // 2 bits per owned bit: its dl (needed per lane only in mixed groups)
    double l0[UPRIOR ? 1 : NB];
    if (UPRIOR) {  // into a scalar register pair
        const double v = bpl_args()->llr0[0];
        l0[0] = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
    }
#define BPL_L0(r) l0[UPRIOR ? 0 : (r)]
#define BPL_AT(a) ((msg_ptr)(uintptr_t)(a))
    // r-th owned bit of this thread (-1 = padding): a uniform base plus a 32-bit byte offset per lane, so the loads outside
    // the iteration loop need no 64-bit per-lane pointers (which the 64-VGPR build kept in scratch)
#define BPL_BIT(r) bpl_table_load(bpl_args()->pos_bit, (unsigned int)tid * 4u, (unsigned int)((((r) & 1) * MP + ((r) >> 1) * NT) * 4))
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
        const int p = tid + j * NT;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int r = 2 * j + b;
            if (!UPRIOR) {
                const int bit = BPL_BIT(r);
                l0[r] = bit >= 0 ? bpl_args()->llr0[bit] : 1.0;  // (UPRIOR: the host has checked that the prior is > 0)
            }
            alo[r] = msg_base + 8u * (unsigned int)bpl_args()->pos_alo[b * MP + p];
            ahi[r] = msg_base + 8u * (unsigned int)bpl_args()->pos_ahi[b * MP + p];
            dl[r] = __builtin_amdgcn_readfirstlane(bpl_args()->grp_dl[b * (MP >> 6) + (p >> 6)]);  // uniform per wave
            dlpack |= (unsigned int)bpl_args()->pos_dl[b * MP + p] << (2 * r);
        }
    }

    const int want_llr_s = __builtin_amdgcn_readfirstlane(bpl_args()->out_llr != nullptr ? 1 : 0);
    // tested from ONE scalar register wherever it is needed (as a loop-invariant condition it became a lane mask plus a
    // vector register that went to scratch)
    auto want_llr = [&]() -> bool {
        int w = want_llr_s;
        asm volatile("" : "+s"(w));
        return w != 0;
    };
    for (;;) {
        if (tid == 0) {
            int zero = 0;
            asm volatile("" : "+v"(zero));  // (formed here: as a loop-invariant register pair it went to scratch)
            sh[0] = zero;
            sh[1] = zero;
            sh[2] = atomicAdd(&bpl_args()->counters[0], 1);
        }
        __syncthreads();
        const long long s = __builtin_amdgcn_readfirstlane(sh[2]);
        if (s >= P.B) {
            // the chunk loop of the host-pointer API launches the next chunk's kernels when this one's tail begins
            if (s == P.B && tid == 0 && bpl_args()->tail_flag) *(volatile int*)bpl_args()->tail_flag = 1;
            break;
        }

        // ---- syndrome bits of my checks; the mismatch bitmap (indexed by position) starts as the syndrome
        bool sbit[CPT];
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
            const int c = bpl_table_load(bpl_args()->pos_chk, (unsigned int)tid * 4u, (unsigned int)(j * NT * 4));
            sbit[j] = (c >= 0) ? (PACKED ? bp_synd_bit(bpl_args()->synd, 1, s, m, c) : ((bpl_args()->synd[(size_t)s * m + c] & 1) != 0)) : false;
            const unsigned long long bal = __ballot(sbit[j]);
            if (lane == 0) {
                const int w0 = ((wave << 6) + j * NT) >> 5;
                diffw[w0] = (unsigned int)bal;
                diffw[w0 + 1] = (unsigned int)(bal >> 32);
                if (bal) sh[0] = 1;
            }
        }
        if (!UPRIOR && bpl_args()->sel) {
#pragma unroll
            for (int j = 0; j < CPT; ++j)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const int i = BPL_BIT(2 * j + b);
                    if (i >= 0) l0[2 * j + b] = bpl_args()->sel[(size_t)s * n + i] ? bpl_args()->llr0_alt[i] : bpl_args()->llr0[i];
                }
        }
        // ---- a3: every edge's bit->check message starts at the prior (two in LDS, one in a register)
        double loc[NB];
#define BPL_LLRT (bpl_args()->llr_tmp + (size_t)blockIdx.x * n)
        unsigned int decmask = 0u;  // bit r: hard decision of my r-th bit
#pragma unroll
        for (int r = 0; r < NB; ++r) {
            double lp = BPL_L0(r);
            if (UPRIOR) asm volatile("" : "+s"(lp));  // copied from the scalar pair here, not kept in a vector pair
            loc[r] = lp;
            *BPL_AT(alo[r]) = lp;
            *BPL_AT(ahi[r]) = lp;
        }
        if (want_llr()) {  // a syndrome that needs no iteration reports the priors
#pragma unroll
            for (int r = 0; r < NB; ++r) {
                const int i = BPL_BIT(r);
                double lp = BPL_L0(r);
                if (UPRIOR) asm volatile("" : "+s"(lp));
                if (i >= 0) BPL_LLRT[i] = lp;
            }
        }
        __syncthreads();

        int it_done = 0;
        bool conv = (sh[0] == 0);
        if (!conv) {
#pragma clang loop unroll(disable)
            for (int it = 1;; ++it) {
                const int fi = it & 1;
                {
                    // the wave's 64 positions of group j are one aligned pair of bitmap words: one broadcast read per group
                    // from a wave-uniform address (no per-lane address or mask registers live across the loop)
                    unsigned long long mis = 0ull;
#pragma unroll
                    for (int j = 0; j < CPT; ++j)
                        mis |= *(const volatile __attribute__((address_space(3))) unsigned long long*)(uintptr_t)(
                            diffw_base + (unsigned int)(((wave << 6) + j * NT) >> 3));  // (padding positions never raise their bits)
                    if (lane == 0 && mis) sh[fi] = 1;
                }
                if (it > P.max_iter) {
                    __syncthreads();
                    conv = (sh[fi] == 0);
                    it_done = P.max_iter;
                    break;
                }
                // =================== check -> bit pass (a4), speculative for it >= 2 ===========
                const unsigned long long alpha_u = alpha_bits_for_iteration(P.ms_scaling, it);  // scalar instructions
                const int alpha_lo = (int)(unsigned int)alpha_u, alpha_hi = (int)(unsigned int)(alpha_u >> 32), nalpha_hi = alpha_hi ^ (int)0x80000000;
                double vl[EARLY ? CPT : 1][4];
                if (EARLY) {
#pragma unroll
                    for (int j = 0; j < CPT; ++j)
#pragma unroll
                        for (int k = 0; k < 4; ++k) vl[j][k] = msg[(tid + j * NT) + k * MP];
                }
#pragma unroll
                for (int j = 0; j < CPT; ++j) {
                    msg_ptr mc = msg + (tid + j * NT);
                    double v[6];
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] = EARLY ? vl[j][k] : mc[k * MP];
                    v[4] = loc[2 * j];
                    v[5] = loc[2 * j + 1];
                    double pre[6], suf[6];
                    pre[0] = __DBL_MAX__;
#pragma unroll
                    for (int k = 1; k < 6; ++k) pre[k] = min_abs(pre[k - 1], v[k - 1]);
                    suf[5] = __DBL_MAX__;
#pragma unroll
                    for (int k = 4; k >= 0; --k) suf[k] = min_abs(suf[k + 1], v[k + 1]);
                    // signs: a message counts as negative when b2c <= 0 (a zero too, as in the reference); the outgoing
                    // message is mag * ((-1)^(syndrome + #negatives + own) * alpha).  The sign rides on the multiplier:
                    // one select of alpha's high word per edge (mag * (-alpha) == -(mag * alpha) bit for bit).
#if BPOSD_BPL_SIGN_ON_MAG
                    // the sign as a lane mask in scalar registers; it lands on the magnitude's high word through one select
                    // with a negated source (an instruction the compiler does not form from C), alpha is a scalar operand
                    unsigned long long negm[6];
                    unsigned long long parm = __ballot(sbit[j]);
#pragma unroll
                    for (int k = 0; k < 6; ++k) {
                        negm[k] = __ballot(v[k] <= 0.0);
                        parm ^= negm[k];
                    }
                    const double alpha_d = __hiloint2double(alpha_hi, alpha_lo);
#pragma unroll
                    for (int k = 0; k < 6; ++k) {
                        const double mag = (k == 0) ? suf[0] : (k == 5 ? pre[5] : min_pos(pre[k], suf[k]));
                        const unsigned long long selm = parm ^ negm[k];
                        int mh = __double2hiint(mag);
                        asm("v_cndmask_b32_e64 %0, %1, -%1, %2" : "=v"(mh) : "v"(mh), "s"(selm));
                        const double o = __hiloint2double(mh, __double2loint(mag)) * alpha_d;
                        if (k < 4) mc[k * MP] = o;
                        else loc[2 * j + (k - 4)] = o;
                    }
#else
                    bool neg[6];
                    bool par = sbit[j];
#pragma unroll
                    for (int k = 0; k < 6; ++k) {
                        neg[k] = (v[k] <= 0.0);
                        par ^= neg[k];
                    }
#pragma unroll
                    for (int k = 0; k < 6; ++k) {
                        const double mag = (k == 0) ? suf[0] : (k == 5 ? pre[5] : min_pos(pre[k], suf[k]));
                        const double sa = __hiloint2double((par ^ neg[k]) ? nalpha_hi : alpha_hi, alpha_lo);
                        const double o = mag * sa;
                        if (k < 4) mc[k * MP] = o;
                        else loc[2 * j + (k - 4)] = o;
                    }
#endif
                }
                __syncthreads();
                if (sh[fi] == 0) {
                    conv = true;
                    it_done = it - 1;
                    break;
                }
                if (tid == 0) sh[fi ^ 1] = 0;
                // ============ bit pass: posterior, decision, bit -> check (a6 / a7) ============
                const bool keep_llr = (it == P.max_iter) || want_llr();  // uniform
                // the two LDS messages of every owned bit: all of them up front (latency hidden inside the thread), or,
                // for the register-capped high-occupancy variant, two bits at a time
#ifndef BPOSD_BPL_BATCH
#define BPOSD_BPL_BATCH 2
#endif
                constexpr int BATCH = (MINW >= 8 && NB > 2) ? BPOSD_BPL_BATCH : NB;
                double X[NB], Y[NB];
#pragma unroll
                for (int r = 0; r < NB; ++r) {
                    if (r % BATCH == 0) {
#pragma unroll
                        for (int q = r; q < r + BATCH; ++q) {
                            X[q] = *BPL_AT(alo[q]);
                            Y[q] = *BPL_AT(ahi[q]);
                        }
                    }
                    double oR, oX, oY, t;
                    if (dl[r] == 0) bit_update<0>(BPL_L0(r), loc[r], X[r], Y[r], t, oR, oX, oY);
                    else if (dl[r] == 1) bit_update<1>(BPL_L0(r), loc[r], X[r], Y[r], t, oR, oX, oY);
                    else if (dl[r] == 2) bit_update<2>(BPL_L0(r), loc[r], X[r], Y[r], t, oR, oX, oY);
                    else bit_update_mixed((int)((dlpack >> (2 * r)) & 3u), BPL_L0(r), loc[r], X[r], Y[r], t, oR, oX, oY);
                    if (keep_llr) {
                        const int bi = BPL_BIT(r);
                        if (bi >= 0) BPL_LLRT[bi] = t;
                    }
                    loc[r] = oR;
                    *BPL_AT(alo[r]) = oX;
                    *BPL_AT(ahi[r]) = oY;
                    const unsigned int dnew = (t <= 0.0) ? 1u : 0u;
                    if (dnew != ((decmask >> r) & 1u)) {  // (a padding bit never gets here: see the tables above)
                        decmask ^= 1u << r;
                        unsigned int pa = alo[r], pb = ahi[r];
                        asm volatile("" : "+v"(pa), "+v"(pb));  // keep the rare path's address arithmetic in the branch
                        const int ca = (int)((pa - msg_base) >> 3) & (MP - 1), cb = (int)((pb - msg_base) >> 3) & (MP - 1);
                        const int co = tid + (r >> 1) * NT;  // slot = k * MP + position
                        atomicXor(&diffw[ca >> 5], 1u << (ca & 31));
                        atomicXor(&diffw[cb >> 5], 1u << (cb & 31));
                        atomicXor(&diffw[co >> 5], 1u << (co & 31));
                    }
                }
                __syncthreads();
            }
        }

        // ---- results
        const bool to_osd = (!conv) && bpl_args()->osd_enabled;
        if (tid == 0) {
            if (to_osd) {
                const int slot = atomicAdd(&bpl_args()->counters[1], 1);
                bpl_args()->osd_list[slot] = (int)s;
                sh[3] = slot;
            }
            if (bpl_args()->out_conv) bpl_args()->out_conv[s] = conv ? 1 : 0;
            if (bpl_args()->out_iters) bpl_args()->out_iters[s] = it_done;
            if (it_done) atomicAdd(bpl_args()->iter_total, (unsigned long long)it_done);
        }
        __syncthreads();
        const int slot = to_osd ? sh[3] : 0;
        constexpr bool packed = PACKED;
        if (packed)  // result rows as 64-bit words through an LDS bitmap (the messages are dead)
            bp_store_packed_rows<NB>((unsigned int*)smem, tid, NT, n, s, to_osd, (unsigned long long*)bpl_args()->out_bp,
                                     (unsigned long long*)bpl_args()->out_osd0, (unsigned long long*)bpl_args()->out_osdw,
                                     [&](int r) { return BPL_BIT(r); }, [&](int r) { return ((decmask >> r) & 1u) != 0u; });
#pragma unroll
        for (int r = 0; r < NB; ++r) {
            const int i = BPL_BIT(r);
            if (i >= 0) {
                const size_t o = (size_t)s * n + i;
                const uint8_t b = (uint8_t)((decmask >> r) & 1u);
                if (!packed) {
                    if (bpl_args()->out_bp) bpl_args()->out_bp[o] = b;
                    if (!to_osd) {
                        bpl_args()->out_osdw[o] = b;
                        if (bpl_args()->out_osd0) bpl_args()->out_osd0[o] = b;
                    }
                }
                if (to_osd) bpl_args()->llr_ws[(size_t)slot * n + i] = BPL_LLRT[i];
                if (want_llr()) bpl_args()->out_llr[o] = BPL_LLRT[i];
            }
        }
        __syncthreads();
    }
}

#undef BPL_L0
#undef BPL_AT
#undef BPL_BIT
#undef BPL_LLRT

}  // namespace bposd
