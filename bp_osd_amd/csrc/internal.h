// internal.h -- host-side state shared by the translation units of libbposd_mi355x.so (not part of the C-ABI).
// bposd_capi.hip holds the C-ABI, table construction and the decode calls; every launch_*.hip holds the instantiations
// and launch code of one kernel family, so that the families compile in parallel (bp_osd_amd/build.py).
#pragma once
#include "../../include/bposd_mi355x.h"
#include "../../include/bposd_mi355x_debug.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "bp_kernel.hip.h"   // BpParams, bp_lds_bytes (templates only: nothing is instantiated by including it)
#include "osd_kernel.hip.h"  // OsdParams

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
};

// Per-call state.  A handle owns BPOSD_LANES of these and alternates between them: consecutive decode calls (and the
// chunks of one host-pointer call) run on different HIP streams with their own workspaces, so the persistent
// workgroups of call k + 1 pick up the CUs that call k's last max_iter stragglers and its OSD kernel leave idle.
constexpr int BPOSD_LANES = 4;  // large codes (HBM-resident workspaces of several GB per lane) use two of them
constexpr int BPOSD_MAX_CHUNKS = 16;  // chunks of one host-pointer call (bposd_decode_batch)

struct Lane {
    hipStream_t stream = nullptr;
    // The OSD kernel of a call runs on a stream of its own at the highest priority (ordered behind the call's BP kernel
    // and in front of whatever follows on `stream` by events): its few, fat workgroups otherwise queue behind the full
    // grid of the NEXT call's BP kernel for every CU that frees up and take many times their own run time.
    hipStream_t osd_stream = nullptr;
    hipEvent_t ev_bp = nullptr, ev_osd = nullptr;
    hipEvent_t ev_done = nullptr;  // both kernels of the lane's last (non-lean) call have ended
    bool done_recorded = false;
    void* h_stage = nullptr;     // page-locked, device-visible staging for small host-pointer calls (zero-copy path)
    size_t h_stage_bytes = 0;
    hipEvent_t ev_up = nullptr;  // host-pointer calls: this lane's chunk has been uploaded (uploads go one at a time, in
                                 // chunk order: the first chunk's kernels then start after one chunk's copy time)
    DevBuf bpl_msg, bpl_llr;  // large BP workspaces (bpl_llr also serves the local-edge kernel: LLRs of the current syndrome)
    DevBuf osdl_ws;           // large OSD workspaces (matrix, sort keys, pivots, weights) carved from one allocation
    DevBuf osd_rows_ws;       // OSD kernel's per-workgroup spill area for finished row words
    DevBuf llr_ws, osd_list, io_synd, io_osdw, io_osd0, io_bp, io_conv, io_iters, io_llr, io_sel;
    // host-pointer calls: the outputs are downloaded on a copy stream of the lane's own right after the BP kernel (event-
    // ordered), the rows the OSD kernel rewrites come from compact copies [list slot][n] once it has run
    DevBuf io_cmp0, io_cmpw;
    // bit-packed host I/O (bposd_decode_batch_packed): packed syndromes in, packed rows out, packed compact OSD rows
    DevBuf io_psynd, io_posdw, io_posd0, io_pbp, io_pcmp;
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev_copy = nullptr;    // the chunk's downloads have left the lane's io buffers
    int* h_list = nullptr;           // page-locked copy of the chunk's OSD list (syndrome index per slot)
    size_t h_list_cap = 0;
    bool copy_pending = false;
    long long* d_osd_dbg = nullptr;  // diagnostics (BPOSD_OSD_DEBUG=1): phase timestamps
    int* d_counters = nullptr;       // 4 ints
    // per-shot channel of a device-pointer call (bposd_decode_batch_select_device): priors and weights of the alternative
    // channel, 2n doubles, copied from a page-locked staging block on the lane's own stream -- consecutive select calls
    // overlap like plain ones (the first version drained every lane and made two blocking copies per call)
    double* d_alt = nullptr;
    double* h_alt = nullptr;
    hipEvent_t ev_alt = nullptr;     // the staging block has been read
    bool alt_busy = false;
    int* h_tail = nullptr;           // page-locked, device-visible: the BP kernel of this lane's current call has entered its tail
};

// What bposd_last_timing reports: one record per kernel pair launched by the last call (one per chunk for a
// host-pointer call).  Events and the pinned counter copies live in the handle so that records outlive lane reuse.
struct CallRecord {
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
    int* h_counters = nullptr;               // pinned: 4 ints
    unsigned long long* h_iter_total = nullptr;  // pinned
    bool ran_osd = false;
    bool recorded = false;  // the counters (and, when timed, the events) have been recorded at least once
    bool timed = false;     // the three events bracket the kernels of this record (not on the lean small-call path)
};

struct bposd_handle {
    bposd_config cfg{};
    int device = 0;
    Lane lanes[BPOSD_LANES];
    Lane* cur = nullptr;      // lane of the call being enqueued
    int nlanes = BPOSD_LANES; // lanes this handle cycles through
    int next_lane = 0;
    CallRecord lane_rec[BPOSD_LANES];  // device-pointer calls: the record of the last call queued on each lane
    CallRecord rec[BPOSD_MAX_CHUNKS];  // host-pointer calls: one record per chunk
    CallRecord* currec = nullptr;
    int nrec = 0;             // > 0: the last call was a host-pointer call of that many chunks
    int last_lane = 0;        // lane of the last device-pointer call
    int num_cu = 0;
    size_t lds_per_cu = 160 * 1024;
    int m = 0, n = 0, E = 0;
    int dc_max = 0, dv_max = 0;
    bool regular = false;
    // local-edge BP kernel (bp_local_kernel.hip.h): available for (3,6)-regular codes with n = 2m, min-sum
    bool local_ok = false;
    int local_mp = 0;
    long long local_passes = 0;  // modelled ds_read_b64 cycles of the bit pass in the chosen layout (floor: 4 * MP / 32)
    long long local_wcycles = 0; // modelled ds_write_b64 cycles of the bit pass (floor: 6 * 4 * MP / 64)
    int *d_lpos_chk = nullptr, *d_lpos_bit = nullptr, *d_lpos_alo = nullptr, *d_lpos_ahi = nullptr, *d_lgrp_dl = nullptr, *d_lpos_dl = nullptr;
    // class BP kernel (bp_class_kernel.hip.h): every check has the same degree, bit degrees inside one compiled range
    bool bp_any = false;  // degrees beyond the compiled kernels: bp_anydeg_kernel.hip.h (run-time degree loops, messages in HBM)
    bool class_ok = false;
    int class_dclo = 0, class_dc = 0, class_dvlo = 0, class_dvhi = 0, class_mp = 0, class_nt = 0;
    long class_read_cycles = 0, class_write_cycles = 0, class_read_floor = 0, class_write_floor = 0;  // modelled, one bit pass
    int *d_cpos_chk = nullptr, *d_cpos_bit = nullptr, *d_cbit_slot = nullptr, *d_cgrp_deg = nullptr, *d_cgrp_cdeg = nullptr;
    bool large = false;   // beyond the register-resident OSD kernel: HBM-resident matrix, device rank probe
    bool bp_hbm = false;  // BP messages do not fit one CU's LDS either: HBM-resident BP kernel
    int max_iter = 0;
    int rank = 0, kprime = 0, ncand = 0;
    bool probs_uniform = true;
    int bp_variant = 0;
    int last_bp_kernel = -1;  // BPOSD_BP_KERNEL_* of the last BP launch
    int osd_variant = 0;      // 0 auto, 1 = one workgroup per elimination (osd_kernel), 2 = one wave per elimination where it applies
    int last_osd_kernel = -1; // 0 none yet, 1 osd_kernel, 2 osd_wave_kernel, 3 osd_large_kernel
    // host copies
    std::vector<int> rp, ci;
    std::vector<double> probs;
    // device tables
    int *d_rp = nullptr, *d_ci = nullptr;
    int *d_chk_deg = nullptr, *d_var_deg = nullptr, *d_var_pos = nullptr, *d_pos_bit = nullptr, *d_var_ck = nullptr;
    int large_form = 0;  // form of the last bp_large_kernel launch: 0 per-edge messages (product-sum), 1 check records in the workspace, 2 per-check data in LDS
    int tab_np = 0;
    long layout_cost = 0, layout_cost_natural = 0, layout_cost_ideal = 0;  // simulated LDS cycles of the bit pass
    double* d_llr0 = nullptr;
    double* d_cost = nullptr;  // log(1/p_i): OSD-W weights of the ldpc-v2 weight function
    double *d_llr0_alt = nullptr, *d_cost_alt = nullptr;  // alternative channel of the two-valued per-shot form
    bool fp_weights = false;   // non-uniform (or degenerate) channel: candidate weights need the fp64 sums
    // serial schedule (cfg.schedule == 1): CSC view and level lists
    int *d_cp = nullptr, *d_ce = nullptr, *d_erow = nullptr, *d_lvl_ptr = nullptr, *d_lvl_bits = nullptr;
    int nlevels = 0;
    int tab_dc = 0, tab_dv = 0, tab_mp = 0;  // layout the tables were built for
    bool have_timing = false;
    long long batch_hint = 0;          // > 0 while a chunked host call is being enqueued: its whole batch size
    bool async_pending = false;        // a device-pointer call may still be running on some lane
    hipStream_t osd_now = nullptr;     // stream the OSD kernel of the call being enqueued goes to
    uint8_t *cmp_osd0 = nullptr, *cmp_osdw = nullptr;  // compact OSD rows of the chunk being enqueued (host-pointer calls)
    bool bp_only = false;              // the call being enqueued wants BP's outputs only (bposd_posterior_llr): no OSD kernel
    bool lane_alt = false;             // the call being enqueued takes the alternative channel from its lane's buffers
    bool packed_now = false;           // the call being enqueued hands the kernels packed syndromes and takes packed result rows
    bool tail_gate = false;            // the call being enqueued is a chunk of a host-pointer call: its BP kernel reports its tail
    std::string err;
};

namespace bposd_host {

int fail(bposd_handle* h, int code, const char* fmt, ...) __attribute__((format(printf, 3, 4)));

#define HIP_TRY(h, expr)                                                                       \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess)                                                                  \
            return bposd_host::fail(h, BPOSD_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                                    __FILE__, __LINE__);                                       \
    } while (0)

int sync_all_lanes(bposd_handle* h);
int set_max_lds(bposd_handle* h, const void* kernel, size_t lds);
int cached_occupancy(bposd_handle* h, const void* kernel, int nt, size_t lds, int* out);
int ensure(bposd_handle* h, DevBuf& b, size_t bytes);
int ensure_lanes(bposd_handle* h, DevBuf Lane::*member, size_t bytes);
void release(DevBuf& b);

// ---- table construction (bposd_capi.hip)
int build_tables(bposd_handle* h, int DC, int DV, int MP, int NT, int VPT);

// ---- LDS-resident BP kernel: workgroup shapes (launch_bp_lds.hip)
bool is_reg63(const bposd_handle* h);
int shape_cpt(int shape);
int shape_threads(const bposd_handle* h, int shape);
int pick_shape(const bposd_handle* h);

// ---- kernel launches, one translation unit per family
int launch_bp(bposd_handle* h, bposd::BpParams& P);                 // launch_bp_lds.hip    (bp_kernel)
int launch_bp_local(bposd_handle* h, const bposd::BpParams& P);     // launch_bp_local.hip  (bp_local_kernel)
int launch_bp_class(bposd_handle* h, const bposd::BpParams& P);     // launch_bp_class.hip  (bp_class_kernel)
bool class_preferred(const bposd_handle* h);
int launch_bp_large(bposd_handle* h, const bposd::BpParams& P);     // launch_bp_misc.hip   (bp_large_kernel)
int launch_bp_serial(bposd_handle* h, const bposd::BpParams& P);    //                      (bp_serial_kernel)
int launch_bp_any(bposd_handle* h, const bposd::BpParams& P);       //                      (bp_anydeg_kernel)
int bp_serial_max_dv();
size_t bp_large_lds_need(int m, int n);
int launch_osd(bposd_handle* h, const bposd::OsdParams& P, long long B);  // launch_osd.hip (osd_kernel, osd_wave_kernel)
int osd_words(int n);
int launch_osd_large(bposd_handle* h, const bposd::OsdParams& P, long long B, int* d_rank_out);  // launch_osd_large.hip
int osd_large_maxspan(bool cs);
// do the kernels this handle runs read packed syndromes / write packed rows themselves (else unpack / pack kernels surround them)?
inline bool native_packed(const bposd_handle* h) {
    // (forced local-edge variants 16 .. 26 have no packed instantiation; the LDS and class kernels switch at run time)
    return !h->bp_any && h->cfg.schedule == 0 && h->bp_variant != 64 && !(h->bp_variant >= 16 && h->bp_variant <= 26);
}

}  // namespace bposd_host
