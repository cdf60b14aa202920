// launch_bp_class.hip -- bp_class_kernel (one check degree, bits in degree classes): launch
// One translation unit of libbposd_mi355x.so: the kernels of this family are instantiated here and nowhere else.
#include "internal.h"

#include "bp_class_kernel.hip.h"
// occupancy targets of the class kernel instances (minimum waves per SIMD the register allocation must allow)
#ifndef BPOSD_CLASS7_MINW
#define BPOSD_CLASS7_MINW 8
#endif
#ifndef BPOSD_CLASS7_MINW_PS
#define BPOSD_CLASS7_MINW_PS 7  // (8 would cap the SGPRs at 80 and spill them)
#endif
#ifndef BPOSD_CLASS6_MINW_PS
#define BPOSD_CLASS6_MINW_PS 7
#endif

using namespace bposd;
using namespace bposd_host;

namespace bposd_host {
constexpr int kClassVPT = 2;

template <int DCLO, int DC, int DVLO, int DVHI, int MP, int MINW, int METHOD, bool UPRIOR>
static int launch_bp_class_t(bposd_handle* h, const BpClassParams& C) {
    auto k = bp_class_kernel<DCLO, DC, DVLO, DVHI, 1, kClassVPT, MP, MP, MINW, METHOD, UPRIOR>;
    const int nt = h->class_nt;
    const size_t lds = bp_class_lds_bytes(DC, MP, MP);
    { int rc_lds = set_max_lds(h, (const void*)k, lds); if (rc_lds) return rc_lds; }
    int wg_per_cu = 1;
    { int rc_occ = cached_occupancy(h, (const void*)k, nt, lds, &wg_per_cu); if (rc_occ) return rc_occ; }
    if (getenv("BPOSD_DEBUG_OCC")) fprintf(stderr, "[bposd] class BP kernel <%d..%d;%d..%d> stride %d: %d threads, %zu B LDS, %d workgroups per CU\n", DCLO, DC, DVLO, DVHI, MP, nt, lds, wg_per_cu);
    wg_per_cu = std::max(1, std::min(wg_per_cu, 16));
    if (const char* e = getenv("BPOSD_CLASS_WG_CAP")) wg_per_cu = std::max(1, std::min(wg_per_cu, atoi(e)));
    long long grid = std::min<long long>(C.B, (long long)h->num_cu * wg_per_cu);
    if (grid < 1) grid = 1;
    int rc = ensure_lanes(h, &Lane::bpl_llr, sizeof(double) * (size_t)grid * h->n);
    if (rc) return rc;
    BpClassParams Cq = C;
    Cq.llr_tmp = (double*)h->cur->bpl_llr.p;
    // syndromes per queue atomic: (what is left) / (2 x grid), at most eight, one at the end (guided self-scheduling)
    // -- for codes of up to 160 checks only, where the queue atomic is the bound (surface code d = 5: 0.93 -> 0.61 ms per
    // 65536 syndromes); larger codes lose 4-6 % to the coarser tail (tools/bp_iteration_cost.py, A/B in one run)
    Cq.queue_batch = h->m <= 160 ? 8 : 1;
    Cq.queue_shift = 1;
    while ((1ll << Cq.queue_shift) < grid * 2) Cq.queue_shift++;
    if (const char* e = getenv("BPOSD_CLASS_QUEUE_BATCH")) Cq.queue_batch = std::max(1, std::min(64, atoi(e)));
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(nt), lds, h->cur->stream, Cq);
    HIP_TRY(h, hipGetLastError());
    return 0;
}

template <int DCLO, int DC, int DVLO, int DVHI, int MINW_MS, int MINW_PS>
static int launch_bp_class_shape(bposd_handle* h, const BpClassParams& C, bool uprior) {
    const bool ms = h->cfg.bp_method == BPOSD_BP_MIN_SUM;
#define BPOSD_CLASS_MP(MPV)                                                                                              \
    if (h->class_mp == MPV) {                                                                                            \
        if (ms) return uprior ? launch_bp_class_t<DCLO, DC, DVLO, DVHI, MPV, MINW_MS, 1, true>(h, C) : launch_bp_class_t<DCLO, DC, DVLO, DVHI, MPV, MINW_MS, 1, false>(h, C); \
        if (h->cfg.ps_math_form) /* product-sum, two divisions per edge */                                               \
            return uprior ? launch_bp_class_t<DCLO, DC, DVLO, DVHI, MPV, MINW_PS, 0, true>(h, C) : launch_bp_class_t<DCLO, DC, DVLO, DVHI, MPV, MINW_PS, 0, false>(h, C); \
        return uprior ? launch_bp_class_t<DCLO, DC, DVLO, DVHI, MPV, MINW_PS, 2, true>(h, C) : launch_bp_class_t<DCLO, DC, DVLO, DVHI, MPV, MINW_PS, 2, false>(h, C); \
    }
    BPOSD_CLASS_MP(256)
    BPOSD_CLASS_MP(512)
    BPOSD_CLASS_MP(1024)
#undef BPOSD_CLASS_MP
    return fail(h, BPOSD_ERR_UNSUPPORTED, "no class BP kernel for stride %d", h->class_mp);
}

// Measured (tools/bp_iteration_cost.py): the class kernel wins everywhere except product-sum on a (3,6)-regular code of
// ~1000 checks (H1922: 8.3 against 6.4 ps per edge-iteration) -- VALU-bound, and the two-checks-per-thread shape of the
// regular LDS kernel interleaves two division chains per thread.
bool class_preferred(const bposd_handle* h) {
    return !(h->cfg.bp_method != BPOSD_BP_MIN_SUM && is_reg63(h) && h->class_mp == 1024);
}

int launch_bp_class(bposd_handle* h, const BpParams& P) {
    BpClassParams C{};
    C.m = P.m; C.n = P.n; C.B = P.B; C.max_iter = P.max_iter; C.ms_scaling = P.ms_scaling; C.ps_clip = P.ps_clip; C.osd_enabled = P.osd_enabled;
    C.synd = P.synd; C.llr0 = P.llr0; C.sel = P.sel; C.llr0_alt = P.llr0_alt;
    C.pos_chk = h->d_cpos_chk; C.pos_bit = h->d_cpos_bit; C.bit_slot = h->d_cbit_slot; C.grp_deg = h->d_cgrp_deg; C.grp_cdeg = h->d_cgrp_cdeg;
    C.out_bp = P.out_bp; C.out_osd0 = P.out_osd0; C.out_osdw = P.out_osdw; C.out_conv = P.out_conv; C.out_iters = P.out_iters;
    C.out_llr = P.out_llr; C.llr_ws = P.llr_ws; C.osd_list = P.osd_list; C.counters = P.counters; C.iter_total = P.iter_total; C.tail_flag = P.tail_flag; C.packed_io = P.packed_io;
    const bool uprior = h->probs_uniform && P.sel == nullptr && h->probs[0] > 0.0 && h->probs[0] < 0.5;
    if (h->class_dc == 7) return launch_bp_class_shape<7, 7, 3, 4, BPOSD_CLASS7_MINW, BPOSD_CLASS7_MINW_PS>(h, C, uprior);
    if (h->class_dc == 6) return launch_bp_class_shape<6, 6, 3, 3, 8, BPOSD_CLASS6_MINW_PS>(h, C, uprior);
    if (h->class_dc == 4 && h->class_dclo == 4) return launch_bp_class_shape<4, 4, 2, 2, 8, 7>(h, C, uprior);
    if (h->class_dc == 4 && h->class_dclo == 3) return launch_bp_class_shape<3, 4, 1, 2, 8, 7>(h, C, uprior);
    if (h->class_dc == 8) return launch_bp_class_shape<8, 8, 4, 4, 7, 6>(h, C, uprior);
    return fail(h, BPOSD_ERR_UNSUPPORTED, "no class BP kernel for check degree %d", h->class_dc);
}
}  // namespace bposd_host
