// launch_bp_misc.hip -- bp_large_kernel (messages in HBM), bp_serial_kernel (schedule = serial), bp_anydeg_kernel (any degree): launch
// One translation unit of libbposd_mi355x.so: the kernels of this family are instantiated here and nowhere else.
#include "internal.h"

#include "bp_large_kernel.hip.h"
#include "bp_serial_kernel.hip.h"
#include "bp_anydeg_kernel.hip.h"

using namespace bposd;
using namespace bposd_host;

namespace bposd_host {
template <int DC, int DV, int METHOD>
static int launch_bp_large_tm(bposd_handle* h, BpLargeParams& P) {
    h->large_form = METHOD;
    const size_t lds = bp_large_lds_bytes(h->m, h->n, METHOD == 2, DC);
    auto k = bp_large_kernel<DC, DV, METHOD>;
    // persistent workgroups: what registers and LDS admit per CU (the message workspace is per workgroup)
    int wg_per_cu = 1;
    { int rc_lds = set_max_lds(h, (const void*)k, lds); if (rc_lds) return rc_lds; }
    { int rc_occ = cached_occupancy(h, (const void*)k, bp_large_threads(METHOD), lds, &wg_per_cu); if (rc_occ) return rc_occ; }
    const int occ = std::max(1, std::min(wg_per_cu, 4));
    // The check records of the min-sum form are gathered ~11 times each in a bit pass; with one workgroup per CU the 256 record
    // arrays (465 KB each on 14520 x 29524) stay in the memory-side cache between uses: 72-74 -> 61-62 ms per 1024 syndromes.
    wg_per_cu = METHOD >= 1 ? 1 : occ;
    if (const char* e = getenv("BPOSD_LARGE_WG_CAP")) wg_per_cu = std::max(1, std::min(occ, atoi(e)));
    long long grid = std::max<long long>(1, std::min<long long>(P.B, (long long)h->num_cu * wg_per_cu));
    if (const char* e = getenv("BPOSD_LARGE_BP_GRID")) grid = std::max<long long>(1, std::min<long long>(grid, atoll(e)));  // (probe: a bandwidth-bound kernel on fewer CUs)
    constexpr int SLOTS = METHOD == 1 ? DC + 4 : (METHOD == 2 ? DC + 1 : DC);  // message planes (+ the check records of the min-sum form)
    int rc;
    if ((rc = ensure_lanes(h, &Lane::bpl_msg, sizeof(double) * (size_t)grid * SLOTS * P.mp))) return rc;
    if ((rc = ensure_lanes(h, &Lane::bpl_llr, sizeof(double) * (size_t)grid * h->n))) return rc;
    P.msg_ws = (double*)h->cur->bpl_msg.p;
    P.llr_tmp = (double*)h->cur->bpl_llr.p;
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(bp_large_threads(METHOD)), lds, h->cur->stream, P);
    HIP_TRY(h, hipGetLastError());
    return 0;
}

template <int DC, int DV>
static int launch_bp_large_t(bposd_handle* h, BpLargeParams& P) {
    if (h->cfg.bp_method != BPOSD_BP_MIN_SUM) return launch_bp_large_tm<DC, DV, 0>(h, P);
    // min-sum: a1 and the flags of every check in LDS where they fit (bp_variant 63 forces the form with whole records in the workspace)
    const bool a1_in_lds = bp_large_lds_bytes(h->m, h->n, true, DC) <= h->lds_per_cu && h->bp_variant != 63;
    return a1_in_lds ? launch_bp_large_tm<DC, DV, 2>(h, P) : launch_bp_large_tm<DC, DV, 1>(h, P);
}

int launch_bp_large(bposd_handle* h, const BpParams& G) {
    BpLargeParams P{};
    P.m = G.m; P.n = G.n; P.B = G.B; P.max_iter = G.max_iter; P.ms_scaling = G.ms_scaling; P.ps_clip = G.ps_clip; P.ps_form = h->cfg.ps_math_form;
    P.osd_enabled = G.osd_enabled; P.mp = h->tab_mp;
    P.synd = G.synd; P.llr0 = G.llr0; P.sel = G.sel; P.llr0_alt = G.llr0_alt;
    P.chk_deg = h->d_chk_deg; P.var_deg = h->d_var_deg; P.var_pos = h->d_var_pos; P.var_ck = h->d_var_ck;
    P.out_bp = G.out_bp; P.out_osd0 = G.out_osd0; P.out_osdw = G.out_osdw; P.out_conv = G.out_conv;
    P.out_iters = G.out_iters; P.out_llr = G.out_llr; P.llr_ws = G.llr_ws; P.osd_list = G.osd_list;
    P.counters = G.counters; P.iter_total = G.iter_total; P.tail_flag = G.tail_flag; P.packed_io = G.packed_io;
    if (h->dc_max <= 12 && h->dv_max <= 6) return launch_bp_large_t<12, 6>(h, P);
    if (h->dc_max <= 16 && h->dv_max <= 8) return launch_bp_large_t<16, 8>(h, P);
    return fail(h, BPOSD_ERR_UNSUPPORTED, "check degree %d / bit degree %d exceed the built kernels (16 / 8)", h->dc_max, h->dv_max);
}

int launch_bp_serial(bposd_handle* h, const BpParams& P) {
    BpSerialParams S{};
    S.m = P.m; S.n = P.n; S.E = h->E; S.B = P.B; S.max_iter = P.max_iter; S.bp_method = h->cfg.bp_method;
    S.ms_scaling = P.ms_scaling; S.ps_clip = P.ps_clip; S.ps_form = h->cfg.ps_math_form; S.osd_enabled = P.osd_enabled; S.nlevels = h->nlevels;
    S.synd = P.synd; S.llr0 = P.llr0; S.sel = P.sel; S.llr0_alt = P.llr0_alt;
    S.rp = h->d_rp; S.ci = h->d_ci; S.cp = h->d_cp; S.ce = h->d_ce; S.erow = h->d_erow;
    S.lvl_ptr = h->d_lvl_ptr; S.lvl_bits = h->d_lvl_bits;
    S.out_bp = P.out_bp; S.out_osd0 = P.out_osd0; S.out_osdw = P.out_osdw; S.out_conv = P.out_conv; S.out_iters = P.out_iters;
    S.out_llr = P.out_llr; S.llr_ws = P.llr_ws; S.osd_list = P.osd_list; S.counters = P.counters; S.iter_total = P.iter_total;
    const size_t lds = bp_serial_lds_bytes(h->n);
    const int wg_per_cu = std::max<int>(1, std::min<size_t>(8, h->lds_per_cu / std::max<size_t>(lds, 1)));
    const long long grid = std::max<long long>(1, std::min<long long>(P.B, (long long)h->num_cu * wg_per_cu));
    int rc;
    if ((rc = ensure_lanes(h, &Lane::bpl_msg, sizeof(double) * (size_t)grid * h->E))) return rc;
    if ((rc = ensure_lanes(h, &Lane::bpl_llr, sizeof(double) * (size_t)grid * h->n))) return rc;
    S.msg_ws = (double*)h->cur->bpl_msg.p;
    S.llr_tmp = (double*)h->cur->bpl_llr.p;
    { int rc_lds = set_max_lds(h, (const void*)bp_serial_kernel, lds); if (rc_lds) return rc_lds; }
    hipLaunchKernelGGL(bp_serial_kernel, dim3((unsigned)grid), dim3(BPS_NT), lds, h->cur->stream, S);
    HIP_TRY(h, hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------ any-degree BP (check degree > 16 or bit degree > 8)
int launch_bp_any(bposd_handle* h, const BpParams& P) {
    BpAnyParams A{};
    A.m = P.m; A.n = P.n; A.E = h->E; A.B = P.B; A.max_iter = P.max_iter; A.bp_method = h->cfg.bp_method;
    A.ms_scaling = P.ms_scaling; A.ps_clip = P.ps_clip; A.ps_form = h->cfg.ps_math_form; A.osd_enabled = P.osd_enabled;
    A.synd = P.synd; A.llr0 = P.llr0; A.sel = P.sel; A.llr0_alt = P.llr0_alt;
    A.rp = h->d_rp; A.ci = h->d_ci; A.cp = h->d_cp; A.ce = h->d_ce;
    A.out_bp = P.out_bp; A.out_osd0 = P.out_osd0; A.out_osdw = P.out_osdw; A.out_conv = P.out_conv; A.out_iters = P.out_iters;
    A.out_llr = P.out_llr; A.llr_ws = P.llr_ws; A.osd_list = P.osd_list; A.counters = P.counters; A.iter_total = P.iter_total;
    A.tail_flag = P.tail_flag;
    const size_t lds = bp_anydeg_lds_bytes(h->n);
    const int wg_per_cu = std::max<int>(1, std::min<size_t>(8, h->lds_per_cu / std::max<size_t>(lds, 1)));
    const long long grid = std::max<long long>(1, std::min<long long>(P.B, (long long)h->num_cu * wg_per_cu));
    int rc;
    if ((rc = ensure_lanes(h, &Lane::bpl_msg, sizeof(double) * (size_t)grid * 3 * h->E))) return rc;
    if ((rc = ensure_lanes(h, &Lane::bpl_llr, sizeof(double) * (size_t)grid * h->n))) return rc;
    A.msg_ws = (double*)h->cur->bpl_msg.p;
    A.llr_tmp = (double*)h->cur->bpl_llr.p;
    { int rc_lds = set_max_lds(h, (const void*)bp_anydeg_kernel, lds); if (rc_lds) return rc_lds; }
    hipLaunchKernelGGL(bp_anydeg_kernel, dim3((unsigned)grid), dim3(BPA_NT), lds, h->cur->stream, A);
    HIP_TRY(h, hipGetLastError());
    return 0;
}

int bp_serial_max_dv() { return BPS_MAXDV; }
size_t bp_large_lds_need(int m, int n) { return bp_large_lds_bytes(m, n); }

}  // namespace bposd_host
