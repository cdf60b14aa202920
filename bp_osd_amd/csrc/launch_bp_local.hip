// launch_bp_local.hip -- bp_local_kernel ((3,6)-regular codes with n = 2m, min-sum): launch
// One translation unit of libbposd_mi355x.so: the kernels of this family are instantiated here and nowhere else.
#include "internal.h"

#include "bp_local_kernel.hip.h"

using namespace bposd;
using namespace bposd_host;

namespace bposd_host {
template <int CPT, int MP, int MINW, bool EARLY, bool UPRIOR, bool PACKED>
static int launch_bp_local_tp(bposd_handle* h, const BpLocalParams& L);

// the packed-I/O form exists for the instances auto-selection takes (internal.h: native_packed() asks for bp_variant == 0)
template <int CPT, int MP, int MINW, bool EARLY, bool UPRIOR = false>
static int launch_bp_local_t(bposd_handle* h, const BpLocalParams& L) {
    constexpr bool has_packed = !EARLY && ((CPT == 1 && MP == 1024 && MINW == 8) || (CPT == 2 && MP == 1024 && (MINW == 8 || MINW == 6)) || MP == 2048);
    if constexpr (has_packed) {
        if (L.packed_io) return launch_bp_local_tp<CPT, MP, MINW, EARLY, UPRIOR, true>(h, L);
    } else {
        if (L.packed_io) return fail(h, BPOSD_ERR_UNSUPPORTED, "this BP kernel variant has no packed-I/O form");
    }
    return launch_bp_local_tp<CPT, MP, MINW, EARLY, UPRIOR, false>(h, L);
}

template <int CPT, int MP, int MINW, bool EARLY, bool UPRIOR, bool PACKED>
static int launch_bp_local_tp(bposd_handle* h, const BpLocalParams& L) {
    auto k = bp_local_kernel<CPT, MP, MINW, EARLY, UPRIOR, PACKED>;
    const int nt = MP / CPT;
    const size_t lds = bp_local_lds_bytes(L.mp);
    { int rc_lds = set_max_lds(h, (const void*)k, lds); if (rc_lds) return rc_lds; }
    int wg_per_cu = 1;
    { int rc_occ = cached_occupancy(h, (const void*)k, nt, lds, &wg_per_cu); if (rc_occ) return rc_occ; }
    if (getenv("BPOSD_DEBUG_OCC")) fprintf(stderr, "[bposd] local-edge BP kernel: %d threads, %zu B LDS, %d workgroups per CU\n", nt, lds, wg_per_cu);
    wg_per_cu = std::max(1, std::min(wg_per_cu, 8));
    long long grid = std::min<long long>(L.B, (long long)h->num_cu * wg_per_cu);
    if (grid < 1) grid = 1;
    int rc = ensure_lanes(h, &Lane::bpl_llr, sizeof(double) * (size_t)grid * h->n);
    if (rc) return rc;
    BpLocalParams Lq = L;
    Lq.llr_tmp = (double*)h->cur->bpl_llr.p;
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(nt), lds, h->cur->stream, Lq);
    HIP_TRY(h, hipGetLastError());
    return 0;
}

int launch_bp_local(bposd_handle* h, const BpParams& P) {
    BpLocalParams L{};
    L.m = P.m; L.n = P.n; L.B = P.B; L.max_iter = P.max_iter; L.ms_scaling = P.ms_scaling; L.osd_enabled = P.osd_enabled;
    L.mp = h->local_mp;
    L.synd = P.synd; L.llr0 = P.llr0; L.sel = P.sel; L.llr0_alt = P.llr0_alt;
    L.pos_chk = h->d_lpos_chk; L.pos_bit = h->d_lpos_bit; L.pos_alo = h->d_lpos_alo; L.pos_ahi = h->d_lpos_ahi;
    L.grp_dl = h->d_lgrp_dl; L.pos_dl = h->d_lpos_dl;
    L.out_bp = P.out_bp; L.out_osd0 = P.out_osd0; L.out_osdw = P.out_osdw; L.out_conv = P.out_conv; L.out_iters = P.out_iters;
    L.out_llr = P.out_llr; L.llr_ws = P.llr_ws; L.osd_list = P.osd_list; L.counters = P.counters; L.iter_total = P.iter_total; L.tail_flag = P.tail_flag; L.packed_io = P.packed_io;
    if (h->local_mp == 2048) return launch_bp_local_t<2, 2048, 4, false>(h, L);  // 1024 threads, one workgroup per CU
    if (h->bp_variant == 17) return launch_bp_local_t<2, 1024, 8, false>(h, L);   // 512 threads, <= 64 VGPRs: 4 workgroups per CU
    if (h->bp_variant == 18) return launch_bp_local_t<1, 1024, 8, false>(h, L);   // 1024 threads, <= 64 VGPRs: 2 workgroups per CU
    if (h->bp_variant == 19) return launch_bp_local_t<4, 1024, 4, true>(h, L);    // 256 threads, <= 128 VGPRs: 4 workgroups per CU
    if (h->bp_variant == 20) return launch_bp_local_t<2, 1024, 6, true>(h, L);    // as the default with early check-pass loads
    if (h->bp_variant == 21) return launch_bp_local_t<4, 1024, 3, true>(h, L);    // 256 threads, <= 168 VGPRs: 3 workgroups per CU
    // one finite positive prior for every bit: it can live in scalar registers (positive: the padding positions share it)
    const bool uprior = h->probs_uniform && !L.sel && h->probs[0] > 0.0 && h->probs[0] < 0.5;
    // Small calls are latency-bound (a max_iter straggler runs ~2000 dependent iterations, a lone syndrome ~60): one check
    // per thread (16 waves per syndrome) iterates 25-30 % faster per syndrome, two checks per thread (4 workgroups per
    // CU) have the higher throughput.  Measured crossover on the [[1922,50]] code: 32768 syndromes per call (2048: 2.5
    // against 3.3 ms, 8192: 4.1 / 5.1, 32768: 9.6 / 10.0, 131072: 31.0 / 28.2).  A chunked host call counts as a whole.
    const long long work = h->batch_hint > 0 ? h->batch_hint : L.B;
    const bool small_call = h->bp_variant == 0 && work <= 40000;
    if (small_call) return uprior ? launch_bp_local_t<1, 1024, 8, false, true>(h, L) : launch_bp_local_t<1, 1024, 8, false>(h, L);
    if ((h->bp_variant == 22 || h->bp_variant == 0) && uprior) return launch_bp_local_t<2, 1024, 8, false, true>(h, L);  // <= 64 VGPRs: 4 workgroups per CU
    if (h->bp_variant == 23 && uprior) return launch_bp_local_t<2, 1024, 6, true, true>(h, L);
    if (h->bp_variant == 24 && uprior) return launch_bp_local_t<2, 1024, 6, false, true>(h, L);
    if (h->bp_variant == 25 && uprior) return launch_bp_local_t<2, 1024, 8, true, true>(h, L);   // 22 with early check-pass loads
    if (h->bp_variant == 26 && uprior) return launch_bp_local_t<1, 1024, 8, false, true>(h, L);  // 18 with the scalar prior
    return launch_bp_local_t<2, 1024, 6, false>(h, L);                            // 512 threads, <= 80 VGPRs: 3 workgroups per CU
}
}  // namespace bposd_host
