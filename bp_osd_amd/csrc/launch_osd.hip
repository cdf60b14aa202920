// launch_osd.hip -- osd_kernel (one workgroup per elimination) and osd_wave_kernel (one wave per elimination): launch
// One translation unit of libbposd_mi355x.so: the kernels of this family are instantiated here and nowhere else.
#include "internal.h"

#include "osd_wave_kernel.hip.h"
#include "osd_mw_kernel.hip.h"

using namespace bposd;
using namespace bposd_host;

namespace bposd_host {
template <int W, bool PACKED>
static int launch_osd_tp(bposd_handle* h, const OsdParams& P, long long B) {
    // OSD_RPT rows per thread: 4 waves cover 1024 rows
    const int rows_per_thread = OSD_RPT;
    const int NT = std::min(64 * OSD_MAXW, std::max(64, ((h->m + rows_per_thread - 1) / rows_per_thread + 63) / 64 * 64));
    const size_t lds = osd_lds_bytes(W, NT * OSD_RPT);
    auto k = osd_kernel<W, PACKED>;
    { int rc_lds = set_max_lds(h, (const void*)k, lds); if (rc_lds) return rc_lds; }
    // persistent workgroups: as many per CU as registers and LDS admit (H1922: one 8-wave workgroup; the reference's
    // [[400,16,6]] code: four 2-wave workgroups -- with one per CU its 32 k eliminations per batch took longer than BP)
    int wg_per_cu = 1;
    { int rc_occ = cached_occupancy(h, (const void*)k, NT, lds, &wg_per_cu); if (rc_occ) return rc_occ; }
    wg_per_cu = std::max(1, std::min(wg_per_cu, 8));
    long long grid = std::min<long long>(B, (long long)h->num_cu * wg_per_cu);
    if (grid < 1) grid = 1;
    int rc = ensure_lanes(h, &Lane::osd_rows_ws, sizeof(unsigned long long) * (size_t)grid * W * NT * OSD_RPT);
    if (rc) return rc;
    OsdParams Q = P;
    Q.rows_ws = (unsigned long long*)h->cur->osd_rows_ws.p;
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(NT), lds, h->osd_now ? h->osd_now : h->cur->osd_stream, Q);
    HIP_TRY(h, hipGetLastError());
    return 0;
}

template <int W>
static int launch_osd_t(bposd_handle* h, const OsdParams& P, long long B) {
    return P.packed_io ? launch_osd_tp<W, true>(h, P, B) : launch_osd_tp<W, false>(h, P, B);
}

int osd_words(int n) {
    const int need = (n + 1 + 63) / 64;
    for (int w : {1, 2, 4, 8, 16, 24, 31, 32})
        if (w >= need) return w;
    return 0;
}

// one wave per elimination (osd_wave_kernel.hip.h): small codes, integer weights
template <int RPL, int W, bool PACKED>
static int launch_osd_wave_tp(bposd_handle* h, const OsdParams& P, long long B);
template <int RPL, int W>
static int launch_osd_wave_t(bposd_handle* h, const OsdParams& P, long long B) {
    return P.packed_io ? launch_osd_wave_tp<RPL, W, true>(h, P, B) : launch_osd_wave_tp<RPL, W, false>(h, P, B);
}
template <int RPL, int W, bool PACKED>
static int launch_osd_wave_tp(bposd_handle* h, const OsdParams& P, long long B) {
    auto k = osd_wave_kernel<RPL, W, PACKED>;
    const size_t lds = OSDW_WAVES * osdw_lds_per_wave(osdw_nsort(h->n), RPL, W);
    { int rc_lds = set_max_lds(h, (const void*)k, lds); if (rc_lds) return rc_lds; }
    int wg_per_cu = 1;
    { int rc_occ = cached_occupancy(h, (const void*)k, 64 * OSDW_WAVES, lds, &wg_per_cu); if (rc_occ) return rc_occ; }
    wg_per_cu = std::max(1, std::min(wg_per_cu, 8));
    long long grid = std::min<long long>((B + OSDW_WAVES - 1) / OSDW_WAVES, (long long)h->num_cu * wg_per_cu);
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(64 * OSDW_WAVES), lds, h->osd_now ? h->osd_now : h->cur->osd_stream, P);
    HIP_TRY(h, hipGetLastError());
    return 0;
}

// 0 = the code / configuration stays on osd_kernel.  A lone elimination is faster on a workgroup of its own (measured,
// tools/latency_reference_codes.py: a decode() that needs OSD 1.23 against 1.92 ms on [[900,36,10]], 0.50 / 0.57 ms on
// [[400,16,6]]); the wave kernel is for throughput, so auto takes it for calls of at least 4096 syndromes.
static int osd_wave_shape(const bposd_handle* h, const OsdParams& P, long long B) {
    static const bool on = !(getenv("BPOSD_OSD_WAVE") && getenv("BPOSD_OSD_WAVE")[0] == '0');
    if (!on || h->osd_variant == 1 || P.cost != nullptr || P.dbg != nullptr) return 0;  // switched off; fp64 weights; diagnostics
    if (h->osd_variant == 0 && std::max<long long>(B, h->batch_hint) < 4096) return 0;
    if (P.osd_method == BPOSD_OSD_E && P.osd_order > OSDW_MAX_E) return 0;
    const int m = h->m, n1 = h->n + 1;
    if (m <= 64 && n1 <= 128) return 1;
    if (m <= 128 && n1 <= 256) return 2;
    if (m <= 192 && n1 <= 448) return 3;
    if (m <= 320 && n1 <= 640) return 4;
    // (a seven-rows x fifteen-words instance held 256 VGPRs + 51 AGPRs, one wave per SIMD, and lost to one workgroup per
    // elimination -- round 3, tools/surface_probe.py; removed)
    return 0;
}

// a few waves per elimination, rows in registers (osd_mw_kernel.hip.h): mid-size codes, integer weights
template <int NWV, int RPL, int W, int MINW, bool PACKED>
static int launch_osd_mw_tp(bposd_handle* h, const OsdParams& P, long long B);
template <int NWV, int RPL, int W, int MINW>
static int launch_osd_mw_t(bposd_handle* h, const OsdParams& P, long long B) {
    return P.packed_io ? launch_osd_mw_tp<NWV, RPL, W, MINW, true>(h, P, B) : launch_osd_mw_tp<NWV, RPL, W, MINW, false>(h, P, B);
}
template <int NWV, int RPL, int W, int MINW, bool PACKED>
static int launch_osd_mw_tp(bposd_handle* h, const OsdParams& P, long long B) {
    auto k = osd_mw_kernel<NWV, RPL, W, MINW, PACKED>;
    const size_t lds = osdm_lds_bytes(osdw_nsort(h->n), NWV, RPL, W);
    { int rc_lds = set_max_lds(h, (const void*)k, lds); if (rc_lds) return rc_lds; }
    int wg_per_cu = 1;
    { int rc_occ = cached_occupancy(h, (const void*)k, 64 * NWV, lds, &wg_per_cu); if (rc_occ) return rc_occ; }
    wg_per_cu = std::max(1, std::min(wg_per_cu, 16));
    if (getenv("BPOSD_DEBUG_OCC")) fprintf(stderr, "[bposd] osd_mw_kernel<%d,%d,%d>: %zu B LDS, %d workgroups per CU\n", NWV, RPL, W, lds, wg_per_cu);
    long long grid = std::min<long long>(B, (long long)h->num_cu * wg_per_cu);
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(64 * NWV), lds, h->osd_now ? h->osd_now : h->cur->osd_stream, P);
    HIP_TRY(h, hipGetLastError());
    return 0;
}

// 0 = not this kernel.  Shapes: 1 = 2 waves x 4 rows x 15 words (m <= 512, n <= 959: [[900,36,10]], surface codes d = 19 ... 21),
// 2 = 4 waves x 3 rows x 20 words (m <= 768, n <= 1279: surface d = 23 ... 25), 3 = 8 waves x 2 rows x 31 words (m <= 1024,
// n <= 1983: H1922 -- only when asked for with variant 2), 11 = shape 1 as 4 waves x 2 rows (A/B: BPOSD_OSD_MW_4X2=1)
static int osd_mw_shape(const bposd_handle* h, const OsdParams& P, long long B) {
    static const bool on = !(getenv("BPOSD_OSD_MW") && getenv("BPOSD_OSD_MW")[0] == '0');
    static const bool alt = getenv("BPOSD_OSD_MW_4X2") && getenv("BPOSD_OSD_MW_4X2")[0] == '1';
    static const long long min_batch = getenv("BPOSD_OSD_MW_MIN_BATCH") ? atoll(getenv("BPOSD_OSD_MW_MIN_BATCH")) : 2048;
    if (!on || h->osd_variant == 1 || P.cost != nullptr || P.dbg != nullptr) return 0;  // switched off; fp64 weights; diagnostics
    if (h->osd_variant == 0 && std::max<long long>(B, h->batch_hint) < min_batch) return 0;
    if (P.osd_method == BPOSD_OSD_E && P.osd_order > OSDW_MAX_E) return 0;
    const int m = h->m, n1 = h->n + 1;
    if (m <= 320 && n1 <= 640) return 0;  // the one-wave kernel's
    if (m <= 512 && n1 <= 960) return alt ? 11 : 1;
    if (m <= 768 && n1 <= 1280) return h->osd_variant == 2 ? 2 : 0;  // (measured slower than osd_kernel<24>: surface code d = 25)
    if (m <= 1024 && n1 <= 1984) return h->osd_variant == 2 ? 3 : 0;
    return 0;
}

int launch_osd(bposd_handle* h, const OsdParams& P, long long B) {
    if (const int shp = osd_mw_shape(h, P, B)) {
        h->last_osd_kernel = 4;
        switch (shp) {
            case 1: return launch_osd_mw_t<2, 4, 15, 3>(h, P, B);
            case 11: return launch_osd_mw_t<4, 2, 15, 4>(h, P, B);
            case 2: return launch_osd_mw_t<4, 3, 20, 2>(h, P, B);
            case 3: return launch_osd_mw_t<8, 2, 31, 2>(h, P, B);
        }
    }
    h->last_osd_kernel = osd_wave_shape(h, P, B) ? 2 : 1;
    switch (osd_wave_shape(h, P, B)) {
        case 1: return launch_osd_wave_t<1, 2>(h, P, B);
        case 2: return launch_osd_wave_t<2, 4>(h, P, B);
        case 3: return launch_osd_wave_t<3, 7>(h, P, B);
        case 4: return launch_osd_wave_t<5, 10>(h, P, B);
    }
    switch (osd_words(h->n)) {
        case 1: return launch_osd_t<1>(h, P, B);
        case 2: return launch_osd_t<2>(h, P, B);
        case 4: return launch_osd_t<4>(h, P, B);
        case 8: return launch_osd_t<8>(h, P, B);
        case 16: return launch_osd_t<16>(h, P, B);
        case 24: return launch_osd_t<24>(h, P, B);
        case 31: return launch_osd_t<31>(h, P, B);
        case 32: return launch_osd_t<32>(h, P, B);
    }
    return fail(h, BPOSD_ERR_UNSUPPORTED, "code too large for the register-resident OSD kernel (n=%d)", h->n);
}
}  // namespace bposd_host
