// launch_osd_large.hip -- osd_large_kernel (matrix in HBM): workspace carving and launch
// One translation unit of libbposd_mi355x.so: the kernels of this family are instantiated here and nowhere else.
#include "internal.h"

#include "osd_large_kernel.hip.h"

using namespace bposd;
using namespace bposd_host;

namespace bposd_host {
static int osdl_rpt(int m) {
    for (int r : {2, 4, 8, 16})
        if (m <= OSDL_NT * r) return r;
    return 0;
}

int launch_osd_large(bposd_handle* h, const OsdParams& P, long long B, int* d_rank_out) {
    const int RPT = osdl_rpt(h->m);
    if (!RPT) return fail(h, BPOSD_ERR_UNSUPPORTED, "m=%d beyond the HBM-resident OSD kernel (16384)", h->m);
    OsdLargeParams Q{};
    Q.m = h->m; Q.n = h->n; Q.W = (h->n + 1 + 63) / 64;
    Q.rank = P.rank; Q.osd_method = P.osd_method; Q.osd_order = P.osd_order; Q.tie_policy = P.tie_policy; Q.e_msb_first = P.e_msb_first;
    Q.nsort = 1;
    while (Q.nsort < h->n) Q.nsort <<= 1;
    Q.mrl = OSDL_NT * RPT;
    Q.synd = P.synd; Q.rp = P.rp; Q.ci = P.ci; Q.llr_ws = P.llr_ws; Q.osd_list = P.osd_list; Q.counters = P.counters;
    Q.packed_io = P.packed_io; Q.out_osd0 = P.out_osd0; Q.out_osdw = P.out_osdw; Q.cmp_osd0 = P.cmp_osd0; Q.cmp_osdw = P.cmp_osdw; Q.rank_out = d_rank_out; Q.dbg = P.dbg;
    // fp64 index-order candidate weights (non-uniform channel) -- only OSD-E / OSD-CS rank candidates
    const bool fpw = P.cost != nullptr && Q.osd_method >= BPOSD_OSD_E && Q.osd_order > 0;
    Q.cost = fpw ? P.cost : nullptr; Q.sel = fpw ? P.sel : nullptr; Q.cost_alt = P.cost_alt;
    Q.wdn = std::max(64 * Q.W, 1 << OSDL_MAXSPAN);
    long long grid = std::min<long long>(B, h->num_cu);
    if (grid < 1) grid = 1;
    auto a256 = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const bool wide_cs = Q.osd_method == BPOSD_OSD_CS && Q.osd_order > OSDL_MAXSPAN;
    const bool wide_fp = wide_cs && fpw;  // fp64 weights over a pair span beyond 16: 64-bit column words per row / per bit in HBM
    const bool gauss = Q.osd_method != BPOSD_OSD_CS;  // Gaussian elimination + back-substitution: every pivot group keeps its rows
    Q.pro_stride = gauss ? osd_large_pro_rows(Q.W) * 64 : (size_t)OSDL_K * Q.W * 64;
    size_t sizes[20];
    // every sub-array is [grid][count], laid out back to back in one allocation
    auto layout = [&](size_t g) {
        const size_t sz[20] = {g * sizeof(unsigned long long) * (size_t)Q.W * Q.mrl,  // mat
                               g * sizeof(unsigned long long) * (size_t)Q.nsort,     // keys
                               g * sizeof(int) * (size_t)Q.nsort,                    // kidx
                               g * sizeof(int) * (size_t)h->n,                       // inv
                               g * sizeof(int) * (size_t)64 * Q.W,                   // pivrow
                               g * sizeof(int) * (size_t)Q.mrl,                      // rowpos
                               g * sizeof(int) * (size_t)64 * Q.W,                   // wt
                               g * (size_t)h->n,                                     // xout
                               g * sizeof(unsigned long long) * (size_t)OSDL_K * Q.mrl,        // tmo
                               g * sizeof(unsigned long long) * Q.pro_stride,                  // pro
                               fpw ? g * sizeof(double) * (size_t)h->n : 0,                    // costs_ws
                               fpw ? g * sizeof(double) * (size_t)Q.wdn : 0,                   // wd_ws
                               fpw ? g * sizeof(unsigned short) * (size_t)Q.mrl : 0,           // am_ws
                               g * sizeof(int) * (size_t)Q.mrl,                                // alist
                               wide_cs ? g * sizeof(unsigned long long) * (size_t)OSDL_MAXSPAN_CS * RPT * OSDL_NW : 0,  // colvec_ws
                               wide_fp ? g * sizeof(unsigned long long) * (size_t)Q.mrl : 0,                   // am64_ws
                               wide_fp ? g * sizeof(unsigned long long) * (size_t)h->n : 0,                    // cm64_ws
                               gauss ? g * sizeof(unsigned long long) * (size_t)64 * Q.W : 0,                  // pmask
                               g * sizeof(unsigned long long) * (size_t)Q.mrl,                                 // cnz
                               g * sizeof(unsigned long long) * (size_t)OSDL_K * 64};                          // gcnz
        size_t t = 0;
        for (int i = 0; i < 20; ++i) { sizes[i] = sz[i]; t += a256(sz[i]); }
        return t;
    };
    size_t total = layout((size_t)grid);
    // The workspace is allocated on every lane of the handle (the packed matrix plus, in Gaussian mode, the kept pivot rows of
    // every group: 61 + 54.5 MB per workgroup on the 14520 x 29524 code = 29.6 GB per lane at 256 workgroups).  Where the device
    // cannot hold that, fewer workgroups share the queue instead of a hipMalloc failing in the middle of a decode.
    if (total > h->lanes[0].osdl_ws.bytes) {
        size_t free_b = 0, total_b = 0;
        HIP_TRY(h, hipMemGetInfo(&free_b, &total_b));
        size_t have = 0;
        for (int l = 0; l < h->nlanes; ++l) have += h->lanes[l].osdl_ws.bytes;   // (released before the larger block is allocated)
        const size_t budget = (size_t)((double)(free_b + have) * 0.92);
        const size_t per_wg = layout(1) * (size_t)h->nlanes;
        if (per_wg > budget)
            return fail(h, BPOSD_ERR_HIP, "the HBM-resident OSD kernel needs %.2f GB of workspace per workgroup (%d lanes); %.2f GB are free", per_wg * 1e-9,
                        h->nlanes, free_b * 1e-9);
        if (total * (size_t)h->nlanes > budget) grid = std::max<long long>(1, (long long)(budget / per_wg));
        total = layout((size_t)grid);
    }
    int rc = ensure_lanes(h, &Lane::osdl_ws, total);
    if (rc) return rc;
    unsigned char* ptrs[20];
    {
        unsigned char* base = (unsigned char*)h->cur->osdl_ws.p;
        for (int i = 0; i < 20; ++i) { ptrs[i] = base; base += a256(sizes[i]); }
    }
    Q.alist = (int*)ptrs[13];
    Q.colvec_ws = wide_cs ? (unsigned long long*)ptrs[14] : nullptr;
    Q.am64_ws = wide_fp ? (unsigned long long*)ptrs[15] : nullptr;
    Q.cm64_ws = wide_fp ? (unsigned long long*)ptrs[16] : nullptr;
    Q.costs_ws = (double*)ptrs[10];
    Q.wd_ws = (double*)ptrs[11];
    Q.am_ws = (unsigned short*)ptrs[12];
    Q.mat = (unsigned long long*)ptrs[0];
    Q.keys = (unsigned long long*)ptrs[1];
    Q.kidx = (int*)ptrs[2];
    Q.inv = (int*)ptrs[3];
    Q.pivrow = (int*)ptrs[4];
    Q.rowpos = (int*)ptrs[5];
    Q.wt = (int*)ptrs[6];
    Q.xout = (uint8_t*)ptrs[7];
    Q.tmo = (unsigned long long*)ptrs[8];
    Q.pro = (unsigned long long*)ptrs[9];
    Q.pmask = gauss ? (unsigned long long*)ptrs[17] : nullptr;
    Q.cnz = (unsigned long long*)ptrs[18];
    Q.gcnz = (unsigned long long*)ptrs[19];
    const size_t lds = osd_large_lds_bytes(Q.W, RPT, fpw ? h->n : 0);
    if (lds > h->lds_per_cu) return fail(h, BPOSD_ERR_UNSUPPORTED, "large OSD kernel needs %zu bytes of LDS", lds);
#define OSDL_LAUNCH(R)                                                                                      \
    case R: {                                                                                               \
        auto k = osd_large_kernel<R>;                                                                       \
        { int rc_lds = set_max_lds(h, (const void*)k, lds); if (rc_lds) return rc_lds; } \
        hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(OSDL_NT), lds, h->osd_now ? h->osd_now : h->cur->osd_stream, Q); \
    } break;
    switch (RPT) {
        OSDL_LAUNCH(2)
        OSDL_LAUNCH(4)
        OSDL_LAUNCH(8)
        OSDL_LAUNCH(16)
    }
#undef OSDL_LAUNCH
    HIP_TRY(h, hipGetLastError());
    return 0;
}

int osd_large_maxspan(bool cs) { return cs ? OSDL_MAXSPAN_CS : OSDL_MAXSPAN; }

}  // namespace bposd_host
