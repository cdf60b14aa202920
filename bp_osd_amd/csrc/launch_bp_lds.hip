// launch_bp_lds.hip -- bp_kernel (every message in LDS; irregular codes, product-sum on H1922): shapes and launch
// One translation unit of libbposd_mi355x.so: the kernels of this family are instantiated here and nowhere else.
#include "internal.h"

#include <climits>

using namespace bposd;
using namespace bposd_host;

namespace bposd_host {
template <int DC, int DV, int CPT, int VPT, int MAXNT, int MINW, bool REG, int MPT>
static int launch_bp_t(bposd_handle* h, const BpParams& P, int NT) {
    const size_t lds = bp_lds_bytes(DC, P.mp);
    int wg_per_cu = (int)std::min<size_t>(h->lds_per_cu / lds, (size_t)(2048 / NT));
    wg_per_cu = std::max(1, std::min(wg_per_cu, 8));
    long long grid = std::min<long long>(P.B, (long long)h->num_cu * wg_per_cu);
    if (grid < 1) grid = 1;
    if (h->cfg.bp_method == BPOSD_BP_MIN_SUM) {
        auto k = bp_kernel<DC, DV, CPT, VPT, MAXNT, MINW, REG, 1, MPT>;
        { int rc_lds = set_max_lds(h, (const void*)k, lds); if (rc_lds) return rc_lds; }
        hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(NT), lds, h->cur->stream, P);
    } else if (h->cfg.ps_math_form) {  // product-sum, two divisions per edge
        auto k = bp_kernel<DC, DV, CPT, VPT, MAXNT, MINW, REG, 0, MPT>;
        { int rc_lds = set_max_lds(h, (const void*)k, lds); if (rc_lds) return rc_lds; }
        hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(NT), lds, h->cur->stream, P);
    } else {                            // product-sum in the reference's operation order (the default)
        auto k = bp_kernel<DC, DV, CPT, VPT, MAXNT, MINW, REG, 2, MPT>;
        { int rc_lds = set_max_lds(h, (const void*)k, lds); if (rc_lds) return rc_lds; }
        hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(NT), lds, h->cur->stream, P);
    }
    HIP_TRY(h, hipGetLastError());
    return 0;
}

static int pow2_at_least(int x) {
    int p = 64;
    while (p < x) p <<= 1;
    return p;
}

// shape id: 1 -> (CPT 1, VPT 2, <=1024 threads), 2 -> (2, 4, <=512), 4 -> (4, 8, <=256).
// Threads per workgroup are a power of two so that the check stride MP = threads * CPT is one.
// The regular (6,3) kernels are compiled for MP = 1024 exactly (H1922: 961 checks, 1922 bits).
bool is_reg63(const bposd_handle* h) { return h->regular && h->dc_max == 6 && h->dv_max == 3; }

// shape 8 = "mid-size": 2 checks / 4 bits per thread with 1024 threads (1024 < m <= 2048, n <= 4096): one workgroup per CU
int shape_cpt(int shape) { return shape == 8 ? 2 : shape; }

int shape_threads(const bposd_handle* h, int shape) {
    const int cpt = shape_cpt(shape), vpt = 2 * cpt;
    int nt = pow2_at_least(std::max((h->m + cpt - 1) / cpt, (h->n + vpt - 1) / vpt));
    if (is_reg63(h) && shape != 8 && nt <= 1024 / shape) nt = 1024 / shape;
    return nt;
}

int pick_shape(const bposd_handle* h) {
    const int caps[3][2] = {{1, 1024}, {2, 512}, {4, 256}};
    if (h->bp_variant) {
        for (auto& c : caps)
            if (c[0] == h->bp_variant && shape_threads(h, c[0]) <= c[1] && (c[0] != 4 || is_reg63(h))) return c[0];
    }
    if (is_reg63(h) && shape_threads(h, 2) <= 512) return 2;
    // generic kernels: one check per thread when that fits, else two
    if (shape_threads(h, 1) <= 1024) return 1;
    if (shape_threads(h, 2) <= 512) return 2;
    if (shape_threads(h, 8) <= 1024) return 8;
    return 0;
}

template <int DC, int DV, bool REG>
static int launch_bp_shape(bposd_handle* h, const BpParams& P, int shape, int NT) {
    // occupancy targets: LDS admits 3 workgroups per CU for H1922 (46 KB each); the regular
    // (6,3) kernels are register-capped for that (2 x 1024, 3 x 512 or 3 x 256 threads per CU)
    if (shape == 1) return launch_bp_t<DC, DV, 1, 2, 1024, (REG ? 8 : 4), REG, (REG ? 1024 : 0)>(h, P, NT);
#ifndef BPOSD_SHAPE2_MINW
#define BPOSD_SHAPE2_MINW 6
#endif
    if (shape == 2) return launch_bp_t<DC, DV, 2, 4, 512, (REG ? BPOSD_SHAPE2_MINW : 2), REG, (REG ? 1024 : 0)>(h, P, NT);
    if constexpr (REG) {
        if (shape == 4) return launch_bp_t<DC, DV, 4, 8, 256, 3, REG, 1024>(h, P, NT);
    } else {
        if (shape == 8) return launch_bp_t<DC, DV, 2, 4, 1024, 4, false, 0>(h, P, NT);
    }
    return fail(h, BPOSD_ERR_UNSUPPORTED, "no BP kernel shape %d for this code", shape);
}

int launch_bp(bposd_handle* h, BpParams& P) {
    int shape = pick_shape(h);
    if (!shape) return fail(h, BPOSD_ERR_UNSUPPORTED, "code too large for the LDS-resident BP kernel (m=%d n=%d)", h->m, h->n);
    const int NT = shape_threads(h, shape);
    const int MP = NT * shape_cpt(shape);
    const int NPOS = NT * 2 * shape_cpt(shape);
    if (MP != h->tab_mp || NPOS != h->tab_np) {
        { int rcs = sync_all_lanes(h); if (rcs) return rcs; }  // kernels in flight still read the old tables
        int rc = build_tables(h, h->tab_dc, h->tab_dv, MP, NT, 2 * shape_cpt(shape));
        if (rc) return rc;
        P.chk_deg = h->d_chk_deg;
        P.var_deg = h->d_var_deg;
        P.var_pos = h->d_var_pos;
        P.pos_bit = h->d_pos_bit;
    }
    P.mp = MP;
    P.np = NPOS;
    if (bp_lds_bytes(h->tab_dc, MP) > h->lds_per_cu)
        return fail(h, BPOSD_ERR_UNSUPPORTED, "BP messages (%zu B) exceed one CU's LDS", bp_lds_bytes(h->tab_dc, MP));
    if (is_reg63(h) && MP == 1024) return launch_bp_shape<6, 3, true>(h, P, shape, NT);
    switch (h->tab_dc) {
        case 4: return launch_bp_shape<4, 2, false>(h, P, shape, NT);
        case 6: return launch_bp_shape<6, 3, false>(h, P, shape, NT);
        case 8: return launch_bp_shape<8, 4, false>(h, P, shape, NT);
        case 12: return launch_bp_shape<12, 6, false>(h, P, shape, NT);
        case 16: return launch_bp_shape<16, 8, false>(h, P, shape, NT);
    }
    return fail(h, BPOSD_ERR_UNSUPPORTED, "no BP kernel for check degree %d / bit degree %d", h->dc_max, h->dv_max);
}
}  // namespace bposd_host
