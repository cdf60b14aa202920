// bposd_capi.hip -- host side of libbposd_mi355x.so: the C-ABI declared in
// include/bposd_mi355x.h, Tanner-graph table construction, kernel dispatch, workspace
// and HIP-event timing.  gfx950 only; there is no CPU fallback anywhere in this file:
// without a HIP device every entry point fails with BPOSD_ERR_NO_DEVICE.
//
// Reference interface replaced: the `bposd_decoder` / `BpOsdDecoder` object of the
// third-party `ldpc` package as used at /root/reference/README.md:178-202 and
// /root/reference/src/bposd/css_decode_sim.py:444-463,174-202.
#include "../../include/bposd_mi355x.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <climits>
#include <cstring>
#include <string>
#include <tuple>
#include <functional>
#include <map>
#include <chrono>
#include <mutex>
#include <thread>
#include <vector>

#include "internal.h"
#include <array>
#include <vector>
#include <algorithm>
#include "local_layout.h"
#include "class_layout.h"

using namespace bposd;
using namespace bposd_host;

namespace {

thread_local std::string g_create_error;

}  // namespace

namespace bposd_host {

int fail(bposd_handle* h, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (h) h->err = buf;
    else g_create_error = buf;
    return code;
}
// Every entry point works on the handle's device and puts the caller's current device back on exit (a process that
// also drives torch, or handles on other GPUs, must not find its thread's device changed by a decode call).
struct DeviceGuard {
    int prev = -1, dev = -1;
    hipError_t err = hipSuccess;
    explicit DeviceGuard(int d) : dev(d) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) err = hipSetDevice(dev);
    }
    ~DeviceGuard() {
        if (prev >= 0 && prev != dev) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};
int sync_all_lanes(bposd_handle* h) {
    for (auto& l : h->lanes) {
        if (l.stream) HIP_TRY(h, hipStreamSynchronize(l.stream));
        if (l.osd_stream) HIP_TRY(h, hipStreamSynchronize(l.osd_stream));
        if (l.copy_stream) HIP_TRY(h, hipStreamSynchronize(l.copy_stream));
    }
    h->async_pending = false;
    return 0;
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a property of the kernel on a device, shared by every handle: it is
// only ever raised (a handle that needs less launches fine under a larger limit), and the API is called only when it
// has to be -- it costs microseconds on the one-syndrome path.
int set_max_lds(bposd_handle* h, const void* kernel, size_t lds) {
    static std::mutex mu;
    static std::map<std::pair<int, const void*>, size_t> limit;
    std::lock_guard<std::mutex> lock(mu);
    size_t& cur = limit[{h->device, kernel}];
    if (lds <= cur) return 0;
    HIP_TRY(h, hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    cur = lds;
    return 0;
}

// Resident workgroups per CU of a kernel at a workgroup size and dynamic-LDS size: asked of the runtime once per process,
// device and configuration (the query costs microseconds on the one-syndrome path).
int cached_occupancy(bposd_handle* h, const void* kernel, int nt, size_t lds, int* out) {
    static std::mutex mu;
    static std::map<std::tuple<int, const void*, int, size_t>, int> memo;
    std::lock_guard<std::mutex> lock(mu);
    auto key = std::make_tuple(h->device, kernel, nt, lds);
    auto it = memo.find(key);
    if (it == memo.end()) {
        int v = 1;
        HIP_TRY(h, hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, kernel, nt, lds));
        it = memo.emplace(key, std::max(v, 1)).first;
    }
    *out = it->second;
    return 0;
}

int ensure(bposd_handle* h, DevBuf& b, size_t bytes) {
    if (bytes <= b.bytes && b.p) return 0;
    if (b.p) {
        HIP_TRY(h, hipFree(b.p));
        b.p = nullptr;
        b.bytes = 0;
    }
    size_t want = std::max<size_t>(bytes, 256);
    HIP_TRY(h, hipMalloc(&b.p, want));
    b.bytes = want;
    return 0;
}

// Workspaces are per lane; when one has to grow, the same buffer of every lane this handle cycles through grows with it,
// so that the allocation (and the page mapping behind it) is paid by the first call of a size class, not by the first
// call that happens to land on each lane.
int ensure_lanes(bposd_handle* h, DevBuf Lane::*member, size_t bytes) {
    for (int l = 0; l < h->nlanes; ++l) {
        int rc = ensure(h, h->lanes[l].*member, bytes);
        if (rc) return rc;
    }
    return 0;
}

void release(DevBuf& b) {
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.bytes = 0;
}
// GF(2) rank of the pcm by packed elimination (ctor-time, host).  a1: upstream's ctor
// eliminates H once to learn rank and k' = n - rank.
int gf2_rank_host(int m, int n, const std::vector<int>& rp, const std::vector<int>& ci) {
    const int W = (n + 63) / 64;
    std::vector<uint64_t> a((size_t)m * W, 0);
    for (int r = 0; r < m; ++r)
        for (int e = rp[r]; e < rp[r + 1]; ++e) a[(size_t)r * W + (ci[e] >> 6)] |= 1ull << (ci[e] & 63);
    int rank = 0;
    for (int j = 0; j < n && rank < m; ++j) {
        const int w = j >> 6;
        const uint64_t bit = 1ull << (j & 63);
        int p = -1;
        for (int r = rank; r < m; ++r)
            if (a[(size_t)r * W + w] & bit) { p = r; break; }
        if (p < 0) continue;
        if (p != rank)
            for (int x = 0; x < W; ++x) std::swap(a[(size_t)p * W + x], a[(size_t)rank * W + x]);
        for (int r = rank + 1; r < m; ++r)
            if (a[(size_t)r * W + w] & bit)
                for (int x = w; x < W; ++x) a[(size_t)r * W + x] ^= a[(size_t)rank * W + x];
        ++rank;
    }
    return rank;
}

struct DegPair { int dc, dv; };
const DegPair kPairs[] = {{4, 2}, {6, 3}, {8, 4}, {12, 6}, {16, 8}};

bool pick_pair(int dc, int dv, DegPair* out) {
    for (const auto& p : kPairs)
        if (p.dc >= dc && p.dv >= dv) { *out = p; return true; }
    return false;
}

int upload_priors(bposd_handle* h) {
    // a3: prior LLR = log((1 - p) / p), evaluated on the host in fp64 (same libm call the
    // CPU path makes) so that device arithmetic is add / compare / multiply only.
    std::vector<double> l0(h->n);
    for (int i = 0; i < h->n; ++i) l0[i] = std::log((1 - h->probs[i]) / h->probs[i]);
    HIP_TRY(h, hipMemcpy(h->d_llr0, l0.data(), sizeof(double) * h->n, hipMemcpyHostToDevice));
    h->probs_uniform = true;
    for (int i = 1; i < h->n; ++i)
        if (h->probs[i] != h->probs[0]) { h->probs_uniform = false; break; }
    // a11: weight(x) = sum over set bits of log(1/p_i) (ldpc v2).  For a uniform 0 < p < 1 every term is
    // the same positive number, so the sums order candidates exactly like Hamming weights (identical
    // partial sums, strictly increasing in the count) and the integer path is used.
    std::vector<double> cost(h->n);
    for (int i = 0; i < h->n; ++i) cost[i] = std::log(1 / h->probs[i]);
    HIP_TRY(h, hipMemcpy(h->d_cost, cost.data(), sizeof(double) * h->n, hipMemcpyHostToDevice));
    h->fp_weights = (h->cfg.weight_fn == 0) &&
                    !(h->probs_uniform && h->probs[0] > 0.0 && h->probs[0] < 1.0);
    return 0;
}
// ---------------------------------------------------------------------------------------------
// Bit-pass layout.  The check pass is bank-conflict free by construction (lane c <-> slot k*MP + c).
// The bit pass gathers/scatters slot (k*MP + c) for the d-th edge of each of 64 lanes; its conflicts
// depend only on which bits share a 32-lane (ds_read_b64: 64 banks) / 16-lane (ds_write_b64: 32 banks)
// group.  The order of bits over lanes is free (tables are position-indexed), so the host simulates
// the LDS cycles (MI355X_MICROARCH.md §LDS banking model) of a family of orders -- natural, and
// two-block orders where each block of (outer x inner) bits is laid out inner-major or outer-major with
// groups padded to a multiple of 32 lanes (the shapes hypergraph-product codes have) -- and keeps the
// cheapest.  For H1922 (31x31 | 31x31) the outer-major order of the first block is conflict free.
struct EdgeSlot { int slot; };

static long bit_pass_cycles(const std::vector<int>& bit_of_pos, int NP, int NT, int VPT, int dv_max,
                            const std::vector<int>& cptr, const std::vector<int>& eslot, long stop_at) {
    long total = 0;
    int cnt[64];
    int first[64];
    for (int r = 0; r < VPT; ++r) {
        for (int w0 = 0; w0 < NT; w0 += 64) {
            for (int d = 0; d < dv_max; ++d) {
                int slots[64];
                bool any = false;
                for (int l = 0; l < 64; ++l) {
                    const int p = r * NT + w0 + l;
                    const int i = p < NP ? bit_of_pos[p] : -1;
                    slots[l] = (i >= 0 && cptr[i] + d < cptr[i + 1]) ? eslot[cptr[i] + d] : -1;
                    any |= slots[l] >= 0;
                }
                if (!any) continue;
                // reads: two 32-lane groups, an 8-byte access covers banks 2*slot, 2*slot+1 of 64
                for (int g = 0; g < 64; g += 32) {
                    int worst = 0;
                    for (int b = 0; b < 32; ++b) { cnt[b] = 0; first[b] = -1; }
                    for (int l = g; l < g + 32; ++l) {
                        if (slots[l] < 0) continue;
                        const int b = slots[l] & 31;
                        // distinct addresses on the same bank serialise (identical ones broadcast; cannot
                        // happen here: every edge has its own slot)
                        ++cnt[b];
                        worst = std::max(worst, cnt[b]);
                    }
                    total += std::max(worst, 1);
                }
                // writes: four 16-lane groups, 32 banks -> 16 slot classes
                for (int g = 0; g < 64; g += 16) {
                    int worst = 0;
                    for (int b = 0; b < 16; ++b) cnt[b] = 0;
                    for (int l = g; l < g + 16; ++l) {
                        if (slots[l] < 0) continue;
                        const int b = slots[l] & 15;
                        ++cnt[b];
                        worst = std::max(worst, cnt[b]);
                    }
                    total += std::max(worst, 1);
                }
                if (total >= stop_at) return total;
            }
        }
    }
    return total;
}

static int round32(int x) { return (x + 31) / 32 * 32; }

// positions of a block of `count` bits starting at bit `b0`, viewed as outer x inner with the given inner
// size, laid out inner-major (transposed = false) or outer-major (transposed = true), groups padded to 32
static int place_block(std::vector<int>& bit_of_pos, int p0, int b0, int count, int inner, bool transposed, int NP) {
    if (count == 0) return p0;
    if (inner <= 0 || count % inner != 0) return -1;
    const int outer = count / inner;
    const int gsz = transposed ? round32(outer) : round32(inner);
    const int ngr = transposed ? inner : outer;
    if ((long)p0 + (long)gsz * ngr > NP) return -1;
    for (int a = 0; a < outer; ++a)
        for (int b = 0; b < inner; ++b) {
            const int p = transposed ? p0 + b * gsz + a : p0 + a * gsz + b;
            bit_of_pos[p] = b0 + a * inner + b;
        }
    return p0 + gsz * ngr;
}

static void choose_bit_layout(bposd_handle* h, int MP, int NT, int VPT, std::vector<int>& best_bit_of_pos) {
    const int n = h->n, NP = NT * VPT;
    // CSC view with the LDS slot of every edge
    std::vector<int> cptr(n + 1, 0), fill(n, 0);
    for (int e = 0; e < h->E; ++e) cptr[h->ci[e] + 1]++;
    for (int i = 0; i < n; ++i) cptr[i + 1] += cptr[i];
    std::vector<int> eslot(h->E);
    for (int c = 0; c < h->m; ++c)
        for (int e = h->rp[c]; e < h->rp[c + 1]; ++e) {
            const int i = h->ci[e];
            eslot[cptr[i] + fill[i]++] = (e - h->rp[c]) * MP + c;  // ascending row within a column
        }
    std::vector<int> cand(NP, -1);
    for (int i = 0; i < n; ++i) cand[i] = i;
    long best = bit_pass_cycles(cand, NP, NT, VPT, h->dv_max, cptr, eslot, LONG_MAX);
    best_bit_of_pos = cand;
    h->layout_cost_natural = best;
    // ideal: every instruction that touches a real bit costs 2 read + 4 write group-cycles
    long ninstr = 0;
    for (int r = 0; r < VPT; ++r)
        for (int w0 = 0; w0 < NT; w0 += 64)
            if (r * NT + w0 < n) ninstr += h->dv_max;
    h->layout_cost_ideal = ninstr * 6;
    if (best <= h->layout_cost_ideal + h->layout_cost_ideal / 20) { h->layout_cost = best; return; }
    // two-block family: bits [0, s) as (s/p1 x p1), bits [s, n) as ((n-s)/p2 x p2)
    for (int p1 = 2; p1 <= 64; ++p1) {
        for (int o1 = 0; o1 <= 64 && o1 * p1 <= n; ++o1) {
            const int s0 = o1 * p1;
            const int rest = n - s0;
            for (int p2 = 2; p2 <= 64; ++p2) {
                if (rest % p2 != 0 || rest / p2 > 64) continue;
                if (s0 == 0 && p1 != 2) continue;  // a single block: p1 is irrelevant, visit once
                for (int t = 0; t < 4; ++t) {
                    std::fill(cand.begin(), cand.end(), -1);
                    int q = place_block(cand, 0, 0, s0, p1, (t & 1) != 0, NP);
                    if (q < 0) continue;
                    q = place_block(cand, q, s0, rest, p2, (t & 2) != 0, NP);
                    if (q < 0) continue;
                    const long c = bit_pass_cycles(cand, NP, NT, VPT, h->dv_max, cptr, eslot, best);
                    if (c < best) { best = c; best_bit_of_pos = cand; }
                }
            }
        }
    }
    h->layout_cost = best;
}

int build_tables(bposd_handle* h, int DC, int DV, int MP, int NT, int VPT) {
    // LDS slot of the k-th edge of check c is k * MP + c (MP = checks padded to threads x CPT)
    const int m = h->m, n = h->n, NP = NT * VPT;
    std::vector<int> bit_of_pos;
    choose_bit_layout(h, MP, NT, VPT, bit_of_pos);
    std::vector<int> pos_of_bit(n, -1);
    for (int p = 0; p < NP; ++p)
        if (bit_of_pos[p] >= 0) pos_of_bit[bit_of_pos[p]] = p;
    std::vector<int> chk_deg(m), var_deg(NP, 0);
    std::vector<int> var_pos((size_t)DV * NP, 0);
    for (int c = 0; c < m; ++c) {
        chk_deg[c] = h->rp[c + 1] - h->rp[c];
        for (int e = h->rp[c]; e < h->rp[c + 1]; ++e) {
            const int k = e - h->rp[c];
            const int p = pos_of_bit[h->ci[e]];
            const int d = var_deg[p]++;  // rows visited ascending => ascending row within a column
            var_pos[(size_t)d * NP + p] = k * MP + c;
        }
    }
    auto up = [&](int** dst, const std::vector<int>& v) -> int {
        if (*dst) { (void)hipFree(*dst); *dst = nullptr; }
        HIP_TRY(h, hipMalloc((void**)dst, sizeof(int) * std::max<size_t>(v.size(), 1)));
        HIP_TRY(h, hipMemcpy(*dst, v.data(), sizeof(int) * v.size(), hipMemcpyHostToDevice));
        return 0;
    };
    int rc;
    if ((rc = up(&h->d_chk_deg, chk_deg))) return rc;
    if ((rc = up(&h->d_var_deg, var_deg))) return rc;
    if ((rc = up(&h->d_var_pos, var_pos))) return rc;
    if ((rc = up(&h->d_pos_bit, bit_of_pos))) return rc;
    h->tab_dc = DC;
    h->tab_dv = DV;
    h->tab_mp = MP;
    h->tab_np = NP;
    return 0;
}
int build_tables_local(bposd_handle* h) {
    using namespace local_layout;
    h->local_ok = false;
    const int m = h->m, n = h->n;
    if (!(h->regular && h->dc_max == 6 && h->dv_max == 3 && n == 2 * m)) return 0;
    const int MP = m <= 1024 ? 1024 : 2048;  // the kernels are compiled for 1024 (H1922: 961 checks) and 2048 positions
    if (m > MP) return 0;
    Graph g;
    Layout best;
    if (!local_layout_host(h->rp, h->ci, m, n, MP, g, best)) return 0;
    if (getenv("BPOSD_DEBUG_OCC"))
        fprintf(stderr, "[bposd] local-edge layout: bit pass %lld read cycles (floor %d) + %lld write cycles (floor %d), %d mixed pairs, %d uniform positions\n",
                best.passes, 4 * (MP / 32), best.wcycles, 6 * 4 * (MP / 64), best.mixed, best.nfull);
    h->local_passes = best.passes;
    h->local_wcycles = best.wcycles;

    // ---- tables
    const int G = MP / 64;
    const std::vector<int>&owner = best.owner, &load = best.load, &pos_of = best.pos_of, &pos_chk = best.pos_chk;
    std::vector<int> grp_dl(2 * (size_t)G, 0);
    for (int gq = 0; gq < G; ++gq)
        for (int b = 0; b < 2; ++b) {
            int code = -1;
            for (int p = 64 * gq; p < 64 * gq + 64; ++p) {
                const int c = pos_chk[p];
                if (c < 0) continue;
                const int d = g.rank_of(load[2 * c + b], c);
                code = (code < 0 || code == d) ? d : 3;
            }
            grp_dl[(size_t)b * G + gq] = code < 0 ? 0 : code;
        }
    // LDS slot of (check c, bit i) for the check's four non-local edges, ascending column order
    auto slot_of = [&](int c, int i) {
        int k = 0;
        for (int e = h->rp[c]; e < h->rp[c + 1]; ++e) {
            const int j = h->ci[e];
            if (owner[j] == c) continue;
            if (j == i) return k * MP + pos_of[c];
            ++k;
        }
        return -1;
    };
    // Padding positions (pos_chk < 0): the two "bits" of such a position are wired to the position's own four LDS slots
    // (slot k * MP + p), a closed toy graph that needs no predicate in the kernel (bp_local_kernel.hip.h).
    std::vector<int> pos_bit(2 * (size_t)MP, -1), pos_alo(2 * (size_t)MP, 0), pos_ahi(2 * (size_t)MP, 0), pos_dl(2 * (size_t)MP, 0);
    for (int p = 0; p < MP; ++p)
        for (int b = 0; b < 2; ++b) {
            pos_alo[(size_t)b * MP + p] = (2 * b) * MP + p;
            pos_ahi[(size_t)b * MP + p] = (2 * b + 1) * MP + p;
        }
    for (int c = 0; c < m; ++c) {
        const int p = pos_of[c];
        for (int b = 0; b < 2; ++b) {
            const int i = load[2 * c + b];
            int o[2];
            g.others(i, c, o);
            const int sx = slot_of(o[0], i), sy = slot_of(o[1], i);
            if (sx < 0 || sy < 0) return 0;
            pos_dl[(size_t)b * MP + p] = g.rank_of(i, c);
            pos_bit[(size_t)b * MP + p] = i;
            pos_alo[(size_t)b * MP + p] = sx;
            pos_ahi[(size_t)b * MP + p] = sy;
        }
    }
    auto up = [&](int** dst, const std::vector<int>& v) -> int {
        if (*dst) { (void)hipFree(*dst); *dst = nullptr; }
        HIP_TRY(h, hipMalloc((void**)dst, sizeof(int) * std::max<size_t>(v.size(), 1)));
        HIP_TRY(h, hipMemcpy(*dst, v.data(), sizeof(int) * v.size(), hipMemcpyHostToDevice));
        return 0;
    };
    int rc;
    if ((rc = up(&h->d_lpos_chk, pos_chk))) return rc;
    if ((rc = up(&h->d_lpos_bit, pos_bit))) return rc;
    if ((rc = up(&h->d_lpos_alo, pos_alo))) return rc;
    if ((rc = up(&h->d_lpos_ahi, pos_ahi))) return rc;
    if ((rc = up(&h->d_lgrp_dl, grp_dl))) return rc;
    if ((rc = up(&h->d_lpos_dl, pos_dl))) return rc;
    h->local_mp = MP;
    h->local_ok = true;
    return 0;
}
// ------------------------------------------------------------------ class BP kernel: tables + launch
// Instances: (check degrees; bit degrees) = (7; 3..4) -- the reference's three example codes --, (6; 3) -- H1922 with
// product-sum, other (3,6)-regular codes --, (4; 2) -- toric codes, hgp(ring_code) --, (8; 4), and (3..4; 1..2) -- surface
// codes, hgp(rep_code) --; LDS stride 256 / 512 / 1024, two bit slots per thread.
struct ClassShape { int dclo, dc, dvlo, dvhi; };
const ClassShape kClassShapes[] = {{7, 7, 3, 4}, {6, 6, 3, 3}, {4, 4, 2, 2}, {8, 8, 4, 4}, {3, 4, 1, 2}};
constexpr int kClassVPT = 2;

// the first instance whose degree ranges cover the code's, or null
const ClassShape* class_shape_for(const std::vector<int>& rp, const std::vector<int>& ci, int m, int n) {
    int clo = 1 << 30, chi = 0, lo = 1 << 30, hi = 0;
    for (int c = 0; c < m; ++c) {
        const int d = rp[c + 1] - rp[c];
        clo = std::min(clo, d); chi = std::max(chi, d);
    }
    std::vector<int> vdeg(n, 0);
    for (int e : ci) vdeg[e]++;
    for (int d : vdeg) { lo = std::min(lo, d); hi = std::max(hi, d); }
    for (const auto& k : kClassShapes)
        if (k.dclo <= clo && chi <= k.dc && k.dvlo <= lo && hi <= k.dvhi) return &k;
    return nullptr;
}

int build_tables_class(bposd_handle* h) {
    h->class_ok = false;
    if (h->bp_hbm || h->m > 1024) return 0;
    const ClassShape* shp = class_shape_for(h->rp, h->ci, h->m, h->n);
    if (!shp) return 0;
    class_layout::Tables T;
    bool ok = false;
    int MP = 0;
    const int iters = getenv("BPOSD_LAYOUT_ITERS") ? atoi(getenv("BPOSD_LAYOUT_ITERS")) : 200000;
    for (int mp : {256, 512, 1024}) {
        if (h->m > mp) continue;
        if (class_layout::build(h->rp, h->ci, h->m, h->n, shp->dclo, shp->dc, shp->dvlo, shp->dvhi, kClassVPT, mp, mp, iters, T)) { ok = true; MP = mp; break; }
    }
    if (!ok) return 0;
    if (getenv("BPOSD_DEBUG_OCC"))
        fprintf(stderr, "[bposd] class BP layout: %d threads, stride %d, bit pass %ld read cycles (floor %ld) + %ld write cycles (floor %ld)\n", T.NT,
                MP, T.read_cycles, T.read_floor, T.write_cycles, T.write_floor);
    auto up = [&](int** dst, const std::vector<int>& v) -> int {
        if (*dst) { (void)hipFree(*dst); *dst = nullptr; }
        HIP_TRY(h, hipMalloc((void**)dst, sizeof(int) * std::max<size_t>(v.size(), 1)));
        HIP_TRY(h, hipMemcpy(*dst, v.data(), sizeof(int) * v.size(), hipMemcpyHostToDevice));
        return 0;
    };
    int rc;
    if ((rc = up(&h->d_cpos_chk, T.pos_chk))) return rc;
    if ((rc = up(&h->d_cpos_bit, T.pos_bit))) return rc;
    if ((rc = up(&h->d_cbit_slot, T.bit_slot))) return rc;
    if ((rc = up(&h->d_cgrp_deg, T.grp_deg))) return rc;
    if ((rc = up(&h->d_cgrp_cdeg, T.grp_cdeg))) return rc;
    h->class_dclo = shp->dclo; h->class_dc = shp->dc; h->class_dvlo = shp->dvlo; h->class_dvhi = shp->dvhi; h->class_mp = MP; h->class_nt = T.NT;
    h->class_read_cycles = T.read_cycles; h->class_write_cycles = T.write_cycles;
    h->class_read_floor = T.read_floor; h->class_write_floor = T.write_floor;
    h->class_ok = true;
    return 0;
}
// ------------------------------------------------------------------------ large-code BP launch
int build_tables_large(bposd_handle* h, int DV, int MP) {
    const int m = h->m, n = h->n;
    std::vector<int> chk_deg(m), var_deg(n, 0);
    std::vector<int> var_pos((size_t)DV * n, 0), var_ck((size_t)DV * n, 0);
    for (int c = 0; c < m; ++c) {
        chk_deg[c] = h->rp[c + 1] - h->rp[c];
        for (int e = h->rp[c]; e < h->rp[c + 1]; ++e) {
            const int i = h->ci[e];
            const int d = var_deg[i]++;
            var_pos[(size_t)d * n + i] = (e - h->rp[c]) * MP + c;
            var_ck[(size_t)d * n + i] = c * 16 + (e - h->rp[c]);  // (slot < 16: the large-code kernels are built for check degree <= 16)
        }
    }
    auto up = [&](int** dst, const std::vector<int>& v) -> int {
        if (*dst) { (void)hipFree(*dst); *dst = nullptr; }
        HIP_TRY(h, hipMalloc((void**)dst, sizeof(int) * std::max<size_t>(v.size(), 1)));
        HIP_TRY(h, hipMemcpy(*dst, v.data(), sizeof(int) * v.size(), hipMemcpyHostToDevice));
        return 0;
    };
    int rc;
    if ((rc = up(&h->d_chk_deg, chk_deg))) return rc;
    if ((rc = up(&h->d_var_deg, var_deg))) return rc;
    if ((rc = up(&h->d_var_pos, var_pos))) return rc;
    if ((rc = up(&h->d_var_ck, var_ck))) return rc;
    h->tab_mp = MP;
    return 0;
}
// ------------------------------------------------------------------ serial-schedule BP: tables + launch
int build_tables_serial(bposd_handle* h) {
    const int m = h->m, n = h->n, E = h->E;
    std::vector<int> cp(n + 1, 0), ce(E), erow(E), fill(n, 0);
    for (int e = 0; e < E; ++e) cp[h->ci[e] + 1]++;
    for (int i = 0; i < n; ++i) cp[i + 1] += cp[i];
    for (int c = 0; c < m; ++c)
        for (int e = h->rp[c]; e < h->rp[c + 1]; ++e) {
            erow[e] = c;
            ce[cp[h->ci[e]] + fill[h->ci[e]]++] = e;  // ascending row within a column
        }
    // level(j) = 1 + the highest level among the earlier bits that share a check with j
    std::vector<int> last(m, 0), level(n, 0);
    int nlev = 0;
    for (int i = 0; i < n; ++i) {
        int lv = 0;
        for (int k = cp[i]; k < cp[i + 1]; ++k) lv = std::max(lv, last[erow[ce[k]]]);
        level[i] = lv + 1;
        for (int k = cp[i]; k < cp[i + 1]; ++k) last[erow[ce[k]]] = lv + 1;
        nlev = std::max(nlev, lv + 1);
    }
    std::vector<int> lptr(nlev + 1, 0), lbits(n);
    for (int i = 0; i < n; ++i) lptr[level[i]]++;  // level l (1-based) counted into slot l
    for (int l = 0; l < nlev; ++l) lptr[l + 1] += lptr[l];
    {
        std::vector<int> pos(lptr.begin(), lptr.end() - 1);
        for (int i = 0; i < n; ++i) lbits[pos[level[i] - 1]++] = i;  // ascending bit index inside a level
    }
    auto up = [&](int** dst, const std::vector<int>& v) -> int {
        HIP_TRY(h, hipMalloc((void**)dst, sizeof(int) * std::max<size_t>(v.size(), 1)));
        HIP_TRY(h, hipMemcpy(*dst, v.data(), sizeof(int) * v.size(), hipMemcpyHostToDevice));
        return 0;
    };
    int rc;
    if ((rc = up(&h->d_cp, cp))) return rc;
    if ((rc = up(&h->d_ce, ce))) return rc;
    if ((rc = up(&h->d_erow, erow))) return rc;
    if ((rc = up(&h->d_lvl_ptr, lptr))) return rc;
    if ((rc = up(&h->d_lvl_bits, lbits))) return rc;
    h->nlevels = nlev;
    return 0;
}
// rank of a large code: one elimination of the zero syndrome on the device (the host routine is O(m^2 n / 64))
int probe_rank_large(bposd_handle* h, int* rank) {
    DevBuf tmp;
    const size_t n = h->n, m = h->m;
    const size_t off_llr = 0, off_synd = off_llr + sizeof(double) * n, off_out = off_synd + ((m + 255) & ~(size_t)255),
                 off_cnt = off_out + ((n + 255) & ~(size_t)255), total = off_cnt + 64;
    int rc = ensure(h, tmp, total);
    if (rc) return rc;
    unsigned char* b = (unsigned char*)tmp.p;
    HIP_TRY(h, hipMemsetAsync(b, 0, total, h->cur->stream));
    const int cnt[8] = {0, 1, 0, 0, /*osd_list*/ 0, /*rank_out*/ -1, 0, 0};
    HIP_TRY(h, hipMemcpyAsync(b + off_cnt, cnt, sizeof(cnt), hipMemcpyHostToDevice, h->cur->stream));
    OsdParams P{};
    P.m = h->m; P.n = h->n; P.rank = std::min(h->m, h->n);
    P.osd_method = BPOSD_OSD_0; P.osd_order = 0; P.tie_policy = 0;
    P.synd = b + off_synd; P.rp = h->d_rp; P.ci = h->d_ci; P.llr_ws = (const double*)(b + off_llr);
    P.osd_list = (const int*)(b + off_cnt) + 4; P.counters = (int*)(b + off_cnt);
    P.out_osd0 = nullptr; P.out_osdw = b + off_out;
    HIP_TRY(h, hipEventRecord(h->cur->ev_bp, h->cur->stream));
    HIP_TRY(h, hipStreamWaitEvent(h->cur->osd_stream, h->cur->ev_bp, 0));
    rc = launch_osd_large(h, P, 1, (int*)(b + off_cnt) + 5);
    if (!rc) {
        int got[8];
        hipError_t e = hipStreamSynchronize(h->cur->osd_stream);
        if (e == hipSuccess) e = hipMemcpy(got, b + off_cnt, sizeof(got), hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail(h, BPOSD_ERR_HIP, "rank probe failed: %s", hipGetErrorString(e));
        else if (got[5] < 0 || got[5] > std::min(h->m, h->n)) rc = fail(h, BPOSD_ERR_HIP, "rank probe returned %d", got[5]);
        else *rank = got[5];
    }
    release(tmp);
    return rc;
}

int num_candidates(const bposd_handle* h) {
    const int w = h->cfg.osd_order;
    if (h->cfg.osd_method <= BPOSD_OSD_0 || w == 0) return 0;
    if (h->cfg.osd_method == BPOSD_OSD_E) return (1 << w) - 1;
    return h->kprime + w * (w - 1) / 2;
}
}  // namespace bposd_host

__global__ void pack_rows_kernel(const uint8_t* __restrict__ in, long long B, int n, int wpr,
                                 unsigned long long* __restrict__ out) {
    // one wave per 64-bit output word: lane l supplies bit l (ballot); grid-stride over words
    const long long nwords = B * wpr;
    const int lane = threadIdx.x & 63;
    const long long wave0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long nwave = ((long long)gridDim.x * blockDim.x) >> 6;
    for (long long w = wave0; w < nwords; w += nwave) {
        const long long b = w / wpr;
        const int i = (int)(w - b * wpr) * 64 + lane;
        const bool bit = (i < n) && (in[b * n + i] & 1);
        const unsigned long long v = __ballot(bit);
        if (lane == 0) out[w] = v;
    }
}

// the inverse: B rows of wpr little-endian 64-bit words -> B rows of n 0/1 bytes (the form the decode kernels read)
__global__ void unpack_rows_kernel(const unsigned long long* __restrict__ in, long long B, int n, int wpr, uint8_t* __restrict__ out) {
    const long long total = B * (long long)n;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const long long b = e / n;
        const int i = (int)(e - b * n);
        out[e] = (uint8_t)((in[b * wpr + (i >> 6)] >> (i & 63)) & 1ull);
    }
}

namespace {

int launch_pack(bposd_handle* h, hipStream_t st, const uint8_t* d_bytes, long long B, int n, unsigned long long* d_words) {
    if (B <= 0) return 0;
    const int wpr = (n + 63) / 64, threads = 256;
    const long long want = ((long long)B * wpr * 64 + threads - 1) / threads;
    const unsigned grid = (unsigned)std::min<long long>(want, (long long)h->num_cu * 2);  // few, fat workgroups: see launch_unpack
    hipLaunchKernelGGL(pack_rows_kernel, dim3(grid), dim3(threads), 0, st, d_bytes, B, n, wpr, d_words);
    HIP_TRY(h, hipGetLastError());
    return 0;
}

int launch_unpack(bposd_handle* h, hipStream_t st, const unsigned long long* d_words, long long B, int n, uint8_t* d_bytes) {
    if (B <= 0) return 0;
    const int wpr = (n + 63) / 64, threads = 256;
    const long long want = ((long long)B * n + threads - 1) / threads;
    // few, fat workgroups: next to a persistent BP grid that takes every slot that frees up, a grid of thousands of short
    // workgroups is starved after its first placements (traced: 6 ms per pack kernel instead of 0.4 ms)
    const unsigned grid = (unsigned)std::min<long long>(want, (long long)h->num_cu * 2);
    hipLaunchKernelGGL(unpack_rows_kernel, dim3(grid), dim3(threads), 0, st, d_words, B, n, wpr, d_bytes);
    HIP_TRY(h, hipGetLastError());
    return 0;
}

}  // namespace

// ================================================================================ C-ABI
extern "C" {

int bposd_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char* bposd_version(void) { return "bposd_mi355x 0.1 (gfx950)"; }

const char* bposd_last_error(bposd_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

void bposd_destroy(bposd_handle* h) {
    if (!h) return;
    DeviceGuard dev_guard(h->device);
    for (auto& l : h->lanes) {
        if (l.stream) (void)hipStreamSynchronize(l.stream);
        if (l.h_list) (void)hipHostFree(l.h_list);
        if (l.ev_copy) (void)hipEventDestroy(l.ev_copy);
        if (l.copy_stream) (void)hipStreamDestroy(l.copy_stream);
        for (DevBuf* b : {&l.io_cmp0, &l.io_cmpw, &l.io_psynd, &l.io_posdw, &l.io_posd0, &l.io_pbp, &l.io_pcmp}) release(*b);
        for (DevBuf* b : {&l.bpl_msg, &l.bpl_llr, &l.osdl_ws, &l.io_sel, &l.osd_rows_ws, &l.llr_ws, &l.osd_list, &l.io_synd, &l.io_osdw,
                          &l.io_osd0, &l.io_bp, &l.io_conv, &l.io_iters, &l.io_llr})
            release(*b);
        for (void* p : {(void*)l.d_counters, (void*)l.d_osd_dbg})  // (d_iter_total lives inside the d_counters block)
            if (p) (void)hipFree(p);
        if (l.osd_stream) (void)hipStreamSynchronize(l.osd_stream);
        if (l.h_stage) (void)hipHostFree(l.h_stage);
        if (l.h_tail) (void)hipHostFree(l.h_tail);
        if (l.d_alt) (void)hipFree(l.d_alt);
        if (l.h_alt) (void)hipHostFree(l.h_alt);
        if (l.ev_alt) (void)hipEventDestroy(l.ev_alt);
        if (l.ev_bp) (void)hipEventDestroy(l.ev_bp);
        if (l.ev_osd) (void)hipEventDestroy(l.ev_osd);
        if (l.ev_done) (void)hipEventDestroy(l.ev_done);
        if (l.ev_up) (void)hipEventDestroy(l.ev_up);
        if (l.osd_stream) (void)hipStreamDestroy(l.osd_stream);
        if (l.stream) (void)hipStreamDestroy(l.stream);
    }
    for (void* p : {(void*)h->d_rp, (void*)h->d_ci, (void*)h->d_chk_deg, (void*)h->d_var_deg,
                    (void*)h->d_var_pos, (void*)h->d_var_ck, (void*)h->d_pos_bit, (void*)h->d_llr0, (void*)h->d_cost, (void*)h->d_llr0_alt, (void*)h->d_cost_alt,
                    (void*)h->d_lpos_chk, (void*)h->d_lpos_bit,
                    (void*)h->d_lpos_alo, (void*)h->d_lpos_ahi, (void*)h->d_lgrp_dl, (void*)h->d_lpos_dl,
                    (void*)h->d_cpos_chk, (void*)h->d_cpos_bit, (void*)h->d_cbit_slot, (void*)h->d_cgrp_deg, (void*)h->d_cgrp_cdeg,
                    (void*)h->d_cp, (void*)h->d_ce, (void*)h->d_erow, (void*)h->d_lvl_ptr, (void*)h->d_lvl_bits})
        if (p) (void)hipFree(p);
    for (CallRecord* rs : {h->rec, h->lane_rec})
        for (int k = 0; k < (rs == h->rec ? BPOSD_MAX_CHUNKS : BPOSD_LANES); ++k) {
            if (rs[k].h_counters) (void)hipHostFree(rs[k].h_counters);
            for (auto& e : rs[k].ev)
                if (e) (void)hipEventDestroy(e);
        }
    delete h;
}

int bposd_create(const bposd_config* cfg, const int32_t* indptr, const int32_t* indices, int32_t m,
                 int32_t n, const double* channel_probs, bposd_handle** out) {
    if (out) *out = nullptr;
    if (!cfg || !indptr || !indices || !channel_probs || !out)
        return fail(nullptr, BPOSD_ERR_INVALID, "null argument");
    if (m <= 0 || n <= 0) return fail(nullptr, BPOSD_ERR_INVALID, "empty parity-check matrix (%d x %d)", m, n);
    if (cfg->bp_method != BPOSD_BP_PRODUCT_SUM && cfg->bp_method != BPOSD_BP_MIN_SUM)
        return fail(nullptr, BPOSD_ERR_INVALID, "bp_method must be 0 (product-sum) or 1 (min-sum)");
    if (cfg->osd_method < BPOSD_OSD_OFF || cfg->osd_method > BPOSD_OSD_CS)
        return fail(nullptr, BPOSD_ERR_INVALID, "osd_method out of range");
    if (cfg->max_iter < 0 || cfg->osd_order < 0) return fail(nullptr, BPOSD_ERR_INVALID, "negative max_iter / osd_order");
    if (cfg->sort_tie_policy < 0 || cfg->sort_tie_policy > 1 || cfg->weight_fn < 0 || cfg->weight_fn > 1)
        return fail(nullptr, BPOSD_ERR_INVALID, "sort_tie_policy / weight_fn out of range");
    if (cfg->osd_e_bit_order < 0 || cfg->osd_e_bit_order > 1)
        return fail(nullptr, BPOSD_ERR_INVALID, "osd_e_bit_order must be 0 (LSB first) or 1 (MSB first)");
    if (cfg->ps_math_form < 0 || cfg->ps_math_form > 1)
        return fail(nullptr, BPOSD_ERR_INVALID, "ps_math_form must be 0 (the reference's operation order) or 1 (two divisions per edge)");
    if (cfg->schedule != 0 && cfg->schedule != 1) return fail(nullptr, BPOSD_ERR_INVALID, "schedule must be 0 (parallel) or 1 (serial)");
    if (!(cfg->ps_clip >= 0.0) || std::isinf(cfg->ps_clip)) return fail(nullptr, BPOSD_ERR_INVALID, "ps_clip must be 0 (off) or a finite positive bound");
    if (indptr[0] != 0) return fail(nullptr, BPOSD_ERR_INVALID, "csr_indptr[0] must be 0");
    for (int c = 0; c < m; ++c) {
        if (indptr[c + 1] < indptr[c]) return fail(nullptr, BPOSD_ERR_INVALID, "csr_indptr not monotone");
        for (int e = indptr[c]; e < indptr[c + 1]; ++e) {
            if (indices[e] < 0 || indices[e] >= n) return fail(nullptr, BPOSD_ERR_INVALID, "column index out of range");
            if (e > indptr[c] && indices[e] <= indices[e - 1])
                return fail(nullptr, BPOSD_ERR_INVALID, "column indices must be strictly ascending within a row");
        }
    }
    for (int i = 0; i < n; ++i)
        if (!(channel_probs[i] >= 0.0 && channel_probs[i] <= 1.0))
            return fail(nullptr, BPOSD_ERR_INVALID, "channel_probs[%d] = %g is not a probability", i, channel_probs[i]);

    int ndev = bposd_device_count();
    if (ndev <= 0) return fail(nullptr, BPOSD_ERR_NO_DEVICE, "no HIP device visible: the MI355X decoder has no CPU path");
    if (cfg->device < 0 || cfg->device >= ndev)
        return fail(nullptr, BPOSD_ERR_INVALID, "device %d out of range (%d visible)", cfg->device, ndev);

    bposd_handle* h = new bposd_handle();
    h->cfg = *cfg;
    h->device = cfg->device;
    h->m = m;
    h->n = n;
    h->E = indptr[m];
    h->rp.assign(indptr, indptr + m + 1);
    h->ci.assign(indices, indices + h->E);
    h->probs.assign(channel_probs, channel_probs + n);
    h->max_iter = cfg->max_iter > 0 ? cfg->max_iter : n;  // A.1: 0 => block length

#define CREATE_TRY(expr)                                                                        \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess) {                                                                 \
            fail(nullptr, BPOSD_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(_e));        \
            bposd_destroy(h);                                                                   \
            return BPOSD_ERR_HIP;                                                               \
        }                                                                                       \
    } while (0)
#define CREATE_RC(expr)                                                                         \
    do {                                                                                        \
        int _rc = (expr);                                                                       \
        if (_rc) {                                                                              \
            g_create_error = h->err;                                                            \
            bposd_destroy(h);                                                                   \
            return _rc;                                                                         \
        }                                                                                       \
    } while (0)

    DeviceGuard dev_guard(h->device);
    CREATE_TRY(dev_guard.err);
    hipDeviceProp_t prop;
    CREATE_TRY(hipGetDeviceProperties(&prop, h->device));
    h->num_cu = prop.multiProcessorCount;
    if (prop.maxSharedMemoryPerMultiProcessor > 0) h->lds_per_cu = prop.maxSharedMemoryPerMultiProcessor;
    int prio_least = 0, prio_greatest = 0;
    CREATE_TRY(hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
    for (auto& l : h->lanes) {
        CREATE_TRY(hipStreamCreateWithFlags(&l.stream, hipStreamNonBlocking));
        CREATE_TRY(hipStreamCreateWithPriority(&l.osd_stream, hipStreamNonBlocking, prio_greatest));
        CREATE_TRY(hipEventCreateWithFlags(&l.ev_bp, hipEventDisableTiming));
        CREATE_TRY(hipEventCreateWithFlags(&l.ev_osd, hipEventDisableTiming));
        CREATE_TRY(hipEventCreateWithFlags(&l.ev_done, hipEventDisableTiming));
        CREATE_TRY(hipEventCreateWithFlags(&l.ev_up, hipEventDisableTiming));
        CREATE_TRY(hipStreamCreateWithFlags(&l.copy_stream, hipStreamNonBlocking));
        CREATE_TRY(hipEventCreateWithFlags(&l.ev_copy, hipEventDisableTiming));
        CREATE_TRY(hipMalloc((void**)&l.d_counters, 32));  // 4 counters + the 64-bit iteration total: one memset, one copy
        CREATE_TRY(hipHostMalloc((void**)&l.h_tail, 64, hipHostMallocMapped));
        *l.h_tail = 0;
    }
    for (CallRecord* rs : {h->rec, h->lane_rec})
        for (int k = 0; k < (rs == h->rec ? BPOSD_MAX_CHUNKS : BPOSD_LANES); ++k) {
            for (auto& e : rs[k].ev) CREATE_TRY(hipEventCreate(&e));
            CREATE_TRY(hipHostMalloc((void**)&rs[k].h_counters, 32));
            rs[k].h_iter_total = (unsigned long long*)(rs[k].h_counters + 4);
            rs[k].h_counters[0] = rs[k].h_counters[1] = 0;
            *rs[k].h_iter_total = 0;
        }
    h->cur = &h->lanes[0];
    h->currec = &h->rec[0];

    // degrees
    std::vector<int> vdeg(n, 0);
    int dc_min = 1 << 30, dv_min = 1 << 30;
    for (int c = 0; c < m; ++c) {
        const int d = indptr[c + 1] - indptr[c];
        h->dc_max = std::max(h->dc_max, d);
        dc_min = std::min(dc_min, d);
        for (int e = indptr[c]; e < indptr[c + 1]; ++e) vdeg[indices[e]]++;
    }
    for (int i = 0; i < n; ++i) {
        h->dv_max = std::max(h->dv_max, vdeg[i]);
        dv_min = std::min(dv_min, vdeg[i]);
    }
    h->regular = (dc_min == h->dc_max) && (dv_min == h->dv_max);

    DegPair pair{16, 8};
    if (is_reg63(h)) pair = {6, 3};
    else if (!pick_pair(h->dc_max, h->dv_max, &pair)) h->bp_any = true;  // beyond the compiled degrees: run-time degree loops
    // small path: messages in LDS, OSD rows in registers.  Anything beyond goes to the HBM-resident kernels.
    // BP in LDS whenever a workgroup shape holds the messages (up to 2048 checks); OSD in registers up to m = 1024 /
    // n = 2047.  Anything beyond goes to the HBM-resident kernels, BP and OSD independently.
    const int shp = h->bp_any ? 0 : pick_shape(h);
    h->bp_hbm = !h->bp_any && (!shp || bp_lds_bytes(pair.dc, shape_threads(h, shp) * shape_cpt(shp)) > h->lds_per_cu);
    h->large = (m > 1024) || (osd_words(n) == 0) || h->bp_hbm;
    if (const char* e = getenv("BPOSD_FORCE_LARGE_OSD")) h->large = h->large || e[0] == '1';  // (probe: the HBM-resident OSD kernel on a small code)
    if (h->large) {
        // Two lanes, three where BP is HBM-resident too: its workgroups take a whole CU like the eliminations', a call is a
        // 35-50 ms BP launch followed by an OSD launch that lasts as long as its slowest elimination (87 ms on L29k, twice the
        // median), and with two calls in flight the CUs the fast eliminations free stay idle until the next call's BP kernel is
        // launched (l29k_ms_e15: 88 ms per step with two lanes, 82-83 with three or four -- the sum of the kernels' CU time).
        h->nlanes = h->bp_hbm ? 3 : 2;
        if (const char* e = getenv("BPOSD_LARGE_LANES")) h->nlanes = std::max(1, std::min(BPOSD_LANES, atoi(e)));
        if (n > 32767 || m > 16384 || (h->bp_hbm && bp_large_lds_need(m, n) > h->lds_per_cu)) {
            fail(nullptr, BPOSD_ERR_UNSUPPORTED, "code too large even for the HBM-resident kernels (m=%d n=%d; limits 16384 / 32767)", m, n);
            bposd_destroy(h);
            return BPOSD_ERR_UNSUPPORTED;
        }
        const int span_cap = osd_large_maxspan(cfg->osd_method == BPOSD_OSD_CS);
        if (cfg->osd_method >= BPOSD_OSD_E && cfg->osd_order > span_cap) {
            fail(nullptr, BPOSD_ERR_UNSUPPORTED,
                 "osd order %d > %d is not supported by the HBM-resident OSD kernel (m=%d n=%d)", cfg->osd_order,
                 span_cap, m, n);
            bposd_destroy(h);
            return BPOSD_ERR_UNSUPPORTED;
        }
    }

    h->rank = h->large ? std::min(m, n) : gf2_rank_host(m, n, h->rp, h->ci);  // large: probed on the device below
    h->kprime = n - h->rank;
    if (cfg->osd_method != BPOSD_OSD_OFF && !h->large) {
        if (m > 1024 || osd_words(n) == 0) {
            fail(nullptr, BPOSD_ERR_UNSUPPORTED,
                 "code too large for the register-resident OSD kernel (m=%d > 1024 or n=%d > 2047): large-code path not built yet",
                 m, n);
            bposd_destroy(h);
            return BPOSD_ERR_UNSUPPORTED;
        }
        if (cfg->osd_method >= BPOSD_OSD_E && cfg->osd_order > h->kprime) {
            fail(nullptr, BPOSD_ERR_INVALID, "osd_order %d exceeds the number of non-pivot columns n - rank = %d",
                 cfg->osd_order, h->kprime);
            bposd_destroy(h);
            return BPOSD_ERR_INVALID;
        }
        if (cfg->osd_method == BPOSD_OSD_E && cfg->osd_order > 20) {
            fail(nullptr, BPOSD_ERR_UNSUPPORTED, "osd_e order %d > 20 not supported", cfg->osd_order);
            bposd_destroy(h);
            return BPOSD_ERR_UNSUPPORTED;
        }
        if (cfg->osd_method == BPOSD_OSD_CS && cfg->osd_order > 64) {
            fail(nullptr, BPOSD_ERR_UNSUPPORTED, "osd_cs order %d > 64 not supported", cfg->osd_order);
            bposd_destroy(h);
            return BPOSD_ERR_UNSUPPORTED;
        }
    }
    h->ncand = num_candidates(h);

    auto upi = [&](int** dst, const std::vector<int>& v) -> hipError_t {
        hipError_t e = hipMalloc((void**)dst, sizeof(int) * std::max<size_t>(v.size(), 1));
        if (e != hipSuccess) return e;
        return hipMemcpy(*dst, v.data(), sizeof(int) * v.size(), hipMemcpyHostToDevice);
    };
    CREATE_TRY(upi(&h->d_rp, h->rp));
    CREATE_TRY(upi(&h->d_ci, h->ci));
    CREATE_TRY(hipMalloc((void**)&h->d_llr0, sizeof(double) * n));
    CREATE_TRY(hipMalloc((void**)&h->d_cost, sizeof(double) * n));
    CREATE_TRY(hipMalloc((void**)&h->d_llr0_alt, sizeof(double) * n));
    CREATE_TRY(hipMalloc((void**)&h->d_cost_alt, sizeof(double) * n));
    if (h->bp_any) CREATE_RC(build_tables_serial(h));  // (its CSC edge map is what the any-degree kernel walks)
    else if (h->bp_hbm) CREATE_RC(build_tables_large(h, h->dv_max <= 6 ? 6 : 8, (m + 63) / 64 * 64));
    else CREATE_RC(build_tables(h, pair.dc, pair.dv, shape_threads(h, shp) * shape_cpt(shp), shape_threads(h, shp), 2 * shape_cpt(shp)));
    if (!h->bp_any && !h->bp_hbm && cfg->bp_method == BPOSD_BP_MIN_SUM) CREATE_RC(build_tables_local(h));
    if (!h->bp_any && !h->bp_hbm && (!(h->local_ok && cfg->bp_method == BPOSD_BP_MIN_SUM) || getenv("BPOSD_CLASS_ALWAYS"))) CREATE_RC(build_tables_class(h));
    if (cfg->schedule == 1) {
        if (h->dv_max > bp_serial_max_dv()) {
            fail(h, BPOSD_ERR_UNSUPPORTED, "serial schedule: bit degree %d exceeds %d", h->dv_max, bp_serial_max_dv());
            CREATE_RC(BPOSD_ERR_UNSUPPORTED);
        }
        if (!h->bp_any) CREATE_RC(build_tables_serial(h));
    }
    CREATE_RC(upload_priors(h));
    if (h->large) {
        CREATE_RC(probe_rank_large(h, &h->rank));
        h->kprime = n - h->rank;
        if (cfg->osd_method >= BPOSD_OSD_E && cfg->osd_order > h->kprime) {
            fail(h, BPOSD_ERR_INVALID, "osd_order %d exceeds the number of non-pivot columns n - rank = %d",
                 cfg->osd_order, h->kprime);
            CREATE_RC(BPOSD_ERR_INVALID);
        }
        h->ncand = num_candidates(h);
    }
    *out = h;
    return BPOSD_OK;
#undef CREATE_TRY
#undef CREATE_RC
}

int bposd_update_channel_probs(bposd_handle* h, const double* channel_probs) {
    if (!h || !channel_probs) return fail(h, BPOSD_ERR_INVALID, "null argument");
    for (int i = 0; i < h->n; ++i)
        if (!(channel_probs[i] >= 0.0 && channel_probs[i] <= 1.0))
            return fail(h, BPOSD_ERR_INVALID, "channel_probs[%d] = %g is not a probability", i, channel_probs[i]);
    DeviceGuard dev_guard(h->device);
    HIP_TRY(h, dev_guard.err);
    { int rcs = sync_all_lanes(h); if (rcs) return rcs; }
    h->probs.assign(channel_probs, channel_probs + h->n);
    return upload_priors(h);
}

int bposd_set_bp_variant(bposd_handle* h, int32_t variant) {
    if (!h) return BPOSD_ERR_INVALID;
    if (variant != 0 && variant != 1 && variant != 2 && variant != 4 && !(variant >= 16 && variant <= 26) && variant != 32 && variant != 63 && variant != 64)
        return fail(h, BPOSD_ERR_INVALID, "bp variant must be 0 (auto), 1, 2, 4 (LDS kernel shapes), 16 .. 26 (local-edge kernel), 32 (class kernel), 63 (HBM-resident min-sum with whole check records in the workspace) or 64 (any-degree kernel)");
    if (variant == 64 && !h->d_cp) {  // the any-degree kernel as a second implementation for cross-checks: its CSC edge map
        DeviceGuard dev_guard(h->device);
        HIP_TRY(h, dev_guard.err);
        int rc_any = sync_all_lanes(h);
        if (!rc_any) rc_any = build_tables_serial(h);
        if (rc_any) return rc_any;
    }
    if (variant == 32 && !h->class_ok)
        return fail(h, BPOSD_ERR_UNSUPPORTED, "the class BP kernel needs one check degree and bit degrees of a compiled range");
    if (variant >= 16 && variant <= 26 && !(h->local_ok && h->cfg.bp_method == BPOSD_BP_MIN_SUM))
        return fail(h, BPOSD_ERR_UNSUPPORTED, "the local-edge BP kernel needs a (3,6)-regular code with n = 2m and min-sum");
    h->bp_variant = variant;
    return BPOSD_OK;
}

int bposd_info(bposd_handle* h, int32_t* rank, int32_t* ncand, int32_t* max_iter, int32_t* nnz) {
    if (!h) return BPOSD_ERR_INVALID;
    if (rank) *rank = h->rank;
    if (ncand) *ncand = h->ncand;
    if (max_iter) *max_iter = h->max_iter;
    if (nnz) *nnz = h->E;
    return BPOSD_OK;
}

int bposd_pack_rows_device(bposd_handle* h, const uint8_t* d_bytes, int64_t B, int32_t n, uint64_t* d_words) {
    if (!h) return BPOSD_ERR_INVALID;
    return bposd_pack_rows_device_lane(h, h->last_lane, d_bytes, B, n, d_words);
}

int bposd_pack_rows_device_lane(bposd_handle* h, int32_t lane, const uint8_t* d_bytes, int64_t B, int32_t n, uint64_t* d_words) {
    if (!h) return BPOSD_ERR_INVALID;
    if (lane < 0 || lane >= h->nlanes) return fail(h, BPOSD_ERR_INVALID, "lane %d out of range", lane);
    if (B < 0 || n <= 0 || (B > 0 && (!d_bytes || !d_words))) return fail(h, BPOSD_ERR_INVALID, "bad pack arguments");
    if (B == 0) return BPOSD_OK;
    DeviceGuard dev_guard(h->device);
    HIP_TRY(h, dev_guard.err);
    const int wpr = (n + 63) / 64;
    const long long nwords = (long long)B * wpr;
    const int threads = 256;
    const long long want = (nwords * 64 + threads - 1) / threads;
    const unsigned grid = (unsigned)std::min<long long>(want, (long long)h->num_cu * 16);
    // queued behind the device-pointer decode that ran on this lane (stream order)
    hipLaunchKernelGGL(pack_rows_kernel, dim3(grid), dim3(threads), 0, h->lanes[lane].stream, d_bytes, (long long)B, (int)n, wpr,
                       (unsigned long long*)d_words);
    HIP_TRY(h, hipGetLastError());
    return BPOSD_OK;
}

int bposd_layout_info(bposd_handle* h, int64_t* natural, int64_t* chosen, int64_t* ideal) {
    if (!h) return BPOSD_ERR_INVALID;
    if (natural) *natural = h->layout_cost_natural;
    if (chosen) *chosen = h->layout_cost;
    if (ideal) *ideal = h->layout_cost_ideal;
    return BPOSD_OK;
}

int bposd_set_osd_variant(bposd_handle* h, int32_t variant) {
    if (!h) return BPOSD_ERR_INVALID;
    if (variant < 0 || variant > 2) return fail(h, BPOSD_ERR_INVALID, "osd variant must be 0 (auto), 1 (workgroup kernel) or 2 (wave kernel where it applies)");
    h->osd_variant = variant;
    return BPOSD_OK;
}

int bposd_last_osd_kernel(bposd_handle* h) { return h ? h->last_osd_kernel : BPOSD_ERR_INVALID; }

int bposd_bp_kernel_info(bposd_handle* h, int32_t* kernel, int64_t* lds_model) {
    if (!h) return BPOSD_ERR_INVALID;
    if (kernel) *kernel = h->last_bp_kernel;
    if (lds_model) {
        for (int k = 0; k < 4; ++k) lds_model[k] = 0;
        if (h->last_bp_kernel == BPOSD_BP_KERNEL_LOCAL) {
            lds_model[0] = h->local_passes; lds_model[1] = 4 * (h->local_mp / 32);
            lds_model[2] = h->local_wcycles; lds_model[3] = 6 * 4 * (h->local_mp / 64);
        } else if (h->last_bp_kernel == BPOSD_BP_KERNEL_CLASS) {
            lds_model[0] = h->class_read_cycles; lds_model[1] = h->class_read_floor;
            lds_model[2] = h->class_write_cycles; lds_model[3] = h->class_write_floor;
        } else if (h->last_bp_kernel == BPOSD_BP_KERNEL_LARGE) {
            lds_model[0] = h->large_form;  // (no bank model: which form of the kernel ran -- bp_large_kernel.hip.h)
        }
    }
    return BPOSD_OK;
}

int bposd_synchronize(bposd_handle* h) {
    if (!h) return BPOSD_ERR_INVALID;
    DeviceGuard dev_guard(h->device);
    HIP_TRY(h, dev_guard.err);
    return sync_all_lanes(h);
}

// One BP launch + one OSD launch on a lane.  lane < 0: a stand-alone device-pointer call, which takes the handle's next
// lane and is record 0 of a new "last call"; lane >= 0, rec_idx: chunk `rec_idx` of a host-pointer call on that lane.
static int decode_device_impl(bposd_handle* h, const uint8_t* d_synd, int64_t B, const uint8_t* d_sel,
                              uint8_t* d_osdw, uint8_t* d_osd0, uint8_t* d_bp, uint8_t* d_conv,
                              int32_t* d_iters, double* d_llr, int lane = -1, int rec_idx = 0, bool lean = false) {
    // lean (the small host-pointer call): everything on the lane's own stream in program order -- no events, no second
    // stream -- so that the call costs a memset, two launches, one 32-byte copy and one synchronisation
    if (!h) return BPOSD_ERR_INVALID;
    if (B < 0 || B > 0x7fffffffLL) return fail(h, BPOSD_ERR_INVALID, "batch size %lld out of range", (long long)B);
    if (B == 0) return BPOSD_OK;
    if (!d_synd || !d_osdw) return fail(h, BPOSD_ERR_INVALID, "syndromes and osdw buffers are required");
    DeviceGuard dev_guard(h->device);
    HIP_TRY(h, dev_guard.err);
    if (lane < 0) {
        lane = h->next_lane;
        h->next_lane = (h->next_lane + 1) % h->nlanes;
        h->last_lane = lane;
        h->nrec = 0;
        h->currec = &h->lane_rec[lane];  // stream order on the lane: its previous call has filled the record by now
        h->async_pending = true;
    } else {
        h->currec = &h->rec[rec_idx];
    }
    h->cur = &h->lanes[lane];
    h->osd_now = lean ? h->cur->stream : h->cur->osd_stream;
    const bool osd_on = h->cfg.osd_method != BPOSD_OSD_OFF && !h->bp_only;
    int rc;
    if (osd_on) {
        if ((rc = ensure_lanes(h, &Lane::llr_ws, sizeof(double) * (size_t)B * h->n))) return rc;
        if ((rc = ensure_lanes(h, &Lane::osd_list, sizeof(int) * (size_t)B))) return rc;
    }
    HIP_TRY(h, hipMemsetAsync(h->cur->d_counters, 0, 32, h->cur->stream));

    BpParams P{};
    P.m = h->m;
    P.n = h->n;
    P.B = B;
    P.max_iter = h->max_iter;
    P.ms_scaling = h->cfg.ms_scaling_factor;
    P.ps_clip = h->cfg.ps_clip;
    P.osd_enabled = osd_on ? 1 : 0;
    P.synd = d_synd;
    P.llr0 = h->d_llr0;
    P.sel = d_sel;
    P.llr0_alt = h->lane_alt ? h->cur->d_alt : h->d_llr0_alt;
    P.chk_deg = h->d_chk_deg;
    P.var_deg = h->d_var_deg;
    P.var_pos = h->d_var_pos;
    P.pos_bit = h->d_pos_bit;
    P.out_bp = d_bp;
    P.out_osd0 = d_osd0;
    P.out_osdw = d_osdw;
    P.out_conv = d_conv;
    P.out_iters = d_iters;
    P.out_llr = d_llr;
    P.llr_ws = (double*)h->cur->llr_ws.p;
    P.osd_list = (int*)h->cur->osd_list.p;
    P.counters = h->cur->d_counters;
    P.iter_total = (unsigned long long*)(h->cur->d_counters + 4);
    P.tail_flag = h->tail_gate ? h->cur->h_tail : nullptr;
    P.packed_io = h->packed_now ? 1 : 0;

    // At most TWO calls have kernels on the device: this call's BP kernel waits for the END of the call two back (its OSD
    // kernel included).  (i) A third BP kernel that became ready meanwhile would share the slots the first one frees with the
    // second from the first workgroup on -- both then take twice as long and finish together.  (ii) The OSD kernel needs a
    // whole CU (131 KB of LDS for H1922); while ANY persistent BP grid is waiting, every slot a draining CU frees goes to a BP
    // workgroup, which fits, and the eliminations of call k starve until the pipeline runs dry (traced with three and four
    // asynchronous host calls in flight: completions came in bursts of three).  With this rule OSD(k) gets its CUs at the
    // start of BP(k + 1)'s tail and BP(k + 2) follows ~1 ms later, whatever the number of calls queued.
    // (Not where BP is HBM-resident: there both kernels need a whole CU, the OSD stream has the higher priority, and three calls in
    // flight are what fills the CUs -- see the lane count in bposd_create.  BPOSD_TWO_BACK=0/1 overrides, for probes.)
    static const char* two_back_env = getenv("BPOSD_TWO_BACK");
    const bool two_back_rule = two_back_env ? two_back_env[0] == '1' : !h->bp_hbm;
    if (!lean && !h->tail_gate && h->nlanes >= 3 && two_back_rule) {  // (the chunks of a synchronous host call are released one by one by the host: tail_gate)
        Lane& two_back = h->lanes[(lane + h->nlanes - 2) % h->nlanes];
        if (two_back.done_recorded) HIP_TRY(h, hipStreamWaitEvent(h->cur->stream, two_back.ev_done, 0));
    }
    if (!lean) HIP_TRY(h, hipEventRecord(h->currec->ev[0], h->cur->stream));
    if (h->cfg.schedule == 1) {
        h->last_bp_kernel = BPOSD_BP_KERNEL_SERIAL;
        if ((rc = launch_bp_serial(h, P))) return rc;
    } else if (h->bp_any || h->bp_variant == 64) {
        h->last_bp_kernel = BPOSD_BP_KERNEL_ANYDEG;
        if ((rc = launch_bp_any(h, P))) return rc;
    } else if (h->bp_hbm) {
        h->last_bp_kernel = BPOSD_BP_KERNEL_LARGE;
        if ((rc = launch_bp_large(h, P))) return rc;
    } else if (h->local_ok && h->cfg.bp_method == BPOSD_BP_MIN_SUM && (h->bp_variant == 0 || (h->bp_variant >= 16 && h->bp_variant <= 26))) {
        h->last_bp_kernel = BPOSD_BP_KERNEL_LOCAL;
        if ((rc = launch_bp_local(h, P))) return rc;
    } else if (h->class_ok && (h->bp_variant == 32 || (h->bp_variant == 0 && class_preferred(h)))) {
        h->last_bp_kernel = BPOSD_BP_KERNEL_CLASS;
        if ((rc = launch_bp_class(h, P))) return rc;
    } else {
        h->last_bp_kernel = BPOSD_BP_KERNEL_LDS;
        if ((rc = launch_bp(h, P))) return rc;
    }
    if (!lean) HIP_TRY(h, hipEventRecord(h->currec->ev[1], h->cur->stream));
    h->currec->ran_osd = false;
    if (!lean && (osd_on || h->tail_gate)) HIP_TRY(h, hipEventRecord(h->cur->ev_bp, h->cur->stream));  // the BP kernel has ended
    if (osd_on) {
        if (!lean) HIP_TRY(h, hipStreamWaitEvent(h->cur->osd_stream, h->cur->ev_bp, 0));
        OsdParams Q{};
        Q.m = h->m;
        Q.n = h->n;
        Q.rank = h->rank;
        Q.osd_method = h->cfg.osd_order == 0 ? BPOSD_OSD_0 : h->cfg.osd_method;
        Q.osd_order = h->cfg.osd_order;
        Q.tie_policy = h->cfg.sort_tie_policy;
        Q.e_msb_first = h->cfg.osd_e_bit_order;
        Q.synd = d_synd;
        Q.rp = h->d_rp;
        Q.ci = h->d_ci;
        Q.llr_ws = (const double*)h->cur->llr_ws.p;
        Q.osd_list = (const int*)h->cur->osd_list.p;
        Q.counters = h->cur->d_counters;
        Q.out_osd0 = d_osd0;
        Q.out_osdw = d_osdw;
        Q.cmp_osd0 = d_osd0 ? h->cmp_osd0 : nullptr;
        Q.cmp_osdw = h->cmp_osdw;
        Q.dbg = nullptr;
        Q.packed_io = h->packed_now ? 1 : 0;
        Q.cost = (h->fp_weights || (d_sel && h->cfg.weight_fn == 0)) ? h->d_cost : nullptr;
        Q.sel = d_sel;
        Q.cost_alt = h->lane_alt ? h->cur->d_alt + h->n : h->d_cost_alt;
        const char* dbg_env = getenv("BPOSD_OSD_DEBUG");
        if (dbg_env && dbg_env[0] == '1') {
            if (!h->cur->d_osd_dbg) HIP_TRY(h, hipMalloc((void**)&h->cur->d_osd_dbg, 8192 * sizeof(long long)));
            HIP_TRY(h, hipMemsetAsync(h->cur->d_osd_dbg, 0, 8192 * sizeof(long long), h->osd_now));
            Q.dbg = h->cur->d_osd_dbg;
        }
        if (h->large) {
            h->last_osd_kernel = 3;
            if ((rc = launch_osd_large(h, Q, B, nullptr))) return rc;
            if (Q.dbg) {
                long long st[28];
                HIP_TRY(h, hipStreamSynchronize(h->osd_now));
                HIP_TRY(h, hipMemcpy(st, h->cur->d_osd_dbg, sizeof(st), hipMemcpyDeviceToHost));
                fprintf(stderr, "[bposd large osd, sparse apply passes %lld: %lld ticks, %lld listed rows, %lld mask bits; word of the last search column %lld]\n", st[12], st[24], st[25], st[26], st[27]);
                fprintf(stderr, "[bposd large osd, s_memtime ticks, list slot 0] sort %lld  build %lld  E1 %lld  E2 %lld  E3 %lld  apply %lld  "
                        "sweep %lld (back-substitution %lld, column vectors %lld, candidates %lld, write-out %lld) | words %lld groups %lld applies %lld | apply look-ups/thread %lld row-words/thread %lld | apply pass: row walks %lld, wait for the slowest walker %lld, own table build %lld, wait for the builders %lld, list builds %lld\n", st[0], st[1], st[2], st[3], st[4], st[5] + st[17] + st[18] + st[19] + st[20], st[6], st[13], st[14], st[15], st[16], st[7], st[8], st[9], st[10], st[11], st[5], st[19], st[17], st[18], st[20]);
                {   // every elimination of the launch (osd_large_kernel writes 16 numbers per list slot behind the first 32)
                    static long long all[8192];
                    HIP_TRY(h, hipMemcpy(all, h->cur->d_osd_dbg, sizeof(all), hipMemcpyDeviceToHost));
                    std::vector<std::array<long long, 16>> v;
                    for (int i = 0; i < 500; ++i)
                        if (all[32 + i * 16] > 0) {
                            std::array<long long, 16> a;
                            for (int k = 0; k < 16; ++k) a[k] = all[32 + i * 16 + k];
                            v.push_back(a);
                        }
                    if (v.size() > 1) {
                        std::sort(v.begin(), v.end());
                        fprintf(stderr, "[bposd large osd, all %zu eliminations of the launch, sorted by ticks] M ticks: total | sort build E2 E3 row-walks sweep | words groups applies | own-table-build E1c E2c-one-wave sparse-apply-passes | wave 0's pivot search alone (E2 column = the rest of the panel phase; E2c-one-wave = the wait for the other waves' share of E3 after it)\n", v.size());
                        for (size_t i = 0; i < v.size(); i += (i + 8 < v.size() ? v.size() / 8 : 1)) {
                            const auto& a = v[i];
                            fprintf(stderr, "  [%3zu] %.0f | %.1f %.1f %.1f %.1f %.1f %.1f | %lld %lld %lld | %.1f %.1f %.1f %.1f | %.1f\n", i, a[0] * 1e-6, a[1] * 1e-6, a[2] * 1e-6, a[4] * 1e-6, a[5] * 1e-6,
                                    a[6] * 1e-6, a[7] * 1e-6, a[8], a[9], a[10], a[12] * 1e-6, a[13] * 1e-6, a[14] * 1e-6, a[15] * 1e-6, a[3] * 1e-6);
                        }
                    }
                }
                Q.dbg = nullptr;
            }
        } else if ((rc = launch_osd(h, Q, B))) return rc;
        h->currec->ran_osd = true;
        if (Q.dbg) {
            static long long st[2048];
            HIP_TRY(h, hipStreamSynchronize(h->osd_now));
            HIP_TRY(h, hipMemcpy(st, h->cur->d_osd_dbg, sizeof(st), hipMemcpyDeviceToHost));
            if (const char* dump = getenv("BPOSD_OSD_DUMP")) {
                if (FILE* f = fopen(dump, "wb")) { fwrite(st, sizeof(long long), 2048, f); fclose(f); }
            }
            fprintf(stderr, "[bposd osd phases, s_memtime ticks] sort %lld  rowbuild %lld  eliminate %lld  osd0 %lld  sweep %lld  write %lld\n",
                    st[1] - st[0], st[2] - st[1], st[3] - st[2], st[4] - st[3], st[5] - st[4], st[6] - st[5]);
            fprintf(stderr, "[bposd osd elimination] panel phase %lld (claims + barrier %lld, solve + tables %lld of which the six steps %lld, absorb %lld)  trailing phase %lld (publish %lld, tables %lld)  pivots %lld\n", st[1190], st[1195], st[1196], st[1189], st[1197], st[1191], st[1193], st[1194], st[1192]);
        }
    }
    if (osd_on && !lean) {  // whatever follows on the lane's stream comes after the OSD kernel
        HIP_TRY(h, hipEventRecord(h->cur->ev_osd, h->cur->osd_stream));
        HIP_TRY(h, hipStreamWaitEvent(h->cur->stream, h->cur->ev_osd, 0));
    }
    if (!lean) {
        HIP_TRY(h, hipEventRecord(h->currec->ev[2], h->cur->stream));
        HIP_TRY(h, hipEventRecord(h->cur->ev_done, h->cur->stream));  // both kernels of this call have ended
        h->cur->done_recorded = true;
    }
    HIP_TRY(h, hipMemcpyAsync(h->currec->h_counters, h->cur->d_counters, 32, hipMemcpyDeviceToHost, h->cur->stream));
    h->osd_now = nullptr;
    h->currec->recorded = true;
    h->currec->timed = !lean;
    h->have_timing = true;
    return BPOSD_OK;
}

int bposd_decode_batch_device(bposd_handle* h, const uint8_t* d_synd, int64_t B, uint8_t* d_osdw,
                              uint8_t* d_osd0, uint8_t* d_bp, uint8_t* d_conv, int32_t* d_iters,
                              double* d_llr) {
    return decode_device_impl(h, d_synd, B, nullptr, d_osdw, d_osd0, d_bp, d_conv, d_iters, d_llr);
}

int bposd_decode_batch_device_packed(bposd_handle* h, const uint64_t* d_synd_words, int64_t B, uint64_t* d_osdw_words,
                                     uint64_t* d_osd0_words, uint64_t* d_bp_words, uint8_t* d_conv, int32_t* d_iters) {
    if (!h) return BPOSD_ERR_INVALID;
    if (!native_packed(h))
        return fail(h, BPOSD_ERR_UNSUPPORTED, "this code's kernels take byte rows (any-degree or serial-schedule kernel): "
                    "use bposd_decode_batch_device and bposd_pack_rows_device");
    h->packed_now = true;
    const int rc = decode_device_impl(h, (const uint8_t*)d_synd_words, B, nullptr, (uint8_t*)d_osdw_words, (uint8_t*)d_osd0_words,
                                      (uint8_t*)d_bp_words, d_conv, d_iters, nullptr);
    h->packed_now = false;
    return rc;
}

static int alt_channel_tables(bposd_handle* h, const double* alt, double* l0, double* cost) {
    if (!alt) return fail(h, BPOSD_ERR_INVALID, "channel_probs_alt is required");
    for (int i = 0; i < h->n; ++i) {
        if (!(alt[i] >= 0.0 && alt[i] <= 1.0))
            return fail(h, BPOSD_ERR_INVALID, "channel_probs_alt[%d] = %g is not a probability", i, alt[i]);
        l0[i] = std::log((1 - alt[i]) / alt[i]);
        cost[i] = std::log(1 / alt[i]);
    }
    return 0;
}

static int upload_alt_channel(bposd_handle* h, const double* alt) {
    std::vector<double> l0(h->n), cost(h->n);
    { int rca = alt_channel_tables(h, alt, l0.data(), cost.data()); if (rca) return rca; }
    DeviceGuard dev_guard(h->device);
    HIP_TRY(h, dev_guard.err);
    { int rcs = sync_all_lanes(h); if (rcs) return rcs; }  // earlier calls may still read the old tables
    HIP_TRY(h, hipMemcpy(h->d_llr0_alt, l0.data(), sizeof(double) * h->n, hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->d_cost_alt, cost.data(), sizeof(double) * h->n, hipMemcpyHostToDevice));
    return 0;
}

int bposd_decode_batch_select_device(bposd_handle* h, const uint8_t* d_synd, int64_t B, const uint8_t* d_sel,
                                     const double* alt, uint8_t* d_osdw, uint8_t* d_osd0, uint8_t* d_bp,
                                     uint8_t* d_conv, int32_t* d_iters, double* d_llr) {
    if (!h) return BPOSD_ERR_INVALID;
    if (!d_sel) return fail(h, BPOSD_ERR_INVALID, "select is required");
    // asynchronous like the plain device-pointer call: the alternative channel goes to the buffers of the lane this call
    // will run on, through that lane's stream
    DeviceGuard dev_guard(h->device);
    HIP_TRY(h, dev_guard.err);
    Lane& L = h->lanes[h->next_lane];
    const size_t bytes = sizeof(double) * 2 * (size_t)h->n;
    if (!L.d_alt) {
        HIP_TRY(h, hipMalloc((void**)&L.d_alt, bytes));
        HIP_TRY(h, hipHostMalloc((void**)&L.h_alt, bytes, hipHostMallocDefault));
        HIP_TRY(h, hipEventCreateWithFlags(&L.ev_alt, hipEventDisableTiming));
    }
    if (L.alt_busy) HIP_TRY(h, hipEventSynchronize(L.ev_alt));  // the copy of this lane's previous select call has read the staging block
    { int rca = alt_channel_tables(h, alt, L.h_alt, L.h_alt + h->n); if (rca) return rca; }
    HIP_TRY(h, hipMemcpyAsync(L.d_alt, L.h_alt, bytes, hipMemcpyHostToDevice, L.stream));
    HIP_TRY(h, hipEventRecord(L.ev_alt, L.stream));
    L.alt_busy = true;
    h->lane_alt = true;
    const int rc = decode_device_impl(h, d_synd, B, d_sel, d_osdw, d_osd0, d_bp, d_conv, d_iters, d_llr);
    h->lane_alt = false;
    return rc;
}

static int decode_host_impl(bposd_handle* h, const uint8_t* synd, int64_t B, const uint8_t* sel, uint8_t* osdw,
                            uint8_t* osd0, uint8_t* bp, uint8_t* conv, int32_t* iters, double* llr, bool packed = false);

int bposd_decode_batch(bposd_handle* h, const uint8_t* synd, int64_t B, uint8_t* osdw, uint8_t* osd0,
                       uint8_t* bp, uint8_t* conv, int32_t* iters, double* llr) {
    return decode_host_impl(h, synd, B, nullptr, osdw, osd0, bp, conv, iters, llr);
}

int bposd_decode_batch_packed(bposd_handle* h, const uint64_t* synd_words, int64_t B, uint64_t* osdw_words, uint64_t* osd0_words,
                              uint64_t* bp_words, uint8_t* conv, int32_t* iters) {
    return decode_host_impl(h, (const uint8_t*)synd_words, B, nullptr, (uint8_t*)osdw_words, (uint8_t*)osd0_words, (uint8_t*)bp_words,
                            conv, iters, nullptr, /*packed=*/true);
}

int bposd_posterior_llr(bposd_handle* h, const uint8_t* synd, int64_t B, double* llr, uint8_t* bp, uint8_t* conv, int32_t* iters) {
    if (!h) return BPOSD_ERR_INVALID;
    if (!llr) return fail(h, BPOSD_ERR_INVALID, "llr buffer is required");
    std::vector<uint8_t> scratch;
    if (!bp) { scratch.resize((size_t)std::max<int64_t>(B, 0) * h->n); bp = scratch.data(); }
    h->bp_only = true;  // the osdw slot of the call receives BP's hard decisions (what a decoder with osd_method "osd_off" returns)
    const int rc = decode_host_impl(h, synd, B, nullptr, bp, nullptr, nullptr, conv, iters, llr);
    h->bp_only = false;
    return rc;
}

int bposd_decode_batch_select(bposd_handle* h, const uint8_t* synd, int64_t B, const uint8_t* sel,
                              const double* alt, uint8_t* osdw, uint8_t* osd0, uint8_t* bp, uint8_t* conv,
                              int32_t* iters, double* llr) {
    if (!h) return BPOSD_ERR_INVALID;
    if (!sel) return fail(h, BPOSD_ERR_INVALID, "select is required");
    int rc = upload_alt_channel(h, alt);
    if (rc) return rc;
    return decode_host_impl(h, synd, B, sel, osdw, osd0, bp, conv, iters, llr);
}

// Host-pointer decode: the batch is cut into chunks that alternate between the handle's lanes, so that the upload of
// chunk c + 1, the kernels of chunk c and the download of chunk c - 1 overlap, and the BP workgroups of chunk c + 1 take
// over the CUs that chunk c's stragglers and OSD kernel leave idle.  Within a lane everything is stream-ordered
// (upload, BP, OSD, downloads), so a lane's staging buffers are reused safely two chunks later.  Page-locked host
// buffers (bposd_host_alloc) make the copies asynchronous; with pageable memory the host thread blocks inside each
// copy while the other lane's kernels keep running.
// (inside the chunk loop: what has been enqueued is drained before an error is returned -- downloads into the caller's
// buffers, or into a buffer local to a caller of this function, may be in flight)
#define HIP_TRY_DRAIN(h, expr)                                                                 \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) {                                                                \
            (void)sync_all_lanes(h);                                                           \
            return fail(h, BPOSD_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
        }                                                                                      \
    } while (0)

// packed: synd / osdw / osd0 / bp are rows of ceil(m / 64) resp. ceil(n / 64) little-endian 64-bit words (bit i & 63 of
// word i >> 6 = entry i) -- one eighth of the bytes over PCIe; the device unpacks the syndromes in front of the BP kernel and
// packs the result rows behind it (sel and llr are not offered in this form).
static int decode_host_impl(bposd_handle* h, const uint8_t* synd, int64_t B, const uint8_t* sel, uint8_t* osdw,
                            uint8_t* osd0, uint8_t* bp, uint8_t* conv, int32_t* iters, double* llr, bool packed) {
    if (!h) return BPOSD_ERR_INVALID;
    if (B < 0 || B > 0x7fffffffLL) return fail(h, BPOSD_ERR_INVALID, "batch size %lld out of range", (long long)B);
    if (B == 0) return BPOSD_OK;
    if (!synd || !osdw) return fail(h, BPOSD_ERR_INVALID, "syndromes and osdw buffers are required");
    DeviceGuard dev_guard(h->device);
    HIP_TRY(h, dev_guard.err);
    // the records and lanes are about to be reused: earlier asynchronous calls must have drained
    if (h->async_pending) { int rcs = sync_all_lanes(h); if (rcs) return rcs; }
    // ---- small calls (the reference's one-syndrome `.decode()`): no copy commands at all.  The kernels read the
    // syndromes from, and write every result to, a page-locked staging area that the device addresses directly; the call
    // costs two host memcpys, the launches and one stream synchronisation.
    {
        static const bool zero_copy = !(getenv("BPOSD_ZERO_COPY") && getenv("BPOSD_ZERO_COPY")[0] == '0');
        const size_t n8 = (size_t)h->n, m8 = (size_t)h->m, b8 = (size_t)B;
        auto a64 = [](size_t x) { return (x + 63) & ~(size_t)63; };
        const size_t o_syn = 0, o_sel = o_syn + a64(b8 * m8), o_osdw = o_sel + (sel ? a64(b8 * n8) : 0),
                     o_osd0 = o_osdw + a64(b8 * n8), o_bp = o_osd0 + (osd0 ? a64(b8 * n8) : 0),
                     o_conv = o_bp + (bp ? a64(b8 * n8) : 0), o_it = o_conv + a64(b8), o_llr = o_it + a64(b8 * 4),
                     total = o_llr + (llr ? a64(b8 * n8 * 8) : 0);
        if (zero_copy && !packed && total <= (size_t)1 << 20) {
            Lane& L = h->lanes[0];
            h->cur = &L;
            if (L.h_stage_bytes < total) {
                if (L.h_stage) (void)hipHostFree(L.h_stage);
                L.h_stage = nullptr;
                L.h_stage_bytes = 0;
                const size_t want = std::max<size_t>(total, (size_t)1 << 16);
                HIP_TRY(h, hipHostMalloc(&L.h_stage, want, hipHostMallocMapped));
                L.h_stage_bytes = want;
            }
            unsigned char* st = (unsigned char*)L.h_stage;
            memcpy(st + o_syn, synd, b8 * m8);
            if (sel) memcpy(st + o_sel, sel, b8 * n8);
            int rcz = decode_device_impl(h, st + o_syn, B, sel ? st + o_sel : nullptr, st + o_osdw, osd0 ? st + o_osd0 : nullptr,
                                         bp ? st + o_bp : nullptr, st + o_conv, (int32_t*)(st + o_it),
                                         llr ? (double*)(st + o_llr) : nullptr, 0, 0, /*lean=*/!getenv("BPOSD_OSD_DEBUG"));
            if (rcz) { (void)sync_all_lanes(h); return rcz; }
            h->nrec = 1;
            if (getenv("BPOSD_OSD_DEBUG")) { rcz = sync_all_lanes(h); if (rcz) return rcz; }
            else HIP_TRY(h, hipStreamSynchronize(L.stream));
            memcpy(osdw, st + o_osdw, b8 * n8);
            if (osd0) memcpy(osd0, st + o_osd0, b8 * n8);
            if (bp) memcpy(bp, st + o_bp, b8 * n8);
            if (conv) memcpy(conv, st + o_conv, b8);
            if (iters) memcpy(iters, st + o_it, b8 * 4);
            if (llr) memcpy(llr, st + o_llr, b8 * n8 * 8);
            return BPOSD_OK;
        }
    }
    // Chunks of ~32768 syndromes (measured on the headline workload, 131072 syndromes: 4 chunks on the 4 lanes 31.9 ms,
    // 8 chunks 35.6 ms -- a lane's next chunk waits for the previous one's OSD kernel and download, and every chunk pays
    // its own straggler tail -- 2 chunks 33.0 ms), at most BPOSD_MAX_CHUNKS; BPOSD_HOST_CHUNK overrides the target size.
    long long target = 32768;
    if (const char* e = getenv("BPOSD_HOST_CHUNK")) target = std::max(1LL, atoll(e));
    int nchunks = (int)std::min<long long>(BPOSD_MAX_CHUNKS, std::max<long long>(1, (B + target / 2) / target));
    if (h->large) nchunks = (int)std::min<long long>(nchunks, std::max<long long>(1, B / (4LL * h->num_cu)));
    long long CH = (B + nchunks - 1) / nchunks;  // capacity of a lane's io buffers = the largest chunk
    nchunks = (int)((B + CH - 1) / CH);
    // chunk boundaries: equal sizes, except that a four-chunk call (one chunk per lane) tapers 7 : 7 : 6 : 4 -- what
    // stays exposed at the end of the call is the LAST chunk's download and straggler tail
    std::vector<long long> clo(nchunks + 1);
    for (int c = 0; c <= nchunks; ++c) clo[c] = std::min<long long>((long long)B, (long long)c * CH);
    static const bool taper = !(getenv("BPOSD_HOST_TAPER") && getenv("BPOSD_HOST_TAPER")[0] == '0');
    if (taper && nchunks == 4 && h->nlanes >= 4 && B >= 4096) {
        const long long w[4] = {7, 7, 6, 4};
        long long acc = 0;
        for (int c = 0; c < 4; ++c) { clo[c] = acc; acc += (B * w[c] / 24 + 63) / 64 * 64; }
        clo[4] = B;
        for (int c = 0; c < 4; ++c) clo[c] = std::min<long long>(clo[c], (long long)B);
        CH = 0;
        for (int c = 0; c < 4; ++c) CH = std::max(CH, clo[c + 1] - clo[c]);
    }
    struct HintScope { bposd_handle* h; ~HintScope() { h->batch_hint = 0; } } hint_scope{h};
    h->batch_hint = B;  // kernel variants are chosen for the call, not for a chunk
    const size_t n = (size_t)h->n, m = (size_t)h->m;
    const size_t rsn = packed ? (n + 63) / 64 * 8 : n, rsm = packed ? (m + 63) / 64 * 8 : m;  // host row strides in bytes
    const bool native = packed && native_packed(h);  // the kernels read packed syndromes / write packed rows themselves
    int rc;
    static const bool gate_env = !(getenv("BPOSD_HOST_GATE") && getenv("BPOSD_HOST_GATE")[0] == '0');
    const bool gate = gate_env && nchunks > 1 && h->cfg.schedule == 0;  // (the serial-schedule kernel does not report its tail)
    const bool osd_on = h->cfg.osd_method != BPOSD_OSD_OFF && !h->bp_only;  // (bposd_posterior_llr: no OSD stage, no OSD list)
    // rows of chunk c that the OSD kernel rewrote: from the compact copies into the caller's arrays (the lane is idle)
    auto patch_osd_rows = [&](int c) -> int {
        Lane& L = h->lanes[c % h->nlanes];
        HIP_TRY(h, hipStreamSynchronize(L.stream));       // chunk c's kernels and its counter copy
        HIP_TRY(h, hipStreamSynchronize(L.copy_stream));  // its bulk downloads (the patched rows must land after them)
        L.copy_pending = false;
        if (!osd_on || !h->rec[c].ran_osd) return 0;
        const long long lo = clo[c];
        const int count = h->rec[c].h_counters[1];
        if (count <= 0) return 0;
        std::vector<uint8_t> rows((size_t)count * rsn);
        for (int which = 0; which < 2; ++which) {
            uint8_t* dst = which ? osd0 : osdw;
            if (!dst) continue;
            const void* src = which ? L.io_cmp0.p : L.io_cmpw.p;
            if (packed && !native) {  // the compact rows, packed on the (idle) lane's stream
                int rcp = launch_pack(h, L.stream, (const uint8_t*)src, count, (int)n, (unsigned long long*)L.io_pcmp.p);
                if (rcp) return rcp;
                HIP_TRY(h, hipMemcpyAsync(rows.data(), L.io_pcmp.p, rows.size(), hipMemcpyDeviceToHost, L.stream));
                HIP_TRY(h, hipStreamSynchronize(L.stream));
            } else {
                HIP_TRY(h, hipMemcpy(rows.data(), src, rows.size(), hipMemcpyDeviceToHost));
            }
            for (int k = 0; k < count; ++k) memcpy(dst + ((size_t)lo + (size_t)L.h_list[k]) * rsn, rows.data() + (size_t)k * rsn, rsn);
        }
        return 0;
    };
    for (int c = 0; c < nchunks; ++c) {
        const long long lo = clo[c], cnt = clo[c + 1] - clo[c];
        const int lane = c % h->nlanes;
        if (cnt <= 0) { h->rec[c].recorded = false; h->rec[c].ran_osd = false; continue; }
        Lane& L = h->lanes[lane];
        h->cur = &L;
        const size_t bn = (size_t)cnt * n, bm = (size_t)cnt * m;
        if (c >= h->nlanes && (rc = patch_osd_rows(c - h->nlanes))) { (void)sync_all_lanes(h); return rc; }  // the lane's previous chunk
        if ((rc = ensure(h, L.io_synd, (size_t)CH * m))) { (void)sync_all_lanes(h); return rc; }
        if ((rc = ensure(h, L.io_osdw, (size_t)CH * n))) { (void)sync_all_lanes(h); return rc; }
        if (osd0 && (rc = ensure(h, L.io_osd0, (size_t)CH * n))) { (void)sync_all_lanes(h); return rc; }
        if (bp && (rc = ensure(h, L.io_bp, (size_t)CH * n))) { (void)sync_all_lanes(h); return rc; }
        if (conv && (rc = ensure(h, L.io_conv, (size_t)CH))) { (void)sync_all_lanes(h); return rc; }
        if (iters && (rc = ensure(h, L.io_iters, sizeof(int) * (size_t)CH))) { (void)sync_all_lanes(h); return rc; }
        if (llr && (rc = ensure(h, L.io_llr, sizeof(double) * (size_t)CH * n))) { (void)sync_all_lanes(h); return rc; }
        if (osd_on) {
            if ((rc = ensure(h, L.io_cmpw, (size_t)CH * n))) { (void)sync_all_lanes(h); return rc; }
            if (osd0 && (rc = ensure(h, L.io_cmp0, (size_t)CH * n))) { (void)sync_all_lanes(h); return rc; }
            if (L.h_list_cap < (size_t)CH) {
                if (L.h_list) (void)hipHostFree(L.h_list);
                L.h_list = nullptr; L.h_list_cap = 0;
                HIP_TRY_DRAIN(h, hipHostMalloc((void**)&L.h_list, sizeof(int) * (size_t)CH, hipHostMallocDefault));
                L.h_list_cap = (size_t)CH;
            }
        }
        if (c > 0) HIP_TRY_DRAIN(h, hipStreamWaitEvent(L.stream, h->lanes[(c - 1) % h->nlanes].ev_up, 0));
        if (packed) {
            if ((rc = ensure(h, L.io_psynd, (size_t)CH * rsm))) { (void)sync_all_lanes(h); return rc; }
            if ((rc = ensure(h, L.io_posdw, (size_t)CH * rsn))) { (void)sync_all_lanes(h); return rc; }
            if (osd0 && (rc = ensure(h, L.io_posd0, (size_t)CH * rsn))) { (void)sync_all_lanes(h); return rc; }
            if (bp && (rc = ensure(h, L.io_pbp, (size_t)CH * rsn))) { (void)sync_all_lanes(h); return rc; }
            if (osd_on && (rc = ensure(h, L.io_pcmp, (size_t)CH * rsn))) { (void)sync_all_lanes(h); return rc; }
            HIP_TRY_DRAIN(h, hipMemcpyAsync(L.io_psynd.p, synd + (size_t)lo * rsm, (size_t)cnt * rsm, hipMemcpyHostToDevice, L.stream));
        } else {
            HIP_TRY_DRAIN(h, hipMemcpyAsync(L.io_synd.p, synd + (size_t)lo * m, bm, hipMemcpyHostToDevice, L.stream));
        }
        if (sel) {
            if ((rc = ensure(h, L.io_sel, (size_t)CH * n))) { (void)sync_all_lanes(h); return rc; }
            HIP_TRY_DRAIN(h, hipMemcpyAsync(L.io_sel.p, sel + (size_t)lo * n, bn, hipMemcpyHostToDevice, L.stream));
        }
        HIP_TRY_DRAIN(h, hipEventRecord(L.ev_up, L.stream));
        if (packed && !native && (rc = launch_unpack(h, L.stream, (const unsigned long long*)L.io_psynd.p, cnt, (int)m, (uint8_t*)L.io_synd.p))) {
            (void)sync_all_lanes(h);
            return rc;
        }
        // Chunk c's kernels are released when chunk c - 1's BP kernel has handed out its last syndrome (its tail begins; the
        // flag is written by that kernel into page-locked memory) or has ended: the chunks then run in order, each filling
        // the previous one's tail, instead of sharing the CUs from the start and all finishing at the end of the call.
        if (c > 0 && gate) {
            Lane& Pv = h->lanes[(c - 1) % h->nlanes];
            // poll the flag (a plain load from page-locked memory); the runtime is asked only every so often and the
            // thread backs off after a short spin -- a chunk's BP kernel runs for milliseconds
            hipError_t qe = hipErrorNotReady;
            for (unsigned spins = 0; *(volatile int*)Pv.h_tail == 0; ++spins) {
                if ((spins & 63) == 63) {
                    qe = hipEventQuery(Pv.ev_bp);
                    if (qe != hipErrorNotReady) break;
                }
                if (spins < 2000) __builtin_ia32_pause();
                else std::this_thread::sleep_for(std::chrono::microseconds(20));
            }
            if (qe != hipErrorNotReady && qe != hipSuccess) {
                (void)sync_all_lanes(h);
                return fail(h, BPOSD_ERR_HIP, "hipEventQuery failed while waiting for chunk %d: %s", c - 1, hipGetErrorString(qe));
            }
        }
        *(volatile int*)L.h_tail = 0;
        h->tail_gate = true;  // (also makes the call record ev_bp, which the downloads below wait for)
        h->cmp_osdw = osd_on ? (uint8_t*)L.io_cmpw.p : nullptr;
        h->cmp_osd0 = (osd_on && osd0) ? (uint8_t*)L.io_cmp0.p : nullptr;
        h->packed_now = native;
        if (native)
            rc = decode_device_impl(h, (const uint8_t*)L.io_psynd.p, cnt, nullptr, (uint8_t*)L.io_posdw.p, osd0 ? (uint8_t*)L.io_posd0.p : nullptr,
                                    bp ? (uint8_t*)L.io_pbp.p : nullptr, conv ? (uint8_t*)L.io_conv.p : nullptr,
                                    iters ? (int32_t*)L.io_iters.p : nullptr, nullptr, lane, c);
        else
            rc = decode_device_impl(h, (const uint8_t*)L.io_synd.p, cnt, sel ? (const uint8_t*)L.io_sel.p : nullptr,
                                    (uint8_t*)L.io_osdw.p, osd0 ? (uint8_t*)L.io_osd0.p : nullptr,
                                    bp ? (uint8_t*)L.io_bp.p : nullptr, conv ? (uint8_t*)L.io_conv.p : nullptr,
                                    iters ? (int32_t*)L.io_iters.p : nullptr, llr ? (double*)L.io_llr.p : nullptr, lane, c);
        h->packed_now = false;
        h->tail_gate = false;
        h->cmp_osdw = h->cmp_osd0 = nullptr;
        if (rc) { (void)sync_all_lanes(h); return rc; }
        // Downloads: everything the BP kernel wrote is final when it ends, except the osdw / osd0 rows of its non-converged
        // syndromes -- those are patched from the compact copies once the OSD kernel has run.  (Queued behind the OSD kernel
        // on the lane's stream, as in the first version, a chunk's downloads started a whole chunk late: the OSD kernel
        // needs a drained CU and the next chunk's persistent BP workgroups take every slot that frees up.)
        hipStream_t cs = L.copy_stream;
        HIP_TRY_DRAIN(h, hipStreamWaitEvent(cs, L.ev_bp, 0));
        if (packed) {
            // (the pack kernels wait for a free workgroup slot like any kernel: with the next chunk's persistent BP grid
            // resident that is that chunk's tail -- the rows then leave one eighth as large)
            struct { uint8_t* host; const DevBuf* bytes; const DevBuf* words; } outs[3] = {{osdw, &L.io_osdw, &L.io_posdw}, {osd0, &L.io_osd0, &L.io_posd0}, {bp, &L.io_bp, &L.io_pbp}};
            for (auto& o : outs) {
                if (!o.host) continue;
                if (!native && (rc = launch_pack(h, cs, (const uint8_t*)o.bytes->p, cnt, (int)n, (unsigned long long*)o.words->p))) { (void)sync_all_lanes(h); return rc; }
                HIP_TRY_DRAIN(h, hipMemcpyAsync(o.host + (size_t)lo * rsn, o.words->p, (size_t)cnt * rsn, hipMemcpyDeviceToHost, cs));
            }
        } else {
            HIP_TRY_DRAIN(h, hipMemcpyAsync(osdw + (size_t)lo * n, L.io_osdw.p, bn, hipMemcpyDeviceToHost, cs));
            if (osd0) HIP_TRY_DRAIN(h, hipMemcpyAsync(osd0 + (size_t)lo * n, L.io_osd0.p, bn, hipMemcpyDeviceToHost, cs));
            if (bp) HIP_TRY_DRAIN(h, hipMemcpyAsync(bp + (size_t)lo * n, L.io_bp.p, bn, hipMemcpyDeviceToHost, cs));
        }
        if (conv) HIP_TRY_DRAIN(h, hipMemcpyAsync(conv + lo, L.io_conv.p, (size_t)cnt, hipMemcpyDeviceToHost, cs));
        if (iters) HIP_TRY_DRAIN(h, hipMemcpyAsync(iters + lo, L.io_iters.p, sizeof(int) * (size_t)cnt, hipMemcpyDeviceToHost, cs));
        if (llr) HIP_TRY_DRAIN(h, hipMemcpyAsync(llr + (size_t)lo * n, L.io_llr.p, sizeof(double) * bn, hipMemcpyDeviceToHost, cs));
        if (osd_on) HIP_TRY_DRAIN(h, hipMemcpyAsync(L.h_list, L.osd_list.p, sizeof(int) * (size_t)cnt, hipMemcpyDeviceToHost, cs));
        L.copy_pending = true;
    }
    for (int c = std::max(0, nchunks - h->nlanes); c < nchunks; ++c)
        if ((rc = patch_osd_rows(c))) { (void)sync_all_lanes(h); return rc; }
    h->nrec = nchunks;
    return sync_all_lanes(h);
}

// Asynchronous host-pointer decode: ONE lane, everything in stream order -- upload (packed rows: + unpack kernel), BP,
// OSD, (pack kernels,) downloads -- and the call returns once that is enqueued.  Consecutive calls take consecutive lanes,
// so call k + 1's BP workgroups fill the straggler tail of call k and call k's OSD kernel, packing and downloads run
// under call k + 1's BP kernel: the host-to-host rate of a stream of batches approaches the device-resident one (a lone
// synchronous call always pays its own upload, its 1922-iteration tail and its download).  The caller's buffers must be
// page-locked (bposd_host_alloc) for the copies to be asynchronous, and stay untouched until bposd_synchronize_lane().
static int decode_host_async_impl(bposd_handle* h, const uint8_t* synd, int64_t B, uint8_t* osdw, uint8_t* osd0, uint8_t* bp,
                                  uint8_t* conv, int32_t* iters, double* llr, bool packed) {
    if (!h) return BPOSD_ERR_INVALID;
    if (B < 0 || B > 0x7fffffffLL) return fail(h, BPOSD_ERR_INVALID, "batch size %lld out of range", (long long)B);
    if (B == 0) return BPOSD_OK;
    if (!synd || !osdw) return fail(h, BPOSD_ERR_INVALID, "syndromes and osdw buffers are required");
    if (packed && llr) return fail(h, BPOSD_ERR_INVALID, "the packed form has no LLR output");
    DeviceGuard dev_guard(h->device);
    HIP_TRY(h, dev_guard.err);
    // a synchronous host-pointer call leaves no work behind, but its per-chunk records and staging are per lane too: nothing
    // to drain here.  Buffers grow on every lane at once (no allocation inside a later call of the same size).
    const size_t n = (size_t)h->n, m = (size_t)h->m, b8 = (size_t)B;
    const size_t rsn = packed ? (n + 63) / 64 * 8 : n, rsm = packed ? (m + 63) / 64 * 8 : m;
    int rc;
    bool grew = false;
    auto need = [&](DevBuf Lane::*member, size_t bytes) -> int {
        bool have = true;
        for (int l = 0; l < h->nlanes; ++l) have = have && (h->lanes[l].*member).p && (h->lanes[l].*member).bytes >= bytes;
        if (have) return 0;
        if (!grew && h->async_pending) { int rcs = sync_all_lanes(h); if (rcs) return rcs; }  // earlier calls may still use the old buffers
        grew = true;
        return ensure_lanes(h, member, bytes);
    };
    if ((rc = need(&Lane::io_synd, b8 * m))) return rc;
    if ((rc = need(&Lane::io_osdw, b8 * n))) return rc;
    if (osd0 && (rc = need(&Lane::io_osd0, b8 * n))) return rc;
    if (bp && (rc = need(&Lane::io_bp, b8 * n))) return rc;
    if ((rc = need(&Lane::io_conv, b8))) return rc;
    if ((rc = need(&Lane::io_iters, sizeof(int) * b8))) return rc;
    if (llr && (rc = need(&Lane::io_llr, sizeof(double) * b8 * n))) return rc;
    if (packed) {
        if ((rc = need(&Lane::io_psynd, b8 * rsm))) return rc;
        if ((rc = need(&Lane::io_posdw, b8 * rsn))) return rc;
        if (osd0 && (rc = need(&Lane::io_posd0, b8 * rsn))) return rc;
        if (bp && (rc = need(&Lane::io_pbp, b8 * rsn))) return rc;
    }
    Lane& L = h->lanes[h->next_lane];  // the lane decode_device_impl is about to take
    if (L.copy_pending) { HIP_TRY(h, hipStreamSynchronize(L.copy_stream)); L.copy_pending = false; }
    // The copies and the pack / unpack kernels go to the lane's HIGH-PRIORITY stream (the one its OSD kernel runs on), ordered
    // against the lane's main stream by events: when workgroup slots free up in the tail of another call's BP kernel these small
    // kernels are dispatched first instead of competing with the next persistent BP grid for every slot.
    hipStream_t hs = L.osd_stream;
    HIP_TRY(h, hipEventRecord(L.ev_copy, L.stream));   // the lane's previous call (its downloads included) has finished
    HIP_TRY(h, hipStreamWaitEvent(hs, L.ev_copy, 0));
    const bool native = packed && native_packed(h);  // the kernels read packed syndromes / write packed rows themselves
    if (packed) {
        HIP_TRY(h, hipMemcpyAsync(L.io_psynd.p, synd, b8 * rsm, hipMemcpyHostToDevice, hs));
        if (!native && (rc = launch_unpack(h, hs, (const unsigned long long*)L.io_psynd.p, B, (int)m, (uint8_t*)L.io_synd.p))) return rc;
    } else {
        HIP_TRY(h, hipMemcpyAsync(L.io_synd.p, synd, b8 * m, hipMemcpyHostToDevice, hs));
    }
    HIP_TRY(h, hipEventRecord(L.ev_up, hs));
    HIP_TRY(h, hipStreamWaitEvent(L.stream, L.ev_up, 0));
    h->packed_now = native;
    if (native)
        rc = decode_device_impl(h, (const uint8_t*)L.io_psynd.p, B, nullptr, (uint8_t*)L.io_posdw.p, osd0 ? (uint8_t*)L.io_posd0.p : nullptr,
                                bp ? (uint8_t*)L.io_pbp.p : nullptr, (uint8_t*)L.io_conv.p, (int32_t*)L.io_iters.p, nullptr);
    else
        rc = decode_device_impl(h, (const uint8_t*)L.io_synd.p, B, nullptr, (uint8_t*)L.io_osdw.p, osd0 ? (uint8_t*)L.io_osd0.p : nullptr,
                                bp ? (uint8_t*)L.io_bp.p : nullptr, (uint8_t*)L.io_conv.p, (int32_t*)L.io_iters.p,
                                llr ? (double*)L.io_llr.p : nullptr);
    h->packed_now = false;
    if (rc) { (void)sync_all_lanes(h); return rc; }
    // (decode_device_impl has made L.stream wait for the OSD kernel: an event on it covers both kernels)
    HIP_TRY_DRAIN(h, hipEventRecord(L.ev_copy, L.stream));
    HIP_TRY_DRAIN(h, hipStreamWaitEvent(hs, L.ev_copy, 0));
    struct { uint8_t* host; const DevBuf* bytes; const DevBuf* words; } outs[3] = {{osdw, &L.io_osdw, &L.io_posdw}, {osd0, &L.io_osd0, &L.io_posd0}, {bp, &L.io_bp, &L.io_pbp}};
    if (conv) HIP_TRY_DRAIN(h, hipMemcpyAsync(conv, L.io_conv.p, b8, hipMemcpyDeviceToHost, hs));
    if (iters) HIP_TRY_DRAIN(h, hipMemcpyAsync(iters, L.io_iters.p, sizeof(int) * b8, hipMemcpyDeviceToHost, hs));
    for (auto& o : outs) {
        if (!o.host) continue;
        if (packed) {
            if (!native && (rc = launch_pack(h, hs, (const uint8_t*)o.bytes->p, B, (int)n, (unsigned long long*)o.words->p))) { (void)sync_all_lanes(h); return rc; }
            HIP_TRY_DRAIN(h, hipMemcpyAsync(o.host, o.words->p, b8 * rsn, hipMemcpyDeviceToHost, hs));
        } else {
            HIP_TRY_DRAIN(h, hipMemcpyAsync(o.host, o.bytes->p, b8 * n, hipMemcpyDeviceToHost, hs));
        }
    }
    if (llr) HIP_TRY_DRAIN(h, hipMemcpyAsync(llr, L.io_llr.p, sizeof(double) * b8 * n, hipMemcpyDeviceToHost, hs));
    HIP_TRY_DRAIN(h, hipEventRecord(L.ev_osd, hs));               // bposd_synchronize_lane waits on the main stream:
    HIP_TRY_DRAIN(h, hipStreamWaitEvent(L.stream, L.ev_osd, 0));  // make it cover the downloads
    return BPOSD_OK;
}

int bposd_decode_batch_async(bposd_handle* h, const uint8_t* synd, int64_t B, uint8_t* osdw, uint8_t* osd0, uint8_t* bp,
                             uint8_t* conv, int32_t* iters, double* llr) {
    return decode_host_async_impl(h, synd, B, osdw, osd0, bp, conv, iters, llr, /*packed=*/false);
}

int bposd_decode_batch_packed_async(bposd_handle* h, const uint64_t* synd_words, int64_t B, uint64_t* osdw_words, uint64_t* osd0_words,
                                    uint64_t* bp_words, uint8_t* conv, int32_t* iters) {
    return decode_host_async_impl(h, (const uint8_t*)synd_words, B, (uint8_t*)osdw_words, (uint8_t*)osd0_words, (uint8_t*)bp_words, conv,
                                  iters, nullptr, /*packed=*/true);
}

static int record_timing(bposd_handle* h, CallRecord* recs, int count, double* bp_ms, double* osd_ms,
                         int64_t* bp_iterations, int64_t* osd_invocations) {
    double a_sum = 0.0, b_sum = 0.0;
    int64_t it_sum = 0, osd_sum = 0;
    for (int r = 0; r < count; ++r) {
        CallRecord& R = recs[r];
        if (!R.recorded) continue;
        float a = 0.f, b = 0.f;
        if (R.timed) {  // (the lean small-call path records counters only: its times read 0)
            HIP_TRY(h, hipEventElapsedTime(&a, R.ev[0], R.ev[1]));
            HIP_TRY(h, hipEventElapsedTime(&b, R.ev[1], R.ev[2]));
        }
        a_sum += a;
        if (R.ran_osd) b_sum += b;
        it_sum += (int64_t)*R.h_iter_total;
        osd_sum += R.h_counters[1];
    }
    if (bp_ms) *bp_ms = a_sum;
    if (osd_ms) *osd_ms = b_sum;
    if (bp_iterations) *bp_iterations = it_sum;
    if (osd_invocations) *osd_invocations = osd_sum;
    return BPOSD_OK;
}

int bposd_last_timing(bposd_handle* h, double* bp_ms, double* osd_ms, int64_t* bp_iterations,
                      int64_t* osd_invocations) {
    if (!h) return BPOSD_ERR_INVALID;
    if (!h->have_timing) return fail(h, BPOSD_ERR_INVALID, "no decode call has been made on this handle");
    DeviceGuard dev_guard(h->device);
    HIP_TRY(h, dev_guard.err);
    if (h->nrec > 0) {
        // a host-pointer call is several kernel pairs (one per chunk); their durations are summed (chunks overlap on
        // the device, so the sum can exceed the call's wall time), the counters add up to the batch's totals
        int rcs = sync_all_lanes(h);
        if (rcs) return rcs;
        return record_timing(h, h->rec, h->nrec, bp_ms, osd_ms, bp_iterations, osd_invocations);
    }
    HIP_TRY(h, hipStreamSynchronize(h->lanes[h->last_lane].stream));
    return record_timing(h, &h->lane_rec[h->last_lane], 1, bp_ms, osd_ms, bp_iterations, osd_invocations);
}

int bposd_num_lanes(bposd_handle* h) { return h ? h->nlanes : BPOSD_LANES; }

int bposd_last_lane(bposd_handle* h) { return h ? h->last_lane : BPOSD_ERR_INVALID; }

int bposd_synchronize_lane(bposd_handle* h, int32_t lane) {
    if (!h) return BPOSD_ERR_INVALID;
    if (lane < 0 || lane >= BPOSD_LANES) return fail(h, BPOSD_ERR_INVALID, "lane %d out of range", lane);
    DeviceGuard dev_guard(h->device);
    HIP_TRY(h, dev_guard.err);
    HIP_TRY(h, hipStreamSynchronize(h->lanes[lane].stream));
    return BPOSD_OK;
}

int bposd_lane_timing(bposd_handle* h, int32_t lane, double* bp_ms, double* osd_ms, int64_t* bp_iterations,
                      int64_t* osd_invocations) {
    if (!h) return BPOSD_ERR_INVALID;
    if (lane < 0 || lane >= BPOSD_LANES) return fail(h, BPOSD_ERR_INVALID, "lane %d out of range", lane);
    if (!h->have_timing) return fail(h, BPOSD_ERR_INVALID, "no decode call has been made on this handle");
    DeviceGuard dev_guard(h->device);
    HIP_TRY(h, dev_guard.err);
    HIP_TRY(h, hipStreamSynchronize(h->lanes[lane].stream));
    return record_timing(h, &h->lane_rec[lane], 1, bp_ms, osd_ms, bp_iterations, osd_invocations);
}

int bposd_debug_local_layout(const int32_t* indptr, const int32_t* indices, int32_t m, int32_t n, int64_t* out) {
    // host-only: the ownership / position layout the local-edge BP kernel would use for this pcm.
    // out[0] simulated LDS passes, out[1] ideal passes, out[2] positions in uniform groups, out[3] mixed (group, slot) pairs,
    // out[4] positions MP, out[5..13] class sizes, out[14] modelled ds_write_b64 cycles of the bit pass, out[15] their floor
    if (!indptr || !indices || !out || n != 2 * m) return BPOSD_ERR_INVALID;
    std::vector<int> rp(indptr, indptr + m + 1), ci(indices, indices + indptr[m]);
    const int MP = m <= 1024 ? 1024 : 2048;
    if (m > MP) return BPOSD_ERR_UNSUPPORTED;
    for (int c = 0; c < m; ++c)
        if (rp[c + 1] - rp[c] != 6) return BPOSD_ERR_UNSUPPORTED;
    std::vector<int> deg(n, 0);
    for (int e : ci) {
        if (e < 0 || e >= n) return BPOSD_ERR_INVALID;
        deg[e]++;
    }
    for (int i = 0; i < n; ++i)
        if (deg[i] != 3) return BPOSD_ERR_UNSUPPORTED;
    local_layout::Graph g;
    local_layout::Layout best;
    if (!local_layout_host(rp, ci, m, n, MP, g, best)) return BPOSD_ERR_UNSUPPORTED;
    int mixed = 0;
    for (int gq = 0; gq < MP / 64; ++gq)
        for (int b = 0; b < 2; ++b) {
            int code = -1;
            for (int p = 64 * gq; p < 64 * gq + 64; ++p) {
                const int c = best.pos_chk[p];
                if (c < 0) continue;
                const int d = g.rank_of(best.load[2 * c + b], c);
                code = (code < 0 || code == d) ? d : 3;
            }
            mixed += code == 3;
        }
    out[0] = best.passes; out[1] = 4 * (MP / 32); out[2] = best.nfull; out[3] = mixed; out[4] = MP;
    out[14] = best.wcycles; out[15] = 6 * 4 * (MP / 64);
    for (int k = 0; k < 9; ++k) out[5 + k] = 0;
    for (int c = 0; c < m; ++c) {
        int a = g.rank_of(best.load[2 * c], c), b = g.rank_of(best.load[2 * c + 1], c);
        if (a > b) std::swap(a, b);
        out[5 + a * 3 + b]++;
    }
    return BPOSD_OK;
}

int bposd_debug_class_layout(const int32_t* indptr, const int32_t* indices, int32_t m, int32_t n, int32_t* pos_chk, int32_t* pos_bit,
                             int32_t* bit_slot, int32_t* grp_deg, int32_t* grp_cdeg, int64_t* info) {
    // host-only: the tables bp_class_kernel would be launched with for this pcm (tests check their invariants without a GPU).
    // info[0..10]: DC, DVLO, DVHI, VPT, MP (= NTMAX), threads per workgroup, modelled read cycles, their floor, modelled write cycles, their
    // floor, DCLO
    if (!indptr || !indices || !info || m < 1 || n < 1) return BPOSD_ERR_INVALID;
    std::vector<int> rp(indptr, indptr + m + 1), ci(indices, indices + indptr[m]);
    for (int e : ci)
        if (e < 0 || e >= n) return BPOSD_ERR_INVALID;
    const ClassShape* shp = class_shape_for(rp, ci, m, n);
    if (!shp || m > 1024) return BPOSD_ERR_UNSUPPORTED;
    class_layout::Tables T;
    bool ok = false;
    for (int mp : {256, 512, 1024}) {
        if (m > mp) continue;
        if (class_layout::build(rp, ci, m, n, shp->dclo, shp->dc, shp->dvlo, shp->dvhi, kClassVPT, mp, mp, 50000, T)) { ok = true; break; }
    }
    if (!ok) return BPOSD_ERR_UNSUPPORTED;
    info[0] = shp->dc; info[1] = shp->dvlo; info[2] = shp->dvhi; info[3] = kClassVPT; info[4] = T.MP; info[5] = T.NT;
    info[6] = T.read_cycles; info[7] = T.read_floor; info[8] = T.write_cycles; info[9] = T.write_floor; info[10] = shp->dclo;
    if (pos_chk) std::copy(T.pos_chk.begin(), T.pos_chk.end(), pos_chk);    // [MP]
    if (pos_bit) std::copy(T.pos_bit.begin(), T.pos_bit.end(), pos_bit);    // [VPT * MP]
    if (bit_slot) std::copy(T.bit_slot.begin(), T.bit_slot.end(), bit_slot);  // [DVHI * VPT * MP]
    if (grp_deg) std::copy(T.grp_deg.begin(), T.grp_deg.end(), grp_deg);    // [VPT * MP / 64]
    if (grp_cdeg) std::copy(T.grp_cdeg.begin(), T.grp_cdeg.end(), grp_cdeg);  // [MP / 64]
    return BPOSD_OK;
}

void* bposd_host_alloc(size_t bytes) {
    void* p = nullptr;
    if (bytes == 0 || hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}

void bposd_host_free(void* p) {
    if (p) (void)hipHostFree(p);
}

}  // extern "C"
