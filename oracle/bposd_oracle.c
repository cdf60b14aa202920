/*
 * oracle/bposd_oracle.c -- CPU restatement (plain C, fp64) of the BP+OSD decode path.
 *
 * TEST INFRASTRUCTURE ONLY -- see bposd_oracle.h for the parity status ("parity
 * unpinned": the algorithm lives in the third-party `ldpc>=2.0.0` package that the
 * reference imports at /root/reference/src/bposd/__init__.py:1 and
 * /root/reference/src/bposd/css_decode_sim.py:6 and that is absent here).
 *
 * Each function names the reference call site whose behaviour it restates and the
 * SURVEY.md Appendix A clause (the behavioural spec of upstream ldpc v2) it follows.
 * The OSD part is deliberately written the "literal" way -- one explicit linear
 * solve, one explicit solution vector and one explicit weight sum per candidate --
 * so that it shares no algorithmic shortcut with the GPU kernels it checks.
 */
#include "bposd_oracle.h"
/* the one file shared with the product: the bit-reproducible tanh / log the GPU kernels evaluate (ps_math = 1) */
#include "../bp_osd_amd/csrc/portable_math.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

struct oracle_decoder {
    oracle_config cfg;
    int m, n, E;
    int *rp, *ci;   /* CSR: edges e in [rp[c], rp[c+1]) ascending column           */
    int *cp, *ce;   /* CSC: edge ids of column i, ascending row                     */
    int *erow;      /* row of edge e                                                */
    double *p;      /* channel probabilities                                        */
    double *llr0;   /* prior log-likelihood ratios                                  */
    double *b2c, *c2b, *llr;
    uint8_t *dec, *cand;
    int max_iter;
    int rank, kprime;
    /* OSD scratch */
    int W, WA;
    uint64_t *U, *L, *tmpU, *tmpL;
    uint64_t *tp, *xp, *tvec;
    int *order, *pivcol, *nonpiv, *sort_tmp;
    uint8_t *sol, *best, *tsyn;
    uint8_t *ispiv;
    /* diagnostics of the last BP run */
    int diag_first_nonfinite, diag_has_inf, diag_has_nan;
};

static int parity64(uint64_t x) { return __builtin_parityll(x); }

/* ---------------------------------------------------------------------------------
 * construction  (a1: README.md:178-187, css_decode_sim.py:444-463; Appendix A.1)
 * ------------------------------------------------------------------------------- */
static void set_priors(oracle_decoder *d) {
    /* Appendix A.3 / a3: prior LLR = log((1-p)/p) */
    for (int i = 0; i < d->n; i++) d->llr0[i] = log((1 - d->p[i]) / d->p[i]);
}

static int gf2_rank_of_H(oracle_decoder *d);

int oracle_create(const oracle_config *cfg, const int32_t *indptr, const int32_t *indices,
                  int32_t m, int32_t n, const double *channel_probs, oracle_decoder **out) {
    if (!cfg || !indptr || !indices || !channel_probs || !out || m <= 0 || n <= 0) return -1;
    oracle_decoder *d = (oracle_decoder *)calloc(1, sizeof(*d));
    if (!d) return -2;
    d->cfg = *cfg;
    d->m = m;
    d->n = n;
    d->E = indptr[m];
    int E = d->E;
    d->rp = (int *)malloc(sizeof(int) * (m + 1));
    d->ci = (int *)malloc(sizeof(int) * (E > 0 ? E : 1));
    d->cp = (int *)calloc(n + 1, sizeof(int));
    d->ce = (int *)malloc(sizeof(int) * (E > 0 ? E : 1));
    d->erow = (int *)malloc(sizeof(int) * (E > 0 ? E : 1));
    d->p = (double *)malloc(sizeof(double) * n);
    d->llr0 = (double *)malloc(sizeof(double) * n);
    d->b2c = (double *)malloc(sizeof(double) * (E > 0 ? E : 1));
    d->c2b = (double *)malloc(sizeof(double) * (E > 0 ? E : 1));
    d->llr = (double *)malloc(sizeof(double) * n);
    d->dec = (uint8_t *)malloc(n);
    d->cand = (uint8_t *)malloc(m);
    memcpy(d->rp, indptr, sizeof(int) * (m + 1));
    memcpy(d->ci, indices, sizeof(int) * E);
    for (int c = 0; c < m; c++) {
        for (int e = d->rp[c]; e < d->rp[c + 1]; e++) {
            if (d->ci[e] < 0 || d->ci[e] >= n) { oracle_destroy(d); return -3; }
            if (e > d->rp[c] && d->ci[e] <= d->ci[e - 1]) { oracle_destroy(d); return -4; }
            d->erow[e] = c;
            d->cp[d->ci[e] + 1]++;
        }
    }
    for (int i = 0; i < n; i++) d->cp[i + 1] += d->cp[i];
    int *fill = (int *)calloc(n, sizeof(int));
    for (int e = 0; e < E; e++) { /* ascending e == ascending row within a column */
        int i = d->ci[e];
        d->ce[d->cp[i] + fill[i]++] = e;
    }
    free(fill);
    memcpy(d->p, channel_probs, sizeof(double) * n);
    set_priors(d);
    /* Appendix A.1: max_iter == 0 => block length n */
    d->max_iter = cfg->max_iter > 0 ? cfg->max_iter : n;

    d->W = (n + 63) / 64;
    d->WA = (m + 63) / 64;
    d->U = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)m * d->W);
    d->L = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)m * d->WA);
    d->tmpU = (uint64_t *)malloc(sizeof(uint64_t) * d->W);
    d->tmpL = (uint64_t *)malloc(sizeof(uint64_t) * d->WA);
    d->tp = (uint64_t *)malloc(sizeof(uint64_t) * d->WA);
    d->tvec = (uint64_t *)malloc(sizeof(uint64_t) * d->WA);
    d->xp = (uint64_t *)malloc(sizeof(uint64_t) * d->W);
    d->order = (int *)malloc(sizeof(int) * n);
    d->sort_tmp = (int *)malloc(sizeof(int) * n);
    d->pivcol = (int *)malloc(sizeof(int) * (m < n ? m : n));
    d->nonpiv = (int *)malloc(sizeof(int) * n);
    d->sol = (uint8_t *)malloc(n);
    d->best = (uint8_t *)malloc(n);
    d->tsyn = (uint8_t *)malloc(m);
    d->ispiv = (uint8_t *)malloc(n);

    /* Appendix A.1 [M]: ctor eliminates H once to learn rank, k' = n - rank */
    d->rank = gf2_rank_of_H(d);
    d->kprime = n - d->rank;
    if (cfg->osd_method >= 2 && cfg->osd_order > d->kprime) { oracle_destroy(d); return -5; }
    if (cfg->osd_method == 2 && cfg->osd_order > 24) { oracle_destroy(d); return -6; }
    *out = d;
    return 0;
}

void oracle_destroy(oracle_decoder *d) {
    if (!d) return;
    free(d->rp); free(d->ci); free(d->cp); free(d->ce); free(d->erow); free(d->p);
    free(d->llr0); free(d->b2c); free(d->c2b); free(d->llr); free(d->dec); free(d->cand);
    free(d->U); free(d->L); free(d->tmpU); free(d->tmpL); free(d->tp); free(d->tvec);
    free(d->xp); free(d->order); free(d->sort_tmp); free(d->pivcol); free(d->nonpiv);
    free(d->sol); free(d->best); free(d->tsyn); free(d->ispiv);
    free(d);
}

/* a12: update_channel_probs -- css_decode_sim.py:229,248 */
int oracle_update_channel_probs(oracle_decoder *d, const double *channel_probs) {
    if (!d || !channel_probs) return -1;
    memcpy(d->p, channel_probs, sizeof(double) * d->n);
    set_priors(d);
    return 0;
}

int oracle_rank(const oracle_decoder *d) { return d->rank; }

int oracle_num_candidates(const oracle_decoder *d) {
    int w = d->cfg.osd_order;
    if (d->cfg.osd_method <= 1 || w == 0) return 0;
    if (d->cfg.osd_method == 2) return (1 << w) - 1;
    return d->kprime + w * (w - 1) / 2;
}

/* ---------------------------------------------------------------------------------
 * BP, parallel (flooding) schedule  (a3-a7; Appendix A.3)
 * ------------------------------------------------------------------------------- */
static void bp_decode(oracle_decoder *d, const uint8_t *syn, uint8_t *converged, int32_t *iters) {
    const int m = d->m, n = d->n;
    int conv = 0, it_done = 0;
    d->diag_first_nonfinite = d->diag_has_inf = d->diag_has_nan = 0;
    /* a3: every edge's bit->check message starts at the prior */
    for (int i = 0; i < n; i++)
        for (int k = d->cp[i]; k < d->cp[i + 1]; k++) d->b2c[d->ce[k]] = d->llr0[i];

    for (int it = 1; it <= d->max_iter; it++) {
        if (d->cfg.bp_method == 0) {
            /* a5 product-sum: forward/backward partial products of tanh(b2c/2) */
            const int pm = d->cfg.ps_math;  /* 0 libm, 1 portable routines with two divisions per edge, 2 portable routines in the reference's operation order */
            for (int c = 0; c < m; c++) {
                d->cand[c] = 0;
                double temp = 1.0;
                for (int e = d->rp[c]; e < d->rp[c + 1]; e++) {
                    d->c2b[e] = temp;
                    temp *= pm ? pm_ps_tanh_half(d->b2c[e], pm == 1) : tanh(d->b2c[e] / 2);
                }
                temp = 1;
                for (int e = d->rp[c + 1] - 1; e >= d->rp[c]; e--) {
                    d->c2b[e] *= temp;
                    int message_sign = syn[c] ? -1 : 1;
                    /* ps_math = 1: the kernels' routines (portable_math.h, "round 4": same operation order, the quotient folded
                     * into the logarithm) */
                    d->c2b[e] = message_sign * (pm ? pm_ps_log_ratio(d->c2b[e], pm == 1) : log((1 + d->c2b[e]) / (1 - d->c2b[e])));
                    if (d->cfg.ps_clip > 0) { /* build-owned switch; upstream does not clip (Appendix A.3 [M]) */
                        if (d->c2b[e] > d->cfg.ps_clip) d->c2b[e] = d->cfg.ps_clip;
                        if (d->c2b[e] < -d->cfg.ps_clip) d->c2b[e] = -d->cfg.ps_clip;
                    }
                    if (!d->diag_first_nonfinite && !isfinite(d->c2b[e])) d->diag_first_nonfinite = it;
                    temp *= pm ? pm_ps_tanh_half(d->b2c[e], pm == 1) : tanh(d->b2c[e] / 2);
                }
            }
        } else {
            /* a4 min-sum; alpha = 1 - 2^-it when the scaling factor is 0 (README.md:184) */
            double alpha;
            if (d->cfg.ms_scaling_factor == 0.0) alpha = 1.0 - pow(2.0, -1.0 * it);
            else alpha = d->cfg.ms_scaling_factor;
            for (int c = 0; c < m; c++) {
                d->cand[c] = 0;
                int total_sgn = syn[c], sgn;
                double temp = DBL_MAX;
                for (int e = d->rp[c]; e < d->rp[c + 1]; e++) {
                    if (d->b2c[e] <= 0) total_sgn += 1;
                    d->c2b[e] = temp;
                    double a = fabs(d->b2c[e]);
                    if (a < temp) temp = a;
                }
                temp = DBL_MAX;
                for (int e = d->rp[c + 1] - 1; e >= d->rp[c]; e--) {
                    sgn = total_sgn;
                    if (d->b2c[e] <= 0) sgn += 1;
                    if (temp < d->c2b[e]) d->c2b[e] = temp;
                    int message_sign = (sgn % 2 == 0) ? 1 : -1;
                    d->c2b[e] *= message_sign * alpha;
                    double a = fabs(d->b2c[e]);
                    if (a < temp) temp = a;
                }
            }
        }
        /* a6: posterior (prefix sums from the top of the column), hard decision,
         * candidate syndrome */
        for (int i = 0; i < n; i++) {
            double temp = d->llr0[i];
            for (int k = d->cp[i]; k < d->cp[i + 1]; k++) {
                int e = d->ce[k];
                d->b2c[e] = temp;
                temp += d->c2b[e];
            }
            d->llr[i] = temp;
            if (!d->diag_first_nonfinite && !isfinite(temp)) d->diag_first_nonfinite = it;
            if (temp <= 0) {
                d->dec[i] = 1;
                for (int k = d->cp[i]; k < d->cp[i + 1]; k++) d->cand[d->erow[d->ce[k]]] ^= 1;
            } else {
                d->dec[i] = 0;
            }
        }
        conv = memcmp(d->cand, syn, m) == 0;
        it_done = it;
        if (conv) break;
        /* a7: bit->check completion, suffix sums from the bottom of the column */
        for (int i = 0; i < n; i++) {
            double temp = 0;
            for (int k = d->cp[i + 1] - 1; k >= d->cp[i]; k--) {
                int e = d->ce[k];
                d->b2c[e] += temp;
                temp += d->c2b[e];
            }
        }
    }
    for (int i = 0; i < n && it_done > 0; i++) {
        if (isnan(d->llr[i])) d->diag_has_nan = 1;
        else if (isinf(d->llr[i])) d->diag_has_inf = 1;
    }
    *converged = (uint8_t)conv;
    *iters = it_done;
}

int oracle_last_bp_diag(const oracle_decoder *d, int32_t *first_nonfinite_iter, int32_t *final_has_inf,
                        int32_t *final_has_nan) {
    if (!d) return -1;
    if (first_nonfinite_iter) *first_nonfinite_iter = d->diag_first_nonfinite;
    if (final_has_inf) *final_has_inf = d->diag_has_inf;
    if (final_has_nan) *final_has_nan = d->diag_has_nan;
    return 0;
}

/* ---------------------------------------------------------------------------------
 * BP, serial schedule  (f4: ldpc v2 `schedule="serial"`, never used by the reference; restated from memory)
 * Per iteration the bits are visited in ascending index.  For bit i: LLR := prior; for each of its checks, top to
 * bottom: the check->bit message is recomputed from the CURRENT bit->check messages of the check's other edges
 * (min-sum: alpha * (-1)^(syndrome + #{others <= 0}) * min |others|; product-sum: (-1)^syndrome *
 * log((1 + prod)/(1 - prod)), prod over the others of tanh(b2c / 2) in row order), the edge's bit->check message
 * becomes the running LLR (prefix), the LLR takes the new message; decision = (LLR <= 0); then, bottom to top, the
 * suffix sums are added to the bit->check messages.  Convergence is tested after every full sweep.
 * ------------------------------------------------------------------------------- */
static void bp_decode_serial(oracle_decoder *d, const uint8_t *syn, uint8_t *converged, int32_t *iters) {
    const int m = d->m, n = d->n;
    int conv = 0, it_done = 0;
    d->diag_first_nonfinite = d->diag_has_inf = d->diag_has_nan = 0;
    const int pm = d->cfg.ps_math;  /* 0 libm, 1 portable routines with two divisions per edge, 2 portable routines in the reference's operation order */
    for (int i = 0; i < n; i++)
        for (int k = d->cp[i]; k < d->cp[i + 1]; k++) d->b2c[d->ce[k]] = d->llr0[i];
    for (int it = 1; it <= d->max_iter; it++) {
        double alpha;
        if (d->cfg.ms_scaling_factor == 0.0) alpha = 1.0 - pow(2.0, -1.0 * it);
        else alpha = d->cfg.ms_scaling_factor;
        for (int i = 0; i < n; i++) {
            double llr = d->llr0[i];
            for (int k = d->cp[i]; k < d->cp[i + 1]; k++) {
                const int e = d->ce[k], c = d->erow[e];
                double msg;
                if (d->cfg.bp_method == 0) {
                    double prod = 1.0;
                    for (int g = d->rp[c]; g < d->rp[c + 1]; g++)
                        if (g != e) prod *= pm ? pm_ps_tanh_half(d->b2c[g], pm == 1) : tanh(d->b2c[g] / 2);
                    msg = (syn[c] ? -1 : 1) * (pm ? pm_ps_log_ratio(prod, pm == 1) : log((1 + prod) / (1 - prod)));
                    if (d->cfg.ps_clip > 0) {
                        if (msg > d->cfg.ps_clip) msg = d->cfg.ps_clip;
                        if (msg < -d->cfg.ps_clip) msg = -d->cfg.ps_clip;
                    }
                } else {
                    int sgn = syn[c];
                    double temp = DBL_MAX;
                    for (int g = d->rp[c]; g < d->rp[c + 1]; g++) {
                        if (g == e) continue;
                        const double a = fabs(d->b2c[g]);
                        if (a < temp) temp = a;
                        if (d->b2c[g] <= 0) sgn += 1;
                    }
                    const double message_sign = (sgn % 2 == 0) ? 1.0 : -1.0;
                    msg = alpha * message_sign * temp;
                }
                d->c2b[e] = msg;
                d->b2c[e] = llr;
                llr += msg;
            }
            d->llr[i] = llr;
            d->dec[i] = (llr <= 0) ? 1 : 0;
            double temp = 0;
            for (int k = d->cp[i + 1] - 1; k >= d->cp[i]; k--) {
                const int e = d->ce[k];
                d->b2c[e] += temp;
                temp += d->c2b[e];
            }
        }
        memset(d->cand, 0, (size_t)m);
        for (int i = 0; i < n; i++)
            if (d->dec[i])
                for (int k = d->cp[i]; k < d->cp[i + 1]; k++) d->cand[d->erow[d->ce[k]]] ^= 1;
        conv = memcmp(d->cand, syn, m) == 0;
        it_done = it;
        if (conv) break;
    }
    for (int i = 0; i < n && it_done > 0; i++) {
        if (isnan(d->llr[i])) d->diag_has_nan = 1;
        else if (isinf(d->llr[i])) d->diag_has_inf = 1;
    }
    *converged = (uint8_t)conv;
    *iters = it_done;
}

/* ---------------------------------------------------------------------------------
 * OSD  (a8-a11; Appendix A.4)
 * ------------------------------------------------------------------------------- */

/* a8: bit indices by ascending LLR.  Upstream: C qsort on (value,index) comparing
 * values only; glibc's merge sort keeps equal keys in index order => stable. */
static void sort_columns(oracle_decoder *d, const double *llr) {
    int n = d->n;
    int *a = d->order, *b = d->sort_tmp;
    for (int i = 0; i < n; i++) a[i] = d->cfg.sort_tie_policy == 1 ? n - 1 - i : i;
    /* bottom-up stable merge sort; comparison is on the double values alone */
    for (int width = 1; width < n; width *= 2) {
        for (int lo = 0; lo < n; lo += 2 * width) {
            int mid = lo + width < n ? lo + width : n;
            int hi = lo + 2 * width < n ? lo + 2 * width : n;
            int i = lo, j = mid, k = lo;
            while (i < mid && j < hi) {
                if (llr[a[j]] < llr[a[i]]) b[k++] = a[j++];
                else b[k++] = a[i++];
            }
            while (i < mid) b[k++] = a[i++];
            while (j < hi) b[k++] = a[j++];
        }
        int *t = a; a = b; b = t;
    }
    if (a != d->order) memcpy(d->order, a, sizeof(int) * n);
}

/* a9: eliminate H in the column order `order`; greedy pivot columns; the row
 * operations are recorded in L so that L * H[:,order] = U (row echelon). */
static int eliminate(oracle_decoder *d, const int *order) {
    const int m = d->m, n = d->n, W = d->W, WA = d->WA;
    memset(d->U, 0, sizeof(uint64_t) * (size_t)m * W);
    memset(d->L, 0, sizeof(uint64_t) * (size_t)m * WA);
    /* inverse permutation in sort_tmp */
    int *inv = d->sort_tmp;
    for (int j = 0; j < n; j++) inv[order[j]] = j;
    for (int r = 0; r < m; r++) {
        for (int e = d->rp[r]; e < d->rp[r + 1]; e++) {
            int j = inv[d->ci[e]];
            d->U[(size_t)r * W + (j >> 6)] |= 1ULL << (j & 63);
        }
        d->L[(size_t)r * WA + (r >> 6)] |= 1ULL << (r & 63);
    }
    int rank = 0;
    int max_rank = m < n ? m : n;
    memset(d->ispiv, 0, (size_t)n);
    for (int j = 0; j < n && rank < max_rank; j++) {
        int w = j >> 6;
        uint64_t bit = 1ULL << (j & 63);
        int p = -1;
        for (int r = rank; r < m; r++)
            if (d->U[(size_t)r * W + w] & bit) { p = r; break; }
        if (p < 0) continue;
        if (p != rank) {
            memcpy(d->tmpU, d->U + (size_t)p * W, sizeof(uint64_t) * W);
            memcpy(d->U + (size_t)p * W, d->U + (size_t)rank * W, sizeof(uint64_t) * W);
            memcpy(d->U + (size_t)rank * W, d->tmpU, sizeof(uint64_t) * W);
            memcpy(d->tmpL, d->L + (size_t)p * WA, sizeof(uint64_t) * WA);
            memcpy(d->L + (size_t)p * WA, d->L + (size_t)rank * WA, sizeof(uint64_t) * WA);
            memcpy(d->L + (size_t)rank * WA, d->tmpL, sizeof(uint64_t) * WA);
        }
        for (int r = rank + 1; r < m; r++) {
            if (d->U[(size_t)r * W + w] & bit) {
                for (int x = w; x < W; x++) d->U[(size_t)r * W + x] ^= d->U[(size_t)rank * W + x];
                for (int x = 0; x < WA; x++) d->L[(size_t)r * WA + x] ^= d->L[(size_t)rank * WA + x];
            }
        }
        d->pivcol[rank] = j;
        d->ispiv[j] = 1;
        rank++;
    }
    return rank;
}

static int gf2_rank_of_H(oracle_decoder *d) {
    for (int i = 0; i < d->n; i++) d->order[i] = i;
    return eliminate(d, d->order);
}

/* unique solution on the pivot columns of  H[:,order] x = t  (x = 0 elsewhere);
 * result in sorted-position space as bytes xs[n] */
static void lu_solve(oracle_decoder *d, int rank, const uint8_t *t, uint8_t *xs) {
    const int m = d->m, n = d->n, W = d->W, WA = d->WA;
    memset(d->tvec, 0, sizeof(uint64_t) * WA);
    for (int r = 0; r < m; r++)
        if (t[r]) d->tvec[r >> 6] |= 1ULL << (r & 63);
    /* forward: t' = L t */
    memset(d->tp, 0, sizeof(uint64_t) * WA);
    for (int r = 0; r < m; r++) {
        uint64_t acc = 0;
        for (int x = 0; x < WA; x++) acc ^= d->L[(size_t)r * WA + x] & d->tvec[x];
        if (parity64(acc)) d->tp[r >> 6] |= 1ULL << (r & 63);
    }
    /* backward substitution over the pivots */
    memset(d->xp, 0, sizeof(uint64_t) * W);
    memset(xs, 0, n);
    for (int k = rank - 1; k >= 0; k--) {
        int j = d->pivcol[k];
        uint64_t acc = 0;
        for (int x = j >> 6; x < W; x++) acc ^= d->U[(size_t)k * W + x] & d->xp[x];
        int v = parity64(acc) ^ (int)((d->tp[k >> 6] >> (k & 63)) & 1);
        if (v) {
            d->xp[j >> 6] |= 1ULL << (j & 63);
            xs[j] = 1;
        }
    }
}

/* a11: weight of a solution given in ORIGINAL bit order */
static double solution_weight(const oracle_decoder *d, const uint8_t *x) {
    double wgt = 0;
    if (d->cfg.weight_fn == 1) {
        for (int i = 0; i < d->n; i++) wgt += x[i];
    } else {
        for (int i = 0; i < d->n; i++)
            if (x[i] == 1) wgt += log(1 / d->p[i]);
    }
    return wgt;
}

int oracle_osd(oracle_decoder *d, const uint8_t *syn, const double *llr, uint8_t *osdw,
               uint8_t *osd0, int32_t *order_out, int32_t *pivot_flag_out) {
    const int m = d->m, n = d->n;
    sort_columns(d, llr);
    int *order = (int *)malloc(sizeof(int) * n);
    memcpy(order, d->order, sizeof(int) * n); /* eliminate() clobbers sort_tmp only, keep a copy anyway */
    int rank = eliminate(d, order);
    if (order_out) memcpy(order_out, order, sizeof(int) * n);
    if (pivot_flag_out)
        for (int j = 0; j < n; j++) pivot_flag_out[j] = d->ispiv[j];

    uint8_t *xs = (uint8_t *)malloc(n);
    /* a9: OSD-0 */
    lu_solve(d, rank, syn, xs);
    for (int j = 0; j < n; j++) d->best[order[j]] = xs[j];
    if (osd0) memcpy(osd0, d->best, n);

    int w = d->cfg.osd_order;
    if (d->cfg.osd_method >= 2 && w > 0) {
        /* non-pivot columns keep the sorted order (Appendix A.4) */
        int kp = 0;
        for (int j = 0; j < n; j++)
            if (!d->ispiv[j]) d->nonpiv[kp++] = j;
        double min_weight = solution_weight(d, d->best);
        int ncand = d->cfg.osd_method == 2 ? (1 << w) - 1 : kp + w * (w - 1) / 2;
        int pa = 0, pb = 0; /* pair enumeration state for osd_cs */
        for (int c = 0; c < ncand; c++) {
            int tsel[32];
            int nt = 0;
            if (d->cfg.osd_method == 2) {
                /* a10 osd_e: pattern i = c+1, least-significant bit -> T position 0 */
                int pat = c + 1;
                for (int b = 0; b < w; b++)
                    if ((pat >> b) & 1) tsel[nt++] = d->cfg.osd_e_bit_order ? w - 1 - b : b;
            } else if (c < kp) {
                /* a10 osd_cs: all k' weight-1 patterns first */
                tsel[nt++] = c;
            } else {
                /* then pairs (i<j<w), i outer, j inner */
                if (c == kp) { pa = 0; pb = 1; }
                tsel[nt++] = pa;
                tsel[nt++] = pb;
                pb++;
                if (pb >= w) { pa++; pb = pa + 1; }
            }
            memcpy(d->tsyn, syn, m);
            for (int k = 0; k < nt; k++) {
                int col = order[d->nonpiv[tsel[k]]];
                for (int q = d->cp[col]; q < d->cp[col + 1]; q++) d->tsyn[d->erow[d->ce[q]]] ^= 1;
            }
            lu_solve(d, rank, d->tsyn, xs);
            for (int k = 0; k < nt; k++) xs[d->nonpiv[tsel[k]]] = 1;
            for (int j = 0; j < n; j++) d->sol[order[j]] = xs[j];
            double cw = solution_weight(d, d->sol);
            if (cw < min_weight) { /* strictly lighter: first found wins ties */
                min_weight = cw;
                memcpy(d->best, d->sol, n);
            }
        }
    }
    if (osdw) memcpy(osdw, d->best, n);
    free(xs);
    free(order);
    return 0;
}

/* ---------------------------------------------------------------------------------
 * a2: decode() dispatcher  (README.md:197; css_decode_sim.py:174-202; Appendix A.2)
 * ------------------------------------------------------------------------------- */
int oracle_decode(oracle_decoder *d, const uint8_t *syn, uint8_t *osdw, uint8_t *osd0,
                  uint8_t *bp, uint8_t *converged, int32_t *iters, double *llr) {
    const int m = d->m, n = d->n;
    int zero = 1;
    for (int c = 0; c < m; c++)
        if (syn[c]) { zero = 0; break; }
    if (zero) {
        /* all-zero syndrome: zeros, converge = True, BP not run */
        if (osdw) memset(osdw, 0, n);
        if (osd0) memset(osd0, 0, n);
        if (bp) memset(bp, 0, n);
        if (converged) *converged = 1;
        if (iters) *iters = 0;
        if (llr) memcpy(llr, d->llr0, sizeof(double) * n);
        return 0;
    }
    uint8_t conv;
    int32_t its;
    if (d->cfg.schedule == 1) bp_decode_serial(d, syn, &conv, &its);
    else bp_decode(d, syn, &conv, &its);
    if (bp) memcpy(bp, d->dec, n);
    if (converged) *converged = conv;
    if (iters) *iters = its;
    if (llr) memcpy(llr, d->llr, sizeof(double) * n);
    if (conv || d->cfg.osd_method == 0) {
        /* converged: osd0 = osdw = bp decoding, OSD skipped */
        if (osdw) memcpy(osdw, d->dec, n);
        if (osd0) memcpy(osd0, d->dec, n);
        return 0;
    }
    return oracle_osd(d, syn, d->llr, osdw, osd0, NULL, NULL);
}

/* vectorised access to bp_osd_amd/csrc/portable_math.h for tests/test_portable_math.py: which = 0 tanh, 1 log, 2 expm1 */
void oracle_portable_math(int32_t which, const double *x, double *y, int64_t count) {
    for (int64_t i = 0; i < count; i++) y[i] = which == 0 ? pm_tanh(x[i]) : (which == 1 ? pm_log(x[i]) : pm_expm1(x[i]));
}

/* round 4: tanh(x / 2) with one division, log(A / B) with one division */
void oracle_portable_tanh_half(const double *x, double *y, int64_t count) {
    for (int64_t i = 0; i < count; i++) y[i] = pm_tanh_half(x[i]);
}
void oracle_portable_log_quot(const double *a, const double *b, double *y, int64_t count) {
    for (int64_t i = 0; i < count; i++) y[i] = pm_log_quot(a[i], b[i]);
}

int oracle_decode_batch_diag(oracle_decoder *d, const uint8_t *syndromes, int64_t B, uint8_t *osdw,
                             uint8_t *osd0, uint8_t *bp, uint8_t *converged, int32_t *iters, double *llr,
                             int32_t *first_nonfinite_iter, uint8_t *final_has_inf, uint8_t *final_has_nan) {
    for (int64_t b = 0; b < B; b++) {
        d->diag_first_nonfinite = d->diag_has_inf = d->diag_has_nan = 0; /* zero syndrome: BP is not run */
        int rc = oracle_decode(d, syndromes + b * d->m, osdw ? osdw + b * d->n : NULL,
                               osd0 ? osd0 + b * d->n : NULL, bp ? bp + b * d->n : NULL,
                               converged ? converged + b : NULL, iters ? iters + b : NULL,
                               llr ? llr + b * d->n : NULL);
        if (rc) return rc;
        if (first_nonfinite_iter) first_nonfinite_iter[b] = d->diag_first_nonfinite;
        if (final_has_inf) final_has_inf[b] = (uint8_t)d->diag_has_inf;
        if (final_has_nan) final_has_nan[b] = (uint8_t)d->diag_has_nan;
    }
    return 0;
}

int oracle_decode_batch(oracle_decoder *d, const uint8_t *syndromes, int64_t B, uint8_t *osdw,
                        uint8_t *osd0, uint8_t *bp, uint8_t *converged, int32_t *iters,
                        double *llr) {
    for (int64_t b = 0; b < B; b++) {
        int rc = oracle_decode(d, syndromes + b * d->m, osdw ? osdw + b * d->n : NULL,
                               osd0 ? osd0 + b * d->n : NULL, bp ? bp + b * d->n : NULL,
                               converged ? converged + b : NULL, iters ? iters + b : NULL,
                               llr ? llr + b * d->n : NULL);
        if (rc) return rc;
    }
    return 0;
}
