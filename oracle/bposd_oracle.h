/*
 * oracle/bposd_oracle.h -- CPU restatement of the reference's BP+OSD decode path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under bp_osd_amd/ may include, link or call
 * this; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
 *
 * PARITY STATUS: "parity unpinned" in the strict sense of the build contract.
 * The reference repository (/root/reference) contains no decoder arithmetic: the
 * path `bposd_decoder(...).decode(syndrome)` is implemented by the un-vendored,
 * un-pinned third-party dependency `ldpc>=2.0.0` (/root/reference/setup.py:30,
 * imported at /root/reference/src/bposd/__init__.py:1 and
 * /root/reference/src/bposd/css_decode_sim.py:6).  That package is absent from this
 * pipeline, and the reference's own tests never call decode().  This file restates
 * the published algorithm of ldpc v2 (BpDecoder parallel schedule, OsdDecoder,
 * soft_decision_col_sort, RowReduce) as described in SURVEY.md Appendix A, and is
 * pinned by the only reference-authored datum for the path -- the worked example
 * at /root/reference/README.md:178-216 -- plus mathematical invariants
 * (tests/test_oracle.py).
 */
#ifndef BPOSD_ORACLE_H
#define BPOSD_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct oracle_decoder oracle_decoder;

/* bp_method: 0 product-sum, 1 min-sum.  osd_method: 0 off, 1 osd0, 2 osd_e, 3 osd_cs.
 * sort_tie_policy: 0 stable (ties keep ascending bit index), 1 ties descending index.
 * weight_fn: 0 sum log(1/p_i) in index order (ldpc v2), 1 Hamming weight (ldpc v1). */
typedef struct {
    int32_t bp_method;
    double ms_scaling_factor; /* 0 => 1 - 2^-it */
    int32_t max_iter;         /* 0 => n */
    int32_t osd_method;
    int32_t osd_order;
    int32_t sort_tie_policy;
    int32_t weight_fn;
    double ps_clip;           /* product-sum only: 0 = none (upstream: no clipping, messages reach +-inf and
                                 NaN); C > 0 = every check->bit message is clamped to [-C, C] (build-owned switch) */
    int32_t schedule;         /* 0 = parallel (flooding) schedule -- the only one the reference uses
                                 (css_decode_sim.py:444-463 never passes `schedule`); 1 = serial schedule of ldpc v2
                                 (bits in ascending index, each bit's check->bit messages recomputed from the latest
                                 bit->check messages; SURVEY.md §8 f4, restated from memory like Appendix A) */
    int32_t ps_math;          /* product-sum only: 0 = tanh / log of the platform libm (what the reference calls);
                                 1 / 2 = the bit-reproducible routines of bp_osd_amd/csrc/portable_math.h, which is what the
                                 GPU kernels evaluate: 2 in the reference's operation order (the kernels' default,
                                 ps_math_form 0), 1 with two divisions per edge (ps_math_form 1) -- with the matching mode
                                 the oracle and the GPU agree bit for bit, with 0 they differ by the libm's last-bit
                                 behaviour (tests state both bars) */
    int32_t osd_e_bit_order;  /* osd_e: bit b of pattern i stands for T position b (0, LSB first) or w - 1 - b (1); only
                                 ties between equally light patterns depend on it.  UNVERIFIED upstream behaviour, a
                                 switch like sort_tie_policy */
} oracle_config;

/* pcm as CSR with sorted column indices; channel_probs[n]. Returns 0 or <0. */
int oracle_create(const oracle_config *cfg, const int32_t *indptr, const int32_t *indices,
                  int32_t m, int32_t n, const double *channel_probs, oracle_decoder **out);
void oracle_destroy(oracle_decoder *d);
int oracle_update_channel_probs(oracle_decoder *d, const double *channel_probs);
int oracle_rank(const oracle_decoder *d);
int oracle_num_candidates(const oracle_decoder *d);

/* One syndrome (uint8[m], 0/1).  Outputs uint8[n] each (nullable), double llr[n] (nullable). */
int oracle_decode(oracle_decoder *d, const uint8_t *syndrome, uint8_t *osdw, uint8_t *osd0,
                  uint8_t *bp, uint8_t *converged, int32_t *iters, double *llr);

/* B syndromes row-major; serial loop over oracle_decode. */
int oracle_decode_batch(oracle_decoder *d, const uint8_t *syndromes, int64_t B, uint8_t *osdw,
                        uint8_t *osd0, uint8_t *bp, uint8_t *converged, int32_t *iters,
                        double *llr);

/* y[i] = f(x[i]) with f from bp_osd_amd/csrc/portable_math.h (which: 0 tanh, 1 log, 2 expm1); for the accuracy test. */
void oracle_portable_math(int32_t which, const double *x, double *y, int64_t count);
void oracle_portable_tanh_half(const double *x, double *y, int64_t count);
void oracle_portable_log_quot(const double *a, const double *b, double *y, int64_t count);

/* oracle_decode_batch plus per-shot BP diagnostics (see oracle_last_bp_diag); the three arrays are nullable. */
int oracle_decode_batch_diag(oracle_decoder *d, const uint8_t *syndromes, int64_t B, uint8_t *osdw,
                             uint8_t *osd0, uint8_t *bp, uint8_t *converged, int32_t *iters, double *llr,
                             int32_t *first_nonfinite_iter, uint8_t *final_has_inf, uint8_t *final_has_nan);

/* Diagnostics of the LAST oracle_decode call's BP run: first iteration in which a check->bit message or a
 * posterior LLR was not finite (0 = never), and whether the final LLRs contain +-inf / NaN. */
int oracle_last_bp_diag(const oracle_decoder *d, int32_t *first_nonfinite_iter, int32_t *final_has_inf,
                        int32_t *final_has_nan);

/* OSD alone on caller-supplied LLRs (used to pin OSD semantics independently of BP). */
int oracle_osd(oracle_decoder *d, const uint8_t *syndrome, const double *llr, uint8_t *osdw,
               uint8_t *osd0, int32_t *order_out /* n, nullable */,
               int32_t *pivot_flag_out /* n by sorted position, nullable */);

#ifdef __cplusplus
}
#endif
#endif
