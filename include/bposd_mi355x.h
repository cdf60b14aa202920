/*
 * bposd_mi355x.h -- C-ABI of libbposd_mi355x.so, the MI355X (gfx950) BP+OSD decoder.
 *
 * This is the drop-in boundary for the reference's decode path.  The reference has
 * no C/FFI plugin interface of its own: its "operator API" is the Python class
 * `bposd_decoder` / `BpOsdDecoder` that it imports from the third-party `ldpc`
 * package (/root/reference/src/bposd/__init__.py:1,
 * /root/reference/src/bposd/css_decode_sim.py:6).  Each entry point below names the
 * reference interface it replaces; bp_osd_amd/decoder.py binds them with ctypes and
 * re-creates that Python class on top (INTEGRATION.md shows the one-line switch).
 *
 * Plain C: opaque handle, plain pointers and sizes, no C++ or torch types.
 * Return value 0 = OK, negative = error (message via bposd_last_error).  The library
 * never aborts the process.  One handle <-> one device <-> one caller thread at a
 * time; different handles may be driven from different threads.
 */
#ifndef BPOSD_MI355X_H
#define BPOSD_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The library is built with -fvisibility=hidden: only what this header declares is exported. */
#pragma GCC visibility push(default)

typedef struct bposd_handle bposd_handle;

/* Values of bposd_config fields. */
enum { BPOSD_BP_PRODUCT_SUM = 0, BPOSD_BP_MIN_SUM = 1 };
enum { BPOSD_OSD_OFF = 0, BPOSD_OSD_0 = 1, BPOSD_OSD_E = 2, BPOSD_OSD_CS = 3 };

/* Error codes. */
enum {
    BPOSD_OK = 0,
    BPOSD_ERR_INVALID = -1,     /* bad argument / shape / option          -> ValueError   */
    BPOSD_ERR_UNSUPPORTED = -2, /* valid request this build cannot run    -> ValueError   */
    BPOSD_ERR_HIP = -3,         /* HIP runtime failure                    -> RuntimeError */
    BPOSD_ERR_NO_DEVICE = -4    /* no gfx950 device visible               -> RuntimeError */
};

/*
 * Constructor options.  Replaces the kwargs of
 *   bposd_decoder(pcm, error_rate, max_iter, bp_method, ms_scaling_factor,
 *                 channel_probs, osd_method, osd_order)      /root/reference/README.md:178-187
 *   BpOsdDecoder(pcm, channel_probs=..., max_iter=..., bp_method=...,
 *                ms_scaling_factor=..., osd_method=..., osd_order=...)
 *                                         /root/reference/src/bposd/css_decode_sim.py:444-463
 */
typedef struct {
    int32_t device;            /* HIP device ordinal (>= 0)                                  */
    int32_t bp_method;         /* BPOSD_BP_*                                                 */
    double ms_scaling_factor;  /* 0 => variable scaling 1 - 2^-it (README.md:184)            */
    int32_t max_iter;          /* 0 => block length n                                        */
    int32_t osd_method;        /* BPOSD_OSD_*                                                */
    int32_t osd_order;         /* osd_e: patterns on the first w non-pivots; osd_cs: pair span */
    int32_t sort_tie_policy;   /* 0 = stable ascending index among equal LLRs, 1 = descending */
    int32_t weight_fn;         /* 0 = sum log(1/p_i) (ldpc v2), 1 = Hamming weight (ldpc v1)  */
    int32_t schedule;          /* 0 = parallel (flooding) BP schedule -- what the reference runs; 1 = ldpc's
                                  "serial" schedule (bits in ascending index).  Was a reserved
                                  word: a zero-filled old config means parallel                */
    double ps_clip;            /* product-sum only.  0 = upstream behaviour: no clipping, so check->bit messages
                                  reach +-inf once tanh rounds to 1 and NaN follows;
                                  C > 0 = every check->bit message is clamped to [-C, C] (build-owned switch,
                                  DESIGN.md "Product-sum").  Occupies two of the four formerly reserved words: a
                                  zero-filled old config means "no clipping"                  */
    int32_t osd_e_bit_order;   /* osd_e: which T position bit b of pattern i = 1 .. 2^w - 1 stands for.  0 = position b
                                  (LSB first; the restatement's reading of upstream), 1 = position
                                  w - 1 - b.  Only the tie between equally light patterns depends on it (the first one
                                  enumerated wins).  An UNVERIFIED-upstream-behaviour switch like sort_tie_policy /
                                  weight_fn; was reserved[0]: a zero-filled old config means LSB first */
    int32_t ps_math_form;      /* product-sum: evaluation order of the check update (bp_osd_amd/csrc/portable_math.h).  0 =
                                  the reference's operation order -- tanh(b2c / 2), then log of the rounded quotient
                                  (1 + x) / (1 - x): four divisions per edge; closest to the platform libm the reference calls
                                  (188 of 2048 clipped BASELINE configs[2] shots differ from it in an integer output).  1 = two
                                  divisions per edge (pm_tanh_half, pm_log_quot): 1.4 x the throughput, 207 of 2048.  Was
                                  reserved[0]: a zero-filled old config means the reference order */
} bposd_config;

/* Number of visible HIP devices (0 if none / runtime unavailable). */
int bposd_device_count(void);

/* Library version string, e.g. "bposd_mi355x 0.1 (gfx950)". */
const char *bposd_version(void);

/*
 * Create a decoder.  pcm is CSR (indptr[m+1], indices[E], column indices strictly
 * ascending within a row); channel_probs[n] are the per-bit error probabilities
 * (`channel_probs`, or `error_rate` broadcast: README.md:180-181).  The library
 * copies everything; the caller keeps ownership of its arrays.
 * Replaces: the ctor call sites README.md:178-187, css_decode_sim.py:444-463.
 */
int bposd_create(const bposd_config *cfg, const int32_t *csr_indptr, const int32_t *csr_indices,
                 int32_t m, int32_t n, const double *channel_probs, bposd_handle **out);

/* Replaces `.update_channel_probs(p)` -- css_decode_sim.py:229,248. */
int bposd_update_channel_probs(bposd_handle *h, const double *channel_probs);

/*
 * Decode B syndromes held in HOST memory (row-major uint8[B*m], values 0/1) and write
 * the results to HOST buffers.  Synchronous.  osdw is required; osd0, bp, converged,
 * iters, llr may be NULL.
 *   osdw[B*n]  -> `.osdw_decoding`  (README.md:202; css_decode_sim.py:257-258)
 *   osd0[B*n]  -> `.osd0_decoding`  (css_decode_sim.py:294-295)
 *   bp[B*n]    -> `.bp_decoding`    (css_decode_sim.py:338-339)
 *   converged[B] -> `.converge`     (css_decode_sim.py:331-336)
 *   iters[B]   -> `.iter`;  llr[B*n] -> `.log_prob_ratios` (final BP LLRs, fp64)
 * Replaces: `.decode(syndrome)` -- README.md:197; css_decode_sim.py:174-202 (B = 1),
 * and is the batched form the MI355X path is built around.
 */
int bposd_decode_batch(bposd_handle *h, const uint8_t *syndromes, int64_t B, uint8_t *osdw,
                       uint8_t *osd0, uint8_t *bp, uint8_t *converged, int32_t *iters, double *llr);

/*
 * The same call with bit-packed rows (host memory): syndromes as B rows of ceil(m/64) little-endian 64-bit words, osdw /
 * osd0 / bp as B rows of ceil(n/64) words -- bit (i & 63) of word (i >> 6) of a row is entry i, padding bits are zero
 * (numpy: np.packbits(rows, axis=1, bitorder="little") padded to a multiple of 8 bytes).  One eighth of the bytes cross
 * PCIe; the device unpacks the syndromes in front of the BP kernel and packs the result rows behind it.  This is the
 * form SURVEY.md 8(d)(i) / 8(e) recommend for the host-to-host metric.  osd0_words, bp_words, converged, iters may be
 * NULL.  Synchronous.  Replaces the same call sites as bposd_decode_batch.
 */
int bposd_decode_batch_packed(bposd_handle *h, const uint64_t *syndrome_words, int64_t B, uint64_t *osdw_words,
                              uint64_t *osd0_words, uint64_t *bp_words, uint8_t *converged, int32_t *iters);

/*
 * Asynchronous forms of bposd_decode_batch / bposd_decode_batch_packed for a STREAM of batches: the whole call -- upload,
 * kernels, download -- is enqueued on the handle's next lane (see "Lanes" below) and the function returns; consecutive
 * calls overlap on the device, which a synchronous call cannot (it pays its own upload, its longest-running syndrome and
 * its download).  The buffers should come from bposd_host_alloc (with pageable memory the copies block) and must stay
 * untouched until bposd_synchronize_lane(h, lane) with lane = bposd_last_lane(h) read right after the call, or
 * bposd_synchronize(h).  At most bposd_num_lanes(h) calls are in flight; a further call queues behind the oldest.
 * (No counterpart in the reference: its decode is synchronous.)
 */
int bposd_decode_batch_async(bposd_handle *h, const uint8_t *syndromes, int64_t B, uint8_t *osdw, uint8_t *osd0,
                             uint8_t *bp, uint8_t *converged, int32_t *iters, double *llr);
int bposd_decode_batch_packed_async(bposd_handle *h, const uint64_t *syndrome_words, int64_t B, uint64_t *osdw_words,
                                    uint64_t *osd0_words, uint64_t *bp_words, uint8_t *converged, int32_t *iters);

/*
 * Same, but every pointer is a DEVICE pointer on the handle's device (inputs already
 * resident in HBM).  Asynchronous on the handle's stream: call bposd_synchronize()
 * before reading the outputs.
 */
int bposd_decode_batch_device(bposd_handle *h, const uint8_t *d_syndromes, int64_t B,
                              uint8_t *d_osdw, uint8_t *d_osd0, uint8_t *d_bp,
                              uint8_t *d_converged, int32_t *d_iters, double *d_llr);

/* The device-pointer call with bit-packed rows (the layout of bposd_decode_batch_packed; every pointer a device pointer):
 * the kernels read packed syndromes and write packed result rows themselves -- no pack kernel between the decode and the
 * multi-GPU gather (the HBM-resident kernels included, since round 5).  Codes on the any-degree BP kernel or with the serial
 * schedule return BPOSD_ERR_UNSUPPORTED: there bposd_decode_batch_device + bposd_pack_rows_device do the same.
 * Asynchronous like bposd_decode_batch_device. */
int bposd_decode_batch_device_packed(bposd_handle *h, const uint64_t *d_syndrome_words, int64_t B, uint64_t *d_osdw_words,
                                     uint64_t *d_osd0_words, uint64_t *d_bp_words, uint8_t *d_converged, int32_t *d_iters);

/*
 * Per-syndrome channel, two values per bit: bit i of syndrome b is decoded with probability
 * channel_probs_alt[i] where select[b*n + i] != 0 and with the handle's channel_probs[i] elsewhere
 * (BP priors and OSD-W weights alike).  This is exactly what the reference's per-shot Bayesian channel
 * update produces -- /root/reference/src/bposd/css_decode_sim.py:207-248 sets, per shot,
 * p_i = py/(px+py) where the first decoder flipped bit i and pz/(1-px-py) elsewhere, then calls
 * `.update_channel_probs(p)` before the second `.decode` -- batched over B shots.
 * Host-pointer form, synchronous; `select` and `channel_probs_alt` are required here.
 */
int bposd_decode_batch_select(bposd_handle *h, const uint8_t *syndromes, int64_t B,
                              const uint8_t *select, const double *channel_probs_alt, uint8_t *osdw,
                              uint8_t *osd0, uint8_t *bp, uint8_t *converged, int32_t *iters,
                              double *llr);

/* Device-pointer form of the above (d_select on the device; channel_probs_alt stays a HOST array of n
 * doubles, it is uploaded by the call).  Asynchronous like bposd_decode_batch_device. */
int bposd_decode_batch_select_device(bposd_handle *h, const uint8_t *d_syndromes, int64_t B,
                                     const uint8_t *d_select, const double *channel_probs_alt,
                                     uint8_t *d_osdw, uint8_t *d_osd0, uint8_t *d_bp,
                                     uint8_t *d_converged, int32_t *d_iters, double *d_llr);

/*
 * Bit-pack B rows of n 0/1 bytes (device) into B rows of ceil(n/64) little-endian 64-bit words (device):
 * bit (i & 63) of word (i >> 6) of row b = d_bytes[b*n + i] & 1.  Used to shrink the one exchange step of
 * the multi-GPU path (the gather of corrections) 8x.  Asynchronous on the handle's stream.
 */
int bposd_pack_rows_device(bposd_handle *h, const uint8_t *d_bytes, int64_t B, int32_t n,
                           uint64_t *d_words);
/* The same on a given lane.  bposd_pack_rows_device queues the kernel on the lane of the MOST RECENT device-pointer
 * decode: it must be called before the next decode call is enqueued, or the rows of an earlier call would be packed on a
 * stream that is not ordered behind the kernel still writing them.  A caller that packs later passes the lane its decode
 * ran on (bposd_last_lane right after that call). */
int bposd_pack_rows_device_lane(bposd_handle *h, int32_t lane, const uint8_t *d_bytes, int64_t B, int32_t n,
                                uint64_t *d_words);

/* Wait for all work queued on the handle (every lane, see below). */
int bposd_synchronize(bposd_handle *h);

/*
 * Lanes.  A handle owns bposd_num_lanes(h) HIP streams with their own workspaces and uses them in turn: consecutive
 * device-pointer calls (and the chunks of one host-pointer call) overlap on the device, so the next call's workgroups
 * take over the CUs that the previous call's last max_iter stragglers and its OSD kernel leave idle.  Consequences
 * for a caller of the asynchronous device-pointer API: two consecutive calls are NOT ordered against each other (give
 * them different output buffers, or synchronise in between); a call is ordered behind the call bposd_num_lanes(h)
 * calls earlier (same lane).  bposd_last_lane() names the lane of the last device-pointer call; bposd_synchronize_lane() waits
 * for that lane only; bposd_lane_timing() is bposd_last_timing() for the last call queued on one lane.
 * (No counterpart in the reference: its decoder is a synchronous single-thread object.)
 */
int bposd_num_lanes(bposd_handle *h); /* lanes this handle cycles through (4; 2 for HBM-resident codes); NULL: the maximum */
int bposd_last_lane(bposd_handle *h);
int bposd_synchronize_lane(bposd_handle *h, int32_t lane);
int bposd_lane_timing(bposd_handle *h, int32_t lane, double *bp_ms, double *osd_ms, int64_t *bp_iterations,
                      int64_t *osd_invocations);

/*
 * Page-locked host memory for the host-pointer API: with buffers from here bposd_decode_batch's chunked uploads and
 * downloads are truly asynchronous (pageable memory works too; the host thread then blocks inside each copy).
 * Returns NULL on failure.
 */
void *bposd_host_alloc(size_t bytes);
void bposd_host_free(void *p);

/*
 * Timing and work counters of the LAST decode call, measured with HIP events on the
 * stream the kernels were launched on (waits for that call to finish):
 *   bp_ms, osd_ms    kernel durations (a host-pointer call runs in chunks: the sum over its chunks, which
 *                    overlap on the device; a host-pointer call of up to 1 MB of staging -- the one-syndrome
 *                    decode -- runs without events and reports 0.0 for both)
 *   bp_iterations    sum over syndromes of BP iterations executed
 *   osd_invocations  syndromes that went through OSD (BP did not converge)
 * Any pointer may be NULL.
 */
int bposd_last_timing(bposd_handle *h, double *bp_ms, double *osd_ms, int64_t *bp_iterations,
                      int64_t *osd_invocations);

/* Decoder facts: rank of the pcm over GF(2), number of OSD-W candidates per decode,
 * effective max_iter, nonzeros E.  Any pointer may be NULL. */
int bposd_info(bposd_handle *h, int32_t *rank, int32_t *num_candidates, int32_t *max_iter,
               int32_t *nnz);

/* BP only: posterior log-likelihood ratios (and, optionally, BP's hard decisions, converge flags and iteration counts) of
 * B host syndromes, without the OSD stage whatever osd_method the handle was created with.  This is what the
 * `log_prob_ratios` attribute of the reference's decoder object holds after a decode (SURVEY.md 8 b); the Python class
 * calls it when the attribute is read after a `decode()` that did not ask for LLRs.  llr [B, n] required; bp [B, n], conv
 * [B], iters [B] nullable.  Synchronous. */
int bposd_posterior_llr(bposd_handle *h, const uint8_t *syndromes, int64_t B, double *llr, uint8_t *bp,
                        uint8_t *converged, int32_t *iters);

/* Tuning / test knob for the small-code OSD stage: 0 = auto, 1 = one workgroup per elimination (osd_kernel.hip.h),
 * 2 = one wave per elimination (osd_wave_kernel.hip.h) where it applies (uniform channel, m <= 448, n <= 959, osd_e order
 * <= 12; auto picks it there for calls of at least 4096 syndromes -- a lone elimination is faster on a workgroup of its own).
 * Identical results.  bposd_last_osd_kernel: 1 / 2 as above, 3 = the HBM-resident kernel,
 * -1 before the first OSD launch. */
int bposd_set_osd_variant(bposd_handle *h, int32_t variant);
int bposd_last_osd_kernel(bposd_handle *h);

/* Tuning knob (not part of the reference surface): which BP kernel / workgroup shape runs.
 * 0 = auto; 1, 2, 4 = LDS kernel with 1 / 2 / 4 checks per thread; 16, 17, 18 = local-edge kernel (a third of the
 * messages in registers; (3,6)-regular codes with n = 2m and min-sum only, BPOSD_ERR_UNSUPPORTED otherwise):
 * 2 checks per thread at <= 80 / <= 64 VGPRs, 1 check per thread.  Auto picks 16 where it applies.  All variants
 * return identical results.  32 = class kernel (one check degree, bit degrees of a compiled range; auto picks it where it
 * applies and the local-edge kernel does not).  63 = HBM-resident min-sum kernel with whole check records in the workspace
 * (the form a code whose per-check data exceed the CU's LDS gets; ignored by the other kernels).  64 = the any-degree
 * kernel on any code (slow; a second implementation for cross-checks, also of the HBM-resident BP kernel). */
int bposd_set_bp_variant(bposd_handle *h, int32_t variant);

/* Message for the last error on this handle (h == NULL: last create() failure). */
const char *bposd_last_error(bposd_handle *h);

void bposd_destroy(bposd_handle *h);

#pragma GCC visibility pop

#ifdef __cplusplus
}
#endif
#endif /* BPOSD_MI355X_H */
