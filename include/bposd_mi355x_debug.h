/*
 * bposd_mi355x_debug.h -- diagnostics of libbposd_mi355x.so that have no counterpart in the reference's interface:
 * which kernel ran, the bank-conflict model of its LDS layout, and the layout tables themselves (host only) for the
 * CPU tests.  Not needed to use the decoder; include/bposd_mi355x.h is the drop-in boundary.
 */
#ifndef BPOSD_MI355X_DEBUG_H
#define BPOSD_MI355X_DEBUG_H

#include "bposd_mi355x.h"

#ifdef __cplusplus
extern "C" {
#endif
#pragma GCC visibility push(default)

/* Diagnostics: simulated LDS cycles of one bit pass (bank-conflict model) for the natural bit order, the
 * order the library chose, and the conflict-free ideal.  Any pointer may be NULL. */
int bposd_layout_info(bposd_handle *h, int64_t *natural_cycles, int64_t *chosen_cycles, int64_t *ideal_cycles);

/* Diagnostics: which BP kernel the last decode call launched, and the bank-conflict model of its bit pass.
 * kernel: BPOSD_BP_KERNEL_*.  lds_model[4] (local-edge and class kernels, else zeros): modelled ds_read_b64 cycles of one
 * bit pass per workgroup, their conflict-free floor, modelled ds_write_b64 cycles, their floor; after a bp_large_kernel launch
 * lds_model[0] is the form that ran (0: per-edge messages both ways (product-sum); 1: min-sum, one 32-byte record per check in the
 * workspace; 2: min-sum, per-check data in LDS).  Any pointer may be NULL. */
#define BPOSD_BP_KERNEL_LDS 0    /* bp_kernel: every message in LDS, per-lane degree predicates */
#define BPOSD_BP_KERNEL_LOCAL 1  /* bp_local_kernel: (3,6)-regular codes, a third of the messages in registers */
#define BPOSD_BP_KERNEL_CLASS 2  /* bp_class_kernel: one check degree, bits sorted into degree classes */
#define BPOSD_BP_KERNEL_LARGE 3  /* bp_large_kernel: messages in HBM */
#define BPOSD_BP_KERNEL_SERIAL 4 /* bp_serial_kernel: schedule = serial */
#define BPOSD_BP_KERNEL_ANYDEG 5 /* bp_anydeg_kernel: check degree > 16 or bit degree > 8 (run-time degree loops) */
int bposd_bp_kernel_info(bposd_handle *h, int32_t *kernel, int64_t *lds_model);

/* Diagnostics, host only (needs no device): the ownership / position layout the local-edge BP kernel would use for a
 * (3,6)-regular pcm with n = 2m.  out[16]: modelled ds_read_b64 cycles of one bit pass, their conflict-free
 * floor, positions in select-free (uniform) groups, mixed (group, slot) pairs, positions, nine class sizes, modelled
 * ds_write_b64 cycles of one bit pass, their floor. */
int bposd_debug_local_layout(const int32_t *csr_indptr, const int32_t *csr_indices, int32_t m, int32_t n, int64_t *out);

/* Diagnostics, host only: the tables bp_class_kernel would run with for a pcm whose check and bit degrees fall inside one
 * compiled instance -- (check degrees; bit degrees) = (7; 3..4), (6; 3), (4; 2), (8; 4), (3..4; 1..2) -- and
 * BPOSD_ERR_UNSUPPORTED otherwise.  info[11]: highest check degree, lowest / highest bit degree, bit slots per thread, LDS
 * stride MP, threads per workgroup, modelled read cycles of one bit pass and their floor, modelled write cycles and their
 * floor, lowest check degree.  Nullable outputs: pos_chk [MP], pos_bit [slots * MP], bit_slot [DVHI * slots * MP],
 * grp_deg [slots * MP / 64], grp_cdeg [MP / 64] -- callers size them for MP = 1024, 2 slots, DVHI = 4. */
int bposd_debug_class_layout(const int32_t *csr_indptr, const int32_t *csr_indices, int32_t m, int32_t n, int32_t *pos_chk,
                             int32_t *pos_bit, int32_t *bit_slot, int32_t *grp_deg, int32_t *grp_cdeg, int64_t *info);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* BPOSD_MI355X_DEBUG_H */
