/* rg_pgen.h -- C ABI of the PLINK2 .pgen hardcall input of the Step-1 path.
 *
 * It replaces, for `regenie --step 1 --pgen PFX`, the reference's use of its vendored pgenlib:
 *   rg_pgen_open            PgenReader::Load + the checks of prep_pgen        (Geno.cpp:1071-1103,
 *                                                                             pgenlibr.cpp:37-160)
 *   rg_pgen_info            GetRawSampleCt / GetVariantCt / GetMaxAlleleCt /
 *                           DosagePresent                                      (Geno.cpp:1090-1101)
 *   rg_pgen_read_bed_rows   the per-variant ReadHardcalls loop of the Step-1
 *                           block reader                                       (Geno.cpp:1781-1798)
 *   rg_pgen_read_hardcalls  PgenReader::ReadHardcalls(buf, n, thr, idx, 1)     (pgenlibr.cpp:296-321)
 *   rg_pgen_read_dosages    PgenReader::Read(buf, n, thr, idx, 1)              (pgenlibr.cpp:323-349)
 *
 * The rows come out in PLINK1 .bed 2-bit coding (00 hom-ALT, 01 missing, 10 het, 11 hom-REF; sample i
 * in bits 2*(i%4) of byte i/4; padding bits zero), which is what rg_l0_blocks (rg_step1.h) takes, so a
 * pgen run hands the GPU the same bytes as the equivalent bed run.  Host-only code: no device work.
 *
 * Conventions as in rg_step1.h: 0 on success, <0 on error, rg_pgen_last_error(h) gives the message.
 * rg_pgen_open always stores a handle in *out (also on failure, so the message can be read); free it
 * with rg_pgen_close.  Calls on one handle must not overlap (rg_pgen_read_bed_rows spreads its own work over the
 * threads set with rg_pgen_set_threads).
 */
#ifndef RG_PGEN_H
#define RG_PGEN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rg_pgen rg_pgen;

#define RG_PGEN_OK 0
#define RG_PGEN_ERR_ARG (-1)
#define RG_PGEN_ERR_FORMAT (-2)      /* not a pgen file, malformed header or record, read failure */
#define RG_PGEN_ERR_UNSUPPORTED (-3) /* 2-bit rows asked of a file with dosage tracks; multiallelic variants */

/* Opens PATH (the .pgen file itself, not the prefix) and parses its header. */
int rg_pgen_open(rg_pgen** out, const char* path);
void rg_pgen_close(rg_pgen* h);
const char* rg_pgen_last_error(const rg_pgen* h);

/* Any pointer may be NULL.  phase_present: hardcall phase tracks exist (they are ignored, as by ReadHardcalls).
 * dosage_present: some variant carries a dosage track -- regenie then sets dosage_mode (Geno.cpp:1101) and reads every
 * variant with Read(); rg_pgen_read_bed_rows refuses such a file, rg_pgen_read_dosages serves it. */
int rg_pgen_info(const rg_pgen* h, int64_t* n_samples, int64_t* n_variants, int32_t* max_alleles,
                 int32_t* phase_present, int32_t* dosage_present);

/* Worker threads used by rg_pgen_read_bed_rows (default 1): the counterpart of the OpenMP loop over the block's
 * variants in the reference's reader (Geno.cpp:1777-1781, PgenReader::Load(.., nthr)). */
int rg_pgen_set_threads(rg_pgen* h, int32_t n_threads);

/* Decodes n variants (0-based file indices, any order; ascending order keeps the LD-base cache warm)
 * into rows[k * row_stride .. + ceil(n_samples/4)). */
int rg_pgen_read_bed_rows(rg_pgen* h, int64_t n, const int64_t* variant_idx, uint8_t* rows, int64_t row_stride);

/* One variant as PgenReader::Read(buf, n, thr, idx, 1) gives it (pgenlibr.cpp:323-349): the ALT dosage (16-bit value /
 * 16384, in [0, 2]) where the record stores one for the sample, the hardcall 0/1/2 elsewhere, -3 = missing; n_samples
 * doubles.  All three dosage layouts (list, per-sample, bit array); phase tracks in front of them are stepped over. */
int rg_pgen_read_dosages(rg_pgen* h, int64_t variant_idx, double* out);
/* The same for n variants at once, spread over the worker threads of rg_pgen_set_threads: rows[k * row_stride .. + n_samples)
 * doubles -- a block for rg_l0_blocks_f64. */
int rg_pgen_read_dosage_rows(rg_pgen* h, int64_t n, const int64_t* variant_idx, double* rows, int64_t row_stride);

/* One variant as ALT-allele counts 0/1/2, -3 = missing (n_samples doubles): the parity hook against
 * PgenReader::ReadHardcalls. */
int rg_pgen_read_hardcalls(rg_pgen* h, int64_t variant_idx, double* out);

#ifdef __cplusplus
}
#endif
#endif
