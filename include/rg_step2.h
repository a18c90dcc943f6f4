/* rg_step2.h -- C ABI of the Step-2 quantitative-trait score test (first slice).
 *
 * What it replaces in the reference (`regenie --step 2 --qt`, dense genotypes, the default non-strict mode):
 *   rg_s2_set_null    the per-chromosome constants the tests read: new_cov (orthonormal covariate basis), the scaled
 *                     LOCO residuals `res`, masked_indivs and scf_sv = scale_Y * p_sd_yres      Data.cpp:2386-2400 (compute_res)
 *   rg_s2_qt_block    for one block of variants: mean imputation of the missing entries            Geno.cpp:3183-3188
 *                     G <- G - X (X^T G), scale_fac = |G| / sqrt(n - C), "ignored" below numtol    Geno.cpp:3242-3260 (residualize_geno)
 *                     num = res^T G * gsc, denum = gsc^2 * mask^T G^2, stats = num / sqrt(denum),
 *                     bhat = stats * scf_sv / sqrt(denum), se = bhat / stats, chisq = stats^2      Step2_Models.cpp:343-468 (compute_score_qt)
 *                     called per variant from compute_tests_mt                                      Data.cpp:2476-2555
 *   rg_s2_qt_block_packed  the same for hard calls as they lie in a .bed file, INCLUDING the sparse-genotype branch the
 *                     reference takes per variant (check_sparse_G, Geno.cpp:3165-3177; Step2_Models.cpp:402-413)
 * Not in this slice (the host keeps doing them): reading the LOCO file (blup_read_chr), the MAC / INFO filters, --strict,
 * mse_full, MCC, the p-value and the output lines.  Both entries make check_sparse_G's per-variant choice (rg_s2_set_sparse_rule).
 *
 * Layout: every matrix is row-major with the SAMPLE index fastest -- G is [bs][ldg], X is [C][n], yres and mask are [P][n];
 * n = samples in the analysis, in the caller's order.  A missing genotype is NaN or any value < 0 (regenie's -3).
 * Conventions: 0 on success, <0 on error with rg_s2_last_error(ctx); the library never falls back to the CPU.
 */
#ifndef RG_STEP2_H
#define RG_STEP2_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rg_s2_ctx rg_s2_ctx;

#define RG_S2_OK 0
#define RG_S2_ERR_ARG (-1)
#define RG_S2_ERR_HIP (-2)

#define RG_S2_MAX_COV 64
#define RG_S2_MAX_PHENO 64

/* n samples in the analysis, C covariate basis columns (intercept included, as in new_cov), P phenotypes. */
int rg_s2_create(rg_s2_ctx** out, int device, int64_t n, int32_t n_cov, int32_t n_pheno);
void rg_s2_destroy(rg_s2_ctx* ctx);
const char* rg_s2_last_error(const rg_s2_ctx* ctx);

/* Per chromosome (the LOCO residuals change with it): X [C][n] orthonormal, yres [P][n] already masked and scaled,
 * mask [P][n] bytes (0 / 1), scf_sv [P].  Host pointers; copied to the device. */
int rg_s2_set_null(rg_s2_ctx* ctx, const double* X, const double* yres, const uint8_t* mask, const double* scf_sv);

typedef struct rg_s2_qt_out {
  double* stats;     /* [bs][P]  num / sqrt(denum)  (NaN for an ignored variant)       */
  double* bhat;      /* [bs][P]  effect size on the raw genotype scale                   */
  double* scale_fac; /* [bs]     block_info->scale_fac                                   */
  double* mean;      /* [bs]     mean over the non-missing samples (= 2 * allele freq.)  */
  int32_t* n_obs;    /* [bs]     non-missing samples                                     */
  int32_t* ignored;  /* [bs]     1 when scale_fac < numtol (or nothing observed)         */
  double* total_p;   /* [bs][P]  rg_s2_qt_block_packed only: allele count over the samples observed for the variant and the trait
                                 (what update_trait_counts leaves in af / mac per trait, Geno.cpp:2948-2959)        */
  int32_t* n_obs_p;  /* [bs][P]  rg_s2_qt_block_packed only: those samples' number (ns per trait)                   */
} rg_s2_qt_out;      /* every pointer is a HOST pointer and may be NULL                  */

/* One block of bs variants.  G is a host pointer, or a device pointer when g_on_device != 0 (then it is read in place).
 * se = bhat / stats and chisq = stats^2 are left to the caller. */
int rg_s2_qt_block(rg_s2_ctx* ctx, const double* G, int64_t ldg, int32_t bs, int32_t g_on_device, double numtol,
                   const rg_s2_qt_out* out);

/* The same statistic for HARD CALLS handed over as they lie in a .bed file: row j holds the 2-bit codes of the n analysed samples in
 * the caller's order, sample-fastest, 4 per byte, low bits first (00 -> 2 copies of the counted allele, 01 -> missing, 10 -> 1,
 * 11 -> 0: buildLookupTable, Geno.cpp:2833-2856), rows ld >= ceil(n / 4) bytes apart; flip != 0 counts the other allele
 * (2 - g, the reference's --ref-first).  This entry follows compute_tests_mt for hard calls to the letter: check_sparse_G
 * (Geno.cpp:3165-3177) sends a variant with at most n_samples * (1 - prop_zero_thr) non-zero entries down the sparse branch of
 * compute_score_qt (Step2_Models.cpp:402-413: no residualisation, scale_fac = 1, per-trait denominators with the reference's
 * "X'X = I for every trait" approximation) and the others down the dense branch (exact mask_p^T r^2) -- the two are the same
 * number when mask == 1 everywhere, and differ when phenotypes differ in their missing values (then C * P extra contraction
 * columns are carried; at most 4096 columns in all).  The contractions run on the i8 matrix cores with exact integer sums
 * (csrc/step2_qt.hip); the genotypes are read at 2 bits each.  rows: host pointer, or device when rows_on_device. */
int rg_s2_qt_block_packed(rg_s2_ctx* ctx, const uint8_t* rows, int64_t ld, int32_t bs, int32_t rows_on_device, int32_t flip,
                          double numtol, const rg_s2_qt_out* out);

/* The same statistic for dosages that are INTEGERS in units of 1 / scale -- 8-bit .bgen probabilities (scale 255: G * 255 = p_het + 2 p_hom),
 * .pgen dosages (scale 16384) -- handed over as uint16 rows [bs][ld >= n], 0xFFFF = missing, values <= 2 * scale (scale <= 16384).
 * The row is split into two or three balanced base-128 digit planes on the device and contracted on the i8 matrix cores like the hard
 * calls (exact integer sums; one division by scale at the end), 2 B per genotype over PCIe instead of 8; phenotypes that differ in
 * their missing values are served through masked-sample lists (no extra contraction columns).  Per-variant choice of branch as
 * rg_s2_qt_block_packed; total_p / n_obs_p are not filled.  G: host pointer, or device when g_on_device. */
int rg_s2_qt_block_int(rg_s2_ctx* ctx, const uint16_t* G, int64_t ld, int32_t bs, int32_t g_on_device, int32_t scale, double numtol,
                       const rg_s2_qt_out* out);

/* The contraction primitive under rg_s2_qt_block_packed, for tests of the same shape (the score tests of binary and count traits are
 * functions of such sums: Step2_Models.cpp:471-552 compute_score_bt needs sum w g~^2, X^T W g~ and g~ . (y - p^) per trait).
 * rg_s2_set_columns: n_col fixed fp64 columns [n_col][n] (host, sample-fastest; at most 4096), split once into int8 digit planes.
 * rg_s2_contract_packed: for every row of 2-bit hard calls (coding and flip as in rg_s2_qt_block_packed)
 *   sums   [bs][2][n_col]  sum_i g0_i col_c(i) and sum_i miss_i col_c(i)   (g0 = the call with 0 at missing entries, miss = its indicator;
 *                          the second is left 0 when the block has no missing call)
 *   sq     [bs][n_sq]      sum_i g0_i^2 col_c(i) for the first n_sq columns
 *   counts [bs][4]         number of calls equal to 1, equal to 2, missing, 0
 * exact up to the truncation of a column at 2^-54 of its largest entry; pointers are HOST pointers and may be NULL. */
typedef struct rg_s2_contract_out {
  double* sums;
  double* sq;
  int32_t* counts;
  double* vstat;   /* rg_s2_contract_int only: [bs][4] sum of the observed entries and of their squares (integer units), observed count,
                      observed non-zero count -- exact integers held in doubles */
} rg_s2_contract_out;
int rg_s2_set_columns(rg_s2_ctx* ctx, int32_t n_col, const double* cols, int32_t n_sq);
int rg_s2_contract_packed(rg_s2_ctx* ctx, const uint8_t* rows, int64_t ld, int32_t bs, int32_t rows_on_device, int32_t flip,
                          const rg_s2_contract_out* out);
/* The same sums for integer dosages (uint16 rows in units of 1 / scale, 0xFFFF = missing; see rg_s2_qt_block_int), in genotype units:
 * sums and sq as above (sq in fp64 on the vector units), vstat instead of counts. */
int rg_s2_contract_int(rg_s2_ctx* ctx, const uint16_t* G, int64_t ld, int32_t bs, int32_t g_on_device, int32_t scale,
                       const rg_s2_contract_out* out);

/* check_sparse_G's constants (Geno.cpp:3165-3177): n_samples = params.n_samples (every kept sample of the file, >= n; default n),
 * prop_zero_thr = --prop-zero-thr (default 0.5).  zero_count_rule = 0: a variant is sparse when the non-zero entries of its
 * mean-imputed vector number <= n_samples * (1 - prop_zero_thr) (.bed / .bgen input, n_zero == -1); 1: when its observed zero
 * entries number >= n_samples * prop_zero_thr (.pgen input, which counts n_zero while reading, Geno.cpp:2582-2594). */
int rg_s2_set_sparse_rule(rg_s2_ctx* ctx, int64_t n_samples, double prop_zero_thr, int32_t zero_count_rule);

/* ---- binary and count traits: the score test and its corrections ---------------------------------------------------
 * What it replaces in the reference (`regenie --step 2 --bt [--firth --approx | --spa]`, `--ct`):
 *   rg_s2_bt_set_null     what compute_res_bin / compute_res_count leave per chromosome (Data.cpp:2439-2455): the fitted mean of the null
 *                         model with the LOCO prediction as offset (fit_null_logistic / fit_null_poisson, Step1_Models.cpp:54-140, :225-288 --
 *                         the caller fits it: a C-parameter IRLS), from which the library forms Gamma_sqrt^2 = w, the weighted covariates
 *                         and (X^T W X)^-1; firth_offset = LOCO prediction + X beta of the null Firth model (fit_null_firth,
 *                         Step2_Models.cpp:985-1060), needed by the approximate Firth correction only
 *   rg_s2_bt_score_packed compute_score_bt / compute_score_ct + get_sumstats for a block of hard calls (Step2_Models.cpp:471-622, :2031-2041):
 *   rg_s2_bt_score_int    z = g~ . (y - p^) / sqrt(denum), denum = sum w g~^2 - (X^T W g~)^T (X^T W X)^-1 (X^T W g~), BETA = z / sqrt(denum);
 *                         the same for integer dosages.  The contractions run on the i8 matrix cores (rg_s2_contract_*), the C x C
 *                         algebra per (variant, trait) on the host inside the library.  The block stays on the device for ...
 *   rg_s2_bt_correct      check_pval_snp's second look at the tests the caller flags (|z| above its threshold, Step2_Models.cpp:1987-2029):
 *                         RG_S2_BT_FIRTH_APPROX = fit_firth_logistic_snp_fast (:1158-1253), RG_S2_BT_SPA = run_SPA_test_snp (:2072-2297).
 *                         One workgroup per (variant, trait) pair iterates on the device; fast[t] != 0 selects the reference's carriers-only
 *                         form (sparse variants: check_sparse_G's verdict is returned by the score call; Firth adds MAC < 50).  The reference
 *                         tests the MINOR allele (flip_geno, Geno.cpp:3150-3162: 2 - g when the mean dosage exceeds 1, BETA negated back):
 *                         the statistics returned are those of the coding given, `sparse` and the carriers of the fast forms those of the
 *                         coding the reference tests -- the only two things the flip changes.  The numerator of the statistic is the reference's
 *                         to the digit: Gres . yres for a dense variant, GW . yres -- the genotype not projected -- for a sparse one
 *                         (Step2_Models.cpp:517 / :519, :603 / :605; they differ by the null model's score at its stopping point).
 * Layouts as above: [P][n] / [C][n] sample-fastest host arrays.  The exact Firth test (--firth without --approx) is not behind this ABI. */
typedef struct rg_s2_bt_null {
  int32_t family;             /* 0: binary trait (logistic null model), 1: count trait (Poisson; no corrections) */
  int32_t niter_max;          /* --niter (params->niter_max, Regenie.hpp; 0 = its default of 50): fit_firth_pseudo hands a fit to the
                                 Newton solvers when one of its logistic steps took more iterations (Step2_Models.cpp:1625) */
  const double* X;            /* [C][n] covariates (new_cov) */
  const double* y;            /* [P][n] raw phenotype */
  const uint8_t* mask;        /* [P][n] masked_indivs */
  const double* fitted;       /* [P][n] fitted mean of the null model (probability / rate) */
  const double* firth_offset; /* [P][n] or NULL */
  const uint8_t* pass;        /* [P] or NULL: 0 = the null model of the trait failed, its tests come back as ignored */
} rg_s2_bt_null;
int rg_s2_bt_set_null(rg_s2_ctx* ctx, const rg_s2_bt_null* null_model);

typedef struct rg_s2_bt_out {   /* HOST pointers, each may be NULL */
  double* stats;          /* [bs][P] z (0 for an ignored test) */
  double* bhat;           /* [bs][P] z / sqrt(denum) */
  double* denum;          /* [bs][P] */
  uint8_t* test_ignored;  /* [bs][P] 1: failed null model, or denum below numtol (Step2_Models.cpp:512-517, :596) */
  double* mean;           /* [bs] mean of the observed entries (the imputed value) */
  int32_t* ignored;       /* [bs] 1: nothing observed */
  uint8_t* sparse;        /* [bs] check_sparse_G's verdict (rg_s2_set_sparse_rule) on the allele the reference tests (flip_geno) */
  int32_t* counts;        /* [bs][4] hard calls: calls equal to 1, equal to 2, missing, 0 */
  double* vstat;          /* [bs][4] integer dosages: sum (units of 1 / scale), sum of squares, observed, observed non-zero */
  double* total_p;        /* [bs][P] hard calls: the trait's allele count minus the variant's (update_trait_counts, Geno.cpp:2948-2959) */
  int32_t* n_obs_p;       /* [bs][P] hard calls: the trait's observed-sample count minus the variant's */
} rg_s2_bt_out;
int rg_s2_bt_score_packed(rg_s2_ctx* ctx, const uint8_t* rows, int64_t ld, int32_t bs, int32_t rows_on_device, int32_t flip, double numtol,
                          const rg_s2_bt_out* out);
int rg_s2_bt_score_int(rg_s2_ctx* ctx, const uint16_t* G, int64_t ld, int32_t bs, int32_t g_on_device, int32_t scale, double numtol,
                       const rg_s2_bt_out* out);

#define RG_S2_BT_FIRTH_APPROX 1
#define RG_S2_BT_SPA 2
typedef struct rg_s2_bt_corr {
  double beta, se, chisq;
  double logp;     /* SPA: -log10 of the saddlepoint p-value; Firth: -1 (the caller takes the p-value of chisq = the likelihood ratio) */
  int32_t fail;    /* 1: TEST_FAIL (no convergence, statistic outside the range K' can reach, ...) */
  int32_t reserved;
} rg_s2_bt_corr;
/* Corrections of npair tests of the block LAST scored: (variant[t], trait[t]) = (row of the block, trait).  firth_se != 0: --firth-se. */
int rg_s2_bt_correct(rg_s2_ctx* ctx, int32_t kind, int32_t npair, const int32_t* variant, const int32_t* trait, const uint8_t* fast, int32_t firth_se,
                     rg_s2_bt_corr* out);

/* Device time of the kernels of the last rg_s2_qt_block / rg_s2_qt_block_packed call (hipEvents on the library's stream), in ms. */
double rg_s2_last_kernel_ms(const rg_s2_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif
