/*
 * rg_step1.h -- C ABI of librg_step1_hip.so: regenie's Step-1 stacked ridge regression
 * hot path on MI355X (gfx950).  Plain C, plain pointers and sizes, no torch / C++ types.
 *
 * regenie has no plugin/FFI layer; the seam this library replaces is the function seam
 * inside Data::run_step1 (reference src/Data.cpp:95-133):
 *
 *   get_G -> residualize_genotypes -> calc_cv_matrices -> ridge_level_0      (per SNP block)
 *   ridge_level_1 -> output/make_predictions -> write_predictions (LOCO)      (per phenotype)
 *
 * Each entry point below cites the reference function(s) it stands in for.
 *
 * Conventions
 *   - every function returns 0 on success, <0 on error; rg_last_error(ctx) gives a message whose
 *     text mirrors the reference's `throw` strings (the C++ host re-throws them, Regenie.cpp:72-91)
 *   - matrices passed from the host are column-major doubles exactly as the reference's Eigen
 *     MatrixXd members are laid out (phenodt::phenotypes N x P, new_cov N x C, masked_indivs N x P)
 *   - "N" is params.n_samples (samples kept after --keep/--remove), "N_file" the .fam size
 *   - one rg_ctx per process/GPU; calls on one ctx are serialised by the caller
 */
#ifndef RG_STEP1_H
#define RG_STEP1_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rg_ctx rg_ctx;

#define RG_OK 0
#define RG_ERR_ARG (-1)
#define RG_ERR_HIP (-2)
#define RG_ERR_LOW_VARIANCE (-3) /* Data.cpp:207-209 "has low variance" */
#define RG_ERR_NOT_SPD (-4)
#define RG_ERR_STATE (-5)

#define RG_MEM_HOST 0
#define RG_MEM_DEVICE 1

/* The per-run inputs of the level-0/level-1 math: struct param / phenodt / filter subset that
 * Data::level_0_calculations (Data.cpp:594-727) and ridge_level_1 (Step1_Models.cpp:772-872) read. */
typedef struct rg_problem {
  int64_t n_samples;       /* N: params.n_samples */
  int64_t n_file;          /* N_file: rows of .fam (bed rows hold ceil(N_file/4) bytes per SNP) */
  int32_t n_pheno;         /* P */
  int32_t n_cov;           /* C = params.ncov (columns of the orthonormal basis, incl. intercept) */
  int32_t cv_folds;        /* K in [2,32], or 0 = leave-one-out CV (params.use_loocv) */
  int32_t n_ridge_l0;      /* R0 */
  int32_t ref_first;       /* --ref-first (Geno.cpp:1746) */
  int32_t family;                      /* 0 = logistic (--bt); 1 = Poisson (--ct: ridge_poisson_level_1[_loocv],
                                          Step1_Models.cpp:1429-1758; yraw = counts, offset = null Poisson eta) */
  int64_t n_analyzed;      /* params.n_analyzed */
  const int32_t* cv_sizes; /* K entries, params.cv_sizes (Data.cpp:401-426), sum = N */
  const double* lambda;    /* R0 entries, ALREADY scaled: M*(1-h)/h (Data.cpp:607) */
  const double* X;         /* N x C  col-major: phenodt::new_cov after getBasis, masked rows zero */
  const double* Y;         /* N x P  col-major: phenodt::phenotypes (residualised, scaled) */
  const uint8_t* mask;     /* N x P  col-major: phenodt::masked_indivs */
  const uint8_t* ind_in_analysis; /* N: filter::ind_in_analysis */
  const uint8_t* ind_ignore;      /* N_file or NULL: filter::ind_ignore */
  const double* neff;      /* P: phenodt::Neff */
  int32_t n_blocks_total;  /* B: total level-0 blocks of the whole run (columns of W = B*R0) */
  int32_t max_block_size;  /* params.block_size */
} rg_problem;

/* ---- lifecycle ---------------------------------------------------------------------------- */
/* stream: a hipStream_t to enqueue on (e.g. torch's current stream) or NULL for a private one. */
int rg_create(rg_ctx** out, int device_id, void* hip_stream);
void rg_destroy(rg_ctx* ctx);
const char* rg_last_error(const rg_ctx* ctx);
/* replaces setmem/set_folds state (Data.cpp:401-577): uploads X, Y, masks; builds the fold-aligned
 * sample layout used by every kernel. */
int rg_set_problem(rg_ctx* ctx, const rg_problem* p);
/* Optional, BEFORE rg_set_problem: sizing of the level-0 workspaces (the reference allocates its block matrices per block,
 * Data.cpp:652; here a batch of blocks is worked on at once and its workspaces are allocated once).  max_batch_blocks > 0 caps
 * the SNP blocks of one batch, pipelines > 0 sets the number of level-0 pipelines (default 2), budget_bytes > 0 bounds the
 * bytes of all pipelines' per-block workspaces (default 64 GB; never more than half of the device memory that is free when
 * rg_set_problem runs).  0 = library default.  A driver that streams a small input from files asks for small batches: the
 * set-up cost of a process grows with the bytes it allocates (and the next process waits for them to be reclaimed). */
int rg_set_l0_workspace(rg_ctx* ctx, int32_t max_batch_blocks, int32_t pipelines, int64_t budget_bytes);

/* Optional: caller-owned storage for the level-0 predictors W (e.g. a torch tensor that RCCL
 * all-gathers in place).  Layout: [B*R0][P][Np] doubles, Np = rg_w_rows(ctx).  If never called the
 * library allocates it on first use. */
int64_t rg_w_rows(const rg_ctx* ctx);  /* Np: fold-aligned padded sample count */
int64_t rg_w_bytes(const rg_ctx* ctx); /* bytes of the whole W buffer */
int rg_set_w_buffer(rg_ctx* ctx, void* dev_ptr, int64_t bytes);
/* A rank that only ever produces the predictors of blocks [first_block, first_block + n_blocks) -- a GPU of a sharded
 * run that hands them on by phenotype (rg_l0_finish / an all-to-all by the caller) -- can keep W for that range alone:
 * rg_w_bytes then reports n_blocks*R0*P*Np doubles, rg_w_device_ptr / rg_set_w_buffer refer to a buffer that STARTS at
 * block first_block, and level 1 runs on the exchanged view (rg_set_l1_view).  Call after rg_set_problem, before level 0. */
int rg_set_block_range(rg_ctx* ctx, int32_t first_block, int32_t n_blocks);
void* rg_w_device_ptr(rg_ctx* ctx);

/* ---- level 0 -------------------------------------------------------------------------------
 * Replaces, for nblk SNP blocks at once: get_G/readChunkFromBedFileToG decode+impute
 * (Geno.cpp:1702-1769), residualize_genotypes (Data.cpp:190-224), calc_cv_matrices K-fold branch
 * (Data.cpp:741-751) and ridge_level_0 (Step1_Models.cpp:458-613).
 *   block_ids[b] : global block index (W columns block*R0 .. block*R0+R0-1, Step1_Models.cpp:511)
 *   bs[b]        : SNPs in the block
 *   bed_rows[b]  : pointer to bs[b] packed .bed rows (2 bits/sample, N_file samples), row b*stride
 *   row_stride   : bytes between consecutive SNP rows (>= ceil(N_file/4))
 *   mem_kind     : RG_MEM_HOST or RG_MEM_DEVICE (where bed_rows[b] live)
 * Work is asynchronous: batches alternate between the library's level-0 pipelines, each on a stream of its own, and are joined back
 * onto the ctx stream by rg_sync, rg_l0_finish, rg_l0_get_w / rg_l0_set_w and the level-1 entries -- W (rg_w_device_ptr, a buffer given
 * by rg_set_w_buffer) holds the blocks' predictors only after one of those calls, not merely stream-ordered after rg_l0_blocks. */
int rg_l0_blocks(rg_ctx* ctx, int32_t nblk, const int32_t* block_ids, const int32_t* bs,
                 const uint8_t* const* bed_rows, int64_t row_stride, int mem_kind);
/* Level 0 on NON-INTEGER genotypes (dosages): what level_0_calculations does after readChunkFromPGENFileToG in dosage_mode
 * (Geno.cpp:1773-1822: PgenReader::Read, mean imputation over the analysed non-missing samples, zero for the others) or
 * after a BGEN read.  rows[b] holds bs[b] variants x n_file doubles (row j at rows[b] + j * row_stride doubles, file sample
 * order, ALT dosage in [0, 2], -3 = missing: exactly what rg_pgen_read_dosages / PgenReader::Read return).  Same results
 * contract as rg_l0_blocks (the predictors of the blocks land in W); the arithmetic is fp64 throughout -- standardised
 * genotypes materialised per block, fold Grams and G~ Y as fp64 MFMA GEMMs, the same batched Cholesky -- where the 2-bit path
 * uses exact integer Grams.  K-fold and leave-one-out CV as set in rg_problem.  A value outside [0, 2]
 * other than -3 is reported by the next rg_sync ("... has a value not in [0,2] or missing", Geno.cpp:1819-1820). */
int rg_l0_blocks_f64(rg_ctx* ctx, int32_t nblk, const int32_t* block_ids, const int32_t* bs,
                     const double* const* rows, int64_t row_stride, int mem_kind);
int rg_sync(rg_ctx* ctx); /* waits and reports deferred device-side errors (low variance, not SPD) */

/* Readback in the reference's own layout (= the `--lowmem` file of write_l0_file,
 * Step1_Models.cpp:728-734): out is N x R0 col-major for (block, pheno). */
int rg_l0_get_w(rg_ctx* ctx, int32_t block_id, int32_t pheno, double* out_host);
/* Injects externally produced level-0 predictors (read_l0 analogue, Step1_Models.cpp:1921-1953). */
int rg_l0_set_w(rg_ctx* ctx, int32_t block_id, int32_t pheno, const double* in_host);

/* ---- level 1, quantitative traits, K-fold ---------------------------------------------------
 * Replaces ridge_level_1 (Step1_Models.cpp:772-872), the tau selection of Data::output
 * (Data.cpp:1025-1037), make_predictions (Data.cpp:1196-1266) and the LOCO assembly of
 * write_predictions (Data.cpp:1846-1858).
 *   tau          : R1 x P col-major, ALREADY scaled (check_l0, Step1_Models.cpp:2115-2117)
 *   cols_per_chr : nchr entries: number of W columns (blocks*R0) of each chromosome that has blocks,
 *                  in file order (Data.cpp:1246)
 *   cumsum_out   : 5 x R1 x P (Sx,Sy,Sx2,Sy2,Sxy; ridgel1::cumsum_values)
 *   best_out     : P  (argmin MSE, first minimum)
 *   pred_out     : N x nchr x P  col-major per phenotype (Data::predictions[0]); host memory */
int rg_l1_qt(rg_ctx* ctx, int32_t n_ridge_l1, const double* tau, int32_t nchr,
             const int32_t* cols_per_chr, double* cumsum_out, int32_t* best_out, double* pred_out);

/* ---- multi-GPU level 1 (optional) ----------------------------------------------------------------
 * After the level-0 predictors have been all-gathered (every rank holds the full W), rg_l1_qt can
 * share its two heavy steps among `world` ranks: the fold-Gram tiles are computed tile-cyclically and
 * the K*R1 ridge systems in contiguous ranges; each step is completed by ONE in-place sum all-reduce
 * of a device buffer, performed by the caller's callback (RCCL through torch.distributed, ncclAllReduce
 * in a C++ host, ...) between the library's kernels.  Every rank must then call rg_l1_qt with the same
 * arguments and receives identical results.  The reference has no counterpart (level 1 is single-node,
 * Step1_Models.cpp:772-872); the decomposition follows SURVEY.md section 8(e). */
typedef int (*rg_allreduce_fn)(void* user, void* dev_ptr, int64_t n_doubles);
int rg_set_collective(rg_ctx* ctx, int32_t world, int32_t rank, rg_allreduce_fn fn, void* user);

/* ---- one node, several GPUs: the level-0 hand-off (optional) --------------------------------------------------
 * The reference splits level 0 over jobs through files: write_l0_master deals contiguous block ranges (Data.cpp:270-302),
 * every job writes PFX_job<k>_l0_Y<ph> (write_l0_file, Step1_Models.cpp:728-734) and the --run-l1 job reads them back
 * (prep_parallel_l1, Data.cpp:862-908; read_l0, Step1_Models.cpp:1921-1987).  Here the jobs are the GPUs of a node: one
 * context per GPU, one host thread per context, and ONE exchange of the predictors over xGMI instead of the files.
 *   rg_group_create : binds n contexts (same rg_problem on each) to a transport: RG_TRANSPORT_RCCL (one RCCL communicator
 *                     per context, librccl resolved at run time) or RG_TRANSPORT_PEER (direct device-to-device copies;
 *                     also valid for contexts sharing one device, which is how world-size-2 runs are tested on one GPU).
 *   rg_l0_finish    : called by EVERY rank from its own thread after its rg_l0_blocks calls.  block_begin[n+1] = the
 *                     contiguous block range of each rank.  pheno_begin[n+1] != NULL (needs P >= n): all-to-all by
 *                     phenotype -- the rank receives the rows of every rank's blocks for its phenotypes only and its
 *                     level-1 view is set to that range (rg_set_l1_view); NULL: all-gather of the block slabs, after
 *                     which rg_l1_qt shares its Gram tiles / ridge systems among the ranks (rg_set_collective with the
 *                     group's RCCL all-reduce).  Returns after the exchange has completed on this rank.
 *   rg_group_prepare: optional, per rank, any time after rg_group_create: allocates that rank's exchange buffers (the
 *                     phenotype view and, for RCCL, the packed send buffer) so that rg_l0_finish finds them ready.
 *   rg_group_abort  : a rank whose host side failed for good (file error, exception) calls this INSTEAD of its next group
 *                     call: the group is broken, every rank waiting in rg_l0_finish or in a shared level-1 all-reduce -- now
 *                     or later -- returns an error instead of waiting for the failed rank.  rg_l0_finish itself agrees on
 *                     success among all ranks before any of them enters a collective.  A broken group cannot be reused. */
#define RG_TRANSPORT_RCCL 0
#define RG_TRANSPORT_PEER 1
typedef struct rg_group rg_group;
int rg_group_create(rg_group** out, int32_t n, rg_ctx* const* ctxs, int transport);
void rg_group_destroy(rg_group* g);
int rg_l0_finish(rg_group* g, int32_t rank, const int32_t* block_begin, const int32_t* pheno_begin);
int rg_group_prepare(rg_group* g, int32_t rank, const int32_t* block_begin, const int32_t* pheno_begin);
void rg_group_abort(rg_group* g, int32_t rank);

/* ---- pinned host memory for streamed ingest (optional) ------------------------------------------------------------
 * Rows handed to rg_l0_blocks with RG_MEM_HOST cross PCIe by asynchronous copies only if they live in page-locked
 * memory; a reader thread can then fill the next buffer while the GPU works on the previous ones (the block loop of
 * Data.cpp:636-678 with the file read taken off the critical path).  rg_ingest_fence blocks the calling host thread
 * until every host-to-device copy issued so far by rg_l0_blocks / rg_l0_blocks_f64 has completed, i.e. until the buffers
 * passed to those calls may be overwritten. */
int32_t rg_l0_batch_blocks(const rg_ctx* ctx); /* SNP blocks the library works on as one batch: the natural size of one rg_l0_blocks call */
void* rg_host_alloc(int64_t bytes);
void rg_host_free(void* p);
/* Page-locks a range the caller owns, so that rows inside it cross PCIe by asynchronous copies without an intermediate buffer.
 * read_only = 1 for memory the caller may only read -- a file mapping: the pages of the page cache then ARE the source of the copies
 * (a .bed mapped and registered this way needs neither a read nor a host buffer).  0 on success; rg_host_unregister before unmapping. */
int rg_host_register(void* ptr, int64_t bytes, int read_only);
int rg_host_unregister(void* ptr);
int rg_ingest_fence(rg_ctx* ctx);

/* ---- the genotype file staged in HBM as a whole (optional; one GPU, the C++ driver's default for a .bed that fits) ----
 * An MI355X holds 288 GB: the 62.5 GB .bed of BASELINE configs[2] fits beside W and the workspaces, and nothing about its bytes depends
 * on the phenotypes -- so the driver starts copying the file the moment the runtime is up, while it still parses the text files, and
 * level 0 reads the rows in place (rg_l0_blocks with RG_MEM_DEVICE, rows at the file's own pitch).  The reference reads a block when
 * the block loop reaches it (Data.cpp:636-678 -> Geno.cpp:1702-1769).
 *   rg_stage_alloc : `bytes` of device memory on the context's device, or NULL (also when `bytes` exceeds `max_frac_of_free` of the
 *                    memory that is free at the time: the caller then streams the file through rg_l0_blocks(RG_MEM_HOST) as before)
 *   rg_stage_copy  : host -> device copy of `bytes` (pageable or page-locked source) on a stream of its own, from any ONE thread at a
 *                    time; returns when the bytes have arrived.  Not ordered with the context's streams: hand rows to rg_l0_blocks
 *                    only after the copy that covers them has returned.
 *   rg_stage_free  : frees the buffer (after rg_sync; or when the caller gives the stage up: rg_stage_fits). */
void* rg_stage_alloc(rg_ctx* ctx, int64_t bytes, double max_frac_of_free);
int rg_stage_copy(rg_ctx* ctx, void* dev_dst, const void* host_src, int64_t bytes);
void rg_stage_free(rg_ctx* ctx, void* dev_ptr);
/* 1 when `bytes` more of device memory are free on the context's device right now (the driver asks once the phenotypes are parsed whether W,
 * the workspaces and level 1 still fit beside the staged file, and gives the stage up if not), else 0. */
int rg_stage_fits(rg_ctx* ctx, int64_t bytes);

/* ---- phenotype-sharded level 1 (optional) ------------------------------------------------------------
 * With P >= world phenotypes the ranks can exchange predictor slabs by phenotype instead of all-gathering W:
 * rank g receives, from every rank, the columns of that rank's blocks for ITS phenotypes only (an all-to-all
 * of 1/world of the all-gather volume) into a buffer laid out [L][count][Np] (Np = rg_w_rows), and then runs
 * the level-1 entry points on that phenotype range.  tau / yraw / offset inputs and every output of the
 * rg_l1_* calls are then arrays for `pheno_count` phenotypes.  w_dev = NULL with the full range restores the
 * default view (the context's own W, all phenotypes); w_dev = NULL with a sub-range runs level 1 of those phenotypes on the
 * context's own W (the C++ driver takes the phenotypes one at a time and writes each one's files while the next is computed). */
int rg_set_l1_view(rg_ctx* ctx, const void* w_dev, int32_t pheno_begin, int32_t pheno_count);

/* ---- LOCO output of the level-1 entry points (optional) ---------------------------------------------------
 * Replaces the assembly loop of write_predictions (Data.cpp:1846-1858): pred_loco[:, c] = rowsum(predictions) -
 * predictions[:, idx(c)] for each of `nchrom` chromosomes (chromosomes without blocks get the full sum).
 * After rg_set_loco_output(ctx, nchrom, chrom_ids) the rg_l1_* calls write, per phenotype, nchrom x N doubles
 * (row c-1 = LOCO predictions leaving chromosome c out) into pred_out instead of the nchr x N per-chromosome
 * predictions; chrom_ids[k] (1-based, <= nchrom) is the chromosome of the k-th entry of cols_per_chr.
 * nchrom = 0 restores the per-chromosome output. */
int rg_set_loco_output(rg_ctx* ctx, int32_t nchrom, const int32_t* chrom_ids, int32_t n_ids);

/* ---- level 1, quantitative traits, leave-one-out CV ------------------------------------------
 * Replaces ridge_level_1_loocv (Step1_Models.cpp:875-962), the tau selection of Data::output and
 * make_predictions_loocv (Data.cpp:1269-1342).  Requires a problem set up with cv_folds = 0.
 * Arguments as rg_l1_qt; cumsum_out rows Sy / Sy2 hold the reference's presets 0 and Neff - ncov. */
int rg_l1_qt_loocv(rg_ctx* ctx, int32_t n_ridge_l1, const double* tau, int32_t nchr,
                   const int32_t* cols_per_chr, double* cumsum_out, int32_t* best_out,
                   double* pred_out);

/* ---- level 1, binary traits: logistic ridge, K-fold or LOOCV (by the problem's cv_folds) ------
 * Replaces ridge_logistic_level_1 (Step1_Models.cpp:966-1156) / ridge_logistic_level_1_loocv +
 * run_log_ridge_loocv (Step1_Models.cpp:1159-1374), the -logLik tau selection (Data.cpp:1030) and
 * make_predictions_binary / make_predictions_binary_loocv (Data.cpp:1346-1427, :1484-1571).
 *   yraw, offset  : N x P col-major: phenodt::phenotypes_raw (0/1) and ests::offset_nullreg
 *   opt           : iteration limits / tolerances (NULL = the reference defaults below)
 *   cumsum_out    : 6 x R1 x P (Sx,Sy,Sx2,Sy2,Sxy,-logLik)
 *   converged_out : P; 0 marks pheno_l1_not_converged (its predictions are skipped, Data.cpp:1016) */
typedef struct rg_bt_options {
  int32_t niter_max_ridge;             /* 100   params.niter_max_ridge */
  int32_t niter_max_line_search_ridge; /* 100   params.niter_max_line_search_ridge */
  int32_t niter_max_line_search;       /* 25    params.niter_max_line_search (LOOCV Newton) */
  int32_t family;                      /* 0 = logistic (--bt); 1 = Poisson (--ct: ridge_poisson_level_1[_loocv],
                                          Step1_Models.cpp:1429-1758; yraw = counts, offset = null Poisson eta) */
  double l1_ridge_tol;                 /* 1e-4  params.l1_ridge_tol */
  double tol;                          /* 1e-8  params.tol */
  /* optional outputs of the K-fold route (NULL = not wanted; ignored by LOOCV), the carriers of ridgel1 (Step1_Models.hpp:52-75): */
  double* beta_out;                    /* [P][K][n_ridge_l1][L]: beta_hat_level_1[ph][fold] -- the coefficients of every fold model at every ridge value */
  double* fold_cumsum_out;             /* [P][K][6][n_ridge_l1]: each fold's own share of cumsum_out (its held-out Sx, Sy, Sx2, Sy2, Sxy, -logLik) */
} rg_bt_options;
int rg_l1_bt(rg_ctx* ctx, int32_t n_ridge_l1, const double* tau, const double* yraw,
             const double* offset, const rg_bt_options* opt, int32_t nchr,
             const int32_t* cols_per_chr, double* cumsum_out, int32_t* converged_out,
             int32_t* best_out, double* pred_out);

/* ---- level 1, time-to-event traits (--t2e): Cox ridge, K-fold ------------------------------------
 * Replaces ridge_cox_level_1 (Step1_Models.cpp:2228-2305) with cox_ridge / cox_ridge_path (cox_ridge.cpp:8-302) and
 * survival_data::setup (survival_data.cpp:9-150), the penalty grid of check_l0 (Step1_Models.cpp:2105-2113) from
 * getCoxLambdaMax (:446-450), the deviance selection (Data.cpp:1026-1050) and make_predictions_cox (Data.cpp:1714-1755).
 * One trait per call: `pheno` is the phenotype of the problem whose level-0 predictors and sample mask belong to the TIME column.
 *   time, event, offset : N each, sample order: phenodt::phenotypes_raw of the time and the event column (0 / 1; entries of masked
 *                         samples are not read) and ests::offset_nullreg (the null Cox model's linear predictor)
 *   opt                 : iteration limits / tolerances (NULL = the reference defaults below)
 *   tau_out, deviance_out : n_ridge_l1 each: the penalties (largest first) and the held-out deviances summed over the folds
 *   converged_out       : 0 marks pheno_l1_not_converged (predictions skipped, Data.cpp:1016); best_out: index of the smallest deviance
 *   pred_out            : nchr x N (or nchrom x N in LOCO output mode) for this trait */
typedef struct rg_cox_options {
  int32_t niter_max;                   /* 50    params.niter_max (not used by level 1; kept next to its line-search twin) */
  int32_t niter_max_line_search;       /* 25    params.niter_max_line_search */
  int32_t niter_max_ridge;             /* 100   params.niter_max_ridge */
  int32_t niter_max_line_search_ridge; /* 100   params.niter_max_line_search_ridge */
  double numtol_cox;                   /* 2.5e-4 params.numtol_cox */
  double l1_ridge_tol;                 /* 1e-4  params.l1_ridge_tol */
  const double* tau;                   /* NULL: the path lambda_max * 1e-6^(j/(R1-1)) from the score at beta = 0 (check_l0, Step1_Models.cpp:2111-2113);
                                        * else n_ridge_l1 penalties of the caller's, e.g. --t2e-l1-pi6: L (1 - h_j) / h_j * 6 / pi^2 (:2106-2110) */
} rg_cox_options;
int rg_l1_cox(rg_ctx* ctx, int32_t pheno, int32_t n_ridge_l1, const double* time, const double* event, const double* offset,
              const rg_cox_options* opt, int32_t nchr, const int32_t* cols_per_chr, double* tau_out, double* deviance_out,
              int32_t* converged_out, int32_t* best_out, double* pred_out);

/* ---- introspection used by bench.py (timing of the dominant kernels with HIP events) --------- */
typedef struct rg_timing {
  double ms_prep, ms_xy, ms_gram, ms_assemble, ms_chol, ms_solve, ms_pred, ms_l1_gram, ms_l1_chol,
      ms_l1_pred;
  int64_t n_gram_launches, n_chol_launches;
  /* the iterative level-1 models (rg_l1_bt, rg_l1_cox): what their time is made of.  The counts are kept whether or not timing is
   * enabled; the three ms fields need rg_enable_timing (they bracket the launches with events and wait for them). */
  double ms_wgram;          /* weighted Grams X^T W X (+ slice reduction, + X^T W z row) */
  double ms_irls_solve;     /* the ridge systems of the IRLS steps (batched Cholesky) / the Cox sweep */
  double ms_irls_stream;    /* the passes over the predictors that evaluate a state: eta, weights, deviance, score */
  int64_t n_wgram;          /* chain Grams formed: one per (fold model or LOOCV model, IRLS step) */
  int64_t n_irls_rounds;    /* lock-step rounds (a round advances every unfinished fold model by one IRLS step) */
  int64_t wgram_positions;  /* sum over the chain Grams of the sample positions contracted (their flop count is positions * L * (L + 1)) */
  int64_t n_wgram_approx_rounds; /* rounds whose Grams were the quasi-Newton ones (16-bit operand planes, wgram_bf16.hip) rather than fp64 */
  int64_t n_irls_passes;         /* streaming passes over a phenotype's predictors in ms_irls_stream: one per k_bt_eval / k_bt_score launch */
} rg_timing;
int rg_enable_timing(rg_ctx* ctx, int on); /* wraps kernel groups in hipEvents on the ctx stream */
int rg_get_timing(rg_ctx* ctx, rg_timing* out);

/* ---- single-kernel entry points (device pointers) used by the parity tests -------------------
 * These expose the individual HIP kernels so tests/ can check each against the oracle. */
/* C[m][n] (int32, ldc) = sum_k dec(A[m][k]) * dec(B[n][k]) over packed 2-bit rows (cleaned layout:
 * code 00->2, 10->1, 11->0, 01->missing->0 [a_miss/b_miss=0] or indicator [=1]); K4 = bytes per row
 * to contract (multiple of 16). */
int rg_k_gram_i8(void* stream, const uint8_t* A, int64_t lda, int a_miss, const uint8_t* B,
                 int64_t ldb, int b_miss, int32_t m, int32_t n, int64_t k_bytes, int32_t* C,
                 int64_t ldc);
/* C[m][n] (int32, ldc) = sum_k A4[m][k] * B4[n][k] over FP4-E2M1 rows (two samples per byte, values
 * 0000 = 0, 0010 = 1, 0100 = 2) on v_mfma_scale_f32_32x32x64_f8f6f4 with exact flushing to int32;
 * k_bytes = bytes per row to contract (multiple of 128 = one LDS stage), lda/ldb multiples of 16. */
int rg_k_gram_fp4(void* stream, const uint8_t* A4, int64_t lda, const uint8_t* B4, int64_t ldb,
                  int32_t m, int32_t n, int64_t k_bytes, int32_t* C, int64_t ldc);
/* batched in-place Cholesky with appended RHS rows: mats[b] is (n_pad + rhs_pad) x n_pad row-major
 * (ld = n_pad), lower triangle referenced; on exit rows < n_pad hold L, RHS rows hold solutions x
 * (A x = rhs).  n_pad and rhs_pad multiples of 64. */
int rg_k_chol_solve(void* stream, double* mats, int64_t mat_stride, int32_t batch, int32_t n_pad,
                    int32_t rhs_pad, int32_t nrhs /* valid RHS rows <= rhs_pad */,
                    double* dinv_ws /* batch*(T + 10*ceil(T/4))*4096 doubles, T = n_pad/64 */,
                    int32_t* info /* device int, set !=0 if not SPD */);
/* C[m][n] = sum_k A[m][k]*B[n][k] (fp64 MFMA, row-major, K multiple of 64, m,n multiples of 64) */
int rg_k_dgemm_nt(void* stream, const double* A, int64_t lda, const double* B, int64_t ldb,
                  int32_t m, int32_t n, int64_t k, double* C, int64_t ldc);

/* register-only MFMA issue-rate micro-benchmark: kind 0 = v_mfma_f64_16x16x4_f64 (TFLOP/s),
 * kind 1 = v_mfma_i32_32x32x32_i8 (TOP/s); the measured ceilings bench.py quotes next to the peaks. */
int rg_k_mfma_peak(int kind, int iters, double* tera_ops_out);

#ifdef __cplusplus
}
#endif
#endif /* RG_STEP1_H */
