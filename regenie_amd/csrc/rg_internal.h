// Internal declarations shared by the HIP translation units of librg_step1_hip.so.
// Product code: nothing here may reference oracle/.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <functional>
#include <string>
#include <vector>

#include "../../include/rg_step1.h"

#define RG_MAX_SEG 32      // max CV folds handled by the fold-aligned layout
#define RG_TILE_G 128      // i8 Gram output tile
#define RG_TILE_C 64       // fp64 Cholesky / Gram tile

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef double v4d __attribute__((ext_vector_type(4)));

static inline int64_t rg_round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }
// doubles of workspace (`dinv`) one batched Cholesky call needs: the 64x64 tile inverses (T per system) followed by the
// operand images of the diagonal blocks (10 tiles per group of 4 tile columns), see chol.hip
static inline size_t rg_chol_ws_doubles(size_t batch, int n64) {
  const size_t T = (size_t)n64 / 64;
  return batch * (T + (T + 3) / 4 * 10) * 4096;
}

// Fold-aligned sample layout ("position space"): fold f occupies positions
// [pos_start[f], pos_start[f]+len[f]) which map to .fam indices [file_start[f], +len[f]);
// every fold starts at a multiple of 256 positions, so a fold boundary never cuts a 128-byte LDS stage
// of the FP4 Gram kernel, a packed 16-byte K-step of the i8 Gram kernel nor a 64-sample chunk of the
// fp64 kernels.
struct SegLayout {
  int32_t nseg;
  int32_t pad_;
  int64_t pos_start[RG_MAX_SEG];
  int64_t file_start[RG_MAX_SEG];
  int64_t len[RG_MAX_SEG];      // in file samples
  int64_t plen[RG_MAX_SEG];     // padded length (multiple of 256)
};

struct rg_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t st_stage = nullptr;    // rg_stage_copy's own (non-blocking) stream, created on first use
  hipStream_t st_part[3] = {};       // further streams of a level-0 batch's Cholesky (rg_api.hip: ranges of systems side by side)
  hipEvent_t ev_fork = nullptr, ev_part[3] = {};
  bool timing_serial = false;        // RG_CHOL_SPLIT=0 behaviour forced (unused)
  bool own_stream = false;
  std::string err;
  bool have_problem = false;
  bool loocv = false;            // cv_folds == 0: leave-one-out CV, one sample segment
  int64_t rtot_wk = 0;           // rows of one ridge system workspace (LOOCV: + Np appended sample rows)
  int nsys = 0;                  // systems per block: K*R0 (K-fold) or R0 (LOOCV)
  double* d_gt = nullptr;        // [nblk][loo_chunk][n64] standardised genotypes of a chunk of samples, sample-major (LOOCV)
  double* d_zt = nullptr;        // [nblk][n64][loo_chunk] their transforms Q^T g~ (loocv_tri.hip)
  double* d_triws = nullptr;     // tridiagonal reductions of a batch: A, Q^T, d, e, step vectors, recurrence tables
  int64_t loo_chunk = 0;
  double* d_lpart = nullptr;     // LOOCV standardisation partial sums

  // problem
  int64_t N = 0, Nfile = 0, Np = 0, n_analyzed = 0;
  int P = 0, C = 0, K = 0, R0 = 0, ref_first = 0, B_total = 0, bs_max = 0;
  int n_active = 0;
  std::vector<int64_t> fold_cstart;  // compact start of each fold (K+1)
  std::vector<int64_t> h_posc;       // position of each compact sample (host copy)
  std::vector<double> lambda, neff;
  SegLayout seg;

  // device: per-position side arrays
  int32_t* d_cidx = nullptr;     // [Np] compact index or -1
  uint8_t* d_act = nullptr;      // [Np/4] 2-bit 11 for samples in the analysis
  double* d_V = nullptr;         // [(C+P)][Np]  X then Y in position space
  double* d_maskp = nullptr;     // [P][Np] 0/1 masked_indivs in position space
  double* d_Q = nullptr;         // [nseg][C][C]   X_f^T X_f
  double* d_XtY = nullptr;       // [nseg][C][P]   X_f^T Y_f
  double* d_lambda = nullptr;    // [R0]
  double* d_neff = nullptr;      // [P]
  uint8_t* d_keptp = nullptr;    // [Np] 1 if the position is one of the N kept samples
  int64_t* d_posc = nullptr;     // [N] position of each compact sample
  double* d_zero = nullptr;      // [Np] zeros (padding rows of the level-1 Gram)

  // W
  double* d_W = nullptr;         // [B*R0][P][Np]
  bool own_W = false;
  int64_t W_bytes = 0;
  int w_b0 = 0, w_nb = 0;        // W holds the predictor rows of blocks [w_b0, w_b0 + w_nb) only (rg_set_block_range; default: all)

  // level-0 workspaces (sized for NBLK blocks); the caller's sizing wishes (rg_set_l0_workspace), 0 = library default
  int ws_nblk = 0, ws_pipes = 0;
  int64_t ws_budget = 0;
  int nblk_cap = 0;
  int n128 = 0, n64 = 0, rtot = 0;  // paddings for bs_max
  uint8_t* d_raw = nullptr;  int64_t raw_ld = 0;    // staged raw rows [nblk][bs_max][raw_ld]
  uint8_t* d_pk = nullptr;   int64_t pk_ld = 0;     // cleaned packed   [nblk][n128][pk_ld]
  uint8_t* d_pk4 = nullptr;  int64_t pk4_ld = 0;    // FP4 dosage plane [nblk][n128][pk4_ld] (gram_fp4.hip)
  bool gram_fp4 = true;          // dosage x dosage Gram on the FP4 matrix cores (RG_GRAM=i8 selects the i8 kernel)
  double* d_mu = nullptr;        // [nblk][n128]
  int32_t* d_nmiss = nullptr;    // [nblk]
  double* d_xypart = nullptr;    // [nblk][nchunk][n128][2][Cv]
  int32_t xy_nchunk = 0;
  std::vector<int32_t> h_chunk_seg; std::vector<int64_t> h_chunk_pos, h_chunk_len;
  int32_t* d_chunk_seg = nullptr; int64_t* d_chunk_pos = nullptr; int64_t* d_chunk_len = nullptr;
  // finer position chunk tables: 1024 positions (level-0 predictions), 256 (level-1 CV / predictions)
  int32_t n_c1k = 0, n_c256 = 0;
  int32_t* d_c1k_seg = nullptr; int64_t* d_c1k_pos = nullptr; int64_t* d_c1k_len = nullptr;
  int32_t* d_c256_seg = nullptr; int64_t* d_c256_pos = nullptr; int64_t* d_c256_len = nullptr;
  int32_t* d_S = nullptr;        // [nblk][nseg][2*n128][2*n128] int32 stacked Gram
  double* d_F = nullptr;         // [nblk][nseg][n128][C]
  double* d_Bm = nullptr;        // [nblk][n128][C]
  double* d_BQ = nullptr;        // [nblk][nseg][n128][C]
  double* d_GYt = nullptr;       // [nblk][nseg][n128][P]
  double* d_sc = nullptr;        // [nblk][n128]  scale_G
  double* d_fold = nullptr;      // [nblk][nseg][rtot][n64]
  double* d_sum = nullptr;       // [nblk][rtot][n64]
  double* d_wk = nullptr;        // [nblk][nseg*R0][rtot][n64]
  double* d_dinv = nullptr;      // [nblk*nseg*R0][n64/64][64*64]
  double* d_beta = nullptr;      // [nblk][nseg*R0][P][n64]  beta / scale_G
  double* d_cb = nullptr;        // [nblk][nseg*R0][P][C]
  int8_t* d_vd = nullptr;        // xy_i8.hip: digit planes of V = [X | Y]  [Cv][8][Np]
  double* d_vsc = nullptr;       //            column scales [Cv]
  int32_t* d_xyS = nullptr;      //            integer sums [nblk][2][nseg][n128][128]
  int32_t* d_segid = nullptr;    //            [nseg] = 0 .. nseg-1 (the per-fold partials are read with chunk == fold)
  int8_t* d_bplanes = nullptr;   // pred_i8.hip: digit planes [nblk][nseg][ngrp][2][8][64][n128]
  double* d_bsc = nullptr;       //              row scales   [nblk][nseg][ngrp][2][64]
  uint8_t* d_pkT = nullptr;      //              SNP-contiguous packed rows [nblk][Np][n128/4]
  double* d_psum = nullptr;      // [nblk][n_c256][P][8][2]
  double* d_pstat = nullptr;     // [nblk][P][8][2] column mean and 1/sd
  int32_t* d_info = nullptr;     // [4] deferred error flags: [0]=low variance, [1]=not SPD
  int32_t* d_bs = nullptr;       // [nblk]
  int32_t* d_blockid = nullptr;  // [nblk]
  const uint8_t** d_rawptr = nullptr;  // [nblk] device pointers to the raw .bed rows of each block
  std::vector<const uint8_t*> h_rawptr;
  std::vector<int> block_done;

  // level-1 phenotype view (rg_set_l1_view): phenotypes [v_p0, v_p0 + v_np) with predictors in v_W [L][v_np][Np]
  // (v_W == nullptr: the context's own W with all P phenotypes)
  const double* v_W = nullptr;
  int v_p0 = 0, v_np = 0;

  // LOCO output mode of the level-1 entry points (rg_set_loco_output): nchrom rows per phenotype instead of nchr
  int loco_nchrom = 0;
  std::vector<int32_t> loco_chrom;   // chromosome (1-based) of each cols_per_chr entry

  // multi-GPU level 1: tile-sharded Gram and system-sharded solves, completed by caller-provided all-reduces
  int coll_world = 1, coll_rank = 0;
  rg_allreduce_fn coll_allreduce = nullptr;
  void* coll_user = nullptr;

  // further level-0 pipelines: a chain of child contexts, each with its own workspaces and HIP stream, sharing W.
  // Batches are dealt round-robin, so the streaming phases (ingest, covariate products, FP4 Gram, assemble, predictions) of
  // one batch overlap the latency/HBM-bound Cholesky of the other.  Joined back onto `stream` with events.
  rg_ctx* twin = nullptr;       // next context of the pipeline chain
  int n_pipe = 1;               // contexts in the chain (set on the parent)
  bool is_child = false;
  hipEvent_t ev_tw_fork = nullptr, ev_tw_join = nullptr;
  bool join_pending = false;    // children hold level-0 work not yet joined onto `stream` (lazy join, rg_api.hip)
  int pipe_rr = 0;              // round-robin cursor over the pipelines, persistent across rg_l0_blocks calls

  // streamed ingest: recorded on this context's stream right after the host-to-device copies of a level-0 batch
  hipEvent_t ev_ingest = nullptr;
  bool ingest_pending = false;

  // level-1 workspaces, kept across calls (hipMalloc/hipFree per call costs milliseconds)
  void* ws_ptr[16] = {};
  size_t ws_bytes[16] = {};

  // fp64 genotype input (l0_f64.hip): grows-only device buffers of the dosage path
  void* f64_ptr[10] = {};
  size_t f64_bytes[10] = {};

  // timing
  bool timing = false;
  rg_timing tm{};
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
};

// address of row 0 of block 0 in W's [B*R0][P][Np] indexing (d_W itself starts at block w_b0)
static inline double* rg_w_base(const rg_ctx* c) { return c->d_W ? c->d_W - (int64_t)c->w_b0 * c->R0 * c->P * c->Np : nullptr; }

// grows-only device workspace slot; returns nullptr on allocation failure
static inline void* rg_ws(rg_ctx* ctx, int slot, size_t bytes) {
  if (bytes == 0) bytes = 8;
  if (ctx->ws_bytes[slot] < bytes) {
    if (ctx->ws_ptr[slot]) hipFree(ctx->ws_ptr[slot]);
    ctx->ws_ptr[slot] = nullptr; ctx->ws_bytes[slot] = 0;
    if (hipMalloc(&ctx->ws_ptr[slot], bytes) != hipSuccess) return nullptr;
    ctx->ws_bytes[slot] = bytes;
  }
  return ctx->ws_ptr[slot];
}

#if defined(__HIPCC__)
// direct global -> LDS copy of 16 bytes per lane (1 KB per wave) issued through inline assembly: hipcc orders every LDS
// read behind ALL pending global_load_lds it knows of (an s_waitcnt vmcnt(0) in front of the first ds_read of each unit),
// which drains a prefetch ring, and it bunches builtin copies up instead of leaving them between MFMAs; issued this way
// the copies are ordered by the caller's own counted waits (chol.hip) or sit where they are put (gram_fp4.hip).
// sbase: wave-uniform base, voff: per-lane byte offset, lds_addr: wave-uniform LDS byte address of the 1 KB destination.
__device__ __forceinline__ void glds16(const void* sbase, uint32_t voff, uint32_t lds_addr) {
  const uint64_t a = reinterpret_cast<uint64_t>(sbase);   // uniform by construction; pin it to scalar registers
  const uint64_t sa = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) |
                      (uint32_t)__builtin_amdgcn_readfirstlane((int)a);
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sa),
               "s"(__builtin_amdgcn_readfirstlane((int)lds_addr))
               : "memory", "m0");
}

// the same copy with a per-lane 64-bit global address (rows that are not at a fixed stride from one base: the level-1 Gram's
// predictor / padding rows); lds_addr: wave-uniform LDS byte address of the 1 KB destination (lane l lands at + 16 l)
__device__ __forceinline__ void glds16p(const void* gptr, uint32_t lds_addr) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gptr),
               "s"(__builtin_amdgcn_readfirstlane((int)lds_addr))
               : "memory", "m0");
}

#endif

#define RG_HIP(call)                                                                   \
  do {                                                                                 \
    hipError_t e_ = (call);                                                            \
    if (e_ != hipSuccess) {                                                            \
      ctx->err = std::string(#call) + ": " + hipGetErrorString(e_);                    \
      return RG_ERR_HIP;                                                               \
    }                                                                                  \
  } while (0)

// ---- launchers implemented in the .hip files ------------------------------------------------
// bed_prep.hip
void rg_launch_bed_prep(hipStream_t st, const uint8_t* const* rawptr, int64_t raw_ld,
                        uint8_t* pk, int64_t pk_ld, int64_t pk_blk_stride, const int32_t* d_bs,
                        int nblk, int n128, const uint8_t* act, SegLayout seg, int64_t Np,
                        int ref_first, int n_active, double* mu, int32_t* nmiss, uint8_t* pk4,
                        int64_t pk4_ld, int64_t pk4_blk_stride);
void rg_launch_geno_xy(hipStream_t st, const uint8_t* pk, int64_t pk_ld, int64_t pk_blk_stride,
                       const int32_t* d_bs, const int32_t* nmiss, int nblk, int n128, const double* V, int64_t Np,
                       int Cv, const int64_t* chunk_pos, const int64_t* chunk_len, int nchunk, double* part);
// gram_i8.hip
void rg_launch_gram_blocks(hipStream_t st, const uint8_t* pk, int64_t pk_ld, int64_t pk_blk_stride,
                           int nblk, int n128, SegLayout seg, const int32_t* nmiss, int32_t* S, int miss_only);
// gram_fp4.hip
void rg_launch_gram_fp4_blocks(hipStream_t st, const uint8_t* pk4, int64_t pk4_ld, int64_t pk4_blk_stride, int nblk,
                               int n128, SegLayout seg, int32_t* S);
void rg_launch_gram_fp4_generic(hipStream_t st, const uint8_t* A, int64_t lda, const uint8_t* B, int64_t ldb, int m,
                                int n, int64_t kbytes, int32_t* C, int64_t ldc);
void rg_launch_gram_generic(hipStream_t st, const uint8_t* A, int64_t lda, int a_miss,
                            const uint8_t* B, int64_t ldb, int b_miss, int m, int n, int64_t kbytes,
                            int32_t* C, int64_t ldc);
// assemble.hip
struct AsmArgs {
  int nblk, nseg, n128, n64, rtot, C, P, Cv, nchunk;
  int64_t n_analyzed;
  const int32_t* bs; const int32_t* chunk_seg;
  const double* part; const double* mu; const int32_t* S; const int32_t* nmiss;
  const double* Q; const double* XtY;
  double *F, *Bm, *BQ, *GYt, *sc, *fold, *sum;
  int32_t* info;
  int diff_mode;   // 1: fold[f] holds (sum over folds) - (fold f), i.e. the training-fold matrix; sum is not written
  int embed = 0;   // 1: the right-hand sides are rows bs .. bs + P - 1 of a block's matrices (chol.hip, "embedded right-hand sides")
};
void rg_launch_rowstats(hipStream_t st, const AsmArgs& a);
void rg_launch_assemble(hipStream_t st, const AsmArgs& a);
// chol.hip  (path: 0 = group-wise / throughput, 1 = per-column / latency, -1 = by batch size)
void rg_launch_chol_solve(hipStream_t st, double* mats, int64_t mat_stride, int batch, int n64,
                          int rhs_pad, int nrhs, double* dinv, int32_t* info, int64_t* n_launch, int path = 1);
void rg_launch_chol_solve_formed_x(hipStream_t st, const double* sum, int64_t sum_stride, const double* fold,
                                   int64_t fold_stride, int nfold, const double* shift, int nshift,
                                   const int32_t* d_n, int n_fixed, int nouter, double* mats,
                                   int64_t mat_stride, int n64, int rhs_pad, int nrhs, double* dinv,
                                   int32_t* info, int64_t* n_launch, int subtract, const double* extra,
                                   int64_t extra_stride, int extra_row0, int n_div = 1, int b_offset = 0,
                                   int b_count = -1, int path = 1, int embed = 0);
void rg_launch_dgemm_nt(hipStream_t st, const double* A, int64_t lda, const double* B, int64_t ldb,
                        int m, int n, int64_t k, double* C, int64_t ldc);
// pred.hip
struct PredArgs {
  int nblk, nseg, R0, P, C, n128, n64, rtot, B_total;
  int64_t Np, pk_ld, pk_blk_stride;
  SegLayout seg;
  const uint8_t* pk; const double* mu; const double* sc; const double* Bm; const double* wk;
  const double* V; const double* maskp; const uint8_t* keptp; const int32_t* bs;
  const int32_t* blockid; const double* neff; const int32_t* nmiss;
  double *beta, *cb, *psum, *W;
  // exact i8 route of the many-row predictions (pred_i8.hip); null = not available for this problem
  int8_t* bplanes = nullptr; double* bsc = nullptr; uint8_t* pkT = nullptr;
  int embed = 0;   // 1: the solutions sit in rows bs .. bs + P - 1 of the factored systems instead of rows n64 ..
};
struct ChunkTab { const int32_t* seg; const int64_t* pos; const int64_t* len; int n; };
void rg_launch_l0_pred_impl(hipStream_t st, const PredArgs& a, const ChunkTab& c1k, const ChunkTab& c256, double* stats);
// xy_i8.hip
void rg_launch_v_split(hipStream_t st, const double* V, int64_t Np, int Cv, int8_t* vd, double* vsc);
void rg_launch_xy_i8(hipStream_t st, const uint8_t* pk, int64_t pk_ld, int64_t pk_blk_stride, const int32_t* d_bs, const int32_t* nmiss,
                     int nblk, int n128, const SegLayout& seg, const int8_t* vd, const double* vsc, int64_t Np, int Cv, int32_t* S32,
                     double* part);
#define RG_XY_LUT_DOSAGE 0x00010002u   // byte k = value of .bed code k: 00 -> 2, 01 (missing) -> 0, 10 -> 1, 11 -> 0
#define RG_XY_LUT_SQUARE 0x00010004u   // the square of the allele count
void rg_launch_xy_i8_sums(hipStream_t st, const uint8_t* pk, int64_t pk_ld, const int32_t* d_bs, const int32_t* nmiss, int ncols, int n128,
                          const SegLayout& seg, const int8_t* vd, int64_t Np, unsigned lut0, int32_t* S32);
void rg_launch_xy_i8_both(hipStream_t st, const uint8_t* pk, int64_t pk_ld, const int32_t* d_bs, int ncols, int n128, const SegLayout& seg,
                          const int8_t* vd, int64_t Np, unsigned lut0, int32_t* S32);
void rg_launch_xy_i8_planes(hipStream_t st, const int8_t* aplanes, int64_t a_set_stride, int nset, const int32_t* d_bs, int ncols, int n128,
                            const SegLayout& seg, const int8_t* vd, int64_t Np, int32_t* S32);
void rg_launch_l0_pred_i8(hipStream_t st, const PredArgs& a, const ChunkTab& c256, int pg, int ngrp, int8_t* planes, double* psc,
                          uint8_t* pkT);
void rg_launch_w_gather(hipStream_t st, const double* W, int64_t Np, int P, int p, int col0, int R0,
                        const int64_t* posc, int64_t N, double* out);
void rg_launch_w_scatter(hipStream_t st, double* W, int64_t Np, int P, int p, int col0, int R0,
                         const int64_t* posc, int64_t N, const double* in);
// loocv.hip
struct LoocvArgs {
  int nblk, R0, P, C, n128, n64, rtot, row_g0;
  int64_t Np, pk_ld, pk_blk_stride;
  const uint8_t* pk; const double* mu; const double* sc; const double* Bm; const double* V;
  const double* maskp; const double* neff; const int32_t* bs; const int32_t* blockid;
  const double* wk; double* gt; double* W;
  int64_t gt_pos0 = 0, gt_len = 0;   // the chunk of sample positions k_decode_gt fills (gt: [nblk][gt_len][n64])
};
// loocv_tri.hip: the leave-one-out predictors of a batch of assembled blocks through one tridiagonal reduction per block
size_t rg_loocv_tri_ws_doubles(int nblk, int n64, int P, int R0);
int rg_l0_loocv_tri(rg_ctx* ctx, hipStream_t st, const LoocvArgs& la, int max_bs, const double* d_sum, int rtot, double* ws, double* zt,
                    int64_t chunk, const std::function<void(int64_t, int64_t)>& gt_chunk);
void rg_launch_decode_gt(hipStream_t st, const LoocvArgs& a);
void rg_launch_l0_loocv(hipStream_t st, const LoocvArgs& a, double* part0, double* part1, int nchunk);
// l1.hip: fold Grams of a row-major matrix with the level-1 Gram kernel (64x64 tile per wave)
void rg_launch_fold_gram_rows(hipStream_t st, const double* G, int64_t ld, int L, int n64, int rtot, const double* zero,
                              const SegLayout& seg, double* out);
// l0_f64.hip: level 0 on non-integer genotypes (dosages)
int rg_l0_blocks_f64_impl(rg_ctx* ctx, int nblk, const int32_t* block_ids, const int32_t* bs, const double* const* rows,
                          int64_t row_stride, int mem_kind);
// l1.hip
// copies the per-chromosome predictions of one phenotype (device, [nchr][N]) to the caller: as they are, or assembled
// into LOCO rows [nchrom][N] first (rg_set_loco_output).  Returns an RG_* code.
int rg_emit_pred(rg_ctx* ctx, hipStream_t st, const double* d_pred, int nchr, int p, double* pred_out);
struct L1Args;
int rg_l1_qt_impl(rg_ctx* ctx, int R1, const double* tau, int nchr, const int32_t* cols_per_chr,
                  double* cumsum_out, int32_t* best_out, double* pred_out);
// wgram_bf16.hip: partial tiles of the quasi-Newton weighted Gram (one fp16 operand plane, or bf16 hi + lo planes, on the matrix cores); returns the K slices written (0 = failed)
int rg_launch_wgram_bf16(rg_ctx* ctx, hipStream_t st, const double* W, int64_t Np, int L, int P, int p, int n64, const double* wv, double* sw, int nchain,
                         const int32_t* d_chainmap, const int32_t* h_chainmap, int nslot, int excl_own, double* part, int64_t out_stride,
                         int max_slices);
// l1x.hip
int rg_l1_qt_loocv_impl(rg_ctx* ctx, int R1, const double* tau, int nchr, const int32_t* cols_per_chr,
                        double* cumsum_out, int32_t* best_out, double* pred_out);
int rg_l1_cox_impl(rg_ctx* ctx, int pheno, int R1, const double* time, const double* event, const double* offset, const rg_cox_options* opt,
                   int nchr, const int32_t* cols_per_chr, double* tau_out, double* deviance_out, int32_t* converged_out, int32_t* best_out,
                   double* pred_out);
int rg_l1_bt_impl(rg_ctx* ctx, int R1, const double* tau, const double* yraw, const double* offset,
                  const rg_bt_options* opt, int nchr, const int32_t* cols_per_chr, double* cumsum_out,
                  int32_t* converged_out, int32_t* best_out, double* pred_out);
