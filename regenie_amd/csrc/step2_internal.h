// Internal state of the Step-2 context shared by step2_qt.hip (quantitative traits, the contraction primitive) and step2_bt.hip (binary /
// count traits: score test, approximate Firth and saddlepoint corrections).  Product code: nothing here may reference oracle/.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <vector>

#include "../../include/rg_step2.h"
#include "rg_internal.h"

struct BtState;
struct rg_s2_ctx {
  int dev = 0;
  int64_t n = 0;
  int C = 0, P = 0;
  bool have_null = false;
  hipStream_t st = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  double *dX = nullptr, *dY = nullptr, *dscf = nullptr;
  uint8_t* dM = nullptr;
  void* buf[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  size_t cap[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int vpb = 4, ept = 4;         // tile of the two streaming kernels (resolved in rg_s2_create)
  double last_ms = 0.0;
  // hard-call route (rg_s2_qt_block_packed): built lazily after rg_s2_set_null
  bool complete = false;        // every mask byte is 1
  bool static_ready = false;    // planes of the columns that depend on X and the masks only
  bool planes_mask_cols = false;   // the planes include the x_c mask_p / mask_p columns
  bool res_ready = false;       // planes of the res columns, res^T X
  std::vector<double> hX;       // host copies of the last X / mask (the planes are kept while they do not change)
  std::vector<uint8_t> hM;
  int64_t Np = 0;               // samples padded to a multiple of 128 * 32
  int Cvt = 0, cm0 = 0;         // columns in all, first mask column (a multiple of 16); complete problems: Cvt = cm0 = C + P
  int64_t rule_n = 0;           // check_sparse_G: params.n_samples (0 = the analysed samples)
  double rule_thr = 0.5;        // params.prop_zero_thr
  int rule_zero_count = 0;      // 1: the .pgen form of the rule (observed zeros >= n_samples * thr)
  double* dV = nullptr;         // [Cvt (padded to 16)][Np]  X | res | x_c mask_p | 0 | mask_p, zero padded
  int8_t* dvd = nullptr;        // the digit planes of dV's columns, [col][8][Np]
  double *dvsc = nullptr, *dYtX = nullptr;   // [col] plane scales, [P][C] res_p^T x_c
  double *dQ = nullptr, *dMsum = nullptr;    // [P][C][C] X^T diag(mask_p) X, [P] sum of mask_p
  // hard-call route, masked problems whose lists of masked samples sum to at most n entries (round 6): the contraction against the mask
  // columns x_c mask_p / mask_p (C P + P columns over all n samples) is taken as (all samples) - (the samples masked for p), the second
  // term over a COMPACT sample axis: the masked samples of phenotype 0, padded, then those of phenotype 1, ... -- at 5 % missing values in
  // each of 10 phenotypes that is n / 2 positions against ONE group of C + 1 columns instead of n positions against 8 groups.
  bool compact = false;
  int64_t Ncp = 0;              // compact positions in all (a multiple of 128)
  int c_spt = 1, c_chunk = 1;   // segments per phenotype, phenotypes per contraction launch (c_spt * c_chunk <= RG_MAX_SEG)
  std::vector<int64_t> c_off;   // [P + 1] first compact position of a phenotype (its range: a multiple of 128 * c_spt positions)
  int32_t* d_clist = nullptr;   // [Ncp] analysed-sample index of the position, -1 = padding
  int32_t* d_cmeta = nullptr;   // [c_K][P][2]: first 32-bit word and words of the (sample chunk, phenotype) range (k_s2_compact_rows)
  int c_K = 0;                  // sample chunks of S2_CS samples
  int32_t* d_cw0 = nullptr;     // [P + 1] first 32-bit word of a phenotype's range in the compact row (k_s2_count_traits)
  double* dVc = nullptr;        // [C padded to 16][Ncp]: x_0 .. x_{C-1} of the listed samples
  int8_t* dvdc = nullptr;       // their digit planes
  double* dvscc = nullptr;
  // generic contraction (rg_s2_set_columns / rg_s2_contract_packed): caller-defined columns
  int g_ncol = 0, g_nsq = 0;
  double* gV = nullptr;         // [ncol padded to 16][Np]
  int8_t* gvd = nullptr;
  double* gvsc = nullptr;
  // dosage route (rg_s2_qt_block), masked problems: per phenotype the samples masked for it
  bool lists_ready = false;
  int32_t* d_mlist = nullptr;
  int64_t* d_moff = nullptr;    // [P + 1]
  double* d_xl = nullptr;       // [list entries][C]: the listed samples' covariate rows
  double* d_xq = nullptr;       // [P][C][C]: X^T X over the samples listed for each phenotype
  void* pbuf[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  size_t pcap[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int32_t hdr[3] = {0, 0, 0};   // staged {total_miss = 0, bs, 0} of the block in flight
  std::string err;
  // ---- binary / count traits behind the ABI (step2_bt.hip): null model per chromosome, the block last scored, correction buffers ----
  struct BtState* bt = nullptr;
};


int rg_xy_i8_launch_groups(int ncols);      // xy_i8.hip: groups of workgroups rg_launch_xy_i8_sums / _both launch for that many columns (1 when two groups go in one pass)

static inline int rg_s2_fail(rg_s2_ctx* ctx, int code, const std::string& msg) {
  if (ctx) ctx->err = msg;
  return code;
}
#define S2_HIP(call)                                                                                               \
  do {                                                                                                             \
    hipError_t e_ = (call);                                                                                        \
    if (e_ != hipSuccess) return rg_s2_fail(ctx, RG_S2_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
  } while (0)

static inline int rg_s2_ensure_in(rg_s2_ctx* ctx, void** buf, size_t* cap, int slot, size_t bytes) {
  if (cap[slot] >= bytes) return RG_S2_OK;
  if (buf[slot]) S2_HIP(hipFree(buf[slot]));
  buf[slot] = nullptr;
  cap[slot] = 0;
  S2_HIP(hipMalloc(&buf[slot], bytes));
  cap[slot] = bytes;
  return RG_S2_OK;
}
// slots of rg_s2_ctx::pbuf used by the contraction entries (the block's staged rows stay there for the corrections of step2_bt.hip)
enum { RG_S2_Q_PK = 0, RG_S2_Q_CNT = 1, RG_S2_Q_S = 2, RG_S2_Q_A = 3, RG_S2_Q_VAR = 4 };
// slot of rg_s2_ctx::buf holding the uint16 dosage rows of the block handed to the integer-dosage entries from the host
enum { RG_S2_B_G = 0 };
void rg_s2_bt_free(rg_s2_ctx* ctx);   // step2_bt.hip
