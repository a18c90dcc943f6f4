// Level-0 out-of-fold predictions for many (phenotype, ridge value) rows on the i8 matrix cores, exactly.
//
// out[m][pos] = sum_j beta~_m[j] * g~_j(pos)  is the (R0*P) x bs x N contraction of ridge_level_0
// (reference src/Step1_Models.cpp:496-511: pred = beta^T G_fold), 5*10^10 multiply-adds per block at
// 500,000 samples and 10 phenotypes -- the largest kernel of BASELINE configs[2] on the fp64 matrix cores
// (k_l0_pred_mfma, pred.hip: 96 of 236 ms per step and GPU).  The genotype operand is an exact small integer
// (dosage 0/1/2, missing indicator 0/1), so the fp64 operand can be taken apart instead of rounded:
//
//   * every coefficient row is written in fixed point against its own largest entry, 2^e > max_j |beta~_m[j]|:
//     q_j = rint(beta~_m[j] * 2^(54-e)), |q_j| < 2^54, and q_j is split into eight balanced base-128 digits
//     d_k[j] in [-64, 63] (k_beta_split) -- int8 planes;
//   * S_k[m][pos] = sum_j d_k[m][j] * g_j(pos) runs on v_mfma_i32_32x32x32_i8 with int32 accumulation, EXACT
//     (|S_k| <= 64 * 2 * bs < 2^31);
//   * out = 2^(e-54) * sum_k 128^k S_k: digit pairs are combined in int32 (128 S_{k+1} + S_k, still exact), the four pairs by
//     Horner in fp64 (every product with a power of two is exact; the sum rounds at 2^-53 of its value).
//   The only approximation is the truncation of beta~ at 2^-54 of the row's largest coefficient: an absolute error
//   below bs * 2 * 2^-55 * max|beta~| per prediction, the size of the rounding an fp64 dot product of bs terms
//   carries anyway.  No tolerance changes anywhere: the parity tests stay at 1e-8.
//   Missing calls are mean-imputed by the same route: a second plane set holds the digits of beta~_m[j] * mu_j and is
//   contracted with the missing indicator (only for blocks that have missing calls).
//
// Layout: the contraction index is the SNP, so the kernel reads a SNP-contiguous copy of the block's cleaned 2-bit
// rows (k_pk_transpose: 16 SNPs per dword, stored in the order the lanes of a wave read them) -- one dword per lane and MFMA, expanded to
// sixteen int8 with v_perm_b32 as a byte LUT (the idiom of gram_i8.hip).  A wave owns 32 positions and keeps their expanded
// genotype operand for 512 SNPs at a time (16 x 16 bytes per lane) in registers across the 8 digit planes; the coefficient
// digits of one (row tile, plane, half) are staged in LDS per workgroup (32 rows x 512 bytes) and shared by its eight waves.  Epilogue as in pred.hip: covariate term, mask, store, per-row sums in a fixed order.
#include <algorithm>
#include <cstdlib>
#include "rg_internal.h"

#define PI8_NPIECE 8
#define PI8_ROWS 64                 // rows (phenotype, ridge value) of a group, two MFMA row tiles
#define PI8_KMAX 1024               // SNPs per block served (n128 <= 1024); larger blocks keep the fp64 kernel
#define PI8_KHALF 512                // SNPs whose expanded genotype operand a lane holds in registers at a time
#define PI8_PITCH (PI8_KHALF + 16)  // LDS row pitch of the staged digit plane: 16 consecutive rows on distinct bank groups
#define LUT_DOSAGE 0x00010002u      // cleaned 2-bit code -> dosage (00 -> 2, 01 -> missing = 0 here, 10 -> 1, 11 -> 0)
#define LUT_MISS 0x00000100u        //                     -> missing indicator

// ---- SNP-contiguous copy of the cleaned packed rows: pkT[blk][pos][j / 4], two bits per SNP -----------------------------
// grid (Np / 128, n128 / 128, nblk); a 128 SNP x 128 position tile through LDS: 32-byte pieces on both sides
__global__ __launch_bounds__(256) void k_pk_transpose(const uint8_t* __restrict__ pk, int64_t pk_ld, int64_t pk_blk_stride, int n128,
                                                      int64_t Np, uint8_t* __restrict__ pkT) {
  __shared__ __attribute__((aligned(16))) uint8_t s[128][32 + 16];
  const int blk = blockIdx.z, j0 = blockIdx.y * 128;
  const int64_t pos0 = (int64_t)blockIdx.x * 128;
  const uint8_t* src = pk + (int64_t)blk * pk_blk_stride + (int64_t)j0 * pk_ld + pos0 / 4;
  {
    const int j = threadIdx.x >> 1, h = threadIdx.x & 1;               // 128 rows x 32 bytes, 16 bytes per thread
    *reinterpret_cast<uint4*>(&s[j][16 * h]) = *reinterpret_cast<const uint4*>(src + (int64_t)j * pk_ld + 16 * h);
  }
  __syncthreads();
  const int pl = threadIdx.x >> 1, hh = threadIdx.x & 1;               // output: 128 positions x 32 bytes, 16 bytes (64 SNPs) per thread
  const int sh = 2 * (pl & 3), pb = pl >> 2;
  uint32_t o[4];
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    uint32_t w = 0;
#pragma unroll
    for (int v = 0; v < 16; ++v) w |= (uint32_t)((s[64 * hh + 16 * d + v][pb] >> sh) & 3u) << (2 * v);
    o[d] = w;
  }
  // lane-ordered layout for the prediction kernel: dword (pos, SNP group of 16 = (t, kb)) at
  //   (((blk * Np/32 + pos/32) * (n128/32) + t) * 2 + kb) * 32 + pos%32   -- a wave's operand load for one K step is 256 contiguous bytes
  uint32_t* out32 = reinterpret_cast<uint32_t*>(pkT);
  const int64_t pg32 = (pos0 + pl) >> 5;
  const int cc = (int)((pos0 + pl) & 31);
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    const int sg = (j0 + 64 * hh + 16 * d) >> 4;          // SNP group of 16: t = sg >> 1, kb = sg & 1
    out32[((((int64_t)blk * (Np >> 5) + pg32) * (n128 >> 5) + (sg >> 1)) * 2 + (sg & 1)) * 32 + cc] = o[d];
  }
}

// ---- fixed-point digit planes of the coefficient rows ------------------------------------------------------------------
// grid (PI8_ROWS, nseg * ngrp, nblk), 256 threads: one row m of one (block, fold, phenotype group).
// planes [blk][s][grp][set][k][m][n128] int8 (set 0: beta~, set 1: beta~ * mu), psc [blk][s][grp][set][m] = 2^(e-54).
__global__ __launch_bounds__(256) void k_beta_split(PredArgs a, int pg, int ngrp, int8_t* __restrict__ planes, double* __restrict__ psc) {
  __shared__ double red[2][4];
  __shared__ double smax[2];
  const int m = blockIdx.x, s = blockIdx.y / ngrp, grp = blockIdx.y % ngrp, blk = blockIdx.z;
  const int p0 = grp * pg, npg = min(pg, a.P - p0), nrow = npg * a.R0;
  const int R0 = a.R0, nm = a.nseg * R0;
  const int bs = a.bs[blk];
  const bool live = m < nrow;
  const int pl = live ? m / R0 : 0, r = live ? m % R0 : 0;
  const double* be = a.beta + (((int64_t)blk * nm + s * R0 + r) * a.P + p0 + pl) * a.n64;
  const double* mu = a.mu + (int64_t)blk * a.n128;
  const bool has_miss = a.nmiss[blk] > 0;
  double mx0 = 0.0, mx1 = 0.0;
  for (int j = threadIdx.x; j < a.n128; j += 256) {
    const double b = (live && j < bs) ? be[j] : 0.0;
    mx0 = fmax(mx0, fabs(b));
    mx1 = fmax(mx1, fabs(b * mu[min(j, a.n128 - 1)]));
  }
  for (int o = 32; o > 0; o >>= 1) { mx0 = fmax(mx0, __shfl_down(mx0, o)); mx1 = fmax(mx1, __shfl_down(mx1, o)); }
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = mx0; red[1][threadIdx.x >> 6] = mx1; }
  __syncthreads();
  if (threadIdx.x < 2) smax[threadIdx.x] = fmax(fmax(red[threadIdx.x][0], red[threadIdx.x][1]), fmax(red[threadIdx.x][2], red[threadIdx.x][3]));
  __syncthreads();
  const int64_t grp_idx = ((int64_t)blk * a.nseg + s) * ngrp + grp;
  for (int set = 0; set < 2; ++set) {
    if (set == 1 && !has_miss) break;
    const double mx = smax[set];
    int e = 0;
    if (mx > 0.0) { (void)frexp(mx, &e); }              // mx = f * 2^e, 0.5 <= f < 1  =>  |b| < 2^e
    const double up = mx > 0.0 ? ldexp(1.0, 54 - e) : 0.0;
    if (threadIdx.x == 0) psc[(grp_idx * 2 + set) * PI8_ROWS + m] = mx > 0.0 ? ldexp(1.0, e - 54) : 0.0;
    int8_t* pl0 = planes + ((grp_idx * 2 + set) * PI8_NPIECE) * (int64_t)PI8_ROWS * a.n128 + (int64_t)m * a.n128;
    for (int j = threadIdx.x; j < a.n128; j += 256) {
      double b = (live && j < bs) ? be[j] : 0.0;
      if (set == 1) b *= mu[j];
      long long q = llrint(b * up);                     // |q| <= 2^54
#pragma unroll
      for (int k = 0; k < PI8_NPIECE; ++k) {
        const int d = (int)(((q & 127) ^ 64) - 64);     // balanced digit in [-64, 63]
        pl0[(int64_t)k * PI8_ROWS * a.n128 + j] = (int8_t)d;
        q = (q - d) >> 7;
      }
    }
  }
}

__device__ __forceinline__ unsigned pi8_expand4(unsigned b, unsigned lut) {
  unsigned x = b | (b << 6);
  x = x | (x << 12);
  x &= 0x03030303u;
  return __builtin_amdgcn_perm(lut, lut, x);
}

// ---- epilogue of one row tile (32 rows x the workgroup's 256 positions): covariate term, mask, store, per-row sums ------------
// out[r]: row tile*32 + (r&3) + 8(r>>2) + 4kb at position pos, for the lane (c, kb) of wave `wave`.
// Every global load of the lane is issued before its first store (a load that follows a store through pointers that may alias
// waits for it: 16 load -> store round trips per tile were three quarters of this kernel's time), and the sums over the 32
// positions of a wave go through LDS -- tmp[wave][row][position], one thread per (wave, row, moment) adding 32 values in a fixed
// order -- instead of 160 dependent cross-lane shuffles per lane.  tmp: 8 x 32 x 33 doubles (67.6 KB) of the staging area, free
// once the contraction is done (the caller has passed a barrier).
#define PI8_TMP_PITCH 33
#define PI8_CPRE 4      // covariate values of the lane's position held in registers across the contraction (more columns: loaded in the epilogue)
// Round 5: the epilogue no longer waits for memory.  With one workgroup per CU (132 KB of staged planes) nothing hides a load, and the
// epilogue of a tile used to pay four exposed round trips -- the rows' covariate products (staged through LDS behind a barrier), the
// covariate values of the position one after the other, the masks -- about as long as the tile's 512 MFMAs.  Now the covariate products of
// BOTH tiles are staged once at kernel start (scb), the lane's covariate values (xv) and the masks of its 2 x 16 rows (one bit each) are
// loaded before the first MFMA and ride along in 18 registers.
__device__ __forceinline__ void pi8_epilogue(const PredArgs& a, int tile, int64_t pos, int wave, int c, int kb, const double (&out)[16],
                                             const double (*scb)[PI8_CPRE + 1], const int* srow_w, const double (&xv)[PI8_CPRE], unsigned mkbits,
                                             double* tmp, double (*sred)[PI8_ROWS][2]) {
  double corr[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) corr[r] = 0.0;
  const int npre = min(a.C, PI8_CPRE);
#pragma unroll
  for (int cc = 0; cc < PI8_CPRE; ++cc) {
    if (cc < npre) {
#pragma unroll
      for (int r = 0; r < 16; ++r) corr[r] = fma(scb[tile * 32 + (r & 3) + 8 * (r >> 2) + 4 * kb][cc], xv[cc], corr[r]);
    }
  }
  int wrow[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) wrow[r] = srow_w[tile * 32 + (r & 3) + 8 * (r >> 2) + 4 * kb];
  double* trow = tmp + ((int64_t)wave * 32) * PI8_TMP_PITCH + c;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const bool live = wrow[r] >= 0;
    const double v = (live && ((mkbits >> r) & 1u)) ? (out[r] - corr[r]) : 0.0;
    if (live) a.W[(int64_t)wrow[r] * a.Np + pos] = v;
    trow[((r & 3) + 8 * (r >> 2) + 4 * kb) * PI8_TMP_PITCH] = v;
  }
  __syncthreads();
  {
    const int w = threadIdx.x >> 6, ml = (threadIdx.x >> 1) & 31, q = threadIdx.x & 1;
    const double* src = tmp + ((int64_t)w * 32 + ml) * PI8_TMP_PITCH;
    double t = 0.0;
#pragma unroll
    for (int pp = 0; pp < 32; ++pp) {
      const double x = src[pp];
      t += q ? x * x : x;
    }
    sred[w][tile * 32 + ml][q] = t;
  }
}
// covariate columns past PI8_CPRE (rare: more than eight covariates): the remaining products, loaded in the epilogue as before
__device__ __forceinline__ void pi8_corr_tail(const PredArgs& a, int tile, int nrow, int blk, int s, int p0, int64_t pos, int kb, double (&out)[16]) {
  const int R0 = a.R0, nm = a.nseg * R0;
  for (int cc = PI8_CPRE; cc < a.C; ++cc) {
    const double xvv = a.V[(int64_t)cc * a.Np + pos];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = tile * 32 + (r & 3) + 8 * (r >> 2) + 4 * kb;
      const int mc = m < nrow ? m : 0;
      const double cb = a.cb[(((int64_t)blk * nm + s * R0 + mc % R0) * a.P + p0 + mc / R0) * a.C + cc];
      out[r] = fma(-(m < nrow ? cb : 0.0), xvv, out[r]);
    }
  }
}

// ---- the contraction + epilogue ------------------------------------------------------------------------------------------
// grid (n_c256, ngrp, nblk), 512 threads = 8 waves x 32 positions (a chunk is 256 positions of one fold).
// LDS: the eight digit planes of one (row tile, set, half): 8 x 32 rows x 512 bytes (+ pad) = 132 KB, staged once per round, so a
// round is one pair of barriers and 8 x 16 back-to-back MFMAs per wave.
// FULL: every half is 512 SNPs wide (n128 a multiple of 512: the block widths of the BASELINE configurations) -- trip counts
// are then compile-time constants and the loops carry no predicates (a predicate per MFMA put every MFMA into its own basic
// block, which kept the scheduler from moving the LDS reads of the next step above it).
#define PI8_PLANE (32 * PI8_PITCH)
template <bool FULL>
__global__ __launch_bounds__(512) void k_l0_pred_i8(PredArgs a, ChunkTab ct, int pg, int ngrp, const int8_t* __restrict__ planes,
                                                    const double* __restrict__ psc, const uint8_t* __restrict__ pkT) {
  extern __shared__ __attribute__((aligned(16))) int8_t smem[];
  int8_t* sA = smem;                                                       // [8 planes][32 rows][PI8_PITCH]
  double (*sred)[PI8_ROWS][2] = reinterpret_cast<double (*)[PI8_ROWS][2]>(smem + PI8_NPIECE * PI8_PLANE);   // [8 waves][64][2]
  __shared__ double scb[PI8_ROWS][PI8_CPRE + 1];       // cb of the group's rows (both tiles), the first PI8_CPRE covariates
  __shared__ int srow_w[PI8_ROWS], srow_p[PI8_ROWS];   // W row (col0 + r) * P + p and phenotype of every row m (-1: dead)
  const int blk = blockIdx.z, ch = blockIdx.x, grp = blockIdx.y, p0 = grp * pg;
  const int npg = min(pg, a.P - p0), nrow = npg * a.R0;
  const int s = ct.seg[ch];
  const int64_t pos0 = ct.pos[ch];
  const int R0 = a.R0, nm = a.nseg * R0;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 31, kb = lane >> 5;
  const int64_t pos = pos0 + wave * 32 + c;
  const bool has_miss = a.nmiss[blk] > 0;
  const int n128 = a.n128, nstep = n128 / 32;          // MFMA K steps of 32 SNPs
  const int64_t grp_idx = ((int64_t)blk * a.nseg + s) * ngrp + grp;
  // this lane's dword of K step t: ((pos group * nstep + t) * 2 + kb) * 32 + c  (k_pk_transpose)
  const uint32_t* brow = reinterpret_cast<const uint32_t*>(pkT) + (((int64_t)blk * (a.Np >> 5) + (pos >> 5)) * nstep) * 64 + kb * 32 + c;
  const int col0 = a.blockid[blk] * R0;
  if (threadIdx.x < PI8_ROWS) {
    const int m = threadIdx.x;
    const bool live = m < nrow;
    const int pl = live ? m / R0 : 0, rr = live ? m % R0 : 0;
    srow_w[m] = live ? (col0 + rr) * a.P + p0 + pl : -1;
    srow_p[m] = p0 + pl;
  }
  for (int e = threadIdx.x; e < PI8_ROWS * PI8_CPRE; e += 512) {      // the rows' covariate products, both tiles, once
    const int m = e / PI8_CPRE, cc = e % PI8_CPRE;
    const int mc = m < nrow ? m : 0;
    const double* cb = a.cb + (((int64_t)blk * nm + s * R0 + mc % R0) * a.P + p0 + mc / R0) * a.C;
    scb[m][cc] = (cc < a.C && m < nrow) ? cb[min(cc, a.C - 1)] : 0.0;
  }
  // what the epilogues need from memory, requested before the first MFMA: the position's covariate values and, one bit per row, the masks
  double xv[PI8_CPRE];
#pragma unroll
  for (int cc = 0; cc < PI8_CPRE; ++cc) xv[cc] = a.V[(int64_t)min(cc, a.C - 1) * a.Np + pos];
  unsigned mkbits[2] = {0u, 0u};
  {
    double mk[2][16];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * kb;
        const int pl = m < nrow ? m / R0 : 0;
        mk[t][r] = a.maskp[(int64_t)(p0 + pl) * a.Np + pos];
      }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) mkbits[t] |= (mk[t][r] != 0.0 ? 1u : 0u) << r;
  }
  // ---- stages: (tile, set, half) in this order; the operands of stage i + 1 are REQUESTED before the MFMAs of stage i ----
  // One workgroup per CU (132 KB of staged planes): nothing else hides a load.  Measured with cycle counters per phase (round 5): the
  // genotype dwords (a load followed at once by its expansion) were 32 % of a workgroup's time, the plane pieces (load -> LDS store)
  // 20 %, the 128 MFMAs per wave and stage 17 %.  The raw genotype dwords of the next stage (16 registers) now travel while the matrix
  // cores work: loaded right after the barrier that releases the MFMA loop, expanded at the top of the next stage.  (The 64 registers
  // of plane pieces do not fit beside them: the compiler parks them in scratch, which waits for the loads before the MFMAs start.)
  const int ntile = nrow > 32 ? 2 : 1, nset = has_miss ? 2 : 1, nhalf = (n128 + PI8_KHALF - 1) / PI8_KHALF;
  const int nstage = ntile * nset * nhalf;
  uint32_t gw[PI8_KHALF / 32];       // raw genotype dwords of the stage about to be expanded
  auto stage_nst = [&](int half) { return FULL ? PI8_KHALF / 32 : min(PI8_KHALF / 32, nstep - half * (PI8_KHALF / 32)); };
  auto request = [&](int st) {       // issue the genotype loads of stage st
    const int half = st % nhalf;
    const int nst = stage_nst(half);
#pragma unroll
    for (int t = 0; t < PI8_KHALF / 32; ++t) {
      const int tc = half * (PI8_KHALF / 32) + ((FULL || t < nst) ? t : nst - 1);
      gw[t] = brow[(int64_t)tc * 64];
    }
  };
  request(0);
  double out[16], oi[16];
#pragma unroll 1
  for (int st = 0; st < nstage; ++st) {
    const int half = st % nhalf, set = (st / nhalf) % nset, tile = st / (nhalf * nset);
    const int nst = stage_nst(half);
    const unsigned lut = set == 0 ? LUT_DOSAGE : LUT_MISS;
    if (half == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) oi[r] = 0.0;
    }
    __syncthreads();                 // the previous stage's MFMAs (and a tile's epilogue) are done with the staging area
    {  // stage 8 planes x 32 rows x (this half's) digits: 16-byte pieces, coalesced along the SNP index; all loads of a thread are
       // issued before its first LDS store
      const int8_t* src = planes + ((grp_idx * 2 + set) * PI8_NPIECE) * (int64_t)PI8_ROWS * n128 + (int64_t)tile * 32 * n128 + half * PI8_KHALF;
      const int per_row = nst * 2;
      uint4 pv[PI8_KHALF / 32];
#pragma unroll
      for (int it = 0; it < PI8_KHALF / 32; ++it) {
        const int e = threadIdx.x + 512 * it;
        const int rowk = e / per_row, pc = e - rowk * per_row;
        const int k = rowk >> 5, row = rowk & 31;
        const bool on = FULL || it < nst;
        pv[it] = *reinterpret_cast<const uint4*>(src + (int64_t)(on ? k : 0) * PI8_ROWS * n128 + (int64_t)(on ? row : 0) * n128 + (on ? pc : 0) * 16);
      }
#pragma unroll
      for (int it = 0; it < PI8_KHALF / 32; ++it) {
        const int e = threadIdx.x + 512 * it;
        const int rowk = e / per_row, pc = e - rowk * per_row;
        const int k = rowk >> 5, row = rowk & 31;
        if (FULL || it < nst) *reinterpret_cast<uint4*>(sA + k * PI8_PLANE + row * PI8_PITCH + pc * 16) = pv[it];
      }
    }
    v4i bf[PI8_KHALF / 32];
#pragma unroll
    for (int t = 0; t < PI8_KHALF / 32; ++t) {
      const uint32_t w = gw[t];
      bf[t] = (v4i){(int)pi8_expand4(w & 0xFFu, lut), (int)pi8_expand4((w >> 8) & 0xFFu, lut),
                    (int)pi8_expand4((w >> 16) & 0xFFu, lut), (int)pi8_expand4(w >> 24, lut)};
    }
    __syncthreads();
    if (st + 1 < nstage) request(st + 1);
    // digit planes from the most significant pair down: a pair is combined in int32 (|S_k| <= 64 * 2 * 512 = 2^16 per half, so
    // 128 S_{k+1} + S_k < 2^24) -- the second chain simply accumulates onto the shifted sums -- and the pairs by Horner in fp64
    // (each pair's exact int32 sum enters the fp64 total times its power of two, 16384^kp: every product is exact, the running sum rounds
    // at 2^-53 of its value -- the Horner form this replaces kept sixteen more doubles alive per lane)
#pragma unroll 1
    for (int kp = PI8_NPIECE / 2 - 1; kp >= 0; --kp) {
      v16i acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0;
#pragma unroll
      for (int hl = 1; hl >= 0; --hl) {
        const int8_t* arow = sA + (2 * kp + hl) * PI8_PLANE + c * PI8_PITCH + 16 * kb;
#pragma unroll
        for (int t = 0; t < PI8_KHALF / 32; ++t) {
          if (FULL || t < nst) {
            const v4i af = *reinterpret_cast<const v4i*>(arow + 32 * t);
            acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(af, bf[t], acc, 0, 0, 0);
          }
        }
        if (hl == 1) {
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[r] *= 128;
        }
      }
      const double w14 = kp == 3 ? 4398046511104.0 : (kp == 2 ? 268435456.0 : (kp == 1 ? 16384.0 : 1.0));     // 16384^kp
#pragma unroll
      for (int r = 0; r < 16; ++r) oi[r] = fma((double)acc[r], w14, oi[r]);
    }
    if (half == nhalf - 1) {
      // the rows' 2^(e-54): a power of two times an integer -- exact up to the one rounding of the sum above
      const double* scrow = psc + (grp_idx * 2 + set) * PI8_ROWS + tile * 32 + 4 * kb;
#pragma unroll
      for (int r = 0; r < 16; ++r) out[r] = set == 0 ? oi[r] * scrow[(r & 3) + 8 * (r >> 2)] : fma(oi[r], scrow[(r & 3) + 8 * (r >> 2)], out[r]);
      if (set == nset - 1) {
        // ---- epilogue of the tile (pi8_epilogue): the staged planes are no longer needed, their LDS carries the per-wave sums ----
        __syncthreads();
        if (a.C > PI8_CPRE) pi8_corr_tail(a, tile, nrow, blk, s, p0, pos, kb, out);
        pi8_epilogue(a, tile, pos, wave, c, kb, out, scb, srow_w, xv, tile == 0 ? mkbits[0] : mkbits[1], reinterpret_cast<double*>(sA), sred);
      }
    }
  }
  __syncthreads();
  if (threadIdx.x < nrow * 2) {
    const int m = threadIdx.x >> 1, q = threadIdx.x & 1;
    const int pl = m / R0, rr = m % R0;
    double tsum = 0.0;
#pragma unroll
    for (int w = 0; w < 8; ++w) tsum += sred[w][m][q];
    a.psum[((((int64_t)blk * ct.n + ch) * a.P + p0 + pl) * 8 + rr) * 2 + q] = tsum;
  }
}

// ---- the same contraction with asynchronous staging (round 5) ---------------------------------------------------------------------------
// Per-phase cycle counters of the kernel above: genotype loads + expansion 32 %, plane staging (global -> registers -> LDS) 20 %, MFMAs 17 %,
// barriers 15 % -- the phases of the one workgroup a CU holds run back to back.  Here the K loop is outermost: a stage is 128 SNPs of ALL
// eight digit planes of a 32-row tile (32 KB), copied straight from memory into one of two LDS buffers (global_load_lds, 16 B per lane,
// issued through inline assembly before the MFMAs of the stage in flight: wave w copies plane w, four instructions of 8 rows x 128 B;
// the eight 16-byte slots of a row are XOR-swizzled by (row >> 1) & 7 so that a ds_read_b128 lane group falls on 16 distinct bank groups);
// every plane has its own int32 accumulator (|S_k| <= 64 * 2 * 1,024 = 2^17 over the whole block: no overflow, no pair trick), the genotype
// dwords of the next stage are requested a stage ahead and expanded right before their four MFMA groups (3.5 vector instructions per MFMA),
// and the planes meet in fp64 once per tile: out = 2^(e-54) sum_k 128^k S_k (every term exact).  One barrier per stage.
#define PV2_STAGE 32768        // bytes of one staged K step group: 8 planes x 32 rows x 128 SNPs
__global__ __launch_bounds__(512) void k_l0_pred_i8v2(PredArgs a, ChunkTab ct, int pg, int ngrp, const int8_t* __restrict__ planes,
                                                     const double* __restrict__ psc, const uint8_t* __restrict__ pkT) {
  extern __shared__ __attribute__((aligned(16))) int8_t smem[];
  int8_t* sA = smem;                                                       // two stage buffers; the epilogue's tmp afterwards
  double (*sred)[PI8_ROWS][2] = reinterpret_cast<double (*)[PI8_ROWS][2]>(smem + 8 * 32 * PI8_TMP_PITCH * 8);   // [8 waves][64][2], past the tmp area
  __shared__ double scb[PI8_ROWS][PI8_CPRE + 1];
  __shared__ int srow_w[PI8_ROWS], srow_p[PI8_ROWS];
  const int blk = blockIdx.z, ch = blockIdx.x, grp = blockIdx.y, p0 = grp * pg;
  const int npg = min(pg, a.P - p0), nrow = npg * a.R0;
  const int s = ct.seg[ch];
  const int64_t pos0 = ct.pos[ch];
  const int R0 = a.R0, nm = a.nseg * R0;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c = lane & 31, kb = lane >> 5;
  const int64_t pos = pos0 + wave * 32 + c;
  const bool has_miss = a.nmiss[blk] > 0;
  const int n128 = a.n128, nstep = n128 / 32, nstage = n128 / 128;
  const int64_t grp_idx = ((int64_t)blk * a.nseg + s) * ngrp + grp;
  const uint32_t* brow = reinterpret_cast<const uint32_t*>(pkT) + (((int64_t)blk * (a.Np >> 5) + (pos >> 5)) * nstep) * 64 + kb * 32 + c;
  const int col0 = a.blockid[blk] * R0;
  if (threadIdx.x < PI8_ROWS) {
    const int m = threadIdx.x;
    const bool live = m < nrow;
    const int pl = live ? m / R0 : 0, rr = live ? m % R0 : 0;
    srow_w[m] = live ? (col0 + rr) * a.P + p0 + pl : -1;
    srow_p[m] = p0 + pl;
  }
  for (int e = threadIdx.x; e < PI8_ROWS * PI8_CPRE; e += 512) {
    const int m = e / PI8_CPRE, cc = e % PI8_CPRE;
    const int mc = m < nrow ? m : 0;
    const double* cb = a.cb + (((int64_t)blk * nm + s * R0 + mc % R0) * a.P + p0 + mc / R0) * a.C;
    scb[m][cc] = (cc < a.C && m < nrow) ? cb[min(cc, a.C - 1)] : 0.0;
  }
  double xv[PI8_CPRE];
#pragma unroll
  for (int cc = 0; cc < PI8_CPRE; ++cc) xv[cc] = a.V[(int64_t)min(cc, a.C - 1) * a.Np + pos];
  unsigned mkbits[2] = {0u, 0u};
  {
    double mk[2][16];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * kb;
        const int pl = m < nrow ? m / R0 : 0;
        mk[t][r] = a.maskp[(int64_t)(p0 + pl) * a.Np + pos];
      }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) mkbits[t] |= (mk[t][r] != 0.0 ? 1u : 0u) << r;
  }
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) int8_t*)smem;
  const int ntile = nrow > 32 ? 2 : 1, nset = has_miss ? 2 : 1;
  // this lane's part of a stage copy: plane `wave`, rows 8 j + (lane >> 3), physical slot lane & 7 = logical slot ^ ((row >> 1) & 7)
  int roff[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = 8 * j + (lane >> 3), sl = (lane & 7) ^ ((row >> 1) & 7);
    roff[j] = row * n128 + sl * 16;
  }
  const int xs = (c >> 1) & 7;
#pragma unroll 1
  for (int tile = 0; tile < ntile; ++tile) {
    double out[16];
#pragma unroll
    for (int set = 0; set < 2; ++set) {
      if (set >= nset) break;
      const unsigned lut = set == 0 ? LUT_DOSAGE : LUT_MISS;
      const int8_t* psrc = planes + (((grp_idx * 2 + set) * PI8_NPIECE + wave) * (int64_t)PI8_ROWS + (int64_t)tile * 32) * n128;
      auto issue = [&](int st, int buf) {
#pragma unroll
        for (int j = 0; j < 4; ++j) glds16p(psrc + roff[j] + st * 128, lds0 + buf * PV2_STAGE + (wave * 4 + j) * 1024);
      };
      v16i acc[PI8_NPIECE];
#pragma unroll
      for (int k = 0; k < PI8_NPIECE; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][r] = 0;
      uint32_t gw[4], gn[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) gw[t] = brow[(int64_t)t * 64];
      __syncthreads();                           // the previous tile's epilogue is done with the staging area
      issue(0, 0);
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll 1
      for (int st = 0; st < nstage; ++st) {
        const int8_t* cur = sA + (st & 1) * PV2_STAGE;
        if (st + 1 < nstage) {
          issue(st + 1, (st + 1) & 1);
#pragma unroll
          for (int t = 0; t < 4; ++t) gn[t] = brow[(int64_t)((st + 1) * 4 + t) * 64];
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const uint32_t w = gw[t];
          const v4i bf = (v4i){(int)pi8_expand4(w & 0xFFu, lut), (int)pi8_expand4((w >> 8) & 0xFFu, lut),
                               (int)pi8_expand4((w >> 16) & 0xFFu, lut), (int)pi8_expand4(w >> 24, lut)};
          const int8_t* arow = cur + c * 128 + (((2 * t + kb) ^ xs) << 4);
#pragma unroll
          for (int k = 0; k < PI8_NPIECE; ++k) {
            const v4i af = *reinterpret_cast<const v4i*>(arow + k * 4096);
            acc[k] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af, bf, acc[k], 0, 0, 0);
          }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) gw[t] = gn[t];
        // the next stage has landed and everybody is done with this one
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      }
      // the planes meet: sum_k 128^k S_k (exact terms), times the row's 2^(e-54)
      {
        const double* scrow = psc + (grp_idx * 2 + set) * PI8_ROWS + tile * 32 + 4 * kb;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          double oi = (double)acc[7][r];
#pragma unroll
          for (int k = PI8_NPIECE - 2; k >= 0; --k) oi = fma(oi, 128.0, (double)acc[k][r]);
          out[r] = set == 0 ? oi * scrow[(r & 3) + 8 * (r >> 2)] : fma(oi, scrow[(r & 3) + 8 * (r >> 2)], out[r]);
        }
      }
    }
    if (a.C > PI8_CPRE) pi8_corr_tail(a, tile, nrow, blk, s, p0, pos, kb, out);
    pi8_epilogue(a, tile, pos, wave, c, kb, out, scb, srow_w, xv, tile == 0 ? mkbits[0] : mkbits[1], reinterpret_cast<double*>(sA), sred);
  }
  __syncthreads();
  if (threadIdx.x < nrow * 2) {
    const int m = threadIdx.x >> 1, q = threadIdx.x & 1;
    const int pl = m / R0, rr = m % R0;
    double tsum = 0.0;
#pragma unroll
    for (int w = 0; w < 8; ++w) tsum += sred[w][m][q];
    a.psum[((((int64_t)blk * ct.n + ch) * a.P + p0 + pl) * 8 + rr) * 2 + q] = tsum;
  }
}

// planes: nblk * nseg * ngrp * 2 * 8 * 64 * n128 bytes; psc: nblk * nseg * ngrp * 2 * 64 doubles; pkT: nblk * Np * n128 / 4 bytes
void rg_launch_l0_pred_i8(hipStream_t st, const PredArgs& a, const ChunkTab& c256, int pg, int ngrp, int8_t* planes, double* psc,
                          uint8_t* pkT) {
  hipLaunchKernelGGL(k_pk_transpose, dim3((unsigned)(a.Np / 128), a.n128 / 128, a.nblk), dim3(256), 0, st, a.pk, a.pk_ld, a.pk_blk_stride,
                     a.n128, a.Np, pkT);
  hipLaunchKernelGGL(k_beta_split, dim3(PI8_ROWS, a.nseg * ngrp, a.nblk), dim3(256), 0, st, a, pg, ngrp, planes, psc);
  const size_t lds = (size_t)PI8_NPIECE * PI8_PLANE + sizeof(double) * 8 * PI8_ROWS * 2;     // 135,168 + 8,192 bytes
  static const bool v1 = getenv("RG_PRED_V1") != nullptr;     // the register-staged kernel above (round 2 - 4), kept for comparison
  if (!v1) {
    const size_t lds2 = (size_t)8 * 32 * PI8_TMP_PITCH * 8 + sizeof(double) * 8 * PI8_ROWS * 2;      // max(two stages, the epilogue's tmp) + sums
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_l0_pred_i8v2), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
    hipLaunchKernelGGL(k_l0_pred_i8v2, dim3(c256.n, ngrp, a.nblk), dim3(512), lds2, st, a, c256, pg, ngrp, (const int8_t*)planes,
                       (const double*)psc, (const uint8_t*)pkT);
    return;
  }
  // more than 64 KB of dynamic LDS needs the attribute; set per launch (it is per device, and a process may drive several)
  if (a.n128 % PI8_KHALF == 0) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_l0_pred_i8<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k_l0_pred_i8<true>, dim3(c256.n, ngrp, a.nblk), dim3(512), lds, st, a, c256, pg, ngrp, (const int8_t*)planes,
                       (const double*)psc, (const uint8_t*)pkT);
  } else {
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_l0_pred_i8<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k_l0_pred_i8<false>, dim3(c256.n, ngrp, a.nblk), dim3(512), lds, st, a, c256, pg, ngrp, (const int8_t*)planes,
                       (const double*)psc, (const uint8_t*)pkT);
  }
}
