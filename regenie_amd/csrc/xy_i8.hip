// G~X and G~Y (the two skinny products of residualize_genotypes and calc_cv_matrices, reference src/Data.cpp:199 and :746)
// on the i8 matrix cores, exactly.
//
// out[j][c] = sum_pos g_j(pos) * V_c(pos) over the positions of a fold, V = [X | Y] (C + P fp64 columns): the contraction index
// is the sample, which is how the packed rows are stored, and the genotype operand is an exact small integer, so -- as in
// pred_i8.hip -- the fp64 operand is taken apart instead of rounded: every column of V is written in fixed point against its
// largest entry (2^e > max |V_c|, q = rint(V_c(pos) * 2^(54-e))) and split into eight balanced base-128 digits d_k in
// [-64, 63] (k_v_split, once per problem: V does not change between blocks); S_k[j][c] = sum_pos g_j(pos) d_k[c](pos) runs on
// v_mfma_i32_32x32x32_i8 with exact int32 sums (|S_k| <= 64 * 2 * n_fold < 2^31 up to 16 million samples per fold); the value is
// 2^(e-54) * sum_k 128^k S_k in fp64 (k_xy_combine).  The truncation of V at 2^-54 of the column's largest entry is below the
// rounding the chunked fp64 sums of k_geno_xy (bed_prep.hip) carry.  Missing calls: the same contraction with the missing
// indicator, for blocks that have any.
// Tile: 128 SNP rows x 128 (column, digit) pairs per 256-thread workgroup, K step = 64 positions: the packed row piece is
// expanded to int8 with v_perm_b32 as a byte LUT and the digit rows are copied, both into 80-byte-pitch LDS images (the tile
// idiom of gram_i8.hip).  At 500,000 samples and 13 columns this is 1.3*10^11 integer multiply-adds per block against the
// 29 ms per 32 blocks of the fp64 VALU kernel.
#include "rg_internal.h"
#include <cstdlib>

#define XT 128
#define X_PITCH 80
#define X_NPIECE 8
#define X_LUT_MISS 0x00000100u      // byte k = value of .bed code k: 01 -> 1

// ---- digit planes of V: vd [Cv][8][Np] int8, vsc [Cv] = 2^(e-54); grid (Cv), 256 threads -----------------------------------
__global__ __launch_bounds__(256) void k_v_split(const double* __restrict__ V, int64_t Np, int8_t* __restrict__ vd, double* __restrict__ vsc) {
  __shared__ double red[4];
  __shared__ double smax;
  const int c = blockIdx.x;
  const double* v = V + (int64_t)c * Np;
  double mx = 0.0;
  for (int64_t i = threadIdx.x; i < Np; i += 256) mx = fmax(mx, fabs(v[i]));
  for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_down(mx, o));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) smax = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
  __syncthreads();
  mx = smax;
  int e = 0;
  if (mx > 0.0) (void)frexp(mx, &e);
  const double up = mx > 0.0 ? ldexp(1.0, 54 - e) : 0.0;
  if (threadIdx.x == 0) vsc[c] = mx > 0.0 ? ldexp(1.0, e - 54) : 0.0;
  int8_t* d0 = vd + (int64_t)c * X_NPIECE * Np;
  for (int64_t i = threadIdx.x; i < Np; i += 256) {
    long long q = llrint(v[i] * up);
#pragma unroll
    for (int k = 0; k < X_NPIECE; ++k) {
      const int d = (int)(((q & 127) ^ 64) - 64);
      d0[(int64_t)k * Np + i] = (int8_t)d;
      q = (q - d) >> 7;
    }
  }
}

__device__ __forceinline__ unsigned x_expand4(unsigned b, unsigned lut) {
  unsigned x = b | (b << 6);
  x = x | (x << 12);
  x &= 0x03030303u;
  return __builtin_amdgcn_perm(lut, lut, x);
}

// ---- S[blk][set][fold][row][col] = sum over the fold's positions; grid (n128 / 128, nseg, nblk * 2), set = z & 1 ----------------
__global__ __launch_bounds__(256, 4) void k_xy_i8(const uint8_t* __restrict__ pk, int64_t pk_ld, int64_t pk_blk_stride, const int32_t* __restrict__ d_bs,
                                               const int32_t* __restrict__ nmiss, int n128, SegLayout seg, const int8_t* __restrict__ vd,
                                               int64_t vd_blk_stride, int meta_bcast /* 1: every blk reads d_bs[0] / nmiss[0] */, int64_t Np,
                                               int ncol_all /* Cv * 8 <= 128 */, int ncol_last /* >= 0: of the last blk */,
                                               unsigned lut0 /* set 0: RG_XY_LUT_* */, int32_t* __restrict__ S) {
  constexpr int PITCH = 144;      // 128 positions per step + 16: 16-byte reads of 32 consecutive rows spread over the banks
  __shared__ __attribute__((aligned(16))) uint8_t sA[XT * PITCH];
  __shared__ __attribute__((aligned(16))) uint8_t sB[XT * PITCH];
  const int blk = blockIdx.z >> 1, set = blockIdx.z & 1, f = blockIdx.y, tr = blockIdx.x;
  const int mb = meta_bcast ? 0 : blk;
  const int ncol = (ncol_last >= 0 && blk == (int)(gridDim.z >> 1) - 1) ? ncol_last : ncol_all;
  if (set == 1 && nmiss[mb] == 0) return;
  const int bs = d_bs[mb];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const unsigned lut = set ? X_LUT_MISS : lut0;
  const int64_t pos0 = seg.pos_start[f], kbytes = seg.plen[f] / 4;
  v16i acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0;
  // staging: every thread expands 32 positions of one packed row (8 bytes -> 32 int8) AND copies 32 bytes of one digit row, so the
  // byte-LUT work is spread over all four waves (it was the first two waves' alone: two of the four SIMDs carried all of it)
  const int srow = tid >> 1, half = tid & 1;
  const int arow = tr * XT + srow;
  const bool validA = arow < bs;
  const uint8_t* ga = pk + (int64_t)blk * pk_blk_stride + (int64_t)(validA ? arow : 0) * pk_ld + pos0 / 4 + half * 16;
  const bool validB = srow < ncol;
  const int8_t* gb = vd + (int64_t)blk * vd_blk_stride + (int64_t)(validB ? srow : 0) * Np + pos0 + half * 64;   // (column, digit) row srow = c * 8 + k of vd [Cv][8][Np]
  uint8_t* lrowA = sA + srow * PITCH + half * 64;
  uint8_t* lrowB = sB + srow * PITCH + half * 64;
  // the global loads of step kb + 32 are issued before the MFMAs of step kb (registers), so their latency hides behind the math
  uint4 w = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);   // code 11 -> 0 under both LUTs
  uint4 v0 = make_uint4(0, 0, 0, 0), v1 = v0, v2 = v0, v3 = v0;
  if (kbytes > 0) {
    if (validA) w = *reinterpret_cast<const uint4*>(ga);
    if (validB) { const uint4* src = reinterpret_cast<const uint4*>(gb); v0 = src[0]; v1 = src[1]; v2 = src[2]; v3 = src[3]; }
  }
  for (int64_t kb = 0; kb < kbytes; kb += 32) {      // 128 positions per step (one barrier pair per 128)
    {
      const unsigned ws[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        uint4 o;
        o.x = x_expand4(ws[d] & 0xFFu, lut);
        o.y = x_expand4((ws[d] >> 8) & 0xFFu, lut);
        o.z = x_expand4((ws[d] >> 16) & 0xFFu, lut);
        o.w = x_expand4(ws[d] >> 24, lut);
        *reinterpret_cast<uint4*>(lrowA + d * 16) = o;
      }
      *reinterpret_cast<uint4*>(lrowB) = v0;
      *reinterpret_cast<uint4*>(lrowB + 16) = v1;
      *reinterpret_cast<uint4*>(lrowB + 32) = v2;
      *reinterpret_cast<uint4*>(lrowB + 48) = v3;
    }
    __syncthreads();
    if (kb + 32 < kbytes) {
      if (validA) w = *reinterpret_cast<const uint4*>(ga + kb + 32);
      if (validB) { const uint4* src = reinterpret_cast<const uint4*>(gb + (kb + 32) * 4); v0 = src[0]; v1 = src[1]; v2 = src[2]; v3 = src[3]; }
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      v4i af[2], bf[2];
      const int koff = ks * 32 + (lane >> 5) * 16;
#pragma unroll
      for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const v4i*>(sA + (wr * 64 + i * 32 + (lane & 31)) * PITCH + koff);
#pragma unroll
      for (int j = 0; j < 2; ++j) bf[j] = *reinterpret_cast<const v4i*>(sB + (wc * 64 + j * 32 + (lane & 31)) * PITCH + koff);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (wc * 64 + j * 32 >= ncol) continue;      // a 32-column block past the last (column, digit) pair: nothing but zeros (wave-uniform)
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[i], bf[j], acc[i][j], 0, 0, 0);
      }
    }
    __syncthreads();
  }
  // C/D map of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  int32_t* Sf = S + ((((int64_t)blk * 2 + set) * seg.nseg + f) * n128) * (int64_t)XT;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = tr * XT + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int col = wc * 64 + j * 32 + (lane & 31);
        Sf[(int64_t)row * XT + col] = acc[i][j][r];
      }
}

// ---- both sets (allele count and missing indicator) of ONE block of rows against ngrp column groups in one pass: the packed rows are
// read and the digit rows staged once for the two contractions (step2_qt.hip, blocks that have a missing call).
// S[grp][set][seg][row][col]; grid (n128 / 128, nseg, ngrp) ------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void k_xy_i8_both(const uint8_t* __restrict__ pk, int64_t pk_ld, const int32_t* __restrict__ d_bs, int n128, SegLayout seg,
                                                       const int8_t* __restrict__ vd, int64_t Np, int ncol_last, unsigned lut0, int32_t* __restrict__ S) {
  constexpr int PITCH = 144;      // 128 positions per step + 16: 16-byte reads of 32 consecutive rows spread over the banks
  __shared__ __attribute__((aligned(16))) uint8_t sA[2][XT * PITCH];
  __shared__ __attribute__((aligned(16))) uint8_t sB[XT * PITCH];
  const int grp = blockIdx.z, f = blockIdx.y, tr = blockIdx.x;
  const int ncol = grp == (int)gridDim.z - 1 ? ncol_last : 16 * X_NPIECE;
  const int bs = d_bs[0];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int64_t pos0 = seg.pos_start[f], kbytes = seg.plen[f] / 4;
  v16i acc[2][2][2];
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][i][j][r] = 0;
  const int srow = tid >> 1, half = tid & 1;
  const int arow = tr * XT + srow;
  const bool validA = arow < bs, validB = srow < ncol;
  const uint8_t* ga = pk + (int64_t)(validA ? arow : 0) * pk_ld + pos0 / 4 + half * 16;
  const int8_t* gb = vd + (int64_t)grp * 16 * X_NPIECE * Np + (int64_t)(validB ? srow : 0) * Np + pos0 + half * 64;
  uint8_t* lrowA0 = sA[0] + srow * PITCH + half * 64;
  uint8_t* lrowA1 = sA[1] + srow * PITCH + half * 64;
  uint8_t* lrowB = sB + srow * PITCH + half * 64;
  uint4 w = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);   // code 11 -> 0 under both LUTs
  uint4 v0 = make_uint4(0, 0, 0, 0), v1 = v0, v2 = v0, v3 = v0;
  if (kbytes > 0) {
    if (validA) w = *reinterpret_cast<const uint4*>(ga);
    if (validB) { const uint4* src = reinterpret_cast<const uint4*>(gb); v0 = src[0]; v1 = src[1]; v2 = src[2]; v3 = src[3]; }
  }
  for (int64_t kb = 0; kb < kbytes; kb += 32) {      // 128 positions per step (one barrier pair per 128)
    {
      const unsigned ws[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        uint4 o, m;
        o.x = x_expand4(ws[d] & 0xFFu, lut0);          m.x = x_expand4(ws[d] & 0xFFu, X_LUT_MISS);
        o.y = x_expand4((ws[d] >> 8) & 0xFFu, lut0);   m.y = x_expand4((ws[d] >> 8) & 0xFFu, X_LUT_MISS);
        o.z = x_expand4((ws[d] >> 16) & 0xFFu, lut0);  m.z = x_expand4((ws[d] >> 16) & 0xFFu, X_LUT_MISS);
        o.w = x_expand4(ws[d] >> 24, lut0);            m.w = x_expand4(ws[d] >> 24, X_LUT_MISS);
        *reinterpret_cast<uint4*>(lrowA0 + d * 16) = o;
        *reinterpret_cast<uint4*>(lrowA1 + d * 16) = m;
      }
      *reinterpret_cast<uint4*>(lrowB) = v0;
      *reinterpret_cast<uint4*>(lrowB + 16) = v1;
      *reinterpret_cast<uint4*>(lrowB + 32) = v2;
      *reinterpret_cast<uint4*>(lrowB + 48) = v3;
    }
    __syncthreads();
    if (kb + 32 < kbytes) {
      if (validA) w = *reinterpret_cast<const uint4*>(ga + kb + 32);
      if (validB) { const uint4* src = reinterpret_cast<const uint4*>(gb + (kb + 32) * 4); v0 = src[0]; v1 = src[1]; v2 = src[2]; v3 = src[3]; }
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      v4i bf[2];
      const int koff = ks * 32 + (lane >> 5) * 16;
#pragma unroll
      for (int j = 0; j < 2; ++j) bf[j] = *reinterpret_cast<const v4i*>(sB + (wc * 64 + j * 32 + (lane & 31)) * PITCH + koff);
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        v4i af[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const v4i*>(sA[q] + (wr * 64 + i * 32 + (lane & 31)) * PITCH + koff);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if (wc * 64 + j * 32 >= ncol) continue;
#pragma unroll
          for (int i = 0; i < 2; ++i) acc[q][i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[i], bf[j], acc[q][i][j], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    int32_t* Sf = S + ((((int64_t)grp * 2 + q) * seg.nseg + f) * n128) * (int64_t)XT;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = tr * XT + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          const int col = wc * 64 + j * 32 + (lane & 31);
          Sf[(int64_t)row * XT + col] = acc[q][i][j][r];
        }
  }
}

// ---- TWO column groups in one pass when the second is narrow (round 6) ---------------------------------------------------------------------
// BASELINE configs[4]'s Step 2 contracts 10 covariates + 10 residual columns = 160 (column, digit) pairs: one full group of 128 and a second
// group of 32.  As two passes the second one expanded every packed row again for a quarter of the products (its workgroups are staging- and
// barrier-bound).  Here the workgroup of group 0 also stages the E <= 2 extra 32-pair blocks of group 1 (sB rows 128 .. 128 + 32 E - 1) and
// every wave takes a share of their products on the A fragments it holds anyway.  S layout unchanged ([grp][set][seg][row][col]: the
// extra results land in group 1's columns 0 .. 32 E - 1).
// both sets: grid (n128 / 128, nseg); wave (wr, wc) takes set wc of the extra blocks for its own 64 rows.
template <int E>
__global__ __launch_bounds__(256, 2) void k_xy_i8_both_x(const uint8_t* __restrict__ pk, int64_t pk_ld, const int32_t* __restrict__ d_bs, int n128, SegLayout seg,
                                                         const int8_t* __restrict__ vd, int64_t Np, int ncol1 /* pairs of group 1 */, unsigned lut0,
                                                         int32_t* __restrict__ S) {
  constexpr int PITCH = 144;
  __shared__ __attribute__((aligned(16))) uint8_t sA[2][XT * PITCH];
  __shared__ __attribute__((aligned(16))) uint8_t sB[(XT + 32 * E) * PITCH];
  const int f = blockIdx.y, tr = blockIdx.x;
  const int bs = d_bs[0];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int64_t pos0 = seg.pos_start[f], kbytes = seg.plen[f] / 4;
  v16i acc[2][2][2], accx[E][2];
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][i][j][r] = 0;
#pragma unroll
  for (int e = 0; e < E; ++e)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) accx[e][i][r] = 0;
  const int srow = tid >> 1, half = tid & 1;
  const int arow = tr * XT + srow;
  const bool validA = arow < bs;
  const bool validX = tid < 64 * E && srow < ncol1;                       // threads 0 .. 64 E - 1 also copy a row of group 1
  const uint8_t* ga = pk + (int64_t)(validA ? arow : 0) * pk_ld + pos0 / 4 + half * 16;
  const int8_t* gb = vd + (int64_t)srow * Np + pos0 + half * 64;
  const int8_t* gx = vd + (int64_t)16 * X_NPIECE * Np + (int64_t)(validX ? srow : 0) * Np + pos0 + half * 64;
  uint8_t* lrowA0 = sA[0] + srow * PITCH + half * 64;
  uint8_t* lrowA1 = sA[1] + srow * PITCH + half * 64;
  uint8_t* lrowB = sB + srow * PITCH + half * 64;
  uint8_t* lrowX = sB + (XT + srow) * PITCH + half * 64;
  uint4 w = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);   // code 11 -> 0 under both LUTs
  uint4 v0 = make_uint4(0, 0, 0, 0), v1 = v0, v2 = v0, v3 = v0, x0 = v0, x1 = v0, x2 = v0, x3 = v0;
  if (kbytes > 0) {
    if (validA) w = *reinterpret_cast<const uint4*>(ga);
    { const uint4* src = reinterpret_cast<const uint4*>(gb); v0 = src[0]; v1 = src[1]; v2 = src[2]; v3 = src[3]; }
    if (validX) { const uint4* src = reinterpret_cast<const uint4*>(gx); x0 = src[0]; x1 = src[1]; x2 = src[2]; x3 = src[3]; }
  }
  for (int64_t kb = 0; kb < kbytes; kb += 32) {
    {
      const unsigned ws[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        uint4 o, m;
        o.x = x_expand4(ws[d] & 0xFFu, lut0);          m.x = x_expand4(ws[d] & 0xFFu, X_LUT_MISS);
        o.y = x_expand4((ws[d] >> 8) & 0xFFu, lut0);   m.y = x_expand4((ws[d] >> 8) & 0xFFu, X_LUT_MISS);
        o.z = x_expand4((ws[d] >> 16) & 0xFFu, lut0);  m.z = x_expand4((ws[d] >> 16) & 0xFFu, X_LUT_MISS);
        o.w = x_expand4(ws[d] >> 24, lut0);            m.w = x_expand4(ws[d] >> 24, X_LUT_MISS);
        *reinterpret_cast<uint4*>(lrowA0 + d * 16) = o;
        *reinterpret_cast<uint4*>(lrowA1 + d * 16) = m;
      }
      *reinterpret_cast<uint4*>(lrowB) = v0;
      *reinterpret_cast<uint4*>(lrowB + 16) = v1;
      *reinterpret_cast<uint4*>(lrowB + 32) = v2;
      *reinterpret_cast<uint4*>(lrowB + 48) = v3;
      if (tid < 64 * E) {
        *reinterpret_cast<uint4*>(lrowX) = x0;
        *reinterpret_cast<uint4*>(lrowX + 16) = x1;
        *reinterpret_cast<uint4*>(lrowX + 32) = x2;
        *reinterpret_cast<uint4*>(lrowX + 48) = x3;
      }
    }
    __syncthreads();
    if (kb + 32 < kbytes) {
      if (validA) w = *reinterpret_cast<const uint4*>(ga + kb + 32);
      { const uint4* src = reinterpret_cast<const uint4*>(gb + (kb + 32) * 4); v0 = src[0]; v1 = src[1]; v2 = src[2]; v3 = src[3]; }
      if (validX) { const uint4* src = reinterpret_cast<const uint4*>(gx + (kb + 32) * 4); x0 = src[0]; x1 = src[1]; x2 = src[2]; x3 = src[3]; }
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      v4i bf[2], bx[E];
      const int koff = ks * 32 + (lane >> 5) * 16;
#pragma unroll
      for (int j = 0; j < 2; ++j) bf[j] = *reinterpret_cast<const v4i*>(sB + (wc * 64 + j * 32 + (lane & 31)) * PITCH + koff);
#pragma unroll
      for (int e = 0; e < E; ++e) bx[e] = *reinterpret_cast<const v4i*>(sB + (XT + e * 32 + (lane & 31)) * PITCH + koff);
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        v4i af[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const v4i*>(sA[q] + (wr * 64 + i * 32 + (lane & 31)) * PITCH + koff);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int i = 0; i < 2; ++i) acc[q][i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[i], bf[j], acc[q][i][j], 0, 0, 0);
        if (q == wc) {
#pragma unroll
          for (int e = 0; e < E; ++e)
#pragma unroll
            for (int i = 0; i < 2; ++i) accx[e][i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[i], bx[e], accx[e][i], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    int32_t* Sf = S + ((((int64_t)0 * 2 + q) * seg.nseg + f) * n128) * (int64_t)XT;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = tr * XT + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          const int col = wc * 64 + j * 32 + (lane & 31);
          Sf[(int64_t)row * XT + col] = acc[q][i][j][r];
        }
  }
  {
    int32_t* Sx = S + ((((int64_t)1 * 2 + wc) * seg.nseg + f) * n128) * (int64_t)XT;
#pragma unroll
    for (int e = 0; e < E; ++e)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = tr * XT + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          Sx[(int64_t)row * XT + e * 32 + (lane & 31)] = accx[e][i][r];
        }
  }
}

// one set per workgroup (grid (n128 / 128, nseg, 2), set = blockIdx.z; set 1 skipped when *nmiss == 0): wave (wr, wc) takes the extra blocks'
// rows of its row block 2 wr + wc (its own A fragment af[wc]).
template <int E>
__global__ __launch_bounds__(256, 3) void k_xy_i8_sums_x(const uint8_t* __restrict__ pk, int64_t pk_ld, const int32_t* __restrict__ d_bs,
                                                         const int32_t* __restrict__ nmiss, int n128, SegLayout seg, const int8_t* __restrict__ vd, int64_t Np,
                                                         int ncol1 /* pairs of group 1 */, unsigned lut0, int32_t* __restrict__ S) {
  constexpr int PITCH = 144;
  __shared__ __attribute__((aligned(16))) uint8_t sA[XT * PITCH];
  __shared__ __attribute__((aligned(16))) uint8_t sB[(XT + 32 * E) * PITCH];
  const int set = blockIdx.z, f = blockIdx.y, tr = blockIdx.x;
  if (set == 1 && nmiss[0] == 0) return;
  const int bs = d_bs[0];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const unsigned lut = set ? X_LUT_MISS : lut0;
  const int64_t pos0 = seg.pos_start[f], kbytes = seg.plen[f] / 4;
  v16i acc[2][2], accx[E];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0;
#pragma unroll
  for (int e = 0; e < E; ++e)
#pragma unroll
    for (int r = 0; r < 16; ++r) accx[e][r] = 0;
  const int srow = tid >> 1, half = tid & 1;
  const int arow = tr * XT + srow;
  const bool validA = arow < bs;
  const bool validX = tid < 64 * E && srow < ncol1;
  const uint8_t* ga = pk + (int64_t)(validA ? arow : 0) * pk_ld + pos0 / 4 + half * 16;
  const int8_t* gb = vd + (int64_t)srow * Np + pos0 + half * 64;
  const int8_t* gx = vd + (int64_t)16 * X_NPIECE * Np + (int64_t)(validX ? srow : 0) * Np + pos0 + half * 64;
  uint8_t* lrowA = sA + srow * PITCH + half * 64;
  uint8_t* lrowB = sB + srow * PITCH + half * 64;
  uint8_t* lrowX = sB + (XT + srow) * PITCH + half * 64;
  uint4 w = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
  uint4 v0 = make_uint4(0, 0, 0, 0), v1 = v0, v2 = v0, v3 = v0, x0 = v0, x1 = v0, x2 = v0, x3 = v0;
  if (kbytes > 0) {
    if (validA) w = *reinterpret_cast<const uint4*>(ga);
    { const uint4* src = reinterpret_cast<const uint4*>(gb); v0 = src[0]; v1 = src[1]; v2 = src[2]; v3 = src[3]; }
    if (validX) { const uint4* src = reinterpret_cast<const uint4*>(gx); x0 = src[0]; x1 = src[1]; x2 = src[2]; x3 = src[3]; }
  }
  for (int64_t kb = 0; kb < kbytes; kb += 32) {
    {
      const unsigned ws[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        uint4 o;
        o.x = x_expand4(ws[d] & 0xFFu, lut);
        o.y = x_expand4((ws[d] >> 8) & 0xFFu, lut);
        o.z = x_expand4((ws[d] >> 16) & 0xFFu, lut);
        o.w = x_expand4(ws[d] >> 24, lut);
        *reinterpret_cast<uint4*>(lrowA + d * 16) = o;
      }
      *reinterpret_cast<uint4*>(lrowB) = v0;
      *reinterpret_cast<uint4*>(lrowB + 16) = v1;
      *reinterpret_cast<uint4*>(lrowB + 32) = v2;
      *reinterpret_cast<uint4*>(lrowB + 48) = v3;
      if (tid < 64 * E) {
        *reinterpret_cast<uint4*>(lrowX) = x0;
        *reinterpret_cast<uint4*>(lrowX + 16) = x1;
        *reinterpret_cast<uint4*>(lrowX + 32) = x2;
        *reinterpret_cast<uint4*>(lrowX + 48) = x3;
      }
    }
    __syncthreads();
    if (kb + 32 < kbytes) {
      if (validA) w = *reinterpret_cast<const uint4*>(ga + kb + 32);
      { const uint4* src = reinterpret_cast<const uint4*>(gb + (kb + 32) * 4); v0 = src[0]; v1 = src[1]; v2 = src[2]; v3 = src[3]; }
      if (validX) { const uint4* src = reinterpret_cast<const uint4*>(gx + (kb + 32) * 4); x0 = src[0]; x1 = src[1]; x2 = src[2]; x3 = src[3]; }
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      v4i af[2], bf[2];
      const int koff = ks * 32 + (lane >> 5) * 16;
#pragma unroll
      for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const v4i*>(sA + (wr * 64 + i * 32 + (lane & 31)) * PITCH + koff);
#pragma unroll
      for (int j = 0; j < 2; ++j) bf[j] = *reinterpret_cast<const v4i*>(sB + (wc * 64 + j * 32 + (lane & 31)) * PITCH + koff);
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[i], bf[j], acc[i][j], 0, 0, 0);
      const v4i ax = wc ? af[1] : af[0];
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const v4i bx = *reinterpret_cast<const v4i*>(sB + (XT + e * 32 + (lane & 31)) * PITCH + koff);
        accx[e] = __builtin_amdgcn_mfma_i32_32x32x32_i8(ax, bx, accx[e], 0, 0, 0);
      }
    }
    __syncthreads();
  }
  int32_t* Sf = S + ((((int64_t)0 * 2 + set) * seg.nseg + f) * n128) * (int64_t)XT;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = tr * XT + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int col = wc * 64 + j * 32 + (lane & 31);
        Sf[(int64_t)row * XT + col] = acc[i][j][r];
      }
  int32_t* Sx = S + ((((int64_t)1 * 2 + set) * seg.nseg + f) * n128) * (int64_t)XT;
#pragma unroll
  for (int e = 0; e < E; ++e)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = tr * XT + wr * 64 + wc * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      Sx[(int64_t)row * XT + e * 32 + (lane & 31)] = accx[e][r];
    }
}

// ---- the same contraction when the row operand already is int8 (digit planes of integer dosages, step2_qt.hip): A rows are copied like
// the digit rows.  aplanes [nset][rows][Np]; S[grp][set][seg][row][col]; grid (n128 / 128, nseg, ngrp * nset) --------------------------
__global__ __launch_bounds__(256, 4) void k_xy_i8_planes(const int8_t* __restrict__ aplanes, int64_t a_set_stride, int nset, const int32_t* __restrict__ d_bs,
                                                         int n128, SegLayout seg, const int8_t* __restrict__ vd, int64_t Np, int ncol_last,
                                                         int32_t* __restrict__ S) {
  __shared__ __attribute__((aligned(16))) uint8_t sA[XT * X_PITCH];
  __shared__ __attribute__((aligned(16))) uint8_t sB[XT * X_PITCH];
  const int grp = blockIdx.z / nset, set = blockIdx.z - grp * nset, f = blockIdx.y, tr = blockIdx.x;
  const int ngrp = gridDim.z / nset;
  const int ncol = grp == ngrp - 1 ? ncol_last : 16 * X_NPIECE;
  const int bs = d_bs[0];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int64_t pos0 = seg.pos_start[f], klen = seg.plen[f];
  v16i acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0;
  const int srow = tid >> 1, half = tid & 1;
  const int arow = tr * XT + srow;
  const bool validA = arow < bs, validB = srow < ncol;
  const int8_t* ga = aplanes + (int64_t)set * a_set_stride + (int64_t)(validA ? arow : 0) * Np + pos0 + half * 32;
  const int8_t* gb = vd + (int64_t)grp * 16 * X_NPIECE * Np + (int64_t)(validB ? srow : 0) * Np + pos0 + half * 32;
  uint8_t* lrowA = sA + srow * X_PITCH + half * 32;
  uint8_t* lrowB = sB + srow * X_PITCH + half * 32;
  const uint4 z4 = make_uint4(0, 0, 0, 0);
  uint4 a0 = z4, a1 = z4, v0 = z4, v1 = z4;
  if (klen > 0) {
    if (validA) { const uint4* src = reinterpret_cast<const uint4*>(ga); a0 = src[0]; a1 = src[1]; }
    if (validB) { const uint4* src = reinterpret_cast<const uint4*>(gb); v0 = src[0]; v1 = src[1]; }
  }
  for (int64_t k = 0; k < klen; k += 64) {      // 64 positions per step
    *reinterpret_cast<uint4*>(lrowA) = a0;
    *reinterpret_cast<uint4*>(lrowA + 16) = a1;
    *reinterpret_cast<uint4*>(lrowB) = v0;
    *reinterpret_cast<uint4*>(lrowB + 16) = v1;
    __syncthreads();
    if (k + 64 < klen) {
      if (validA) { const uint4* src = reinterpret_cast<const uint4*>(ga + k + 64); a0 = src[0]; a1 = src[1]; }
      if (validB) { const uint4* src = reinterpret_cast<const uint4*>(gb + k + 64); v0 = src[0]; v1 = src[1]; }
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      v4i af[2], bf[2];
      const int koff = ks * 32 + (lane >> 5) * 16;
#pragma unroll
      for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const v4i*>(sA + (wr * 64 + i * 32 + (lane & 31)) * X_PITCH + koff);
#pragma unroll
      for (int j = 0; j < 2; ++j) bf[j] = *reinterpret_cast<const v4i*>(sB + (wc * 64 + j * 32 + (lane & 31)) * X_PITCH + koff);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (wc * 64 + j * 32 >= ncol) continue;
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[i], bf[j], acc[i][j], 0, 0, 0);
      }
    }
    __syncthreads();
  }
  int32_t* Sf = S + ((((int64_t)grp * nset + set) * seg.nseg + f) * n128) * (int64_t)XT;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = tr * XT + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int col = wc * 64 + j * 32 + (lane & 31);
        Sf[(int64_t)row * XT + col] = acc[i][j][r];
      }
}

// ---- part[blk][fold][row][set][c0 + c] = vsc[c0 + c] * sum_k 128^k S[.][c*8+k] for the ncg columns of one group of 16;
// grid (ceil(n128*ncg / 256), nseg, nblk) ------------------------------------------------------------------------------------------
__global__ void k_xy_combine(const int32_t* __restrict__ S, const double* __restrict__ vsc, const int32_t* __restrict__ nmiss, int n128, int nseg,
                             int Cv, int c0, int ncg, double* __restrict__ part) {
  const int blk = blockIdx.z, f = blockIdx.y;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n128 * ncg) return;
  const int j = t / ncg, c = t - j * ncg;
  const int nset = nmiss[blk] > 0 ? 2 : 1;
  for (int set = 0; set < nset; ++set) {
    const int32_t* s = S + (((((int64_t)blk * 2 + set) * nseg + f) * n128 + j) * (int64_t)XT) + c * X_NPIECE;
    double v = 0.0, w = 1.0;
#pragma unroll
    for (int k = 0; k < X_NPIECE; ++k) { v = fma((double)s[k], w, v); w *= 128.0; }
    part[((((int64_t)blk * nseg + f) * n128 + j) * 2 + set) * Cv + c0 + c] = v * vsc[c0 + c];
  }
}

void rg_launch_v_split(hipStream_t st, const double* V, int64_t Np, int Cv, int8_t* vd, double* vsc) {
  hipLaunchKernelGGL(k_v_split, dim3(Cv), dim3(256), 0, st, V, Np, vd, vsc);
}

// the digit sums alone (step2_qt.hip sums them over the segments itself), ONE block of rows against ncols columns = ngrp groups of 16
// in one launch (the group takes the kernel's block index: its planes start at vd + grp * 16 * 8 * Np; the last group's missing columns
// are neither loaded nor multiplied): S32 [ngrp][2][nseg][n128][128]; lut0 = what set 0 contracts (the allele count, or its square); set 1 is the missing
// indicator, skipped when *nmiss == 0
// Two column groups whose second holds at most 32 (column, digit) pairs -- 17 to 20 columns -- go in ONE pass (k_xy_i8_*_x): the caller sizes its segments for that
// many groups of workgroups (RG_XY_NO_FUSE=1: always one pass per group).
int rg_xy_i8_launch_groups(int ncols) {
  static const bool no_fuse = getenv("RG_XY_NO_FUSE") && atoi(getenv("RG_XY_NO_FUSE")) != 0;
  const int ngrp = (ncols + 15) / 16;
  return (!no_fuse && ngrp == 2 && (ncols - 16) * X_NPIECE <= 32) ? 1 : ngrp;      // (E = 2, up to 24 columns: the both-sets kernel spills at 256 registers)
}
void rg_launch_xy_i8_sums(hipStream_t st, const uint8_t* pk, int64_t pk_ld, const int32_t* d_bs, const int32_t* nmiss, int ncols, int n128,
                          const SegLayout& seg, const int8_t* vd, int64_t Np, unsigned lut0, int32_t* S32) {
  const int ngrp = (ncols + 15) / 16;
  if (ngrp == 2 && rg_xy_i8_launch_groups(ncols) == 1) {
    const int n1 = (ncols - 16) * X_NPIECE;
    hipLaunchKernelGGL(k_xy_i8_sums_x<1>, dim3(n128 / XT, seg.nseg, 2), dim3(256), 0, st, pk, pk_ld, d_bs, nmiss, n128, seg, vd, Np, n1, lut0, S32);
    return;
  }
  hipLaunchKernelGGL(k_xy_i8, dim3(n128 / XT, seg.nseg, ngrp * 2), dim3(256), 0, st, pk, pk_ld, (int64_t)0, d_bs, nmiss, n128, seg, vd,
                     (int64_t)16 * X_NPIECE * Np, 1, Np, 16 * X_NPIECE, (ncols - (ngrp - 1) * 16) * X_NPIECE, lut0, S32);
}

// both sets in one pass (same S32 layout as rg_launch_xy_i8_sums)
void rg_launch_xy_i8_both(hipStream_t st, const uint8_t* pk, int64_t pk_ld, const int32_t* d_bs, int ncols, int n128, const SegLayout& seg,
                          const int8_t* vd, int64_t Np, unsigned lut0, int32_t* S32) {
  const int ngrp = (ncols + 15) / 16;
  if (ngrp == 2 && rg_xy_i8_launch_groups(ncols) == 1) {
    const int n1 = (ncols - 16) * X_NPIECE;
    hipLaunchKernelGGL(k_xy_i8_both_x<1>, dim3(n128 / XT, seg.nseg, 1), dim3(256), 0, st, pk, pk_ld, d_bs, n128, seg, vd, Np, n1, lut0, S32);
    return;
  }
  hipLaunchKernelGGL(k_xy_i8_both, dim3(n128 / XT, seg.nseg, ngrp), dim3(256), 0, st, pk, pk_ld, d_bs, n128, seg, vd, Np,
                     (ncols - (ngrp - 1) * 16) * X_NPIECE, lut0, S32);
}

// int8 row planes against ncols columns: S32 [ngrp][nset][nseg][n128][128]
void rg_launch_xy_i8_planes(hipStream_t st, const int8_t* aplanes, int64_t a_set_stride, int nset, const int32_t* d_bs, int ncols, int n128,
                            const SegLayout& seg, const int8_t* vd, int64_t Np, int32_t* S32) {
  const int ngrp = (ncols + 15) / 16;
  // (the one-pass form of a narrow second group -- k_xy_i8_sums_x -- was built for the planes too: 0.72 ms against 0.65 ms for the two passes at 1,024 rows; not kept)
  hipLaunchKernelGGL(k_xy_i8_planes, dim3(n128 / XT, seg.nseg, ngrp * nset), dim3(256), 0, st, aplanes, a_set_stride, nset, d_bs, n128, seg, vd, Np,
                     (ncols - (ngrp - 1) * 16) * X_NPIECE, S32);
}

// S32: nblk * 2 * nseg * n128 * 128 int32; part: [nblk][nseg][n128][2][Cv] (rowstats reads it with nchunk = nseg, chunk_seg = identity).
// More than 16 columns (many phenotypes: BASELINE configs[3] has 3 + 50): one pass per group of 16 columns, the sums of a group combined
// into its columns of `part` before the next group reuses S32 (stream order).
void rg_launch_xy_i8(hipStream_t st, const uint8_t* pk, int64_t pk_ld, int64_t pk_blk_stride, const int32_t* d_bs, const int32_t* nmiss,
                     int nblk, int n128, const SegLayout& seg, const int8_t* vd, const double* vsc, int64_t Np, int Cv, int32_t* S32,
                     double* part) {
  for (int c0 = 0; c0 < Cv; c0 += 16) {
    const int ncg = Cv - c0 < 16 ? Cv - c0 : 16;
    hipLaunchKernelGGL(k_xy_i8, dim3(n128 / XT, seg.nseg, nblk * 2), dim3(256), 0, st, pk, pk_ld, pk_blk_stride, d_bs, nmiss, n128, seg,
                       vd + (int64_t)c0 * X_NPIECE * Np, (int64_t)0, 0, Np, ncg * X_NPIECE, -1, RG_XY_LUT_DOSAGE, S32);
    hipLaunchKernelGGL(k_xy_combine, dim3((n128 * ncg + 255) / 256, seg.nseg, nblk), dim3(256), 0, st, (const int32_t*)S32, vsc, nmiss, n128, seg.nseg,
                       Cv, c0, ncg, part);
  }
}
