// Exact integer fold-Gram of a SNP block straight from 2-bit packed genotypes.
//
// Replaces the `Gmat * Gmat.transpose()` of calc_cv_matrices (reference src/Data.cpp:748) and the
// decode of readChunkFromBedFileToG (src/Geno.cpp:1724-1763): the raw dosages {0,1,2} and the
// missing indicator {0,1} are contracted on the i8 MFMA (v_mfma_i32_32x32x32_i8) with exact int32
// accumulation (every partial sum <= 4*N < 2^31); mean imputation, covariate residualisation and
// scaling are applied afterwards as an fp64 rank-(C+1) correction on the bs x bs result
// (assemble.hip), which is algebraically identical to the reference's order of operations.
//
// Tile: 128 x 128 outputs per 256-thread workgroup (4 waves, 64 x 64 each = 2 x 2 MFMA 32x32),
// K-step = 64 samples = 16 packed bytes per SNP row.  Each thread stages one packed 16-byte row
// piece per K-step, expands it to 64 int8 with v_perm_b32 as a 4-entry byte LUT and writes it to an
// LDS image with an 80-byte row pitch (conflict-free ds_read_b128 fragment reads).
#include "rg_internal.h"

#define GT 128
#define LDS_PITCH 80  // 64 data bytes + 16 pad: 16 consecutive rows hit 16 distinct 4-bank groups

// cleaned 2-bit code -> dosage / missing indicator, as byte LUTs for v_perm_b32 (selector = code)
//   00 -> 2 (hom. first allele)   01 -> missing   10 -> 1 (het)   11 -> 0     (Geno.cpp:2838-2843)
#define LUT_DOSAGE 0x00010002u
#define LUT_MISS 0x00000100u

__device__ __forceinline__ unsigned expand4(unsigned b, unsigned lut) {
  unsigned x = b | (b << 6);
  x = x | (x << 12);
  x &= 0x03030303u;
  return __builtin_amdgcn_perm(lut, lut, x);
}

__device__ __forceinline__ void stage_row(const uint8_t* gptr, bool valid, unsigned lut,
                                          uint8_t* lds_row) {
  uint4 w;
  if (valid) w = *reinterpret_cast<const uint4*>(gptr);
  else w = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);  // code 11 -> 0
  unsigned ws[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    uint4 o;
    o.x = expand4(ws[d] & 0xFFu, lut);
    o.y = expand4((ws[d] >> 8) & 0xFFu, lut);
    o.z = expand4((ws[d] >> 16) & 0xFFu, lut);
    o.w = expand4(ws[d] >> 24, lut);
    *reinterpret_cast<uint4*>(lds_row + d * 16) = o;
  }
}

// One 128x128 tile: C[r][c] = sum_k dec(A[r][k]) * dec(B[c][k]).
__device__ __forceinline__ void gram_tile(const uint8_t* __restrict__ A, int64_t lda, int a_rows,
                                          unsigned a_lut, const uint8_t* __restrict__ B,
                                          int64_t ldb, int b_rows, unsigned b_lut, bool same,
                                          int64_t kbytes, int32_t* __restrict__ C, int64_t ldc,
                                          int c_rows, int c_cols, uint8_t* sA, uint8_t* sB) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;

  v16i acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0;

  const bool isA = tid < 128;
  const int srow = isA ? tid : tid - 128;
  const uint8_t* gbase = isA ? A + (int64_t)srow * lda : B + (int64_t)srow * ldb;
  const bool valid = isA ? (srow < a_rows) : (srow < b_rows);
  const unsigned lut = isA ? a_lut : b_lut;
  uint8_t* lrow = (isA ? sA : sB) + srow * LDS_PITCH;
  const bool do_stage = isA || !same;
  const uint8_t* fB = same ? sA : sB;

  for (int64_t kb = 0; kb < kbytes; kb += 16) {
    if (do_stage) stage_row(gbase + kb, valid, lut, lrow);
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      v4i af[2], bf[2];
      const int koff = ks * 32 + (lane >> 5) * 16;
#pragma unroll
      for (int i = 0; i < 2; ++i)
        af[i] = *reinterpret_cast<const v4i*>(sA + (wr * 64 + i * 32 + (lane & 31)) * LDS_PITCH + koff);
#pragma unroll
      for (int j = 0; j < 2; ++j)
        bf[j] = *reinterpret_cast<const v4i*>(fB + (wc * 64 + j * 32 + (lane & 31)) * LDS_PITCH + koff);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }

  // C/D map of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int col = wc * 64 + j * 32 + (lane & 31);
        if (row < c_rows && col < c_cols) C[(int64_t)row * ldc + col] = acc[i][j][r];
      }
}

__global__ __launch_bounds__(256) void k_gram_generic(const uint8_t* A, int64_t lda, int a_miss,
                                                      const uint8_t* B, int64_t ldb, int b_miss,
                                                      int m, int n, int64_t kbytes, int32_t* C,
                                                      int64_t ldc) {
  __shared__ __attribute__((aligned(16))) uint8_t smem[2 * GT * LDS_PITCH];
  const int tr = blockIdx.y, tc = blockIdx.x;
  gram_tile(A + (int64_t)tr * GT * lda, lda, m - tr * GT, a_miss ? LUT_MISS : LUT_DOSAGE,
            B + (int64_t)tc * GT * ldb, ldb, n - tc * GT, b_miss ? LUT_MISS : LUT_DOSAGE, false,
            kbytes, C + (int64_t)tr * GT * ldc + tc * GT, ldc, m - tr * GT, n - tc * GT, smem,
            smem + GT * LDS_PITCH);
}

// Production launch: grid.x = lower-triangular tile index over the stacked [dosage; missing] rows
// (2*nt tile rows), grid.y = fold, grid.z = block of the batch.  Tiles that involve the missing
// indicator exit at once when the block has no missing call (nmiss[blk] == 0).
__global__ __launch_bounds__(256) void k_gram_blocks(const uint8_t* pk, int64_t pk_ld,
                                                     int64_t pk_blk_stride, int n128, SegLayout seg,
                                                     const int32_t* nmiss, int32_t* S, int miss_only) {
  __shared__ __attribute__((aligned(16))) uint8_t smem[2 * GT * LDS_PITCH];
  const int nt = n128 / GT;
  // XCD-aware remap: consecutive tile ids share operand panels; keep them on one XCD's L2.
  int tidx = blockIdx.x;
  {
    const int nwg = gridDim.x;
    const int q = nwg / 8, r = nwg % 8, xcd = tidx % 8, k = tidx / 8;
    tidx = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  }
  // unrank lower-triangular (tr >= tc) over 2*nt
  int tr = (int)((sqrtf(8.0f * tidx + 1.0f) - 1.0f) * 0.5f);
  while ((tr + 1) * (tr + 2) / 2 <= tidx) ++tr;
  while (tr * (tr + 1) / 2 > tidx) --tr;
  const int tc = tidx - tr * (tr + 1) / 2;
  const int blk = blockIdx.z, f = blockIdx.y;
  const bool a_miss = tr >= nt, b_miss = tc >= nt;
  if ((a_miss || b_miss) && nmiss[blk] == 0) return;
  if (miss_only && !(a_miss || b_miss)) return;   // dosage x dosage tiles come from gram_fp4.hip
  const int ar = (a_miss ? tr - nt : tr), br = (b_miss ? tc - nt : tc);
  const uint8_t* base = pk + (int64_t)blk * pk_blk_stride + seg.pos_start[f] / 4;
  const int64_t ldS = 2 * (int64_t)n128;
  int32_t* Sf = S + ((int64_t)blk * seg.nseg + f) * ldS * ldS;
  gram_tile(base + (int64_t)ar * GT * pk_ld, pk_ld, GT, a_miss ? LUT_MISS : LUT_DOSAGE,
            base + (int64_t)br * GT * pk_ld, pk_ld, GT, b_miss ? LUT_MISS : LUT_DOSAGE,
            (tr == tc), seg.plen[f] / 4, Sf + (int64_t)tr * GT * ldS + tc * GT, ldS, GT, GT, smem,
            smem + GT * LDS_PITCH);
}

void rg_launch_gram_blocks(hipStream_t st, const uint8_t* pk, int64_t pk_ld, int64_t pk_blk_stride,
                           int nblk, int n128, SegLayout seg, const int32_t* nmiss, int32_t* S, int miss_only) {
  const int nt2 = 2 * (n128 / GT);
  dim3 grid(nt2 * (nt2 + 1) / 2, seg.nseg, nblk);
  hipLaunchKernelGGL(k_gram_blocks, grid, dim3(256), 0, st, pk, pk_ld, pk_blk_stride, n128, seg,
                     nmiss, S, miss_only);
}

void rg_launch_gram_generic(hipStream_t st, const uint8_t* A, int64_t lda, int a_miss,
                            const uint8_t* B, int64_t ldb, int b_miss, int m, int n, int64_t kbytes,
                            int32_t* C, int64_t ldc) {
  dim3 grid((n + GT - 1) / GT, (m + GT - 1) / GT);
  hipLaunchKernelGGL(k_gram_generic, grid, dim3(256), 0, st, A, lda, a_miss, B, ldb, b_miss, m, n,
                     kbytes, C, ldc);
}
