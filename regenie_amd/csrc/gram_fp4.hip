// Exact fold-Gram of a SNP block on the CDNA4 FP4 matrix cores.
//
// Replaces the `Gmat * Gmat.transpose()` of calc_cv_matrices (reference src/Data.cpp:748) for the
// dosage x dosage part.  Raw dosages {0,1,2} are exactly representable in FP4 E2M1 (0 = 0000, 1 = 0010,
// 2 = 0100); v_mfma_scale_f32_32x32x64_f8f6f4 with unit block scales multiplies them exactly
// (products in {0,1,2,4}) and accumulates in fp32, which is exact while every partial sum stays below
// 2^24 -- the K loop flushes the fp32 accumulators into the int32 output every 2^20 samples, so the
// result is the exact integer Gram for any fold length.  Against the i8 MFMA this doubles the matrix
// rate (~10 vs ~5 POP/s dense) and needs no 2-bit -> int8 expansion in the inner loop: the ingest
// kernel (bed_prep.hip) writes the FP4 plane once per block (two samples per byte, position space)
// and the Gram kernel only copies bytes HBM -> LDS -> registers.
// Tiles that involve the missing-call indicator stay on the i8 kernel (gram_i8.hip); they exist only
// for blocks that have missing genotypes.
//
// Tile: 256 x 256 outputs per 1024-thread workgroup: 16 waves (8 x 2), 32 x 128 per wave (1 x 4 MFMA 32x32x64, 64 fp32
// accumulators per lane), four waves per SIMD.  K stage = 256 samples = 128 B per row, 512 rows (A and B operands),
// double-buffered in LDS and filled by direct global -> LDS copies (global_load_lds_dwordx4: no staging registers).
// Those copies land at wave-base + lane*16, so the LDS rows are unpadded (128 B pitch) and bank conflicts are avoided by
// an XOR swizzle of the 16-byte slot index applied to the per-lane GLOBAL address: slot' = slot ^ ((row >> 1) & 7), which
// makes the 16 rows of every ds_read_b128 lane group hit 16 distinct 4-bank groups.
// Why so many waves: a stage is 64 KB = 64 copy instructions per workgroup, and each costs its wave ~150 cycles of issue
// during which that wave issues no MFMA.  With one 128 x 128 wave per SIMD (16 copies, 64 MFMAs of 32 cycles per stage)
// the copies took longer than the MFMAs (46 % of the FP4 peak); with four waves per SIMD the copy-issue and LDS-read
// phases of one wave run under the MFMAs of the others (53 %).  Spreading the copies between the MFMAs of a single wave
// instead was slower.
// Every fold segment of the position space is a multiple of 256 samples (SegLayout), so a stage
// never straddles a fold boundary and there is no K tail.
#include <algorithm>
#include "rg_internal.h"

#define FT 256
#define FROWB 128     // bytes per row per stage (256 samples)
typedef float v16f __attribute__((ext_vector_type(16)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;

#define FP4_FLUSH_STAGES 4096  // 2^20 samples: partial sums <= 4 * 2^20 = 2^22 < 2^24

struct G4Operand { const uint8_t* base; int64_t ld; int rows; };

// one stage = 512 (or 256 when A == B) rows x 128 B; 1 KB (8 rows) per wave instruction
template <bool SAME>
__device__ __forceinline__ void g4_stage(const G4Operand& A, const G4Operand& B, int64_t kb0, uint8_t* buf) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int NPER = SAME ? 2 : 4;    // 16 waves
#pragma unroll
  for (int i = 0; i < NPER; ++i) {
    const int g = wave * NPER + i;
    const int row = g * 8 + (lane >> 3);
    const bool isA = SAME || row < FT;
    const int r = isA ? row : row - FT;
    const int rows = isA ? A.rows : B.rows;
    const int rc = r < rows ? r : rows - 1;                 // clamp: results of rows >= rows are never stored
    const int slot = (lane & 7) ^ ((row >> 1) & 7);         // global 16-byte chunk that lands in LDS slot lane&7
    const uint8_t* gp = (isA ? A.base + (int64_t)rc * A.ld : B.base + (int64_t)rc * B.ld) + kb0 + slot * 16;
    __builtin_amdgcn_global_load_lds((glb_void_t*)gp, (lds_void_t*)(buf + g * 1024), 16, 0, 0);
  }
}

// C[r][c] = (or, ATOMIC: +=) sum_k A[r][k] * B[c][k] over `nstage` stages of 128 bytes of FP4 pairs per row.
// nstage <= FP4_FLUSH_STAGES keeps every fp32 partial sum an exact integer.
template <bool SAME>
__device__ __forceinline__ void gram4_tile(G4Operand A, G4Operand B, int nstage, bool atomic, int32_t* __restrict__ C,
                                           int64_t ldc, int c_rows, int c_cols, uint8_t* smem) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wr = wave >> 1, wc = wave & 1;       // 8 waves: 4 x 2, each 64 x 128 (two waves per SIMD cover each other's
  const int l31 = lane & 31, h = lane >> 5;      // copy-issue and LDS-read phases)
  v16f acc[1][4];
#pragma unroll
  for (int i = 0; i < 1; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  uint8_t* buf0 = smem;
  uint8_t* buf1 = smem + 2 * FT * FROWB;
  g4_stage<SAME>(A, B, 0, buf0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // fragment addresses: row R -> slot (2*ks + h) ^ ((R >> 1) & 7); the rows of one lane differ by multiples of 32,
  // so the swizzle term is the same for all four fragments of an operand
  const int ra = wr * 32 + l31, rb = (SAME ? 0 : FT) + wc * 128 + l31;
  const int xa = (ra >> 1) & 7, xb = (rb >> 1) & 7;
  for (int s = 0; s < nstage; ++s) {
    uint8_t* cur = (s & 1) ? buf1 : buf0;
    uint8_t* nxt = (s & 1) ? buf0 : buf1;
    if (s + 1 < nstage) g4_stage<SAME>(A, B, (int64_t)(s + 1) * FROWB, nxt);
    const uint8_t* sa = cur + ra * FROWB;
    const uint8_t* sb = cur + rb * FROWB;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      v4i af[1], bf[4];
#pragma unroll
      for (int i = 0; i < 1; ++i)
        af[i] = *reinterpret_cast<const v4i*>(sa + i * 32 * FROWB + (((2 * ks + h) ^ xa) << 4));
#pragma unroll
      for (int j = 0; j < 4; ++j)
        bf[j] = *reinterpret_cast<const v4i*>(sb + j * 32 * FROWB + (((2 * ks + h) ^ xb) << 4));
#pragma unroll
      for (int i = 0; i < 1; ++i) {
        const v8i a8 = __builtin_shufflevector(af[i], af[i], 0, 1, 2, 3, -1, -1, -1, -1);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const v8i b8 = __builtin_shufflevector(bf[j], bf[j], 0, 1, 2, 3, -1, -1, -1, -1);
          acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc[i][j], 4, 4, 0, 0x7F7F7F7F, 0,
                                                                      0x7F7F7F7F);
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the next stage has landed in LDS
    __syncthreads();
  }
  // the fp32 sums are exact integers
#pragma unroll
  for (int i = 0; i < 1; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wr * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const int col = wc * 128 + j * 32 + l31;
        if (row < c_rows && col < c_cols) {
          int32_t* p = C + (int64_t)row * ldc + col;
          const int32_t v = __float2int_rn(acc[i][j][r]);
          if (atomic) atomicAdd(p, v);
          else *p = v;
        }
      }
}

// ---- test / generic entry: C[m][n] = A4 * B4^T; grid.z splits K into super-chunks of FP4_FLUSH_STAGES stages ----
__global__ __launch_bounds__(1024) void k_gram_fp4_generic(const uint8_t* A, int64_t lda, const uint8_t* B,
                                                          int64_t ldb, int m, int n, int64_t kbytes,
                                                          int32_t* C, int64_t ldc) {
  __shared__ __attribute__((aligned(16))) uint8_t smem[4 * FT * FROWB];
  const int tr = blockIdx.y, tc = blockIdx.x;
  const int64_t k0 = (int64_t)blockIdx.z * FP4_FLUSH_STAGES * FROWB;
  const int nstage = (int)(min(kbytes - k0, (int64_t)FP4_FLUSH_STAGES * FROWB) / FROWB);
  G4Operand a{A + (int64_t)tr * FT * lda + k0, lda, m - tr * FT};
  G4Operand b{B + (int64_t)tc * FT * ldb + k0, ldb, n - tc * FT};
  gram4_tile<false>(a, b, nstage, gridDim.z > 1, C + (int64_t)tr * FT * ldc + tc * FT, ldc, m - tr * FT, n - tc * FT,
                    smem);
}

// ---- production: one workgroup per (lower-triangular 256-tile, K super-chunk, fold, block); output into the
//      dosage x dosage quadrant of S (ld = 2*n128) --------
// Placement: the hardware deals workgroups to the 8 XCDs round-robin by LINEAR workgroup id, and each XCD has its own L2.
// The ntile tiles of one (block, fold) read the same nt operand panels (each panel by up to nt tiles: 4x re-reads at
// nt = 4 when the tiles land on different XCDs), so the grid is one-dimensional and work item w = xcd * ceil(total / 8) +
// slot (xcd = id & 7, slot = id >> 3) is decoded tile-fastest: every XCD walks its own contiguous range of (block, fold)
// groups and the tiles of a group are co-resident on it, streaming the same K window through one L2.
__global__ __launch_bounds__(1024) void k_gram_fp4_blocks(const uint8_t* pk4, int64_t pk4_ld, int64_t pk4_blk_stride,
                                                         int n128, SegLayout seg, int ntile, int nsuper, int total,
                                                         int32_t* S) {
  __shared__ __attribute__((aligned(16))) uint8_t smem[4 * FT * FROWB];
  const int per_xcd = (total + 7) >> 3;
  const int w = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
  if ((int)(blockIdx.x >> 3) >= per_xcd || w >= total) return;
  const int per_group = ntile * nsuper;
  const int t = w % per_group, grp = w / per_group;
  const int tidx = t % ntile, sup = t / ntile;
  const int f = grp % seg.nseg, blk = grp / seg.nseg;
  int tr = (int)((sqrtf(8.0f * tidx + 1.0f) - 1.0f) * 0.5f);
  while ((tr + 1) * (tr + 2) / 2 <= tidx) ++tr;
  while (tr * (tr + 1) / 2 > tidx) --tr;
  const int tc = tidx - tr * (tr + 1) / 2;
  const int64_t kbytes = seg.plen[f] / 2;
  const int64_t k0 = (int64_t)sup * FP4_FLUSH_STAGES * FROWB;
  if (k0 >= kbytes) return;
  const int nstage = (int)(min(kbytes - k0, (int64_t)FP4_FLUSH_STAGES * FROWB) / FROWB);
  const uint8_t* base = pk4 + (int64_t)blk * pk4_blk_stride + seg.pos_start[f] / 2 + k0;
  const int64_t ldS = 2 * (int64_t)n128;
  int32_t* Sf = S + ((int64_t)blk * seg.nseg + f) * ldS * ldS;
  G4Operand a{base + (int64_t)tr * FT * pk4_ld, pk4_ld, n128 - tr * FT};
  G4Operand b{base + (int64_t)tc * FT * pk4_ld, pk4_ld, n128 - tc * FT};
  int32_t* Ct = Sf + (int64_t)tr * FT * ldS + tc * FT;
  if (tr == tc) gram4_tile<true>(a, b, nstage, nsuper > 1, Ct, ldS, n128 - tr * FT, n128 - tc * FT, smem);
  else gram4_tile<false>(a, b, nstage, nsuper > 1, Ct, ldS, n128 - tr * FT, n128 - tc * FT, smem);
}

void rg_launch_gram_fp4_blocks(hipStream_t st, const uint8_t* pk4, int64_t pk4_ld, int64_t pk4_blk_stride, int nblk,
                               int n128, SegLayout seg, int32_t* S) {
  const int nt = (n128 + FT - 1) / FT;
  const int ntile = nt * (nt + 1) / 2;
  int64_t maxlen = 0;
  for (int f = 0; f < seg.nseg; ++f) maxlen = std::max(maxlen, seg.plen[f]);
  const int64_t per = (int64_t)FP4_FLUSH_STAGES * FROWB * 2;   // samples per super-chunk
  const int nsuper = (int)((maxlen + per - 1) / per);
  if (nsuper > 1) {  // K split: the super-chunks accumulate with integer atomics into a zeroed quadrant
    const int64_t ldS = 2 * (int64_t)n128;
    hipMemsetAsync(S, 0, sizeof(int32_t) * (size_t)nblk * seg.nseg * ldS * ldS, st);
  }
  const int total = ntile * nsuper * seg.nseg * nblk;
  dim3 grid((unsigned)(((total + 7) / 8) * 8));
  hipLaunchKernelGGL(k_gram_fp4_blocks, grid, dim3(1024), 0, st, pk4, pk4_ld, pk4_blk_stride, n128, seg, ntile, nsuper, total, S);
}

void rg_launch_gram_fp4_generic(hipStream_t st, const uint8_t* A, int64_t lda, const uint8_t* B, int64_t ldb, int m,
                                int n, int64_t kbytes, int32_t* C, int64_t ldc) {
  const int64_t per = (int64_t)FP4_FLUSH_STAGES * FROWB;
  const int nsuper = (int)((kbytes + per - 1) / per);
  if (nsuper > 1)
    for (int r = 0; r < m; ++r) hipMemsetAsync(C + (int64_t)r * ldc, 0, sizeof(int32_t) * n, st);
  dim3 grid((n + FT - 1) / FT, (m + FT - 1) / FT, nsuper);
  hipLaunchKernelGGL(k_gram_fp4_generic, grid, dim3(1024), 0, st, A, lda, B, ldb, m, n, kbytes, C, ldc);
}
