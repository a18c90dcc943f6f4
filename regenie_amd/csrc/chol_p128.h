// Batched Cholesky of the level-0 ridge systems, round 6: left-looking over PANELS OF 128 COLUMNS, one launch per panel.
//
// The reference solves (A - A_f + lambda_r I) beta = b - b_f through one eigendecomposition per fold (src/Step1_Models.cpp:484-505); here
// every (fold, ridge value) is its own SPD system of order <= n64 (see chol.hip), and a batch holds K * R0 * blocks of them.
//
// Why this form (rounds 2-5 ran the group-of-4-tile-columns kernels of chol.hip: update / gfact / gstrip, 0.34 of the fp64 matrix peak):
//   * the diagonal block of a group was factored by ONE WAVE per system (k_chol_gfact: 27 % of the time for 3 % of the flops), in a launch
//     of its own that the matrix cores sat out; here a diagonal block is the tail of a work item INSIDE the panel launch, started first,
//     so its pivot chains run under the matrix products of the other items;
//   * every product is LDS-staged with a 128 x 128 tile per workgroup (16 flop per byte through L2 -> LDS, the scheme of k_dgemm_nt128,
//     0.88 of the peak) instead of 64 x 64 register-fed tiles (8 flop per byte: k_chol_update sat on the L2 fabric) or 64 x 256 strips;
//   * one launch per panel (8 for order 1,024) instead of three per group (11).
//
// Launch j (j = -1 .. Tp - 2) per system b, every item one workgroup of 4 waves:
//   successor item : tile (j+1, j) as below, then the DIAGONAL block j+1:  D = X[j+1][j+1] - L[j+1][0..j] L[j+1][0..j]^T  (K = 128 (j+1), lower
//                    16-blocks only, row blocks paired w | 7-w over the waves so that each wave issues 9 of the 16 block products),
//                    factored and inverted as 2 x 2 tiles of 64 in LDS (diag_factor_lds), leaves L, the 128 x 128 inverse and the 64 x 64
//                    tile inverses of the back substitution
//   plain item i   : tile (i, j), i > j+1:  U = X[i][j] - L[i][0..j-1] L[j][0..j-1]^T  (K = 128 j),  L[i][j] = U Linv_jj^T
// A tile is formed from the source matrices (FormSrc) when it is first touched, i.e. at its own panel: read once, written once.
//
// Registers: a wave owns 32 rows x 128 columns of the tile, accumulated TRANSPOSED -- accT[n][m] = mfma(L_j rows, tile rows): lane (i, q)
// holds U[row 16m + i][columns 16n + q + 4r] -- which is the A-operand layout (k = 16n + q + 4r) of the triangular multiply that follows,
// so U never leaves the registers.  LDS: two 32 KB stage buffers; K-chunk stages = 16 k of the tile's 128 rows | the panel's 128 rows (16-byte
// slots XOR-swizzled by (row >> 1) & 7); row stages = 64 rows x 128 columns (X from the sources, then Linv), slots XOR-swizzled by row & 15.
//
// The same bits whatever the batch: an item's arithmetic depends on its system alone (no atomics, fixed order).
#pragma once
#include "chol_common.h"

#ifndef RG_HOST_EMU
#define C128_LDS_ADDR(p) ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)(p))
#define C128_STAGE_SYNC() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define c128_glds16 glds16
#define C128_RFL(x) __builtin_amdgcn_readfirstlane(x)
// barrier for LDS hazards only: __syncthreads() also waits for every outstanding GLOBAL store of the wave (s_waitcnt vmcnt(0)) -- in the
// factorization of a diagonal block, which stores L and the inverses between its LDS phases, that was a store round trip per barrier
#define C128_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
// a lane id the compiler cannot connect with the kernel's own: what a phase derives from it cannot be hoisted above the phases before it
// (hipcc computed the addresses of the LATER phases ahead of the K loop and spilled the K loop's own operands to scratch -- with an
// s_waitcnt vmcnt in front of every reload, i.e. behind the stage copies just issued)
#define C128_LAUNDER(x) asm volatile("" : "+v"(x))
// end of a K stage: the products must have been issued before the wave waits for the next stage's copies (hipcc sank half of them below the barrier)
#define C128_STAGE_SYNC_ACC(acc)                                                                                                        \
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier"                                                                            \
               : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]), "+v"(acc[2][0]), "+v"(acc[2][1]), "+v"(acc[3][0]), \
                 "+v"(acc[3][1]), "+v"(acc[4][0]), "+v"(acc[4][1]), "+v"(acc[5][0]), "+v"(acc[5][1]), "+v"(acc[6][0]), "+v"(acc[6][1]), \
                 "+v"(acc[7][0]), "+v"(acc[7][1])                                                                                       \
               :                                                                                                                        \
               : "memory")
// (the diagonal block's accumulators: row block w has no columns past block 3)
#define C128_STAGE_SYNC_ACCD(acc)                                                                                                       \
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier"                                                                            \
               : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]), "+v"(acc[2][0]), "+v"(acc[2][1]), "+v"(acc[3][0]), \
                 "+v"(acc[3][1]), "+v"(acc[4][1]), "+v"(acc[5][1]), "+v"(acc[6][1]), "+v"(acc[7][1])                                   \
               :                                                                                                                        \
               : "memory")
#else
#define C128_LAUNDER(x) ((void)0)
#define C128_LDS_BARRIER() __syncthreads()
#define C128_STAGE_SYNC_ACC(acc) C128_STAGE_SYNC()
#define C128_STAGE_SYNC_ACCD(acc) C128_STAGE_SYNC()
#endif

// K ring: units of 16 KB = 8 k of the tile's 128 rows (8 KB) | 8 k of the panel's 128 rows (8 KB), four of them, copies issued three units
// ahead (a stage of 16 k issued ONE ahead left a wave waiting for its copy after every 64 products: the tile's rows come from HBM, 2+ us,
// against 0.85 us of products -- first device run, profiles/r6_batch_sequence_first_p128.md)
#define C128_UNIT 16384
#define C128_NBUF 4
#define C128_DIST 3
#ifndef RG_HOST_EMU
template <int N>
__device__ __forceinline__ void c128_wait_upto(int n) {   // s_waitcnt vmcnt(4 * min(n, N)), immediate operands only
  if (N > 0 && n >= N) { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * N) : "memory"); return; }
  if (N > 0) c128_wait_upto<(N > 0 ? N - 1 : 0)>(n);
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
// unit u has landed (at most `later` units issued after it may still be in flight) and every wave is done with the unit before it
#define C128_UNIT_SYNC(later)                                          \
  do {                                                                 \
    c128_wait_upto<C128_DIST - 1>(later);                              \
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");    \
  } while (0)
#else
#define C128_UNIT_SYNC(later) __syncthreads()
#endif
#define C128_STAGE 32768
#define C128_LDS (2 * C128_STAGE + 4096)      // the diagonal items alias sA | sB (64 x 66 doubles each) + dv (64) onto the stage buffers

struct C128Args {
  double* mats; int64_t mat_stride; int n64;
  double* linv;         // [batch][Tp][128 * 128]: inverse of the diagonal block of a panel, row-major
  double* dinv;         // [batch][n64 / 64][64 * 64]: tile inverses for the back substitution (k_chol_backsolve*)
  int32_t* info;
  int j, Tp, batch, R, nplain;
  int flags;            // diagnostics (RG_C128_FLAGS): timing only, wrong results: 2 = every source row block is read from
                        // the system's first 16 rows, 4 = no products in the triangular multiply, 8 = the two tile factorizations skipped,
                        // 16 = every K unit of a tile re-reads the tile's first (is the product loop waiting for memory?)
  unsigned long long* dbg;      // RG_C128_DBG=1: per-phase time sums (100 MHz ticks of s_memrealtime, thread 0 of every workgroup); else nullptr
  FormSrc fs;
};
#ifndef RG_HOST_EMU
// (the sums are kept in the last 1.5 KB of the workgroup's LDS and flushed with the workgroup's last instructions: a global atomic per phase
// made the NEXT phase wait for it at its first vmcnt(0) and showed up there -- 155 -> 420 thousand workgroup-us in "L21/M1/D22/save")
#define C128_T(k)                                                                        \
  do {                                                                                   \
    if (a.dbg && threadIdx.x == 0) {                                                     \
      const unsigned long long t_ = __builtin_amdgcn_s_memrealtime();                    \
      tl[k] += t_ - t_prev;                                                              \
      t_prev = t_;                                                                       \
    }                                                                                    \
  } while (0)
#define C128_TFLUSH()                                                                    \
  do {                                                                                   \
    if (a.dbg && threadIdx.x == 0)                                                       \
      for (int k_ = 1; k_ < 16; ++k_)                                                    \
        if (tl[k_]) atomicAdd(&a.dbg[k_], tl[k_]);                                       \
  } while (0)
#define C128_T0() (a.dbg ? __builtin_amdgcn_s_memrealtime() : 0ull)
#else
#define C128_T(k) ((void)0)
#define C128_TFLUSH() ((void)0)
#define C128_T0() 0ull
#endif

// ---- blocked (16) factorization + inverse of the 64x64 tile held in LDS (diag_factor_lds of chol_common.h with a straight-line 16 x 16 step) ----
// On return: lower triangle + diagonal of s = L, strict upper triangle = Linv^T, dv[r] = Linv[r][r].
// Returns true (in some thread) when a pivot was not positive.
__device__ __forceinline__ bool c128_factor64(double (&s)[CT][CT + 2], double (&dv)[CT], unsigned long long* dbg = nullptr) {
#ifndef RG_HOST_EMU
#define C128_FT(k) do { if (dbg && threadIdx.x == 0) { const unsigned long long t_ = __builtin_amdgcn_s_memrealtime(); dbg[k] += t_ - tf; tf = t_; } } while (0)      // dbg: the workgroup's LDS sums
  unsigned long long tf = dbg ? __builtin_amdgcn_s_memrealtime() : 0ull;
#else
#define C128_FT(k) ((void)0)
#endif
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lq = lane >> 4;
  // ---- blocked (16) factorization + inverse, the 16x16x16 block products on the fp64 MFMA ---------------
  // MFMA 16x16x4: lane (i = lane&15, q = lane>>4) supplies A[i][kk], B[kk][i] and owns D[q + 4r][i], r = 0..3;
  // a K = 16 block product is 4 instructions with kk(q, s) chosen per product (any permutation of K is fine
  // as long as A and B use the same one).
  // element (r, c) of the inverse of a DIAGONAL 16-block at offset o (0 above the diagonal)
  auto inv_diag = [&](int o, int r, int c) -> double {
    return (c < r) ? s[o + c][o + r] : ((c == r) ? dv[o + r] : 0.0);
  };
  bool bad = false;
  for (int sb = 0; sb < 4; ++sb) {
    const int o = sb * 16;
    const int nb = 3 - sb;   // 16-row blocks below the diagonal block
    // (i) diagonal block: lanes 0..15 of wave 0 hold one row each in registers; then its inverse, one column each
    if (tid < 16) {
      // Straight-line code (round 6): the entries above the diagonal of the 16 x 16 block are don't-cares, so every update runs in all 16
      // lanes without an exec mask per statement, a bad pivot is flagged without a branch, and the inverse is accumulated right-looking
      // (16 independent chains instead of one dependent chain of r multiply-adds per entry): 13 instead of 32 us per tile.
      // Cross-lane values are broadcast with v_readlane (compile-time lane index, no LDS round trip); square root and reciprocal come from
      // one v_rsq_f64 + two Newton steps (~1 ulp).
      double a[16], rdv[16];
#pragma unroll
      for (int c = 0; c < 16; ++c) a[c] = s[o + tid][o + c];
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const double p0 = bcast_lane(a[c], c);
        const bool ok = p0 > 0.0;
        bad |= !ok;
        const double piv = ok ? p0 : 1.0;
        double r0 = __builtin_amdgcn_rsq(piv);
        r0 = r0 * fma(-0.5 * piv * r0, r0, 1.5);
        r0 = r0 * fma(-0.5 * piv * r0, r0, 1.5);
        double d = piv * r0;
        d = fma(0.5 * r0, fma(-d, d, piv), d);     // sqrt(piv)
        const double rd = fma(r0, fma(-d, r0, 1.0), r0);        // 1 / sqrt(piv)
        rdv[c] = rd;
        a[c] = tid == c ? d : a[c] * rd;
#pragma unroll
        for (int c2 = c + 1; c2 < 16; ++c2) a[c2] = fma(-a[c], bcast_lane(a[c], c2), a[c2]);      // L[c2][c] from lane c2
      }
#pragma unroll
      for (int c = 0; c < 16; ++c)
        if (c <= tid) s[o + tid][o + c] = a[c];
      // inverse of the 16x16 triangle, lane = column: x[j] = v[j] / L[j][j], then v[r] -= L[r][j] x[j] for the rows below
      double v[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = (r == tid) ? 1.0 : 0.0;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        v[j] *= rdv[j];
#pragma unroll
        for (int r = j + 1; r < 16; ++r) v[r] = fma(-bcast_lane(a[j], r), v[j], v[r]);      // L[r][j] from lane r
      }
      dv[o + tid] = v[tid];
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (r > tid) s[o + tid][o + r] = v[r];   // Linv[r][tid], transposed into the upper triangle
    }
    C128_FT(11);
    C128_LDS_BARRIER();
    C128_FT(14);
    // (ii) rows below: L21 = A21 * Linv11^T, one 16-row block per wave
    if (wave < nb) {
      const int rb = o + 16 + 16 * wave;
      v4d acc = (v4d){0, 0, 0, 0};
#pragma unroll
      for (int st = 0; st < 4; ++st)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(s[rb + li][o + 4 * lq + st], inv_diag(o, li, 4 * lq + st), acc, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) s[rb + lq + 4 * r][o + li] = acc[r];
    }
    C128_LDS_BARRIER();
    // (iii) trailing update inside the tile: A22 -= L21 L21^T (lower blocks; only the lower triangle of the
    //       diagonal blocks is written -- their upper triangle will hold the inverse), blocks dealt to the waves
    {
      const int nblk2 = nb * (nb + 1) / 2;
      for (int idx = wave; idx < nblk2; idx += 4) {
        int bi = 0, rem = idx;
        while (rem > bi) { rem -= bi + 1; ++bi; }
        const int bj = rem;
        const int ri = o + 16 + 16 * bi, rj = o + 16 + 16 * bj;
        v4d acc = (v4d){0, 0, 0, 0};
#pragma unroll
        for (int st = 0; st < 4; ++st)
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(s[ri + li][o + 4 * lq + st], s[rj + li][o + 4 * lq + st], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (bi != bj || li <= lq + 4 * r) s[ri + lq + 4 * r][rj + li] -= acc[r];
      }
    }
    C128_LDS_BARRIER();
    C128_FT(12);
  }
  C128_FT(12);
  // off-diagonal blocks of the inverse (i > j), by sub-diagonal distance:
  //   Linv[i][j] = -Linv[i][i] * sum_{kb=j}^{i-1} L[i][kb] Linv[kb][j]      (Linv[kb][j] at s[16j + .][16kb + .]^T)
  for (int dist = 1; dist < 4; ++dist) {
    const int j = wave, ib = wave + dist;
    if (ib < 4) {
      v4d m1 = (v4d){0, 0, 0, 0};
      for (int kb = j; kb < ib; ++kb) {
#pragma unroll
        for (int st = 0; st < 4; ++st) {
          const int kk = 4 * lq + st;
          const double bval = (kb == j) ? inv_diag(16 * j, kk, li) : s[16 * j + li][16 * kb + kk];
          m1 = __builtin_amdgcn_mfma_f64_16x16x4f64(s[16 * ib + li][16 * kb + kk], bval, m1, 0, 0, 0);
        }
      }
      // second product with kk(q, st) = q + 4 st: the B operand M1[kk][li] is exactly register st of m1
      v4d acc = (v4d){0, 0, 0, 0};
#pragma unroll
      for (int st = 0; st < 4; ++st)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(inv_diag(16 * ib, li, lq + 4 * st), m1[st], acc, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) s[16 * j + li][16 * ib + lq + 4 * r] = -acc[r];
    }
    C128_LDS_BARRIER();
  }
  C128_FT(13);
  return bad;
}

// ---- one ring unit: accT[n][m] += L_j rows(n) x tile rows(m)^T over 8 k -----------------------------------------------------------------
// rows of 64 bytes, the four 16-byte slots XOR-swizzled by f((row >> 2) & 3), f = (0, 3, 2, 1): every ds_read_b128 lane group then falls on
// 16 distinct bank quads.  Lane (i, q) takes slot q: k = 2q, 2q + 1.
// DIAG: both operands are the panel's own rows; wave W issues the blocks n <= W (m = 0: row block W) and n <= 7 - W (m = 1: row block 7 - W).
template <bool DIAG>
__device__ __forceinline__ void c128_kunit(const uint8_t* cur, int aoff0, int aoff1, int boff, int sq, v4d (&acc)[8][2], int W) {
  double2 a[2], b[8];
  a[0] = *reinterpret_cast<const double2*>(cur + aoff0 + sq);
  a[1] = *reinterpret_cast<const double2*>(cur + aoff1 + sq);
#pragma unroll
  for (int n = 0; n < 8; ++n) b[n] = *reinterpret_cast<const double2*>(cur + boff + n * 1024 + sq);
#pragma unroll
  for (int kk = 0; kk < 2; ++kk)
#pragma unroll
    for (int n = 0; n < 8; ++n)
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        if (DIAG && (m == 0 ? n > 3 : false)) continue;                 // row blocks 0-3 have no columns past block 3
        if (DIAG && n > (m == 0 ? W : 7 - W)) continue;                  // wave-uniform: a scalar branch around the product
        acc[n][m] = __builtin_amdgcn_mfma_f64_16x16x4f64(kk == 0 ? b[n].x : b[n].y, kk == 0 ? a[m].x : a[m].y, acc[n][m], 0, 0, 0);
      }
}

// ---- stage copies (direct global -> LDS, 16 bytes per lane) -------------------------------------------------------------------------------
// ring unit: waves 0, 1 copy the A half (8 k of 128 rows from ap), waves 2, 3 the B half (from bp); piece p of a wave = rows 64 (w & 1) + 16 p .. + 15,
// lane l -> row + (l >> 2), slot (l & 3); the swizzle f((row >> 2) & 3) = f(l >> 4) is the same for every piece: one lane offset, the rest scalar
__device__ __forceinline__ void c128_issue_unit(const double* ap, const double* bp, int buf, int wave, int n64, uint32_t lds0, uint32_t uoff) {
  const double* base = (wave < 2 ? ap : bp) + (int64_t)(64 * (wave & 1)) * n64;
  const uint32_t dst = lds0 + buf * C128_UNIT + wave * 4096;
#pragma unroll
  for (int p = 0; p < 4; ++p) c128_glds16(base + (int64_t)(16 * p) * n64, uoff, dst + p * 1024);
}
__device__ __forceinline__ uint32_t c128_unit_off(int lane, int n64) {
  const int slot = (lane & 3) ^ ((4 - (lane >> 4)) & 3);
  return (uint32_t)(((int64_t)(lane >> 2) * n64 + 2 * slot) * 8);
}
// row stage: 64 rows x 128 doubles (1 KB per row), wave w copies rows 16 w .. 16 w + 15; lane l of row p takes the 16-byte slot l ^ p
__device__ __forceinline__ void c128_issue_rows(const double* src, int64_t ld, int wave, int lane, uint32_t lds0) {
  const double* base = src + (int64_t)(16 * wave) * ld;
  const uint32_t dst = lds0 + wave * 16384;
#pragma unroll
  for (int p = 0; p < 16; ++p) c128_glds16(base + (int64_t)p * ld, (uint32_t)((lane ^ p) << 4), dst + p * 1024);
}
// ring unit of 16 rows x 128 doubles (1 KB per row): wave w copies rows 4 w .. 4 w + 3; lane l of row p takes the 16-byte slot l ^ p
__device__ __forceinline__ void c128_issue_rows16(const double* src, int64_t ld, int buf, int wave, int lane, uint32_t lds0) {
  const uint32_t dst = lds0 + buf * C128_UNIT + wave * 4096;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int row = 4 * wave + p;
    c128_glds16(src + (int64_t)row * ld, (uint32_t)((lane ^ row) << 4), dst + p * 1024);
  }
}
// ring unit of 128 rows x 16 doubles (the source's column block of a tile; 128 bytes per row, slots XOR-swizzled by (row >> 1) & 7): wave w
// copies ITS OWN rows 32 w .. 32 w + 31 (4 pieces of 8 rows); the swizzle of row 8 p + (lane >> 3) depends on the parity of p alone
__device__ __forceinline__ void c128_issue_cols16(const double* src, int buf, int wave, int n64, uint32_t lds0, const uint32_t (&xoff)[2]) {
  const double* base = src + (int64_t)(32 * wave) * n64;
  const uint32_t dst = lds0 + buf * C128_UNIT + wave * 4096;
#pragma unroll
  for (int p = 0; p < 4; ++p) c128_glds16(base + (int64_t)(8 * p) * n64, xoff[p & 1], dst + p * 1024);
}
__device__ __forceinline__ void c128_cols_off(int lane, int n64, uint32_t (&xoff)[2]) {
#pragma unroll
  for (int p = 0; p < 2; ++p) xoff[p] = (uint32_t)(((int64_t)(lane >> 3) * n64 + 2 * ((lane & 7) ^ ((4 * p + (lane >> 4)) & 7))) * 8);
}
// element (row, column q + 4 r) of such a unit; row = 16 blk + i
__device__ __forceinline__ double c128_cols16_at(const uint8_t* cur, int row, int r, int i, int q) {
  return *reinterpret_cast<const double*>(cur + row * 128 + ((((q >> 1) + 2 * r) ^ ((i >> 1) & 7)) << 4) + 8 * (q & 1));
}
// element (row i, column 16 n + q + 4 r) of such a unit
__device__ __forceinline__ double c128_rows16_at(const uint8_t* cur, int n, int r, int i, int q) {
  return *reinterpret_cast<const double*>(cur + i * 1024 + (((8 * n + 2 * r + (q >> 1)) ^ i) << 4) + 8 * (q & 1));
}
// element (row block rbl of the stage, row i, column 16 n + q + 4 r) of a row stage
__device__ __forceinline__ double c128_rows_at(const uint8_t* smem, int rbl, int n, int r, int i, int q) {
  return *reinterpret_cast<const double*>(smem + (16 * rbl + i) * 1024 + (((8 * n + 2 * r + (q >> 1)) ^ i) << 4) + 8 * (q & 1));
}

// MODE 0: launch j = -1 (diagonal block 0 alone: no tile, no products); 1: j = 0 (tiles of panel 0: no products before the triangular multiply);
// 2: j >= 1.  Compile-time so that the K loops have no bypass path (with one, hipcc kept a second copy of the 128 accumulator registers
// alive across the loop and spilled the loop's own operands).
template <int MODE>
__global__ __launch_bounds__(256, 2) void k_c128_panel(C128Args a) {
  __shared__ __attribute__((aligned(16))) uint8_t smem[C128_LDS];
  const unsigned long long t_start = C128_T0();
  const int wave = C128_RFL((int)threadIdx.x >> 6);
  int b, g;
  bool succ;
  {
    const int wg = blockIdx.x;
    const int nsucc = (int)xcd_affine_grid(1, a.batch, a.R);
    if (wg < nsucc) {
      succ = true;
      if (!xcd_affine(wg, 1, a.batch, a.R, b, g)) return;
    } else {
      succ = false;
      if (!xcd_affine(wg - nsucc, a.nplain, a.batch, a.R, b, g)) return;
    }
  }
  b = C128_RFL(b);
  g = C128_RFL(g);
  const FormIdx fx = form_idx(a.fs, b);
  const int nb = fx.n, nrhs = fx.nrhs;                 // order of the matrix; + the embedded right-hand-side rows
  const int n64 = a.n64;
  int tpb = a.Tp;                                      // panels of this system that hold data (the rest is identity padding: never touched)
  if (a.fs.skip_pad && a.fs.d_n) { const int t = (nrhs + 127) >> 7; tpb = t < tpb ? t : tpb; }
  const int j = a.j;
  const int it = succ ? j + 1 : j + 2 + g;
  if (it >= tpb) return;
  double* M = a.mats + (int64_t)b * a.mat_stride;
  const uint32_t lds0 = C128_LDS_ADDR(smem);
  unsigned long long t_prev = C128_T0();
  unsigned long long* tl = reinterpret_cast<unsigned long long*>(smem + C128_LDS - 1024);      // [16] phase sums of this workgroup (diagnostic)
  if (a.dbg && threadIdx.x == 0) {
    for (int k_ = 0; k_ < 16; ++k_) tl[k_] = 0;
    atomicAdd(&a.dbg[succ ? 17 : 16], 1ull);
    atomicAdd(&a.dbg[19], t_prev - t_start);
#ifndef RG_HOST_EMU
    // how long a slot of this CU stood empty before this workgroup started: its start against the latest end recorded for the CU
    // (a.dbg[64 + CU], CU = XCC id : SE : SH : CU of the hardware id registers); gaps above 50 us (launch boundaries) are not counted
    const unsigned cu = ((unsigned)__builtin_amdgcn_s_getreg((3 << 11) | 20) << 8) | (((unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4) >> 8) & 0xFFu);
    tl[15] = cu;
    const unsigned long long le = __hip_atomic_load(&a.dbg[64 + cu], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (le && t_start > le && t_start - le < 5000) { atomicAdd(&a.dbg[56], t_start - le); atomicAdd(&a.dbg[57], 1ull); }
#endif
  }

  v4d acc[8][2];
#pragma unroll
  for (int n = 0; n < 8; ++n)
#pragma unroll
    for (int m = 0; m < 2; ++m) acc[n][m] = (v4d){0, 0, 0, 0};

  // ======== part 1: tile (it, j) = (X - L[it][0..j-1] L[j][0..j-1]^T) Linv_jj^T ==========================================================
  // ONE ring pipeline per tile, copies three units ahead throughout: 16 j K units, then the tile's 8 row blocks of the source(s) (X: 16 rows x
  // 128 columns each, consumed by the wave that owns the rows), then the 8 row blocks of Linv (the triangular multiply, block cb of the
  // result stored under the products of block cb + 1).  (First build: X and Linv as 64 KB stages with nothing in flight behind them --
  // 17 + 24 us per tile next to 47 us of K units, RG_C128_DBG=1.)
  const int nsrc = fx.F ? 2 : 1;
  if (MODE >= 1) {
    const int rb0 = 32 * wave;                          // this wave's rows of the tile: rb0 + 16 m + i
    const int nK = MODE >= 2 ? 16 * j : 0, nX = 8 * nsrc, total = nK + nX + 8;
    const int kpg = nK / nX;                            // K units in front of every source row block: the X copies (HBM, and slow next to the other
    const double* arows = M + (int64_t)(128 * it) * n64;   // workgroup's K stream: 2.5 us per unit) land under this tile's own products
    const double* brows = M + (int64_t)(128 * j) * n64;
    const double* Li = a.linv + ((int64_t)b * a.Tp + j) * (128 * 128);
    int lane_k = threadIdx.x & 63;
    C128_LAUNDER(lane_k);
    const uint32_t uoff = c128_unit_off(lane_k, n64);
    uint32_t xoff[2];
    c128_cols_off(lane_k, n64, xoff);
    // issue cursor (wave-uniform): kpg K units, one source row block, ... then the remaining K units, then the row blocks of Linv
    int is_n = 0, is_k = 0, is_x = 0, is_c = 0;
    auto issue_next = [&]() {
      if (is_n >= total) return;
      const int buf = is_n & (C128_NBUF - 1);
      if (is_x < nX && (is_c == kpg || is_k == nK)) {
        c128_issue_cols16((is_x >= 8 ? fx.F : fx.S) + ((a.flags & 2) ? 0 : (int64_t)(128 * it) * n64 + 128 * j + 16 * (is_x & 7)), buf, wave, n64, lds0, xoff);
        ++is_x; is_c = 0;
      } else if (is_k < nK) {
        c128_issue_unit(arows + ((a.flags & 16) ? 0 : 8 * is_k), brows + ((a.flags & 16) ? 0 : 8 * is_k), buf, wave, n64, lds0, uoff);
        ++is_k; ++is_c;
      } else {
        c128_issue_rows16(Li + (int64_t)(16 * (is_n - nK - nX)) * 128, 128, buf, wave, lane_k, lds0);
      }
      ++is_n;
    };
    int u = 0, kc = 0;
    const int ik = lane_k & 15, qk = lane_k >> 4;
    const int sq = (qk ^ ((4 - (ik >> 2)) & 3)) << 4;
    const int aoff0 = (rb0 + ik) * 64, aoff1 = (rb0 + 16 + ik) * 64, boff = 8192 + ik * 64;
    auto k_unit = [&]() {
      C128_UNIT_SYNC(total - 1 - u);
      issue_next();
      c128_kunit<false>(smem + (u & (C128_NBUF - 1)) * C128_UNIT, aoff0, aoff1, boff, sq, acc, 0);
      ++u; ++kc;
    };
#pragma unroll
    for (int p = 0; p < C128_DIST; ++p) issue_next();      // total >= 16
#pragma unroll 1
    for (int src = 0; src < nsrc; ++src) {
      const double sgn = src ? 1.0 : -1.0;
#pragma unroll
      for (int rb = 0; rb < 8; ++rb) {
        if (MODE >= 2) {
#pragma unroll 1
          for (int t = 0; t < kpg; ++t) k_unit();
        }
        int lane = threadIdx.x & 63;
        C128_LAUNDER(lane);       // per unit: the element masks below must not be hoisted out (they were, as spilled SGPR pairs)
        const int i = lane & 15, q = lane >> 4;
        C128_UNIT_SYNC(total - 1 - u);
        issue_next();
        {   // unit rb = the tile's column block rb (128 rows x 16 columns): every lane takes its two rows' four columns q + 4 r
          const uint8_t* cur = smem + (u & (C128_NBUF - 1)) * C128_UNIT;
#pragma unroll
          for (int m = 0; m < 2; ++m) {
            // element (gi, gj) holds data when gj < nb and gi < nrhs: one limit per lane on the column offset 16 n + 4 r (+ q)
            const int gi = 128 * it + rb0 + 16 * m + i;
            const int lim = gi < nrhs ? nb - 128 * j - q : -1;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              double x = c128_cols16_at(cur, rb0 + 16 * m + i, r, i, q);
              x = (16 * rb + 4 * r < lim) ? x : 0.0;
              acc[rb][m][r] = fma(sgn, x, acc[rb][m][r]);
            }
          }
        }
        ++u;
      }
    }
    if (MODE >= 2) {
#pragma unroll 1
      while (kc < nK) k_unit();
    }
    C128_T(1);
    {
      // ---- T = U Linv^T by row blocks cb of Linv: out[cb] = sum_{n <= cb} U[.][16n..] Linv[16cb..][16n..]^T
      int lane = threadIdx.x & 63;
      C128_LAUNDER(lane);
      const int i = lane & 15, q = lane >> 4;
      double* Trow = M + (int64_t)(128 * it + rb0 + q) * n64 + 128 * j + i;      // element (row rb0 + q, column i) of the tile
      v4d pend[2];                                        // the previous block's results: stored under the next block's products
      int pend_cb = -1;
      auto store_pend = [&]() {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int r = 0; r < 4; ++r) Trow[(int64_t)(16 * m + 4 * r) * n64 + 16 * pend_cb] = pend[m][r];
      };
#pragma unroll
      for (int cb = 0; cb < 8; ++cb) {
        C128_UNIT_SYNC(total - 1 - u);
        issue_next();
        const uint8_t* cur = smem + (u & (C128_NBUF - 1)) * C128_UNIT;
        if (pend_cb >= 0) store_pend();
        v4d out[2];
        out[0] = (v4d){0, 0, 0, 0};
        out[1] = (v4d){0, 0, 0, 0};
#pragma unroll
        for (int n = 0; n < 8; ++n) {
          if (n > cb || (a.flags & 4)) continue;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const double lv = c128_rows16_at(cur, n, r, i, q);
            out[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(-acc[n][0][r], lv, out[0], 0, 0, 0);
            out[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(-acc[n][1][r], lv, out[1], 0, 0, 0);
          }
        }
        pend[0] = out[0];
        pend[1] = out[1];
        pend_cb = cb;
        ++u;
      }
      store_pend();
    }
    C128_T(3);
    if (!succ) {
      C128_TFLUSH();
      if (a.dbg && threadIdx.x == 0) { atomicAdd(&a.dbg[18], t_prev - t_start); atomicAdd(&a.dbg[21 + j], t_prev - t_start); atomicMin(&a.dbg[33 + j], t_start); atomicMax(&a.dbg[45 + j], t_prev); atomicMax(&a.dbg[64 + tl[15]], C128_T0()); }
      return;
    }
    __threadfence_block();                              // the diagonal block below reads this tile back: same workgroup, same L2 (an agent-scope
    C128_STAGE_SYNC();                                  // fence wrote the XCD's whole L2 back: 25 us per item)
#pragma unroll
    for (int n = 0; n < 8; ++n)
#pragma unroll
      for (int m = 0; m < 2; ++m) acc[n][m] = (v4d){0, 0, 0, 0};
  }

  // ======== part 2 (successor item): diagonal block jd = j + 1 ============================================================================
  // row blocks of 16: wave w holds block w (m = 0, columns n <= w) and block 7 - w (m = 1, columns n <= 7 - w).  The same ring: 8 jd K units
  // of 16 k (the A half holds 8 k of the panel's rows, the B half the next 8 k of the SAME rows), then the 8 row blocks of the source(s).
  const int jd = j + 1;
  {
    const int nK = MODE >= 1 ? 8 * jd : 0, nX = 8 * nsrc, total = nK + nX;
    const int kpg = nK / nX > 0 ? nK / nX : (nK > 0 ? 1 : 0);      // K units (16 k each) in front of every source row block
    const double* arows = M + (int64_t)(128 * jd) * n64;
    int lane_k = threadIdx.x & 63;
    C128_LAUNDER(lane_k);
    const uint32_t uoff = c128_unit_off(lane_k, n64);
    uint32_t xoff[2];
    c128_cols_off(lane_k, n64, xoff);
    int is_n = 0, is_k = 0, is_x = 0, is_c = 0;
    auto issue_next = [&]() {
      if (is_n >= total) return;
      const int buf = is_n & (C128_NBUF - 1);
      if (is_x < nX && (is_c == kpg || is_k == nK)) {
        c128_issue_cols16((is_x >= 8 ? fx.F : fx.S) + ((a.flags & 2) ? 0 : (int64_t)(128 * jd) * n64 + 128 * jd + 16 * (is_x & 7)), buf, wave, n64, lds0, xoff);
        ++is_x; is_c = 0;
      } else {
        c128_issue_unit(arows + ((a.flags & 16) ? 0 : 16 * is_k), arows + ((a.flags & 16) ? 0 : 16 * is_k) + 8, buf, wave, n64, lds0, uoff);
        ++is_k; ++is_c;
      }
      ++is_n;
    };
    int u = 0, kc = 0;
    const int ik = lane_k & 15, qk = lane_k >> 4;
    const int sq = (qk ^ ((4 - (ik >> 2)) & 3)) << 4;
    const int aoff0 = (16 * wave + ik) * 64, aoff1 = (16 * (7 - wave) + ik) * 64, boff = ik * 64;
    auto k_unit = [&]() {
      C128_UNIT_SYNC(total - 1 - u);
      issue_next();
      const uint8_t* cur = smem + (u & (C128_NBUF - 1)) * C128_UNIT;
      c128_kunit<true>(cur, aoff0, aoff1, boff, sq, acc, wave);
      c128_kunit<true>(cur + 8192, aoff0, aoff1, boff, sq, acc, wave);
      ++u; ++kc;
    };
#pragma unroll
    for (int p = 0; p < C128_DIST; ++p) issue_next();      // total >= 8
    const double sh = fx.sh;
#pragma unroll 1
    for (int src = 0; src < nsrc; ++src) {
#pragma unroll
      for (int rb = 0; rb < 8; ++rb) {
        if (MODE >= 1) {
#pragma unroll 1
          for (int t = 0; t < kpg && kc < nK; ++t) k_unit();
        }
        int lane = threadIdx.x & 63;
        C128_LAUNDER(lane);
        const int i = lane & 15, q = lane >> 4;
        C128_UNIT_SYNC(total - 1 - u);
        issue_next();
        {   // unit rb = column block rb of the diagonal block: this wave's row blocks w (m = 0) and 7 - w (m = 1) where they lie on or below it
          const uint8_t* cur = smem + (u & (C128_NBUF - 1)) * C128_UNIT;
#pragma unroll
          for (int m = 0; m < 2; ++m) {
            if (m == 0 && rb > 3) continue;                            // row blocks 0-3 have no columns past block 3
            const int rbk = m == 0 ? wave : 7 - wave;
            if (rb > rbk) continue;                                    // wave-uniform: above the diagonal
            const int gi = 128 * jd + 16 * rbk + i;
            const int lim = gi < nrhs ? nb - 128 * jd - q : -1;        // off the diagonal: data when 16 n + 4 r < lim
            const int de = 16 * rbk + i - q;                           // the diagonal element sits at 16 n + 4 r == de
            // on the diagonal: + shift inside the matrix, 2^100 on an embedded right-hand-side row, 1 in the identity padding
            const double dadd = gi < nb ? sh : (gi < nrhs ? RG_EMBED_DIAG : 1.0), dmul = gi < nb ? 1.0 : 0.0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              double x = c128_cols16_at(cur, 16 * rbk + i, r, i, q);
              x = (16 * rb + 4 * r < lim) ? x : 0.0;
              if (src == 0) {
                x = (16 * rb + 4 * r == de) ? fma(x, dmul, dadd) : x;
                acc[rb][m][r] -= x;
              } else {
                acc[rb][m][r] += x;
              }
            }
          }
        }
        ++u;
      }
    }
    if (MODE >= 1) {
#pragma unroll 1
      while (kc < nK) k_unit();
    }
    C128_STAGE_SYNC();
  }
  C128_T(4);
  // ---- factor the 128 x 128 block as 2 x 2 tiles of 64 (acc = -D: lane (i, q) holds -D[16 rb + i][16 n + q + 4 r]) ------------------------
  int tid = threadIdx.x;
  C128_LAUNDER(tid);
  const int lane = tid & 63;
  const int i = lane & 15, q = lane >> 4;
  double (&sA)[CT][CT + 2] = *reinterpret_cast<double (*)[CT][CT + 2]>(smem);
  double (&sB)[CT][CT + 2] = *reinterpret_cast<double (*)[CT][CT + 2]>(smem + CT * (CT + 2) * 8);
  double (&dv)[CT] = *reinterpret_cast<double (*)[CT]>(smem + 2 * CT * (CT + 2) * 8);
  const int li = i, lq = q;
  bool bad = false;
  double* Dg = M + (int64_t)(128 * jd) * n64 + 128 * jd;                 // the block in the system
  double* Lv = a.linv + ((int64_t)b * a.Tp + jd) * (128 * 128);          // its inverse
  double* Iv = a.dinv + ((int64_t)b * (n64 / CT) + 2 * jd) * (CT * CT);  // the two tile inverses
  auto inv_at = [&](int r, int c) -> double {          // Linv[r][c] of the tile factored in sA (0 above the diagonal)
    return sA[c][r] * (c < r ? 1.0 : 0.0) + dv[r] * (c == r ? 1.0 : 0.0);
  };
  // tile of 64 held in sA after diag_factor_lds -> L into the system, Linv into the 128-inverse at (o, o) and into tile inverse t
  auto save_tile = [&](int o, int t) {
    for (int e = tid; e < CT * CT; e += 256) {
      const int r = e >> 6, c = e & 63;
      if (c <= r) Dg[(int64_t)(o + r) * n64 + o + c] = sA[r][c];
      const double lv = (c < r) ? sA[c][r] : (c == r ? dv[r] : 0.0);
      Lv[(o + r) * 128 + o + c] = lv;
      Iv[(int64_t)t * CT * CT + e] = lv;
    }
  };
  // D11 -> sA (rows 16 w + i of wave w, m = 0)
#pragma unroll
  for (int n = 0; n < 4; ++n)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int rr = 16 * wave + i, cc = 16 * n + q + 4 * r;
      sA[rr][cc] = (cc <= rr) ? -acc[n][0][r] : 0.0;
    }
  C128_LDS_BARRIER();
  C128_T(6);
  if (!(a.flags & 8)) bad |= c128_factor64(sA, dv, a.dbg ? tl : nullptr);
  C128_T(7);
  // L21 = D21 Linv11^T: this wave's row block 7 - w of the block = rows 16 (3 - w) of the lower half
  const int lrb = 3 - wave;
  {
    v4d out[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
      out[cb] = (v4d){0, 0, 0, 0};
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        if (n > cb) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r)
          out[cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(-acc[n][1][r], inv_at(16 * cb + li, 16 * n + lq + 4 * r), out[cb], 0, 0, 0);
      }
    }
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        sB[16 * lrb + lq + 4 * r][16 * cb + li] = out[cb][r];
        Dg[(int64_t)(64 + 16 * lrb + lq + 4 * r) * n64 + 16 * cb + li] = out[cb][r];
      }
  }
  C128_LDS_BARRIER();
  // D22 -= L21 L21^T (blocks n - 4 <= 3 - w of this wave's row block), and M1 = L21 Linv11 (this wave's 16 rows), both from sA / sB
  v4d m1[4];
  {
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) m1[cb] = (v4d){0, 0, 0, 0};
#pragma unroll 4
    for (int ks = 0; ks < 16; ++ks) {
      const int k = 4 * ks + lq;
      const double av = sB[16 * lrb + li][k];
#pragma unroll
      for (int nn = 0; nn < 4; ++nn) {
        if (nn <= lrb) acc[4 + nn][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(sB[16 * nn + li][k], av, acc[4 + nn][1], 0, 0, 0);
        // M1[.][16 nn + li] = sum_k L21[.][k] Linv11[k][16 nn + li]: Linv11[k][c] is sA[c][k] below the diagonal of the tile
        const int c = 16 * nn + li;
        const double bv = sA[c][k] * (c < k ? 1.0 : 0.0) + dv[k] * (c == k ? 1.0 : 0.0);
        m1[nn] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, m1[nn], 0, 0, 0);
      }
    }
  }
  C128_LDS_BARRIER();
  save_tile(0, 0);
  C128_LDS_BARRIER();
  // M1 -> sB (L21 is no longer needed), D22 -> sA
#pragma unroll
  for (int cb = 0; cb < 4; ++cb)
#pragma unroll
    for (int r = 0; r < 4; ++r) sB[16 * lrb + lq + 4 * r][16 * cb + li] = m1[cb][r];
#pragma unroll
  for (int nn = 0; nn < 4; ++nn)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int rr = 16 * lrb + i, cc = 16 * nn + q + 4 * r;
      sA[rr][cc] = (cc <= rr) ? -acc[4 + nn][1][r] : 0.0;
    }
  C128_LDS_BARRIER();
  C128_T(8);
  if (!(a.flags & 8)) bad |= c128_factor64(sA, dv, a.dbg ? tl : nullptr);
  C128_T(9);
  // I21 = -Linv22 M1: wave w takes rows 16 w .. of the lower half
  {
    v4d o[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) o[cb] = (v4d){0, 0, 0, 0};
#pragma unroll 4
    for (int ks = 0; ks < 16; ++ks) {
      const int k = 4 * ks + lq;
      const int rr = 16 * wave + li;
      const double av = sA[k][rr] * (k < rr ? 1.0 : 0.0) + dv[rr] * (k == rr ? 1.0 : 0.0);      // Linv22[rr][k]
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) o[cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, sB[k][16 * cb + li], o[cb], 0, 0, 0);
    }
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
      for (int r = 0; r < 4; ++r) Lv[(64 + 16 * wave + lq + 4 * r) * 128 + 16 * cb + li] = -o[cb][r];
  }
  save_tile(64, 1);
  C128_T(10);
  C128_TFLUSH();
  if (a.dbg && threadIdx.x == 0) { atomicAdd(&a.dbg[18], t_prev - t_start); atomicAdd(&a.dbg[21 + j], t_prev - t_start); atomicMin(&a.dbg[33 + j], t_start); atomicMax(&a.dbg[45 + j], t_prev); atomicMax(&a.dbg[64 + tl[15]], C128_T0()); }
  if (bad && !a.flags) atomicMax(a.info, 1);
}

// factorization of `batch` systems, all tiles formed from `first` (group-wise path of rg_launch_chol_solve_src); n64 % 128 == 0, no separate
// right-hand-side rows (they are embedded or absent).  `linv` needs batch * (n64 / 128) * 16384 doubles.
static void c128_launch_factor(hipStream_t st, double* mats, int64_t mat_stride, int batch, int n64, double* dinv, double* linv,
                               int32_t* info, const FormSrc& first, int R, int64_t& nl) {
  C128Args a;
  a.mats = mats; a.mat_stride = mat_stride; a.n64 = n64; a.linv = linv; a.dinv = dinv; a.info = info;
  static const int flags = getenv("RG_C128_FLAGS") ? atoi(getenv("RG_C128_FLAGS")) : 0;
  if (flags & 1) R = 1;       // diagnostic: the ridge shifts of a fold matrix are NOT co-located
  a.flags = flags;
  a.Tp = n64 / 128; a.batch = batch; a.R = R; a.fs = first;
  static const bool dbg = getenv("RG_C128_DBG") && atoi(getenv("RG_C128_DBG")) != 0;
  a.dbg = nullptr;
  if (dbg && hipMalloc(&a.dbg, (64 + 2048) * sizeof(unsigned long long)) == hipSuccess) {
    (void)hipMemsetAsync(a.dbg, 0, (64 + 2048) * sizeof(unsigned long long), st);
    (void)hipMemsetAsync(a.dbg + 32, 0xFF, 12 * sizeof(unsigned long long), st);      // [32 + launch]: earliest workgroup start, [44 + launch]: latest end
  }
  for (int j = -1; j <= a.Tp - 2; ++j) {
    a.j = j;
    a.nplain = j < 0 ? 0 : std::max(0, a.Tp - j - 2);
    const unsigned grid = xcd_affine_grid(1, batch, R) + (a.nplain ? xcd_affine_grid(a.nplain, batch, R) : 0u);
    if (j < 0) hipLaunchKernelGGL(k_c128_panel<0>, dim3(grid), dim3(256), 0, st, a);
    else if (j == 0) hipLaunchKernelGGL(k_c128_panel<1>, dim3(grid), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(k_c128_panel<2>, dim3(grid), dim3(256), 0, st, a);
    ++nl;
  }
  if (a.dbg) {      // diagnostic: per-phase sums over the workgroups, in microseconds of workgroup time (s_memrealtime ticks at 100 MHz)
    unsigned long long h[64];
    (void)hipStreamSynchronize(st);
    (void)hipMemcpy(h, a.dbg, sizeof(h), hipMemcpyDeviceToHost);
    (void)hipFree(a.dbg);
    static const char* nm[11] = {"", "X1", "K1", "T1", "X2", "K2", "->sA", "factor11", "L21/M1/D22/save", "factor22", "I21/save"};
    fprintf(stderr, "c128 phases (batch %d, n64 %d): plain items %llu, successor items %llu; workgroup-us:", batch, n64, h[16], h[17]);
    for (int k = 1; k <= 10; ++k) fprintf(stderr, " %s %.0f", nm[k], h[k] / 100.0);
    fprintf(stderr, " | prologue %.0f, workgroup lifetimes %.0f; per launch j = -1 ..:", h[19] / 100.0, h[18] / 100.0);
    for (int k = 0; k < a.Tp && k < 11; ++k) fprintf(stderr, " %.0f", h[20 + k] / 100.0);
    fprintf(stderr, " | inside the tile factorizations: 16x16 steps %.0f, barrier after them %.0f, rows below + trailing update %.0f, inverse blocks %.0f", h[11] / 100.0, h[14] / 100.0, h[12] / 100.0, h[13] / 100.0);
    // per launch: first start .. last end, and the workgroups in flight on average over that span (512 = every slot of the 256 CUs taken)
    fprintf(stderr, " | per launch span us (workgroups in flight):");
    for (int k = 0; k < a.Tp && k < 11; ++k) {
      const double span = (double)(h[44 + k] - h[32 + k]) / 100.0;
      fprintf(stderr, " %.0f (%.0f)", span, span > 0 ? h[20 + k] / 100.0 / span : 0.0);
    }
    fprintf(stderr, "; first start to last end of the factorization %.0f us; a CU's latest workgroup end to the next start on it: %.2f us on average over %llu starts\n",
            (double)(h[44 + a.Tp - 1] - h[32]) / 100.0, h[57] ? (double)h[56] / 100.0 / (double)h[57] : 0.0, h[57]);
  }
}
