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
#define C128_STAGE_SYNC_ACC(acc) C128_STAGE_SYNC()
#define C128_STAGE_SYNC_ACCD(acc) C128_STAGE_SYNC()
#endif

#define C128_STAGE 32768
#define C128_LDS (2 * C128_STAGE + 4096)      // the diagonal items alias sA | sB (64 x 66 doubles each) + dv (64) onto the stage buffers

struct C128Args {
  double* mats; int64_t mat_stride; int n64;
  double* linv;         // [batch][Tp][128 * 128]: inverse of the diagonal block of a panel, row-major
  double* dinv;         // [batch][n64 / 64][64 * 64]: tile inverses for the back substitution (k_chol_backsolve*)
  int32_t* info;
  int j, Tp, batch, R, nplain;
  FormSrc fs;
};

// ---- one K-chunk stage: accT[n][m] += L_j rows(n) x tile rows(m)^T over 16 k ------------------------------------------------------------
// DIAG: both operands are the panel's own rows (A part of the stage); wave W issues the blocks n <= W (m = 0: row block W) and n <= 7 - W
// (m = 1: row block 7 - W).
template <bool DIAG, int W>
__device__ __forceinline__ void c128_kstage(const uint8_t* cur, int aoff0, int aoff1, int boff, int s0, int s1, v4d (&acc)[8][2]) {
  double2 a[2][2];
  a[0][0] = *reinterpret_cast<const double2*>(cur + aoff0 + s0);
  a[0][1] = *reinterpret_cast<const double2*>(cur + aoff0 + s1);
  a[1][0] = *reinterpret_cast<const double2*>(cur + aoff1 + s0);
  a[1][1] = *reinterpret_cast<const double2*>(cur + aoff1 + s1);
#pragma unroll
  for (int nh = 0; nh < 2; ++nh) {        // the panel's row blocks in two halves: 32 operand registers live instead of 64
    double2 b[4][2];
#pragma unroll
    for (int nl = 0; nl < 4; ++nl) {
      const int n = 4 * nh + nl;
      if (DIAG && n > W && n > 7 - W) continue;
      b[nl][0] = *reinterpret_cast<const double2*>(cur + boff + n * 2048 + s0);
      b[nl][1] = *reinterpret_cast<const double2*>(cur + boff + n * 2048 + s1);
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int nl = 0; nl < 4; ++nl)
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          const int n = 4 * nh + nl;
          if (DIAG && n > (m == 0 ? W : 7 - W)) continue;
          const double av = kk == 0 ? a[m][0].x : (kk == 1 ? a[m][0].y : (kk == 2 ? a[m][1].x : a[m][1].y));
          const double bv = kk == 0 ? b[nl][0].x : (kk == 1 ? b[nl][0].y : (kk == 2 ? b[nl][1].x : b[nl][1].y));
          acc[n][m] = __builtin_amdgcn_mfma_f64_16x16x4f64(bv, av, acc[n][m], 0, 0, 0);
        }
  }
}

// ---- stage copies (direct global -> LDS, 16 bytes per lane) -------------------------------------------------------------------------------
// K chunk: piece p of wave w = rows 64 (w & 1) + 8 p .. + 7 of the A part (w < 2: the tile's rows) or the B part (the panel's rows); the slot
// swizzle (row >> 1) & 7 of row 8 p + (lane >> 3) depends on the parity of p alone: two lane offsets, the rest is scalar
__device__ __forceinline__ void c128_issue_k(const double* arows, const double* brows, int s, int buf, int wave, int n64, uint32_t lds0,
                                             const uint32_t (&koff)[2]) {
  const double* base = (wave < 2 ? arows : brows);      // brows == nullptr: the A part only
  if (base) {
    base += 16 * s + (int64_t)(64 * (wave & 1)) * n64;
    const uint32_t dst = lds0 + buf * C128_STAGE + wave * 8192;
#pragma unroll
    for (int p = 0; p < 8; ++p) c128_glds16(base + (int64_t)(8 * p) * n64, koff[p & 1], dst + p * 1024);
  }
}
// row stage: 64 rows x 128 doubles (1 KB per row), wave w copies rows 16 w .. 16 w + 15; lane l of row p takes the 16-byte slot l ^ p
__device__ __forceinline__ void c128_issue_rows(const double* src, int64_t ld, int wave, int lane, uint32_t lds0) {
  const double* base = src + (int64_t)(16 * wave) * ld;
  const uint32_t dst = lds0 + wave * 16384;
#pragma unroll
  for (int p = 0; p < 16; ++p) c128_glds16(base + (int64_t)p * ld, (uint32_t)((lane ^ p) << 4), dst + p * 1024);
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

  v4d acc[8][2];
#pragma unroll
  for (int n = 0; n < 8; ++n)
#pragma unroll
    for (int m = 0; m < 2; ++m) acc[n][m] = (v4d){0, 0, 0, 0};

  // ======== part 1: tile (it, j) = (X - L[it][0..j-1] L[j][0..j-1]^T) Linv_jj^T ==========================================================
  if (MODE >= 1) {
    const int rb0 = 32 * wave;                          // this wave's rows of the tile: rb0 + 16 m + i
    {
      // ---- X: acc = -(S - F) from the sources, two row stages per source (rows 0-63 | 64-127); waves 2h, 2h+1 own the rows of stage h
#pragma unroll 1
      for (int src = 0; src < (fx.F ? 2 : 1); ++src) {
        const double* Sx = (src ? fx.F : fx.S) + (int64_t)(128 * it) * n64 + 128 * j;
        const double sgn = src ? 1.0 : -1.0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          int lane = threadIdx.x & 63;
          C128_LAUNDER(lane);       // per stage: the element masks below must not be hoisted out of the stage (they were, as spilled SGPR pairs)
          const int i = lane & 15, q = lane >> 4;
          c128_issue_rows(Sx + (int64_t)(64 * h) * n64, n64, wave, lane, lds0);
          C128_STAGE_SYNC();
          if ((wave >> 1) == h) {
#pragma unroll
            for (int m = 0; m < 2; ++m) {
              // element (gi, gj) holds data when gj < nb and gi < nrhs: one limit per lane on the column offset e = 16 n + 4 r (+ q)
              const int gi = 128 * it + rb0 + 16 * m + i;
              const int lim = gi < nrhs ? nb - 128 * j - q : -1;
#pragma unroll
              for (int n = 0; n < 8; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                  double x = c128_rows_at(smem, 2 * (wave & 1) + m, n, r, i, q);
                  x = (16 * n + 4 * r < lim) ? x : 0.0;
                  acc[n][m][r] = fma(sgn, x, acc[n][m][r]);
                }
            }
          }
          C128_STAGE_SYNC();
        }
      }
    }
    {
      // ---- K loop: 8 j stages of 16 k
      int lane = threadIdx.x & 63;
      C128_LAUNDER(lane);
      const int i = lane & 15, q = lane >> 4;
      const int xs = (i >> 1) & 7;
      const int s0 = (q ^ xs) << 4, s1 = ((q + 4) ^ xs) << 4;
      uint32_t koff[2];
#pragma unroll
      for (int p = 0; p < 2; ++p) koff[p] = (uint32_t)(((int64_t)(lane >> 3) * n64 + 2 * ((lane & 7) ^ (((8 * p + (lane >> 3)) >> 1) & 7))) * 8);
      const int ns = 8 * j;
      const double* arows = M + (int64_t)(128 * it) * n64;
      const double* brows = M + (int64_t)(128 * j) * n64;
      const int aoff0 = (rb0 + i) * 128, aoff1 = (rb0 + 16 + i) * 128, boff = 16384 + i * 128;
      if (MODE >= 2) {
        c128_issue_k(arows, brows, 0, 0, wave, n64, lds0, koff);
        C128_STAGE_SYNC();
        int s = 0;
#pragma unroll 1
        do {
          if (s + 1 < ns) c128_issue_k(arows, brows, s + 1, (s + 1) & 1, wave, n64, lds0, koff);
          c128_kstage<false, 0>(smem + (s & 1) * C128_STAGE, aoff0, aoff1, boff, s0, s1, acc);
          C128_STAGE_SYNC_ACC(acc);
        } while (++s < ns);
      }
    }
    {
      // ---- T = U Linv^T by row blocks cb of Linv: out[cb] = sum_{n <= cb} U[.][16n..] Linv[16cb..][16n..]^T; two row stages of Linv
      int lane = threadIdx.x & 63;
      C128_LAUNDER(lane);
      const int i = lane & 15, q = lane >> 4;
      const double* Li = a.linv + ((int64_t)b * a.Tp + j) * (128 * 128);
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
      for (int h = 0; h < 2; ++h) {
        c128_issue_rows(Li + (int64_t)(64 * h) * 128, 128, wave, lane, lds0);
        C128_STAGE_SYNC();
#pragma unroll
        for (int cl = 0; cl < 4; ++cl) {
          const int cb = 4 * h + cl;
          if (pend_cb >= 0) store_pend();
          v4d out[2];
          out[0] = (v4d){0, 0, 0, 0};
          out[1] = (v4d){0, 0, 0, 0};
#pragma unroll
          for (int n = 0; n < 8; ++n) {
            if (n > cb) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const double lv = c128_rows_at(smem, cl, n, r, i, q);
              out[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(-acc[n][0][r], lv, out[0], 0, 0, 0);
              out[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(-acc[n][1][r], lv, out[1], 0, 0, 0);
            }
          }
          pend[0] = out[0];
          pend[1] = out[1];
          pend_cb = cb;
        }
        C128_STAGE_SYNC();
      }
      store_pend();
    }
    if (!succ) return;
    __threadfence();                                    // the diagonal block below reads this tile back (same workgroup, through L2)
    C128_STAGE_SYNC();
#pragma unroll
    for (int n = 0; n < 8; ++n)
#pragma unroll
      for (int m = 0; m < 2; ++m) acc[n][m] = (v4d){0, 0, 0, 0};
  }

  // ======== part 2 (successor item): diagonal block jd = j + 1 ============================================================================
  const int jd = j + 1;
  // row blocks of 16: wave w holds block w (m = 0, columns n <= w) and block 7 - w (m = 1, columns n <= 7 - w)
  {
    const double sh = fx.sh;
#pragma unroll 1
    for (int src = 0; src < (fx.F ? 2 : 1); ++src) {
      const double* Sx = (src ? fx.F : fx.S) + (int64_t)(128 * jd) * n64 + 128 * jd;
#pragma unroll
      for (int h = 0; h < 2; ++h) {                     // stage h holds row blocks 4h .. 4h + 3: m = h for every wave
        int lane = threadIdx.x & 63;
        C128_LAUNDER(lane);
        const int i = lane & 15, q = lane >> 4;
        c128_issue_rows(Sx + (int64_t)(64 * h) * n64, n64, wave, lane, lds0);
        C128_STAGE_SYNC();
        const int rbk = h == 0 ? wave : 7 - wave;       // this wave's row block of the stage
        const int gi = 128 * jd + 16 * rbk + i;
        const int lim = gi < nrhs ? nb - 128 * jd - q : -1;        // off the diagonal: data when 16 n + 4 r < lim
        const int de = 16 * rbk + i - q;                           // the diagonal element sits at 16 n + 4 r == de
        // on the diagonal: + shift inside the matrix, 2^100 on an embedded right-hand-side row, 1 in the identity padding
        const double dadd = gi < nb ? sh : (gi < nrhs ? RG_EMBED_DIAG : 1.0), dmul = gi < nb ? 1.0 : 0.0;
#pragma unroll
        for (int n = 0; n < 8; ++n) {
          if (h == 0 && n > 3) continue;                // row blocks 0-3 have no columns past block 3
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            double x = c128_rows_at(smem, rbk - 4 * h, n, r, i, q);
            x = (16 * n + 4 * r < lim) ? x : 0.0;
            if (src == 0) {
              x = (16 * n + 4 * r == de) ? fma(x, dmul, dadd) : x;
              acc[n][h][r] = -x;
            } else {
              acc[n][h][r] += x;
            }
          }
        }
        C128_STAGE_SYNC();
      }
    }
  }
  {
    int lane = threadIdx.x & 63;
    C128_LAUNDER(lane);
    const int i = lane & 15, q = lane >> 4;
    const int xs = (i >> 1) & 7;
    const int s0 = (q ^ xs) << 4, s1 = ((q + 4) ^ xs) << 4;
    uint32_t koff[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) koff[p] = (uint32_t)(((int64_t)(lane >> 3) * n64 + 2 * ((lane & 7) ^ (((8 * p + (lane >> 3)) >> 1) & 7))) * 8);
    const int ns = 8 * jd;
    const double* arows = M + (int64_t)(128 * jd) * n64;
    const int aoff0 = (16 * wave + i) * 128, aoff1 = (16 * (7 - wave) + i) * 128, boff = i * 128;
    if (MODE >= 1) {
      c128_issue_k(arows, nullptr, 0, 0, wave, n64, lds0, koff);
      C128_STAGE_SYNC();
      int s = 0;
#pragma unroll 1
      do {
        if (s + 1 < ns) c128_issue_k(arows, nullptr, s + 1, (s + 1) & 1, wave, n64, lds0, koff);
        const uint8_t* cur = smem + (s & 1) * C128_STAGE;
        if (wave == 0) c128_kstage<true, 0>(cur, aoff0, aoff1, boff, s0, s1, acc);
        else if (wave == 1) c128_kstage<true, 1>(cur, aoff0, aoff1, boff, s0, s1, acc);
        else if (wave == 2) c128_kstage<true, 2>(cur, aoff0, aoff1, boff, s0, s1, acc);
        else c128_kstage<true, 3>(cur, aoff0, aoff1, boff, s0, s1, acc);
        C128_STAGE_SYNC_ACCD(acc);
      } while (++s < ns);
    }
  }
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
  __syncthreads();
  bad |= diag_factor_lds(sA, dv);
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
  __syncthreads();
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
  __syncthreads();
  save_tile(0, 0);
  __syncthreads();
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
  __syncthreads();
  bad |= diag_factor_lds(sA, dv);
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
  if (bad) atomicMax(a.info, 1);
}

// factorization of `batch` systems, all tiles formed from `first` (group-wise path of rg_launch_chol_solve_src); n64 % 128 == 0, no separate
// right-hand-side rows (they are embedded or absent).  `linv` needs batch * (n64 / 128) * 16384 doubles.
static void c128_launch_factor(hipStream_t st, double* mats, int64_t mat_stride, int batch, int n64, double* dinv, double* linv,
                               int32_t* info, const FormSrc& first, int R, int64_t& nl) {
  C128Args a;
  a.mats = mats; a.mat_stride = mat_stride; a.n64 = n64; a.linv = linv; a.dinv = dinv; a.info = info;
  a.Tp = n64 / 128; a.batch = batch; a.R = R; a.fs = first;
  for (int j = -1; j <= a.Tp - 2; ++j) {
    a.j = j;
    a.nplain = j < 0 ? 0 : std::max(0, a.Tp - j - 2);
    const unsigned grid = xcd_affine_grid(1, batch, R) + (a.nplain ? xcd_affine_grid(a.nplain, batch, R) : 0u);
    if (j < 0) hipLaunchKernelGGL(k_c128_panel<0>, dim3(grid), dim3(256), 0, st, a);
    else if (j == 0) hipLaunchKernelGGL(k_c128_panel<1>, dim3(grid), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(k_c128_panel<2>, dim3(grid), dim3(256), 0, st, a);
    ++nl;
  }
}
