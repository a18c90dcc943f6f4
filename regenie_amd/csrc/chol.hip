// Batched multi-lambda ridge solves in fp64.
//
// The reference solves (A - A_i + lambda_r I) beta = (b - b_i) for every fold i and ridge value r
// through one symmetric eigendecomposition per fold (src/Step1_Models.cpp:484-494 for level 0,
// :827-838 for level 1).  K*R factorizations of order bs (or L) are latency-hostile as eigen
// problems on a GPU; K*R Cholesky factorizations of well conditioned (lambda >= ~1e3) matrices are
// fewer flops (R*n^3/3 < 9 n^3) and batch perfectly, and agree with the eigen route to ~1e-12.
//
// Layout of one system: row-major (n64 + rhs_pad) x n64, ld = n64; rows < n64 hold the lower
// triangle of the SPD matrix, rows >= n64 hold the right-hand sides as ROWS, so the forward
// substitution happens for free as part of the factorization (the RHS rows are just more strip rows).
// Padded diagonal entries are 1.
// Embedded right-hand sides (FormSrc::embed = P, level 0 whenever bs + P <= n64): the P right-hand sides are rows n .. n + P - 1 of the
// matrix itself -- the padding rows of its last tile -- with 2^100 on their diagonal: the system [[A, .], [b^T, D]] has the factor
// [[L, 0], [y^T, .]], y = L^-1 b, so the forward substitution is still part of the factorization, but there is no sixty-four-row tile
// row for (at level 0) ONE right-hand side: that tile row cost 22 % of the strip kernel's products and a launch of its own for the
// last group.  The back substitution then reads y from row n + p with the columns >= n masked.
//
// Tile algorithm, tile 64, tile columns in groups of 4, left-looking over the groups -- per group [k0, k0+nc):
//   k_chol_update : the group's DIAGONAL block (<= 10 tiles) -= L[.][q < k0] L[.][q < k0]^T, one wave per tile
//   k_chol_gfact  : one WAVE per system factors that block (<= 256 x 256) tile column by tile column: diagonal tile
//                   factored and inverted in LDS (16x16x16 MFMA block products), the <= 3 tiles below it accumulated
//                   transposed so that the update's result registers are the A operand of the triangular multiply;
//                   leaves the tile inverses and the byte images the strip kernel stages
//   k_chol_gstrip : every tile row below the block, one workgroup each: C = A - L[t][q < k0] L[g][q < k0]^T (K = 64*k0)
//                   and then T = C * Lgg^-T by forward substitution over the group's tile columns, all in registers;
//                   every operand streams through a ring of LDS units filled by direct global -> LDS copies
// then k_chol_backsolve (one workgroup per system) solves L^T x = y for the RHS rows (HBM-bound: reads L once).
// Every tile below the diagonal blocks is read (or formed from the sources) ONCE and written ONCE; a group takes three
// launches.  The systems are never materialised by a separate pass: the first kernel that touches a tile reads it from
// the source matrices (FormSrc).
//
// fp64 MFMA (v_mfma_f64_16x16x4_f64): lane l supplies A[i = l&15][k = l>>4] and B[k = l>>4][j = l&15]
// and owns D[row = (l>>4) + 4*reg][col = l&15].  Both operands are K-contiguous rows here, and a
// contraction is invariant under any K permutation applied to both operands alike, so a lane takes a few
// CONSECUTIVE k of a chunk (one 16- or 32-byte piece of its row), from HBM/L2 (k_chol_update, gfact) or from LDS.
// What bounds the register-fed kernels is the L2 -> CU fabric (~9 TB/s chip-wide: a 64x64 tile per wave needs 8 flop/B);
// the strip kernel shares its operands through LDS instead.
// Every global load sits in straight-line code: hipcc turns a load under a data-dependent `if` into branch + load +
// s_waitcnt vmcnt(0) (one full memory round trip each); conditions are applied to addresses (clamping) and values
// (0/1 masks) instead.
#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include "rg_internal.h"

#define CT 64

template <int MT, int NT>
__device__ __forceinline__ void dmma_load(const double* const* arow, const double* const* brow,
                                          double (&av)[MT][16], double (&bv)[NT][16]) {
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const double4* p = reinterpret_cast<const double4*>(arow[m]);
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const double4 x = p[v];
      av[m][4 * v] = x.x; av[m][4 * v + 1] = x.y; av[m][4 * v + 2] = x.z; av[m][4 * v + 3] = x.w;
    }
  }
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const double4* p = reinterpret_cast<const double4*>(brow[n]);
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const double4 x = p[v];
      bv[n][4 * v] = x.x; bv[n][4 * v + 1] = x.y; bv[n][4 * v + 2] = x.z; bv[n][4 * v + 3] = x.w;
    }
  }
}

template <int MT, int NT>
__device__ __forceinline__ void dmma_fma(const double (&av)[MT][16], const double (&bv)[NT][16],
                                         v4d (&acc)[MT][NT]) {
#pragma unroll
  for (int s = 0; s < 16; ++s)
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n)
        acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[m][s], bv[n][s], acc[m][n], 0, 0, 0);
}

// ---- generic C = A * B^T (test entry + reuse) -----------------------------------------------
__global__ __launch_bounds__(256) void k_dgemm_nt(const double* A, int64_t lda, const double* B,
                                                  int64_t ldb, int64_t K, double* C, int64_t ldc) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int i = lane & 15, q = lane >> 4;
  const int64_t r0 = (int64_t)blockIdx.y * CT + wr * 32, c0 = (int64_t)blockIdx.x * CT + wc * 32;
  v4d acc[2][2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n) acc[m][n] = (v4d){0, 0, 0, 0};
  for (int64_t k = 0; k < K; k += 64) {
    const double* ar[2] = {A + (r0 + i) * lda + k + 16 * q, A + (r0 + 16 + i) * lda + k + 16 * q};
    const double* br[2] = {B + (c0 + i) * ldb + k + 16 * q, B + (c0 + 16 + i) * ldb + k + 16 * q};
    double av[2][16], bv[2][16];
    dmma_load<2, 2>(ar, br, av, bv);
    dmma_fma<2, 2>(av, bv, acc);
  }
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        C[(r0 + m * 16 + q + 4 * r) * ldc + c0 + n * 16 + i] = acc[m][n][r];
}

// ---- the same product through LDS: one WORKGROUP per 128 x 128 macro tile (4 waves, 2 x 2 of 64 x 64) ---------------------------------
// The register-fed kernel above pulls 8 flop per byte through the L2 -> CU fabric and stays near 30 TFLOP/s (29 measured on the 1,024 x
// 131,072 x 1,024 products of the leave-one-out level 0, profiles/r5_loocv_level0_*).  This is the staging scheme of k_l1_gram128 (l1.hip)
// with two operand matrices: the 128 rows of A and the 128 rows of B of a macro tile are staged 16 K-positions at a time (32 KB per stage,
// two stages) by direct global -> LDS copies (global_load_lds, 16 B per lane) and shared by the four waves; the eight 16-byte slots of a row
// are XOR-swizzled by ((row >> 1) & 7) so that every ds_read_b128 lane group hits 16 distinct bank quads; one barrier per stage.
// Work order: workgroup id w runs on XCD w % 8; an XCD walks super tiles of SM x SN macro tiles (<= 64 = its resident workgroups), so that
// the A and B rows a super tile needs are fetched once into that XCD's L2 and shared by its SM * SN products.
// m, n: multiples of 64 (rows past them are clamped for the loads, wave tiles past them are not stored); K: a multiple of 16.
struct Gemm128 { const double* A; const double* B; double* C; int64_t lda, ldb, ldc, K; int m, n, MT, NTl, SM, SN, SMcount; };
__global__ __launch_bounds__(256, 2) void k_dgemm_nt128(Gemm128 g) {
  __shared__ __attribute__((aligned(16))) uint8_t smem[2 * 32768];
  int mt, nt;
  {
    const int w = blockIdx.x, x = w & 7, s = w >> 3;
    const int per = g.SM * g.SN, u = s / per, within = s % per;
    const int mi = within % g.SM, ni = within / g.SM;
    const int sm = u % g.SMcount, snl = u / g.SMcount;
    mt = sm * g.SM + mi;
    nt = (snl * 8 + x) * g.SN + ni;
    if (mt >= g.MT || nt >= g.NTl) return;
  }
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int i = lane & 15, q = lane >> 4;
  const int ns = (int)(g.K / 16);
  const double* src[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int rl = 8 * (wave * 8 + j) + (lane >> 3);
    const int slot = (lane & 7) ^ ((rl >> 1) & 7);
    src[j] = (rl < 128 ? g.A + (int64_t)min(mt * 128 + rl, g.m - 1) * g.lda : g.B + (int64_t)min(nt * 128 + rl - 128, g.n - 1) * g.ldb) + 2 * slot;
  }
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)smem;
  const uint32_t wdst = __builtin_amdgcn_readfirstlane(lds0 + wave * 8192);
  auto issue = [&](int buf) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      glds16p(src[j], wdst + buf * 32768 + j * 1024);
      src[j] += 16;
    }
  };
  const int row0 = mt * 128 + wr * 64, col0 = nt * 128 + wc * 64;
  const bool live = row0 < g.m && col0 < g.n;
  v4d acc[4][4];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[m][n] = (v4d){0, 0, 0, 0};
  const int xs = (i >> 1) & 7;
  const int oa = (wr * 64 + i) * 128 + ((q ^ xs) << 4), ob = 16384 + (wc * 64 + i) * 128 + ((q ^ xs) << 4);
  const int o4 = (((q + 4) ^ xs) << 4) - ((q ^ xs) << 4);
  if (ns > 0) issue(0);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  for (int s = 0; s < ns; ++s) {
    const uint8_t* cur = smem + (s & 1) * 32768;
    if (s + 1 < ns) issue((s + 1) & 1);
    if (live) {
      double2 a[4][2], b[4][2];
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        a[m][0] = *reinterpret_cast<const double2*>(cur + oa + m * 2048);
        a[m][1] = *reinterpret_cast<const double2*>(cur + oa + m * 2048 + o4);
        b[m][0] = *reinterpret_cast<const double2*>(cur + ob + m * 2048);
        b[m][1] = *reinterpret_cast<const double2*>(cur + ob + m * 2048 + o4);
      }
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int n = 0; n < 4; ++n) {
            const double av = kk == 0 ? a[m][0].x : (kk == 1 ? a[m][0].y : (kk == 2 ? a[m][1].x : a[m][1].y));
            const double bv = kk == 0 ? b[n][0].x : (kk == 1 ? b[n][0].y : (kk == 2 ? b[n][1].x : b[n][1].y));
            acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[m][n], 0, 0, 0);
          }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
  if (!live) return;
  double* O = g.C + (int64_t)row0 * g.ldc + col0;
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) O[(int64_t)(m * 16 + q + 4 * r) * g.ldc + n * 16 + i] = acc[m][n][r];
}

void rg_launch_dgemm_nt(hipStream_t st, const double* A, int64_t lda, const double* B, int64_t ldb,
                        int m, int n, int64_t k, double* C, int64_t ldc) {
  static const bool old = getenv("RG_DGEMM_REG") != nullptr;
  const bool al = ((uintptr_t)A % 16 == 0) && ((uintptr_t)B % 16 == 0) && lda % 2 == 0 && ldb % 2 == 0;
  if (old || !al || k % 16 != 0 || (int64_t)m * n < 128 * 128) {
    hipLaunchKernelGGL(k_dgemm_nt, dim3(n / CT, m / CT), dim3(256), 0, st, A, lda, B, ldb, k, C, ldc);
    return;
  }
  Gemm128 g;
  g.A = A; g.B = B; g.C = C; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.K = k; g.m = m; g.n = n;
  g.MT = (m + 127) / 128; g.NTl = (n + 127) / 128;
  g.SM = std::min(g.MT, 8); g.SN = std::max(1, std::min(g.NTl, 64 / g.SM));
  g.SMcount = (g.MT + g.SM - 1) / g.SM;
  const int nsup = (g.NTl + g.SN - 1) / g.SN;
  const int64_t per_xcd = (int64_t)g.SMcount * ((nsup + 7) / 8) * g.SM * g.SN;
  hipLaunchKernelGGL(k_dgemm_nt128, dim3((unsigned)(per_xcd * 8)), dim3(256), 0, st, g);
}

#include "chol_common.h"

// ==== small-batch path (a few dozen systems: level 1, the logistic IRLS steps) ========================================
// With fewer systems than SIMDs the wave-per-system block factorization is a long serial chain per system; here every
// tile column is two launches that spread one system over many waves: k_chol_diag (one workgroup per system: in-group
// update, factorization and inverse of the diagonal tile) and k_chol_panel (one wave per tile below it), followed per
// group by the wide trailing update.
// Left-looking inside a column group: before factoring, the tile is updated with the group's earlier tile
// columns [kc0, kc0 + nkc):  A[k][k] -= sum_q L[k][q] L[k][q]^T  (fp64 MFMA, operands straight from HBM/L2).
__global__ __launch_bounds__(256) void k_chol_diag(double* mats, int64_t mat_stride, int n64, int k, int kc0,
                                                   int nkc, double* dinv, int32_t* info, FormSrc fs) {
  // ONE 64 x 66 LDS array holds both results (34 KB -> 4 workgroups per CU, a whole 800-system batch resident):
  //   lower triangle + diagonal : L            strict upper triangle : Linv^T  (Linv[r][c] at s[c][r], r > c)
  //   dv[r] = Linv[r][r] = 1 / L[r][r]
  __shared__ double s[CT][CT + 2];
  __shared__ double dv[CT];
  const int b = blockIdx.x;
  const int tid = threadIdx.x;
  double* D = mats + (int64_t)b * mat_stride + (int64_t)k * CT * n64 + k * CT;
  if (fs.enabled) {
    const FormIdx fx = form_idx(fs, b);
    const int md = form_mode_rows(fx, k * CT, (k + 1) * CT);   // a diagonal tile never lies in the extra rows: 0 or 1
#pragma unroll
    for (int u = 0; u < CT * CT / 256; ++u) {   // unconditional loads (the upper triangle of the source exists), then select
      const int e = tid + 256 * u;
      const int r = e >> 6, c = e & 63;
      const int gi = k * CT + r, gj = k * CT + c;
      const int64_t ge = (int64_t)gi * n64 + gj;
      const double v = md == 0 ? form_val<0>(fx, gi, gj, ge) : (md == 1 ? form_val<1>(fx, gi, gj, ge) : form_val<2>(fx, gi, gj, ge));
      s[r][c] = (c <= r) ? v : 0.0;
    }
  } else {
#pragma unroll
    for (int u = 0; u < CT * CT / 256; ++u) {
      const int e = tid + 256 * u;
      const int r = e >> 6, c = e & 63;
      const double v = D[(int64_t)r * n64 + c];
      s[r][c] = (c <= r) ? v : 0.0;
    }
  }
  __syncthreads();
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lq = lane >> 4;
  if (nkc > 0) {
    const int wr = wave >> 1, wc = wave & 1;
    const int i = li, q = lq;
    const double* Mrow = mats + (int64_t)b * mat_stride + (int64_t)k * CT * n64 + (int64_t)kc0 * CT + 16 * q;
    v4d acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int n = 0; n < 2; ++n) acc[m][n] = (v4d){0, 0, 0, 0};
    if (wc <= wr) {   // the strictly upper 32x32 block is never referenced
      for (int kk = 0; kk < nkc; ++kk) {
        const double* ar[2] = {Mrow + (int64_t)(wr * 32 + i) * n64 + kk * CT, Mrow + (int64_t)(wr * 32 + 16 + i) * n64 + kk * CT};
        const double* br[2] = {Mrow + (int64_t)(wc * 32 + i) * n64 + kk * CT, Mrow + (int64_t)(wc * 32 + 16 + i) * n64 + kk * CT};
        double av[2][16], bv[2][16];
        dmma_load<2, 2>(ar, br, av, bv);
        dmma_fma<2, 2>(av, bv, acc);
      }
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int rr = wr * 32 + m * 16 + q + 4 * r, cc = wc * 32 + n * 16 + i;
            if (cc <= rr) s[rr][cc] -= acc[m][n][r];
          }
    }
    __syncthreads();
  }
  if (diag_factor_lds(s, dv)) atomicMax(info, 1);
  double* I = dinv + ((int64_t)b * (n64 / CT) + k) * CT * CT;
  for (int e = tid; e < CT * CT; e += 256) {
    const int rr = e >> 6, c = e & 63;
    if (c <= rr) D[(int64_t)rr * n64 + c] = s[rr][c];
    I[e] = (c < rr) ? s[c][rr] : ((c == rr) ? dv[rr] : 0.0);
  }
}

// ---- panel: L[t][k] = (A[t][k] - sum_q L[t][q] L[k][q]^T) * Linv^T, q over the group's earlier tile columns ----
// (left-looking inside the column group: the tile is read once and written once).  One WAVE per 64x64 tile, four
// tiles of the same tile column per workgroup (they share L[k][q] and Linv, the latter staged once in LDS).
// The in-group update is accumulated TRANSPOSED -- acc[n][m] += L[k]-rows(n) x L[t]-rows(m)^T -- so that lane
// (i, q) ends up holding U[row 16m + i][cols 16n + q + 4r] of the updated tile U: one row, 16 column values per
// 16-column block, which is exactly an MFMA A-operand layout for the triangular multiply T = U * Linv^T with the K
// assignment kk(q, step = (n, r)) = 16n + q + 4r.  No LDS round trip, no barrier between update and multiply; the
// K steps with n > (column block of the output) are skipped because Linv is lower triangular.
#define PL_PITCH 66
__global__ __launch_bounds__(256, 2) void k_chol_panel(double* mats, int64_t mat_stride, int n64, int k, int kc0,
                                                       int nkc, const double* dinv, int ngrp4, int batch, int R,
                                                       int ntile, FormSrc fs) {
  __shared__ double sLi[CT * PL_PITCH];
  int b, g;
  if (!xcd_affine(blockIdx.x, ngrp4, batch, R, b, g)) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, q = lane >> 4;
  double* M = mats + (int64_t)b * mat_stride;
  {  // stage Linv (64 x 64 doubles) with coalesced 32-byte loads
    const double* I = dinv + ((int64_t)b * (n64 / CT) + k) * CT * CT;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e4 = threadIdx.x + 256 * u;          // index of a double4
      const int r = e4 >> 4, c4 = (e4 & 15) * 4;
      const double4 v = *reinterpret_cast<const double4*>(I + r * CT + c4);
      double* d = sLi + r * PL_PITCH + c4;
      d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
  }
  __syncthreads();
  const int idx = g * 4 + wave;
  if (idx >= ntile) return;
  const int t = k + 1 + idx;
  double* T = M + (int64_t)t * CT * n64 + k * CT;
  // accT[n][m][r] = -(U[row 16m + i][col 16n + q + 4r])
  v4d acc[4][4];
  {
    auto init = [&](auto mode) {
      FormIdx fx{};
      if (decltype(mode)::value >= 0) fx = form_idx(fs, b);
#pragma unroll
      for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int lr = m * 16 + i, lc = n * 16 + q + 4 * r;
            if (decltype(mode)::value < 0) acc[n][m][r] = -T[(int64_t)lr * n64 + lc];
            else {
              const int gi = t * CT + lr, gj = k * CT + lc;
              acc[n][m][r] = -form_val<(decltype(mode)::value < 0 ? 0 : decltype(mode)::value)>(fx, gi, gj, (int64_t)gi * n64 + gj);
            }
          }
    };
    if (!fs.enabled) init(std::integral_constant<int, -1>{});
    else {
      const FormIdx f0 = form_idx(fs, b);
      const int md = form_mode_rows(f0, t * CT, (t + 1) * CT);
      if (md == 0) init(std::integral_constant<int, 0>{});
      else if (md == 1) init(std::integral_constant<int, 1>{});
      else if (md == 3) init(std::integral_constant<int, 3>{});
      else init(std::integral_constant<int, 2>{});
    }
  }
  if (nkc > 0) {
    const double* A = M + ((int64_t)t * CT + i) * n64 + kc0 * CT + 2 * q;   // rows of the tile's own tile row
    const double* B = M + ((int64_t)k * CT + i) * n64 + kc0 * CT + 2 * q;   // rows of tile row k (the column's L[k][q])
    const int nk8 = nkc * 8;
    auto load8 = [&](double2 (&av)[4], double2 (&bv)[4], int kc) {
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        av[m] = *reinterpret_cast<const double2*>(A + (int64_t)m * 16 * n64 + kc * 8);
        bv[m] = *reinterpret_cast<const double2*>(B + (int64_t)m * 16 * n64 + kc * 8);
      }
    };
    auto mma8 = [&](const double2 (&av)[4], const double2 (&bv)[4]) {
#pragma unroll
      for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[n][m] = __builtin_amdgcn_mfma_f64_16x16x4f64(bv[n].x, av[m].x, acc[n][m], 0, 0, 0);
#pragma unroll
      for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[n][m] = __builtin_amdgcn_mfma_f64_16x16x4f64(bv[n].y, av[m].y, acc[n][m], 0, 0, 0);
    };
    double2 a0[4], b0[4], a1[4], b1[4];
    load8(a0, b0, 0);
    for (int kc = 0; kc < nk8; kc += 2) {
      load8(a1, b1, kc + 1);
      mma8(a0, b0);
      if (kc + 2 < nk8) load8(a0, b0, kc + 2);
      mma8(a1, b1);
    }
  }
  // triangular multiply, one 16-row block at a time: out[cb] = sum_{n <= cb, r} (-acc[n][m][r]) x Linv[16cb + i][16n + q + 4r]
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    v4d out[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
      out[cb] = (v4d){0, 0, 0, 0};
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        if (n > cb) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r)
          out[cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(-acc[n][m][r], sLi[(cb * 16 + i) * PL_PITCH + n * 16 + q + 4 * r],
                                                         out[cb], 0, 0, 0);
      }
    }
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
      for (int r = 0; r < 4; ++r) T[(int64_t)(m * 16 + q + 4 * r) * n64 + cb * 16 + i] = out[cb][r];
  }
}

// ---- update: A[r][c] -= sum_{q in [kc0, kc0+nkc)} L[r][q] L[c][q]^T for tile columns c in [c_lo, c_hi),
//      rows r in [c, Ttot) (matrix tiles below/on the diagonal plus the RHS row tiles) ---------------------
// One WAVE per 64x64 tile (4 x 4 MFMA sub-tiles, 64 accumulator doubles per lane): per 16-deep K chunk a
// wave loads 2 x 4 row pieces of 32 bytes per lane and issues 64 MFMAs (~4K cycles), i.e. ~4 B/clk of
// operands per wave -- half of a 32x32-per-wave tiling; the four waves of a workgroup take consecutive
// tiles of one tile column, so they share the B operand in L1.
__global__ __launch_bounds__(256, 2) void k_chol_update(double* mats, int64_t mat_stride, int n64,
                                                        int Ttot, int c_lo, int c_hi, int ntile,
                                                        int kc0, int nkc, int batch, int R, FormSrc fs) {
  int b, g;
  if (!xcd_affine(blockIdx.x, (ntile + 3) / 4, batch, R, b, g)) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int idx = g * 4 + wave, tc = c_lo;
  if (idx >= ntile) return;
  while (tc < c_hi && idx >= Ttot - tc) { idx -= Ttot - tc; ++tc; }
  const int tr = tc + idx;
  {
    const int T = n64 / CT;
    if (tr < T && tr >= sys_tiles(fs, b, T)) return;   // identity padding of a system smaller than the batch's order
  }
  const int i = lane & 15, q = lane >> 4;
  double* M = mats + (int64_t)b * mat_stride;
  const double* A = M + ((int64_t)tr * CT + i) * n64 + kc0 * CT + 2 * q;
  const double* B = M + ((int64_t)tc * CT + i) * n64 + kc0 * CT + 2 * q;
  double* C = M + (int64_t)tr * CT * n64 + tc * CT;
  v4d acc[4][4];
  if (fs.enabled) {
    const FormIdx fx = form_idx(fs, b);
    auto init = [&](auto mode) {
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int gi = tr * CT + m * 16 + q + 4 * r, gj = tc * CT + n * 16 + i;
            acc[m][n][r] = -form_val<decltype(mode)::value>(fx, gi, gj, (int64_t)gi * n64 + gj);
          }
    };
    const int md = form_mode_rows(fx, tr * CT, (tr + 1) * CT);
    if (md == 0) init(std::integral_constant<int, 0>{});
    else if (md == 1) init(std::integral_constant<int, 1>{});
    else if (md == 3) init(std::integral_constant<int, 3>{});
    else init(std::integral_constant<int, 2>{});
  } else {
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          acc[m][n][r] = -C[(int64_t)(m * 16 + q + 4 * r) * n64 + n * 16 + i];
  }
  // K loop in 8-deep chunks, software pipelined over two register sets: the 8 x 16-byte loads of chunk c+1 are in
  // flight while the 32 MFMAs (~2K cycles) of chunk c issue, so a wave no longer stalls for the full load latency in
  // front of every MFMA burst (with only two waves per SIMD that stall was ~1/3 of the time).  Lane (i, q) supplies
  // k = 2q + s of a chunk to MFMA step s.
  const int nk8 = nkc * 8;
  auto load8 = [&](double2 (&av)[4], double2 (&bv)[4], int kc) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      av[m] = *reinterpret_cast<const double2*>(A + (int64_t)m * 16 * n64 + kc * 8);
      bv[m] = *reinterpret_cast<const double2*>(B + (int64_t)m * 16 * n64 + kc * 8);
    }
  };
  auto mma8 = [&](const double2 (&av)[4], const double2 (&bv)[4]) {
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[m].x, bv[n].x, acc[m][n], 0, 0, 0);
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[m].y, bv[n].y, acc[m][n], 0, 0, 0);
  };
  {
    double2 a0[4], b0[4], a1[4], b1[4];
    load8(a0, b0, 0);
    for (int kc = 0; kc < nk8; kc += 2) {   // nk8 is a multiple of 8
      load8(a1, b1, kc + 1);
      mma8(a0, b0);
      if (kc + 2 < nk8) load8(a0, b0, kc + 2);
      mma8(a1, b1);
    }
  }
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        C[(int64_t)(m * 16 + q + 4 * r) * n64 + n * 16 + i] = -acc[m][n][r];
}


// =====================================================================================================================
// Group-wise (left-looking) factorization: per group of 4 tile columns [k0, k0+nc)
//   k_chol_update (reused) : the group's DIAGONAL block (<= 10 tiles) -= L[.][q < k0] L[.][q < k0]^T
//   k_chol_gfact           : one workgroup per system factors that block (<= 256 x 256) tile column by tile column
//                            (diagonal tile in LDS, the <= 3 tiles below it by 16-row slabs) and emits the tile inverses
//   k_chol_gstrip          : every tile row below the block, one workgroup each: C = A - L[t][q < k0] L[g][q < k0]^T
//                            (K = 64*k0 contracted through LDS stages) and then T = C * Lgg^-T by forward substitution
//                            over the group's tile columns, all in registers: each wave owns 16 rows x 256 columns
// Every tile below the diagonal blocks is read (or formed from the sources) ONCE and written ONCE, the L panels are
// read through LDS stages shared by the four waves of a workgroup, and a group takes 3 launches.
// =====================================================================================================================
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;

// ---- operand images of a group's diagonal block ------------------------------------------------------------------------
// gfact leaves, per system and group, the 64x64 tiles that k_chol_gstrip's substitution consumes, in consumption order
// (for c: L[c][0..c-1], then the inverse of L[c][c]) and in the exact byte image the consumer wants in LDS, so that staging
// is a straight copy.  A tile is two 16 KB halves (columns 0-31 | 32-63), each [64 rows][32 columns]; inside a half, element
// (row, col = 16n + q + 4r) sits at row*32 + 2*(((8(n&1) + 2q + (r >> 1)) ^ (row & 15))) + (r & 1): the four k = 16n + q + 4r,
// r = 0..3 a lane needs are two aligned 16-byte slots, XOR-swizzled by the row so that every ds_read_b128 lane group
// falls on 16 distinct bank groups.
#define GT_TILE (CT * CT)
#define GT_NIMG 10
__device__ __forceinline__ int img_pos(int row, int col) {
  const int n = col >> 4, qq = col & 3, r = (col >> 2) & 3;
  return (n >> 1) * (GT_TILE / 2) + row * 32 + ((((8 * (n & 1) + 2 * qq + (r >> 1)) ^ (row & 15)) << 1) | (r & 1));
}
static inline size_t chol_ws_img_offset(size_t batch, int n64) { return batch * (size_t)(n64 / CT) * GT_TILE; }

// ---- wave-level blocked (16) factorization + inverse of the 64x64 tile held in LDS, one wave, no barriers ----------------
// On return: lower triangle + diagonal of s = L, strict upper triangle = Linv^T, dv[r] = Linv[r][r]; returns true (in some
// lane) when a pivot was not positive.  MFMA 16x16x4: lane (i = lane&15, q = lane>>4) supplies A[i][kk], B[kk][i] and owns
// D[q + 4r][i]; a K = 16 block product is 4 instructions with kk(q, s) chosen per product.
__device__ __forceinline__ bool diag_factor_wave(double (*s)[CT + 2], double* dv) {
  const int lane = threadIdx.x & 63;
  const int li = lane & 15, lq = lane >> 4;
  // unconditional LDS reads combined with 0/1 factors: a load under a data-dependent condition becomes a branch + wait
  auto inv_diag = [&](int o, int r, int c) -> double {
    return s[o + c][o + r] * (c < r ? 1.0 : 0.0) + dv[o + r] * (c == r ? 1.0 : 0.0);
  };
  bool bad = false;
#pragma unroll 1
  for (int sb = 0; sb < 4; ++sb) {
    const int o = sb * 16;
    const int nb = 3 - sb;
    if (lane < 16) {
      double a[16], rdv[16];
#pragma unroll
      for (int c = 0; c < 16; ++c) a[c] = s[o + lane][o + c];
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const double piv = bcast_lane(a[c], c);
        double d, rd;
        if (piv > 0.0) {
          double r0 = __builtin_amdgcn_rsq(piv);
          r0 = r0 * fma(-0.5 * piv * r0, r0, 1.5);
          r0 = r0 * fma(-0.5 * piv * r0, r0, 1.5);
          d = piv * r0;
          d = fma(0.5 * r0, fma(-d, d, piv), d);
          rd = fma(r0, fma(-d, r0, 1.0), r0);
        } else { d = 1.0; rd = 1.0; bad = true; }
        rdv[c] = rd;
        if (lane > c) a[c] *= rd;
        else if (lane == c) a[c] = d;
#pragma unroll
        for (int c2 = c + 1; c2 < 16; ++c2) {
          const double l = bcast_lane(a[c], c2);
          if (lane >= c2) a[c2] = fma(-a[c], l, a[c2]);
        }
      }
#pragma unroll
      for (int c = 0; c < 16; ++c)
        if (c <= lane) s[o + lane][o + c] = a[c];
      double x[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        double v = (r == lane) ? 1.0 : 0.0;
#pragma unroll
        for (int j = 0; j < r; ++j) v = fma(-bcast_lane(a[j], r), x[j], v);
        x[r] = v * rdv[r];
      }
      dv[o + lane] = x[lane];
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (r > lane) s[o + lane][o + r] = x[r];
    }
    // rows below: L21 = A21 * Linv11^T
#pragma unroll 1
    for (int wv = 0; wv < nb; ++wv) {
      const int rb = o + 16 + 16 * wv;
      v4d acc = (v4d){0, 0, 0, 0};
#pragma unroll
      for (int st = 0; st < 4; ++st)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(s[rb + li][o + 4 * lq + st], inv_diag(o, li, 4 * lq + st), acc, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) s[rb + lq + 4 * r][o + li] = acc[r];
    }
    // trailing update inside the tile
#pragma unroll 1
    for (int bi = 0; bi < nb; ++bi)
#pragma unroll 1
      for (int bj = 0; bj <= bi; ++bj) {
        const int ri = o + 16 + 16 * bi, rj = o + 16 + 16 * bj;
        v4d acc = (v4d){0, 0, 0, 0};
#pragma unroll
        for (int st = 0; st < 4; ++st)
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(s[ri + li][o + 4 * lq + st], s[rj + li][o + 4 * lq + st], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) s[ri + lq + 4 * r][rj + li] -= acc[r];   // upper triangles of later diagonal blocks: scratch until factored
      }
  }
  // off-diagonal blocks of the inverse, by sub-diagonal distance
#pragma unroll 1
  for (int dist = 1; dist < 4; ++dist)
#pragma unroll 1
    for (int j = 0; j + dist < 4; ++j) {
      const int ib = j + dist;
      v4d m1 = (v4d){0, 0, 0, 0};
#pragma unroll 1
      for (int kb = j; kb < ib; ++kb) {
#pragma unroll
        for (int st = 0; st < 4; ++st) {
          const int kk = 4 * lq + st;
          const double bval = (kb == j) ? inv_diag(16 * j, kk, li) : s[16 * j + li][16 * kb + kk];
          m1 = __builtin_amdgcn_mfma_f64_16x16x4f64(s[16 * ib + li][16 * kb + kk], bval, m1, 0, 0, 0);
        }
      }
      v4d acc = (v4d){0, 0, 0, 0};
#pragma unroll
      for (int st = 0; st < 4; ++st)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(inv_diag(16 * ib, li, lq + 4 * st), m1[st], acc, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) s[16 * j + li][16 * ib + lq + 4 * r] = -acc[r];
    }
  return bad;
}

// ---- diagonal block of a group: <= 4 tile columns, ONE WAVE per system (four systems per workgroup) ---------------------
// A 64x64 factorization is a chain of 64 dependent pivots; with one system per wave nothing waits at a barrier, the
// four SIMDs of a CU run four independent chains, and the tile never leaves the wave's LDS slice.  Per tile column k:
//   tile (k,k) -= sum_q L[k][q] L[k][q]^T over the group's earlier columns (accumulated in registers), factored and
//   inverted in LDS; then the tiles below it inside the block: L[t][k] = (A[t][k] - sum_q L[t][q] L[k][q]^T) Linv^T,
//   accumulated transposed (acc[n][m] = mfma(b, a): lane (i, q) ends up with one row and 16 column values per 16-column
//   block, exactly an MFMA A-operand layout for T = U * Linv^T with k(q, step = (n, r)) = 16n + q + 4r).  Tiles written in one step are re-read by the same wave in the next.
// value of element (gi, gj) of the system: from the workspace (md < 0) or formed from the sources; the mode is resolved
// around whole unrolled load loops (dispatch_md) so that every loop body is straight-line loads
template <int MD>
__device__ __forceinline__ double sys_val(const FormIdx& fx, const double* M, int n64, int gi, int gj) {
  const int64_t e = (int64_t)gi * n64 + gj;
  if (MD < 0) return M[e];
  return form_val<(MD < 0 ? 0 : MD)>(fx, gi, gj, e);
}
// the diagonal block of a group consists of matrix rows only (extra rows start past the right-hand sides): -1, 0 or 1
template <typename F>
__device__ __forceinline__ void dispatch_md(int md, F&& f) {
  if (md < 0) f(std::integral_constant<int, -1>{});
  else if (md == 0) f(std::integral_constant<int, 0>{});
  else f(std::integral_constant<int, 1>{});
}

__device__ __forceinline__ void gfact_wave(int md, double (*s)[CT + 2], double* dv, double* M, int b, int n64, int k0, int nc,
                                           double* dinvb, double* img, int32_t* info, const FormSrc& fs) {
  const int lane = threadIdx.x & 63;
  const int i = lane & 15, q = lane >> 4;
  FormIdx fx{};
  if (md >= 0) fx = form_idx(fs, b);
  bool bad = false;
#pragma unroll 1
  for (int c = 0; c < nc; ++c) {
    const int k = k0 + c;
    double* D = M + (int64_t)k * CT * n64 + k * CT;
    {
      // acc[m][n][r] = -(tile[16m + q + 4r][16n + i]), lower blocks only
      v4d acc[4][4];
      dispatch_md(md, [&](auto mode) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int n = 0; n < 4; ++n) {
            if (n > m) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r)
              acc[m][n][r] = -sys_val<decltype(mode)::value>(fx, M, n64, k * CT + 16 * m + q + 4 * r, k * CT + 16 * n + i);
          }
      });
      if (c > 0) {
        const double* Rk = M + ((int64_t)k * CT + i) * n64 + k0 * CT + 2 * q;
        // 8-deep chunks, software pipelined over two register sets (a single wave has nobody to hide a load behind)
        auto ld = [&](double2 (&av)[4], int kc) {
#pragma unroll
          for (int m = 0; m < 4; ++m) av[m] = *reinterpret_cast<const double2*>(Rk + (int64_t)m * 16 * n64 + kc * 8);
        };
        auto mm = [&](const double2 (&av)[4]) {
#pragma unroll
          for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n) {
              if (n > m) continue;
              acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[m].x, av[n].x, acc[m][n], 0, 0, 0);
            }
#pragma unroll
          for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n) {
              if (n > m) continue;
              acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[m].y, av[n].y, acc[m][n], 0, 0, 0);
            }
        };
        double2 u0[4], u1[4];
        const int nk8 = c * 8;
        ld(u0, 0);
#pragma unroll 1
        for (int kc = 0; kc < nk8; kc += 2) {
          ld(u1, kc + 1);
          mm(u0);
          ld(u0, kc + 2 < nk8 ? kc + 2 : kc);
          mm(u1);
        }
      }
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int lr = 16 * m + q + 4 * r, lc = 16 * n + i;
            s[lr][lc] = (n < m || (n == m && lc <= lr)) ? -acc[n <= m ? m : 0][n <= m ? n : 0][r] : 0.0;
          }
    }
    bad |= diag_factor_wave(s, dv);
    {
      double* I = dinvb + (int64_t)k * GT_TILE;
      double* G = img + (int64_t)(c * (c + 1) / 2 + c) * GT_TILE;
#pragma unroll 4
      for (int rr = 0; rr < CT; ++rr) {
        const int cc = lane;
        if (cc <= rr) D[(int64_t)rr * n64 + cc] = s[rr][cc];
        const double lin = s[cc][rr] * (cc < rr ? 1.0 : 0.0) + dv[rr] * (cc == rr ? 1.0 : 0.0);
        I[rr * CT + cc] = lin;
        G[img_pos(rr, cc)] = lin;
      }
    }
    // tiles (t, k) below the diagonal tile inside the block, one at a time
#pragma unroll 1
    for (int t = k + 1; t < k0 + nc; ++t) {
      double* T = M + (int64_t)t * CT * n64 + k * CT;
      v4d acc[4][4];   // accT[n][m][r] = -(U[row 16m + i][col 16n + q + 4r])
      dispatch_md(md, [&](auto mode) {
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
          for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              acc[n][m][r] = -sys_val<decltype(mode)::value>(fx, M, n64, t * CT + m * 16 + i, k * CT + n * 16 + q + 4 * r);
      });
      if (c > 0) {
        const double* A = M + ((int64_t)t * CT + i) * n64 + k0 * CT + 2 * q;
        const double* B = M + ((int64_t)k * CT + i) * n64 + k0 * CT + 2 * q;
        auto ld = [&](double2 (&av)[4], double2 (&bv)[4], int kc) {
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            av[m] = *reinterpret_cast<const double2*>(A + (int64_t)m * 16 * n64 + kc * 8);
            bv[m] = *reinterpret_cast<const double2*>(B + (int64_t)m * 16 * n64 + kc * 8);
          }
        };
        auto mm = [&](const double2 (&av)[4], const double2 (&bv)[4]) {
#pragma unroll
          for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int m = 0; m < 4; ++m) acc[n][m] = __builtin_amdgcn_mfma_f64_16x16x4f64(bv[n].x, av[m].x, acc[n][m], 0, 0, 0);
#pragma unroll
          for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int m = 0; m < 4; ++m) acc[n][m] = __builtin_amdgcn_mfma_f64_16x16x4f64(bv[n].y, av[m].y, acc[n][m], 0, 0, 0);
        };
        double2 a0[4], b0[4], a1[4], b1[4];
        const int nk8 = c * 8;
        ld(a0, b0, 0);
#pragma unroll 1
        for (int kc = 0; kc < nk8; kc += 2) {
          ld(a1, b1, kc + 1);
          mm(a0, b0);
          ld(a0, b0, kc + 2 < nk8 ? kc + 2 : kc);
          mm(a1, b1);
        }
      }
      const int ct = t - k0;
      double* G = img + (int64_t)(ct * (ct + 1) / 2 + c) * GT_TILE;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        v4d out[4];
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
          out[cb] = (v4d){0, 0, 0, 0};
#pragma unroll
          for (int n = 0; n < 4; ++n) {
            if (n > cb) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int lr = 16 * cb + i, lc = 16 * n + q + 4 * r;
              const double li = (n < cb) ? s[lc][lr] : s[lc][lr] * (lc < lr ? 1.0 : 0.0) + dv[lr] * (lc == lr ? 1.0 : 0.0);
              out[cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(-acc[n][m][r], li, out[cb], 0, 0, 0);
            }
          }
        }
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int lr = m * 16 + q + 4 * r, lc = cb * 16 + i;
            T[(int64_t)lr * n64 + lc] = out[cb][r];
            G[img_pos(lr, lc)] = out[cb][r];
          }
      }
    }
    __threadfence_block();
  }
  if (bad) atomicMax(info, 1);
}

__global__ __launch_bounds__(256) void k_chol_gfact(double* mats, int64_t mat_stride, int n64, int k0, int nc, int batch,
                                                    double* dinv, double* dimg, int ngrp, int32_t* info, FormSrc fs) {
  __shared__ double S[4][CT][CT + 2];
  __shared__ double DV[4][CT];
  const int wave = threadIdx.x >> 6;
  const int b = blockIdx.x * 4 + wave;
  if (b >= batch) return;     // no workgroup barriers below: the waves are independent
  double* M = mats + (int64_t)b * mat_stride;
  double* dinvb = dinv + (int64_t)b * (n64 / CT) * GT_TILE;
  double* img = dimg + ((int64_t)b * ngrp + k0 / 4) * GT_NIMG * GT_TILE;
  int md = -1;
  if (fs.enabled) {
    const FormIdx f0 = form_idx(fs, b);
    md = f0.F ? 1 : 0;   // form_mode_rows for rows < n64 <= extra_row0
  }
  const int tb = sys_tiles(fs, b, n64 / CT);
  if (k0 >= tb) return;                 // the whole group is identity padding for this system
  if (k0 + nc > tb) nc = tb - k0;
  gfact_wave(md, S[wave], DV[wave], M, b, n64, k0, nc, dinvb, img, info, fs);
}

// ---- tile rows below the diagonal block -----------------------------------------------------------------------------
// 256 threads = 4 waves; a wave owns NS slabs of 16 rows, the workgroup 64*NS rows.  Lane (i, q) holds, per slab,
// xn[n][r] = -X[row i][col 16n + q + 4r], n = 0..15 over the group's 256 columns: the transposed-accumulation layout
// (acc = mfma(L-rows, strip-rows)), which is also the B-operand layout (k = 16n + q + 4r) of the substitution products.
// EVERYTHING the workgroup reads streams through one ring of 16 KB LDS units filled by direct global -> LDS copies issued
// NBUF-1 units ahead; a step never waits for more than "the unit after this one has landed" (counted vmcnt, raw s_barrier):
//   A units    : 64 strip rows x 32 k (two 16-k stages), copied to registers when their turn comes
//   B units    : 16 k x 128 rows of the group (two units per stage), 16-byte slots XOR-swizzled by ((row >> 1) & 7);
//                lane (i, q) takes the logical slots q and q + 4 (k = 2q, 2q+1, 8+2q, 9+2q)     -> 32 MFMAs per wave and slab
//   X units    : 64 strip rows x 32 columns of the tile being solved, from the workspace or from the source matrices
//                (first touch: S, and F to subtract), slots swizzled by (row & 15), read back in the xn layout
//   image units: half tiles of the diagonal block (L[c][c'] and the tile inverses) exactly as gfact left them
//     for c:  Xn_c = sum over the K units - X_c;  Xn_c += sum_{c' < c} L[c][c'] (x) T_c';  T_c = -(Linv_cc (x) Xn_c), stored
// NS = 1 is what ships: two workgroups per CU cover each other's barrier and wait phases.  NS = 2 (one workgroup of 128
// rows per CU, 9-unit ring) halves the L2 -> LDS traffic per flop but measured 40 % slower: a lone wave per SIMD cannot
// keep the MFMA pipe fed across the per-unit barriers.
#define GU_UNIT 2048            // doubles per ring unit (16 KB)

template <int N>
__device__ __forceinline__ void gu_wait_upto(int n) {   // s_waitcnt vmcnt(4 * min(n, N)), immediate operands only
  if (N > 0 && n >= N) { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * N) : "memory"); return; }
  if (N > 0) gu_wait_upto<(N > 0 ? N - 1 : 0)>(n);
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int NS>
__global__ __launch_bounds__(256, (NS == 1 ? 2 : 1)) void k_chol_gstrip(double* mats, int64_t mat_stride, int n64, int Ttot,
                                                                      int k0, int nc, const double* dimg, int ngrp,
                                                                      int nitem, int batch, int R, int row_end,
                                                                      FormSrc fs) {
  constexpr int NBUF = NS == 1 ? 5 : 9;     // ring units: 80 KB (two workgroups per CU) or 144 KB
  constexpr int DIST = NBUF - 1;            // prefetch distance in units
  constexpr int PAIR = NS + 4;              // K units per pair of stages: NS A units, 2 x 2 B units
  __shared__ __attribute__((aligned(16))) double ring[NBUF * GU_UNIT];
  int b, g;
  if (!xcd_affine(blockIdx.x, nitem, batch, R, b, g)) return;
  b = __builtin_amdgcn_readfirstlane(b);   // uniform, but the integer divisions leave them in vector registers;
  g = __builtin_amdgcn_readfirstlane(g);   // the copy bases must be scalar
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, q = lane >> 4;
  const int k1 = k0 + nc;
  const int tr0 = k1 + NS * g;              // first tile row of the workgroup; slab t of wave w: rows 64(tr0+t) + 16w + i
  {
    const int T = n64 / CT;
    const int tb = __builtin_amdgcn_readfirstlane(sys_tiles(fs, b, T));
    if (k0 >= tb || (tr0 < T && tr0 >= tb)) return;   // group or tile row in the identity padding of a smaller system
    if (k1 > tb) nc = tb - k0;                        // only the tile columns that hold data are solved for
  }
  double* M = mats + (int64_t)b * mat_stride;
  const double* img = dimg + ((int64_t)b * ngrp + k0 / 4) * GT_NIMG * GT_TILE;
  bool act[NS];                             // wave-uniform: slabs of padding rows only take part in the staging
  double* Xr[NS];
  int trc[NS];                              // a slab row block past the end re-reads the last one (results unused)
#pragma unroll
  for (int t = 0; t < NS; ++t) {
    act[t] = (tr0 + t) * CT + 16 * wave < row_end;
    trc[t] = tr0 + t < Ttot ? tr0 + t : Ttot - 1;
    Xr[t] = M + ((int64_t)trc[t] * CT + 16 * wave + i) * n64 + k0 * CT;
  }
  const int nact = act[NS - 1] ? NS : (act[0] ? 1 : 0);     // slabs are in row order: slab 1 active implies slab 0 active
  // compute bodies are instantiated per active-slab count so that the MFMA streams stay branch-free
  auto with_nact = [&](auto&& f) {
    if (nact == NS) f(std::integral_constant<int, NS>{});
    else if (NS > 1 && nact == 1) f(std::integral_constant<int, 1>{});
  };
  // sources of the strip's own tiles: X = srcA - srcB (srcB optional); strip rows are never on the diagonal
  const double* srcA = M;
  const double* srcB = nullptr;
  if (fs.enabled) {
    const FormIdx fx = form_idx(fs, b);
    if (fx.X && tr0 * CT >= fx.x0) srcA = fx.X - fx.xoff;
    else { srcA = fx.S; srcB = fx.F; }
  }
  const int nsrc = srcB ? 2 : 1;
  const int npair = k0 * 2;                                     // pairs of 16-k stages left of the group (k0 * 64 / 32)
  const int nK = npair * PAIR;
  auto cnt = [&](int c) { return 2 * (NS * nsrc + c + 1); };    // units of tile column c: X (per slab row block), L[c][0..c-1], inverse
  int total = nK;
  for (int c = 0; c < nc; ++c) total += cnt(c);
  // per-lane byte offsets of the pieces this lane copies: B units (2 halves x 4 pieces), A / X units (4 pieces), images
  uint32_t koff[2][4], xoff[4], ioff[4];
  {
    const int maxrow = Ttot * CT - 1;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int lr = (wave * 4 + j) * 8 + (lane >> 3);
        int grow = k0 * CT + h * 128 + lr;
        grow = grow < maxrow ? grow : maxrow;
        const int slot = (lane & 7) ^ ((lr >> 1) & 7);
        koff[h][j] = (uint32_t)(((int64_t)grow * n64 + slot * 2) * 8);
      }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int lr = (wave * 4 + j) * 4 + (lane >> 4);
      const int slot = (lane & 15) ^ (lr & 15);
      xoff[j] = (uint32_t)(((int64_t)lr * n64 + slot * 2) * 8);
      ioff[j] = (uint32_t)((wave * 512 + j * 128 + lane * 2) * 8);
    }
  }
  const uint32_t ring_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) double*)ring;
  const uint32_t wave_lds = __builtin_amdgcn_readfirstlane(ring_lds + wave * 4096);     // this wave's 4 KB of every unit
  // prefetch cursor: units are staged strictly in order
  int pf_u = 0, pf_slot = 0, pf_pp = 0, pf_r = 0, pf_c = 0, pf_v = 0;
  auto stage_next = [&]() {
    const uint32_t dst = wave_lds + pf_slot * (GU_UNIT * 8);
    if (pf_u < nK) {
      if (pf_r < NS) {                      // A unit of slab row block pf_r: 64 rows x 32 k
        const double* base = M + (int64_t)trc[pf_r < NS ? pf_r : 0] * CT * n64 + pf_pp * 32;
#pragma unroll
        for (int j = 0; j < 4; ++j) glds16(base, xoff[j], dst + j * 1024);
      } else {                              // B unit: stage 2 pp + (r' >> 1), half r' & 1
        const int rr = pf_r - NS;
        const double* base = M + (2 * pf_pp + (rr >> 1)) * 16;
        if (rr & 1) {
#pragma unroll
          for (int j = 0; j < 4; ++j) glds16(base, koff[1][j], dst + j * 1024);
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) glds16(base, koff[0][j], dst + j * 1024);
        }
      }
    } else {
      if (pf_v < 2 * NS * nsrc) {           // X unit: (slab row block t, source, column half)
        const int xsh = nsrc == 2 ? 2 : 1;                      // 2 * nsrc is 2 or 4: no integer division in scalar code
        const int t = pf_v >> xsh, rem = pf_v & (2 * nsrc - 1);
        const double* src = (rem >> 1) ? srcB : srcA;
        const double* base = src + (int64_t)trc[t < NS ? t : 0] * CT * n64 + (k0 + pf_c) * CT + (rem & 1) * 32;
#pragma unroll
        for (int j = 0; j < 4; ++j) glds16(base, xoff[j], dst + j * 1024);
      } else {
        const int w = pf_v - 2 * NS * nsrc;
        const double* base = img + (int64_t)(pf_c * (pf_c + 1) / 2 + (w >> 1)) * GT_TILE + (w & 1) * (GT_TILE / 2);
#pragma unroll
        for (int j = 0; j < 4; ++j) glds16(base, ioff[j], dst + j * 1024);
      }
    }
    {  // advance the cursor with plain value selects (conditional ++ of either pair ends up as an indexed stack slot)
      const bool ink = pf_u < nK;
      const int nr = pf_r + 1, nv = pf_v + 1;
      const bool wk = nr == PAIR, wc = nv == cnt(pf_c);
      pf_r = ink ? (wk ? 0 : nr) : pf_r;
      pf_pp = ink ? pf_pp + (wk ? 1 : 0) : pf_pp;
      pf_v = ink ? pf_v : (wc ? 0 : nv);
      pf_c = ink ? pf_c : pf_c + (wc ? 1 : 0);
    }
    ++pf_u;
    pf_slot = pf_slot + 1 == NBUF ? 0 : pf_slot + 1;
  };
  // ring bookkeeping: unit u lives in slot u % NBUF; units 0 .. DIST-1 are issued up front, unit u + DIST at the start of unit u
  int u = 0, cur_slot = 0;
  for (int p = 0; p < DIST && p < total; ++p) stage_next();
  auto unit_begin = [&]() -> const double* {
    if (pf_u < total) stage_next();
    return ring + cur_slot * GU_UNIT;
  };
  auto unit_end = [&]() {            // the next unit must have landed: at most 4 pieces per later unit may be in flight
    gu_wait_upto<DIST - 1>(total - 2 - u);
    // raw barrier: __syncthreads() would fence with vmcnt(0) and drain the copies still in flight for later units
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    ++u;
    cur_slot = cur_slot + 1 == NBUF ? 0 : cur_slot + 1;
  };
  v4d xn[NS][16];
#pragma unroll
  for (int t = 0; t < NS; ++t)
#pragma unroll
    for (int n = 0; n < 16; ++n) xn[t][n] = (v4d){0, 0, 0, 0};
  gu_wait_upto<DIST - 1>(total - 1);   // unit 0
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  // ---- phase 1: K units ----
  {
    const int xs = (i >> 1) & 7;
    const int s0 = ((q ^ xs) << 1), s1 = (((q + 4) ^ xs) << 1);
    const int ob = i * 16;
    for (int pp = 0; pp < npair; ++pp) {
      double2 av[NS][2][2];              // [slab][stage of the pair][k pair]: this lane's strip row, k = 2q, 2q+1 | 8+2q, 9+2q
#pragma unroll
      for (int t = 0; t < NS; ++t) {
        const double* cur = unit_begin();
        const double* rp = cur + (16 * wave + i) * 32;
#pragma unroll
        for (int sp = 0; sp < 2; ++sp) {
          av[t][sp][0] = *reinterpret_cast<const double2*>(rp + (((8 * sp + q) ^ i) << 1));
          av[t][sp][1] = *reinterpret_cast<const double2*>(rp + (((8 * sp + 4 + q) ^ i) << 1));
        }
        unit_end();
      }
#pragma unroll
      for (int sp = 0; sp < 2; ++sp)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const double* cur = unit_begin();
          with_nact([&](auto na) {
            constexpr int NA = decltype(na)::value;
            double2 b0[2][2], b1[2][2];
            auto rd = [&](int j, int set) {
#pragma unroll
              for (int t2 = 0; t2 < 2; ++t2) {
                b0[set][t2] = *reinterpret_cast<const double2*>(cur + ob + (2 * j + t2) * 256 + s0);
                b1[set][t2] = *reinterpret_cast<const double2*>(cur + ob + (2 * j + t2) * 256 + s1);
              }
            };
            rd(0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              if (j < 3) rd(j + 1, (j + 1) & 1);
              const int st = j & 1, n0 = 8 * h + 2 * j;
#pragma unroll
              for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
                  for (int t = 0; t < NA; ++t) {
                    const double bb = kk == 0 ? b0[st][t2].x : (kk == 1 ? b0[st][t2].y : (kk == 2 ? b1[st][t2].x : b1[st][t2].y));
                    const double aa = kk == 0 ? av[t][sp][0].x : (kk == 1 ? av[t][sp][0].y : (kk == 2 ? av[t][sp][1].x : av[t][sp][1].y));
                    xn[t][n0 + t2] = __builtin_amdgcn_mfma_f64_16x16x4f64(bb, aa, xn[t][n0 + t2], 0, 0, 0);
                  }
            }
          });
          unit_end();
        }
    }
  }
  // ---- phase 2: per tile column X units, L tiles, inverse ----
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    if (c >= nc) break;
#pragma unroll
    for (int t = 0; t < NS; ++t)
      for (int src = 0; src < nsrc; ++src) {
        const double sgn = src ? 1.0 : -1.0;          // xn = -X + ..., X = A - B
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const double* cur = unit_begin();
          if (act[t]) {
            const double* rp = cur + (16 * wave + i) * 32 + (q & 1);
#pragma unroll
            for (int nbl = 0; nbl < 2; ++nbl)
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int slot = (8 * nbl + 2 * r + (q >> 1)) ^ i;
                xn[t][4 * c + 2 * h + nbl][r] = fma(sgn, rp[slot << 1], xn[t][4 * c + 2 * h + nbl][r]);
              }
          }
          unit_end();
        }
      }
    // a operand of block (nb, n) in a half-tile image: row 16nb + i, the two 16-byte slots (8(n&1) + 2q + hh) ^ i
    auto a4 = [&](const double* cur, int nb, int nl, double (&a)[4]) {
      const double* rowp = cur + (16 * nb + i) * 32;
      const double2 lo = *reinterpret_cast<const double2*>(rowp + (((8 * nl + 2 * q) ^ i) << 1));
      const double2 hi = *reinterpret_cast<const double2*>(rowp + (((8 * nl + 2 * q + 1) ^ i) << 1));
      a[0] = lo.x; a[1] = lo.y; a[2] = hi.x; a[3] = hi.y;
    };
#pragma unroll
    for (int cp = 0; cp < c; ++cp) {          // Xn_c += L[c][cp] (x) T_cp
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const double* cur = unit_begin();
        with_nact([&](auto na) {
          constexpr int NA = decltype(na)::value;
#pragma unroll
          for (int nl = 0; nl < 2; ++nl)
#pragma unroll
            for (int nbp = 0; nbp < 2; ++nbp) {
              double a[2][4];
              a4(cur, 2 * nbp, nl, a[0]);
              a4(cur, 2 * nbp + 1, nl, a[1]);
#pragma unroll
              for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
                  for (int t = 0; t < NA; ++t)
                    xn[t][4 * c + 2 * nbp + t2] = __builtin_amdgcn_mfma_f64_16x16x4f64(
                        a[t2][r], xn[t][4 * cp + 2 * h + nl][r], xn[t][4 * c + 2 * nbp + t2], 0, 0, 0);
            }
        });
        unit_end();
      }
    }
    {                                         // T_c = -(Linv_cc (x) Xn_c), lower triangular: blocks n <= nb
      v4d out[NS][4];
#pragma unroll
      for (int t = 0; t < NS; ++t)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) out[t][nb] = (v4d){0, 0, 0, 0};
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const double* cur = unit_begin();
        with_nact([&](auto na) {
          constexpr int NA = decltype(na)::value;
#pragma unroll
          for (int nl = 0; nl < 2; ++nl)
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
              const int n = 2 * h + nl;
              if (n > nb) continue;
              double a[4];
              a4(cur, nb, nl, a);
#pragma unroll
              for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int t = 0; t < NA; ++t)
                  out[t][nb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[r], xn[t][4 * c + n][r], out[t][nb], 0, 0, 0);
            }
        });
        if (h == 1) {
#pragma unroll
          for (int t = 0; t < NS; ++t) {
            if (!act[t]) continue;
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) xn[t][4 * c + nb] = -out[t][nb];
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)
#pragma unroll
              for (int r = 0; r < 4; ++r) Xr[t][c * CT + 16 * nb + q + 4 * r] = xn[t][4 * c + nb][r];
          }
        }
        unit_end();
      }
    }
  }
}

// ---- back substitution L^T x = y for the RHS rows (in place), one workgroup per system -------------
// The four waves split either the RHS rows (pg = 4 or 2 per pass) or, for few RHS, the 64-row
// contraction of each tile (parts = 4 / pg), so a single right-hand side still keeps 4 waves of
// loads in flight.
__global__ __launch_bounds__(256) void k_chol_backsolve(double* mats, int64_t mat_stride, int n64,
                                                        int nrhs, const double* dinv, FormSrc fs) {
  __shared__ double xk[4][CT];
  __shared__ double yk[4][CT];
  __shared__ double red[4][CT];
  const int b = blockIdx.x;
  const int Tfull = n64 / CT;
  const int T = sys_tiles(fs, b, Tfull);    // solution entries past the system's own order are not produced (nor read)
  const int w = threadIdx.x >> 6, c = threadIdx.x & 63;
  const int pg = nrhs >= 4 ? 4 : (nrhs >= 2 ? 2 : 1);
  const int parts = 4 / pg;
  const int p = w % pg, part = w / pg;
  const int rlen = CT / parts, r0 = part * rlen;
  double* M = mats + (int64_t)b * mat_stride;
  // embedded right-hand sides: y is row n + p of the system, its entries past column n belong to the factor of the padding
  int nsys = n64;
  if (fs.embed) {
    const int bb = b + fs.b_offset;
    nsys = fs.d_n[(bb / (fs.nfold * fs.nshift)) / fs.n_div];
  }
  for (int p0 = 0; p0 < nrhs; p0 += pg) {
    const bool act = (p0 + p) < nrhs;
    double* Y = M + (int64_t)(nsys + p0 + p) * n64;
    for (int k = T - 1; k >= 0; --k) {
      __syncthreads();
      if (part == 0) yk[p][c] = (act && k * CT + c < nsys) ? Y[k * CT + c] : 0.0;
      __syncthreads();
      const double* I = dinv + ((int64_t)b * Tfull + k) * CT * CT;
      double x = 0.0;
      for (int r = r0; r < r0 + rlen; ++r) x = fma(yk[p][r], I[r * CT + c], x);  // Linv is lower: zeros above
      red[w][c] = x;
      __syncthreads();
      if (part == 0) {
        double t = 0.0;
        for (int s2 = 0; s2 < parts; ++s2) t += red[p + s2 * pg][c];
        xk[p][c] = t;
        if (act) Y[k * CT + c] = t;
      }
      __syncthreads();
      for (int j = 0; j < k; ++j) {
        const double* Lt = M + (int64_t)(k * CT + r0) * n64 + j * CT + c;
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        for (int r = 0; r < rlen; r += 4) {
          a0 = fma(xk[p][r0 + r], Lt[(int64_t)r * n64], a0);
          a1 = fma(xk[p][r0 + r + 1], Lt[(int64_t)(r + 1) * n64], a1);
          a2 = fma(xk[p][r0 + r + 2], Lt[(int64_t)(r + 2) * n64], a2);
          a3 = fma(xk[p][r0 + r + 3], Lt[(int64_t)(r + 3) * n64], a3);
        }
        const double v = (a0 + a1) + (a2 + a3);
        if (parts == 1) {
          if (act) Y[j * CT + c] -= v;
        } else {
          __syncthreads();
          red[w][c] = v;
          __syncthreads();
          if (part == 0 && act) {
            double t = 0.0;
            for (int s2 = 0; s2 < parts; ++s2) t += red[p + s2 * pg][c];
            Y[j * CT + c] -= t;
          }
        }
      }
    }
  }
}

// ---- back substitution for SEVERAL right-hand sides on the matrix cores ---------------------------------------------------------------
// k_chol_backsolve above walks L once per group of four right-hand sides with VALU dot products: at BASELINE configs[2] (P = 10) that is three
// passes over every factor and 31 % of the level-0 Cholesky's time with no matrix instruction in it.  Here ONE pass serves sixteen
// right-hand sides.  One workgroup per system; wave w OWNS the solution tiles j = w, w + 4, w + 8, w + 12 (64 entries x 16 right-hand
// sides each) and keeps them in registers from the first touch to the last: the only memory traffic is L itself, read once, 32 bytes
// per lane and K step.  Walking the tile rows upwards, for k = T-1 .. 0:
//   the owner of k finishes x_k = y_k Linv_kk   (y_k through LDS into A-operand layout; 64 products 16 x 16 x 4)
//   barrier; every wave subtracts x_k L[k][j] from the tiles j < k it owns (A = -x_k from LDS, B = the tile of L straight from memory)
// v_mfma_f64_16x16x4: lane (i, q) supplies A[right-hand side i][k = 4 s + q] and B[k = 4 s + q][n = i]; a lane loads FOUR consecutive
// columns 4 i .. 4 i + 3 of row 4 s + q (one 32-byte load) and feeds component t to the product whose output columns are {4 i' + t}:
// the four products of a K step then cover the tile's 64 columns, and lane (i, q) owns x[rhs q + 4 r][column 4 i + t] in acc[.][t][r].
#define BS_PITCH 66
__global__ __launch_bounds__(256, 2) void k_chol_backsolve_mfma(double* mats, int64_t mat_stride, int n64, int nrhs, int rhs_row0,
                                                                const double* dinv, FormSrc fs) {
  __shared__ double xk[2][16][BS_PITCH];
  __shared__ double ytmp[4][16][BS_PITCH];
  const int b = blockIdx.x;
  const int Tfull = n64 / CT;
  const int T = sys_tiles(fs, b, Tfull);
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 15, q = lane >> 4;
  double* M = mats + (int64_t)b * mat_stride;
  int nsys = n64, yrow0 = rhs_row0;
  if (fs.embed) {      // embedded right-hand sides: y is row n + p of the system, its entries past column n belong to the factor of the padding
    const int bb = b + fs.b_offset;
    nsys = fs.d_n[(bb / (fs.nfold * fs.nshift)) / fs.n_div];
    yrow0 = nsys;
  }
  for (int p0 = 0; p0 < nrhs; p0 += 16) {
    const int np = nrhs - p0 < 16 ? nrhs - p0 : 16;
    v4d acc[4][4];      // [owned tile jj][column phase t]: rows (right-hand sides) q + 4 r
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int j = 4 * jj + w;
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int pr = q + 4 * r, col = j * CT + 4 * i + t;
          const bool ok = j < T && pr < np && col < nsys;
          const int64_t e = ok ? (int64_t)(yrow0 + p0 + pr) * n64 + col : 0;      // clamped address, masked value
          const double v = M[e];
          acc[jj][t][r] = ok ? v : 0.0;
        }
    }
    for (int k = T - 1; k >= 0; --k) {
      const int ow = k & 3, kj = k >> 2;
      if (w == ow) {     // x_k = y_k Linv_kk
        double (*yt)[BS_PITCH] = ytmp[w];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
          if (jj == kj) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
              for (int r = 0; r < 4; ++r) yt[q + 4 * r][4 * i + t] = acc[jj][t][r];
          }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // same wave writes and reads: no barrier needed
        const double* I = dinv + ((int64_t)b * Tfull + k) * CT * CT + (int64_t)q * CT + 4 * i;
        v4d out[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) out[t] = (v4d){0, 0, 0, 0};
#pragma unroll 1
        for (int s8 = 0; s8 < 4; ++s8) {      // four K steps at a time: the owned tiles and `out` leave room for 32 operand registers
          double4 bv[4];
#pragma unroll
          for (int s = 0; s < 4; ++s) bv[s] = *reinterpret_cast<const double4*>(I + (int64_t)(4 * (4 * s8 + s)) * CT);
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const double a = yt[i][4 * (4 * s8 + s) + q];
            out[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bv[s].x, out[0], 0, 0, 0);
            out[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bv[s].y, out[1], 0, 0, 0);
            out[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bv[s].z, out[2], 0, 0, 0);
            out[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bv[s].w, out[3], 0, 0, 0);
          }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int pr = q + 4 * r, col = 4 * i + t;
            xk[k & 1][pr][col] = -out[t][r];                     // stored negated: the updates below add A x B
            if (pr < np) M[(int64_t)(yrow0 + p0 + pr) * n64 + k * CT + col] = out[t][r];
          }
      }
      __syncthreads();
      if (k == 0) break;
      const double (*xs)[BS_PITCH] = xk[k & 1];
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const int j = 4 * jj + w;
        if (j >= k) continue;
        const double* Lp = M + ((int64_t)k * CT + q) * n64 + j * CT + 4 * i;
#pragma unroll
        for (int s8 = 0; s8 < 2; ++s8) {
          double4 bv[8];
#pragma unroll
          for (int s = 0; s < 8; ++s) bv[s] = *reinterpret_cast<const double4*>(Lp + (int64_t)(4 * (8 * s8 + s)) * n64);
#pragma unroll
          for (int s = 0; s < 8; ++s) {
            const double a = xs[i][4 * (8 * s8 + s) + q];
            acc[jj][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bv[s].x, acc[jj][0], 0, 0, 0);
            acc[jj][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bv[s].y, acc[jj][1], 0, 0, 0);
            acc[jj][2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bv[s].z, acc[jj][2], 0, 0, 0);
            acc[jj][3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bv[s].w, acc[jj][3], 0, 0, 0);
          }
        }
      }
    }
    __syncthreads();      // the next group of right-hand sides reuses xk
  }
}

// ---- back substitution for FEW systems: one launch per tile row k, one workgroup per tile column j < k -----------------
// k_chol_backsolve walks a system with a single workgroup: 2,346 dependent tile steps for order 4,416 (the shared level 1 of an
// 8-GPU run: 6 ms of the 15 ms its four systems took).  Here launch k applies x_k to every y_j, j < k, in parallel
// (y_j -= L[k][j]^T x_k, the four waves splitting the tile's 64 rows), and the workgroup of column k-1 then finishes
// x_{k-1} = Linv_{k-1}^T y_{k-1} in place, so the next launch finds it ready.  T launches of T-k workgroups per system.
// k == T: only the final step for row T-1 (grid (1, batch)).
__global__ __launch_bounds__(256) void k_chol_backsolve_row(double* mats, int64_t mat_stride, int n64, int nrhs, const double* dinv,
                                                            int k) {
  __shared__ double xs[CT];
  __shared__ double red[4][CT];
  __shared__ double ys[CT];
  const int T = n64 / CT;
  const int b = blockIdx.y, j = (k == T) ? T - 1 : (int)blockIdx.x;
  const int w = threadIdx.x >> 6, c = threadIdx.x & 63;
  double* M = mats + (int64_t)b * mat_stride;
  double l[16];
  if (k < T) {
    const double* Lt = M + (int64_t)(k * CT + w * 16) * n64 + j * CT + c;
#pragma unroll
    for (int r = 0; r < 16; ++r) l[r] = Lt[(int64_t)r * n64];
  }
  const bool finish = (k == T) || (j == k - 1);
  const double* I = dinv + ((int64_t)b * T + j) * CT * CT;
  for (int p = 0; p < nrhs; ++p) {
    double* Y = M + (int64_t)(n64 + p) * n64;
    double yv = 0.0;
    if (k < T) {
      __syncthreads();
      if (w == 0) xs[c] = Y[k * CT + c];
      __syncthreads();
      double a = 0.0;
#pragma unroll
      for (int r = 0; r < 16; ++r) a = fma(xs[w * 16 + r], l[r], a);
      red[w][c] = a;
      __syncthreads();
      if (w == 0) {
        yv = Y[j * CT + c] - ((red[0][c] + red[1][c]) + (red[2][c] + red[3][c]));
        if (!finish) Y[j * CT + c] = yv;
      }
    } else if (w == 0) {
      yv = Y[j * CT + c];
    }
    if (finish) {   // x_j = Linv_j^T y_j (Linv is lower triangular: zeros above the diagonal as stored)
      __syncthreads();
      if (w == 0) ys[c] = yv;
      __syncthreads();
      double x = 0.0;
#pragma unroll
      for (int r = 0; r < 16; ++r) x = fma(ys[w * 16 + r], I[(w * 16 + r) * CT + c], x);
      red[w][c] = x;
      __syncthreads();
      if (w == 0) Y[j * CT + c] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
    }
  }
}

static void launch_backsolve_rows(hipStream_t st, double* mats, int64_t mat_stride, int batch, int n64, int nrhs, const double* dinv,
                                  int64_t& nl) {
  const int T = n64 / CT;
  hipLaunchKernelGGL(k_chol_backsolve_row, dim3(1, batch), dim3(256), 0, st, mats, mat_stride, n64, nrhs, dinv, T);
  ++nl;
  for (int k = T - 1; k >= 1; --k) {
    hipLaunchKernelGGL(k_chol_backsolve_row, dim3(k, batch), dim3(256), 0, st, mats, mat_stride, n64, nrhs, dinv, k);
    ++nl;
  }
}

#include "chol_p128.h"

void rg_launch_chol_solve_src(hipStream_t st, double* mats, int64_t mat_stride, int batch, int n64,
                              int rhs_pad, int nrhs, double* dinv, int32_t* info, int64_t* n_launch,
                              const FormSrc* src, int path) {
  const int T = n64 / CT, Tr = rhs_pad / CT, Ttot = T + Tr;
  FormSrc off{};
  off.enabled = 0; off.extra = nullptr; off.n_div = 1; off.b_offset = 0; off.skip_pad = 0; off.d_n = nullptr; off.nfold = off.nshift = 1;
  off.embed = 0;
  int64_t nl = 0;
  // path: 0 = group-wise (throughput: level 0, whatever the batch size, so that results do not depend on how the blocks
  // are batched), 1 = per-column (latency: level 1 and the logistic steps, a few dozen systems), -1 = by batch size
  // (test entry; RG_CHOL_SMALL moves the threshold)
  int small_max = path == 1 ? 0x7fffffff : -1;
  if (path < 0) {
    const char* se = getenv("RG_CHOL_SMALL");
    small_max = se ? atoi(se) : 64;
  }
  if (batch <= small_max) {   // small-batch path: per-column diag / panel launches, wide updates per group
    const int G = 4;
  for (int k0 = 0; k0 < T; k0 += G) {
    const int k1 = std::min(T, k0 + G);
    // first touch of every tile happens in the first group: tile columns < G by diag/panel, the rest by the
    // wide update; systems that share a source matrix (the shifts) are co-located on one XCD for those launches
    const FormSrc& first = (src && k0 == 0) ? *src : off;
    const int R = (src && k0 == 0) ? std::max(1, src->nshift) : 1;
    for (int j = k0; j < k1; ++j) {
      hipLaunchKernelGGL(k_chol_diag, dim3(batch), dim3(256), 0, st, mats, mat_stride, n64, j, k0, j - k0, dinv,
                         info, first);
      ++nl;
      if (Ttot - 1 - j > 0) {
        const int npt = Ttot - 1 - j;   // panel tiles of this column, four per workgroup
        hipLaunchKernelGGL(k_chol_panel, dim3(xcd_affine_grid((npt + 3) / 4, batch, R)), dim3(256), 0, st, mats,
                           mat_stride, n64, j, k0, j - k0, dinv, (npt + 3) / 4, batch, R, npt, first);
        ++nl;
      }
    }
    if (k1 < T) {  // wide trailing update with the whole group (K = 64 * (k1 - k0))
      int ntile = 0;
      for (int c = k1; c < T; ++c) ntile += Ttot - c;
      hipLaunchKernelGGL(k_chol_update, dim3(xcd_affine_grid((ntile + 3) / 4, batch, R)), dim3(256), 0, st, mats,
                         mat_stride, n64, Ttot, k1, T, ntile, k0, k1 - k0, batch, R, first);
      ++nl;
    }
  }
  if (nrhs > 0) {
    // few, large systems: row-parallel back substitution; many small ones: one workgroup per system
    if ((int64_t)batch * 4 <= 256 && T >= 8) launch_backsolve_rows(st, mats, mat_stride, batch, n64, nrhs, dinv, nl);
    else {
      hipLaunchKernelGGL(k_chol_backsolve, dim3(batch), dim3(256), 0, st, mats, mat_stride, n64, nrhs, dinv, off);
      ++nl;
    }
  }
  if (n_launch) *n_launch += nl;
    return;
  }

  // First touch of a tile (read from the source matrices instead of the workspace): group 0 by gfact/gstrip, later groups
  // by the diagonal-block update and gstrip.  Systems that share a source matrix (the shifts) are co-located on one XCD.
  const int R = src ? std::max(1, src->nshift) : 1;
  const int ngrp = (T + 3) / 4;
  // right-hand-side rows beyond the ones that are solved for are padding: their 16-row slabs are skipped.  (nrhs == 0:
  // forward substitution only, every appended row is real -- the LOOCV / inverse callers.)
  const int row_end = n64 + (nrhs > 0 ? nrhs : rhs_pad);
  double* dimg = dinv + chol_ws_img_offset((size_t)batch, n64);   // the workspace holds the tile inverses, then the images
  FormSrc first = src ? *src : off;
  first.skip_pad = (src && src->d_n && !getenv("RG_CHOL_FULLPAD")) ? 1 : 0;
  FormSrc later = first;          // launches past a tile's first touch: workspace values, but still the per-system orders
  later.enabled = 0;
  // round 6: panels of 128 columns, one launch per panel (chol_p128.h) -- whenever the right-hand sides are embedded (or absent), the order
  // is a multiple of 128 and every tile is formed from the sources; RG_CHOL_GROUP4=1 keeps the group-of-four kernels below
  static const bool group4 = getenv("RG_CHOL_GROUP4") && atoi(getenv("RG_CHOL_GROUP4")) != 0;
  const bool p128 = !group4 && src && src->enabled && !src->extra && Ttot == T && n64 % 128 == 0;
  if (p128) c128_launch_factor(st, mats, mat_stride, batch, n64, dinv, dimg, info, first, R, nl);
  for (int k0 = 0; k0 < T && !p128; k0 += 4) {
    const int nc = std::min(4, T - k0), k1 = k0 + nc;
    if (k0 > 0) {   // diagonal block: tiles (r, c), k0 <= c <= r < k1, K = 64 * k0
      const int ntile = nc * (nc + 1) / 2;
      hipLaunchKernelGGL(k_chol_update, dim3(xcd_affine_grid((ntile + 3) / 4, batch, R)), dim3(256), 0, st, mats,
                         mat_stride, n64, k1, k0, k1, ntile, 0, k0, batch, R, first);
      ++nl;
    }
    hipLaunchKernelGGL(k_chol_gfact, dim3((batch + 3) / 4), dim3(256), 0, st, mats, mat_stride, n64, k0, nc, batch, dinv,
                       dimg, ngrp, info, k0 == 0 ? first : later);
    ++nl;
    if (Ttot > k1) {
      const int nitem = Ttot - k1;
      hipLaunchKernelGGL(k_chol_gstrip<1>, dim3(xcd_affine_grid(nitem, batch, R)), dim3(256), 0, st, mats, mat_stride,
                         n64, Ttot, k0, nc, dimg, ngrp, nitem, batch, R, row_end, first);
      ++nl;
    }
  }
  if (nrhs > 0) {
    // several right-hand sides and at most 16 tile rows: one pass over L on the matrix cores; else the VALU kernel (a single right-hand
    // side is bound by reading L either way).  RG_BACKSOLVE_VALU=1 keeps the VALU kernel.
    static const bool valu = getenv("RG_BACKSOLVE_VALU") && atoi(getenv("RG_BACKSOLVE_VALU")) != 0;
    if (nrhs >= 2 && T <= 16 && !valu)
      hipLaunchKernelGGL(k_chol_backsolve_mfma, dim3(batch), dim3(256), 0, st, mats, mat_stride, n64, nrhs, n64, dinv, later);
    else
      hipLaunchKernelGGL(k_chol_backsolve, dim3(batch), dim3(256), 0, st, mats, mat_stride, n64, nrhs, dinv, later);
    ++nl;
  }
  if (n_launch) *n_launch += nl;
}

void rg_launch_chol_solve(hipStream_t st, double* mats, int64_t mat_stride, int batch, int n64,
                          int rhs_pad, int nrhs, double* dinv, int32_t* info, int64_t* n_launch, int path) {
  rg_launch_chol_solve_src(st, mats, mat_stride, batch, n64, rhs_pad, nrhs, dinv, info, n_launch, nullptr, path);
}

// General form: subtract = 0 drops the held-out-fold term (LOOCV); rows >= extra_row0 of every system are
// read from extra[outer] (shared by the nfold*nshift systems of one outer index).
void rg_launch_chol_solve_formed_x(hipStream_t st, const double* sum, int64_t sum_stride, const double* fold,
                                   int64_t fold_stride, int nfold, const double* shift, int nshift,
                                   const int32_t* d_n, int n_fixed, int nouter, double* mats,
                                   int64_t mat_stride, int n64, int rhs_pad, int nrhs, double* dinv,
                                   int32_t* info, int64_t* n_launch, int subtract, const double* extra,
                                   int64_t extra_stride, int extra_row0, int n_div, int b_offset, int b_count, int path, int embed) {
  FormSrc f;
  f.embed = (embed > 0 && d_n && path == 0) ? embed : 0;
  f.sum = sum; f.sum_stride = sum_stride; f.fold = fold; f.fold_stride = fold_stride; f.shift = shift;
  f.d_n = d_n; f.nfold = nfold; f.nshift = nshift; f.n_fixed = n_fixed; f.enabled = 1;
  f.subtract = subtract; f.extra = extra; f.extra_stride = extra_stride; f.extra_row0 = extra_row0; f.n64 = n64;
  f.n_div = n_div; f.b_offset = b_offset; f.skip_pad = 0;
  rg_launch_chol_solve_src(st, mats, mat_stride, b_count >= 0 ? b_count : nouter * nfold * nshift, n64, rhs_pad, nrhs,
                           dinv, info, n_launch, &f, path);
}
