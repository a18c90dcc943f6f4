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
// substitution happens for free as part of the panel/update steps (the RHS rows are just one more
// row tile).  Padded diagonal entries are 1.
//
// Right-looking tile algorithm, tile 64: per step k
//   k_chol_diag   : factor the 64x64 diagonal tile in LDS, also emit its inverse
//   k_chol_panel  : L[t][k] = A[t][k] * inv(L[k][k])^T               (fp64 MFMA, NT form)
//   k_chol_update : A[r][c] -= L[r][k] * L[c][k]^T  for k < c <= r    (fp64 MFMA, NT form)
// then k_chol_backsolve (one workgroup per system) solves L^T x = y for the RHS rows.
//
// fp64 MFMA (v_mfma_f64_16x16x4_f64): lane l supplies A[i = l&15][k = l>>4] and B[k = l>>4][j = l&15]
// and owns D[row = (l>>4) + 4*reg][col = l&15].  Both operands are K-contiguous rows here, and a
// contraction is invariant under any K permutation applied to both operands alike, so lane (i, q)
// takes the 16 CONSECUTIVE k = 16q .. 16q+15 of a 64-chunk (one 128-byte piece of its row): four
// 32-byte global loads per row, no LDS staging.
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

void rg_launch_dgemm_nt(hipStream_t st, const double* A, int64_t lda, const double* B, int64_t ldb,
                        int m, int n, int64_t k, double* C, int64_t ldc) {
  hipLaunchKernelGGL(k_dgemm_nt, dim3(n / CT, m / CT), dim3(256), 0, st, A, lda, B, ldb, k, C, ldc);
}

// ---- form: wk[(o, f, r)] = sum[o] - fold[o][f] + shift[r] on the diagonal ----------------------
__global__ __launch_bounds__(256) void k_form(const double* sum, int64_t sum_stride,
                                              const double* fold, int64_t fold_stride, int nfold,
                                              const double* shift, int nshift, const int32_t* d_n,
                                              int n_fixed, int n64, int rtot, double* wk) {
  const int o = blockIdx.z;
  const int f = blockIdx.y / nshift, r = blockIdx.y % nshift;
  const int n = d_n ? d_n[o] : n_fixed;
  const int64_t msz = (int64_t)rtot * n64;
  const double* S = sum + (int64_t)o * sum_stride;
  const double* F = fold + ((int64_t)o * nfold + f) * fold_stride;
  double* W = wk + (((int64_t)o * nfold + f) * nshift + r) * msz;
  const double sh = shift[r];
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < msz; e += (int64_t)gridDim.x * 256) {
    const int i = (int)(e / n64), j = (int)(e % n64);
    double v = 0.0;
    if (i >= n64 || j <= i) {
      v = S[e] - F[e];
      if (i == j) v = (i < n) ? v + sh : 1.0;
    }
    W[e] = v;
  }
}

void rg_launch_form(hipStream_t st, const double* sum, int64_t sum_stride, const double* fold,
                    int64_t fold_stride, int nfold, const double* shift, int nshift,
                    const int32_t* d_n, int n_fixed, int nouter, int n64, int rtot, double* wk) {
  const int64_t msz = (int64_t)rtot * n64;
  int gx = (int)((msz + 256 * 8 - 1) / (256 * 8));
  if (gx < 1) gx = 1;
  hipLaunchKernelGGL(k_form, dim3(gx, nfold * nshift, nouter), dim3(256), 0, st, sum, sum_stride,
                     fold, fold_stride, nfold, shift, nshift, d_n, n_fixed, n64, rtot, wk);
}

// ---- diagonal tile: potf2 + triangular inverse in LDS ---------------------------------------------
__global__ __launch_bounds__(256) void k_chol_diag(double* mats, int64_t mat_stride, int n64, int k,
                                                   double* dinv, int32_t* info) {
  __shared__ double s[CT][CT + 1];
  __shared__ double si[CT][CT + 1];
  const int b = blockIdx.x;
  const int tid = threadIdx.x;
  double* D = mats + (int64_t)b * mat_stride + (int64_t)k * CT * n64 + k * CT;
  for (int e = tid; e < CT * CT; e += 256) {
    const int r = e >> 6, c = e & 63;
    s[r][c] = (c <= r) ? D[(int64_t)r * n64 + c] : 0.0;
    si[r][c] = 0.0;
  }
  __syncthreads();
  // left-looking column Cholesky: 4 threads per row split the dot product
  const int r = tid >> 2, part = tid & 3;
  for (int c = 0; c < CT; ++c) {
    double p = 0.0;
    for (int j = part; j < c; j += 4) p = fma(s[r][j], s[c][j], p);
    p += __shfl_xor(p, 1);
    p += __shfl_xor(p, 2);
    __syncthreads();
    if (part == 0 && r >= c) s[r][c] -= p;
    __syncthreads();
    const double piv = s[c][c];
    double d = sqrt(piv);
    if (!(piv > 0.0)) {
      d = 1.0;
      if (tid == 0) atomicMax(info, 1);
    }
    __syncthreads();
    if (part == 0) {
      if (r > c) s[r][c] /= d;
      else if (r == c) s[r][c] = d;
    }
    __syncthreads();
  }
  // inverse of the lower-triangular factor: thread (c = tid>>2, part) builds column c
  {
    const int c = tid >> 2;
    for (int rr = 0; rr < CT; ++rr) {
      double p = 0.0;
      for (int j = part; j < rr; j += 4) p = fma(s[rr][j], si[j][c], p);
      p += __shfl_xor(p, 1);
      p += __shfl_xor(p, 2);
      if (part == 0) si[rr][c] = (((rr == c) ? 1.0 : 0.0) - p) / s[rr][rr];
      __syncthreads();
    }
  }
  double* I = dinv + ((int64_t)b * (n64 / CT) + k) * CT * CT;
  for (int e = tid; e < CT * CT; e += 256) {
    const int rr = e >> 6, c = e & 63;
    if (c <= rr) D[(int64_t)rr * n64 + c] = s[rr][c];
    I[e] = (c <= rr) ? si[rr][c] : 0.0;
  }
}

// ---- panel: L[t][k] = A[t][k] * Linv^T ---------------------------------------------------------------
__global__ __launch_bounds__(256) void k_chol_panel(double* mats, int64_t mat_stride, int n64, int k,
                                                    const double* dinv) {
  const int b = blockIdx.y, t = k + 1 + blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int i = lane & 15, q = lane >> 4;
  double* T = mats + (int64_t)b * mat_stride + (int64_t)t * CT * n64 + k * CT;
  const double* I = dinv + ((int64_t)b * (n64 / CT) + k) * CT * CT;
  const double* ar[2] = {T + (int64_t)(wr * 32 + i) * n64 + 16 * q,
                         T + (int64_t)(wr * 32 + 16 + i) * n64 + 16 * q};
  const double* br[2] = {I + (wc * 32 + i) * CT + 16 * q, I + (wc * 32 + 16 + i) * CT + 16 * q};
  double av[2][16], bv[2][16];
  dmma_load<2, 2>(ar, br, av, bv);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();  // every wave holds its operands before any wave overwrites the tile
  v4d acc[2][2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n) acc[m][n] = (v4d){0, 0, 0, 0};
  dmma_fma<2, 2>(av, bv, acc);
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        T[(int64_t)(wr * 32 + m * 16 + q + 4 * r) * n64 + wc * 32 + n * 16 + i] = acc[m][n][r];
}

// ---- trailing update: A[r][c] -= L[r][k] L[c][k]^T -----------------------------------------------------
__global__ __launch_bounds__(256) void k_chol_update(double* mats, int64_t mat_stride, int n64,
                                                     int ntile_mat, int k) {
  const int b = blockIdx.y;
  const int nrem = ntile_mat - 1 - k;
  const int tri = nrem * (nrem + 1) / 2;
  int idx = blockIdx.x, tr, tc;
  if (idx < tri) {
    int rr = (int)((sqrtf(8.0f * idx + 1.0f) - 1.0f) * 0.5f);
    while ((rr + 1) * (rr + 2) / 2 <= idx) ++rr;
    while (rr * (rr + 1) / 2 > idx) --rr;
    tr = k + 1 + rr;
    tc = k + 1 + (idx - rr * (rr + 1) / 2);
  } else {
    idx -= tri;
    tr = ntile_mat + idx / nrem;
    tc = k + 1 + idx % nrem;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int i = lane & 15, q = lane >> 4;
  double* M = mats + (int64_t)b * mat_stride;
  const double* A = M + (int64_t)tr * CT * n64 + k * CT;
  const double* B = M + (int64_t)tc * CT * n64 + k * CT;
  double* C = M + (int64_t)tr * CT * n64 + tc * CT;
  const double* ar[2] = {A + (int64_t)(wr * 32 + i) * n64 + 16 * q,
                         A + (int64_t)(wr * 32 + 16 + i) * n64 + 16 * q};
  const double* br[2] = {B + (int64_t)(wc * 32 + i) * n64 + 16 * q,
                         B + (int64_t)(wc * 32 + 16 + i) * n64 + 16 * q};
  double av[2][16], bv[2][16];
  dmma_load<2, 2>(ar, br, av, bv);
  v4d acc[2][2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        acc[m][n][r] = -C[(int64_t)(wr * 32 + m * 16 + q + 4 * r) * n64 + wc * 32 + n * 16 + i];
  dmma_fma<2, 2>(av, bv, acc);
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        C[(int64_t)(wr * 32 + m * 16 + q + 4 * r) * n64 + wc * 32 + n * 16 + i] = -acc[m][n][r];
}

// ---- back substitution L^T x = y for the RHS rows (in place), one workgroup per system -------------
#define BS_PG 4  // RHS rows per pass: 4 x 64 outputs = 256 threads
__global__ __launch_bounds__(256) void k_chol_backsolve(double* mats, int64_t mat_stride, int n64,
                                                        int nrhs, const double* dinv) {
  __shared__ double xk[BS_PG][CT];
  __shared__ double yk[BS_PG][CT];
  const int b = blockIdx.x;
  const int T = n64 / CT;
  const int p = threadIdx.x >> 6, c = threadIdx.x & 63;
  double* M = mats + (int64_t)b * mat_stride;
  for (int p0 = 0; p0 < nrhs; p0 += BS_PG) {
    const bool act = (p0 + p) < nrhs;
    double* Y = M + (int64_t)(n64 + p0 + p) * n64;  // this thread's RHS row
    for (int k = T - 1; k >= 0; --k) {
      __syncthreads();
      yk[p][c] = act ? Y[k * CT + c] : 0.0;
      __syncthreads();
      // x_k[c] = sum_r y_k[r] * Linv[r][c]   (Linv lower: r >= c)
      const double* I = dinv + ((int64_t)b * T + k) * CT * CT;
      double x = 0.0;
      for (int r = c; r < CT; ++r) x = fma(yk[p][r], I[r * CT + c], x);
      xk[p][c] = x;
      if (act) Y[k * CT + c] = x;
      __syncthreads();
      // y_j[c] -= sum_r x_k[r] * L[k*64 + r][j*64 + c]   for j < k
      for (int j = 0; j < k; ++j) {
        const double* Lt = M + (int64_t)k * CT * n64 + j * CT + c;
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        for (int r = 0; r < CT; r += 4) {
          a0 = fma(xk[p][r], Lt[(int64_t)r * n64], a0);
          a1 = fma(xk[p][r + 1], Lt[(int64_t)(r + 1) * n64], a1);
          a2 = fma(xk[p][r + 2], Lt[(int64_t)(r + 2) * n64], a2);
          a3 = fma(xk[p][r + 3], Lt[(int64_t)(r + 3) * n64], a3);
        }
        if (act) Y[j * CT + c] -= (a0 + a1) + (a2 + a3);
      }
    }
  }
}

void rg_launch_chol_solve(hipStream_t st, double* mats, int64_t mat_stride, int batch, int n64,
                          int rhs_pad, int nrhs, double* dinv, int32_t* info, int64_t* n_launch) {
  const int T = n64 / CT, Tr = rhs_pad / CT;
  int64_t nl = 0;
  for (int k = 0; k < T; ++k) {
    hipLaunchKernelGGL(k_chol_diag, dim3(batch), dim3(256), 0, st, mats, mat_stride, n64, k, dinv, info);
    ++nl;
    const int below = T - 1 - k + Tr;
    if (below > 0) {
      hipLaunchKernelGGL(k_chol_panel, dim3(below, batch), dim3(256), 0, st, mats, mat_stride, n64, k, dinv);
      ++nl;
    }
    const int nrem = T - 1 - k;
    const int ntile = nrem * (nrem + 1) / 2 + Tr * nrem;
    if (ntile > 0) {
      hipLaunchKernelGGL(k_chol_update, dim3(ntile, batch), dim3(256), 0, st, mats, mat_stride, n64, T, k);
      ++nl;
    }
  }
  if (nrhs > 0) {
    hipLaunchKernelGGL(k_chol_backsolve, dim3(batch), dim3(256), 0, st, mats, mat_stride, n64, nrhs, dinv);
    ++nl;
  }
  if (n_launch) *n_launch += nl;
}
