// Level-0 leave-one-out predictors through ONE orthogonal reduction per block (reference src/Data.cpp:755-767 +
// src/Step1_Models.cpp:615-726).
//
// The reference diagonalises A = G G^T = V D V^T once per block and serves every ridge value from it:
//     h_ir = sum_k (V^T g_i)_k^2 / (d_k + lambda_r),     num_irp = sum_k (V^T g_i)_k (V^T G Y)_kp / (d_k + lambda_r).
// What makes that cheap is not the diagonal form but that ONE orthogonal transform of the N genotype columns serves all R0 shifts.  A
// symmetric TRIDIAGONAL form does the same: with A = Q T Q^T (Householder, no iteration, no convergence test),
//     (A + lambda I)^-1 = Q (T + lambda I)^-1 Q^T,      T + lambda I = L_r D_r L_r^T  (unit lower BIdiagonal L_r, 2 n flops),
//     z_i = Q^T g_i,   y = L_r^-1 z_i  (a two-term recurrence),   h_ir = sum_k y_k^2 / delta_rk,   num_irp = sum_k y_k u_rpk / delta_rk
// with u_rp = L_r^-1 Q^T (G Y)_p.  Per block: 4/3 bs^3 flops for the reduction (1 % of the work), ONE fp64 GEMM Z = Q^T G~ of 2 N bs^2 flops
// on the matrix cores, and N bs R0 (3 + P) fused multiply-adds of recurrences -- against R0 triangular solves with N right-hand sides each
// (R0 N bs^2 flops through the batched Cholesky's strip kernel, 403 ms per 1,000-SNP block at 500,000 samples; profiles/r5_loocv_*).
//
// Kernels:
//   k_tri_init    A (full symmetric working copy of the assembled lower triangle), Qt = I, d = diag
//   k_tri_step    step k of the reduction for every block of the batch, one launch per step (the steps are a dependency chain; within a step
//                 the trailing matrix is updated by row-cyclic workgroups and Qt by column slabs, no inter-workgroup synchronisation):
//                 every workgroup re-derives w_{k-1} and the reflector v_k from O(n) data (previous p, v, and the pivot row handed over through
//                 a side buffer), then applies A -= v w^T + w v^T to its rows WHILE forming p_k = tau_k A v_k from the updated values
//   k_tri_tables  delta, l of the R0 factorizations, u = L^-1 Q^T b_p, laid out per k for the recurrence kernel
//   k_tri_rec     one thread per sample: the R0 recurrences over k, leverage and numerators in registers, writes W
// Standardisation of the columns afterwards: k_loocv_std (loocv.hip).
#include <functional>
#include "rg_internal.h"

#define TRI_NA 16   // workgroups per matrix on the trailing-matrix update (rows dealt cyclically)
#define TRI_NT 256

struct TriArgs {
  int nblk, n64, P, R0, rtot, Ppad;   // Ppad: P rounded up to the recurrence kernel's phenotype group (the table rows are padded with zeros)
  const int32_t* bs;
  const double* sum;   // [nblk][rtot][n64] assembled systems: lower triangle of A in rows < n64, b_p^T in rows n64 + p
  double* A;           // [nblk][n64][n64]
  double* Qt;          // [nblk][n64][n64]
  double* dv;          // [nblk][n64]
  double* ev;          // [nblk][n64]
  double* v;           // [nblk][2][n64] reflector of step k at parity k & 1
  double* p;           // [nblk][2][n64] tau_k A v_k
  double* tau;         // [nblk][2]
  double* side;        // [nblk][2][n64] row k of A^(k-1), handed from step k-1 to step k
  const double* lambda;
  double* tab;         // [nblk][n64][R0][2 + Ppad]: (l_{k-1}, 1 / delta_k, u_pk / delta_k)
  int32_t* info;       // set to 1 when a delta is not positive (A + lambda I not positive definite)
};

__global__ __launch_bounds__(TRI_NT) void k_tri_init(TriArgs a) {
  const int blk = blockIdx.z, n64 = a.n64, n = a.bs[blk];
  const int64_t e = (int64_t)blockIdx.x * TRI_NT + threadIdx.x;
  if (e >= (int64_t)n64 * n64) return;
  const int i = (int)(e / n64), j = (int)(e % n64);
  const double* S = a.sum + (int64_t)blk * a.rtot * n64;
  const bool in = i < n && j < n;
  const int hi = in ? max(i, j) : 0, lo = in ? min(i, j) : 0;
  const double val = S[(int64_t)hi * n64 + lo];
  a.A[(int64_t)blk * n64 * n64 + e] = in ? val : 0.0;
  a.Qt[(int64_t)blk * n64 * n64 + e] = (i == j) ? 1.0 : 0.0;
  if (i == j) {
    a.dv[(int64_t)blk * n64 + i] = in ? val : 0.0;
    a.ev[(int64_t)blk * n64 + i] = 0.0;
  }
}

__device__ __forceinline__ double tri_block_sum(double x, double* red) {
  for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = x;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

// grid (TRI_NA + n64 / 64, nblk); dynamic LDS: 4 * n64 doubles
__global__ __launch_bounds__(TRI_NT) void k_tri_step(TriArgs a, int k) {
  extern __shared__ double sm[];
  __shared__ double red[4];
  __shared__ double ysum[4][64];
  const int blk = blockIdx.y, n64 = a.n64, n = a.bs[blk];
  if (k > n - 2) return;
  double* vprev = sm;
  double* wprev = sm + n64;
  double* vcur = sm + 2 * n64;
  double* row = sm + 3 * n64;
  const int tid = threadIdx.x;
  double* A = a.A + (int64_t)blk * n64 * n64;
  const int par = k & 1, ppar = par ^ 1;
  // ---- w_{k-1} = p - (tau / 2)(p . v) v, from the previous step's vectors; row k of A^(k): the side row brought up to date ----
  // (the loads of v, p and the side row are independent: issued together, up to four columns per thread)
  double vk_prev = 0.0, wk_prev = 0.0;
  double xn2 = 0.0;
  if (k > 0) {
    const double* gv = a.v + ((int64_t)blk * 2 + ppar) * n64;
    const double* gp = a.p + ((int64_t)blk * 2 + ppar) * n64;
    const double* gs = a.side + ((int64_t)blk * 2 + par) * n64;
    const double taup = a.tau[blk * 2 + ppar];
    double dot = 0.0;
    for (int jb = k + tid; jb < n; jb += 4 * TRI_NT) {
      double vj[4], pj[4], sj[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int j = min(jb + t * TRI_NT, n - 1);
        vj[t] = gv[j]; pj[t] = gp[j]; sj[t] = gs[j];
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int j = jb + t * TRI_NT;
        if (j < n) {
          vprev[j] = vj[t];
          wprev[j] = pj[t];
          row[j] = sj[t];
          dot = fma(pj[t], vj[t], dot);
        }
      }
    }
    dot = tri_block_sum(dot, red);
    const double f = 0.5 * taup * dot;
    for (int j = k + tid; j < n; j += TRI_NT) wprev[j] = fma(-f, vprev[j], wprev[j]);
    __syncthreads();
    vk_prev = vprev[k];
    wk_prev = wprev[k];
    for (int j = k + tid; j < n; j += TRI_NT) {
      const double r = row[j] - (vk_prev * wprev[j] + wk_prev * vprev[j]);
      row[j] = r;
      if (j >= k + 2) xn2 = fma(r, r, xn2);
    }
  } else {
    for (int j = tid; j < n; j += TRI_NT) {
      const double r = A[j];
      row[j] = r;
      if (j >= 2) xn2 = fma(r, r, xn2);
    }
  }
  {
    xn2 = tri_block_sum(xn2, red);
    const double alpha = row[k + 1];
    double beta = alpha, tauk = 0.0, scale = 0.0;
    if (n - k - 1 > 1 && xn2 > 0.0) {
      beta = -copysign(sqrt(fma(alpha, alpha, xn2)), alpha);
      tauk = (beta - alpha) / beta;
      scale = 1.0 / (alpha - beta);
    }
    for (int j = k + 1 + tid; j < n; j += TRI_NT) vcur[j] = (j == k + 1) ? 1.0 : row[j] * scale;
    __syncthreads();
    if (blockIdx.x == 0) {
      double* gv = a.v + ((int64_t)blk * 2 + par) * n64;
      for (int j = k + 1 + tid; j < n; j += TRI_NT) gv[j] = vcur[j];
      if (tid == 0) {
        a.tau[blk * 2 + par] = tauk;
        a.dv[(int64_t)blk * n64 + k] = row[k];
        a.ev[(int64_t)blk * n64 + k] = beta;
      }
    }
    // ---- the step itself ----
    const int lane = tid & 63, wave = tid >> 6;
    if ((int)blockIdx.x < TRI_NA) {
      // rows k+1 ..: A[i][j] -= v_{k-1,i} w_{k-1,j} + w_{k-1,i} v_{k-1,j} (j >= k+1), p_k[i] = tau_k sum_j A[i][j] v_k[j]
      double* gp = a.p + ((int64_t)blk * 2 + par) * n64;
      double* gside = a.side + ((int64_t)blk * 2 + ppar) * n64;
      const int j0 = (k + 1) & ~63;
      // eight loads of a row are issued before the first use (the store of an updated value would otherwise fence the next load: one
      // L2 round trip per 64 columns and row)
      for (int i = k + 1 + (int)blockIdx.x * 4 + wave; i < n; i += 4 * TRI_NA) {
        double* Ai = A + (int64_t)i * n64;
        const double vi = (k > 0) ? vprev[i] : 0.0, wi = (k > 0) ? wprev[i] : 0.0;
        double acc = 0.0;
        for (int jb = j0; jb < n; jb += 512) {
          double x[8];
#pragma unroll
          for (int t = 0; t < 8; ++t) x[t] = Ai[min(jb + 64 * t + lane, n64 - 1)];
#pragma unroll
          for (int t = 0; t < 8; ++t) {
            const int j = jb + 64 * t + lane;
            if (j >= k + 1 && j < n) {
              double xv = x[t];
              if (k > 0) {
                xv -= vi * wprev[j] + wi * vprev[j];
                Ai[j] = xv;
              }
              acc = fma(xv, vcur[j], acc);
              if (j == i) a.dv[(int64_t)blk * n64 + i] = xv;
              if (i == k + 1) gside[j] = xv;
            }
          }
        }
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
        if (lane == 0) gp[i] = tauk * acc;
      }
    } else if (tauk != 0.0) {
      // Qt <- H_k Qt on a slab of 64 columns: y_c = sum_i v_i Qt[i][c], Qt[i][c] -= tau v_i y_c  (rows i >= k+1)
      const int c = ((int)blockIdx.x - TRI_NA) * 64 + lane;
      if (((int)blockIdx.x - TRI_NA) * 64 >= n) return;
      double* Q = a.Qt + (int64_t)blk * n64 * n64 + c;
      double y = 0.0;
      for (int ib = k + 1 + wave; ib < n; ib += 32) {
        double x[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) x[t] = Q[(int64_t)min(ib + 4 * t, n - 1) * n64];
#pragma unroll
        for (int t = 0; t < 8; ++t)
          if (ib + 4 * t < n) y = fma(vcur[ib + 4 * t], x[t], y);
      }
      ysum[wave][lane] = y;
      __syncthreads();
      y = tauk * ((ysum[0][lane] + ysum[1][lane]) + (ysum[2][lane] + ysum[3][lane]));
      for (int ib = k + 1 + wave; ib < n; ib += 32) {
        double x[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) x[t] = Q[(int64_t)min(ib + 4 * t, n - 1) * n64];
#pragma unroll
        for (int t = 0; t < 8; ++t)
          if (ib + 4 * t < n) Q[(int64_t)(ib + 4 * t) * n64] = fma(-vcur[ib + 4 * t], y, x[t]);
      }
    }
  }
}

// t_p = Qt b_p: grid (n64 / 64, nblk, P), one wave per 16 rows
__global__ __launch_bounds__(TRI_NT) void k_tri_qtb(TriArgs a, double* tq /*[nblk][P][n64]*/) {
  const int blk = blockIdx.y, p = blockIdx.z, n64 = a.n64, n = a.bs[blk];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const double* Q = a.Qt + (int64_t)blk * n64 * n64;
  const double* b = a.sum + (int64_t)blk * a.rtot * n64 + (int64_t)(n64 + p) * n64;
  for (int rr = 0; rr < 16; ++rr) {
    const int k = blockIdx.x * 64 + wave * 16 + rr;
    if (k >= n) break;
    double acc = 0.0;
    for (int j = lane; j < n; j += 64) acc = fma(Q[(int64_t)k * n64 + j], b[j], acc);
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
    if (lane == 0) tq[((int64_t)blk * a.P + p) * n64 + k] = acc;
  }
}

// The R0 factorizations T + lambda_r I = L D L^T (delta_0 = d_0 + lambda, l_k = e_k / delta_k, delta_{k+1} = d_{k+1} + lambda - l_k e_k)
// and u = L^-1 t_p, one thread per (r, p) -- each repeats the scalar delta recurrence of its r rather than wait for another thread's --
// written per k for the recurrence kernel: (l_{k-1}, 1 / delta_k, u_pk / delta_k).  grid (nblk), blockDim = 64 * ceil(R0 * max(P, 1) / 64);
// dynamic LDS: 2 * n64 doubles (d, e)
__global__ void k_tri_tables(TriArgs a, const double* tq) {
  extern __shared__ double sm[];
  const int blk = blockIdx.x, n64 = a.n64, n = a.bs[blk], P = a.P, R0 = a.R0;
  double* sd = sm;
  double* se = sm + n64;
  for (int j = threadIdx.x; j < n; j += blockDim.x) {
    sd[j] = a.dv[(int64_t)blk * n64 + j];
    se[j] = a.ev[(int64_t)blk * n64 + j];
  }
  __syncthreads();
  const int rp = threadIdx.x;
  if (rp >= R0 * P) return;
  const int r = rp / P, p = rp % P;
  const int st = 2 + a.Ppad;
  double* tab = a.tab + (int64_t)blk * n64 * R0 * st;
  const double* t = tq + ((int64_t)blk * P + p) * n64;
  const double lam = a.lambda[r];
  double delta = sd[0] + lam, lprev = 0.0, u = 0.0;
  bool bad = false;
  for (int k0 = 0; k0 < n; k0 += 8) {
    double tv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) tv[i] = t[min(k0 + i, n64 - 1)];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int k = k0 + i;
      if (k < n) {
        if (!(delta > 0.0)) bad = true;
        const double dinv = 1.0 / delta;
        u = fma(-lprev, u, tv[i]);
        double* e = tab + ((int64_t)k * R0 + r) * st;
        if (p == 0) { e[0] = lprev; e[1] = dinv; }
        e[2 + p] = u * dinv;
        if (k + 1 < n) {
          lprev = se[k] * dinv;
          delta = sd[k + 1] + lam - lprev * se[k];
        }
      }
    }
  }
  if (bad) *a.info = 1;
}

// The recurrences: y_r(k) = z(k) - l_r(k-1) y_r(k-1); h_r += y^2 / delta; num_rp += y u_rp / delta.
// One WAVE per ridge value r and group of <= TRI_PG phenotypes, four samples per lane (256 samples per wave).  The table entries of a k are
// the same for every lane: they are read at wave-uniform addresses (scalar loads into SGPRs, prefetched four k ahead by the unrolled body) and
// enter the v_fma_f64 as scalar operands -- staged through LDS, their broadcast reads would cost as many issue slots of the CU's one LDS pipe
// as the multiply-adds cost on its four SIMDs.  The waves of a workgroup (one per r) walk the same z columns and share them through L1.
// grid (ceil(chunk / 256), nblk, ceil(P / TRI_PG)), block 64 * R0; zt: [nblk][n64][chunk]
struct TriRecArgs {
  int nblk, n64, P, R0, C, Ppad;
  int64_t Np, pos0, chunk;
  const int32_t* bs; const int32_t* blockid;
  const double* zt; const double* tab; const double* V; double* W;
};
#define TRI_PG 10
#define TRI_S 2
template <int PG>
__global__ __launch_bounds__(512) void k_tri_rec(TriRecArgs a) {
  const int blk = blockIdx.y, n = a.bs[blk], n64 = a.n64, P = a.P, R = a.R0;
  const int r = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int p0 = blockIdx.z * PG, np = min(PG, P - p0);
  const int stq = 2 + a.Ppad;
  const int64_t c0 = (int64_t)blockIdx.x * (64 * TRI_S) + lane;
  const double* __restrict__ z = a.zt + (int64_t)blk * n64 * a.chunk;
  // this wave's rows of the table: (l, 1 / delta) and the PG consecutive u / delta of its phenotype group (padding entries are zero)
  const double* __restrict__ tab = a.tab + ((int64_t)blk * n64 * R + r) * stq;
  const double* __restrict__ tabc = tab + 2 + p0;
  int64_t cp[TRI_S];
#pragma unroll
  for (int s = 0; s < TRI_S; ++s) cp[s] = min(c0 + 64 * s, a.chunk - 1);
  double y[TRI_S], h[TRI_S], num[TRI_S][PG];
#pragma unroll
  for (int s = 0; s < TRI_S; ++s) {
    y[s] = 0.0; h[s] = 0.0;
#pragma unroll
    for (int q = 0; q < PG; ++q) num[s][q] = 0.0;
  }
  const int n4 = n & ~3;
  // the z values of the NEXT four k are requested before the multiply-adds of the current four (first touch of a z column is an HBM round trip)
  double zn[4][TRI_S];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int s = 0; s < TRI_S; ++s) zn[i][s] = z[(int64_t)min(i, n - 1) * a.chunk + cp[s]];
  for (int k0 = 0; k0 < n4; k0 += 4) {
    double zk[4][TRI_S];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int s = 0; s < TRI_S; ++s) {
        zk[i][s] = zn[i][s];
        zn[i][s] = z[(int64_t)min(k0 + 4 + i, n - 1) * a.chunk + cp[s]];
      }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const double* __restrict__ e = tab + (int64_t)(k0 + i) * R * stq;
      const double* __restrict__ ec = tabc + (int64_t)(k0 + i) * R * stq;
      const double lm = e[0], di = e[1];
      double cq[PG];
#pragma unroll
      for (int q = 0; q < PG; ++q) cq[q] = ec[q];
#pragma unroll
      for (int s = 0; s < TRI_S; ++s) {
        const double ys = fma(-lm, y[s], zk[i][s]);
        y[s] = ys;
        h[s] = fma(ys * ys, di, h[s]);
#pragma unroll
        for (int q = 0; q < PG; ++q) num[s][q] = fma(ys, cq[q], num[s][q]);
      }
    }
  }
  for (int k = n4; k < n; ++k) {
    const double* __restrict__ e = tab + (int64_t)k * R * stq;
    const double lm = e[0], di = e[1];
#pragma unroll
    for (int s = 0; s < TRI_S; ++s) {
      const double ys = fma(-lm, y[s], z[(int64_t)k * a.chunk + cp[s]]);
      y[s] = ys;
      h[s] = fma(ys * ys, di, h[s]);
#pragma unroll
      for (int q = 0; q < PG; ++q) num[s][q] = fma(ys, e[2 + p0 + q], num[s][q]);
    }
  }
  const int col = a.blockid[blk] * R + r;
#pragma unroll
  for (int s = 0; s < TRI_S; ++s) {
    if (c0 + 64 * s >= a.chunk) continue;
    const int64_t pos = a.pos0 + c0 + 64 * s;
#pragma unroll
    for (int q = 0; q < PG; ++q)
      if (q < np) {
        const double yv = a.V[(int64_t)(a.C + p0 + q) * a.Np + pos];
        a.W[((int64_t)col * P + p0 + q) * a.Np + pos] = (num[s][q] - h[s] * yv) / (1.0 - h[s]);
      }
  }
}

static void launch_rec(hipStream_t st, const TriRecArgs& ra) {
  const unsigned gx = (unsigned)((ra.chunk + 64 * TRI_S - 1) / (64 * TRI_S));
  if (ra.P <= 2) hipLaunchKernelGGL((k_tri_rec<2>), dim3(gx, ra.nblk, (ra.P + 1) / 2), dim3(64 * ra.R0), 0, st, ra);
  else if (ra.P <= 5) hipLaunchKernelGGL((k_tri_rec<5>), dim3(gx, ra.nblk, 1), dim3(64 * ra.R0), 0, st, ra);
  else hipLaunchKernelGGL((k_tri_rec<TRI_PG>), dim3(gx, ra.nblk, (ra.P + TRI_PG - 1) / TRI_PG), dim3(64 * ra.R0), 0, st, ra);
}

static int tri_ppad(int P) { return P <= 2 ? 2 : (P <= 5 ? 5 : (P + TRI_PG - 1) / TRI_PG * TRI_PG); }
size_t rg_loocv_tri_ws_doubles(int nblk, int n64, int P, int R0) {
  return (size_t)nblk * ((size_t)2 * n64 * n64 + (size_t)8 * n64 + 2 + (size_t)n64 * R0 * (2 + tri_ppad(P)) + (size_t)n64 * std::max(P, 1));
}

// The leave-one-out predictors of `nblk` assembled blocks.  gt_chunk(pos0, len) must fill la.gt as [nblk][len][n64] (standardised genotypes,
// sample-major) for the sample positions [pos0, pos0 + len); zt: [nblk][n64][chunk] scratch; ws: rg_loocv_tri_ws_doubles.
int rg_l0_loocv_tri(rg_ctx* ctx, hipStream_t st, const LoocvArgs& la, int max_bs, const double* d_sum, int rtot, double* ws, double* zt,
                    int64_t chunk, const std::function<void(int64_t, int64_t)>& gt_chunk) {
  const int nblk = la.nblk, n64 = la.n64, P = la.P, R0 = la.R0;
  if (R0 > 8) { ctx->err = "leave-one-out level 0: more than 8 ridge values"; return RG_ERR_ARG; }
  TriArgs a;
  a.nblk = nblk; a.n64 = n64; a.P = P; a.R0 = R0; a.rtot = rtot; a.bs = la.bs; a.sum = d_sum;
  double* w = ws;
  a.A = w; w += (size_t)nblk * n64 * n64;
  a.Qt = w; w += (size_t)nblk * n64 * n64;
  a.dv = w; w += (size_t)nblk * n64;
  a.ev = w; w += (size_t)nblk * n64;
  a.v = w; w += (size_t)nblk * 2 * n64;
  a.p = w; w += (size_t)nblk * 2 * n64;
  a.side = w; w += (size_t)nblk * 2 * n64;
  a.tau = w; w += (size_t)nblk * 2;
  a.Ppad = tri_ppad(P);
  a.tab = w; w += (size_t)nblk * n64 * R0 * (2 + a.Ppad);
  hipMemsetAsync(a.tab, 0, sizeof(double) * (size_t)nblk * n64 * R0 * (2 + a.Ppad), st);
  double* tq = w;
  a.lambda = ctx->d_lambda; a.info = ctx->d_info + 1;
  hipLaunchKernelGGL(k_tri_init, dim3((unsigned)(((int64_t)n64 * n64 + TRI_NT - 1) / TRI_NT), 1, nblk), dim3(TRI_NT), 0, st, a);
  const size_t lds = sizeof(double) * 4 * n64;
  for (int k = 0; k <= max_bs - 2; ++k)
    hipLaunchKernelGGL(k_tri_step, dim3(TRI_NA + n64 / 64, nblk), dim3(TRI_NT), lds, st, a, k);
  if (R0 * P > 1024) { ctx->err = "leave-one-out level 0: more than 1,024 (ridge value, phenotype) pairs"; return RG_ERR_ARG; }
  hipLaunchKernelGGL(k_tri_qtb, dim3(n64 / 64, nblk, P), dim3(TRI_NT), 0, st, a, tq);
  hipLaunchKernelGGL(k_tri_tables, dim3(nblk), dim3(64 * ((R0 * P + 63) / 64)), sizeof(double) * 2 * n64, st, a, (const double*)tq);
  if (hipGetLastError() != hipSuccess) { ctx->err = "leave-one-out level 0: kernel launch failed"; return RG_ERR_HIP; }
  const int mk = (int)rg_round_up(max_bs, 64);
  for (int64_t pos0 = 0; pos0 < la.Np; pos0 += chunk) {
    const int64_t len = std::min(chunk, la.Np - pos0);
    gt_chunk(pos0, len);
    for (int b = 0; b < nblk; ++b)
      rg_launch_dgemm_nt(st, a.Qt + (size_t)b * n64 * n64, n64, la.gt + (size_t)b * len * n64, n64, mk, (int)len, mk, zt + (size_t)b * n64 * len, len);
    TriRecArgs ra;
    ra.nblk = nblk; ra.n64 = n64; ra.P = P; ra.R0 = R0; ra.C = la.C; ra.Ppad = a.Ppad; ra.Np = la.Np; ra.pos0 = pos0; ra.chunk = len;
    ra.bs = la.bs; ra.blockid = la.blockid; ra.zt = zt; ra.tab = a.tab; ra.V = la.V; ra.W = la.W;
    launch_rec(st, ra);
  }
  if (hipGetLastError() != hipSuccess) { ctx->err = "leave-one-out level 0: kernel launch failed"; return RG_ERR_HIP; }
  return RG_OK;
}
