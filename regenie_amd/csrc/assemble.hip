// fp64 assembly of the per-fold ridge systems from the exact integer Gram.
//
// With G~ the mean-imputed, analysis-masked dosages of a block (Geno.cpp:1749-1761), X the
// orthonormal covariate basis and s_j the residual scale, the reference forms (Data.cpp:196-211,
// :741-751)       G = D_s^-1 (G~ - (G~ X) X^T),   A_f = G_f G_f^T,   b_f = G_f Y_f   per fold f.
// Here, with S_f = G0_f G0_f^T, T_f = M_f G0_f^T, U_f = M_f M_f^T the int32 outputs of gram_i8.hip
// (G0 raw dosages with missing -> 0, M the missing indicator), F_f = G~_f X_f, B = sum_f F_f,
// Q_f = X_f^T X_f:
//   A~_f      = S_f + D_mu T_f + T_f^T D_mu + D_mu U_f D_mu
//   R_f       = A~_f - F_f B^T - B F_f^T + B Q_f B^T              ( = (G~_f - B X_f^T)(...)^T )
//   s_j^2     = sum_f R_f[j][j] / (n_analyzed - C)                 (Data.cpp:203)
//   A_f       = D_s^-1 R_f D_s^-1 ,   b_f = D_s^-1 (G~_f Y_f - B X_f^T Y_f)
// which is the same arithmetic re-associated; it agrees with the reference order to ~1e-15.
#include "rg_internal.h"

// thread = one SNP row of one block
__global__ __launch_bounds__(64) void k_rowstats(AsmArgs a) {
  const int blk = blockIdx.y;
  const int j = blockIdx.x * 64 + threadIdx.x;
  if (j >= a.n128) return;
  const int bs = a.bs[blk];
  const int C = a.C, P = a.P, Cv = a.Cv, ns = a.nseg;
  double* F = a.F + (int64_t)blk * ns * a.n128 * C;
  double* GY = a.GYt + (int64_t)blk * ns * a.n128 * P;
  double* Bm = a.Bm + ((int64_t)blk * a.n128 + j) * C;
  double* BQ = a.BQ + (int64_t)blk * ns * a.n128 * C;
  for (int s = 0; s < ns; ++s) {
    for (int c = 0; c < C; ++c) F[((int64_t)s * a.n128 + j) * C + c] = 0.0;
    for (int p = 0; p < P; ++p) GY[((int64_t)s * a.n128 + j) * P + p] = 0.0;
  }
  for (int c = 0; c < C; ++c) Bm[c] = 0.0;
  if (j >= bs) {
    a.sc[(int64_t)blk * a.n128 + j] = 1.0;
    for (int s = 0; s < ns; ++s)
      for (int c = 0; c < C; ++c) BQ[((int64_t)s * a.n128 + j) * C + c] = 0.0;
    return;
  }
  const double mu = a.mu[(int64_t)blk * a.n128 + j];
  const bool has_miss = a.nmiss[blk] > 0;
  // fixed-order reduction of the position-chunk partials (deterministic)
  for (int ch = 0; ch < a.nchunk; ++ch) {
    const int s = a.chunk_seg[ch];
    const double* pp = a.part + ((((int64_t)blk * a.nchunk + ch) * a.n128 + j) * 2) * Cv;
    for (int c = 0; c < C; ++c) {
      double v = pp[c];
      if (has_miss) v = fma(mu, pp[Cv + c], v);
      F[((int64_t)s * a.n128 + j) * C + c] += v;
    }
    for (int p = 0; p < P; ++p) {
      double v = pp[C + p];
      if (has_miss) v = fma(mu, pp[Cv + C + p], v);
      GY[((int64_t)s * a.n128 + j) * P + p] += v;
    }
  }
  for (int s = 0; s < ns; ++s)
    for (int c = 0; c < C; ++c) Bm[c] += F[((int64_t)s * a.n128 + j) * C + c];
  const int64_t ldS = 2 * (int64_t)a.n128;
  double d = 0.0;
  for (int s = 0; s < ns; ++s) {
    const double* Q = a.Q + (int64_t)s * C * C;
    const double* XtY = a.XtY + (int64_t)s * C * P;
    const double* Fs = F + ((int64_t)s * a.n128 + j) * C;
    double* BQs = BQ + ((int64_t)s * a.n128 + j) * C;
    double fb = 0.0, bqb = 0.0;
    for (int c2 = 0; c2 < C; ++c2) {
      double t = 0.0;
      for (int c = 0; c < C; ++c) t = fma(Bm[c], Q[c * C + c2], t);
      BQs[c2] = t;
      bqb = fma(t, Bm[c2], bqb);
      fb = fma(Fs[c2], Bm[c2], fb);
    }
    for (int p = 0; p < P; ++p) {
      double t = 0.0;
      for (int c = 0; c < C; ++c) t = fma(Bm[c], XtY[c * P + p], t);
      GY[((int64_t)s * a.n128 + j) * P + p] -= t;
    }
    const int32_t* S = a.S + ((int64_t)blk * ns + s) * ldS * ldS;
    double at = (double)S[(int64_t)j * ldS + j];
    if (has_miss) {
      at += 2.0 * mu * (double)S[(int64_t)(a.n128 + j) * ldS + j];
      at += mu * mu * (double)S[(int64_t)(a.n128 + j) * ldS + a.n128 + j];
    }
    d += at - 2.0 * fb + bqb;
  }
  double sc = sqrt(d / (double)(a.n_analyzed - C));
  if (!(sc >= 1e-6)) {  // params.numtol, Data.cpp:207 (also catches NaN)
    atomicMax(a.info, j + 1 + (blk << 20));
    sc = 1.0;
  }
  a.sc[(int64_t)blk * a.n128 + j] = sc;
}

// grid (n64/32, rtot/8, nblk), block (32, 8): element (row i = by*8+ty, col k = bx*32+tx).
// The per-fold values of one element live in registers (up to AF folds) so that in diff_mode the training-fold
// matrices (total - fold f) are written in a single pass; more folds fall back to a second pass through memory.
#define AF 8
__global__ __launch_bounds__(256) void k_assemble(AsmArgs a) {
  const int blk = blockIdx.z;
  const int k = blockIdx.x * 32 + threadIdx.x;
  const int i = blockIdx.y * 8 + threadIdx.y;
  const int bs = a.bs[blk];
  const int C = a.C, P = a.P, ns = a.nseg, n128 = a.n128, n64 = a.n64;
  if (i >= a.rtot || k >= n64) return;
  const int64_t msz = (int64_t)a.rtot * n64;
  double* __restrict__ fold = a.fold + (int64_t)blk * ns * msz;
  double* __restrict__ sum = a.sum + (int64_t)blk * msz;
  const int64_t e = (int64_t)i * n64 + k;
  const double* __restrict__ sc = a.sc + (int64_t)blk * n128;
  const bool inreg = ns <= AF;
  double vf[AF];
#pragma unroll
  for (int s = 0; s < AF; ++s) vf[s] = 0.0;
  double tot = 0.0;
  bool write_zero = false, skip = false;
  if (i >= n64) {  // RHS rows: b_f^T
    const int p = i - n64;
    for (int s = 0; s < ns; ++s) {
      double v = 0.0;
      if (p < P && k < bs) v = a.GYt[(((int64_t)blk * ns + s) * n128 + k) * P + p] / sc[k];
      if (inreg) {
#pragma unroll
        for (int t = 0; t < AF; ++t) if (t == s) vf[t] = v;
      } else fold[(int64_t)s * msz + e] = v;
      tot += v;
    }
  } else if (k > i) {  // upper triangle: never referenced, except inside diagonal 64-tiles (kept finite)
    if ((k >> 6) == (i >> 6)) write_zero = true;
    else skip = true;
  } else if (i >= bs) {  // padding (k <= i)
    write_zero = true;
  } else {
    const bool has_miss = a.nmiss[blk] > 0;
    const double mui = a.mu[(int64_t)blk * n128 + i], muk = a.mu[(int64_t)blk * n128 + k];
    const double inv = 1.0 / (sc[i] * sc[k]);
    const double* __restrict__ Bi = a.Bm + ((int64_t)blk * n128 + i) * C;
    const double* __restrict__ Bk = a.Bm + ((int64_t)blk * n128 + k) * C;
    const int64_t ldS = 2 * (int64_t)n128;
    for (int s = 0; s < ns; ++s) {
      const int32_t* __restrict__ S = a.S + ((int64_t)blk * ns + s) * ldS * ldS;
      double at = (double)S[(int64_t)i * ldS + k];
      if (has_miss) {
        at += mui * (double)S[(int64_t)(n128 + i) * ldS + k];
        at += muk * (double)S[(int64_t)(n128 + k) * ldS + i];
        at += mui * muk * (double)S[(int64_t)(n128 + i) * ldS + n128 + k];
      }
      const double* __restrict__ Fi = a.F + (((int64_t)blk * ns + s) * n128 + i) * C;
      const double* __restrict__ Fk = a.F + (((int64_t)blk * ns + s) * n128 + k) * C;
      const double* __restrict__ BQi = a.BQ + (((int64_t)blk * ns + s) * n128 + i) * C;
      double corr = 0.0;
      for (int c = 0; c < C; ++c) corr += BQi[c] * Bk[c] - Fi[c] * Bk[c] - Bi[c] * Fk[c];
      const double v = (at + corr) * inv;
      if (inreg) {
#pragma unroll
        for (int t = 0; t < AF; ++t) if (t == s) vf[t] = v;
      } else fold[(int64_t)s * msz + e] = v;
      tot += v;
    }
  }
  if (skip) return;
  if (write_zero) {
    for (int s = 0; s < ns; ++s) fold[(int64_t)s * msz + e] = 0.0;
    if (!a.diff_mode) sum[e] = 0.0;
    return;
  }
  // diff_mode: fold[f] := (sum over folds) - (fold f) = the training-fold system of CV fold f, so the
  // Cholesky's first touch reads ONE matrix; otherwise fold[f] and their sum are both written (LOOCV path).
  if (inreg) {
#pragma unroll
    for (int t = 0; t < AF; ++t)
      if (t < ns) fold[(int64_t)t * msz + e] = a.diff_mode ? tot - vf[t] : vf[t];
  } else if (a.diff_mode) {
    for (int s = 0; s < ns; ++s) fold[(int64_t)s * msz + e] = tot - fold[(int64_t)s * msz + e];
  }
  if (!a.diff_mode) sum[e] = tot;
}

void rg_launch_rowstats(hipStream_t st, const AsmArgs& a) {
  hipLaunchKernelGGL(k_rowstats, dim3((a.n128 + 63) / 64, a.nblk), dim3(64), 0, st, a);
}
void rg_launch_assemble(hipStream_t st, const AsmArgs& a) {
  hipLaunchKernelGGL(k_assemble, dim3((a.n64 + 31) / 32, (a.rtot + 7) / 8, a.nblk), dim3(32, 8), 0,
                     st, a);
}
