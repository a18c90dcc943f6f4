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
#include <type_traits>
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
__global__ __launch_bounds__(256) void k_assemble_generic(AsmArgs a) {
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
  int64_t e_rhs = e;
  if (i >= n64) {  // RHS rows: b_f^T
    const int p = i - n64;
    if (a.embed) {   // embedded: row bs + p of the matrix part (its padding rows), nothing past n64
      if (p >= P) return;
      e_rhs = (int64_t)(bs + p) * n64 + k;
    }
    for (int s = 0; s < ns; ++s) {
      double v = 0.0;
      if (p < P && k < bs) v = a.GYt[(((int64_t)blk * ns + s) * n128 + k) * P + p] / sc[k];
      if (inreg) {
#pragma unroll
        for (int t = 0; t < AF; ++t) if (t == s) vf[t] = v;
      } else fold[(int64_t)s * msz + e_rhs] = v;
      tot += v;
    }
  } else if (a.embed && i >= bs && i < bs + P) {  // an embedded right-hand-side row: written by the thread of row n64 + p
    skip = true;
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
      if (t < ns) fold[(int64_t)t * msz + e_rhs] = a.diff_mode ? tot - vf[t] : vf[t];
  } else if (a.diff_mode) {
    for (int s = 0; s < ns; ++s) fold[(int64_t)s * msz + e_rhs] = tot - fold[(int64_t)s * msz + e_rhs];
  }
  if (!a.diff_mode) sum[e_rhs] = tot;
}

// ---- tiled assemble (the default: up to AF folds) --------------------------------------------------------------
// grid (n64/32, ceil(rtot/32), nblk), block (32, 8): a 32 x 32 tile per workgroup, thread (tx, ty) owns column
// k = k0 + tx of rows i0 + ty + {0, 8, 16, 24}.  The small per-row / per-column covariate terms
//   E_s[i] = B Q_s [i] - F_s[i],   F_s[k],   B[i],   B[k]            (C numbers each)
// are staged once per tile in LDS (chunks of ACC covariates); correction_s(i,k) = sum_c E_s[i][c] B[k][c] - B[i][c] F_s[k][c].
// Every global load is UNCONDITIONAL with a clamped address and the value is selected afterwards: hipcc turns a load
// under a data-dependent `if` into branch + load + s_waitcnt vmcnt(0), i.e. one full memory round trip per load,
// whereas unconditional loads of an unrolled loop are all in flight together.
#define ACC 8
// NF = number of folds carried in registers / LDS (5: the default K; 8: the maximum of this kernel) -- sizing the per-fold
// arrays for 8 when K = 5 cost a third of the registers and left ONE workgroup resident per CU.
template <int NF>
__global__ __launch_bounds__(256, (NF <= 5 ? 2 : 1)) void k_assemble_tiled(AsmArgs a) {
  __shared__ double sE[NF][32][ACC + 1];
  __shared__ double sF[NF][32][ACC + 1];
  __shared__ double sBi[32][ACC + 1];
  __shared__ double sBk[32][ACC + 1];
  const int blk = blockIdx.z;
  const int tx = threadIdx.x, ty = threadIdx.y, tid = ty * 32 + tx;
  const int k0 = blockIdx.x * 32, i0 = blockIdx.y * 32;
  const int bs = a.bs[blk];
  const int C = a.C, P = a.P, ns = a.nseg, n128 = a.n128, n64 = a.n64;
  const int64_t msz = (int64_t)a.rtot * n64;
  double* __restrict__ fold = a.fold + (int64_t)blk * ns * msz;
  double* __restrict__ sum = a.sum + (int64_t)blk * msz;
  const double* __restrict__ sc = a.sc + (int64_t)blk * n128;
  const int k = k0 + tx;                       // < n64 (grid.x = n64 / 32)
  const int kc = min(k, n128 - 1);
  if (i0 >= n64) {  // RHS rows: b_f^T
    const double isck = 1.0 / sc[kc];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int i = i0 + ty + 8 * rr;
      const int p = i - n64;
      const bool live = (i < a.rtot) && (p < P) && (k < bs);
      const int pc = live ? p : 0;
      double vf[NF], tot = 0.0;
#pragma unroll
      for (int s = 0; s < NF; ++s) {
        const int ss = min(s, ns - 1);
        const double g = a.GYt[(((int64_t)blk * ns + ss) * n128 + kc) * P + pc];
        vf[s] = (live && s < ns) ? g * isck : 0.0;
        tot += vf[s];
      }
      if (a.embed ? p < P : i < a.rtot) {     // embedded: row bs + p of the matrix part (its padding rows), nothing past n64
        const int64_t e = (int64_t)(a.embed ? bs + p : i) * n64 + k;
#pragma unroll
        for (int s = 0; s < NF; ++s)
          if (s < ns) fold[(int64_t)s * msz + e] = a.diff_mode ? tot - vf[s] : vf[s];
        if (!a.diff_mode) sum[e] = tot;
      }
    }
    return;
  }
  if (k0 > i0 + 31) {  // tile strictly above the diagonal: only the part inside a diagonal 64-tile is kept finite
    if ((k0 >> 6) == (i0 >> 6)) {
      for (int rr = 0; rr < 4; ++rr) {
        const int64_t e = (int64_t)(i0 + ty + 8 * rr) * n64 + k;
        for (int s = 0; s < ns; ++s) fold[(int64_t)s * msz + e] = 0.0;
        if (!a.diff_mode) sum[e] = 0.0;
      }
    }
    return;
  }
  const bool has_miss = a.nmiss[blk] > 0;      // uniform per block
  const int64_t ldS = 2 * (int64_t)n128;
  // integer Gram values of this thread's 4 elements, all folds: independent unconditional loads
  double at[4][NF];
  const double muk = a.mu[(int64_t)blk * n128 + kc];
  // the missing-call terms are a per-block (uniform) property: the branch is taken ONCE, around the whole unrolled
  // loop, so that each variant is straight-line code with all its loads in flight together
  auto gram_vals = [&](auto hm) {
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int i = i0 + ty + 8 * rr;            // < n64 <= n128
      const bool live = (k <= i) && (i < bs);
      const int il = live ? i : 0, kl = live ? k : 0;
      const double mui = a.mu[(int64_t)blk * n128 + i];
#pragma unroll
      for (int s = 0; s < NF; ++s) {
        const int ss = min(s, ns - 1);
        const int32_t* __restrict__ S = a.S + ((int64_t)blk * ns + ss) * ldS * ldS;
        double v = (double)S[(int64_t)il * ldS + kl];
        if (decltype(hm)::value) {
          v += mui * (double)S[(int64_t)(n128 + il) * ldS + kl];
          v += muk * (double)S[(int64_t)(n128 + kl) * ldS + il];
          v += mui * muk * (double)S[(int64_t)(n128 + il) * ldS + n128 + kl];
        }
        // multiply by a 0/1 mask instead of selecting: a select lets the optimizer sink the load back under the
        // condition (and serialise it); v is an exact integer-valued double, so v * 0.0 is exactly 0
        at[rr][s] = v * ((live && s < ns) ? 1.0 : 0.0);
      }
    }
  };
  if (has_miss) gram_vals(std::true_type{});
  else gram_vals(std::false_type{});
  double corr[4][NF];
#pragma unroll
  for (int rr = 0; rr < 4; ++rr)
#pragma unroll
    for (int s = 0; s < NF; ++s) corr[rr][s] = 0.0;
  for (int c0 = 0; c0 < C; c0 += ACC) {
    const int cn = min(ACC, C - c0);
    __syncthreads();
    // stage (32 rows x ACC) B[i], B[k] and (ns x 32 x ACC) E_s[i], F_s[k]; rows of this tile are all < n128
    for (int t = tid; t < 32 * ACC; t += 256) {
      const int r = t / ACC, c = t % ACC;
      const int cc = c0 + min(c, cn - 1);
      const double bi = a.Bm[((int64_t)blk * n128 + i0 + r) * C + cc];
      const double bk = a.Bm[((int64_t)blk * n128 + k0 + r) * C + cc];
      sBi[r][c] = (c < cn) ? bi : 0.0;
      sBk[r][c] = (c < cn) ? bk : 0.0;
    }
    for (int t = tid; t < ns * 32 * ACC; t += 256) {
      const int s = t / (32 * ACC), r = (t / ACC) % 32, c = t % ACC;
      const int cc = c0 + min(c, cn - 1);
      const int64_t oi = (((int64_t)blk * ns + s) * n128 + i0 + r) * C + cc;
      const int64_t ok = (((int64_t)blk * ns + s) * n128 + k0 + r) * C + cc;
      const double e = a.BQ[oi] - a.F[oi], f = a.F[ok];
      sE[s][r][c] = (c < cn) ? e : 0.0;
      sF[s][r][c] = (c < cn) ? f : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < ACC; ++c) {
      const double bk = sBk[tx][c];
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const double bi = sBi[ty + 8 * rr][c];
#pragma unroll
        for (int s = 0; s < NF; ++s)
          if (s < ns) corr[rr][s] = fma(sE[s][ty + 8 * rr][c], bk, fma(-bi, sF[s][tx][c], corr[rr][s]));
      }
    }
  }
  const double sck = sc[kc];
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const int i = i0 + ty + 8 * rr;
    const int64_t e = (int64_t)i * n64 + k;
    const double inv = 1.0 / (sc[i] * sck);
    if (a.embed && i >= bs && i < bs + P) continue;     // an embedded right-hand-side row: written by the workgroup of row n64 + p
    if (k > i) {  // upper triangle inside a diagonal tile
      if ((k >> 6) == (i >> 6)) {
        for (int s = 0; s < ns; ++s) fold[(int64_t)s * msz + e] = 0.0;
        if (!a.diff_mode) sum[e] = 0.0;
      }
      continue;
    }
    // rows >= bs (padding): at and the covariate terms are zero -> zeros are written
    double vf[NF], tot = 0.0;
#pragma unroll
    for (int s = 0; s < NF; ++s) {
      vf[s] = (s < ns && i < bs) ? (at[rr][s] + corr[rr][s]) * inv : 0.0;
      tot += vf[s];
    }
#pragma unroll
    for (int s = 0; s < NF; ++s)
      if (s < ns) fold[(int64_t)s * msz + e] = a.diff_mode ? tot - vf[s] : vf[s];
    if (!a.diff_mode) sum[e] = tot;
  }
}

void rg_launch_rowstats(hipStream_t st, const AsmArgs& a) {
  hipLaunchKernelGGL(k_rowstats, dim3((a.n128 + 63) / 64, a.nblk), dim3(64), 0, st, a);
}
void rg_launch_assemble(hipStream_t st, const AsmArgs& a) {
  if (a.nseg <= AF)
    if (a.nseg <= 5)
      hipLaunchKernelGGL(k_assemble_tiled<5>, dim3((a.n64 + 31) / 32, (a.rtot + 31) / 32, a.nblk), dim3(32, 8), 0, st, a);
    else
      hipLaunchKernelGGL(k_assemble_tiled<AF>, dim3((a.n64 + 31) / 32, (a.rtot + 31) / 32, a.nblk), dim3(32, 8), 0, st, a);
  else
    hipLaunchKernelGGL(k_assemble_generic, dim3((a.n64 + 31) / 32, (a.rtot + 7) / 8, a.nblk), dim3(32, 8), 0, st, a);
}
