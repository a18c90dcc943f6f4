// Step-2 quantitative-trait score test for one block of variants (include/rg_step2.h; SURVEY.md 8(f) row 1, first slice).
//
// Reference path: compute_tests_mt (Data.cpp:2476-2555) -> residualize_geno (Geno.cpp:3242-3260) -> compute_score_qt
// (Step2_Models.cpp:343-468), dense genotypes, non-strict mode.  With r = g~ - X (X^T g~) (g~ = mean-imputed genotypes) the
// reference's num = res^T (r / sf) * sf and denum = sf^2 * mask^T (r / sf)^2 are res^T r and mask^T r^2, so the scaled
// genotypes are never materialised: two streaming passes over the block, both HBM-bound (8 B per genotype each).
//
//   pass 1  k_s2_proj   per variant: sum and count of the observed entries, A_c = sum g0 x_c and M_c = sum miss x_c
//                       (g0 = g with 0 at the missing entries)  ->  mu = sum / count, beta_c = A_c + mu M_c
//   pass 2  k_s2_score  r = g~ - sum_c beta_c x_c;  |r|^2, and per phenotype res_p . r and mask_p . r^2
//
// A workgroup owns VPB variants x 256 * EPT samples so that every X / res / mask element it loads is used for VPB variants;
// each thread keeps its EPT samples of the VPB variants in registers.  The partial sums a thread carries per covariate (or
// per phenotype) are reduced across the wave eight at a time with a butterfly that halves the value count at each of the
// first three exchange steps (10 exchanges instead of 48).  Partials per (variant, sample chunk) are summed in chunk order:
// deterministic.  The tile (VPB x EPT) is a tuning knob: RG_S2_TILE=4x4|4x8|8x4|8x8|16x4 (read at rg_s2_create).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "step2_internal.h"

// Hard calls (rg_s2_qt_block_packed): the same statistic from the 2-bit rows, exactly, on the i8 matrix cores.  The statistic needs
// nothing but genotype COUNTS and CONTRACTIONS of the genotype row with columns that are fixed per chromosome:
//   A_c = sum g0 x_c, M_c = sum miss x_c, R_p = sum g0 res_p, T_p = sum miss res_p        (g0 in {0,1,2}, miss in {0,1})
//   mu = (n1 + 2 n2) / (n - nmiss),  beta = A + mu M = X^T g~,  |r|^2 = n1 + 4 n2 + nmiss mu^2 - |beta|^2   (X orthonormal)
//   num_p = R_p + mu T_p - (res_p^T X) beta
// and, when every analysed sample is observed for every phenotype, denum_p = |r|^2.  When phenotypes differ in their missing values
// the per-trait denominators also need the contractions with x_c * mask_p (C * P columns) and of g0^2 and miss with mask_p:
//   XtGm_pc = sum g~ mask_p x_c,  g2m_p = sum mask_p g~^2 = sum mask_p g0^2 + mu^2 sum mask_p miss
//   dense variant :  denum_p = g2m_p - 2 XtGm_p . beta + beta^T Q_p beta,  Q_p = X^T diag(mask_p) X      (= mask_p^T r^2, exact)
//   sparse variant:  denum_p = g2m_p - 2 XtGm_p . beta + |beta|^2    (the reference's approximation, Step2_Models.cpp:402-413)
// where "sparse" is check_sparse_G's rule (Geno.cpp:3165-3177: at most n_samples * (1 - prop_zero_thr) non-zero entries after the
// mean imputation) -- the reference takes that branch without residualising or rescaling the variant (Data.cpp:2513-2515).
// Those contractions are what xy_i8.hip evaluates for Step 1: the columns are split once into eight balanced base-128 digit
// planes (k_v_split; only the res columns change with the chromosome), v_mfma_i32_32x32x32_i8 accumulates exact int32 digit sums
// per sample segment, the segments are added in int64 and the digits recombined in fp64 (k_s2_combine).  The genotype operand is
// read at 2 bits per call -- 0.25 B per genotype against 16 B over the two fp64 passes above -- and never leaves integer form.

namespace {

constexpr double NNZ_UNIT = 67108864.0;   // 2^26: k_s2_proj carries (non-zero count) * 2^26 + (observed count) in one exact fp64 sum (n < 2^26)
constexpr int DEFAULT_VPB = 4, DEFAULT_EPT = 4;   // variants per workgroup, samples per thread (best of the sweep in
                                                 // profiles/r1_step2_qt.md)

__device__ __forceinline__ bool is_missing(double g) { return !(g >= 0.0); }   // NaN, or regenie's -3

// v[0..7] are eight independent per-lane partial sums.  On return v[0] of every lane holds the wave total of value
// (lane >> 3); the summation tree is fixed.
__device__ __forceinline__ void wave_reduce8(double (&v)[8]) {
  const int lane = threadIdx.x & 63;
  {
    const bool hi = lane & 32;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const double keep = hi ? v[4 + i] : v[i], send = hi ? v[i] : v[4 + i];
      v[i] = keep + __shfl_xor(send, 32);
    }
  }
  {
    const bool hi = lane & 16;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const double keep = hi ? v[2 + i] : v[i], send = hi ? v[i] : v[2 + i];
      v[i] = keep + __shfl_xor(send, 16);
    }
  }
  {
    const bool hi = lane & 8;
    const double keep = hi ? v[1] : v[0], send = hi ? v[0] : v[1];
    v[0] = keep + __shfl_xor(send, 8);
  }
  v[0] += __shfl_xor(v[0], 4);
  v[0] += __shfl_xor(v[0], 2);
  v[0] += __shfl_xor(v[0], 1);
}

// Wave totals of a[v] and b[v] (v < VPB) into row[v * stride + off_a] and row[v * stride + off_b], eight values per butterfly.
template <int VPB>
__device__ __forceinline__ void reduce_pairs(const double (&a)[VPB], const double (&b)[VPB], double* row, int stride, int off_a,
                                             int off_b) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int grp = 0; grp < 2 * VPB / 8; ++grp) {
    double v8[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int t = grp * 8 + i;
      v8[i] = t < VPB ? a[t < VPB ? t : 0] : b[t < VPB ? 0 : t - VPB];
    }
    wave_reduce8(v8);
    if ((lane & 7) == 0) {
      const int t = grp * 8 + (lane >> 3);
      row[(t % VPB) * stride + (t < VPB ? off_a : off_b)] = v8[0];
    }
  }
}

// The per-covariate / per-phenotype loops are latency-bound when each iteration waits for its own loads (measured: 2.5 us per
// workgroup-iteration at 3 waves per SIMD), so the next row is fetched into registers before the current one is consumed.
template <int EPT>
__device__ __forceinline__ void load_row(const double* __restrict__ src, int64_t row, int64_t n, int64_t base, double (&dst)[EPT]) {
  const double* p = src + row * n;
#pragma unroll
  for (int k = 0; k < EPT; ++k) {
    const int64_t pos = base + (int64_t)k * 256;
    dst[k] = pos < n ? p[pos] : 0.0;
  }
}
template <int EPT>
__device__ __forceinline__ void load_mask_row(const uint8_t* __restrict__ src, int64_t row, int64_t n, int64_t base, uint32_t (&dst)[EPT]) {
  const uint8_t* p = src + row * n;
#pragma unroll
  for (int k = 0; k < EPT; ++k) {
    const int64_t pos = base + (int64_t)k * 256;
    dst[k] = pos < n ? p[pos] : 0u;
  }
}

// A workgroup owns VPB variants x (256 * EPT) samples.  grid (ceil(bs / VPB), nchunk), dynamic LDS 4 * VPB * Q1 doubles.
// part1[(j * nchunk + chunk) * Q1 + q], Q1 = 2 + 2C: sum, count (+ 2^26 * non-zero count), A_c (C), M_c (C)
template <int VPB, int EPT>
__global__ __launch_bounds__(256) void k_s2_proj(const double* __restrict__ G, int64_t ldg, int bs, int64_t n,
                                                 const double* __restrict__ X, int C, double* __restrict__ part1) {
  extern __shared__ double dyn_lds[];
  const int Q1 = 2 + 2 * C;
  double* red = dyn_lds;                                   // [4][VPB][Q1]
  const int j0 = blockIdx.x * VPB, w = threadIdx.x >> 6;
  double* row = red + (size_t)w * VPB * Q1;
  const int64_t base = (int64_t)blockIdx.y * (256 * EPT) + threadIdx.x;
  double g[VPB][EPT];
  uint32_t miss[VPB];
  double qa[VPB], qb[VPB];
#pragma unroll
  for (int v = 0; v < VPB; ++v) {
    miss[v] = 0;
    double s = 0.0, cnt = 0.0;
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      const int64_t pos = base + (int64_t)k * 256;
      double x = 0.0;
      if (pos < n && j0 + v < bs) {
        x = G[(int64_t)(j0 + v) * ldg + pos];
        if (is_missing(x)) { x = 0.0; miss[v] |= 1u << k; } else cnt += x != 0.0 ? 1.0 + NNZ_UNIT : 1.0;   // observed, and non-zero (exact integers)
      }
      g[v][k] = x;
      s += x;
    }
    qa[v] = s;
    qb[v] = cnt;
  }
  reduce_pairs<VPB>(qa, qb, row, Q1, 0, 1);
  double xn[EPT];
  load_row<EPT>(X, 0, n, base, xn);
  for (int c = 0; c < C; ++c) {
    double xk[EPT];
#pragma unroll
    for (int k = 0; k < EPT; ++k) xk[k] = xn[k];
    if (c + 1 < C) load_row<EPT>(X, c + 1, n, base, xn);
#pragma unroll
    for (int v = 0; v < VPB; ++v) {
      double a = 0.0, m = 0.0;
#pragma unroll
      for (int k = 0; k < EPT; ++k) {
        a += g[v][k] * xk[k];
        m += ((miss[v] >> k) & 1u) ? xk[k] : 0.0;
      }
      qa[v] = a;
      qb[v] = m;
    }
    reduce_pairs<VPB>(qa, qb, row, Q1, 2 + c, 2 + C + c);
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < VPB * Q1; idx += 256) {
    const int v = idx / Q1;
    if (j0 + v < bs)
      part1[((int64_t)(j0 + v) * gridDim.y + blockIdx.y) * Q1 + idx % Q1] =
          (red[idx] + red[VPB * Q1 + idx]) + (red[2 * VPB * Q1 + idx] + red[3 * VPB * Q1 + idx]);
  }
}

// thread = (variant j, covariate c): fixed-order sums over the chunks, mu = sum / count, beta_c = A_c + mu M_c
__global__ void k_s2_beta(const double* __restrict__ part1, int nchunk, int C, int bs, double* __restrict__ beta /*[bs][64]*/,
                          double* __restrict__ mu, int32_t* __restrict__ nobs, int32_t* __restrict__ nnz) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= bs * C) return;
  const int j = t / C, c = t % C, Q1 = 2 + 2 * C;
  double s = 0.0, cnt = 0.0, a = 0.0, m = 0.0;
  for (int ch = 0; ch < nchunk; ++ch) {
    const double* p = part1 + ((int64_t)j * nchunk + ch) * Q1;
    s += p[0]; cnt += p[1]; a += p[2 + c]; m += p[2 + C + c];
  }
  const double nz = floor(cnt / NNZ_UNIT);
  cnt -= nz * NNZ_UNIT;
  const double mean = s / cnt;   // NaN when nothing is observed: the variant comes out as ignored
  beta[(int64_t)j * RG_S2_MAX_COV + c] = a + mean * m;
  if (c == 0) { mu[j] = mean; nobs[j] = (int32_t)cnt; nnz[j] = (int32_t)nz; }
}

// grid (ceil(bs / VPB), nchunk), dynamic LDS VPB * C + VPB + 4 * VPB * (Q2 + 1) doubles.
// part2[(j * nchunk + chunk) * Q2 + q], Q2 = 1 + 2P: |r|^2, res_p . r (P), mask_p . r^2 (P)
template <int VPB, int EPT>
__global__ __launch_bounds__(256) void k_s2_score(const double* __restrict__ G, int64_t ldg, int bs, int64_t n,
                                                  const double* __restrict__ X, int C, const double* __restrict__ Y,
                                                  const uint8_t* __restrict__ M, int P, const double* __restrict__ beta,
                                                  const double* __restrict__ mu, double* __restrict__ part2) {
  extern __shared__ double dyn_lds[];
  const int Q2 = 1 + 2 * P, QS = Q2 + 1;                   // one spare column takes the unused half of the |r|^2 butterfly
  double* sb = dyn_lds;                                    // [VPB][C]
  double* smu = sb + VPB * C;                              // [VPB]
  double* red = smu + VPB;                                 // [4][VPB][QS]
  const int j0 = blockIdx.x * VPB, w = threadIdx.x >> 6;
  double* row = red + (size_t)w * VPB * QS;
  const int64_t base = (int64_t)blockIdx.y * (256 * EPT) + threadIdx.x;
  for (int idx = threadIdx.x; idx < VPB * C; idx += 256) {
    const int v = idx / C, c = idx % C;
    sb[idx] = j0 + v < bs ? beta[(int64_t)(j0 + v) * RG_S2_MAX_COV + c] : 0.0;
  }
  if (threadIdx.x < VPB) smu[threadIdx.x] = j0 + threadIdx.x < bs ? mu[j0 + threadIdx.x] : 0.0;
  __syncthreads();
  double r[VPB][EPT];
#pragma unroll
  for (int v = 0; v < VPB; ++v)
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      const int64_t pos = base + (int64_t)k * 256;
      double x = 0.0;
      if (pos < n && j0 + v < bs) {
        x = G[(int64_t)(j0 + v) * ldg + pos];
        if (is_missing(x)) x = smu[v];
      }
      r[v][k] = x;
    }
  double xn[EPT], yn[EPT];
  uint32_t mn[EPT];
  load_row<EPT>(X, 0, n, base, xn);
  for (int c = 0; c < C; ++c) {
    double xk[EPT];
#pragma unroll
    for (int k = 0; k < EPT; ++k) xk[k] = xn[k];
    if (c + 1 < C) load_row<EPT>(X, c + 1, n, base, xn);
    else { load_row<EPT>(Y, 0, n, base, yn); load_mask_row<EPT>(M, 0, n, base, mn); }
#pragma unroll
    for (int v = 0; v < VPB; ++v) {
      const double b = sb[v * C + c];
#pragma unroll
      for (int k = 0; k < EPT; ++k) r[v][k] -= b * xk[k];
    }
  }
  double qa[VPB], qb[VPB];
#pragma unroll
  for (int v = 0; v < VPB; ++v) {
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      s += r[v][k] * r[v][k];
    }
    qa[v] = s;
    qb[v] = 0.0;
  }
  reduce_pairs<VPB>(qa, qb, row, QS, 0, Q2);
  for (int p = 0; p < P; ++p) {
    double yk[EPT], mk[EPT];
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      yk[k] = yn[k];
      mk[k] = mn[k] ? 1.0 : 0.0;
    }
    if (p + 1 < P) { load_row<EPT>(Y, p + 1, n, base, yn); load_mask_row<EPT>(M, p + 1, n, base, mn); }
#pragma unroll
    for (int v = 0; v < VPB; ++v) {
      double num = 0.0, den = 0.0;
#pragma unroll
      for (int k = 0; k < EPT; ++k) {
        num += yk[k] * r[v][k];
        den += mk[k] * (r[v][k] * r[v][k]);
      }
      qa[v] = num;
      qb[v] = den;
    }
    reduce_pairs<VPB>(qa, qb, row, QS, 1 + p, 1 + P + p);
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < VPB * Q2; idx += 256) {
    const int v = idx / Q2, q = idx % Q2, o = v * QS + q;
    if (j0 + v < bs)
      part2[((int64_t)(j0 + v) * gridDim.y + blockIdx.y) * Q2 + q] =
          (red[o] + red[VPB * QS + o]) + (red[2 * VPB * QS + o] + red[3 * VPB * QS + o]);
  }
}

// grid (bs, P), 256 threads: sums over the samples masked for phenotype p (mlist[moff[p] .. moff[p + 1])) of the mean-imputed variant j:
// corr[(j * P + p) * (C + 1) + c] = sum g~_i x_c(i) for c < C, [C] = sum g~_i^2 -- what the per-trait denominators of the sparse branch
// lack relative to the all-sample sums (X^T (g~ o mask_p) = beta - corr, sum mask_p g~^2 = |g~|^2 - corr_C).
__global__ __launch_bounds__(256) void k_s2_masked_corr(const double* __restrict__ G, int64_t ldg, int64_t n, const double* __restrict__ X, int C,
                                                        const int32_t* __restrict__ mlist, const int64_t* __restrict__ moff,
                                                        const double* __restrict__ mu, int P, double* __restrict__ corr) {
  __shared__ double red[4][8];
  const int j = blockIdx.x, p = blockIdx.y;
  const int64_t e0 = moff[p], e1 = moff[p + 1];
  const double m = mu[j];
  const double* g = G + (int64_t)j * ldg;
  double* out = corr + ((int64_t)j * P + p) * (C + 1);
  for (int c0 = 0; c0 < C + 1; c0 += 8) {      // eight sums per sweep over the list
    double acc[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    for (int64_t e = e0 + threadIdx.x; e < e1; e += 256) {
      const int64_t i = mlist[e];
      double v = g[i];
      if (is_missing(v)) v = m;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int c = c0 + k;
        if (c < C) acc[k] = fma(v, X[(int64_t)c * n + i], acc[k]);
        else if (c == C) acc[k] = fma(v, v, acc[k]);
      }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      double v = acc[k];
      for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
      if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < 8 && c0 + (int)threadIdx.x <= C)
      out[c0 + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    __syncthreads();
  }
}

// thread = (variant j, phenotype p): the statistic and the effect size (Step2_Models.cpp:402-431).  check_sparse_G's rule picks the branch:
// a sparse variant is tested on its raw scale (scale_fac = 1, never "ignored" by residualize_geno) with the reference's approximate
// per-trait denominator |g~ o m_p|^2 - 2 (X^T (g~ o m_p)) . beta + |beta|^2 = |r|^2 - corr_C + 2 corr . beta; a dense one with mask_p^T r^2.
__global__ void k_s2_final(const double* __restrict__ part2, int nchunk, int P, int C, int bs, int64_t n, double numtol, double nz_max,
                           double zeros_min /* < 0: the non-zero form of the rule */,
                           const double* __restrict__ scf_sv, const int32_t* __restrict__ nobs, const int32_t* __restrict__ nnz,
                           const double* __restrict__ mu, const double* __restrict__ beta, const double* __restrict__ corr /* null: no masked sample */,
                           double* __restrict__ stats, double* __restrict__ bhat, double* __restrict__ scale_fac, int32_t* __restrict__ ignored) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= bs * P) return;
  const int j = t / P, p = t % P, Q2 = 1 + 2 * P;
  double ss = 0.0, num = 0.0, den = 0.0;
  for (int ch = 0; ch < nchunk; ++ch) {
    const double* q = part2 + ((int64_t)j * nchunk + ch) * Q2;
    ss += q[0]; num += q[1 + p]; den += q[1 + P + p];
  }
  const double m = mu[j];
  const bool sparse = zeros_min >= 0.0 ? (double)(nobs[j] - nnz[j]) >= zeros_min                      // .pgen: observed zeros
                                       : (double)nnz[j] + ((m != 0.0) ? (double)(n - nobs[j]) : 0.0) <= nz_max;   // non-zero entries of the mean-imputed vector
  double sf = 1.0;
  if (sparse) {
    den = ss;
    if (corr) {
      const double* cr = corr + ((int64_t)j * P + p) * (C + 1);
      const double* b = beta + (int64_t)j * RG_S2_MAX_COV;
      double cb = 0.0;
      for (int c = 0; c < C; ++c) cb = fma(cr[c], b[c], cb);
      den = ss - cr[C] + 2.0 * cb;
    }
  } else {
    sf = sqrt(ss) / sqrt((double)(n - C));              // residualize_geno: norm / sqrt(n_analyzed - X.cols())
  }
  const bool ign = nobs[j] == 0 || (!sparse && !(sf >= numtol));      // also catches NaN
  const double sd = sqrt(den);
  const double z = num / sd;
  stats[(int64_t)j * P + p] = ign ? NAN : z;
  bhat[(int64_t)j * P + p] = ign ? NAN : z * scf_sv[p] / sd;
  if (p == 0) { scale_fac[j] = sf; ignored[j] = ign ? 1 : 0; }
}

// ---- hard-call route ---------------------------------------------------------------------------------------------------------
// One workgroup per staged row [ldp bytes = Np / 4]: positions >= n become code 11 (0 copies, not missing), the counted allele
// is swapped when flip != 0 (00 <-> 11, the reference's --ref-first), and the calls are counted (.bed coding, Geno.cpp:2833-2856:
// 00 -> 2 copies, 01 -> missing, 10 -> 1, 11 -> 0).  cnt [bs][4] = n1, n2, nmiss, 0; *total_miss accumulates nmiss.
__global__ __launch_bounds__(256) void k_s2_rows(uint8_t* __restrict__ pk, int64_t ldp, int64_t n, int flip, int32_t* __restrict__ cnt,
                                                 int32_t* __restrict__ total_miss, double* __restrict__ vstat) {
  __shared__ int red[3][4];
  uint32_t* w = reinterpret_cast<uint32_t*>(pk + (int64_t)blockIdx.x * ldp);
  const int64_t nwords = ldp / 4;
  int c1 = 0, c2 = 0, cm = 0;
  for (int64_t i = threadIdx.x; i < nwords; i += 256) {
    const uint32_t x0 = w[i];
    uint32_t x = x0;
    if (flip) {
      const uint32_t eq = ~(x ^ (x >> 1)) & 0x55555555u;
      x ^= eq | (eq << 1);
    }
    const int64_t p0 = i * 16;
    if (p0 + 16 > n) {
      const int64_t valid = n > p0 ? n - p0 : 0;
      const uint32_t keep = valid == 0 ? 0u : ((1u << (2 * valid)) - 1u);   // valid < 16 here
      x |= ~keep;
    }
    if (x != x0) w[i] = x;
    const uint32_t lo = x & 0x55555555u, hi = (x >> 1) & 0x55555555u;
    c2 += __popc(~lo & ~hi & 0x55555555u);
    cm += __popc(lo & ~hi);
    c1 += __popc(hi & ~lo);
  }
  for (int o = 32; o > 0; o >>= 1) { c1 += __shfl_down(c1, o); c2 += __shfl_down(c2, o); cm += __shfl_down(cm, o); }
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = c1; red[1][threadIdx.x >> 6] = c2; red[2][threadIdx.x >> 6] = cm; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int a = red[0][0] + red[0][1] + red[0][2] + red[0][3], b = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    const int m = red[2][0] + red[2][1] + red[2][2] + red[2][3];
    cnt[blockIdx.x * 4 + 0] = a; cnt[blockIdx.x * 4 + 1] = b; cnt[blockIdx.x * 4 + 2] = m; cnt[blockIdx.x * 4 + 3] = 0;
    if (vstat) {   // sum, sum of squares, observed, observed non-zero (k_s2_packed_final)
      double* vs = vstat + (int64_t)blockIdx.x * 4;
      vs[0] = (double)(a + 2 * b); vs[1] = (double)(a + 4 * b); vs[2] = (double)(n - m); vs[3] = (double)(a + b);
    }
    if (m) atomicAdd(total_miss, m);
  }
}

// YtX[p][c] = sum_i res_p(i) x_c(i); grid (P * C), 256 threads, fixed summation tree
__global__ __launch_bounds__(256) void k_s2_ytx(const double* __restrict__ X, const double* __restrict__ Y, int64_t n, int C, double* __restrict__ ytx) {
  __shared__ double red[4];
  const int p = blockIdx.x / C, c = blockIdx.x % C;
  const double* x = X + (int64_t)c * n;
  const double* y = Y + (int64_t)p * n;
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += 256) s = fma(x[i], y[i], s);
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) ytx[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// V[C + P + p * C + c][i] = x_c(i) mask_p(i), V[cm0 + p][i] = mask_p(i); grid (ceil(n / 256), P)
__global__ void k_s2_mask_cols(const double* __restrict__ X, const uint8_t* __restrict__ M, int64_t n, int64_t Np, int C, int P, int cm0,
                               double* __restrict__ V) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int p = blockIdx.y;
  if (i >= n) return;
  const double m = M[(int64_t)p * n + i] ? 1.0 : 0.0;
  for (int c = 0; c < C; ++c) V[(int64_t)(C + P + p * C + c) * Np + i] = X[(int64_t)c * n + i] * m;
  V[(int64_t)(cm0 + p) * Np + i] = m;
}

// out[j][set][c] = vsc[c] * sum_k 128^k (sum over segments of S[c >> 4][set][f][j][(c & 15) * 8 + k]), c < Cv: the segment sums are exact in
// int64, so the value does not depend on how the samples were cut into segments.  thread = (j, c); set 1 only when total_miss says the
// block has a missing call (total_miss == nullptr: set 0 only).
__global__ void k_s2_combine(const int32_t* __restrict__ S, const double* __restrict__ vsc, const int32_t* __restrict__ total_miss, int bs,
                             int n128, int nseg, int Cv, double* __restrict__ out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= bs * Cv) return;
  const int j = t / Cv, c = t - j * Cv, grp = c >> 4, cl = c & 15;
  const int nset = (total_miss && *total_miss > 0) ? 2 : 1;
  for (int set = 0; set < 2; ++set) {
    double v = 0.0;
    if (set < nset) {
      long long tk[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int f = 0; f < nseg; ++f) {
        const int32_t* s = S + (((((int64_t)grp * 2 + set) * nseg + f) * n128 + j) * 128) + cl * 8;
#pragma unroll
        for (int k = 0; k < 8; ++k) tk[k] += s[k];
      }
#pragma unroll
      for (int k = 7; k >= 0; --k) v = fma(v, 128.0, (double)tk[k]);
      v *= vsc[c];
    }
    out[((int64_t)j * 2 + set) * Cv + c] = v;
  }
}

// ---- compact sample axis of a masked problem (hard calls) ----------------------------------------------------------------------------
// out[row][e] = the 2-bit code of sample clist[e] (code 11 = 0 copies, not missing, for padding).  The compact axis is built chunk-aware
// (ensure_planes): within a phenotype's range the entries of sample chunk k (S2_CS samples) come first padded to 16, then those of chunk
// k + 1, ... so that a workgroup = (chunk k, 8 rows) holds the rows' bytes of that chunk in LDS (64 KB), reads every list entry of the chunk
// ONCE for its 8 rows and writes whole 32-bit words.  The LDS image is byte-transposed -- the 8 rows' bytes of byte position B are the 8
// bytes at 8 B -- so ONE ds_read_b64 serves an entry for all 8 rows (random byte reads of a row-major image kept the kernel at the LDS's
// random-access rate: 0.59 ms per 8,192 rows; one workgroup per row with the whole row in LDS and the whole 1 MB list read by every
// workgroup: 1.34 ms).
// cmeta [K][P][2]: first 32-bit word of the (chunk, phenotype) range in the compact row, its words.  grid (K, ceil(bs / 8)), S2_CT threads.
#define S2_CS 32768
#define S2_CR 8
#define S2_CT 1024      // threads: a chunk has about one 32-bit word of list entries per thread (256 threads, four words each in turn: latency bound)
// (Taking k_s2_rows' work -- allele swap, padding codes, call counts -- into this kernel so that the staged rows are read once was built and
// measured: 1.03 ms against 0.54 + 0.26 ms for the two kernels; the counts' atomics and the write-back sit badly in the chunked grid.)
__global__ __launch_bounds__(S2_CT) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_s2_compact_rows(const uint8_t* __restrict__ pk, int64_t ldp, const int32_t* __restrict__ clist,
                                                           const int32_t* __restrict__ cmeta, int P, int bs, uint8_t* __restrict__ out, int64_t ldc) {
  extern __shared__ __attribute__((aligned(16))) uint8_t srow[];      // [S2_CS / 4 + 1][8]: byte position, row; the last position is 0xFF (padding entries read it)
  __shared__ int pre[RG_S2_MAX_PHENO + 1];
  __shared__ int cnt[RG_S2_MAX_PHENO];
  const int k = blockIdx.x, r0 = blockIdx.y * S2_CR, nr = min(S2_CR, bs - r0), tid = threadIdx.x;
  constexpr int cb = S2_CS / 4;
  const int64_t b0 = (int64_t)k * cb;
  const int nd = (int)(min((int64_t)cb, ldp - b0) / 4);               // dwords of a row in this chunk (ldp is a multiple of 16)
  for (int o = tid; o < cb / 4; o += S2_CT) {
    uint32_t d[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) d[r] = (r < nr && o < nd) ? reinterpret_cast<const uint32_t*>(pk + (int64_t)(r0 + r) * ldp + b0)[o] : 0xFFFFFFFFu;
    // byte transpose: q[b][h] = byte b of rows 4 h .. 4 h + 3
    uint32_t q[4][2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const uint32_t t01l = __builtin_amdgcn_perm(d[4 * h + 1], d[4 * h], 0x05010400u), t01h = __builtin_amdgcn_perm(d[4 * h + 1], d[4 * h], 0x07030602u);
      const uint32_t t23l = __builtin_amdgcn_perm(d[4 * h + 3], d[4 * h + 2], 0x05010400u), t23h = __builtin_amdgcn_perm(d[4 * h + 3], d[4 * h + 2], 0x07030602u);
      q[0][h] = __builtin_amdgcn_perm(t23l, t01l, 0x05040100u);
      q[1][h] = __builtin_amdgcn_perm(t23l, t01l, 0x07060302u);
      q[2][h] = __builtin_amdgcn_perm(t23h, t01h, 0x05040100u);
      q[3][h] = __builtin_amdgcn_perm(t23h, t01h, 0x07060302u);
    }
    uint4* dst = reinterpret_cast<uint4*>(srow + (size_t)o * 32);
    dst[0] = make_uint4(q[0][0], q[0][1], q[1][0], q[1][1]);
    dst[1] = make_uint4(q[2][0], q[2][1], q[3][0], q[3][1]);
  }
  if (tid == 0) *reinterpret_cast<uint2*>(srow + (size_t)cb * 8) = make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
  // word counts of the chunk's P ranges (loaded side by side), then the prefix sums
  if (tid < P) cnt[tid] = cmeta[((int64_t)k * P + tid) * 2 + 1];
  __syncthreads();
  if (tid == 0) {
    int acc = 0;
    for (int p = 0; p < P; ++p) { pre[p] = acc; acc += cnt[p]; }
    pre[P] = acc;
  }
  __syncthreads();
  const int tot = pre[P], base = k * S2_CS;
  for (int i = tid; i < tot; i += S2_CT) {
    int p = 0;
    while (pre[p + 1] <= i) ++p;
    const int64_t w = (int64_t)cmeta[((int64_t)k * P + p) * 2] + (i - pre[p]);
    const int4* ci = reinterpret_cast<const int4*>(clist + w * 16);
    int ids[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) { const int4 id = ci[q]; ids[4 * q] = id.x; ids[4 * q + 1] = id.y; ids[4 * q + 2] = id.z; ids[4 * q + 3] = id.w; }
    uint32_t word[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int loc = ids[e] - base;
      const int ad = ids[e] < 0 ? cb * 8 : (loc >> 2) * 8;
      const int sh = ids[e] < 0 ? 0 : 2 * (loc & 3);
      const uint2 v = *reinterpret_cast<const uint2*>(srow + ad);
      const uint32_t lo = v.x >> sh, hi = v.y >> sh;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        word[r] |= __builtin_amdgcn_ubfe(lo, 8u * r, 2u) << (2 * e);
        word[4 + r] |= __builtin_amdgcn_ubfe(hi, 8u * r, 2u) << (2 * e);
      }
    }
#pragma unroll
    for (int r = 0; r < 8; ++r)
      if (r < nr) reinterpret_cast<uint32_t*>(out + (int64_t)(r0 + r) * ldc)[w] = word[r];
  }
}

// (Counting inside k_s2_compact_rows, on the words it holds in registers, with one 64-bit LDS atomic per row and word: 0.79 ms for that kernel
// against 0.49 + 0.14 ms for the two -- the lanes of a wave add to the same (row, phenotype) counter.)
// call counts over a phenotype's range of the compact rows: n1, n2, nmiss [row][p][4] (the column of ones of the compact contraction, both sets,
// and the squares, without a contraction: sum g = n1 + 2 n2, sum g^2 = n1 + 4 n2, missing = nmiss; padding = code 11 counts nowhere).
// grid (bs, P), 256 threads; w0 [P + 1] = first 32-bit word of a phenotype's range.
__global__ __launch_bounds__(256) void k_s2_count_traits(const uint8_t* __restrict__ pkc, int64_t ldc, const int32_t* __restrict__ w0, int32_t* __restrict__ out) {
  __shared__ int red[3][4];
  const int p = blockIdx.y, P = gridDim.y;
  const uint32_t* w = reinterpret_cast<const uint32_t*>(pkc + (int64_t)blockIdx.x * ldc);
  int c1 = 0, c2 = 0, cm = 0;
  for (int i = w0[p] + threadIdx.x; i < w0[p + 1]; i += 256) {
    const uint32_t x = w[i];
    const uint32_t lo = x & 0x55555555u, hi = (x >> 1) & 0x55555555u;
    c2 += __popc(~lo & ~hi & 0x55555555u);
    cm += __popc(lo & ~hi);
    c1 += __popc(hi & ~lo);
  }
  for (int o = 32; o > 0; o >>= 1) { c1 += __shfl_down(c1, o); c2 += __shfl_down(c2, o); cm += __shfl_down(cm, o); }
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = c1; red[1][threadIdx.x >> 6] = c2; red[2][threadIdx.x >> 6] = cm; }
  __syncthreads();
  if (threadIdx.x == 0) {
    int32_t* o = out + ((int64_t)blockIdx.x * P + p) * 4;
    o[0] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    o[1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    o[2] = red[2][0] + red[2][1] + red[2][2] + red[2][3];
    o[3] = 0;
  }
}

// Lc[j][set][p][c] = vsc[c] * sum_k 128^k (sum over phenotype p's segments of S[c >> 4][set][f][j][(c & 15) * 8 + k]), c < C; c = C (the column
// of ones) and Lq[j][p] (the squares) come from the call counts of k_s2_count_traits.  thread = (j, tt, c <= C);
// phenotypes t0 .. t0 + nt - 1 of this launch, spt segments each.
__global__ void k_s2_combine_traits(const int32_t* __restrict__ S, const int32_t* __restrict__ tcnt, const double* __restrict__ vsc,
                                    const int32_t* __restrict__ total_miss, int bs, int n128, int nseg, int spt, int t0, int nt, int P, int C,
                                    double* __restrict__ Lc, double* __restrict__ Lq) {
  const int Cc = C + 1;
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (int64_t)bs * nt * Cc) return;
  const int c = (int)(t % Cc);
  const int tt = (int)((t / Cc) % nt);
  const int j = (int)(t / ((int64_t)Cc * nt));
  const int p = t0 + tt;
  if (c == C) {
    const int32_t* q = tcnt + ((int64_t)j * P + p) * 4;
    Lc[(((int64_t)j * 2 + 0) * P + p) * Cc + C] = (double)(q[0] + 2 * q[1]);
    Lc[(((int64_t)j * 2 + 1) * P + p) * Cc + C] = (double)q[2];
    Lq[(int64_t)j * P + p] = (double)(q[0] + 4 * q[1]);
    return;
  }
  const int grp = c >> 4, cl = c & 15;
  const int nset = (*total_miss > 0) ? 2 : 1;
  for (int set = 0; set < 2; ++set) {
    double v = 0.0;
    if (set < nset) {
      long long tk[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int f = tt * spt; f < (tt + 1) * spt; ++f) {
        const int32_t* s = S + (((((int64_t)grp * 2 + set) * nseg + f) * n128 + j) * 128) + cl * 8;
#pragma unroll
        for (int k = 0; k < 8; ++k) tk[k] += s[k];
      }
#pragma unroll
      for (int k = 7; k >= 0; --k) v = fma(v, 128.0, (double)tk[k]);
      v *= vsc[c];
    }
    Lc[(((int64_t)j * 2 + set) * P + p) * Cc + c] = v;
  }
}

// vstat [bs][4] (doubles, exact integers): sum of the observed entries, sum of their squares (both in the row's integer units), observed
// count, observed non-zero count.  Hard calls: k_s2_rows fills it from the call counts; integer dosages: k_s2_int_rows.
struct PackedFinal {
  const double* A;      // [bs][2][Cvt]: (row . column) and (missing indicator . column), in genotype units
  const double* Sq;     // [bs][2][CvB] (hard calls, masked problems) or null
  const double* vstat;  // [bs][4]
  const double* corr;   // [bs][P][C + 2] (integer dosages, masked problems: k_s2_masked_int) or null
  const double *Lc, *Lq; // masked == 3: [bs][2][P][C + 1] sums over the samples masked for the trait against x_0 .. x_{C-1}, 1; [bs][P] of the squares
  const double *ytx, *Q, *msum, *scf_sv;
  int bs, C, P, Cvt, cm0, CvB, sqoff, masked;   // masked: 0 none, 1 = mask columns (hard calls), 2 = masked-sample lists (integer dosages),
                                                // 3 = complement over the compact axis (hard calls)
  int64_t n;
  double inv_scale;        // genotype units per integer unit (1 for hard calls, 1 / 255 for 8-bit .bgen probabilities, ...)
  double numtol, nz_max;   // a variant is "sparse" when its non-zero entries number <= nz_max ...
  double zeros_min;        // ... or (>= 0, the .pgen form) when its observed zeros number >= zeros_min
  double *stats, *bhat, *scale_fac, *mean, *total_p;
  int32_t *nobs, *ignored, *nobs_p;
};

// thread = (variant, phenotype): the statistic from the contractions and the counts (formulas at the top of the file)
__global__ void k_s2_packed_final(PackedFinal a) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= a.bs * a.P) return;
  const int j = t / a.P, p = t - j * a.P, C = a.C, P = a.P;
  const double* vs = a.vstat + (int64_t)j * 4;
  const double nobs = vs[2], nm = (double)a.n - nobs, nnz = vs[3];
  const double tot_all = vs[0] * a.inv_scale, sq_all = vs[1] * a.inv_scale * a.inv_scale;
  const double mu = nobs > 0 ? tot_all / nobs : 0.0;
  const double* a0 = a.A + (int64_t)j * 2 * a.Cvt;
  const double* a1 = a0 + a.Cvt;
  double b2 = 0.0, corr = 0.0;
  for (int c = 0; c < C; ++c) {
    const double b = fma(mu, a1[c], a0[c]);
    b2 = fma(b, b, b2);
    corr = fma(a.ytx[p * C + c], b, corr);
  }
  const double num = fma(mu, a1[C + p], a0[C + p]) - corr;
  const double ss = (sq_all + nm * mu * mu) - b2;                             // |g~ - X beta|^2 over every analysed sample
  const bool sparse = a.zeros_min >= 0.0 ? (nobs - nnz) >= a.zeros_min
                                         : (nnz + (mu != 0.0 ? nm : 0.0)) <= a.nz_max;            // check_sparse_G on the mean-imputed vector
  const double sf = sparse ? 1.0 : sqrt(ss) / sqrt((double)(a.n - C));       // residualize_geno only runs for dense variants
  const bool ign = nobs <= 0 || (!sparse && !(sf >= a.numtol));
  double den = ss, tot = tot_all, nobs_p = nobs;
  if (a.masked == 1) {
    const double* sq = a.Sq + (int64_t)j * 2 * a.CvB + a.sqoff;
    const double g2m = sq[p] + mu * mu * a1[a.cm0 + p];
    const double* x0 = a0 + C + P + p * C;
    const double* x1 = a1 + C + P + p * C;
    double cross = 0.0;
    for (int c = 0; c < C; ++c) cross = fma(fma(mu, x1[c], x0[c]), fma(mu, a1[c], a0[c]), cross);
    double last = b2;
    if (!sparse) {
      last = 0.0;
      const double* q = a.Q + (int64_t)p * C * C;
      for (int c = 0; c < C; ++c) {
        double row = 0.0;
        for (int d = 0; d < C; ++d) row = fma(q[c * C + d], fma(mu, a1[d], a0[d]), row);
        last = fma(row, fma(mu, a1[c], a0[c]), last);
      }
    }
    den = g2m - 2.0 * cross + last;
    tot = a0[a.cm0 + p];
    nobs_p = a.msum[p] - a1[a.cm0 + p];
  } else if (a.masked == 3) {
    // the mask-column sums of mode 1 as (every sample) - (the samples masked for the trait): sum mask_p g x_c = a[c] - l[c],
    // sum mask_p g = (sum g) - l[C], sum mask_p [missing] = (missing calls) - l1[C], sum mask_p g^2 = (sum g^2) - Lq
    const int Cc = C + 1;
    const double* l0 = a.Lc + (((int64_t)j * 2) * P + p) * Cc;
    const double* l1 = a.Lc + (((int64_t)j * 2 + 1) * P + p) * Cc;
    const double m0 = tot_all - l0[C], m1 = nm - l1[C];
    const double g2m = (sq_all - a.Lq[(int64_t)j * P + p]) + mu * mu * m1;
    double cross = 0.0;
    for (int c = 0; c < C; ++c) cross = fma(fma(mu, a1[c] - l1[c], a0[c] - l0[c]), fma(mu, a1[c], a0[c]), cross);
    double last = b2;
    if (!sparse) {
      last = 0.0;
      const double* q = a.Q + (int64_t)p * C * C;
      for (int c = 0; c < C; ++c) {
        double row = 0.0;
        for (int d = 0; d < C; ++d) row = fma(q[c * C + d], fma(mu, a1[d], a0[d]), row);
        last = fma(row, fma(mu, a1[c], a0[c]), last);
      }
    }
    den = g2m - 2.0 * cross + last;
    tot = m0;
    nobs_p = a.msum[p] - m1;
  } else if (a.masked == 2) {
    // sums over the samples masked for the trait: corr_c = sum g~ x_c, corr2 = sum g~^2, rs2 = sum (g~ - x . beta)^2
    const double* cr = a.corr + ((int64_t)j * P + p) * (C + 2);
    if (sparse) {
      double cb = 0.0;
      for (int c = 0; c < C; ++c) cb = fma(cr[c], fma(mu, a1[c], a0[c]), cb);
      den = ss - cr[C] + 2.0 * cb;          // |g~ o m|^2 - 2 (X^T (g~ o m)) . beta + |beta|^2
    } else {
      den = ss - cr[C + 1];                 // mask^T r^2
    }
  }
  const double sd = sqrt(den);
  const double z = num / sd;
  a.stats[(int64_t)j * P + p] = ign ? NAN : z;
  a.bhat[(int64_t)j * P + p] = ign ? NAN : z * a.scf_sv[p] / sd;
  a.total_p[(int64_t)j * P + p] = tot;
  a.nobs_p[(int64_t)j * P + p] = (int32_t)nearbyint(nobs_p);
  if (p == 0) { a.scale_fac[j] = sf; a.mean[j] = mu; a.nobs[j] = (int32_t)nobs; a.ignored[j] = ign ? 1 : 0; }
}

// ---- integer dosages (rg_s2_qt_block_int) -----------------------------------------------------------------------------------------
// One workgroup per row of uint16 dosages in units of 1 / scale (0xFFFF = missing): balanced base-128 digit planes d_0 .. d_{D-1}
// (G = sum_k 128^k d_k, d_k in [-64, 63]) and the missing indicator as plane D, zero padded to Np; vstat as for hard calls (exact: the
// sums stay below 2^53).  planes [D + 1][bs][Np].
template <bool VEC>
__global__ __launch_bounds__(256) void k_s2_int_rows(const uint16_t* __restrict__ G, int64_t ld, int64_t n, int64_t Np, int D, int8_t* __restrict__ planes,
                                                     int64_t set_stride, double* __restrict__ vstat, int32_t* __restrict__ total_miss) {
  __shared__ long long red[4][4];
  const int j = blockIdx.x;
  const uint16_t* g = G + (int64_t)j * ld;
  int8_t* out = planes + (int64_t)j * Np;
  long long sg = 0, sg2 = 0, no = 0, nz = 0;
  for (int64_t i0 = (int64_t)threadIdx.x * 8; i0 < Np; i0 += 256 * 8) {      // eight entries per thread: one 16-byte load, one 8-byte store per plane
    unsigned v8[8];
    if (VEC && i0 + 8 <= n) {
      const uint4 w = *reinterpret_cast<const uint4*>(g + i0);
      v8[0] = w.x & 0xFFFFu; v8[1] = w.x >> 16; v8[2] = w.y & 0xFFFFu; v8[3] = w.y >> 16;
      v8[4] = w.z & 0xFFFFu; v8[5] = w.z >> 16; v8[6] = w.w & 0xFFFFu; v8[7] = w.w >> 16;
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v8[e] = i0 + e < n ? g[i0 + e] : 0u;
    }
    unsigned long long dig[3] = {0ull, 0ull, 0ull}, mbits = 0ull;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const bool miss = v8[e] == 0xFFFFu;
      int v = miss ? 0 : (int)v8[e];
      if (i0 + e < n && !miss) { sg += v; sg2 += (long long)v * v; ++no; nz += v != 0; }
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int d = ((v & 127) ^ 64) - 64;
        dig[k] |= (unsigned long long)(uint8_t)(int8_t)d << (8 * e);
        v = (v - d) >> 7;
      }
      mbits |= (unsigned long long)(miss ? 1u : 0u) << (8 * e);
    }
    for (int k = 0; k < D; ++k) *reinterpret_cast<unsigned long long*>(out + (int64_t)k * set_stride + i0) = dig[k];
    *reinterpret_cast<unsigned long long*>(out + (int64_t)D * set_stride + i0) = mbits;
  }
  long long vals[4] = {sg, sg2, no, nz};
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    long long x = vals[q];
    for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o);
    if ((threadIdx.x & 63) == 0) red[q][threadIdx.x >> 6] = x;
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    const long long x = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
    vstat[(int64_t)j * 4 + threadIdx.x] = (double)x;
    if (threadIdx.x == 2 && x < n) atomicAdd(total_miss, (int)(n - x));
  }
}

// sq[j][q] = sum over the observed entries of (G_i / scale)^2 col_q(i), q < nsq (fp64; the binary-trait test's sum w g^2 for dosages).
// grid (bs, ceil(nsq / 8)), 256 threads
__global__ __launch_bounds__(256) void k_s2_int_sq(const uint16_t* __restrict__ G, int64_t ld, int64_t n, int64_t Np, double inv_scale,
                                                   const double* __restrict__ V, int nsq, double* __restrict__ sq) {
  __shared__ double red[4][8];
  const int j = blockIdx.x, q0 = blockIdx.y * 8;
  const uint16_t* g = G + (int64_t)j * ld;
  double acc[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  for (int64_t i = threadIdx.x; i < n; i += 256) {
    const unsigned v0 = g[i];
    if (v0 == 0xFFFFu) continue;
    const double v = (double)v0 * inv_scale, v2 = v * v;
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (q0 + k < nsq) acc[k] = fma(v2, V[(int64_t)(q0 + k) * Np + i], acc[k]);
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    double v = acc[k];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < 8 && q0 + (int)threadIdx.x < nsq)
    sq[(int64_t)j * nsq + q0 + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// out[j][0][c] = vsc[c] / scale * sum_k 128^k (T_k^0 + 128 T_k^1 + 128^2 T_k^2), out[j][1][c] = vsc[c] * sum_k 128^k T_k^D, T_k^s = the sum over
// the segments of S[c >> 4][s][f][j][(c & 15) * 8 + k] (int64: exact).  thread = (j, c)
__global__ void k_s2_int_combine(const int32_t* __restrict__ S, const double* __restrict__ vsc, int bs, int n128, int nseg, int nset, int D, double inv_scale,
                                 int Cv, double* __restrict__ out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= bs * Cv) return;
  const int j = t / Cv, c = t - j * Cv, grp = c >> 4, cl = c & 15;
  long long u[8] = {0, 0, 0, 0, 0, 0, 0, 0}, m[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int set = 0; set < nset; ++set) {
    const long long wgt = set == 0 ? 1 : (set == 1 ? 128 : 16384);
    for (int f = 0; f < nseg; ++f) {
      const int32_t* s = S + (((((int64_t)grp * nset + set) * nseg + f) * n128 + j) * 128) + cl * 8;
      if (set < D) {
#pragma unroll
        for (int k = 0; k < 8; ++k) u[k] += wgt * s[k];
      } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) m[k] += s[k];
      }
    }
  }
  double v = 0.0, w = 0.0;
#pragma unroll
  for (int k = 7; k >= 0; --k) { v = fma(v, 128.0, (double)u[k]); w = fma(w, 128.0, (double)m[k]); }
  out[((int64_t)j * 2 + 0) * Cv + c] = v * vsc[c] * inv_scale;
  out[((int64_t)j * 2 + 1) * Cv + c] = w * vsc[c];
}

// beta[j][c] = A0 + mu A1 (thread = (j, c)), for the masked-sample sums below
__global__ void k_s2_beta_from_sums(const double* __restrict__ A, const double* __restrict__ vstat, double inv_scale, int bs, int C, int Cv,
                                    double* __restrict__ beta /*[bs][RG_S2_MAX_COV]*/) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= bs * C) return;
  const int j = t / C, c = t - j * C;
  const double nobs = vstat[(int64_t)j * 4 + 2];
  const double mu = nobs > 0 ? vstat[(int64_t)j * 4] * inv_scale / nobs : 0.0;
  beta[(int64_t)j * RG_S2_MAX_COV + c] = fma(mu, A[((int64_t)j * 2 + 1) * Cv + c], A[((int64_t)j * 2) * Cv + c]);
}

// grid (bs, P), 256 threads: over the samples masked for phenotype p (mlist[moff[p] .. moff[p + 1])) of the mean-imputed variant j:
// corr[(j * P + p) * (C + 2) + c] = sum g~_i x_c(i) (c < C), [C] = sum g~_i^2, [C + 1] = sum (g~_i - x_i . beta_j)^2.
// xl [entry][C]: the covariate rows of the listed samples, compact and C-contiguous (ensure_lists) -- one 8 C-byte read per entry instead
// of C reads n doubles apart.
__global__ __launch_bounds__(256) void k_s2_masked_int(const uint16_t* __restrict__ G, int64_t ld, double inv_scale, const double* __restrict__ xl, int C,
                                                       const int32_t* __restrict__ mlist, const int64_t* __restrict__ moff,
                                                       const double* __restrict__ vstat, const double* __restrict__ beta, int P, double* __restrict__ corr) {
  __shared__ double red[4][18];
  __shared__ double sbeta[RG_S2_MAX_COV];
  const int j = blockIdx.x, p = blockIdx.y;
  const int64_t e0 = moff[p], e1 = moff[p + 1];
  const double nobs = vstat[(int64_t)j * 4 + 2];
  const double mu = nobs > 0 ? vstat[(int64_t)j * 4] * inv_scale / nobs : 0.0;
  const uint16_t* g = G + (int64_t)j * ld;
  double* out = corr + ((int64_t)j * P + p) * (C + 2);
  if (threadIdx.x < C) sbeta[threadIdx.x] = beta[(int64_t)j * RG_S2_MAX_COV + threadIdx.x];
  __syncthreads();
  for (int c0 = 0; c0 < C; c0 += 16) {         // sixteen covariates per sweep; the first sweep also carries the two quadratic sums
    const int nc = min(16, C - c0);
    double acc[18];
#pragma unroll
    for (int k = 0; k < 18; ++k) acc[k] = 0.0;
    for (int64_t e = e0 + threadIdx.x; e < e1; e += 256) {
      const unsigned v0 = g[mlist[e]];
      const double v = v0 == 0xFFFFu ? mu : (double)v0 * inv_scale;
      const double* x = xl + e * C;
#pragma unroll
      for (int k = 0; k < 16; ++k)
        if (k < nc) acc[k] = fma(v, x[c0 + k], acc[k]);
      if (c0 == 0) {
        double r = v;
        for (int d = 0; d < C; ++d) r = fma(-x[d], sbeta[d], r);
        acc[16] = fma(v, v, acc[16]);
        acc[17] = fma(r, r, acc[17]);
      }
    }
#pragma unroll
    for (int k = 0; k < 18; ++k) {
      double v = acc[k];
      for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
      if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < 18) {
      const double v = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
      if ((int)threadIdx.x < nc) out[c0 + threadIdx.x] = v;
      else if (c0 == 0 && threadIdx.x >= 16) out[C + (threadIdx.x - 16)] = v;
    }
    __syncthreads();
  }
}

// The same sums for SIXTEEN variants per workgroup on the fp64 matrix cores: grid (ceil(bs / 16), P).  k_s2_masked_int reads the listed
// samples' covariate rows once per (variant, phenotype) -- 20 MB per variant at 500,000 samples, 10 phenotypes with 5 % missing values each --
// and that traffic was 90 % of the integer-dosage route's time.  Here the rows are read once per sixteen variants: v_mfma_f64_16x16x4 with
// A[variant][entry] = the mean-imputed dosage of the entry's sample (a 2-byte gather per lane) and B[entry][covariate] = the entry's row,
// four entries per instruction, the four waves taking every fourth group of entries.  sum g^2 rides along per lane; the residual sum follows
// from the products: sum (g - x.beta)^2 = sum g^2 - 2 beta.(X^T g) + beta^T Q_p beta, Q_p = X^T X over the phenotype's list (ensure_lists).
__global__ __launch_bounds__(256) void k_s2_masked_int_mfma(const uint16_t* __restrict__ G, int64_t ld, double inv_scale, const double* __restrict__ xl,
                                                            const double* __restrict__ xq, int C, const int32_t* __restrict__ mlist,
                                                            const int64_t* __restrict__ moff, const double* __restrict__ vstat,
                                                            const double* __restrict__ beta, int bs, int P, double* __restrict__ corr) {
  __shared__ double sacc[4][2][16][17];
  __shared__ double svv[4][16];
  const int jt = blockIdx.x, p = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, kq = lane >> 4;
  const int j = min(jt * 16 + i, bs - 1);
  const int64_t e0 = moff[p], e1 = moff[p + 1];
  const double nobs = vstat[(int64_t)j * 4 + 2];
  const double mu = nobs > 0 ? vstat[(int64_t)j * 4] * inv_scale / nobs : 0.0;
  const uint16_t* g = G + (int64_t)j * ld;
  const bool two = C > 16;
  const bool col0 = i < C, col1 = 16 + i < C;
  v4d acc0 = (v4d){0, 0, 0, 0}, acc1 = (v4d){0, 0, 0, 0};
  double vv = 0.0;
  // groups of four entries; wave w takes groups w, w + 4, ...; four groups per trip so that their gathers are in flight together
  const int64_t ngrp = (e1 - e0 + 3) / 4;
  for (int64_t gq = wave; gq < ngrp; gq += 16) {
    double a[4], b0[4], b1[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t e = e0 + 4 * (gq + 4 * u) + kq;
      const bool live = e < e1;
      const int64_t ec = live ? e : e0;                       // clamped address, value selected afterwards (no load under a condition)
      const unsigned v0 = g[mlist[ec]];
      const double v = v0 == 0xFFFFu ? mu : (double)v0 * inv_scale;
      a[u] = live ? v : 0.0;
      const double* x = xl + ec * C;
      const double x0 = x[col0 ? i : 0];
      b0[u] = (live && col0) ? x0 : 0.0;
      if (two) { const double x1 = x[col1 ? 16 + i : 0]; b1[u] = (live && col1) ? x1 : 0.0; }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b0[u], acc0, 0, 0, 0);
      if (two) acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b1[u], acc1, 0, 0, 0);
      vv = fma(a[u], a[u], vv);
    }
  }
  // D[row = kq + 4 r][col = i]: rows are the variants, columns the covariates
#pragma unroll
  for (int r = 0; r < 4; ++r) { sacc[wave][0][kq + 4 * r][i] = acc0[r]; sacc[wave][1][kq + 4 * r][i] = two ? acc1[r] : 0.0; }
  vv += __shfl_xor(vv, 16);
  vv += __shfl_xor(vv, 32);
  if (kq == 0) svv[wave][i] = vv;
  __syncthreads();
  {
    const int vi = threadIdx.x >> 4, n = threadIdx.x & 15;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      const double s = (sacc[0][ct][vi][n] + sacc[1][ct][vi][n]) + (sacc[2][ct][vi][n] + sacc[3][ct][vi][n]);
      __syncthreads();
      sacc[0][ct][vi][n] = s;
    }
  }
  __syncthreads();
  if (threadIdx.x < 16) {
    const int vi = threadIdx.x, jg = jt * 16 + vi;
    if (jg < bs) {
      double* out = corr + ((int64_t)jg * P + p) * (C + 2);
      const double* bj = beta + (int64_t)jg * RG_S2_MAX_COV;
      const double* Q = xq + (int64_t)p * C * C;
      const double gg = (svv[0][vi] + svv[1][vi]) + (svv[2][vi] + svv[3][vi]);
      double bx = 0.0, bqb = 0.0;
      for (int c = 0; c < C; ++c) {
        const double xv = sacc[0][c >> 4][vi][c & 15];
        out[c] = xv;
        bx = fma(bj[c], xv, bx);
        double t = 0.0;
        for (int d = 0; d < C; ++d) t = fma(Q[c * C + d], bj[d], t);
        bqb = fma(bj[c], t, bqb);
      }
      out[C] = gg;
      out[C + 1] = gg - 2.0 * bx + bqb;
    }
  }
}

}  // namespace

namespace {
int fail(rg_s2_ctx* ctx, int code, const std::string& msg) { return rg_s2_fail(ctx, code, msg); }
int ensure_in(rg_s2_ctx* ctx, void** buf, size_t* cap, int slot, size_t bytes) { return rg_s2_ensure_in(ctx, buf, cap, slot, bytes); }
// The sample axis (Np, a multiple of 128 * 32) is cut into nseg equal segments, one workgroup per (row tile, segment, column group, set):
// enough workgroups to fill the 256 CUs (>= 768), as few segments as that allows -- every segment costs a 64 KB tile of partial sums.
int pick_segments(int64_t Np, int tiles, int ngrp, SegLayout& seg) {
  int nseg = 1;
  while (nseg < RG_MAX_SEG && (int64_t)tiles * nseg * ngrp < 768 && Np / (nseg * 2) >= 1024) nseg *= 2;
  // the int32 sums of a segment must not wrap: |digit| <= 64 times a LUT value of at most 4 (the squared allele count) per sample
  while (nseg < RG_MAX_SEG && Np / nseg > (1 << 23)) nseg *= 2;
  memset(&seg, 0, sizeof(seg));
  seg.nseg = nseg;
  for (int f = 0; f < nseg; ++f) { seg.pos_start[f] = f * (Np / nseg); seg.file_start[f] = seg.pos_start[f]; seg.len[f] = Np / nseg; seg.plen[f] = Np / nseg; }
  return nseg;
}
int ensure(rg_s2_ctx* ctx, int slot, size_t bytes) { return ensure_in(ctx, ctx->buf, ctx->cap, slot, bytes); }
// The int8 digit planes of the contraction columns [X | res (| x_c mask_p | mask_p)]: rebuilt when X or the masks changed (everything but
// res) and after every rg_s2_set_null (the res columns, res^T X).  want_mask_cols: the hard-call route's extra columns for phenotypes that
// differ in their missing values (the integer-dosage route uses masked-sample lists instead).
int ensure_planes(rg_s2_ctx* ctx, bool want_mask_cols) {
  const int64_t n = ctx->n;
  const int C = ctx->C, P = ctx->P;
  const bool mask_cols = want_mask_cols && !ctx->complete;
  if (!ctx->static_ready || ctx->planes_mask_cols != mask_cols) {   // X or the masks changed: column layout, the planes of every column but res, Q_p
    // the masked samples as a compact axis (step2_internal.h) when their lists hold at most n entries in all; RG_S2_MASK_COLS=1 keeps the
    // mask columns of rounds 4 - 5
    bool compact = false;
    if (mask_cols && !(getenv("RG_S2_MASK_COLS") && atoi(getenv("RG_S2_MASK_COLS")) != 0)) {
      int64_t listed = 0;
      for (size_t e = 0; e < (size_t)P * n; ++e) listed += ctx->hM[e] ? 0 : 1;
      compact = listed <= n;
    }
    ctx->compact = compact;
    const bool wide = mask_cols && !compact;      // the x_c mask_p / mask_p columns are part of the planes
    const int cv1 = !wide ? C + P : C + P + C * P;
    const int cm0 = !wide ? cv1 : (cv1 + 15) / 16 * 16, cvt = !wide ? cv1 : cm0 + P;
    if (cvt > 4096) return fail(ctx, RG_S2_ERR_ARG, "covariates x phenotypes with differing missing values > 4096 columns");
    const int ngrp = (cvt + 15) / 16;
    ctx->Np = (n + 128 * RG_MAX_SEG - 1) / (128 * RG_MAX_SEG) * (128 * RG_MAX_SEG);   // 32 pieces of a multiple of 128 samples
    for (void** q : {(void**)&ctx->dV, (void**)&ctx->dvd, (void**)&ctx->dvsc, (void**)&ctx->dYtX, (void**)&ctx->dQ, (void**)&ctx->dMsum})
      if (*q) { S2_HIP(hipFree(*q)); *q = nullptr; }
    ctx->Cvt = cvt; ctx->cm0 = cm0;
    S2_HIP(hipMalloc((void**)&ctx->dV, sizeof(double) * ngrp * 16 * ctx->Np));
    S2_HIP(hipMalloc((void**)&ctx->dvd, (size_t)ngrp * 16 * 8 * ctx->Np));
    S2_HIP(hipMalloc((void**)&ctx->dvsc, sizeof(double) * ngrp * 16));
    S2_HIP(hipMalloc((void**)&ctx->dYtX, sizeof(double) * P * C));
    S2_HIP(hipMalloc((void**)&ctx->dQ, sizeof(double) * P * C * C));
    S2_HIP(hipMalloc((void**)&ctx->dMsum, sizeof(double) * P));
    S2_HIP(hipMemsetAsync(ctx->dV, 0, sizeof(double) * ngrp * 16 * ctx->Np, ctx->st));
    S2_HIP(hipMemcpy2DAsync(ctx->dV, ctx->Np * sizeof(double), ctx->dX, n * sizeof(double), n * sizeof(double), C, hipMemcpyDeviceToDevice, ctx->st));
    if (mask_cols) {
      if (wide) hipLaunchKernelGGL(k_s2_mask_cols, dim3((unsigned)((n + 255) / 256), P), dim3(256), 0, ctx->st, ctx->dX, ctx->dM, n, ctx->Np, C, P, cm0, ctx->dV);
      // Q_p = X^T diag(mask_p) X = I - sum over the samples masked for p of x x^T (X is orthonormal on the analysed samples)
      std::vector<double> Q((size_t)P * C * C, 0.0), msum(P, 0.0), xi(C);
      for (int p = 0; p < P; ++p) {
        double* q = Q.data() + (size_t)p * C * C;
        for (int c = 0; c < C; ++c) q[c * C + c] = 1.0;
        const uint8_t* m = ctx->hM.data() + (size_t)p * n;
        int64_t kept = 0;
        for (int64_t i = 0; i < n; ++i) {
          if (m[i]) { ++kept; continue; }
          for (int c = 0; c < C; ++c) xi[c] = ctx->hX[(size_t)c * n + i];
          for (int c = 0; c < C; ++c)
            for (int d = 0; d < C; ++d) q[c * C + d] -= xi[c] * xi[d];
        }
        msum[p] = (double)kept;
      }
      S2_HIP(hipMemcpyAsync(ctx->dQ, Q.data(), sizeof(double) * Q.size(), hipMemcpyHostToDevice, ctx->st));
      S2_HIP(hipMemcpyAsync(ctx->dMsum, msum.data(), sizeof(double) * P, hipMemcpyHostToDevice, ctx->st));
      S2_HIP(hipStreamSynchronize(ctx->st));      // Q / msum are locals
      if (compact) {
        // compact axis: phenotype p's masked samples at positions c_off[p] ..., padded with -1 to a multiple of 128 * c_spt (c_spt equal
        // segments per phenotype, as many as RG_MAX_SEG allows for the phenotypes of one launch)
        ctx->c_chunk = std::min(P, RG_MAX_SEG);
        ctx->c_spt = RG_MAX_SEG / ctx->c_chunk;
        const int64_t unit = 128 * (int64_t)ctx->c_spt;
        const int K = (int)((ctx->Np + S2_CS - 1) / S2_CS);
        std::vector<int32_t> cl, cmeta((size_t)K * P * 2, 0);
        ctx->c_off.assign(P + 1, 0);
        for (int p = 0; p < P; ++p) {
          const uint8_t* m = ctx->hM.data() + (size_t)p * n;
          for (int k = 0; k < K; ++k) {      // chunk-aware: the entries of sample chunk k, padded to whole 32-bit words of the compact row
            const size_t lo = cl.size();
            for (int64_t i = (int64_t)k * S2_CS; i < std::min<int64_t>(n, (int64_t)(k + 1) * S2_CS); ++i) if (!m[i]) cl.push_back((int32_t)i);
            cl.resize((cl.size() + 15) / 16 * 16, -1);
            if (k == K - 1) {                // the phenotype's range: a multiple of 128 * c_spt positions, at least one unit
              const int64_t len = (int64_t)cl.size() - ctx->c_off[p];
              cl.resize((size_t)(ctx->c_off[p] + (std::max<int64_t>(len, 1) + unit - 1) / unit * unit), -1);
            }
            cmeta[((size_t)k * P + p) * 2] = (int32_t)(lo / 16);
            cmeta[((size_t)k * P + p) * 2 + 1] = (int32_t)((cl.size() - lo) / 16);
          }
          ctx->c_off[p + 1] = (int64_t)cl.size();
        }
        ctx->c_K = K;
        const int64_t Ncp = ctx->Ncp = (int64_t)cl.size();
        const int ngc = (C + 15) / 16;      // (the column of ones and the squares come from call counts: k_s2_count_traits)
        std::vector<double> vc((size_t)ngc * 16 * Ncp, 0.0);
        for (int64_t e = 0; e < Ncp; ++e) {
          if (cl[e] < 0) continue;
          for (int c = 0; c < C; ++c) vc[(size_t)c * Ncp + e] = ctx->hX[(size_t)c * n + cl[e]];
        }
        std::vector<int32_t> cw0(P + 1);
        for (int p = 0; p <= P; ++p) cw0[p] = (int32_t)(ctx->c_off[p] / 16);
        for (void** q : {(void**)&ctx->d_clist, (void**)&ctx->d_cmeta, (void**)&ctx->d_cw0, (void**)&ctx->dVc, (void**)&ctx->dvdc, (void**)&ctx->dvscc})
          if (*q) { S2_HIP(hipFree(*q)); *q = nullptr; }
        S2_HIP(hipMalloc((void**)&ctx->d_clist, sizeof(int32_t) * Ncp));
        S2_HIP(hipMalloc((void**)&ctx->dVc, sizeof(double) * vc.size()));
        S2_HIP(hipMalloc((void**)&ctx->dvdc, (size_t)ngc * 16 * 8 * Ncp));
        S2_HIP(hipMalloc((void**)&ctx->dvscc, sizeof(double) * ngc * 16));
        S2_HIP(hipMemcpy(ctx->d_clist, cl.data(), sizeof(int32_t) * Ncp, hipMemcpyHostToDevice));
        S2_HIP(hipMalloc((void**)&ctx->d_cw0, sizeof(int32_t) * (P + 1)));
        S2_HIP(hipMemcpy(ctx->d_cw0, cw0.data(), sizeof(int32_t) * (P + 1), hipMemcpyHostToDevice));
        S2_HIP(hipMalloc((void**)&ctx->d_cmeta, sizeof(int32_t) * cmeta.size()));
        S2_HIP(hipMemcpy(ctx->d_cmeta, cmeta.data(), sizeof(int32_t) * cmeta.size(), hipMemcpyHostToDevice));
        S2_HIP(hipMemcpy(ctx->dVc, vc.data(), sizeof(double) * vc.size(), hipMemcpyHostToDevice));
        rg_launch_v_split(ctx->st, ctx->dVc, Ncp, ngc * 16, ctx->dvdc, ctx->dvscc);
      }
    }
    rg_launch_v_split(ctx->st, ctx->dV, ctx->Np, C, ctx->dvd, ctx->dvsc);
    if (cvt > C + P)
      rg_launch_v_split(ctx->st, ctx->dV + (int64_t)(C + P) * ctx->Np, ctx->Np, ngrp * 16 - (C + P), ctx->dvd + (size_t)(C + P) * 8 * ctx->Np, ctx->dvsc + C + P);
    S2_HIP(hipGetLastError());
    ctx->static_ready = true;
    ctx->planes_mask_cols = mask_cols;
    ctx->res_ready = false;
  }
  const int64_t Np = ctx->Np;
  if (!ctx->res_ready) {      // once per rg_s2_set_null: the planes of the res columns and res^T X
    S2_HIP(hipMemcpy2DAsync(ctx->dV + (int64_t)C * Np, Np * sizeof(double), ctx->dY, n * sizeof(double), n * sizeof(double), P, hipMemcpyDeviceToDevice, ctx->st));
    rg_launch_v_split(ctx->st, ctx->dV + (int64_t)C * Np, Np, P, ctx->dvd + (size_t)C * 8 * Np, ctx->dvsc + C);
    hipLaunchKernelGGL(k_s2_ytx, dim3(P * C), dim3(256), 0, ctx->st, ctx->dX, ctx->dY, n, C, ctx->dYtX);
    S2_HIP(hipGetLastError());
    ctx->res_ready = true;
  }
  return RG_S2_OK;
}

// Per phenotype the analysed samples masked for it (index lists on the device), rebuilt when the masks changed.
int ensure_lists(rg_s2_ctx* ctx) {
  if (ctx->lists_ready) return RG_S2_OK;
  const int64_t n = ctx->n;
  const int P = ctx->P;
  std::vector<int32_t> lst;
  std::vector<int64_t> off(P + 1, 0);
  for (int p = 0; p < P; ++p) {
    const uint8_t* m = ctx->hM.data() + (size_t)p * n;
    for (int64_t i = 0; i < n; ++i) if (!m[i]) lst.push_back((int32_t)i);
    off[p + 1] = (int64_t)lst.size();
  }
  if (ctx->d_mlist) { S2_HIP(hipFree(ctx->d_mlist)); ctx->d_mlist = nullptr; }
  if (!ctx->d_moff) S2_HIP(hipMalloc((void**)&ctx->d_moff, sizeof(int64_t) * (P + 1)));
  S2_HIP(hipMalloc((void**)&ctx->d_mlist, sizeof(int32_t) * std::max<size_t>(1, lst.size())));
  S2_HIP(hipMemcpy(ctx->d_mlist, lst.data(), sizeof(int32_t) * lst.size(), hipMemcpyHostToDevice));
  S2_HIP(hipMemcpy(ctx->d_moff, off.data(), sizeof(int64_t) * (P + 1), hipMemcpyHostToDevice));
  {  // the listed samples' covariate rows, compact and C-contiguous
    const int C = ctx->C;
    std::vector<double> xl(std::max<size_t>(1, lst.size() * C));
    for (size_t e = 0; e < lst.size(); ++e)
      for (int c = 0; c < C; ++c) xl[e * C + c] = ctx->hX[(size_t)c * n + lst[e]];
    if (ctx->d_xl) { S2_HIP(hipFree(ctx->d_xl)); ctx->d_xl = nullptr; }
    S2_HIP(hipMalloc((void**)&ctx->d_xl, sizeof(double) * xl.size()));
    S2_HIP(hipMemcpy(ctx->d_xl, xl.data(), sizeof(double) * xl.size(), hipMemcpyHostToDevice));
    // X^T X over each phenotype's list: sum (g - x.beta)^2 = sum g^2 - 2 beta.(X^T g) + beta^T (X^T X) beta needs no second pass over the entries
    std::vector<double> xq((size_t)P * C * C, 0.0);
    for (int p = 0; p < P; ++p)
      for (int64_t e = off[p]; e < off[p + 1]; ++e)
        for (int a = 0; a < C; ++a)
          for (int b = 0; b < C; ++b) xq[((size_t)p * C + a) * C + b] += xl[(size_t)e * C + a] * xl[(size_t)e * C + b];
    if (ctx->d_xq) { S2_HIP(hipFree(ctx->d_xq)); ctx->d_xq = nullptr; }
    S2_HIP(hipMalloc((void**)&ctx->d_xq, sizeof(double) * xq.size()));
    S2_HIP(hipMemcpy(ctx->d_xq, xq.data(), sizeof(double) * xq.size(), hipMemcpyHostToDevice));
  }
  ctx->lists_ready = true;
  return RG_S2_OK;
}

int ensure_p(rg_s2_ctx* ctx, int slot, size_t bytes) { return ensure_in(ctx, ctx->pbuf, ctx->pcap, slot, bytes); }
}  // namespace

extern "C" {

int rg_s2_create(rg_s2_ctx** out, int device, int64_t n, int32_t n_cov, int32_t n_pheno) {
  if (!out) return RG_S2_ERR_ARG;
  rg_s2_ctx* ctx = new rg_s2_ctx();
  *out = ctx;
  if (n <= 0 || n >= (1 << 26) || n_cov < 1 || n_cov > RG_S2_MAX_COV || n_pheno < 1 || n_pheno > RG_S2_MAX_PHENO || n <= n_cov)
    return fail(ctx, RG_S2_ERR_ARG, "rg_s2_create: need n > n_cov, 1 <= n_cov <= 64, 1 <= n_pheno <= 64");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(ctx, RG_S2_ERR_HIP, "rg_s2_create: no HIP device (this library has no CPU path)");
  if (device < 0 || device >= ndev) return fail(ctx, RG_S2_ERR_ARG, "rg_s2_create: device index out of range");
  ctx->dev = device; ctx->n = n; ctx->C = n_cov; ctx->P = n_pheno;
  ctx->vpb = DEFAULT_VPB; ctx->ept = DEFAULT_EPT;
  if (const char* t = getenv("RG_S2_TILE")) {
    int a = 0, b = 0;
    if (sscanf(t, "%dx%d", &a, &b) != 2 || !((a == 4 && b == 4) || (a == 4 && b == 8) || (a == 8 && b == 4) || (a == 8 && b == 8) || (a == 16 && b == 4)))
      return fail(ctx, RG_S2_ERR_ARG, "RG_S2_TILE must be one of 4x4, 4x8, 8x4, 8x8, 16x4");
    ctx->vpb = a; ctx->ept = b;
  }
  S2_HIP(hipSetDevice(device));
  S2_HIP(hipStreamCreateWithFlags(&ctx->st, hipStreamNonBlocking));
  S2_HIP(hipEventCreate(&ctx->e0));
  S2_HIP(hipEventCreate(&ctx->e1));
  S2_HIP(hipMalloc((void**)&ctx->dX, sizeof(double) * n * n_cov));
  S2_HIP(hipMalloc((void**)&ctx->dY, sizeof(double) * n * n_pheno));
  S2_HIP(hipMalloc((void**)&ctx->dM, (size_t)n * n_pheno));
  S2_HIP(hipMalloc((void**)&ctx->dscf, sizeof(double) * n_pheno));
  return RG_S2_OK;
}

void rg_s2_destroy(rg_s2_ctx* ctx) {
  if (!ctx) return;
  if (ctx->st) {
    (void)hipSetDevice(ctx->dev);
    (void)hipStreamSynchronize(ctx->st);
    for (void* p : ctx->buf) if (p) (void)hipFree(p);
    for (void* p : ctx->pbuf) if (p) (void)hipFree(p);
    if (ctx->dV) (void)hipFree(ctx->dV);
    if (ctx->dvd) (void)hipFree(ctx->dvd);
    if (ctx->dvsc) (void)hipFree(ctx->dvsc);
    if (ctx->dYtX) (void)hipFree(ctx->dYtX);
    if (ctx->dQ) (void)hipFree(ctx->dQ);
    if (ctx->dMsum) (void)hipFree(ctx->dMsum);
    if (ctx->gV) (void)hipFree(ctx->gV);
    if (ctx->gvd) (void)hipFree(ctx->gvd);
    if (ctx->gvsc) (void)hipFree(ctx->gvsc);
    if (ctx->d_clist) (void)hipFree(ctx->d_clist);
    if (ctx->d_cmeta) (void)hipFree(ctx->d_cmeta);
    if (ctx->d_cw0) (void)hipFree(ctx->d_cw0);
    if (ctx->dVc) (void)hipFree(ctx->dVc);
    if (ctx->dvdc) (void)hipFree(ctx->dvdc);
    if (ctx->dvscc) (void)hipFree(ctx->dvscc);
    if (ctx->d_mlist) (void)hipFree(ctx->d_mlist);
    if (ctx->d_moff) (void)hipFree(ctx->d_moff);
    if (ctx->d_xl) (void)hipFree(ctx->d_xl);
    if (ctx->d_xq) (void)hipFree(ctx->d_xq);
    if (ctx->dX) (void)hipFree(ctx->dX);
    if (ctx->dY) (void)hipFree(ctx->dY);
    if (ctx->dM) (void)hipFree(ctx->dM);
    if (ctx->dscf) (void)hipFree(ctx->dscf);
    rg_s2_bt_free(ctx);
    if (ctx->e0) (void)hipEventDestroy(ctx->e0);
    if (ctx->e1) (void)hipEventDestroy(ctx->e1);
    (void)hipStreamDestroy(ctx->st);
  }
  delete ctx;
}

const char* rg_s2_last_error(const rg_s2_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int rg_s2_set_null(rg_s2_ctx* ctx, const double* X, const double* yres, const uint8_t* mask, const double* scf_sv) {
  if (!ctx || !ctx->st) return fail(ctx, RG_S2_ERR_ARG, "rg_s2_set_null: context was not created");
  if (!X || !yres || !mask || !scf_sv) return fail(ctx, RG_S2_ERR_ARG, "rg_s2_set_null: null argument");
  S2_HIP(hipSetDevice(ctx->dev));
  S2_HIP(hipMemcpyAsync(ctx->dX, X, sizeof(double) * ctx->n * ctx->C, hipMemcpyHostToDevice, ctx->st));
  S2_HIP(hipMemcpyAsync(ctx->dY, yres, sizeof(double) * ctx->n * ctx->P, hipMemcpyHostToDevice, ctx->st));
  S2_HIP(hipMemcpyAsync(ctx->dM, mask, (size_t)ctx->n * ctx->P, hipMemcpyHostToDevice, ctx->st));
  S2_HIP(hipMemcpyAsync(ctx->dscf, scf_sv, sizeof(double) * ctx->P, hipMemcpyHostToDevice, ctx->st));
  S2_HIP(hipStreamSynchronize(ctx->st));
  ctx->have_null = true;
  ctx->res_ready = false;
  const size_t nx = (size_t)ctx->n * ctx->C, nm = (size_t)ctx->n * ctx->P;
  if (ctx->hX.size() != nx || ctx->hM.size() != nm || memcmp(ctx->hX.data(), X, nx * sizeof(double)) != 0 || memcmp(ctx->hM.data(), mask, nm) != 0) {
    ctx->hX.assign(X, X + nx);
    ctx->hM.assign(mask, mask + nm);
    ctx->static_ready = false;
    ctx->lists_ready = false;
    ctx->complete = true;
    for (size_t i = 0; i < nm; ++i)
      if (!mask[i]) { ctx->complete = false; break; }
  }
  return RG_S2_OK;
}

int rg_s2_set_sparse_rule(rg_s2_ctx* ctx, int64_t n_samples, double prop_zero_thr, int32_t zero_count_rule) {
  if (!ctx || !ctx->st) return fail(ctx, RG_S2_ERR_ARG, "rg_s2_set_sparse_rule: context was not created");
  if (n_samples < ctx->n || !(prop_zero_thr >= 0.0 && prop_zero_thr <= 1.0))
    return fail(ctx, RG_S2_ERR_ARG, "rg_s2_set_sparse_rule: need n_samples >= n and 0 <= prop_zero_thr <= 1");
  ctx->rule_n = n_samples; ctx->rule_thr = prop_zero_thr; ctx->rule_zero_count = zero_count_rule ? 1 : 0;
  return RG_S2_OK;
}

int rg_s2_qt_block(rg_s2_ctx* ctx, const double* G, int64_t ldg, int32_t bs, int32_t g_on_device, double numtol,
                   const rg_s2_qt_out* out) {
  if (!ctx || !ctx->st) return fail(ctx, RG_S2_ERR_ARG, "rg_s2_qt_block: context was not created");
  if (!ctx->have_null) return fail(ctx, RG_S2_ERR_ARG, "rg_s2_qt_block: rg_s2_set_null has not been called");
  if (!G || !out || bs < 1 || ldg < ctx->n) return fail(ctx, RG_S2_ERR_ARG, "rg_s2_qt_block: bad arguments (need bs >= 1, ldg >= n)");
  const int64_t n = ctx->n;
  const int C = ctx->C, P = ctx->P, Q1 = 2 + 2 * C, Q2 = 1 + 2 * P;
  int vpb = ctx->vpb, ept = ctx->ept;
  size_t lds1 = sizeof(double) * 4 * vpb * Q1, lds2 = sizeof(double) * ((size_t)vpb * C + vpb + 4 * (size_t)vpb * (Q2 + 1));
  if (lds1 > 60 * 1024 || lds2 > 60 * 1024) {   // many covariates / phenotypes: the documented default tile always fits (<= 17 KB)
    vpb = DEFAULT_VPB; ept = DEFAULT_EPT;
    lds1 = sizeof(double) * 4 * vpb * Q1; lds2 = sizeof(double) * ((size_t)vpb * C + vpb + 4 * (size_t)vpb * (Q2 + 1));
  }
  const int nchunk = (int)((n + 256 * ept - 1) / (256 * ept));
  const unsigned gv = (unsigned)((bs + vpb - 1) / vpb);
  S2_HIP(hipSetDevice(ctx->dev));
  enum { B_G, B_PART, B_BETA, B_VAR, B_NOBS, B_STAT, B_CORR };
  const size_t part_elems = (size_t)bs * nchunk * (size_t)(Q1 > Q2 ? Q1 : Q2);
  int rc;
  if ((rc = ensure(ctx, B_PART, part_elems * sizeof(double)))) return rc;
  if ((rc = ensure(ctx, B_BETA, (size_t)bs * RG_S2_MAX_COV * sizeof(double)))) return rc;
  if ((rc = ensure(ctx, B_VAR, (size_t)bs * 2 * sizeof(double)))) return rc;       // mu | scale_fac
  if ((rc = ensure(ctx, B_NOBS, (size_t)bs * 3 * sizeof(int32_t)))) return rc;     // nobs | ignored | nnz
  const bool masked = !ctx->complete;
  if (masked && (rc = ensure(ctx, B_CORR, (size_t)bs * P * (C + 1) * sizeof(double)))) return rc;
  if (masked && (rc = ensure_lists(ctx))) return rc;
  if ((rc = ensure(ctx, B_STAT, (size_t)bs * P * 2 * sizeof(double)))) return rc;  // stats | bhat
  const double* dG = G;
  int64_t ld = ldg;
  if (!g_on_device) {
    if ((rc = ensure(ctx, B_G, (size_t)bs * n * sizeof(double)))) return rc;
    S2_HIP(hipMemcpy2DAsync(ctx->buf[B_G], n * sizeof(double), G, ldg * sizeof(double), n * sizeof(double), bs,
                            hipMemcpyHostToDevice, ctx->st));
    dG = (const double*)ctx->buf[B_G];
    ld = n;
  }
  double* part = (double*)ctx->buf[B_PART];
  double* beta = (double*)ctx->buf[B_BETA];
  double* mu = (double*)ctx->buf[B_VAR];
  double* sf = mu + bs;
  int32_t* nobs = (int32_t*)ctx->buf[B_NOBS];
  int32_t* ign = nobs + bs;
  int32_t* nnz = ign + bs;
  double* stats = (double*)ctx->buf[B_STAT];
  double* bhat = stats + (size_t)bs * P;
  double* corr = masked ? (double*)ctx->buf[B_CORR] : nullptr;
  S2_HIP(hipEventRecord(ctx->e0, ctx->st));
#define S2_TILE(V, E, KERNEL_CALL) if (vpb == V && ept == E) { constexpr int TV = V, TE = E; KERNEL_CALL; }
#define S2_PROJ hipLaunchKernelGGL((k_s2_proj<TV, TE>), dim3(gv, nchunk), dim3(256), lds1, ctx->st, dG, ld, bs, n, ctx->dX, C, part)
#define S2_SCORE                                                                                                              \
  hipLaunchKernelGGL((k_s2_score<TV, TE>), dim3(gv, nchunk), dim3(256), lds2, ctx->st, dG, ld, bs, n, ctx->dX, C, ctx->dY, ctx->dM, \
                     P, beta, mu, part)
  S2_TILE(4, 4, S2_PROJ) S2_TILE(4, 8, S2_PROJ) S2_TILE(8, 4, S2_PROJ) S2_TILE(8, 8, S2_PROJ) S2_TILE(16, 4, S2_PROJ)
  hipLaunchKernelGGL(k_s2_beta, dim3((bs * C + 255) / 256), dim3(256), 0, ctx->st, part, nchunk, C, bs, beta, mu, nobs, nnz);
  if (masked)
    hipLaunchKernelGGL(k_s2_masked_corr, dim3(bs, P), dim3(256), 0, ctx->st, dG, ld, n, ctx->dX, C, ctx->d_mlist, ctx->d_moff, mu, P, corr);
  S2_TILE(4, 4, S2_SCORE) S2_TILE(4, 8, S2_SCORE) S2_TILE(8, 4, S2_SCORE) S2_TILE(8, 8, S2_SCORE) S2_TILE(16, 4, S2_SCORE)
#undef S2_TILE
#undef S2_PROJ
#undef S2_SCORE
  hipLaunchKernelGGL(k_s2_final, dim3((bs * P + 255) / 256), dim3(256), 0, ctx->st, part, nchunk, P, C, bs, n, numtol,
                     (double)(ctx->rule_n > 0 ? ctx->rule_n : n) * (1.0 - ctx->rule_thr),
                     ctx->rule_zero_count ? (double)(ctx->rule_n > 0 ? ctx->rule_n : n) * ctx->rule_thr : -1.0, ctx->dscf, nobs, nnz, mu, beta, corr, stats,
                     bhat, sf, ign);
  S2_HIP(hipEventRecord(ctx->e1, ctx->st));
  S2_HIP(hipGetLastError());
  if (out->stats) S2_HIP(hipMemcpyAsync(out->stats, stats, sizeof(double) * bs * P, hipMemcpyDeviceToHost, ctx->st));
  if (out->bhat) S2_HIP(hipMemcpyAsync(out->bhat, bhat, sizeof(double) * bs * P, hipMemcpyDeviceToHost, ctx->st));
  if (out->scale_fac) S2_HIP(hipMemcpyAsync(out->scale_fac, sf, sizeof(double) * bs, hipMemcpyDeviceToHost, ctx->st));
  if (out->mean) S2_HIP(hipMemcpyAsync(out->mean, mu, sizeof(double) * bs, hipMemcpyDeviceToHost, ctx->st));
  if (out->n_obs) S2_HIP(hipMemcpyAsync(out->n_obs, nobs, sizeof(int32_t) * bs, hipMemcpyDeviceToHost, ctx->st));
  if (out->ignored) S2_HIP(hipMemcpyAsync(out->ignored, ign, sizeof(int32_t) * bs, hipMemcpyDeviceToHost, ctx->st));
  S2_HIP(hipStreamSynchronize(ctx->st));
  float ms = 0.f;
  S2_HIP(hipEventElapsedTime(&ms, ctx->e0, ctx->e1));
  ctx->last_ms = ms;
  return RG_S2_OK;
}

int rg_s2_qt_block_packed(rg_s2_ctx* ctx, const uint8_t* rows, int64_t ld, int32_t bs, int32_t rows_on_device, int32_t flip, double numtol,
                          const rg_s2_qt_out* out) {
  if (!ctx || !ctx->st) return fail(ctx, RG_S2_ERR_ARG, "rg_s2_qt_block_packed: context was not created");
  if (!ctx->have_null) return fail(ctx, RG_S2_ERR_ARG, "rg_s2_qt_block_packed: rg_s2_set_null has not been called");
  const int64_t n = ctx->n, nbytes = (n + 3) / 4;
  if (!rows || !out || bs < 1 || ld < nbytes) return fail(ctx, RG_S2_ERR_ARG, "rg_s2_qt_block_packed: bad arguments (need bs >= 1, ld >= ceil(n / 4))");
  const int C = ctx->C, P = ctx->P;
  S2_HIP(hipSetDevice(ctx->dev));
  int rc;
  if ((rc = ensure_planes(ctx, true))) return rc;
  const int Cvt = ctx->Cvt, cm0 = ctx->cm0, ngrp = (Cvt + 15) / 16, masked = ctx->complete ? 0 : (ctx->compact ? 3 : 1);
  const int64_t Np = ctx->Np, ldp = Np / 4;
  const int n128 = (int)((bs + 127) / 128 * 128);
  const int gm0 = cm0 / 16, ngrpB = masked == 1 ? ngrp - gm0 : 0, CvB = ngrpB * 16;      // column groups of the g0^2 contraction (the mask columns)
  // masked == 3: the compact axis (step2_internal.h): one contraction launch per c_chunk phenotypes, c_spt segments each
  const int64_t Ncp = masked == 3 ? ctx->Ncp : 0, ldc = Ncp / 4;
  const int Cc = C + 1, ngc = (C + 15) / 16, nsegc = masked == 3 ? ctx->c_chunk * ctx->c_spt : 0;
  // enough workgroups to fill the 256 CUs (tiles x segments x column groups >= 768), as few segments as that allows: every segment
  // costs a 64 KB tile of partial sums per workgroup
  SegLayout seg, segB;
  const int nseg = pick_segments(Np, n128 / 128, rg_xy_i8_launch_groups(Cvt), seg), nsegB = pick_segments(Np, n128 / 128, std::max(1, ngrpB), segB);
  enum { Q_PK, Q_CNT, Q_S, Q_A, Q_VAR, Q_STAT, Q_PKC, Q_L };
  const size_t s_grp = (size_t)2 * nseg * n128 * 128, s_grpB = (size_t)2 * nsegB * n128 * 128, s_grpC = (size_t)2 * nsegc * n128 * 128;
  if ((rc = ensure_p(ctx, Q_PK, (size_t)bs * ldp))) return rc;
  if ((rc = ensure_p(ctx, Q_CNT, ((size_t)bs * 4 + 8) * sizeof(int32_t)))) return rc;            // counts | total_miss | bs | 0
  if ((rc = ensure_p(ctx, Q_S, ((size_t)ngrp * s_grp + (size_t)ngrpB * s_grpB + (size_t)(masked == 3 ? ngc : 0) * s_grpC) * sizeof(int32_t)))) return rc;
  if (masked == 3) {
    if ((rc = ensure_p(ctx, Q_PKC, (size_t)bs * ldc))) return rc;
    if ((rc = ensure_p(ctx, Q_L, (size_t)bs * P * ((2 * Cc + 1) * sizeof(double) + 4 * sizeof(int32_t))))) return rc;       // Lc [bs][2][P][Cc] | Lq [bs][P] | call counts [bs][P][4]
  }
  if ((rc = ensure_p(ctx, Q_A, (size_t)bs * 2 * (Cvt + CvB) * sizeof(double)))) return rc;
  if ((rc = ensure_p(ctx, Q_VAR, (size_t)bs * (6 * sizeof(double) + 2 * sizeof(int32_t))))) return rc;   // scale_fac | mean | vstat[4] | nobs | ignored
  if ((rc = ensure_p(ctx, Q_STAT, (size_t)bs * P * (3 * sizeof(double) + sizeof(int32_t))))) return rc;  // stats | bhat | total_p | nobs_p
  uint8_t* pk = (uint8_t*)ctx->pbuf[Q_PK];
  int32_t* cnt = (int32_t*)ctx->pbuf[Q_CNT];
  int32_t* total_miss = cnt + (size_t)bs * 4;
  int32_t* d_bs = total_miss + 1;
  int32_t* d_zero = total_miss + 2;
  int32_t* S = (int32_t*)ctx->pbuf[Q_S];
  double* A = (double*)ctx->pbuf[Q_A];
  double* Sq = A + (size_t)bs * 2 * Cvt;
  double* sf = (double*)ctx->pbuf[Q_VAR];
  double* mu = sf + bs;
  double* vstat = mu + bs;
  int32_t* nobs = (int32_t*)(vstat + (size_t)bs * 4);
  int32_t* ign = nobs + bs;
  double* stats = (double*)ctx->pbuf[Q_STAT];
  double* bhat = stats + (size_t)bs * P;
  double* total_p = bhat + (size_t)bs * P;
  int32_t* nobs_p = (int32_t*)(total_p + (size_t)bs * P);
  ctx->hdr[0] = 0; ctx->hdr[1] = bs; ctx->hdr[2] = 0;
  S2_HIP(hipMemcpyAsync(total_miss, ctx->hdr, sizeof(ctx->hdr), hipMemcpyHostToDevice, ctx->st));
  S2_HIP(hipMemcpy2DAsync(pk, ldp, rows, ld, nbytes, bs, rows_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, ctx->st));
  S2_HIP(hipEventRecord(ctx->e0, ctx->st));
  hipLaunchKernelGGL(k_s2_rows, dim3(bs), dim3(256), 0, ctx->st, pk, ldp, n, flip ? 1 : 0, cnt, total_miss, vstat);
  // a block with a missing call contracts both sets (allele count, missing indicator) in one pass over the rows; the count comes back
  // from k_s2_rows first (4 bytes, one stream synchronisation: tens of microseconds against the milliseconds of the contraction)
  int32_t h_miss = 0;
  S2_HIP(hipMemcpyAsync(&h_miss, total_miss, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->st));
  S2_HIP(hipStreamSynchronize(ctx->st));
  if (h_miss > 0) rg_launch_xy_i8_both(ctx->st, pk, ldp, d_bs, Cvt, n128, seg, ctx->dvd, Np, RG_XY_LUT_DOSAGE, S);
  else rg_launch_xy_i8_sums(ctx->st, pk, ldp, d_bs, total_miss, Cvt, n128, seg, ctx->dvd, Np, RG_XY_LUT_DOSAGE, S);
  hipLaunchKernelGGL(k_s2_combine, dim3((bs * Cvt + 255) / 256), dim3(256), 0, ctx->st, (const int32_t*)S, ctx->dvsc, total_miss, bs, n128, nseg, Cvt, A);
  if (masked == 1) {   // sum mask_p g0^2: the square LUT against the mask columns only (the missing-indicator set is skipped: d_zero)
    rg_launch_xy_i8_sums(ctx->st, pk, ldp, d_bs, d_zero, Cvt - gm0 * 16, n128, segB, ctx->dvd + (size_t)gm0 * 16 * 8 * Np, Np, RG_XY_LUT_SQUARE, S + (size_t)ngrp * s_grp);
    hipLaunchKernelGGL(k_s2_combine, dim3((bs * CvB + 255) / 256), dim3(256), 0, ctx->st, (const int32_t*)(S + (size_t)ngrp * s_grp), ctx->dvsc + gm0 * 16,
                       (const int32_t*)nullptr, bs, n128, nsegB, CvB, Sq);
  }
  double *Lc = nullptr, *Lq = nullptr;
  if (masked == 3) {
    // the rows over the compact axis, then per launch of c_chunk phenotypes: allele counts (and missing indicators) against x_0 .. x_{C-1}, 1
    // and squared allele counts against the column of ones, summed over each phenotype's own segments
    uint8_t* pkc = (uint8_t*)ctx->pbuf[Q_PKC];
    Lc = (double*)ctx->pbuf[Q_L];
    Lq = Lc + (size_t)bs * 2 * P * Cc;
    int32_t* Sc = S + (size_t)ngrp * s_grp;
    int32_t* tcnt = (int32_t*)(Lq + (size_t)bs * P);
    const size_t lds_c = (size_t)(S2_CS / 4 + 1) * 8;      // (per device, and cheap: set for every block)
    S2_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_s2_compact_rows), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_c));
    hipLaunchKernelGGL(k_s2_compact_rows, dim3(ctx->c_K, (bs + S2_CR - 1) / S2_CR), dim3(S2_CT), lds_c, ctx->st, (const uint8_t*)pk, ldp, (const int32_t*)ctx->d_clist,
                       (const int32_t*)ctx->d_cmeta, P, bs, pkc, ldc);
    hipLaunchKernelGGL(k_s2_count_traits, dim3(bs, P), dim3(256), 0, ctx->st, (const uint8_t*)pkc, ldc, (const int32_t*)ctx->d_cw0, tcnt);
    for (int t0 = 0; t0 < P; t0 += ctx->c_chunk) {
      const int nt = std::min(ctx->c_chunk, P - t0), spt = ctx->c_spt;
      SegLayout sc;
      memset(&sc, 0, sizeof(sc));
      sc.nseg = nt * spt;
      for (int tt = 0; tt < nt; ++tt) {
        const int64_t o = ctx->c_off[t0 + tt], len = (ctx->c_off[t0 + tt + 1] - o) / spt;
        for (int sg = 0; sg < spt; ++sg) {
          const int f = tt * spt + sg;
          sc.pos_start[f] = o + sg * len; sc.file_start[f] = sc.pos_start[f]; sc.len[f] = len; sc.plen[f] = len;
        }
      }
      // (the S layout of a launch is [group][set][sc.nseg][n128][128]: a last launch with fewer phenotypes uses the front of the buffer)
      if (h_miss > 0) rg_launch_xy_i8_both(ctx->st, pkc, ldc, d_bs, C, n128, sc, ctx->dvdc, Ncp, RG_XY_LUT_DOSAGE, Sc);
      else rg_launch_xy_i8_sums(ctx->st, pkc, ldc, d_bs, total_miss, C, n128, sc, ctx->dvdc, Ncp, RG_XY_LUT_DOSAGE, Sc);
      const int64_t nthr = (int64_t)bs * nt * Cc;
      hipLaunchKernelGGL(k_s2_combine_traits, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, ctx->st, (const int32_t*)Sc, (const int32_t*)tcnt,
                         (const double*)ctx->dvscc, (const int32_t*)total_miss, bs, n128, sc.nseg, spt, t0, nt, P, C, Lc, Lq);
    }
  }
  PackedFinal fa;
  fa.A = A; fa.Sq = masked == 1 ? Sq : nullptr; fa.vstat = vstat; fa.corr = nullptr; fa.inv_scale = 1.0;
  fa.Lc = Lc; fa.Lq = Lq;
  fa.ytx = ctx->dYtX; fa.Q = ctx->dQ; fa.msum = ctx->dMsum; fa.scf_sv = ctx->dscf;
  fa.bs = bs; fa.C = C; fa.P = P; fa.Cvt = Cvt; fa.cm0 = cm0; fa.CvB = CvB; fa.sqoff = cm0 - gm0 * 16; fa.masked = masked;
  fa.n = n; fa.numtol = numtol; fa.nz_max = (double)(ctx->rule_n > 0 ? ctx->rule_n : n) * (1.0 - ctx->rule_thr);
  fa.zeros_min = ctx->rule_zero_count ? (double)(ctx->rule_n > 0 ? ctx->rule_n : n) * ctx->rule_thr : -1.0;
  fa.stats = stats; fa.bhat = bhat; fa.scale_fac = sf; fa.mean = mu; fa.total_p = total_p; fa.nobs = nobs; fa.ignored = ign; fa.nobs_p = nobs_p;
  hipLaunchKernelGGL(k_s2_packed_final, dim3((bs * P + 255) / 256), dim3(256), 0, ctx->st, fa);
  S2_HIP(hipEventRecord(ctx->e1, ctx->st));
  S2_HIP(hipGetLastError());
  if (out->stats) S2_HIP(hipMemcpyAsync(out->stats, stats, sizeof(double) * bs * P, hipMemcpyDeviceToHost, ctx->st));
  if (out->bhat) S2_HIP(hipMemcpyAsync(out->bhat, bhat, sizeof(double) * bs * P, hipMemcpyDeviceToHost, ctx->st));
  if (out->scale_fac) S2_HIP(hipMemcpyAsync(out->scale_fac, sf, sizeof(double) * bs, hipMemcpyDeviceToHost, ctx->st));
  if (out->mean) S2_HIP(hipMemcpyAsync(out->mean, mu, sizeof(double) * bs, hipMemcpyDeviceToHost, ctx->st));
  if (out->n_obs) S2_HIP(hipMemcpyAsync(out->n_obs, nobs, sizeof(int32_t) * bs, hipMemcpyDeviceToHost, ctx->st));
  if (out->ignored) S2_HIP(hipMemcpyAsync(out->ignored, ign, sizeof(int32_t) * bs, hipMemcpyDeviceToHost, ctx->st));
  if (out->total_p) S2_HIP(hipMemcpyAsync(out->total_p, total_p, sizeof(double) * bs * P, hipMemcpyDeviceToHost, ctx->st));
  if (out->n_obs_p) S2_HIP(hipMemcpyAsync(out->n_obs_p, nobs_p, sizeof(int32_t) * bs * P, hipMemcpyDeviceToHost, ctx->st));
  S2_HIP(hipStreamSynchronize(ctx->st));
  float ms = 0.f;
  S2_HIP(hipEventElapsedTime(&ms, ctx->e0, ctx->e1));
  ctx->last_ms = ms;
  return RG_S2_OK;
}

int rg_s2_qt_block_int(rg_s2_ctx* ctx, const uint16_t* G, int64_t ld, int32_t bs, int32_t g_on_device, int32_t scale, double numtol,
                       const rg_s2_qt_out* out) {
  if (!ctx || !ctx->st) return fail(ctx, RG_S2_ERR_ARG, "rg_s2_qt_block_int: context was not created");
  if (!ctx->have_null) return fail(ctx, RG_S2_ERR_ARG, "rg_s2_qt_block_int: rg_s2_set_null has not been called");
  const int64_t n = ctx->n;
  if (!G || !out || bs < 1 || ld < n || scale < 1 || scale > 16384)
    return fail(ctx, RG_S2_ERR_ARG, "rg_s2_qt_block_int: bad arguments (need bs >= 1, ld >= n, 1 <= scale <= 16384)");
  const int C = ctx->C, P = ctx->P, Cv = C + P;
  S2_HIP(hipSetDevice(ctx->dev));
  int rc;
  if ((rc = ensure_planes(ctx, ctx->planes_mask_cols))) return rc;       // [X | res] lead the planes whatever else they hold
  const int masked = ctx->complete ? 0 : 2;
  if (masked && (rc = ensure_lists(ctx))) return rc;
  const int64_t Np = ctx->Np;
  const int D = 2 * scale <= 8127 ? 2 : 3, nset = D + 1;                 // balanced base-128 digits of a dosage <= 2 * scale (two reach 63 + 128 * 63), then the missing indicator
  const int n128 = (int)((bs + 127) / 128 * 128), ngrp = (Cv + 15) / 16;
  SegLayout seg;
  int nseg = pick_segments(Np, n128 / 128, ngrp * nset, seg);
  while (Np / nseg > 262144 && nseg < RG_MAX_SEG) { nseg *= 2; }         // |digit x digit| <= 4096: int32 sums hold 2^19 samples per segment
  if (Np / nseg > 524288) return fail(ctx, RG_S2_ERR_ARG, "rg_s2_qt_block_int: more than 16.7 million samples");
  memset(&seg, 0, sizeof(seg));
  seg.nseg = nseg;
  for (int f = 0; f < nseg; ++f) { seg.pos_start[f] = f * (Np / nseg); seg.file_start[f] = seg.pos_start[f]; seg.len[f] = Np / nseg; seg.plen[f] = Np / nseg; }
  enum { Q_PK, Q_CNT, Q_S, Q_A, Q_VAR, Q_STAT };
  enum { B_G, B_PART, B_BETA, B_VAR, B_NOBS, B_STAT, B_CORR, B_PLANES };
  if ((rc = ensure_p(ctx, Q_CNT, ((size_t)bs * 4 + 8) * sizeof(int32_t)))) return rc;
  if ((rc = ensure_p(ctx, Q_S, (size_t)ngrp * nset * nseg * n128 * 128 * sizeof(int32_t)))) return rc;
  if ((rc = ensure_p(ctx, Q_A, (size_t)bs * 2 * ctx->Cvt * sizeof(double)))) return rc;
  if ((rc = ensure_p(ctx, Q_VAR, (size_t)bs * (6 * sizeof(double) + 2 * sizeof(int32_t))))) return rc;
  if ((rc = ensure_p(ctx, Q_STAT, (size_t)bs * P * (3 * sizeof(double) + sizeof(int32_t))))) return rc;
  if ((rc = ensure(ctx, B_PLANES, (size_t)nset * bs * Np))) return rc;
  const int64_t ld8 = (n + 7) / 8 * 8;       // staged rows start on 16-byte boundaries
  if (!g_on_device && (rc = ensure(ctx, B_G, (size_t)bs * ld8 * sizeof(uint16_t)))) return rc;
  if (masked) {
    if ((rc = ensure(ctx, B_BETA, (size_t)bs * RG_S2_MAX_COV * sizeof(double)))) return rc;
    if ((rc = ensure(ctx, B_CORR, (size_t)bs * P * (C + 2) * sizeof(double)))) return rc;
  }
  const uint16_t* dG = G;
  int64_t ldg = ld;
  if (!g_on_device) {
    if (ld == ld8) S2_HIP(hipMemcpyAsync(ctx->buf[B_G], G, ((size_t)(bs - 1) * ld8 + n) * sizeof(uint16_t), hipMemcpyHostToDevice, ctx->st));     // rows already on the staged pitch: one flat copy
    else S2_HIP(hipMemcpy2DAsync(ctx->buf[B_G], ld8 * sizeof(uint16_t), G, ld * sizeof(uint16_t), n * sizeof(uint16_t), bs, hipMemcpyHostToDevice, ctx->st));
    dG = (const uint16_t*)ctx->buf[B_G];
    ldg = ld8;
  }
  int32_t* cnt = (int32_t*)ctx->pbuf[Q_CNT];
  int32_t* total_miss = cnt + (size_t)bs * 4;
  int32_t* d_bs = total_miss + 1;
  int32_t* S = (int32_t*)ctx->pbuf[Q_S];
  double* A = (double*)ctx->pbuf[Q_A];
  double* sf = (double*)ctx->pbuf[Q_VAR];
  double* mu = sf + bs;
  double* vstat = mu + bs;
  int32_t* nobs = (int32_t*)(vstat + (size_t)bs * 4);
  int32_t* ign = nobs + bs;
  double* stats = (double*)ctx->pbuf[Q_STAT];
  double* bhat = stats + (size_t)bs * P;
  double* total_p = bhat + (size_t)bs * P;
  int32_t* nobs_p = (int32_t*)(total_p + (size_t)bs * P);
  int8_t* planes = (int8_t*)ctx->buf[B_PLANES];
  const double inv_scale = 1.0 / (double)scale;
  ctx->hdr[0] = 0; ctx->hdr[1] = bs; ctx->hdr[2] = 0;
  S2_HIP(hipMemcpyAsync(total_miss, ctx->hdr, sizeof(ctx->hdr), hipMemcpyHostToDevice, ctx->st));
  S2_HIP(hipEventRecord(ctx->e0, ctx->st));
  if (ldg % 8 == 0 && ((uintptr_t)dG & 15) == 0)
    hipLaunchKernelGGL(k_s2_int_rows<true>, dim3(bs), dim3(256), 0, ctx->st, dG, ldg, n, Np, D, planes, (int64_t)bs * Np, vstat, total_miss);
  else
    hipLaunchKernelGGL(k_s2_int_rows<false>, dim3(bs), dim3(256), 0, ctx->st, dG, ldg, n, Np, D, planes, (int64_t)bs * Np, vstat, total_miss);
  rg_launch_xy_i8_planes(ctx->st, planes, (int64_t)bs * Np, nset, d_bs, Cv, n128, seg, ctx->dvd, Np, S);
  // A is laid out with the planes' full column count so that k_s2_packed_final indexes it as for hard calls; only [X | res] are filled
  hipLaunchKernelGGL(k_s2_int_combine, dim3((bs * Cv + 255) / 256), dim3(256), 0, ctx->st, (const int32_t*)S, ctx->dvsc, bs, n128, nseg, nset, D, inv_scale, Cv,
                     A);
  double* corr = nullptr;
  if (masked) {
    double* beta = (double*)ctx->buf[B_BETA];
    corr = (double*)ctx->buf[B_CORR];
    hipLaunchKernelGGL(k_s2_beta_from_sums, dim3((bs * C + 255) / 256), dim3(256), 0, ctx->st, (const double*)A, (const double*)vstat, inv_scale, bs, C, Cv, beta);
    static const bool per_variant = getenv("RG_S2_MASKED_OLD") && atoi(getenv("RG_S2_MASKED_OLD")) != 0;
    if (per_variant || C > 32)      // the matrix-core form carries two column tiles
      hipLaunchKernelGGL(k_s2_masked_int, dim3(bs, P), dim3(256), 0, ctx->st, dG, ldg, inv_scale, (const double*)ctx->d_xl, C, ctx->d_mlist, ctx->d_moff,
                         (const double*)vstat, (const double*)beta, P, corr);
    else
      hipLaunchKernelGGL(k_s2_masked_int_mfma, dim3((bs + 15) / 16, P), dim3(256), 0, ctx->st, dG, ldg, inv_scale, (const double*)ctx->d_xl, (const double*)ctx->d_xq,
                         C, ctx->d_mlist, ctx->d_moff, (const double*)vstat, (const double*)beta, bs, P, corr);
  }
  PackedFinal fa;
  fa.A = A; fa.Sq = nullptr; fa.vstat = vstat; fa.corr = corr; fa.inv_scale = inv_scale;
  fa.Lc = nullptr; fa.Lq = nullptr;
  fa.ytx = ctx->dYtX; fa.Q = ctx->dQ; fa.msum = ctx->dMsum; fa.scf_sv = ctx->dscf;
  fa.bs = bs; fa.C = C; fa.P = P; fa.Cvt = Cv; fa.cm0 = Cv; fa.CvB = 0; fa.sqoff = 0; fa.masked = masked;
  fa.n = n; fa.numtol = numtol; fa.nz_max = (double)(ctx->rule_n > 0 ? ctx->rule_n : n) * (1.0 - ctx->rule_thr);
  fa.zeros_min = ctx->rule_zero_count ? (double)(ctx->rule_n > 0 ? ctx->rule_n : n) * ctx->rule_thr : -1.0;
  fa.stats = stats; fa.bhat = bhat; fa.scale_fac = sf; fa.mean = mu; fa.total_p = total_p; fa.nobs = nobs; fa.ignored = ign; fa.nobs_p = nobs_p;
  hipLaunchKernelGGL(k_s2_packed_final, dim3((bs * P + 255) / 256), dim3(256), 0, ctx->st, fa);
  S2_HIP(hipEventRecord(ctx->e1, ctx->st));
  S2_HIP(hipGetLastError());
  if (out->stats) S2_HIP(hipMemcpyAsync(out->stats, stats, sizeof(double) * bs * P, hipMemcpyDeviceToHost, ctx->st));
  if (out->bhat) S2_HIP(hipMemcpyAsync(out->bhat, bhat, sizeof(double) * bs * P, hipMemcpyDeviceToHost, ctx->st));
  if (out->scale_fac) S2_HIP(hipMemcpyAsync(out->scale_fac, sf, sizeof(double) * bs, hipMemcpyDeviceToHost, ctx->st));
  if (out->mean) S2_HIP(hipMemcpyAsync(out->mean, mu, sizeof(double) * bs, hipMemcpyDeviceToHost, ctx->st));
  if (out->n_obs) S2_HIP(hipMemcpyAsync(out->n_obs, nobs, sizeof(int32_t) * bs, hipMemcpyDeviceToHost, ctx->st));
  if (out->ignored) S2_HIP(hipMemcpyAsync(out->ignored, ign, sizeof(int32_t) * bs, hipMemcpyDeviceToHost, ctx->st));
  S2_HIP(hipStreamSynchronize(ctx->st));
  float ms = 0.f;
  S2_HIP(hipEventElapsedTime(&ms, ctx->e0, ctx->e1));
  ctx->last_ms = ms;
  return RG_S2_OK;
}

int rg_s2_set_columns(rg_s2_ctx* ctx, int32_t n_col, const double* cols, int32_t n_sq) {
  if (!ctx || !ctx->st) return fail(ctx, RG_S2_ERR_ARG, "rg_s2_set_columns: context was not created");
  if (!cols || n_col < 1 || n_col > 4096 || n_sq < 0 || n_sq > n_col)
    return fail(ctx, RG_S2_ERR_ARG, "rg_s2_set_columns: need 1 <= n_col <= 4096 and 0 <= n_sq <= n_col");
  const int64_t n = ctx->n;
  S2_HIP(hipSetDevice(ctx->dev));
  const int ngrp = (n_col + 15) / 16;
  const int64_t Np = (n + 128 * RG_MAX_SEG - 1) / (128 * RG_MAX_SEG) * (128 * RG_MAX_SEG);
  if (ctx->g_ncol != n_col || !ctx->gV) {
    for (void** q : {(void**)&ctx->gV, (void**)&ctx->gvd, (void**)&ctx->gvsc})
      if (*q) { S2_HIP(hipFree(*q)); *q = nullptr; }
    ctx->g_ncol = 0;
    S2_HIP(hipMalloc((void**)&ctx->gV, sizeof(double) * ngrp * 16 * Np));
    S2_HIP(hipMalloc((void**)&ctx->gvd, (size_t)ngrp * 16 * 8 * Np));
    S2_HIP(hipMalloc((void**)&ctx->gvsc, sizeof(double) * ngrp * 16));
    S2_HIP(hipMemsetAsync(ctx->gV, 0, sizeof(double) * ngrp * 16 * Np, ctx->st));
  }
  S2_HIP(hipMemcpy2DAsync(ctx->gV, Np * sizeof(double), cols, n * sizeof(double), n * sizeof(double), n_col, hipMemcpyHostToDevice, ctx->st));
  rg_launch_v_split(ctx->st, ctx->gV, Np, ngrp * 16, ctx->gvd, ctx->gvsc);
  S2_HIP(hipGetLastError());
  S2_HIP(hipStreamSynchronize(ctx->st));
  ctx->g_ncol = n_col; ctx->g_nsq = n_sq;
  return RG_S2_OK;
}

int rg_s2_contract_packed(rg_s2_ctx* ctx, const uint8_t* rows, int64_t ld, int32_t bs, int32_t rows_on_device, int32_t flip,
                          const rg_s2_contract_out* out) {
  if (!ctx || !ctx->st) return fail(ctx, RG_S2_ERR_ARG, "rg_s2_contract_packed: context was not created");
  if (ctx->g_ncol < 1) return fail(ctx, RG_S2_ERR_ARG, "rg_s2_contract_packed: rg_s2_set_columns has not been called");
  const int64_t n = ctx->n, nbytes = (n + 3) / 4;
  if (!rows || !out || bs < 1 || ld < nbytes) return fail(ctx, RG_S2_ERR_ARG, "rg_s2_contract_packed: bad arguments (need bs >= 1, ld >= ceil(n / 4))");
  S2_HIP(hipSetDevice(ctx->dev));
  const int ncol = ctx->g_ncol, nsq = ctx->g_nsq, ngrp = (ncol + 15) / 16, ngrpB = (nsq + 15) / 16, CvB = ngrpB * 16;
  const int64_t Np = (n + 128 * RG_MAX_SEG - 1) / (128 * RG_MAX_SEG) * (128 * RG_MAX_SEG), ldp = Np / 4;
  const int n128 = (int)((bs + 127) / 128 * 128);
  SegLayout seg, segB;
  const int nseg = pick_segments(Np, n128 / 128, rg_xy_i8_launch_groups(ncol), seg), nsegB = pick_segments(Np, n128 / 128, std::max(1, ngrpB), segB);
  enum { Q_PK, Q_CNT, Q_S, Q_A };
  const size_t s_grp = (size_t)2 * nseg * n128 * 128, s_grpB = (size_t)2 * nsegB * n128 * 128;
  int rc;
  if ((rc = ensure_p(ctx, Q_PK, (size_t)bs * ldp))) return rc;
  if ((rc = ensure_p(ctx, Q_CNT, ((size_t)bs * 4 + 8) * sizeof(int32_t)))) return rc;
  if ((rc = ensure_p(ctx, Q_S, ((size_t)ngrp * s_grp + (size_t)ngrpB * s_grpB) * sizeof(int32_t)))) return rc;
  if ((rc = ensure_p(ctx, Q_A, (size_t)bs * 2 * (ncol + CvB) * sizeof(double)))) return rc;
  uint8_t* pk = (uint8_t*)ctx->pbuf[Q_PK];
  int32_t* cnt = (int32_t*)ctx->pbuf[Q_CNT];
  int32_t* total_miss = cnt + (size_t)bs * 4;
  int32_t* d_bs = total_miss + 1;
  int32_t* d_zero = total_miss + 2;
  int32_t* S = (int32_t*)ctx->pbuf[Q_S];
  double* A = (double*)ctx->pbuf[Q_A];
  double* Sq = A + (size_t)bs * 2 * ncol;
  ctx->hdr[0] = 0; ctx->hdr[1] = bs; ctx->hdr[2] = 0;
  S2_HIP(hipMemcpyAsync(total_miss, ctx->hdr, sizeof(ctx->hdr), hipMemcpyHostToDevice, ctx->st));
  S2_HIP(hipMemcpy2DAsync(pk, ldp, rows, ld, nbytes, bs, rows_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, ctx->st));
  S2_HIP(hipEventRecord(ctx->e0, ctx->st));
  hipLaunchKernelGGL(k_s2_rows, dim3(bs), dim3(256), 0, ctx->st, pk, ldp, n, flip ? 1 : 0, cnt, total_miss, (double*)nullptr);
  int32_t h_miss = 0;      // as in rg_s2_qt_block_packed: both sets in one pass when the block has a missing call
  S2_HIP(hipMemcpyAsync(&h_miss, total_miss, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->st));
  S2_HIP(hipStreamSynchronize(ctx->st));
  if (h_miss > 0) rg_launch_xy_i8_both(ctx->st, pk, ldp, d_bs, ncol, n128, seg, ctx->gvd, Np, RG_XY_LUT_DOSAGE, S);
  else rg_launch_xy_i8_sums(ctx->st, pk, ldp, d_bs, total_miss, ncol, n128, seg, ctx->gvd, Np, RG_XY_LUT_DOSAGE, S);
  hipLaunchKernelGGL(k_s2_combine, dim3((bs * ncol + 255) / 256), dim3(256), 0, ctx->st, (const int32_t*)S, ctx->gvsc, total_miss, bs, n128, nseg, ncol, A);
  if (nsq > 0) {
    rg_launch_xy_i8_sums(ctx->st, pk, ldp, d_bs, d_zero, nsq, n128, segB, ctx->gvd, Np, RG_XY_LUT_SQUARE, S + (size_t)ngrp * s_grp);
    hipLaunchKernelGGL(k_s2_combine, dim3((bs * CvB + 255) / 256), dim3(256), 0, ctx->st, (const int32_t*)(S + (size_t)ngrp * s_grp), ctx->gvsc,
                       (const int32_t*)nullptr, bs, n128, nsegB, CvB, Sq);
  }
  S2_HIP(hipEventRecord(ctx->e1, ctx->st));
  S2_HIP(hipGetLastError());
  if (out->sums) S2_HIP(hipMemcpyAsync(out->sums, A, sizeof(double) * bs * 2 * ncol, hipMemcpyDeviceToHost, ctx->st));
  if (out->counts) S2_HIP(hipMemcpyAsync(out->counts, cnt, sizeof(int32_t) * bs * 4, hipMemcpyDeviceToHost, ctx->st));
  if (out->sq && nsq > 0)     // set 0 of [bs][2][CvB]: the first nsq entries of each row
    S2_HIP(hipMemcpy2DAsync(out->sq, sizeof(double) * nsq, Sq, sizeof(double) * 2 * CvB, sizeof(double) * nsq, bs, hipMemcpyDeviceToHost, ctx->st));
  S2_HIP(hipStreamSynchronize(ctx->st));
  float ms = 0.f;
  S2_HIP(hipEventElapsedTime(&ms, ctx->e0, ctx->e1));
  ctx->last_ms = ms;
  return RG_S2_OK;
}

int rg_s2_contract_int(rg_s2_ctx* ctx, const uint16_t* G, int64_t ld, int32_t bs, int32_t g_on_device, int32_t scale, const rg_s2_contract_out* out) {
  if (!ctx || !ctx->st) return fail(ctx, RG_S2_ERR_ARG, "rg_s2_contract_int: context was not created");
  if (ctx->g_ncol < 1) return fail(ctx, RG_S2_ERR_ARG, "rg_s2_contract_int: rg_s2_set_columns has not been called");
  const int64_t n = ctx->n;
  if (!G || !out || bs < 1 || ld < n || scale < 1 || scale > 16384)
    return fail(ctx, RG_S2_ERR_ARG, "rg_s2_contract_int: bad arguments (need bs >= 1, ld >= n, 1 <= scale <= 16384)");
  S2_HIP(hipSetDevice(ctx->dev));
  const int ncol = ctx->g_ncol, nsq = ctx->g_nsq, ngrp = (ncol + 15) / 16;
  const int64_t Np = (n + 128 * RG_MAX_SEG - 1) / (128 * RG_MAX_SEG) * (128 * RG_MAX_SEG);
  const int D = 2 * scale <= 8127 ? 2 : 3, nset = D + 1;
  const int n128 = (int)((bs + 127) / 128 * 128);
  SegLayout seg;
  int nseg = pick_segments(Np, n128 / 128, ngrp * nset, seg);
  while (Np / nseg > 262144 && nseg < RG_MAX_SEG) nseg *= 2;
  if (Np / nseg > 524288) return fail(ctx, RG_S2_ERR_ARG, "rg_s2_contract_int: more than 16.7 million samples");
  memset(&seg, 0, sizeof(seg));
  seg.nseg = nseg;
  for (int f = 0; f < nseg; ++f) { seg.pos_start[f] = f * (Np / nseg); seg.file_start[f] = seg.pos_start[f]; seg.len[f] = Np / nseg; seg.plen[f] = Np / nseg; }
  enum { Q_PK, Q_CNT, Q_S, Q_A, Q_VAR };
  enum { B_G, B_PART, B_BETA, B_VAR, B_NOBS, B_STAT, B_CORR, B_PLANES };
  int rc;
  if ((rc = ensure_p(ctx, Q_CNT, ((size_t)bs * 4 + 8) * sizeof(int32_t)))) return rc;
  if ((rc = ensure_p(ctx, Q_S, (size_t)ngrp * nset * nseg * n128 * 128 * sizeof(int32_t)))) return rc;
  if ((rc = ensure_p(ctx, Q_A, ((size_t)bs * 2 * ncol + (size_t)bs * std::max(1, nsq)) * sizeof(double)))) return rc;
  if ((rc = ensure_p(ctx, Q_VAR, (size_t)bs * (6 * sizeof(double) + 2 * sizeof(int32_t))))) return rc;
  if ((rc = ensure(ctx, B_PLANES, (size_t)nset * bs * Np))) return rc;
  const int64_t ld8 = (n + 7) / 8 * 8;
  if (!g_on_device && (rc = ensure(ctx, B_G, (size_t)bs * ld8 * sizeof(uint16_t)))) return rc;
  const uint16_t* dG = G;
  int64_t ldg = ld;
  if (!g_on_device) {
    if (ld == ld8) S2_HIP(hipMemcpyAsync(ctx->buf[B_G], G, ((size_t)(bs - 1) * ld8 + n) * sizeof(uint16_t), hipMemcpyHostToDevice, ctx->st));     // rows already on the staged pitch: one flat copy
    else S2_HIP(hipMemcpy2DAsync(ctx->buf[B_G], ld8 * sizeof(uint16_t), G, ld * sizeof(uint16_t), n * sizeof(uint16_t), bs, hipMemcpyHostToDevice, ctx->st));
    dG = (const uint16_t*)ctx->buf[B_G];
    ldg = ld8;
  }
  int32_t* cnt = (int32_t*)ctx->pbuf[Q_CNT];
  int32_t* total_miss = cnt + (size_t)bs * 4;
  int32_t* d_bs = total_miss + 1;
  int32_t* S = (int32_t*)ctx->pbuf[Q_S];
  double* A = (double*)ctx->pbuf[Q_A];
  double* Sq = A + (size_t)bs * 2 * ncol;
  double* vstat = (double*)ctx->pbuf[Q_VAR] + (size_t)bs * 2;
  int8_t* planes = (int8_t*)ctx->buf[B_PLANES];
  const double inv_scale = 1.0 / (double)scale;
  ctx->hdr[0] = 0; ctx->hdr[1] = bs; ctx->hdr[2] = 0;
  S2_HIP(hipMemcpyAsync(total_miss, ctx->hdr, sizeof(ctx->hdr), hipMemcpyHostToDevice, ctx->st));
  S2_HIP(hipEventRecord(ctx->e0, ctx->st));
  if (ldg % 8 == 0 && ((uintptr_t)dG & 15) == 0)
    hipLaunchKernelGGL(k_s2_int_rows<true>, dim3(bs), dim3(256), 0, ctx->st, dG, ldg, n, Np, D, planes, (int64_t)bs * Np, vstat, total_miss);
  else
    hipLaunchKernelGGL(k_s2_int_rows<false>, dim3(bs), dim3(256), 0, ctx->st, dG, ldg, n, Np, D, planes, (int64_t)bs * Np, vstat, total_miss);
  rg_launch_xy_i8_planes(ctx->st, planes, (int64_t)bs * Np, nset, d_bs, ncol, n128, seg, ctx->gvd, Np, S);
  hipLaunchKernelGGL(k_s2_int_combine, dim3((bs * ncol + 255) / 256), dim3(256), 0, ctx->st, (const int32_t*)S, ctx->gvsc, bs, n128, nseg, nset, D, inv_scale, ncol, A);
  if (nsq > 0)
    hipLaunchKernelGGL(k_s2_int_sq, dim3(bs, (nsq + 7) / 8), dim3(256), 0, ctx->st, dG, ldg, n, Np, inv_scale, (const double*)ctx->gV, nsq, Sq);
  S2_HIP(hipEventRecord(ctx->e1, ctx->st));
  S2_HIP(hipGetLastError());
  if (out->sums) S2_HIP(hipMemcpyAsync(out->sums, A, sizeof(double) * bs * 2 * ncol, hipMemcpyDeviceToHost, ctx->st));
  if (out->sq && nsq > 0) S2_HIP(hipMemcpyAsync(out->sq, Sq, sizeof(double) * bs * nsq, hipMemcpyDeviceToHost, ctx->st));
  if (out->vstat) S2_HIP(hipMemcpyAsync(out->vstat, vstat, sizeof(double) * bs * 4, hipMemcpyDeviceToHost, ctx->st));
  S2_HIP(hipStreamSynchronize(ctx->st));
  float ms = 0.f;
  S2_HIP(hipEventElapsedTime(&ms, ctx->e0, ctx->e1));
  ctx->last_ms = ms;
  return RG_S2_OK;
}

double rg_s2_last_kernel_ms(const rg_s2_ctx* ctx) { return ctx ? ctx->last_ms : 0.0; }

}  // extern "C"
