// Step-2 score test of binary and count traits, with the approximate Firth and saddlepoint corrections, behind the C ABI
// (include/rg_step2.h: rg_s2_bt_set_null / rg_s2_bt_score_packed / rg_s2_bt_score_int / rg_s2_bt_correct).
//
// Reference path: compute_res_bin / compute_res_count leave, per chromosome and trait, the null model's fitted mean p^ (offset = LOCO
// prediction), Gamma_sqrt = sqrt(w) with w = p^ (1 - p^) (the Poisson rate for counts) and the projector of Gamma X (Data.cpp:2439-2455,
// Step1_Models.cpp:54-140, :225-288); compute_score_bt / compute_score_ct (Step2_Models.cpp:471-622) then need per variant and trait
//   sum_i w g~^2,   X^T W g~ (C numbers)   and   g~ . (y - p^)             (g~ = the mean-imputed genotype)
// -- contractions of the genotype row with P (C + 3) fixed columns [w | w x_c | y - p^ | mask], which the contraction primitive of
// step2_qt.hip / xy_i8.hip evaluates exactly on the i8 matrix cores (hard calls at 2 bits, integer dosages as uint16) -- followed by
// C x C algebra per (variant, trait):  denum = sum w g~^2 - (X^T W g~)^T (X^T W X)^-1 (X^T W g~),  z = g~ . (y - p^) / sqrt(denum).
// check_pval_snp (Step2_Models.cpp:1987-2029) re-tests what lies above the threshold:
//   --firth --approx   fit_firth_logistic_snp_fast (:1158-1253): one-parameter penalised fit of Gres / Gamma_sqrt with the covariate
//                      effects of the null Firth model in the offset (carriers only for rare sparse variants)
//   --spa              run_SPA_test_snp (:2072-2297): Lugannani-Rice with the cumulant generating function of sum gm_i (Y_i - p_i) / c
// Both are O(n) per Newton step and independent per (variant, trait): ONE workgroup per flagged pair walks the samples (residualised
// genotype kept in a scratch row, excluded samples marked), every sum reduced in a fixed order, every decision taken from the reduced
// values by all threads alike.  The iterations follow the host restatements that are pinned against regenie's own output
// (oracle/regenie_step2_bt.py; the C++ driver used to run them on host threads).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>

#include "step2_internal.h"

struct BtPairDev {
  int32_t variant, trait, fast;
  int32_t flip;                  // the reference tests 2 - g (flip_geno, Geno.cpp:3150-3162: mean dosage > 1): only WHO is a carrier depends on it
  double mu, stats, denum;
  double tc[RG_S2_MAX_COV];      // (X^T W X)^-1 X^T W g~: the covariate coefficients taken out of g~
};

struct BtState {
  int family = 0;
  int niter_max = 50;                         // --niter (params->niter_max): fit_firth_pseudo gives up on a logistic step that took more iterations
  bool have_null = false, have_firth = false;
  int ncol = 0;
  std::vector<double> xwx_inv, msum, xres, rsum;     // xres [P][C] = (X^T W X)^-1 X^T ((y - fitted) mask): the null model's residual score, ~0; rsum [P] = sum of (y - fitted) mask
  std::vector<uint8_t> pass;
  double *dX = nullptr, *dY = nullptr, *dFit = nullptr, *dFo = nullptr;
  uint8_t* dM = nullptr;
  // the block last scored
  int bs = 0, kind = 0, scale = 0;          // kind 1: packed rows in pbuf[RG_S2_Q_PK], 2: uint16 rows
  const uint8_t* d_pk = nullptr; int64_t ldp = 0;
  const uint16_t* d_g16 = nullptr; int64_t ldg = 0;
  std::vector<double> sums, sq, mu, denum, stats;
  std::vector<uint8_t> flip;                // [bs] flip_geno's verdict: the reference tests the other allele of this variant
  int32_t* d_two = nullptr; size_t two_cap = 0;      // [bs] observed entries of an integer-dosage row that are exactly 2
  // correction scratch
  double* d_v = nullptr; size_t v_cap = 0;
  void* d_pairs = nullptr; size_t pairs_cap = 0;
  double* d_res = nullptr; size_t res_cap = 0;
};

namespace {

// ---- device helpers ---------------------------------------------------------------------------------------------------------------
constexpr double kNumtolEps = 10 * 2.220446049250313e-16;      // Regenie.hpp:225
__device__ __forceinline__ double bt_pvec(double eta) {        // get_pvec (Step1_Models.cpp:1799-1806)
  double pr = 1.0 - 1.0 / (exp(eta) + 1.0);
  if (eta < -30.0) pr = kNumtolEps / (1.0 + kNumtolEps);
  if (eta > 30.0) pr = 1.0 / (1.0 + kNumtolEps);
  return pr;
}
// sum of NV per-thread values over the 256 threads of the workgroup, the same total in every thread; fixed tree
template <int NV>
__device__ __forceinline__ void bt_block_sum(double (&x)[NV], double (*red)[4]) {
#pragma unroll
  for (int v = 0; v < NV; ++v)
    for (int o = 32; o > 0; o >>= 1) x[v] += __shfl_down(x[v], o);
  __syncthreads();
  if ((threadIdx.x & 63) == 0)
#pragma unroll
    for (int v = 0; v < NV; ++v) red[v][threadIdx.x >> 6] = x[v];
  __syncthreads();
#pragma unroll
  for (int v = 0; v < NV; ++v) x[v] = (red[v][0] + red[v][1]) + (red[v][2] + red[v][3]);
}
// the mean-imputed genotype of sample i: hard calls (.bed coding after k_s2_rows: 00 -> 2, 01 -> missing, 10 -> 1, 11 -> 0) or uint16 dosages
__device__ __forceinline__ double bt_geno(int kind, const uint8_t* pk, const uint16_t* g16, int64_t i, double mu, double inv_scale) {
  if (kind == 1) {
    const unsigned code = (pk[i >> 2] >> (2 * (i & 3))) & 3u;
    return code == 0 ? 2.0 : (code == 2 ? 1.0 : (code == 3 ? 0.0 : mu));
  }
  const uint16_t v = g16[i];
  return v == 0xFFFFu ? mu : (double)v * inv_scale;
}

// ---- integer dosages: observed entries of a row that are exactly 2 copies (the zeros of the flipped coding 2 - g) -------------------------
__global__ __launch_bounds__(256) void k_bt_count_two(const uint16_t* __restrict__ g16, int64_t ldg, int64_t n, unsigned two, int32_t* __restrict__ out) {
  __shared__ int red[4];
  const uint16_t* g = g16 + (int64_t)blockIdx.x * ldg;
  int c = 0;
  for (int64_t i = threadIdx.x; i < n; i += 256) c += (unsigned)g[i] == two;
  for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// ---- residualised genotype of every flagged pair: v_i = g~_i - sum_c x_ci tc_c on the samples that enter the exact terms, NaN elsewhere ----
// aux [pair][8] = a_all = sum v p^, lo = sum of the negative v, hi = sum of the positive v (all over the unmasked samples), then over the
// carriers of a fast (sparse) pair: sum (v Gamma_sqrt)^2, sum v p^; number of samples kept.
__global__ __launch_bounds__(256) void k_bt_prep(const BtPairDev* __restrict__ pairs, int corr_kind, int kind, const uint8_t* __restrict__ pk, int64_t ldp,
                                                 const uint16_t* __restrict__ g16, int64_t ldg, double inv_scale, unsigned two, const double* __restrict__ X,
                                                 const uint8_t* __restrict__ M, const double* __restrict__ fit, int64_t n, int C, double* __restrict__ V,
                                                 double* __restrict__ aux) {
  __shared__ double red[6][4];
  __shared__ double stc[RG_S2_MAX_COV];
  const BtPairDev& pr = pairs[blockIdx.x];
  if (threadIdx.x < C) stc[threadIdx.x] = pr.tc[threadIdx.x];
  __syncthreads();
  const uint8_t* row = pk + (int64_t)pr.variant * ldp;
  const uint16_t* grow = g16 + (int64_t)pr.variant * ldg;
  const uint8_t* mq = M + (int64_t)pr.trait * n;
  const double* pq = fit + (int64_t)pr.trait * n;
  double* v = V + (int64_t)blockIdx.x * n;
  double s[6] = {0, 0, 0, 0, 0, 0};
  for (int64_t i = threadIdx.x; i < n; i += 256) {
    double out = __longlong_as_double(0x7ff8000000000000LL);      // NaN: not part of the exact terms
    if (mq[i]) {
      const double gt = bt_geno(kind, row, grow, i, pr.mu, inv_scale);
      double r = gt;
      for (int c = 0; c < C; ++c) r -= X[(int64_t)c * n + i] * stc[c];
      const double ph = pq[i];
      s[0] += r * ph;
      if (r < 0) s[1] += r; else s[2] += r;
      // carriers only in the fast forms: fit_firth_logistic_snp_fast drops G <= 1e-4, run_SPA_test_snp's fastSPA the exact zeros -- of the
      // coding the reference tests: 2 - g for a flipped variant (flip_geno).  The residual of 2 - g is -r (the intercept is in the span of X),
      // the score -stats, and every quantity below is odd or even in that sign as the reference's own BETA = -BETA back-flip needs: the carriers
      // are all the flip changes.  Integer dosages: the zero of 2 - g is decided on the integers.
      double gc = gt;
      if (pr.flip) gc = (kind == 2 && grow[i] != 0xFFFFu) ? (double)((int)two - (int)grow[i]) * inv_scale : 2.0 - gt;
      const bool keep = !pr.fast || (corr_kind == RG_S2_BT_FIRTH_APPROX ? gc > 1e-4 : gc != 0.0);
      if (keep) {
        out = r;
        s[5] += 1.0;
        if (pr.fast) { s[3] += r * r * ph * (1.0 - ph); s[4] += r * ph; }
      }
    }
    v[i] = out;
  }
  bt_block_sum<6>(s, red);
  if (threadIdx.x == 0)
#pragma unroll
    for (int t = 0; t < 6; ++t) aux[(int64_t)blockIdx.x * 8 + t] = s[t];
}

// ---- fit_firth_logistic_snp_fast with its one-parameter solver: Fisher scoring with step halving on the penalised deviance ----------------
// res [pair][4] = beta, se, lrt, fail
__global__ __launch_bounds__(256) void k_bt_firth1(const BtPairDev* __restrict__ pairs, const double* __restrict__ V, const double* __restrict__ Y,
                                                   const double* __restrict__ Fo, const double* __restrict__ aux, int64_t n, double* __restrict__ res, int niter_max) {
  __shared__ double red[2][4];
  __shared__ double red4[4][4];
  const BtPairDev& pr = pairs[blockIdx.x];
  if (aux[(int64_t)blockIdx.x * 8 + 5] == 0.0) {      // no sample enters the fit
    if (threadIdx.x == 0) { double* r = res + (int64_t)blockIdx.x * 4; r[0] = r[1] = r[2] = 0.0; r[3] = 1.0; }
    return;
  }
  const double* v = V + (int64_t)blockIdx.x * n;
  const double* y = Y + (int64_t)pr.trait * n;
  const double* o = Fo + (int64_t)pr.trait * n;
  auto state = [&](double b, double& xtwx, double& dev) {
    double s[2] = {0.0, 0.0};
    for (int64_t i = threadIdx.x; i < n; i += 256) {
      const double g = v[i];
      if (!(g == g)) continue;
      const double p = bt_pvec(o[i] + g * b);
      s[0] -= (y[i] == 0.0) ? log(1.0 - p) : log(p);
      s[1] += g * g * p * (1.0 - p);
    }
    bt_block_sum<2>(s, red);
    xtwx = s[1];
    dev = 2.0 * s[0] - log(s[1]);
  };
  auto score_at = [&](double b, double xtwx) {
    double s[1] = {0.0};
    for (int64_t i = threadIdx.x; i < n; i += 256) {
      const double g = v[i];
      if (!(g == g)) continue;
      const double p = bt_pvec(o[i] + g * b);
      s[0] += g * (y[i] - p + g * g * p * (1.0 - p) / xtwx * (0.5 - p));
    }
    bt_block_sum<1>(s, red);
    return s[0];
  };
  double xtwx, dev, dev0, beta = 0.0;
  state(0.0, xtwx, dev0);
  // ---- the reference's first solver, to the letter: fit_firth_pseudo (Step2_Models.cpp:1548-1665), one parameter, started at 0 -- IRLS on the pseudo-response
  // y* = y + h (0.5 - p), stopped at |modified score| < numtol_firth = 2.5e-4: BETA / SE / LRT are those of the iterate it stops at, not of the root.  Every
  // thread follows the same scalars (the block sums return the same value to all).  Fit states 1 - 4 (too slow, a growing step, p = 0, LRT < 0) leave
  // `pseudo` at 0 and the root finder below takes over, where the reference runs its Newton solvers.
  {
    const int niter = pr.fast ? 125 : 50;            // niter_max_firth / 2, at most 50 unless the fit is on the carriers (:1166, :1186)
    const double tol = 2.5e-4;
    auto sums_at = [&](double b, double xt, double bstar, bool want_state, double& o_score, double& o_xtwx, double& o_dev, double& o_zero) {
      // score = sum g (y* - p(b)) with y* = y + h(bstar) (0.5 - p(bstar)), h = g^2 w / xt; and, when asked, sum g^2 w(b), the deviance and zeros of w at b
      double t[4] = {0.0, 0.0, 0.0, 0.0};
      for (int64_t i = threadIdx.x; i < n; i += 256) {
        const double g = v[i];
        if (!(g == g)) continue;
        const double ps = bt_pvec(o[i] + g * bstar);
        const double ystar = y[i] + g * g * ps * (1.0 - ps) / xt * (0.5 - ps);
        const double pb = (b == bstar) ? ps : bt_pvec(o[i] + g * b);
        t[0] += g * (ystar - pb);
        if (want_state) {
          const double w = pb * (1.0 - pb);
          t[1] += g * g * w;
          t[2] -= (y[i] == 0.0) ? log(1.0 - pb) : log(pb);
          t[3] += (w == 0.0) ? 1.0 : 0.0;
        }
      }
      bt_block_sum<4>(t, red4);
      o_score = t[0]; o_xtwx = t[1]; o_dev = 2.0 * t[2]; o_zero = t[3];
    };
    int pseudo = 0;           // 1: stopped by its criterion
    double b = 0.0, b14 = 0.0, xt = xtwx, dv = dev0, se_x = xtwx;
    for (int itp = 1; itp <= niter && !pseudo; ++itp) {
      double sc, x_b, d_b, zz;
      sums_at(b, xt, b, true, sc, x_b, d_b, zz);             // (xt is sum g^2 w at b: kept from the last update, or from state(0))
      dv = d_b - log(xt);
      if (fabs(sc) < tol && itp >= 2) { pseudo = 1; se_x = xt; break; }
      if (itp == 14) b14 = b;
      if (itp == 15 && fabs(b - b14) > 0.1) break;
      double bdiff = 1e16, bnew = b, xin = xt;
      const double bstar = b, xstar = xt;
      bool bad = false, done = false;
      int nlog = 26;                                          // the reference's niter_log after its `while (niter_log++ < 25)`: 26 when the loop ran out
      for (int il = 0; il < 25; ++il) {
        const double step = sc / xin, bd = fabs(step);
        if (bd > bdiff) { bad = true; break; }
        const double mx = bd / 5.0;
        bnew = b + (mx > 1.0 ? step / mx : step);
        double s2, x2, d2, z2;
        sums_at(bnew, xstar, bstar, true, s2, x2, d2, z2);
        sc = s2;
        if (fabs(sc) < tol) { done = true; xin = x2; nlog = il + 1; break; }
        if (z2 > 0.0) { bad = true; break; }
        xin = x2; b = bnew; bdiff = bd;
      }
      (void)done;
      if (bad) break;
      if (nlog > niter_max) break;                            // `if (niter_log > params->niter_max) return 1` (Step2_Models.cpp:1625): only with --niter below 25
      b = bnew;
      // sum g^2 w at the new b (the inner loop left it in xin when it updated there; after its `break` on the score it belongs to bnew as well)
      xt = xin;
    }
    if (pseudo) {
      const double lrt = dev0 - dv;
      if (lrt >= 0) {
        if (threadIdx.x == 0) { double* r = res + (int64_t)blockIdx.x * 4; r[0] = b; r[1] = sqrt(1.0 / se_x); r[2] = lrt; r[3] = 0.0; }
        return;
      }
    }
  }
  dev = dev0;
  bool conv = false;
  int it = 0;
  for (; it < 500; ++it) {
    const double sc = score_at(beta, xtwx);
    double step = sc / xtwx;
    if (fabs(step) < 1e-9) { conv = true; break; }
    if (fabs(step) > 5.0) step = step > 0 ? 5.0 : -5.0;
    double x_n = xtwx, dev_n = dev;
    bool ok = false;
    for (int hs = 0; hs < 60; ++hs) {
      state(beta + step, x_n, dev_n);
      if (dev_n <= dev || fabs(step) < 1e-6) { ok = true; break; }
      step /= 2.0;
    }
    if (!ok) { state(beta, xtwx, dev); conv = fabs(sc / xtwx) < 1e-6; break; }
    beta += step; xtwx = x_n; dev = dev_n;
  }
  if (threadIdx.x == 0) {
    const double lrt = dev0 - dev;
    double* r = res + (int64_t)blockIdx.x * 4;
    r[0] = beta; r[1] = sqrt(1.0 / xtwx); r[2] = lrt;
    r[3] = (!conv || !(lrt >= 0)) ? 1.0 : 0.0;
  }
}

// ---- run_SPA_test_snp: the two saddlepoints by Newton with the bisection safeguard, Lugannani-Rice -----------------------------------------
// res [pair][4] = p-value, -, -, fail
__global__ __launch_bounds__(256) void k_bt_spa(const BtPairDev* __restrict__ pairs, const double* __restrict__ V, const double* __restrict__ fit,
                                                const double* __restrict__ aux, int64_t n, double* __restrict__ res) {
  __shared__ double red[2][4];
  const BtPairDev& pr = pairs[blockIdx.x];
  const double* gm = V + (int64_t)blockIdx.x * n;
  const double* ph = fit + (int64_t)pr.trait * n;
  const double* ax = aux + (int64_t)blockIdx.x * 8;
  const bool fast = pr.fast != 0;
  const double denum = pr.denum, c = sqrt(denum), tol = pow(2.220446049250313e-16, 0.25);
  const double a = ax[0], lo = ax[1] - a, hi = ax[2] - a, b = fast ? denum - ax[3] : denum, d = ax[4];
  double* out = res + (int64_t)blockIdx.x * 4;
  if (pr.stats * c < lo || pr.stats * c > hi || ax[5] == 0.0) { if (threadIdx.x == 0) { out[0] = 0.0; out[3] = 1.0; } return; }
  auto K = [&](double t) {
    double s[1] = {0.0};
    for (int64_t i = threadIdx.x; i < n; i += 256) {
      const double g = gm[i];
      if (!(g == g)) continue;
      s[0] += log(1.0 - ph[i] + ph[i] * exp(t / c * g));
    }
    bt_block_sum<1>(s, red);
    return s[0] + (fast ? -t * d / c + t * t / 2.0 / denum * b : -t * a / c);
  };
  auto K1 = [&](double t) {
    double s[1] = {0.0};
    for (int64_t i = threadIdx.x; i < n; i += 256) {
      const double g = gm[i];
      if (!(g == g)) continue;
      s[0] += (g * ph[i] / c) / (ph[i] + (1.0 - ph[i]) * exp(-t / c * g));
    }
    bt_block_sum<1>(s, red);
    return s[0] + (fast ? -d / c + t / denum * b : -a / c);
  };
  auto K2 = [&](double t) {      // 0 when an exponent passes MAX_EXP_LIM (the reference returns 0 there)
    double s[2] = {0.0, 0.0};
    for (int64_t i = threadIdx.x; i < n; i += 256) {
      const double g = gm[i];
      if (!(g == g)) continue;
      const double vexp = -t / c * g;
      if (vexp > 708.0) { s[1] += 1.0; continue; }
      const double e = exp(vexp), den = ph[i] + (1.0 - ph[i]) * e, gs2 = ph[i] * (1.0 - ph[i]);
      s[0] += (g * g * gs2 / (c * c) * e) / (den * den);
    }
    bt_block_sum<2>(s, red);
    if (s[1] > 0.0) return 0.0;
    return s[0] + (fast ? b / denum : 0.0);
  };
  const double tval = pr.stats >= 0 ? -pr.stats : pr.stats;
  double pv = 0.0;
  bool failed = false;
  for (int lam = 1; lam >= -1 && !failed; lam -= 2) {
    double min_x = tval >= 0 ? 0.0 : -1.7976931348623157e308, max_x = tval >= 0 ? 1.7976931348623157e308 : 0.0;
    double t_old = 0.0, f_old = lam * K1(lam * t_old) - tval, t_new = -1.0, f_new = 0.0;
    bool conv = false;
    for (int it = 0; it < 1000; ++it) {
      const double hess = K2(lam * t_old);
      if (hess == 0.0) { failed = true; break; }
      t_new = t_old - f_old / hess;
      f_new = lam * K1(lam * t_new) - tval;
      if (fabs(f_new) < tol) { conv = true; break; }
      if (t_new != 0.0 && t_new > min_x && t_new < max_x) { if (f_new > 0) max_x = t_new; else min_x = t_new; }
      else {
        t_new = (min_x + max_x) / 2.0;
        f_new = lam * K1(lam * t_new) - tval;
        if (f_new <= 0) min_x = t_new; else max_x = t_new;
      }
      t_old = t_new; f_old = f_new;
    }
    if (failed) break;
    if (!conv) { failed = true; break; }
    const double root = t_new, kval = K(lam * root), k2val = K2(lam * root);
    if (k2val == 0.0) { failed = true; break; }
    const double wval = (root > 0 ? 1.0 : root < 0 ? -1.0 : 0.0) * sqrt(2.0 * (root * tval - kval)), vval = root * sqrt(k2val);
    if (vval == 0.0) pv += 0.5;
    else { const double rval = wval + log(vval / wval) / wval; pv += 0.5 * erfc(-rval / sqrt(2.0)); }
  }
  if (!failed && !(pv <= 1.0)) failed = true;
  if (threadIdx.x == 0) { out[0] = pv; out[3] = failed ? 1.0 : 0.0; }
}

// ---- host helpers ------------------------------------------------------------------------------------------------------------------
// inverse of a small SPD matrix by Gaussian elimination with partial pivoting; false when singular
bool small_inverse(std::vector<double> A, int n, std::vector<double>& inv) {
  inv.assign((size_t)n * n, 0.0);
  for (int i = 0; i < n; ++i) inv[(size_t)i * n + i] = 1.0;
  for (int k = 0; k < n; ++k) {
    int piv = k;
    for (int i = k + 1; i < n; ++i) if (std::fabs(A[(size_t)i * n + k]) > std::fabs(A[(size_t)piv * n + k])) piv = i;
    if (A[(size_t)piv * n + k] == 0.0) return false;
    if (piv != k)
      for (int j = 0; j < n; ++j) { std::swap(A[(size_t)k * n + j], A[(size_t)piv * n + j]); std::swap(inv[(size_t)k * n + j], inv[(size_t)piv * n + j]); }
    const double d = A[(size_t)k * n + k];
    for (int j = 0; j < n; ++j) { A[(size_t)k * n + j] /= d; inv[(size_t)k * n + j] /= d; }
    for (int i = 0; i < n; ++i) {
      if (i == k) continue;
      const double f = A[(size_t)i * n + k];
      if (f == 0.0) continue;
      for (int j = 0; j < n; ++j) { A[(size_t)i * n + j] -= f * A[(size_t)k * n + j]; inv[(size_t)i * n + j] -= f * inv[(size_t)k * n + j]; }
    }
  }
  return true;
}

// quantile of the standard normal distribution: Wichura's algorithm AS 241 (PPND16), relative accuracy ~1e-16
double norm_quantile(double p) {
  const double q = p - 0.5;
  if (std::fabs(q) <= 0.425) {
    const double r = 0.180625 - q * q;
    return q * (((((((2509.0809287301226727 * r + 33430.575583588128105) * r + 67265.770927008700853) * r + 45921.953931549871457) * r + 13731.693765509461125) * r +
                   1971.5909503065514427) * r + 133.14166789178437745) * r + 3.387132872796366608) /
           (((((((5226.495278852545925 * r + 28729.085735721942674) * r + 39307.89580009271061) * r + 21213.794301586595867) * r + 5394.1960214247511077) * r +
              687.1870074920579083) * r + 42.313330701600911252) * r + 1.0);
  }
  double r = q < 0 ? p : 1.0 - p;
  r = std::sqrt(-std::log(r));
  double val;
  if (r <= 5.0) {
    r -= 1.6;
    val = (((((((7.7454501427834140764e-4 * r + 0.0227238449892691845833) * r + 0.24178072517745061177) * r + 1.27045825245236838258) * r + 3.64784832476320460504) * r +
             5.7694972214606914055) * r + 4.6303378461565452959) * r + 1.42343711074968357734) /
          (((((((1.05075007164441684324e-9 * r + 5.475938084995344946e-4) * r + 0.0151986665636164571966) * r + 0.14810397642748007459) * r + 0.68976733498510000455) * r +
              1.6763848301838038494) * r + 2.05319162663775882187) * r + 1.0);
  } else {
    r -= 5.0;
    val = (((((((2.01033439929228813265e-7 * r + 2.71155556874348757815e-5) * r + 0.0012426609473880784386) * r + 0.026532189526576123093) * r + 0.29656057182850489123) * r +
             1.7848265399172913358) * r + 5.4637849111641143699) * r + 6.6579046435011037772) /
          (((((((2.04426310338993978564e-15 * r + 1.4215117583164458887e-7) * r + 1.8463183175100546818e-5) * r + 7.868691311456132591e-4) * r + 0.0148753612908506148525) * r +
              0.13692988092273580531) * r + 0.59983224962614106254) * r + 1.0);
  }
  return q < 0 ? -val : val;
}

int grow(rg_s2_ctx* ctx, void** p, size_t* cap, size_t bytes) {
  if (*cap >= bytes) return RG_S2_OK;
  if (*p) S2_HIP(hipFree(*p));
  *p = nullptr; *cap = 0;
  S2_HIP(hipMalloc(p, bytes));
  *cap = bytes;
  return RG_S2_OK;
}

// the score test of every (variant, trait) of the block from the contraction sums (compute_score_bt / compute_score_ct with get_sumstats)
// n_two [bs]: observed entries equal to 2 (integer dosages; hard calls have it in counts)
int score_from_sums(rg_s2_ctx* ctx, int bs, double numtol, const int32_t* counts, const double* vstat, const int32_t* n_two, int scale, const rg_s2_bt_out* out) {
  BtState& bt = *ctx->bt;
  const int P = ctx->P, C = ctx->C, ncol = bt.ncol;
  const int64_t n = ctx->n;
  const double Nrule = ctx->rule_n > 0 ? (double)ctx->rule_n : (double)n;
  bt.mu.assign(bs, 0.0); bt.denum.assign((size_t)bs * P, 0.0); bt.stats.assign((size_t)bs * P, 0.0); bt.flip.assign(bs, 0);
  for (int j = 0; j < bs; ++j) {
    double nobs, tot, nnz, ntwo;
    if (vstat) { nobs = vstat[(size_t)j * 4 + 2]; tot = vstat[(size_t)j * 4] / scale; nnz = vstat[(size_t)j * 4 + 3]; ntwo = n_two ? (double)n_two[j] : 0.0; }
    else {
      const double n1 = counts[(size_t)j * 4], n2 = counts[(size_t)j * 4 + 1], nm = counts[(size_t)j * 4 + 2];
      nobs = (double)n - nm; tot = n1 + 2.0 * n2; nnz = n1 + n2; ntwo = n2;
    }
    const double mu = nobs > 0 ? tot / nobs : 0.0;
    bt.mu[j] = mu;
    if (out->mean) out->mean[j] = mu;
    if (out->ignored) out->ignored[j] = nobs > 0 ? 0 : 1;
    // flip_geno (Geno.cpp:3150-3162; params.with_flip: additive tests of binary / count traits, Data.cpp:2108): a variant whose mean dosage
    // exceeds 1 is tested as 2 - g and its BETA negated back.  Nothing printed depends on it but check_sparse_G's verdict (the non-zero
    // entries of 2 - g: the observed g != 2 and, unless 2 - mean = 0, the imputed ones) and, with it, who the carriers of the fast forms are.
    // The .pgen form of the rule counts the observed zeros BEFORE the flip (prep_snp_stats, Geno.cpp:2582-2594) and stays as it is.
    const bool flipped = mu > 1.0;
    bt.flip[j] = flipped ? 1 : 0;
    const double nnz_t = flipped ? nobs - ntwo : nnz, mu_t = flipped ? 2.0 - mu : mu;
    // check_sparse_G (Geno.cpp:3165-3177)
    const bool sparse_j = ctx->rule_zero_count ? (nobs - nnz) >= Nrule * ctx->rule_thr : (nnz_t + (mu_t != 0.0 ? (double)n - nobs : 0.0)) <= Nrule * (1.0 - ctx->rule_thr);
    if (out->sparse) out->sparse[j] = sparse_j ? 1 : 0;
    const double* s0 = bt.sums.data() + (size_t)j * 2 * ncol;
    const double* s1 = s0 + ncol;
    for (int q = 0; q < P; ++q) {
      const int cm = P + P * C + P + q, cr = P + P * C + q;
      if (out->total_p) out->total_p[(size_t)j * P + q] = vstat ? 0.0 : std::nearbyint(s0[cm]) - tot;      // per-trait corrections of the allele count ...
      if (out->n_obs_p) out->n_obs_p[(size_t)j * P + q] = vstat ? 0 : (int32_t)(std::nearbyint(bt.msum[q] - s1[cm]) - nobs);   // ... and of the sample count
      uint8_t ign = 0;
      double st = 0.0, bh = 0.0, denum = 0.0;
      if (!bt.pass[q]) ign = 1;
      else {
        const double sw2 = bt.sq[(size_t)j * P + q] + mu * mu * s1[q];                                         // sum w g~^2
        double quad = 0.0, rcorr = 0.0;                                                                         // (X^T W g~)^T (X^T W X)^-1 (X^T W g~)
        const double* inv = bt.xwx_inv.data() + (size_t)q * C * C;
        for (int a = 0; a < C; ++a) {
          const double ua = s0[P + q * C + a] + mu * s1[P + q * C + a];
          double row = 0.0;
          for (int c = 0; c < C; ++c) row += inv[(size_t)a * C + c] * (s0[P + q * C + c] + mu * s1[P + q * C + c]);
          quad += ua * row;
          rcorr += ua * bt.xres[(size_t)q * C + a];
        }
        denum = sw2 - quad;
        const double sd = std::sqrt(denum);
        if (bt.family == 1 ? !(denum >= numtol) : !(sd >= numtol)) ign = 1;                                    // Step2_Models.cpp:512-517, :596
        else {
          // The numerator.  Dense form: Gres . yres (Step2_Models.cpp:519, :605) = g~ . r - (X^T W g~)^T (X^T W X)^-1 X^T r with r = (y - p^) mask: yres is not
          // projected (Data.cpp:2443-2445), Gres is.  SPARSE form: GW . yres (:517, :603) -- the genotype of the coding the reference tests, not projected -- i.e.
          // g~ . r as it stands, or for a flipped variant (2 - g~) . r, which is -(g~ . r - 2 sum r) in the coding given.  The forms differ by the null model's
          // score at its stopping point (|score| < 1e-6): the seventh digit of the statistic, and every tenth printed line.
          const double gr = s0[cr] + mu * s1[cr];
          const double num = !sparse_j ? gr - rcorr : (flipped ? gr - 2.0 * bt.rsum[q] : gr);
          st = num / sd; bh = st / sd;                                                                                    // get_sumstats (Step2_Models.cpp:2031-2041)
        }
      }
      bt.denum[(size_t)j * P + q] = ign ? 0.0 : denum;
      bt.stats[(size_t)j * P + q] = st;
      if (out->test_ignored) out->test_ignored[(size_t)j * P + q] = ign;
      if (out->stats) out->stats[(size_t)j * P + q] = st;
      if (out->bhat) out->bhat[(size_t)j * P + q] = bh;
      if (out->denum) out->denum[(size_t)j * P + q] = ign ? 0.0 : denum;
    }
  }
  return RG_S2_OK;
}

}  // namespace

void rg_s2_bt_free(rg_s2_ctx* ctx) {
  if (!ctx || !ctx->bt) return;
  BtState* bt = ctx->bt;
  for (void* p : {(void*)bt->dX, (void*)bt->dY, (void*)bt->dFit, (void*)bt->dFo, (void*)bt->dM, (void*)bt->d_v, bt->d_pairs, (void*)bt->d_res, (void*)bt->d_two})
    if (p) (void)hipFree(p);
  delete bt;
  ctx->bt = nullptr;
}

extern "C" {

int rg_s2_bt_set_null(rg_s2_ctx* ctx, const rg_s2_bt_null* nm) {
  if (!ctx || !ctx->st) return rg_s2_fail(ctx, RG_S2_ERR_ARG, "rg_s2_bt_set_null: context was not created");
  if (!nm || !nm->X || !nm->y || !nm->mask || !nm->fitted || nm->family < 0 || nm->family > 1)
    return rg_s2_fail(ctx, RG_S2_ERR_ARG, "rg_s2_bt_set_null: null argument / family must be 0 (binary) or 1 (count)");
  S2_HIP(hipSetDevice(ctx->dev));
  if (!ctx->bt) ctx->bt = new BtState();
  BtState& bt = *ctx->bt;
  const int64_t n = ctx->n;
  const int P = ctx->P, C = ctx->C;
  bt.family = nm->family;
  bt.niter_max = nm->niter_max > 0 ? nm->niter_max : 50;
  bt.ncol = P * (C + 3);
  if (bt.ncol > 4096) return rg_s2_fail(ctx, RG_S2_ERR_ARG, "rg_s2_bt_set_null: phenotypes x (covariates + 3) > 4096 contraction columns");
  bt.pass.assign(P, 1);
  if (nm->pass) bt.pass.assign(nm->pass, nm->pass + P);
  bt.msum.assign(P, 0.0);
  bt.rsum.assign(P, 0.0);
  bt.xwx_inv.assign((size_t)P * C * C, 0.0);
  bt.xres.assign((size_t)P * C, 0.0);
  // the contraction columns [w_q] (also against g^2) | [w_q x_c] | [y_q - fitted_q] | [mask_q], and (X^T W X)^-1 per trait
  std::vector<double> cols((size_t)bt.ncol * n, 0.0), A((size_t)C * C), inv;
  for (int q = 0; q < P; ++q) {
    const double* yq = nm->y + (size_t)q * n;
    const double* fq = nm->fitted + (size_t)q * n;
    const uint8_t* mq = nm->mask + (size_t)q * n;
    double* cw = cols.data() + (size_t)q * n;
    double* cr = cols.data() + (size_t)(P + P * C + q) * n;
    double* cm = cols.data() + (size_t)(P + P * C + P + q) * n;
    std::fill(A.begin(), A.end(), 0.0);
    std::vector<double> xr(C, 0.0);
    double ms = 0.0;
    for (int64_t k = 0; k < n; ++k) {
      const double m = mq[k] ? 1.0 : 0.0;
      const double w = (bt.family == 1 ? fq[k] : fq[k] * (1.0 - fq[k])) * m;      // Gamma_sqrt^2 on the unmasked samples
      cw[k] = w; cr[k] = (yq[k] - fq[k]) * m; cm[k] = m; ms += m;
      if (!bt.pass[q]) continue;
      for (int a = 0; a < C; ++a) {
        const double xa = nm->X[(size_t)a * n + k] * w;
        cols[(size_t)(P + q * C + a) * n + k] = xa;
        xr[a] += nm->X[(size_t)a * n + k] * cr[k];
        for (int c = 0; c <= a; ++c) A[(size_t)a * C + c] += xa * nm->X[(size_t)c * n + k];
      }
    }
    bt.msum[q] = ms;
    { double rs = 0.0; for (int64_t k = 0; k < n; ++k) rs += cr[k]; bt.rsum[q] = rs; }
    if (!bt.pass[q]) continue;
    for (int a = 0; a < C; ++a) for (int c = a + 1; c < C; ++c) A[(size_t)a * C + c] = A[(size_t)c * C + a];
    if (!small_inverse(A, C, inv)) return rg_s2_fail(ctx, RG_S2_ERR_ARG, "rg_s2_bt_set_null: X'WX is singular in the null model of trait " + std::to_string(q + 1));
    std::copy(inv.begin(), inv.end(), bt.xwx_inv.begin() + (size_t)q * C * C);
    // the dense form's numerator is Gres . yres with Gres = the genotype projected off the covariates and yres NOT projected (Step2_Models.cpp:503, :519;
    // Data.cpp:2443-2445) = g~ . r - (X^T W g~)^T (X^T W X)^-1 X^T r, r = (y - p^) mask: the second term is the null model's score at its stopping point
    // (|score| < 1e-6) -- tiny, not zero, and absent from the sparse form (score_from_sums)
    for (int a = 0; a < C; ++a) {
      double t = 0.0;
      for (int c = 0; c < C; ++c) t += inv[(size_t)a * C + c] * xr[c];
      bt.xres[(size_t)q * C + a] = t;
    }
  }
  int rc = rg_s2_set_columns(ctx, bt.ncol, cols.data(), P);
  if (rc) return rc;
  // device copies for the corrections
  if (!bt.dX) S2_HIP(hipMalloc((void**)&bt.dX, sizeof(double) * n * C));
  if (!bt.dY) S2_HIP(hipMalloc((void**)&bt.dY, sizeof(double) * n * P));
  if (!bt.dFit) S2_HIP(hipMalloc((void**)&bt.dFit, sizeof(double) * n * P));
  if (!bt.dM) S2_HIP(hipMalloc((void**)&bt.dM, (size_t)n * P));
  S2_HIP(hipMemcpyAsync(bt.dX, nm->X, sizeof(double) * n * C, hipMemcpyHostToDevice, ctx->st));
  S2_HIP(hipMemcpyAsync(bt.dY, nm->y, sizeof(double) * n * P, hipMemcpyHostToDevice, ctx->st));
  S2_HIP(hipMemcpyAsync(bt.dFit, nm->fitted, sizeof(double) * n * P, hipMemcpyHostToDevice, ctx->st));
  S2_HIP(hipMemcpyAsync(bt.dM, nm->mask, (size_t)n * P, hipMemcpyHostToDevice, ctx->st));
  bt.have_firth = nm->firth_offset != nullptr;
  if (bt.have_firth) {
    if (!bt.dFo) S2_HIP(hipMalloc((void**)&bt.dFo, sizeof(double) * n * P));
    S2_HIP(hipMemcpyAsync(bt.dFo, nm->firth_offset, sizeof(double) * n * P, hipMemcpyHostToDevice, ctx->st));
  }
  S2_HIP(hipStreamSynchronize(ctx->st));
  bt.have_null = true;
  bt.bs = 0; bt.kind = 0;
  return RG_S2_OK;
}

int rg_s2_bt_score_packed(rg_s2_ctx* ctx, const uint8_t* rows, int64_t ld, int32_t bs, int32_t rows_on_device, int32_t flip, double numtol,
                          const rg_s2_bt_out* out) {
  if (!ctx || !ctx->bt || !ctx->bt->have_null) return rg_s2_fail(ctx, RG_S2_ERR_ARG, "rg_s2_bt_score_packed: rg_s2_bt_set_null has not been called");
  if (!out) return rg_s2_fail(ctx, RG_S2_ERR_ARG, "rg_s2_bt_score_packed: null output");
  BtState& bt = *ctx->bt;
  const int P = ctx->P;
  bt.sums.resize((size_t)bs * 2 * bt.ncol); bt.sq.resize((size_t)bs * P);
  std::vector<int32_t> counts((size_t)bs * 4);
  rg_s2_contract_out co;
  memset(&co, 0, sizeof(co));
  co.sums = bt.sums.data(); co.sq = bt.sq.data(); co.counts = counts.data();
  const int rc = rg_s2_contract_packed(ctx, rows, ld, bs, rows_on_device, flip, &co);
  if (rc) return rc;
  bt.bs = bs; bt.kind = 1; bt.scale = 0;
  const int64_t Np = (ctx->n + 128 * RG_MAX_SEG - 1) / (128 * RG_MAX_SEG) * (128 * RG_MAX_SEG);
  bt.d_pk = (const uint8_t*)ctx->pbuf[RG_S2_Q_PK]; bt.ldp = Np / 4;       // the staged rows (flip applied, padding = 0 copies) stay for the corrections
  bt.d_g16 = nullptr; bt.ldg = 0;
  if (out->counts) memcpy(out->counts, counts.data(), sizeof(int32_t) * counts.size());
  return score_from_sums(ctx, bs, numtol, counts.data(), nullptr, nullptr, 1, out);
}

int rg_s2_bt_score_int(rg_s2_ctx* ctx, const uint16_t* G, int64_t ld, int32_t bs, int32_t g_on_device, int32_t scale, double numtol,
                       const rg_s2_bt_out* out) {
  if (!ctx || !ctx->bt || !ctx->bt->have_null) return rg_s2_fail(ctx, RG_S2_ERR_ARG, "rg_s2_bt_score_int: rg_s2_bt_set_null has not been called");
  if (!out) return rg_s2_fail(ctx, RG_S2_ERR_ARG, "rg_s2_bt_score_int: null output");
  BtState& bt = *ctx->bt;
  const int P = ctx->P;
  bt.sums.resize((size_t)bs * 2 * bt.ncol); bt.sq.resize((size_t)bs * P);
  std::vector<double> vstat((size_t)bs * 4);
  rg_s2_contract_out co;
  memset(&co, 0, sizeof(co));
  co.sums = bt.sums.data(); co.sq = bt.sq.data(); co.vstat = vstat.data();
  const int rc = rg_s2_contract_int(ctx, G, ld, bs, g_on_device, scale, &co);
  if (rc) return rc;
  bt.bs = bs; bt.kind = 2; bt.scale = scale;
  bt.d_pk = nullptr; bt.ldp = 0;
  if (g_on_device) { bt.d_g16 = G; bt.ldg = ld; }
  else { bt.d_g16 = (const uint16_t*)ctx->buf[RG_S2_B_G]; bt.ldg = (ctx->n + 7) / 8 * 8; }
  if (out->vstat) memcpy(out->vstat, vstat.data(), sizeof(double) * vstat.size());
  // the zeros of the flipped coding (flip_geno): entries that are exactly 2 copies, counted on the staged rows
  std::vector<int32_t> n_two(bs, 0);
  int rc2;
  if ((rc2 = grow(ctx, (void**)&bt.d_two, &bt.two_cap, sizeof(int32_t) * (size_t)bs))) return rc2;
  hipLaunchKernelGGL(k_bt_count_two, dim3(bs), dim3(256), 0, ctx->st, bt.d_g16, bt.ldg, ctx->n, (unsigned)(2 * scale), bt.d_two);
  S2_HIP(hipGetLastError());
  S2_HIP(hipMemcpyAsync(n_two.data(), bt.d_two, sizeof(int32_t) * (size_t)bs, hipMemcpyDeviceToHost, ctx->st));
  S2_HIP(hipStreamSynchronize(ctx->st));
  return score_from_sums(ctx, bs, numtol, nullptr, vstat.data(), n_two.data(), scale, out);
}

int rg_s2_bt_correct(rg_s2_ctx* ctx, int32_t kind, int32_t npair, const int32_t* variant, const int32_t* trait, const uint8_t* fast, int32_t firth_se,
                     rg_s2_bt_corr* out) {
  if (!ctx || !ctx->bt || !ctx->bt->have_null || ctx->bt->bs < 1)
    return rg_s2_fail(ctx, RG_S2_ERR_ARG, "rg_s2_bt_correct: no block has been scored (rg_s2_bt_score_packed / rg_s2_bt_score_int)");
  BtState& bt = *ctx->bt;
  if (kind != RG_S2_BT_FIRTH_APPROX && kind != RG_S2_BT_SPA) return rg_s2_fail(ctx, RG_S2_ERR_ARG, "rg_s2_bt_correct: kind must be RG_S2_BT_FIRTH_APPROX or RG_S2_BT_SPA");
  if (bt.family != 0) return rg_s2_fail(ctx, RG_S2_ERR_ARG, "rg_s2_bt_correct: the corrections are those of the binary-trait test");
  if (kind == RG_S2_BT_FIRTH_APPROX && !bt.have_firth) return rg_s2_fail(ctx, RG_S2_ERR_ARG, "rg_s2_bt_correct: the null model was set without firth_offset");
  if (npair < 0 || (npair > 0 && (!variant || !trait || !out))) return rg_s2_fail(ctx, RG_S2_ERR_ARG, "rg_s2_bt_correct: null argument");
  if (npair == 0) return RG_S2_OK;
  S2_HIP(hipSetDevice(ctx->dev));
  const int64_t n = ctx->n;
  const int P = ctx->P, C = ctx->C, ncol = bt.ncol;
  std::vector<BtPairDev> pairs(npair);
  for (int t = 0; t < npair; ++t) {
    const int j = variant[t], q = trait[t];
    if (j < 0 || j >= bt.bs || q < 0 || q >= P) return rg_s2_fail(ctx, RG_S2_ERR_ARG, "rg_s2_bt_correct: pair out of range");
    BtPairDev& pd = pairs[t];
    memset(&pd, 0, sizeof(pd));
    pd.variant = j; pd.trait = q; pd.fast = fast && fast[t] ? 1 : 0; pd.flip = bt.flip[j];
    pd.mu = bt.mu[j]; pd.stats = bt.stats[(size_t)j * P + q]; pd.denum = bt.denum[(size_t)j * P + q];
    const double* s0 = bt.sums.data() + (size_t)j * 2 * ncol;
    const double* s1 = s0 + ncol;
    const double* inv = bt.xwx_inv.data() + (size_t)q * C * C;
    for (int a = 0; a < C; ++a) {
      double tc = 0.0;
      for (int c = 0; c < C; ++c) tc += inv[(size_t)a * C + c] * (s0[P + q * C + c] + pd.mu * s1[P + q * C + c]);
      pd.tc[a] = tc;
    }
  }
  // pairs in batches whose residual rows fit 2 GB
  const int per = (int)std::max<int64_t>(1, std::min<int64_t>(npair, (int64_t)2000000000 / (8 * n)));
  int rc;
  if ((rc = grow(ctx, (void**)&bt.d_v, &bt.v_cap, sizeof(double) * (size_t)per * n))) return rc;
  if ((rc = grow(ctx, &bt.d_pairs, &bt.pairs_cap, sizeof(BtPairDev) * (size_t)per))) return rc;
  if ((rc = grow(ctx, (void**)&bt.d_res, &bt.res_cap, sizeof(double) * (size_t)per * 12))) return rc;
  double* d_aux = bt.d_res + (size_t)per * 4;
  std::vector<double> hres((size_t)per * 4);
  const double inv_scale = bt.scale > 0 ? 1.0 / (double)bt.scale : 1.0;
  float ms_total = 0.f;
  for (int t0 = 0; t0 < npair; t0 += per) {
    const int np = std::min(per, npair - t0);
    S2_HIP(hipMemcpyAsync(bt.d_pairs, pairs.data() + t0, sizeof(BtPairDev) * np, hipMemcpyHostToDevice, ctx->st));
    S2_HIP(hipEventRecord(ctx->e0, ctx->st));
    hipLaunchKernelGGL(k_bt_prep, dim3(np), dim3(256), 0, ctx->st, (const BtPairDev*)bt.d_pairs, kind, bt.kind, bt.d_pk, bt.ldp, bt.d_g16, bt.ldg, inv_scale,
                       (unsigned)(2 * std::max(bt.scale, 0)), (const double*)bt.dX, (const uint8_t*)bt.dM, (const double*)bt.dFit, n, C, bt.d_v, d_aux);
    if (kind == RG_S2_BT_FIRTH_APPROX)
      hipLaunchKernelGGL(k_bt_firth1, dim3(np), dim3(256), 0, ctx->st, (const BtPairDev*)bt.d_pairs, (const double*)bt.d_v, (const double*)bt.dY,
                         (const double*)bt.dFo, (const double*)d_aux, n, bt.d_res, bt.niter_max);
    else
      hipLaunchKernelGGL(k_bt_spa, dim3(np), dim3(256), 0, ctx->st, (const BtPairDev*)bt.d_pairs, (const double*)bt.d_v, (const double*)bt.dFit,
                         (const double*)d_aux, n, bt.d_res);
    S2_HIP(hipEventRecord(ctx->e1, ctx->st));
    S2_HIP(hipGetLastError());
    S2_HIP(hipMemcpyAsync(hres.data(), bt.d_res, sizeof(double) * (size_t)np * 4, hipMemcpyDeviceToHost, ctx->st));
    S2_HIP(hipStreamSynchronize(ctx->st));
    float ms = 0.f;
    S2_HIP(hipEventElapsedTime(&ms, ctx->e0, ctx->e1));
    ms_total += ms;
    for (int t = 0; t < np; ++t) {
      rg_s2_bt_corr& o = out[t0 + t];
      const double* r = hres.data() + (size_t)t * 4;
      memset(&o, 0, sizeof(o));
      o.logp = -1.0;
      if (r[3] != 0.0) { o.fail = 1; continue; }
      const BtPairDev& pd = pairs[t0 + t];
      if (kind == RG_S2_BT_FIRTH_APPROX) {
        o.beta = r[0]; o.chisq = r[2];
        o.se = (firth_se && r[2] > 0) ? std::fabs(r[0]) / std::sqrt(r[2]) : r[1];                  // --firth-se: back_correct_se (Step2_Models.cpp:2008-2009)
      } else {   // check_pval_snp (Step2_Models.cpp:2012-2020): chi-square from the p-value, SE of the score test, the sign of the score
        const double pval = std::max(10.0 * std::numeric_limits<double>::min(), r[0]);             // get_logp(pv, ...) (Regenie.cpp:1859-1873)
        const double z = norm_quantile(0.5 * pval);
        o.chisq = z * z;
        o.se = 1.0 / std::sqrt(pd.denum);
        o.beta = (pd.stats > 0 ? 1.0 : pd.stats < 0 ? -1.0 : 0.0) * std::sqrt(o.chisq) * o.se;
        o.logp = -std::log10(pval);
      }
    }
  }
  ctx->last_ms = ms_total;
  return RG_S2_OK;
}

}  // extern "C"
