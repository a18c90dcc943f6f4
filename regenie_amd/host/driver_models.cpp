// regenie-amd, the C++ host driver (see driver.h): covariate-only null models and host-side statistics.
#include "driver.h"

namespace rgdrv {

// ---- binary traits: covariate-only logistic regression (Step1_Models.cpp:54-222) ---------------------------
double get_pvec1(double eta) {                                // Step1_Models.cpp:1799-1806
  double pr = 1.0 - 1.0 / (std::exp(eta) + 1.0);
  if (eta < -30.0) pr = NUMTOL_EPS / (1.0 + NUMTOL_EPS);
  if (eta > 30.0) pr = 1.0 / (1.0 + NUMTOL_EPS);
  return pr;
}
// dense solve by Gaussian elimination with partial pivoting (order = number of covariates)
bool solve_dense(std::vector<double> A, std::vector<double> b, int n, std::vector<double>& x) {
  for (int k = 0; k < n; ++k) {
    int piv = k;
    for (int i = k + 1; i < n; ++i) if (std::fabs(A[(size_t)i * n + k]) > std::fabs(A[(size_t)piv * n + k])) piv = i;
    if (A[(size_t)piv * n + k] == 0.0) return false;
    if (piv != k) { for (int j = 0; j < n; ++j) std::swap(A[(size_t)k * n + j], A[(size_t)piv * n + j]); std::swap(b[k], b[piv]); }
    for (int i = k + 1; i < n; ++i) {
      const double f = A[(size_t)i * n + k] / A[(size_t)k * n + k];
      for (int j = k; j < n; ++j) A[(size_t)i * n + j] -= f * A[(size_t)k * n + j];
      b[i] -= f * b[k];
    }
  }
  x.assign(n, 0.0);
  for (int i = n - 1; i >= 0; --i) {
    double v = b[i];
    for (int j = i + 1; j < n; ++j) v -= A[(size_t)i * n + j] * x[j];
    x[i] = v / A[(size_t)i * n + i];
  }
  return true;
}
// fit_logistic (Step1_Models.cpp:156-222) for one phenotype; offset may be null (zero); eta_out = offset + X beta on success,
// pv_out (optional) the fitted probabilities
// `resume`: the reference calls fit_logistic twice -- with and then without the deviance test of the step halving -- on the SAME pivec / etavec /
// betavec (Step1_Models.cpp:88: `fit_logistic(.., true, ..) || fit_logistic(.., false, ..)`, all three passed by reference): the second attempt goes
// on from wherever the first one stopped (with --niter 2 that is four Newton steps; found by tests/golden/fuzz_oracle_vs_reference.py).  resume != nullptr
// holds that state: empty vectors = start at beta = 0, and in every case the state the attempt leaves, converged or not.
bool fit_logistic(const double* y, const double* X, const uint8_t* mask, int64_t N, int C, const Params& prm,
                  bool check_hs_dev, std::vector<double>& eta, const double* offset, std::vector<double>* pv_out, std::vector<double>* beta_out, LogisticState* resume) {
  std::vector<double> beta(C, 0.0), betanew(C, 0.0), pv(N), w(N);
  auto dev = [&](const std::vector<double>& pp) {
    double t = 0.0;
    for (int64_t i = 0; i < N; ++i) if (mask[i]) t -= (y[i] == 0.0) ? std::log(1.0 - pp[i]) : std::log(pp[i]);
    return 2.0 * t;
  };
  if (resume && (int64_t)resume->pv.size() == N && (int)resume->beta.size() == C) { beta = resume->beta; betanew = beta; pv = resume->pv; eta = resume->eta; }
  else {
    eta.assign(N, 0.0);
    for (int64_t i = 0; i < N; ++i) { eta[i] = offset ? offset[i] : 0.0; pv[i] = get_pvec1(eta[i]); }
  }
  struct Leave { LogisticState* st; std::vector<double>&b, &p, &e; ~Leave() { if (st) { st->beta = b; st->pv = p; st->eta = e; } } } leave{resume, beta, pv, eta};
  double dev_old = dev(pv), dev_new = dev_old, diff_dev = 0.0;
  int niter = 0;
  bool small_score = false;
  while (true) {
    if (++niter > prm.niter_max) break;
    for (int64_t i = 0; i < N; ++i) { w[i] = mask[i] ? pv[i] * (1.0 - pv[i]) : 1.0; if (w[i] == 0.0) return false; }
    std::vector<double> A((size_t)C * C, 0.0), b(C, 0.0);
    for (int64_t i = 0; i < N; ++i) {
      if (!mask[i]) continue;
      const double z = eta[i] - (offset ? offset[i] : 0.0) + (y[i] - pv[i]) / w[i];
      for (int a = 0; a < C; ++a) {
        const double xa = X[(size_t)a * N + i] * w[i];
        b[a] += xa * z;
        for (int c = 0; c < C; ++c) A[(size_t)a * C + c] += xa * X[(size_t)c * N + i];
      }
    }
    if (!solve_dense(A, b, C, betanew)) return false;
    bool ok_search = false;
    for (int ls = 0; ls < prm.niter_max_line_search; ++ls) {
      bool inside = true;
      for (int64_t i = 0; i < N; ++i) {
        double e = offset ? offset[i] : 0.0;
        for (int a = 0; a < C; ++a) e += X[(size_t)a * N + i] * betanew[a];
        eta[i] = e;
        pv[i] = get_pvec1(e);
        if (mask[i] && !(pv[i] > 0.0 && pv[i] < 1.0)) inside = false;
      }
      dev_new = dev(pv);
      if (inside && (!check_hs_dev || dev_new < dev_old)) { ok_search = true; break; }
      for (int a = 0; a < C; ++a) betanew[a] = (beta[a] + betanew[a]) / 2;
    }
    if (!ok_search) return false;
    double smax = 0.0;
    for (int a = 0; a < C; ++a) {
      double sc = 0.0;
      for (int64_t i = 0; i < N; ++i) if (mask[i]) sc += X[(size_t)a * N + i] * (y[i] - pv[i]);
      smax = std::max(smax, std::fabs(sc));
    }
    if (smax < NUMTOL) break;
    if (!small_score && niter < 20 && smax < 1) small_score = true;
    if (small_score && niter > 20 && smax > 5) return false;
    diff_dev = std::fabs(dev_new - dev_old) / (0.1 + std::fabs(dev_new));
    beta = betanew;
    dev_old = dev_new;
  }
  if ((diff_dev == 0 || diff_dev >= NUMTOL) && niter > prm.niter_max) return false;
  if (pv_out) *pv_out = pv;
  if (beta_out) *beta_out = betanew;
  return true;
}

// Standard normal quantile (what boost::math::quantile(normal(0,1), p) returns in rint_pheno, Pheno.cpp:2002-2008):
// Wichura's algorithm AS 241 (PPND16), relative accuracy about 1e-16.
double norm_quantile(double p) {
  const double q = p - 0.5;
  if (std::fabs(q) <= 0.425) {
    const double r = 0.180625 - q * q;
    const double num = (((((((2.5090809287301226727e3 * r + 3.3430575583588128105e4) * r + 6.7265770927008700853e4) * r + 4.5921953931549871457e4) * r +
                           1.3731693765509461125e4) * r + 1.9715909503065514427e3) * r + 1.3314166789178437745e2) * r + 3.3871328727963666080e0);
    const double den = (((((((5.2264952788528545610e3 * r + 2.8729085735721942674e4) * r + 3.9307895800092710610e4) * r + 2.1213794301586595867e4) * r +
                           5.3941960214247511077e3) * r + 6.8718700749205790830e2) * r + 4.2313330701600911252e1) * r + 1.0);
    return q * num / den;
  }
  double r = q < 0 ? p : 1.0 - p;
  r = std::sqrt(-std::log(r));
  double v;
  if (r <= 5.0) {
    r -= 1.6;
    const double num = (((((((7.74545014278341407640e-4 * r + 2.27238449892691845833e-2) * r + 2.41780725177450611770e-1) * r + 1.27045825245236838258e0) * r +
                           3.64784832476320460504e0) * r + 5.76949722146069140550e0) * r + 4.63033784615654529590e0) * r + 1.42343711074968357734e0);
    const double den = (((((((1.05075007164441684324e-9 * r + 5.47593808499534494600e-4) * r + 1.51986665636164571966e-2) * r + 1.48103976427480074590e-1) * r +
                           6.89767334985100004550e-1) * r + 1.67638483018380384940e0) * r + 2.05319162663775882187e0) * r + 1.0);
    v = num / den;
  } else {
    r -= 5.0;
    const double num = (((((((2.01033439929228813265e-7 * r + 2.71155556874348757815e-5) * r + 1.24266094738807843860e-3) * r + 2.65321895265761230930e-2) * r +
                           2.96560571828504891230e-1) * r + 1.78482653991729133580e0) * r + 5.46378491116411436990e0) * r + 6.65790464350110377720e0);
    const double den = (((((((2.04426310338993978564e-15 * r + 1.42151175831644588870e-7) * r + 1.84631831751005468180e-5) * r + 7.86869131145613259100e-4) * r +
                           1.48753612908506148525e-2) * r + 1.36929880922735805310e-1) * r + 5.99832206555887937690e-1) * r + 1.0);
    v = num / den;
  }
  return q < 0 ? -v : v;
}

// ---- time-to-event traits: the null Cox model of step 1 (fit_null_cox, Step1_Models.cpp:353-440) ------------------------------------------
// cox_ridge with lambda = 0 on the covariates (cox_ridge.cpp:8-178; survival_data::setup, survival_data.cpp:9-100): IRLS on the diagonal of
// the Hessian, one cyclic pass over the C coordinates per iteration, step halving on the deviance.  X: col-major N x C.  eta = X beta on
// the unmasked samples, 0 elsewhere.  (Level 1 -- the same model on the thousands of level-0 predictors -- runs in the library: rg_l1_cox.)
bool cox_null_fit(const double* time, const double* event, const uint8_t* mask, const double* X, int64_t N, int C, const Params& prm, std::vector<double>& eta) {
  const int64_t n = N;
  double neff = 0;
  for (int64_t i = 0; i < n; ++i) neff += mask[i];
  const double w = 1.0 / neff;
  std::vector<int64_t> ord(n);
  for (int64_t i = 0; i < n; ++i) ord[i] = i;
  auto st = [&](int64_t i) { return mask[i] ? event[i] : -999.0; };
  std::stable_sort(ord.begin(), ord.end(), [&](int64_t a, int64_t b) { return time[a] != time[b] ? time[a] < time[b] : st(a) > st(b); });
  std::vector<uint8_t> keep(n), dd(n, 0), ev1(n, 0);
  std::vector<double> ww(n, 0.0), wsub;
  std::vector<int64_t> evs;
  for (int64_t i = 0; i < n; ++i) {
    const int64_t s = ord[i];
    keep[i] = mask[s];
    if (mask[s] && event[s] == 1.0) { ev1[i] = dd[i] = 1; ww[i] = w; evs.push_back(i); }
  }
  for (size_t a = 0; a < evs.size();) {
    size_t b = a + 1;
    while (b < evs.size() && time[ord[evs[b]]] == time[ord[evs[a]]]) ++b;
    if (b - a > 1) { for (size_t t = a + 1; t < b; ++t) { dd[evs[t]] = 0; ww[evs[t]] = 0.0; } ww[evs[a]] = (double)(b - a) * w; }
    wsub.push_back((double)(b - a) * w);
    a = b;
  }
  double lsat = 0;
  for (double x : wsub) lsat -= x * std::log(x);
  std::vector<double> beta(C, 0.0), beta_old(C), g(n), h(n), z(n), rsk(n);
  eta.assign(n, 0.0);
  auto deviance = [&]() {
    double run = 0, ll = 0;
    for (int64_t i = n - 1; i >= 0; --i) {
      const double e = eta[ord[i]];
      if (keep[i]) run += w * std::exp(e);
      if (keep[i] && ev1[i]) ll += w * e;
      if (keep[i] && dd[i]) ll -= ww[i] * std::log(run);
    }
    return 2.0 * (lsat - ll);
  };
  auto grad = [&]() {      // coxGrad (cox_ridge.cpp:60-82): g, h in sample order
    double mean = 0;
    for (int64_t i = 0; i < n; ++i) if (mask[i]) mean += eta[i];
    mean *= w;
    double run = 0;
    for (int64_t i = n - 1; i >= 0; --i) { if (keep[i]) run += w * std::exp(eta[ord[i]] - mean); rsk[i] = run; }
    double A = 0, B = 0;
    for (int64_t i = 0; i < n; ++i) {
      if (keep[i] && dd[i]) { A += ww[i] / rsk[i]; B += ww[i] / (rsk[i] * rsk[i]); }
      const int64_t s = ord[i];
      if (keep[i]) { const double we = w * std::exp(eta[s] - mean); g[s] = w * (ev1[i] ? 1.0 : 0.0) - we * A; h[s] = we * we * B - we * A; }
      else { g[s] = 0; h[s] = 0; }
    }
  };
  auto set_eta = [&]() {
    for (int64_t i = 0; i < n; ++i) {
      double e = 0;
      if (mask[i]) for (int c = 0; c < C; ++c) e += X[(size_t)c * N + i] * beta[c];
      eta[i] = e;
    }
  };
  const double tol = 2.5e-4;        // numtol_cox, Regenie.hpp:221
  double dev_prev = deviance(), obj_prev = dev_prev;
  for (int t = 1; t <= prm.niter_max; ++t) {
    beta_old = beta;
    grad();
    for (int64_t i = 0; i < n; ++i) z[i] = (mask[i] ? eta[i] : 0.0) - (h[i] != 0 ? g[i] / h[i] : 0.0);
    for (int k = 0; k < C; ++k) {
      const double* xk = X + (size_t)k * N;
      double rx = 0, s2 = 0;
      for (int64_t i = 0; i < n; ++i) { rx += h[i] * (z[i] - eta[i]) * xk[i]; s2 += xk[i] * xk[i] * h[i]; }
      const double b1 = (rx + beta[k] * s2) / s2;                 // lambda = 0
      for (int64_t i = 0; i < n; ++i) if (mask[i]) eta[i] += xk[i] * (b1 - beta[k]);
      beta[k] = b1;
    }
    double dev = deviance(), obj = dev;
    if (dev - dev_prev > tol) {
      int ii = 0;
      while (dev - dev_prev > tol) {
        if (++ii > prm.niter_max_line_search) return false;
        for (int c = 0; c < C; ++c) beta[c] = (beta[c] + beta_old[c]) / 2;
        set_eta();
        dev = obj = deviance();
      }
    }
    double score = 0;
    for (int k = 0; k < C; ++k) {
      double sx = 0;
      for (int64_t i = 0; i < n; ++i) sx += g[i] * X[(size_t)k * N + i];
      score = std::max(score, std::fabs(sx));
    }
    const bool stop = std::fabs(obj - obj_prev) / (0.1 + std::fabs(obj)) < tol || score < tol;
    dev_prev = dev; obj_prev = obj;
    if (stop) return true;
  }
  return false;
}

// The reference's fall-back when the coordinate descent above does not converge (fit_null_cox, Step1_Models.cpp:415-436): the Newton
// solver of cox_firth.cpp without the Firth term (cox_firth::fit, :137-219, likelihood and derivatives :43-135) from beta = 0 -- full
// Hessian sum_k ww_k (S2 / S0 - S1 S1^T / S0^2) over the risk sets of the distinct event times, steps clamped to maxstep_null (25), step
// halving while the log-likelihood drops by more than `stephalf_tol` -- tried with stephalf_tol = numtol_cox_stephalf (2.5e-4) and, if
// that fails, with 0.  Converged: max |score| < numtol_cox, or a full step that moves no coefficient by 1e-8 (numtol_beta_cox).
bool cox_null_newton(const double* time, const double* event, const uint8_t* mask, const double* X, int64_t N, int C, const Params& prm, double stephalf_tol,
                     std::vector<double>& eta) {
  const int64_t n = N;
  double neff = 0;
  for (int64_t i = 0; i < n; ++i) neff += mask[i];
  const double w = 1.0 / neff;
  std::vector<int64_t> ord(n);
  for (int64_t i = 0; i < n; ++i) ord[i] = i;
  auto stt = [&](int64_t i) { return mask[i] ? event[i] : -999.0; };
  std::stable_sort(ord.begin(), ord.end(), [&](int64_t a, int64_t b) { return time[a] != time[b] ? time[a] < time[b] : stt(a) > stt(b); });
  std::vector<double> ww(n, 0.0);       // tie-collapsed event weight at the first event of every distinct event time (survival_data.cpp:40-60)
  {
    int64_t a = 0;
    while (a < n) {
      int64_t b = a;
      double cnt = 0;
      while (b < n && time[ord[b]] == time[ord[a]]) { if (mask[ord[b]] && event[ord[b]] == 1.0) cnt += 1.0; ++b; }
      if (cnt > 0) ww[a] = cnt * w;      // events sort first inside a time, so position a is an event
      a = b;
    }
  }
  std::vector<double> beta(C, 0.0), betanew(C, 0.0), score(C), H((size_t)C * C), S1(C), S2((size_t)C * C), we(n), lam0(n), steps;
  eta.assign(n, 0.0);
  double loglik = 0;
  auto likelihood = [&](const std::vector<double>& b) {     // eta, loglik, score = X^T residual, H = -second derivative (positive definite)
    for (int64_t i = 0; i < n; ++i) {
      double e = 0;
      if (mask[i]) for (int c = 0; c < C; ++c) e += X[(size_t)c * N + i] * b[c];
      eta[i] = e;
    }
    std::fill(H.begin(), H.end(), 0.0); std::fill(S1.begin(), S1.end(), 0.0); std::fill(S2.begin(), S2.end(), 0.0);
    double S0 = 0, ll = 0;
    for (int64_t i = n - 1; i >= 0; --i) {
      const int64_t s = ord[i];
      we[i] = mask[s] ? w * std::exp(eta[s]) : 0.0;
      if (mask[s]) {
        S0 += we[i];
        for (int a = 0; a < C; ++a) {
          const double xa = X[(size_t)a * N + s] * we[i];
          S1[a] += xa;
          for (int c2 = 0; c2 <= a; ++c2) S2[(size_t)a * C + c2] += xa * X[(size_t)c2 * N + s];
        }
        if (event[s] == 1.0) ll += w * eta[s];
      }
      lam0[i] = S0;
      if (ww[i] > 0) {
        ll -= ww[i] * std::log(S0);
        for (int a = 0; a < C; ++a)
          for (int c2 = 0; c2 <= a; ++c2) H[(size_t)a * C + c2] += ww[i] * (S2[(size_t)a * C + c2] / S0 - S1[a] * S1[c2] / (S0 * S0));
      }
    }
    for (int a = 0; a < C; ++a) for (int c2 = a + 1; c2 < C; ++c2) H[(size_t)a * C + c2] = H[(size_t)c2 * C + a];
    std::fill(score.begin(), score.end(), 0.0);
    double A = 0;
    for (int64_t i = 0; i < n; ++i) {     // cumulative hazard in time order, residual = w (status - mu)
      if (ww[i] > 0) A += ww[i] / lam0[i];
      const int64_t s = ord[i];
      if (!mask[s]) continue;
      const double res = w * (event[s] == 1.0 ? 1.0 : 0.0) - we[i] * A;
      for (int a = 0; a < C; ++a) score[a] += X[(size_t)a * N + s] * res;
    }
    loglik = ll;
  };
  const double tol = 2.5e-4, betatol = 1e-8, maxstep = 25.0;    // numtol_cox, numtol_beta_cox, maxstep_null (Regenie.hpp:221-223, :340)
  likelihood(beta);
  double ll_prev = loglik;
  for (int it = 1; it <= prm.niter_max; ++it) {
    if (!solve_dense(H, score, C, steps)) return false;
    for (double& v : steps) if (std::fabs(v) >= maxstep) v = v > 0 ? maxstep : -maxstep;
    for (int c = 0; c < C; ++c) betanew[c] = beta[c] + steps[c];
    likelihood(betanew);
    int ii = 0;
    while (ll_prev - loglik > stephalf_tol) {
      if (++ii > prm.niter_max_line_search) {       // "cannot correct step size, add eps" (:186-194)
        for (int c = 0; c < C; ++c) betanew[c] = beta[c] + steps[c] + 1e-6;
        likelihood(betanew);
        break;
      }
      for (int c = 0; c < C; ++c) betanew[c] = (beta[c] + betanew[c]) / 2;
      likelihood(betanew);
    }
    double smax = 0, dmax = 0;
    for (int c = 0; c < C; ++c) { smax = std::max(smax, std::fabs(score[c])); dmax = std::max(dmax, std::fabs(beta[c] - betanew[c])); }
    beta = betanew;
    ll_prev = loglik;
    if (smax < tol || (ii <= 1 && dmax < betatol)) return true;     // eta is the linear predictor at beta
  }
  return false;
}

// fit_null_poisson + fit_poisson (Step1_Models.cpp:225-345) for one phenotype; offset may be null (zero); eta_out = offset + X beta on
// success, pv_out (optional) the fitted rates
bool fit_poisson(const double* y, const double* X, const uint8_t* mask, int64_t N, int C, const Params& prm, std::vector<double>& eta,
                 const double* offset, std::vector<double>* pv_out) {
  std::vector<double> beta(C, 0.0), betanew(C, 0.0), pv(N);
  auto dev = [&](const std::vector<double>& pp) {
    double t = 0.0;
    for (int64_t i = 0; i < N; ++i) if (mask[i]) t -= y[i] * std::log(pp[i]) - pp[i];
    return 2.0 * t;
  };
  auto any_zero = [&]() { for (int64_t i = 0; i < N; ++i) if (mask[i] && pv[i] == 0.0) return true; return false; };
  eta.assign(N, 0.0);
  double esum = 0.0;
  for (int64_t i = 0; i < N; ++i) {  // starting values: p = y + 0.1, eta = log p on the analysed samples, intercept = mean(eta)
    pv[i] = y[i] + 1e-1;
    eta[i] = mask[i] ? std::log(pv[i]) : 0.0;
    esum += eta[i];
  }
  beta[0] = esum / (double)N;
  if (offset) { double osum = 0.0; for (int64_t i = 0; i < N; ++i) osum += offset[i]; beta[0] -= osum / (double)N; }   // Step1_Models.cpp:247
  double dev_old = dev(pv), dev_new = dev_old;
  int niter = 0;
  bool dev_conv = false;
  while (true) {
    if (++niter > prm.niter_max) break;
    if (any_zero()) return false;
    std::vector<double> A((size_t)C * C, 0.0), b(C, 0.0);
    for (int64_t i = 0; i < N; ++i) {
      if (!mask[i]) continue;
      const double z = eta[i] - (offset ? offset[i] : 0.0) + (y[i] - pv[i]) / pv[i];
      for (int a = 0; a < C; ++a) {
        const double xa = X[(size_t)a * N + i] * pv[i];
        b[a] += xa * z;
        for (int c = 0; c < C; ++c) A[(size_t)a * C + c] += xa * X[(size_t)c * N + i];
      }
    }
    if (!solve_dense(A, b, C, betanew)) return false;
    for (int ls = 0; ls < prm.niter_max_line_search; ++ls) {
      for (int64_t i = 0; i < N; ++i) {
        double e = offset ? offset[i] : 0.0;
        for (int a = 0; a < C; ++a) e += X[(size_t)a * N + i] * betanew[a];
        eta[i] = e;
        pv[i] = std::exp(e);
      }
      dev_new = dev(pv);
      if (!any_zero()) break;
      for (int a = 0; a < C; ++a) betanew[a] = (beta[a] + betanew[a]) / 2;
    }
    double smax = 0.0;
    for (int a = 0; a < C; ++a) {
      double sc = 0.0;
      for (int64_t i = 0; i < N; ++i) if (mask[i]) sc += X[(size_t)a * N + i] * (y[i] - pv[i]);
      smax = std::max(smax, std::fabs(sc));
    }
    dev_conv = std::fabs(dev_new - dev_old) / (0.1 + std::fabs(dev_new)) < 1e-8;  // params->tol
    if (smax < 1e-8) break;
    beta = betanew;
    dev_old = dev_new;
  }
  if (!dev_conv && niter > prm.niter_max) return false;
  if (pv_out) *pv_out = pv;
  return true;
}

// -log10 p of a 1-df chi-square statistic (get_logp, Regenie.cpp:1843-1856)
double get_logp(double t) {
  if (t < 0 && std::fabs(t) < 1e-6) return 0.0;
  if (t < 0) return -1.0;
  const double pv = std::erfc(std::sqrt(t / 2.0));   // cdf(complement(chi_squared(1), t))
  const double lp = pv == 0 ? std::log10(2.0) - 0.5 * std::log10(2 * M_PI * t) - 0.5 * t * M_LOG10E : std::log10(pv);
  return -lp;
}

// ---- `--step 2`: single-variant additive tests (Data::test_snps_fast, Data.cpp:2230-2360) --------------------------------------------
// Host side: the LOCO reader (blup_read / blup_read_chr, Pheno.cpp:1241-1391, Step2_Models.cpp:51-140), per chromosome compute_res
// (Data.cpp:2386-2400) or the null logistic / Poisson (/ Firth) model of compute_res_bin / compute_res_count, the per-variant bookkeeping
// of parseSnpfromBed / parseSnpfromBGEN / readChunkFromPGENFileToG (allele counts, the MAC and INFO filters, allele frequencies, per-trait
// counts for samples with missing phenotypes) and the output lines (print_sum_stats_head / print_sum_stats_single, Step2_Models.cpp:
// 2410-2530).  Device side (include/rg_step2.h): every per-variant O(n) contraction -- the QT statistic whole (hard calls: 2-bit rows;
// dosages: uint16 rows; both on the i8 matrix cores), the sums the binary / count trait score tests are functions of.  Phenotypes may
// differ in their missing values: the library makes check_sparse_G's per-variant choice between the sparse and the dense branch of
// compute_score_qt.  The binary-trait score test and its approximate-Firth / saddlepoint corrections are library calls too (rg_s2_bt_*);
// the null models (C parameters) and the exact Firth test of flagged variants (C + 1 parameters) are fitted on the host.

// ---- approximate Firth correction of the binary-trait test (--firth --approx) ---------------------------------------------------------
// regenie reaches the maximisers below through a chain of solvers and fall-backs (fit_firth_nr, the pseudo-data IRLS of fit_firth_pseudo,
// step halving, restarts: Step2_Models.cpp:899-984, :1254-1737) that stop at |modified score| < 50 * numtol (null model) or < 2.5e-4 (per
// variant).  The penalised likelihood has one maximiser; here it is found to machine precision by Fisher scoring with step halving on
// the penalised deviance, which agrees with regenie's printed numbers to its stopping tolerance (1e-5 relative on BETA; the reference
// and its own golden file differ by as much).

// log |A| and A^-1 of a small SPD matrix (Cholesky); false when not positive definite
bool spd_logdet_inv(const std::vector<double>& A, int n, double& logdet, std::vector<double>* inv) {
  std::vector<double> L(A);
  logdet = 0.0;
  for (int j = 0; j < n; ++j) {
    double d = L[(size_t)j * n + j];
    for (int k = 0; k < j; ++k) d -= L[(size_t)j * n + k] * L[(size_t)j * n + k];
    if (!(d > 0.0)) return false;
    d = std::sqrt(d);
    L[(size_t)j * n + j] = d;
    logdet += 2.0 * std::log(d);
    for (int i = j + 1; i < n; ++i) {
      double v = L[(size_t)i * n + j];
      for (int k = 0; k < j; ++k) v -= L[(size_t)i * n + k] * L[(size_t)j * n + k];
      L[(size_t)i * n + j] = v / d;
    }
  }
  if (inv) {
    inv->assign((size_t)n * n, 0.0);
    std::vector<double> col(n);
    for (int c = 0; c < n; ++c) {   // solve L L^T x = e_c
      for (int i = 0; i < n; ++i) { double v = i == c ? 1.0 : 0.0; for (int k = 0; k < i; ++k) v -= L[(size_t)i * n + k] * col[k]; col[i] = v / L[(size_t)i * n + i]; }
      for (int i = n - 1; i >= 0; --i) { double v = col[i]; for (int k = i + 1; k < n; ++k) v -= L[(size_t)k * n + i] * col[k]; col[i] = v / L[(size_t)i * n + i]; }
      for (int i = 0; i < n; ++i) (*inv)[(size_t)i * n + c] = col[i];
    }
  }
  return true;
}

// fit_firth_nr with cols_incl = nfree (Step2_Models.cpp:1267-1385): maximise l(beta) + 0.5 log |X^T W X| over the FIRST nfree coefficients (the
// others stay where they start); penalty and hat diagonal always use every column.  cols: K column pointers (sample-fastest, n each).
// beta in: start, out: the maximiser; dev_out: the penalised deviance there; inv_out (optional): (X^T W X)^-1.  false = no convergence.
bool firth_fit_cols(const double* y, const std::vector<const double*>& cols, const uint8_t* mask, const double* offset, int64_t n, int nfree, double maxstep,
                    std::vector<double>& beta, double* dev_out, std::vector<double>* inv_out, double stop_tol) {
  const int K = (int)cols.size();
  std::vector<double> pv(n), w(n), A((size_t)K * K), Ainv, Afree((size_t)nfree * nfree), Afinv, score(nfree), step(K, 0.0), bnew(K), hx(K);
  // A sample masked for the trait is in neither the likelihood nor the score -- but it IS in X^T W X, with weight 1: the reference's get_wvec
  // returns mask.select(p (1 - p), 1) (Step1_Models.cpp:1809-1811) and fit_firth_nr builds X^T W X from it without the mask (:1287-1290,
  // :1325-1328), so penalty, hat diagonal and Newton matrix see the masked rows.  regenie's single-trait run of the same trait (the samples
  // dropped) gives the other value; its multi-trait run is the reference of a multi-trait run (found by tests/golden/fuzz_oracle_vs_reference.py).
  std::vector<double> Amasked((size_t)K * K, 0.0);
  for (int64_t i = 0; i < n; ++i)
    if (!mask[i])
      for (int a = 0; a < K; ++a) { const double xa = cols[a][i]; for (int c = 0; c <= a; ++c) Amasked[(size_t)a * K + c] += xa * cols[c][i]; }
  auto pen_dev = [&](const std::vector<double>& b, double& dev) {
    double ll = 0.0;
    A = Amasked;
    for (int64_t i = 0; i < n; ++i) {
      if (!mask[i]) continue;
      double e = offset[i];
      for (int c = 0; c < K; ++c) e += cols[c][i] * b[c];
      const double pr = get_pvec1(e);
      pv[i] = pr; w[i] = pr * (1.0 - pr);
      ll -= (y[i] == 0.0) ? std::log(1.0 - pr) : std::log(pr);
      for (int a = 0; a < K; ++a) { const double xa = cols[a][i] * w[i]; for (int c = 0; c <= a; ++c) A[(size_t)a * K + c] += xa * cols[c][i]; }
    }
    for (int a = 0; a < K; ++a) for (int c = a + 1; c < K; ++c) A[(size_t)a * K + c] = A[(size_t)c * K + a];
    double logdet;
    if (!spd_logdet_inv(A, K, logdet, &Ainv)) return false;
    dev = 2.0 * ll - logdet;
    return true;
  };
  double dev;
  if (!pen_dev(beta, dev)) return false;
  for (int it = 0; it < 2000; ++it) {
    std::fill(score.begin(), score.end(), 0.0);
    for (int64_t i = 0; i < n; ++i) {
      if (!mask[i]) continue;
      double h = 0.0;                                  // h_i = w_i x_i^T (X^T W X)^-1 x_i
      for (int a = 0; a < K; ++a) { double t = 0.0; for (int c = 0; c < K; ++c) t += Ainv[(size_t)a * K + c] * cols[c][i]; hx[a] = t; }
      for (int a = 0; a < K; ++a) h += cols[a][i] * hx[a];
      h *= w[i];
      const double u = y[i] - pv[i] + h * (0.5 - pv[i]);
      for (int a = 0; a < nfree; ++a) score[a] += cols[a][i] * u;
    }
    const std::vector<double>* Finv = &Ainv;
    if (nfree < K) {                                   // the step solves with the free block of the information alone (:1311-1314)
      for (int a = 0; a < nfree; ++a) for (int c = 0; c < nfree; ++c) Afree[(size_t)a * nfree + c] = A[(size_t)a * K + c];
      double ld;
      if (!spd_logdet_inv(Afree, nfree, ld, &Afinv)) return false;
      Finv = &Afinv;
    }
    // stop_tol > 0: fit_firth_nr's stopping rule (Step2_Models.cpp:1320-1323) -- |modified score| below the tolerance, from the second iteration on; the
    // estimates are then those of the iterate the reference stops at (with masked samples in X^T W X the iteration converges linearly: a tolerance off the root)
    if (stop_tol > 0 && it >= 1) {
      double smax = 0.0;
      for (int a = 0; a < nfree; ++a) smax = std::max(smax, std::fabs(score[a]));
      if (smax < stop_tol) { if (dev_out) *dev_out = dev; if (inv_out) *inv_out = Ainv; return true; }
    }
    const int F = nfree < K ? nfree : K;
    double mx = 0.0;
    for (int a = 0; a < nfree; ++a) { double t = 0.0; for (int c = 0; c < nfree; ++c) t += (*Finv)[(size_t)a * F + c] * score[c]; step[a] = t; mx = std::max(mx, std::fabs(t)); }
    if (mx < 1e-10) { if (dev_out) *dev_out = dev; if (inv_out) *inv_out = Ainv; return true; }
    if (mx > maxstep) for (int a = 0; a < nfree; ++a) step[a] *= maxstep / mx;
    double dev_new = dev;
    bool ok = false;
    for (int hs = 0; hs < 60; ++hs) {
      double smx = 0.0;
      for (int a = 0; a < K; ++a) { bnew[a] = beta[a] + (a < nfree ? step[a] : 0.0); if (a < nfree) smx = std::max(smx, std::fabs(step[a])); }
      if (pen_dev(bnew, dev_new) && (dev_new < dev + 1e-12 || smx < 1e-6)) { ok = true; break; }      // (steps that small change the deviance by less than its rounding)
      for (int a = 0; a < nfree; ++a) step[a] /= 2.0;
    }
    if (!ok) return false;
    beta = bnew; dev = dev_new;
  }
  return false;
}

// fit_approx_firth_null (Step2_Models.cpp:899-984): the covariate-only penalised fit, offset = LOCO prediction.  X [C][n] sample-fastest.
bool firth_null_fit(const double* y, const double* X, const uint8_t* mask, const double* offset, int64_t n, int C, std::vector<double>& beta) {
  std::vector<const double*> cols(C);
  for (int c = 0; c < C; ++c) cols[c] = X + (size_t)c * n;
  return firth_fit_cols(y, cols, mask, offset, n, C, 25.0, beta, nullptr, nullptr, 50 * NUMTOL);      // maxstep_null; tol = 50 numtol (fit_approx_firth_null, :906)
}

}  // namespace rgdrv
