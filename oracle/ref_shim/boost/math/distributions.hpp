// TEST INFRASTRUCTURE (oracle/ref_shim): a stand-in for the subset of Boost.Math that regenie v4.1.2
// names (src/Regenie.hpp:70 includes <boost/math/distributions.hpp> in every translation unit), written on
// <cmath> so that the reference sources under /root/reference can be compiled where Boost is absent.
// Not Boost code: the distribution functions are implemented here from the textbook series / continued
// fractions (regularised incomplete gamma and beta), Wichura's AS 241 for the normal quantile, and
// bracketing root finders for the other quantiles.  Accuracy target: ~1e-14 relative on the values the
// Step-1 / Step-2 single-variant paths use (chi-square tails, normal cdf/quantile).
#ifndef RG_SHIM_BOOST_MATH_DISTRIBUTIONS_HPP
#define RG_SHIM_BOOST_MATH_DISTRIBUTIONS_HPP
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <stdexcept>

namespace boost { namespace math {

// classification by bit pattern: the reference is built with -ffast-math, under which `x != x` folds away
inline uint64_t shim_bits(double x) { uint64_t u; std::memcpy(&u, &x, 8); return u; }
template <class T> inline bool isnan(T x) {
  uint64_t u = shim_bits((double)x) & 0x7fffffffffffffffull; return u > 0x7ff0000000000000ull;
}
template <class T> inline bool isnormal(T x) {
  uint64_t e = (shim_bits((double)x) >> 52) & 0x7ff; return e != 0 && e != 0x7ff;
}
inline double lgamma(double x) { return ::lgamma(x); }

namespace shim_detail {
const double EPS = 1e-16;
const double TINY = 1e-300;

// regularised lower incomplete gamma P(a,x) by its power series (x < a+1)
inline double gamma_p_series(double a, double x) {
  double sum = 1.0 / a, term = sum, ap = a;
  for (int n = 0; n < 100000; ++n) {
    ap += 1.0; term *= x / ap; sum += term;
    if (std::fabs(term) < std::fabs(sum) * EPS) break;
  }
  return sum * std::exp(-x + a * std::log(x) - ::lgamma(a));
}
// regularised upper incomplete gamma Q(a,x) by the modified Lentz continued fraction (x >= a+1)
inline double gamma_q_cf(double a, double x) {
  double b = x + 1.0 - a, c = 1.0 / TINY, d = 1.0 / b, h = d;
  for (int i = 1; i < 100000; ++i) {
    double an = -i * (i - a);
    b += 2.0;
    d = an * d + b; if (std::fabs(d) < TINY) d = TINY;
    c = b + an / c; if (std::fabs(c) < TINY) c = TINY;
    d = 1.0 / d;
    double del = d * c; h *= del;
    if (std::fabs(del - 1.0) < EPS) break;
  }
  return std::exp(-x + a * std::log(x) - ::lgamma(a)) * h;
}
inline double gamma_p(double a, double x) {
  if (x <= 0) return 0.0;
  if (x < a + 1.0) return gamma_p_series(a, x);
  return 1.0 - gamma_q_cf(a, x);
}
inline double gamma_q(double a, double x) {
  if (x <= 0) return 1.0;
  if (x < a + 1.0) return 1.0 - gamma_p_series(a, x);
  return gamma_q_cf(a, x);
}
// continued fraction of the incomplete beta function
inline double beta_cf(double a, double b, double x) {
  double qab = a + b, qap = a + 1.0, qam = a - 1.0;
  double c = 1.0, d = 1.0 - qab * x / qap;
  if (std::fabs(d) < TINY) d = TINY;
  d = 1.0 / d; double h = d;
  for (int m = 1; m < 100000; ++m) {
    int m2 = 2 * m;
    double aa = m * (b - m) * x / ((qam + m2) * (a + m2));
    d = 1.0 + aa * d; if (std::fabs(d) < TINY) d = TINY;
    c = 1.0 + aa / c; if (std::fabs(c) < TINY) c = TINY;
    d = 1.0 / d; h *= d * c;
    aa = -(a + m) * (qab + m) * x / ((a + m2) * (qap + m2));
    d = 1.0 + aa * d; if (std::fabs(d) < TINY) d = TINY;
    c = 1.0 + aa / c; if (std::fabs(c) < TINY) c = TINY;
    d = 1.0 / d; double del = d * c; h *= del;
    if (std::fabs(del - 1.0) < EPS) break;
  }
  return h;
}
// regularised incomplete beta I_x(a,b) and its complement
inline void ibeta_pair(double a, double b, double x, double& p, double& q) {
  if (x <= 0) { p = 0; q = 1; return; }
  if (x >= 1) { p = 1; q = 0; return; }
  double bt = std::exp(::lgamma(a + b) - ::lgamma(a) - ::lgamma(b) + a * std::log(x) + b * std::log1p(-x));
  if (x < (a + 1.0) / (a + b + 2.0)) { p = bt * beta_cf(a, b, x) / a; q = 1.0 - p; }
  else { q = bt * beta_cf(b, a, 1.0 - x) / b; p = 1.0 - q; }
}
inline double norm_cdf(double z) { return 0.5 * std::erfc(-z * M_SQRT1_2); }
// Wichura, Algorithm AS 241 (PPND16)
inline double norm_quantile(double p) {
  if (p <= 0) return -std::numeric_limits<double>::infinity();
  if (p >= 1) return std::numeric_limits<double>::infinity();
  double q = p - 0.5, r, val;
  if (std::fabs(q) <= 0.425) {
    r = 0.180625 - q * q;
    val = q * (((((((2.5090809287301226727e3 * r + 3.3430575583588128105e4) * r + 6.7265770927008700853e4) * r
      + 4.5921953931549871457e4) * r + 1.3731693765509461125e4) * r + 1.9715909503065514427e3) * r
      + 1.3314166789178437745e2) * r + 3.3871328727963666080e0)
      / (((((((5.2264952788528545610e3 * r + 2.8729085735721942674e4) * r + 3.9307895800092710610e4) * r
      + 2.1213794301586595867e4) * r + 5.3941960214247511077e3) * r + 6.8718700749205790830e2) * r
      + 4.2313330701600911252e1) * r + 1.0);
    return val;
  }
  r = q < 0 ? p : 1.0 - p;
  r = std::sqrt(-std::log(r));
  if (r <= 5.0) {
    r -= 1.6;
    val = (((((((7.74545014278341407640e-4 * r + 2.27238449892691845833e-2) * r + 2.41780725177450611770e-1) * r
      + 1.27045825245236838258e0) * r + 3.64784832476320460504e0) * r + 5.76949722146069140550e0) * r
      + 4.63033784615654529590e0) * r + 1.42343711074968357734e0)
      / (((((((1.05075007164441684324e-9 * r + 5.47593808499534494600e-4) * r + 1.51986665636164571966e-2) * r
      + 1.48103976427480074590e-1) * r + 6.89767334985100004550e-1) * r + 1.67638483018380384940e0) * r
      + 2.05319162663775882187e0) * r + 1.0);
  } else {
    r -= 5.0;
    val = (((((((2.01033439929228813265e-7 * r + 2.71155556874348757815e-5) * r + 1.24266094738807843860e-3) * r
      + 2.65321895265761230930e-2) * r + 2.96560571828504891230e-1) * r + 1.78482653991729133580e0) * r
      + 5.46378491116411436990e0) * r + 6.65790464350110377720e0)
      / (((((((2.04426310338993978564e-15 * r + 1.42151175831644588870e-7) * r + 1.84631831751005468180e-5) * r
      + 7.86869131145613259100e-4) * r + 1.48753612908506148525e-2) * r + 1.36929880922735805310e-1) * r
      + 5.99832206555887937690e-1) * r + 1.0);
  }
  return q < 0 ? -val : val;
}
// monotone increasing F on [lo,hi]: find x with F(x)=target (bisection + secant polish)
template <class F> inline double invert_increasing(F f, double target, double lo, double hi) {
  for (int it = 0; it < 400; ++it) {
    double mid = 0.5 * (lo + hi);
    if (mid == lo || mid == hi) break;
    if (f(mid) < target) lo = mid; else hi = mid;
  }
  return 0.5 * (lo + hi);
}
} // namespace shim_detail

template <class Dist> struct complemented2_type {
  const Dist& dist; double param;
  complemented2_type(const Dist& d, double x) : dist(d), param(x) {}
};
template <class Dist> inline complemented2_type<Dist> complement(const Dist& d, double x) {
  return complemented2_type<Dist>(d, x);
}

// ---- normal -------------------------------------------------------------------------------------
template <class T = double> struct normal_distribution {
  T m, s;
  normal_distribution(T mean = 0, T sd = 1) : m(mean), s(sd) {}
  T mean() const { return m; } T standard_deviation() const { return s; }
};
typedef normal_distribution<double> normal;
inline double cdf(const normal& d, double x) { return shim_detail::norm_cdf((x - d.m) / d.s); }
inline double cdf(const complemented2_type<normal>& c) { return shim_detail::norm_cdf(-(c.param - c.dist.m) / c.dist.s); }
inline double pdf(const normal& d, double x) { double z = (x - d.m) / d.s; return std::exp(-0.5 * z * z) / (d.s * std::sqrt(2 * M_PI)); }
inline double quantile(const normal& d, double p) { return d.m + d.s * shim_detail::norm_quantile(p); }
inline double quantile(const complemented2_type<normal>& c) { return c.dist.m - c.dist.s * shim_detail::norm_quantile(c.param); }

// ---- gamma / chi-squared ------------------------------------------------------------------------
template <class T = double> struct gamma_distribution {
  T k, theta;
  gamma_distribution(T shape, T scale = 1) : k(shape), theta(scale) {}
};
inline double cdf(const gamma_distribution<double>& d, double x) { return shim_detail::gamma_p(d.k, x / d.theta); }
inline double cdf(const complemented2_type<gamma_distribution<double> >& c) { return shim_detail::gamma_q(c.dist.k, c.param / c.dist.theta); }

template <class T = double> struct chi_squared_distribution {
  T df;
  chi_squared_distribution(T v) : df(v) {}
  T degrees_of_freedom() const { return df; }
};
typedef chi_squared_distribution<double> chi_squared;
inline double cdf(const chi_squared& d, double x) { return shim_detail::gamma_p(0.5 * d.df, 0.5 * x); }
inline double cdf(const complemented2_type<chi_squared>& c) { return shim_detail::gamma_q(0.5 * c.dist.df, 0.5 * c.param); }
inline double pdf(const chi_squared& d, double x) {
  if (x <= 0) return 0.0;
  double a = 0.5 * d.df;
  return std::exp((a - 1) * std::log(x) - 0.5 * x - a * M_LN2 - ::lgamma(a));
}
inline double quantile(const chi_squared& d, double p) {
  if (p <= 0) return 0.0;
  if (p >= 1) return std::numeric_limits<double>::infinity();
  double hi = d.df + 10.0; while (shim_detail::gamma_p(0.5 * d.df, 0.5 * hi) < p && hi < 1e300) hi *= 2;
  const chi_squared dd = d;
  return shim_detail::invert_increasing([&dd](double x) { return shim_detail::gamma_p(0.5 * dd.df, 0.5 * x); }, p, 0.0, hi);
}
inline double quantile(const complemented2_type<chi_squared>& c) {
  double q = c.param;
  if (q >= 1) return 0.0;
  if (q <= 0) return std::numeric_limits<double>::infinity();
  double df = c.dist.df;
  double hi = df + 10.0; while (shim_detail::gamma_q(0.5 * df, 0.5 * hi) > q && hi < 1e300) hi *= 2;
  // -Q is increasing
  return shim_detail::invert_increasing([df](double x) { return -shim_detail::gamma_q(0.5 * df, 0.5 * x); }, -q, 0.0, hi);
}

// ---- non-central chi-squared (Poisson mixture) ----------------------------------------------------
template <class T = double> struct non_central_chi_squared_distribution {
  T df, ncp;
  non_central_chi_squared_distribution(T v, T lambda) : df(v), ncp(lambda) {}
};
typedef non_central_chi_squared_distribution<double> non_central_chi_squared;
inline double cdf(const non_central_chi_squared& d, double x) {
  if (x <= 0) return 0.0;
  double h = 0.5 * d.ncp, sum = 0.0;
  int j0 = (int)h; // start at the mode of the Poisson weights, walk both ways
  double w0 = std::exp(-h + j0 * std::log(h > 0 ? h : 1.0) - ::lgamma(j0 + 1.0));
  if (h <= 0) { return shim_detail::gamma_p(0.5 * d.df, 0.5 * x); }
  double w = w0;
  for (int j = j0; j < j0 + 100000; ++j) { double t = w * shim_detail::gamma_p(0.5 * d.df + j, 0.5 * x); sum += t; w *= h / (j + 1.0); if (w < 1e-18 && j > j0 + 10) break; }
  w = w0;
  for (int j = j0 - 1; j >= 0; --j) { w *= (j + 1.0) / h; sum += w * shim_detail::gamma_p(0.5 * d.df + j, 0.5 * x); if (w < 1e-18) break; }
  return sum;
}
inline double cdf(const complemented2_type<non_central_chi_squared>& c) {
  const non_central_chi_squared& d = c.dist; double x = c.param;
  if (x <= 0) return 1.0;
  double h = 0.5 * d.ncp, sum = 0.0;
  if (h <= 0) { return shim_detail::gamma_q(0.5 * d.df, 0.5 * x); }
  int j0 = (int)h;
  double w0 = std::exp(-h + j0 * std::log(h) - ::lgamma(j0 + 1.0));
  double w = w0;
  for (int j = j0; j < j0 + 100000; ++j) { sum += w * shim_detail::gamma_q(0.5 * d.df + j, 0.5 * x); w *= h / (j + 1.0); if (w < 1e-18 && j > j0 + 10) break; }
  w = w0;
  for (int j = j0 - 1; j >= 0; --j) { w *= (j + 1.0) / h; sum += w * shim_detail::gamma_q(0.5 * d.df + j, 0.5 * x); if (w < 1e-18) break; }
  return sum;
}

// ---- beta -----------------------------------------------------------------------------------------
template <class T = double> struct beta_distribution {
  T a, b;
  beta_distribution(T alpha = 1, T beta = 1) : a(alpha), b(beta) {}
  T alpha() const { return a; } T beta() const { return b; }
};
inline double cdf(const beta_distribution<double>& d, double x) { double p, q; shim_detail::ibeta_pair(d.a, d.b, x, p, q); return p; }
inline double cdf(const complemented2_type<beta_distribution<double> >& c) { double p, q; shim_detail::ibeta_pair(c.dist.a, c.dist.b, c.param, p, q); return q; }
inline double pdf(const beta_distribution<double>& d, double x) {
  if (x < 0 || x > 1) return 0.0;
  return std::exp(::lgamma(d.a + d.b) - ::lgamma(d.a) - ::lgamma(d.b) + (d.a - 1) * std::log(x) + (d.b - 1) * std::log1p(-x));
}
inline double quantile(const beta_distribution<double>& d, double p) {
  if (p <= 0) return 0.0;
  if (p >= 1) return 1.0;
  const beta_distribution<double> dd = d;
  return shim_detail::invert_increasing([&dd](double x) { double pp, qq; shim_detail::ibeta_pair(dd.a, dd.b, x, pp, qq); return pp; }, p, 0.0, 1.0);
}

// ---- Student t, Fisher F, Cauchy ------------------------------------------------------------------
template <class T = double> struct students_t_distribution { T df; students_t_distribution(T v) : df(v) {} };
typedef students_t_distribution<double> students_t;
inline double cdf(const complemented2_type<students_t>& c) {
  double t = c.param, v = c.dist.df, p, q;
  shim_detail::ibeta_pair(0.5 * v, 0.5, v / (v + t * t), p, q);
  return t >= 0 ? 0.5 * p : 1.0 - 0.5 * p;
}
inline double cdf(const students_t& d, double t) { return 1.0 - cdf(complement(d, t)); }

template <class T = double> struct fisher_f_distribution { T d1, d2; fisher_f_distribution(T a, T b) : d1(a), d2(b) {} };
typedef fisher_f_distribution<double> fisher_f;
inline double cdf(const fisher_f& d, double x) { double p, q; shim_detail::ibeta_pair(0.5 * d.d1, 0.5 * d.d2, d.d1 * x / (d.d1 * x + d.d2), p, q); return p; }
inline double cdf(const complemented2_type<fisher_f>& c) { double p, q; shim_detail::ibeta_pair(0.5 * c.dist.d1, 0.5 * c.dist.d2, c.dist.d1 * c.param / (c.dist.d1 * c.param + c.dist.d2), p, q); return q; }

template <class T = double> struct cauchy_distribution { T x0, g; cauchy_distribution(T loc = 0, T scale = 1) : x0(loc), g(scale) {} };
typedef cauchy_distribution<double> cauchy;
inline double cdf(const cauchy& d, double x) { return 0.5 + std::atan((x - d.x0) / d.g) / M_PI; }
inline double cdf(const complemented2_type<cauchy>& c) { return 0.5 - std::atan((c.param - c.dist.x0) / c.dist.g) / M_PI; }
inline double quantile(const cauchy& d, double p) { return d.x0 + d.g * std::tan(M_PI * (p - 0.5)); }
inline double quantile(const complemented2_type<cauchy>& c) { return c.dist.x0 - c.dist.g * std::tan(M_PI * (c.param - 0.5)); }

template <class T> inline T binomial_coefficient(unsigned n, unsigned k) {
  if (k > n) return 0;
  return (T)std::floor(0.5 + std::exp(::lgamma(n + 1.0) - ::lgamma(k + 1.0) - ::lgamma(n - k + 1.0)));
}

}} // namespace boost::math
#endif
