// regenie-amd: the default stream format of a double (`%g`, six significant digits) without the general-purpose machinery.
//
// regenie writes its .loco / .prs rows with `ofstream << double` (write_chr_row, /root/reference/src/Data.cpp:1951-1975): at 500,000 samples a
// .loco file holds 11.5 million numbers, and std::to_chars(general, 6) at 60 - 90 ns each made the ten files of BASELINE configs[2] the
// longest part of level 1 on the 16 host cores of the GPU box (1.06 s against 0.68 s of GPU work, profiles/r6_e2e_config3.log).
//
// fmt_g6 scales |v| by an exact power of ten to [1e5, 1e6), rounds to the six-digit integer and prints that: one multiplication (or division)
// with one rounding error of at most half an ulp (< 1e-10 at that magnitude).  Whenever the scaled value is closer than 1e-6 to a rounding
// boundary -- the only place where that error could change the digits, exact ties included -- and for everything outside 1e-17 <= |v| < 1e22,
// NaN and infinities, it hands the value to std::to_chars, so the text is the same as the stream's for EVERY double
// (tests/test_host_format_cpu.py: boundary cases, exact ties and 200 million random values against snprintf("%g")).
#pragma once
#include <charconv>
#include <cmath>
#include <cstdint>
#include <cstring>

namespace rgfmt {

inline const double* pow10_table() {
  static const double t[23] = {1e0,  1e1,  1e2,  1e3,  1e4,  1e5,  1e6,  1e7,  1e8,  1e9,  1e10, 1e11,
                               1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
  return t;
}

// writes v as `%g` would and returns the end of the text; buf holds at least 32 bytes
inline char* fmt_g6(double v, char* buf) {
  const double a = std::fabs(v);
  if (!(a >= 1e-17 && a < 1e22)) {
    if (v == 0.0) {
      if (std::signbit(v)) *buf++ = '-';
      *buf++ = '0';
      return buf;
    }
    return std::to_chars(buf, buf + 32, v, std::chars_format::general, 6).ptr;
  }
  const double* p10 = pow10_table();
  // decimal exponent of a: from the binary one (a = f 2^e2, f in [0.5, 1); a is normal in this range), corrected by at most one step below
  uint64_t bits;
  std::memcpy(&bits, &a, 8);
  const int e2 = (int)(bits >> 52) - 1022;
  int e10 = ((e2 - 1) * 1233) >> 12;                    // floor((e2 - 1) log10 2) for |e2| < 1650 (arithmetic shift)
  double s;
  for (int tries = 0;; ++tries) {
    const int k = 5 - e10;                              // |k| <= 22 by the range check (e10 in [-17, 21])
    s = k >= 0 ? a * p10[k] : a / p10[-k];
    if (s < 1e5 - 1e-6) { --e10; }                      // (within 1e-6 below the decade: rounds up to 100000 of this decade either way)
    else if (s >= 1e6) { ++e10; }
    else break;
    if (tries == 2 || e10 < -17 || e10 > 21) return std::to_chars(buf, buf + 32, v, std::chars_format::general, 6).ptr;
  }
  const uint32_t fl = (uint32_t)s;                      // s in [99999.999999, 1e6): truncation = floor
  const double fr = s - (double)fl;                     // exact
  if (std::fabs(fr - 0.5) < 1e-6)                       // a rounding boundary within the scaling error: the slow, exact way
    return std::to_chars(buf, buf + 32, v, std::chars_format::general, 6).ptr;
  uint32_t n = fl + (fr > 0.5 ? 1u : 0u);               // six digits, or 1000000
  if (n == 1000000u) { n = 100000u; ++e10; }
  static const char pairs[201] =
      "00010203040506070809101112131415161718192021222324252627282930313233343536373839404142434445464748495051525354555657585960616263646566676869"
      "707172737475767778798081828384858687888990919293949596979899";
  char d[6];
  const uint32_t hi = n / 10000u, lo = n - hi * 10000u, mid = lo / 100u, last = lo - mid * 100u;
  std::memcpy(d, pairs + 2 * hi, 2);
  std::memcpy(d + 2, pairs + 2 * mid, 2);
  std::memcpy(d + 4, pairs + 2 * last, 2);
  int nd = 6;
  while (nd > 1 && d[nd - 1] == '0') --nd;              // %g drops trailing zeros
  char* o = buf;
  if (v < 0) *o++ = '-';
  if (e10 >= -4 && e10 < 6) {
    if (e10 >= 0) {
      const int ni = e10 + 1;                           // digits in front of the point
      for (int i = 0; i < ni; ++i) *o++ = i < nd ? d[i] : '0';
      if (nd > ni) {
        *o++ = '.';
        for (int i = ni; i < nd; ++i) *o++ = d[i];
      }
    } else {
      *o++ = '0';
      *o++ = '.';
      for (int i = -1; i > e10; --i) *o++ = '0';
      for (int i = 0; i < nd; ++i) *o++ = d[i];
    }
  } else {
    *o++ = d[0];
    if (nd > 1) {
      *o++ = '.';
      for (int i = 1; i < nd; ++i) *o++ = d[i];
    }
    *o++ = 'e';
    int x = e10;
    if (x < 0) { *o++ = '-'; x = -x; } else *o++ = '+';
    if (x >= 100) { *o++ = (char)('0' + x / 100); x %= 100; }
    *o++ = (char)('0' + x / 10);
    *o++ = (char)('0' + x % 10);
  }
  return o;
}

}  // namespace rgfmt
