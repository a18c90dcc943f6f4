// tests/test_host_format_cpu.py: rgfmt::fmt_g6 (regenie_amd/host/fmt_g6.h, the number format of the .loco / .prs rows) against snprintf("%g") --
// what `ofstream << double` prints in regenie's write_chr_row (/root/reference/src/Data.cpp:1951-1975).  argv[1] = random draws per family.
#include "../../regenie_amd/host/fmt_g6.h"
#include <cstdio>
#include <random>
#include <chrono>
#include <vector>
#include <string>
int main(int argc, char** argv) {
  long n = argc > 1 ? atol(argv[1]) : 20000000;
  std::mt19937_64 g(7);
  long bad = 0, slow = 0;
  char a[64], b[64];
  auto chk = [&](double v) {
    char* e = rgfmt::fmt_g6(v, a); *e = 0;
    snprintf(b, sizeof b, "%g", v);
    if (strcmp(a, b)) { if (bad < 20) printf("MISMATCH %.17g: %s vs %s\n", v, a, b); ++bad; }
  };
  // edge cases
  const double edges[] = {0.0, -0.0, 1.0, -1.0, 0.1, 0.001, 1e-4, 1e-5, 9.99995e-5, 9.999995e-5, 99999.95, 999999.5, 999999.4999, 1e6, 1e5, 100000.5, 123456.5, 123455.5, 0.5, 0.25,
    1.0000005, 1.0000015, 2.5e-5, 1e21, 9.999999e21, 1e22, 1e-17, 9.9e-18, 1e-300, 1e300, 5e-324, 1.7976931348623157e308, 0.000123456789, 1234567.0, 12345678.9, 0.3, 2.0/3, 1e-10, 1.5e-7};
  for (double v : edges) { chk(v); chk(-v); }
  chk(NAN); chk(INFINITY); chk(-INFINITY);
  // exact ties at six digits: k + 0.5 for six-digit k, and scaled by powers of two
  for (int k = 100000; k < 1000000; k += 37) { chk(k + 0.5); chk((k + 0.5) / 1024); chk((k + 0.5) * 64); chk((k + 0.5) / 8); }
  // near-ties: decimal strings with 7 digits ending in 5
  for (int k = 100000; k < 1000000; k += 13) for (int e = -8; e <= 8; ++e) { char t[64]; snprintf(t, sizeof t, "%d5e%d", k, e - 6); chk(strtod(t, nullptr)); snprintf(t, sizeof t, "%d49999999999e%d", k, e - 16); chk(strtod(t, nullptr)); }
  std::uniform_real_distribution<double> u(-1, 1), ue(-25, 25);
  std::normal_distribution<double> nd(0, 1);
  for (long i = 0; i < n; ++i) {
    chk(nd(g)); chk(u(g) * std::pow(10.0, ue(g)));
    uint64_t bits = g(); double v; memcpy(&v, &bits, 8); chk(v);
    // short decimals (what a rounded table would hold)
    chk(std::round(u(g) * 1e6) / 1e6); chk(std::round(u(g) * 1e4) / 1e7);
  }
  printf("checked, mismatches %ld\n", bad);
  // speed
  std::vector<double> x(2000000); for (auto& v : x) v = nd(g) * 0.3;
  auto t0 = std::chrono::steady_clock::now(); size_t tot = 0;
  for (double v : x) tot += rgfmt::fmt_g6(v, a) - a;
  auto t1 = std::chrono::steady_clock::now();
  for (double v : x) tot += std::to_chars(a, a + 48, v, std::chars_format::general, 6).ptr - a;
  auto t2 = std::chrono::steady_clock::now();
  printf("fmt_g6 %.1f ns, to_chars %.1f ns (%zu)\n", std::chrono::duration<double>(t1 - t0).count() * 1e9 / x.size(), std::chrono::duration<double>(t2 - t1).count() * 1e9 / x.size(), tot);
  return bad != 0;
}
