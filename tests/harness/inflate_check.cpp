// Test harness for regenie_amd/csrc/inflate_fast.h: every stream is decoded by rgflate::inflate_zlib and by zlib's uncompress();
// the two must agree byte for byte whenever rgflate accepts, rgflate must accept every valid stream produced here, and must never
// touch memory outside its buffers (built with -fsanitize=address,undefined by tests/test_inflate_cpu.py).
#include <zlib.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../../regenie_amd/csrc/inflate_fast.h"

static std::vector<uint8_t> deflate_with(const std::vector<uint8_t>& in, int level, int strategy, int memlevel) {
  z_stream zs{};
  deflateInit2(&zs, level, Z_DEFLATED, 15, memlevel, strategy);
  std::vector<uint8_t> out(deflateBound(&zs, in.size()) + in.size() / 4 + 1024);
  zs.next_in = (Bytef*)in.data(); zs.avail_in = (uInt)in.size();
  zs.next_out = out.data(); zs.avail_out = (uInt)out.size();
  const int rc = deflate(&zs, Z_FINISH);
  if (rc != Z_STREAM_END) { fprintf(stderr, "deflate failed rc=%d level %d strategy %d memlevel %d n=%zu out=%zu\n", rc, level, strategy, memlevel, in.size(), out.size()); exit(2); }
  out.resize(zs.total_out);
  deflateEnd(&zs);
  return out;
}

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 300;
  std::mt19937_64 rng(12345);
  rgflate::Tables* t = new rgflate::Tables;
  long accepted = 0, valid = 0, damaged = 0, damaged_accepted = 0;
  for (int r = 0; r < rounds; ++r) {
    // inputs: genotype-block-like bytes (few distinct values, runs), text-like, random, empty, tiny
    size_t n = r % 7 == 0 ? rng() % 40 : (size_t)(rng() % 200000);
    if (r == 1) n = 0;
    if (r == 2) n = 1500010;
    std::vector<uint8_t> in(n);
    const int kind = r % 5;
    for (size_t i = 0; i < n; ++i) {
      const uint64_t x = rng();
      switch (kind) {
        case 0: in[i] = (x % 10 < 7) ? ((x >> 8) % 3 == 0 ? 255 : 0) : (uint8_t)(x >> 16); break;        // probability bytes
        case 1: in[i] = (uint8_t)(x >> 20); break;                                                          // incompressible
        case 2: in[i] = (uint8_t)("acgtn \n"[x % 7]); break;
        case 3: in[i] = i > 300 && x % 3 ? in[i - 1 - (x >> 8) % 300] : (uint8_t)(x >> 24); break;          // short-distance matches
        default: in[i] = (uint8_t)((i / 1000) & 255); break;                                                // long runs (distance 1)
      }
    }
    static const int levels[] = {0, 1, 1, 6, 9}, strategies[] = {Z_DEFAULT_STRATEGY, Z_FIXED, Z_HUFFMAN_ONLY, Z_RLE, Z_FILTERED};
    const std::vector<uint8_t> z = deflate_with(in, levels[(r / 5) % 5], strategies[(r / 25) % 5], 1 + (r % 9));
    ++valid;
    {
      // the output buffer is exactly n bytes: any write past it is caught by the sanitizer
      std::vector<uint8_t> out(n);
      if (!rgflate::inflate_zlib(out.data(), n, z.data(), z.size(), *t)) { fprintf(stderr, "round %d: a valid stream was refused (n=%zu)\n", r, n); return 1; }
      if (out != in) { fprintf(stderr, "round %d: wrong bytes\n", r); return 1; }
      ++accepted;
      // wrong size expectations must be refused
      if (n > 0) {
        std::vector<uint8_t> o2(n - 1);
        if (rgflate::inflate_zlib(o2.data(), n - 1, z.data(), z.size(), *t)) { fprintf(stderr, "round %d: short output accepted\n", r); return 1; }
      }
      std::vector<uint8_t> o3(n + 1);
      if (rgflate::inflate_zlib(o3.data(), n + 1, z.data(), z.size(), *t)) { fprintf(stderr, "round %d: long output accepted\n", r); return 1; }
    }
    // damage: flipped bytes, truncation -- whatever rgflate accepts must equal what zlib makes of the same bytes
    for (int d = 0; d < 6 && z.size() > 8; ++d) {
      std::vector<uint8_t> zz = z;
      if (d < 4) zz[rng() % zz.size()] ^= (uint8_t)(1u << (rng() % 8));
      else zz.resize(zz.size() - 1 - rng() % std::min<size_t>(zz.size() - 7, 50));
      ++damaged;
      std::vector<uint8_t> out(n), ref(n);
      const bool ok = rgflate::inflate_zlib(out.data(), n, zz.data(), zz.size(), *t);
      if (ok) {
        ++damaged_accepted;
        uLongf dl = (uLongf)n;
        const int zr = uncompress(ref.data(), &dl, zz.data(), (uLong)zz.size());
        if (zr != Z_OK || dl != n || out != ref) { fprintf(stderr, "round %d: damaged stream accepted with bytes zlib does not give\n", r); return 1; }
      }
    }
  }
  printf("valid %ld accepted %ld damaged %ld damaged_accepted %ld\n", valid, accepted, damaged, damaged_accepted);
  if (argc > 2) {      // timing on a file holding one zlib stream: argv[2] = path, argv[3] = inflated size
    FILE* f = fopen(argv[2], "rb");
    std::vector<uint8_t> z(1 << 24);
    z.resize(fread(z.data(), 1, z.size(), f));
    fclose(f);
    const size_t n = (size_t)atol(argv[3]);
    std::vector<uint8_t> a(n), b(n);
    for (int which = 0; which < 2; ++which) {
      auto t0 = std::chrono::steady_clock::now();
      bool ok = true;
      for (int k = 0; k < 30; ++k) {
        if (which == 0) ok &= rgflate::inflate_zlib(a.data(), n, z.data(), z.size(), *t);
        else { uLongf dl = (uLongf)n; ok &= uncompress(b.data(), &dl, z.data(), (uLong)z.size()) == Z_OK; }
      }
      const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / 30;
      printf("%s: %.2f ms per block, %.0f MB/s %s\n", which == 0 ? "rgflate" : "zlib   ", ms, n / ms / 1e3, ok ? "" : "(FAILED)");
    }
    if (a != b) { printf("timing outputs differ\n"); return 1; }
  }
  return 0;
}
