// TEST INFRASTRUCTURE (oracle/ref_shim): regenie parses .bgen headers and variant identifying data through
// the external BGEN library v1.1.7 (src/bgen_to_vcf.hpp wraps genfile::bgen::*); that library is not under
// /root/reference.  This header provides the handful of names bgen_to_vcf.hpp uses, written from the
// published BGEN v1.1/v1.2 file layout (offset, header block, sample identifier block, variant
// identifying data, genotype data block), so the reference sources compile and read the reference's own
// example .bgen files.  Not BGEN-library code.
#ifndef RG_SHIM_GENFILE_BGEN_HPP
#define RG_SHIM_GENFILE_BGEN_HPP
#include <cstdint>
#include <cstring>
#include <istream>
#include <stdexcept>
#include <string>
#include <vector>
#include <zlib.h>

extern "C" size_t ZSTD_decompress(void* dst, size_t dstCapacity, const void* src, size_t compressedSize);
extern "C" unsigned ZSTD_isError(size_t code);

namespace genfile {
typedef uint8_t byte_t;
enum OrderType { eUnknownOrderType = 0, eUnorderedList = 1, eOrderedList = 2, ePerUnorderedGenotype = 3,
                 ePerOrderedHaplotype = 4, ePerUnorderedHaplotype = 5, ePerAllele = 6, ePerSample = 7 };
enum ValueType { eUnknownValueType = 0, eProbability = 1, eAlleleIndex = 2, eDosage = 3 };
struct MissingValue {};

namespace bgen {
enum FlagMask { e_NoFlags = 0, e_CompressedSNPBlocks = 0x3, e_Layout = 0x3C };
enum Compression { e_NoCompression = 0, e_ZlibCompression = 1, e_ZstdCompression = 2 };
enum Layout { e_Layout0 = 0x0, e_Layout1 = 0x4, e_Layout2 = 0x8 };
enum Structure { e_SampleIdentifiers = 0x80000000 };

struct Context {
  Context() : number_of_samples(0), number_of_variants(0), magic("bgen"), flags(0) {}
  uint32_t number_of_samples, number_of_variants;
  std::string magic, free_data;
  uint32_t flags;
};

namespace shim {
template <class T> inline bool rd(std::istream& s, T* v) {
  unsigned char b[sizeof(T)];
  s.read((char*)b, sizeof(T));
  if (!s) return false;
  T r = 0;
  for (size_t i = 0; i < sizeof(T); ++i) r |= (T)((T)b[i] << (8 * i));
  *v = r;
  return true;
}
template <class L> inline bool rd_str(std::istream& s, std::string* out) {
  L n;
  if (!rd(s, &n)) return false;
  out->resize(n);
  if (n) s.read(&(*out)[0], n);
  return (bool)s;
}
}

inline void read_offset(std::istream& s, uint32_t* offset) {
  if (!shim::rd(s, offset)) throw std::runtime_error("bgen: cannot read offset");
}
inline std::size_t read_header_block(std::istream& s, Context* ctx) {
  uint32_t hlen = 0, nv = 0, ns = 0, flags = 0;
  char magic[4];
  if (!shim::rd(s, &hlen) || !shim::rd(s, &nv) || !shim::rd(s, &ns)) throw std::runtime_error("bgen: bad header");
  s.read(magic, 4);
  if (!s || hlen < 20) throw std::runtime_error("bgen: bad header");
  if (std::memcmp(magic, "bgen", 4) != 0 && std::memcmp(magic, "\0\0\0\0", 4) != 0) throw std::runtime_error("bgen: bad magic");
  std::string free_data(hlen - 20, '\0');
  if (hlen > 20) s.read(&free_data[0], hlen - 20);
  if (!shim::rd(s, &flags)) throw std::runtime_error("bgen: bad header");
  ctx->number_of_samples = ns; ctx->number_of_variants = nv; ctx->magic.assign(magic, 4);
  ctx->free_data = free_data; ctx->flags = flags;
  return hlen;
}
template <class Setter>
inline std::size_t read_sample_identifier_block(std::istream& s, Context const& ctx, Setter setter) {
  uint32_t blen = 0, n = 0;
  if (!shim::rd(s, &blen) || !shim::rd(s, &n)) throw std::runtime_error("bgen: bad sample block");
  if (n != ctx.number_of_samples) throw std::runtime_error("bgen: sample count mismatch");
  for (uint32_t i = 0; i < n; ++i) {
    std::string id;
    if (!shim::rd_str<uint16_t>(s, &id)) throw std::runtime_error("bgen: bad sample block");
    setter(id);
  }
  return blen;
}
template <class NSetter, class ASetter>
inline bool read_snp_identifying_data(std::istream& s, Context const& ctx, std::string* SNPID, std::string* RSID,
                                      std::string* chromosome, uint32_t* position, NSetter set_n, ASetter set_allele) {
  uint32_t layout = ctx.flags & e_Layout;
  if (layout == e_Layout1 || layout == e_Layout0) {
    uint32_t n;
    if (!shim::rd(s, &n)) return false;
  }
  if (!shim::rd_str<uint16_t>(s, SNPID)) return false;
  if (!shim::rd_str<uint16_t>(s, RSID) || !shim::rd_str<uint16_t>(s, chromosome) || !shim::rd(s, position))
    throw std::runtime_error("bgen: truncated variant record");
  uint16_t nall = 2;
  if (layout == e_Layout2) { if (!shim::rd(s, &nall)) throw std::runtime_error("bgen: truncated variant record"); }
  set_n(nall);
  for (uint16_t a = 0; a < nall; ++a) {
    std::string al;
    if (!shim::rd_str<uint32_t>(s, &al)) throw std::runtime_error("bgen: truncated variant record");
    set_allele(a, al);
  }
  return true;
}
inline void ignore_genotype_data_block(std::istream& s, Context const& ctx) {
  uint32_t layout = ctx.flags & e_Layout;
  if (layout == e_Layout2 || (ctx.flags & e_CompressedSNPBlocks)) {
    uint32_t len;
    if (!shim::rd(s, &len)) throw std::runtime_error("bgen: truncated genotype block");
    s.ignore(len);
  } else s.ignore(6 * (std::streamsize)ctx.number_of_samples);
}
inline void shim_inflate(Context const& ctx, const byte_t* src, size_t n, std::vector<byte_t>* out, size_t outlen) {
  out->resize(outlen);
  uint32_t comp = ctx.flags & e_CompressedSNPBlocks;
  if (comp == e_ZlibCompression) {
    uLongf d = outlen;
    if (uncompress(&(*out)[0], &d, src, n) != Z_OK || d != outlen) throw std::runtime_error("bgen: zlib failure");
  } else if (comp == e_ZstdCompression) {
    size_t d = ZSTD_decompress(&(*out)[0], outlen, src, n);
    if (ZSTD_isError(d) || d != outlen) throw std::runtime_error("bgen: zstd failure");
  } else { if (n != outlen) throw std::runtime_error("bgen: size mismatch"); std::memcpy(&(*out)[0], src, n); }
}
// parse one genotype data block and hand the probabilities to the setter (see the comments in
// src/bgen_to_vcf.hpp:12-76 for the callback protocol)
template <class Setter>
inline void read_and_parse_genotype_data_block(std::istream& s, Context const& ctx, Setter& setter,
                                               std::vector<byte_t>* buf1, std::vector<byte_t>* buf2) {
  uint32_t layout = ctx.flags & e_Layout, comp = ctx.flags & e_CompressedSNPBlocks;
  const uint32_t N = ctx.number_of_samples;
  if (layout == e_Layout2) {
    uint32_t total = 0, dlen = 0;
    if (!shim::rd(s, &total)) throw std::runtime_error("bgen: truncated genotype block");
    size_t clen = total;
    if (comp) { if (!shim::rd(s, &dlen)) throw std::runtime_error("bgen: truncated genotype block"); clen = total - 4; }
    else dlen = total;
    buf1->resize(clen);
    if (clen) s.read((char*)&(*buf1)[0], clen);
    if (!s) throw std::runtime_error("bgen: truncated genotype block");
    shim_inflate(ctx, &(*buf1)[0], clen, buf2, dlen);
    const byte_t* p = &(*buf2)[0]; const byte_t* end = p + dlen;
    if (dlen < 10 + (size_t)N) throw std::runtime_error("bgen: short genotype block");
    uint32_t n; std::memcpy(&n, p, 4); p += 4;
    uint16_t K; std::memcpy(&K, p, 2); p += 2;
    unsigned pmin = *p++, pmax = *p++;
    const byte_t* ploidy = p; p += n;
    bool phased = (*p++) != 0; unsigned bits = *p++;
    if (n != N) throw std::runtime_error("bgen: sample count mismatch in block");
    auto nchoose = [](unsigned a, unsigned b) { double r = 1; for (unsigned i = 1; i <= b; ++i) r = r * (a - b + i) / i; return (uint32_t)(r + 0.5); };
    auto entries = [&](unsigned pl) { return phased ? pl * K : nchoose(pl + K - 1, K - 1); };
    setter.initialise(N, K);
    setter.set_min_max_ploidy(pmin, pmax, entries(pmin), entries(pmax));
    uint64_t bitpos = 0; const double denom = (double)((1ull << bits) - 1);
    if (bits < 1 || bits > 32) throw std::runtime_error("bgen: unsupported number of bits");
    auto next = [&]() -> double {
      uint64_t v = 0; size_t byte = bitpos >> 3; unsigned sh = bitpos & 7;
      for (unsigned i = 0; i < 5; ++i) if (p + byte + i < end) v |= (uint64_t)p[byte + i] << (8 * i);
      v = (v >> sh) & ((1ull << bits) - 1);
      bitpos += bits; return v / denom;
    };
    for (uint32_t i = 0; i < N; ++i) {
      unsigned pl = ploidy[i] & 0x3F; bool miss = ploidy[i] & 0x80;
      unsigned ne = entries(pl);
      unsigned stored = phased ? pl * (K - 1) : ne - 1;
      if (!setter.set_sample(i)) { bitpos += (uint64_t)stored * bits; continue; }
      setter.set_number_of_entries(pl, ne, phased ? ePerOrderedHaplotype : ePerUnorderedGenotype, eProbability);
      if (miss) { for (unsigned e = 0; e < ne; ++e) setter.set_value(e, MissingValue()); bitpos += (uint64_t)stored * bits; continue; }
      if (phased) {
        unsigned e = 0;
        for (unsigned h = 0; h < pl; ++h) { double sum = 0; for (unsigned a = 0; a + 1 < K; ++a) { double v = next(); sum += v; setter.set_value(e++, v); } setter.set_value(e++, 1.0 - sum); }
      } else {
        double sum = 0;
        for (unsigned e = 0; e + 1 < ne; ++e) { double v = next(); sum += v; setter.set_value(e, v); }
        setter.set_value(ne - 1, 1.0 - sum);
      }
    }
    setter.finalise();
  } else {
    // layout 1: three 16-bit probabilities per sample, scaled by 32768; all zero = missing
    size_t dlen = 6 * (size_t)N;
    if (comp) {
      uint32_t clen; if (!shim::rd(s, &clen)) throw std::runtime_error("bgen: truncated genotype block");
      buf1->resize(clen); s.read((char*)&(*buf1)[0], clen);
      shim_inflate(ctx, &(*buf1)[0], clen, buf2, dlen);
    } else { buf2->resize(dlen); s.read((char*)&(*buf2)[0], dlen); }
    if (!s) throw std::runtime_error("bgen: truncated genotype block");
    setter.initialise(N, 2);
    setter.set_min_max_ploidy(2, 2, 3, 3);
    for (uint32_t i = 0; i < N; ++i) {
      if (!setter.set_sample(i)) continue;
      setter.set_number_of_entries(2, 3, ePerUnorderedGenotype, eProbability);
      uint16_t a[3]; std::memcpy(a, &(*buf2)[6 * (size_t)i], 6);
      if (a[0] == 0 && a[1] == 0 && a[2] == 0) for (unsigned e = 0; e < 3; ++e) setter.set_value(e, MissingValue());
      else for (unsigned e = 0; e < 3; ++e) setter.set_value(e, a[e] / 32768.0);
    }
    setter.finalise();
  }
}
} // namespace bgen
} // namespace genfile
#endif
