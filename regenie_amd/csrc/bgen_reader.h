// BGEN v1.2 reader for regenie's Step-1 `--bgen` input: the host-side counterpart of prep_bgen's variant scan
// (Geno.cpp:38-175) and of the self-contained fast path readChunkFromBGEN / readChunkFromBGENFileToG_fast
// (Geno.cpp:2122-2171, :1574-1699) that regenie uses for "layout 2, 8 bits per probability" files (the UK Biobank format):
// per variant, inflate the genotype block and turn the two stored probabilities of every sample into a dosage,
//     prob2 = max(1 - prob0 - prob1, 0),   G = prob1 + 2 prob0   (or prob1 + 2 prob2 with --ref-first),   missing -> -3.
// The rows go to the fp64 level 0 (rg_l0_blocks_f64), which does the reference's mean imputation.  Written from the
// published BGEN v1.2 layout (the reference links the external BGEN library for everything but that fast path; none of it is
// used here).  Scope = what the fast path accepts: layout 2, unphased, biallelic, diploid, 8-bit probabilities, zlib or zstd
// (zstd through libzstd.so.1 at run time) or no compression.  Anything else is an error, never a silent fallback.
#pragma once
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <memory>
#include <string>
#include <vector>

#include "inflate_fast.h"

namespace rgbgen {

struct Variant {
  uint64_t offset;   // file offset of the variant identifying data (what regenie keeps in snpinfo[].offset)
  uint64_t data;     // file offset of the genotype data block (its 4-byte length field)
  uint32_t csize;    // that length field: bytes of the block after it (compressed files: 4 bytes of inflated length + the stream)
  uint32_t position;
  std::string id, rsid, chrom, a0, a1;
};

class Reader {
 public:
  Reader() = default;
  Reader(const Reader&) = delete;
  Reader& operator=(const Reader&) = delete;
  ~Reader() { close(); }
  void close() {
    if (fd_ >= 0) ::close(fd_);
    fd_ = -1;
  }

  uint32_t n_samples() const { return n_; }
  uint32_t n_variants() const { return m_; }
  int compression() const { return comp_; }
  int layout() const { return layout_; }
  bool has_sample_ids() const { return !ids_.empty(); }
  const std::vector<std::string>& sample_ids() const { return ids_; }
  const std::vector<Variant>& variants() const { return vars_; }

  void open(const std::string& path) {
    close();
    fd_ = ::open(path.c_str(), O_RDONLY);
    if (fd_ < 0) throw std::runtime_error("cannot open file : " + path);
    struct stat st;
    if (fstat(fd_, &st) != 0) throw std::runtime_error("cannot stat file : " + path);
    fsize_ = (uint64_t)st.st_size;
    uint8_t h[24];
    if (fsize_ < 24 || !pread_all(h, 24, 0)) throw std::runtime_error("invalid bgen file (too short) : " + path);
    uint32_t offset, lh;
    std::memcpy(&offset, h, 4);
    std::memcpy(&lh, h + 4, 4);
    std::memcpy(&m_, h + 8, 4);
    std::memcpy(&n_, h + 12, 4);
    if (std::memcmp(h + 16, "bgen", 4) != 0 && std::memcmp(h + 16, "\0\0\0\0", 4) != 0)
      throw std::runtime_error("invalid bgen file (magic number mismatch) : " + path);
    if (lh < 20 || 4ull + lh > fsize_) throw std::runtime_error("invalid bgen header : " + path);
    uint32_t flags;
    if (!pread_all(&flags, 4, 4ull + lh - 4)) throw std::runtime_error("cannot read bgen header : " + path);
    comp_ = (int)(flags & 3);
    layout_ = (int)((flags >> 2) & 15);
    if (layout_ != 2) throw std::runtime_error("bgen layout " + std::to_string(layout_) + " is not supported (layout 2, i.e. BGEN v1.2, is) : " + path);
    if (comp_ > 2) throw std::runtime_error("unknown bgen compression flag : " + path);
    if (n_ == 0 || m_ == 0) throw std::runtime_error("empty bgen file : " + path);
    uint64_t pos = 4ull + lh;
    if (flags >> 31) {  // sample identifier block
      uint8_t b[8];
      if (!pread_all(b, 8, pos)) throw std::runtime_error("cannot read bgen sample block : " + path);
      uint32_t lsi, ns;
      std::memcpy(&lsi, b, 4);
      std::memcpy(&ns, b + 4, 4);
      if (ns != n_) throw std::runtime_error("bgen sample block does not match the header's sample count : " + path);
      if ((uint64_t)lsi > fsize_ || 2ull * ns > fsize_) throw std::runtime_error("malformed bgen sample block : " + path);       // (refused before a buffer of that length is allocated)
      std::vector<uint8_t> blk(lsi > 8 ? lsi - 8 : 0);
      if (!blk.empty() && !pread_all(blk.data(), blk.size(), pos + 8)) throw std::runtime_error("cannot read bgen sample block : " + path);
      size_t p = 0;
      for (uint32_t i = 0; i < ns; ++i) {
        if (p + 2 > blk.size()) throw std::runtime_error("malformed bgen sample block : " + path);
        uint16_t l;
        std::memcpy(&l, blk.data() + p, 2);
        p += 2;
        if (p + l > blk.size()) throw std::runtime_error("malformed bgen sample block : " + path);
        ids_.emplace_back((const char*)blk.data() + p, l);
        p += l;
      }
    }
    // variant scan: identifying data, then skip the genotype block
    pos = 4ull + offset;
    // a variant's identifying data and block length take 24 bytes or more: a damaged count is refused before the table is reserved for it
    if (24ull * m_ > fsize_) throw std::runtime_error("invalid bgen header (more variants than the file can hold) : " + path);
    vars_.reserve(m_);
    // the identifying data of a variant is a few dozen bytes in front of a genotype block that is skipped: one small read
    // per variant (a window that is refilled when a field runs past it) instead of one system call per field
    std::vector<uint8_t> win;
    uint64_t woff = 0;
    auto fetch = [&](uint64_t at, uint64_t len) -> const uint8_t* {
      need(at, len, path);
      if (at < woff || at + len > woff + win.size()) {
        const uint64_t want = std::min<uint64_t>(std::max<uint64_t>(len, 512), fsize_ - at);
        win.resize(want);
        if (!pread_all(win.data(), want, at)) throw std::runtime_error("cannot read bgen file : " + path);
        woff = at;
      }
      return win.data() + (at - woff);
    };
    for (uint32_t j = 0; j < m_; ++j) {
      Variant v;
      v.offset = pos;
      auto str16 = [&](std::string& s) {
        uint16_t l;
        std::memcpy(&l, fetch(pos, 2), 2);
        pos += 2;
        s.assign((const char*)fetch(pos, l), l);
        pos += l;
      };
      str16(v.id);
      str16(v.rsid);
      str16(v.chrom);
      uint16_t k;
      {
        const uint8_t* q = fetch(pos, 6);
        std::memcpy(&v.position, q, 4);
        std::memcpy(&k, q + 4, 2);
      }
      pos += 6;
      if (k != 2) throw std::runtime_error("only bi-allelic variants are accepted (variant '" + v.rsid + "' has " + std::to_string(k) + " alleles).");
      for (int a = 0; a < 2; ++a) {
        uint32_t l;
        std::memcpy(&l, fetch(pos, 4), 4);
        pos += 4;
        (a ? v.a1 : v.a0).assign((const char*)fetch(pos, l), l);
        pos += l;
      }
      v.data = pos;
      uint32_t c;
      std::memcpy(&c, fetch(pos, 4), 4);
      v.csize = c;
      pos += 4ull + c;
      if (pos > fsize_) throw std::runtime_error("bgen file ends inside the data of variant '" + v.rsid + "' : " + path);
      vars_.push_back(std::move(v));
    }
  }

  // Bytes [at, at + len) of the file as they are (the device path ships the stored zlib streams: csrc/bgen_inflate.hip)
  bool read_raw(uint64_t at, uint64_t len, void* dst) const { return fd_ >= 0 && at + len <= fsize_ && pread_all(dst, len, at); }

  // The inflated, checked probability block of variant j (layout 2: N, K, min / max ploidy, N ploidy-and-missingness bytes, phased flag,
  // bits, 2 N probability bytes), in `dst` when it is given (capacity `cap` >= block_bytes()) or in `ubuf`; returns its first byte.
  size_t block_bytes() const { return 10 + 3 * (size_t)n_; }
  const uint8_t* read_block(uint32_t j, std::vector<uint8_t>& cbuf, std::vector<uint8_t>& ubuf, uint8_t* dst = nullptr, size_t cap = 0) const {
    if (fd_ < 0) throw std::runtime_error("bgen file is closed");
    if (j >= m_) throw std::runtime_error("variant index out of range");
    const Variant& v = vars_[j];
    uint32_t c = 0, d = 0;
    if (!pread_all(&c, 4, v.data)) throw std::runtime_error("cannot read bgen file");
    uint8_t* blk;
    size_t blen;
    // a size taken from a corrupt record must not drive the allocations below: the encoding served (layout 2, biallelic,
    // diploid, 8 bits) inflates to 10 + 3 n bytes, and no biallelic diploid block (up to 32 bits) exceeds 10 + 9 n; the
    // exact layout checks (with their own messages) follow after decompression
    const uint64_t most = 64 + 16 * (uint64_t)n_;
    auto room = [&](uint64_t len) -> uint8_t* {      // a block larger than the caller's row (another encoding, trailing bytes) is inflated aside
      if (dst && len <= cap) return dst;
      ubuf.resize(len);
      return ubuf.data();
    };
    if (comp_ == 0) {
      if (c > most) throw std::runtime_error("genotype data block of variant " + v.rsid + " is larger than any biallelic diploid block of " + std::to_string(n_) + " samples");
      blk = room(c);
      if (c && !pread_all(blk, c, v.data + 4)) throw std::runtime_error("cannot read bgen file");
      blen = c;
    } else {
      if (c < 4 || !pread_all(&d, 4, v.data + 4)) throw std::runtime_error("failed to decompress genotype data block for variant: " + v.rsid);
      if (d > most) throw std::runtime_error("genotype data block of variant " + v.rsid + " is larger than any biallelic diploid block of " + std::to_string(n_) + " samples");
      if ((uint64_t)c > fsize_) throw std::runtime_error("failed to decompress genotype data block for variant: " + v.rsid);
      // (neither DEFLATE nor zstd's block format expands beyond ~1,030 : 1 / 2^17 per block byte: a length no stream of c bytes can reach is refused before the buffer for it exists)
      if (comp_ == 1 && (uint64_t)d > 1100ull * c + 64) throw std::runtime_error("failed to decompress genotype data block for variant: " + v.rsid);
      cbuf.resize(c - 4);
      if (c > 4 && !pread_all(cbuf.data(), c - 4, v.data + 8)) throw std::runtime_error("cannot read bgen file");
      blk = room(d);
      bool fail;
      if (comp_ == 1) {
        // the decoder of inflate_fast.h first (1.7x zlib's rate on these blocks); whatever it does not accept goes through zlib, whose
        // verdict is the one reported
        // (owned by the thread: the read paths run on short-lived worker threads, a bare `new` here leaked 44 KB per thread and block)
        static thread_local std::unique_ptr<rgflate::Tables> tabs(new rgflate::Tables);
        static const bool zlib_only = getenv("RG_BGEN_ZLIB") != nullptr;
        fail = zlib_only || !rgflate::inflate_zlib(blk, d, cbuf.data(), c - 4, *tabs);
        if (fail) {
          uLongf dl = d;
          fail = uncompress(blk, &dl, cbuf.data(), c - 4) != Z_OK || dl != d;
        }
      } else {
        fail = zstd()(blk, d, cbuf.data(), c - 4) != d;
      }
      if (fail) throw std::runtime_error("failed to decompress genotype data block for variant: " + v.rsid);  // Geno.cpp:1616-1617
      blen = d;
    }
    if (blen < 10ull + n_) throw std::runtime_error("malformed genotype data block for variant: " + v.rsid);
    uint32_t nind;
    uint16_t k;
    std::memcpy(&nind, blk, 4);
    std::memcpy(&k, blk + 4, 2);
    const uint8_t pmin = blk[6], pmax = blk[7];
    if (nind != n_) throw std::runtime_error("sample count of variant '" + v.rsid + "' does not match the bgen header");
    if (k != 2) throw std::runtime_error("only bi-allelic variants are accepted (variant '" + v.rsid + "').");
    if (pmin != 2 || pmax != 2) throw std::runtime_error("only diploid genotypes are supported (variant '" + v.rsid + "').");
    const uint8_t phased = blk[8 + n_], bits = blk[9 + n_];
    if (phased) throw std::runtime_error("only unphased bgen are supported.");  // Geno.cpp:66-67
    if (bits != 8) throw std::runtime_error("bgen probabilities with " + std::to_string((int)bits) + " bits are not supported (8-bit encoding is) : variant " + v.rsid);
    if (blen < 10ull + n_ + 2ull * n_) throw std::runtime_error("malformed genotype data block for variant: " + v.rsid);
    if (dst && blk != dst) { std::memcpy(dst, blk, block_bytes()); return dst; }     // trailing bytes after the probabilities: not copied
    return blk;
  }

  // Dosage row of variant j: n_samples doubles, -3 = missing.
  void read_dosages(uint32_t j, bool ref_first, double* out, std::vector<uint8_t>& cbuf, std::vector<uint8_t>& ubuf, double* info = nullptr) const {
    const uint8_t* blk = read_block(j, cbuf, ubuf);
    const uint8_t* ploidy = blk + 8;
    const uint8_t* pr = blk + 10 + n_;
    for (uint32_t i = 0; i < n_; ++i) {
      if (ploidy[i] & 0x80) { out[i] = -3.0; if (info) info[i] = 0.0; continue; }
      const double p0 = pr[2 * i] / 255.0, p1 = pr[2 * i + 1] / 255.0;
      const double p2 = std::max(1.0 - p0 - p1, 0.0);
      out[i] = ref_first ? p1 + 2.0 * p2 : p1 + 2.0 * p0;   // Geno.cpp:1672-1679
      if (info) info[i] = (ref_first ? 4.0 * p2 + p1 : 4.0 * p0 + p1) - out[i] * out[i];   // the sample's term of the IMPUTE info score (parseSnpfromBGEN, Geno.cpp:2292-2295)
    }
  }

 private:
  int fd_ = -1;
  uint64_t fsize_ = 0;
  uint32_t m_ = 0, n_ = 0;
  int comp_ = 0, layout_ = 0;
  std::vector<std::string> ids_;
  std::vector<Variant> vars_;

  bool pread_all(void* dst, uint64_t len, uint64_t off) const {
    uint8_t* d = (uint8_t*)dst;
    while (len) {
      const ssize_t r = ::pread(fd_, d, len, (off_t)off);
      if (r <= 0) return false;
      d += r;
      off += (uint64_t)r;
      len -= (uint64_t)r;
    }
    return true;
  }
  void need(uint64_t pos, uint64_t len, const std::string& path) const {
    if (pos + len > fsize_) throw std::runtime_error("bgen file ends inside a variant record : " + path);
  }
  typedef size_t (*zstd_fn)(void*, size_t, const void*, size_t);
  static zstd_fn zstd() {
    static zstd_fn fn = []() -> zstd_fn {
      void* h = dlopen("libzstd.so.1", RTLD_NOW);
      if (!h) h = dlopen("libzstd.so", RTLD_NOW);
      return h ? (zstd_fn)dlsym(h, "ZSTD_decompress") : nullptr;
    }();
    if (!fn) throw std::runtime_error("bgen file is zstd-compressed and libzstd.so.1 is not available on this machine");
    return fn;
  }
};

}  // namespace rgbgen
