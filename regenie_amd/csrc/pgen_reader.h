// PLINK2 .pgen hardcall reader: the host-side input format of regenie's `--pgen` (SURVEY.md section 8
// row a5).  It turns each variant record into the 2-bit PLINK1 .bed row the level-0 kernels consume
// (bed_prep.hip), so a pgen run feeds exactly the same bytes to the GPU as the equivalent bed run.
//
// What it replaces in the reference: PgenReader::Load / ReadHardcalls of the vendored pgenlib
// (external_libs/pgenlib/pgenlibr.cpp:37-160, :296-321), as regenie drives them from prep_pgen
// (Geno.cpp:1071-1103) and the Step-1 block reader (Geno.cpp:1793-1798).  Written from the format as
// pgenlib_read.cc parses it (header: :684-975, :1094-1640; records: :2177-2267, :2494-2595,
// :2597-2731, :2837-2900); no pgenlib code is used.
//
// Scope: storage modes 0x02 (fixed-width 2-bit), 0x10 and 0x11 (variable-width; 4- or 8-bit record
// types, 1-4 record-length bytes).  All eight main-track record types are decoded (2-bit, one-bit +
// exceptions, LD-compressed against the previous non-LD variant, inverted LD, constant + exceptions,
// all-hom-REF).  Phase tracks are stepped over (hardcalls ignore phase, as ReadHardcalls does).  Files
// with a dosage track make regenie switch to dosages (Geno.cpp:1101, :1795-1796), which the 2-bit GPU
// path does not represent: open() refuses them with a message rather than silently using hardcalls.
#pragma once
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace rgpgen {

constexpr uint32_t kVblock = 65536;      // variants per header block
constexpr uint32_t kDifflistGroup = 64;  // entries per difflist group
constexpr uint32_t kMaxDifflistDiv = 8;  // a difflist holds at most N/8 entries

// pgen code -> PLINK1 bed code in every 2-bit slot of a word: 0->11, 1->10, 2->00, 3->01, i.e.
// (hi, lo) -> (~hi, ~(hi ^ lo)).
inline uint64_t pgen_to_bed_word(uint64_t w) {
  const uint64_t m = 0x5555555555555555ull;
  const uint64_t hi = (w >> 1) & m, lo = w & m;
  return ((~hi & m) << 1) | (~(hi ^ lo) & m);
}

// 0 <-> 2 in every 2-bit slot (1 and 3 stay): (hi, lo) -> (hi ^ ~lo, lo).
inline uint64_t invert_word(uint64_t w) {
  const uint64_t m = 0x5555555555555555ull;
  const uint64_t hi = (w >> 1) & m, lo = w & m;
  return (((hi ^ ~lo) & m) << 1) | lo;
}

struct Tables {
  uint16_t spread[256];  // bit k of a byte -> bit 2k of a 16-bit word
  Tables() {
    for (int x = 0; x < 256; ++x) {
      uint16_t v = 0;
      for (int k = 0; k < 8; ++k)
        if (x & (1 << k)) v |= (uint16_t)(1u << (2 * k));
      spread[x] = v;
    }
  }
};

inline const Tables& tables() {
  static const Tables t;
  return t;
}

// Per-thread decode state: record bytes, the current variant and the cached LD base as packed pgen codes.
struct Scratch {
  std::vector<uint8_t> rec, cur, ld;
  int64_t ld_vidx = -1;
  size_t main_end = 0;  // offset in rec of the first byte after the main genotype track of the last decoded variant
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
  int max_alleles() const { return max_alleles_; }
  bool dosage_present() const { return dosage_; }
  bool phase_present() const { return phase_; }
  int64_t bytes_per_row() const { return bpr_; }

  // Parses the header.  Throws std::runtime_error with a message.
  void open(const std::string& path) {
    close();
    fd_ = ::open(path.c_str(), O_RDONLY);
    if (fd_ < 0) throw std::runtime_error("cannot open file : " + path);
    struct stat st;
    if (fstat(fd_, &st) != 0) throw std::runtime_error("cannot stat file : " + path);
    fsize_ = (uint64_t)st.st_size;
    uint8_t h[12];
    if (fsize_ < 12 || !pread_all(h, 12, 0) || h[0] != 0x6c || h[1] != 0x1b)
      throw std::runtime_error("invalid pgen file format (magic number mismatch) : " + path);
    const uint8_t mode = h[2];
    if (mode == 0x01) throw std::runtime_error("file is a PLINK1 bed file; pass it with --bed : " + path);
    if (mode == 0x03 || mode == 0x04) {
      dosage_ = true;
      throw std::runtime_error("pgen storage modes 0x03 / 0x04 (fixed-width unphased / phased dosages) are not decoded; files with per-variant "
                               "dosage tracks (modes 0x10 / 0x11) are : " + path);
    }
    if (mode != 0x02 && mode != 0x10 && mode != 0x11)
      throw std::runtime_error("pgen storage mode is not supported : " + path);
    std::memcpy(&m_, h + 3, 4);
    std::memcpy(&n_, h + 7, 4);
    if (m_ == 0 || n_ == 0 || m_ > 0x7ffffffdu || n_ > 0x7ffffffeu)
      throw std::runtime_error("invalid variant or sample count in pgen file : " + path);
    // every variant owns at least one byte of the file (a fixed-width row, or its record-length byte in the header): a damaged count is
    // refused here, before the per-variant tables (9 bytes each) are allocated for it
    if ((uint64_t)m_ > fsize_) throw std::runtime_error("invalid pgen header (more variants than bytes in the file) : " + path);
    const uint8_t ctrl = h[11];
    bpr_ = ((int64_t)n_ + 3) / 4;
    sample_id_bytes_ = (31 - __builtin_clz(n_)) / 8 + 1;
    vrtypes_.assign((size_t)m_ + 1, 0);  // one trailing zero: "is the next variant LD-compressed" reads it
    fpos_.assign((size_t)m_ + 1, 0);
    if (mode == 0x02) {  // fixed-width 2-bit records
      if (ctrl & 63) throw std::runtime_error("invalid pgen header : " + path);
      const uint64_t off = 12 + (((ctrl >> 6) == 3) ? ((uint64_t)m_ + 7) / 8 : 0);
      if (off + (uint64_t)m_ * (uint64_t)bpr_ != fsize_) throw std::runtime_error("unexpected pgen file size : " + path);
      for (uint64_t j = 0; j <= m_; ++j) fpos_[j] = off + j * (uint64_t)bpr_;
    } else {
      const uint32_t store = ctrl & 15;
      if (store & 8) throw std::runtime_error("pgen header uses a compact single-sample layout that is not supported : " + path);
      const uint32_t ac_bytes = (ctrl >> 4) & 3;
      if (ac_bytes) {  // PgenReader::Load refuses the allele-count bytes outright (pgenlibr.cpp:65-68)
        max_alleles_ = 3;
        throw std::runtime_error("Storing of allele count information is not supported (only bi-allelic variants should be present).");
      }
      const bool nonref_stored = (ctrl >> 6) == 3;
      const uint32_t nblk = (m_ + kVblock - 1) / kVblock;
      const uint32_t rl = 1 + (store & 3);
      uint64_t pos = 12, fpos = 0;
      if (!pread_all(&fpos, 8, pos)) throw std::runtime_error("cannot read pgen header : " + path);
      pos += 8ull * nblk;  // only the first block offset is needed: the record lengths give the rest
      std::vector<uint8_t> buf;
      uint32_t v0 = 0;
      for (uint32_t b = 0; b < nblk; ++b) {
        const uint32_t cnt = (m_ - v0 < kVblock) ? (m_ - v0) : kVblock;
        const uint64_t vt_bytes = (store < 4) ? ((uint64_t)cnt + 1) / 2 : cnt;
        const uint64_t need = vt_bytes + (uint64_t)cnt * rl + (nonref_stored ? ((uint64_t)cnt + 7) / 8 : 0);
        buf.resize(need);
        if (!pread_all(buf.data(), need, pos)) throw std::runtime_error("cannot read pgen header : " + path);
        pos += need;
        const uint8_t* p = buf.data();
        if (store < 4) {
          for (uint32_t j = 0; j < cnt; ++j) vrtypes_[v0 + j] = (p[j >> 1] >> (4 * (j & 1))) & 15;
        } else {
          std::memcpy(&vrtypes_[v0], p, cnt);
        }
        p += vt_bytes;
        for (uint32_t j = 0; j < cnt; ++j) {
          uint64_t len = 0;
          for (uint32_t k = 0; k < rl; ++k) len |= (uint64_t)p[(size_t)j * rl + k] << (8 * k);
          fpos_[v0 + j] = fpos;
          fpos += len;
        }
        v0 += cnt;
      }
      fpos_[m_] = fpos;
      if (pos > fpos_[0] || fpos_[m_] > fsize_) throw std::runtime_error("invalid pgen header : " + path);
      for (uint32_t j = 0; j < m_; ++j) {
        const uint8_t t = vrtypes_[j];
        if (t & 0x60) dosage_ = true;
        if (t & 0x10) phase_ = true;
        if ((t & 0x08) && max_alleles_ < 3) max_alleles_ = 3;
      }
    }
    own_ = make_scratch();
  }

  // Buffers are padded to whole 64-bit words so the row transforms can run word-wise.
  Scratch make_scratch() const {
    Scratch s;
    const size_t padded = ((size_t)bpr_ + 7) / 8 * 8 + 8;
    s.cur.assign(padded, 0);
    s.ld.assign(padded, 0);
    return s;
  }

  // One variant as a PLINK1 bed row (ceil(N/4) bytes; padding bits zero).  The const overloads take the
  // caller's Scratch, so several threads can decode from one open file (pread carries its own offset).
  void read_bed_row(uint32_t vidx, uint8_t* out) { read_bed_row(vidx, out, own_); }
  void read_bed_row(uint32_t vidx, uint8_t* out, Scratch& s) const {
    const uint8_t* g = decode(vidx, s);
    const int64_t words = bpr_ / 8;
    for (int64_t i = 0; i < words; ++i) {
      uint64_t w;
      std::memcpy(&w, g + 8 * i, 8);
      w = pgen_to_bed_word(w);
      std::memcpy(out + 8 * i, &w, 8);
    }
    if (bpr_ & 7) {
      uint64_t w;
      std::memcpy(&w, g + 8 * words, 8);  // the buffers are padded past the row
      w = pgen_to_bed_word(w);
      std::memcpy(out + 8 * words, &w, (size_t)(bpr_ & 7));
    }
    if (n_ & 3) out[bpr_ - 1] &= (uint8_t)((1u << (2 * (n_ & 3))) - 1);
  }

  // One variant as ALT-allele counts 0/1/2 and -3 for missing: ReadHardcalls(.., allele_idx = 1).
  void read_hardcalls(uint32_t vidx, double* out) { read_hardcalls(vidx, out, own_); }
  void read_hardcalls(uint32_t vidx, double* out, Scratch& s) const {
    static const double val[4] = {0.0, 1.0, 2.0, -3.0};
    const uint8_t* g = decode(vidx, s);
    for (uint32_t i = 0; i < n_; ++i) out[i] = val[(g[i >> 2] >> (2 * (i & 3))) & 3];
  }

  // One variant as PgenReader::Read(.., allele_idx = 1) gives it (pgenlibr.cpp:323-349; PgrGet1D / ParseDosage16,
  // pgenlib_read.cc:7185-7330, :7459-7497): the ALT dosage value / 16384 where the record stores one for the sample, the
  // hardcall (0/1/2, -3 missing) elsewhere.  The dosage track follows the main track and, when present, the
  // hardcall-phase track (stepped over: its length follows from the number of heterozygous calls, SkipAux2 :6819-6840).
  // Three layouts (vrtype & 0x60): 0x20 = list of sample ids + one value each, 0x40 = one value per sample (65535 = none),
  // 0x60 = one bit per sample + one value per set bit.
  void read_dosages(uint32_t vidx, double* out) { read_dosages(vidx, out, own_); }
  void read_dosages(uint32_t vidx, double* out, Scratch& s) const {
    read_hardcalls(vidx, out, s);
    const uint8_t vt = vrtypes_[vidx];
    if (!(vt & 0x60)) return;
    if (vt & 0x08) bad(vidx, "multiallelic variant");
    const uint8_t* g = ((vt & 6) == 2 || (vrtypes_[vidx + 1] & 6) != 2) ? s.cur.data() : s.ld.data();
    const uint8_t* p = s.rec.data() + s.main_end;
    const uint8_t* end = s.rec.data() + (fpos_[vidx + 1] - fpos_[vidx]);
    if (vt & 0x10) {
      uint64_t het = 0;  // heterozygous = code 1 = (lo & ~hi) in each 2-bit slot; padding slots of the last byte are masked
      for (int64_t i = 0; i < bpr_; ++i) {
        uint8_t b = g[i];
        if (i == bpr_ - 1 && (n_ & 3)) b &= (uint8_t)((1u << (2 * (n_ & 3))) - 1);
        het += (uint64_t)__builtin_popcount((unsigned)(b & ~(b >> 1) & 0x55));
      }
      uint64_t first = 1 + het / 8;
      if (het == 0) bad(vidx, "phase track on a variant without heterozygous calls");  // as ParseAux2Subset (:6743-6760)
      if ((uint64_t)(end - p) < first) bad(vidx, "phase track runs past the record");
      if (p[0] & 1) {
        uint64_t present = 0;
        for (uint64_t i = 0; i < first; ++i) present += (uint64_t)__builtin_popcount(p[i]);
        if (present < 2) bad(vidx, "phase track without a phased call");
        first += (present - 1 + 7) / 8;
        if ((uint64_t)(end - p) < first) bad(vidx, "phase track runs past the record");
      }
      p += first;
    }
    const uint8_t kind = vt & 0x60;
    const double scale = 1.0 / 16384.0;
    auto value = [&](const uint8_t* q) { return (uint16_t)(q[0] | (q[1] << 8)); };
    if (kind == 0x40) {
      if ((uint64_t)(end - p) < 2ull * n_) bad(vidx, "dosage values run past the record");
      for (uint32_t i = 0; i < n_; ++i) {
        const uint16_t v = value(p + 2 * (size_t)i);
        if (v != 65535) out[i] = v * scale;
      }
    } else if (kind == 0x60) {
      const int64_t nb = ((int64_t)n_ + 7) / 8;
      if (end - p < nb) bad(vidx, "dosage bit array runs past the record");
      const uint8_t* bits = p;
      p += nb;
      uint64_t cnt = 0;
      for (int64_t i = 0; i < nb; ++i) {
        uint8_t b = bits[i];
        if (i == nb - 1 && (n_ & 7)) b &= (uint8_t)((1u << (n_ & 7)) - 1);
        cnt += (uint64_t)__builtin_popcount(b);
      }
      if ((uint64_t)(end - p) < 2 * cnt) bad(vidx, "dosage values run past the record");
      for (uint32_t i = 0; i < n_; ++i)
        if (bits[i >> 3] & (1u << (i & 7))) { out[i] = value(p) * scale; p += 2; }
    } else {  // list: the id stream of a difflist (no replacement codes), then the values
      const uint32_t len = vint(p, end, vidx);
      if (!len) return;
      if (len > n_ / kMaxDifflistDiv) bad(vidx, "dosage list too long");
      const uint32_t groups = (len + kDifflistGroup - 1) / kDifflistGroup;
      const uint64_t index_bytes = (uint64_t)groups * (sample_id_bytes_ + 1) - 1;
      if ((uint64_t)(end - p) < index_bytes) bad(vidx, "dosage list runs past the record");
      const uint8_t* first = p;
      p += index_bytes;
      std::vector<uint32_t> ids(len);
      uint32_t k = 0;
      for (uint32_t gi = 0; gi < groups; ++gi) {
        uint64_t id = 0;
        for (uint32_t b = 0; b < sample_id_bytes_; ++b) id |= (uint64_t)first[(size_t)gi * sample_id_bytes_ + b] << (8 * b);
        const uint32_t stop = (len - k < kDifflistGroup) ? len : k + kDifflistGroup;
        for (;;) {
          if (id >= n_) bad(vidx, "dosage list sample index out of range");
          ids[k] = (uint32_t)id;
          if (++k == stop) break;
          id += vint(p, end, vidx);
        }
      }
      if ((uint64_t)(end - p) < 2ull * len) bad(vidx, "dosage values run past the record");
      for (uint32_t j = 0; j < len; ++j) out[ids[j]] = value(p + 2 * (size_t)j) * scale;
    }
  }

 private:
  int fd_ = -1;
  uint64_t fsize_ = 0;
  uint32_t m_ = 0, n_ = 0;
  int64_t bpr_ = 0;
  uint32_t sample_id_bytes_ = 1;
  int max_alleles_ = 2;
  bool dosage_ = false, phase_ = false;
  std::vector<uint8_t> vrtypes_;
  std::vector<uint64_t> fpos_;
  Scratch own_;  // state of the single-threaded entry points

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

  [[noreturn]] static void bad(uint32_t vidx, const char* what) {
    throw std::runtime_error("malformed pgen record for variant " + std::to_string((uint64_t)vidx + 1) + " (" + what + ")");
  }

  static uint32_t vint(const uint8_t*& p, const uint8_t* end, uint32_t vidx) {
    uint32_t v = 0;
    for (int shift = 0; shift <= 28; shift += 7) {
      if (p >= end) bad(vidx, "varint runs past the record");
      const uint8_t b = *p++;
      v |= (uint32_t)(b & 0x7f) << shift;
      if (!(b & 0x80)) return v;
    }
    bad(vidx, "varint too long");
  }

  // Patches g (packed pgen codes) with the difflist at p: `len`, group first ids, group byte sizes,
  // 2-bit replacement codes, then 63 varint id deltas per group.
  void apply_difflist(const uint8_t*& p, const uint8_t* end, uint8_t* g, uint32_t vidx) const {
    const uint32_t len = vint(p, end, vidx);
    if (!len) return;
    if (len > n_ / kMaxDifflistDiv) bad(vidx, "difflist too long");
    const uint32_t groups = (len + kDifflistGroup - 1) / kDifflistGroup;
    const uint64_t index_bytes = (uint64_t)groups * (sample_id_bytes_ + 1) - 1;
    const uint64_t code_bytes = ((uint64_t)len + 3) / 4;
    if ((uint64_t)(end - p) < index_bytes + code_bytes) bad(vidx, "difflist runs past the record");
    const uint8_t* first = p;
    const uint8_t* codes = p + index_bytes;
    p = codes + code_bytes;
    uint32_t k = 0;
    for (uint32_t gi = 0; gi < groups; ++gi) {
      uint64_t id = 0;
      for (uint32_t b = 0; b < sample_id_bytes_; ++b) id |= (uint64_t)first[(size_t)gi * sample_id_bytes_ + b] << (8 * b);
      const uint32_t stop = (len - k < kDifflistGroup) ? len : k + kDifflistGroup;
      for (;;) {
        if (id >= n_) bad(vidx, "difflist sample index out of range");
        const uint32_t c = (codes[k >> 2] >> (2 * (k & 3))) & 3;
        const uint32_t sh = 2 * (uint32_t)(id & 3);
        uint8_t& byte = g[id >> 2];
        byte = (uint8_t)((byte & ~(3u << sh)) | (c << sh));
        if (++k == stop) break;
        id += vint(p, end, vidx);
      }
    }
  }

  // Main genotype track of variant vidx as packed pgen codes (valid until the next call on the same Scratch).
  const uint8_t* decode(uint32_t vidx, Scratch& s) const {
    if (fd_ < 0) throw std::runtime_error("pgen file is closed");
    if (vidx >= m_) throw std::runtime_error("variant index " + std::to_string((uint64_t)vidx + 1) + " is out of range (1.." + std::to_string(m_) + ")");
    const uint32_t vt = vrtypes_[vidx] & 7;
    if ((vt & 6) == 2) {  // LD-compressed: the last earlier variant that is not, patched (and inverted for 3)
      int64_t base = (int64_t)vidx - 1;
      while (base >= 0 && (vrtypes_[base] & 6) == 2) --base;
      if (base < 0) bad(vidx, "LD-compressed variant without a base variant");
      if (s.ld_vidx != base) {
        s.ld_vidx = -1;
        decode_plain((uint32_t)base, s.ld.data(), s);
        s.ld_vidx = base;
      }
      std::memcpy(s.cur.data(), s.ld.data(), (size_t)bpr_);
      const uint8_t *p, *end;
      load_record(vidx, p, end, s);
      apply_difflist(p, end, s.cur.data(), vidx);
      s.main_end = (size_t)(p - s.rec.data());
      if (vt == 3) {
        const int64_t words = (bpr_ + 7) / 8;
        for (int64_t i = 0; i < words; ++i) {
          uint64_t w;
          std::memcpy(&w, s.cur.data() + 8 * i, 8);
          w = invert_word(w);
          std::memcpy(s.cur.data() + 8 * i, &w, 8);
        }
      }
      return s.cur.data();
    }
    if ((vrtypes_[vidx + 1] & 6) == 2) {  // the next variant will want this one as its base
      s.ld_vidx = -1;
      decode_plain(vidx, s.ld.data(), s);
      s.ld_vidx = vidx;
      return s.ld.data();
    }
    decode_plain(vidx, s.cur.data(), s);
    return s.cur.data();
  }

  void load_record(uint32_t vidx, const uint8_t*& p, const uint8_t*& end, Scratch& s) const {
    const uint64_t len = fpos_[vidx + 1] - fpos_[vidx];
    if (s.rec.size() < len + 1) s.rec.resize(len + 1);
    if (len && !pread_all(s.rec.data(), len, fpos_[vidx])) throw std::runtime_error("cannot read pgen file");
    p = s.rec.data();
    end = p + len;
  }

  void decode_plain(uint32_t vidx, uint8_t* g, Scratch& s) const {  // record types 0, 1, 4..7
    const uint32_t vt = vrtypes_[vidx] & 7;
    const uint8_t *p, *end;
    load_record(vidx, p, end, s);
    if (!(vt & 4)) {
      if (vt & 3) {  // one bit per sample picks one of two codes; exceptions follow as a difflist
        const int64_t nb = ((int64_t)n_ + 7) / 8;
        if (end - p < 1 + nb) bad(vidx, "one-bit track runs past the record");
        const uint8_t c2 = *p++;
        const uint16_t lo = (uint16_t)((c2 >> 2) * 0x5555u), dlt = c2 & 3;
        const Tables& t = tables();
        const int64_t full = bpr_ / 2;  // whole bit-bytes that map to two whole output bytes
        for (int64_t i = 0; i < full; ++i) {
          const uint16_t v = (uint16_t)(lo + t.spread[p[i]] * dlt);
          std::memcpy(g + 2 * i, &v, 2);
        }
        if (bpr_ & 1) g[bpr_ - 1] = (uint8_t)(lo + t.spread[p[full]] * dlt);
        p += nb;
        apply_difflist(p, end, g, vidx);
      } else {
        if (end - p < bpr_) bad(vidx, "2-bit track runs past the record");
        std::memcpy(g, p, (size_t)bpr_);
        p += bpr_;
      }
    } else if ((vt & 3) == 1) {
      std::memset(g, 0, (size_t)bpr_);  // every sample hom-REF; the record is empty
    } else {
      std::memset(g, (int)((vt & 3) * 0x55), (size_t)bpr_);
      apply_difflist(p, end, g, vidx);
    }
    s.main_end = (size_t)(p - s.rec.data());
  }
};

}  // namespace rgpgen
