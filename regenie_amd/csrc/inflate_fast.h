// A DEFLATE (RFC 1951) decoder for zlib streams (RFC 1950) whose inflated size is known in advance -- the genotype blocks of a BGEN file:
// at 500,000 samples each block is 1.5 MB, and inflating them is what bounds `--step 2 --bgen` end to end (the reference inflates with zlib
// under OpenMP, Geno.cpp:2219-2262; zlib 1.2.11's inflate() does 220 MB/s per thread on these blocks).  Written from the two RFCs for this
// one use:
//   * whole-buffer decoding (no streaming state, no window: matches copy from the output itself);
//   * a 64-bit bit buffer refilled with one unaligned 8-byte load, so that a length / distance pair (at most 15 + 5 + 15 + 13 bits) or
//     three literals are decoded per refill;
//   * one table look-up per symbol for codes of up to 11 bits (literal / length) and 8 bits (distance), a second one for longer codes;
//   * matches copied in 8-byte steps when the distance allows it;
//   * the unchecked loop runs while both buffers have slack, a bounds-checked copy of the same loop finishes the tail.
// The Adler-32 of the output is verified as zlib's uncompress() does.  ANY irregularity -- bad header, invalid or over-subscribed code,
// distance beyond the output so far, size mismatch, checksum mismatch -- returns false and the caller repeats the block with zlib itself,
// so error behaviour (and the message the reference prints) stays zlib's; a wrong result cannot pass the checksum.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif

namespace rgflate {

struct Entry {       // one decode-table slot
  uint8_t nbits;     // bits to consume: the Huffman code (the whole code for a second-level slot) plus, for a length / distance, its extra bits
  uint8_t kind;      // K_* | number of extra bits (length / distance) or of second-level index bits (K_SUB)
  uint16_t base;     // literal byte, length / distance base, or offset of the second-level table
};
enum : uint8_t { K_LIT = 0x00, K_LEN = 0x20, K_EOB = 0x40, K_SUB = 0x60, K_BAD = 0x80, K_MASK = 0xE0, K_XBITS = 0x1F };

constexpr int LIT_BITS = 11, DIST_BITS = 8;
constexpr int SLACK = 3 + 258 + 8 + 5;      // three literals, the longest match, the overrun of its 8-byte copies; rounded up
constexpr int LIT_SLOTS = (1 << LIT_BITS) + 288 * 16, DIST_SLOTS = (1 << DIST_BITS) + 32 * 128;

struct Tables {
  Entry lit[LIT_SLOTS];
  Entry dist[DIST_SLOTS];
};

inline uint64_t load64(const uint8_t* p) { uint64_t v; std::memcpy(&v, p, 8); return v; }

// Canonical Huffman code of `n` symbols with lengths len[] (0 = unused) -> decode table with `tb` first-level bits.  `sym_entry(s)` gives the
// (kind, base) of symbol s.  Returns false for an over-subscribed code; slots no code reaches stay K_BAD (an incomplete code is legal for a
// single distance symbol, and harmless otherwise: decoding stops at the first such slot).
template <class SymEntry>
inline bool build_table(const uint8_t* len, int n, int tb, Entry* table, int slots, SymEntry sym_entry) {
  int count[16] = {0};
  for (int s = 0; s < n; ++s) count[len[s]]++;
  count[0] = 0;
  int left = 1;
  for (int l = 1; l <= 15; ++l) { left = left * 2 - count[l]; if (left < 0) return false; }
  uint32_t next_code[16];
  uint32_t code = 0;
  for (int l = 1; l <= 15; ++l) { code = (code + count[l - 1]) << 1; next_code[l] = code; }
  const int first = 1 << tb;
  for (int i = 0; i < first; ++i) table[i] = Entry{0, K_BAD, 0};
  // longest code below every first-level prefix (bit-reversed codes index the table from the low bits)
  uint8_t sub_bits[1 << LIT_BITS];
  std::memset(sub_bits, 0, (size_t)first);
  uint32_t codes[288];
  for (int s = 0; s < n; ++s) {
    const int l = len[s];
    if (!l) continue;
    uint32_t c = next_code[l]++, rev = 0;
    for (int b = 0; b < l; ++b) rev |= ((c >> b) & 1u) << (l - 1 - b);
    codes[s] = rev;
    if (l > tb) { uint8_t& sb = sub_bits[rev & (first - 1)]; if (l - tb > sb) sb = (uint8_t)(l - tb); }
  }
  int used = first;
  for (int i = 0; i < first; ++i)
    if (sub_bits[i]) {
      const int sz = 1 << sub_bits[i];
      if (used + sz > slots) return false;
      table[i] = Entry{(uint8_t)tb, (uint8_t)(K_SUB | sub_bits[i]), (uint16_t)used};
      for (int k = 0; k < sz; ++k) table[used + k] = Entry{0, K_BAD, 0};
      used += sz;
    }
  for (int s = 0; s < n; ++s) {
    const int l = len[s];
    if (!l) continue;
    Entry e = sym_entry(s);
    e.nbits = (uint8_t)(l + ((e.kind & K_MASK) == K_LEN ? (e.kind & K_XBITS) : 0));     // code + extra bits, consumed together
    const uint32_t rev = codes[s];
    if (l <= tb) {
      for (uint32_t i = rev; i < (uint32_t)first; i += 1u << l) table[i] = e;
    } else {
      const Entry& top = table[rev & (first - 1)];
      const int sb = top.kind & K_XBITS;
      for (uint32_t i = rev >> tb; i < (1u << sb); i += 1u << (l - tb)) table[top.base + i] = e;
    }
  }
  return true;
}

inline Entry lit_entry(int s) {
  static const uint16_t lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
  static const uint8_t lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
  if (s < 256) return Entry{0, K_LIT, (uint16_t)s};
  if (s == 256) return Entry{0, K_EOB, 0};
  if (s <= 285) return Entry{0, (uint8_t)(K_LEN | lext[s - 257]), lbase[s - 257]};
  return Entry{0, K_BAD, 0};
}
inline Entry dist_entry(int s) {
  static const uint16_t dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
  static const uint8_t dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
  if (s < 30) return Entry{0, (uint8_t)(K_LEN | dext[s]), dbase[s]};
  return Entry{0, K_BAD, 0};
}

inline uint32_t adler32(const uint8_t* p, size_t n) {
  uint32_t a = 1, b = 0;
  while (n) {
    size_t k = n < 5552 ? n : 5552;      // the largest run for which the 32-bit sums cannot overflow before the reduction
    n -= k;
#if defined(__SSE2__)
    // sixteen bytes per step: a grows by their sum (psadbw), b by 16 a_before + sum (16 - i) byte_i (two pmaddwd against the weights)
    if (k >= 16) {
      const __m128i zero = _mm_setzero_si128();
      const __m128i w_lo = _mm_set_epi16(9, 10, 11, 12, 13, 14, 15, 16), w_hi = _mm_set_epi16(1, 2, 3, 4, 5, 6, 7, 8);
      __m128i va = zero, vps = zero, vb = zero;       // va, vps: two 64-bit lanes; vb: four 32-bit lanes
      const size_t chunks = k / 16;
      for (size_t c = 0; c < chunks; ++c, p += 16) {
        const __m128i x = _mm_loadu_si128((const __m128i*)p);
        vps = _mm_add_epi64(vps, va);
        va = _mm_add_epi64(va, _mm_sad_epu8(x, zero));
        vb = _mm_add_epi32(vb, _mm_madd_epi16(_mm_unpacklo_epi8(x, zero), w_lo));
        vb = _mm_add_epi32(vb, _mm_madd_epi16(_mm_unpackhi_epi8(x, zero), w_hi));
      }
      uint64_t la[2], lp[2];
      uint32_t lb[4];
      _mm_storeu_si128((__m128i*)la, va); _mm_storeu_si128((__m128i*)lp, vps); _mm_storeu_si128((__m128i*)lb, vb);
      const uint64_t sum = la[0] + la[1], ps = lp[0] + lp[1], bw = (uint64_t)lb[0] + lb[1] + lb[2] + lb[3];
      b = (uint32_t)((b + (uint64_t)a * (16 * chunks) + 16 * ps + bw) % 65521u);
      a = (uint32_t)((a + sum) % 65521u);
      k -= 16 * chunks;
    }
#endif
    while (k--) { a += *p++; b += a; }
    a %= 65521u; b %= 65521u;
  }
  return (b << 16) | a;
}

// Bit reader over [p, end): `buf` holds `cnt` valid bits.  Past the end it feeds zeros and counts them (`over`), which the block loop
// turns into a failure.
struct Bits {
  const uint8_t* p;
  const uint8_t* end;
  uint64_t buf = 0;
  int cnt = 0;
  int over = 0;      // bytes of zero padding consumed past the end
  inline void refill_fast() {                 // needs p + 8 <= end
    buf |= load64(p) << cnt;
    p += (63 - cnt) >> 3;
    cnt |= 56;
  }
  inline void refill_safe() {
    while (cnt <= 56) {
      if (p < end) buf |= (uint64_t)*p++ << cnt;
      else ++over;
      cnt += 8;
    }
  }
  inline uint32_t peek(int n) const { return (uint32_t)(buf & ((1ull << n) - 1)); }
  inline void drop(int n) { buf >>= n; cnt -= n; }
};

// One compressed block's symbols.  FAST: no bounds checks -- the caller guarantees 8 readable input bytes at every refill and room for
// SLACK output bytes, and re-enters with FAST = false near either end.  Returns 1 at end of block, 0 to be re-entered (FAST only), -1 on error.
template <bool FAST>
inline int run_block(Bits& br_io, uint8_t* out0, uint8_t*& outp, uint8_t* out_end, const Tables& t) {
  // the reader and the output cursor live in locals (registers): the byte stores to `out` may alias anything reachable through a reference
  Bits br = br_io;
  uint8_t* out = outp;
  const uint8_t* in_stop = br.end - 8;
  const Entry* const lit = t.lit;
  const Entry* const dtab = t.dist;
  int rc;
#define RG_LOOKUP_LIT(e)                                                                                              \
  e = lit[br.buf & ((1u << LIT_BITS) - 1)];                                                                           \
  if ((e.kind & K_MASK) == K_SUB) e = lit[e.base + ((br.buf >> LIT_BITS) & ((1u << (e.kind & K_XBITS)) - 1))];
  for (;;) {
    if (FAST) {
      if (br.p > in_stop || out_end - out < SLACK) { rc = 0; break; }
      br.refill_fast();
    } else {
      br.refill_safe();
      if (br.over > 8) { rc = -1; break; }
    }
    Entry e;
    RG_LOOKUP_LIT(e)
    if (e.kind == K_LIT) {
      // up to three literals per refill (3 x 15 bits <= 56)
      br.drop(e.nbits);
      if (!FAST && out >= out_end) { rc = -1; break; }
      *out++ = (uint8_t)e.base;
      RG_LOOKUP_LIT(e)
      if (e.kind == K_LIT) {
        br.drop(e.nbits);
        if (!FAST && out >= out_end) { rc = -1; break; }
        *out++ = (uint8_t)e.base;
        RG_LOOKUP_LIT(e)
        if (e.kind == K_LIT) {
          br.drop(e.nbits);
          if (!FAST && out >= out_end) { rc = -1; break; }
          *out++ = (uint8_t)e.base;
          continue;
        }
      }
      // a length / end-of-block code follows the literals: it needs up to 48 fresh bits
      if (FAST) { if (br.p > in_stop) { rc = 0; break; } br.refill_fast(); }
      else { br.refill_safe(); if (br.over > 8) { rc = -1; break; } }
    }
    const uint8_t kind = e.kind & K_MASK;
    if (kind == K_EOB) { br.drop(e.nbits); rc = 1; break; }
    if (kind != K_LEN) { rc = -1; break; }
    // `nbits` of a length / distance slot counts the code AND its extra bits: the bit buffer -- the loop-carried dependency -- moves on with
    // one shift per look-up, the extra bits are picked out of the saved copy off that chain
    const int lx = e.kind & K_XBITS;
    const uint32_t length = e.base + ((uint32_t)(br.buf >> (e.nbits - lx)) & ((1u << lx) - 1));
    br.drop(e.nbits);
    Entry d = dtab[br.buf & ((1u << DIST_BITS) - 1)];
    if ((d.kind & K_MASK) == K_SUB) d = dtab[d.base + ((br.buf >> DIST_BITS) & ((1u << (d.kind & K_XBITS)) - 1))];
    if ((d.kind & K_MASK) != K_LEN) { rc = -1; break; }
    const int dx = d.kind & K_XBITS;
    const uint32_t dist = d.base + ((uint32_t)(br.buf >> (d.nbits - dx)) & ((1u << dx) - 1));
    br.drop(d.nbits);
    if (dist > (size_t)(out - out0)) { rc = -1; break; }
    const uint8_t* src = out - dist;
    if (FAST) {
      uint8_t* const stop = out + length;
      if (dist >= 8) {
        do { std::memcpy(out, src, 8); out += 8; src += 8; } while (out < stop);      // may run up to 7 bytes past `stop`: inside the slack
        out = stop;
      } else if (dist == 1) {
        std::memset(out, *src, length);
        out = stop;
      } else {
        do { *out++ = *src++; } while (out < stop);
      }
    } else {
      if (length > (size_t)(out_end - out)) { rc = -1; break; }
      for (uint32_t k = 0; k < length; ++k) out[k] = src[k];
      out += length;
    }
  }
#undef RG_LOOKUP_LIT
  br_io = br;
  outp = out;
  return rc;
}

inline bool fixed_tables(Tables& t) {
  uint8_t len[288];
  for (int s = 0; s < 144; ++s) len[s] = 8;
  for (int s = 144; s < 256; ++s) len[s] = 9;
  for (int s = 256; s < 280; ++s) len[s] = 7;
  for (int s = 280; s < 288; ++s) len[s] = 8;
  if (!build_table(len, 288, LIT_BITS, t.lit, LIT_SLOTS, lit_entry)) return false;
  uint8_t dl[32];
  for (int s = 0; s < 32; ++s) dl[s] = 5;
  return build_table(dl, 32, DIST_BITS, t.dist, DIST_SLOTS, dist_entry);
}

// The code-length code of a dynamic block (RFC 1951 3.2.7) -> the two tables.
inline bool dynamic_tables(Bits& br, Tables& t) {
  br.refill_safe();
  const int hlit = (int)br.peek(5) + 257; br.drop(5);
  const int hdist = (int)br.peek(5) + 1; br.drop(5);
  const int hclen = (int)br.peek(4) + 4; br.drop(4);
  if (hlit > 286 || hdist > 30) return false;
  static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
  uint8_t cl[19] = {0};
  for (int i = 0; i < hclen; ++i) {
    br.refill_safe();
    cl[order[i]] = (uint8_t)br.peek(3);
    br.drop(3);
  }
  Entry clt[1 << 7];
  if (!build_table(cl, 19, 7, clt, 1 << 7, [](int s) { return Entry{0, K_LIT, (uint16_t)s}; })) return false;
  uint8_t len[288 + 32];
  int i = 0;
  const int total = hlit + hdist;
  while (i < total) {
    br.refill_safe();
    if (br.over > 8) return false;
    const Entry e = clt[br.buf & 127];
    if (e.kind != K_LIT) return false;
    br.drop(e.nbits);
    const int s = e.base;
    if (s < 16) { len[i++] = (uint8_t)s; continue; }
    int rep;
    uint8_t v = 0;
    if (s == 16) {
      if (i == 0) return false;
      v = len[i - 1];
      rep = 3 + (int)br.peek(2); br.drop(2);
    } else if (s == 17) { rep = 3 + (int)br.peek(3); br.drop(3); }
    else { rep = 11 + (int)br.peek(7); br.drop(7); }
    if (i + rep > total) return false;
    while (rep--) len[i++] = v;
  }
  if (len[256] == 0) return false;        // no end-of-block code
  uint8_t ll[288] = {0}, dl[32] = {0};
  std::memcpy(ll, len, (size_t)hlit);
  std::memcpy(dl, len + hlit, (size_t)hdist);
  return build_table(ll, 288, LIT_BITS, t.lit, LIT_SLOTS, lit_entry) && build_table(dl, 32, DIST_BITS, t.dist, DIST_SLOTS, dist_entry);
}

// zlib stream [src, src + slen) -> exactly dlen bytes at dst.  `dst` must have at least dlen bytes; nothing is written past dst + dlen.
// False = not decoded (the output is then unspecified): repeat with zlib.
inline bool inflate_zlib(uint8_t* dst, size_t dlen, const uint8_t* src, size_t slen, Tables& t) {
  if (slen < 6) return false;
  const unsigned cmf = src[0], flg = src[1];
  if ((cmf & 15) != 8 || (cmf >> 4) > 7 || ((cmf << 8) | flg) % 31 != 0 || (flg & 0x20)) return false;      // deflate, window <= 32 KB, no preset dictionary
  Bits br;
  br.p = src + 2;
  br.end = src + slen - 4;       // the Adler-32 follows the deflate data
  uint8_t* out = dst;
  uint8_t* const out_end = dst + dlen;
  bool fixed_ready = false, last = false;
  while (!last) {
    br.refill_safe();
    if (br.over > 8) return false;
    last = br.peek(1);
    const uint32_t type = (br.buf >> 1) & 3;
    br.drop(3);
    if (type == 0) {                                   // stored: skip to the byte boundary, LEN, NLEN, bytes
      br.drop(br.cnt & 7);
      // un-read the whole bytes still in the bit buffer (the newest `over` of them are padding, not stream bytes)
      if ((br.cnt >> 3) < br.over) return false;
      const uint8_t* q = br.p - ((br.cnt >> 3) - br.over);
      br.buf = 0; br.cnt = 0; br.over = 0;
      if ((size_t)(br.end - q) < 4) return false;
      const uint32_t n = q[0] | (q[1] << 8), nn = q[2] | (q[3] << 8);
      if ((n ^ nn) != 0xFFFFu) return false;
      q += 4;
      if ((size_t)(br.end - q) < n || (size_t)(out_end - out) < n) return false;
      if (n) std::memcpy(out, q, n);
      out += n;
      br.p = q + n;
      continue;
    }
    if (type == 3) return false;
    if (type == 1) { if (!fixed_tables(t)) return false; fixed_ready = true; }
    else if (!dynamic_tables(br, t)) return false;
    (void)fixed_ready;
    for (;;) {
      int rc = 0;
      if (br.end - br.p >= 16 && out_end - out >= 2 * SLACK) rc = run_block<true>(br, dst, out, out_end, t);
      if (rc == 0) rc = run_block<false>(br, dst, out, out_end, t);
      if (rc < 0) return false;
      if (rc == 1) break;
    }
  }
  if (out != out_end) return false;
  if (br.over > 0 && (int)(br.cnt >> 3) < br.over) return false;     // symbols were decoded from padding, not from the stream
  const uint8_t* a = src + slen - 4;
  const uint32_t want = ((uint32_t)a[0] << 24) | ((uint32_t)a[1] << 16) | ((uint32_t)a[2] << 8) | a[3];
  return adler32(dst, dlen) == want;
}

}  // namespace rgflate
