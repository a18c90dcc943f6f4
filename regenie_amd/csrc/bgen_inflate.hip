// BGEN v1.2 genotype blocks inflated and walked ON THE DEVICE (include/rg_bgen.h, "device path").
//
// The reference inflates one variant's block per OpenMP thread with zlib and walks the probabilities on the host (parseSnpfromBGEN,
// src/Geno.cpp:2186-2330; uncompress at :2200-2210).  At 500,000 samples a block is 1.5 MB and a `--step 2 --bgen` run is bound by that
// host work (32 workers x 2.7 ms per block against 9 ms of device work per 400 variants: profiles/r4_step2_bgen_500k.md).  Here the host
// only reads the stored bytes: the zlib streams of a batch of variants cross PCIe as they are (0.4 MB instead of 1 MB of dosages per variant
// at 500,000 samples) and
//   k_bgen_inflate   decodes them, ONE STREAM PER WAVE: DEFLATE (RFC 1951) is a serial format -- every code's position depends on the
//                    previous code's length, every match on the output so far -- so the parallelism is the batch (400 - 4,000 independent
//                    streams per launch).  The decode state (bit buffer, positions) is wave-uniform and lives in SGPRs; the Huffman tables
//                    (two-level, 10 / 9 first-level bits, zlib's ENOUGH bounds) and the last 4 KB of output sit in the wave's slice of
//                    LDS; the input arrives 256 bytes per global load (one dword per lane), matches are copied by all 64 lanes, the ring
//                    is flushed to memory in 16-byte pieces per lane; a match that reaches further back than the ring reads the flushed
//                    output.  Table construction (canonical codes from the code lengths) is lane-parallel (ballot / popcount ranks,
//                    strided fills).  The symbols of a block are decoded 64 bit positions at a time: every lane decodes the symbol that
//                    WOULD start at its bit (table look-ups of all lanes in flight together), the wave then walks the chain of true
//                    starts with one v_readlane per symbol.  The kernel is bound by scalar issue (one scalar or branch instruction per
//                    SIMD and four clocks, three waves per SIMD; profiles/r5_bgen_decoder_pmc.md), so that walk is written for
//                    instruction count: ~50 scalar + branch instructions per symbol on genotype streams (263,000 symbols per 1.5 MB
//                    block at zlib level 1: four in five are matches of 3 - 8 bytes, one match in four further back than the ring).
//   k_bgen_check     the block's header fields (N, K = 2, ploidy 2 / 2, unphased, 8 bits: the checks of bgen_reader.h) and the
//                    Adler-32 of the inflated bytes against the stream's trailer (what zlib's uncompress() verifies)
//   k_bgen_walk      probabilities -> integer dosages (uint16 rows in units of 1 / 255, the input of the digit-plane kernels of
//                    step2_qt.hip / step2_bt.hip) and the per-variant sums of parseSnpfromBGEN as EXACT integers: sum q, sum (255 (4 b0 +
//                    b1) - q^2) = 65025 x the IMPUTE-info numerator, observed samples, and their per-trait corrections.  The reference
//                    accumulates the same quantities as doubles in sample order; its result is the exact value up to its own rounding
//                    (~1e-13 relative at 500,000 samples), so the exact sums agree with it to every printed digit.
// Anything irregular (bad code, distance before the start, size or checksum mismatch, unsupported header) sets the variant's status; the
// caller repeats such a variant on the host route, whose messages are the reference's.  Product code: nothing here references oracle/.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rg_bgen.h"

namespace {

constexpr int LIT_TB = 10, DIST_TB = 9;
constexpr int LIT_SLOTS = 1408, DIST_SLOTS = 640;     // zlib's ENOUGH for (286, 10, 15) = 1332 and (30, 9, 15) = 592
constexpr int RING = 4096;                            // bytes of output history kept in LDS per stream (12.7 KB of LDS per wave with the tables: 12 waves per CU)
constexpr int WPB = 1;                                // streams (waves) per workgroup: 17 KB of LDS each, nine per CU
enum : uint32_t { K_LIT = 0x00, K_LEN = 0x20, K_EOB = 0x40, K_SUB = 0x60, K_BAD = 0x80, K_MASK = 0xE0, K_XBITS = 0x1F };
// entry: bits 0-7 nbits (code + extra bits), 8-15 kind | extra/sub bits, 16-31 base
__device__ __forceinline__ uint32_t mk_entry(uint32_t nbits, uint32_t kind, uint32_t base) { return nbits | (kind << 8) | (base << 16); }

struct WaveLds {
  uint32_t lit[LIT_SLOTS];
  uint32_t dist[DIST_SLOTS];
  uint8_t ring[RING];
  uint8_t lens[352];       // code lengths of the block being set up: [0, 19) the code-length code, [32, 32 + 286 + 30) the two alphabets
  uint32_t count[16];
  uint32_t next[16];
  uint32_t used;
};

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

__constant__ uint16_t c_lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
__constant__ uint8_t c_lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
__constant__ uint16_t c_dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
__constant__ uint8_t c_dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
__constant__ uint8_t c_clorder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// (kind | extra bits, base) of symbol s of the literal / length alphabet (mode 0), the distance alphabet (1), the code-length alphabet (2)
__device__ __forceinline__ void sym_entry(int mode, int s, uint32_t& kind, uint32_t& base) {
  if (mode == 0) {
    if (s < 256) { kind = K_LIT; base = (uint32_t)s; }
    else if (s == 256) { kind = K_EOB; base = 0; }
    else if (s <= 285) { kind = K_LEN | c_lext[s - 257]; base = c_lbase[s - 257]; }
    else { kind = K_BAD; base = 0; }
  } else if (mode == 1) {
    if (s < 30) { kind = K_LEN | c_dext[s]; base = c_dbase[s]; }
    else { kind = K_BAD; base = 0; }
  } else { kind = K_LIT; base = (uint32_t)s; }
}

// Canonical Huffman code of the n symbols with lengths L.lens[s0 + s] -> two-level decode table with tb first-level bits.  All 64 lanes of
// the wave take part; returns false (in every lane) for an over-subscribed code or a table that does not fit.  Slots no code reaches stay
// K_BAD.  maxlen: 15 (7 for the code-length code).
__device__ __noinline__ bool build_table_impl(WaveLds& L, int s0, int n, int tb, uint32_t* table, int slots, int mode_in) {
  const int mode = mode_in & 3;
  const bool fixed_code = (mode_in & 4) != 0;      // the fixed distance code of RFC 1951 (30 codes of 5 bits) is incomplete by definition: zlib never runs it through inflate_table
  const int lane = lane_id();
  const uint32_t first = 1u << tb;
  if (lane < 16) L.count[lane] = 0;
  for (uint32_t i = lane; i < first; i += 64) table[i] = 0;      // 0 = "no long code below this prefix yet"; finalised below
  __builtin_amdgcn_wave_barrier();
  for (int s = lane; s < n; s += 64) {
    const int l = L.lens[s0 + s];
    if (l) atomicAdd(&L.count[l], 1u);
  }
  __builtin_amdgcn_wave_barrier();
  {
    int left = 1, maxl = 0;
    uint32_t code = 0, prev = 0;
    bool over = false;
    for (int l = 1; l <= 15; ++l) {
      const uint32_t c = uni(L.count[l]);
      left = left * 2 - (int)c;
      if (left < 0) over = true;
      if (c) maxl = l;
      code = (code + prev) << 1;
      prev = c;
      if (lane == 0) L.next[l] = code;
    }
    if (over) return false;
    // an INCOMPLETE set is refused like zlib's inflate_table does (inftrees.c: `left > 0 && (type == CODES || max != 1)`): only a
    // literal / length or distance code that consists of ONE code of one bit may leave slots unreached.  The host route and the reference
    // then give the verdict on such a stream ('failed to decompress'); before round 6 only the Adler-32 check stood behind it.
    if (left > 0 && maxl > 0 && !fixed_code && (mode == 2 || maxl != 1)) return false;
  }
  __builtin_amdgcn_wave_barrier();
  // codes in symbol order: rank of a symbol among the symbols of its length = popcount of the lanes before it with that length, plus
  // what earlier chunks of 64 symbols used (L.next advances chunk by chunk)
  uint32_t mycode[5], mylen[5];
#pragma unroll
  for (int ch = 0; ch < 5; ++ch) {
    const int s = ch * 64 + lane;
    const int l = (s < n) ? L.lens[s0 + s] : 0;
    uint32_t code = 0;
    for (int ll = 1; ll <= 15; ++ll) {
      const uint64_t m = __ballot(l == ll);
      if (m == 0) continue;                                       // uniform
      const uint32_t base = uni(L.next[ll]);
      if (l == ll) code = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
      __builtin_amdgcn_wave_barrier();
      if (lane == 0) L.next[ll] = base + (uint32_t)__popcll(m);
      __builtin_amdgcn_wave_barrier();
    }
    mylen[ch] = (uint32_t)l;
    mycode[ch] = l ? (__brev(code) >> (32 - l)) : 0;              // bit-reversed: the stream delivers a code's first bit in the lowest position
    if (ch * 64 + 64 >= n) {
#pragma unroll
      for (int c2 = ch + 1; c2 < 5; ++c2) { mylen[c2] = 0; mycode[c2] = 0; }
      break;
    }
  }
  // longest code below every first-level prefix
#pragma unroll
  for (int ch = 0; ch < 5; ++ch)
    if ((int)mylen[ch] > tb) atomicMax(&table[mycode[ch] & (first - 1)], mylen[ch] - (uint32_t)tb);
  __builtin_amdgcn_wave_barrier();
  // second-level tables: sizes 2^sub_bits, placed in prefix order (a scan over the first-level slots, 1024 / 64 = 16 per lane)
  {
    const int per = (int)first / 64;
    uint32_t mine = 0;
    for (int i = 0; i < per; ++i) { const uint32_t sb = table[lane * per + i]; mine += sb ? (1u << sb) : 0u; }
    uint32_t incl = mine;
    for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(incl, o); if (lane >= o) incl += t; }
    const uint32_t total = uni(__shfl(incl, 63));
    if (first + total > (uint32_t)slots) return false;
    uint32_t at = first + incl - mine;
    for (int i = 0; i < per; ++i) {
      const uint32_t sb = table[lane * per + i];
      if (sb) { table[lane * per + i] = mk_entry((uint32_t)tb, K_SUB | sb, at); at += 1u << sb; }
      else table[lane * per + i] = mk_entry(0, K_BAD, 0);
    }
    __builtin_amdgcn_wave_barrier();
    for (uint32_t i = first + lane; i < first + total; i += 64) table[i] = mk_entry(0, K_BAD, 0);
  }
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int ch = 0; ch < 5; ++ch) {
    const uint32_t l = mylen[ch];
    if (!l) continue;
    uint32_t kind, base;
    sym_entry(mode, ch * 64 + lane, kind, base);
    const uint32_t e = mk_entry(l + (((kind & K_MASK) == K_LEN) ? (kind & K_XBITS) : 0u), kind, base);
    const uint32_t rev = mycode[ch];
    if ((int)l <= tb) {
      for (uint32_t i = rev; i < first; i += 1u << l) table[i] = e;
    } else {
      const uint32_t top = table[rev & (first - 1)];
      const uint32_t sb = (top >> 8) & K_XBITS, off = top >> 16;
      for (uint32_t i = rev >> tb; i < (1u << sb); i += 1u << (l - tb)) table[off + i] = e;
    }
  }
  __builtin_amdgcn_wave_barrier();
  return true;
}

// (a call's result is "divergent" to the compiler; it is the same in every lane)
__device__ __forceinline__ bool build_table(WaveLds& L, int s0, int n, int tb, uint32_t* table, int slots, int mode) {
  return uni(build_table_impl(L, s0, n, tb, table, slots, mode) ? 1u : 0u) != 0u;
}

struct InflateArgs {
  const uint8_t* comp;      // device: the streams back to back
  const int64_t* off;       // [nvar] byte offset of a variant's zlib stream in comp
  const int32_t* clen;      // [nvar] its length
  const int32_t* ulen;      // [nvar] inflated length the file states
  uint8_t* out;             // [nvar][stride]
  int64_t stride;
  int32_t* status;          // [nvar]
  int nvar;
};

// wave-uniform bit reader over the stream's dwords: `in` holds dword (chunk * 64 + lane) of the aligned stream, `nxt` the following chunk
struct BitIn {
  const uint32_t* base;     // aligned start
  uint32_t ndw;             // dwords that may be read (the stream's bytes rounded up; reads past the end return zeros)
  uint32_t w;               // next dword index
  uint32_t cur, nxt;        // this lane's dword of chunk (w >> 6) and of the next chunk
  uint64_t buf;
  uint32_t cnt;
};
__device__ __forceinline__ uint32_t load_dw(const BitIn& b, uint32_t idx) { return idx < b.ndw ? b.base[idx] : 0u; }
__device__ __forceinline__ void bits_init(BitIn& b, const uint8_t* p, uint32_t nbytes) {
  const uintptr_t a = ((uintptr_t)uni((uint32_t)((uintptr_t)p >> 32)) << 32) | (uintptr_t)uni((uint32_t)(uintptr_t)p);
  const uint32_t mis = (uint32_t)(a & 3);
  b.base = (const uint32_t*)(a - mis);
  b.ndw = (mis + nbytes + 3) >> 2;
  b.w = 0;
  b.cur = load_dw(b, (uint32_t)lane_id());
  b.nxt = load_dw(b, 64u + (uint32_t)lane_id());
  b.buf = 0; b.cnt = 0;
  // first two dwords, minus the bytes before the stream
  const uint32_t d0 = (uint32_t)__builtin_amdgcn_readlane((int)b.cur, 0), d1 = (uint32_t)__builtin_amdgcn_readlane((int)b.cur, 1);
  b.buf = ((uint64_t)d0 | ((uint64_t)d1 << 32)) >> (8 * mis);
  b.cnt = 64 - 8 * mis;
  b.w = 2;
}
// at least 33 bits afterwards
__device__ __forceinline__ void bits_refill(BitIn& b) {
  if (b.cnt <= 32) {
    const uint32_t d = (uint32_t)__builtin_amdgcn_readlane((int)b.cur, (int)(b.w & 63));
    b.buf |= (uint64_t)d << b.cnt;
    b.cnt += 32;
    b.w += 1;
    if ((b.w & 63) == 0) {       // the chunk is used up: the prefetched one takes its place, the one after it is requested
      b.cur = b.nxt;
      b.nxt = load_dw(b, b.w + 64u + (uint32_t)lane_id());
    }
  }
}
__device__ __forceinline__ uint32_t bits_peek(const BitIn& b, uint32_t n) { return (uint32_t)b.buf & ((1u << n) - 1u); }
__device__ __forceinline__ void bits_drop(BitIn& b, uint32_t n) { b.buf >>= n; b.cnt -= n; }
__device__ __forceinline__ uint32_t bits_take(BitIn& b, uint32_t n) { const uint32_t v = bits_peek(b, n); bits_drop(b, n); return v; }

// dword `idx` of the stream, out of the two chunk registers (chunk = idx >> 6 is `chunk` or `chunk + 1`); wave-uniform
__device__ __forceinline__ uint32_t dw_at(uint32_t cur, uint32_t nxt, uint32_t chunk, uint32_t idx) {
  const uint32_t rel = idx - 64u * chunk;
  return rel < 64u ? (uint32_t)__builtin_amdgcn_readlane((int)cur, (int)(rel & 63u)) : (uint32_t)__builtin_amdgcn_readlane((int)nxt, (int)((rel - 64u) & 63u));
}
// the serial reader positioned at absolute bit `bp` (fresh chunk loads: used once per DEFLATE block, after the window decoder)
__device__ __forceinline__ void bits_seek(BitIn& b, uint32_t bp) {
  const uint32_t w0 = bp >> 5, chunk = w0 >> 6, lane = (uint32_t)lane_id();
  b.cur = load_dw(b, 64u * chunk + lane);
  b.nxt = load_dw(b, 64u * chunk + 64u + lane);
  const uint32_t d0 = dw_at(b.cur, b.nxt, chunk, w0), d1 = dw_at(b.cur, b.nxt, chunk, w0 + 1u);
  b.buf = ((uint64_t)d0 | ((uint64_t)d1 << 32)) >> (bp & 31u);
  b.cnt = 64u - (bp & 31u);
  b.w = w0 + 2u;
  if ((b.w >> 6) != chunk) { b.cur = b.nxt; b.nxt = load_dw(b, 64u * (chunk + 2u) + lane); }
}

enum { ST_OK = 0, ST_HEADER = 1, ST_BTYPE = 2, ST_STORED = 3, ST_TABLE = 4, ST_CODE = 5, ST_DIST = 6, ST_OVERRUN = 7, ST_SHORT = 8, ST_INPUT = 9 };

struct OutState {
  uint8_t* out;       // the variant's inflated block
  uint32_t pos;       // bytes produced
  uint32_t flushed;   // bytes already written to `out` (a multiple of 16 until the end)
  uint32_t cap;       // inflated length expected
};
// ring -> memory, 16 bytes per lane, for everything below `upto` (a multiple of 16, or the final position)
__device__ __forceinline__ void ring_flush(WaveLds& L, OutState& o, uint32_t upto) {
  const int lane = lane_id();
  for (uint32_t p = o.flushed + 16u * lane; p + 16u <= upto; p += 16u * 64u)
    *reinterpret_cast<uint4*>(o.out + p) = *reinterpret_cast<const uint4*>(&L.ring[p & (RING - 1)]);
  uint32_t done = o.flushed + ((upto - o.flushed) & ~15u);
  if (done < upto) {          // the tail of the stream
    const uint32_t p = done + lane;
    if (p < upto) o.out[p] = L.ring[p & (RING - 1)];
    done = upto;
  }
  o.flushed = done;
}

// bytes [pos, pos + len) := bytes [pos - dist, pos - dist + len), by all lanes (the caller has checked dist <= pos and pos + len <= cap)
__device__ __forceinline__ void emit_match(WaveLds& L, OutState& o, uint32_t len, uint32_t dist) {
  const int lane = lane_id();
  __builtin_amdgcn_wave_barrier();
  if (dist <= RING - 64) {
    // from the ring
    if (dist >= len || dist >= 64) {
      // the source lies before the bytes being written (or, dist >= 64, before the 64-byte piece being written): plain offsets
      for (uint32_t c0 = 0; c0 < len; c0 += 64) {
        const uint32_t i = c0 + lane;
        if (i < len) L.ring[(o.pos + i) & (RING - 1)] = L.ring[(o.pos + i - dist) & (RING - 1)];
        __builtin_amdgcn_wave_barrier();
      }
    } else {
      // an overlapping match of short distance repeats the last `dist` bytes: lane i reads byte i mod dist of them (a float quotient:
      // i <= 321, dist < 64, exact)
      for (uint32_t c0 = 0; c0 < len; c0 += 64) {
        const uint32_t i = c0 + lane;
        if (i < len) {
          const uint32_t so = (i - dist * (uint32_t)((float)i / (float)dist)) - dist;
          L.ring[(o.pos + i) & (RING - 1)] = L.ring[(o.pos + so) & (RING - 1)];
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
  } else {
    // further back than the ring keeps (one match in ten on zlib level 1 genotype streams): those bytes were flushed to memory at
    // least RING / 2 - 600 bytes ago (pos - flushed < RING / 2 + 258 + 16).  The wave's own stores have to have reached L2
    // (s_waitcnt vmcnt(0)), and the loads go past this CU's L1 (agent-scope atomic loads of the aligned words): no cache
    // maintenance -- a fence here (write-back + invalidate per far match) made the kernel 20x slower
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (uint32_t c0 = 0; c0 < len; c0 += 64) {
      const uint32_t i = c0 + lane;
      if (i < len) {
        const uintptr_t A = (uintptr_t)(o.out + (o.pos + i - dist));
        const uint32_t wd = __hip_atomic_load((const uint32_t*)(A & ~(uintptr_t)3), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        L.ring[(o.pos + i) & (RING - 1)] = (uint8_t)(wd >> (8 * (A & 3)));
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

template <bool PAR>
__global__ __launch_bounds__(64 * WPB) void k_bgen_inflate(InflateArgs a) {
  __shared__ WaveLds lds[WPB];
  // every quantity of the decode loop is the same in all 64 lanes; uni() (v_readfirstlane) tells the compiler so, and the state then lives in
  // SGPRs, the arithmetic runs on the scalar unit and the branches are scalar branches (left in VGPRs -- an LDS load is "divergent" to the
  // compiler -- the same code spent ~1,400 cycles per symbol on vector ALU latencies and exec-mask bookkeeping)
  const int wave = (int)uni(threadIdx.x >> 6), lane = lane_id();
  const int v = blockIdx.x * WPB + wave;
  if (v >= a.nvar) return;                 // no workgroup barriers below: the waves are independent
  WaveLds& L = lds[wave];
  const uint8_t* src = a.comp + a.off[v];
  const uint32_t clen = uni((uint32_t)a.clen[v]);
  OutState o;
  o.out = a.out + (int64_t)v * a.stride; o.pos = 0; o.flushed = 0; o.cap = uni((uint32_t)a.ulen[v]);
  int st = ST_OK;
  BitIn b;
  bits_init(b, src, clen);
  // zlib header (RFC 1950): CM = 8, CINFO <= 7, FCHECK, no preset dictionary
  {
    const uint32_t cmf = bits_take(b, 8), flg = bits_take(b, 8);
    if ((cmf & 15u) != 8u || (cmf >> 4) > 7u || ((cmf << 8) | flg) % 31u != 0u || (flg & 0x20u) || clen < 6u) st = ST_HEADER;
  }
  bool last = false;
  while (st == ST_OK && !last) {
    bits_refill(b);
    last = bits_take(b, 1) != 0;
    const uint32_t btype = bits_take(b, 2);
    if (btype == 0) {
      // stored: skip to the byte boundary, LEN / NLEN, then LEN bytes
      bits_drop(b, b.cnt & 7u);
      bits_refill(b);
      const uint32_t len = bits_take(b, 16);
      bits_refill(b);
      const uint32_t nlen = bits_take(b, 16);
      if ((len ^ 0xFFFFu) != nlen) { st = ST_STORED; break; }
      if (o.pos + len > o.cap) { st = ST_OVERRUN; break; }
      for (uint32_t i = 0; i < len; ++i) {      // (stored blocks do not occur in compressed genotype data: bytes go one at a time)
        bits_refill(b);
        const uint32_t byte = bits_take(b, 8);
        if (lane == 0) L.ring[o.pos & (RING - 1)] = (uint8_t)byte;
        ++o.pos;
        if (o.pos - o.flushed >= RING / 2) { __builtin_amdgcn_wave_barrier(); ring_flush(L, o, o.pos & ~15u); }
      }
      continue;
    }
    if (btype == 3) { st = ST_BTYPE; break; }
    if (btype == 1) {
      for (int s = lane; s < 288; s += 64) L.lens[s] = (uint8_t)(s < 144 ? 8 : (s < 256 ? 9 : (s < 280 ? 7 : 8)));
      if (lane < 32) L.lens[288 + lane] = 5;
      __builtin_amdgcn_wave_barrier();
      if (!build_table(L, 0, 288, LIT_TB, L.lit, LIT_SLOTS, 0) || !build_table(L, 288, 30, DIST_TB, L.dist, DIST_SLOTS, 1 | 4)) { st = ST_TABLE; break; }
    } else {
      bits_refill(b);
      const uint32_t hlit = bits_take(b, 5) + 257u, hdist = bits_take(b, 5) + 1u, hclen = bits_take(b, 4) + 4u;
      if (hlit > 286u || hdist > 30u) { st = ST_TABLE; break; }
      if (lane < 19) L.lens[lane] = 0;
      __builtin_amdgcn_wave_barrier();
      for (uint32_t i = 0; i < hclen; ++i) {
        bits_refill(b);
        const uint32_t l = bits_take(b, 3);
        if (lane == 0) L.lens[c_clorder[i]] = (uint8_t)l;
      }
      __builtin_amdgcn_wave_barrier();
      // the code-length code decodes through the distance table's storage (7-bit first level); its lengths then move to lens[0..]
      if (!build_table(L, 0, 19, 7, L.dist, DIST_SLOTS, 2)) { st = ST_TABLE; break; }
      uint32_t i = 0, prev = 0;
      const uint32_t ntot = hlit + hdist;
      bool bad = false;
      while (i < ntot) {
        bits_refill(b);
        const uint32_t e = uni(L.dist[bits_peek(b, 7)]);
        const uint32_t nb = e & 0xFFu;
        if (((e >> 8) & K_MASK) != K_LIT || nb == 0) { bad = true; break; }
        bits_drop(b, nb);
        const uint32_t sym = e >> 16;
        uint32_t rep = 1, val = sym;
        if (sym == 16) { if (i == 0) { bad = true; break; } rep = 3 + bits_take(b, 2); val = prev; }
        else if (sym == 17) { rep = 3 + bits_take(b, 3); val = 0; }
        else if (sym == 18) { rep = 11 + bits_take(b, 7); val = 0; }
        if (i + rep > ntot) { bad = true; break; }
        // lens[32 + i ...] (the first 19 bytes still hold the code-length code's lengths)
        for (uint32_t k = lane; k < rep; k += 64) L.lens[32 + i + k] = (uint8_t)val;
        i += rep;
        prev = val;
      }
      if (bad) { st = ST_TABLE; break; }
      __builtin_amdgcn_wave_barrier();
      if (uni(L.lens[32 + 256]) == 0) { st = ST_TABLE; break; }      // no end-of-block code
      if (!build_table(L, 32, (int)hlit, LIT_TB, L.lit, LIT_SLOTS, 0) || !build_table(L, 32 + (int)hlit, (int)hdist, DIST_TB, L.dist, DIST_SLOTS, 1)) { st = ST_TABLE; break; }
    }
    // ---- the block's symbols ----
    if (PAR) {
      // Window decoder (round 5): the serial loop below issues ~120 scalar instructions and two or three dependent LDS round trips per
      // symbol (~1 us).  Here all 64 lanes decode SPECULATIVELY the symbol that would start at each of the next 64 bit positions (lane i:
      // bits bp + i ...: literal / length code, extra bits, distance code, extra bits -- at most 48 bits, out of three dwords of the
      // stream gathered across the lanes), and the wave then walks the chain of true starts through the lanes' results: ONE v_readlane
      // per symbol, whose 32 bits hold everything the walk needs --
      //   bits 0-5 the symbol's length in bits | bit 6 match | bit 7 end of block (bit 8 set) or no code (bit 8 clear)
      //   literal: bits 8-15 the byte | match: bits 8-16 the length, bits 17-31 the distance - 1.
      // The scalar unit is what the kernel runs out of (PMC: 114 scalar instructions per symbol in the first form of this loop), so the
      // walk does as little as it can: a literal is written without a bounds test (the window loop runs while 64 more bytes fit -- a window
      // holds at most 64 symbols -- and hands the block's tail to the serial loop below), an end-of-block or invalid code takes the
      // literal's path (its byte lands on a position that is not advanced), and the loop has ONE exit test (a loop with several exits and
      // lane-dependent code inside is rebuilt by the compiler around exit-selector registers: two dozen scalar moves per symbol).
      uint32_t bp = b.w * 32u - b.cnt;
      uint32_t chunk = (bp >> 5) >> 6;
      uint32_t cur = load_dw(b, 64u * chunk + (uint32_t)lane), nxt = load_dw(b, 64u * chunk + 64u + (uint32_t)lane);
      uint32_t stop = 0;                               // 1 = end of block, 2 = the block's tail goes to the serial loop, 16 + ST_* = failure
      // the window loop runs while a match of any length and the literals of a whole window still fit behind it: one test per match, none per literal
      const bool win_ok = o.cap >= 322u;
      const uint32_t lim = o.cap - 322u;
      while (win_ok && stop == 0u && o.pos <= lim) {
        const uint32_t w0 = bp >> 5;
        if (w0 >= 64u * (chunk + 1u)) { ++chunk; cur = nxt; nxt = load_dw(b, 64u * chunk + 64u + (uint32_t)lane); }
        if (w0 > b.ndw + 2u) { stop = 16u + (uint32_t)ST_INPUT; break; }
        uint32_t A;
        {
          // the three dwords the lane's 64 bits span, gathered across the lanes that hold the stream (no scalar work); only a window that
          // reaches into the next chunk (5 in 64) needs the second register
          const uint32_t bb = (bp & 31u) + (uint32_t)lane, sh = bb & 31u;
          const uint32_t base = w0 - 64u * chunk, rel = base + (bb >> 5);      // rel .. rel + 2 <= base + 4 <= 67
          uint32_t lo, mid, hi;
          lo = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(rel << 2), (int)cur);
          mid = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((rel + 1u) << 2), (int)cur);
          hi = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((rel + 2u) << 2), (int)cur);
          if (base + 4u >= 64u) {
            const uint32_t n0 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(rel << 2), (int)nxt), n1 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((rel + 1u) << 2), (int)nxt),
                           n2 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((rel + 2u) << 2), (int)nxt);
            lo = rel >= 64u ? n0 : lo; mid = rel + 1u >= 64u ? n1 : mid; hi = rel + 2u >= 64u ? n2 : hi;
          }
          const uint32_t r0 = __builtin_amdgcn_alignbit(mid, lo, sh), r1 = __builtin_amdgcn_alignbit(hi, mid, sh);
          const uint64_t win = ((uint64_t)r1 << 32) | r0;
          uint32_t e = L.lit[r0 & ((1u << LIT_TB) - 1u)];
          if (((e >> 8) & K_MASK) == K_SUB) {
            const uint32_t sb = (e >> 8) & K_XBITS;
            e = L.lit[(e >> 16) + ((r0 & ((1u << (LIT_TB + sb)) - 1u)) >> LIT_TB)];
          }
          const uint32_t kind = (e >> 8) & 0xFFu, nb = e & 0xFFu;
          A = 0x80u;                                                            // no code
          if ((kind & K_MASK) == K_LIT) A = nb | ((e >> 8) & 0xFF00u);          // the byte: bits 16-23 of the entry
          else if ((kind & K_MASK) == K_EOB) A = nb | 0x180u;
          else if ((kind & K_MASK) == K_LEN) {
            const uint32_t xb = kind & K_XBITS;
            const uint32_t len = (e >> 16) + ((r0 & ((1u << nb) - 1u)) >> (nb - xb));
            const uint32_t w2 = (uint32_t)(win >> nb);
            uint32_t d = L.dist[w2 & ((1u << DIST_TB) - 1u)];
            if (((d >> 8) & K_MASK) == K_SUB) {
              const uint32_t sb = (d >> 8) & K_XBITS;
              d = L.dist[(d >> 16) + ((w2 & ((1u << (DIST_TB + sb)) - 1u)) >> DIST_TB)];
            }
            const uint32_t dk = (d >> 8) & 0xFFu, dn = d & 0xFFu;
            if ((dk & K_MASK) == K_LEN) {
              const uint32_t dxb = dk & K_XBITS;
              const uint32_t dist = (d >> 16) + ((w2 & ((1u << dn) - 1u)) >> (dn - dxb));      // 1 .. 32,768
              // bit 7 of a match: not the common case -- <= 64 bytes out of the ring, none of them written by the match itself -- whose copy is
              // one masked LDS read + write with nothing to decide
              const uint32_t slow = (dist > RING - 64u || dist < len || len > 64u) ? 0x80u : 0u;
              A = (nb + dn) | 0x40u | slow | (len << 8) | ((dist - 1u) << 17);                     // nb + dn <= 48, len <= 258
            }
          }
        }
        __builtin_amdgcn_wave_barrier();
        uint32_t sidx = 0;
        do {
          const uint32_t a_ = (uint32_t)__builtin_amdgcn_readlane((int)A, (int)sidx);
          uint32_t adv = a_ & 0x3Fu;
          if (a_ & 0x40u) {
            const uint32_t len = (a_ >> 8) & 0x1FFu, dist = (a_ >> 17) + 1u;
            if (dist <= o.pos && o.pos <= lim) {
              if (a_ & 0x80u) emit_match(L, o, len, dist);
              else if ((uint32_t)lane < len) L.ring[(o.pos + (uint32_t)lane) & (RING - 1)] = L.ring[(o.pos + (uint32_t)lane - dist) & (RING - 1)];
              o.pos += len;
            } else {                                   // not consumed: the serial loop takes the block's tail, and reports a distance before the block
              stop = o.pos > lim ? 2u : 16u + (uint32_t)ST_DIST;
              adv = 0;
            }
          } else {
            const uint32_t special = (a_ >> 7) & 1u;
            if (lane == 0) L.ring[o.pos & (RING - 1)] = (uint8_t)(a_ >> 8);
            o.pos += special ^ 1u;
            stop = special ? ((a_ & 0x100u) ? 1u : 16u + (uint32_t)ST_CODE) : 0u;
          }
          sidx += adv;
          if (o.pos - o.flushed >= RING / 2) { __builtin_amdgcn_wave_barrier(); ring_flush(L, o, o.pos & ~15u); }
        } while (sidx < 64u && stop == 0u);
        bp += sidx;
      }
      if (stop == 0u) stop = 2u;                       // (left at the top: the tail)
      if (stop > 2u) { st = (int)(stop - 16u); break; }
      bits_seek(b, bp);
      if (stop == 1u) continue;
      // (the last bytes of the block: the serial loop, which tests every symbol against the block's size)
    }
    while (true) {
      if (b.cnt <= 32 && b.w > b.ndw + 2u) { st = ST_INPUT; break; }           // about to read past the stream (zero bits): not a valid stream
      bits_refill(b);
      uint32_t e = uni(L.lit[bits_peek(b, LIT_TB)]);
      if (((e >> 8) & K_MASK) == K_SUB) {
        const uint32_t sb = (e >> 8) & K_XBITS;
        e = uni(L.lit[(e >> 16) + ((bits_peek(b, LIT_TB + sb)) >> LIT_TB)]);
      }
      const uint32_t kind = (e >> 8) & 0xFFu, nb = e & 0xFFu;
      if ((kind & K_MASK) == K_LIT) {
        bits_drop(b, nb);
        if (o.pos >= o.cap) { st = ST_OVERRUN; break; }
        if (lane == 0) L.ring[o.pos & (RING - 1)] = (uint8_t)(e >> 16);
        ++o.pos;
      } else if ((kind & K_MASK) == K_LEN) {
        const uint32_t xb = kind & K_XBITS;
        const uint32_t len = (e >> 16) + ((bits_peek(b, nb) >> (nb - xb)));
        bits_drop(b, nb);
        bits_refill(b);
        uint32_t d = uni(L.dist[bits_peek(b, DIST_TB)]);
        if (((d >> 8) & K_MASK) == K_SUB) {
          const uint32_t sb = (d >> 8) & K_XBITS;
          d = uni(L.dist[(d >> 16) + (bits_peek(b, DIST_TB + sb) >> DIST_TB)]);
        }
        const uint32_t dk = (d >> 8) & 0xFFu, dn = d & 0xFFu;
        if ((dk & K_MASK) != K_LEN) { st = ST_CODE; break; }
        const uint32_t dxb = dk & K_XBITS;
        const uint32_t dist = (d >> 16) + (bits_peek(b, dn) >> (dn - dxb));
        bits_drop(b, dn);
        if (dist > o.pos) { st = ST_DIST; break; }
        if (o.pos + len > o.cap) { st = ST_OVERRUN; break; }
        emit_match(L, o, len, dist);
        o.pos += len;
      } else if ((kind & K_MASK) == K_EOB) {
        bits_drop(b, nb);
        break;
      } else { st = ST_CODE; break; }
      if (o.pos - o.flushed >= RING / 2) { __builtin_amdgcn_wave_barrier(); ring_flush(L, o, o.pos & ~15u); }
    }
  }
  __builtin_amdgcn_wave_barrier();
  if (st == ST_OK) {
    ring_flush(L, o, o.pos);
    if (o.pos != o.cap) st = ST_SHORT;
  }
  if (lane == 0) a.status[v] = st;
}

// ---- header fields + Adler-32 of the inflated block; one workgroup per variant ----
struct CheckArgs {
  const uint8_t* comp; const int64_t* off; const int32_t* clen; const int32_t* ulen;
  const uint8_t* out; int64_t stride; int64_t n_file; int32_t* status; int nvar;
};
enum { ST_ADLER = 10, ST_FORMAT = 11 };
// Adler-32 pieces: a = 1 + sum d_i, b = n + sum (n - i) d_i (mod 65521).  grid (ADLER_SEG, nvar): every workgroup sums a segment of the block as
// exact 64-bit integers (16 bytes per load) into the variant's two accumulators; k_bgen_check reduces them and compares.
#define ADLER_SEG 16
__global__ __launch_bounds__(256) void k_bgen_adler(CheckArgs a, unsigned long long* acc /*[nvar][2]*/) {
  const int v = blockIdx.y;
  if (a.status[v] != ST_OK) return;
  const uint8_t* p = a.out + (int64_t)v * a.stride;            // 64-byte aligned
  const uint32_t n = (uint32_t)a.ulen[v];
  const uint32_t nq = (n + 15) / 16;                            // 16-byte pieces (the block's stride is padded: bytes past n are masked)
  const uint32_t per = (nq + ADLER_SEG - 1) / ADLER_SEG;
  const uint32_t q0 = min(nq, per * blockIdx.x), q1 = min(nq, q0 + per);
  unsigned long long A = 0, B = 0;
  for (uint32_t q = q0 + threadIdx.x; q < q1; q += 256) {
    const uint4 w = *reinterpret_cast<const uint4*>(p + 16ull * q);
    const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
    uint32_t sa = 0, sj = 0;                                    // sum d_j, sum j d_j over the piece
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const uint32_t i = 16u * q + j;
      const uint32_t d = i < n ? (ws[j >> 2] >> (8 * (j & 3))) & 0xFFu : 0u;
      sa += d; sj += j * d;
    }
    A += sa;
    B += (unsigned long long)(n - 16u * q) * sa - sj;           // sum (n - i) d_i, i = 16 q + j
  }
  for (int o = 32; o > 0; o >>= 1) { A += __shfl_down(A, o); B += __shfl_down(B, o); }
  if ((threadIdx.x & 63) == 0) { atomicAdd(&acc[2 * v], A); atomicAdd(&acc[2 * v + 1], B); }
}
// one thread per variant: checksum against the stream's trailer, then the header fields bgen_reader.h checks
__global__ __launch_bounds__(64) void k_bgen_check(CheckArgs a, const unsigned long long* acc) {
  const int v = blockIdx.x * 64 + threadIdx.x;
  if (v >= a.nvar || a.status[v] != ST_OK) return;
  const uint8_t* p = a.out + (int64_t)v * a.stride;
  const uint32_t n = (uint32_t)a.ulen[v];
  const unsigned long long ta = 1ull + acc[2 * v], tb = (unsigned long long)n + acc[2 * v + 1];
  const uint32_t adler = (uint32_t)(((tb % 65521ull) << 16) | (ta % 65521ull));
  const uint8_t* tr = a.comp + a.off[v] + a.clen[v] - 4;
  const uint32_t want = ((uint32_t)tr[0] << 24) | ((uint32_t)tr[1] << 16) | ((uint32_t)tr[2] << 8) | (uint32_t)tr[3];
  int st = ST_OK;
  if (adler != want) st = ST_ADLER;
  const uint32_t N = (uint32_t)a.n_file;
  if (n < 10u + N + 2u * N) st = ST_FORMAT;
  else {
    const uint32_t nind = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
    const uint32_t k = (uint32_t)p[4] | ((uint32_t)p[5] << 8);
    if (nind != N || k != 2 || p[6] != 2 || p[7] != 2 || p[8 + N] != 0 || p[9 + N] != 8) st = ST_FORMAT;
  }
  if (st != ST_OK) a.status[v] = st;
}

// ---- probabilities -> uint16 dosages + exact integer sums; grid (ceil(n / 1024), nvar), 256 threads x 4 samples ----
struct WalkArgs {
  const uint8_t* out; int64_t stride; int64_t n_file, n; const int64_t* file_idx;   // file_idx == nullptr: identity
  const int32_t* status; int ref_first;
  uint16_t* g16; int64_t ld16;
  int P; const unsigned long long* missbits;      // [n][W]: bit p of a sample's words = trait p is MISSING for it; nullptr: no per-trait sums
  int W;                                          // words per sample = ceil(P / 64)
  long long* sum_q; long long* sum_info; long long* n_obs; int* max_q;      // [nvar]
  long long* sum_q_t; long long* sum_info_t; long long* n_obs_t;            // [nvar][P]: what the samples missing for trait p contribute
};
__device__ __forceinline__ long long wave_sum_ll(long long x) {
  for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o);
  return x;
}
__global__ __launch_bounds__(256) void k_bgen_walk(WalkArgs a) {
  extern __shared__ unsigned long long s_trait[];      // [3][P]: what this workgroup's samples contribute to the per-trait corrections
  const int v = blockIdx.y;
  uint16_t* row = a.g16 + (int64_t)v * a.ld16;
  const int64_t k0 = (int64_t)blockIdx.x * 1024 + threadIdx.x;
  if (a.status[v] != ST_OK) return;                    // (the same for the whole workgroup)
  if (a.missbits) {
    for (int i = threadIdx.x; i < 3 * a.P; i += 256) s_trait[i] = 0ull;
    __syncthreads();
  }
  const uint8_t* blk = a.out + (int64_t)v * a.stride;
  const uint8_t* ploidy = blk + 8;
  const uint8_t* pr = blk + 10 + a.n_file;
  long long sq = 0, si = 0, no = 0;
  int mq = 0;
  for (int t = 0; t < 4; ++t) {
    const int64_t k = k0 + 256 * t;
    if (k >= a.ld16) break;
    if (k >= a.n) { row[k] = 0; continue; }
    const int64_t i = a.file_idx ? a.file_idx[k] : k;
    if (ploidy[i] & 0x80) { row[k] = 0xFFFFu; continue; }
    const unsigned b0 = pr[2 * i], b1 = pr[2 * i + 1];
    // G = prob1 + 2 prob0 (prob1 + 2 prob2 with --ref-first, prob2 = max(1 - prob0 - prob1, 0): Geno.cpp:2286-2290), in units of 1 / 255;
    // the sample's term of the IMPUTE-info numerator is (4 probX + prob1) - G^2: 255 (4 bX + b1) - q^2 in units of 1 / 65025
    const unsigned bx = a.ref_first ? (b0 + b1 < 255u ? 255u - b0 - b1 : 0u) : b0;
    const unsigned q = b1 + 2u * bx;
    row[k] = (uint16_t)q;
    const long long inf = 255ll * (long long)(4u * bx + b1) - (long long)q * (long long)q;
    sq += q; si += inf; no += 1;
    mq = max(mq, (int)q);
    if (a.missbits)
      for (int w = 0; w < a.W; ++w) {
        unsigned long long mb = a.missbits[k * a.W + w];        // one load per sample (ten byte loads before): most samples have no bit set
        while (mb) {       // LDS atomics: three global atomics per missing (sample, trait) would all land on the variant's 3 P words
          const int p = 64 * w + __ffsll((long long)mb) - 1;
          mb &= mb - 1ull;
          atomicAdd(&s_trait[p], (unsigned long long)q);
          atomicAdd(&s_trait[a.P + p], (unsigned long long)inf);
          atomicAdd(&s_trait[2 * a.P + p], 1ull);
        }
      }
  }
  if (a.missbits) {
    __syncthreads();
    for (int i = threadIdx.x; i < 3 * a.P; i += 256) {
      const unsigned long long x = s_trait[i];
      if (x == 0ull) continue;
      const int which = i / a.P, p = i - which * a.P;
      long long* dst = which == 0 ? a.sum_q_t : (which == 1 ? a.sum_info_t : a.n_obs_t);
      atomicAdd((unsigned long long*)&dst[(int64_t)v * a.P + p], x);
    }
  }
  sq = wave_sum_ll(sq); si = wave_sum_ll(si); no = wave_sum_ll(no);
  for (int o = 32; o > 0; o >>= 1) mq = max(mq, __shfl_down(mq, o));
  if ((threadIdx.x & 63) == 0) {
    atomicAdd((unsigned long long*)&a.sum_q[v], (unsigned long long)sq);
    atomicAdd((unsigned long long*)&a.sum_info[v], (unsigned long long)si);
    atomicAdd((unsigned long long*)&a.n_obs[v], (unsigned long long)no);
    atomicMax(&a.max_q[v], mq);
  }
}

}  // namespace

// =========================================================================================================
// C ABI (include/rg_bgen.h, device path)
// =========================================================================================================
struct rg_bgen_dev {
  int device = 0;
  hipStream_t st = nullptr;
  std::string err;
  int64_t n_file = 0, n = 0;
  int P = 0;
  bool identity = true;
  int64_t* d_file_idx = nullptr;
  unsigned long long* d_missbits = nullptr;      // [n][ceil(P / 64)]: the traits missing per sample
  // per slot: compressed bytes, descriptors, inflated blocks, dosage rows, sums
  struct Slot {
    uint8_t* d_comp = nullptr; size_t comp_cap = 0;
    uint8_t* d_raw = nullptr; size_t raw_cap = 0;
    uint16_t* d_g16 = nullptr; size_t g16_cap = 0;
    void* d_desc = nullptr; size_t desc_cap = 0;
    void* d_sums = nullptr; size_t sums_cap = 0;
    void* h_sums = nullptr; size_t h_sums_cap = 0;
  } slot[2];
};

#define BD_HIP(call)                                                                       \
  do {                                                                                     \
    hipError_t e_ = (call);                                                                \
    if (e_ != hipSuccess) { h->err = std::string(#call) + ": " + hipGetErrorString(e_); return RG_BGEN_ERR_DEVICE; } \
  } while (0)

static int bd_ensure(rg_bgen_dev* h, void** p, size_t* cap, size_t bytes) {
  if (*cap >= bytes) return RG_BGEN_OK;
  if (*p) BD_HIP(hipFree(*p));
  *p = nullptr; *cap = 0;
  BD_HIP(hipMalloc(p, bytes + bytes / 8 + 4096));
  *cap = bytes + bytes / 8;
  return RG_BGEN_OK;
}

extern "C" {

int rg_bgen_dev_create(rg_bgen_dev** out, int32_t device) {
  if (!out) return RG_BGEN_ERR_ARG;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return RG_BGEN_ERR_DEVICE;
  rg_bgen_dev* h = new (std::nothrow) rg_bgen_dev();
  if (!h) return RG_BGEN_ERR_ARG;
  h->device = device;
  if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&h->st, hipStreamNonBlocking) != hipSuccess) { delete h; return RG_BGEN_ERR_DEVICE; }
  *out = h;
  return RG_BGEN_OK;
}

void rg_bgen_dev_destroy(rg_bgen_dev* h) {
  if (!h) return;
  hipSetDevice(h->device);
  if (h->st) { hipStreamSynchronize(h->st); hipStreamDestroy(h->st); }
  for (void* p : {(void*)h->d_file_idx, (void*)h->d_missbits}) if (p) hipFree(p);
  for (auto& s : h->slot) {
    for (void* p : {(void*)s.d_comp, (void*)s.d_raw, (void*)s.d_g16, s.d_desc, s.d_sums}) if (p) hipFree(p);
    if (s.h_sums) hipHostFree(s.h_sums);
  }
  delete h;
}

const char* rg_bgen_dev_last_error(const rg_bgen_dev* h) { return h ? h->err.c_str() : "null device decoder"; }

int rg_bgen_dev_set_samples(rg_bgen_dev* h, int64_t n_file, int64_t n, const int64_t* file_idx, int32_t P, const uint8_t* mask) {
  if (!h || n_file < 1 || n < 1 || n > n_file || P < 0) return RG_BGEN_ERR_ARG;
  if (mask && P > 2048) { h->err = "rg_bgen_dev_set_samples: more than 2,048 traits with masks (the walk keeps 3 P sums in LDS)"; return RG_BGEN_ERR_ARG; }
  hipSetDevice(h->device);
  for (void** p : {(void**)&h->d_file_idx, (void**)&h->d_missbits}) if (*p) { hipFree(*p); *p = nullptr; }
  h->n_file = n_file; h->n = n; h->P = mask ? P : 0;
  h->identity = file_idx == nullptr;
  if (file_idx) {
    for (int64_t k = 0; k < n; ++k) if (file_idx[k] < 0 || file_idx[k] >= n_file) { h->err = "rg_bgen_dev_set_samples: sample index outside the file"; return RG_BGEN_ERR_ARG; }
    BD_HIP(hipMalloc((void**)&h->d_file_idx, sizeof(int64_t) * (size_t)n));
    BD_HIP(hipMemcpy(h->d_file_idx, file_idx, sizeof(int64_t) * (size_t)n, hipMemcpyHostToDevice));
  }
  if (mask && P > 0) {
    const int W = (P + 63) / 64;
    std::vector<unsigned long long> bits((size_t)n * W, 0ull);
    for (int p = 0; p < P; ++p)
      for (int64_t k = 0; k < n; ++k) if (!mask[(size_t)p * n + k]) bits[(size_t)k * W + p / 64] |= 1ull << (p % 64);
    BD_HIP(hipMalloc((void**)&h->d_missbits, sizeof(unsigned long long) * bits.size()));
    BD_HIP(hipMemcpy(h->d_missbits, bits.data(), sizeof(unsigned long long) * bits.size(), hipMemcpyHostToDevice));
  }
  return RG_BGEN_OK;
}

int rg_bgen_dev_decode(rg_bgen_dev* h, int32_t slot, int32_t nvar, const uint8_t* comp, int64_t comp_bytes, const int64_t* off,
                       const int32_t* clen, const int32_t* ulen, int32_t ref_first, rg_bgen_dev_out* out) {
  if (!h || slot < 0 || slot > 1 || nvar < 1 || !comp || comp_bytes < 1 || !off || !clen || !ulen || !out) return RG_BGEN_ERR_ARG;
  if (h->n < 1) { h->err = "rg_bgen_dev_decode: rg_bgen_dev_set_samples first"; return RG_BGEN_ERR_ARG; }
  hipSetDevice(h->device);
  rg_bgen_dev::Slot& s = h->slot[slot];
  int64_t stride = 0;
  for (int v = 0; v < nvar; ++v) {
    if (off[v] < 0 || clen[v] < 0 || off[v] + clen[v] > comp_bytes || ulen[v] < 0) { h->err = "rg_bgen_dev_decode: stream outside the buffer"; return RG_BGEN_ERR_ARG; }
    stride = std::max<int64_t>(stride, ulen[v]);
  }
  stride = (stride + 63) / 64 * 64;
  const int64_t ld16 = (h->n + 7) / 8 * 8;
  const int P = h->P;
  int rc;
  if ((rc = bd_ensure(h, (void**)&s.d_comp, &s.comp_cap, (size_t)comp_bytes + 1024))) return rc;
  if ((rc = bd_ensure(h, (void**)&s.d_raw, &s.raw_cap, (size_t)nvar * stride))) return rc;
  if ((rc = bd_ensure(h, (void**)&s.d_g16, &s.g16_cap, sizeof(uint16_t) * (size_t)nvar * ld16))) return rc;
  // descriptors: off [nvar] i64 | clen | ulen | status [nvar] i32
  const size_t desc_bytes = (size_t)nvar * (8 + 4 + 4 + 4);
  if ((rc = bd_ensure(h, &s.d_desc, &s.desc_cap, desc_bytes))) return rc;
  // sums: sum_q, sum_info, n_obs [nvar] i64 | per-trait x3 [nvar][P] i64 | max_q [nvar] i32
  const size_t nsum = (size_t)nvar * (3 + 3 * (size_t)P);
  const size_t sums_bytes = nsum * 8 + (size_t)nvar * 4 + 4 /* pad to 8 */ + (size_t)nvar * 16 /* Adler-32 accumulators */;
  if ((rc = bd_ensure(h, &s.d_sums, &s.sums_cap, sums_bytes))) return rc;
  if (s.h_sums_cap < sums_bytes + (size_t)nvar * 4) {
    if (s.h_sums) hipHostFree(s.h_sums);
    s.h_sums = nullptr; s.h_sums_cap = 0;
    BD_HIP(hipHostMalloc(&s.h_sums, sums_bytes + (size_t)nvar * 4 + 4096));
    s.h_sums_cap = sums_bytes + (size_t)nvar * 4;
  }
  hipStream_t st = h->st;
  int64_t* d_off = (int64_t*)s.d_desc;
  int32_t* d_clen = (int32_t*)(d_off + nvar);
  int32_t* d_ulen = d_clen + nvar;
  int32_t* d_status = d_ulen + nvar;
  BD_HIP(hipMemcpyAsync(s.d_comp, comp, (size_t)comp_bytes, hipMemcpyHostToDevice, st));
  BD_HIP(hipMemcpyAsync(d_off, off, sizeof(int64_t) * nvar, hipMemcpyHostToDevice, st));
  BD_HIP(hipMemcpyAsync(d_clen, clen, sizeof(int32_t) * nvar, hipMemcpyHostToDevice, st));
  BD_HIP(hipMemcpyAsync(d_ulen, ulen, sizeof(int32_t) * nvar, hipMemcpyHostToDevice, st));
  BD_HIP(hipMemsetAsync(s.d_sums, 0, sums_bytes, st));
  InflateArgs ia{s.d_comp, d_off, d_clen, d_ulen, s.d_raw, stride, d_status, nvar};
  static const bool serial = getenv("RG_BGEN_SERIAL") != nullptr;      // the one-symbol-at-a-time decoder (kept as the cross-check of the window decoder)
  if (serial) hipLaunchKernelGGL(k_bgen_inflate<false>, dim3((unsigned)((nvar + WPB - 1) / WPB)), dim3(64 * WPB), 0, st, ia);
  else hipLaunchKernelGGL(k_bgen_inflate<true>, dim3((unsigned)((nvar + WPB - 1) / WPB)), dim3(64 * WPB), 0, st, ia);
  CheckArgs ca{s.d_comp, d_off, d_clen, d_ulen, s.d_raw, stride, h->n_file, d_status, nvar};
  unsigned long long* d_adler = (unsigned long long*)((uint8_t*)s.d_sums + ((nsum * 8 + (size_t)nvar * 4 + 7) / 8) * 8);
  hipLaunchKernelGGL(k_bgen_adler, dim3(ADLER_SEG, (unsigned)nvar), dim3(256), 0, st, ca, d_adler);
  hipLaunchKernelGGL(k_bgen_check, dim3((unsigned)((nvar + 63) / 64)), dim3(64), 0, st, ca, (const unsigned long long*)d_adler);
  long long* d_sq = (long long*)s.d_sums;
  long long* d_si = d_sq + nvar;
  long long* d_no = d_si + nvar;
  long long* d_sqt = d_no + nvar;
  long long* d_sit = d_sqt + (size_t)nvar * P;
  long long* d_not = d_sit + (size_t)nvar * P;
  int* d_mq = (int*)(d_not + (size_t)nvar * P);
  WalkArgs wa{s.d_raw, stride, h->n_file, h->n, h->d_file_idx, d_status, ref_first ? 1 : 0, s.d_g16, ld16, P, h->d_missbits, (P + 63) / 64,
              d_sq, d_si, d_no, d_mq, d_sqt, d_sit, d_not};
  hipLaunchKernelGGL(k_bgen_walk, dim3((unsigned)((ld16 + 1023) / 1024), (unsigned)nvar), dim3(256), wa.missbits ? (size_t)(3 * wa.P) * sizeof(unsigned long long) : 0, st, wa);
  BD_HIP(hipGetLastError());
  BD_HIP(hipMemcpyAsync(s.h_sums, s.d_sums, sums_bytes, hipMemcpyDeviceToHost, st));
  BD_HIP(hipMemcpyAsync((uint8_t*)s.h_sums + sums_bytes, d_status, sizeof(int32_t) * nvar, hipMemcpyDeviceToHost, st));
  BD_HIP(hipStreamSynchronize(st));
  const long long* hs = (const long long*)s.h_sums;
  out->g16 = s.d_g16;
  out->ld16 = ld16;
  out->raw = s.d_raw;
  out->raw_stride = stride;
  for (int v = 0; v < nvar; ++v) {
    if (out->sum_q) out->sum_q[v] = hs[v];
    if (out->sum_info) out->sum_info[v] = hs[(size_t)nvar + v];
    if (out->n_obs) out->n_obs[v] = hs[(size_t)2 * nvar + v];
  }
  if (P > 0) {
    const long long* t0 = hs + (size_t)3 * nvar;
    if (out->sum_q_t) std::memcpy(out->sum_q_t, t0, sizeof(int64_t) * (size_t)nvar * P);
    if (out->sum_info_t) std::memcpy(out->sum_info_t, t0 + (size_t)nvar * P, sizeof(int64_t) * (size_t)nvar * P);
    if (out->n_obs_t) std::memcpy(out->n_obs_t, t0 + (size_t)2 * nvar * P, sizeof(int64_t) * (size_t)nvar * P);
  }
  if (out->max_q) std::memcpy(out->max_q, (const uint8_t*)s.h_sums + nsum * 8, sizeof(int32_t) * nvar);
  if (out->status) std::memcpy(out->status, (const uint8_t*)s.h_sums + sums_bytes, sizeof(int32_t) * nvar);
  return RG_BGEN_OK;
}

int rg_bgen_dev_fetch(rg_bgen_dev* h, const void* device_ptr, void* dst, int64_t n) {
  if (!h || !device_ptr || !dst || n < 0) return RG_BGEN_ERR_ARG;
  hipSetDevice(h->device);
  BD_HIP(hipMemcpy(dst, device_ptr, (size_t)n, hipMemcpyDeviceToHost));
  return RG_BGEN_OK;
}

}  // extern "C"
