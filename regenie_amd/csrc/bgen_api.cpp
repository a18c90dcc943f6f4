// C ABI over rgbgen::Reader (include/rg_bgen.h).  Host-only.
#include <algorithm>
#include <new>
#include <string>
#include <atomic>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/rg_bgen.h"
#include "bgen_reader.h"

struct rg_bgen {
  rgbgen::Reader rd;
  std::string err;
  std::mutex err_mu;
  bool ok = false;
  int threads = 1;
};

namespace {
// The read calls only read the handle (every call has its own buffers), so several host threads -- one per GPU in `--step 2 --gpus N` --
// may read through one handle at once; a failing call finds ITS message through the calling thread's copy.
thread_local std::string tl_err;
thread_local const rg_bgen* tl_err_handle = nullptr;
int fail(rg_bgen* h, int code, const std::string& msg) {
  if (h) {
    std::lock_guard<std::mutex> lk(h->err_mu);
    h->err = msg;
    tl_err = msg;
    tl_err_handle = h;
  }
  return code;
}
int classify(const std::string& m) {
  for (const char* k : {"not supported", "only bi-allelic", "only unphased", "only diploid"})
    if (m.find(k) != std::string::npos) return RG_BGEN_ERR_UNSUPPORTED;
  return RG_BGEN_ERR_FORMAT;
}
}  // namespace

extern "C" {

int rg_bgen_open(rg_bgen** out, const char* path) {
  if (!out) return RG_BGEN_ERR_ARG;
  *out = nullptr;
  rg_bgen* h = new (std::nothrow) rg_bgen();
  if (!h) return RG_BGEN_ERR_ARG;
  *out = h;
  if (tl_err_handle == h) tl_err_handle = nullptr;   // a new handle at a closed handle's address must not inherit its message
  if (!path) return fail(h, RG_BGEN_ERR_ARG, "rg_bgen_open: null path");
  try {
    h->rd.open(path);
  } catch (const std::exception& e) {
    h->rd.close();
    return fail(h, classify(e.what()), e.what());
  }
  h->ok = true;
  return RG_BGEN_OK;
}

void rg_bgen_close(rg_bgen* h) {
  if (tl_err_handle == h) tl_err_handle = nullptr;
  delete h;
}
const char* rg_bgen_last_error(const rg_bgen* h) {
  if (!h) return "null bgen handle";
  if (tl_err_handle == h) return tl_err.c_str();
  // a failure raised on another thread: copy it under the lock (that thread may be assigning h->err) into this thread's buffer
  thread_local std::string other;
  {
    std::lock_guard<std::mutex> lk(const_cast<rg_bgen*>(h)->err_mu);
    other = h->err;
  }
  return other.c_str();
}

int rg_bgen_info(const rg_bgen* h, int64_t* n_samples, int64_t* n_variants, int32_t* compression, int32_t* has_sample_ids) {
  if (!h || !h->ok) return RG_BGEN_ERR_ARG;
  if (n_samples) *n_samples = h->rd.n_samples();
  if (n_variants) *n_variants = h->rd.n_variants();
  if (compression) *compression = h->rd.compression();
  if (has_sample_ids) *has_sample_ids = h->rd.has_sample_ids() ? 1 : 0;
  return RG_BGEN_OK;
}

int rg_bgen_sample_id(const rg_bgen* h, int64_t i, const char** id) {
  if (!h || !h->ok || !id || i < 0 || i >= (int64_t)h->rd.sample_ids().size()) return RG_BGEN_ERR_ARG;
  *id = h->rd.sample_ids()[(size_t)i].c_str();
  return RG_BGEN_OK;
}

int rg_bgen_variant(const rg_bgen* h, int64_t j, const char** chrom, uint32_t* position, const char** rsid, const char** allele0,
                    const char** allele1, int64_t* file_offset) {
  if (!h || !h->ok || j < 0 || j >= (int64_t)h->rd.n_variants()) return RG_BGEN_ERR_ARG;
  const rgbgen::Variant& v = h->rd.variants()[(size_t)j];
  if (chrom) *chrom = v.chrom.c_str();
  if (position) *position = v.position;
  if (rsid) *rsid = v.rsid.c_str();
  if (allele0) *allele0 = v.a0.c_str();
  if (allele1) *allele1 = v.a1.c_str();
  if (file_offset) *file_offset = (int64_t)v.offset;
  return RG_BGEN_OK;
}

int rg_bgen_set_threads(rg_bgen* h, int32_t n_threads) {
  if (!h) return RG_BGEN_ERR_ARG;
  if (n_threads < 1) return fail(h, RG_BGEN_ERR_ARG, "rg_bgen_set_threads: thread count must be at least 1");
  h->threads = std::min<int32_t>(n_threads, 256);
  return RG_BGEN_OK;
}

static int read_rows(rg_bgen* h, int64_t n, const int64_t* variant_idx, int32_t ref_first, double* rows, double* info_rows, int64_t row_stride);

int rg_bgen_read_dosages(rg_bgen* h, int64_t n, const int64_t* variant_idx, int32_t ref_first, double* rows, int64_t row_stride) {
  return read_rows(h, n, variant_idx, ref_first, rows, nullptr, row_stride);
}

int rg_bgen_read_dosages_info(rg_bgen* h, int64_t n, const int64_t* variant_idx, int32_t ref_first, double* rows, double* info_rows,
                              int64_t row_stride) {
  if (h && !info_rows) return fail(h, RG_BGEN_ERR_ARG, "rg_bgen_read_dosages_info: bad argument");
  return read_rows(h, n, variant_idx, ref_first, rows, info_rows, row_stride);
}
int rg_bgen_block_bytes(const rg_bgen* h, int64_t* bytes) {
  if (!h || !h->ok || !bytes) return RG_BGEN_ERR_ARG;
  *bytes = (int64_t)h->rd.block_bytes();
  return RG_BGEN_OK;
}

int rg_bgen_read_blocks(rg_bgen* h, int64_t n, const int64_t* variant_idx, uint8_t* blocks, int64_t block_stride, int32_t n_threads) {
  if (!h) return RG_BGEN_ERR_ARG;
  if (!h->ok) return fail(h, RG_BGEN_ERR_ARG, "bgen file is not open");
  if (n < 0 || (n > 0 && (!variant_idx || !blocks)) || block_stride < (int64_t)h->rd.block_bytes())
    return fail(h, RG_BGEN_ERR_ARG, "rg_bgen_read_blocks: bad argument");
  for (int64_t k = 0; k < n; ++k)
    if (variant_idx[k] < 0 || variant_idx[k] >= (int64_t)h->rd.n_variants())
      return fail(h, RG_BGEN_ERR_ARG, "variant index " + std::to_string(variant_idx[k] + 1) + " is out of range");
  const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(n_threads > 0 ? std::min<int32_t>(n_threads, 256) : h->threads, n));
  std::vector<std::string> errs((size_t)nt);
  std::atomic<int64_t> next(0);
  auto work = [&](int t) {
    static thread_local std::vector<uint8_t> cbuf, ubuf;      // a caller's worker thread that reads variant after variant keeps its buffers
    try {
      for (int64_t k; (k = next.fetch_add(1)) < n;)
        h->rd.read_block((uint32_t)variant_idx[k], cbuf, ubuf, blocks + k * block_stride, (size_t)block_stride);
    } catch (const std::exception& e) {
      errs[(size_t)t] = e.what();
      if (errs[(size_t)t].empty()) errs[(size_t)t] = "bgen read failed";
      next = n;
    }
  };
  if (nt == 1) work(0);
  else {
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; ++t) pool.emplace_back(work, t);
    work(0);
    for (auto& th : pool) th.join();
  }
  for (const auto& e : errs)
    if (!e.empty()) return fail(h, classify(e), e);
  return RG_BGEN_OK;
}

// ---- the stored zlib streams of a batch of variants, for the device decoder (bgen_inflate.hip) ----------------------------------
// Stream k lands at dst + off[k] (16-byte aligned offsets), clen[k] bytes long, inflating to ulen[k] bytes.  Variants that follow each other in
// the file are fetched as one range per worker (their identifying data, a few dozen bytes each, comes along and is skipped).
int rg_bgen_compressed_bytes(const rg_bgen* h, int64_t n, const int64_t* variant_idx, int64_t* bytes) {
  if (!h || !h->ok || n < 0 || (n > 0 && !variant_idx) || !bytes) return RG_BGEN_ERR_ARG;
  int64_t tot = 0;
  for (int64_t k = 0; k < n; ++k) {
    if (variant_idx[k] < 0 || variant_idx[k] >= (int64_t)h->rd.n_variants()) return RG_BGEN_ERR_ARG;
    tot += ((int64_t)h->rd.variants()[(size_t)variant_idx[k]].csize + 15) / 16 * 16 + 16;
  }
  *bytes = tot + 64;
  return RG_BGEN_OK;
}

int rg_bgen_read_compressed(rg_bgen* h, int64_t n, const int64_t* variant_idx, uint8_t* dst, int64_t cap, int64_t* off, int32_t* clen,
                            int32_t* ulen, int32_t n_threads) {
  if (!h) return RG_BGEN_ERR_ARG;
  if (!h->ok) return fail(h, RG_BGEN_ERR_ARG, "bgen file is not open");
  if (n < 0 || (n > 0 && (!variant_idx || !dst || !off || !clen || !ulen))) return fail(h, RG_BGEN_ERR_ARG, "rg_bgen_read_compressed: bad argument");
  if (h->rd.compression() != 1) return fail(h, RG_BGEN_ERR_UNSUPPORTED, "rg_bgen_read_compressed: the device decoder takes zlib-compressed files (others are not supported here)");
  int64_t at = 0;
  for (int64_t k = 0; k < n; ++k) {
    if (variant_idx[k] < 0 || variant_idx[k] >= (int64_t)h->rd.n_variants())
      return fail(h, RG_BGEN_ERR_ARG, "variant index " + std::to_string(variant_idx[k] + 1) + " is out of range");
    const rgbgen::Variant& v = h->rd.variants()[(size_t)variant_idx[k]];
    if (v.csize < 4 + 6) return fail(h, RG_BGEN_ERR_FORMAT, "failed to decompress genotype data block for variant: " + v.rsid);
    // the record is placed so that its stream (8 bytes into it: block length, inflated length) starts at a multiple of 16
    off[k] = at + 16;
    clen[k] = (int32_t)(v.csize - 4);
    at = off[k] + ((int64_t)clen[k] + 15) / 16 * 16;
  }
  if (at + 64 > cap) return fail(h, RG_BGEN_ERR_ARG, "rg_bgen_read_compressed: the buffer is smaller than rg_bgen_compressed_bytes");
  const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(n_threads > 0 ? std::min<int32_t>(n_threads, 256) : h->threads, n));
  std::vector<std::string> errs((size_t)nt);
  std::atomic<int64_t> next(0);
  auto work = [&](int t) {
    for (int64_t k; (k = next.fetch_add(1)) < n;) {
      const rgbgen::Variant& v = h->rd.variants()[(size_t)variant_idx[k]];
      // [data, data + 4) block length, [data + 4, data + 8) inflated length, then the stream
      uint8_t* rec = dst + off[k] - 8;
      if (!h->rd.read_raw(v.data, 4ull + v.csize, rec)) { errs[(size_t)t] = "cannot read bgen file"; next = n; return; }
      uint32_t d;
      std::memcpy(&d, rec + 4, 4);
      if (d > 64 + 16 * (uint64_t)h->rd.n_samples()) { errs[(size_t)t] = "genotype data block of variant " + v.rsid + " is larger than any biallelic diploid block of " + std::to_string(h->rd.n_samples()) + " samples"; next = n; return; }
      ulen[k] = (int32_t)d;
    }
  };
  if (nt == 1) work(0);
  else {
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; ++t) pool.emplace_back(work, t);
    work(0);
    for (auto& th : pool) th.join();
  }
  for (const auto& e : errs)
    if (!e.empty()) return fail(h, classify(e), e);
  return RG_BGEN_OK;
}
}  // extern "C"

static int read_rows(rg_bgen* h, int64_t n, const int64_t* variant_idx, int32_t ref_first, double* rows, double* info_rows, int64_t row_stride) {
  if (!h) return RG_BGEN_ERR_ARG;
  if (!h->ok) return fail(h, RG_BGEN_ERR_ARG, "bgen file is not open");
  if (n < 0 || (n > 0 && (!variant_idx || !rows)) || row_stride < (int64_t)h->rd.n_samples())
    return fail(h, RG_BGEN_ERR_ARG, "rg_bgen_read_dosages: bad argument");
  for (int64_t k = 0; k < n; ++k)
    if (variant_idx[k] < 0 || variant_idx[k] >= (int64_t)h->rd.n_variants())
      return fail(h, RG_BGEN_ERR_ARG, "variant index " + std::to_string(variant_idx[k] + 1) + " is out of range");
  const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(h->threads, n / 4));
  std::vector<std::string> errs((size_t)nt);
  auto work = [&](int t) {
    std::vector<uint8_t> cbuf, ubuf;
    const int64_t k0 = n * t / nt, k1 = n * (t + 1) / nt;
    try {
      for (int64_t k = k0; k < k1; ++k)
        h->rd.read_dosages((uint32_t)variant_idx[k], ref_first != 0, rows + k * row_stride, cbuf, ubuf, info_rows ? info_rows + k * row_stride : nullptr);
    } catch (const std::exception& e) {
      errs[(size_t)t] = e.what();
      if (errs[(size_t)t].empty()) errs[(size_t)t] = "bgen read failed";
    }
  };
  if (nt == 1) work(0);
  else {
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; ++t) pool.emplace_back(work, t);
    work(0);
    for (auto& th : pool) th.join();
  }
  for (const auto& e : errs)
    if (!e.empty()) return fail(h, classify(e), e);
  return RG_BGEN_OK;
}
