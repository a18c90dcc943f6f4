// C ABI over rgpgen::Reader (include/rg_pgen.h).  Host-only.
#include <algorithm>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rg_pgen.h"
#include "pgen_reader.h"

struct rg_pgen {
  rgpgen::Reader rd;
  std::string err;
  bool ok = false;
  int threads = 1;
  std::vector<rgpgen::Scratch> scratch;  // one decode state per worker thread
};

namespace {
int fail(rg_pgen* h, int code, const std::string& msg) {
  if (h) h->err = msg;
  return code;
}
}  // namespace

extern "C" {

int rg_pgen_open(rg_pgen** out, const char* path) {
  if (!out) return RG_PGEN_ERR_ARG;
  *out = nullptr;
  rg_pgen* h = new (std::nothrow) rg_pgen();
  if (!h) return RG_PGEN_ERR_ARG;
  *out = h;
  if (!path) return fail(h, RG_PGEN_ERR_ARG, "rg_pgen_open: null path");
  try {
    h->rd.open(path);
  } catch (const std::exception& e) {
    h->rd.close();
    const bool unsupported = h->rd.dosage_present() || h->rd.max_alleles() != 2;
    return fail(h, unsupported ? RG_PGEN_ERR_UNSUPPORTED : RG_PGEN_ERR_FORMAT, e.what());
  }
  if (h->rd.max_alleles() != 2) {  // prep_pgen, Geno.cpp:1098-1099
    h->rd.close();
    return fail(h, RG_PGEN_ERR_UNSUPPORTED, "only bi-allelic variants are accepted.");
  }
  h->ok = true;  // a file with dosage tracks opens: rg_pgen_info reports it and rg_pgen_read_bed_rows refuses it
  return RG_PGEN_OK;
}

void rg_pgen_close(rg_pgen* h) { delete h; }

const char* rg_pgen_last_error(const rg_pgen* h) { return h ? h->err.c_str() : "null pgen handle"; }

int rg_pgen_info(const rg_pgen* h, int64_t* n_samples, int64_t* n_variants, int32_t* max_alleles, int32_t* phase_present,
                 int32_t* dosage_present) {
  if (!h || !h->ok) return RG_PGEN_ERR_ARG;
  if (dosage_present) *dosage_present = h->rd.dosage_present() ? 1 : 0;
  if (n_samples) *n_samples = h->rd.n_samples();
  if (n_variants) *n_variants = h->rd.n_variants();
  if (max_alleles) *max_alleles = h->rd.max_alleles();
  if (phase_present) *phase_present = h->rd.phase_present() ? 1 : 0;
  return RG_PGEN_OK;
}

int rg_pgen_set_threads(rg_pgen* h, int32_t n_threads) {
  if (!h) return RG_PGEN_ERR_ARG;
  if (n_threads < 1) return fail(h, RG_PGEN_ERR_ARG, "rg_pgen_set_threads: thread count must be at least 1");
  h->threads = std::min<int32_t>(n_threads, 256);
  return RG_PGEN_OK;
}

int rg_pgen_read_bed_rows(rg_pgen* h, int64_t n, const int64_t* variant_idx, uint8_t* rows, int64_t row_stride) {
  if (!h) return RG_PGEN_ERR_ARG;
  if (!h->ok) return fail(h, RG_PGEN_ERR_ARG, "pgen file is not open");
  if (n < 0 || (n > 0 && (!variant_idx || !rows)) || row_stride < h->rd.bytes_per_row())
    return fail(h, RG_PGEN_ERR_ARG, "rg_pgen_read_bed_rows: bad argument");
  if (h->rd.dosage_present())  // regenie reads such a file as dosages (Geno.cpp:1101, :1795-1796): 2-bit rows would be other numbers
    return fail(h, RG_PGEN_ERR_UNSUPPORTED, "pgen file has dosages; the GPU path reads hardcall (2-bit) genotypes only");
  for (int64_t k = 0; k < n; ++k)
    if (variant_idx[k] < 0 || variant_idx[k] >= (int64_t)h->rd.n_variants())
      return fail(h, RG_PGEN_ERR_ARG, "variant index " + std::to_string(variant_idx[k] + 1) + " is out of range (1.." +
                                          std::to_string(h->rd.n_variants()) + ")");
  // Variants are independent except that an LD-compressed one needs its base, which each worker caches for itself:
  // contiguous chunks keep those caches warm (as OpenMP's per-thread PgenReader state does, Geno.cpp:1777-1798).
  const int64_t min_chunk = 8;
  const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(h->threads, n / min_chunk));
  while ((int)h->scratch.size() < nt) h->scratch.push_back(h->rd.make_scratch());
  std::vector<std::string> errs((size_t)nt);
  auto work = [&](int t) {
    const int64_t k0 = n * t / nt, k1 = n * (t + 1) / nt;
    try {
      for (int64_t k = k0; k < k1; ++k) h->rd.read_bed_row((uint32_t)variant_idx[k], rows + k * row_stride, h->scratch[(size_t)t]);
    } catch (const std::exception& e) {
      errs[(size_t)t] = e.what();
      if (errs[(size_t)t].empty()) errs[(size_t)t] = "pgen read failed";
    }
  };
  if (nt == 1) {
    work(0);
  } else {
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; ++t) pool.emplace_back(work, t);
    work(0);
    for (auto& th : pool) th.join();
  }
  for (const auto& e : errs)
    if (!e.empty()) return fail(h, RG_PGEN_ERR_FORMAT, e);
  return RG_PGEN_OK;
}

int rg_pgen_read_dosage_rows(rg_pgen* h, int64_t n, const int64_t* variant_idx, double* rows, int64_t row_stride) {
  if (!h) return RG_PGEN_ERR_ARG;
  if (!h->ok) return fail(h, RG_PGEN_ERR_ARG, "pgen file is not open");
  if (n < 0 || (n > 0 && (!variant_idx || !rows)) || row_stride < (int64_t)h->rd.n_samples())
    return fail(h, RG_PGEN_ERR_ARG, "rg_pgen_read_dosage_rows: bad argument");
  for (int64_t k = 0; k < n; ++k)
    if (variant_idx[k] < 0 || variant_idx[k] >= (int64_t)h->rd.n_variants())
      return fail(h, RG_PGEN_ERR_ARG, "variant index " + std::to_string(variant_idx[k] + 1) + " is out of range (1.." +
                                          std::to_string(h->rd.n_variants()) + ")");
  const int64_t min_chunk = 8;
  const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(h->threads, n / min_chunk));
  while ((int)h->scratch.size() < nt) h->scratch.push_back(h->rd.make_scratch());
  std::vector<std::string> errs((size_t)nt);
  auto work = [&](int t) {
    const int64_t k0 = n * t / nt, k1 = n * (t + 1) / nt;
    try {
      for (int64_t k = k0; k < k1; ++k) h->rd.read_dosages((uint32_t)variant_idx[k], rows + k * row_stride, h->scratch[(size_t)t]);
    } catch (const std::exception& e) {
      errs[(size_t)t] = e.what();
      if (errs[(size_t)t].empty()) errs[(size_t)t] = "pgen read failed";
    }
  };
  if (nt == 1) {
    work(0);
  } else {
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; ++t) pool.emplace_back(work, t);
    work(0);
    for (auto& th : pool) th.join();
  }
  for (const auto& e : errs)
    if (!e.empty()) return fail(h, RG_PGEN_ERR_FORMAT, e);
  return RG_PGEN_OK;
}

int rg_pgen_read_dosages(rg_pgen* h, int64_t variant_idx, double* out) {
  if (!h) return RG_PGEN_ERR_ARG;
  if (!h->ok) return fail(h, RG_PGEN_ERR_ARG, "pgen file is not open");
  if (!out || variant_idx < 0 || variant_idx >= (int64_t)h->rd.n_variants())
    return fail(h, RG_PGEN_ERR_ARG, "rg_pgen_read_dosages: bad argument");
  try {
    h->rd.read_dosages((uint32_t)variant_idx, out);
  } catch (const std::exception& e) {
    return fail(h, RG_PGEN_ERR_FORMAT, e.what());
  }
  return RG_PGEN_OK;
}

int rg_pgen_read_hardcalls(rg_pgen* h, int64_t variant_idx, double* out) {
  if (!h) return RG_PGEN_ERR_ARG;
  if (!h->ok) return fail(h, RG_PGEN_ERR_ARG, "pgen file is not open");
  if (!out || variant_idx < 0 || variant_idx >= (int64_t)h->rd.n_variants())
    return fail(h, RG_PGEN_ERR_ARG, "rg_pgen_read_hardcalls: bad argument");
  try {
    h->rd.read_hardcalls((uint32_t)variant_idx, out);
  } catch (const std::exception& e) {
    return fail(h, RG_PGEN_ERR_FORMAT, e.what());
  }
  return RG_PGEN_OK;
}
}
