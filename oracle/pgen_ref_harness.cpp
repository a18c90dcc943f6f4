// TEST INFRASTRUCTURE (oracle/): a C-ABI shim over the REFERENCE's own pgen reader, so the pgen
// restatement (oracle/pgen.py) and the product reader (regenie_amd/csrc/pgen_reader.h) can be
// pinned against what regenie itself would read.  Nothing here is copied from the reference: the
// recipe oracle/Makefile compiles the reference's sources where they lie
// (/root/reference/external_libs/pgenlib) and links this shim against them into
// oracle/_ref/libpgen_ref.so.  Only tests/ may load it.
//
// The calls mirror regenie's use of the class: prep_pgen (Geno.cpp:1071-1103) and the Step-1
// block reader (Geno.cpp:1793-1798: ReadHardcalls(g, n, thread, offset, 1)).
#include <cstdint>
#include <string>
#include <vector>

#include "pgenlibr.h"

extern "C" {

// counts[0..3] = raw sample count, variant count, max allele count, dosage present
int pgen_ref_counts(const char* path, uint32_t n_samples, int64_t* counts) {
  PgenReader pgr;
  pgr.Load(path, n_samples, std::vector<int>(), 1);
  counts[0] = pgr.GetRawSampleCt();
  counts[1] = pgr.GetVariantCt();
  counts[2] = pgr.GetMaxAlleleCt();
  counts[3] = pgr.DosagePresent() ? 1 : 0;
  pgr.Close();
  return 0;
}

// out is n_idx x n_keep doubles (0/1/2 ALT counts, -3 = missing); keep_1based may be empty (all samples)
int pgen_ref_hardcalls(const char* path, uint32_t n_samples, const int32_t* keep_1based, int64_t n_keep_ids,
                       const int64_t* idx, int64_t n_idx, double* out) {
  PgenReader pgr;
  std::vector<int> keep(keep_1based, keep_1based + n_keep_ids);
  pgr.Load(path, n_samples, keep, 1);
  const size_t n = n_keep_ids ? (size_t)n_keep_ids : (size_t)n_samples;
  for (int64_t j = 0; j < n_idx; ++j) pgr.ReadHardcalls(out + (size_t)j * n, n, 0, (int)idx[j], 1);
  pgr.Close();
  return 0;
}
}

// PgenReader::Read(buf, n, thr, idx, 1) -- what regenie calls in dosage_mode (Geno.cpp:1795-1796): ALT dosages where the
// file stores them, hardcalls elsewhere, -3 for missing.
extern "C" int pgen_ref_dosages(const char* path, uint32_t n_samples, const int64_t* idx, int64_t n_idx, double* out) {
  PgenReader pgr;
  pgr.Load(path, n_samples, std::vector<int>(), 1);
  for (int64_t j = 0; j < n_idx; ++j) pgr.Read(out + (size_t)j * n_samples, n_samples, 0, (int)idx[j], 1);
  pgr.Close();
  return 0;
}
