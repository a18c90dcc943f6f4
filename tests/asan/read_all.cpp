// reads every variant of a (possibly damaged) .bgen / .pgen through the C ABI; exit 0 = read or refused cleanly
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#include "rg_bgen.h"
#include "rg_pgen.h"
int main(int argc, char** argv) {
  if (argc < 3) return 2;
  if (!strcmp(argv[1], "bgen")) {
    rg_bgen* h = nullptr;
    if (rg_bgen_open(&h, argv[2]) != 0) { if (h) rg_bgen_close(h); return 0; }
    int64_t n = 0, m = 0; int32_t comp = 0, ids = 0;
    rg_bgen_info(h, &n, &m, &comp, &ids);
    if (n < 0 || m < 0 || n > 1000000 || m > 1000000) { rg_bgen_close(h); return 0; }
    for (int64_t i = 0; i < n && ids; ++i) { const char* s; rg_bgen_sample_id(h, i, &s); }
    std::vector<double> row((size_t)n + 1), info((size_t)n + 1);
    int64_t bb = 0; rg_bgen_block_bytes(h, &bb);
    std::vector<uint8_t> blk((size_t)bb + 16);
    for (int64_t j = 0; j < m; ++j) {
      const char *c, *r, *a0, *a1; uint32_t pos; int64_t off;
      rg_bgen_variant(h, j, &c, &pos, &r, &a0, &a1, &off);
      rg_bgen_read_dosages(h, 1, &j, 0, row.data(), n);
      rg_bgen_read_dosages_info(h, 1, &j, 1, row.data(), info.data(), n);
      rg_bgen_read_blocks(h, 1, &j, blk.data(), bb, 1);
      int64_t cb = 0;
      if (rg_bgen_compressed_bytes(h, 1, &j, &cb) == 0 && cb >= 0 && cb < (1 << 28)) {
        std::vector<uint8_t> dst((size_t)cb + 16); int64_t o; int32_t cl, ul;
        rg_bgen_read_compressed(h, 1, &j, dst.data(), cb, &o, &cl, &ul, 1);
      }
    }
    rg_bgen_close(h);
  } else {
    rg_pgen* h = nullptr;
    if (rg_pgen_open(&h, argv[2]) != 0) { if (h) rg_pgen_close(h); return 0; }
    int64_t n = 0, m = 0; int32_t ma = 0, ph = 0, hd = 0;
    rg_pgen_info(h, &n, &m, &ma, &ph, &hd);
    if (n < 0 || m < 0 || n > 1000000 || m > 1000000) { rg_pgen_close(h); return 0; }
    std::vector<double> row((size_t)n + 1);
    std::vector<uint8_t> bed((size_t)(n + 3) / 4 + 8);
    for (int64_t j = 0; j < m; ++j) {
      rg_pgen_read_bed_rows(h, 1, &j, bed.data(), (n + 3) / 4);
      rg_pgen_read_dosages(h, j, row.data());
      rg_pgen_read_dosage_rows(h, 1, &j, row.data(), n);
      rg_pgen_read_hardcalls(h, j, row.data());
    }
    rg_pgen_close(h);
  }
  return 0;
}
