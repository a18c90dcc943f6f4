#include "driver.h"
extern "C" const char* rg_last_error(const rg_ctx*) { return "none"; }
using namespace rgdrv;
int main(int argc, char** argv) {
  Run r;
  try {
    r.p = parse_args(argc, argv);
    sout.f.open("/dev/null");
    read_bim_fam(r);
    read_pheno_cov(r);
  } catch (const std::exception& e) { return 0; }
  return 0;
}
