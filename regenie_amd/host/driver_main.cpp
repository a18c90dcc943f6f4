// regenie-amd, the C++ host driver (see driver.h): main().
#include "driver.h"

int main(int argc, char** argv) {
  int rc = EXIT_FAILURE;
  try {
    rc = rgdrv::run(argc, argv);
  } catch (const std::exception& e) {  // Regenie.cpp:72-91
    rgdrv::sout << "\nERROR: " << e.what() << "\nFor more information, use option '--help' or visit the website: https://rgcgithub.github.io/regenie/\n";
    return EXIT_FAILURE;
  }
  if (rc == 0 && rgdrv::fast_exit) {   // the run's files are closed (they are locals of run()); the log and stdout are flushed here
    std::cout.flush();
    if (rgdrv::sout.f.is_open()) rgdrv::sout.f.close();
    fflush(nullptr);
    _exit(0);
  }
  return rc;
}
