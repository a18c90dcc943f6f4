// regenie-amd, the C++ host driver (see driver.h): main().
#include "driver.h"

int main(int argc, char** argv) {
  try {
    return rgdrv::run(argc, argv);
  } catch (const std::exception& e) {  // Regenie.cpp:72-91
    rgdrv::sout << "\nERROR: " << e.what() << "\nFor more information, use option '--help' or visit the website: https://rgcgithub.github.io/regenie/\n";
    return EXIT_FAILURE;
  }
}
