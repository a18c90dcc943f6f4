/* TEST INFRASTRUCTURE (oracle/ref_shim): the reference links two Fortran-77 libraries (quadpack's dqags,
 * mvtnorm's mvtdst) for its gene-based tests (src/SKAT.cpp:1898, external_libs/mvtnorm/mvtnorm.cpp).
 * There is no Fortran compiler in this image and neither Step 1 nor the single-variant Step 2 reaches
 * them: these entry points only satisfy the linker and abort if called. */
#include <stdio.h>
#include <stdlib.h>
void dqags_(void) { fprintf(stderr, "oracle/ref_shim: dqags_ (quadpack) is not available in this build\n"); abort(); }
void mvtdst_(void) { fprintf(stderr, "oracle/ref_shim: mvtdst_ (mvtnorm) is not available in this build\n"); abort(); }
