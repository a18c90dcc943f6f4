// Times the set-up primitives the C++ driver pays before its first kernel: runtime initialisation, device allocations of growing
// size (and their release), page-locked host allocations.  Build: hipcc -O2 tools/alloc_probe.cpp -o tools/bin/alloc_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
  double t0 = now();
  hipInit(0);
  double t1 = now();
  hipSetDevice(0);
  hipStream_t st; hipStreamCreate(&st);
  double t2 = now();
  void* p0 = nullptr; hipMalloc(&p0, 1 << 20);
  double t3 = now();
  size_t fr = 0, tot = 0; hipMemGetInfo(&fr, &tot);
  printf("hipInit %.1f ms, device+stream %.1f ms, first 1 MB malloc %.1f ms, free %.1f of %.1f GB\n", t1 - t0, t2 - t1, t3 - t2, fr / 1e9, tot / 1e9);
  std::vector<double> gbs = {0.25, 1, 4, 16, 48};
  if (argc > 1) { gbs.clear(); for (int i = 1; i < argc; ++i) gbs.push_back(atof(argv[i])); }
  for (double gb : gbs) {
    void* p = nullptr;
    double a = now();
    hipError_t e = hipMalloc(&p, (size_t)(gb * 1e9));
    double b = now();
    if (e != hipSuccess) { printf("malloc %.2f GB failed\n", gb); continue; }
    hipMemsetAsync(p, 0, (size_t)(gb * 1e9), st); hipStreamSynchronize(st);
    double c = now();
    hipFree(p);
    double d = now();
    printf("hipMalloc %6.2f GB: %.1f ms, first-touch memset %.1f ms, hipFree %.1f ms\n", gb, b - a, c - b, d - c);
  }
  {  // many small allocations (the library makes ~60 per context)
    double a = now();
    std::vector<void*> ps(64);
    for (auto& p : ps) hipMalloc(&p, 4 << 20);
    double b = now();
    for (auto& p : ps) hipFree(p);
    printf("64 x 4 MB hipMalloc: %.1f ms, free %.1f ms\n", b - a, now() - b);
  }
  for (double mb : {64.0, 256.0, 1024.0}) {
    void* h = nullptr;
    double a = now();
    hipHostMalloc(&h, (size_t)(mb * 1048576), hipHostMallocDefault);
    double b = now();
    hipHostFree(h);
    printf("hipHostMalloc %5.0f MB: %.1f ms, free %.1f ms\n", mb, b - a, now() - b);
  }
  {  // keep 40 GB until exit: what the NEXT process pays when it starts right after
    void* p = nullptr;
    if (getenv("PROBE_HOLD") && hipMalloc(&p, (size_t)40e9) == hipSuccess) { hipMemsetAsync(p, 1, (size_t)40e9, st); hipStreamSynchronize(st); printf("holding 40 GB until exit\n"); }
  }
  printf("total %.1f ms\n", now() - t0);
  return 0;
}
