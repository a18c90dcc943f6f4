// Round 6: what a LARGE device allocation costs on this platform and whether it can be had faster -- one hipMalloc, the same again after a
// free (the driver's scrub of freed memory), chunks of <= 16 GB, several threads at once, and the virtual-memory API (one reserved range
// backed by chunks).  Build: hipcc -O2 tools/alloc_probe2.cpp -o tools/bin/alloc_probe2 -lpthread.  Usage: alloc_probe2 [GB=100]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
  const double GB = argc > 1 ? atof(argv[1]) : 100.0;
  const size_t bytes = (size_t)(GB * 1e9) / (2 << 20) * (2 << 20);
  hipInit(0); hipSetDevice(0);
  hipStream_t st; hipStreamCreate(&st);
  size_t fr = 0, tot = 0; hipMemGetInfo(&fr, &tot);
  printf("free %.1f of %.1f GB\n", fr / 1e9, tot / 1e9);
  auto one = [&](const char* tag) {
    void* p = nullptr; double a = now();
    hipError_t e = hipMalloc(&p, bytes); double b = now();
    if (e != hipSuccess) { printf("%s: failed\n", tag); return; }
    hipMemsetAsync(p, 0, bytes, st); hipStreamSynchronize(st); double c = now();
    hipFree(p); double d = now();
    printf("%s: hipMalloc %.0f GB %.1f ms (%.1f GB/s), memset %.1f ms, hipFree %.1f ms\n", tag, GB, b - a, GB / (b - a) * 1e3, c - b, d - c);
  };
  one("single, first");
  one("single, again right after the free");
  {  // chunks of 8 GB, sequential
    const size_t ch = (size_t)8e9; const int n = (int)(bytes / ch);
    std::vector<void*> ps(n, nullptr); double a = now();
    for (auto& p : ps) hipMalloc(&p, ch);
    double b = now();
    for (auto& p : ps) hipFree(p);
    printf("%d chunks of 8 GB, one thread: %.1f ms (%.1f GB/s); free %.1f ms\n", n, b - a, n * 8.0 / (b - a) * 1e3, now() - b);
  }
  for (int nt : {2, 4, 8}) {  // threads at once
    std::vector<void*> ps(nt, nullptr); std::vector<std::thread> th; double a = now();
    for (int t = 0; t < nt; ++t) th.emplace_back([&, t]() { hipSetDevice(0); hipMalloc(&ps[t], bytes / nt); });
    for (auto& t : th) t.join();
    double b = now();
    for (auto& p : ps) hipFree(p);
    printf("%d threads x %.1f GB: %.1f ms (%.1f GB/s); free %.1f ms\n", nt, GB / nt, b - a, GB / (b - a) * 1e3, now() - b);
  }
  {  // virtual-memory API: reserve the range, back it by 1 GB physical chunks
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    size_t gran = 0; hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended);
    const size_t ch = (size_t)1 << 30; const int n = (int)(bytes / ch);
    void* va = nullptr; double a = now();
    if (hipMemAddressReserve(&va, (size_t)n * ch, 0, nullptr, 0) == hipSuccess) {
      std::vector<hipMemGenericAllocationHandle_t> hs(n);
      bool ok = true;
      for (int k = 0; k < n && ok; ++k) ok = hipMemCreate(&hs[k], ch, &prop, 0) == hipSuccess && hipMemMap((char*)va + (size_t)k * ch, ch, 0, hs[k], 0) == hipSuccess;
      hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
      if (ok) ok = hipMemSetAccess(va, (size_t)n * ch, &acc, 1) == hipSuccess;
      double b = now();
      if (ok) { hipMemsetAsync(va, 0, (size_t)n * ch, st); hipStreamSynchronize(st); }
      double c = now();
      printf("virtual range of %d x 1 GB chunks (granularity %zu): %s, create + map %.1f ms (%.1f GB/s), memset %.1f ms\n", n, gran, ok ? "ok" : "FAILED", b - a, n / (b - a) * 1e3 * 1.074, c - b);
      for (int k = 0; k < n; ++k) { hipMemUnmap((char*)va + (size_t)k * ch, ch); hipMemRelease(hs[k]); }
      hipMemAddressFree(va, (size_t)n * ch);
    } else printf("hipMemAddressReserve failed\n");
  }
  one("single, last");
  return 0;
}
