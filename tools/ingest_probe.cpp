// What bounds the driver's ingest of a .bed that sits in the page cache: pread into page-locked / pageable buffers with 1..16 threads,
// the copy to the device from a page-locked buffer, and the copy straight from a file mapping (registered read-only with the runtime, or
// pageable).  Build: hipcc -O2 tools/ingest_probe.cpp -o tools/bin/ingest_probe -lpthread ; run: ingest_probe /tmp/probe.bin [GB]
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
  const char* fn = argc > 1 ? argv[1] : "/tmp/ingest_probe.bin";
  const size_t gb = argc > 2 ? (size_t)atoi(argv[2]) : 8, bytes = gb << 30;
  {  // the file, written once (stays in the page cache)
    int fd = open(fn, O_CREAT | O_WRONLY | O_TRUNC, 0644);
    std::vector<char> buf(64 << 20, 0x5a);
    double t = now();
    for (size_t o = 0; o < bytes; o += buf.size()) if (write(fd, buf.data(), buf.size()) < 0) return 1;
    close(fd);
    printf("wrote %zu GB in %.1f s\n", gb, now() - t);
  }
  hipSetDevice(0);
  void* dev = nullptr;
  if (hipMalloc(&dev, bytes) != hipSuccess) { printf("device alloc failed\n"); return 1; }
  hipStream_t st; hipStreamCreate(&st);
  int fd = open(fn, O_RDONLY);
  void* pinned = nullptr;
  double t = now();
  hipHostMalloc(&pinned, bytes, hipHostMallocDefault);
  printf("hipHostMalloc %zu GB: %.2f s\n", gb, now() - t);
  char* pageable = (char*)aligned_alloc(4096, bytes);
  memset(pageable, 1, bytes);
  auto pread_rate = [&](char* dst, int nt) {
    double t0 = now();
    std::vector<std::thread> th;
    const size_t piece = 8 << 20;
    for (int k = 0; k < nt; ++k)
      th.emplace_back([&, k]() { for (size_t o = (size_t)k * piece; o < bytes; o += (size_t)nt * piece) if (pread(fd, dst + o, piece, (off_t)o) < 0) break; });
    for (auto& x : th) x.join();
    return gb * 1.073741824 / (now() - t0);
  };
  for (int nt : {1, 4, 8, 16, 32}) printf("pread -> page-locked, %2d threads: %6.1f GB/s   -> pageable: %6.1f GB/s\n", nt, pread_rate((char*)pinned, nt), pread_rate(pageable, nt));
  t = now(); hipMemcpyAsync(dev, pinned, bytes, hipMemcpyHostToDevice, st); hipStreamSynchronize(st);
  printf("H2D from page-locked: %.1f GB/s\n", gb * 1.073741824 / (now() - t));
  void* map = mmap(nullptr, bytes, PROT_READ, MAP_SHARED, fd, 0);
  if (map == MAP_FAILED) { printf("mmap failed\n"); return 1; }
  t = now(); hipMemcpy(dev, map, bytes, hipMemcpyHostToDevice);
  printf("H2D from the file mapping (pageable): %.1f GB/s\n", gb * 1.073741824 / (now() - t));
  for (unsigned flags : {(unsigned)hipHostRegisterReadOnly, (unsigned)hipHostRegisterDefault}) {
    t = now();
    hipError_t e = hipHostRegister(map, bytes, flags);
    const double treg = now() - t;
    if (e != hipSuccess) { printf("hipHostRegister(mapping, flags %u): %s\n", flags, hipGetErrorString(e)); (void)hipGetLastError(); continue; }
    t = now(); e = hipMemcpyAsync(dev, map, bytes, hipMemcpyHostToDevice, st); hipStreamSynchronize(st);
    printf("hipHostRegister(mapping, flags %u): %.2f s; H2D from it: %.1f GB/s (%s)\n", flags, treg, gb * 1.073741824 / (now() - t), hipGetErrorString(e));
    hipHostUnregister(map);
    break;
  }
  unlink(fn);
  return 0;
}
