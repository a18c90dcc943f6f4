// TEST SCAFFOLDING, not product code: a stand-in for <hip/hip_runtime.h> that lets g++ compile a .hip source of this repository for the HOST and
// run its kernels there -- one workgroup at a time, its work-items as cooperatively scheduled fibers on the calling thread (ucontext), so that
// __syncthreads(), __shfl_down() across a 64-lane wavefront and `__shared__` arrays mean what they mean on the device.  tests/ compile single
// library sources against it (with host stand-ins for the kernels of the OTHER sources they call) to execute the device code paths of a change
// in this container, which has no GPU.  It covers what those sources use -- 1-D workgroups, 64-wide shuffles, static shared memory, the memory
// / stream / event calls as synchronous host operations -- and nothing more; nothing under regenie_amd/ includes or links it.
#pragma once
#include <ucontext.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

typedef int hipError_t;
constexpr hipError_t hipSuccess = 0;
inline const char* hipGetErrorString(hipError_t) { return "host stand-in of the HIP runtime"; }
typedef struct hipcpu_stream_t* hipStream_t;
typedef struct hipcpu_event_t* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
inline hipError_t hipMalloc(void** p, size_t b) { *p = calloc(b ? b : 1, 1); return *p ? 0 : 2; }
template <class T> inline hipError_t hipMalloc(T** p, size_t b) { return hipMalloc((void**)p, b); }
inline hipError_t hipFree(void* p) { free(p); return 0; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t b, hipMemcpyKind, hipStream_t = nullptr) { memmove(d, s, b); return 0; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t b, hipMemcpyKind) { memmove(d, s, b); return 0; }
inline hipError_t hipMemsetAsync(void* d, int v, size_t b, hipStream_t = nullptr) { memset(d, v, b); return 0; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return 0; }
inline hipError_t hipDeviceSynchronize() { return 0; }
inline hipError_t hipSetDevice(int) { return 0; }
inline hipError_t hipGetLastError() { return 0; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t = nullptr) { return 0; }
constexpr unsigned hipEventDisableTiming = 2;
constexpr hipError_t hipErrorPeerAccessAlreadyEnabled = 704;
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = reinterpret_cast<hipEvent_t>(malloc(1)); return 0; }
inline hipError_t hipEventCreate(hipEvent_t* e) { return hipEventCreateWithFlags(e, 0); }
inline hipError_t hipEventDestroy(hipEvent_t e) { free(e); return 0; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return 0; }       // (every operation here has completed when its call returns)
inline hipError_t hipDeviceEnablePeerAccess(int, unsigned) { return 0; }
inline hipError_t hipMemcpy2DAsync(void* d, size_t dpitch, const void* s, size_t spitch, size_t width, size_t height, hipMemcpyKind, hipStream_t = nullptr) {
  for (size_t r = 0; r < height; ++r) memmove((char*)d + r * dpitch, (const char*)s + r * spitch, width);
  return 0;
}
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return 0; }

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {}
};

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static thread_local

namespace hipcpu {
struct Idx { unsigned x = 0, y = 0, z = 0; };
struct Bar { int count = 0, size = 0; unsigned gen = 0; };
struct Fiber { ucontext_t ctx; bool done = false; };
struct Block {
  int n = 0, cur = 0, live = 0;
  std::vector<Fiber> f;
  std::vector<char> stacks;
  ucontext_t main;
  Bar all, wave[16];
  uint64_t slot[1024];
  std::function<void()> body;
};
inline thread_local Block g_blk;      // (per host thread: the rank threads of a multi-device test may launch at the same time)
inline thread_local Idx g_block_idx, g_block_dim, g_grid_dim;
constexpr size_t kStack = 256 * 1024;

inline void yield() {
  Block& b = g_blk;
  const int me = b.cur;
  int nx = me;
  do nx = (nx + 1) % b.n; while (b.f[nx].done && nx != me);
  if (nx == me) return;
  b.cur = nx;
  swapcontext(&b.f[me].ctx, &b.f[nx].ctx);
}
inline void bar_wait(Bar& br) {
  const unsigned g = br.gen;
  if (++br.count >= br.size) { br.count = 0; ++br.gen; return; }
  while (br.gen == g) yield();
}
inline void bar_leave(Bar& br) {   // a work-item that has returned no longer takes part (as a finished wavefront at s_barrier)
  --br.size;
  if (br.size > 0 && br.count >= br.size) { br.count = 0; ++br.gen; }
}
inline void trampoline() {
  Block& b = g_blk;
  b.body();
  const int me = b.cur;
  b.f[me].done = true;
  --b.live;
  bar_leave(b.all);
  bar_leave(b.wave[me >> 6]);
  if (b.live == 0) { setcontext(&b.main); }
  int nx = me;
  do nx = (nx + 1) % b.n; while (b.f[nx].done);
  b.cur = nx;
  setcontext(&b.f[nx].ctx);
}
inline void run_block(int n, const std::function<void()>& body) {
  Block& b = g_blk;
  if (n < 1 || n > 1024) { fprintf(stderr, "hipcpu: workgroup of %d work-items\n", n); abort(); }
  b.n = n; b.live = n; b.cur = 0; b.body = body;
  if ((int)b.f.size() < n) b.f.resize(n);
  if (b.stacks.size() < (size_t)n * kStack) b.stacks.resize((size_t)n * kStack);
  b.all = Bar(); b.all.size = n;
  for (int w = 0; w < 16; ++w) { b.wave[w] = Bar(); b.wave[w].size = std::max(0, std::min(64, n - 64 * w)); }
  for (int t = 0; t < n; ++t) {
    b.f[t].done = false;
    getcontext(&b.f[t].ctx);
    b.f[t].ctx.uc_stack.ss_sp = b.stacks.data() + (size_t)t * kStack;
    b.f[t].ctx.uc_stack.ss_size = kStack;
    b.f[t].ctx.uc_link = nullptr;
    makecontext(&b.f[t].ctx, trampoline, 0);
  }
  swapcontext(&b.main, &b.f[0].ctx);
}
template <class K, class... A>
inline void launch(K kernel, dim3 grid, dim3 block, A... args) {
  if (block.y != 1 || block.z != 1) { fprintf(stderr, "hipcpu: 1-D workgroups only\n"); abort(); }
  g_block_dim = Idx{block.x, 1, 1};
  g_grid_dim = Idx{grid.x, grid.y, grid.z};
  for (unsigned z = 0; z < grid.z; ++z)
    for (unsigned y = 0; y < grid.y; ++y)
      for (unsigned x = 0; x < grid.x; ++x) {
        g_block_idx = Idx{x, y, z};
        run_block((int)block.x, [&]() { kernel(args...); });
      }
}
struct ThreadIdx {
  struct X { operator unsigned() const { return (unsigned)g_blk.cur; } } x;
  struct One { operator unsigned() const { return 0u; } } y, z;
};
}  // namespace hipcpu

static const hipcpu::ThreadIdx threadIdx;
#define blockIdx hipcpu::g_block_idx
#define blockDim hipcpu::g_block_dim
#define gridDim hipcpu::g_grid_dim
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) hipcpu::launch(kernel, dim3(grid), dim3(block), __VA_ARGS__)

inline void __syncthreads() { hipcpu::bar_wait(hipcpu::g_blk.all); }
template <class T>
inline T __shfl_down(T v, int delta) {      // lane + delta beyond the wavefront: the caller's own value
  static_assert(sizeof(T) <= 8, "shuffles of up to 64 bits");
  hipcpu::Block& b = hipcpu::g_blk;
  const int t = b.cur, lane = t & 63, w = t >> 6;
  memcpy(&b.slot[t], &v, sizeof(T));
  hipcpu::bar_wait(b.wave[w]);
  T r = v;
  if (lane + delta < 64 && t + delta < b.n) memcpy(&r, &b.slot[t + delta], sizeof(T));
  hipcpu::bar_wait(b.wave[w]);
  return r;
}
inline double __longlong_as_double(long long x) { double d; memcpy(&d, &x, 8); return d; }
inline long long __double_as_longlong(double d) { long long x; memcpy(&x, &d, 8); return x; }
inline int __popc(unsigned x) { return __builtin_popcount(x); }
inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
template <class T> inline T atomicAdd(T* p, T v) { T o = *p; *p += v; return o; }
