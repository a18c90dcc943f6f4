// TEST SCAFFOLDING, not product code (see hip/hip_runtime.h beside this file).  A host stand-in for the nine RCCL entry points rg_group.hip
// resolves with dlsym, built with the soname librccl.so.1 and loaded ahead of the library under test: communicators of ONE process whose ranks
// are host threads, "device" buffers that are host memory.  Point-to-point operations of a group are matched by (source, destination) in the
// order they were issued, as RCCL matches them, and a receive whose count differs from its send ABORTS (the thing a wrong offset / count
// computation in the caller would produce); broadcasts and all-reduces are matched by their order on the communicator.  It checks the
// caller's arithmetic -- who sends what to whom, in which order, into which address -- not RCCL.
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <vector>

namespace {
struct World {
  int n = 0;
  std::mutex m;
  std::condition_variable cv;
  std::map<std::pair<int, int>, std::deque<std::vector<double>>> box;      // (src, dst) -> messages in issue order
  // collectives: one slot per sequence number
  struct Coll { std::vector<std::vector<double>> part; int arrived = 0, left = 0; };
  std::map<uint64_t, Coll> coll;
  int64_t stat_sends = 0, stat_bytes = 0, stat_bcasts = 0, stat_allreduces = 0;
};
struct Comm { World* w; int rank; uint64_t seq = 0; };
struct Op { int kind; const void* src; void* dst; size_t count; int peer; Comm* c; };
thread_local std::vector<Op> t_ops;
thread_local int t_depth = 0;
World* g_last_world = nullptr;

int run(const Op& o) {
  World& w = *o.c->w;
  if (o.kind == 0) {          // send: buffered, never blocks
    std::lock_guard<std::mutex> lk(w.m);
    w.box[{o.c->rank, o.peer}].emplace_back((const double*)o.src, (const double*)o.src + o.count);
    ++w.stat_sends; w.stat_bytes += (int64_t)o.count * 8;
    w.cv.notify_all();
    return 0;
  }
  if (o.kind == 1) {          // receive
    std::unique_lock<std::mutex> lk(w.m);
    auto& q = w.box[{o.peer, o.c->rank}];
    w.cv.wait(lk, [&] { return !q.empty(); });
    if (q.front().size() != o.count) {
      fprintf(stderr, "fake RCCL: rank %d receives %zu doubles from rank %d, which sent %zu\n", o.c->rank, o.count, o.peer, q.front().size());
      abort();
    }
    memcpy(o.dst, q.front().data(), o.count * 8);
    q.pop_front();
    return 0;
  }
  // broadcast (kind 2, root = peer) / all-reduce (kind 3): every rank of the communicator calls them in the same order
  const uint64_t id = o.c->seq++;
  std::unique_lock<std::mutex> lk(w.m);
  World::Coll& c = w.coll[id];
  if (c.part.empty()) c.part.resize(w.n);
  if (o.kind == 3 || o.c->rank == o.peer) c.part[o.c->rank].assign((const double*)o.src, (const double*)o.src + o.count);
  if (o.c->rank == 0) (o.kind == 2 ? w.stat_bcasts : w.stat_allreduces)++;
  ++c.arrived;
  w.cv.notify_all();
  w.cv.wait(lk, [&] { return c.arrived == w.n; });
  if (o.kind == 2) {
    if (c.part[o.peer].size() != o.count) { fprintf(stderr, "fake RCCL: broadcast counts differ between the ranks\n"); abort(); }
    if (o.dst != o.src || o.c->rank != o.peer) memcpy(o.dst, c.part[o.peer].data(), o.count * 8);
  } else {
    std::vector<double> sum(o.count, 0.0);
    for (int r = 0; r < w.n; ++r) {
      if (c.part[r].size() != o.count) { fprintf(stderr, "fake RCCL: all-reduce counts differ between the ranks\n"); abort(); }
      for (size_t i = 0; i < o.count; ++i) sum[i] += c.part[r][i];
    }
    memcpy(o.dst, sum.data(), o.count * 8);
  }
  if (++c.left == w.n) w.coll.erase(id);
  return 0;
}
int issue(const Op& o) {
  if (t_depth > 0) { t_ops.push_back(o); return 0; }
  return run(o);
}
}  // namespace

extern "C" {
int ncclCommInitAll(void** comms, int n, const int*) {
  World* w = new World();
  w->n = n;
  g_last_world = w;
  for (int r = 0; r < n; ++r) comms[r] = new Comm{w, r};
  return 0;
}
int ncclCommDestroy(void* c) { delete (Comm*)c; return 0; }      // (the world object is leaked: test processes are short)
int ncclGroupStart() { ++t_depth; return 0; }
int ncclGroupEnd() {
  if (--t_depth > 0) return 0;
  std::vector<Op> ops;
  ops.swap(t_ops);
  for (const Op& o : ops) if (o.kind == 0) run(o);                 // every send of the group is in flight before any receive waits
  for (const Op& o : ops) if (o.kind != 0) run(o);
  return 0;
}
int ncclSend(const void* buf, size_t count, int dtype, int peer, void* comm, void*) {
  if (dtype != 8) return 4;
  return issue(Op{0, buf, nullptr, count, peer, (Comm*)comm});
}
int ncclRecv(void* buf, size_t count, int dtype, int peer, void* comm, void*) {
  if (dtype != 8) return 4;
  return issue(Op{1, nullptr, buf, count, peer, (Comm*)comm});
}
int ncclBroadcast(const void* src, void* dst, size_t count, int dtype, int root, void* comm, void*) {
  if (dtype != 8) return 4;
  return issue(Op{2, src, dst, count, root, (Comm*)comm});
}
int ncclAllReduce(const void* src, void* dst, size_t count, int dtype, int op, void* comm, void*) {
  if (dtype != 8 || op != 0) return 4;
  return issue(Op{3, src, dst, count, 0, (Comm*)comm});
}
const char* ncclGetErrorString(int) { return "fake RCCL error"; }
// what the last world moved (for the test's bookkeeping)
void fake_rccl_stats(int64_t* out) {
  World* w = g_last_world;
  out[0] = w ? w->stat_sends : 0; out[1] = w ? w->stat_bytes : 0; out[2] = w ? w->stat_bcasts : 0; out[3] = w ? w->stat_allreduces : 0;
}
}
