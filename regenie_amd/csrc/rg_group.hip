// Multi-GPU hand-off of the level-0 predictors (include/rg_step1.h: rg_group_*, rg_l0_finish).
//
// The reference splits level 0 over processes through files: write_l0_master deals contiguous block ranges to jobs
// (src/Data.cpp:270-302), every job writes PFX_job<k>_l0_Y<ph> and the --run-l1 job reads them all back (prep_parallel_l1,
// Data.cpp:862-908; read_l0, Step1_Models.cpp:1921-1987).  Here the jobs are the GPUs of one node -- one context and one host
// thread per GPU -- and the files are replaced by ONE exchange over xGMI:
//   P >= n : all-to-all by phenotype.  Rank r receives, from every rank g, the predictor rows of g's blocks for r's phenotypes
//            only (1/n of the all-gather volume) into Wr [L][np_r][Np], and runs level 1 on that phenotype range.
//   P <  n : all-gather of the block slabs of W (every rank ends with the full W); level 1 then runs on every rank with
//            its fold-Gram tiles and ridge systems shared and completed by sum all-reduces (rg_set_collective).
// Transports:
//   RG_TRANSPORT_RCCL : RCCL (librccl.so.1, resolved at run time so that single-GPU users never load it): one communicator per
//                       context from ncclCommInitAll; the all-to-all is one ncclGroup of send/recv pairs per rank (xGMI is
//                       point to point: every pair gets its own link instead of a ring), packed by strided device copies;
//                       the all-gather is a group of broadcasts of the uneven slabs; the all-reduce is ncclAllReduce.
//   RG_TRANSPORT_PEER : direct device-to-device copies (hipMemcpy2DAsync pulls from the peer's buffer after a host barrier).
//                       Works for contexts that share ONE device, which is how the sharding / exchange / view logic is
//                       tested on a single-GPU box with world size 2; its all-reduce is staged through host memory.
// No compute happens here; nothing in this file is a fallback for a kernel.
#include <dlfcn.h>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include "rg_internal.h"

namespace {

// ---- the few RCCL entry points used, resolved with dlsym ---------------------------------------------------------------------
typedef void* ncclComm_t;
typedef int ncclResult_t;
enum { kNcclDouble = 8, kNcclSum = 0 };   // ncclFloat64 = 8, ncclSum = 0 (rccl.h)
struct Rccl {
  void* h = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool load(std::string& err) {
    if (h) return true;
    const char* named = getenv("RG_RCCL_LIB");        // a particular build of RCCL (path or soname); the tests' stand-in, tests/hipcpu/fake_rccl.cpp
    for (const char* name : {named, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      if (!name || !*name) continue;
      h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (h) break;
    }
    if (!h) { err = std::string("cannot load RCCL: ") + dlerror(); return false; }
    auto sym = [&](const char* n) { void* p = dlsym(h, n); if (!p) err = std::string("RCCL symbol missing: ") + n; return p; };
    CommInitAll = (decltype(CommInitAll))sym("ncclCommInitAll");
    CommDestroy = (decltype(CommDestroy))sym("ncclCommDestroy");
    GroupStart = (decltype(GroupStart))sym("ncclGroupStart");
    GroupEnd = (decltype(GroupEnd))sym("ncclGroupEnd");
    Send = (decltype(Send))sym("ncclSend");
    Recv = (decltype(Recv))sym("ncclRecv");
    Broadcast = (decltype(Broadcast))sym("ncclBroadcast");
    AllReduce = (decltype(AllReduce))sym("ncclAllReduce");
    GetErrorString = (decltype(GetErrorString))sym("ncclGetErrorString");
    return CommInitAll && CommDestroy && GroupStart && GroupEnd && Send && Recv && Broadcast && AllReduce && GetErrorString;
  }
};
Rccl g_rccl;

// Reusable barrier of the rank threads that can be BROKEN: once a rank has failed for good (rg_group_abort, or an error
// inside a group call) every wait -- pending or future -- returns false at once, so that no rank is ever left waiting for
// a peer that will not come.  A broken group stays broken: every later group call fails.
struct Barrier {
  std::mutex m; std::condition_variable cv; int n = 1, count = 0; uint64_t gen = 0; bool broken = false;
  bool wait() {
    std::unique_lock<std::mutex> lk(m);
    if (broken) return false;
    const uint64_t g = gen;
    if (++count == n) { count = 0; ++gen; cv.notify_all(); }
    else cv.wait(lk, [&] { return gen != g || broken; });
    return !broken;
  }
  void abort() {
    { std::lock_guard<std::mutex> lk(m); broken = true; }
    cv.notify_all();
  }
  bool is_broken() { std::lock_guard<std::mutex> lk(m); return broken; }
};

}  // namespace

struct rg_group;
namespace { struct ArUser { rg_group* g; int rank; }; }

struct rg_group {
  int n = 0, transport = 0;
  std::vector<ArUser> users;         // one per rank: the `user` pointer handed to the library's all-reduce callback
  std::vector<rg_ctx*> ctx;
  std::vector<ncclComm_t> comm;
  std::vector<double*> wview;        // per rank: [L][np_r][Np] (phenotype-sharded form), owned here
  std::vector<double*> sendbuf;      // per rank: packed rows for the all-to-all (RCCL transport)
  std::vector<size_t> sendbuf_bytes, wview_bytes;
  std::vector<std::vector<double>> hstage;   // peer transport: host staging of the all-reduce
  std::vector<const double*> ar_ptr;
  std::vector<int> failed;
  // one event per rank, recorded on the rank's stream when its part of an exchange is queued.  The exchanges are ordered by events and
  // stream order only: no host thread waits for the GPU inside rg_l0_finish (round 5; before, each transfer phase ended in a
  // hipStreamSynchronize).  A transfer that fails on the device surfaces at the rank's next rg_sync, like a kernel fault.
  std::vector<hipEvent_t> ev;
  Barrier bar;
  std::string err;
  // every rank passes `agree` right before it enters a collective: all ranks alive and willing, or nobody enters
  bool agree(int rank, bool ok) {
    failed[rank] = ok ? 0 : 1;
    if (!bar.wait()) return false;
    bool all = true;
    for (int k = 0; k < n; ++k) all = all && failed[k] == 0;
    if (!bar.wait()) return false;      // everyone has read the flags before anyone rewrites its own
    return all;
  }
};

namespace {

int rank_error(rg_group* g, int rank, const std::string& msg) {
  g->ctx[rank]->err = msg;
  return RG_ERR_HIP;
}

// sum all-reduce of a device buffer of rank `rank` (called from that rank's thread, by the library, between kernels)
int group_allreduce(void* user, void* dev_ptr, int64_t n) {
  ArUser* u = (ArUser*)user;
  rg_group* g = u->g;
  const int r = u->rank;
  rg_ctx* c = g->ctx[r];
  hipSetDevice(c->device);
  if (g->transport == RG_TRANSPORT_RCCL) {
    // all ranks are here and alive (a rank that failed earlier has broken the group): only then is the collective entered.
    // The all-reduce is ordered on the context's stream like the kernels around it -- no host synchronisation after it.
    if (!g->agree(r, true)) { c->err = "all-reduce: another GPU of the group has failed"; return 1; }
    ncclResult_t e = g_rccl.AllReduce(dev_ptr, dev_ptr, (size_t)n, kNcclDouble, kNcclSum, g->comm[r], c->stream);
    if (e != 0) { c->err = std::string("ncclAllReduce: ") + g_rccl.GetErrorString(e); g->bar.abort(); return 1; }
    return 0;
  }
  // peer transport: every rank stages its buffer in host memory, sums all of them in rank order, uploads
  std::vector<double>& h = g->hstage[r];
  h.resize((size_t)n);
  bool ok = hipMemcpyAsync(h.data(), dev_ptr, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream) == hipSuccess &&
            hipStreamSynchronize(c->stream) == hipSuccess;
  if (!g->agree(r, ok)) { c->err = "all-reduce: another GPU of the group has failed"; return 1; }
  std::vector<double> sum((size_t)n, 0.0);
  for (int k = 0; k < g->n; ++k) {
    const std::vector<double>& o = g->hstage[k];
    for (int64_t i = 0; i < n; ++i) sum[i] += o[i];
  }
  if (!g->bar.wait()) return 1;     // everyone has read every staging buffer
  if (hipMemcpyAsync(dev_ptr, sum.data(), sizeof(double) * n, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
      hipStreamSynchronize(c->stream) != hipSuccess) { g->bar.abort(); return 1; }
  return 0;
}

// Peer transport: rank `rank` has queued its pulls on its stream.  Nobody may overwrite its W (a next run) before every peer has pulled:
// each rank records its event, a host barrier makes sure all events ARE recorded (a wait on an event that has not been recorded yet
// would pass), then every rank's stream waits for every peer's event.  The host threads never wait for the device here.
bool chain_after_pulls(rg_group* g, int rank) {
  rg_ctx* c = g->ctx[rank];
  if (!g->ev[rank] || hipEventRecord(g->ev[rank], c->stream) != hipSuccess) { g->bar.abort(); return false; }
  if (!g->bar.wait()) return false;
  for (int k = 0; k < g->n; ++k)
    if (k != rank && g->ev[k] && hipStreamWaitEvent(c->stream, g->ev[k], 0) != hipSuccess) { g->bar.abort(); return false; }
  // the events must not be re-recorded (a next exchange) before every rank has queued its waits
  return g->bar.wait();
}

// exchange buffers of one rank for the phenotype-sharded form: the view [L][np_r][Np] and (RCCL) the packed send buffer
int ensure_exchange_buffers(rg_group* g, int rank, const int32_t* block_begin, const int32_t* pheno_begin) {
  rg_ctx* c = g->ctx[rank];
  const int R0 = c->R0, P = c->P;
  const int64_t Np = c->Np;
  const int nq = pheno_begin[rank + 1] - pheno_begin[rank];
  if (nq < 1) return rank_error(g, rank, "rg_l0_finish: a rank without phenotypes (use the all-gather form)");
  const size_t need = sizeof(double) * (size_t)c->B_total * R0 * nq * Np;
  if (g->wview_bytes[rank] < need) {
    if (g->wview[rank]) hipFree(g->wview[rank]);
    g->wview[rank] = nullptr; g->wview_bytes[rank] = 0;
    if (hipMalloc((void**)&g->wview[rank], need) != hipSuccess) return rank_error(g, rank, "rg_l0_finish: out of device memory for the phenotype view");
    g->wview_bytes[rank] = need;
  }
  if (g->transport == RG_TRANSPORT_RCCL) {
    const int64_t nlm = (int64_t)(block_begin[rank + 1] - block_begin[rank]) * R0;
    const size_t sneed = sizeof(double) * (size_t)nlm * P * Np;
    if (g->sendbuf_bytes[rank] < sneed || !g->sendbuf[rank]) {
      if (g->sendbuf[rank]) hipFree(g->sendbuf[rank]);
      g->sendbuf[rank] = nullptr; g->sendbuf_bytes[rank] = 0;
      if (hipMalloc((void**)&g->sendbuf[rank], sneed ? sneed : 8) != hipSuccess) return rank_error(g, rank, "rg_l0_finish: out of device memory for the send buffer");
      g->sendbuf_bytes[rank] = sneed;
    }
  }
  return RG_OK;
}

}  // namespace

extern "C" {

int rg_group_create(rg_group** out, int32_t n, rg_ctx* const* ctxs, int transport) {
  if (!out || n < 1 || !ctxs) return RG_ERR_ARG;
  *out = nullptr;
  rg_group* g = new rg_group();
  g->n = n; g->transport = transport;
  g->ctx.assign(ctxs, ctxs + n);
  g->wview.assign(n, nullptr); g->sendbuf.assign(n, nullptr);
  g->sendbuf_bytes.assign(n, 0); g->wview_bytes.assign(n, 0);
  g->hstage.resize(n); g->ar_ptr.assign(n, nullptr);
  g->users.resize(n);
  g->failed.assign(n, 0);
  for (int r = 0; r < n; ++r) g->users[r] = ArUser{g, r};
  g->bar.n = n;
  g->ev.assign(n, nullptr);
  for (int r = 0; r < n; ++r) {
    if (!ctxs[r]) continue;
    hipSetDevice(ctxs[r]->device);
    if (hipEventCreateWithFlags(&g->ev[r], hipEventDisableTiming) != hipSuccess) g->ev[r] = nullptr;
  }
  for (int r = 0; r < n; ++r)
    if (!ctxs[r] || !ctxs[r]->have_problem) { delete g; return RG_ERR_STATE; }
  if (transport == RG_TRANSPORT_RCCL) {
    std::string err;
    if (!g_rccl.load(err)) { ctxs[0]->err = err; delete g; return RG_ERR_HIP; }
    std::vector<int> devs(n);
    for (int r = 0; r < n; ++r) devs[r] = ctxs[r]->device;
    g->comm.assign(n, nullptr);
    ncclResult_t e = g_rccl.CommInitAll(g->comm.data(), n, devs.data());
    if (e != 0) { ctxs[0]->err = std::string("ncclCommInitAll: ") + g_rccl.GetErrorString(e); delete g; return RG_ERR_HIP; }
  } else if (transport == RG_TRANSPORT_PEER) {
    for (int r = 0; r < n; ++r)       // peer access between distinct devices (a shared device needs none)
      for (int k = 0; k < n; ++k)
        if (ctxs[r]->device != ctxs[k]->device) {
          hipSetDevice(ctxs[r]->device);
          hipError_t e = hipDeviceEnablePeerAccess(ctxs[k]->device, 0);
          if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) { ctxs[0]->err = "peer access unavailable"; delete g; return RG_ERR_HIP; }
          (void)hipGetLastError();
        }
  } else { delete g; return RG_ERR_ARG; }
  *out = g;
  return RG_OK;
}

void rg_group_destroy(rg_group* g) {
  if (!g) return;
  for (int r = 0; r < g->n; ++r) {
    hipSetDevice(g->ctx[r]->device);
    rg_set_l1_view(g->ctx[r], nullptr, 0, g->ctx[r]->P);
    rg_set_collective(g->ctx[r], 1, 0, nullptr, nullptr);
    if (g->wview[r]) hipFree(g->wview[r]);
    if (g->sendbuf[r]) hipFree(g->sendbuf[r]);
    if (r < (int)g->ev.size() && g->ev[r]) { hipStreamSynchronize(g->ctx[r]->stream); hipEventDestroy(g->ev[r]); }
    if (g->transport == RG_TRANSPORT_RCCL && r < (int)g->comm.size() && g->comm[r]) g_rccl.CommDestroy(g->comm[r]);
  }
  delete g;
}

// Marks the group broken on behalf of `rank` (a host-side failure of that rank: a reader / file error, an exception in its
// level 1): every rank that waits in a group call -- now or later -- returns with an error instead of waiting for it.
void rg_group_abort(rg_group* g, int32_t rank) {
  if (!g) return;
  if (rank >= 0 && rank < g->n) g->failed[rank] = 1;
  g->bar.abort();
}

// Allocates the exchange buffers of `rank` ahead of time (phenotype-sharded form), so that the 13 - 51 GB of the view and the
// send buffer are not allocated between level 0 and the exchange.  Optional: rg_l0_finish allocates what is missing.
int rg_group_prepare(rg_group* g, int32_t rank, const int32_t* block_begin, const int32_t* pheno_begin) {
  if (!g || rank < 0 || rank >= g->n || !block_begin) return RG_ERR_ARG;
  if (!pheno_begin) return RG_OK;       // all-gather form: the slabs land in W itself
  hipSetDevice(g->ctx[rank]->device);
  return ensure_exchange_buffers(g, rank, block_begin, pheno_begin);
}

// Called by every rank from its own host thread once its level-0 blocks are queued.  block_begin[n+1]: the contiguous
// block range of every rank (Data.cpp:270-302).  pheno_begin[n+1] != NULL selects the phenotype-sharded form (every rank
// needs at least one phenotype); NULL the all-gather form.
// Failure protocol: everything that can fail on the host side of a rank (its level 0, allocations, packing copies) happens
// BEFORE one agreement of all ranks; the collective is entered by all ranks or by none.  An error after that point (a RCCL or
// HIP call that fails) breaks the group, so that the peers' later barriers return instead of waiting.
int rg_l0_finish(rg_group* g, int32_t rank, const int32_t* block_begin, const int32_t* pheno_begin) {
  if (!g || rank < 0 || rank >= g->n || !block_begin) return RG_ERR_ARG;
  rg_ctx* c = g->ctx[rank];
  hipSetDevice(c->device);
  const int n = g->n, R0 = c->R0, P = c->P;
  const int64_t Np = c->Np;
  hipStream_t st = c->stream;
  // ---- phase A (may fail, rank-local): level 0 complete, buffers present, rows packed per destination ----
  int rc = rg_sync(c);                                        // level 0 of this rank is complete (and its deferred errors seen)
  if (rc == RG_OK && !c->d_W) rc = rank_error(g, rank, "rg_l0_finish: no level-0 predictors on this rank");
  if (rc == RG_OK && !pheno_begin && c->w_nb != c->B_total)
    rc = rank_error(g, rank, "rg_l0_finish: the all-gather form needs the full W on every rank (no rg_set_block_range)");
  const int q0 = pheno_begin ? pheno_begin[rank] : 0, nq = pheno_begin ? pheno_begin[rank + 1] - q0 : P;
  const int64_t l0m = (int64_t)block_begin[rank] * R0, nlm = (int64_t)(block_begin[rank + 1] - block_begin[rank]) * R0;
  std::vector<double*> sptr(n, nullptr);
  double* Wv = nullptr;
  if (rc == RG_OK && pheno_begin) {
    rc = ensure_exchange_buffers(g, rank, block_begin, pheno_begin);
    Wv = g->wview[rank];
    if (rc == RG_OK && g->transport == RG_TRANSPORT_RCCL) {
      // pack my rows per destination (strided device copies); my own share goes straight into the view (a single-rank
      // group sends it to itself instead, so that the send / receive path of the transport executes on a one-GPU box too)
      double* sp = g->sendbuf[rank];
      for (int k = 0; k < n && rc == RG_OK; ++k) {
        const int qk = pheno_begin[k], nk = pheno_begin[k + 1] - qk;
        sptr[k] = sp;
        if (nlm > 0 && nk > 0) {
          double* dst = (k == rank && n > 1) ? Wv + l0m * nq * Np : sp;
          if (hipMemcpy2DAsync(dst, sizeof(double) * nk * Np, rg_w_base(c) + (l0m * P + qk) * Np, sizeof(double) * P * Np,
                               sizeof(double) * nk * Np, (size_t)nlm, hipMemcpyDeviceToDevice, st) != hipSuccess)
            rc = rank_error(g, rank, "rg_l0_finish: pack failed");
        }
        sp += (size_t)nlm * nk * Np;
      }
    }
  }
  // ---- agreement: the ranks are threads of one process.  A rank that failed must not leave the others waiting in a
  //      receive, and -- peer transport -- nobody pulls from a W that is still being written ----
  if (!g->agree(rank, rc == RG_OK)) {
    if (rc == RG_OK) { c->err = "rg_l0_finish: level 0 failed on another GPU"; rc = RG_ERR_STATE; }
    return rc;
  }
  // ---- phase B: the exchange.  Errors from here on break the group (the peers are inside the same collective) ----
  auto broke = [&](const std::string& msg) { g->bar.abort(); return rank_error(g, rank, msg); };
  if (pheno_begin) {
    if (g->transport == RG_TRANSPORT_PEER) {
      for (int k = 0; k < n; ++k) {   // pull: rows of rank k's blocks, my phenotypes, straight out of k's W
        const int64_t l0 = (int64_t)block_begin[k] * R0, nl = (int64_t)(block_begin[k + 1] - block_begin[k]) * R0;
        if (nl == 0) continue;
        const double* src = rg_w_base(g->ctx[k]) + (l0 * P + q0) * Np;
        if (hipMemcpy2DAsync(Wv + l0 * nq * Np, sizeof(double) * nq * Np, src, sizeof(double) * P * Np, sizeof(double) * nq * Np,
                             (size_t)nl, hipMemcpyDeviceToDevice, st) != hipSuccess)
          return broke("rg_l0_finish: peer copy failed");
      }
      if (!chain_after_pulls(g, rank)) return rank_error(g, rank, "rg_l0_finish: the exchange failed on another GPU");
    } else {
      // one group of send/recv pairs per rank: a direct exchange in which every pair of GPUs uses its own xGMI link
      ncclResult_t e = g_rccl.GroupStart();
      for (int k = 0; k < n && e == 0; ++k) {
        if (k == rank && n > 1) continue;
        const int nk = pheno_begin[k + 1] - pheno_begin[k];
        const int64_t l0k = (int64_t)block_begin[k] * R0, nlk = (int64_t)(block_begin[k + 1] - block_begin[k]) * R0;
        if (nlm > 0 && nk > 0) e = g_rccl.Send(sptr[k], (size_t)nlm * nk * Np, kNcclDouble, k, g->comm[rank], st);
        if (e == 0 && nlk > 0) e = g_rccl.Recv(Wv + l0k * nq * Np, (size_t)nlk * nq * Np, kNcclDouble, k, g->comm[rank], st);
      }
      ncclResult_t e2 = g_rccl.GroupEnd();
      if (e != 0 || e2 != 0) return broke(std::string("RCCL all-to-all: ") + g_rccl.GetErrorString(e ? e : e2));
      // (stream-ordered: level 1 of this rank queues behind its receives, a next level 0 behind its sends)
    }
    rc = rg_set_l1_view(c, Wv, q0, nq);
    if (rc) return rc;
    return rg_set_collective(c, 1, 0, nullptr, nullptr);
  }
  // ---- all-gather form: every rank ends with the whole W; level 1 is then shared through all-reduces ----
  if (g->transport == RG_TRANSPORT_PEER) {
    for (int k = 0; k < n; ++k) {
      if (k == rank) continue;
      const int64_t l0 = (int64_t)block_begin[k] * R0, nl = (int64_t)(block_begin[k + 1] - block_begin[k]) * R0;
      if (nl == 0) continue;
      if (hipMemcpyAsync(c->d_W + l0 * P * Np, g->ctx[k]->d_W + l0 * P * Np, sizeof(double) * nl * P * Np, hipMemcpyDeviceToDevice, st) != hipSuccess)
        return broke("rg_l0_finish: peer copy failed");
    }
    if (!chain_after_pulls(g, rank)) return rank_error(g, rank, "rg_l0_finish: the exchange failed on another GPU");
  } else {
    ncclResult_t e = g_rccl.GroupStart();
    for (int k = 0; k < n && e == 0; ++k) {
      const int64_t l0 = (int64_t)block_begin[k] * R0, nl = (int64_t)(block_begin[k + 1] - block_begin[k]) * R0;
      if (nl == 0) continue;
      double* slab = c->d_W + l0 * P * Np;
      e = g_rccl.Broadcast(slab, slab, (size_t)nl * P * Np, kNcclDouble, k, g->comm[rank], st);
    }
    ncclResult_t e2 = g_rccl.GroupEnd();
    if (e != 0 || e2 != 0) return broke(std::string("RCCL all-gather: ") + g_rccl.GetErrorString(e ? e : e2));
  }
  for (int b = 0; b < c->B_total; ++b) c->block_done[b] = 1;
  rc = rg_set_l1_view(c, nullptr, 0, P);
  if (rc) return rc;
  // a single rank keeps the callback too: the tile-shared level 1 then runs with one rank and its all-reduces execute
  return rg_set_collective(c, n, rank, group_allreduce, &g->users[rank]);
}

}  // extern "C"
