// TEST SCAFFOLDING (see hip/hip_runtime.h beside this file): regenie_amd/csrc/chol_p128.h -- the panel-of-128 batched Cholesky of the level-0 ridge
// systems -- compiled by g++ for the host and executed workgroup by workgroup, work-items as fibers.  The matrix instruction, the cross-lane
// reads and the direct global -> LDS copy are restated here from their documented lane layouts (the same layouts every fp64 MFMA kernel of the
// library is written against and tested with on the device): v_mfma_f64_16x16x4 -- lane l supplies A[l & 15][l >> 4] and B[l >> 4][l & 15] and
// owns D[(l >> 4) + 4 r][l & 15], r = 0..3; global_load_lds_dwordx4 -- lane l's 16 bytes land at M0 + 16 l.
#define RG_HOST_EMU 1
#include "hip/hip_runtime.h"
#include <cstdint>
#include <cmath>

typedef double v4d __attribute__((vector_size(32)));
struct double2 { double x, y; };
using std::min;
using std::max;

namespace emu {
static thread_local double xa[1024], xb[1024];
static thread_local uint8_t* lds_base = nullptr;
struct Ring { unsigned seq = 0; int val[256]; };
static thread_local Ring ring[1024];
inline void reset() { for (auto& r : ring) r.seq = 0; }
}  // namespace emu

inline v4d __builtin_amdgcn_mfma_f64_16x16x4f64(double a, double b, v4d c, int, int, int) {
  hipcpu::Block& blk = hipcpu::g_blk;
  const int t = blk.cur, w = t >> 6, lane = t & 63, base = t & ~63;
  emu::xa[t] = a;
  emu::xb[t] = b;
  hipcpu::bar_wait(blk.wave[w]);
  const int i = lane & 15, q = lane >> 4;
  v4d d = c;
  for (int r = 0; r < 4; ++r) {
    double s = 0.0;
    for (int k = 0; k < 4; ++k) s += emu::xa[base + (q + 4 * r) + 16 * k] * emu::xb[base + i + 16 * k];
    d[r] += s;
  }
  hipcpu::bar_wait(blk.wave[w]);
  return d;
}
// v_readlane under divergence (only some lanes of the wave execute it, all of them the same sequence of calls): sequence-numbered mailboxes
inline int __builtin_amdgcn_readlane(int v, int src) {
  hipcpu::Block& blk = hipcpu::g_blk;
  const int t = blk.cur, base = t & ~63;
  emu::Ring& me = emu::ring[t];
  me.val[me.seq & 255] = v;
  const unsigned s = me.seq++;
  emu::Ring& o = emu::ring[base + src];
  while (o.seq <= s) hipcpu::yield();
  if (o.seq - s > 200) { fprintf(stderr, "emu: readlane skew\n"); abort(); }
  return o.val[s & 255];
}
inline int __double2loint(double x) { int64_t u; memcpy(&u, &x, 8); return (int)(uint32_t)u; }
inline int __double2hiint(double x) { int64_t u; memcpy(&u, &x, 8); return (int)(uint32_t)(u >> 32); }
inline double __hiloint2double(int hi, int lo) { const uint64_t u = ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo; double d; memcpy(&d, &u, 8); return d; }
inline double __builtin_amdgcn_rsq(double x) { return 1.0 / std::sqrt(x); }
inline int atomicMax(int32_t* p, int v) { const int o = *p; if (v > o) *p = v; return o; }
// (the RG_C128_DBG diagnostic's per-launch spans; never executed here -- a.dbg is null on the host)
inline unsigned long long atomicMin(unsigned long long* p, unsigned long long v) { const unsigned long long o = *p; if (v < o) *p = v; return o; }
inline unsigned long long atomicMax(unsigned long long* p, unsigned long long v) { const unsigned long long o = *p; if (v > o) *p = v; return o; }
inline void __threadfence() {}
inline void __threadfence_block() {}
#define C128_LDS_ADDR(p) (emu::lds_base = (uint8_t*)(p), 0u)
#define C128_STAGE_SYNC() __syncthreads()
#define C128_RFL(x) (x)
inline void c128_glds16(const void* sbase, uint32_t voff, uint32_t lds_addr) {
  memcpy(emu::lds_base + lds_addr + 16 * (hipcpu::g_blk.cur & 63), (const uint8_t*)sbase + voff, 16);
}

#include "../../regenie_amd/csrc/chol_p128.h"

// S: [nouter][n64][n64] source matrices (lower triangle + the embedded right-hand-side rows), shift: [R], d_n: [nouter] orders.
// Systems b = o * R + r.  Outputs: mats [batch][n64][n64], dinv [batch][n64/64][4096], linv [batch][n64/128][16384].
extern "C" int c128_host_factor(const double* S, int nouter, const double* shift, int R, const int32_t* d_n, int n64, int embed, int skip_pad,
                                const double* F, double* mats, double* dinv, double* linv, int32_t* info) {
  FormSrc f{};
  f.sum = S; f.sum_stride = (int64_t)n64 * n64; f.fold = F; f.fold_stride = (int64_t)n64 * n64; f.shift = shift; f.d_n = d_n;
  f.nfold = 1; f.nshift = R; f.n_fixed = n64; f.enabled = 1; f.subtract = F ? 1 : 0; f.extra = nullptr; f.extra_stride = 0; f.extra_row0 = 0;
  f.n64 = n64; f.n_div = 1; f.b_offset = 0; f.skip_pad = skip_pad; f.embed = embed;
  int64_t nl = 0;
  emu::reset();
  c128_launch_factor(nullptr, mats, (int64_t)n64 * n64, nouter * R, n64, dinv, linv, info, f, R, nl);
  return (int)nl;
}
