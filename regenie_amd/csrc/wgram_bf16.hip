// The weighted Gram X^T W X of the logistic / Poisson ridge IRLS (ridge_logistic_level_1, src/Step1_Models.cpp:1041-1101) as a
// QUASI-NEWTON Hessian on the 16-bit matrix cores.
//
// BASELINE configs[3] (50 binary traits, 500,000 samples, L = 2,560 level-0 predictors) spends 93 % of its level 1 in that Gram:
// 57 chain Grams per trait of 0.8 N L (L + 1) flop each, 47 ms apiece at 0.70 of the fp64 matrix peak (profiles/r4_config4_*).  No
// fp64 kernel can make that much cheaper, but the Gram does not have to be an fp64 quantity: the IRLS step
//   beta_new = (X^T W X + tau I)^-1 X^T W z        is the Newton step        beta_new = beta + H^-1 (X^T (y - p) - tau beta),
// and with the score X^T (y - p) - tau beta evaluated exactly (fp64 passes over the predictors, k_bt_score) the iteration
//   beta_new = beta + H~^-1 score(beta)
// has the SAME fixed point for any non-singular H~ -- the penalised maximum-likelihood estimate -- and the same stopping rule (max
// |score| < 1e-4 on the exact score).  H~ only has to be close enough to H that the step stays a Newton step: with ||H^-1 (H~ - H)|| =
// rho the error after a step is rho x the error before it plus the usual quadratic term.  Here H~ is formed from the operand
// V = W sqrt(w) on the 16-bit matrix cores, fp32 accumulators flushed into fp64 partial tiles every 4,096 positions (an fp32 running sum of
// half a million positive terms would otherwise lose its last digits).  Two operand formats:
//   * fp16, ONE plane and ONE product per operand pair (default; v_mfma_f32_32x32x16_f16).  V is standardised predictors times sqrt(w) <= 1/2:
//     well inside fp16's range, 11 significant bits, rounded to nearest.  The products are exact in fp32, the rounding errors of the two
//     operands are independent and average over the positions: an entry of H~ is off by about 2^-10.5 / sqrt(N) of the diagonal (1e-6 at
//     400,000 positions), ||H~ - H|| about 1e-4 of the diagonal, against eigenvalues of H + tau I that are at least tau and in practice 1e-2 of
//     the diagonal (the five ridge values of a block are correlated 0.9 - 0.99): rho = 1e-2, the IRLS takes the same number of rounds
//     (measured at 500,000 samples x 2,560 predictors: 24 rounds and 119 chain Grams against 24 and 120) and each Gram costs 5.3 instead of
//     11.7 ms;
//   * bf16 hi + lo planes, three products per pair (RG_WGRAM_FMT=bf16x3; hi hi^T + hi lo^T + lo hi^T, on the diagonal tiles also lo lo^T): 16
//     significant bits, entry errors 1e-8 of the diagonal -- the first form of this kernel, kept for comparison.
// Either way the results are held to the oracle at 1e-6 (tests/test_l1_models_gpu.py, tests/test_l1_full_width_gpu.py), which is what both
// sides' stopping rule (max |score| < 1e-4) allows.
// A chain that needs more than RG_WGRAM_SWITCH (12) steps at one ridge value finishes on the fp64 Gram (k_wgram128); RG_WGRAM_F64=1
// keeps the fp64 Gram everywhere.  Leave-one-out CV keeps the fp64 Gram (its leverages are read off the matrix itself).
//
// Data flow of one lock-step round over the unfinished fold models ("chains"; slot s = s-th active chain):
//   k_wsplit       V[s][row][chunk] = W[row][pos] * sqrt(w_chain(pos)) as fp16, 64 positions per 128-byte chunk (bf16x3: 32 positions,
//                  64 B of hi, 64 B of lo), the chain's held-out fold squeezed out of the position axis; W is read once for all slots
//   k_wgram_mx     one workgroup (16 waves, 32 x 128 outputs each) per 256 x 256 tile of the lower triangle, slot and K slice: the
//                  design of the FP4 fold Gram (gram_fp4.hip) -- 128-byte rows staged by direct global -> LDS copies, two buffers,
//                  XOR-swizzled 16-byte slots, four waves per SIMD -- with 16 matrix instructions per stage and wave (bf16x3: 24, diagonal tiles 32)
//   k_wg_reduce    (l1x.hip) sums the K slices in a fixed order and puts tau on the diagonal
#include <algorithm>
#include <cstdlib>
#include <string>
#include <vector>
#include "rg_internal.h"

#define WB_T 256          // output tile
#define WB_ROWB 128       // bytes per row and stage: 32 positions, hi then lo
// positions per stage (= per 128-byte chunk of an operand row) and stages between flushes of the fp32 accumulators into the fp64 partial
// tile (4,096 positions either way), per operand format: F16 = one fp16 plane (64 positions per chunk), else bf16 hi | lo (32)
template <bool F16> struct WbFmt { static constexpr int CHUNK = F16 ? 64 : 32, SHIFT = F16 ? 6 : 5, FLUSH = F16 ? 64 : 128; };
typedef float v16f __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;

// sw = sqrt(w) once per round (the operand scale of every predictor row)
__global__ __launch_bounds__(256) void k_sqrtw(const double* w, int64_t n, double* sw) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) sw[i] = sqrt(w[i]);
}

struct WSplitArgs {
  const double* W; int64_t Np; int L, P, p;
  const double* wv;            // [nchain][Np] SQUARE ROOTS of the IRLS weights (0 on masked / held-out / padding positions)
  const int32_t* chainmap;     // [nslot] chain of each slot
  int nslot, excl_own;
  uint8_t* V; int64_t v_row_bytes, v_slot_bytes;   // V[slot][row][chunk][128]
};

// block = 4 waves = 4 consecutive predictor rows x the same 512 positions (the weights are shared through the cache);
// lane i converts positions 8 i .. 8 i + 7 of the range = a quarter of a chunk: 16 bytes of hi, 16 bytes of lo per slot
template <bool F16>
__global__ __launch_bounds__(256) void k_wsplit(WSplitArgs a, SegLayout seg) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = blockIdx.y * 4 + wave;
  if (row >= a.L) return;
  const int64_t pos = (int64_t)blockIdx.x * 512 + 8 * lane;
  const int64_t all = seg.pos_start[seg.nseg - 1] + seg.plen[seg.nseg - 1];
  if (pos >= all) return;
  const double* w = a.W + ((int64_t)row * a.P + a.p) * a.Np + pos;
  double x[8];
#pragma unroll
  for (int k = 0; k < 8; k += 2) { const double2 t = *reinterpret_cast<const double2*>(w + k); x[k] = t.x; x[k + 1] = t.y; }
  constexpr int SH = WbFmt<F16>::SHIFT;
  const int64_t ca = pos >> SH;                      // chunk in position space
  // byte offset of this lane's 8 positions inside the chunk: an eighth of the fp16 chunk, or a quarter of the hi (and of the lo) half
  const int q = F16 ? (lane & 7) * 16 : (lane & 3) * 16;
  for (int s = 0; s < a.nslot; ++s) {
    const int chain = a.chainmap[s];
    int64_t cv = ca;
    if (a.excl_own) {
      const int64_t sk0 = seg.pos_start[chain] >> SH, skn = seg.plen[chain] >> SH;
      if (ca >= sk0 && ca < sk0 + skn) continue;      // the chain's held-out fold is not part of its operand
      if (ca >= sk0 + skn) cv = ca - skn;
    }
    const double* wt = a.wv + (int64_t)chain * a.Np + pos;
    unsigned hi[4], lo[4];
#pragma unroll
    for (int k = 0; k < 8; k += 2) {
      const double2 t = *reinterpret_cast<const double2*>(wt + k);
      const float v0 = (float)(x[k] * t.x), v1 = (float)(x[k + 1] * t.y);
      if (F16) {     // one fp16 value per entry (11 significant bits, round to nearest)
        const _Float16 f0 = (_Float16)v0, f1 = (_Float16)v1;
        hi[k >> 1] = (unsigned)__builtin_bit_cast(unsigned short, f0) | ((unsigned)__builtin_bit_cast(unsigned short, f1) << 16);
        lo[k >> 1] = 0u;
        continue;
      }
      const __bf16 h0 = (__bf16)v0, h1 = (__bf16)v1;
      const __bf16 l0 = (__bf16)(v0 - (float)h0), l1 = (__bf16)(v1 - (float)h1);
      hi[k >> 1] = (unsigned)__builtin_bit_cast(unsigned short, h0) | ((unsigned)__builtin_bit_cast(unsigned short, h1) << 16);
      lo[k >> 1] = (unsigned)__builtin_bit_cast(unsigned short, l0) | ((unsigned)__builtin_bit_cast(unsigned short, l1) << 16);
    }
    uint8_t* dst = a.V + (int64_t)s * a.v_slot_bytes + (int64_t)row * a.v_row_bytes + cv * WB_ROWB + q;
    *reinterpret_cast<uint4*>(dst) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
    if (!F16) *reinterpret_cast<uint4*>(dst + 64) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
  }
}

struct WbItem { int16_t tr, tc, slot, slice; };
struct WbArgs {
  const uint8_t* V; int64_t v_row_bytes, v_slot_bytes;
  int L, n64, nslot, nslice;
  const int32_t* chainmap; int excl_own;
  const WbItem* items; int nitem;
  double* part; int64_t out_stride;     // part[slice][slot][(n64 + 64) x n64], lower 64 x 64 tiles written
};

// one stage = 512 (SAME: 256) rows x 128 B; 1 KB (8 rows) per wave instruction, slot' = slot ^ ((row >> 1) & 7) on the global side
template <bool SAME>
__device__ __forceinline__ void wb_stage(const uint8_t* abase, int arows, const uint8_t* bbase, int brows, int64_t ld, int64_t kb0, uint8_t* buf) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int NPER = SAME ? 2 : 4;
#pragma unroll
  for (int i = 0; i < NPER; ++i) {
    const int g = wave * NPER + i;
    const int row = g * 8 + (lane >> 3);
    const bool isA = SAME || row < WB_T;
    const int r = isA ? row : row - WB_T;
    const int rows = isA ? arows : brows;
    const int rc = r < rows ? r : rows - 1;               // rows past the matrix repeat its last row: their results are never stored
    const int slot = (lane & 7) ^ ((row >> 1) & 7);
    const uint8_t* gp = (isA ? abase : bbase) + (int64_t)rc * ld + kb0 + slot * 16;
    __builtin_amdgcn_global_load_lds((glb_void_t*)gp, (lds_void_t*)(buf + g * 1024), 16, 0, 0);
  }
}

template <bool SAME, bool F16>
__device__ __forceinline__ void wb_tile(const WbArgs& g, const WbItem it, int64_t st0, int nstage, uint8_t* smem) {
  constexpr int WB_FLUSH = WbFmt<F16>::FLUSH;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int l31 = lane & 31, h = lane >> 5;
  const uint8_t* Vs = g.V + (int64_t)it.slot * g.v_slot_bytes + st0 * WB_ROWB;
  const uint8_t* abase = Vs + (int64_t)it.tr * WB_T * g.v_row_bytes;
  const uint8_t* bbase = Vs + (int64_t)it.tc * WB_T * g.v_row_bytes;
  const int arows = g.L - it.tr * WB_T, brows = g.L - it.tc * WB_T;
  v16f acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
  uint8_t* buf0 = smem;
  uint8_t* buf1 = smem + 2 * WB_T * WB_ROWB;
  wb_stage<SAME>(abase, arows, bbase, brows, g.v_row_bytes, 0, buf0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const int ra = wr * 32 + l31, rb = (SAME ? 0 : WB_T) + wc * 128 + l31;
  const int xa = (ra >> 1) & 7, xb = (rb >> 1) & 7;
  double* O = g.part + ((int64_t)it.slice * g.nslot + it.slot) * g.out_stride;
  bool first = true;
  // fp32 accumulators -> the workgroup's own fp64 partial tile: plain stores the first time, fp64 atomic adds afterwards (every element
  // has ONE writer -- this lane -- so the sum is a fixed sequence of additions; the atomics only save the read-back and its registers;
  // the tile stays in L2 between flushes).  Rows / columns in [L, n64) are the padding of the last 64-tile: zeros, not the clamped
  // rows' sums.  Offsets are 32-bit against the uniform tile base (n64^2 < 2^31).
  const int row_b = it.tr * WB_T + wr * 32 + 4 * h, col_b = it.tc * WB_T + wc * 128 + l31;
  auto flush = [&]() {
    int n64v = g.n64;
    asm volatile("" : "+v"(n64v));      // opaque: the 64 element offsets are formed here, not hoisted out of the stage loop (and spilled)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = col_b + j * 32;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = row_b + (r & 3) + 8 * (r >> 2);
        const bool in = row < g.n64 && col < g.n64 && (col >> 6) <= (row >> 6);     // lower 64 x 64 tiles (a diagonal tile: all of it)
        const double v = (row < g.L && col < g.L) ? (double)acc[j][r] : 0.0;
        const int off = row * n64v + col;
        if (in) {
          if (first) O[off] = v;
          else unsafeAtomicAdd(O + off, v);
        }
        acc[j][r] = 0.0f;
      }
      asm volatile("" ::: "memory");     // one column block at a time: keeps the 64 offsets from being live together
    }
    first = false;
  };
  // two loops: the fp32 -> fp64 flush sits between runs of WB_FLUSH stages, outside the stage loop, so that its temporaries do not
  // compete with the loop's registers (inside the loop they pushed the copy addresses into scratch)
  for (int c0 = 0; c0 < nstage; c0 += WB_FLUSH) {
    const int c1 = c0 + WB_FLUSH < nstage ? c0 + WB_FLUSH : nstage;
#pragma unroll 1
    for (int s = c0; s < c1; ++s) {
      uint8_t* cur = (s & 1) ? buf1 : buf0;
      uint8_t* nxt = (s & 1) ? buf0 : buf1;
      if (s + 1 < nstage) wb_stage<SAME>(abase, arows, bbase, brows, g.v_row_bytes, (int64_t)(s + 1) * WB_ROWB, nxt);
      const uint8_t* sa = cur + ra * WB_ROWB;
      const uint8_t* sb = cur + rb * WB_ROWB;
      if (F16) {
        // one product: four K = 16 steps over the chunk's 64 positions, 16-byte slot 2 ks + h of the row
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const v4i av = *reinterpret_cast<const v4i*>(sa + (((2 * ks + h) ^ xa) << 4));
          v4i bv[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) bv[j] = *reinterpret_cast<const v4i*>(sb + j * 32 * WB_ROWB + (((2 * ks + h) ^ xb) << 4));
          const f16x8 a_f = __builtin_bit_cast(f16x8, av);
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_f, __builtin_bit_cast(f16x8, bv[j]), acc[j], 0, 0, 0);
        }
      } else {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const v4i ah = *reinterpret_cast<const v4i*>(sa + (((2 * ks + h) ^ xa) << 4));
        const v4i al = *reinterpret_cast<const v4i*>(sa + (((4 + 2 * ks + h) ^ xa) << 4));
        v4i bh[4], bl[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          bh[j] = *reinterpret_cast<const v4i*>(sb + j * 32 * WB_ROWB + (((2 * ks + h) ^ xb) << 4));
          bl[j] = *reinterpret_cast<const v4i*>(sb + j * 32 * WB_ROWB + (((4 + 2 * ks + h) ^ xb) << 4));
        }
        const bf16x8 a_h = __builtin_bit_cast(bf16x8, ah), a_l = __builtin_bit_cast(bf16x8, al);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const bf16x8 b_h = __builtin_bit_cast(bf16x8, bh[j]), b_l = __builtin_bit_cast(bf16x8, bl[j]);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_h, b_h, acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_h, b_l, acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_l, b_h, acc[j], 0, 0, 0);
          if (SAME) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_l, b_l, acc[j], 0, 0, 0);
        }
      }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the next stage has landed in LDS
      __syncthreads();
    }
    flush();
  }
}

// Work items come from a host-built table (tile row, tile column, slot, K slice); item w = xcd * ceil(n / 8) + k for workgroup id
// 8 k + xcd, so that each of the eight XCDs walks a contiguous range of the table -- the tiles of one (slot, slice) -- and streams
// their common operand panels through its own L2 (as k_gram_fp4_blocks).
template <bool F16>
__global__ __launch_bounds__(1024) void k_wgram_mx(WbArgs g, SegLayout seg) {
  constexpr int WB_CHUNK = WbFmt<F16>::CHUNK;
  __shared__ __attribute__((aligned(16))) uint8_t smem[4 * WB_T * WB_ROWB];
  const int per_xcd = (g.nitem + 7) >> 3;
  const int w = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
  if ((int)(blockIdx.x >> 3) >= per_xcd || w >= g.nitem) return;
  const WbItem it = g.items[w];
  const int chain = g.chainmap[it.slot];
  const int64_t all = (seg.pos_start[seg.nseg - 1] + seg.plen[seg.nseg - 1]) / WB_CHUNK;
  const int64_t vs = all - (g.excl_own ? seg.plen[chain] / WB_CHUNK : 0);        // stages of this chain's operand
  const int64_t s0 = vs * it.slice / g.nslice, s1 = vs * (it.slice + 1) / g.nslice;
  if (it.tr == it.tc) wb_tile<true, F16>(g, it, s0, (int)(s1 - s0), smem);
  else wb_tile<false, F16>(g, it, s0, (int)(s1 - s0), smem);
}

// Forms the partial tiles of the `nslot` chains in `chainmap` into part[slice][slot]; returns the number of K slices (0 on failure).
// wv: [nchain][Np] weights, sw: scratch of the same size for their square roots; V / items live in workspace slots 15 / 13 of the context.
int rg_launch_wgram_bf16(rg_ctx* ctx, hipStream_t st, const double* W, int64_t Np, int L, int P, int p, int n64, const double* wv, double* sw, int nchain,
                         const int32_t* d_chainmap, const int32_t* h_chainmap, int nslot, int excl_own, double* part, int64_t out_stride,
                         int max_slices) {
  const SegLayout& seg = ctx->seg;
  // operand format: one fp16 plane and one product per pair (default), or bf16 hi + lo planes and three products (RG_WGRAM_FMT=bf16x3)
  static const bool F16 = !(getenv("RG_WGRAM_FMT") && std::string(getenv("RG_WGRAM_FMT")) == "bf16x3");
  const int WB_CHUNK = F16 ? WbFmt<true>::CHUNK : WbFmt<false>::CHUNK, WB_FLUSH = F16 ? WbFmt<true>::FLUSH : WbFmt<false>::FLUSH;
  const int64_t all = seg.pos_start[seg.nseg - 1] + seg.plen[seg.nseg - 1];
  int64_t min_vs = all / WB_CHUNK;
  if (excl_own) for (int s = 0; s < nslot; ++s) min_vs = std::min(min_vs, (all - seg.plen[h_chainmap[s]]) / WB_CHUNK);
  if (min_vs < 1) return 0;
  const int nt = (L + WB_T - 1) / WB_T, ntile = nt * (nt + 1) / 2;
  // K slices: enough work items for several rounds of the chip's 256 one-workgroup slots, the last round as full as possible
  int nslice = 1;
  {
    double best = -1.0;
    const int hi = (int)std::min<int64_t>(std::min(max_slices, 16), std::max<int64_t>(1, min_vs / (2 * WB_FLUSH)));
    for (int k = 1; k <= hi; ++k) {
      const int64_t items = (int64_t)ntile * nslot * k;
      const double eff = (double)items / (double)((items + 255) / 256 * 256);
      const double score = eff - (items < 512 ? 1.0 : 0.0) - 0.004 * k;      // at least two rounds; fewer slices among equals
      if (score > best) { best = score; nslice = k; }
    }
  }
  std::vector<WbItem> items;
  for (int s = 0; s < nslot; ++s)
    for (int k = 0; k < nslice; ++k)
      for (int tr = 0; tr < nt; ++tr)
        for (int tc = 0; tc <= tr; ++tc) items.push_back(WbItem{(int16_t)tr, (int16_t)tc, (int16_t)s, (int16_t)k});
  const int64_t v_row_bytes = (all / WB_CHUNK) * WB_ROWB, v_slot_bytes = v_row_bytes * L;
  uint8_t* V = (uint8_t*)rg_ws(ctx, 15, (size_t)v_slot_bytes * nslot);
  WbItem* d_items = (WbItem*)rg_ws(ctx, 13, sizeof(WbItem) * items.size());
  if (!V || !d_items) { ctx->err = "weighted Gram (bf16 operand planes): out of device memory"; return 0; }
  if (hipMemcpyAsync(d_items, items.data(), sizeof(WbItem) * items.size(), hipMemcpyHostToDevice, st) != hipSuccess ||
      hipStreamSynchronize(st) != hipSuccess) { ctx->err = "weighted Gram: copy of the work table failed"; return 0; }
  hipLaunchKernelGGL(k_sqrtw, dim3((unsigned)(((int64_t)nchain * Np + 255) / 256)), dim3(256), 0, st, wv, (int64_t)nchain * Np, sw);
  WSplitArgs sa{W, Np, L, P, p, sw, d_chainmap, nslot, excl_own, V, v_row_bytes, v_slot_bytes};
  if (F16) hipLaunchKernelGGL(k_wsplit<true>, dim3((unsigned)((all + 511) / 512), (unsigned)((L + 3) / 4)), dim3(256), 0, st, sa, seg);
  else hipLaunchKernelGGL(k_wsplit<false>, dim3((unsigned)((all + 511) / 512), (unsigned)((L + 3) / 4)), dim3(256), 0, st, sa, seg);
  WbArgs g{V, v_row_bytes, v_slot_bytes, L, n64, nslot, nslice, d_chainmap, excl_own, d_items, (int)items.size(), part, out_stride};
  if (F16) hipLaunchKernelGGL(k_wgram_mx<true>, dim3((unsigned)(((items.size() + 7) / 8) * 8)), dim3(1024), 0, st, g, seg);
  else hipLaunchKernelGGL(k_wgram_mx<false>, dim3((unsigned)(((items.size() + 7) / 8) * 8)), dim3(1024), 0, st, g, seg);
  if (hipGetLastError() != hipSuccess) return 0;   // the caller reports "0 slices" as a failure of the Gram
  return nslice;
}
