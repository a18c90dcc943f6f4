// Level-1 models beyond the QT K-fold path of l1.hip:
//   * QT leave-one-out CV          ridge_level_1_loocv          src/Step1_Models.cpp:875-962
//                                  make_predictions_loocv       src/Data.cpp:1269-1342
//   * BT logistic ridge, K-fold    ridge_logistic_level_1       src/Step1_Models.cpp:966-1156
//                                  make_predictions_binary      src/Data.cpp:1346-1427
//   * BT logistic ridge, LOOCV     ridge_logistic_level_1_loocv src/Step1_Models.cpp:1159-1286
//                                  run_log_ridge_loocv          src/Step1_Models.cpp:1288-1374
//                                  make_predictions_binary_loocv src/Data.cpp:1484-1571
//
// Building blocks (all fp64):
//   k_wgram        weighted Gram  sum_pos w(pos) W_r(pos) W_c(pos)  on v_mfma_f64_16x16x4_f64 straight from the
//                  [L][P][Np] predictor rows, one "chain" (= CV fold model, or the single LOOCV model) per
//                  grid.y, a per-chain extra row (working response) so the IRLS right-hand side X^T W z comes
//                  out of the same launch, the chain's held-out fold skipped, tau added on the diagonal.
//   batched Cholesky of chol.hip for the solves; for the leave-one-out leverages the identity is appended as
//                  right-hand-side rows, which the factorization turns into Y = L^-T, then H = Y Y^T = A^-1 and
//                  U^T = H W^T are two MFMA GEMMs; h_i = w_i . u_i is a streaming dot product.
// The reference gets the same quantities from a symmetric eigendecomposition (QT) or LLT + explicit
// solves against X^T (BT); the closed forms are identical, the factorization differs (agreement ~1e-11).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "rg_internal.h"

#define CT 64
#define NCH 8  // chains handled per pass of the streaming kernels

// ---------------------------------------------------------------------------------------------------------
struct WgArgs {
  const double* W; const double* zero; int64_t Np; int L, P, p, n64;
  const double* wv;        // [nchain][Np] weights (0 on masked / held-out / padding) or nullptr = 1
  const double* zv;        // [nchain][Np] extra row n64 (working response / y) or nullptr = zero row
  const double* dshift;    // [nchain] added to diagonal entries < L (entries >= L get 1.0); nullptr = none
  const int32_t* chainmap; // [gridDim.y] chain id of each output slot (nullptr = identity)
  int excl_own;            // chain c skips fold segment c (its held-out fold)
  double* out; int64_t out_stride;  // slot s -> out + s*out_stride, (n64+64) x n64 row-major
};
__device__ __forceinline__ const double* wg_row(const WgArgs& a, int chain, int row) {
  if (row < a.L) return a.W + ((int64_t)row * a.P + a.p) * a.Np;
  if (row == a.n64 && a.zv) return a.zv + (int64_t)chain * a.Np;
  return a.zero;
}

// One WAVE per 64x64 tile (4 x 4 MFMA 16x16x4 sub-tiles), four tiles of one tile column per workgroup, K loop in
// 8-deep chunks software-pipelined over two register sets (as k_chol_update / k_l1_gram64); the weights scale the
// A operand on the fly.  Tiles are enumerated column by column: column tc holds rows tc..T (T = the extra-row tile).
__global__ __launch_bounds__(256, 2) void k_wgram(WgArgs a, SegLayout seg, int T) {
  const int slot = blockIdx.y;
  const int chain = a.chainmap ? a.chainmap[slot] : slot;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ntile = T * (T + 1) / 2 + T;
  const int idx = blockIdx.x * 4 + wave;
  if (idx >= ntile) return;
  int tc = 0, rem = idx;
  while (rem >= T + 1 - tc) { rem -= T + 1 - tc; ++tc; }
  const int tr = tc + rem;
  const int i = lane & 15, q = lane >> 4;
  const double* A[4];
  const double* B[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    A[m] = wg_row(a, chain, tr * CT + m * 16 + i) + 2 * q;
    B[m] = wg_row(a, chain, tc * CT + m * 16 + i) + 2 * q;
  }
  const double* wrow = a.wv ? a.wv + (int64_t)chain * a.Np + 2 * q : nullptr;
  v4d acc[4][4];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[m][n] = (v4d){0, 0, 0, 0};
  auto load8 = [&](double2 (&av)[4], double2 (&bv)[4], int64_t p8) {   // p8: position / 8
    double2 w = make_double2(1.0, 1.0);
    if (wrow) w = *reinterpret_cast<const double2*>(wrow + p8 * 8);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const double2 x = *reinterpret_cast<const double2*>(A[m] + p8 * 8);
      av[m] = make_double2(x.x * w.x, x.y * w.y);
      bv[m] = *reinterpret_cast<const double2*>(B[m] + p8 * 8);
    }
  };
  auto mma8 = [&](const double2 (&av)[4], const double2 (&bv)[4]) {
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[m].x, bv[n].x, acc[m][n], 0, 0, 0);
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[m].y, bv[n].y, acc[m][n], 0, 0, 0);
  };
  for (int f = 0; f < seg.nseg; ++f) {
    if (a.excl_own && f == chain) continue;
    const int64_t k0 = seg.pos_start[f] / 8, k1 = k0 + seg.plen[f] / 8;   // (k1 - k0) is a multiple of 32
    double2 a0[4], b0[4], a1[4], b1[4];
    load8(a0, b0, k0);
    for (int64_t kc = k0; kc < k1; kc += 2) {
      load8(a1, b1, kc + 1);
      mma8(a0, b0);
      if (kc + 2 < k1) load8(a0, b0, kc + 2);
      mma8(a1, b1);
    }
  }
  const double sh = a.dshift ? a.dshift[chain] : 0.0;
  double* O = a.out + (int64_t)slot * a.out_stride + (int64_t)tr * CT * a.n64 + tc * CT;
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int lr = m * 16 + q + 4 * r, lc = n * 16 + i;
        double v = acc[m][n][r];
        if (a.dshift && tr == tc && lr == lc) v = (tr * CT + lr < a.L) ? v + sh : 1.0;
        O[(int64_t)lr * a.n64 + lc] = v;
      }
}

// ---- the weighted Gram through LDS (the k_l1_gram128 design of l1.hip): one workgroup per 128 x 128 macro tile ------------------------
// The logistic ridge of BASELINE configs[3] (50 binary traits at 500,000 samples, L = 2,560) spends its time here: K chains x R1 ridge
// values x 5 - 8 IRLS steps of 2 * 0.8 N * L^2 flop per phenotype.  The register-fed kernel above runs at ~30 TFLOP/s (L2 -> CU fabric);
// here the 256 operand rows of a macro tile AND the 16 weights of the stage go through a two-stage direct global -> LDS ring shared by four
// waves, the weights scale the A fragment as it leaves LDS.  A chain's held-out fold is a gap in its position range (the folds are
// contiguous in position space): the stage counter of a K slice simply jumps it.  Slices write partial tiles (k_wg_reduce sums them in a
// fixed order and puts tau on the diagonal); the working-response row X^T W z is k_wg_wz's.
struct WgItem { int16_t mr, mc, slot, slice; };
struct Wg128 {
  WgArgs a; int nslice, nslot; const WgItem* items; double* part;      // part [slice][slot][(n64 + 64) x n64]
};
#define WG128_STAGE (32768 + 128)   // 256 rows x 16 positions x 8 B, then the stage's 16 weights
__global__ __launch_bounds__(256, 2) void k_wgram128(Wg128 g, SegLayout seg) {
  extern __shared__ __attribute__((aligned(16))) uint8_t wg_smem[];
  const WgItem it = g.items[blockIdx.x];
  if (it.slot < 0) return;
  const WgArgs& a = g.a;
  const int chain = a.chainmap ? a.chainmap[it.slot] : it.slot;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int i = lane & 15, q = lane >> 4;
  const int n64 = a.n64;
  // virtual stages of the chain: every position but its held-out fold
  const int64_t all16 = (seg.pos_start[seg.nseg - 1] + seg.plen[seg.nseg - 1]) / 16;
  const int64_t skip_len = a.excl_own ? seg.plen[chain] / 16 : 0, skip_at = a.excl_own ? seg.pos_start[chain] / 16 : all16 + 1;
  const int64_t vs = all16 - skip_len;
  const int64_t v0 = vs * it.slice / g.nslice, v1 = vs * (it.slice + 1) / g.nslice;
  const int ns = (int)(v1 - v0);
  const int64_t pos0 = (v0 < skip_at ? v0 : v0 + skip_len) * 16;
  const double* src[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int rl = 8 * (wave * 8 + j) + (lane >> 3);
    const int row = rl < 128 ? it.mr * 128 + rl : it.mc * 128 + (rl - 128);
    const int slot = (lane & 7) ^ ((rl >> 1) & 7);
    src[j] = (row < a.L ? a.W + ((int64_t)row * a.P + a.p) * a.Np : a.zero) + pos0 + 2 * slot;
  }
  const double* wsrc = a.wv + (int64_t)chain * a.Np + pos0 + 2 * (lane & 7);      // the stage's weights: lanes 0..7 of wave 0
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)wg_smem;
  const uint32_t wdst = __builtin_amdgcn_readfirstlane(lds0 + wave * 8192);
  int64_t vnext = v0;
  auto issue = [&](int buf) {
#pragma unroll
    for (int j = 0; j < 8; ++j) glds16p(src[j], wdst + buf * WG128_STAGE + j * 1024);
    if (wave == 0 && lane < 8) glds16p(wsrc, lds0 + buf * WG128_STAGE + 32768);
    const int64_t adv = (vnext + 1 == skip_at) ? (1 + skip_len) * 16 : 16;
#pragma unroll
    for (int j = 0; j < 8; ++j) src[j] += adv;
    wsrc += adv;
    ++vnext;
  };
  const int row0 = it.mr * 128 + wr * 64, col0 = it.mc * 128 + wc * 64;
  const bool live = row0 >= col0 && row0 < n64 && col0 < n64;
  v4d acc[4][4];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[m][n] = (v4d){0, 0, 0, 0};
  const int xs = (i >> 1) & 7;
  const int oa = (wr * 64 + i) * 128 + ((q ^ xs) << 4), ob = 16384 + (wc * 64 + i) * 128 + ((q ^ xs) << 4);
  const int o4 = (((q + 4) ^ xs) << 4) - ((q ^ xs) << 4);
  if (ns > 0) issue(0);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  for (int s = 0; s < ns; ++s) {
    const uint8_t* cur = wg_smem + (s & 1) * WG128_STAGE;
    if (s + 1 < ns) issue((s + 1) & 1);
    if (live) {
      const double2 w0 = *reinterpret_cast<const double2*>(cur + 32768 + q * 16), w1 = *reinterpret_cast<const double2*>(cur + 32768 + (q + 4) * 16);
      double2 av[4][2], bv[4][2];
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const double2 x0 = *reinterpret_cast<const double2*>(cur + oa + m * 2048), x1 = *reinterpret_cast<const double2*>(cur + oa + m * 2048 + o4);
        av[m][0] = make_double2(x0.x * w0.x, x0.y * w0.y);
        av[m][1] = make_double2(x1.x * w1.x, x1.y * w1.y);
        bv[m][0] = *reinterpret_cast<const double2*>(cur + ob + m * 2048);
        bv[m][1] = *reinterpret_cast<const double2*>(cur + ob + m * 2048 + o4);
      }
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int n = 0; n < 4; ++n) {
            const double x = kk == 0 ? av[m][0].x : (kk == 1 ? av[m][0].y : (kk == 2 ? av[m][1].x : av[m][1].y));
            const double y = kk == 0 ? bv[n][0].x : (kk == 1 ? bv[n][0].y : (kk == 2 ? bv[n][1].x : bv[n][1].y));
            acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc[m][n], 0, 0, 0);
          }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
  if (!live) return;
  double* O = g.part + ((int64_t)it.slice * g.nslot + it.slot) * a.out_stride + (int64_t)row0 * n64 + col0;
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) O[(int64_t)(m * 16 + q + 4 * r) * n64 + n * 16 + i] = acc[m][n][r];
}

// out[slot] (lower 64 x 64 tiles) = sum over the K slices in slice order, + tau on the diagonal (1 on the padded diagonal)
__global__ __launch_bounds__(256) void k_wg_reduce(Wg128 g, int T) {
  const WgArgs& a = g.a;
  const int slot = blockIdx.y;
  const int chain = a.chainmap ? a.chainmap[slot] : slot;
  int tr = (int)((sqrtf(8.0f * blockIdx.x + 1.0f) - 1.0f) * 0.5f);
  while ((tr + 1) * (tr + 2) / 2 <= (int)blockIdx.x) ++tr;
  while (tr * (tr + 1) / 2 > (int)blockIdx.x) --tr;
  const int tc = blockIdx.x - tr * (tr + 1) / 2;
  const double sh = a.dshift ? a.dshift[chain] : 0.0;
  for (int e = threadIdx.x; e < CT * CT; e += 256) {
    const int lr = e >> 6, lc = e & 63;
    const int64_t off = (int64_t)(tr * CT + lr) * a.n64 + tc * CT + lc;
    double v = 0.0;
    for (int sl = 0; sl < g.nslice; ++sl) v += g.part[((int64_t)sl * g.nslot + slot) * a.out_stride + off];
    if (a.dshift && tr == tc && lr == lc) v = (tr * CT + lr < a.L) ? v + sh : 1.0;
    a.out[(int64_t)slot * a.out_stride + off] = v;
  }
}

// row n64 of every slot's matrix: X^T W z of the chain (the right-hand side of the IRLS step), one workgroup per (predictor, slot);
// rows past it in the tile are zeros.  The chain's held-out fold carries zero weights and is skipped.
__global__ __launch_bounds__(256) void k_wg_wz(WgArgs a, SegLayout seg) {
  __shared__ double red[4];
  const int l = blockIdx.x, slot = blockIdx.y;
  const int chain = a.chainmap ? a.chainmap[slot] : slot;
  double* orow = a.out + (int64_t)slot * a.out_stride + (int64_t)a.n64 * a.n64;
  if (l >= a.L) { if (threadIdx.x == 0) orow[l] = 0.0; return; }
  const double* w = a.W + ((int64_t)l * a.P + a.p) * a.Np;
  const double* wt = a.wv + (int64_t)chain * a.Np;
  const double* z = a.zv + (int64_t)chain * a.Np;
  double t0 = 0.0, t1 = 0.0;
  for (int f = 0; f < seg.nseg; ++f) {
    if (a.excl_own && f == chain) continue;
    const int64_t p0 = seg.pos_start[f], p1 = p0 + seg.plen[f];
    for (int64_t e = p0 + 2 * (int64_t)threadIdx.x; e < p1; e += 512) {
      const double2 x = *reinterpret_cast<const double2*>(w + e), ww = *reinterpret_cast<const double2*>(wt + e), zz = *reinterpret_cast<const double2*>(z + e);
      t0 = fma(x.x * ww.x, zz.x, t0);
      t1 = fma(x.y * ww.y, zz.y, t1);
    }
  }
  double t = t0 + t1;
  for (int o = 32; o > 0; o >>= 1) t += __shfl_down(t, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = t;
  __syncthreads();
  if (threadIdx.x == 0) orow[l] = (red[0] + red[1]) + (red[2] + red[3]);
}

// the weighted Gram of `nslot` chains: LDS-staged form when weights are given (the logistic / Poisson ridge), else the register-fed kernel
static int launch_wgram(rg_ctx* ctx, hipStream_t st, const WgArgs& a, int T, int nslot) {
  static const bool old = getenv("RG_WGRAM64") && atoi(getenv("RG_WGRAM64")) != 0;
  if (!a.wv || old) {
    hipLaunchKernelGGL(k_wgram, dim3((T * (T + 1) / 2 + T + 3) / 4, nslot), dim3(256), 0, st, a, ctx->seg, T);
    return RG_OK;
  }
  const int T2 = (a.n64 + 127) / 128, ntile2 = T2 * (T2 + 1) / 2;
  int64_t all16 = (ctx->seg.pos_start[ctx->seg.nseg - 1] + ctx->seg.plen[ctx->seg.nseg - 1]) / 16, min_vs = all16;
  if (a.excl_own) for (int f = 0; f < ctx->seg.nseg; ++f) min_vs = std::min(min_vs, all16 - ctx->seg.plen[f] / 16);
  int nslice = (int)std::min<int64_t>(16, std::max<int64_t>(1, (3072 + (int64_t)ntile2 * nslot - 1) / ((int64_t)ntile2 * nslot)));
  nslice = (int)std::max<int64_t>(1, std::min<int64_t>(nslice, min_vs / 8));
  // work table: the macro tiles of one 4 x 4 super tile of one (slot, slice) together, dealt to the eight XCDs (as l1_build_items)
  std::vector<std::vector<WgItem>> xl(8);
  int gi = 0;
  const int S = 4, TS = (T2 + S - 1) / S;
  for (int sl = 0; sl < nslot; ++sl)
    for (int ks = 0; ks < nslice; ++ks)
      for (int Mr = 0; Mr < TS; ++Mr)
        for (int Mc = 0; Mc <= Mr; ++Mc) {
          std::vector<WgItem>& dst = xl[gi % 8];
          bool any = false;
          for (int mr = Mr * S; mr < std::min(T2, (Mr + 1) * S); ++mr)
            for (int mc = Mc * S; mc < std::min(T2, (Mc + 1) * S); ++mc) {
              if (mc > mr) continue;
              dst.push_back(WgItem{(int16_t)mr, (int16_t)mc, (int16_t)sl, (int16_t)ks});
              any = true;
            }
          if (any) ++gi;
        }
  size_t mx = 0;
  for (auto& v : xl) mx = std::max(mx, v.size());
  std::vector<WgItem> items(mx * 8, WgItem{0, 0, -1, 0});
  for (int x = 0; x < 8; ++x)
    for (size_t j = 0; j < xl[x].size(); ++j) items[8 * j + x] = xl[x][j];
  WgItem* d_items = (WgItem*)rg_ws(ctx, 13, sizeof(WgItem) * items.size());
  double* d_part = (double*)rg_ws(ctx, 14, sizeof(double) * (size_t)nslice * nslot * a.out_stride);
  if (!d_items || !d_part) { ctx->err = "weighted Gram: out of device memory"; return RG_ERR_HIP; }
  RG_HIP(hipMemcpyAsync(d_items, items.data(), sizeof(WgItem) * items.size(), hipMemcpyHostToDevice, st));
  RG_HIP(hipStreamSynchronize(st));      // `items` is pageable host memory
  Wg128 g{a, nslice, nslot, d_items, d_part};
  const size_t lds = 2 * WG128_STAGE;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k_wgram128), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(k_wgram128, dim3((unsigned)items.size()), dim3(256), lds, st, g, ctx->seg);
  hipLaunchKernelGGL(k_wg_reduce, dim3(T * (T + 1) / 2, nslot), dim3(256), 0, st, g, T);
  // the right-hand-side row tile: zeros, then X^T W z in its first row
  RG_HIP(hipMemset2DAsync(a.out + (int64_t)a.n64 * a.n64, sizeof(double) * a.out_stride, 0, sizeof(double) * CT * a.n64, nslot, st));
  if (a.zv) hipLaunchKernelGGL(k_wg_wz, dim3(a.n64, nslot), dim3(256), 0, st, a, ctx->seg);
  return RG_OK;
}

// ---- Wt[pos][c] = W[c][p][pos] (c < L; 0 for L <= c < n64): sample-major copy, the K-contiguous operand of
//      U^T = H W^T.  grid (Np/64, n64/64) --------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_transpose_w(const double* W, int64_t Np, int L, int P, int p, int n64,
                                                     double* Wt) {
  __shared__ double s[64][65];
  const int64_t pos0 = (int64_t)blockIdx.x * 64;
  const int c0 = blockIdx.y * 64;
  {
    const int cl = threadIdx.x >> 2, qd = threadIdx.x & 3;
    const int c = c0 + cl;
    if (c < L) {
      const double* src = W + ((int64_t)c * P + p) * Np + pos0 + qd * 16;
#pragma unroll
      for (int i = 0; i < 16; ++i) s[cl][qd * 16 + i] = src[i];
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) s[cl][qd * 16 + i] = 0.0;
    }
  }
  __syncthreads();
  {
    const int pl = threadIdx.x >> 2, jq = threadIdx.x & 3;
    double* dst = Wt + (pos0 + pl) * n64 + c0 + jq * 16;
#pragma unroll
    for (int i = 0; i < 16; ++i) dst[i] = s[jq * 16 + i][pl];
  }
}

__global__ void k_eye(double* E, int n) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (int64_t)n * n) return;
  E[e] = (e / n == e % n) ? 1.0 : 0.0;
}

// b[r] = sum_k H[r][k] v[k]; one wave per row
__global__ __launch_bounds__(256) void k_symv(const double* H, const double* v, int n, double* b) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= n) return;
  double acc = 0.0;
  for (int k = lane; k < n; k += 64) acc = fma(H[(int64_t)r * n + k], v[k], acc);
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
  if (lane == 0) b[r] = acc;
}

// ---------------------------------------------------------------------------------------------------------
// Leave-one-out closed forms, thread = one position.  Ut[c][pos] = (H w_pos)[c], bv = H W^T y (QT) or beta (BT).
//   QT  (Step1_Models.cpp:934-952):  pred = (w.b - h y) / (1 - h)
//   BT  (Step1_Models.cpp:1221-1262): pred = w.beta - h (y - p) / (1 - h wgt) + offset -> p1 = clamp(sigmoid)
struct LooArgs {
  const double* W; const double* Ut; const double* bv; int64_t Np; int L, P, p;
  const double* y;      // QT: residualised y;  BT: raw 0/1 y            [Np]
  const double* rv;     // BT: y - p (0 on masked)                         [Np]
  const double* wv;     // BT: p(1-p) (0 on masked)                        [Np]
  const double* off;    // BT: offset                                      [Np]
  const double* maskp;  // BT: 0/1                                         [Np]
  int bt;               // 0 = QT, 1 = binary (logistic), 2 = count (Poisson: wv = mean, p1 = exp(eta))
};
#define LOO_NPART 6
__device__ __forceinline__ double block_sum_256(double x, double* sred /*[4]*/) {
  for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = x;
  __syncthreads();
  return (sred[0] + sred[1]) + (sred[2] + sred[3]);
}

// part: [gridDim.x][6] = Sx, Sy, Sx2, Sy2, Sxy, -LL  (QT fills Sx, Sx2, Sxy only)
__global__ __launch_bounds__(256) void k_loo_cv(LooArgs a, double* part) {
  __shared__ double sred[4];
  const int64_t pos = (int64_t)blockIdx.x * 256 + threadIdx.x;
  double t[LOO_NPART] = {0, 0, 0, 0, 0, 0};
  if (pos < a.Np) {
    double h = 0.0, xb = 0.0;
    const double* w = a.W + (int64_t)a.p * a.Np + pos;
    const double* u = a.Ut + pos;
    for (int c = 0; c < a.L; ++c) {
      const double x = w[(int64_t)c * a.P * a.Np];
      h = fma(x, u[(int64_t)c * a.Np], h);
      xb = fma(x, a.bv[c], xb);
    }
    if (!a.bt) {
      const double y = a.y[pos];
      const double pred = (xb - h * y) / (1.0 - h);
      t[0] = pred; t[2] = pred * pred; t[4] = pred * y;
    } else if (a.maskp[pos] != 0.0) {
      const double y = a.y[pos];
      const double eta = xb - h * a.rv[pos] / (1.0 - h * a.wv[pos]) + a.off[pos];
      double p1;
      if (a.bt == 2) {                         // count trait: mean exp(eta), floored (Step1_Models.cpp:1669-1684)
        p1 = fmax(exp(eta), 1e-5);
        t[5] = -(y * log(p1) - p1);
      } else {
        p1 = 1.0 - 1.0 / (exp(eta) + 1.0);
        p1 = fmin(fmax(p1, 1e-5), 1.0 - 1e-5);   // l1_ridge_eps, Step1_Models.cpp:1256-1257
        t[5] = -((y == 0.0) ? log(1.0 - p1) : log(p1));
      }
      t[0] = p1; t[1] = y; t[2] = p1 * p1; t[3] = y * y; t[4] = p1 * y;
    }
  }
#pragma unroll
  for (int k = 0; k < LOO_NPART; ++k) {
    const double s = block_sum_256(t[k], sred);
    if (threadIdx.x == 0) part[(int64_t)blockIdx.x * LOO_NPART + k] = s;
  }
}

// pred[c][n] (compact sample order) = sum_{col in chr c} w_col (b_col - u_col g),
//   QT: g = (y - w.b) / (1 - h)     (Data.cpp:1311-1326)      BT: g = (y - p) / (1 - h wgt)  (Data.cpp:1530-1548)
__global__ __launch_bounds__(256) void k_loo_pred(LooArgs a, const int32_t* chr_col0, int nchr,
                                                  const int32_t* cidx, int64_t N, double* pred) {
  const int64_t pos = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (pos >= a.Np) return;
  const int32_t n = cidx[pos];
  if (n < 0) return;
  const double* w = a.W + (int64_t)a.p * a.Np + pos;
  const double* u = a.Ut + pos;
  double h = 0.0, xb = 0.0;
  for (int c = 0; c < a.L; ++c) {
    const double x = w[(int64_t)c * a.P * a.Np];
    h = fma(x, u[(int64_t)c * a.Np], h);
    xb = fma(x, a.bv[c], xb);
  }
  const double g = a.bt ? a.rv[pos] / (1.0 - h * a.wv[pos]) : (a.y[pos] - xb) / (1.0 - h);
  for (int c = 0; c < nchr; ++c) {
    double s1 = 0.0, s2 = 0.0;
    for (int col = chr_col0[c]; col < chr_col0[c + 1]; ++col) {
      const double x = w[(int64_t)col * a.P * a.Np];
      s1 = fma(x, a.bv[col], s1);
      s2 = fma(x, u[(int64_t)col * a.Np], s2);
    }
    pred[(int64_t)c * N + n] = s1 - s2 * g;
  }
}

// ---------------------------------------------------------------------------------------------------------
// Logistic state at beta for up to NCH chains at once (one read of W), thread = one position.
//   training position of chain ch (mask, not in ch's held-out fold):  eta = off + w.beta, p = get_pvec(eta)
//   (Step1_Models.cpp:1799-1806), wgt = p(1-p), z = w.beta + (y-p)/wgt, r = y - p, deviance partial;
//   held-out position (K-fold only): test p1 = sigmoid(off + w.beta) clamped, the six running sums
//   (Step1_Models.cpp:1113-1134).
struct BtArgs {
  const double* W; int64_t Np; int L, P, p, n64;
  const double* yraw; const double* off; const double* maskp;
  const double* beta;   // [nchain][n64]
  int nchain, kfold;
  double *wv, *zv, *rv; // [nchain][Np]
  int family;           // 0 = logistic, 1 = Poisson (p = exp(eta), wgt = p; Step1_Models.cpp:1813-1818, :1483-1493)
};
#define BT_NPART 8      // Sx, Sy, Sx2, Sy2, Sxy, -LL (held-out), deviance (training), #(wgt == 0)
__device__ __forceinline__ double rg_pvec(double eta) {
  const double eps = 10.0 * 2.220446049250313e-16;   // numtol_eps, Regenie.hpp:225
  double p = 1.0 - 1.0 / (exp(eta) + 1.0);
  if (eta < -30.0) p = eps / (1.0 + eps);           // ETAMINTHR / ETAMAXTHR, Step1_Models.hpp:30-31
  if (eta > 30.0) p = 1.0 / (1.0 + eps);
  return p;
}

__global__ __launch_bounds__(256) void k_bt_eval(BtArgs a, int ch0, const int32_t* chunk_seg,
                                                 const int64_t* chunk_pos, const int64_t* chunk_len,
                                                 double* part /*[nchunk][nchain][BT_NPART]*/) {
  __shared__ double sB[256][NCH];
  __shared__ double sred[4];
  const int chunk = blockIdx.x;
  const int f = chunk_seg[chunk];
  const int64_t p0 = chunk_pos[chunk], plen = chunk_len[chunk];
  const int nc = min(NCH, a.nchain - ch0);
  const bool live = threadIdx.x < plen;   // chunks are <= 256 positions
  const int64_t pos = p0 + threadIdx.x;
  double acc[NCH];
#pragma unroll
  for (int j = 0; j < NCH; ++j) acc[j] = 0.0;
  for (int c0 = 0; c0 < a.L; c0 += 256) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int col = c0 + threadIdx.x;
      sB[threadIdx.x][j] = (j < nc && col < a.L) ? a.beta[(int64_t)(ch0 + j) * a.n64 + col] : 0.0;
    }
    __syncthreads();
    if (live) {
      const int cn = min(256, a.L - c0);
      const double* w = a.W + ((int64_t)c0 * a.P + a.p) * a.Np + pos;
      for (int c = 0; c < cn; ++c) {
        const double x = w[(int64_t)c * a.P * a.Np];
#pragma unroll
        for (int j = 0; j < NCH; ++j) acc[j] = fma(x, sB[c][j], acc[j]);
      }
    }
  }
  const double m = live ? a.maskp[pos] : 0.0;
  const double y = live ? a.yraw[pos] : 0.0;
  const double off = live ? a.off[pos] : 0.0;
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    if (j >= nc) break;
    const int ch = ch0 + j;
    const bool test = a.kfold && (f == ch);
    double t[BT_NPART] = {0, 0, 0, 0, 0, 0, 0, 0};
    double wgt = 0.0, z = 0.0, r = 0.0;
    if (m != 0.0) {
      if (!test) {
        double p;
        if (a.family == 1) {
          p = exp(off + acc[j]);
          wgt = p;
          t[6] = -2.0 * (y * log(p) - p);
        } else {
          p = rg_pvec(off + acc[j]);
          wgt = p * (1.0 - p);
          t[6] = -2.0 * ((y == 0.0) ? log(1.0 - p) : log(p));
        }
        r = y - p;
        if (wgt == 0.0) t[7] = 1.0;
        else z = acc[j] + r / wgt;
      } else {
        double p1;
        if (a.family == 1) {
          p1 = fmax(exp(off + acc[j]), 1e-5);     // l1_ridge_eps floor, Step1_Models.cpp:1560-1561
          t[5] = -(y * log(p1) - p1);
        } else {
          p1 = 1.0 - 1.0 / (exp(off + acc[j]) + 1.0);
          p1 = fmin(fmax(p1, 1e-5), 1.0 - 1e-5);
          t[5] = -((y == 0.0) ? log(1.0 - p1) : log(p1));
        }
        t[0] = p1; t[1] = y; t[2] = p1 * p1; t[3] = y * y; t[4] = p1 * y;
      }
    }
    if (live) {
      a.wv[(int64_t)ch * a.Np + pos] = wgt;
      a.zv[(int64_t)ch * a.Np + pos] = z;
      a.rv[(int64_t)ch * a.Np + pos] = r;
    }
#pragma unroll
    for (int k = 0; k < BT_NPART; ++k) {
      const double s = block_sum_256(t[k], sred);
      if (threadIdx.x == 0) part[((int64_t)chunk * a.nchain + ch) * BT_NPART + k] = s;
    }
  }
}

// score[ch][c] = sum_pos W[c][pos] r_ch(pos) - tau_ch beta_ch[c]   (Step1_Models.cpp:1088-1092 / :1361)
// One workgroup per BT_SROWS predictor rows: the residual vectors of the chains are read once per workgroup, not once per predictor
// (one row per workgroup moved 5 x 4 MB of residuals through L2 for every 4 MB row of W: 8.3 ms per call at 500,000 samples and
// L = 2,560 -- 1.25 TB/s on the bytes of W; the pass is now bound by W itself).  Fixed summation order: thread t takes positions
// t, t + 256, ..., then the block reduction.
// Round 6: 16-byte loads -- thread t takes the position pairs (2t, 2t + 1), (2t + 512, ...), ... (8-byte loads, one position per thread and
// pass of the loop, kept the kernel at 2.9 ms per pass where k_bt_eval reads the same bytes in 1.9; four rows per workgroup instead of eight
// measured slower, 3.05 ms).  Still a fixed summation order.
#define BT_SROWS 8
__global__ __launch_bounds__(256) void k_bt_score(BtArgs a, int ch0, const double* tauc, double* score) {
  __shared__ double sred[4];
  const int c0 = blockIdx.x * BT_SROWS;
  const int nc = min(NCH, a.nchain - ch0);
  const double* w[BT_SROWS];
#pragma unroll
  for (int k = 0; k < BT_SROWS; ++k) w[k] = a.W + ((int64_t)min(c0 + k, a.L - 1) * a.P + a.p) * a.Np;
  double acc[BT_SROWS][NCH];
#pragma unroll
  for (int k = 0; k < BT_SROWS; ++k)
#pragma unroll
    for (int j = 0; j < NCH; ++j) acc[k][j] = 0.0;
  for (int64_t pos = 2 * threadIdx.x; pos < a.Np; pos += 512) {      // Np is a multiple of 256: pos + 1 < Np
    double2 x[BT_SROWS], r[NCH];
#pragma unroll
    for (int k = 0; k < BT_SROWS; ++k) x[k] = *reinterpret_cast<const double2*>(w[k] + pos);
#pragma unroll
    for (int j = 0; j < NCH; ++j) r[j] = *reinterpret_cast<const double2*>(a.rv + (int64_t)(ch0 + (j < nc ? j : 0)) * a.Np + pos);
#pragma unroll
    for (int k = 0; k < BT_SROWS; ++k)
#pragma unroll
      for (int j = 0; j < NCH; ++j) acc[k][j] = fma(x[k].y, r[j].y, fma(x[k].x, r[j].x, acc[k][j]));
  }
#pragma unroll
  for (int k = 0; k < BT_SROWS; ++k)
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      if (j >= nc) break;
      const double s = block_sum_256(acc[k][j], sred);
      if (threadIdx.x == 0 && c0 + k < a.L)
        score[(int64_t)(ch0 + j) * a.n64 + c0 + k] = s - tauc[ch0 + j] * a.beta[(int64_t)(ch0 + j) * a.n64 + c0 + k];
    }
}

// per-chromosome linear predictors with fold-specific coefficient vectors (make_predictions /
// make_predictions_binary, Data.cpp:1242-1258, :1398-1414): alpha of fold f = alpha0 + f * alpha_stride
__global__ __launch_bounds__(256) void k_fold_pred(const double* W, int64_t Np, int L, int P, int p,
                                                   const double* alpha0, int64_t alpha_stride,
                                                   const int32_t* chunk_seg, const int64_t* chunk_pos,
                                                   const int64_t* chunk_len, const int32_t* chr_col0, int nchr,
                                                   const int32_t* cidx, int64_t N, double* pred) {
  extern __shared__ double sAl[];
  const int ch = blockIdx.x;
  const int f = chunk_seg[ch];
  const int64_t p0 = chunk_pos[ch], plen = chunk_len[ch];
  const double* al = alpha0 + (int64_t)f * alpha_stride;
  for (int c = threadIdx.x; c < L; c += 256) sAl[c] = al[c];
  __syncthreads();
  if (threadIdx.x >= plen) return;
  const int64_t pos = p0 + threadIdx.x;
  const int32_t n = cidx[pos];
  if (n < 0) return;
  for (int c = 0; c < nchr; ++c) {
    double acc = 0.0;
    const double* w = W + ((int64_t)chr_col0[c] * P + p) * Np + pos;
    const int nn = chr_col0[c + 1] - chr_col0[c];
    for (int t = 0; t < nn; ++t) acc = fma(w[(int64_t)t * P * Np], sAl[chr_col0[c] + t], acc);
    pred[(int64_t)c * N + n] = acc;
  }
}

// ---------------------------------------------------------------------------------------------------------
namespace {

struct DevBufs {
  std::vector<void*> ptrs;
  ~DevBufs() { for (void* p : ptrs) if (p) hipFree(p); }
  template <class T> hipError_t alloc(T** p, size_t n) {
    *p = nullptr;
    hipError_t e = hipMalloc((void**)p, std::max<size_t>(n, 1) * sizeof(T));
    if (e == hipSuccess) ptrs.push_back(*p);
    return e;
  }
};

#define L1X_HIP(call)                                                                  \
  do {                                                                                 \
    hipError_t e_ = (call);                                                            \
    if (e_ != hipSuccess) {                                                            \
      ctx->err = std::string(#call) + ": " + hipGetErrorString(e_);                    \
      return RG_ERR_HIP;                                                               \
    }                                                                                  \
  } while (0)

// brackets the launches of a scope with the context's two events when timing is on (rg_enable_timing) and adds the elapsed time to
// a slot of rg_timing; a no-op otherwise (the waits would serialise the host with the device)
struct L1Lap {
  rg_ctx* c; hipStream_t st; double* slot;
  L1Lap(rg_ctx* ctx, hipStream_t s, double* sl) : c(ctx), st(s), slot(sl) { if (c->timing) hipEventRecord(c->ev0, st); }
  ~L1Lap() {
    if (!c->timing) return;
    hipEventRecord(c->ev1, st); hipEventSynchronize(c->ev1);
    float ms = 0.f; hipEventElapsedTime(&ms, c->ev0, c->ev1); *slot += ms;
  }
};

struct L1Common {
  rg_ctx* ctx; hipStream_t st;
  int L, P, n64, rtot, T, nchr;   // P: phenotypes of the current view
  const double* Wv; int Pv, p0v; bool view;   // predictor buffer / its phenotype stride / first global phenotype
  int64_t msz, Np, N;
  std::vector<int32_t> col0;
  int32_t* d_col0 = nullptr;
  double* d_pred = nullptr;
  DevBufs bufs;
};

int l1_common_init(rg_ctx* ctx, L1Common& c, int nchr, const int32_t* cols_per_chr, const char* who) {
  if (!ctx->have_problem || !(ctx->d_W || ctx->v_W)) { ctx->err = std::string(who) + ": no level-0 predictors"; return RG_ERR_STATE; }
  c.ctx = ctx; c.st = ctx->stream;
  c.L = ctx->B_total * ctx->R0; c.P = ctx->v_np; c.Np = ctx->Np; c.N = ctx->N; c.nchr = nchr;
  if (!ctx->v_W && ctx->w_nb != ctx->B_total) { ctx->err = std::string(who) + ": W holds a block range only (rg_set_block_range): level 1 needs the exchanged view"; return RG_ERR_STATE; }
  c.view = ctx->v_W != nullptr; c.Wv = c.view ? ctx->v_W : ctx->d_W; c.Pv = c.view ? ctx->v_np : ctx->P; c.p0v = ctx->v_p0;
  c.col0.assign(nchr + 1, 0);
  for (int i = 0; i < nchr; ++i) c.col0[i + 1] = c.col0[i] + cols_per_chr[i];
  if (c.col0[nchr] != c.L) { ctx->err = std::string(who) + ": cols_per_chr does not sum to n_blocks*R0"; return RG_ERR_ARG; }
  c.n64 = (int)rg_round_up(c.L, CT); c.rtot = c.n64 + CT; c.T = c.n64 / CT;
  c.msz = (int64_t)c.rtot * c.n64;
  L1X_HIP(c.bufs.alloc(&c.d_col0, nchr + 1));
  L1X_HIP(c.bufs.alloc(&c.d_pred, (size_t)nchr * c.N));
  L1X_HIP(hipMemcpyAsync(c.d_col0, c.col0.data(), sizeof(int32_t) * (nchr + 1), hipMemcpyHostToDevice, c.st));
  return RG_OK;
}

// compact (N) host vector -> position space (Np), zeros elsewhere
void to_pos(const rg_ctx* ctx, const double* src, std::vector<double>& dst) {
  dst.assign((size_t)ctx->Np, 0.0);
  for (int64_t n = 0; n < ctx->N; ++n) dst[(size_t)ctx->h_posc[n]] = src[n];
}

int check_spd(rg_ctx* ctx, bool* bad) {
  int32_t info[2] = {0, 0};
  L1X_HIP(hipMemcpy(info, ctx->d_info, sizeof(info), hipMemcpyDeviceToHost));
  *bad = info[1] != 0;
  if (*bad) L1X_HIP(hipMemset(ctx->d_info + 1, 0, sizeof(int32_t)));
  return RG_OK;
}

// H = (G + tau I)^-1 for `nsh` shifts at once.  G: (n64+64) x n64 (lower triangle used).
//   sysI: [nsh][2*n64][n64] workspace; on exit rows n64.. of system s hold Y_s = L_s^-T;  Hs[s] = Y_s Y_s^T.
int invert_shifted(rg_ctx* ctx, const L1Common& c, const double* d_G, const double* d_shift, int nsh,
                   const double* d_eye, double* d_sysI, double* d_dinv, double* d_H /*[nsh][n64][n64]*/) {
  const int n64 = c.n64;
  const int64_t ssz = (int64_t)2 * n64 * n64;
  rg_launch_chol_solve_formed_x(c.st, d_G, 0, nullptr, 0, 1, d_shift, nsh, nullptr, c.L, 1, d_sysI, ssz, n64,
                                n64, 0, d_dinv, ctx->d_info + 1, &ctx->tm.n_chol_launches, 0, d_eye, 0, n64);
  for (int s = 0; s < nsh; ++s) {
    const double* Y = d_sysI + s * ssz + (int64_t)n64 * n64;
    rg_launch_dgemm_nt(c.st, Y, n64, Y, n64, n64, n64, n64, d_H + (int64_t)s * n64 * n64, n64);
  }
  return RG_OK;
}

}  // namespace

// =========================================================================================================
// QT, leave-one-out
// =========================================================================================================
int rg_l1_qt_loocv_impl(rg_ctx* ctx, int R1, const double* tau, int nchr, const int32_t* cols_per_chr,
                        double* cumsum_out, int32_t* best_out, double* pred_out) {
  if (R1 < 1 || R1 > 16) { ctx->err = "rg_l1_qt_loocv: n_ridge_l1 must be in [1,16]"; return RG_ERR_ARG; }
  if (!ctx->loocv) { ctx->err = "rg_l1_qt_loocv: the problem was set up for K-fold CV"; return RG_ERR_STATE; }
  L1Common c;
  int rc = l1_common_init(ctx, c, nchr, cols_per_chr, "rg_l1_qt_loocv");
  if (rc) return rc;
  hipStream_t st = c.st;
  const int L = c.L, P = c.P, n64 = c.n64, T = c.T;
  const int64_t Np = c.Np, N = c.N;
  const unsigned gpos = (unsigned)((Np + 255) / 256);
  double *d_G, *d_eye, *d_sysI, *d_dinv, *d_H, *d_tau, *d_Wt, *d_Ut, *d_b, *d_part;
  L1X_HIP(c.bufs.alloc(&d_G, (size_t)c.msz));
  L1X_HIP(c.bufs.alloc(&d_eye, (size_t)n64 * n64));
  L1X_HIP(c.bufs.alloc(&d_sysI, (size_t)R1 * 2 * n64 * n64));
  L1X_HIP(c.bufs.alloc(&d_dinv, rg_chol_ws_doubles((size_t)R1, c.n64)));
  L1X_HIP(c.bufs.alloc(&d_H, (size_t)R1 * n64 * n64));
  L1X_HIP(c.bufs.alloc(&d_tau, (size_t)R1));
  L1X_HIP(c.bufs.alloc(&d_Wt, (size_t)Np * n64));
  L1X_HIP(c.bufs.alloc(&d_Ut, (size_t)Np * n64));
  L1X_HIP(c.bufs.alloc(&d_b, (size_t)n64));
  L1X_HIP(c.bufs.alloc(&d_part, (size_t)gpos * LOO_NPART));
  hipLaunchKernelGGL(k_eye, dim3((unsigned)(((int64_t)n64 * n64 + 255) / 256)), dim3(256), 0, st, d_eye, n64);
  std::vector<double> hpart((size_t)gpos * LOO_NPART);

  for (int p = 0; p < P; ++p) {
    const int pg = c.p0v + p, pw = c.view ? p : pg;   // global phenotype / index inside the predictor buffer
    const double* d_y = ctx->d_V + (int64_t)(ctx->C + pg) * Np;
    // G = W^T W with W^T y as row n64 (xtx, zvec of Step1_Models.cpp:893-907)
    WgArgs g{c.Wv, ctx->d_zero, Np, L, c.Pv, pw, n64, nullptr, d_y, nullptr, nullptr, 0, d_G, c.msz};
    hipLaunchKernelGGL(k_wgram, dim3((T * (T + 1) / 2 + T + 3) / 4, 1), dim3(256), 0, st, g, ctx->seg, T);
    hipLaunchKernelGGL(k_transpose_w, dim3((unsigned)(Np / 64), n64 / 64), dim3(256), 0, st, c.Wv, Np, L, c.Pv, pw,
                       n64, d_Wt);
    L1X_HIP(hipMemcpyAsync(d_tau, tau + (int64_t)p * R1, sizeof(double) * R1, hipMemcpyHostToDevice, st));
    rc = invert_shifted(ctx, c, d_G, d_tau, R1, d_eye, d_sysI, d_dinv, d_H);
    if (rc) return rc;
    const double* d_z = d_G + (int64_t)n64 * n64;  // W^T y (padded with zeros to n64)
    double* cs = cumsum_out + (int64_t)p * 5 * R1;
    for (int t = 0; t < 5 * R1; ++t) cs[t] = 0.0;
    LooArgs la{c.Wv, d_Ut, d_b, Np, L, c.Pv, pw, d_y, nullptr, nullptr, nullptr, nullptr, 0};
    auto prepare = [&](int j) {
      const double* H = d_H + (int64_t)j * n64 * n64;
      hipLaunchKernelGGL(k_symv, dim3((n64 + 3) / 4), dim3(256), 0, st, H, d_z, n64, d_b);
      rg_launch_dgemm_nt(st, H, n64, d_Wt, n64, n64, (int)Np, n64, d_Ut, Np);
    };
    for (int j = 0; j < R1; ++j) {
      prepare(j);
      hipLaunchKernelGGL(k_loo_cv, dim3(gpos), dim3(256), 0, st, la, d_part);
      L1X_HIP(hipMemcpyAsync(hpart.data(), d_part, sizeof(double) * hpart.size(), hipMemcpyDeviceToHost, st));
      L1X_HIP(hipStreamSynchronize(st));
      double sx = 0, sx2 = 0, sxy = 0;
      for (unsigned b = 0; b < gpos; ++b) {
        sx += hpart[(size_t)b * LOO_NPART]; sx2 += hpart[(size_t)b * LOO_NPART + 2]; sxy += hpart[(size_t)b * LOO_NPART + 4];
      }
      cs[0 * R1 + j] = sx; cs[2 * R1 + j] = sx2; cs[4 * R1 + j] = sxy;
      cs[1 * R1 + j] = 0.0;                                  // Sy preset (Step1_Models.cpp:890)
      cs[3 * R1 + j] = ctx->neff[pg] - ctx->C;               // Sy2 = Neff - ncov (:891)
    }
    bool bad = false;
    if ((rc = check_spd(ctx, &bad))) return rc;
    if (bad) { ctx->err = "level 1 ridge system is not positive definite"; return RG_ERR_NOT_SPD; }
    int best = 0; double minv = 1e10;
    for (int j = 0; j < R1; ++j) {
      const double perf = (cs[2 * R1 + j] + cs[3 * R1 + j] - 2 * cs[4 * R1 + j]) / ctx->neff[pg];
      if (perf < minv) { best = j; minv = perf; }
    }
    best_out[p] = best;
    if (best != R1 - 1) prepare(best);
    L1X_HIP(hipMemsetAsync(c.d_pred, 0, sizeof(double) * (size_t)nchr * N, st));
    hipLaunchKernelGGL(k_loo_pred, dim3(gpos), dim3(256), 0, st, la, c.d_col0, nchr, ctx->d_cidx, N, c.d_pred);
    { const int rce = rg_emit_pred(ctx, st, c.d_pred, nchr, p, pred_out); if (rce) return rce; }
    L1X_HIP(hipStreamSynchronize(st));
  }
  return RG_OK;
}

// =========================================================================================================
// BT logistic ridge
// =========================================================================================================
namespace {

struct BtState {
  rg_ctx* ctx; L1Common* c; int p, nchain, nchunk;
  BtArgs a;
  double *d_beta, *d_score, *d_tauc, *d_part, *d_sys, *d_dinv;
  double* d_sw = nullptr;      // [nchain][Np] square roots of the weights (the quasi-Newton Gram of wgram_bf16.hip); null = fp64 Grams only
  // a chain's last quasi-Newton Hessian, kept for steps that do not form a new one (see "Steps on a Hessian that is already there"):
  double* d_G = nullptr;       // [nchain][msz] X^T W X + g_tau I as k_wg_reduce left it (right-hand-side rows zero)
  double* d_fac = nullptr;     // [nchain][msz] its Cholesky factor at fac_tau (lower triangle)
  double* d_delta = nullptr;   // [nchain] diagonal shifts of a re-factorization
  int32_t* d_map;
  std::vector<double> h_part, h_score, h_sol;
};

// state (wgt, z, r, sums) of every chain at the coefficients in hbeta [nchain][n64]
int bt_eval(BtState& s, const std::vector<double>& hbeta, std::vector<double>& sums /*[nchain][BT_NPART]*/) {
  rg_ctx* ctx = s.ctx;
  hipStream_t st = s.c->st;
  L1Lap lap(ctx, st, &ctx->tm.ms_irls_stream);
  L1X_HIP(hipMemcpyAsync(s.d_beta, hbeta.data(), sizeof(double) * hbeta.size(), hipMemcpyHostToDevice, st));
  for (int ch0 = 0; ch0 < s.nchain; ch0 += NCH) {
    hipLaunchKernelGGL(k_bt_eval, dim3(s.nchunk), dim3(256), 0, st, s.a, ch0, ctx->d_c256_seg, ctx->d_c256_pos,
                       ctx->d_c256_len, s.d_part);
    ++ctx->tm.n_irls_passes;
  }
  L1X_HIP(hipMemcpyAsync(s.h_part.data(), s.d_part, sizeof(double) * s.h_part.size(), hipMemcpyDeviceToHost, st));
  L1X_HIP(hipStreamSynchronize(st));
  sums.assign((size_t)s.nchain * BT_NPART, 0.0);
  for (int chk = 0; chk < s.nchunk; ++chk)
    for (int t = 0; t < s.nchain * BT_NPART; ++t) sums[t] += s.h_part[(size_t)chk * s.nchain * BT_NPART + t];
  return RG_OK;
}

// score of every chain at the state of the last bt_eval; returns max |score| per chain
int bt_score(BtState& s, const std::vector<double>& tauc, std::vector<double>& maxabs) {
  rg_ctx* ctx = s.ctx;
  hipStream_t st = s.c->st;
  L1Lap lap(ctx, st, &ctx->tm.ms_irls_stream);
  L1X_HIP(hipMemcpyAsync(s.d_tauc, tauc.data(), sizeof(double) * s.nchain, hipMemcpyHostToDevice, st));
  for (int ch0 = 0; ch0 < s.nchain; ch0 += NCH) {
    hipLaunchKernelGGL(k_bt_score, dim3((s.c->L + BT_SROWS - 1) / BT_SROWS), dim3(256), 0, st, s.a, ch0, s.d_tauc, s.d_score);
    ++ctx->tm.n_irls_passes;
  }
  L1X_HIP(hipMemcpyAsync(s.h_score.data(), s.d_score, sizeof(double) * s.h_score.size(), hipMemcpyDeviceToHost, st));
  L1X_HIP(hipStreamSynchronize(st));
  maxabs.assign(s.nchain, 0.0);
  for (int ch = 0; ch < s.nchain; ++ch)
    for (int k = 0; k < s.c->L; ++k) maxabs[ch] = std::max(maxabs[ch], std::fabs(s.h_score[(size_t)ch * s.c->n64 + k]));
  return RG_OK;
}

// ---- Steps on a Hessian that is already there ---------------------------------------------------------------------------------
// beta + H~^-1 score(beta) converges to the same fixed point for any H~ near the Hessian (wgram_bf16.hip), so H~ need not be formed at the
// current weights either: along the path of ridge values the fitted probabilities move by a few per cent from one value to the next, and
// X^T W X of a few steps ago is as good a quasi-Newton matrix as a freshly rounded one.  A chain therefore keeps its last Gram G (d_G) and
// its Cholesky factor (d_fac) and takes
//   a CHORD step      when the factor is at the current ridge value: two triangular solves (k_tri_solve), no Gram, no factorization;
//   a REFACTORED step when the ridge value has moved on: G + (tau - g_tau) I factored again, no Gram;
//   a FRESH step      (Gram at the current weights) when there is nothing to reuse, or when the last reused step shrank max |score| by
//                     less than RG_WGRAM_REUSE_RATIO (a factor 5), or -- after stepping back -- when it made it larger.
// The score, the stopping rule (max |score| < 1e-4) and the zero-weight halving are those of every other step.  RG_WGRAM_REUSE=0: fresh only.
__global__ void k_add_diag(double* mats, int64_t msz, int n64, int L, const double* delta) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < L) mats[(int64_t)blockIdx.y * msz + (int64_t)i * n64 + i] += delta[blockIdx.y];
}

// x = (F F^T)^-1 b for the factor F that chol.hip leaves in the lower triangle of a row-major n64 x n64 matrix (order L, the padding past it
// is not touched); one workgroup per system, the vector in LDS.  Forward substitution column tile by column tile (the diagonal tile's 64 pivots
// on one wave, then every row below takes its 64-term update), backward substitution the same way up the transposed factor (column sums over
// the rows below, four row groups reduced through LDS).  Each direction reads the 26 MB of the factor (L = 2,560) once; one workgroup is
// latency-bound on them (5.2 ms with 256 threads; a tile-row forward substitution measured slower), hence the 1,024 threads.
#define TS_NT 1024      // threads of a triangular solve: one workgroup per system is latency-bound, so it is as wide as a workgroup gets
__global__ __launch_bounds__(TS_NT) void k_tri_solve(const double* __restrict__ fac, int64_t msz, const int32_t* __restrict__ chainmap, int n64, int L,
                                                   const double* __restrict__ rhs, double* __restrict__ sol) {
  extern __shared__ double tsm[];
  double* y = tsm;                         // [n64]
  double* tile = tsm + n64;                // [64][65]
  double* red = tile + 64 * 65;            // [TS_NT / 64][64]
  const int tid = threadIdx.x;
  const int chain = chainmap[blockIdx.x];
  const double* A = fac + (int64_t)chain * msz;
  for (int i = tid; i < n64; i += TS_NT) y[i] = i < L ? rhs[(int64_t)chain * n64 + i] : 0.0;
  const int nt = (L + 63) / 64;
  auto stage_tile = [&](int k0) {          // lower triangle of the diagonal tile; identity past the order
    const int nk = L - k0 < 64 ? L - k0 : 64;
    for (int e = tid; e < 64 * 64; e += TS_NT) {
      const int r = e >> 6, c = e & 63;
      tile[r * 65 + c] = (r < nk && c <= r) ? A[(int64_t)(k0 + r) * n64 + k0 + c] : (r == c ? 1.0 : 0.0);
    }
  };
  __syncthreads();
  for (int k = 0; k < nt; ++k) {
    const int k0 = k * 64;
    stage_tile(k0);
    __syncthreads();
    if (tid < 64) {
      double v = y[k0 + tid];
      for (int c = 0; c < 64; ++c) {
        const double piv = __shfl(v, c) / tile[c * 65 + c];
        if (tid == c) v = piv;
        else if (tid > c) v -= tile[tid * 65 + c] * piv;
      }
      y[k0 + tid] = v;
    }
    __syncthreads();
    for (int i = k0 + 64 + tid; i < L; i += TS_NT) {
      const double* row = A + (int64_t)i * n64 + k0;
      double a0 = 0.0, a1 = 0.0;
#pragma unroll 8
      for (int c = 0; c < 64; c += 2) {
        const double2 t = *reinterpret_cast<const double2*>(row + c);
        a0 = fma(t.x, y[k0 + c], a0);
        a1 = fma(t.y, y[k0 + c + 1], a1);
      }
      y[i] -= a0 + a1;
    }
    __syncthreads();
  }
  const int cc = tid & 63, gg = tid >> 6;
  for (int k = nt - 1; k >= 0; --k) {
    const int k0 = k * 64;
    double acc = 0.0;
    for (int i = k0 + 64 + gg; i < L; i += TS_NT / 64) acc = fma(A[(int64_t)i * n64 + k0 + cc], y[i], acc);
    red[gg * 64 + cc] = acc;
    stage_tile(k0);
    __syncthreads();
    if (tid < 64) {
      double rs = 0.0;
      for (int g = 0; g < TS_NT / 64; ++g) rs += red[g * 64 + tid];
      double v = y[k0 + tid] - rs;
      for (int c = 63; c >= 0; --c) {
        const double piv = __shfl(v, c) / tile[c * 65 + c];
        if (tid == c) v = piv;
        else if (tid < c) v -= tile[c * 65 + tid] * piv;
      }
      y[k0 + tid] = v;
    }
    __syncthreads();
  }
  for (int i = tid; i < n64; i += TS_NT) sol[(int64_t)chain * n64 + i] = i < L ? y[i] : 0.0;
}

// ---- the same two substitutions for FEW large systems, one LAUNCH per tile row (round 6) ----------------------------------------------------
// k_tri_solve walks a system with ONE workgroup: 2 x 40 dependent tile steps at L = 2,560, each a pass of that workgroup over up to 1.3 MB of
// the factor -- 5.2 ms per call whatever the number of chains (<= 20 workgroups on 256 CUs), 20 % of configs[3]'s level 1
// (profiles/r4_config4_level1_kernel_stats_stored_hessians.md).  Here a step is a launch and every tile row (column) below (left of) the current
// one is a workgroup: forward, launch s: y_i -= F[i][s-1] y_(s-1) for every i >= s, and the workgroup of row s then finishes y_s = F_ss^-1 y_s;
// backward, launch k: y_j -= F[k][j]^T x_k for every j < k, and the workgroup of column k-1 finishes x_(k-1).  2 T launches of a few
// microseconds each (the scheme of k_chol_backsolve_row, chol.hip, which needs the tile inverses a stored factor does not come with).
// grid (rows of this step, systems), 256 threads; sol [chain][n64] holds b, then y, then x.
__global__ __launch_bounds__(256) void k_tri_rows(const double* __restrict__ fac, int64_t msz, const int32_t* __restrict__ chainmap, int n64, int L,
                                                const double* __restrict__ rhs, double* __restrict__ sol, int step, int backward) {
  __shared__ double tile[64][65];
  __shared__ double yv[64];
  __shared__ double red[4][64];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int chain = chainmap[blockIdx.y];
  const double* A = fac + (int64_t)chain * msz;
  double* y = sol + (int64_t)chain * n64;
  const int T = (L + 63) / 64;
  auto stage_diag = [&](int k0) {          // lower triangle of the diagonal tile; identity past the order
    const int nk = L - k0 < 64 ? L - k0 : 64;
    for (int e = tid; e < 64 * 64; e += 256) {
      const int r = e >> 6, c = e & 63;
      tile[r][c] = (r < nk && c <= r) ? A[(int64_t)(k0 + r) * n64 + k0 + c] : (r == c ? 1.0 : 0.0);
    }
  };
  if (!backward) {
    if (step == 0) {                       // b -> sol (the padding past the order: 0), then y_0
      for (int i = tid; i < n64; i += 256) y[i] = i < L ? rhs[(int64_t)chain * n64 + i] : 0.0;
      stage_diag(0);
      __syncthreads();
      if (tid < 64) {
        double v = y[tid];
        const double rd = 1.0 / tile[tid][tid];      // the 64 reciprocals at once: no division inside the chain of 64 dependent steps
        for (int c = 0; c < 64; ++c) {
          const double piv = __shfl(v * rd, c);
          if (tid == c) v = piv;
          else if (tid > c) v -= tile[tid][c] * piv;
        }
        y[tid] = v;
      }
      return;
    }
    const int i = step + blockIdx.x, k0 = (step - 1) * 64, i0 = i * 64;      // y_i -= F[i][step-1] y_(step-1)
    if (i >= T) return;
    if (tid < 64) yv[tid] = y[k0 + tid];
    __syncthreads();
    // thread t: row t >> 2 of the tile, columns 16 (t & 3) .. + 15 -- sixteen independent loads (four adjacent lanes cover the row's 512
    // bytes), then two shuffles (a load per row and six shuffles behind it, row after row, took 24 us per launch: one memory latency per row)
    {
      const int r = tid >> 2, q4 = tid & 3, row = i0 + r;
      const double* ap = A + (int64_t)(row < L ? row : 0) * n64 + k0 + 16 * q4;
      double4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const double4*>(ap + 4 * u);
      double p = 0.0;
#pragma unroll
      for (int u = 0; u < 4; ++u)
        p += (v[u].x * yv[16 * q4 + 4 * u] + v[u].y * yv[16 * q4 + 4 * u + 1]) + (v[u].z * yv[16 * q4 + 4 * u + 2] + v[u].w * yv[16 * q4 + 4 * u + 3]);
      p += __shfl_xor(p, 1);
      p += __shfl_xor(p, 2);
      if (q4 == 0 && row < L) y[row] -= p;
    }
    if (blockIdx.x != 0) return;
    __threadfence_block();
    stage_diag(i0);                        // the row of this step: y_s = F_ss^-1 y_s
    __syncthreads();
    if (tid < 64) {
      double v = y[i0 + tid];
      const double rd = 1.0 / tile[tid][tid];
      for (int c = 0; c < 64; ++c) {
        const double piv = __shfl(v * rd, c);
        if (tid == c) v = piv;
        else if (tid > c) v -= tile[tid][c] * piv;
      }
      y[i0 + tid] = v;
    }
    return;
  }
  // backward: step = k (T-1 .. 1: the updates with x_k), or T for the first solve x_(T-1)
  auto solve_t = [&](int j0) {             // x = F_jj^-T y on the tile staged in `tile`
    if (tid < 64) {
      double v = y[j0 + tid];
      const double rd = 1.0 / tile[tid][tid];
      for (int c = 63; c >= 0; --c) {
        const double piv = __shfl(v * rd, c);
        if (tid == c) v = piv;
        else if (tid < c) v -= tile[c][tid] * piv;
      }
      y[j0 + tid] = v;
    }
  };
  if (step == T) {
    stage_diag((T - 1) * 64);
    __syncthreads();
    solve_t((T - 1) * 64);
    return;
  }
  const int k0 = step * 64, j = blockIdx.x, j0 = j * 64;                  // y_j -= F[k][j]^T x_k
  if (tid < 64) yv[tid] = y[k0 + tid];
  __syncthreads();
  double a = 0.0;
  {
    double l[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {          // sixteen loads in flight (rows past the order: clamped address, zero weight)
      const int row = k0 + 16 * w + r;
      l[r] = A[(int64_t)(row < L ? row : L - 1) * n64 + j0 + lane];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) a = fma(l[r], (k0 + 16 * w + r < L) ? yv[16 * w + r] : 0.0, a);
  }
  red[w][lane] = a;
  __syncthreads();
  if (w == 0) y[j0 + lane] -= (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
  if (j != step - 1) return;
  __threadfence_block();
  stage_diag(j0);
  __syncthreads();
  solve_t(j0);
}

static void launch_tri_rows(hipStream_t st, const double* fac, int64_t msz, const int32_t* d_map, int na, int n64, int L, const double* rhs, double* sol) {
  const int T = (L + 63) / 64;
  for (int s = 0; s < T; ++s)
    hipLaunchKernelGGL(k_tri_rows, dim3(s == 0 ? 1 : T - s, na), dim3(256), 0, st, fac, msz, d_map, n64, L, rhs, sol, s, 0);
  hipLaunchKernelGGL(k_tri_rows, dim3(1, na), dim3(256), 0, st, fac, msz, d_map, n64, L, rhs, sol, T, 1);
  for (int k = T - 1; k >= 1; --k)
    hipLaunchKernelGGL(k_tri_rows, dim3(k, na), dim3(256), 0, st, fac, msz, d_map, n64, L, rhs, sol, k, 1);
}

// what a workgroup may ask for in dynamic LDS on this device (160 KB on gfx950; asked once)
static size_t lds_optin_bytes() {
  static const size_t v = []() -> size_t {
    int dev = 0, b = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&b, hipDeviceAttributeSharedMemPerBlockOptin, dev) != hipSuccess || b <= 0) return 64 * 1024;
    return (size_t)b;
  }();
  return v;
}
static size_t tri_solve_lds(int n64) { return sizeof(double) * ((size_t)n64 + 64 * 65 + TS_NT); }

// chord steps of the chains in `act`: solutions of F F^T x = score with each chain's stored factor, into h_sol
int bt_chord(BtState& s, const std::vector<int32_t>& act) {
  rg_ctx* ctx = s.ctx;
  L1Common& c = *s.c;
  hipStream_t st = c.st;
  const int na = (int)act.size();
  L1Lap lap(ctx, st, &ctx->tm.ms_irls_solve);
  L1X_HIP(hipMemcpyAsync(s.d_map, act.data(), sizeof(int32_t) * na, hipMemcpyHostToDevice, st));
  const size_t lds = tri_solve_lds(c.n64);
  if (lds > 48 * 1024) L1X_HIP(hipFuncSetAttribute((const void*)k_tri_solve, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  double* d_x = s.d_sys;      // [nchain][n64] scratch: the systems' workspace is idle during a chord step
  static const bool one_wg = getenv("RG_TRI_ONE_WG") && atoi(getenv("RG_TRI_ONE_WG")) != 0;      // the round-4 kernel: one workgroup per system
  if (one_wg || c.L < 256)
    hipLaunchKernelGGL(k_tri_solve, dim3(na), dim3(TS_NT), lds, st, (const double*)s.d_fac, c.msz, (const int32_t*)s.d_map, c.n64, c.L, (const double*)s.d_score, d_x);
  else
    launch_tri_rows(st, s.d_fac, c.msz, s.d_map, na, c.n64, c.L, s.d_score, d_x);
  L1X_HIP(hipGetLastError());   // 62 KB of dynamic LDS at L = 2,560: a refused launch must not pass scratch off as the chord step
  for (int i = 0; i < na; ++i)
    L1X_HIP(hipMemcpyAsync(s.h_sol.data() + (size_t)act[i] * c.n64, d_x + (int64_t)act[i] * c.n64, sizeof(double) * c.n64, hipMemcpyDeviceToHost, st));
  L1X_HIP(hipStreamSynchronize(st));
  ++ctx->tm.n_irls_rounds;
  return RG_OK;
}

// (X^T W X + tau I) x = rhs for the chains in `act` (rhs = X^T W z from the extra row, or the score);
// solutions land in h_sol [nchain][n64].  *bad is set when a system is not positive definite.
// approx: H~ from the 16-bit operand planes (wgram_bf16.hip) instead of the fp64 Gram -- only with rhs_is_score, where the right-hand side is
// the exact score and the solution a quasi-Newton step (a fixed point of beta + H~^-1 score(beta) has score = 0 whatever H~ is)
// keep: the quasi-Newton Gram and its factor are stored per chain (d_G, d_fac).  g_tau != nullptr: NO Gram is formed -- the stored one is
// shifted from the ridge value it holds (g_tau[chain]) to tauc[chain] and factored again.
int bt_solve(BtState& s, const std::vector<int32_t>& act, const std::vector<double>& tauc, bool rhs_is_score,
             bool* bad, bool approx = false, bool keep = false, const std::vector<double>* g_tau = nullptr) {
  rg_ctx* ctx = s.ctx;
  L1Common& c = *s.c;
  hipStream_t st = c.st;
  const int na = (int)act.size();
  L1X_HIP(hipMemcpyAsync(s.d_map, act.data(), sizeof(int32_t) * na, hipMemcpyHostToDevice, st));
  L1X_HIP(hipMemcpyAsync(s.d_tauc, tauc.data(), sizeof(double) * s.nchain, hipMemcpyHostToDevice, st));
  WgArgs g{c.Wv, ctx->d_zero, c.Np, c.L, c.Pv, s.p, c.n64, s.a.wv, rhs_is_score ? nullptr : s.a.zv, s.d_tauc,
           s.d_map, s.a.kfold, s.d_sys, c.msz};
  if (g_tau && s.d_G) {
    L1Lap lap(ctx, st, &ctx->tm.ms_irls_solve);
    std::vector<double> delta(na);
    for (int i = 0; i < na; ++i) {
      delta[i] = tauc[act[i]] - (*g_tau)[act[i]];
      L1X_HIP(hipMemcpyAsync(s.d_sys + (int64_t)i * c.msz, s.d_G + (int64_t)act[i] * c.msz, sizeof(double) * c.msz, hipMemcpyDeviceToDevice, st));
    }
    L1X_HIP(hipMemcpyAsync(s.d_delta, delta.data(), sizeof(double) * na, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_add_diag, dim3((c.L + 255) / 256, na), dim3(256), 0, st, s.d_sys, c.msz, c.n64, c.L, (const double*)s.d_delta);
    L1X_HIP(hipGetLastError());
  } else if (approx && rhs_is_score && s.d_sw) {
    L1Lap lap(ctx, st, &ctx->tm.ms_wgram);
    const int MAXSL = 16;
    double* d_part = (double*)rg_ws(ctx, 14, sizeof(double) * (size_t)MAXSL * na * c.msz);
    if (!d_part) { ctx->err = "weighted Gram: out of device memory"; return RG_ERR_HIP; }
    L1X_HIP(hipStreamSynchronize(st));      // `act` (host) must have reached d_map before the table is built from it
    const int ns = rg_launch_wgram_bf16(ctx, st, c.Wv, c.Np, c.L, c.Pv, s.p, c.n64, s.a.wv, s.d_sw, s.nchain, s.d_map, act.data(), na, s.a.kfold,
                                        d_part, c.msz, MAXSL);
    if (ns <= 0) { ctx->err = "rg_l1_bt: the quasi-Newton Gram (k_wgram_mx) could not be launched"; return RG_ERR_HIP; }
    Wg128 g2{g, ns, na, nullptr, d_part};
    hipLaunchKernelGGL(k_wg_reduce, dim3(c.T * (c.T + 1) / 2, na), dim3(256), 0, st, g2, c.T);
    L1X_HIP(hipMemset2DAsync(s.d_sys + (int64_t)c.n64 * c.n64, sizeof(double) * c.msz, 0, sizeof(double) * CT * c.n64, na, st));
    if (keep && s.d_G)
      for (int i = 0; i < na; ++i)
        L1X_HIP(hipMemcpyAsync(s.d_G + (int64_t)act[i] * c.msz, s.d_sys + (int64_t)i * c.msz, sizeof(double) * c.msz, hipMemcpyDeviceToDevice, st));
    ++ctx->tm.n_wgram_approx_rounds;
  } else {
    L1Lap lap(ctx, st, &ctx->tm.ms_wgram);
    const int rcw = launch_wgram(ctx, st, g, c.T, na); if (rcw) return rcw;
  }
  ++ctx->tm.n_irls_rounds;
  if (!(g_tau && s.d_G)) {
    ctx->tm.n_wgram += na;
    const int64_t all = ctx->seg.pos_start[ctx->seg.nseg - 1] + ctx->seg.plen[ctx->seg.nseg - 1];
    for (int i = 0; i < na; ++i) ctx->tm.wgram_positions += all - (s.a.kfold ? ctx->seg.plen[act[i]] : 0);
  }
  L1Lap lap2(ctx, st, &ctx->tm.ms_irls_solve);
  if (rhs_is_score)
    for (int i = 0; i < na; ++i)
      L1X_HIP(hipMemcpyAsync(s.d_sys + (int64_t)i * c.msz + (int64_t)c.n64 * c.n64, s.d_score + (int64_t)act[i] * c.n64,
                             sizeof(double) * c.L, hipMemcpyDeviceToDevice, st));
  rg_launch_chol_solve(st, s.d_sys, c.msz, na, c.n64, CT, 1, s.d_dinv, ctx->d_info + 1, &ctx->tm.n_chol_launches);
  for (int i = 0; i < na; ++i) {
    L1X_HIP(hipMemcpyAsync(s.h_sol.data() + (size_t)act[i] * c.n64, s.d_sys + (int64_t)i * c.msz + (int64_t)c.n64 * c.n64,
                           sizeof(double) * c.n64, hipMemcpyDeviceToHost, st));
    if ((keep || g_tau) && s.d_fac)      // the factor, for chord steps at this ridge value (the square part is enough)
      L1X_HIP(hipMemcpyAsync(s.d_fac + (int64_t)act[i] * c.msz, s.d_sys + (int64_t)i * c.msz, sizeof(double) * (size_t)c.n64 * c.n64, hipMemcpyDeviceToDevice, st));
  }
  L1X_HIP(hipStreamSynchronize(st));
  return check_spd(ctx, bad);
}

// run_log_ridge_loocv (Step1_Models.cpp:1288-1374) for the single all-sample chain; beta in/out (n64 padded).
// On success the device state (wgt, r) is the state at the returned beta.
int bt_newton_loocv(BtState& s, double lam, std::vector<double>& beta, const rg_bt_options& o, bool* ok) {
  const int L = s.c->L;
  *ok = false;
  std::vector<double> sums, maxabs, betanew = beta, step(beta.size(), 0.0);
  std::vector<double> tauc(1, lam);
  std::vector<int32_t> act(1, 0);
  auto pen = [&](const std::vector<double>& b) { double t = 0; for (int k = 0; k < L; ++k) t += b[k] * b[k]; return lam * t; };
  int rc = bt_eval(s, beta, sums);
  if (rc) return rc;
  double fn_start = sums[6] + pen(beta), fn_end = fn_start;
  if (sums[7] != 0.0) return RG_OK;
  if ((rc = bt_score(s, tauc, maxabs))) return rc;
  int niter = 0;
  bool dev_conv = false;
  while (true) {
    if (++niter > o.niter_max_ridge) break;
    bool bad = false;
    // Newton step, or the quasi-Newton one at large N -- for at most RG_WGRAM_SWITCH (12) rounds, as in the K-fold chains: a fit the
    // approximate Hessian has not brought below the tolerance by then finishes on the exact one instead of being reported as not converged
    static const int qn_rounds = getenv("RG_WGRAM_SWITCH") ? atoi(getenv("RG_WGRAM_SWITCH")) : 12;
    if ((rc = bt_solve(s, act, tauc, true, &bad, s.d_sw != nullptr && niter <= qn_rounds))) return rc;
    if (bad) return RG_OK;
    for (int k = 0; k < L; ++k) step[k] = s.h_sol[k];
    for (int ls = 0; ls < o.niter_max_line_search; ++ls) {
      for (int k = 0; k < L; ++k) betanew[k] = beta[k] + step[k];
      if ((rc = bt_eval(s, betanew, sums))) return rc;
      fn_end = sums[6] + pen(betanew);
      if (sums[7] != 0.0) return RG_OK;
      if (fn_end < fn_start + 1e-6) break;   // numtol
      for (int k = 0; k < L; ++k) step[k] *= 0.5;
    }
    if ((rc = bt_score(s, tauc, maxabs))) return rc;
    dev_conv = std::fabs(fn_end - fn_start) / (0.01 + std::fabs(fn_end)) < o.tol;
    if (maxabs[0] < o.l1_ridge_tol) break;
    beta = betanew;
    fn_start = fn_end;
  }
  if (!dev_conv && niter > o.niter_max_ridge) return RG_OK;
  beta = betanew;
  *ok = true;
  return RG_OK;
}

// run_ct_ridge_loocv (Step1_Models.cpp:1694-1758): plain Newton on the penalised Poisson likelihood, no line search;
// beta in/out.  On success the device state (wgt = mean, r) is the state at the returned beta.
int ct_newton_loocv(BtState& s, double lam, std::vector<double>& beta, const rg_bt_options& o, bool* ok) {
  const int L = s.c->L;
  *ok = false;
  std::vector<double> sums, maxabs, betaold = beta, betanew = beta;
  std::vector<double> tauc(1, lam);
  std::vector<int32_t> act(1, 0);
  int rc, niter = 0;
  while (true) {
    if (++niter > o.niter_max_ridge) break;
    if ((rc = bt_eval(s, betaold, sums))) return rc;
    if (sums[7] != 0.0) return RG_OK;            // a zero mean among the analysed samples
    bool bad = false;
    if ((rc = bt_solve(s, act, tauc, false, &bad))) return rc;
    if (bad) return RG_OK;
    for (int k = 0; k < L; ++k) betanew[k] = s.h_sol[k];
    if ((rc = bt_eval(s, betanew, sums))) return rc;
    if (sums[7] != 0.0) return RG_OK;
    if ((rc = bt_score(s, tauc, maxabs))) return rc;
    if (maxabs[0] < o.l1_ridge_tol) break;
    betaold = betanew;
  }
  if (niter > o.niter_max_ridge) return RG_OK;
  beta = betanew;
  *ok = true;
  return RG_OK;
}

}  // namespace

int rg_l1_bt_impl(rg_ctx* ctx, int R1, const double* tau, const double* yraw, const double* offset,
                  const rg_bt_options* opt, int nchr, const int32_t* cols_per_chr, double* cumsum_out,
                  int32_t* converged_out, int32_t* best_out, double* pred_out) {
  if (R1 < 1 || R1 > 16) { ctx->err = "rg_l1_bt: n_ridge_l1 must be in [1,16]"; return RG_ERR_ARG; }
  rg_bt_options o;
  o.niter_max_ridge = 100; o.niter_max_line_search_ridge = 100; o.niter_max_line_search = 25;
  o.l1_ridge_tol = 1e-4; o.tol = 1e-8; o.family = 0; o.beta_out = nullptr; o.fold_cumsum_out = nullptr;
  if (opt) o = *opt;
  if (o.family != 0 && o.family != 1) { ctx->err = "rg_l1_bt: family must be 0 (logistic) or 1 (Poisson)"; return RG_ERR_ARG; }
  const bool poisson = o.family == 1;
  L1Common c;
  int rc = l1_common_init(ctx, c, nchr, cols_per_chr, "rg_l1_bt");
  if (rc) return rc;
  hipStream_t st = c.st;
  const int L = c.L, P = c.P, n64 = c.n64, T = c.T;
  const int64_t Np = c.Np, N = c.N;
  const bool loocv = ctx->loocv;
  const int K = ctx->K;
  const int nchain = loocv ? 1 : K;
  const unsigned gpos = (unsigned)((Np + 255) / 256);

  BtState s;
  s.ctx = ctx; s.c = &c; s.nchain = nchain; s.nchunk = ctx->n_c256;
  double *d_yraw, *d_off, *d_wv, *d_zv, *d_rv, *d_betas = nullptr;
  L1X_HIP(c.bufs.alloc(&d_yraw, (size_t)Np));
  L1X_HIP(c.bufs.alloc(&d_off, (size_t)Np));
  L1X_HIP(c.bufs.alloc(&d_wv, (size_t)nchain * Np));
  L1X_HIP(c.bufs.alloc(&d_zv, (size_t)nchain * Np));
  L1X_HIP(c.bufs.alloc(&d_rv, (size_t)nchain * Np));
  L1X_HIP(c.bufs.alloc(&s.d_beta, (size_t)nchain * n64));
  L1X_HIP(c.bufs.alloc(&s.d_score, (size_t)nchain * n64));
  L1X_HIP(c.bufs.alloc(&s.d_tauc, (size_t)nchain));
  L1X_HIP(c.bufs.alloc(&s.d_part, (size_t)s.nchunk * nchain * BT_NPART));
  L1X_HIP(c.bufs.alloc(&s.d_sys, (size_t)nchain * c.msz));
  L1X_HIP(c.bufs.alloc(&s.d_dinv, rg_chol_ws_doubles((size_t)nchain, c.n64)));
  L1X_HIP(c.bufs.alloc(&s.d_map, (size_t)nchain));
  // K-fold: the weighted Grams of the IRLS steps as quasi-Newton Hessians on the 16-bit matrix cores (wgram_bf16.hip) unless RG_WGRAM_F64=1;
  // a chain that has not converged after RG_WGRAM_SWITCH steps at one ridge value continues on the fp64 Gram
  // -- where the fp64 Gram costs more than a few milliseconds (RG_WGRAM_QUASI_MIN: flop of one chain Gram from which on the quasi-Newton
  // Gram is used, default 2e11 = 4 ms of the fp64 kernel; 0 = always).  Below that the fp64 Gram is free and the iterates are those of
  // the reference's Newton steps digit for digit (small problems are what the reference's own output files pin to the last printed digit;
  // with the quasi-Newton Gram the last iterate differs from the fp64 one by rho x the last step, up to 1e-5 relative on few samples).
  const bool wg_f64 = getenv("RG_WGRAM_F64") && atoi(getenv("RG_WGRAM_F64")) != 0;
  const int wg_switch = getenv("RG_WGRAM_SWITCH") ? atoi(getenv("RG_WGRAM_SWITCH")) : 12;
  const double wg_min = getenv("RG_WGRAM_QUASI_MIN") ? atof(getenv("RG_WGRAM_QUASI_MIN")) : 2e11;
  const double gram_flop = (double)(ctx->seg.pos_start[ctx->seg.nseg - 1] + ctx->seg.plen[ctx->seg.nseg - 1]) * (loocv ? 1.0 : (double)(K - 1) / K) * L * (L + 1.0);
  // (round 5: the leave-one-out Newton iteration of the logistic ridge takes the same quasi-Newton Hessian -- its line search and its stopping
  // rule use the exact fp64 deviance and score; the leave-one-out shortcut afterwards forms the exact fp64 Hessian at the converged weights)
  if (!wg_f64 && gram_flop >= wg_min && !(loocv && poisson)) L1X_HIP(c.bufs.alloc(&s.d_sw, (size_t)nchain * Np));
  // steps on a stored Hessian (chord / refactored, see k_tri_solve): on with the quasi-Newton Gram unless RG_WGRAM_REUSE=0
  const bool reuse = s.d_sw && !loocv && !(getenv("RG_WGRAM_REUSE") && atoi(getenv("RG_WGRAM_REUSE")) == 0) && tri_solve_lds(c.n64) <= lds_optin_bytes();
  const double reuse_ratio = getenv("RG_WGRAM_REUSE_RATIO") ? atof(getenv("RG_WGRAM_REUSE_RATIO")) : 0.2;
  const double reuse_tol = getenv("RG_WGRAM_REUSE_TOL") ? atof(getenv("RG_WGRAM_REUSE_TOL")) : 1e-6;
  if (reuse) {
    L1X_HIP(c.bufs.alloc(&s.d_G, (size_t)nchain * c.msz));
    L1X_HIP(c.bufs.alloc(&s.d_fac, (size_t)nchain * c.msz));
    L1X_HIP(c.bufs.alloc(&s.d_delta, (size_t)nchain));
  }
  L1X_HIP(hipMemsetAsync(s.d_score, 0, sizeof(double) * (size_t)nchain * n64, st));
  s.h_part.resize((size_t)s.nchunk * nchain * BT_NPART);
  s.h_score.resize((size_t)nchain * n64);
  s.h_sol.assign((size_t)nchain * n64, 0.0);
  // LOOCV extras: identity rows, inverse, sample-major W and U^T = H W^T
  double *d_eye = nullptr, *d_sysI = nullptr, *d_H = nullptr, *d_Wt = nullptr, *d_Ut = nullptr, *d_G = nullptr,
         *d_lpart = nullptr, *d_tau1 = nullptr, *d_dinvI = nullptr;
  if (loocv) {
    L1X_HIP(c.bufs.alloc(&d_eye, (size_t)n64 * n64));
    L1X_HIP(c.bufs.alloc(&d_sysI, (size_t)2 * n64 * n64));
    L1X_HIP(c.bufs.alloc(&d_dinvI, rg_chol_ws_doubles(1, c.n64)));
    L1X_HIP(c.bufs.alloc(&d_H, (size_t)n64 * n64));
    L1X_HIP(c.bufs.alloc(&d_G, (size_t)c.msz));
    L1X_HIP(c.bufs.alloc(&d_Wt, (size_t)Np * n64));
    L1X_HIP(c.bufs.alloc(&d_Ut, (size_t)Np * n64));
    L1X_HIP(c.bufs.alloc(&d_lpart, (size_t)gpos * LOO_NPART));
    L1X_HIP(c.bufs.alloc(&d_tau1, 1));
    hipLaunchKernelGGL(k_eye, dim3((unsigned)(((int64_t)n64 * n64 + 255) / 256)), dim3(256), 0, st, d_eye, n64);
  } else {
    L1X_HIP(c.bufs.alloc(&d_betas, (size_t)K * R1 * n64));
  }
  std::vector<double> hpos, hl((size_t)gpos * LOO_NPART);

  for (int p = 0; p < P; ++p) {
    double* cs = cumsum_out + (int64_t)p * 6 * R1;
    for (int t = 0; t < 6 * R1; ++t) cs[t] = 0.0;
    converged_out[p] = 0;
    best_out[p] = 0;
    const double* taup = tau + (int64_t)p * R1;
    to_pos(ctx, yraw + (int64_t)p * N, hpos);
    L1X_HIP(hipMemcpyAsync(d_yraw, hpos.data(), sizeof(double) * Np, hipMemcpyHostToDevice, st));
    L1X_HIP(hipStreamSynchronize(st));
    to_pos(ctx, offset + (int64_t)p * N, hpos);
    L1X_HIP(hipMemcpyAsync(d_off, hpos.data(), sizeof(double) * Np, hipMemcpyHostToDevice, st));
    L1X_HIP(hipStreamSynchronize(st));
    const int pg = c.p0v + p, pw = c.view ? p : pg;   // global phenotype / index inside the predictor buffer
    s.p = pw;
    s.a = BtArgs{c.Wv, Np, L, c.Pv, pw, n64, d_yraw, d_off, ctx->d_maskp + (int64_t)pg * Np, s.d_beta, nchain,
                 loocv ? 0 : 1, d_wv, d_zv, d_rv, o.family};
    bool ok = true;

    if (!loocv) {
      // ---- K-fold: the K fold models advance in lockstep launches, each with its own (tau index, iteration) ----
      std::vector<double> beta((size_t)K * n64, 0.0), betaold = beta, sums, maxabs, tauc(K), hbetas((size_t)K * R1 * n64, 0.0);
      std::vector<int> jj(K, 0), niter(K, 0), solved(K, 0), halv(K, 0), nfresh(K, 0);
      // stored-Hessian state per chain: has a Gram (at ridge value g_tau), has a factor (at fac_tau), the last step reused one, the next must not
      std::vector<char> have_g(K, 0), have_f(K, 0), last_reused(K, 0), need_fresh(K, 0);
      std::vector<double> g_tau(K, 0.0), fac_tau(K, 0.0), prev_abs(K, 0.0);
      int ndone = 0;
      std::vector<char> done(K, 0);
      while (ok && ndone < K) {
        if ((rc = bt_eval(s, beta, sums))) return rc;
        for (int ch = 0; ch < K; ++ch) tauc[ch] = taup[std::min(jj[ch], R1 - 1)];
        bool any_solved = false;
        for (int ch = 0; ch < K; ++ch) any_solved |= (!done[ch] && solved[ch]);
        if (any_solved && (rc = bt_score(s, tauc, maxabs))) return rc;
        std::vector<int32_t> act;
        for (int ch = 0; ch < K && ok; ++ch) {
          if (done[ch]) continue;
          const double* sm = sums.data() + (size_t)ch * BT_NPART;
          if (solved[ch]) {
            if (sm[7] != 0.0) {  // zero weights: halve towards the previous iterate (Step1_Models.cpp:1066-1079)
              if (++halv[ch] > o.niter_max_line_search_ridge) { ok = false; break; }
              for (int k = 0; k < L; ++k)
                beta[(size_t)ch * n64 + k] = 0.5 * (betaold[(size_t)ch * n64 + k] + beta[(size_t)ch * n64 + k]);
              continue;  // re-evaluate, no new solve
            }
            halv[ch] = 0;
            // A Newton step lands far inside the tolerance it is stopped by (the reference's last step starts from max |score| > 1e-4 and
            // ends near 1e-8); steps on a stored Hessian converge linearly and would stop just under it, up to 1e-4 / lambda_min further from
            // the optimum than the reference's iterate.  They are therefore held to a tighter bound (RG_WGRAM_REUSE_TOL, 1e-6).
            const double tol_ch = last_reused[ch] ? std::min(reuse_tol, o.l1_ridge_tol) : o.l1_ridge_tol;
            if (last_reused[ch]) {     // how did the step on the stored Hessian do?
              last_reused[ch] = 0;
              if (maxabs[ch] > prev_abs[ch] && maxabs[ch] >= tol_ch) {      // worse: step back, form the Gram here
                std::memcpy(beta.data() + (size_t)ch * n64, betaold.data() + (size_t)ch * n64, sizeof(double) * n64);
                need_fresh[ch] = 1;
                solved[ch] = 0;
                continue;  // re-evaluate at the previous iterate, no new solve this round
              }
              if (maxabs[ch] > reuse_ratio * prev_abs[ch]) need_fresh[ch] = 1;
            }
            if (maxabs[ch] < tol_ch) {  // converged at tau_j: record beta + held-out sums
              const int j = jj[ch];
              std::memcpy(hbetas.data() + ((size_t)ch * R1 + j) * n64, beta.data() + (size_t)ch * n64, sizeof(double) * n64);
              for (int t = 0; t < 6; ++t) cs[t * R1 + j] += sm[t];
              if (o.fold_cumsum_out)
                for (int t = 0; t < 6; ++t) o.fold_cumsum_out[(((size_t)p * K + ch) * 6 + t) * R1 + j] = sm[t];
              if (++jj[ch] == R1) { done[ch] = 1; ++ndone; continue; }
              niter[ch] = 0; nfresh[ch] = 0;
            }
          } else if (sm[7] != 0.0) { ok = false; break; }
          if (++niter[ch] > o.niter_max_ridge) { ok = false; break; }
          std::memcpy(betaold.data() + (size_t)ch * n64, beta.data() + (size_t)ch * n64, sizeof(double) * n64);
          act.push_back(ch);
        }
        if (!ok) break;
        if (act.empty()) continue;
        // The IRLS step (Step1_Models.cpp:1059-1064) in its Newton form: beta <- beta + (X^T W X + tau I)^-1 (X^T (y - p) - tau beta),
        // the same vector as (X^T W X + tau I)^-1 X^T W z.  The score is exact (k_bt_score); the matrix may then be the quasi-Newton
        // Gram of wgram_bf16.hip.  The score has to be the one at the chain's CURRENT ridge value: a chain that has just moved on to its
        // next value (or has not been scored yet) gets it recomputed.
        bool rescore = !any_solved;
        for (int ch : act) rescore |= (tauc[ch] != taup[std::min(jj[ch], R1 - 1)]);
        for (int ch = 0; ch < K; ++ch) tauc[ch] = taup[std::min(jj[ch], R1 - 1)];
        if (rescore && (rc = bt_score(s, tauc, maxabs))) return rc;
        // quasi-Newton Gram (fresh) / fp64 Gram (a chain that is slow at this ridge value finishes on it) / stored Hessian: chord, refactored
        std::vector<int32_t> act_q, act_x, act_c, act_r;
        for (int ch : act) {
          prev_abs[ch] = maxabs[ch];
          if (!(s.d_sw && nfresh[ch] <= wg_switch)) act_x.push_back(ch);
          else if (reuse && have_g[ch] && !need_fresh[ch]) ((have_f[ch] && fac_tau[ch] == tauc[ch]) ? act_c : act_r).push_back(ch);
          else act_q.push_back(ch);
        }
        for (int pass = 0; pass < 4 && ok; ++pass) {
          const std::vector<int32_t>& aa = pass == 0 ? act_q : pass == 1 ? act_x : pass == 2 ? act_r : act_c;
          if (aa.empty()) continue;
          bool bad = false;
          if (pass == 3) rc = bt_chord(s, aa);
          else rc = bt_solve(s, aa, tauc, true, &bad, pass == 0, pass == 0 && reuse, pass == 2 ? &g_tau : nullptr);
          if (rc) return rc;
          if (bad) { ok = false; break; }
          for (int ch : aa) {
            for (int k = 0; k < L; ++k) beta[(size_t)ch * n64 + k] = betaold[(size_t)ch * n64 + k] + s.h_sol[(size_t)ch * n64 + k];
            solved[ch] = 1;
            if (pass == 0) { ++nfresh[ch]; need_fresh[ch] = 0; if (reuse) { have_g[ch] = have_f[ch] = 1; g_tau[ch] = fac_tau[ch] = tauc[ch]; } }
            else if (pass == 1) { have_g[ch] = have_f[ch] = 0; }
            else { last_reused[ch] = 1; if (pass == 2) { have_f[ch] = 1; fac_tau[ch] = tauc[ch]; } }
          }
        }
      }
      if (o.beta_out)
        for (int ch = 0; ch < K; ++ch)
          for (int j = 0; j < R1; ++j)
            std::memcpy(o.beta_out + (((size_t)p * K + ch) * R1 + j) * L, hbetas.data() + ((size_t)ch * R1 + j) * n64, sizeof(double) * L);
      if (!ok) continue;  // pheno_l1_not_converged: LOCO predictions are skipped (Data.cpp:1016-1021)
      converged_out[p] = 1;
      int best = 0; double minv = 1e10;
      for (int j = 0; j < R1; ++j) {
        const double perf = cs[5 * R1 + j] / ctx->neff[pg];  // -logLik / Neff (Data.cpp:1030)
        if (perf < minv) { best = j; minv = perf; }
      }
      best_out[p] = best;
      L1X_HIP(hipMemcpyAsync(d_betas, hbetas.data(), sizeof(double) * hbetas.size(), hipMemcpyHostToDevice, st));
      L1X_HIP(hipMemsetAsync(c.d_pred, 0, sizeof(double) * (size_t)nchr * N, st));
      hipLaunchKernelGGL(k_fold_pred, dim3(ctx->n_c256), dim3(256), sizeof(double) * L, st, c.Wv, Np, L, c.Pv, pw,
                         d_betas + (int64_t)best * n64, (int64_t)R1 * n64, ctx->d_c256_seg, ctx->d_c256_pos,
                         ctx->d_c256_len, c.d_col0, nchr, ctx->d_cidx, N, c.d_pred);
    } else {
      // ---- LOOCV: warm-started Newton per tau, then the leave-one-out shortcut -----------------------------
      hipLaunchKernelGGL(k_transpose_w, dim3((unsigned)(Np / 64), n64 / 64), dim3(256), 0, st, c.Wv, Np, L, c.Pv, pw,
                         n64, d_Wt);
      std::vector<double> beta((size_t)n64, 0.0);
      LooArgs la{c.Wv, d_Ut, s.d_beta, Np, L, c.Pv, pw, d_yraw, d_rv, d_wv, d_off, ctx->d_maskp + (int64_t)pg * Np, poisson ? 2 : 1};
      auto newton = [&](double lam, std::vector<double>& b, bool* cv) { return poisson ? ct_newton_loocv(s, lam, b, o, cv) : bt_newton_loocv(s, lam, b, o, cv); };
      auto loo_setup = [&](double lam) -> int {   // H = (X^T W X + lam I)^-1 at the current weights, U^T = H W^T
        WgArgs g{c.Wv, ctx->d_zero, Np, L, c.Pv, pw, n64, d_wv, nullptr, nullptr, nullptr, 0, d_G, c.msz};
        { const int rcw = launch_wgram(ctx, st, g, T, 1); if (rcw) return rcw; }
        L1X_HIP(hipMemcpyAsync(d_tau1, &lam, sizeof(double), hipMemcpyHostToDevice, st));
        L1X_HIP(hipStreamSynchronize(st));
        int r2 = invert_shifted(ctx, c, d_G, d_tau1, 1, d_eye, d_sysI, d_dinvI, d_H);
        if (r2) return r2;
        rg_launch_dgemm_nt(st, d_H, n64, d_Wt, n64, n64, (int)Np, n64, d_Ut, Np);
        return RG_OK;
      };
      for (int j = 0; j < R1 && ok; ++j) {
        bool conv = false;
        if ((rc = newton(taup[j], beta, &conv))) return rc;
        if (!conv) { ok = false; break; }
        if ((rc = loo_setup(taup[j]))) return rc;
        hipLaunchKernelGGL(k_loo_cv, dim3(gpos), dim3(256), 0, st, la, d_lpart);
        L1X_HIP(hipMemcpyAsync(hl.data(), d_lpart, sizeof(double) * hl.size(), hipMemcpyDeviceToHost, st));
        L1X_HIP(hipStreamSynchronize(st));
        for (unsigned b = 0; b < gpos; ++b)
          for (int t = 0; t < 6; ++t) cs[t * R1 + j] += hl[(size_t)b * LOO_NPART + t];
        bool bad = false;
        if ((rc = check_spd(ctx, &bad))) return rc;
        if (bad) ok = false;
      }
      if (!ok) continue;
      converged_out[p] = 1;
      int best = 0; double minv = 1e10;
      for (int j = 0; j < R1; ++j) {
        const double perf = cs[5 * R1 + j] / ctx->neff[pg];
        if (perf < minv) { best = j; minv = perf; }
      }
      best_out[p] = best;
      // make_predictions_binary_loocv refits at tau* from beta = 0 (Data.cpp:1499-1503)
      std::fill(beta.begin(), beta.end(), 0.0);
      bool conv = false;
      if ((rc = newton(taup[best], beta, &conv))) return rc;
      if (!conv) { converged_out[p] = 0; continue; }
      if ((rc = loo_setup(taup[best]))) return rc;
      L1X_HIP(hipMemsetAsync(c.d_pred, 0, sizeof(double) * (size_t)nchr * N, st));
      hipLaunchKernelGGL(k_loo_pred, dim3(gpos), dim3(256), 0, st, la, c.d_col0, nchr, ctx->d_cidx, N, c.d_pred);
    }
    { const int rce = rg_emit_pred(ctx, st, c.d_pred, nchr, p, pred_out); if (rce) return rce; }
    L1X_HIP(hipStreamSynchronize(st));
  }
  return RG_OK;
}

// =========================================================================================================
// Time-to-event traits: Cox ridge at level 1 (ridge_cox_level_1, Step1_Models.cpp:2228-2305; cox_ridge.cpp; survival_data.cpp)
// =========================================================================================================
// The reference fits, per CV fold and penalty, the Cox partial likelihood with ridge penalty by IRLS on the DIAGONAL of the Hessian:
// an iteration computes the gradient g and diagonal h of the log partial likelihood at eta (cumulative sums over the samples sorted by
// time), the working response z = (eta - offset) - g / h, and makes ONE cyclic pass over the L coordinates
//   beta_k <- (r . x_k + beta_k s_k) / (s_k - lambda),  r = h (z - eta + offset) with the eta of the moment,  s_k = sum x_k^2 h
// (cox_ridge.cpp:116-178).  With the weights w = -h >= 0 that pass is one Gauss-Seidel sweep on (X^T W X + lambda I) beta = X^T W z
// started at the current beta, so the N-sized work of an iteration is the weighted Gram k_wgram128 already computes for the logistic
// ridge (2 N L^2 flop on the fp64 matrix cores instead of L dependent passes over N), plus
//   k_cox_eta   eta = offset + W^T beta on the samples of the chain (training fold, held-out fold, or all)
//   k_cox_scan  one workgroup: the samples in time order (a permutation the host sorts once per chain) -> mean of eta, the risk-set
//               sums by chunked scans (1,024 threads, a contiguous chunk each, chunk totals scanned in LDS), g, h, the weights and z
//               in position order, and the deviance (cox_ridge.cpp:60-114)
//   k_cox_xtg / k_cox_sweep   X^T g = c - G beta (a row per workgroup), then the sweep by one workgroup with v = c - G beta in LDS,
//               a rank-one update of v per coordinate, on the symmetrised Gram (k_cox_symm)
// The host keeps the reference's control flow: step halving against the previous deviance, the two stopping rules, the warm-started
// path over the penalties (largest first, every fit measured from the FIRST fit's starting deviance, cox_ridge.cpp:254-270), the
// held-out deviance of every solution, the penalty grid from the score at beta = 0.  Chains (folds) run one after the other: an
// iteration is dominated by its Gram, there is nothing to gain from interleaving them.
#define COX_T 1024
struct CoxChain {                 // one risk-set structure: a training fold, a held-out fold, or all samples
  const int32_t* order;           // [n] position of the i-th sample in (time asc, event before censored) order
  const uint8_t* flags;           // [n] bit 0: in the chain's sample set (keep), bit 1: status == 1, bit 2: dd == 1 (first event of its time)
  const double* ww;               // [n] tie-collapsed event weights (0 unless dd)
  int n; double w;                // w = 1 / neff
};
__global__ __launch_bounds__(256) void k_cox_eta(const double* W, int64_t Np, int L, int P, int p, const double* beta, const double* off,
                                                 const double* keepv, double* eta) {
  __shared__ double sB[256];
  const int64_t pos = (int64_t)blockIdx.x * 256 + threadIdx.x;
  double acc = 0.0;
  for (int c0 = 0; c0 < L; c0 += 256) {
    __syncthreads();
    sB[threadIdx.x] = (c0 + threadIdx.x < L) ? beta[c0 + threadIdx.x] : 0.0;
    __syncthreads();
    if (pos < Np) {
      const int cn = min(256, L - c0);
      const double* w = W + ((int64_t)c0 * P + p) * Np + pos;
      for (int c = 0; c < cn; ++c) acc = fma(w[(int64_t)c * P * Np], sB[c], acc);
    }
  }
  if (pos < Np) eta[pos] = keepv[pos] != 0.0 ? acc + off[pos] : 0.0;     // mask.select(X beta + offset, 0)
}

__device__ __forceinline__ double cox_block_sum(double v, double* red) {     // COX_T threads; result in every thread
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double t = 0.0;
  for (int i = 0; i < COX_T / 64; ++i) t += red[i];
  return t;
}
// exclusive scan of one value per thread over the workgroup, in thread order (rev = 0) or reverse thread order (rev = 1)
__device__ __forceinline__ double cox_block_excl(double v, double* buf, int rev) {
  __syncthreads();
  buf[threadIdx.x] = v;
  __syncthreads();
  if (threadIdx.x == 0) {         // 1,024 sequential additions: negligible next to the chunk passes, and a fixed order
    double run = 0.0;
    if (!rev) for (int i = 0; i < COX_T; ++i) { const double x = buf[i]; buf[i] = run; run += x; }
    else for (int i = COX_T - 1; i >= 0; --i) { const double x = buf[i]; buf[i] = run; run += x; }
  }
  __syncthreads();
  return buf[threadIdx.x];
}
// out[0] = log partial likelihood part sum_w eta [event] - sum ww log(risk set), out[1] = mean of eta over the chain's samples
__global__ __launch_bounds__(COX_T) void k_cox_scan(CoxChain ch, const double* eta, const double* off, double* rsk /*[n] scratch*/,
                                                    double* wv, double* zv, double* gv, int want_grad, double* out) {
  __shared__ double red[COX_T / 64];
  __shared__ double buf[COX_T];
  const int n = ch.n, C = (n + COX_T - 1) / COX_T;
  const int i0 = min(n, (int)threadIdx.x * C), i1 = min(n, i0 + C);
  const double w = ch.w;
  double s = 0.0;
  for (int i = i0; i < i1; ++i) if (ch.flags[i] & 1) s += eta[ch.order[i]];
  const double mean = cox_block_sum(s, red) * w;              // sum(eta w_orig) / sum(w_orig), w_orig = 1 / neff on the chain's samples
  // reverse cumulative sums of w exp(eta - mean) (gradient) and w exp(eta) (likelihood)
  double se = 0.0, sl = 0.0;
  for (int i = i0; i < i1; ++i)
    if (ch.flags[i] & 1) { const double e = eta[ch.order[i]]; se += w * exp(e - mean); sl += w * exp(e); }
  double run_e = cox_block_excl(se, buf, 1);
  double run_l = cox_block_excl(sl, buf, 1);
  double ll = 0.0, sa = 0.0, sb = 0.0;
  for (int i = i1 - 1; i >= i0; --i) {
    const uint8_t f = ch.flags[i];
    const double e = eta[ch.order[i]];
    if (f & 1) { run_e += w * exp(e - mean); run_l += w * exp(e); }
    rsk[i] = run_e;
    if ((f & 1) && (f & 2)) ll += w * e;
    if ((f & 1) && (f & 4)) {
      ll -= ch.ww[i] * log(run_l);
      sa += ch.ww[i] / run_e;
      sb += ch.ww[i] / (run_e * run_e);
    }
  }
  const double tot = cox_block_sum(ll, red);
  if (threadIdx.x == 0) { out[0] = tot; out[1] = mean; }
  if (!want_grad) return;
  double A = cox_block_excl(sa, buf, 0);
  double B = cox_block_excl(sb, buf, 0);
  for (int i = i0; i < i1; ++i) {
    const uint8_t f = ch.flags[i];
    const int pos = ch.order[i];
    if ((f & 1) && (f & 4)) { const double r = rsk[i]; A += ch.ww[i] / r; B += ch.ww[i] / (r * r); }
    double g = 0.0, h = 0.0;
    if (f & 1) {
      const double we = w * exp(eta[pos] - mean);
      g = w * ((f & 2) ? 1.0 : 0.0) - we * A;
      h = we * we * B - we * A;
    }
    wv[pos] = -h;
    zv[pos] = (f & 1) ? (eta[pos] - off[pos]) - (h != 0.0 ? g / h : 0.0) : 0.0;
    gv[pos] = g;
  }
}
// xtw[c] = sum_pos W[c][pos] g[pos]
__global__ __launch_bounds__(256) void k_cox_xtv(const double* W, int64_t Np, int P, int p, const double* g, double* out) {
  __shared__ double sred[4];
  const double* w = W + ((int64_t)blockIdx.x * P + p) * Np;
  double acc = 0.0;
  for (int64_t pos = threadIdx.x; pos < Np; pos += 256) acc = fma(w[pos], g[pos], acc);
  const double s = block_sum_256(acc, sred);
  if (threadIdx.x == 0) out[blockIdx.x] = s;
}
// upper triangle <- lower triangle of the n64 x n64 system (32 x 32 tiles)
__global__ __launch_bounds__(256) void k_cox_symm(double* S, int n64) {
  __shared__ double t[32][33];
  const int bi = blockIdx.y, bj = blockIdx.x;
  if (bj > bi) return;
  for (int r = threadIdx.x >> 5; r < 32; r += 8) t[r][threadIdx.x & 31] = S[(int64_t)(bi * 32 + r) * n64 + bj * 32 + (threadIdx.x & 31)];
  __syncthreads();
  for (int r = threadIdx.x >> 5; r < 32; r += 8) {
    const int row = bj * 32 + r, col = bi * 32 + (threadIdx.x & 31);
    if (col > row) S[(int64_t)row * n64 + col] = t[threadIdx.x & 31][r];
  }
}
// xtg[k] = c[k] - sum_j G[k][j] beta[j]  (G = system minus lambda on the diagonal; c = row n64 of the system)
__global__ __launch_bounds__(256) void k_cox_xtg(const double* S, int n64, int L, double lam, const double* beta, double* xtg) {
  __shared__ double sred[4];
  const int k = blockIdx.x;
  const double* row = S + (int64_t)k * n64;
  double acc = 0.0;
  for (int j = threadIdx.x; j < L; j += 256) acc = fma(row[j] - (j == k ? lam : 0.0), beta[j], acc);
  const double s = block_sum_256(acc, sred);
  if (threadIdx.x == 0) xtg[k] = S[(int64_t)n64 * n64 + k] - s;
}
// one cyclic pass: v = c - G beta kept in LDS; coordinate k: beta_k' = (v_k + G_kk beta_k) / (G_kk + lambda), v -= (beta_k' - beta_k) G[k][:]
__global__ __launch_bounds__(COX_T) void k_cox_sweep(const double* S, int n64, int L, double lam, const double* xtg, double* beta) {
  extern __shared__ double v[];      // [L]
  __shared__ double bk[2];
  for (int j = threadIdx.x; j < L; j += COX_T) v[j] = xtg[j];
  __syncthreads();
  for (int k = 0; k < L; ++k) {
    const double* row = S + (int64_t)k * n64;
    if (threadIdx.x == 0) {
      const double skk = row[k], b0 = beta[k];
      const double b1 = (v[k] + (skk - lam) * b0) / skk;
      beta[k] = b1;
      bk[0] = b1 - b0;
    }
    __syncthreads();
    const double d = bk[0];
    for (int j = threadIdx.x; j < L; j += COX_T) v[j] -= d * (row[j] - (j == k ? lam : 0.0));
    __syncthreads();
  }
}

namespace {

struct CoxHostChain {              // survival_data::setup (survival_data.cpp:9-100) for one sample set
  std::vector<int32_t> order; std::vector<uint8_t> flags; std::vector<double> ww;
  double neff = 0, lsat = 0;
};
// pos_of[n]: position of compact sample n; in_set[n]: 1 for the chain's samples
void cox_build_chain(const double* time, const double* event, const std::vector<uint8_t>& in_set, const std::vector<int64_t>& posc, CoxHostChain& c) {
  const int n = (int)in_set.size();
  std::vector<int32_t> idx(n);
  for (int i = 0; i < n; ++i) idx[i] = i;
  auto st = [&](int i) { return in_set[i] ? event[i] : -999.0; };
  std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) {      // _getOrder (:104-127): time, then events first
    if (time[a] != time[b]) return time[a] < time[b];
    return st(a) > st(b);
  });
  c.order.resize(n); c.flags.assign(n, 0); c.ww.assign(n, 0.0);
  c.neff = 0;
  for (int i = 0; i < n; ++i) c.neff += in_set[i] ? 1.0 : 0.0;
  const double w = 1.0 / c.neff;
  std::vector<int> ev;
  for (int i = 0; i < n; ++i) {
    const int s = idx[i];
    c.order[i] = (int32_t)posc[s];
    uint8_t f = 0;
    if (in_set[s]) { f |= 1; if (event[s] == 1.0) { f |= 2 | 4; ev.push_back(i); c.ww[i] = w; } }
    c.flags[i] = f;
  }
  // ties among the event times (_findTies :129-150): the first event of a time carries the weight of all of them
  std::vector<double> wsub;
  for (size_t a = 0; a < ev.size();) {
    size_t b = a + 1;
    while (b < ev.size() && time[idx[ev[b]]] == time[idx[ev[a]]]) ++b;
    if (b - a > 1) {
      for (size_t t = a + 1; t < b; ++t) { c.flags[ev[t]] &= (uint8_t)~4; c.ww[ev[t]] = 0.0; }
      c.ww[ev[a]] = (double)(b - a) * w;
    }
    wsub.push_back((double)(b - a) * w);
    a = b;
  }
  c.lsat = 0;                                                        // _coxDeviance (cox_ridge.cpp:93-114)
  for (double x : wsub) c.lsat -= x * std::log(x);
}

struct CoxState {
  rg_ctx* ctx; L1Common* c; int pw;
  double *d_eta, *d_off, *d_keep, *d_wv, *d_zv, *d_gv, *d_rsk, *d_beta, *d_xtg, *d_out, *d_sys, *d_tau1;
  int32_t* d_order; uint8_t* d_flags; double* d_ww; int32_t* d_map;
  CoxChain dev{};
  double lsat = 0;
};

int cox_load_chain(CoxState& s, const CoxHostChain& h, const std::vector<double>& keep_pos) {
  rg_ctx* ctx = s.ctx;
  hipStream_t st = s.c->st;
  const int n = (int)h.order.size();
  L1X_HIP(hipMemcpyAsync(s.d_order, h.order.data(), sizeof(int32_t) * n, hipMemcpyHostToDevice, st));
  L1X_HIP(hipMemcpyAsync(s.d_flags, h.flags.data(), n, hipMemcpyHostToDevice, st));
  L1X_HIP(hipMemcpyAsync(s.d_ww, h.ww.data(), sizeof(double) * n, hipMemcpyHostToDevice, st));
  L1X_HIP(hipMemcpyAsync(s.d_keep, keep_pos.data(), sizeof(double) * s.c->Np, hipMemcpyHostToDevice, st));
  L1X_HIP(hipStreamSynchronize(st));
  s.dev = CoxChain{s.d_order, s.d_flags, s.d_ww, n, 1.0 / h.neff};
  s.lsat = h.lsat;
  return RG_OK;
}
// eta at beta, then the deviance (and, if asked, gradient / weights / working response) of the loaded chain
int cox_eval(CoxState& s, const std::vector<double>& beta, bool want_grad, double* deviance) {
  rg_ctx* ctx = s.ctx;
  L1Common& c = *s.c;
  hipStream_t st = c.st;
  L1Lap lap(ctx, st, &ctx->tm.ms_irls_stream);
  L1X_HIP(hipMemcpyAsync(s.d_beta, beta.data(), sizeof(double) * c.L, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(k_cox_eta, dim3((unsigned)((c.Np + 255) / 256)), dim3(256), 0, st, c.Wv, c.Np, c.L, c.Pv, s.pw, s.d_beta, s.d_off,
                     s.d_keep, s.d_eta);
  hipLaunchKernelGGL(k_cox_scan, dim3(1), dim3(COX_T), 0, st, s.dev, s.d_eta, s.d_off, s.d_rsk, s.d_wv, s.d_zv, s.d_gv, want_grad ? 1 : 0,
                     s.d_out);
  double out[2];
  L1X_HIP(hipMemcpyAsync(out, s.d_out, sizeof(out), hipMemcpyDeviceToHost, st));
  L1X_HIP(hipStreamSynchronize(st));
  *deviance = 2.0 * (s.lsat - out[0]);
  return RG_OK;
}
// one IRLS iteration's coordinate pass at the state of the last cox_eval(want_grad): beta in/out, xtg = X^T g out
int cox_sweep(CoxState& s, double lam, std::vector<double>& beta, std::vector<double>& xtg) {
  rg_ctx* ctx = s.ctx;
  L1Common& c = *s.c;
  hipStream_t st = c.st;
  const int32_t zero = 0;
  L1X_HIP(hipMemcpyAsync(s.d_map, &zero, sizeof(int32_t), hipMemcpyHostToDevice, st));
  L1X_HIP(hipMemcpyAsync(s.d_tau1, &lam, sizeof(double), hipMemcpyHostToDevice, st));
  L1X_HIP(hipStreamSynchronize(st));
  WgArgs g{c.Wv, ctx->d_zero, c.Np, c.L, c.Pv, s.pw, c.n64, s.d_wv, s.d_zv, s.d_tau1, s.d_map, 0, s.d_sys, c.msz};
  {
    L1Lap lap(ctx, st, &ctx->tm.ms_wgram);
    const int rcw = launch_wgram(ctx, st, g, c.T, 1); if (rcw) return rcw;
  }
  ++ctx->tm.n_wgram; ++ctx->tm.n_irls_rounds;
  ctx->tm.wgram_positions += ctx->seg.pos_start[ctx->seg.nseg - 1] + ctx->seg.plen[ctx->seg.nseg - 1];   // held-out samples carry zero weights but are contracted
  L1Lap lap2(ctx, st, &ctx->tm.ms_irls_solve);
  hipLaunchKernelGGL(k_cox_symm, dim3(c.n64 / 32, c.n64 / 32), dim3(256), 0, st, s.d_sys, c.n64);
  hipLaunchKernelGGL(k_cox_xtg, dim3(c.L), dim3(256), 0, st, s.d_sys, c.n64, c.L, lam, s.d_beta, s.d_xtg);
  hipFuncSetAttribute(reinterpret_cast<const void*>(k_cox_sweep), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(double) * c.L));
  hipLaunchKernelGGL(k_cox_sweep, dim3(1), dim3(COX_T), sizeof(double) * c.L, st, s.d_sys, c.n64, c.L, lam, s.d_xtg, s.d_beta);
  L1X_HIP(hipMemcpyAsync(beta.data(), s.d_beta, sizeof(double) * c.L, hipMemcpyDeviceToHost, st));
  L1X_HIP(hipMemcpyAsync(xtg.data(), s.d_xtg, sizeof(double) * c.L, hipMemcpyDeviceToHost, st));
  L1X_HIP(hipStreamSynchronize(st));
  return RG_OK;
}

}  // namespace

int rg_l1_cox_impl(rg_ctx* ctx, int pheno, int R1, const double* time, const double* event, const double* offset, const rg_cox_options* opt,
                   int nchr, const int32_t* cols_per_chr, double* tau_out, double* deviance_out, int32_t* converged_out, int32_t* best_out,
                   double* pred_out) {
  if (R1 < 2 || R1 > 16) { ctx->err = "rg_l1_cox: n_ridge_l1 must be in [2,16]"; return RG_ERR_ARG; }
  if (ctx->loocv) { ctx->err = "rg_l1_cox: time-to-event traits use K-fold cross-validation (Regenie.cpp:1199-1201)"; return RG_ERR_STATE; }
  rg_cox_options o;
  o.niter_max = 50; o.niter_max_line_search = 25; o.niter_max_ridge = 100; o.niter_max_line_search_ridge = 100;
  o.numtol_cox = 2.5e-4; o.l1_ridge_tol = 1e-4; o.tau = nullptr;
  if (opt) o = *opt;
  L1Common c;
  int rc = l1_common_init(ctx, c, nchr, cols_per_chr, "rg_l1_cox");
  if (rc) return rc;
  if (pheno < c.p0v || pheno >= c.p0v + c.P) { ctx->err = "rg_l1_cox: phenotype outside the level-1 view"; return RG_ERR_ARG; }
  hipStream_t st = c.st;
  const int L = c.L, n64 = c.n64, K = ctx->K;
  const int64_t Np = c.Np, N = c.N;
  const int pw = c.view ? pheno - c.p0v : pheno;
  if ((size_t)L * sizeof(double) > 150 * 1024) { ctx->err = "rg_l1_cox: more level-0 predictors than the coordinate pass holds in LDS (19,200)"; return RG_ERR_ARG; }

  CoxState s;
  s.ctx = ctx; s.c = &c; s.pw = pw;
  L1X_HIP(c.bufs.alloc(&s.d_eta, (size_t)Np)); L1X_HIP(c.bufs.alloc(&s.d_off, (size_t)Np)); L1X_HIP(c.bufs.alloc(&s.d_keep, (size_t)Np));
  L1X_HIP(c.bufs.alloc(&s.d_wv, (size_t)Np)); L1X_HIP(c.bufs.alloc(&s.d_zv, (size_t)Np)); L1X_HIP(c.bufs.alloc(&s.d_gv, (size_t)Np));
  L1X_HIP(c.bufs.alloc(&s.d_rsk, (size_t)N)); L1X_HIP(c.bufs.alloc(&s.d_beta, (size_t)n64)); L1X_HIP(c.bufs.alloc(&s.d_xtg, (size_t)n64));
  L1X_HIP(c.bufs.alloc(&s.d_out, 2)); L1X_HIP(c.bufs.alloc(&s.d_sys, (size_t)c.msz)); L1X_HIP(c.bufs.alloc(&s.d_tau1, 1));
  L1X_HIP(c.bufs.alloc(&s.d_order, (size_t)N)); L1X_HIP(c.bufs.alloc(&s.d_flags, (size_t)N)); L1X_HIP(c.bufs.alloc(&s.d_ww, (size_t)N));
  L1X_HIP(c.bufs.alloc(&s.d_map, 1));
  double* d_betas = nullptr;
  L1X_HIP(c.bufs.alloc(&d_betas, (size_t)K * R1 * n64));
  L1X_HIP(hipMemsetAsync(s.d_beta, 0, sizeof(double) * n64, st));
  L1X_HIP(hipMemsetAsync(s.d_wv, 0, sizeof(double) * Np, st));
  L1X_HIP(hipMemsetAsync(s.d_zv, 0, sizeof(double) * Np, st));
  L1X_HIP(hipMemsetAsync(s.d_gv, 0, sizeof(double) * Np, st));
  std::vector<double> hpos, hmask((size_t)Np);
  to_pos(ctx, offset, hpos);
  L1X_HIP(hipMemcpyAsync(s.d_off, hpos.data(), sizeof(double) * Np, hipMemcpyHostToDevice, st));
  L1X_HIP(hipMemcpyAsync(hmask.data(), ctx->d_maskp + (int64_t)pheno * Np, sizeof(double) * Np, hipMemcpyDeviceToHost, st));
  L1X_HIP(hipStreamSynchronize(st));
  std::vector<uint8_t> mask((size_t)N), in_set((size_t)N);
  std::vector<int> fold((size_t)N, 0);
  for (int64_t n = 0; n < N; ++n) mask[n] = hmask[(size_t)ctx->h_posc[n]] != 0.0;
  for (int f = 0; f < K; ++f)
    for (int64_t n = ctx->fold_cstart[f]; n < ctx->fold_cstart[f + 1]; ++n) fold[n] = f;
  auto load = [&](int kind, int f) -> int {     // kind 0: all samples, 1: training samples of fold f, 2: held-out samples of fold f
    for (int64_t n = 0; n < N; ++n) in_set[n] = mask[n] && (kind == 0 || (kind == 1 ? fold[n] != f : fold[n] == f));
    CoxHostChain h;
    cox_build_chain(time, event, in_set, ctx->h_posc, h);
    if (h.neff < 1) { ctx->err = "rg_l1_cox: a fold without samples"; return RG_ERR_ARG; }
    std::vector<double> kp((size_t)Np, 0.0);
    for (int64_t n = 0; n < N; ++n) if (in_set[n]) kp[(size_t)ctx->h_posc[n]] = 1.0;
    return cox_load_chain(s, h, kp);
  };

  // ---- penalty grid: lambda_max = max |X^T g| / 1e-3 at beta = 0 on all samples (getCoxLambdaMax :446-450), then
  //      tau_j = lambda_max * 1e-6^(j / (R1 - 1))  (check_l0 :2105-2113) ----
  std::vector<double> beta((size_t)L, 0.0), xtg((size_t)L, 0.0), tau((size_t)R1);
  double dev0 = 0;
  if ((rc = load(0, 0))) return rc;
  if ((rc = cox_eval(s, beta, true, &dev0))) return rc;
  hipLaunchKernelGGL(k_cox_xtv, dim3(L), dim3(256), 0, st, c.Wv, Np, c.Pv, pw, s.d_gv, s.d_xtg);
  L1X_HIP(hipMemcpyAsync(xtg.data(), s.d_xtg, sizeof(double) * L, hipMemcpyDeviceToHost, st));
  L1X_HIP(hipStreamSynchronize(st));
  double gmax = 0;
  for (int k = 0; k < L; ++k) gmax = std::max(gmax, std::fabs(xtg[k]));
  const double lam_max = gmax / 1e-3;
  for (int j = 0; j < R1; ++j) tau[j] = std::exp((double)j / (R1 - 1) * std::log(1e-6) + std::log(lam_max));
  if (o.tau)   // --t2e-l1-pi6: the caller's penalties replace the path from lambda_max (check_l0 :2106-2110)
    for (int j = 0; j < R1; ++j) {
      if (!(o.tau[j] > 0)) { ctx->err = "rg_l1_cox: the penalties given in rg_cox_options.tau must be positive"; return RG_ERR_ARG; }
      tau[j] = o.tau[j];
    }
  for (int j = 0; j < R1; ++j) { tau_out[j] = tau[j]; deviance_out[j] = 0.0; }
  *converged_out = 0; *best_out = 0;

  // ---- per fold: the warm-started path, then the held-out deviance of each solution ----
  std::vector<double> hbetas((size_t)K * R1 * n64, 0.0), beta_old((size_t)L);
  bool all_conv = true;
  auto sq = [&](const std::vector<double>& b) { double t = 0; for (int k = 0; k < L; ++k) t += b[k] * b[k]; return t; };
  for (int f = 0; f < K; ++f) {
    if ((rc = load(1, f))) return rc;
    std::fill(beta.begin(), beta.end(), 0.0);
    double dev_first = 0;
    for (int j = 0; j < R1; ++j) {                   // cox_ridge_path::fit (cox_ridge.cpp:246-287)
      const double lam = tau[j];
      double dev_prev, dev;
      if ((rc = cox_eval(s, beta, true, &dev))) return rc;     // state at the start values: gradient for the first pass
      if (j == 0) dev_first = dev;
      dev_prev = dev_first;                                    // every fit of the path starts its deviance record at the first fit's
      double obj_prev = dev_prev + lam * sq(beta) / 2;
      bool conv = false;
      for (int t = 1; t <= o.niter_max_ridge; ++t) {           // cox_ridge::fit (:116-178)
        beta_old = beta;
        if ((rc = cox_sweep(s, lam, beta, xtg))) return rc;
        if ((rc = cox_eval(s, beta, true, &dev))) return rc;
        double obj = dev + lam * sq(beta) / 2;
        if (dev - dev_prev > o.l1_ridge_tol) {                 // step halving towards the previous iterate
          int ii = 0;
          bool gave_up = false;
          while (dev - dev_prev > o.l1_ridge_tol) {
            if (++ii > o.niter_max_line_search_ridge) { gave_up = true; break; }
            for (int k = 0; k < L; ++k) beta[k] = (beta[k] + beta_old[k]) / 2;
            if ((rc = cox_eval(s, beta, true, &dev))) return rc;
            obj = dev + lam * sq(beta) / 2;
          }
          if (gave_up) break;                                  // "cannot correct step size": the fit ends unconverged
        }
        double score = 0;
        for (int k = 0; k < L; ++k) score = std::max(score, std::fabs(xtg[k] - lam * beta[k]));
        const bool stop = std::fabs(obj - obj_prev) / (0.1 + std::fabs(obj)) < o.l1_ridge_tol || score < o.l1_ridge_tol;
        dev_prev = dev; obj_prev = obj;
        if (stop) { conv = true; break; }
      }
      all_conv = all_conv && conv;
      std::memcpy(hbetas.data() + ((size_t)f * R1 + j) * n64, beta.data(), sizeof(double) * L);
    }
    if ((rc = load(2, f))) return rc;                           // held-out deviance (Step1_Models.cpp:2293-2297)
    for (int j = 0; j < R1; ++j) {
      std::vector<double> b(hbetas.begin() + ((size_t)f * R1 + j) * n64, hbetas.begin() + ((size_t)f * R1 + j) * n64 + L);
      double dv = 0;
      if ((rc = cox_eval(s, b, false, &dv))) return rc;
      deviance_out[j] += dv;
    }
  }
  if (!all_conv) return RG_OK;                                  // pheno_l1_not_converged: predictions are skipped (Data.cpp:1016-1021)
  *converged_out = 1;
  int best = 0; double minv = 1e10;
  for (int j = 0; j < R1; ++j) if (deviance_out[j] < minv) { best = j; minv = deviance_out[j]; }      // no division by Neff (Data.cpp:1031)
  *best_out = best;
  L1X_HIP(hipMemcpyAsync(d_betas, hbetas.data(), sizeof(double) * hbetas.size(), hipMemcpyHostToDevice, st));
  L1X_HIP(hipMemsetAsync(c.d_pred, 0, sizeof(double) * (size_t)nchr * N, st));
  hipLaunchKernelGGL(k_fold_pred, dim3(ctx->n_c256), dim3(256), sizeof(double) * L, st, c.Wv, Np, L, c.Pv, pw, d_betas + (int64_t)best * n64,
                     (int64_t)R1 * n64, ctx->d_c256_seg, ctx->d_c256_pos, ctx->d_c256_len, c.d_col0, nchr, ctx->d_cidx, N, c.d_pred);
  { const int rce = rg_emit_pred(ctx, st, c.d_pred, nchr, 0, pred_out); if (rce) return rce; }
  L1X_HIP(hipStreamSynchronize(st));
  return RG_OK;
}
