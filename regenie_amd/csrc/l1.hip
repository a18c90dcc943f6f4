// Level-1 cross-validated ridge for quantitative traits (K-fold) on the level-0 predictors W.
//
// Reference: src/Step1_Models.cpp:772-872 ridge_level_1 (fold Grams X_folds[i] = W_i^T W_i,
// XtY[i] = W_i^T y_i, per fold solve for all tau, out-of-fold p1 = W_i beta, five running sums),
// src/Data.cpp:1025-1037 (argmin MSE) and :1242-1258 make_predictions (per-chromosome W_i beta_i).
//
// W lives in HBM as [L][P][Np] (fold-aligned position space; padding / ignored samples are exact
// zeros so they drop out of every contraction).  The fold Grams run on the fp64 MFMA straight from
// HBM rows (K-contiguous on both sides, see chol.hip); y rides along as one extra row so
// W_f^T y_f and the fold systems' right-hand sides come out of the same launch.  The K*R1 systems
// (sum_f S_f - S_i + tau_j I) are then factored by the same batched Cholesky as level 0.
#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdint>
#include <vector>
#include "rg_internal.h"

#define CT 64

// ---- fold Gram: out[f][tr*64..][tc*64..] = sum_{pos in fold f} row_tr(pos) * row_tc(pos) ------------
// rows < L : W column (i*P + p);  L <= row < n64 : zero row;  row == n64 : y_p;  other: zero row.
struct L1Rows {
  const double* W; const double* y; const double* zero;
  int64_t Np; int L, P, p, n64;
};
__device__ __forceinline__ const double* l1_row(const L1Rows& R, int row) {
  if (row < R.L) return R.W + ((int64_t)row * R.P + R.p) * R.Np;
  if (row == R.n64) return R.y;
  return R.zero;
}

// ---- fold Gram, one WAVE per 64x64 tile (4 x 4 MFMA 16x16x4 sub-tiles, as k_chol_update): 4 B/clk of operand
//      traffic per wave instead of 8 with 32x32 per wave.  Work item = (K slice, tile); tiles are enumerated
//      column by column so the four waves of a workgroup share their B rows in L1; the K range of a fold is cut
//      into `nslice` slices when there are too few tiles to fill the chip (small L), each slice writing its own
//      partial matrix (summed in fixed order by k_reduce_slices: deterministic).  Multi-GPU: rank r computes the
//      tiles with index = r (mod world); the others stay zero and an all-reduce completes the matrices.
struct L1G64 {
  L1Rows R; int T, rtot, nslice, ntile, world, rank;
  double* out; int64_t slice_stride;
};
__global__ __launch_bounds__(256, 2) void k_l1_gram64(L1G64 g, SegLayout seg) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int f = blockIdx.y;
  const int item = blockIdx.x * 4 + wave;
  if (item >= g.ntile * g.nslice) return;
  const int sl = item / g.ntile, idx = item % g.ntile;
  if (idx % g.world != g.rank) return;
  int tc = 0, rem = idx;
  while (rem >= g.T + 1 - tc) { rem -= g.T + 1 - tc; ++tc; }
  const int tr = tc + rem;                       // tr == T: the right-hand-side row tile (y)
  const int i = lane & 15, q = lane >> 4;
  const int64_t nch = seg.plen[f] / 64;
  const int64_t c0 = nch * sl / g.nslice, c1 = nch * (sl + 1) / g.nslice;
  const int64_t p0 = seg.pos_start[f] + 2 * q;
  const double* A[4];
  const double* B[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    A[m] = l1_row(g.R, tr * CT + m * 16 + i) + p0;
    B[m] = l1_row(g.R, tc * CT + m * 16 + i) + p0;
  }
  v4d acc[4][4];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[m][n] = (v4d){0, 0, 0, 0};
  // 8-deep K chunks, two register sets: the loads of chunk c+1 fly while the 32 MFMAs of chunk c issue (see
  // k_chol_update).  Lane (i, q) supplies k = 2q + s of a chunk to MFMA step s.
  auto load8 = [&](double2 (&av)[4], double2 (&bv)[4], int64_t kc) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      av[m] = *reinterpret_cast<const double2*>(A[m] + kc * 8);
      bv[m] = *reinterpret_cast<const double2*>(B[m] + kc * 8);
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
  {
    const int64_t k0 = c0 * 8, k1 = c1 * 8;   // 64-chunks -> 8-chunks; (k1 - k0) is a multiple of 8
    if (k1 > k0) {
      double2 a0[4], b0[4], a1[4], b1[4];
      load8(a0, b0, k0);
      for (int64_t kc = k0; kc < k1; kc += 2) {
        load8(a1, b1, kc + 1);
        mma8(a0, b0);
        if (kc + 2 < k1) load8(a0, b0, kc + 2);
        mma8(a1, b1);
      }
    }
  }
  double* O = g.out + (int64_t)sl * g.slice_stride + (int64_t)f * g.rtot * g.R.n64 + (int64_t)tr * CT * g.R.n64 + tc * CT;
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) O[(int64_t)(m * 16 + q + 4 * r) * g.R.n64 + n * 16 + i] = acc[m][n][r];
}

// ---- fold Gram through LDS: one WORKGROUP per 128 x 128 macro tile (4 waves, 2 x 2 of 64 x 64) --------------------------------
// The register-fed kernel above pulls 8 flop per byte through the L2 -> CU fabric (~9 TB/s chip-wide), which caps it near
// 30 TFLOP/s executed; at BASELINE configs[2] (L = 2,560 - 2,610 predictors, 500,000 samples, 10 phenotypes) it was 54 % of the
// whole Step-1 run.  Here the 256 operand rows of a macro tile are staged 16 positions at a time (32 KB per stage, two stages)
// by direct global -> LDS copies and shared by the four waves: 16 flop per byte from L2, no staging registers.
//   LDS stage: [256 rows][16 doubles]; the eight 16-byte slots of a row are XOR-swizzled by ((row >> 1) & 7) and lane (i, q)
//   reads the logical slots q and q + 4 (k = 2q, 2q+1, 8+2q, 9+2q) of its rows: every ds_read_b128 lane group hits 16
//   distinct bank quads (the layout of chol.hip's strip kernel).  A contraction is invariant under a permutation of K applied
//   to both operands alike, so MFMA step s simply takes the lane's s-th value.
//   One barrier per stage: the copies of stage s+1 are issued (inline assembly, invisible to hipcc's wait-count model) before
//   the 64 MFMAs of stage s and must have landed -- s_waitcnt vmcnt(0) -- before the barrier that ends it.
// Work items come from a host-built table: (macro row, macro column, fold, K slice), ordered so that the workgroups an XCD
// holds at one time belong to one 4 x 4 super tile of one (fold, slice) and stream the same rows through that XCD's L2.
// The y row (W_f^T y_f) is k_l1_wty's; multi-GPU: rank r computes the macro tiles with index = r (mod world).
struct L1Item { int16_t mr, mc, fold, slice; };
struct L1G128 {
  L1Rows R; int rtot, nslice; const L1Item* items; double* out; int64_t slice_stride;
};
#define G128_STAGE 32768   // bytes per LDS stage: 256 rows x 16 positions x 8 B
__device__ __forceinline__ const double* l1_wrow(const L1Rows& R, int row) {     // predictor row, or the zero row past L
  return row < R.L ? R.W + ((int64_t)row * R.P + R.p) * R.Np : R.zero;
}
__global__ __launch_bounds__(256, 2) void k_l1_gram128(L1G128 g, SegLayout seg) {
  __shared__ __attribute__((aligned(16))) uint8_t smem[2 * G128_STAGE];
  const L1Item it = g.items[blockIdx.x];
  if (it.fold < 0) return;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int i = lane & 15, q = lane >> 4;
  const int f = it.fold, n64 = g.R.n64;
  const int64_t nst = seg.plen[f] / 16;
  const int64_t s0 = nst * it.slice / g.nslice, s1 = nst * (it.slice + 1) / g.nslice;
  const int ns = (int)(s1 - s0);
  // the eight 1 KB pieces (8 rows x 128 B) of a stage this wave copies: pieces [8 wave, 8 wave + 8) of 32
  const double* src[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int rl = 8 * (wave * 8 + j) + (lane >> 3);
    const int row = rl < 128 ? it.mr * 128 + rl : it.mc * 128 + (rl - 128);
    const int slot = (lane & 7) ^ ((rl >> 1) & 7);
    src[j] = l1_wrow(g.R, row) + seg.pos_start[f] + s0 * 16 + 2 * slot;
  }
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)smem;
  const uint32_t wdst = __builtin_amdgcn_readfirstlane(lds0 + wave * 8192);
  auto issue = [&](int buf) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      glds16p(src[j], wdst + buf * G128_STAGE + j * 1024);
      src[j] += 16;
    }
  };
  // a wave whose 64 x 64 sub-tile lies strictly above the diagonal, or past the padded order, only takes part in the staging
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
    const uint8_t* cur = smem + (s & 1) * G128_STAGE;
    if (s + 1 < ns) issue((s + 1) & 1);
    if (live) {
      double2 a[4][2], b[4][2];
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        a[m][0] = *reinterpret_cast<const double2*>(cur + oa + m * 2048);
        a[m][1] = *reinterpret_cast<const double2*>(cur + oa + m * 2048 + o4);
        b[m][0] = *reinterpret_cast<const double2*>(cur + ob + m * 2048);
        b[m][1] = *reinterpret_cast<const double2*>(cur + ob + m * 2048 + o4);
      }
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int n = 0; n < 4; ++n) {
            const double av = kk == 0 ? a[m][0].x : (kk == 1 ? a[m][0].y : (kk == 2 ? a[m][1].x : a[m][1].y));
            const double bv = kk == 0 ? b[n][0].x : (kk == 1 ? b[n][0].y : (kk == 2 ? b[n][1].x : b[n][1].y));
            acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[m][n], 0, 0, 0);
          }
    }
    // the next stage has landed and everybody is done with this one
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
  if (!live) return;
  double* O = g.out + (int64_t)it.slice * g.slice_stride + (int64_t)f * g.rtot * n64 + (int64_t)row0 * n64 + col0;
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) O[(int64_t)(m * 16 + q + 4 * r) * n64 + n * 16 + i] = acc[m][n][r];
}

// ---- W_f^T y_f: row n64 of every fold matrix (the right-hand sides), one workgroup per (predictor, fold) -----------------------
// fixed-order reduction (thread-strided partial sums, wave shuffles, four wave sums): the value does not depend on the launch
__global__ __launch_bounds__(256) void k_l1_wty(L1Rows R, SegLayout seg, int rtot, int world, int rank, double* out) {
  __shared__ double red[4];
  const int l = blockIdx.x, f = blockIdx.y;
  if (l % world != rank) return;
  const double* w = R.W + ((int64_t)l * R.P + R.p) * R.Np + seg.pos_start[f];
  const double* y = R.y + seg.pos_start[f];
  const int64_t n = seg.plen[f];
  double t0 = 0.0, t1 = 0.0;
  for (int64_t e = 2 * (int64_t)threadIdx.x; e < n; e += 512) {
    const double2 wv = *reinterpret_cast<const double2*>(w + e), yv = *reinterpret_cast<const double2*>(y + e);
    t0 = fma(wv.x, yv.x, t0);
    t1 = fma(wv.y, yv.y, t1);
  }
  double t = t0 + t1;
  for (int o = 32; o > 0; o >>= 1) t += __shfl_down(t, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = t;
  __syncthreads();
  if (threadIdx.x == 0) out[(int64_t)f * rtot * R.n64 + (int64_t)R.n64 * R.n64 + l] = (red[0] + red[1]) + (red[2] + red[3]);
}

// Work table of k_l1_gram128 for one rank (see the kernel): macro tiles (mr >= mc) of the padded order n64, K folds, nslice K
// slices.  Groups = the macro tiles of one 4 x 4 super tile of one (fold, slice); the groups are dealt to the eight XCDs and
// workgroup id 8 j + x is the j-th item of XCD x's list (ids are dispatched to XCD id mod 8).  Lists are padded with fold = -1.
static std::vector<L1Item> l1_build_items(int n64, int K, int nslice, int world, int rank) {
  const int T2 = (n64 + 127) / 128, S = 4, TS = (T2 + S - 1) / S;
  std::vector<std::vector<L1Item>> xl(8);
  int gi = 0;
  for (int f = 0; f < K; ++f)
    for (int sl = 0; sl < nslice; ++sl)
      for (int Mr = 0; Mr < TS; ++Mr)
        for (int Mc = 0; Mc <= Mr; ++Mc) {
          std::vector<L1Item>& dst = xl[gi % 8];
          bool any = false;
          for (int mr = Mr * S; mr < std::min(T2, (Mr + 1) * S); ++mr)
            for (int mc = Mc * S; mc < std::min(T2, (Mc + 1) * S); ++mc) {
              if (mc > mr) continue;
              if ((mr * (mr + 1) / 2 + mc) % world != rank) continue;
              dst.push_back(L1Item{(int16_t)mr, (int16_t)mc, (int16_t)f, (int16_t)sl});
              any = true;
            }
          if (any) ++gi;
        }
  size_t mx = 0;
  for (auto& v : xl) mx = std::max(mx, v.size());
  std::vector<L1Item> items(mx * 8, L1Item{0, 0, -1, 0});
  for (int x = 0; x < 8; ++x)
    for (size_t j = 0; j < xl[x].size(); ++j) items[8 * j + x] = xl[x][j];
  return items;
}

// The same fold Gram for any row-major matrix G [n64][ld] (rows >= L read as zero): the fp64 genotype path of l0_f64.hip.
// out: [nfold][rtot][n64], lower 64x64 tiles; the right-hand-side row tiles are written as zeros (the caller fills them).
void rg_launch_fold_gram_rows(hipStream_t st, const double* G, int64_t ld, int L, int n64, int rtot, const double* zero,
                              const SegLayout& seg, double* out) {
  L1G64 g;
  g.R = L1Rows{G, zero, zero, ld, L, 1, 0, n64};
  g.T = n64 / CT; g.rtot = rtot; g.nslice = 1; g.ntile = g.T * (g.T + 1) / 2 + g.T; g.world = 1; g.rank = 0;
  g.out = out; g.slice_stride = 0;
  hipLaunchKernelGGL(k_l1_gram64, dim3((g.ntile + 3) / 4, seg.nseg), dim3(256), 0, st, g, seg);
}

__global__ void k_reduce_slices(const double* part, int64_t slice_stride, int nslice, int64_t n, double* out) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  double t = 0.0;
  for (int s = 0; s < nslice; ++s) t += part[(int64_t)s * slice_stride + e];
  out[e] = t;
}

__global__ void k_sum_folds(const double* fold, int64_t msz, int nfold, double* sum) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= msz) return;
  double t = 0.0;
  for (int f = 0; f < nfold; ++f) t += fold[(int64_t)f * msz + e];
  sum[e] = t;
}

// ---- LOCO assembly (write_predictions, Data.cpp:1846-1858): out[c][n] = sum_k pred[k][n] - pred[idx(c)][n] ------------
__global__ __launch_bounds__(256) void k_loco(const double* pred, int nchr, int64_t N, const int32_t* chrom, int nchrom,
                                              double* out) {
  const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  double tot = 0.0;
  for (int k = 0; k < nchr; ++k) tot += pred[(int64_t)k * N + n];      // fixed order: the reference's rowwise().sum()
  for (int c = 0; c < nchrom; ++c) out[(int64_t)c * N + n] = tot;
  for (int k = 0; k < nchr; ++k) out[(int64_t)(chrom[k] - 1) * N + n] = tot - pred[(int64_t)k * N + n];
}

int rg_emit_pred(rg_ctx* ctx, hipStream_t st, const double* d_pred, int nchr, int p, double* pred_out) {
  const int64_t N = ctx->N;
  if (ctx->loco_nchrom <= 0) {
    RG_HIP(hipMemcpyAsync(pred_out + (int64_t)p * nchr * N, d_pred, sizeof(double) * (size_t)nchr * N, hipMemcpyDeviceToHost, st));
    return RG_OK;
  }
  if ((int)ctx->loco_chrom.size() != nchr) { ctx->err = "LOCO output: rg_set_loco_output was given a different number of chromosomes"; return RG_ERR_ARG; }
  const int nchrom = ctx->loco_nchrom;
  double* d_out = (double*)rg_ws(ctx, 11, sizeof(double) * (size_t)nchrom * N + sizeof(int32_t) * (size_t)nchr + 64);
  if (!d_out) { ctx->err = "LOCO output: out of device memory"; return RG_ERR_HIP; }
  int32_t* d_chr = reinterpret_cast<int32_t*>(d_out + (size_t)nchrom * N);
  RG_HIP(hipMemcpyAsync(d_chr, ctx->loco_chrom.data(), sizeof(int32_t) * nchr, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(k_loco, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, d_pred, nchr, N, d_chr, nchrom, d_out);
  RG_HIP(hipMemcpyAsync(pred_out + (int64_t)p * nchrom * N, d_out, sizeof(double) * (size_t)nchrom * N, hipMemcpyDeviceToHost, st));
  return RG_OK;
}

// ---- multi-rank exchange of the fold Grams: only the computed tiles travel ------------------------------------------
// The fold matrices live as full rows (ld = n64) but only the lower-triangle tiles and the y-row tiles are ever non-zero;
// packing them tile by tile ([fold][tile][64][64], tile (tr, tc) -> tr(tr+1)/2 + tc) halves the all-reduce volume.
__global__ __launch_bounds__(256) void k_tiles_pack(double* fold, int64_t msz, int T, int n64, double* packed, int unpack) {
  const int t = blockIdx.x, f = blockIdx.y;
  int tr = (int)((sqrtf(8.0f * t + 1.0f) - 1.0f) * 0.5f);
  while ((tr + 1) * (tr + 2) / 2 <= t) ++tr;
  while (tr * (tr + 1) / 2 > t) --tr;
  int tc = t - tr * (tr + 1) / 2;
  if (tr >= T) { tc = t - T * (T + 1) / 2; tr = T; }      // the y row tiles follow the triangle
  double* src = fold + (int64_t)f * msz + (int64_t)tr * CT * n64 + tc * CT;
  double* pk = packed + ((int64_t)f * gridDim.x + t) * (CT * CT);
  for (int e = threadIdx.x; e < CT * CT; e += 256) {
    const int r = e >> 6, c = e & 63;
    if (unpack) src[(int64_t)r * n64 + c] = pk[e];
    else pk[e] = src[(int64_t)r * n64 + c];
  }
}

// ---- out-of-fold predictions for every tau + the five running sums -----------------------------------
// alpha: [(f*R1 + j)] systems, solution = RHS row 0 (row n64) of each factored system.
// grid (nchunk), 256 threads, thread = one position.  part: [chunk][R1][3] + ysum [chunk][2]
#define R1MAX 8
#define L1_CT 256
__global__ __launch_bounds__(256) void k_l1_cv(L1Rows R, const double* alpha /*[K*R1][n64]*/, int R1,
                                               const int32_t* chunk_seg, const int64_t* chunk_pos,
                                               const int64_t* chunk_len, double* part) {
  __shared__ double sA[L1_CT][R1MAX];
  __shared__ double sred[4][R1MAX * 3 + 2];
  const int ch = blockIdx.x;
  const int f = chunk_seg[ch];
  const int64_t p0 = chunk_pos[ch], plen = chunk_len[ch];
  double tot[R1MAX * 3 + 2];
#pragma unroll
  for (int t = 0; t < R1MAX * 3 + 2; ++t) tot[t] = 0.0;
  for (int64_t sub = 0; sub < plen; sub += 256) {
    const int64_t pos = p0 + sub + threadIdx.x;
    const bool live = sub + threadIdx.x < plen;
    double acc[R1MAX];
#pragma unroll
    for (int j = 0; j < R1MAX; ++j) acc[j] = 0.0;
    for (int c0 = 0; c0 < R.L; c0 += L1_CT) {
      __syncthreads();
      for (int j = 0; j < R1MAX; ++j) {
        const int col = c0 + threadIdx.x;
        sA[threadIdx.x][j] = (j < R1 && col < R.L)
            ? alpha[((int64_t)f * R1 + j) * R.n64 + col] : 0.0;
      }
      __syncthreads();
      if (live) {
        const int cn = min(L1_CT, R.L - c0);
        const double* w = R.W + ((int64_t)c0 * R.P + R.p) * R.Np + pos;
        for (int c = 0; c < cn; ++c) {
          const double x = w[(int64_t)c * R.P * R.Np];
#pragma unroll
          for (int j = 0; j < R1MAX; ++j) acc[j] = fma(x, sA[c][j], acc[j]);
        }
      }
    }
    if (live) {
      const double y = R.y[pos];
#pragma unroll
      for (int j = 0; j < R1MAX; ++j) {
        tot[3 * j] += acc[j];
        tot[3 * j + 1] = fma(acc[j], acc[j], tot[3 * j + 1]);
        tot[3 * j + 2] = fma(acc[j], y, tot[3 * j + 2]);
      }
      tot[R1MAX * 3] += y;
      tot[R1MAX * 3 + 1] = fma(y, y, tot[R1MAX * 3 + 1]);
    }
  }
#pragma unroll
  for (int t = 0; t < R1MAX * 3 + 2; ++t) {
    double x = tot[t];
    for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o);
    if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6][t] = x;
  }
  __syncthreads();
  if (threadIdx.x < R1MAX * 3 + 2)
    part[(int64_t)ch * (R1MAX * 3 + 2) + threadIdx.x] =
        (sred[0][threadIdx.x] + sred[1][threadIdx.x]) + (sred[2][threadIdx.x] + sred[3][threadIdx.x]);
}

// ---- final per-chromosome predictions with the selected tau ------------------------------------------------
// pred: [nchr][N] (compact sample order); thread = one position
__global__ __launch_bounds__(256) void k_l1_pred(L1Rows R, const double* alpha /*[K*R1][n64]*/, int R1,
                                                 int best, const int32_t* chunk_seg,
                                                 const int64_t* chunk_pos, const int64_t* chunk_len,
                                                 const int32_t* chr_col0 /*[nchr+1]*/, int nchr,
                                                 const int32_t* cidx, int64_t N, double* pred) {
  extern __shared__ double sAl[];  // [L]
  const int ch = blockIdx.x;
  const int f = chunk_seg[ch];
  const int64_t p0 = chunk_pos[ch], plen = chunk_len[ch];
  const double* al = alpha + ((int64_t)f * R1 + best) * R.n64;
  for (int c = threadIdx.x; c < R.L; c += 256) sAl[c] = al[c];
  __syncthreads();
  for (int64_t sub = 0; sub < plen; sub += 256) {
    const int64_t pos = p0 + sub + threadIdx.x;
    if (sub + threadIdx.x >= plen) continue;
    const int32_t n = cidx[pos];
    if (n < 0) continue;
    for (int c = 0; c < nchr; ++c) {
      double acc = 0.0;
      const double* w = R.W + ((int64_t)chr_col0[c] * R.P + R.p) * R.Np + pos;
      const int nn = chr_col0[c + 1] - chr_col0[c];
      for (int t = 0; t < nn; ++t) acc = fma(w[(int64_t)t * R.P * R.Np], sAl[chr_col0[c] + t], acc);
      pred[(int64_t)c * N + n] = acc;
    }
  }
}

int rg_l1_qt_impl(rg_ctx* ctx, int R1, const double* tau, int nchr, const int32_t* cols_per_chr,
                  double* cumsum_out, int32_t* best_out, double* pred_out) {
  if (!ctx->have_problem || !(ctx->d_W || ctx->v_W)) { ctx->err = "rg_l1_qt: no level-0 predictors"; return RG_ERR_STATE; }
  if (R1 < 1 || R1 > R1MAX) { ctx->err = "rg_l1_qt: n_ridge_l1 must be in [1,8]"; return RG_ERR_ARG; }
  if (ctx->loocv) { ctx->err = "rg_l1_qt: the problem was set up for LOOCV (use rg_l1_qt_loocv)"; return RG_ERR_STATE; }
  hipStream_t st = ctx->stream;
  const int L = ctx->B_total * ctx->R0, K = ctx->K;
  // phenotype view (rg_set_l1_view): local phenotype p <-> global pg = v_p0 + p; predictors read from v_W laid out
  // [L][v_np][Np] when a view buffer is set, else from the context's W [L][P][Np]
  if (!ctx->v_W && ctx->w_nb != ctx->B_total) { ctx->err = "rg_l1_qt: W holds a block range only (rg_set_block_range): level 1 needs the exchanged view"; return RG_ERR_STATE; }
  const double* Wv = ctx->v_W ? ctx->v_W : ctx->d_W;
  const int Pv = ctx->v_W ? ctx->v_np : ctx->P, P = ctx->v_np, p0v = ctx->v_p0;
  int ltot = 0;
  std::vector<int32_t> col0(nchr + 1, 0);
  for (int c = 0; c < nchr; ++c) { col0[c + 1] = col0[c] + cols_per_chr[c]; }
  ltot = col0[nchr];
  if (ltot != L) { ctx->err = "rg_l1_qt: cols_per_chr does not sum to n_blocks*R0"; return RG_ERR_ARG; }
  const int n64 = (int)rg_round_up(L, CT), rtot = n64 + CT, T = n64 / CT;
  const int64_t msz = (int64_t)rtot * n64;
  const int nsys = K * R1;
  const int nch = ctx->n_c256;
  const int NPART = R1MAX * 3 + 2;
  const int world = ctx->coll_world, rank = ctx->coll_rank;
  // a callback with world == 1 is legal: the shared form then runs with a single rank (its all-reduces are identities)
  const bool multi = ctx->coll_allreduce != nullptr;
  // systems (fold f, tau j) -> b = f*R1 + j; rank r factors the contiguous range [b0, b1)
  const int b0 = multi ? (int)((int64_t)nsys * rank / world) : 0;
  const int b1 = multi ? (int)((int64_t)nsys * (rank + 1) / world) : nsys;
  const int nloc = b1 - b0;
  // Gram work items: tiles (lower triangle + the y row tile) x K slices
  const int ntile = T * (T + 1) / 2 + T;
  int64_t min_nch = INT64_MAX;
  for (int f = 0; f < K; ++f) min_nch = std::min(min_nch, ctx->seg.plen[f] / 64);
  int nslice = (int)std::min<int64_t>(16, std::max<int64_t>(1, (4096 + (int64_t)ntile * K - 1) / ((int64_t)ntile * K)));
  nslice = (int)std::max<int64_t>(1, std::min<int64_t>(nslice, min_nch / 4));
  // LDS-staged Gram (k_l1_gram128): 128 x 128 macro tiles; K slices until the table holds several rounds of the chip's 512
  // workgroup slots (equal-cost items: the tail of the last round is what the slices even out).  RG_L1_GRAM64=1 keeps the
  // register-fed kernel (one wave per 64 x 64 tile).
  static const bool gram64 = getenv("RG_L1_GRAM64") && atoi(getenv("RG_L1_GRAM64")) != 0;
  const int T2 = (n64 + 127) / 128, ntile2 = T2 * (T2 + 1) / 2;
  if (!gram64) {
    nslice = (int)std::min<int64_t>(16, std::max<int64_t>(1, (3072 + (int64_t)ntile2 * K - 1) / ((int64_t)ntile2 * K)));
    nslice = (int)std::max<int64_t>(1, std::min<int64_t>(nslice, min_nch / 2));      // >= 8 stages of 16 positions per slice
  }

  double *d_fold = nullptr, *d_part = nullptr, *d_sum = nullptr, *d_wk = nullptr, *d_dinv = nullptr, *d_tau = nullptr,
         *d_cvp = nullptr, *d_pred = nullptr, *d_alpha = nullptr, *d_pack = nullptr;
  int32_t* d_col0 = nullptr;
#define L1_WS(var, slot, type, count)                                                   \
  var = (type*)rg_ws(ctx, slot, sizeof(type) * (size_t)(count));                        \
  if (!var) { ctx->err = "rg_l1_qt: out of device memory"; return RG_ERR_HIP; }
  L1_WS(d_fold, 0, double, msz * K)
  if (nslice > 1) { L1_WS(d_part, 1, double, msz * K * nslice) }
  L1_WS(d_sum, 2, double, msz)
  L1_WS(d_wk, 3, double, msz * std::max(1, nloc))
  L1_WS(d_dinv, 4, double, rg_chol_ws_doubles((size_t)std::max(1, nloc), n64))
  L1_WS(d_tau, 5, double, R1)
  L1_WS(d_cvp, 6, double, (size_t)nch * NPART)
  L1_WS(d_pred, 7, double, (size_t)nchr * ctx->N)
  L1_WS(d_alpha, 8, double, (size_t)nsys * n64)
  L1_WS(d_col0, 9, int32_t, nchr + 1)
  if (multi) { L1_WS(d_pack, 10, double, (size_t)ntile * K * CT * CT) }
  L1Item* d_items = nullptr;
  std::vector<L1Item> items;
  if (!gram64) {
    items = l1_build_items(n64, K, nslice, multi ? world : 1, multi ? rank : 0);
    L1_WS(d_items, 12, L1Item, items.size())
    RG_HIP(hipMemcpyAsync(d_items, items.data(), sizeof(L1Item) * items.size(), hipMemcpyHostToDevice, st));
    RG_HIP(hipStreamSynchronize(st));     // `items` is pageable host memory
  }
#undef L1_WS
  RG_HIP(hipMemcpyAsync(d_col0, col0.data(), sizeof(int32_t) * (nchr + 1), hipMemcpyHostToDevice, st));
  std::vector<double> hpart((size_t)nch * NPART);
  int rc = RG_OK;
  auto lap = [&](double* slot, hipEvent_t e0, hipEvent_t e1) {
    if (!ctx->timing) return;
    hipEventRecord(e1, st); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); *slot += ms;
    hipEventRecord(e0, st);
  };

  for (int p = 0; p < P && rc == RG_OK; ++p) {
    const int pg = p0v + p, pw = ctx->v_W ? p : pg;
    L1Rows R{Wv, ctx->d_V + (int64_t)(ctx->C + pg) * ctx->Np, ctx->d_zero, ctx->Np, L, Pv, pw, n64};
    hipEvent_t e0 = ctx->ev0, e1 = ctx->ev1;
    if (ctx->timing) hipEventRecord(e0, st);
    // ---- fold Grams X_f = W_f^T W_f with W_f^T y_f as an extra row ------------------------------------------
    double* gout = nslice > 1 ? d_part : d_fold;
    hipMemsetAsync(gout, 0, sizeof(double) * msz * K * nslice, st);
    if (gram64) {
      L1G64 g{R, T, rtot, nslice, ntile, multi ? world : 1, multi ? rank : 0, gout, msz * K};
      hipLaunchKernelGGL(k_l1_gram64, dim3((ntile * nslice + 3) / 4, K), dim3(256), 0, st, g, ctx->seg);
    } else {
      L1G128 g{R, rtot, nslice, d_items, gout, msz * K};
      hipLaunchKernelGGL(k_l1_gram128, dim3((unsigned)items.size()), dim3(256), 0, st, g, ctx->seg);
    }
    if (nslice > 1)
      hipLaunchKernelGGL(k_reduce_slices, dim3((unsigned)((msz * K + 255) / 256)), dim3(256), 0, st, d_part, msz * K,
                         nslice, msz * K, d_fold);
    if (!gram64)
      hipLaunchKernelGGL(k_l1_wty, dim3(L, K), dim3(256), 0, st, R, ctx->seg, rtot, multi ? world : 1, multi ? rank : 0, d_fold);
    if (multi) {  // every rank needs every fold matrix: sum of disjoint tile sets, exchanged tile-packed
      hipLaunchKernelGGL(k_tiles_pack, dim3(ntile, K), dim3(256), 0, st, d_fold, msz, T, n64, d_pack, 0);
      RG_HIP(hipStreamSynchronize(st));
      if (ctx->coll_allreduce(ctx->coll_user, d_pack, (int64_t)ntile * K * CT * CT) != 0) { ctx->err = "rg_l1_qt: all-reduce callback failed"; rc = RG_ERR_STATE; break; }
      hipLaunchKernelGGL(k_tiles_pack, dim3(ntile, K), dim3(256), 0, st, d_fold, msz, T, n64, d_pack, 1);
    }
    hipLaunchKernelGGL(k_sum_folds, dim3((unsigned)((msz + 255) / 256)), dim3(256), 0, st, d_fold, msz, K, d_sum);
    lap(&ctx->tm.ms_l1_gram, e0, e1);
    // ---- the K*R1 systems (sum - X_f + tau_j I) alpha = (sum - X_f^T y_f) -----------------------------------
    hipMemcpyAsync(d_tau, tau + (int64_t)p * R1, sizeof(double) * R1, hipMemcpyHostToDevice, st);
    hipMemsetAsync(d_alpha, 0, sizeof(double) * (size_t)nsys * n64, st);
    if (nloc > 0) {
      rg_launch_chol_solve_formed_x(st, d_sum, 0, d_fold, msz, K, d_tau, R1, nullptr, L, 1, d_wk, msz, n64, CT, 1,
                                    d_dinv, ctx->d_info + 1, &ctx->tm.n_chol_launches, 1, nullptr, 0, 0, 1, b0, nloc);
      RG_HIP(hipMemcpy2DAsync(d_alpha + (int64_t)b0 * n64, sizeof(double) * n64, d_wk + (int64_t)n64 * n64,
                              sizeof(double) * msz, sizeof(double) * n64, nloc, hipMemcpyDeviceToDevice, st));
    }
    if (multi) {
      RG_HIP(hipStreamSynchronize(st));
      if (ctx->coll_allreduce(ctx->coll_user, d_alpha, (int64_t)nsys * n64) != 0) { ctx->err = "rg_l1_qt: all-reduce callback failed"; rc = RG_ERR_STATE; break; }
    }
    lap(&ctx->tm.ms_l1_chol, e0, e1);
    // ---- out-of-fold predictions for every tau, the five sums (every rank: identical, deterministic) -------------
    hipLaunchKernelGGL(k_l1_cv, dim3(nch), dim3(256), 0, st, R, d_alpha, R1, ctx->d_c256_seg,
                       ctx->d_c256_pos, ctx->d_c256_len, d_cvp);
    RG_HIP(hipMemcpyAsync(hpart.data(), d_cvp, sizeof(double) * hpart.size(), hipMemcpyDeviceToHost, st));
    RG_HIP(hipStreamSynchronize(st));
    // cumsum_values (Step1_Models.cpp:854-858): fixed-order host reduction of the chunk partials
    double* cs = cumsum_out + (int64_t)p * 5 * R1;  // [5][R1] row-major per phenotype
    for (int t = 0; t < 5 * R1; ++t) cs[t] = 0.0;
    double sy = 0.0, sy2 = 0.0;
    for (int ch = 0; ch < nch; ++ch) {
      const double* q = hpart.data() + (size_t)ch * NPART;
      for (int j = 0; j < R1; ++j) {
        cs[0 * R1 + j] += q[3 * j];
        cs[2 * R1 + j] += q[3 * j + 1];
        cs[4 * R1 + j] += q[3 * j + 2];
      }
      sy += q[R1MAX * 3];
      sy2 += q[R1MAX * 3 + 1];
    }
    for (int j = 0; j < R1; ++j) { cs[1 * R1 + j] = sy; cs[3 * R1 + j] = sy2; }
    // Data.cpp:1025-1037: first minimum of (Sx2 + Sy2 - 2 Sxy) / Neff
    int best = 0; double minv = 1e10;
    for (int j = 0; j < R1; ++j) {
      const double perf = (cs[2 * R1 + j] + cs[3 * R1 + j] - 2 * cs[4 * R1 + j]) / ctx->neff[pg];
      if (perf < minv) { best = j; minv = perf; }
    }
    best_out[p] = best;
    hipMemsetAsync(d_pred, 0, sizeof(double) * (size_t)nchr * ctx->N, st);
    hipLaunchKernelGGL(k_l1_pred, dim3(nch), dim3(256), sizeof(double) * L, st, R, d_alpha, R1, best,
                       ctx->d_c256_seg, ctx->d_c256_pos, ctx->d_c256_len, d_col0, nchr,
                       ctx->d_cidx, ctx->N, d_pred);
    if ((rc = rg_emit_pred(ctx, st, d_pred, nchr, p, pred_out))) break;
    RG_HIP(hipStreamSynchronize(st));
    if (ctx->timing) { hipEventRecord(e1, st); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); ctx->tm.ms_l1_pred += ms; }
    int32_t info[2] = {0, 0};
    RG_HIP(hipMemcpy(info, ctx->d_info, sizeof(info), hipMemcpyDeviceToHost));
    if (info[1]) { ctx->err = "level 1 ridge system is not positive definite"; rc = RG_ERR_NOT_SPD; }
  }
  return rc;
}
