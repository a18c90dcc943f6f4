// Level-0 out-of-fold predictions, their column standardisation, and W gather/scatter helpers.
//
// Reference: src/Step1_Models.cpp:496-511 (pred = beta^T G_fold masked, running sums) and :539-571
// (centre/scale with mean = p_sum/Neff, invsd = sqrt((Neff-1)/(p_sum2 - Neff mean^2)), applied to
// ALL rows of the fold blocks, masked rows included).
// beta^T G is evaluated on the raw dosages:  beta^T G = (D_s^-1 beta)^T G~ - ((D_s^-1 beta)^T B) X^T
// with G~ = G0 + D_mu M decoded on the fly from the cleaned 2-bit rows, so the standardised
// genotype matrix is never materialised (the reference allocates bs x N doubles per block).
#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include "rg_internal.h"

// ---- beta~ = x / scale_G and cb = beta~^T B --------------------------------------------------------
// grid (nseg*R0, nblk); the solutions x sit in the RHS rows of the factored systems.
__global__ __launch_bounds__(256) void k_beta_post(PredArgs a) {
  __shared__ double red[4];
  const int blk = blockIdx.y, m = blockIdx.x;
  const int bs = a.bs[blk];
  const int nm = a.nseg * a.R0;
  const int64_t msz = (int64_t)a.rtot * a.n64;
  const double* M = a.wk + ((int64_t)blk * nm + m) * msz;
  const double* sc = a.sc + (int64_t)blk * a.n128;
  const double* Bm = a.Bm + (int64_t)blk * a.n128 * a.C;
  for (int p = 0; p < a.P; ++p) {
    double* beta = a.beta + (((int64_t)blk * nm + m) * a.P + p) * a.n64;
    const double* x = M + (int64_t)((a.embed ? bs : a.n64) + p) * a.n64;
    for (int j = threadIdx.x; j < a.n64; j += 256) beta[j] = (j < bs) ? x[j] / sc[j] : 0.0;
    for (int c = 0; c < a.C; ++c) {
      double t = 0.0;
      for (int j = threadIdx.x; j < bs; j += 256) t = fma(x[j] / sc[j], Bm[(int64_t)j * a.C + c], t);
      for (int o = 32; o > 0; o >>= 1) t += __shfl_down(t, o);
      __syncthreads();
      if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = t;
      __syncthreads();
      if (threadIdx.x == 0)
        a.cb[(((int64_t)blk * nm + m) * a.P + p) * a.C + c] = (red[0] + red[1]) + (red[2] + red[3]);
    }
  }
}

// ---- predictions -----------------------------------------------------------------------------------------------
// grid (nchunk, P, nblk); 256 threads, thread = 4 consecutive positions (one packed byte column).
#define RMAX 8
#define JT 128

// NR = number of ridge values carried in registers (5 for the default grid, 8 max).
// The packed genotype tile (JT SNP rows x 1024 positions = 256 bytes per row) is staged through LDS
// with coalesced 16-byte loads so the inner loop never waits on HBM; the inner loop per SNP row is
// decode (2 ops per sample) + NR FMAs per sample; the coefficients of the tile are staged in LDS next to it and read
// back as broadcasts.
template <int NR>
__global__ __launch_bounds__(256) void k_l0_pred(PredArgs a, ChunkTab ct) {
  __shared__ __attribute__((aligned(16))) uint8_t sP[JT][256 + 16];
  __shared__ __attribute__((aligned(16))) double sB[JT][NR + 1];   // coefficients of the tile's SNPs, then their means
  __shared__ double sred[4][RMAX][2];
  const int blk = blockIdx.z, p = blockIdx.y, ch = blockIdx.x;
  const int bs = a.bs[blk];
  const int s = ct.seg[ch];
  const int64_t p0 = ct.pos[ch], plen = ct.len[ch];
  const int R0 = a.R0;
  const int nm = a.nseg * R0;
  const uint8_t* __restrict__ pk = a.pk + (int64_t)blk * a.pk_blk_stride;
  const double* __restrict__ mu = a.mu + (int64_t)blk * a.n128;
  const bool has_miss = a.nmiss[blk] > 0;
  const double* __restrict__ bp[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r)
    bp[r] = a.beta + (((int64_t)blk * nm + s * R0 + (r < R0 ? r : 0)) * a.P + p) * a.n64;
  const int col0 = a.blockid[blk] * R0;
  double tsum[NR], tsq[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) tsum[r] = tsq[r] = 0.0;

  for (int64_t sub = 0; sub < plen; sub += 1024) {
    const int64_t pos = p0 + sub + 4 * (int64_t)threadIdx.x;
    const bool live = (sub + 4 * (int64_t)threadIdx.x) < plen;
    const int64_t nbytes = min((int64_t)256, (plen - sub) / 4);  // valid packed bytes per row in this tile
    double acc[4][NR];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < NR; ++r) acc[i][r] = 0.0;
    for (int jt = 0; jt < bs; jt += JT) {
      __syncthreads();
      {
        const int c16 = (threadIdx.x & 15) * 16;
#pragma unroll
        for (int it = 0; it < JT / 16; ++it) {
          const int row = it * 16 + (threadIdx.x >> 4);
          // unconditional load at a clamped address + select (a load under `if` costs a full round trip each)
          const bool ok = (jt + row < bs) && (c16 < nbytes);
          uint4 v = *reinterpret_cast<const uint4*>(pk + (int64_t)min(jt + row, bs - 1) * a.pk_ld + (p0 + sub) / 4 +
                                                    (c16 < nbytes ? c16 : 0));
          if (!ok) v = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
          *reinterpret_cast<uint4*>(&sP[row][c16]) = v;
        }
        // the tile's coefficients (and means): uniform across the wave, so every lane reads the same LDS word
        // (broadcast).  Left to itself hipcc fetched them with per-SNP vector loads followed by vmcnt(0).
        for (int e = threadIdx.x; e < JT * (NR + 1); e += 256) {
          const int r = e / JT, t = e - r * JT;     // beta is zero-padded beyond bs (n64 entries), mu has n128
          const int j = min(jt + t, a.n64 - 1);     // clamped: rows beyond bs decode to 0 and are never multiplied
          const double v = r < NR ? bp[r < R0 ? r : 0][j] : mu[j];
          sB[t][r] = jt + t < bs ? v : 0.0;
        }
      }
      __syncthreads();
      if (live) {
        const int jn = (min(JT, bs - jt) + 3) & ~3;
        auto run = [&](auto miss) {
#pragma unroll 1
          for (int t0 = 0; t0 < jn; t0 += 4) {
            unsigned bb[4];
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) bb[tt] = sP[t0 + tt][threadIdx.x];
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
              const int t = t0 + tt;
              const unsigned b = bb[tt];
              const unsigned lo = b & 0x55u, hi = (b >> 1) & 0x55u;
              const unsigned dd = (hi & ~lo) | ((~(hi | lo) & 0x55u) << 1);   // 2-bit dosage fields
              double g[4];
#pragma unroll
              for (int i = 0; i < 4; ++i) g[i] = (double)((dd >> (2 * i)) & 3u);
              if (decltype(miss)::value) {
                const unsigned ms = lo & ~hi;
                const double m = sB[t][NR];
#pragma unroll
                for (int i = 0; i < 4; ++i) g[i] = fma((double)((ms >> (2 * i)) & 1u), m, g[i]);
              }
#pragma unroll
              for (int r = 0; r < NR; ++r) {
                const double bt = sB[t][r];
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i][r] = fma(g[i], bt, acc[i][r]);
              }
            }
          }
        };
        if (has_miss) run(std::true_type{});
        else run(std::false_type{});
      }
    }
    if (live) {
      // covariate term: sum_c cb_r[c] X_c(pos).  The X values are loaded once per covariate chunk (unconditional,
      // clamped) and reused by every ridge value; cb_r[c] is wave-uniform (scalar loads)
      double corr[NR][4];
#pragma unroll
      for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int i = 0; i < 4; ++i) corr[r][i] = 0.0;
      for (int c0 = 0; c0 < a.C; c0 += 4) {
        double xv[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const double* x = a.V + (int64_t)min(c0 + u, a.C - 1) * a.Np + pos;
#pragma unroll
          for (int i = 0; i < 4; ++i) xv[u][i] = x[i];
        }
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          const double* cb = a.cb + (((int64_t)blk * nm + s * R0 + (r < R0 ? r : 0)) * a.P + p) * a.C;
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const double cc = cb[min(c0 + u, a.C - 1)] * ((c0 + u < a.C) ? 1.0 : 0.0);
#pragma unroll
            for (int i = 0; i < 4; ++i) corr[r][i] = fma(cc, xv[u][i], corr[r][i]);
          }
        }
      }
      const double* mk = a.maskp + (int64_t)p * a.Np + pos;
      double mkv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) mkv[i] = mk[i];
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        if (r >= R0) break;
        double* w = a.W + ((int64_t)(col0 + r) * a.P + p) * a.Np + pos;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const double v = (acc[i][r] - corr[r][i]) * mkv[i];
          w[i] = v;
          tsum[r] += v;
          tsq[r] = fma(v, v, tsq[r]);
        }
      }
    }
  }
  // block reduction of the running sums -> per-chunk partials
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    double x = tsum[r], y = tsq[r];
    for (int o = 32; o > 0; o >>= 1) {
      x += __shfl_down(x, o);
      y += __shfl_down(y, o);
    }
    if ((threadIdx.x & 63) == 0) {
      sred[threadIdx.x >> 6][r][0] = x;
      sred[threadIdx.x >> 6][r][1] = y;
    }
  }
  __syncthreads();
  if (threadIdx.x < R0 * 2) {
    const int r = threadIdx.x >> 1, q = threadIdx.x & 1;
    a.psum[((((int64_t)blk * ct.n + ch) * a.P + p) * RMAX + r) * 2 + q] =
        (sred[0][r][q] + sred[1][r][q]) + (sred[2][r][q] + sred[3][r][q]);
  }
}

// ---- predictions on the fp64 matrix cores ------------------------------------------------------------------
// out[m][pos] = sum_j beta_m[j] g~_j(pos), rows m = (phenotype, ridge value) pairs: an (R0*P) x bs x N contraction.
// v_mfma_f64_16x16x4: A = beta (16 rows m x 4 SNPs), B = genotypes decoded on the fly (4 SNPs x 16 positions: ONE
// decode per lane and K step, shared by all MB row blocks).  A wave owns 64 positions x MB*16 rows; the K loop runs in
// super-steps of 16 SNPs, lane (i, q) taking SNPs j0 + 4q .. 4q+3 (one 32-byte beta load per row block, four 16-byte
// packed-row loads), double-buffered so the loads of the next super-step fly under the MFMAs of the current one.
// With P phenotypes the decode is amortised over R0*P rows (the VALU kernel above re-decodes per phenotype):
// 10 phenotypes at 500,000 samples: 3.9 -> ~1 ms per block.
// grid (n_c256 chunks, phenotype groups, nblk); 256 threads = 4 waves x 64 positions.
#define PG_MAX 64
template <int MB>
__global__ __launch_bounds__(256, 2) void k_l0_pred_mfma(PredArgs a, ChunkTab ct, int pg /*phenotypes per group*/) {
  constexpr int PJ = 64;                              // SNPs per LDS tile of coefficients
  __shared__ double sred[4][MB * 16][2];
  __shared__ __attribute__((aligned(16))) double sBe[MB * 16][PJ + 2];
  const int blk = blockIdx.z, ch = blockIdx.x, p0 = blockIdx.y * pg;
  const int npg = min(pg, a.P - p0);
  const int nrow = npg * a.R0;                       // live rows of this group, m = pl * R0 + r
  const int bs = a.bs[blk];
  const int s = ct.seg[ch];
  const int64_t cpos = ct.pos[ch];
  const int64_t clen = ct.len[ch];                   // multiple of 64, <= 256
  const int R0 = a.R0, nm = a.nseg * R0;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, q = lane >> 4;
  const bool wlive = (int64_t)wave * 64 < clen;      // wave-uniform
  const int64_t pos0 = cpos + (wlive ? wave * 64 : 0);
  const bool has_miss = a.nmiss[blk] > 0;
  const uint8_t* __restrict__ pk = a.pk + (int64_t)blk * a.pk_blk_stride + pos0 / 4;
  const double* __restrict__ mu = a.mu + (int64_t)blk * a.n128;
  // A operand: the coefficient rows m = pl*R0 + r of the group, staged per tile of PJ SNPs in LDS (shared by the four
  // waves, which differ only in their positions; rows beyond nrow are zero).  Register double-buffering of the
  // coefficients next to the 128 accumulator registers spilled, and every wave fetched the same rows from L2.
  const int sh = 8 * (i >> 2) + 2 * (i & 3);         // bit offset of this lane's sample inside each dword of a 16-byte row piece
  v4d acc[MB][4];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) acc[mb][nb] = (v4d){0, 0, 0, 0};
  const int nsup = (bs + 15) / 16;                    // rows >= bs decode to 0 and have beta = 0 (n64 >= 16*nsup)
  struct Stage { uint4 g[4]; double4 mu4; };
  auto load = [&](Stage& t, int su) {
    const int j = su * 16 + 4 * q;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) t.g[ks] = *reinterpret_cast<const uint4*>(pk + (int64_t)min(j + ks, a.n128 - 1) * a.pk_ld);
    t.mu4 = *reinterpret_cast<const double4*>(mu + min(j, a.n128 - 4));
  };
  auto compute = [&](const Stage& t, int sl /* super-step inside the LDS tile */) {
    double4 be[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) be[mb] = *reinterpret_cast<const double4*>(&sBe[mb * 16 + i][sl * 16 + 4 * q]);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const unsigned wv[4] = {t.g[ks].x, t.g[ks].y, t.g[ks].z, t.g[ks].w};
      const double muj = ks == 0 ? t.mu4.x : (ks == 1 ? t.mu4.y : (ks == 2 ? t.mu4.z : t.mu4.w));
      double bv[4];
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) {
        const unsigned w = wv[nb];
        const unsigned lo = w & 0x55555555u, hi = (w >> 1) & 0x55555555u;
        const unsigned dd = (hi & ~lo) | ((~(hi | lo) & 0x55555555u) << 1);     // 2-bit dosage fields
        double g = (double)((dd >> sh) & 3u);
        if (has_miss) g = fma((double)(((lo & ~hi) >> sh) & 1u), muj, g);       // missing call -> SNP mean
        bv[nb] = g;
      }
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        const double av = ks == 0 ? be[mb].x : (ks == 1 ? be[mb].y : (ks == 2 ? be[mb].z : be[mb].w));
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) acc[mb][nb] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv[nb], acc[mb][nb], 0, 0, 0);
      }
    }
  };
  {
    Stage t0, t1;
    if (wlive) load(t0, 0);
    for (int jt = 0; jt < bs; jt += PJ) {
      __syncthreads();
      // stage the tile: MB*16 rows x PJ coefficients, coalesced along the SNP index (n64 >= jt + PJ: both multiples of 64)
      for (int e = threadIdx.x; e < MB * 16 * (PJ / 2); e += 256) {
        const int m = e / (PJ / 2), c2 = (e - m * (PJ / 2)) * 2;
        const int mc = min(m, nrow - 1);
        const int pl = mc / R0, r = mc - pl * R0;
        const double2 v = *reinterpret_cast<const double2*>(
            a.beta + (((int64_t)blk * nm + s * R0 + r) * a.P + p0 + pl) * a.n64 + jt + c2);
        const double mk = m < nrow ? 1.0 : 0.0;
        sBe[m][c2] = v.x * mk;
        sBe[m][c2 + 1] = v.y * mk;
      }
      __syncthreads();
      if (wlive) {
        const int su0 = jt / 16, sun = min(PJ / 16, nsup - su0);   // super-steps of this tile
        for (int sl = 0; sl < sun; sl += 2) {
          const int su = su0 + sl;
          if (su + 1 < nsup) load(t1, su + 1);
          compute(t0, sl);
          if (su + 2 < nsup) load(t0, su + 2);
          if (sl + 1 < sun) compute(t1, sl + 1);
        }
      }
    }
  }
  // ---- epilogue: covariate term, mask, store, per-row sums --------------------------------------------------------
  // lane holds rows m = mb*16 + q + 4r', positions pos0 + nb*16 + i
  const int col0 = a.blockid[blk] * R0;
  double xs[4][4];   // X_c at the lane's 4 positions, chunk of 4 covariates
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int m = mb * 16 + q + 4 * rr;
      const int mc = min(m, nrow - 1);
      const int pl = mc / R0, r = mc % R0;
      const bool live = wlive && (m < nrow);
      const double* cb = a.cb + (((int64_t)blk * nm + s * R0 + r) * a.P + p0 + pl) * a.C;
      double corr[4] = {0, 0, 0, 0};
      for (int c0 = 0; c0 < a.C; c0 += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int cc = min(c0 + u, a.C - 1);
          const double cv = cb[cc] * ((c0 + u < a.C) ? 1.0 : 0.0);
#pragma unroll
          for (int nb = 0; nb < 4; ++nb) {
            xs[u][nb] = a.V[(int64_t)cc * a.Np + pos0 + nb * 16 + i];
            corr[nb] = fma(cv, xs[u][nb], corr[nb]);
          }
        }
      }
      const double* mk = a.maskp + (int64_t)(p0 + pl) * a.Np + pos0 + i;
      double* w = a.W + ((int64_t)(col0 + r) * a.P + p0 + pl) * a.Np + pos0 + i;
      double sx = 0.0, sq = 0.0;
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) {
        const double v = (acc[mb][nb][rr] - corr[nb]) * mk[nb * 16] * (live ? 1.0 : 0.0);
        if (live) w[nb * 16] = v;
        sx += v;
        sq = fma(v, v, sq);
      }
      // reduce over the 16 lanes i of this q group
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) {
        sx += __shfl_xor(sx, o, 16);
        sq += __shfl_xor(sq, o, 16);
      }
      if (i == 0) { sred[wave][m][0] = sx; sred[wave][m][1] = sq; }
    }
  __syncthreads();
  if (threadIdx.x < nrow * 2) {
    const int m = threadIdx.x >> 1, t = threadIdx.x & 1;
    const int pl = m / R0, r = m % R0;
    a.psum[((((int64_t)blk * ct.n + ch) * a.P + p0 + pl) * RMAX + r) * 2 + t] =
        (sred[0][m][t] + sred[1][m][t]) + (sred[2][m][t] + sred[3][m][t]);
  }
}

// ---- column statistics: mean and 1/sd of every (block, phenotype, ridge value) column, once ---------------------------
// stats: [nblk][P][RMAX][2]; fixed-order reduction of the chunk partials (deterministic)
// one 64-lane wave per (block, phenotype, ridge value): lane-strided partial sums over the position chunks, then a fixed-order
// shuffle tree (the value does not depend on the launch); a single thread walking ~2,000 chunks with two dependent loads each was
// 0.8 ms per batch at 500,000 samples
__global__ __launch_bounds__(64) void k_l0_stats(PredArgs a, int nchunk, double* stats) {
  const int blk = blockIdx.y, t = blockIdx.x;
  const int p = t / a.R0, r = t % a.R0;
  double sx = 0.0, sq = 0.0;
  for (int ch = threadIdx.x; ch < nchunk; ch += 64) {
    const double2 q = *reinterpret_cast<const double2*>(a.psum + ((((int64_t)blk * nchunk + ch) * a.P + p) * RMAX + r) * 2);
    sx += q.x;
    sq += q.y;
  }
  for (int o = 32; o > 0; o >>= 1) { sx += __shfl_down(sx, o); sq += __shfl_down(sq, o); }
  if (threadIdx.x != 0) return;
  const double neff = a.neff[p];
  const double mean = sx / neff;
  double* o = stats + (((int64_t)blk * a.P + p) * RMAX + r) * 2;
  o[0] = mean;
  o[1] = sqrt((neff - 1.0) / (sq - neff * mean * mean));
}

// centre / scale one predictor row in place: 32 bytes per thread, both loads of the lane before its stores (a load that follows a
// store through the same pointer waits for it)
__global__ __launch_bounds__(256) void k_l0_scale(PredArgs a, const double* stats) {
  const int blk = blockIdx.z, r = blockIdx.y % a.R0, p = blockIdx.y / a.R0;
  const double* o = stats + (((int64_t)blk * a.P + p) * RMAX + r) * 2;
  const double mean = o[0], invsd = o[1];
  double* w = a.W + ((int64_t)(a.blockid[blk] * a.R0 + r) * a.P + p) * a.Np;
  const int64_t pos = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (pos >= a.Np) return;
  const double2 x0 = *reinterpret_cast<const double2*>(w + pos), x1 = *reinterpret_cast<const double2*>(w + pos + 2);
  const uint32_t kp = *reinterpret_cast<const uint32_t*>(a.keptp + pos);
  double2 y0, y1;
  y0.x = (kp & 0xFFu) ? (x0.x - mean) * invsd : 0.0;
  y0.y = (kp & 0xFF00u) ? (x0.y - mean) * invsd : 0.0;
  y1.x = (kp & 0xFF0000u) ? (x1.x - mean) * invsd : 0.0;
  y1.y = (kp & 0xFF000000u) ? (x1.y - mean) * invsd : 0.0;
  *reinterpret_cast<double2*>(w + pos) = y0;
  *reinterpret_cast<double2*>(w + pos + 2) = y1;
}

void rg_launch_l0_pred_impl(hipStream_t st, const PredArgs& a, const ChunkTab& c1k, const ChunkTab& c256, double* stats) {
  hipLaunchKernelGGL(k_beta_post, dim3(a.nseg * a.R0, a.nblk), dim3(256), 0, st, a);
  const int pg = std::max(1, std::min(a.P, PG_MAX / a.R0));      // phenotypes per group: pg * R0 <= 64 rows
  const int ngrp = (a.P + pg - 1) / pg;
  const int mb = (std::min(a.P, pg) * a.R0 + 15) / 16;
  int nchunk;
  if (mb <= 1 && a.R0 <= RMAX) {
    // few rows (one phenotype): the VALU kernel with scalar-operand coefficients is faster than a 16-row MFMA block
    nchunk = c1k.n;
    if (a.R0 <= 5) hipLaunchKernelGGL(k_l0_pred<5>, dim3(c1k.n, a.P, a.nblk), dim3(256), 0, st, a, c1k);
    else hipLaunchKernelGGL(k_l0_pred<8>, dim3(c1k.n, a.P, a.nblk), dim3(256), 0, st, a, c1k);
  } else if (mb >= 2 && a.bplanes && a.n128 <= 1024 && !getenv("RG_PRED_F64")) {
    // many rows: exact fixed-point split of the coefficients on the i8 matrix cores (pred_i8.hip); RG_PRED_F64=1 keeps fp64
    nchunk = c256.n;
    rg_launch_l0_pred_i8(st, a, c256, pg, ngrp, a.bplanes, a.bsc, a.pkT);
  } else {
    nchunk = c256.n;
    const dim3 grid(c256.n, ngrp, a.nblk);
    if (mb <= 1) hipLaunchKernelGGL(k_l0_pred_mfma<1>, grid, dim3(256), 0, st, a, c256, pg);
    else if (mb == 2) hipLaunchKernelGGL(k_l0_pred_mfma<2>, grid, dim3(256), 0, st, a, c256, pg);
    else if (mb == 3) hipLaunchKernelGGL(k_l0_pred_mfma<3>, grid, dim3(256), 0, st, a, c256, pg);
    else hipLaunchKernelGGL(k_l0_pred_mfma<4>, grid, dim3(256), 0, st, a, c256, pg);
  }
  hipLaunchKernelGGL(k_l0_stats, dim3(a.P * a.R0, a.nblk), dim3(64), 0, st, a, nchunk, stats);
  hipLaunchKernelGGL(k_l0_scale, dim3((unsigned)((a.Np / 4 + 255) / 256), a.R0 * a.P, a.nblk),
                     dim3(256), 0, st, a, (const double*)stats);
}

// ---- W gather / scatter between position space and the reference's N x R0 column-major slab ---------
__global__ void k_w_gather(const double* W, int64_t Np, int P, int p, int col0, int R0,
                           const int64_t* posc, int64_t N, double* out) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int r = blockIdx.y;
  if (n >= N) return;
  out[(int64_t)r * N + n] = W[((int64_t)(col0 + r) * P + p) * Np + posc[n]];
}
__global__ void k_w_scatter(double* W, int64_t Np, int P, int p, int col0, int R0,
                            const int64_t* posc, int64_t N, const double* in) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int r = blockIdx.y;
  if (n >= N) return;
  W[((int64_t)(col0 + r) * P + p) * Np + posc[n]] = in[(int64_t)r * N + n];
}
void rg_launch_w_gather(hipStream_t st, const double* W, int64_t Np, int P, int p, int col0, int R0,
                        const int64_t* posc, int64_t N, double* out) {
  hipLaunchKernelGGL(k_w_gather, dim3((unsigned)((N + 255) / 256), R0), dim3(256), 0, st, W, Np, P, p,
                     col0, R0, posc, N, out);
}
void rg_launch_w_scatter(hipStream_t st, double* W, int64_t Np, int P, int p, int col0, int R0,
                         const int64_t* posc, int64_t N, const double* in) {
  hipLaunchKernelGGL(k_w_scatter, dim3((unsigned)((N + 255) / 256), R0), dim3(256), 0, st, W, Np, P,
                     p, col0, R0, posc, N, in);
}
