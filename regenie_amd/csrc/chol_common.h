// Shared pieces of the batched Cholesky kernels (chol.hip and chol_p128.h): work placement, the lazily formed systems (FormSrc), and the
// factorization + inverse of a 64 x 64 tile held in LDS.  Device code only; included once per translation unit.
#pragma once
#ifndef CT
#define CT 64
#endif

// ---- XCD-affine work order -----------------------------------------------------------------------
// MI355X dispatches workgroup id w to XCD (w % 8), each XCD with its own 4 MiB L2.  All `ngrp` work items
// of one system share that system's panel rows, so system b is pinned to one XCD and its items run back to
// back there.  R consecutive systems (the ridge shifts of one fold matrix) are pinned to the SAME XCD and
// advance item by item together, so that a first-touch tile of the shared matrix is fetched from HBM once
// and served to the other R-1 systems by that XCD's L2:
//   id w -> xcd = w % 8, s = w / 8, r = s % R, item = (s / R) % ngrp, group = s / (R * ngrp),
//   system = (group * 8 + xcd) * R + r.
// Only speed depends on the placement; the mapping is a bijection onto (system, item) for any dispatch.
__device__ __forceinline__ bool xcd_affine(int w, int ngrp, int batch, int R, int& b, int& g) {
  const int xcd = w & 7, s = w >> 3;
  const int r = s % R, t = s / R;
  g = t % ngrp;
  b = ((t / ngrp) * 8 + xcd) * R + r;
  return b < batch;
}
__host__ __device__ static inline unsigned xcd_affine_grid(int ngrp, int batch, int R) {
  return (unsigned)(((batch + 8 * R - 1) / (8 * R)) * 8 * R * ngrp);
}

// ---- lazy "form": value of element (i,j) of system b = sum[o] - fold[o][f] + shift[r] on the diagonal ----
// Every tile of a system is first touched exactly once during the first column group of the
// factorization, so the systems are never materialised by a separate pass: the first-touch kernels
// read (sum, fold) instead of the workspace.
// `extra`: rows >= extra_row0 of every system come from a shared matrix (per outer index) instead of
// sum/fold -- the LOOCV paths append the sample-major predictor rows there so that the factorization
// forward-substitutes them (z_i = L^-1 x_i gives the LOO leverages ||z_i||^2 without any inverse).
// `subtract` = 0: the system is sum + shift (no held-out fold), used by LOOCV.
struct FormSrc {
  const double* sum; int64_t sum_stride;
  const double* fold; int64_t fold_stride;
  const double* shift; const int32_t* d_n;
  const double* extra; int64_t extra_stride;
  int nfold, nshift, n_fixed, enabled, subtract, extra_row0, n64, n_div, b_offset;
  int skip_pad;   // group-wise path: tile rows / columns past a system's own order (identity padding) are not computed
  int embed;      // > 0: that many right-hand sides sit in rows n .. n + embed - 1 of the system (needs d_n; group-wise path only)
};
#define RG_EMBED_DIAG 0x1p100   // diagonal of an embedded right-hand-side row: D - y^T y stays positive and sqrt(D) = 2^50 overflows nothing
// Tile columns of system b that hold data: ceil(n_b / 64) when the caller gave per-system orders (level 0: the SNP count of
// the block, so that a chromosome-end block of 300 SNPs is factored at order 320 instead of the batch's 1024), else T.
// Tile rows [T_b, T) and tile columns >= T_b of such a system are identity padding: never read, never written.
__device__ __forceinline__ int sys_tiles(const FormSrc& f, int b_local, int T) {
  if (!f.skip_pad || !f.d_n) return T;
  const int b = b_local + f.b_offset;
  const int o = b / (f.nfold * f.nshift);
  const int tb = (f.d_n[o / f.n_div] + f.embed + CT - 1) / CT;
  return tb < T ? tb : T;
}
struct FormIdx { const double* S; const double* F; const double* X; double sh; int64_t xoff; int n, x0, nrhs; };
__device__ __forceinline__ FormIdx form_idx(const FormSrc& f, int b_local) {
  const int b = b_local + f.b_offset;   // a rank / caller may own a contiguous sub-range of the systems
  const int per = f.nfold * f.nshift;
  const int o = b / per, rem = b % per, fo = rem / f.nshift, r = rem % f.nshift;
  FormIdx x;
  x.S = f.sum + (int64_t)o * f.sum_stride;
  x.F = f.subtract ? f.fold + ((int64_t)o * f.nfold + fo) * f.fold_stride : nullptr;
  x.X = f.extra ? f.extra + (int64_t)o * f.extra_stride : nullptr;
  x.x0 = f.extra ? f.extra_row0 : 0x7fffffff;
  x.xoff = (int64_t)f.extra_row0 * f.n64;
  x.sh = f.shift[r];
  x.n = f.d_n ? f.d_n[o / f.n_div] : f.n_fixed;
  x.nrhs = x.n + f.embed;
  return x;
}
// e = i * n64 + j.  MODE is resolved OUTSIDE the unrolled element loops: hipcc turns every load under an `if` into
// branch + load + s_waitcnt vmcnt(0) -- one full memory round trip per element -- whereas straight-line loads of an
// unrolled loop are all in flight together.  MODE 0: S only (one source matrix); 1: S - F; 2: general (extra rows).
template <int MODE>
__device__ __forceinline__ double form_val(const FormIdx& x, int i, int j, int64_t e) {
  double v;
  if (MODE == 3) return x.X[e - x.xoff];       // a tile that lies entirely in the extra rows (never on the diagonal)
  if (MODE == 2) {
    if (i >= x.x0) return x.X[e - x.xoff];   // (i - extra_row0) * n64 + j
    v = x.S[e];
    if (x.F) v -= x.F[e];
  } else if (MODE == 1) {
    v = x.S[e] - x.F[e];
  } else {
    v = x.S[e];
  }
  if (i == j) v = (i < x.n) ? v + x.sh : (i < x.nrhs ? RG_EMBED_DIAG : 1.0);
  return v;
}
__device__ __forceinline__ int form_mode(const FormIdx& x) { return x.X ? 2 : (x.F ? 1 : 0); }
// Mode of a tile whose rows are [row_lo, row_hi): the extra rows start at a tile boundary in every caller, so a tile is
// either all matrix / right-hand-side rows (modes 0, 1: straight-line loads) or all extra rows (mode 3); the per-element
// branch of mode 2 -- one serialized memory round trip per element -- is only a fallback for a straddling tile.
__device__ __forceinline__ int form_mode_rows(const FormIdx& x, int row_lo, int row_hi) {
  if (!x.X || row_hi <= x.x0) return x.F ? 1 : 0;
  return row_lo >= x.x0 ? 3 : 2;
}

// ---- diagonal tile: blocked (16) potf2 + blocked triangular inverse, all in LDS -------------------
// value of `x` in lane `l` (compile-time constant), broadcast through SGPRs
__device__ __forceinline__ double bcast_lane(double x, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(x), l);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(x), l);
  return __hiloint2double(hi, lo);
}

// ---- blocked (16) factorization + inverse of the 64x64 tile held in LDS (lower triangle of s), 256 threads -------
// On return: lower triangle + diagonal of s = L, strict upper triangle = Linv^T, dv[r] = Linv[r][r].
// Returns true (in some thread) when a pivot was not positive.
__device__ __forceinline__ bool diag_factor_lds(double (&s)[CT][CT + 2], double (&dv)[CT]) {
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lq = lane >> 4;
  // ---- blocked (16) factorization + inverse, the 16x16x16 block products on the fp64 MFMA ---------------
  // MFMA 16x16x4: lane (i = lane&15, q = lane>>4) supplies A[i][kk], B[kk][i] and owns D[q + 4r][i], r = 0..3;
  // a K = 16 block product is 4 instructions with kk(q, s) chosen per product (any permutation of K is fine
  // as long as A and B use the same one).
  // element (r, c) of the inverse of a DIAGONAL 16-block at offset o (0 above the diagonal)
  auto inv_diag = [&](int o, int r, int c) -> double {
    return (c < r) ? s[o + c][o + r] : ((c == r) ? dv[o + r] : 0.0);
  };
  bool bad = false;
  for (int sb = 0; sb < 4; ++sb) {
    const int o = sb * 16;
    const int nb = 3 - sb;   // 16-row blocks below the diagonal block
    // (i) diagonal block: lanes 0..15 of wave 0 hold one row each in registers; then its inverse, one column each
    if (tid < 16) {
      // cross-lane values are broadcast with v_readlane (compile-time lane index, no LDS round trip); square root
      // and reciprocal come from one v_rsq_f64 + two Newton steps (~1 ulp), far shorter than sqrt() followed by a division
      double a[16], rdv[16];
#pragma unroll
      for (int c = 0; c < 16; ++c) a[c] = s[o + tid][o + c];
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const double piv = bcast_lane(a[c], c);
        double d, rd;
        if (piv > 0.0) {
          double r0 = __builtin_amdgcn_rsq(piv);
          r0 = r0 * fma(-0.5 * piv * r0, r0, 1.5);
          r0 = r0 * fma(-0.5 * piv * r0, r0, 1.5);
          d = piv * r0;
          d = fma(0.5 * r0, fma(-d, d, piv), d);     // sqrt(piv)
          rd = fma(r0, fma(-d, r0, 1.0), r0);        // 1 / sqrt(piv)
        } else { d = 1.0; rd = 1.0; bad = true; }
        rdv[c] = rd;
        if (tid > c) a[c] *= rd;
        else if (tid == c) a[c] = d;
#pragma unroll
        for (int c2 = c + 1; c2 < 16; ++c2) {
          const double l = bcast_lane(a[c], c2);  // L[c2][c]
          if (tid >= c2) a[c2] = fma(-a[c], l, a[c2]);
        }
      }
#pragma unroll
      for (int c = 0; c < 16; ++c)
        if (c <= tid) s[o + tid][o + c] = a[c];
      // inverse of the 16x16 triangle: lane = column, forward substitution; L[r][j] comes from lane r's registers
      double x[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        double v = (r == tid) ? 1.0 : 0.0;
#pragma unroll
        for (int j = 0; j < r; ++j) v = fma(-bcast_lane(a[j], r), x[j], v);
        x[r] = v * rdv[r];
      }
      dv[o + tid] = x[tid];
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (r > tid) s[o + tid][o + r] = x[r];   // Linv[r][tid], transposed into the upper triangle
    }
    __syncthreads();
    // (ii) rows below: L21 = A21 * Linv11^T, one 16-row block per wave
    if (wave < nb) {
      const int rb = o + 16 + 16 * wave;
      v4d acc = (v4d){0, 0, 0, 0};
#pragma unroll
      for (int st = 0; st < 4; ++st)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(s[rb + li][o + 4 * lq + st], inv_diag(o, li, 4 * lq + st), acc, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) s[rb + lq + 4 * r][o + li] = acc[r];
    }
    __syncthreads();
    // (iii) trailing update inside the tile: A22 -= L21 L21^T (lower blocks; only the lower triangle of the
    //       diagonal blocks is written -- their upper triangle will hold the inverse), blocks dealt to the waves
    {
      const int nblk2 = nb * (nb + 1) / 2;
      for (int idx = wave; idx < nblk2; idx += 4) {
        int bi = 0, rem = idx;
        while (rem > bi) { rem -= bi + 1; ++bi; }
        const int bj = rem;
        const int ri = o + 16 + 16 * bi, rj = o + 16 + 16 * bj;
        v4d acc = (v4d){0, 0, 0, 0};
#pragma unroll
        for (int st = 0; st < 4; ++st)
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(s[ri + li][o + 4 * lq + st], s[rj + li][o + 4 * lq + st], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (bi != bj || li <= lq + 4 * r) s[ri + lq + 4 * r][rj + li] -= acc[r];
      }
    }
    __syncthreads();
  }
  // off-diagonal blocks of the inverse (i > j), by sub-diagonal distance:
  //   Linv[i][j] = -Linv[i][i] * sum_{kb=j}^{i-1} L[i][kb] Linv[kb][j]      (Linv[kb][j] at s[16j + .][16kb + .]^T)
  for (int dist = 1; dist < 4; ++dist) {
    const int j = wave, ib = wave + dist;
    if (ib < 4) {
      v4d m1 = (v4d){0, 0, 0, 0};
      for (int kb = j; kb < ib; ++kb) {
#pragma unroll
        for (int st = 0; st < 4; ++st) {
          const int kk = 4 * lq + st;
          const double bval = (kb == j) ? inv_diag(16 * j, kk, li) : s[16 * j + li][16 * kb + kk];
          m1 = __builtin_amdgcn_mfma_f64_16x16x4f64(s[16 * ib + li][16 * kb + kk], bval, m1, 0, 0, 0);
        }
      }
      // second product with kk(q, st) = q + 4 st: the B operand M1[kk][li] is exactly register st of m1
      v4d acc = (v4d){0, 0, 0, 0};
#pragma unroll
      for (int st = 0; st < 4; ++st)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(inv_diag(16 * ib, li, lq + 4 * st), m1[st], acc, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) s[16 * j + li][16 * ib + lq + 4 * r] = -acc[r];
    }
    __syncthreads();
  }
  return bad;
}

