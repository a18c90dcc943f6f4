// Level-0 leave-one-out path (reference src/Data.cpp:755-767 + src/Step1_Models.cpp:615-726).
//
// Reference: A = G G^T = V D V^T once per block, then per sample i and ridge value r
//     h_ir = sum_k (V^T g_i)_k^2 / (d_k + lambda_r),
//     pred_irp = ( sum_k (V^T g_i)_k (V^T G Y)_kp / (d_k + lambda_r) - h_ir y_ip ) / (1 - h_ir).
// Here, without any eigendecomposition: with A + lambda_r I = L_r L_r^T,
//     h_ir = || L_r^-1 g_i ||^2,     numerator = (L_r^-1 g_i) . (L_r^-1 b_p),
// and z_i = L_r^-1 g_i is what the batched Cholesky's panel/update kernels produce when the
// standardised genotype rows g_i^T (sample-major, k_decode_gt) are appended to the system as extra
// right-hand-side ROWS: forward substitution is just more row tiles of the same MFMA kernels.
// Afterwards (Step1_Models.cpp:693-704): mask, centre by colsum/Neff, re-mask, scale by norm/sqrt(Neff-1).
#include "rg_internal.h"

// ---- Gt[blk][pos][j] = (g~_j(pos) - B_j . X(pos)) / s_j  (fp64, sample-major, ld = n64) ---------------
// grid (Np/64, n64/64, nblk), 256 threads; a 64(SNP) x 64(pos) tile goes through LDS for the transpose.
__global__ __launch_bounds__(256) void k_decode_gt(LoocvArgs a) {
  __shared__ double s[64][65];
  const int blk = blockIdx.z;
  const int bs = a.bs[blk];
  const int64_t pos0 = (int64_t)blockIdx.x * 64;
  const int j0 = blockIdx.y * 64;
  {
    const int jl = threadIdx.x >> 2, qd = threadIdx.x & 3;  // SNP row, 16-position quarter
    const int j = j0 + jl;
    double out[16];
    if (j < bs) {
      const unsigned w = *reinterpret_cast<const unsigned*>(a.pk + (int64_t)blk * a.pk_blk_stride +
                                                             (int64_t)j * a.pk_ld + pos0 / 4 + qd * 4);
      const double mu = a.mu[(int64_t)blk * a.n128 + j];
      const double inv = 1.0 / a.sc[(int64_t)blk * a.n128 + j];
      const double* B = a.Bm + ((int64_t)blk * a.n128 + j) * a.C;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const unsigned code = (w >> (2 * i)) & 3u;
        double g = (code == 0u) ? 2.0 : ((code == 2u) ? 1.0 : ((code == 1u) ? mu : 0.0));
        const int64_t pos = pos0 + qd * 16 + i;
        for (int c = 0; c < a.C; ++c) g = fma(-B[c], a.V[(int64_t)c * a.Np + pos], g);
        out[i] = g * inv;
      }
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) out[i] = 0.0;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) s[jl][qd * 16 + i] = out[i];
  }
  __syncthreads();
  {
    const int pl = threadIdx.x >> 2, jq = threadIdx.x & 3;  // position, 16-SNP quarter
    double* dst = a.gt + ((int64_t)blk * a.Np + pos0 + pl) * a.n64 + j0 + jq * 16;
#pragma unroll
    for (int i = 0; i < 16; ++i) dst[i] = s[jq * 16 + i][pl];
  }
}

// ---- LOO predictions from the forward-substituted rows -------------------------------------------------
// system (blk, r): rows [row_g0, row_g0 + Np) hold z_i = L^-1 g_i, rows [n64, n64+P) hold L^-1 b_p.
// grid (ceil(Np/256), R0, nblk), 256 threads = 4 waves, one wave per sample row at a time.
#define LP_MAX 4
__global__ __launch_bounds__(256) void k_l0_loocv_pred(LoocvArgs a) {
  extern __shared__ double syb[];  // [P][n64]
  const int blk = blockIdx.z, r = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t msz = (int64_t)a.rtot * a.n64;
  const double* M = a.wk + ((int64_t)blk * a.R0 + r) * msz;
  const int col = a.blockid[blk] * a.R0 + r;
  for (int p0 = 0; p0 < a.P; p0 += LP_MAX) {
    const int np = min(LP_MAX, a.P - p0);
    __syncthreads();
    for (int t = threadIdx.x; t < np * a.n64; t += 256) syb[t] = M[(int64_t)(a.n64 + p0) * a.n64 + t];
    __syncthreads();
    for (int rr = wave; rr < 256; rr += 4) {
      const int64_t pos = (int64_t)blockIdx.x * 256 + rr;
      if (pos >= a.Np) break;
      const double* z = M + ((int64_t)a.row_g0 + pos) * a.n64;
      double h = 0.0, num[LP_MAX];
#pragma unroll
      for (int p = 0; p < LP_MAX; ++p) num[p] = 0.0;
      // columns past the block's own tile count are identity padding the factorization does not produce (chol.hip: sys_tiles)
      const int kend = min(a.n64, (a.bs[blk] + 63) & ~63);
      for (int k = lane * 2; k < kend; k += 128) {
        const double2 zz = *reinterpret_cast<const double2*>(z + k);
        h = fma(zz.x, zz.x, h);
        h = fma(zz.y, zz.y, h);
#pragma unroll
        for (int p = 0; p < LP_MAX; ++p)
          if (p < np) {
            num[p] = fma(zz.x, syb[p * a.n64 + k], num[p]);
            num[p] = fma(zz.y, syb[p * a.n64 + k + 1], num[p]);
          }
      }
      for (int o = 32; o > 0; o >>= 1) {
        h += __shfl_down(h, o);
#pragma unroll
        for (int p = 0; p < LP_MAX; ++p) num[p] += __shfl_down(num[p], o);
      }
      if (lane == 0) {
#pragma unroll
        for (int p = 0; p < LP_MAX; ++p)
          if (p < np) {
            const double y = a.V[(int64_t)(a.C + p0 + p) * a.Np + pos];
            a.W[((int64_t)col * a.P + p0 + p) * a.Np + pos] = (num[p] - h * y) / (1.0 - h);
          }
      }
    }
  }
}

// ---- LOOCV column standardisation of W columns (also used by nothing else): three passes ----------------
// columns = (blk, r, p); pass 0: v *= mask, S1 = sum v;  pass 1: v = (v - S1/Neff) * mask, S2 = sum v^2;
// pass 2: v /= sqrt(S2 / (Neff - 1)).   part: [ncols][nchunk] partial sums (fixed-order reduction).
__global__ __launch_bounds__(256) void k_loocv_std(LoocvArgs a, int pass, int nchunk, double* part,
                                                   const double* prev) {
  __shared__ double red[4];
  const int blk = blockIdx.z, rp = blockIdx.y, ch = blockIdx.x;
  const int r = rp / a.P, p = rp % a.P;
  const int ncol = a.R0 * a.P;
  double* w = a.W + ((int64_t)(a.blockid[blk] * a.R0 + r) * a.P + p) * a.Np;
  const double* mk = a.maskp + (int64_t)p * a.Np;
  const double neff = a.neff[p];
  double stat = 0.0;
  if (pass > 0) {
    const double* q = prev + ((int64_t)blk * ncol + rp) * nchunk;
    for (int c = 0; c < nchunk; ++c) stat += q[c];
  }
  const double mean = stat / neff;                      // pass 1
  const double sd = sqrt(stat / (neff - 1.0));          // pass 2
  double acc = 0.0;
  const int64_t per = (a.Np + nchunk - 1) / nchunk;
  const int64_t lo = (int64_t)ch * per, hi = min(a.Np, lo + per);
  for (int64_t i = lo + threadIdx.x; i < hi; i += 256) {
    double v = w[i];
    if (pass == 0) { v *= mk[i]; acc += v; }
    else if (pass == 1) { v = (v - mean) * mk[i]; acc = fma(v, v, acc); }
    else v /= sd;
    w[i] = v;
  }
  if (pass < 2) {
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[((int64_t)blk * ncol + rp) * nchunk + ch] = (red[0] + red[1]) + (red[2] + red[3]);
  }
}

void rg_launch_l0_loocv(hipStream_t st, const LoocvArgs& a, double* part0, double* part1, int nchunk) {
  hipLaunchKernelGGL(k_l0_loocv_pred, dim3((unsigned)((a.Np + 255) / 256), a.R0, a.nblk), dim3(256),
                     sizeof(double) * LP_MAX * a.n64, st, a);
  dim3 g(nchunk, a.R0 * a.P, a.nblk);
  hipLaunchKernelGGL(k_loocv_std, g, dim3(256), 0, st, a, 0, nchunk, part0, (const double*)nullptr);
  hipLaunchKernelGGL(k_loocv_std, g, dim3(256), 0, st, a, 1, nchunk, part1, (const double*)part0);
  hipLaunchKernelGGL(k_loocv_std, g, dim3(256), 0, st, a, 2, nchunk, (double*)nullptr, (const double*)part1);
}

void rg_launch_decode_gt(hipStream_t st, const LoocvArgs& a) {
  hipLaunchKernelGGL(k_decode_gt, dim3((unsigned)(a.Np / 64), a.n64 / 64, a.nblk), dim3(256), 0, st, a);
}
