// Level-0 leave-one-out path (reference src/Data.cpp:755-767 + src/Step1_Models.cpp:615-726): the pieces around loocv_tri.hip.
//   k_decode_gt   the standardised genotypes of a chunk of samples, sample-major (the B operand of Z = Q^T G~)
//   k_loocv_std   afterwards (Step1_Models.cpp:693-704): mask, centre by colsum / Neff, re-mask, scale by norm / sqrt(Neff - 1)
#include "rg_internal.h"

// ---- Gt[blk][pos][j] = (g~_j(pos) - B_j . X(pos)) / s_j  (fp64, sample-major, ld = n64) ---------------
// grid (Np/64, n64/64, nblk), 256 threads; a 64(SNP) x 64(pos) tile goes through LDS for the transpose.
__global__ __launch_bounds__(256) void k_decode_gt(LoocvArgs a) {
  __shared__ double s[64][65];
  __shared__ double sx[16][64];     // covariate basis of the tile's 64 positions, up to sixteen columns at a time
  const int blk = blockIdx.z;
  const int bs = a.bs[blk];
  const int64_t pos0 = a.gt_pos0 + (int64_t)blockIdx.x * 64;
  const int j0 = blockIdx.y * 64;
  {
    const int jl = threadIdx.x >> 2, qd = threadIdx.x & 3;  // SNP row, 16-position quarter
    const int j = j0 + jl;
    const bool on = j < bs;
    double out[16];
    const unsigned w = *reinterpret_cast<const unsigned*>(a.pk + (int64_t)blk * a.pk_blk_stride + (int64_t)(on ? j : 0) * a.pk_ld + pos0 / 4 + qd * 4);
    const double mu = a.mu[(int64_t)blk * a.n128 + (on ? j : 0)];
    const double inv = on ? 1.0 / a.sc[(int64_t)blk * a.n128 + j] : 0.0;
    const double* B = a.Bm + ((int64_t)blk * a.n128 + (on ? j : 0)) * a.C;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const unsigned code = (w >> (2 * i)) & 3u;
      out[i] = (code == 0u) ? 2.0 : ((code == 2u) ? 1.0 : ((code == 1u) ? mu : 0.0));
    }
    for (int c0 = 0; c0 < a.C; c0 += 16) {
      const int cn = min(16, a.C - c0);
      __syncthreads();
      for (int e = threadIdx.x; e < cn * 64; e += 256) sx[e >> 6][e & 63] = a.V[(int64_t)(c0 + (e >> 6)) * a.Np + pos0 + (e & 63)];
      __syncthreads();
      for (int c = 0; c < cn; ++c) {
        const double b = B[c0 + c];
#pragma unroll
        for (int i = 0; i < 16; ++i) out[i] = fma(-b, sx[c][qd * 16 + i], out[i]);
      }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) s[jl][qd * 16 + i] = out[i] * inv;
  }
  __syncthreads();
  {
    // 32 lanes write the 64 SNPs of one position (512 B contiguous), two positions per wave instruction
    double* dst0 = a.gt + ((int64_t)blk * a.gt_len + (pos0 - a.gt_pos0)) * a.n64 + j0;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int idx = it * 256 + threadIdx.x;
      const int pl = idx >> 5, jj = (idx & 31) * 2;
      *reinterpret_cast<double2*>(dst0 + (int64_t)pl * a.n64 + jj) = make_double2(s[jj][pl], s[jj + 1][pl]);
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
  dim3 g(nchunk, a.R0 * a.P, a.nblk);
  hipLaunchKernelGGL(k_loocv_std, g, dim3(256), 0, st, a, 0, nchunk, part0, (const double*)nullptr);
  hipLaunchKernelGGL(k_loocv_std, g, dim3(256), 0, st, a, 1, nchunk, part1, (const double*)part0);
  hipLaunchKernelGGL(k_loocv_std, g, dim3(256), 0, st, a, 2, nchunk, (double*)nullptr, (const double*)part1);
}

// standardised genotypes of the sample positions [a.gt_pos0, a.gt_pos0 + a.gt_len) of every block, sample-major: gt[blk][pos - gt_pos0][j]
void rg_launch_decode_gt(hipStream_t st, const LoocvArgs& a) {
  hipLaunchKernelGGL(k_decode_gt, dim3((unsigned)(a.gt_len / 64), a.n64 / 64, a.nblk), dim3(256), 0, st, a);
}
