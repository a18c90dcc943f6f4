// Level 0 on NON-INTEGER genotypes (pgen dosages, later BGEN): the fp64 sibling of the 2-bit block pipeline.
//
// Reference: readChunkFromPGENFileToG (Geno.cpp:1773-1822: Read(), total = mean over analysed non-missing samples,
// mean_impute_g), residualize_genotypes (Data.cpp:190-224), calc_cv_matrices (Data.cpp:741-767), ridge_level_0
// (Step1_Models.cpp:458-613).  The 2-bit path never materialises the standardised genotypes: it works from exact integer
// Grams of the packed codes plus rank-C corrections.  Dosages have no such structure, so this path does what the reference
// does, per block, in device memory:
//   G  [n64][Np]  dosage rows in position space (fold-aligned sample order), mean-imputed, zero outside the analysis
//   GX = G X^T    (fp64 MFMA GEMM against the covariate basis)            G <- (G - GX X) / scale_G        (low variance -> flag)
//   fold Grams F_f = G_f G_f^T and right-hand sides Y_f G_f^T (fp64 MFMA GEMMs over the fold's position range), then the
//   training-fold systems sum - F_f in the workspace layout of the 2-bit path, the SAME batched Cholesky with the lambda
//   shifts formed on the fly, predictions beta^T G_f masked per phenotype, and the same column statistics / scaling kernels.
// Blocks go in mini-batches whose G stay resident (8 n64 Np bytes each: 0.4 GB at 50,000 samples, 4 GB at 500,000; <= 24 GB in
// all) so that one batched Cholesky serves all their systems.
#include <algorithm>
#include <vector>

#include "rg_internal.h"

#define F64_HIP(x)                                                             \
  do {                                                                         \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess) { ctx->err = std::string("HIP: ") + hipGetErrorString(e_); return RG_ERR_HIP; } \
  } while (0)

namespace {

__device__ __forceinline__ double block_sum256(double x, double* sred /*[4]*/) {
  for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = x;
  __syncthreads();
  return (sred[0] + sred[1]) + (sred[2] + sred[3]);
}

__device__ __forceinline__ bool act_at(const uint8_t* act, int64_t pos) { return ((act[pos >> 2] >> (2 * (pos & 3))) & 3) == 3; }

// grid (bs): mean of a variant over the analysed, non-missing samples (Geno.cpp:1805-1806); any value < -3 or > 2 raises
// the out-of-range flag (Geno.cpp:1799)
__global__ __launch_bounds__(256) void k_f64_rowstat(const double* rows, int64_t ld, const int64_t* pos2file, const uint8_t* act,
                                                     int64_t Np, double* mu, int32_t* info) {
  __shared__ double sred[4];
  const int j = blockIdx.x;
  const double* row = rows + (int64_t)j * ld;
  double sum = 0.0, cnt = 0.0;
  bool bad = false;
  for (int64_t pos = threadIdx.x; pos < Np; pos += 256) {
    const int64_t pf = pos2file[pos];
    if (pf < 0) continue;
    const double g = row[pf];
    if (g < -3.0 || g > 2.0) bad = true;
    if (act_at(act, pos) && g != -3.0) { sum += g; cnt += 1.0; }
  }
  const double s = block_sum256(sum, sred);
  const double c = block_sum256(cnt, sred);
  if (threadIdx.x == 0) mu[j] = s / c;
  if (bad) atomicMax(info + 2, 1);
}

// grid (Np / 256, n64): position-space fill with mean imputation; rows >= bs and padding positions are zero
__global__ __launch_bounds__(256) void k_f64_fill(const double* rows, int64_t ld, const int64_t* pos2file, const uint8_t* act,
                                                  int64_t Np, const double* mu, int bs, double* G) {
  const int j = blockIdx.y;
  const int64_t pos = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (pos >= Np) return;
  double v = 0.0;
  if (j < bs) {
    const int64_t pf = pos2file[pos];
    if (pf >= 0 && act_at(act, pos)) {
      const double g = rows[(int64_t)j * ld + pf];
      v = (g == -3.0) ? mu[j] : g;
    }
  }
  G[(int64_t)j * Np + pos] = v;
}

// grid (Np / 256, bs): partial products of a variant with the C covariate basis columns over one 256-position chunk
__global__ __launch_bounds__(256) void k_f64_gxpart(const double* G, int64_t Np, const double* X /*[64][Np]*/, int C,
                                                    double* part /*[bs][nchunk][C]*/) {
  __shared__ double sred[4];
  const int j = blockIdx.y;
  const int64_t pos = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const double g = pos < Np ? G[(int64_t)j * Np + pos] : 0.0;
  for (int c = 0; c < C; ++c) {
    const double s = block_sum256(pos < Np ? g * X[(int64_t)c * Np + pos] : 0.0, sred);
    if (threadIdx.x == 0) part[((int64_t)j * gridDim.x + blockIdx.x) * C + c] = s;
  }
}
// gx[j][c] = fixed-order sum of the chunk partials
__global__ void k_f64_gxsum(const double* part, int nchunk, int C, int bs, double* gx /*[n64][64]*/) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= bs * C) return;
  const int j = t / C, c = t % C;
  double s = 0.0;
  for (int ch = 0; ch < nchunk; ++ch) s += part[((int64_t)j * nchunk + ch) * C + c];
  gx[(int64_t)j * 64 + c] = s;
}

// right-hand sides: rhs_f[p][j] = sum over the positions of fold f of Y[p][pos] G~[j][pos], written into the system layout
// (rows n64 + p of fold f's [rtot][n64] block; the padding rows up to rtot are zeroed).  thread = (fold, row p, variant j)
__global__ void k_f64_gysum(const double* part /*[bs][nchunk][P]*/, int nchunk, int P, int bs, int n64, int rhs_pad, SegLayout seg,
                            double* F0, int64_t msz) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t per = (int64_t)rhs_pad * n64;
  if (t >= per * seg.nseg) return;
  const int f = (int)(t / per), p = (int)((t % per) / n64), j = (int)(t % n64);
  double s = 0.0;
  if (p < P && j < bs) {
    const int c0 = (int)(seg.pos_start[f] / 256), c1 = (int)((seg.pos_start[f] + seg.plen[f]) / 256);
    for (int ch = c0; ch < c1; ++ch) s += part[((int64_t)j * nchunk + ch) * P + p];
  }
  F0[(int64_t)f * msz + (int64_t)(n64 + p) * n64 + j] = s;
}

// grid (Np / 256, bs): G <- G - (G X^T) X, and the partial sums of squares of the residual rows
__global__ __launch_bounds__(256) void k_f64_resid(double* G, int64_t Np, const double* gx /*[n64][64]*/, const double* X /*[64][Np]*/,
                                                   int C, double* part /*[n64][nchunk]*/) {
  __shared__ double sred[4];
  const int j = blockIdx.y;
  const int64_t pos = (int64_t)blockIdx.x * 256 + threadIdx.x;
  double v = 0.0;
  if (pos < Np) {
    v = G[(int64_t)j * Np + pos];
    for (int c = 0; c < C; ++c) v = fma(-gx[(int64_t)j * 64 + c], X[(int64_t)c * Np + pos], v);
    G[(int64_t)j * Np + pos] = v;
  }
  const double s = block_sum256(v * v, sred);
  if (threadIdx.x == 0) part[(int64_t)j * gridDim.x + blockIdx.x] = s;
}

// one thread per variant: scale_G = ||row|| / sqrt(n_analyzed - ncov) (Data.cpp:203-209); low variance -> deferred flag
__global__ void k_f64_scale1(const double* part, int nchunk, int bs, double denom, int blk, double* sc, int32_t* info) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= bs) return;
  double ss = 0.0;
  for (int c = 0; c < nchunk; ++c) ss += part[(int64_t)j * nchunk + c];
  const double s = sqrt(ss) / denom;
  sc[j] = s;
  if (s < 1e-6) atomicMax(info, (blk << 20) | (j + 1));
}

__global__ __launch_bounds__(256) void k_f64_scale2(double* G, int64_t Np, const double* sc) {
  const int j = blockIdx.y;
  const int64_t pos = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (pos < Np) G[(int64_t)j * Np + pos] /= sc[j];
}

// fold[f] <- (sum over folds) - fold[f]: the training-fold system of fold f, matrix and right-hand-side rows alike
__global__ __launch_bounds__(256) void k_f64_train(double* fold, int K, int64_t msz) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= msz) return;
  double s = 0.0;
  for (int f = 0; f < K; ++f) s += fold[(int64_t)f * msz + e];
  for (int f = 0; f < K; ++f) fold[(int64_t)f * msz + e] = s - fold[(int64_t)f * msz + e];
}

// predictions of one block: grid (n_c256, P); thread = position of a 256-position chunk inside fold f:
//   W[(blockid R0 + r) P + p][pos] = mask_p(pos) * sum_j beta_{f, r, p}[j] G[j][pos]     (Step1_Models.cpp:496-507)
// and the chunk's partial sums of the values and their squares (the column statistics of :539-571)
#define F64_JT 256
#define F64_RMAX 8
__global__ __launch_bounds__(256) void k_f64_pred(const double* G, int64_t Np, int bs, int n64, int rtot, int R0, int P,
                                                  const double* wk, const double* maskp, const int32_t* cseg,
                                                  const int64_t* cpos, const int64_t* clen, int nchunk, int blockid,
                                                  double* W, double* psum) {
  __shared__ double sB[F64_RMAX][F64_JT];
  __shared__ double sred[4];
  const int ch = blockIdx.x, p = blockIdx.y;
  const int f = cseg[ch];
  const bool live = threadIdx.x < clen[ch];
  const int64_t pos = cpos[ch] + threadIdx.x;
  const int64_t msz = (int64_t)rtot * n64;
  double acc[F64_RMAX];
#pragma unroll
  for (int r = 0; r < F64_RMAX; ++r) acc[r] = 0.0;
  for (int j0 = 0; j0 < bs; j0 += F64_JT) {
    __syncthreads();
    for (int r = 0; r < R0; ++r) {
      const int j = j0 + threadIdx.x;
      sB[r][threadIdx.x] = (j < bs) ? wk[((int64_t)f * R0 + r) * msz + (int64_t)(n64 + p) * n64 + j] : 0.0;
    }
    __syncthreads();
    if (live) {
      const int jn = min(F64_JT, bs - j0);
      for (int j = 0; j < jn; ++j) {
        const double g = G[(int64_t)(j0 + j) * Np + pos];
#pragma unroll
        for (int r = 0; r < F64_RMAX; ++r)
          if (r < R0) acc[r] = fma(sB[r][j], g, acc[r]);
      }
    }
  }
  const double m = live ? maskp[(int64_t)p * Np + pos] : 0.0;
  for (int r = 0; r < R0; ++r) {
    const double v = (m != 0.0) ? acc[r] : 0.0;
    if (live) W[((int64_t)(blockid * R0 + r) * P + p) * Np + pos] = v;
    const double s1 = block_sum256(v, sred);
    const double s2 = block_sum256(v * v, sred);
    if (threadIdx.x == 0) {
      double* q = psum + (((int64_t)ch * P + p) * F64_RMAX + r) * 2;
      q[0] = s1;
      q[1] = s2;
    }
  }
}

// column statistics and scaling: the kernels of pred.hip, restated on this path's chunk partials (block slot 0)
__global__ void k_f64_stats(const double* psum, int nchunk, int P, int R0, const double* neff, double* stats) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= P * R0) return;
  const int p = t / R0, r = t % R0;
  double sx = 0.0, sq = 0.0;
  for (int ch = 0; ch < nchunk; ++ch) {
    const double* q = psum + (((int64_t)ch * P + p) * F64_RMAX + r) * 2;
    sx += q[0];
    sq += q[1];
  }
  const double n = neff[p];
  const double mean = sx / n;
  stats[((int64_t)p * F64_RMAX + r) * 2] = mean;
  stats[((int64_t)p * F64_RMAX + r) * 2 + 1] = sqrt((n - 1.0) / (sq - n * mean * mean));
}

__global__ __launch_bounds__(256) void k_f64_wscale(double* W, int64_t Np, int R0, int P, int blockid, const uint8_t* keptp,
                                                    const double* stats) {
  const int r = blockIdx.y % R0, p = blockIdx.y / R0;
  const double mean = stats[((int64_t)p * F64_RMAX + r) * 2], invsd = stats[((int64_t)p * F64_RMAX + r) * 2 + 1];
  double* w = W + ((int64_t)(blockid * R0 + r) * P + p) * Np;
  const int64_t pos = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (pos < Np) w[pos] = keptp[pos] ? (w[pos] - mean) * invsd : 0.0;
}

// gt[pos][j] = G[j][pos]: the sample-major standardised genotypes the leave-one-out path appends to its systems as extra
// right-hand-side rows (loocv.hip); grid (Np / 64, n64 / 64), a 64 x 64 tile through LDS
__global__ __launch_bounds__(256) void k_f64_transpose(const double* G, int64_t Np, int n64, double* gt, int64_t pos0) {
  __shared__ double t[64][65];
  const int64_t p0 = (int64_t)blockIdx.x * 64;
  const int j0 = blockIdx.y * 64;
  for (int e = threadIdx.x; e < 64 * 64; e += 256) {
    const int jl = e >> 6, pl = e & 63;
    t[jl][pl] = G[(int64_t)(j0 + jl) * Np + pos0 + p0 + pl];
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 64 * 64; e += 256) {
    const int pl = e >> 6, jl = e & 63;
    gt[(p0 + pl) * n64 + j0 + jl] = t[jl][pl];
  }
}

template <class T>
T* f64_buf(rg_ctx* ctx, int slot, size_t count) {
  const size_t bytes = std::max<size_t>(8, count * sizeof(T));
  if (ctx->f64_bytes[slot] < bytes) {
    if (ctx->f64_ptr[slot]) hipFree(ctx->f64_ptr[slot]);
    ctx->f64_ptr[slot] = nullptr;
    ctx->f64_bytes[slot] = 0;
    if (hipMalloc(&ctx->f64_ptr[slot], bytes) != hipSuccess) return nullptr;
    ctx->f64_bytes[slot] = bytes;
  }
  return (T*)ctx->f64_ptr[slot];
}

}  // namespace

int rg_l0_blocks_f64_impl(rg_ctx* ctx, int nblk, const int32_t* block_ids, const int32_t* bs, const double* const* rows,
                          int64_t row_stride, int mem_kind) {
  if (ctx->C > 64) { ctx->err = "rg_l0_blocks_f64: more than 64 covariate basis columns"; return RG_ERR_ARG; }
  if (ctx->R0 > F64_RMAX) { ctx->err = "rg_l0_blocks_f64: more than 8 level-0 ridge values"; return RG_ERR_ARG; }
  hipStream_t st = ctx->stream;
  const int K = ctx->K, R0 = ctx->R0, P = ctx->P, C = ctx->C, n64 = ctx->n64, rtot = ctx->rtot;
  const int64_t Np = ctx->Np, Nfile = ctx->Nfile;
  const int64_t msz = (int64_t)rtot * n64;
  const int rhs_pad = rtot - n64;
  const unsigned gpos = (unsigned)((Np + 255) / 256);
  // blocks per mini-batch: their standardised genotypes stay resident so that ONE batched Cholesky serves all of them
  int nbb = 1;
  if (!ctx->loocv) {
    const double per = 8.0 * (double)n64 * (double)Np;
    nbb = (int)std::max(1.0, std::min((double)std::min(nblk, ctx->nblk_cap), 24e9 / per));
  }
  // buffers: 0 G [nbb][n64][Np], 1 Xpad [64][Np], 2 Ypad [rhs_pad][Np], 3 pos2file, 4 staged host rows, 5 GX [n64][64], 6 mu,
  //          7 partials [n64][nchunk][max(C,P,1)], 8 scale
  const bool first = ctx->f64_ptr[1] == nullptr;
  double* Gall = f64_buf<double>(ctx, 0, (size_t)nbb * n64 * Np);
  double* Xp = f64_buf<double>(ctx, 1, (size_t)64 * Np);
  double* Yp = f64_buf<double>(ctx, 2, (size_t)rhs_pad * Np);
  int64_t* p2f = f64_buf<int64_t>(ctx, 3, (size_t)Np);
  double* gx = f64_buf<double>(ctx, 5, (size_t)n64 * 64);
  double* mu = f64_buf<double>(ctx, 6, (size_t)n64);
  double* part = f64_buf<double>(ctx, 7, (size_t)n64 * gpos * std::max(std::max(C, P), 1));
  double* sc = f64_buf<double>(ctx, 8, (size_t)n64);
  double* stage = nullptr;
  if (mem_kind != RG_MEM_DEVICE) stage = f64_buf<double>(ctx, 4, (size_t)ctx->bs_max * Nfile);
  if (!Gall || !Xp || !Yp || !p2f || !gx || !mu || !part || !sc || (mem_kind != RG_MEM_DEVICE && !stage)) {
    ctx->err = "rg_l0_blocks_f64: out of device memory";
    return RG_ERR_HIP;
  }
  if (first) {   // covariate basis and phenotypes padded to whole 64-row GEMM operands; position -> file index
    F64_HIP(hipMemsetAsync(Xp, 0, sizeof(double) * 64 * Np, st));
    F64_HIP(hipMemsetAsync(Yp, 0, sizeof(double) * (size_t)rhs_pad * Np, st));
    F64_HIP(hipMemcpyAsync(Xp, ctx->d_V, sizeof(double) * (size_t)C * Np, hipMemcpyDeviceToDevice, st));
    F64_HIP(hipMemcpyAsync(Yp, ctx->d_V + (size_t)C * Np, sizeof(double) * (size_t)P * Np, hipMemcpyDeviceToDevice, st));
    std::vector<int64_t> h((size_t)Np, -1);
    for (int f = 0; f < ctx->seg.nseg; ++f)
      for (int64_t i = 0; i < ctx->seg.len[f]; ++i) h[(size_t)(ctx->seg.pos_start[f] + i)] = ctx->seg.file_start[f] + i;
    F64_HIP(hipMemcpyAsync(p2f, h.data(), sizeof(int64_t) * Np, hipMemcpyHostToDevice, st));
    F64_HIP(hipStreamSynchronize(st));
  }
  const double denom = sqrt((double)(ctx->n_analyzed - C));
  for (int b0 = 0; b0 < nblk; b0 += nbb) {
    const int nb_ = std::min(nbb, nblk - b0);
    F64_HIP(hipMemcpyAsync(ctx->d_bs, bs + b0, sizeof(int32_t) * nb_, hipMemcpyHostToDevice, st));
    F64_HIP(hipMemcpyAsync(ctx->d_blockid, block_ids + b0, sizeof(int32_t) * nb_, hipMemcpyHostToDevice, st));
    // ---- per block: ingest, residualise, fold Grams --------------------------------------------------------------------
    for (int i = 0; i < nb_; ++i) {
      const int b = b0 + i, nb = bs[b];
      double* G = Gall + (size_t)i * n64 * Np;
      const double* src = rows[b];
      int64_t ld = row_stride;
      if (mem_kind != RG_MEM_DEVICE) {   // stream order protects the staging buffer: the copy queues behind its last readers
        F64_HIP(hipMemcpy2DAsync(stage, sizeof(double) * Nfile, rows[b], sizeof(double) * row_stride, sizeof(double) * Nfile, nb,
                                 hipMemcpyHostToDevice, st));
        src = stage;
        ld = Nfile;
      }
      hipLaunchKernelGGL(k_f64_rowstat, dim3(nb), dim3(256), 0, st, src, ld, p2f, ctx->d_act, Np, mu, ctx->d_info);
      hipLaunchKernelGGL(k_f64_fill, dim3(gpos, n64), dim3(256), 0, st, src, ld, p2f, ctx->d_act, Np, mu, nb, G);
      // residualize_genotypes
      hipLaunchKernelGGL(k_f64_gxpart, dim3(gpos, nb), dim3(256), 0, st, G, Np, Xp, C, part);
      hipLaunchKernelGGL(k_f64_gxsum, dim3((nb * C + 255) / 256), dim3(256), 0, st, part, (int)gpos, C, nb, gx);
      hipLaunchKernelGGL(k_f64_resid, dim3(gpos, nb), dim3(256), 0, st, G, Np, gx, Xp, C, part);
      hipLaunchKernelGGL(k_f64_scale1, dim3((nb + 63) / 64), dim3(64), 0, st, part, (int)gpos, nb, denom, i, sc, ctx->d_info);
      hipLaunchKernelGGL(k_f64_scale2, dim3(gpos, nb), dim3(256), 0, st, G, Np, sc);
      if (ctx->loocv) {
        // calc_cv_matrices, LOOCV branch (Data.cpp:755-767): the full Gram and G~ Y; the standardised genotypes, sample-major,
        // ride along as extra right-hand-side rows of the (A + lambda_r I) systems (loocv.hip) -- the same launches as the
        // 2-bit path from here on, one block at a time (nbb == 1)
        rg_launch_fold_gram_rows(st, G, Np, nb, n64, rtot, ctx->d_zero, ctx->seg, ctx->d_sum);
        hipLaunchKernelGGL(k_f64_gxpart, dim3(gpos, nb), dim3(256), 0, st, G, Np, Yp, P, part);
        hipLaunchKernelGGL(k_f64_gysum, dim3((unsigned)(((int64_t)rhs_pad * n64 + 255) / 256)), dim3(256), 0, st, part, (int)gpos, P, nb, n64,
                           rhs_pad, ctx->seg, ctx->d_sum, msz);
        LoocvArgs la;
        la.nblk = 1; la.R0 = R0; la.P = P; la.C = C; la.n128 = ctx->n128; la.n64 = n64; la.rtot = (int)ctx->rtot_wk;
        la.row_g0 = rtot; la.Np = Np; la.pk_ld = ctx->pk_ld; la.pk_blk_stride = 0; la.pk = nullptr;
        la.mu = nullptr; la.sc = nullptr; la.Bm = nullptr; la.V = ctx->d_V; la.maskp = ctx->d_maskp;
        la.neff = ctx->d_neff; la.bs = ctx->d_bs; la.blockid = ctx->d_blockid; la.wk = ctx->d_wk; la.gt = ctx->d_gt;
        la.W = rg_w_base(ctx);
        // one block at a time (nbb == 1): d_bs / d_blockid entry 0 must be THIS block's
        F64_HIP(hipMemcpyAsync(ctx->d_bs, bs + b, sizeof(int32_t), hipMemcpyHostToDevice, st));
        F64_HIP(hipMemcpyAsync(ctx->d_blockid, block_ids + b, sizeof(int32_t), hipMemcpyHostToDevice, st));
        const int rcl = rg_l0_loocv_tri(ctx, st, la, nb, ctx->d_sum, rtot, ctx->d_triws, ctx->d_zt, ctx->loo_chunk, [&](int64_t pos0, int64_t len) {
          hipLaunchKernelGGL(k_f64_transpose, dim3((unsigned)(len / 64), n64 / 64), dim3(256), 0, st, G, Np, n64, ctx->d_gt, pos0);
        });
        if (rcl) return rcl;
        rg_launch_l0_loocv(st, la, ctx->d_lpart, ctx->d_lpart + (size_t)R0 * P * 64, 64);
        continue;
      }
      // calc_cv_matrices: per-fold Gram (lower tiles, all folds in one launch) and right-hand sides, straight into the
      // [rtot][n64] system layout of block slot i; then fold f <- sum - fold f
      double* F0 = ctx->d_fold + (int64_t)i * K * msz;
      rg_launch_fold_gram_rows(st, G, Np, nb, n64, rtot, ctx->d_zero, ctx->seg, F0);
      hipLaunchKernelGGL(k_f64_gxpart, dim3(gpos, nb), dim3(256), 0, st, G, Np, Yp, P, part);   // chunk partials of Y G~^T
      hipLaunchKernelGGL(k_f64_gysum, dim3((unsigned)(((int64_t)rhs_pad * n64 * K + 255) / 256)), dim3(256), 0, st, part, (int)gpos, P, nb,
                         n64, rhs_pad, ctx->seg, F0, msz);
      hipLaunchKernelGGL(k_f64_train, dim3((unsigned)((msz + 255) / 256)), dim3(256), 0, st, F0, K, msz);
    }
    if (!ctx->loocv) {
      // ---- ridge_level_0: the K * R0 shifted systems of every block of the mini-batch in one batched Cholesky --------------
      rg_launch_chol_solve_formed_x(st, ctx->d_fold, msz, nullptr, 0, 1, ctx->d_lambda, R0, ctx->d_bs, 0, nb_ * K, ctx->d_wk, msz,
                                    n64, rhs_pad, P, ctx->d_dinv, ctx->d_info + 1, &ctx->tm.n_chol_launches, 0, nullptr, 0, 0, K,
                                    0, -1, 0);
      // ---- predictions and their standardisation ----------------------------------------------------------------------
      for (int i = 0; i < nb_; ++i) {
        const int b = b0 + i;
        const double* G = Gall + (size_t)i * n64 * Np;
        hipLaunchKernelGGL(k_f64_pred, dim3(ctx->n_c256, P), dim3(256), 0, st, G, Np, bs[b], n64, rtot, R0, P,
                           ctx->d_wk + (int64_t)i * K * R0 * msz, ctx->d_maskp, ctx->d_c256_seg, ctx->d_c256_pos, ctx->d_c256_len,
                           ctx->n_c256, block_ids[b], rg_w_base(ctx), ctx->d_psum);
        hipLaunchKernelGGL(k_f64_stats, dim3((P * R0 + 63) / 64), dim3(64), 0, st, ctx->d_psum, ctx->n_c256, P, R0, ctx->d_neff,
                           ctx->d_pstat);
        hipLaunchKernelGGL(k_f64_wscale, dim3(gpos, R0 * P), dim3(256), 0, st, rg_w_base(ctx), Np, R0, P, block_ids[b], ctx->d_keptp,
                           ctx->d_pstat);
      }
    }
    for (int i = 0; i < nb_; ++i) ctx->block_done[block_ids[b0 + i]] = 1;
  }
  return RG_OK;
}
