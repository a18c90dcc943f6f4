// C-ABI entry points of librg_step1_hip.so (see include/rg_step1.h) and the host-side
// orchestration of the level-0 block pipeline.  No CPU compute fallback exists: every numerical
// step is a HIP kernel launch; if no GPU is present rg_create fails.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include "rg_internal.h"

namespace {

template <class T>
int dev_alloc(rg_ctx* ctx, T** p, size_t n) {
  if (*p) { hipFree(*p); *p = nullptr; }
  if (n == 0) n = 1;
  RG_HIP(hipMalloc((void**)p, n * sizeof(T)));
  return RG_OK;
}
template <class T>
int dev_upload(rg_ctx* ctx, T** p, const std::vector<T>& v) {
  int rc = dev_alloc(ctx, p, v.size());
  if (rc) return rc;
  if (!v.empty()) RG_HIP(hipMemcpy(*p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  return RG_OK;
}

void free_all(rg_ctx* c) {
  void* ptrs[] = {c->d_cidx, c->d_act, c->d_V, c->d_maskp, c->d_Q, c->d_XtY, c->d_lambda, c->d_neff,
                  c->d_keptp, c->d_posc, c->d_zero, c->d_raw, c->d_pk, c->d_pk4, c->d_mu, c->d_nmiss,
                  c->d_xypart, c->d_chunk_seg, c->d_chunk_pos, c->d_chunk_len, c->d_S, c->d_F, c->d_Bm,
                  c->d_BQ, c->d_GYt, c->d_sc, c->d_fold, c->d_sum, c->d_wk, c->d_dinv, c->d_beta,
                  c->d_cb, c->d_psum, c->d_pstat, c->d_info, c->d_bs, c->d_blockid, (void*)c->d_rawptr, c->d_c1k_seg, c->d_c1k_pos,
                  c->d_c1k_len, c->d_c256_seg, c->d_c256_pos, c->d_c256_len, c->d_gt, c->d_zt, c->d_triws, c->d_lpart, c->d_bplanes, c->d_bsc, c->d_pkT, c->d_vd, c->d_vsc, c->d_xyS, c->d_segid};
  for (void* p : ptrs)
    if (p) hipFree(p);
  if (c->own_W && c->d_W) hipFree(c->d_W);
  for (int i = 0; i < 16; ++i)
    if (c->ws_ptr[i]) { hipFree(c->ws_ptr[i]); c->ws_ptr[i] = nullptr; c->ws_bytes[i] = 0; }
  for (int i = 0; i < 10; ++i)
    if (c->f64_ptr[i]) { hipFree(c->f64_ptr[i]); c->f64_ptr[i] = nullptr; c->f64_bytes[i] = 0; }
}

struct StageTimer {
  rg_ctx* c; double* slot;
  StageTimer(rg_ctx* ctx, double* s) : c(ctx), slot(s) { if (c->timing) hipEventRecord(c->ev0, c->stream); }
  ~StageTimer() {
    if (!c->timing) return;
    hipEventRecord(c->ev1, c->stream);
    hipEventSynchronize(c->ev1);
    float ms = 0;
    hipEventElapsedTime(&ms, c->ev0, c->ev1);
    *slot += ms;
  }
};

}  // namespace

extern "C" {

int rg_create(rg_ctx** out, int device_id, void* hip_stream) {
  if (!out) return RG_ERR_ARG;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device_id >= ndev) return RG_ERR_HIP;
  if (hipSetDevice(device_id) != hipSuccess) return RG_ERR_HIP;
  rg_ctx* c = new rg_ctx();
  c->device = device_id;
  if (hip_stream) c->stream = (hipStream_t)hip_stream;
  else {
    if (hipStreamCreate(&c->stream) != hipSuccess) { delete c; return RG_ERR_HIP; }
    c->own_stream = true;
  }
  hipEventCreate(&c->ev0);
  hipEventCreate(&c->ev1);
  *out = c;
  return RG_OK;
}

void rg_destroy(rg_ctx* c) {
  if (!c) return;
  hipSetDevice(c->device);
  if (c->twin) { rg_destroy(c->twin); c->twin = nullptr; }
  if (c->ev_ingest) hipEventDestroy(c->ev_ingest);
  if (c->ev_tw_fork) hipEventDestroy(c->ev_tw_fork);
  if (c->ev_tw_join) hipEventDestroy(c->ev_tw_join);
  hipStreamSynchronize(c->stream);
  free_all(c);
  if (c->ev0) hipEventDestroy(c->ev0);
  if (c->ev1) hipEventDestroy(c->ev1);
  if (c->ev_fork) {
    hipEventDestroy(c->ev_fork);
    for (int k = 0; k < 3; ++k) { hipStreamSynchronize(c->st_part[k]); hipStreamDestroy(c->st_part[k]); hipEventDestroy(c->ev_part[k]); }
  }
  if (c->st_stage) { hipStreamSynchronize(c->st_stage); hipStreamDestroy(c->st_stage); }
  if (c->own_stream) hipStreamDestroy(c->stream);
  delete c;
}

const char* rg_last_error(const rg_ctx* c) { return c ? c->err.c_str() : "null context"; }

int rg_set_problem(rg_ctx* ctx, const rg_problem* p) {
  if (!ctx || !p) return RG_ERR_ARG;
  hipSetDevice(ctx->device);
  // RG_TIMING: where the set-up goes (host layout + uploads | level-0 workspaces | digit planes of V | the other pipelines)
  static const bool sp_timing = getenv("RG_TIMING") != nullptr;
  const auto sp_t0 = std::chrono::steady_clock::now();
  auto sp_mark = [&](const char* what) {
    if (sp_timing) fprintf(stderr, "[timing] rg_set_problem%s: %s at %.0f ms\n", ctx->is_child ? " (second pipeline)" : "", what,
                           std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - sp_t0).count());
  };
  // cv_folds == 0 selects leave-one-out CV (params.use_loocv: cv_folds := n_samples, Data.cpp:363)
  if (p->cv_folds != 0 && (p->cv_folds < 2 || p->cv_folds > RG_MAX_SEG)) { ctx->err = "cv_folds must be 0 (LOOCV) or in [2,32]"; return RG_ERR_ARG; }
  ctx->loocv = (p->cv_folds == 0);
  if (p->n_ridge_l0 < 1 || p->n_ridge_l0 > 8) { ctx->err = "n_ridge_l0 must be in [1,8]"; return RG_ERR_ARG; }
  if (p->n_pheno < 1 || p->n_pheno > 64 || p->n_cov < 1 || p->n_cov > 64) { ctx->err = "n_pheno / n_cov must be in [1,64]"; return RG_ERR_ARG; }
  if (p->n_samples < 1 || p->n_file < p->n_samples || p->max_block_size < 1 || p->n_blocks_total < 1) { ctx->err = "bad sizes"; return RG_ERR_ARG; }
  ctx->N = p->n_samples; ctx->Nfile = p->n_file; ctx->P = p->n_pheno; ctx->C = p->n_cov;
  ctx->K = ctx->loocv ? 1 : p->cv_folds;  // LOOCV: one segment holding every sample
  ctx->R0 = p->n_ridge_l0; ctx->ref_first = p->ref_first;
  ctx->n_analyzed = p->n_analyzed; ctx->B_total = p->n_blocks_total; ctx->bs_max = p->max_block_size;
  const int64_t N = ctx->N, Nf = ctx->Nfile;
  const int P = ctx->P, C = ctx->C, K = ctx->K;
  ctx->lambda.assign(p->lambda, p->lambda + ctx->R0);
  ctx->neff.assign(p->neff, p->neff + P);

  // compact -> file index
  std::vector<int64_t> file_of_c(N);
  {
    int64_t n = 0;
    for (int64_t i = 0; i < Nf; ++i) {
      if (p->ind_ignore && p->ind_ignore[i]) continue;
      if (n >= N) { ctx->err = "ind_ignore keeps more than n_samples samples"; return RG_ERR_ARG; }
      file_of_c[n++] = i;
    }
    if (n != N) { ctx->err = "ind_ignore does not keep n_samples samples"; return RG_ERR_ARG; }
  }
  ctx->fold_cstart.assign(K + 1, 0);
  for (int f = 0; f < K; ++f) {
    const int64_t sz = ctx->loocv ? N : (int64_t)p->cv_sizes[f];
    if (sz < 1) { ctx->err = "empty CV fold"; return RG_ERR_ARG; }
    ctx->fold_cstart[f + 1] = ctx->fold_cstart[f] + sz;
  }
  if (ctx->fold_cstart[K] != N) { ctx->err = "cv_sizes do not sum to n_samples"; return RG_ERR_ARG; }

  SegLayout& sg = ctx->seg;
  memset(&sg, 0, sizeof(sg));
  sg.nseg = K;
  int64_t pos = 0;
  for (int f = 0; f < K; ++f) {
    const int64_t fs = (f == 0) ? 0 : file_of_c[ctx->fold_cstart[f]];
    const int64_t fe = (f == K - 1) ? Nf : file_of_c[ctx->fold_cstart[f + 1]];
    sg.file_start[f] = fs;
    sg.len[f] = fe - fs;
    sg.plen[f] = rg_round_up(fe - fs, 256);  // one LDS stage of the FP4 Gram kernel (gram_fp4.hip)
    sg.pos_start[f] = pos;
    pos += sg.plen[f];
  }
  ctx->Np = pos;
  const int64_t Np = ctx->Np;

  std::vector<int32_t> cidx(Np, -1);
  std::vector<uint8_t> act(Np / 4, 0), keptp(Np, 0);
  std::vector<int64_t> posc(N);
  const int Cv = C + P;
  std::vector<double> V((size_t)Cv * Np, 0.0), maskp((size_t)P * Np, 0.0);
  std::vector<double> Q((size_t)K * C * C, 0.0), XtY((size_t)K * C * P, 0.0);
  int n_active = 0;
  {
    int64_t n = 0;
    for (int f = 0; f < K; ++f)
      for (int64_t i = sg.file_start[f]; i < sg.file_start[f] + sg.len[f]; ++i) {
        if (p->ind_ignore && p->ind_ignore[i]) continue;
        const int64_t ps = sg.pos_start[f] + (i - sg.file_start[f]);
        cidx[ps] = (int32_t)n;
        posc[n] = ps;
        keptp[ps] = 1;
        if (p->ind_in_analysis[n]) { act[ps >> 2] |= (uint8_t)(3u << (2 * (ps & 3))); ++n_active; }
        for (int c = 0; c < C; ++c) V[(size_t)c * Np + ps] = p->X[(size_t)c * N + n];
        for (int q = 0; q < P; ++q) {
          V[(size_t)(C + q) * Np + ps] = p->Y[(size_t)q * N + n];
          maskp[(size_t)q * Np + ps] = p->mask[(size_t)q * N + n] ? 1.0 : 0.0;
        }
        // the fold of a compact sample is defined by cv_sizes, which the segments reproduce
        for (int c = 0; c < C; ++c) {
          const double xc = p->X[(size_t)c * N + n];
          for (int c2 = 0; c2 < C; ++c2) Q[((size_t)f * C + c) * C + c2] += xc * p->X[(size_t)c2 * N + n];
          for (int q = 0; q < P; ++q) XtY[((size_t)f * C + c) * P + q] += xc * p->Y[(size_t)q * N + n];
        }
        ++n;
      }
    if (n != N) { ctx->err = "internal: fold layout does not cover all samples"; return RG_ERR_STATE; }
  }
  ctx->n_active = n_active;

  // position chunks (<= 4096 positions, inside one fold)
  ctx->h_chunk_seg.clear(); ctx->h_chunk_pos.clear(); ctx->h_chunk_len.clear();
  for (int f = 0; f < K; ++f)
    for (int64_t o = 0; o < sg.plen[f]; o += 4096) {
      ctx->h_chunk_seg.push_back(f);
      ctx->h_chunk_pos.push_back(sg.pos_start[f] + o);
      ctx->h_chunk_len.push_back(std::min<int64_t>(4096, sg.plen[f] - o));
    }
  ctx->xy_nchunk = (int)ctx->h_chunk_seg.size();
  for (int pass = 0; pass < 2; ++pass) {
    const int64_t step = pass == 0 ? 1024 : 256;
    std::vector<int32_t> cs; std::vector<int64_t> cp, cl;
    for (int f = 0; f < K; ++f)
      for (int64_t o = 0; o < sg.plen[f]; o += step) {
        cs.push_back(f); cp.push_back(sg.pos_start[f] + o); cl.push_back(std::min<int64_t>(step, sg.plen[f] - o));
      }
    int rc2;
    if (pass == 0) {
      ctx->n_c1k = (int)cs.size();
      if ((rc2 = dev_upload(ctx, &ctx->d_c1k_seg, cs)) || (rc2 = dev_upload(ctx, &ctx->d_c1k_pos, cp)) ||
          (rc2 = dev_upload(ctx, &ctx->d_c1k_len, cl))) return rc2;
    } else {
      ctx->n_c256 = (int)cs.size();
      if ((rc2 = dev_upload(ctx, &ctx->d_c256_seg, cs)) || (rc2 = dev_upload(ctx, &ctx->d_c256_pos, cp)) ||
          (rc2 = dev_upload(ctx, &ctx->d_c256_len, cl))) return rc2;
    }
  }

  int rc;
  if ((rc = dev_upload(ctx, &ctx->d_cidx, cidx))) return rc;
  if ((rc = dev_upload(ctx, &ctx->d_act, act))) return rc;
  if ((rc = dev_upload(ctx, &ctx->d_keptp, keptp))) return rc;
  if ((rc = dev_upload(ctx, &ctx->d_posc, posc))) return rc;
  ctx->h_posc = posc;
  if ((rc = dev_upload(ctx, &ctx->d_V, V))) return rc;
  if ((rc = dev_upload(ctx, &ctx->d_maskp, maskp))) return rc;
  if ((rc = dev_upload(ctx, &ctx->d_Q, Q))) return rc;
  if ((rc = dev_upload(ctx, &ctx->d_XtY, XtY))) return rc;
  if ((rc = dev_upload(ctx, &ctx->d_lambda, ctx->lambda))) return rc;
  if ((rc = dev_upload(ctx, &ctx->d_neff, ctx->neff))) return rc;
  if ((rc = dev_upload(ctx, &ctx->d_chunk_seg, ctx->h_chunk_seg))) return rc;
  if ((rc = dev_upload(ctx, &ctx->d_chunk_pos, ctx->h_chunk_pos))) return rc;
  if ((rc = dev_upload(ctx, &ctx->d_chunk_len, ctx->h_chunk_len))) return rc;
  if ((rc = dev_alloc(ctx, &ctx->d_zero, (size_t)Np))) return rc;
  RG_HIP(hipMemset(ctx->d_zero, 0, sizeof(double) * Np));
  if ((rc = dev_alloc(ctx, &ctx->d_info, 4))) return rc;
  RG_HIP(hipMemset(ctx->d_info, 0, sizeof(int32_t) * 4));

  sp_mark("host layout + uploads done");
  // level-0 workspaces
  const int bsm = ctx->bs_max;
  ctx->n128 = (int)rg_round_up(bsm, 128);
  ctx->n64 = (int)rg_round_up(bsm, 64);
  ctx->rtot = ctx->n64 + (int)rg_round_up(P, 64);
  // LOOCV: no ridge systems at level 0 (loocv_tri.hip: one tridiagonal reduction per block serves every ridge value)
  ctx->rtot_wk = (int64_t)ctx->rtot;
  ctx->nsys = ctx->loocv ? 1 : K * ctx->R0;
  ctx->loo_chunk = std::min<int64_t>(Np, 65536);   // sample positions whose transformed genotypes Q^T g~ are resident at a time
  // blocks per batch: more systems per launch hide the Cholesky dependency chain.  (Sizing batches to one round of the
  // block factorization -- one wave per system, 4 per CU: 1024 systems -- was measured: 37.9 ms per step with 3 batches of 37
  // blocks against 36.1 ms with 2 of 55 at BASELINE configs[1]; what the extra batch costs elsewhere outweighs the saved round.)
  int nb = ctx->ws_nblk > 0 ? ctx->ws_nblk : 64;
  if (const char* e = getenv("RG_NBLK")) nb = std::max(1, atoi(e));
  nb = std::min(nb, ctx->B_total);
  {  // memory: the per-block workspaces of all pipelines stay inside a budget (default 64 GB, RG_WS_GB): at 500,000 samples a
     // block costs ~0.95 GB (packed planes 0.5, the 25 system workspaces 0.22, integer Grams 0.08, ...), so 2 x 64 blocks
     // would take 120 GB next to W, the exchange buffers and the resident genotypes of a 2-GPU BASELINE configs[2] run
    double budget = ctx->ws_budget > 0 ? (double)ctx->ws_budget : 64e9;
    if (const char* e = getenv("RG_WS_GB")) budget = std::max(1.0, atof(e)) * 1e9;
    {  // never more than half of what the device has free right now (W, the exchange buffers and level 1 come on top)
      size_t fr = 0, tot = 0;
      if (!ctx->is_child && hipMemGetInfo(&fr, &tot) == hipSuccess && fr > 0) budget = std::min(budget, 0.5 * (double)fr);
    }
    const double n128d = (double)rg_round_up(ctx->bs_max, 128), n64d = (double)rg_round_up(ctx->bs_max, 64);
    const double rtotd = n64d + (double)rg_round_up(P, 64);
    const double per_blk = n128d * (Np / 4.0) * 2.0 /* pk + pkT */ + n128d * (Np / 2.0) /* FP4 plane */ + (double)K * 4.0 * n128d * n128d * 4.0 /* S */ +
                           ((double)K + 1.0 + (double)K * ctx->R0) * rtotd * n64d * 8.0 /* fold, sum, wk */ +
                           (double)K * ctx->R0 * (n64d / 64.0 + 10.0 * ((n64d / 64.0 + 3.0) / 4.0)) * 4096.0 * 8.0 /* dinv + images */;
    int npipe = ctx->ws_pipes > 0 ? ctx->ws_pipes : 2;
    if (const char* e = getenv("RG_PIPELINES")) npipe = std::max(1, atoi(e));
    if (!ctx->loocv) nb = (int)std::max(4.0, std::min((double)nb, budget / (npipe * per_blk)));
    nb = std::min(nb, ctx->B_total);
  }
  {  // balanced batches: B blocks go in ceil(B / nb) batches of (almost) equal size -- the workspaces are sized for those,
     // not for the cap (109 blocks: 2 x 55 instead of 64 + 45; allocation and first-touch time of the set-up scale with it)
    const int nbatch = (ctx->B_total + nb - 1) / nb;
    nb = (ctx->B_total + nbatch - 1) / nbatch;
  }
  if (ctx->loocv) {  // bound the chunk buffers of the leave-one-out path (g~ and Q^T g~ of a chunk of samples, fp64: ~16 GB)
    const double per_blk = 2.0 * 8.0 * ctx->n64 * (double)ctx->loo_chunk;
    nb = (int)std::max(1.0, std::min((double)nb, 16e9 / per_blk));
  }
  ctx->nblk_cap = nb;
  const int nseg = K, R0 = ctx->R0, n128 = ctx->n128, n64 = ctx->n64, rtot = ctx->rtot;
  ctx->raw_ld = rg_round_up((Nf + 3) / 4, 16);
  ctx->pk_ld = Np / 4;
  const size_t msz = (size_t)rtot * n64;
  if (ctx->d_raw) { hipFree(ctx->d_raw); ctx->d_raw = nullptr; }   // staging of host rows: allocated by the first RG_MEM_HOST batch
  if ((rc = dev_alloc(ctx, &ctx->d_pk, (size_t)nb * n128 * ctx->pk_ld + 16))) return rc;
  ctx->gram_fp4 = true;
  if (const char* e = getenv("RG_GRAM")) ctx->gram_fp4 = std::string(e) != "i8";
  ctx->pk4_ld = Np / 2;
  if (ctx->gram_fp4) {
    if ((rc = dev_alloc(ctx, &ctx->d_pk4, (size_t)nb * n128 * ctx->pk4_ld + 16))) return rc;
  } else if (ctx->d_pk4) { hipFree(ctx->d_pk4); ctx->d_pk4 = nullptr; }
  if ((rc = dev_alloc(ctx, &ctx->d_mu, (size_t)nb * n128 * 2))) return rc;  // mu + int scratch
  if ((rc = dev_alloc(ctx, &ctx->d_nmiss, (size_t)nb))) return rc;
  if ((rc = dev_alloc(ctx, &ctx->d_xypart, (size_t)nb * ctx->xy_nchunk * n128 * 2 * Cv))) return rc;
  if ((rc = dev_alloc(ctx, &ctx->d_S, (size_t)nb * nseg * 4 * n128 * n128))) return rc;
  if ((rc = dev_alloc(ctx, &ctx->d_F, (size_t)nb * nseg * n128 * C))) return rc;
  if ((rc = dev_alloc(ctx, &ctx->d_Bm, (size_t)nb * n128 * C))) return rc;
  if ((rc = dev_alloc(ctx, &ctx->d_BQ, (size_t)nb * nseg * n128 * C))) return rc;
  if ((rc = dev_alloc(ctx, &ctx->d_GYt, (size_t)nb * nseg * n128 * P))) return rc;
  if ((rc = dev_alloc(ctx, &ctx->d_sc, (size_t)nb * n128))) return rc;
  if ((rc = dev_alloc(ctx, &ctx->d_fold, (size_t)nb * nseg * msz))) return rc;
  if ((rc = dev_alloc(ctx, &ctx->d_sum, (size_t)nb * msz))) return rc;
  if ((rc = dev_alloc(ctx, &ctx->d_wk, (size_t)nb * ctx->nsys * ctx->rtot_wk * n64))) return rc;
  if ((rc = dev_alloc(ctx, &ctx->d_dinv, rg_chol_ws_doubles((size_t)nb * ctx->nsys, n64)))) return rc;
  if (ctx->loocv) {
    if ((rc = dev_alloc(ctx, &ctx->d_gt, (size_t)nb * ctx->loo_chunk * n64))) return rc;
    if ((rc = dev_alloc(ctx, &ctx->d_zt, (size_t)nb * ctx->loo_chunk * n64))) return rc;
    if ((rc = dev_alloc(ctx, &ctx->d_triws, rg_loocv_tri_ws_doubles(nb, n64, P, R0)))) return rc;
    if ((rc = dev_alloc(ctx, &ctx->d_lpart, (size_t)2 * nb * R0 * P * 64))) return rc;
  }
  if ((rc = dev_alloc(ctx, &ctx->d_beta, (size_t)nb * nseg * R0 * P * n64))) return rc;
  if ((rc = dev_alloc(ctx, &ctx->d_cb, (size_t)nb * nseg * R0 * P * C))) return rc;
  if ((rc = dev_alloc(ctx, &ctx->d_psum, (size_t)nb * ctx->n_c256 * P * 8 * 2))) return rc;
  sp_mark("level-0 workspaces allocated");
  // G~X / G~Y on the i8 matrix cores (xy_i8.hip): digit planes of V = [X | Y], once per problem (RG_XY_F64=1 keeps the fp64 kernel)
  if (ctx->d_vd) { hipFree(ctx->d_vd); ctx->d_vd = nullptr; }
  if (ctx->d_vsc) { hipFree(ctx->d_vsc); ctx->d_vsc = nullptr; }
  if (ctx->d_xyS) { hipFree(ctx->d_xyS); ctx->d_xyS = nullptr; }
  if (ctx->d_segid) { hipFree(ctx->d_segid); ctx->d_segid = nullptr; }
  if (!getenv("RG_XY_F64")) {      // any number of columns: the kernel takes them sixteen at a time
    if ((rc = dev_alloc(ctx, &ctx->d_vd, (size_t)Cv * 8 * Np))) return rc;
    if ((rc = dev_alloc(ctx, &ctx->d_vsc, (size_t)Cv))) return rc;
    if ((rc = dev_alloc(ctx, &ctx->d_xyS, (size_t)nb * 2 * nseg * n128 * 128))) return rc;
    std::vector<int32_t> ids(nseg);
    for (int f = 0; f < nseg; ++f) ids[f] = f;
    if ((rc = dev_upload(ctx, &ctx->d_segid, ids))) return rc;
    rg_launch_v_split(ctx->stream, ctx->d_V, Np, Cv, ctx->d_vd, ctx->d_vsc);
    RG_HIP(hipStreamSynchronize(ctx->stream));
  }
  // many (phenotype, ridge value) rows: the exact i8 route of the predictions (pred_i8.hip) needs digit planes and a
  // SNP-contiguous copy of the packed rows; few rows (one phenotype) stay on the fp64 VALU kernel
  if (ctx->d_bplanes) { hipFree(ctx->d_bplanes); ctx->d_bplanes = nullptr; }
  if (ctx->d_bsc) { hipFree(ctx->d_bsc); ctx->d_bsc = nullptr; }
  if (ctx->d_pkT) { hipFree(ctx->d_pkT); ctx->d_pkT = nullptr; }
  if (!ctx->loocv && P * R0 > 16 && R0 <= 8 && n128 <= 1024) {
    const int pg = std::max(1, std::min(P, 64 / R0)), ngrp = (P + pg - 1) / pg;
    if ((rc = dev_alloc(ctx, &ctx->d_bplanes, (size_t)nb * nseg * ngrp * 2 * 8 * 64 * n128))) return rc;
    if ((rc = dev_alloc(ctx, &ctx->d_bsc, (size_t)nb * nseg * ngrp * 2 * 64))) return rc;
    if ((rc = dev_alloc(ctx, &ctx->d_pkT, (size_t)nb * Np * (n128 / 4)))) return rc;
  }
  if ((rc = dev_alloc(ctx, &ctx->d_pstat, (size_t)nb * P * 8 * 2))) return rc;
  if ((rc = dev_alloc(ctx, &ctx->d_bs, (size_t)nb))) return rc;
  if ((rc = dev_alloc(ctx, &ctx->d_blockid, (size_t)nb))) return rc;
  if ((rc = dev_alloc(ctx, &ctx->d_rawptr, (size_t)nb))) return rc;
  ctx->w_b0 = 0; ctx->w_nb = ctx->B_total;
  ctx->W_bytes = (int64_t)sizeof(double) * ctx->B_total * R0 * P * Np;
  if (ctx->own_W && ctx->d_W) { hipFree(ctx->d_W); }
  ctx->d_W = nullptr; ctx->own_W = false;
  ctx->block_done.assign(ctx->B_total, 0);
  ctx->v_W = nullptr; ctx->v_p0 = 0; ctx->v_np = P;
  ctx->have_problem = true;
  ctx->join_pending = false; ctx->pipe_rr = 0; ctx->ingest_pending = false;
  memset(&ctx->tm, 0, sizeof(ctx->tm));
  sp_mark("this pipeline ready");
  // further pipelines (see rg_ctx::twin): a chain of child contexts; RG_PIPELINES=1 keeps a single one
  if (ctx->twin) { rg_destroy(ctx->twin); ctx->twin = nullptr; }
  int npipe = ctx->ws_pipes > 0 ? ctx->ws_pipes : 2;
  if (const char* e = getenv("RG_PIPELINES")) npipe = atoi(e);
  ctx->n_pipe = 1;
  if (!ctx->is_child && npipe >= 2 && !ctx->loocv && ctx->B_total > 1) {
    rg_ctx* tail = ctx;
    for (int k = 1; k < npipe && k < ctx->B_total; ++k) {
      rg_ctx* ch = nullptr;
      if (rg_create(&ch, ctx->device, nullptr) != RG_OK || !ch) break;
      ch->is_child = true;
      ch->ws_nblk = ctx->nblk_cap; ch->ws_pipes = npipe; ch->ws_budget = ctx->ws_budget;   // the same batch size as the parent
      if (rg_set_problem(ch, p) != RG_OK) { rg_destroy(ch); break; }   // e.g. not enough memory for another workspace set
      if (!ch->ev_tw_join) hipEventCreateWithFlags(&ch->ev_tw_join, hipEventDisableTiming);
      tail->twin = ch;
      tail = ch;
      ++ctx->n_pipe;
    }
    if (ctx->n_pipe > 1 && !ctx->ev_tw_fork) hipEventCreateWithFlags(&ctx->ev_tw_fork, hipEventDisableTiming);
  }
  sp_mark("all pipelines ready");
  return RG_OK;
}

int rg_set_l0_workspace(rg_ctx* ctx, int32_t max_batch_blocks, int32_t pipelines, int64_t budget_bytes) {
  if (!ctx) return RG_ERR_ARG;
  if (max_batch_blocks < 0 || pipelines < 0 || budget_bytes < 0) { ctx->err = "rg_set_l0_workspace: negative argument"; return RG_ERR_ARG; }
  ctx->ws_nblk = max_batch_blocks; ctx->ws_pipes = pipelines; ctx->ws_budget = budget_bytes;
  return RG_OK;
}

int64_t rg_w_rows(const rg_ctx* c) { return c ? c->Np : 0; }
int32_t rg_l0_batch_blocks(const rg_ctx* c) { return c ? c->nblk_cap : 0; }
int64_t rg_w_bytes(const rg_ctx* c) { return c ? c->W_bytes : 0; }
void* rg_w_device_ptr(rg_ctx* c) { return c ? c->d_W : nullptr; }

int rg_set_block_range(rg_ctx* ctx, int32_t first_block, int32_t n_blocks) {
  if (!ctx || !ctx->have_problem) return RG_ERR_STATE;
  if (first_block < 0 || n_blocks < 0 || first_block + n_blocks > ctx->B_total) { ctx->err = "rg_set_block_range: range out of bounds"; return RG_ERR_ARG; }
  if (ctx->own_W && ctx->d_W) hipFree(ctx->d_W);
  ctx->d_W = nullptr; ctx->own_W = false;
  ctx->w_b0 = first_block; ctx->w_nb = n_blocks;
  ctx->W_bytes = (int64_t)sizeof(double) * std::max(1, n_blocks) * ctx->R0 * ctx->P * ctx->Np;
  for (rg_ctx* t = ctx->twin; t; t = t->twin) { t->d_W = nullptr; t->own_W = false; t->w_b0 = first_block; t->w_nb = n_blocks; t->W_bytes = ctx->W_bytes; }
  return RG_OK;
}

int rg_set_w_buffer(rg_ctx* ctx, void* dev_ptr, int64_t bytes) {
  if (!ctx || !ctx->have_problem) return RG_ERR_STATE;
  if (bytes < ctx->W_bytes) { ctx->err = "W buffer too small"; return RG_ERR_ARG; }
  if (ctx->own_W && ctx->d_W) hipFree(ctx->d_W);
  ctx->d_W = (double*)dev_ptr;
  ctx->own_W = false;
  for (rg_ctx* t = ctx->twin; t; t = t->twin) { t->d_W = ctx->d_W; t->own_W = false; }
  return RG_OK;
}

static int ensure_W(rg_ctx* ctx) {
  if (ctx->d_W) return RG_OK;
  RG_HIP(hipMalloc((void**)&ctx->d_W, (size_t)ctx->W_bytes));
  RG_HIP(hipMemsetAsync(ctx->d_W, 0, (size_t)ctx->W_bytes, ctx->stream));
  ctx->own_W = true;
  for (rg_ctx* t = ctx->twin; t; t = t->twin) { t->d_W = ctx->d_W; t->own_W = false; }
  return RG_OK;
}

static int l0_batch(rg_ctx* ctx, int nblk, const int32_t* block_ids, const int32_t* bs,
                    const uint8_t* const* bed_rows, int64_t row_stride, int mem_kind) {
  hipStream_t st = ctx->stream;
  const int nseg = ctx->K, R0 = ctx->R0, P = ctx->P, C = ctx->C, Cv = C + P;
  const int n128 = ctx->n128, n64 = ctx->n64, rtot = ctx->rtot;
  const int64_t bytes_row = (ctx->Nfile + 3) / 4;
  const int64_t raw_blk = (int64_t)ctx->bs_max * ctx->raw_ld, pk_blk = (int64_t)n128 * ctx->pk_ld;
  const int64_t msz = (int64_t)rtot * n64;
  RG_HIP(hipMemcpyAsync(ctx->d_bs, bs, sizeof(int32_t) * nblk, hipMemcpyHostToDevice, st));
  RG_HIP(hipMemcpyAsync(ctx->d_blockid, block_ids, sizeof(int32_t) * nblk, hipMemcpyHostToDevice, st));
  {
    StageTimer t(ctx, &ctx->tm.ms_prep);
    // host rows are staged into d_raw; rows that already live in device memory are read in place
    std::vector<const uint8_t*>& hp = ctx->h_rawptr;
    hp.resize(nblk);
    int64_t ld = row_stride;
    if (mem_kind == RG_MEM_DEVICE) {
      for (int b = 0; b < nblk; ++b) hp[b] = bed_rows[b];
    } else {
      ld = ctx->raw_ld;
      if (!ctx->d_raw) {
        const int rca = dev_alloc(ctx, &ctx->d_raw, (size_t)ctx->nblk_cap * ctx->bs_max * ctx->raw_ld + 16);
        if (rca) return rca;
      }
      // A block whose rows are no further apart than the staging pitch travels as ONE linear copy and is read at the caller's pitch
      // (k_bed_prep_rows takes rows at any byte alignment): a pitched host -> device copy of 1,000 rows of 125 KB runs at 13 GB/s, a
      // linear one at the PCIe rate (55 GB/s; tools/ingest_probe.cpp) -- at 500,000 samples that was most of a run from files.
      const bool linear = row_stride <= ctx->raw_ld;
      if (linear) ld = row_stride;
      for (int b = 0; b < nblk; ++b) {
        if (linear)
          RG_HIP(hipMemcpyAsync(ctx->d_raw + (int64_t)b * raw_blk, bed_rows[b], (size_t)(bs[b] - 1) * row_stride + bytes_row, hipMemcpyHostToDevice, st));
        else
          RG_HIP(hipMemcpy2DAsync(ctx->d_raw + (int64_t)b * raw_blk, ctx->raw_ld, bed_rows[b], row_stride, bytes_row,
                                  bs[b], hipMemcpyHostToDevice, st));
        hp[b] = ctx->d_raw + (int64_t)b * raw_blk;
      }
    }
    if (mem_kind != RG_MEM_DEVICE) {   // the caller's host buffers are free again once this event has passed (rg_ingest_fence)
      if (!ctx->ev_ingest) RG_HIP(hipEventCreateWithFlags(&ctx->ev_ingest, hipEventDisableTiming));
      RG_HIP(hipEventRecord(ctx->ev_ingest, st));
      ctx->ingest_pending = true;
    }
    RG_HIP(hipMemcpyAsync(ctx->d_rawptr, hp.data(), sizeof(uint8_t*) * nblk, hipMemcpyHostToDevice, st));
    rg_launch_bed_prep(st, ctx->d_rawptr, ld, ctx->d_pk, ctx->pk_ld, pk_blk, ctx->d_bs,
                       nblk, n128, ctx->d_act, ctx->seg, ctx->Np, ctx->ref_first, ctx->n_active,
                       ctx->d_mu, ctx->d_nmiss, ctx->gram_fp4 ? ctx->d_pk4 : nullptr, ctx->pk4_ld,
                       (int64_t)n128 * ctx->pk4_ld);
  }
  {
    StageTimer t(ctx, &ctx->tm.ms_xy);
    if (ctx->d_vd)
      rg_launch_xy_i8(st, ctx->d_pk, ctx->pk_ld, pk_blk, ctx->d_bs, ctx->d_nmiss, nblk, n128, ctx->seg, ctx->d_vd, ctx->d_vsc, ctx->Np, Cv,
                      ctx->d_xyS, ctx->d_xypart);
    else
      rg_launch_geno_xy(st, ctx->d_pk, ctx->pk_ld, pk_blk, ctx->d_bs, ctx->d_nmiss, nblk, n128, ctx->d_V, ctx->Np, Cv,
                        ctx->d_chunk_pos, ctx->d_chunk_len, ctx->xy_nchunk, ctx->d_xypart);
  }
  {
    StageTimer t(ctx, &ctx->tm.ms_gram);
    if (ctx->gram_fp4)
      rg_launch_gram_fp4_blocks(st, ctx->d_pk4, ctx->pk4_ld, (int64_t)n128 * ctx->pk4_ld, nblk, n128, ctx->seg, ctx->d_S);
    rg_launch_gram_blocks(st, ctx->d_pk, ctx->pk_ld, pk_blk, nblk, n128, ctx->seg, ctx->d_nmiss, ctx->d_S,
                          ctx->gram_fp4 ? 1 : 0);
    ctx->tm.n_gram_launches += 1;
  }
  // K-fold level 0 with room in the padding rows of the blocks' last tiles: the right-hand sides are embedded in the systems
  // (chol.hip); RG_NO_EMBED=1 keeps them in a tile row of their own
  static const bool no_embed = getenv("RG_NO_EMBED") && atoi(getenv("RG_NO_EMBED")) != 0;
  const int embed = (!ctx->loocv && !no_embed && ctx->bs_max + P <= n64) ? P : 0;
  {
    StageTimer t(ctx, &ctx->tm.ms_assemble);
    AsmArgs a;
    a.nblk = nblk; a.nseg = nseg; a.n128 = n128; a.n64 = n64; a.rtot = rtot; a.C = C; a.P = P; a.Cv = Cv;
    a.nchunk = ctx->d_vd ? nseg : ctx->xy_nchunk; a.n_analyzed = ctx->n_analyzed; a.bs = ctx->d_bs;
    a.chunk_seg = ctx->d_vd ? ctx->d_segid : ctx->d_chunk_seg; a.part = ctx->d_xypart; a.mu = ctx->d_mu; a.S = ctx->d_S;
    a.nmiss = ctx->d_nmiss; a.Q = ctx->d_Q; a.XtY = ctx->d_XtY; a.F = ctx->d_F; a.Bm = ctx->d_Bm;
    a.BQ = ctx->d_BQ; a.GYt = ctx->d_GYt; a.sc = ctx->d_sc; a.fold = ctx->d_fold; a.sum = ctx->d_sum;
    a.info = ctx->d_info;
    a.diff_mode = ctx->loocv ? 0 : 1;
    a.embed = embed ? 1 : 0;
    rg_launch_rowstats(st, a);
    rg_launch_assemble(st, a);
  }
  if (ctx->loocv) {
    LoocvArgs la;
    la.nblk = nblk; la.R0 = R0; la.P = P; la.C = C; la.n128 = n128; la.n64 = n64; la.rtot = (int)ctx->rtot_wk;
    la.row_g0 = rtot; la.Np = ctx->Np; la.pk_ld = ctx->pk_ld; la.pk_blk_stride = pk_blk; la.pk = ctx->d_pk;
    la.mu = ctx->d_mu; la.sc = ctx->d_sc; la.Bm = ctx->d_Bm; la.V = ctx->d_V; la.maskp = ctx->d_maskp;
    la.neff = ctx->d_neff; la.bs = ctx->d_bs; la.blockid = ctx->d_blockid; la.wk = ctx->d_wk; la.gt = ctx->d_gt;
    la.W = rg_w_base(ctx);
    {
      StageTimer t(ctx, &ctx->tm.ms_chol);
      int max_bs = 0;
      for (int b = 0; b < nblk; ++b) max_bs = std::max(max_bs, (int)bs[b]);
      const int rcl = rg_l0_loocv_tri(ctx, st, la, max_bs, ctx->d_sum, rtot, ctx->d_triws, ctx->d_zt, ctx->loo_chunk,
                                      [&](int64_t pos0, int64_t len) { la.gt_pos0 = pos0; la.gt_len = len; rg_launch_decode_gt(st, la); });
      if (rcl) return rcl;
    }
    {
      StageTimer t(ctx, &ctx->tm.ms_pred);
      rg_launch_l0_loocv(st, la, ctx->d_lpart, ctx->d_lpart + (size_t)nblk * R0 * P * 64, 64);
    }
  } else {
  {
    StageTimer t(ctx, &ctx->tm.ms_chol);
    // d_fold[(blk, f)] holds the training-fold system of fold f (assemble.hip diff_mode): one source matrix per
    // (block, fold), shared by the R0 shifted systems that are co-located on one XCD for their first touch.
    // Round 6 (panel-128 kernels, chol_p128.h): a panel launch ends with a tail -- its long items (tile + diagonal block + factorization of a
    // successor) finish while slots stand empty -- and launch j + 1 of a system only needs launch j of THAT system: ranges of the batch on
    // streams of their own fill each other's tails: RG_CHOL_PARTS=2 takes a batch ALONE from 11.7 to 11.5 ms (0.429 -> 0.436 of the matrix peak),
    // but a STEP from 30.2 to 32.1 ms -- the two level-0 pipelines already put another batch's kernels into those tails, more streams only add
    // contention -- so the default is one range.  (Round 5's RG_CHOL_SPLIT -- the systems past the last full round of k_chol_gfact as a second range -- went with that kernel's
    // default role.)  The ranges start at multiples of 8 R0 systems: the placement groups of xcd_affine.
    const int nsys_all = nblk * nseg * R0;
    static const int parts_env = getenv("RG_CHOL_PARTS") ? atoi(getenv("RG_CHOL_PARTS")) : 1;
    int parts = std::max(1, std::min(4, parts_env));
    while (parts > 1 && nsys_all < parts * 8 * 8 * R0) --parts;
    if (parts > 1) {
      const int grp = 8 * R0, ngrp = (nsys_all + grp - 1) / grp;
      if (!ctx->ev_fork) {
        RG_HIP(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
        for (int k = 0; k < 3; ++k) {
          RG_HIP(hipStreamCreateWithFlags(&ctx->st_part[k], hipStreamNonBlocking));
          RG_HIP(hipEventCreateWithFlags(&ctx->ev_part[k], hipEventDisableTiming));
        }
      }
      RG_HIP(hipEventRecord(ctx->ev_fork, st));
      for (int k = 0; k < parts; ++k) {
        const int s0 = std::min(nsys_all, (ngrp * k / parts) * grp), s1 = std::min(nsys_all, (ngrp * (k + 1) / parts) * grp);
        if (s1 <= s0) continue;
        hipStream_t sk = k == 0 ? st : ctx->st_part[k - 1];
        if (k > 0) RG_HIP(hipStreamWaitEvent(sk, ctx->ev_fork, 0));
        rg_launch_chol_solve_formed_x(sk, ctx->d_fold, msz, nullptr, 0, 1, ctx->d_lambda, R0, ctx->d_bs, 0, nblk * nseg,
                                      ctx->d_wk + (int64_t)s0 * msz, msz, n64, embed ? 0 : rtot - n64, P, ctx->d_dinv + rg_chol_ws_doubles((size_t)s0, n64),
                                      ctx->d_info + 1, &ctx->tm.n_chol_launches, 0, nullptr, 0, 0, nseg, s0, s1 - s0, 0, embed);
        if (k > 0) {
          RG_HIP(hipEventRecord(ctx->ev_part[k - 1], sk));
          RG_HIP(hipStreamWaitEvent(st, ctx->ev_part[k - 1], 0));
        }
      }
    } else
    // (the per-column kernels of level 1 -- k_chol_diag / k_chol_panel -- measured on these 1,375 systems: 17.7 ms per batch against 15.5 ms
    // for the group-of-four path without embedded right-hand sides, 14.5 ms with them)
    rg_launch_chol_solve_formed_x(st, ctx->d_fold, msz, nullptr, 0, 1, ctx->d_lambda, R0, ctx->d_bs, 0, nblk * nseg,
                                  ctx->d_wk, msz, n64, embed ? 0 : rtot - n64, P, ctx->d_dinv, ctx->d_info + 1,
                                  &ctx->tm.n_chol_launches, 0, nullptr, 0, 0, nseg, 0, -1, 0, embed);
  }
  {
    StageTimer t(ctx, &ctx->tm.ms_pred);
    PredArgs pa;
    pa.nblk = nblk; pa.nseg = nseg; pa.R0 = R0; pa.P = P; pa.C = C; pa.n128 = n128; pa.n64 = n64;
    pa.rtot = rtot; pa.B_total = ctx->B_total; pa.Np = ctx->Np; pa.pk_ld = ctx->pk_ld;
    pa.pk_blk_stride = pk_blk; pa.seg = ctx->seg; pa.pk = ctx->d_pk; pa.mu = ctx->d_mu; pa.sc = ctx->d_sc;
    pa.Bm = ctx->d_Bm; pa.wk = ctx->d_wk; pa.V = ctx->d_V; pa.maskp = ctx->d_maskp;
    pa.keptp = ctx->d_keptp; pa.bs = ctx->d_bs; pa.blockid = ctx->d_blockid; pa.neff = ctx->d_neff; pa.nmiss = ctx->d_nmiss;
    pa.beta = ctx->d_beta; pa.cb = ctx->d_cb; pa.psum = ctx->d_psum; pa.W = rg_w_base(ctx);
    pa.bplanes = ctx->d_bplanes; pa.bsc = ctx->d_bsc; pa.pkT = ctx->d_pkT;
    pa.embed = embed ? 1 : 0;
    rg_launch_l0_pred_impl(st, pa, ChunkTab{ctx->d_c1k_seg, ctx->d_c1k_pos, ctx->d_c1k_len, ctx->n_c1k},
                           ChunkTab{ctx->d_c256_seg, ctx->d_c256_pos, ctx->d_c256_len, ctx->n_c256}, ctx->d_pstat);
  }
  }
  for (int b = 0; b < nblk; ++b) ctx->block_done[block_ids[b]] = 1;
  return RG_OK;
}

int rg_l0_blocks(rg_ctx* ctx, int32_t nblk, const int32_t* block_ids, const int32_t* bs,
                 const uint8_t* const* bed_rows, int64_t row_stride, int mem_kind) {
  if (!ctx || !ctx->have_problem) return RG_ERR_STATE;
  hipSetDevice(ctx->device);
  if (nblk < 1 || !block_ids || !bs || !bed_rows) { ctx->err = "rg_l0_blocks: bad arguments"; return RG_ERR_ARG; }
  if (row_stride < (ctx->Nfile + 3) / 4) { ctx->err = "rg_l0_blocks: row_stride < ceil(N_file/4)"; return RG_ERR_ARG; }
  for (int b = 0; b < nblk; ++b) {
    if (bs[b] < 1 || bs[b] > ctx->bs_max) { ctx->err = "rg_l0_blocks: block size out of range"; return RG_ERR_ARG; }
    if (block_ids[b] < ctx->w_b0 || block_ids[b] >= ctx->w_b0 + ctx->w_nb) { ctx->err = "rg_l0_blocks: block id out of range"; return RG_ERR_ARG; }
  }
  int rc = ensure_W(ctx);
  if (rc) return rc;
  // balanced batches: ceil(nblk / cap) batches of (almost) equal size.
  // Several pipelines: batch number (pipe_rr + ib) runs in context ((pipe_rr + ib) mod n_pipe) of the chain, each on its own
  // stream.  The children are ordered after everything queued on ctx->stream before the FIRST level-0 call of a sequence
  // (fork event) and joined back lazily -- by rg_sync and every entry point that reads W -- so that consecutive calls (a
  // driver streaming one batch per call from a reader thread) keep alternating pipelines and overlap, exactly like the
  // batches of one big call.  The per-stage timing mode keeps a single pipeline so that its HIP-event brackets stay meaningful.
  const int npipe = (ctx->twin && !ctx->timing) ? ctx->n_pipe : 1;
  int nbatch = (nblk + ctx->nblk_cap - 1) / ctx->nblk_cap;
  {
    // RG_BATCH_ROUND=1 rounds the number of batches of a call up to a multiple of the pipelines
    static const bool round_up = getenv("RG_BATCH_ROUND") && atoi(getenv("RG_BATCH_ROUND")) != 0;
    if (npipe > 1 && nblk >= 2 * npipe && round_up) nbatch = (nbatch + npipe - 1) / npipe * npipe;
  }
  const int per = (nblk + nbatch - 1) / nbatch;
  const bool multi = npipe > 1;
  if (multi && !ctx->join_pending) {
    RG_HIP(hipEventRecord(ctx->ev_tw_fork, ctx->stream));
    for (rg_ctx* t = ctx->twin; t; t = t->twin) {
      t->d_W = ctx->d_W; t->own_W = false;
      RG_HIP(hipStreamWaitEvent(t->stream, ctx->ev_tw_fork, 0));
    }
    ctx->join_pending = true;
  }
  for (int b0 = 0; b0 < nblk; b0 += per) {
    const int nb = std::min(per, nblk - b0);
    // within a context the small H2D descriptor copies of the next batch must not overtake the kernels of the
    // previous one: everything of a pipeline is ordered on its stream.
    rg_ctx* c = ctx;
    if (multi) {
      for (int k = ctx->pipe_rr % npipe; k > 0 && c->twin; --k) c = c->twin;
      ++ctx->pipe_rr;
    }
    rc = l0_batch(c, nb, block_ids + b0, bs + b0, bed_rows + b0, row_stride, mem_kind);
    if (rc) { if (c != ctx) ctx->err = c->err; return rc; }
    if (c != ctx) for (int b = 0; b < nb; ++b) ctx->block_done[block_ids[b0 + b]] = 1;
  }
  return RG_OK;
}

// joins the level-0 pipelines back onto ctx->stream (see rg_l0_blocks)
static int join_pipes(rg_ctx* ctx) {
  if (!ctx->join_pending) return RG_OK;
  for (rg_ctx* t = ctx->twin; t; t = t->twin) {
    RG_HIP(hipEventRecord(t->ev_tw_join, t->stream));
    RG_HIP(hipStreamWaitEvent(ctx->stream, t->ev_tw_join, 0));
  }
  ctx->join_pending = false;
  ctx->pipe_rr = 0;
  return RG_OK;
}

int rg_l0_blocks_f64(rg_ctx* ctx, int32_t nblk, const int32_t* block_ids, const int32_t* bs, const double* const* rows,
                     int64_t row_stride, int mem_kind) {
  if (!ctx || !ctx->have_problem) return RG_ERR_STATE;
  hipSetDevice(ctx->device);
  if (nblk < 1 || !block_ids || !bs || !rows) { ctx->err = "rg_l0_blocks_f64: bad arguments"; return RG_ERR_ARG; }
  if (row_stride < ctx->Nfile) { ctx->err = "rg_l0_blocks_f64: row_stride < N_file"; return RG_ERR_ARG; }
  for (int b = 0; b < nblk; ++b) {
    if (block_ids[b] < ctx->w_b0 || block_ids[b] >= ctx->w_b0 + ctx->w_nb) { ctx->err = "rg_l0_blocks_f64: block id out of range"; return RG_ERR_ARG; }
    if (bs[b] < 1 || bs[b] > ctx->bs_max) { ctx->err = "rg_l0_blocks_f64: block size out of range"; return RG_ERR_ARG; }
    if (!rows[b]) { ctx->err = "rg_l0_blocks_f64: null row pointer"; return RG_ERR_ARG; }
  }
  { const int rcw = ensure_W(ctx); if (rcw) return rcw; }
  { const int rcj = join_pipes(ctx); if (rcj) return rcj; }
  const int rc = rg_l0_blocks_f64_impl(ctx, nblk, block_ids, bs, rows, row_stride, mem_kind);
  if (rc == RG_OK && mem_kind != RG_MEM_DEVICE) {   // coarse: the rows are free once the whole call has run
    if (!ctx->ev_ingest) RG_HIP(hipEventCreateWithFlags(&ctx->ev_ingest, hipEventDisableTiming));
    RG_HIP(hipEventRecord(ctx->ev_ingest, ctx->stream));
    ctx->ingest_pending = true;
  }
  return rc;
}

void* rg_host_alloc(int64_t bytes) {
  void* p = nullptr;
  if (bytes <= 0 || hipHostMalloc(&p, (size_t)bytes, hipHostMallocDefault) != hipSuccess) return nullptr;
  return p;
}
void rg_host_free(void* p) { if (p) hipHostFree(p); }
int rg_host_register(void* ptr, int64_t bytes, int read_only) {
  if (!ptr || bytes <= 0) return RG_ERR_ARG;
  const unsigned flags = hipHostRegisterPortable | (read_only ? hipHostRegisterReadOnly : 0u);
  if (hipHostRegister(ptr, (size_t)bytes, flags) != hipSuccess) { (void)hipGetLastError(); return RG_ERR_HIP; }
  return RG_OK;
}
int rg_host_unregister(void* ptr) { return (ptr && hipHostUnregister(ptr) == hipSuccess) ? RG_OK : RG_ERR_HIP; }

void* rg_stage_alloc(rg_ctx* ctx, int64_t bytes, double max_frac_of_free) {
  if (!ctx || bytes <= 0) return nullptr;
  hipSetDevice(ctx->device);
  size_t fr = 0, tot = 0;
  if (hipMemGetInfo(&fr, &tot) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  if (max_frac_of_free > 0 && (double)bytes > max_frac_of_free * (double)fr) return nullptr;
  void* p = nullptr;
  if (hipMalloc(&p, (size_t)bytes) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  return p;
}
int rg_stage_copy(rg_ctx* ctx, void* dev_dst, const void* host_src, int64_t bytes) {
  if (!ctx || !dev_dst || !host_src || bytes < 0) return RG_ERR_ARG;
  if (bytes == 0) return RG_OK;
  hipSetDevice(ctx->device);
  // a stream that does not synchronise with the null stream or the context's: the copies run beside rg_set_problem's uploads and level 0
  if (!ctx->st_stage && hipStreamCreateWithFlags(&ctx->st_stage, hipStreamNonBlocking) != hipSuccess) { ctx->err = "rg_stage_copy: no stream"; return RG_ERR_HIP; }
  if (hipMemcpyAsync(dev_dst, host_src, (size_t)bytes, hipMemcpyHostToDevice, ctx->st_stage) != hipSuccess ||
      hipStreamSynchronize(ctx->st_stage) != hipSuccess) {
    (void)hipGetLastError();
    return RG_ERR_HIP;
  }
  return RG_OK;
}
int rg_stage_fits(rg_ctx* ctx, int64_t bytes) {
  if (!ctx) return 0;
  hipSetDevice(ctx->device);
  size_t fr = 0, tot = 0;
  if (hipMemGetInfo(&fr, &tot) != hipSuccess) { (void)hipGetLastError(); return 0; }
  return bytes <= (int64_t)fr ? 1 : 0;
}
void rg_stage_free(rg_ctx* ctx, void* dev_ptr) {
  if (!ctx || !dev_ptr) return;
  hipSetDevice(ctx->device);
  hipFree(dev_ptr);
}

int rg_ingest_fence(rg_ctx* ctx) {
  if (!ctx) return RG_ERR_ARG;
  hipSetDevice(ctx->device);
  for (rg_ctx* c = ctx; c; c = c->twin)
    if (c->ingest_pending) {
      if (hipEventSynchronize(c->ev_ingest) != hipSuccess) { ctx->err = "rg_ingest_fence: event wait failed"; return RG_ERR_HIP; }
      c->ingest_pending = false;
    }
  return RG_OK;
}

int rg_sync(rg_ctx* ctx) {
  if (!ctx) return RG_ERR_ARG;
  hipSetDevice(ctx->device);
  { const int rcj = join_pipes(ctx); if (rcj) return rcj; }
  RG_HIP(hipStreamSynchronize(ctx->stream));
  if (ctx->twin) {   // deferred device-side errors of the other pipelines (each child syncs the rest of the chain)
    int rc2 = rg_sync(ctx->twin);
    if (rc2) { ctx->err = ctx->twin->err; return rc2; }
  }
  if (!ctx->d_info) return RG_OK;
  int32_t info[4] = {0, 0, 0, 0};
  RG_HIP(hipMemcpy(info, ctx->d_info, sizeof(info), hipMemcpyDeviceToHost));
  if (info[0]) {
    const int j = (info[0] & 0xFFFFF) - 1, blk = info[0] >> 20;
    ctx->err = "!! Uh-oh, SNP #" + std::to_string(j) + " of batch block " + std::to_string(blk) + " has low variance.";
    hipMemset(ctx->d_info, 0, sizeof(info));
    return RG_ERR_LOW_VARIANCE;
  }
  if (info[2]) {  // Geno.cpp:1799-1800, :1819-1820
    ctx->err = "there is a variant in the block that has a value not in [0,2] or missing";
    hipMemset(ctx->d_info, 0, sizeof(info));
    return RG_ERR_ARG;
  }
  if (info[1]) {
    ctx->err = "ridge system is not positive definite";
    hipMemset(ctx->d_info, 0, sizeof(info));
    return RG_ERR_NOT_SPD;
  }
  return RG_OK;
}

int rg_l0_get_w(rg_ctx* ctx, int32_t block_id, int32_t pheno, double* out_host) {
  if (!ctx || !ctx->have_problem || !ctx->d_W) return RG_ERR_STATE;
  if (block_id < ctx->w_b0 || block_id >= ctx->w_b0 + ctx->w_nb || pheno < 0 || pheno >= ctx->P || !out_host) return RG_ERR_ARG;
  hipSetDevice(ctx->device);
  { const int rcj = join_pipes(ctx); if (rcj) return rcj; }
  double* tmp = nullptr;
  RG_HIP(hipMalloc((void**)&tmp, sizeof(double) * ctx->N * ctx->R0));
  rg_launch_w_gather(ctx->stream, rg_w_base(ctx), ctx->Np, ctx->P, pheno, block_id * ctx->R0, ctx->R0, ctx->d_posc, ctx->N, tmp);
  hipError_t e = hipMemcpyAsync(out_host, tmp, sizeof(double) * ctx->N * ctx->R0, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  hipFree(tmp);
  if (e != hipSuccess) { ctx->err = hipGetErrorString(e); return RG_ERR_HIP; }
  return RG_OK;
}

int rg_l0_set_w(rg_ctx* ctx, int32_t block_id, int32_t pheno, const double* in_host) {
  if (!ctx || !ctx->have_problem) return RG_ERR_STATE;
  if (block_id < ctx->w_b0 || block_id >= ctx->w_b0 + ctx->w_nb || pheno < 0 || pheno >= ctx->P || !in_host) return RG_ERR_ARG;
  hipSetDevice(ctx->device);
  int rc = ensure_W(ctx);
  if (rc) return rc;
  if ((rc = join_pipes(ctx))) return rc;
  double* tmp = nullptr;
  RG_HIP(hipMalloc((void**)&tmp, sizeof(double) * ctx->N * ctx->R0));
  hipError_t e = hipMemcpyAsync(tmp, in_host, sizeof(double) * ctx->N * ctx->R0, hipMemcpyHostToDevice, ctx->stream);
  rg_launch_w_scatter(ctx->stream, rg_w_base(ctx), ctx->Np, ctx->P, pheno, block_id * ctx->R0, ctx->R0, ctx->d_posc, ctx->N, tmp);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  hipFree(tmp);
  if (e != hipSuccess) { ctx->err = hipGetErrorString(e); return RG_ERR_HIP; }
  ctx->block_done[block_id] = 1;
  return RG_OK;
}

int rg_l1_qt(rg_ctx* ctx, int32_t n_ridge_l1, const double* tau, int32_t nchr,
             const int32_t* cols_per_chr, double* cumsum_out, int32_t* best_out, double* pred_out) {
  if (!ctx) return RG_ERR_ARG;
  hipSetDevice(ctx->device);
  if (!tau || nchr < 1 || !cols_per_chr || !cumsum_out || !best_out || !pred_out) { ctx->err = "rg_l1_qt: bad arguments"; return RG_ERR_ARG; }
  int rc = rg_sync(ctx);
  if (rc) return rc;
  return rg_l1_qt_impl(ctx, n_ridge_l1, tau, nchr, cols_per_chr, cumsum_out, best_out, pred_out);
}

int rg_l1_qt_loocv(rg_ctx* ctx, int32_t n_ridge_l1, const double* tau, int32_t nchr,
                   const int32_t* cols_per_chr, double* cumsum_out, int32_t* best_out, double* pred_out) {
  if (!ctx) return RG_ERR_ARG;
  hipSetDevice(ctx->device);
  if (!tau || nchr < 1 || !cols_per_chr || !cumsum_out || !best_out || !pred_out) { ctx->err = "rg_l1_qt_loocv: bad arguments"; return RG_ERR_ARG; }
  int rc = rg_sync(ctx);
  if (rc) return rc;
  return rg_l1_qt_loocv_impl(ctx, n_ridge_l1, tau, nchr, cols_per_chr, cumsum_out, best_out, pred_out);
}

int rg_l1_bt(rg_ctx* ctx, int32_t n_ridge_l1, const double* tau, const double* yraw, const double* offset,
             const rg_bt_options* opt, int32_t nchr, const int32_t* cols_per_chr, double* cumsum_out,
             int32_t* converged_out, int32_t* best_out, double* pred_out) {
  if (!ctx) return RG_ERR_ARG;
  hipSetDevice(ctx->device);
  if (!tau || !yraw || !offset || nchr < 1 || !cols_per_chr || !cumsum_out || !converged_out || !best_out || !pred_out) {
    ctx->err = "rg_l1_bt: bad arguments"; return RG_ERR_ARG;
  }
  int rc = rg_sync(ctx);
  if (rc) return rc;
  return rg_l1_bt_impl(ctx, n_ridge_l1, tau, yraw, offset, opt, nchr, cols_per_chr, cumsum_out, converged_out,
                       best_out, pred_out);
}

int rg_l1_cox(rg_ctx* ctx, int32_t pheno, int32_t n_ridge_l1, const double* time, const double* event, const double* offset,
              const rg_cox_options* opt, int32_t nchr, const int32_t* cols_per_chr, double* tau_out, double* deviance_out,
              int32_t* converged_out, int32_t* best_out, double* pred_out) {
  if (!ctx) return RG_ERR_ARG;
  hipSetDevice(ctx->device);
  if (!time || !event || !offset || nchr < 1 || !cols_per_chr || !tau_out || !deviance_out || !converged_out || !best_out || !pred_out ||
      pheno < 0 || pheno >= ctx->P) {
    ctx->err = "rg_l1_cox: bad arguments"; return RG_ERR_ARG;
  }
  int rc = rg_sync(ctx);
  if (rc) return rc;
  return rg_l1_cox_impl(ctx, pheno, n_ridge_l1, time, event, offset, opt, nchr, cols_per_chr, tau_out, deviance_out, converged_out,
                        best_out, pred_out);
}

int rg_set_loco_output(rg_ctx* ctx, int32_t nchrom, const int32_t* chrom_ids, int32_t n_ids) {
  if (!ctx) return RG_ERR_ARG;
  if (nchrom <= 0) { ctx->loco_nchrom = 0; ctx->loco_chrom.clear(); return RG_OK; }
  if (!chrom_ids || n_ids < 1) { ctx->err = "rg_set_loco_output: chromosome ids missing"; return RG_ERR_ARG; }
  for (int k = 0; k < n_ids; ++k)
    if (chrom_ids[k] < 1 || chrom_ids[k] > nchrom) { ctx->err = "rg_set_loco_output: chromosome id out of range"; return RG_ERR_ARG; }
  ctx->loco_nchrom = nchrom;
  ctx->loco_chrom.assign(chrom_ids, chrom_ids + n_ids);
  return RG_OK;
}

int rg_set_l1_view(rg_ctx* ctx, const void* w_dev, int32_t pheno_begin, int32_t pheno_count) {
  if (!ctx || !ctx->have_problem) return RG_ERR_STATE;
  if (pheno_begin < 0 || pheno_count < 1 || pheno_begin + pheno_count > ctx->P) { ctx->err = "rg_set_l1_view: phenotype range out of bounds"; return RG_ERR_ARG; }
  // w_dev == NULL with a sub-range: level 1 of those phenotypes on the context's own W (a caller that takes the phenotypes one at a
  // time to write each one's files while the next is computed)
  ctx->v_W = (const double*)w_dev; ctx->v_p0 = pheno_begin; ctx->v_np = pheno_count;
  return RG_OK;
}

int rg_set_collective(rg_ctx* ctx, int32_t world, int32_t rank, rg_allreduce_fn fn, void* user) {
  if (!ctx) return RG_ERR_ARG;
  if (world < 1 || rank < 0 || rank >= world || (world > 1 && !fn)) { ctx->err = "rg_set_collective: bad arguments"; return RG_ERR_ARG; }
  ctx->coll_world = world; ctx->coll_rank = rank; ctx->coll_allreduce = fn; ctx->coll_user = user;
  return RG_OK;
}

int rg_enable_timing(rg_ctx* ctx, int on) {
  if (!ctx) return RG_ERR_ARG;
  ctx->timing = on != 0;
  memset(&ctx->tm, 0, sizeof(ctx->tm));
  return RG_OK;
}
int rg_get_timing(rg_ctx* ctx, rg_timing* out) {
  if (!ctx || !out) return RG_ERR_ARG;
  *out = ctx->tm;
  return RG_OK;
}

// ---- single-kernel entry points ------------------------------------------------------------------------------
int rg_k_gram_i8(void* stream, const uint8_t* A, int64_t lda, int a_miss, const uint8_t* B,
                 int64_t ldb, int b_miss, int32_t m, int32_t n, int64_t k_bytes, int32_t* C, int64_t ldc) {
  if (!A || !B || !C || m < 1 || n < 1 || k_bytes < 16 || (k_bytes & 15) || (lda & 15) || (ldb & 15)) return RG_ERR_ARG;
  rg_launch_gram_generic((hipStream_t)stream, A, lda, a_miss, B, ldb, b_miss, m, n, k_bytes, C, ldc);
  return hipGetLastError() == hipSuccess ? RG_OK : RG_ERR_HIP;
}

int rg_k_gram_fp4(void* stream, const uint8_t* A, int64_t lda, const uint8_t* B, int64_t ldb, int32_t m,
                  int32_t n, int64_t k_bytes, int32_t* C, int64_t ldc) {
  if (!A || !B || !C || m < 1 || n < 1 || k_bytes < 128 || (k_bytes & 127) || (lda & 15) || (ldb & 15)) return RG_ERR_ARG;
  rg_launch_gram_fp4_generic((hipStream_t)stream, A, lda, B, ldb, m, n, k_bytes, C, ldc);
  return hipGetLastError() == hipSuccess ? RG_OK : RG_ERR_HIP;
}

int rg_k_chol_solve(void* stream, double* mats, int64_t mat_stride, int32_t batch, int32_t n_pad,
                    int32_t rhs_pad, int32_t nrhs, double* dinv_ws, int32_t* info) {
  if (!mats || !dinv_ws || !info || batch < 1 || n_pad < 64 || (n_pad & 63) || (rhs_pad & 63) || nrhs > rhs_pad) return RG_ERR_ARG;
  rg_launch_chol_solve((hipStream_t)stream, mats, mat_stride, batch, n_pad, rhs_pad, nrhs, dinv_ws, info, nullptr, -1);
  return hipGetLastError() == hipSuccess ? RG_OK : RG_ERR_HIP;
}

int rg_k_dgemm_nt(void* stream, const double* A, int64_t lda, const double* B, int64_t ldb, int32_t m,
                  int32_t n, int64_t k, double* C, int64_t ldc) {
  if (!A || !B || !C || (m & 63) || (n & 63) || (k & 63) || m < 64 || n < 64 || k < 64) return RG_ERR_ARG;
  rg_launch_dgemm_nt((hipStream_t)stream, A, lda, B, ldb, m, n, k, C, ldc);
  return hipGetLastError() == hipSuccess ? RG_OK : RG_ERR_HIP;
}

}  // extern "C"
