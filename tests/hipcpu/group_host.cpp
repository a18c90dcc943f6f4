// TEST SCAFFOLDING, not product code (see hip/hip_runtime.h beside this file).  What rg_group.hip needs from rg_api.hip, for a test that runs the
// multi-GPU hand-off of the level-0 predictors with host threads as ranks and host memory as device memory: contexts that hold a W
// ([blocks of the rank's range * R0][P][Np]) and nothing else, rg_sync / rg_set_l1_view / rg_set_collective as in rg_api.hip.
#include "../../regenie_amd/csrc/rg_internal.h"

extern "C" {
int rg_sync(rg_ctx*) { return RG_OK; }
int rg_set_l1_view(rg_ctx* ctx, const void* w_dev, int32_t pheno_begin, int32_t pheno_count) {
  if (!ctx || !ctx->have_problem) return RG_ERR_STATE;
  if (pheno_begin < 0 || pheno_count < 1 || pheno_begin + pheno_count > ctx->P) { ctx->err = "rg_set_l1_view: phenotype range out of bounds"; return RG_ERR_ARG; }
  ctx->v_W = (const double*)w_dev; ctx->v_p0 = pheno_begin; ctx->v_np = pheno_count;
  return RG_OK;
}
int rg_set_collective(rg_ctx* ctx, int32_t world, int32_t rank, rg_allreduce_fn fn, void* user) {
  if (!ctx) return RG_ERR_ARG;
  if (world < 1 || rank < 0 || rank >= world || (world > 1 && !fn)) { ctx->err = "rg_set_collective: bad arguments"; return RG_ERR_ARG; }
  ctx->coll_world = world; ctx->coll_rank = rank; ctx->coll_allreduce = fn; ctx->coll_user = user;
  return RG_OK;
}
// a context whose W holds blocks [w_b0, w_b0 + w_nb)
rg_ctx* emu_ctx_create(int device, int R0, int P, int64_t Np, int B_total, int w_b0, int w_nb) {
  rg_ctx* c = new rg_ctx();
  c->device = device; c->R0 = R0; c->P = P; c->Np = Np; c->B_total = B_total; c->w_b0 = w_b0; c->w_nb = w_nb;
  c->have_problem = true;
  c->stream = reinterpret_cast<hipStream_t>(c);
  c->block_done.assign(B_total, 0);
  c->d_W = (double*)malloc(sizeof(double) * (size_t)w_nb * R0 * P * Np);
  return c;
}
void emu_ctx_destroy(rg_ctx* c) { free(c->d_W); delete c; }
double* emu_ctx_w(rg_ctx* c) { return c->d_W; }
const char* emu_ctx_error(rg_ctx* c) { return c->err.c_str(); }
void emu_ctx_view(rg_ctx* c, const double** w, int32_t* p0, int32_t* np, int32_t* world, int32_t* rank) {
  *w = c->v_W; *p0 = c->v_p0; *np = c->v_np; *world = c->coll_world; *rank = c->coll_rank;
}
int emu_ctx_allreduce(rg_ctx* c, double* buf, int64_t n) { return c->coll_allreduce ? c->coll_allreduce(c->coll_user, buf, n) : -1; }
int emu_blocks_done(rg_ctx* c) { int s = 0; for (int b : c->block_done) s += b; return s; }
}
