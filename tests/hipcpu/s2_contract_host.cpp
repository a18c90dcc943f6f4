// TEST SCAFFOLDING, not product code (see hip/hip_runtime.h beside this file).  Host stand-ins for the entries of regenie_amd/csrc/step2_qt.hip
// that step2_bt.hip calls, written from their contract in include/rg_step2.h (plain fp64 loops, no matrix cores): the context, the sparse rule,
// rg_s2_set_columns, rg_s2_contract_packed / rg_s2_contract_int -- sums, squares, counts, and the block's rows left staged where the corrections
// of step2_bt.hip read them (hard calls: --ref-first applied, positions past n as "0 copies"; uint16 dosages: rows of (n + 7) / 8 * 8 entries).
// With step2_bt.hip compiled against the shim this makes a library that tests/ drive through regenie_amd.step2 like the real one.
#include "../../regenie_amd/csrc/step2_internal.h"

namespace {
struct HostCols { std::vector<double> cols; int ncol = 0, nsq = 0; };
HostCols& cols_of(rg_s2_ctx* ctx) { return *reinterpret_cast<HostCols*>(ctx->gV); }
}  // namespace

extern "C" {

int rg_s2_create(rg_s2_ctx** out, int device, int64_t n, int32_t C, int32_t P) {
  if (!out || n < 1 || C < 1 || P < 1) return RG_S2_ERR_ARG;
  rg_s2_ctx* ctx = new rg_s2_ctx();
  ctx->dev = device; ctx->n = n; ctx->C = C; ctx->P = P;
  ctx->st = reinterpret_cast<hipStream_t>(ctx);                       // (the entries take a null stream for "context was not created")
  ctx->gV = reinterpret_cast<double*>(new HostCols());
  *out = ctx;
  return RG_S2_OK;
}
void rg_s2_destroy(rg_s2_ctx* ctx) {
  if (!ctx) return;
  rg_s2_bt_free(ctx);
  delete &cols_of(ctx);
  for (void* p : ctx->buf) free(p);
  for (void* p : ctx->pbuf) free(p);
  delete ctx;
}
const char* rg_s2_last_error(const rg_s2_ctx* ctx) { return ctx ? ctx->err.c_str() : "no context"; }
double rg_s2_last_kernel_ms(const rg_s2_ctx* ctx) { return ctx ? ctx->last_ms : 0.0; }
int rg_s2_set_sparse_rule(rg_s2_ctx* ctx, int64_t n_samples, double prop_zero_thr, int32_t zero_count_rule) {
  ctx->rule_n = n_samples; ctx->rule_thr = prop_zero_thr; ctx->rule_zero_count = zero_count_rule ? 1 : 0;
  return RG_S2_OK;
}
int rg_s2_set_columns(rg_s2_ctx* ctx, int32_t n_col, const double* cols, int32_t n_sq) {
  HostCols& h = cols_of(ctx);
  h.cols.assign(cols, cols + (size_t)n_col * ctx->n);
  h.ncol = n_col; h.nsq = n_sq;
  ctx->g_ncol = n_col; ctx->g_nsq = n_sq;
  return RG_S2_OK;
}

int rg_s2_contract_packed(rg_s2_ctx* ctx, const uint8_t* rows, int64_t ld, int32_t bs, int32_t, int32_t flip, const rg_s2_contract_out* out) {
  const HostCols& h = cols_of(ctx);
  const int64_t n = ctx->n;
  const int64_t Np = (n + 128 * RG_MAX_SEG - 1) / (128 * RG_MAX_SEG) * (128 * RG_MAX_SEG), ldp = Np / 4;
  int rc = rg_s2_ensure_in(ctx, ctx->pbuf, ctx->pcap, RG_S2_Q_PK, (size_t)bs * ldp);
  if (rc) return rc;
  uint8_t* pk = (uint8_t*)ctx->pbuf[RG_S2_Q_PK];
  memset(pk, 0xFF, (size_t)bs * ldp);                                     // code 11 = 0 copies
  for (int j = 0; j < bs; ++j) {
    int32_t n1 = 0, n2 = 0, nm = 0;
    std::vector<double> g0(n), ms(n);
    for (int64_t i = 0; i < n; ++i) {
      unsigned code = (rows[(size_t)j * ld + (i >> 2)] >> (2 * (i & 3))) & 3u;
      if (flip && (code == 0 || code == 3)) code ^= 3u;                     // 00 <-> 11
      uint8_t& b = pk[(size_t)j * ldp + (i >> 2)];
      b = (uint8_t)((b & ~(3u << (2 * (i & 3)))) | (code << (2 * (i & 3))));
      g0[i] = code == 0 ? 2.0 : code == 2 ? 1.0 : 0.0;
      ms[i] = code == 1 ? 1.0 : 0.0;
      n1 += code == 2; n2 += code == 0; nm += code == 1;
    }
    if (out->counts) { int32_t* c = out->counts + (size_t)j * 4; c[0] = n1; c[1] = n2; c[2] = nm; c[3] = 0; }
    for (int c = 0; c < h.ncol; ++c) {
      const double* col = h.cols.data() + (size_t)c * n;
      double s = 0.0, sm = 0.0, s2 = 0.0;
      for (int64_t i = 0; i < n; ++i) { s += g0[i] * col[i]; sm += ms[i] * col[i]; s2 += g0[i] * g0[i] * col[i]; }
      if (out->sums) { out->sums[((size_t)j * 2) * h.ncol + c] = s; out->sums[((size_t)j * 2 + 1) * h.ncol + c] = sm; }
      if (out->sq && c < h.nsq) out->sq[(size_t)j * h.nsq + c] = s2;
    }
  }
  return RG_S2_OK;
}

int rg_s2_contract_int(rg_s2_ctx* ctx, const uint16_t* G, int64_t ld, int32_t bs, int32_t, int32_t scale, const rg_s2_contract_out* out) {
  const HostCols& h = cols_of(ctx);
  const int64_t n = ctx->n, ldg = (n + 7) / 8 * 8;
  int rc = rg_s2_ensure_in(ctx, ctx->buf, ctx->cap, RG_S2_B_G, sizeof(uint16_t) * (size_t)bs * ldg);
  if (rc) return rc;
  uint16_t* st = (uint16_t*)ctx->buf[RG_S2_B_G];
  for (int j = 0; j < bs; ++j) {
    double sg = 0, sg2 = 0, no = 0, nz = 0;
    std::vector<double> g0(n), ms(n);
    for (int64_t i = 0; i < n; ++i) {
      const uint16_t v = G[(size_t)j * ld + i];
      st[(size_t)j * ldg + i] = v;
      const bool miss = v == 0xFFFFu;
      g0[i] = miss ? 0.0 : (double)v / (double)scale;
      ms[i] = miss ? 1.0 : 0.0;
      if (!miss) { sg += v; sg2 += (double)v * v; no += 1; nz += v != 0; }
    }
    for (int64_t i = n; i < ldg; ++i) st[(size_t)j * ldg + i] = 0;
    if (out->vstat) { double* vs = out->vstat + (size_t)j * 4; vs[0] = sg; vs[1] = sg2; vs[2] = no; vs[3] = nz; }
    for (int c = 0; c < h.ncol; ++c) {
      const double* col = h.cols.data() + (size_t)c * n;
      double s = 0.0, sm = 0.0, s2 = 0.0;
      for (int64_t i = 0; i < n; ++i) { s += g0[i] * col[i]; sm += ms[i] * col[i]; s2 += g0[i] * g0[i] * col[i]; }
      if (out->sums) { out->sums[((size_t)j * 2) * h.ncol + c] = s; out->sums[((size_t)j * 2 + 1) * h.ncol + c] = sm; }
      if (out->sq && c < h.nsq) out->sq[(size_t)j * h.nsq + c] = s2;
    }
  }
  return RG_S2_OK;
}

}  // extern "C"
