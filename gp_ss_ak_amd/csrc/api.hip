// api.hip -- the C-ABI of libgpak_hip.so (see include/gpak.h for the contract and the
// reference members each entry point replaces).  Orchestration only; kernels live in
// gram.hip / gemm.hip / potrf.hip / solve.hip / predict.hip.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>

#include "../../include/gpak_dist.h"
#include "gpak_internal.h"

int gpak_calibrate_impl(gpak_ctx *ctx, double *scratch, size_t scratch_bytes, double *tflops, double *gbs);
int gpak_predict_impl(gpak_ctx *ctx, const double *Xte, long M, double *mean, double *var, const double *pool_sum,
                      long pool_M);
// multi.hip
#include <functional>
int gpak_multi_create(gpak_multi **out, int n, const int *devices, int precision, std::string &err,
                      const gpak_dist_engine *const *engines);
void gpak_multi_destroy(gpak_multi *g);
const char *gpak_multi_error(gpak_multi *g);
int gpak_multi_set_train(gpak_multi *g, const double *X, const double *y, int N, int d);
int gpak_multi_set_params(gpak_multi *g, const double *expans, double bias, double sn2, int dist_mode);
int gpak_multi_nlz(gpak_multi *g, double *nlz, double *quad, double *sumlp, double *logdet);
int gpak_multi_alpha(gpak_multi *g, double *alpha_host);
int gpak_multi_on_replica0(gpak_multi *g, const std::function<int(gpak_ctx *)> &f, bool wants_factor);
int gpak_multi_failed_column(gpak_multi *g);
int gpak_multi_set_kernel(gpak_multi *g, int nterms, const int *kinds, const double *pars, double bias, double white,
                          double sn2, int dist_mode);
int gpak_multi_grad_hyb(gpak_multi *g, double *grad, int ng);
const char *gpak_multi_transport(gpak_multi *g);
int gpak_multi_stats(gpak_multi *g, int r, gpak_dist_stats *out);
int gpak_multi_predict(gpak_multi *g, const double *Xte, long M, int d, double *mean, double *var);
int gpak_multi_grad(gpak_multi *g, double *grad10);
int gpak_multi_timing(gpak_multi *g, gpak_phase_times *out);
int gpak_multi_n(gpak_multi *g);
#define GPAK_MULTI_ERR(rc_) do { int v_ = (rc_); if (v_) ctx->err = gpak_multi_error(ctx->multi); return v_; } while (0)
int gpak_solve_chol_impl(gpak_ctx *ctx, double *X_host, int k);
int gpak_grad_impl(gpak_ctx *ctx, double *g, int ng);

static std::string g_global_err;

// ---- the tuning set (gpak_internal.h): defaults, overridden once by the GPAK_* environment ------------------
static GpakTuning read_tuning_env() {
  GpakTuning t;
  auto geti = [](const char *name, int &v) { if (const char *e = getenv(name)) v = atoi(e); };
  auto getb = [](const char *name, bool &v) { if (const char *e = getenv(name)) v = atoi(e) != 0; };
  geti("GPAK_NB_OUTER", t.nb_outer);
  geti("GPAK_NB_WIDE", t.nb_wide); t.nb_wide = t.nb_wide / GPAK_TILE * GPAK_TILE;
  geti("GPAK_NB_WIDE_ROWS", t.nb_wide_rows);
  geti("GPAK_NB_XWIDE", t.nb_xwide); t.nb_xwide = t.nb_xwide / GPAK_TILE * GPAK_TILE;
  geti("GPAK_NB_XWIDE_ROWS", t.nb_xwide_rows);
  getb("GPAK_FIRST_NARROW", t.first_narrow);
  geti("GPAK_TAIL_ROWS", t.tail_rows);
  getb("GPAK_SUB_NEXT", t.sub_next);
  geti("GPAK_NEXT_SPLIT_ROWS", t.next_split_rows);
  getb("GPAK_INV512", t.inv512);
  geti("GPAK_BWD_FUSED", t.bwd_fused);
  geti("GPAK_SBASE_ROWS", t.sbase_rows);
  geti("GPAK_BWD_BLOCK", t.bwd_block);
  getb("GPAK_LOOKAHEAD", t.lookahead);
  getb("GPAK_FWD_IN_FACTOR", t.fwd_in_factor);
  geti("GPAK_POTRF_CO", t.potrf_co);
  geti("GPAK_TAIL_MASK", t.tail_mask);
  geti("GPAK_TAIL_MASK_STRIDE", t.tail_mask_stride);
  getb("GPAK_BULK_QUEUE", t.bulk_queue);
  if (const char *e = getenv("GPAK_LD_PAD")) t.ld_pad = atol(e) / 2 * 2;
  geti("GPAK_GEMM_SMALL", t.gemm_small);
  geti("GPAK_GEMM_SMALL_ROWS", t.gemm_small_rows);
  geti("GPAK_SUPER_LR", t.super_lr);
  getb("GPAK_FILL_FAST", t.fill_fast);
  getb("GPAK_KMV_SYM", t.kmv_sym);
  if (const char *e = getenv("GPAK_F32_ACC")) t.f32_wide = strcmp(e, "plain") != 0;
  geti("GPAK_F32_RSD", t.f32_rsd);
  geti("GPAK_F32_TILE", t.f32_tile);
  geti("GPAK_PRED_BATCH", t.pred_batch);
  geti("GPAK_PRED_LD_SKEW", t.pred_ld_skew);
  if (const char *e = getenv("GPAK_FS_LEVELS_F32")) {   // "128,512,...": ascending, each a multiple of the one before
    int n = 0, lv[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (const char *p = e; *p && n < 8;) {
      const int v = atoi(p);
      if (v >= GPAK_TILE && v % GPAK_TILE == 0 && (n == 0 ? v == GPAK_TILE : v > lv[n - 1] && v % lv[n - 1] == 0)) lv[n++] = v;
      while (*p && *p != ',') p++;
      if (*p == ',') p++;
    }
    if (n > 0) memcpy(t.fs_levels, lv, sizeof(lv));
  }
  return t;
}
static GpakTuning &tuning_storage() {
  static GpakTuning t = read_tuning_env();
  return t;
}
const GpakTuning &gpak_tuning() { return tuning_storage(); }
extern "C" void gpak_reload_tuning(void) { tuning_storage() = read_tuning_env(); }

static int round_up(int v, int m) { return (v + m - 1) / m * m; }

static void free_points(DevPoints &p) {
  if (p.base) hipFree(p.base);
  p = DevPoints();
}
int gpak_alloc_points(gpak_ctx *ctx, DevPoints &p, int cap) {
  if (p.cap >= cap) return GPAK_OK;
  free_points(p);
  double *base = nullptr;
  if (hipMalloc(&base, sizeof(double) * GPAK_PT * GPAK_MAX_TERMS * (size_t)cap) != hipSuccess) {
    ctx->err = "hipMalloc(points) failed";
    return GPAK_ENOMEM;
  }
  p.base = base;
  p.cap = cap;
  return GPAK_OK;
}

// sigInv = Rot * lambda * Rot^T, Kernel.cpp:1399-1425 (host, 3x3)
void gpak_build_siginv(const double *e, double *A) {
  const double alpha = e[0], beta = e[2], teta = e[4];
  const double lam[3] = {e[1], e[3], e[5]};
  double R[9];
  const double ca = std::cos(alpha), sa = std::sin(alpha), cb = std::cos(beta), sb = std::sin(beta);
  const double ct = std::cos(teta), st = std::sin(teta);
  R[0 + 0 * 3] = ca * ct + sa * sb * st;
  R[0 + 1 * 3] = -sa * ct + ca * sb * st;
  R[0 + 2 * 3] = -cb * st;
  R[1 + 0 * 3] = sa * cb;
  R[1 + 1 * 3] = ca * cb;
  R[1 + 2 * 3] = sb;
  R[2 + 0 * 3] = ca * st - sa * sb * ct;
  R[2 + 1 * 3] = -sa * st - ca * sb * ct;
  R[2 + 2 * 3] = cb * ct;
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) {
      double s = 0.0;
      for (int k = 0; k < 3; k++) s += R[r + k * 3] * lam[k] * R[c + k * 3];
      A[r + c * 3] = s;
    }
}

// HybKerns composition -> KernParams (Kernel.cpp:140-154 and the children's parameter lists; shared by gpak_set_kernel
// and by the distributed paths, which carry the composition as one serialized array: gpak_dev.h GPAK_DIST_HYB)
int gpak_build_kp(int nterms, const int *kinds, const double *pars, double bias, double white, int dist_mode,
                  KernParams *out, double *kdiag_out) {
  KernParams kp;
  memset(&kp, 0, sizeof(kp));
  if (nterms < 1 || nterms > GPAK_MAX_TERMS) return GPAK_EINVAL;
  kp.nterms = nterms;
  double kdiag = bias + white;
  const double *p = pars;
  for (int t = 0; t < nterms; t++) {
    KernTerm &T = kp.term[t];
    if (kinds[t] == GPAK_KERN_EXPANS) {
      gpak_build_siginv(p, T.A);
      T.a33 = p[7];
      T.var2 = p[6] * p[6]; T.profile = GPAK_PROFILE_EXPSQRT;
      p += 8;
    } else if (kinds[t] == GPAK_KERN_EXP) {   // {Hayper_Euc_Exp, Sigma_Exp}, Kernel.cpp:576-600
      const double s = 1.0 / p[0];             // mlA: X * hyp^-2 on one side == both sides scaled by 1/hyp
      T.A[0] = T.A[4] = T.A[8] = T.a33 = s;  // EuclDist treats every input column alike (Kernel.cpp:1356-1362)
      T.var2 = p[1] * p[1]; T.profile = GPAK_PROFILE_EXPSQRT;
      p += 2;
    } else if (kinds[t] == GPAK_KERN_RBF) {   // {Hayper_Euc_RBF, inverseWidth_RBF, Sigma_RBF}, Kernel.cpp:411-428
      const double s = 1.0 / p[0];
      T.A[0] = T.A[4] = T.A[8] = T.a33 = s;
      T.iw = p[1]; T.var2 = p[2] * p[2]; T.profile = GPAK_PROFILE_RBF;
      p += 3;
    } else {
      return GPAK_EINVAL;
    }
    kdiag += T.var2;
  }
  kp.bias = bias; kp.white = white; kp.mode = dist_mode;
  *out = kp;
  if (kdiag_out) *kdiag_out = kdiag;
  return GPAK_OK;
}

// pooled mean of X1 u X2 exactly as Kernel.cpp:1391-1392 computes it
void gpak_pooled_mean(const double *s1, long n, const double *s2, long m, double *mu) {
  for (int k = 0; k < 4; k++) {  // s1, s2, mu hold four column sums (the 4th is 0 for 3-D inputs)
    double mX1 = (double)n / (double)(n + m) * s1[k] / (double)n;
    mu[k] = (double)m / (double)(n + m) * s2[k] / (double)m + mX1;
  }
}

static void release_train(gpak_ctx *ctx) {
  if (ctx->dX) hipFree(ctx->dX);
  if (ctx->dy) hipFree(ctx->dy);
  if (ctx->dM) hipFree(ctx->dM);
  if (ctx->dInv) hipFree(ctx->dInv);
  if (ctx->dInv512) hipFree(ctx->dInv512);
  if (ctx->dT512) hipFree(ctx->dT512);
  if (ctx->dAlpha) hipFree(ctx->dAlpha);
  if (ctx->dWork) hipFree(ctx->dWork);
  if (ctx->dF) hipFree(ctx->dF);
  gpak_grad_release(ctx);
  ctx->dX = ctx->dy = ctx->dM = ctx->dInv = ctx->dInv512 = ctx->dT512 = ctx->dAlpha = ctx->dWork = ctx->dF = nullptr;
  free_points(ctx->U);
  if (ctx->dLf) hipFree(ctx->dLf);
  if (ctx->dInvf) hipFree(ctx->dInvf);
  ctx->dLf = ctx->dInvf = nullptr;
  ctx->lf_ok = false;
  ctx->N = ctx->Np = 0;
  ctx->mstate = gpak_ctx::M_NONE;
  ctx->alpha_ok = ctx->nlz_ok = false;
}

extern "C" {

const char *gpak_global_error(void) { return g_global_err.c_str(); }

int gpak_create(gpak_ctx **out, int device, int precision) {
  if (!out) return GPAK_EINVAL;
  *out = nullptr;
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0) {
    g_global_err = std::string("no HIP device available: ") + (e != hipSuccess ? hipGetErrorString(e) : "count=0") +
                   " (libgpak_hip has no CPU fallback)";
    return GPAK_EHIP;
  }
  if (device < 0 || device >= count) { g_global_err = "device ordinal out of range"; return GPAK_EINVAL; }
  if (precision != GPAK_F64 && precision != GPAK_F32) { g_global_err = "precision must be GPAK_F64 or GPAK_F32"; return GPAK_EINVAL; }
  if ((e = hipSetDevice(device)) != hipSuccess) { g_global_err = hipGetErrorString(e); return GPAK_EHIP; }
  hipDeviceProp_t prop;
  if ((e = hipGetDeviceProperties(&prop, device)) != hipSuccess) { g_global_err = hipGetErrorString(e); return GPAK_EHIP; }
  if (std::string(prop.gcnArchName).find("gfx950") == std::string::npos) {
    g_global_err = std::string("device is ") + prop.gcnArchName + ", libgpak_hip is built for gfx950 only";
    return GPAK_EHIP;
  }
  gpak_ctx *ctx = new gpak_ctx();
  ctx->tune = gpak_tuning();
  ctx->device = device;
  ctx->precision = precision;
  memset(&ctx->times, 0, sizeof(ctx->times));
  int prio_lo = 0, prio_hi = 0;
  hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);  // numerically lower = higher priority
  if ((e = hipStreamCreateWithPriority(&ctx->stream, hipStreamNonBlocking, prio_lo)) != hipSuccess ||
      (e = hipStreamCreateWithPriority(&ctx->stream_hi, hipStreamNonBlocking, prio_hi)) != hipSuccess ||
      (e = hipStreamCreateWithPriority(&ctx->stream_fs, hipStreamNonBlocking, prio_hi)) != hipSuccess ||
      (e = hipStreamCreateWithPriority(&ctx->stream_x, hipStreamNonBlocking, prio_hi)) != hipSuccess) {
    g_global_err = hipGetErrorString(e); delete ctx; return GPAK_EHIP;
  }
  // a copy of the main stream that may not use the first GPAK_TAIL_MASK (default 8) compute units; the bulk
  // updates of the chain-bound tail of the factorisation go there so that the panel chain finds idle CUs
  {
    const int skip = ctx->tune.tail_mask;            // 0 switches it off; measured: 8 CUs, rows <= 12288: 183.4 -> 181.2 ms
    if (skip > 0 && skip < prop.multiProcessorCount) {
      std::vector<uint32_t> mask((prop.multiProcessorCount + 31) / 32, 0xffffffffu);
      const int stride = ctx->tune.tail_mask_stride;
      for (int c = 0; c < skip; c++) { const int bit = (c * stride) % prop.multiProcessorCount; mask[bit / 32] &= ~(1u << (bit % 32)); }
      if (hipExtStreamCreateWithCUMask(&ctx->stream_tail, (uint32_t)mask.size(), mask.data()) != hipSuccess) {
        ctx->stream_tail = nullptr;
        (void)hipGetLastError();
      }
    }
  }
  // The bulk trailing updates get a queue of their own, created through the same call as the tail's but with every CU
  // enabled.  Measured, not understood: on the context's ordinary low-priority stream the dependants of "bulk update b
  // done" (update b+1 in the same stream, the next column's update in the panel stream) start 75-125 us after it in
  // the middle of the factorisation with nothing else running, on a queue created by hipExtStreamCreateWithCUMask
  // 23 us: 175.1 -> 172.8 ms per factorisation at N = 32768 (same-box A/B).  The same for the panel stream (which then
  // loses its priority): 203 ms; for the forward-substitution or the main stream: +1.3 ms.  GPAK_BULK_QUEUE=0: off.
  if (ctx->tune.bulk_queue) {
    std::vector<uint32_t> mask((prop.multiProcessorCount + 31) / 32, 0xffffffffu);
    if (hipExtStreamCreateWithCUMask(&ctx->stream_bulk, (uint32_t)mask.size(), mask.data()) != hipSuccess) {
      ctx->stream_bulk = nullptr;
      (void)hipGetLastError();
    }
  }
  for (int i = 0; i < 10; i++) hipEventCreate(&ctx->ev[i]);
  hipMalloc(&ctx->dRed, sizeof(double) * 64);
  hipMalloc(&ctx->dInfo, sizeof(int) * 4);
  ctx->fwd_in_factor = ctx->tune.fwd_in_factor;
  ctx->lookahead = ctx->tune.lookahead;   // diagnostics: 0 = one stream
  ctx->nb_outer = ctx->tune.nb_outer;
  *out = ctx;
  return GPAK_OK;
}

int gpak_create_multi(gpak_ctx **out, int n_gpus, const int *devices, int precision) {
  if (!out || n_gpus < 1) return GPAK_EINVAL;
  *out = nullptr;
  if (precision != GPAK_F64 && precision != GPAK_F32) { g_global_err = "precision must be GPAK_F64 or GPAK_F32"; return GPAK_EINVAL; }
  gpak_multi *g = nullptr;
  int rc = gpak_multi_create(&g, n_gpus, devices, precision, g_global_err, nullptr);
  if (rc) return rc;
  gpak_ctx *ctx = new gpak_ctx();
  ctx->multi = g;
  ctx->precision = precision;
  memset(&ctx->times, 0, sizeof(ctx->times));
  *out = ctx;
  return GPAK_OK;
}

// TEST entry point (gpak_dist.h): the same group -- one host thread per rank, the in-process transport, the gpak_ctx
// surface of logLikelihood / alpha / gradient -- over caller-supplied engines, so that the thread-per-GPU host logic
// can be exercised (and run under ThreadSanitizer / AddressSanitizer) on a box without a GPU.
int gpak_create_multi_with_engines(gpak_ctx **out, int n_ranks, const gpak_dist_engine *const *engines) {
  if (!out || n_ranks < 1 || !engines) return GPAK_EINVAL;
  *out = nullptr;
  for (int r = 0; r < n_ranks; r++) if (!engines[r]) return GPAK_EINVAL;
  gpak_multi *g = nullptr;
  int rc = gpak_multi_create(&g, n_ranks, nullptr, GPAK_F64, g_global_err, engines);
  if (rc) return rc;
  gpak_ctx *ctx = new gpak_ctx();
  ctx->multi = g;
  ctx->precision = GPAK_F64;
  memset(&ctx->times, 0, sizeof(ctx->times));
  *out = ctx;
  return GPAK_OK;
}

void gpak_destroy(gpak_ctx *ctx) {
  if (!ctx) return;
  if (ctx->multi) { gpak_multi_destroy(ctx->multi); delete ctx; return; }
  hipSetDevice(ctx->device);
  hipStreamSynchronize(ctx->stream);
  release_train(ctx);
  gpak_predict_release(ctx);
  if (ctx->dRed) hipFree(ctx->dRed);
  if (ctx->dInfo) hipFree(ctx->dInfo);
  for (int i = 0; i < 10; i++) hipEventDestroy(ctx->ev[i]);
  for (auto e : ctx->ev_pool) hipEventDestroy(e);
  for (auto e : ctx->ev_sync) hipEventDestroy(e);
  hipStreamDestroy(ctx->stream_hi);
  hipStreamDestroy(ctx->stream_fs);
  hipStreamDestroy(ctx->stream_x);
  if (ctx->stream_tail) hipStreamDestroy(ctx->stream_tail);
  if (ctx->stream_bulk) hipStreamDestroy(ctx->stream_bulk);
  hipStreamDestroy(ctx->stream);
  delete ctx;
}

const char *gpak_last_error(const gpak_ctx *ctx) { return ctx ? ctx->err.c_str() : g_global_err.c_str(); }

int gpak_set_option(gpak_ctx *ctx, int option, long value) {
  if (!ctx) return GPAK_EINVAL;
  if (ctx->multi) return GPAK_OK;   // schedule options of the single-GPU factorisation: nothing to set on a group
  switch (option) {
    case GPAK_OPT_MEMOISE: ctx->memoise = value != 0; return GPAK_OK;
    case GPAK_OPT_NB_OUTER:
      if (value < 128 || value % 128) { ctx->err = "nb_outer must be a positive multiple of 128"; return GPAK_EINVAL; }
      ctx->nb_outer = (int)value;
      return GPAK_OK;
    case GPAK_OPT_PROFILE: ctx->profile = value != 0; return GPAK_OK;
    case GPAK_OPT_LOOKAHEAD: ctx->lookahead = value != 0; return GPAK_OK;
    case GPAK_OPT_NB_WIDE:
      if (value < 0 || value % 128) { ctx->err = "nb_wide must be 0 or a multiple of 128"; return GPAK_EINVAL; }
      ctx->tune.nb_wide = (int)value;
      return GPAK_OK;
    case GPAK_OPT_NB_WIDE_ROWS: ctx->tune.nb_wide_rows = (int)value; return GPAK_OK;
    case GPAK_OPT_TAIL_ROWS: ctx->tune.tail_rows = (int)value; return GPAK_OK;
    case GPAK_OPT_FIRST_NARROW: ctx->tune.first_narrow = value != 0; return GPAK_OK;
    case GPAK_OPT_INV512: ctx->tune.inv512 = value != 0; return GPAK_OK;
    case GPAK_OPT_POTRF_CO:
      if (value < 0 || value > 2) { ctx->err = "potrf_co must be 0, 1 or 2"; return GPAK_EINVAL; }
      ctx->tune.potrf_co = (int)value;
      return GPAK_OK;
    case GPAK_OPT_PRED_BATCH: ctx->tune.pred_batch = (int)value; return GPAK_OK;
    case GPAK_OPT_BWD_FUSED:
      if (value < 0 || value > 2) { ctx->err = "bwd_fused must be 0, 1 or 2"; return GPAK_EINVAL; }
      ctx->tune.bwd_fused = (int)value;
      return GPAK_OK;
  }
  ctx->err = "unknown option";
  return GPAK_EINVAL;
}

int gpak_set_train(gpak_ctx *ctx, const double *X, const double *y, int N, int d) {
  if (!ctx || !X || !y || N <= 0) return GPAK_EINVAL;
  if (ctx->multi) { ctx->N = N; ctx->d = d; GPAK_MULTI_ERR(gpak_multi_set_train(ctx->multi, X, y, N, d)); }
  // d = 3: x, y, z; d = 4: + rock-type column with its own inverse width (Kernel.cpp:1411-1424)
  if (d != 3 && d != 4) { ctx->err = "inputs must have 3 or 4 columns"; return GPAK_ENOTIMPL; }
  GPAK_HIP(hipSetDevice(ctx->device));
  release_train(ctx);
  const int Np = round_up(N, GPAK_TILE);
  long pad = Np >= 1024 ? 32 : 0;
  if (ctx->tune.ld_pad >= 0) pad = ctx->tune.ld_pad;
  const long ld = Np + pad;
  ctx->N = N; ctx->Np = Np; ctx->ld = (int)ld; ctx->d = d;
  ctx->hX.assign(X, X + (size_t)N * d);
  ctx->xsum[3] = 0.0;
  for (int k = 0; k < d; k++) {
    double s = 0.0;
    for (int i = 0; i < N; i++) s += X[i + (size_t)k * N];
    ctx->xsum[k] = s;
  }
  const int T = Np / GPAK_TILE;
  // width of the back substitution's explicit diagonal-block inverses (fixed per training set: it sizes dInv512)
  int bw = ctx->tune.bwd_block;
  if (bw != 512 && bw != 1024 && bw != 2048) bw = 512;
  ctx->bwd_bw = bw;
  if (hipMalloc(&ctx->dX, sizeof(double) * 4 * (size_t)Np) != hipSuccess ||
      hipMalloc(&ctx->dy, sizeof(double) * (size_t)Np) != hipSuccess ||
      hipMalloc(&ctx->dM, sizeof(double) * (size_t)ld * Np) != hipSuccess ||
      hipMalloc(&ctx->dInv, sizeof(double) * (size_t)T * 2 * GPAK_TILE * GPAK_TILE) != hipSuccess ||
      hipMalloc(&ctx->dInv512, sizeof(double) * (size_t)((Np + bw - 1) / bw) * bw * bw) != hipSuccess ||
      hipMalloc(&ctx->dT512, sizeof(double) * (size_t)(2 * ((Np + bw - 1) / bw)) * bw * bw) != hipSuccess ||
      hipMalloc(&ctx->dAlpha, sizeof(double) * (size_t)Np) != hipSuccess ||
      hipMalloc(&ctx->dF, sizeof(double) * (size_t)Np) != hipSuccess ||
      // 70 Np doubles of vectors + the back substitution's scratch for the widest block ((8 + bw / 32) * bw for the
      // three-launch step, 34 * bw for the fused one: solve.hip)
      hipMalloc(&ctx->dWork, sizeof(double) * (70 * (size_t)Np + (size_t)std::max(8 + bw / 32, 34) * bw)) != hipSuccess) {
    ctx->err = "device allocation failed for the training set";
    release_train(ctx);
    return GPAK_ENOMEM;
  }
  int rc = gpak_alloc_points(ctx, ctx->U, Np);
  if (rc) return rc;
  GPAK_HIP(hipMemsetAsync(ctx->dX, 0, sizeof(double) * 4 * (size_t)Np, ctx->stream));
  GPAK_HIP(hipMemsetAsync(ctx->dy, 0, sizeof(double) * (size_t)Np, ctx->stream));
  // the inverse blocks are triangular: zero once, potrf128 only writes inside the triangles
  GPAK_HIP(hipMemsetAsync(ctx->dInv, 0, sizeof(double) * (size_t)T * 2 * GPAK_TILE * GPAK_TILE, ctx->stream));
  GPAK_HIP(hipMemsetAsync(ctx->dAlpha, 0, sizeof(double) * (size_t)Np, ctx->stream));
  for (int k = 0; k < d; k++)
    GPAK_HIP(hipMemcpyAsync(ctx->dX + (size_t)k * Np, X + (size_t)k * N, sizeof(double) * N,
                            hipMemcpyHostToDevice, ctx->stream));
  GPAK_HIP(hipMemcpyAsync(ctx->dy, y, sizeof(double) * N, hipMemcpyHostToDevice, ctx->stream));
  GPAK_HIP(hipStreamSynchronize(ctx->stream));
  memset(&ctx->times, 0, sizeof(ctx->times));
  ctx->times.n = N; ctx->times.n_padded = Np;
  ctx->U.n = 0;  // transformed points are rebuilt on the next use
  return GPAK_OK;
}

int gpak_set_params(gpak_ctx *ctx, const double *expans, double bias, double sn2, int dist_mode) {
  if (!ctx || !expans) return GPAK_EINVAL;
  if (dist_mode != GPAK_DIST_EXPANSION && dist_mode != GPAK_DIST_DIRECT) { ctx->err = "bad dist_mode"; return GPAK_EINVAL; }
  if (ctx->multi) { ctx->sn2 = sn2; GPAK_MULTI_ERR(gpak_multi_set_params(ctx->multi, expans, bias, sn2, dist_mode)); }
  bool same = ctx->have_params && memcmp(expans, ctx->expans, sizeof(double) * 8) == 0 && bias == ctx->bias &&
              sn2 == ctx->sn2 && dist_mode == ctx->dist_mode;
  memcpy(ctx->expans, expans, sizeof(double) * 8);
  ctx->bias = bias; ctx->sn2 = sn2; ctx->dist_mode = dist_mode;
  ctx->have_params = true;
  ctx->expans_only = true;
  ctx->kinds[0] = GPAK_KERN_EXPANS;
  ctx->kp.nterms = 1;
  gpak_build_siginv(expans, ctx->kp.term[0].A);
  ctx->kp.term[0].a33 = expans[7];  // InversewidthR: lambda(3,3) for a 4th input column (Kernel.cpp:1421-1424)
  ctx->kp.term[0].var2 = expans[6] * expans[6];
  ctx->kp.term[0].iw = 0.0;
  ctx->kp.term[0].profile = GPAK_PROFILE_EXPSQRT;
  ctx->kp.bias = bias;
  ctx->kp.white = 0.0;
  ctx->kp.mode = dist_mode;
  ctx->kdiag = expans[6] * expans[6] + bias;
  if (!(same && ctx->memoise)) {
    // GP_Utils.cpp:132-133: setKUpdateStat(false) -> K, alpha and the likelihood are stale
    ctx->mstate = gpak_ctx::M_NONE;
    ctx->alpha_ok = ctx->nlz_ok = false;
    ctx->U.n = 0;
  }
  return GPAK_OK;
}

// general composition: kinds[t] in {GPAK_KERN_EXPANS, GPAK_KERN_EXP, GPAK_KERN_RBF}, pars = the children's
// parameters concatenated in the reference's order (8 / 2 / 3 values)
int gpak_set_kernel(gpak_ctx *ctx, int nterms, const int *kinds, const double *pars, double bias, double white,
                    double sn2, int dist_mode) {
  if (!ctx || !kinds || !pars || nterms < 1 || nterms > GPAK_MAX_TERMS) return GPAK_EINVAL;
  if (dist_mode != GPAK_DIST_EXPANSION && dist_mode != GPAK_DIST_DIRECT) { ctx->err = "bad dist_mode"; return GPAK_EINVAL; }
  if (ctx->multi) { ctx->sn2 = sn2; GPAK_MULTI_ERR(gpak_multi_set_kernel(ctx->multi, nterms, kinds, pars, bias, white, sn2, dist_mode)); }
  KernParams kp;
  double kdiag = 0.0;
  int rc = gpak_build_kp(nterms, kinds, pars, bias, white, dist_mode, &kp, &kdiag);
  if (rc) { ctx->err = "unknown kernel kind"; return rc; }
  {
    const double *p = pars;
    for (int t = 0; t < nterms; t++) {
      if (kinds[t] == GPAK_KERN_EXPANS) memcpy(ctx->expans, p, sizeof(double) * 8);  // the (single) ExpAns child, wherever it sits
      p += kinds[t] == GPAK_KERN_EXPANS ? 8 : kinds[t] == GPAK_KERN_EXP ? 2 : 3;
      ctx->kinds[t] = kinds[t];
    }
  }
  ctx->kp = kp;
  ctx->kdiag = kdiag;
  ctx->expans_only = (nterms == 1 && kinds[0] == GPAK_KERN_EXPANS && white == 0.0);
  ctx->bias = bias; ctx->sn2 = sn2; ctx->dist_mode = dist_mode;
  ctx->have_params = true;
  ctx->mstate = gpak_ctx::M_NONE;
  ctx->alpha_ok = ctx->nlz_ok = false;
  ctx->U.n = 0;
  return GPAK_OK;
}

}  // extern "C"

// transformed training points for the train x train Gram (pooled mean of X u X)
int gpak_ensure_U(gpak_ctx *ctx) {
  if (!ctx->N) { ctx->err = "no training set (gpak_set_train)"; return GPAK_ESTATE; }
  if (!ctx->have_params) { ctx->err = "no parameters (gpak_set_params)"; return GPAK_ESTATE; }
  if (ctx->U.n == ctx->N) return GPAK_OK;
  gpak_pooled_mean(ctx->xsum, ctx->N, ctx->xsum, ctx->N, ctx->kp.mu);
  ctx->kp.d = ctx->d;
  gpak_launch_transform(ctx->stream, ctx->dX, ctx->Np, ctx->N, ctx->kp, ctx->U);
  return GPAK_OK;
}

static int ensure_factor(gpak_ctx *ctx) {
  if (ctx->mstate == gpak_ctx::M_L) return GPAK_OK;
  int rc = gpak_ensure_U(ctx);
  if (rc) return rc;
  GPAK_HIP(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  GPAK_HIP(hipEventRecord(ctx->ev[0], st));
  // B = I + (sW sW') % K  with W = 1/sn2  (GP_Utils.cpp:898-902); lower tiles only
  gpak_launch_fill(st, ctx->U, ctx->U, ctx->Np, ctx->Np, ctx->kp, 1.0 / ctx->sn2, 1.0, 1.0, 1, ctx->dM,
                   ctx->ld, nullptr);
  ctx->mstate = gpak_ctx::M_B;
  ctx->z_ok = false;
  ctx->inv512_ok = false;
  ctx->t512_mode = 0;
  ctx->lf_ok = false;
  if (ctx->fwd_in_factor) gpak_launch_scale(st, ctx->Np, ctx->dy, 1.0 / ctx->sn2, ctx->dWork);  // rhs = y/sn2
  GPAK_HIP(hipEventRecord(ctx->ev[1], st));
  rc = gpak_potrf_blocked(ctx);
  GPAK_HIP(hipEventRecord(ctx->ev[2], st));
  GPAK_HIP(hipEventSynchronize(ctx->ev[2]));
  float ms = 0;
  GPAK_HIP(hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]));
  ctx->times.gram_ms = ms;
  GPAK_HIP(hipEventElapsedTime(&ms, ctx->ev[1], ctx->ev[2]));
  ctx->times.factor_ms = ms;
  ctx->times.accumulated_ms[0] += ctx->times.gram_ms;
  ctx->times.accumulated_ms[1] += ctx->times.factor_ms;
  ctx->times.evaluations++;
  ctx->times.gram_bytes = 8.0 * ((double)ctx->Np * (ctx->Np + GPAK_TILE) / 2.0);
  if (rc == GPAK_ENOTPD) {
    ctx->mstate = gpak_ctx::M_NONE;
    ctx->err = "B = I + K/sn2 is not positive definite";
    return rc;
  }
  if (rc) return rc;
  ctx->mstate = gpak_ctx::M_L;
  ctx->z_ok = ctx->fwd_in_factor;
  ctx->t512_mode = ctx->fwd_in_factor && ctx->tune.inv512 ? ctx->tune.bwd_fused : 0;
  ctx->inv512_ok = ctx->fwd_in_factor && ctx->tune.inv512 && ctx->t512_mode != 2;   // mode 2 builds [R ; T] elsewhere
  return GPAK_OK;
}

static int ensure_alpha(gpak_ctx *ctx) {
  int rc = ensure_factor(ctx);
  if (rc) return rc;
  if (ctx->alpha_ok) return GPAK_OK;
  hipStream_t st = ctx->stream;
  double *w0 = ctx->dWork, *w1 = ctx->dWork + ctx->Np;
  GPAK_HIP(hipEventRecord(ctx->ev[3], st));
  // alpha = (K + sn2 I)^-1 y = B^-1 (y / sn2)
  if (!ctx->z_ok) {
    gpak_launch_scale(st, ctx->Np, ctx->dy, 1.0 / ctx->sn2, w0);
    gpak_launch_trsv_fwd(st, ctx->Np, ctx->dM, ctx->ld, ctx->dInv, w0, w1);
  }
  ctx->z_ok = false;  // the back substitution consumes w1
  if (ctx->t512_mode)
    gpak_launch_trsv_bwd3(st, ctx->Np, ctx->dM, ctx->ld, w1, ctx->dAlpha, ctx->dWork + 2 * (size_t)ctx->Np,
                          ctx->t512_mode == 2 ? ctx->dT512 : ctx->dInv512,
                          (size_t)(ctx->t512_mode == 2 ? 2 : 1) * ctx->bwd_bw * ctx->bwd_bw,
                          (ctx->t512_mode == 2 ? 2 : 1) * ctx->bwd_bw, ctx->t512_mode == 2, ctx->bwd_bw);
  else
    gpak_launch_trsv_bwd2(st, ctx->Np, ctx->dM, ctx->ld, ctx->dInv, w1, ctx->dAlpha, ctx->dWork + 2 * (size_t)ctx->Np,
                          ctx->inv512_ok ? ctx->dInv512 : nullptr, ctx->bwd_bw);
  GPAK_HIP(hipEventRecord(ctx->ev[4], st));
  GPAK_HIP(hipEventSynchronize(ctx->ev[4]));
  float ms = 0;
  GPAK_HIP(hipEventElapsedTime(&ms, ctx->ev[3], ctx->ev[4]));
  ctx->times.solve_ms = ms;
  ctx->times.accumulated_ms[2] += ms;
  ctx->alpha_ok = true;
  return GPAK_OK;
}

static int ensure_nlz(gpak_ctx *ctx) {
  int rc = ensure_alpha(ctx);
  if (rc) return rc;
  if (ctx->nlz_ok) return GPAK_OK;
  hipStream_t st = ctx->stream;
  double *f = ctx->dF;
  double *scratch = ctx->dWork + 4 * (size_t)ctx->Np;  // 64 * Np doubles available
  GPAK_HIP(hipEventRecord(ctx->ev[5], st));
  int splits = gpak_kmatvec_splits(ctx->N, ctx->N);
  gpak_launch_kmatvec(st, ctx->U, 0, ctx->N, ctx->dAlpha, ctx->U, ctx->kp, scratch, splits, f, 64);  // f = K*Alpha
  GPAK_HIP(hipEventRecord(ctx->ev[8], st));
  if (ctx->kp.white != 0.0) gpak_launch_axpy(st, ctx->N, ctx->kp.white, ctx->dAlpha, f);  // Kern_White diagonal
  gpak_launch_logdet(st, ctx->N, ctx->dM, ctx->ld, ctx->dRed);
  gpak_launch_nlz_terms(st, ctx->N, ctx->dy, f, ctx->dAlpha, ctx->sn2, ctx->dRed);
  double red[3];
  GPAK_HIP(hipMemcpyAsync(red, ctx->dRed, sizeof(red), hipMemcpyDeviceToHost, st));
  GPAK_HIP(hipEventRecord(ctx->ev[6], st));
  GPAK_HIP(hipEventSynchronize(ctx->ev[6]));
  float ms = 0;
  GPAK_HIP(hipEventElapsedTime(&ms, ctx->ev[5], ctx->ev[6]));
  ctx->times.nlz_ms = ms;
  ctx->times.accumulated_ms[3] += ms;
  GPAK_HIP(hipEventElapsedTime(&ms, ctx->ev[5], ctx->ev[8]));
  ctx->times.kmatvec_ms = ms;
  ctx->logdet = red[0]; ctx->quad = red[1]; ctx->sumlp = red[2];
  ctx->nlz = ctx->quad - ctx->sumlp + ctx->logdet;  // GP_Utils.cpp:1159
  ctx->nlz_ok = true;
  return GPAK_OK;
}

// multi.hip: a context of a group takes over the factor its device's rank already holds as packed panels (the
// result of the group's distributed logLikelihood() for the CURRENT parameters) instead of factoring again:
// device-to-device copies of N^2/2 doubles where ensure_factor would spend N^3/3 flops.  The context must hold the
// same training set and parameters (multi.hip's ensure_replica); afterwards it is in the state ensure_nlz leaves.
int gpak_import_factor(gpak_ctx *ctx, const gpak_dist_factor_view *v) {
  if (!ctx || !v) return GPAK_EINVAL;
  if (!ctx->N || v->N != ctx->N || v->Np != ctx->Np || !ctx->have_params) {
    ctx->err = "gpak_import_factor: the context does not hold the group's training set / parameters";
    return GPAK_ESTATE;
  }
  GPAK_HIP(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const int Np = ctx->Np;
  const long ld = ctx->ld;
  for (int b = 0; b < v->nJ; b++) {
    const int J = b * v->nb, W = std::min(v->nb, Np - J), rows = Np - J;
    GPAK_HIP(hipMemcpy2DAsync(ctx->dM + J + (size_t)J * ld, sizeof(double) * ld, v->panels[b], sizeof(double) * rows,
                              sizeof(double) * rows, W, hipMemcpyDeviceToDevice, st));
    GPAK_HIP(hipMemcpyAsync(ctx->dInv + (size_t)(J / GPAK_TILE) * 2 * GPAK_TILE * GPAK_TILE, v->invs[b],
                            sizeof(double) * (W / GPAK_TILE) * 2 * GPAK_TILE * GPAK_TILE, hipMemcpyDeviceToDevice, st));
  }
  GPAK_HIP(hipMemcpyAsync(ctx->dAlpha, v->alpha, sizeof(double) * Np, hipMemcpyDeviceToDevice, st));
  GPAK_HIP(hipMemcpyAsync(ctx->dF, v->f, sizeof(double) * Np, hipMemcpyDeviceToDevice, st));
  GPAK_HIP(hipStreamSynchronize(st));
  ctx->mstate = gpak_ctx::M_L;
  ctx->z_ok = false;         // dWork does not hold L^-1 (y/sn2)
  ctx->inv512_ok = false;    // the 512-block inverses were not imported: back substitutions use the 128-blocks
  ctx->t512_mode = 0;
  ctx->lf_ok = false;        // the fp32 image (GPAK_F32 prediction) is rebuilt from this factor on first use
  ctx->alpha_ok = true; ctx->nlz_ok = true;
  ctx->failed_col = 0;
  ctx->quad = v->quad; ctx->sumlp = v->sumlp; ctx->logdet = v->logdet; ctx->nlz = v->nlz;
  return GPAK_OK;
}

// copy an n x m block (ld) of a device matrix to a dense host column-major array
static int copy_out(gpak_ctx *ctx, const double *dsrc, long ld, int n, int m, double *host) {
  GPAK_HIP(hipMemcpy2DAsync(host, sizeof(double) * n, dsrc, sizeof(double) * ld, sizeof(double) * n, m,
                            hipMemcpyDeviceToHost, ctx->stream));
  GPAK_HIP(hipStreamSynchronize(ctx->stream));
  return GPAK_OK;
}

extern "C" {

int gpak_gram(gpak_ctx *ctx, double *K_host, double *D2_host) {
  if (!ctx) return GPAK_EINVAL;
  if (ctx->multi) GPAK_MULTI_ERR(gpak_multi_on_replica0(ctx->multi, [&](gpak_ctx *c) { return gpak_gram(c, K_host, D2_host); }, false));
  GPAK_HIP(hipSetDevice(ctx->device));
  int rc = gpak_ensure_U(ctx);
  if (rc) return rc;
  double *dD2 = nullptr;
  if (D2_host && hipMalloc(&dD2, sizeof(double) * (size_t)ctx->ld * ctx->Np) != hipSuccess) {
    ctx->err = "device allocation failed for D2";
    return GPAK_ENOMEM;
  }
  // the matrix buffer is reused: whatever factor it held is gone
  ctx->mstate = gpak_ctx::M_NONE;
  ctx->alpha_ok = ctx->nlz_ok = false;
  gpak_launch_fill(ctx->stream, ctx->U, ctx->U, ctx->Np, ctx->Np, ctx->kp, 1.0, 0.0, 0.0, 0, ctx->dM, ctx->ld,
                   dD2);
  rc = GPAK_OK;
  if (K_host) rc = copy_out(ctx, ctx->dM, ctx->ld, ctx->N, ctx->N, K_host);
  if (!rc && D2_host) rc = copy_out(ctx, dD2, ctx->ld, ctx->N, ctx->N, D2_host);
  GPAK_HIP(hipStreamSynchronize(ctx->stream));
  if (dD2) hipFree(dD2);
  return rc;
}

int gpak_compute_k(gpak_ctx *ctx, const double *X1, int n, const double *X2, int m, int d, double *K_host,
                   double *D2_host) {
  if (!ctx || !X1 || !X2 || n <= 0 || m <= 0) return GPAK_EINVAL;
  if (ctx->multi)
    GPAK_MULTI_ERR(gpak_multi_on_replica0(ctx->multi, [&](gpak_ctx *c) { return gpak_compute_k(c, X1, n, X2, m, d, K_host, D2_host); }, false));
  if (d != 3 && d != 4) { ctx->err = "inputs must have 3 or 4 columns"; return GPAK_ENOTIMPL; }
  if (!ctx->have_params) { ctx->err = "no parameters (gpak_set_params)"; return GPAK_ESTATE; }
  GPAK_HIP(hipSetDevice(ctx->device));
  const int np = round_up(n, GPAK_TILE), mp = round_up(m, GPAK_TILE);
  double s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
  for (int k = 0; k < d; k++) {
    for (int i = 0; i < n; i++) s1[k] += X1[i + (size_t)k * n];
    for (int i = 0; i < m; i++) s2[k] += X2[i + (size_t)k * m];
  }
  KernParams kp = ctx->kp;
  kp.d = d;
  gpak_pooled_mean(s1, n, s2, m, kp.mu);
  if (!(X1[0] == X2[0] && n == m)) kp.white = 0.0;  // Kern_White::computeK, Kernel.cpp:260-262
  DevPoints P, Q;
  double *dx = nullptr, *dK = nullptr, *dD2 = nullptr;
  int rc = gpak_alloc_points(ctx, P, np);
  if (!rc) rc = gpak_alloc_points(ctx, Q, mp);
  if (!rc && (hipMalloc(&dx, sizeof(double) * 4 * (size_t)(np + mp)) != hipSuccess ||
              hipMalloc(&dK, sizeof(double) * (size_t)np * mp) != hipSuccess ||
              (D2_host && hipMalloc(&dD2, sizeof(double) * (size_t)np * mp) != hipSuccess))) {
    ctx->err = "device allocation failed in gpak_compute_k";
    rc = GPAK_ENOMEM;
  }
  if (!rc) {
    hipStream_t st = ctx->stream;
    hipMemsetAsync(dx, 0, sizeof(double) * 4 * (size_t)(np + mp), st);
    double *dx2 = dx + 4 * (size_t)np;
    for (int k = 0; k < d; k++) {
      hipMemcpyAsync(dx + (size_t)k * np, X1 + (size_t)k * n, sizeof(double) * n, hipMemcpyHostToDevice, st);
      hipMemcpyAsync(dx2 + (size_t)k * mp, X2 + (size_t)k * m, sizeof(double) * m, hipMemcpyHostToDevice, st);
    }
    gpak_launch_transform(st, dx, np, n, kp, P);
    gpak_launch_transform(st, dx2, mp, m, kp, Q);
    gpak_launch_fill(st, P, Q, np, mp, kp, 1.0, 0.0, 0.0, 0, dK, np, dD2);
    if (K_host) rc = copy_out(ctx, dK, np, n, m, K_host);
    if (!rc && D2_host) rc = copy_out(ctx, dD2, np, n, m, D2_host);
    hipStreamSynchronize(st);
  }
  free_points(P); free_points(Q);
  if (dx) hipFree(dx);
  if (dK) hipFree(dK);
  if (dD2) hipFree(dD2);
  return rc;
}

int gpak_factor(gpak_ctx *ctx) {
  if (!ctx) return GPAK_EINVAL;
  if (ctx->multi) { double v; GPAK_MULTI_ERR(gpak_multi_nlz(ctx->multi, &v, nullptr, nullptr, nullptr)); }
  return ensure_factor(ctx);
}

int gpak_failed_column(const gpak_ctx *ctx) {
  if (!ctx) return 0;
  return ctx->multi ? gpak_multi_failed_column(ctx->multi) : ctx->failed_col;
}
int gpak_n_gpus(const gpak_ctx *ctx) { return !ctx ? 0 : (ctx->multi ? gpak_multi_n(ctx->multi) : 1); }
const char *gpak_transport(const gpak_ctx *ctx) { return !ctx ? "" : (ctx->multi ? gpak_multi_transport(ctx->multi) : "none"); }
int gpak_group_rank_stats(gpak_ctx *ctx, int rank, gpak_dist_stats *out) {
  if (!ctx || !out || !ctx->multi) return GPAK_EINVAL;
  return gpak_multi_stats(ctx->multi, rank, out);
}

int gpak_get_chol_upper(gpak_ctx *ctx, double *R_host) {
  if (!ctx || !R_host) return GPAK_EINVAL;
  if (ctx->multi) GPAK_MULTI_ERR(gpak_multi_on_replica0(ctx->multi, [&](gpak_ctx *c) { return gpak_get_chol_upper(c, R_host); }, true));
  int rc = ensure_factor(ctx);
  if (rc) return rc;
  const int N = ctx->N;
  std::vector<double> tmp((size_t)N * N);
  rc = copy_out(ctx, ctx->dM, ctx->ld, N, N, tmp.data());
  if (rc) return rc;
  for (int j = 0; j < N; j++)
    for (int i = 0; i < N; i++) R_host[i + (size_t)j * N] = i <= j ? tmp[j + (size_t)i * N] : 0.0;  // R = L^T
  return GPAK_OK;
}

int gpak_solve_alpha(gpak_ctx *ctx, double *alpha_host) {
  if (!ctx) return GPAK_EINVAL;
  if (ctx->multi) GPAK_MULTI_ERR(gpak_multi_alpha(ctx->multi, alpha_host));
  int rc = ensure_alpha(ctx);
  if (rc) return rc;
  if (alpha_host) {
    GPAK_HIP(hipMemcpyAsync(alpha_host, ctx->dAlpha, sizeof(double) * ctx->N, hipMemcpyDeviceToHost, ctx->stream));
    GPAK_HIP(hipStreamSynchronize(ctx->stream));
  }
  return GPAK_OK;
}

int gpak_solve_chol(gpak_ctx *ctx, double *X_host, int k) {
  if (!ctx || !X_host || k <= 0) return GPAK_EINVAL;
  if (ctx->multi) GPAK_MULTI_ERR(gpak_multi_on_replica0(ctx->multi, [&](gpak_ctx *c) { return gpak_solve_chol(c, X_host, k); }, true));
  int rc = ensure_factor(ctx);
  if (rc) return rc;
  return gpak_solve_chol_impl(ctx, X_host, k);
}

int gpak_nlz(gpak_ctx *ctx, double *nlz) {
  if (!ctx || !nlz) return GPAK_EINVAL;
  if (ctx->multi) GPAK_MULTI_ERR(gpak_multi_nlz(ctx->multi, nlz, nullptr, nullptr, nullptr));
  int rc = ensure_nlz(ctx);
  if (rc) {
    *nlz = std::numeric_limits<double>::quiet_NaN();  // GP_Utils.cpp:1146, 1157
    return rc;
  }
  *nlz = ctx->nlz;
  return GPAK_OK;
}

int gpak_nlz_terms(gpak_ctx *ctx, double *quad, double *sumlp, double *logdet) {
  if (!ctx) return GPAK_EINVAL;
  if (ctx->multi) { double v; GPAK_MULTI_ERR(gpak_multi_nlz(ctx->multi, &v, quad, sumlp, logdet)); }
  int rc = ensure_nlz(ctx);
  if (rc) return rc;
  if (quad) *quad = ctx->quad;
  if (sumlp) *sumlp = ctx->sumlp;
  if (logdet) *logdet = ctx->logdet;
  return GPAK_OK;
}

int gpak_predict(gpak_ctx *ctx, const double *Xte, long M, int d, double *mean, double *var, int compat_flags) {
  if (!ctx || !Xte || !mean || M <= 0) return GPAK_EINVAL;
  int rc;
  if (ctx->multi) {
    rc = gpak_multi_predict(ctx->multi, Xte, M, d, mean, var);
    if (rc) { ctx->err = gpak_multi_error(ctx->multi); return rc; }
  } else {
    if (d != ctx->d) { ctx->err = "test points must have as many columns as the training set"; return GPAK_EINVAL; }
    // _postVar calls logLikelihood() (GP_Utils.cpp:980); _postMean calls updateAlpha() (:961)
    rc = ensure_nlz(ctx);
    if (rc) return rc;
    rc = gpak_predict_impl(ctx, Xte, M, mean, var, nullptr, 0);
    if (rc) return rc;
  }
  if (var) {
    if (compat_flags & GPAK_COMPAT_VARCLAMP) {
      // GP_Utils.cpp:1002-1003: `uvec ind = varSigma < 0; varSigma.elem(ind) = 0` -- the 0/1
      // mask is an index list, so element 0 is zeroed when any entry is >= 0 and element 1
      // when any entry is < 0.
      bool any_neg = false, any_nonneg = false;
      for (long i = 0; i < M; i++) { if (var[i] < 0) any_neg = true; else any_nonneg = true; }
      if (any_nonneg) var[0] = 0.0;
      if (any_neg && M > 1) var[1] = 0.0;
    } else {
      for (long i = 0; i < M; i++) if (var[i] < 0) var[i] = 0.0;
    }
    if (!((compat_flags & GPAK_COMPAT_SN2SKIP) && ctx->sn2 == 1.0))  // GP_Utils.cpp:1036-1040
      for (long i = 0; i < M; i++) var[i] += ctx->sn2;
  }
  return GPAK_OK;
}

int gpak_grad(gpak_ctx *ctx, double *g) {
  if (!ctx || !g) return GPAK_EINVAL;
  if (ctx->multi) GPAK_MULTI_ERR(gpak_multi_grad(ctx->multi, g));
  if (!ctx->expans_only) { ctx->err = "gpak_grad handles the ExpAns(+Bias) composition only"; return GPAK_ENOTIMPL; }
  int rc = ensure_nlz(ctx);  // GradLL re-enters logLikelihood(): GP_Utils.cpp:1173-1174
  if (rc) return rc;
  return gpak_grad_impl(ctx, g, 10);
}

int gpak_grad_hyb(gpak_ctx *ctx, double *g, int ng) {
  if (!ctx || !g) return GPAK_EINVAL;
  if (ctx->multi) GPAK_MULTI_ERR(gpak_multi_grad_hyb(ctx->multi, g, ng));
  int rc = ensure_nlz(ctx);
  if (rc) return rc;
  return gpak_grad_impl(ctx, g, ng);
}

int gpak_timing(gpak_ctx *ctx, gpak_phase_times *out) {
  if (!ctx || !out) return GPAK_EINVAL;
  if (ctx->multi) return gpak_multi_timing(ctx->multi, out);
  *out = ctx->times;
  return GPAK_OK;
}

int gpak_calibrate(gpak_ctx *ctx, double *mfma_f64_tflops, double *hbm_write_gbs) {
  if (!ctx || !mfma_f64_tflops || !hbm_write_gbs) return GPAK_EINVAL;
  if (ctx->multi)
    GPAK_MULTI_ERR(gpak_multi_on_replica0(ctx->multi, [&](gpak_ctx *c) { return gpak_calibrate(c, mfma_f64_tflops, hbm_write_gbs); }, false));
  GPAK_HIP(hipSetDevice(ctx->device));
  const size_t bytes = (size_t)4 << 30;
  double *scratch = nullptr;
  if (hipMalloc(&scratch, bytes) != hipSuccess) { ctx->err = "calibration scratch allocation failed"; return GPAK_ENOMEM; }
  int rc = gpak_calibrate_impl(ctx, scratch, bytes, mfma_f64_tflops, hbm_write_gbs);
  hipFree(scratch);
  return rc;
}

}  // extern "C"
