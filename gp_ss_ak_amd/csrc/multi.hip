// multi.hip -- ONE process driving all GPUs of a node: gpak_create_multi (include/gpak.h), what
// `gp_ss_ak --gpus n` runs on.  One host thread per device; each thread owns
//   * one rank of the C++ block-column-cyclic schedule (csrc/dist.hip) -- logLikelihood / alpha on all GPUs;
//   * lazily, a full single-GPU context on its device (a replica of the model) -- the prediction is sharded over
//     the test points with the factor replicated ("embarrassingly parallel if L is replicated"),
//     and the calls that are not distributed (Gram copies, solve_chol, gradients of other compositions) run on the
//     replica of device 0.  The ExpAns(+Bias) gradient is distributed (gpak_dist_grad).
// Collectives between the threads: RCCL (one communicator per thread, ncclCommInitRank with a shared id), or --
// when RCCL cannot be used (several ranks on ONE device, which is all a test box has; GPAK_MULTI_TRANSPORT=local;
// a failed RCCL start-up) -- an in-process transport: the root publishes its buffer and an event, the receivers
// pull it with hipMemcpyPeerAsync on their own communication streams.
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <limits>
#include <mutex>
#include <set>
#include <thread>

#include "../../include/gpak_dist.h"
#include "gpak_internal.h"

int gpak_predict_impl(gpak_ctx *ctx, const double *Xte, long M, double *mean, double *var, const double *pool_sum,
                      long pool_M);

namespace {

// ---- in-process transport ------------------------------------------------------------------------
struct LocalGroup {
  int P = 1;
  std::mutex m;
  std::condition_variable cv;
  // broadcast rendezvous: one slot, used by one broadcast at a time (every rank issues them in the same order)
  struct Slot {
    bool active = false, root_done = false;
    long seq = -1;
    const void *src = nullptr;
    int src_dev = 0;
    hipEvent_t ready = nullptr;
    int acks = 0;
    std::vector<hipEvent_t> copied;
  } slot;
  // all-reduce staging
  std::vector<std::vector<double>> stage_d;
  std::vector<std::vector<int>> stage_i;
  int arrived = 0;
  long generation = 0;
  void barrier(std::unique_lock<std::mutex> &lk) {
    const long g = generation;
    if (++arrived == P) { arrived = 0; generation++; cv.notify_all(); }
    else cv.wait(lk, [&] { return generation != g; });
  }
};

struct LocalRank {
  LocalGroup *g = nullptr;
  int rank = 0, dev = 0;
  long seq = 0;
  hipEvent_t ready = nullptr, copied = nullptr;
};

int lt_bcast(void *self, void *st, double *buf, size_t count, int root) {
  LocalRank *r = (LocalRank *)self;
  LocalGroup *g = r->g;
  if (g->P == 1) return GPAK_OK;
  const long seq = r->seq++;
  hipStream_t s = (hipStream_t)st;
  LocalGroup::Slot &sl = g->slot;
  std::unique_lock<std::mutex> lk(g->m);
  if (r->rank == root) {
    if (hipEventRecord(r->ready, s) != hipSuccess) return GPAK_EHIP;
    g->cv.wait(lk, [&] { return !sl.active; });                 // the previous broadcast has drained
    sl.active = true; sl.root_done = false; sl.seq = seq; sl.src = buf; sl.src_dev = r->dev; sl.ready = r->ready;
    sl.acks = 0;
    sl.copied.clear();
    g->cv.notify_all();
    // the broadcast is complete on the root's stream once every receiver has read the buffer
    g->cv.wait(lk, [&] { return (int)sl.copied.size() == g->P - 1; });
    int rc = GPAK_OK;
    for (hipEvent_t e : sl.copied)
      if (hipStreamWaitEvent(s, e, 0) != hipSuccess) rc = GPAK_EHIP;
    sl.root_done = true;
    g->cv.notify_all();
    return rc;
  }
  g->cv.wait(lk, [&] { return sl.active && sl.seq == seq; });
  const void *src = sl.src;
  const int src_dev = sl.src_dev;
  hipEvent_t ready = sl.ready;
  lk.unlock();
  int rc = GPAK_OK;
  if (hipStreamWaitEvent(s, ready, 0) != hipSuccess) rc = GPAK_EHIP;
  hipError_t e = (src_dev == r->dev) ? hipMemcpyAsync(buf, src, sizeof(double) * count, hipMemcpyDeviceToDevice, s)
                                     : hipMemcpyPeerAsync(buf, r->dev, src, src_dev, sizeof(double) * count, s);
  if (e != hipSuccess || hipEventRecord(r->copied, s) != hipSuccess) rc = GPAK_EHIP;
  lk.lock();
  sl.copied.push_back(r->copied);
  g->cv.notify_all();
  // r->copied is re-recorded by this rank's next broadcast: leave only after the root has queued its wait on it
  g->cv.wait(lk, [&] { return sl.root_done; });
  if (++sl.acks == g->P - 1) { sl.active = false; sl.root_done = false; g->cv.notify_all(); }
  return rc;
}

template <typename T, typename Op>
int lt_allreduce(LocalRank *r, hipStream_t s, T *buf, size_t count, std::vector<std::vector<T>> &stage, Op op) {
  LocalGroup *g = r->g;
  if (g->P == 1) return GPAK_OK;
  std::vector<T> mine(count);
  if (hipMemcpyAsync(mine.data(), buf, sizeof(T) * count, hipMemcpyDeviceToHost, s) != hipSuccess ||
      hipStreamSynchronize(s) != hipSuccess) return GPAK_EHIP;
  std::vector<T> out(count);
  {
    std::unique_lock<std::mutex> lk(g->m);
    stage[r->rank] = std::move(mine);
    g->barrier(lk);
    out = stage[0];
    for (int q = 1; q < g->P; q++)            // rank order: the same sum, bit for bit, on every rank
      for (size_t i = 0; i < count; i++) out[i] = op(out[i], stage[q][i]);
    g->barrier(lk);                           // nobody overwrites its stage entry before everyone has read it
  }
  if (hipMemcpyAsync(buf, out.data(), sizeof(T) * count, hipMemcpyHostToDevice, s) != hipSuccess ||
      hipStreamSynchronize(s) != hipSuccess) return GPAK_EHIP;
  return GPAK_OK;
}
int lt_allreduce_sum(void *self, void *st, double *buf, size_t count) {
  LocalRank *r = (LocalRank *)self;
  return lt_allreduce(r, (hipStream_t)st, buf, count, r->g->stage_d, [](double a, double b) { return a + b; });
}
int lt_allreduce_min_int(void *self, void *st, int *buf, size_t count) {
  LocalRank *r = (LocalRank *)self;
  return lt_allreduce(r, (hipStream_t)st, buf, count, r->g->stage_i, [](int a, int b) { return a < b ? a : b; });
}

}  // namespace

// ---- the group -------------------------------------------------------------------------------------
struct gpak_multi {
  int P = 1;
  int precision = GPAK_F64;
  std::vector<int> devices;
  std::vector<gpak_dist *> ranks;
  std::vector<gpak_ctx *> replicas;          // lazily created full contexts, one per device
  std::vector<LocalRank> local_ranks;
  LocalGroup local;
  bool use_rccl = false;
  std::string transport_name;
  // worker threads and the job they all run
  std::vector<std::thread> threads;
  std::mutex m;
  std::condition_variable cv;
  long job_seq = 0;
  int pending = 0;
  bool quit = false;
  std::function<int(int)> job;
  std::vector<int> rc;
  // model state mirrored for the replicas
  std::vector<double> X, y;
  int N = 0, d = 0;
  bool have_params = false, general_kernel = false;
  double expans[8] = {0}, bias = 0, sn2 = 0;
  int dist_mode = GPAK_DIST_DIRECT;
  std::vector<char> replica_train_ok, replica_params_ok;
  int nb = 512;
  std::string err;
  int failed_col = 0;

  int run(std::function<int(int)> f) {
    std::unique_lock<std::mutex> lk(m);
    job = std::move(f);
    rc.assign(P, GPAK_OK);
    pending = P;
    job_seq++;
    cv.notify_all();
    cv.wait(lk, [&] { return pending == 0; });
    for (int r = 0; r < P; r++) if (rc[r] != GPAK_OK) return rc[r];
    return GPAK_OK;
  }
  void worker(int r) {
    hipSetDevice(devices[r]);
    long seen = 0;
    for (;;) {
      std::function<int(int)> f;
      {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return quit || job_seq != seen; });
        if (quit) return;
        seen = job_seq;
        f = job;
      }
      const int v = f(r);
      std::unique_lock<std::mutex> lk(m);
      rc[r] = v;
      if (--pending == 0) cv.notify_all();
    }
  }
};

static int ensure_replica(gpak_multi *g, int r) {
  if (!g->replicas[r]) {
    int rc = gpak_create(&g->replicas[r], g->devices[r], g->precision);
    if (rc) { g->err = std::string("replica context: ") + gpak_global_error(); return rc; }
  }
  gpak_ctx *c = g->replicas[r];
  if (!g->replica_train_ok[r]) {
    int rc = gpak_set_train(c, g->X.data(), g->y.data(), g->N, g->d);
    if (rc) { g->err = gpak_last_error(c); return rc; }
    g->replica_train_ok[r] = 1;
    g->replica_params_ok[r] = 0;
  }
  if (!g->replica_params_ok[r]) {
    if (!g->have_params) { g->err = "no parameters (gpak_set_params)"; return GPAK_ESTATE; }
    int rc = gpak_set_params(c, g->expans, g->bias, g->sn2, g->dist_mode);
    if (rc) { g->err = gpak_last_error(c); return rc; }
    g->replica_params_ok[r] = 1;
  }
  return GPAK_OK;
}

// ---- entry points used by api.hip when ctx->multi is set -------------------------------------------------
void gpak_multi_destroy(gpak_multi *g) {
  if (!g) return;
  g->run([&](int r) {
    if (g->ranks[r]) gpak_dist_destroy(g->ranks[r]);
    if (g->replicas[r]) gpak_destroy(g->replicas[r]);
    if (g->local_ranks[r].ready) hipEventDestroy(g->local_ranks[r].ready);
    if (g->local_ranks[r].copied) hipEventDestroy(g->local_ranks[r].copied);
    return GPAK_OK;
  });
  {
    std::unique_lock<std::mutex> lk(g->m);
    g->quit = true;
    g->cv.notify_all();
  }
  for (std::thread &t : g->threads) t.join();
  delete g;
}

int gpak_multi_create(gpak_multi **out, int n, const int *devices, int precision, std::string &err) {
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) { err = "no HIP device available (libgpak_hip has no CPU fallback)"; return GPAK_EHIP; }
  gpak_multi *g = new gpak_multi();
  g->P = n; g->precision = precision;
  // GPAK_MULTI_DEVICES="0,0,...": ordinals for a caller that passes none (the CLI) -- tests rehearse several ranks on
  // the one GPU of a test box with it
  std::vector<int> from_env;
  if (!devices && getenv("GPAK_MULTI_DEVICES")) {
    for (const char *p = getenv("GPAK_MULTI_DEVICES"); *p;) {
      from_env.push_back(atoi(p));
      while (*p && *p != ',') p++;
      if (*p == ',') p++;
    }
    if ((int)from_env.size() == n) devices = from_env.data();
  }
  for (int r = 0; r < n; r++) {
    const int dev = devices ? devices[r] : r;
    if (dev < 0 || dev >= count) { err = "gpak_create_multi: device ordinal out of range (more ranks than GPUs?)"; delete g; return GPAK_EINVAL; }
    g->devices.push_back(dev);
  }
  g->ranks.assign(n, nullptr); g->replicas.assign(n, nullptr);
  g->replica_train_ok.assign(n, 0); g->replica_params_ok.assign(n, 0);
  g->local_ranks.resize(n);
  g->local.P = n; g->local.stage_d.resize(n); g->local.stage_i.resize(n);
  // RCCL needs one device per rank; several ranks on one device (a test box) use the in-process transport
  const bool distinct = std::set<int>(g->devices.begin(), g->devices.end()).size() == (size_t)n;
  const char *tr = getenv("GPAK_MULTI_TRANSPORT");
  g->use_rccl = n > 1 && distinct && !(tr && !strcmp(tr, "local"));
  char id[GPAK_DIST_ID_BYTES];
  if (g->use_rccl && gpak_dist_rccl_unique_id(id) != GPAK_OK) g->use_rccl = false;   // librccl missing: in-process transport
  for (int r = 0; r < n; r++) g->threads.emplace_back(&gpak_multi::worker, g, r);
  for (int attempt = 0; attempt < 2; attempt++) {
    std::atomic<int> rccl_failed{0};
    int rc = g->run([&](int r) {
      if (g->ranks[r]) { gpak_dist_destroy(g->ranks[r]); g->ranks[r] = nullptr; }
      LocalRank &lr = g->local_ranks[r];
      lr.g = &g->local; lr.rank = r; lr.dev = g->devices[r]; lr.seq = 0;
      if (!lr.ready && (hipEventCreateWithFlags(&lr.ready, hipEventDisableTiming) != hipSuccess ||
                        hipEventCreateWithFlags(&lr.copied, hipEventDisableTiming) != hipSuccess)) return (int)GPAK_EHIP;
      gpak_dist_transport t;
      t.self = &lr; t.bcast = lt_bcast; t.allreduce_sum = lt_allreduce_sum; t.allreduce_min_int = lt_allreduce_min_int;
      int v = gpak_dist_create(&g->ranks[r], r, g->P, g->devices[r], nullptr, g->use_rccl ? nullptr : &t);
      if (v) return v;
      if (g->use_rccl && gpak_dist_init_rccl(g->ranks[r], id) != GPAK_OK) { rccl_failed++; return (int)GPAK_OK; }
      return (int)GPAK_OK;
    });
    if (rc) { err = "gpak_create_multi: rank creation failed"; gpak_multi_destroy(g); return rc; }
    if (g->use_rccl && rccl_failed.load()) { g->use_rccl = false; continue; }   // all ranks again, in-process transport
    if (g->P > 1) {
      // start-up self-check on every rank at once (collectives on the side stream, CU-masked bulk stream)
      rc = g->run([&](int r) { return gpak_dist_selfcheck(g->ranks[r], nullptr); });
      if (rc && g->use_rccl) { g->use_rccl = false; continue; }
      if (rc) { err = std::string("gpak_create_multi: ") + gpak_dist_last_error(g->ranks[0]); gpak_multi_destroy(g); return rc; }
    }
    break;
  }
  g->transport_name = g->use_rccl ? "rccl" : (n > 1 ? "in-process peer copies" : "none");
  *out = g;
  return GPAK_OK;
}

const char *gpak_multi_error(gpak_multi *g) { return g->err.c_str(); }
const char *gpak_multi_transport(gpak_multi *g) { return g->transport_name.c_str(); }

int gpak_multi_set_train(gpak_multi *g, const double *X, const double *y, int N, int d) {
  if (d != 3) { g->err = "a multi-GPU context handles 3-D inputs"; return GPAK_ENOTIMPL; }
  g->X.assign(X, X + (size_t)N * d); g->y.assign(y, y + N);
  g->N = N; g->d = d;
  std::fill(g->replica_train_ok.begin(), g->replica_train_ok.end(), 0);
  int rc = g->run([&](int r) { return gpak_dist_set_train(g->ranks[r], g->X.data(), g->y.data(), N, d, g->nb); });
  if (rc) g->err = gpak_dist_last_error(g->ranks[0]);
  return rc;
}

int gpak_multi_set_params(gpak_multi *g, const double *expans, double bias, double sn2, int dist_mode) {
  memcpy(g->expans, expans, sizeof(double) * 8);
  g->bias = bias; g->sn2 = sn2; g->dist_mode = dist_mode; g->have_params = true;
  std::fill(g->replica_params_ok.begin(), g->replica_params_ok.end(), 0);
  for (int r = 0; r < g->P; r++) {
    int rc = gpak_dist_set_params(g->ranks[r], expans, bias, sn2, dist_mode);
    if (rc) { g->err = gpak_dist_last_error(g->ranks[r]); return rc; }
  }
  return GPAK_OK;
}

int gpak_multi_nlz(gpak_multi *g, double *nlz, double *quad, double *sumlp, double *logdet) {
  if (!g->N) { g->err = "no training set (gpak_set_train)"; return GPAK_ESTATE; }
  std::vector<double> v(g->P, std::numeric_limits<double>::quiet_NaN());
  int rc = g->run([&](int r) { return gpak_dist_nlz(g->ranks[r], &v[r]); });
  if (nlz) *nlz = v[0];
  if (rc) { g->err = gpak_dist_last_error(g->ranks[0]); return rc; }
  if (quad || sumlp || logdet) rc = gpak_dist_nlz_terms(g->ranks[0], quad, sumlp, logdet);
  return rc;
}

int gpak_multi_alpha(gpak_multi *g, double *alpha_host) {
  double v;
  int rc = gpak_multi_nlz(g, &v, nullptr, nullptr, nullptr);
  if (rc) return rc;
  if (!alpha_host) return GPAK_OK;
  rc = g->run([&](int r) { return r == 0 ? gpak_dist_get_alpha(g->ranks[0], alpha_host) : (int)GPAK_OK; });
  if (rc) g->err = gpak_dist_last_error(g->ranks[0]);
  return rc;
}

// GP_utils::GradLL on the group: B^-1 by row blocks over the ranks (gpak_dist_grad)
int gpak_multi_grad(gpak_multi *g, double *grad10) {
  std::vector<std::vector<double>> gs(g->P, std::vector<double>(10, 0.0));
  int rc = g->run([&](int r) { return gpak_dist_grad(g->ranks[r], gs[r].data()); });
  if (rc) { g->err = gpak_dist_last_error(g->ranks[0]); return rc; }
  memcpy(grad10, gs[0].data(), sizeof(double) * 10);
  return GPAK_OK;
}

int gpak_multi_stats(gpak_multi *g, int r, gpak_dist_stats *out) { return gpak_dist_get_stats(g->ranks[r], out); }

// the calls that are not distributed run on the replica of device 0 (on its own thread: a HIP context per thread)
int gpak_multi_on_replica0(gpak_multi *g, const std::function<int(gpak_ctx *)> &f) {
  int rc = g->run([&](int r) {
    if (r != 0) return (int)GPAK_OK;
    int v = ensure_replica(g, 0);
    if (v) return v;
    v = f(g->replicas[0]);
    if (v) g->err = gpak_last_error(g->replicas[0]);
    return v;
  });
  return rc;
}

// GP_utils::posteriorMeanVar with the test points sharded over the devices; every device factors its own replica
// (N^3/3 each, concurrently) and predicts a contiguous slice with the pooled mean of ALL test points
int gpak_multi_predict(gpak_multi *g, const double *Xte, long M, int d, double *mean, double *var) {
  if (d != g->d) { g->err = "test points must have as many columns as the training set"; return GPAK_EINVAL; }
  double s2[4] = {0, 0, 0, 0};
  for (int k = 0; k < d; k++)
    for (long i = 0; i < M; i++) s2[k] += Xte[i + (size_t)k * M];
  const long per = ((M + g->P - 1) / g->P + 255) / 256 * 256;
  return g->run([&](int r) {
    const long m0 = std::min(M, r * per), m1 = std::min(M, (r + 1) * per);
    if (m1 <= m0) return (int)GPAK_OK;
    int v = ensure_replica(g, r);
    if (v) return v;
    gpak_ctx *c = g->replicas[r];
    double nlz;
    v = gpak_nlz(c, &nlz);   // _postVar calls logLikelihood() (GP_Utils.cpp:980)
    if (v) { g->err = gpak_last_error(c); return v; }
    // the slice as its own column-major (m1-m0) x d array
    std::vector<double> xs((size_t)(m1 - m0) * d);
    for (int k = 0; k < d; k++) memcpy(xs.data() + (size_t)k * (m1 - m0), Xte + (size_t)k * M + m0, sizeof(double) * (m1 - m0));
    v = gpak_predict_impl(c, xs.data(), m1 - m0, mean + m0, var ? var + m0 : nullptr, s2, M);
    if (v) g->err = gpak_last_error(c);
    return v;
  });
}

// phase times of a group: the dist rank 0 view of the last step, mapped onto gpak_phase_times
int gpak_multi_timing(gpak_multi *g, gpak_phase_times *out) {
  memset(out, 0, sizeof(*out));
  gpak_dist_stats st;
  int rc = gpak_dist_get_stats(g->ranks[0], &st);
  if (rc) return rc;
  out->gram_ms = st.fill_ms; out->factor_ms = st.factor_ms; out->solve_ms = st.solve_ms; out->nlz_ms = st.nlz_ms;
  out->trailing_ms = st.bulk_ms; out->trailing_flops = st.bulk_flops;
  out->n = st.n; out->n_padded = st.n_padded;
  if (g->replicas[0]) {
    gpak_phase_times t;
    if (gpak_timing(g->replicas[0], &t) == GPAK_OK) { out->predict_ms = t.predict_ms; out->grad_ms = t.grad_ms; }
  }
  return GPAK_OK;
}
int gpak_multi_n(gpak_multi *g) { return g->P; }
