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
// dist.hip (not part of the C-ABI): all communicators of one process made by ONE ncclCommInitAll call
extern "C" int gpak_dist_rccl_init_all(int n, const int *devices, void **comms, std::string &err);
extern "C" void gpak_dist_rccl_destroy(void *comm);
extern "C" int gpak_dist_adopt_rccl(gpak_dist *h, void *comm);
void gpak_dist_hip_services(gpak_dist_engine *e);   // memory / stream / event entries of the built-in HIP engine

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
    void *ready = nullptr;
    int acks = 0;
    std::vector<void *> copied;
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
  // memory / stream / event services of the rank's engine (the HIP services of dist.hip, or a test's callback table):
  // the rendezvous below never calls HIP itself, so a CPU harness can drive it thread per rank under a sanitizer
  gpak_dist_engine E;
  bool hip = false;            // built-in HIP engine: buffers of different devices are copied with hipMemcpyPeerAsync
  void *ready = nullptr, *copied = nullptr;
};

int lt_bcast(void *self, void *st, double *buf, size_t count, int root) {
  LocalRank *r = (LocalRank *)self;
  LocalGroup *g = r->g;
  if (g->P == 1) return GPAK_OK;
  const gpak_dist_engine &E = r->E;
  const long seq = r->seq++;
  LocalGroup::Slot &sl = g->slot;
  std::unique_lock<std::mutex> lk(g->m);
  if (r->rank == root) {
    if (E.event_record(E.self, r->ready, st) != GPAK_OK) return GPAK_EHIP;
    g->cv.wait(lk, [&] { return !sl.active; });                 // the previous broadcast has drained
    sl.active = true; sl.root_done = false; sl.seq = seq; sl.src = buf; sl.src_dev = r->dev; sl.ready = r->ready;
    sl.acks = 0;
    sl.copied.clear();
    g->cv.notify_all();
    // the broadcast is complete on the root's stream once every receiver has read the buffer
    g->cv.wait(lk, [&] { return (int)sl.copied.size() == g->P - 1; });
    int rc = GPAK_OK;
    for (void *e : sl.copied)
      if (E.stream_wait_event(E.self, st, e) != GPAK_OK) rc = GPAK_EHIP;
    sl.root_done = true;
    g->cv.notify_all();
    return rc;
  }
  g->cv.wait(lk, [&] { return sl.active && sl.seq == seq; });
  const void *src = sl.src;
  const int src_dev = sl.src_dev;
  void *ready = sl.ready;
  lk.unlock();
  int rc = GPAK_OK;
  if (E.stream_wait_event(E.self, st, ready) != GPAK_OK) rc = GPAK_EHIP;
  if (r->hip && src_dev != r->dev) {
    if (hipMemcpyPeerAsync(buf, r->dev, src, src_dev, sizeof(double) * count, (hipStream_t)st) != hipSuccess) rc = GPAK_EHIP;
  } else if (E.copy(E.self, st, buf, src, sizeof(double) * count) != GPAK_OK) {
    rc = GPAK_EHIP;
  }
  if (E.event_record(E.self, r->copied, st) != GPAK_OK) rc = GPAK_EHIP;
  lk.lock();
  sl.copied.push_back(r->copied);
  g->cv.notify_all();
  // r->copied is re-recorded by this rank's next broadcast: leave only after the root has queued its wait on it
  g->cv.wait(lk, [&] { return sl.root_done; });
  if (++sl.acks == g->P - 1) { sl.active = false; sl.root_done = false; g->cv.notify_all(); }
  return rc;
}

template <typename T, typename Op>
int lt_allreduce(LocalRank *r, void *s, T *buf, size_t count, std::vector<std::vector<T>> &stage, Op op) {
  LocalGroup *g = r->g;
  if (g->P == 1) return GPAK_OK;
  const gpak_dist_engine &E = r->E;
  std::vector<T> mine(count);
  if (E.download(E.self, s, mine.data(), buf, sizeof(T) * count) != GPAK_OK) return GPAK_EHIP;   // complete on return
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
  if (E.upload(E.self, s, buf, out.data(), sizeof(T) * count) != GPAK_OK || E.stream_sync(E.self, s) != GPAK_OK)
    return GPAK_EHIP;                         // `out` is a local: the copy must have left it before this returns
  return GPAK_OK;
}
int lt_allreduce_sum(void *self, void *st, double *buf, size_t count) {
  LocalRank *r = (LocalRank *)self;
  return lt_allreduce(r, st, buf, count, r->g->stage_d, [](double a, double b) { return a + b; });
}
int lt_allreduce_min_int(void *self, void *st, int *buf, size_t count) {
  LocalRank *r = (LocalRank *)self;
  return lt_allreduce(r, st, buf, count, r->g->stage_i, [](int a, int b) { return a < b ? a : b; });
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
  std::vector<gpak_dist_engine> test_engines;   // gpak_create_multi_with_engines: caller-supplied engine per rank (tests)
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
  // a general HybKerns composition (gpak_set_kernel on the group): kinds / parameters / Kern_White as given
  int nterms = 1, kinds[GPAK_MAX_TERMS] = {0, 0, 0};
  std::vector<double> pars;
  double white = 0;
  std::vector<char> replica_train_ok, replica_params_ok;
  int nb = 512;
  std::string err;                   // written by the caller's thread only, after run(): see fail()
  std::vector<std::string> rank_err; // written by worker r only (slot r), read after run()
  int failed_col = 0;
  double acc_ms[4] = {0, 0, 0, 0};   // fill / factor / solve / nlz of rank 0, summed over the evaluations since set_train
  int evaluations = 0;
  bool factor_current = false;       // the ranks hold the result of the current parameters (set by gpak_multi_nlz)
  std::vector<char> replica_factor_ok;   // the replica holds the CURRENT factor (imported from the rank's packed panels)

  // worker r reports a failure: its own slot, no shared string
  int fail(int r, int rc, const std::string &what) { rank_err[r] = what; return rc; }
  // caller's thread, after run(): the first rank's message becomes the group's
  void collect_errors() {
    for (int r = 0; r < P; r++)
      if (!rank_err[r].empty()) { err = "rank " + std::to_string(r) + ": " + rank_err[r]; break; }
  }

  int run(std::function<int(int)> f) {
    std::unique_lock<std::mutex> lk(m);
    job = std::move(f);
    rc.assign(P, GPAK_OK);
    for (std::string &e : rank_err) e.clear();
    pending = P;
    job_seq++;
    cv.notify_all();
    cv.wait(lk, [&] { return pending == 0; });
    collect_errors();
    for (int r = 0; r < P; r++) if (rc[r] != GPAK_OK) return rc[r];
    return GPAK_OK;
  }
  void worker(int r) {
    if (test_engines.empty()) hipSetDevice(devices[r]);
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
  if (!g->test_engines.empty()) return g->fail(r, GPAK_ENOTIMPL, "a group over caller-supplied engines has no device replicas");
  if (!g->replicas[r]) {
    int rc = gpak_create(&g->replicas[r], g->devices[r], g->precision);
    if (rc) return g->fail(r, rc, std::string("replica context: ") + gpak_global_error());
  }
  gpak_ctx *c = g->replicas[r];
  if (!g->replica_train_ok[r]) {
    int rc = gpak_set_train(c, g->X.data(), g->y.data(), g->N, g->d);
    if (rc) return g->fail(r, rc, gpak_last_error(c));
    g->replica_train_ok[r] = 1;
    g->replica_params_ok[r] = 0;
  }
  if (!g->replica_params_ok[r]) {
    if (!g->have_params) return g->fail(r, GPAK_ESTATE, "no parameters (gpak_set_params)");
    int rc = g->general_kernel ? gpak_set_kernel(c, g->nterms, g->kinds, g->pars.data(), g->bias, g->white, g->sn2, g->dist_mode)
                               : gpak_set_params(c, g->expans, g->bias, g->sn2, g->dist_mode);
    if (rc) return g->fail(r, rc, gpak_last_error(c));
    g->replica_params_ok[r] = 1;
    g->replica_factor_ok[r] = 0;
  }
  return GPAK_OK;
}

// ---- entry points used by api.hip when ctx->multi is set -------------------------------------------------
void gpak_multi_destroy(gpak_multi *g) {
  if (!g) return;
  g->run([&](int r) {
    if (g->ranks[r]) gpak_dist_destroy(g->ranks[r]);
    if (g->replicas[r]) gpak_destroy(g->replicas[r]);
    LocalRank &lr = g->local_ranks[r];
    if (lr.ready) lr.E.event_destroy(lr.E.self, lr.ready);
    if (lr.copied) lr.E.event_destroy(lr.E.self, lr.copied);
    lr.ready = lr.copied = nullptr;
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

// engines: nullptr (the product: built-in HIP engine per device) or n caller-supplied tables (tests: the thread-per-rank
// host logic and the in-process transport driven on a box without a GPU, e.g. under a sanitizer)
int gpak_multi_create(gpak_multi **out, int n, const int *devices, int precision, std::string &err,
                      const gpak_dist_engine *const *engines) {
  *out = nullptr;
  int count = 0;
  if (engines) count = n;
  else if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) { err = "no HIP device available (libgpak_hip has no CPU fallback)"; return GPAK_EHIP; }
  gpak_multi *g = new gpak_multi();
  g->P = n; g->precision = precision;
  if (engines) for (int r = 0; r < n; r++) g->test_engines.push_back(*engines[r]);
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
  g->replica_train_ok.assign(n, 0); g->replica_params_ok.assign(n, 0); g->replica_factor_ok.assign(n, 0);
  g->rank_err.assign(n, std::string());
  g->local_ranks.resize(n);
  g->local.P = n; g->local.stage_d.resize(n); g->local.stage_i.resize(n);
  // RCCL needs one device per rank; several ranks on one device (a test box) use the in-process transport
  const bool distinct = std::set<int>(g->devices.begin(), g->devices.end()).size() == (size_t)n;
  const char *tr = getenv("GPAK_MULTI_TRANSPORT");
  g->use_rccl = n > 1 && distinct && !engines && !(tr && !strcmp(tr, "local"));
  // Whether RCCL can be used is decided HERE, on the caller's thread, before any rank exists: ncclCommInitAll makes
  // the communicators of all ranks in one call -- all of them or none -- so no worker can be left waiting in a
  // rendezvous that a failed peer never joins (ncclCommInitRank per thread had exactly that failure mode).
  std::vector<void *> comms(n, nullptr);
  std::string rccl_note;
  if (g->use_rccl) {
    std::string why;
    if (gpak_dist_rccl_init_all(n, g->devices.data(), comms.data(), why) != GPAK_OK) {
      g->use_rccl = false;
      rccl_note = " (RCCL start-up failed: " + why + ")";
      (void)hipGetLastError();
    }
  }
  for (int r = 0; r < n; r++) g->threads.emplace_back(&gpak_multi::worker, g, r);
  for (int attempt = 0; attempt < 2; attempt++) {
    int rc = g->run([&](int r) {
      if (g->ranks[r]) { gpak_dist_destroy(g->ranks[r]); g->ranks[r] = nullptr; }
      LocalRank &lr = g->local_ranks[r];
      lr.g = &g->local; lr.rank = r; lr.dev = g->devices[r]; lr.seq = 0;
      if (!lr.ready) {
        lr.hip = g->test_engines.empty();
        if (lr.hip) gpak_dist_hip_services(&lr.E); else lr.E = g->test_engines[r];
        lr.ready = lr.E.event_create(lr.E.self, 0);
        lr.copied = lr.E.event_create(lr.E.self, 0);
        if (!lr.ready || !lr.copied) return g->fail(r, GPAK_EHIP, "event creation failed");
      }
      // the in-process transport pulls with hipMemcpyPeerAsync: map every other device of the group into this one
      // (without it the copy is staged through the host by the runtime -- correct, but not what xGMI is for)
      if (attempt == 0 && lr.hip)
        for (int q = 0; q < g->P; q++) {
          if (g->devices[q] == g->devices[r]) continue;
          int can = 0;
          if (hipDeviceCanAccessPeer(&can, g->devices[r], g->devices[q]) == hipSuccess && can) {
            hipError_t e = hipDeviceEnablePeerAccess(g->devices[q], 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
          }
          (void)hipGetLastError();
        }
      gpak_dist_transport t;
      t.self = &lr; t.bcast = lt_bcast; t.allreduce_sum = lt_allreduce_sum; t.allreduce_min_int = lt_allreduce_min_int;
      int v = gpak_dist_create(&g->ranks[r], r, g->P, g->devices[r], lr.hip ? nullptr : &lr.E, g->use_rccl ? nullptr : &t);
      if (v) return g->fail(r, v, "gpak_dist_create failed");
      if (g->use_rccl) {
        v = gpak_dist_adopt_rccl(g->ranks[r], comms[r]);   // the handle owns the communicator from here on
        if (v) return g->fail(r, v, gpak_dist_last_error(g->ranks[r]));
        comms[r] = nullptr;
      }
      return (int)GPAK_OK;
    });
    if (rc) { err = "gpak_create_multi: rank creation failed: " + g->err; gpak_multi_destroy(g); return rc; }
    if (g->P > 1) {
      // start-up self-check on every rank at once (collectives on the side stream, CU-masked bulk stream)
      rc = g->run([&](int r) {
        int v = gpak_dist_selfcheck(g->ranks[r], nullptr);
        return v ? g->fail(r, v, gpak_dist_last_error(g->ranks[r])) : (int)GPAK_OK;
      });
      if (rc && g->use_rccl) { g->use_rccl = false; rccl_note = " (RCCL self-check failed: " + g->err + ")"; continue; }   // all ranks again
      if (rc) { err = std::string("gpak_create_multi: ") + g->err; gpak_multi_destroy(g); return rc; }
    }
    break;
  }
  for (void *c : comms) gpak_dist_rccl_destroy(c);   // communicators nobody adopted (the fall-back path)
  g->transport_name = g->use_rccl ? "rccl" : (n > 1 ? "in-process peer copies" + rccl_note : "none");
  *out = g;
  return GPAK_OK;
}

const char *gpak_multi_error(gpak_multi *g) { return g->err.c_str(); }
const char *gpak_multi_transport(gpak_multi *g) { return g->transport_name.c_str(); }
int gpak_multi_failed_column(gpak_multi *g) { return g->failed_col; }

int gpak_multi_set_train(gpak_multi *g, const double *X, const double *y, int N, int d) {
  if (d != 3 && d != 4) { g->err = "inputs must have 3 or 4 columns"; return GPAK_ENOTIMPL; }
  g->X.assign(X, X + (size_t)N * d); g->y.assign(y, y + N);
  g->N = N; g->d = d;
  std::fill(g->replica_train_ok.begin(), g->replica_train_ok.end(), 0);
  std::fill(g->replica_factor_ok.begin(), g->replica_factor_ok.end(), 0);
  g->acc_ms[0] = g->acc_ms[1] = g->acc_ms[2] = g->acc_ms[3] = 0;
  g->evaluations = 0;
  g->failed_col = 0;
  return g->run([&](int r) {
    int v = gpak_dist_set_train(g->ranks[r], g->X.data(), g->y.data(), N, d, g->nb);
    return v ? g->fail(r, v, gpak_dist_last_error(g->ranks[r])) : (int)GPAK_OK;
  });
}

int gpak_multi_set_params(gpak_multi *g, const double *expans, double bias, double sn2, int dist_mode) {
  memcpy(g->expans, expans, sizeof(double) * 8);
  g->bias = bias; g->sn2 = sn2; g->dist_mode = dist_mode; g->have_params = true;
  g->general_kernel = false; g->white = 0.0;
  std::fill(g->replica_params_ok.begin(), g->replica_params_ok.end(), 0);
  std::fill(g->replica_factor_ok.begin(), g->replica_factor_ok.end(), 0);
  g->factor_current = false;
  for (int r = 0; r < g->P; r++) {
    int rc = gpak_dist_set_params(g->ranks[r], expans, bias, sn2, dist_mode);
    if (rc) { g->err = gpak_dist_last_error(g->ranks[r]); return rc; }
  }
  return GPAK_OK;
}

// HybKerns of other children on the group (gpak_set_kernel): logLikelihood / alpha distributed with the serialized
// composition, prediction and solve_chol on the imported factor, the children's gradients on device 0 (gpak_multi_grad_hyb)
int gpak_multi_set_kernel(gpak_multi *g, int nterms, const int *kinds, const double *pars, double bias, double white,
                          double sn2, int dist_mode) {
  int np = 0;
  for (int t = 0; t < nterms; t++) {
    g->kinds[t] = kinds[t];
    if (kinds[t] == GPAK_KERN_EXPANS) memcpy(g->expans, pars + np, sizeof(double) * 8);
    np += kinds[t] == GPAK_KERN_EXPANS ? 8 : kinds[t] == GPAK_KERN_EXP ? 2 : 3;
  }
  g->nterms = nterms;
  g->pars.assign(pars, pars + np);
  g->bias = bias; g->white = white; g->sn2 = sn2; g->dist_mode = dist_mode; g->have_params = true;
  g->general_kernel = true;
  std::fill(g->replica_params_ok.begin(), g->replica_params_ok.end(), 0);
  std::fill(g->replica_factor_ok.begin(), g->replica_factor_ok.end(), 0);
  g->factor_current = false;
  for (int r = 0; r < g->P; r++) {
    int rc = gpak_dist_set_kernel(g->ranks[r], nterms, kinds, pars, bias, white, sn2, dist_mode);
    if (rc) { g->err = gpak_dist_last_error(g->ranks[r]); return rc; }
  }
  return GPAK_OK;
}

int gpak_multi_nlz(gpak_multi *g, double *nlz, double *quad, double *sumlp, double *logdet) {
  if (!g->N) { g->err = "no training set (gpak_set_train)"; return GPAK_ESTATE; }
  std::vector<double> v(g->P, std::numeric_limits<double>::quiet_NaN());
  const bool fresh = !g->factor_current;
  int rc = g->run([&](int r) {
    int s = gpak_dist_nlz(g->ranks[r], &v[r]);
    return s ? g->fail(r, s, gpak_dist_last_error(g->ranks[r])) : (int)GPAK_OK;
  });
  if (nlz) *nlz = v[0];
  g->failed_col = gpak_dist_failed_column(g->ranks[0]);   // min-reduced inside factor(): the same on every rank
  if (fresh) {   // a new evaluation (not a cached result): rank 0's phase times join the accumulated totals
    gpak_dist_stats st;
    if (gpak_dist_get_stats(g->ranks[0], &st) == GPAK_OK) {
      g->acc_ms[0] += st.fill_ms; g->acc_ms[1] += st.factor_ms;
      if (rc == GPAK_OK) { g->acc_ms[2] += st.solve_ms; g->acc_ms[3] += st.nlz_ms; }
      g->evaluations++;
    }
  }
  if (rc) return rc;
  g->factor_current = true;
  if (quad || sumlp || logdet) rc = gpak_dist_nlz_terms(g->ranks[0], quad, sumlp, logdet);
  return rc;
}

int gpak_multi_alpha(gpak_multi *g, double *alpha_host) {
  double v;
  int rc = gpak_multi_nlz(g, &v, nullptr, nullptr, nullptr);
  if (rc) return rc;
  if (!alpha_host) return GPAK_OK;
  return g->run([&](int r) {
    if (r != 0) return (int)GPAK_OK;
    int s = gpak_dist_get_alpha(g->ranks[0], alpha_host);
    return s ? g->fail(0, s, gpak_dist_last_error(g->ranks[0])) : (int)GPAK_OK;
  });
}

// GP_utils::GradLL on the group: B^-1 by row blocks over the ranks (gpak_dist_grad)
int gpak_multi_grad(gpak_multi *g, double *grad10) {
  if (g->general_kernel) { g->err = "gpak_grad handles the ExpAns(+Bias) composition only (use gpak_grad_hyb)"; return GPAK_ENOTIMPL; }
  double v;
  int rc = gpak_multi_nlz(g, &v, nullptr, nullptr, nullptr);   // GradLL re-enters logLikelihood(): GP_Utils.cpp:1173-1174
  if (rc) return rc;
  std::vector<std::vector<double>> gs(g->P, std::vector<double>(10, 0.0));
  rc = g->run([&](int r) {
    int s = gpak_dist_grad(g->ranks[r], gs[r].data());
    return s ? g->fail(r, s, gpak_dist_last_error(g->ranks[r])) : (int)GPAK_OK;
  });
  if (rc) return rc;
  memcpy(grad10, gs[0].data(), sizeof(double) * 10);
  return GPAK_OK;
}

// GradLL of any composition on a group: ExpAns(+Bias) is distributed (gpak_multi_grad); other compositions take the
// children's getGradients on device 0 from the DISTRIBUTED factor (imported into the replica, nothing is factored again)
int gpak_multi_on_replica0(gpak_multi *g, const std::function<int(gpak_ctx *)> &f, bool wants_factor);
int gpak_multi_grad_hyb(gpak_multi *g, double *grad, int ng) {
  if (!g->general_kernel) {
    if (ng != 10) { g->err = "the ExpAns(+Bias) composition has 10 gradient entries"; return GPAK_EINVAL; }
    return gpak_multi_grad(g, grad);
  }
  return gpak_multi_on_replica0(g, [&](gpak_ctx *c) { return gpak_grad_hyb(c, grad, ng); }, true);
}

int gpak_multi_stats(gpak_multi *g, int r, gpak_dist_stats *out) {
  if (r < 0 || r >= g->P) return GPAK_EINVAL;
  return gpak_dist_get_stats(g->ranks[r], out);
}

// bring the replica of rank r up to the group's CURRENT factor without factoring again: every rank keeps the whole
// factor as packed panels (+ the inverted diagonal blocks, alpha and f), on the replica's own device
static int replica_with_factor(gpak_multi *g, int r) {
  int v = ensure_replica(g, r);
  if (v) return v;
  if (g->replica_factor_ok[r]) return GPAK_OK;
  gpak_dist_factor_view view;
  v = gpak_dist_factor_view_get(g->ranks[r], &view);
  if (v) return g->fail(r, v, gpak_dist_last_error(g->ranks[r]));
  v = gpak_import_factor(g->replicas[r], &view);
  if (v) return g->fail(r, v, gpak_last_error(g->replicas[r]));
  g->replica_factor_ok[r] = 1;
  return GPAK_OK;
}

// the calls that are not distributed run on the replica of device 0 (on its own thread: a HIP context per thread);
// calls that only READ the factor (solve_chol, the factor copy) get it imported, the others rebuild what they need
int gpak_multi_on_replica0(gpak_multi *g, const std::function<int(gpak_ctx *)> &f, bool wants_factor) {
  if (wants_factor) {
    double v;
    int rc = gpak_multi_nlz(g, &v, nullptr, nullptr, nullptr);
    if (rc) return rc;
  }
  return g->run([&](int r) {
    if (r != 0) return (int)GPAK_OK;
    int v = wants_factor ? replica_with_factor(g, 0) : ensure_replica(g, 0);
    if (v) return v;
    if (!wants_factor) g->replica_factor_ok[0] = 0;   // such a call may overwrite the replica's matrix buffer
    v = f(g->replicas[0]);
    return v ? g->fail(0, v, gpak_last_error(g->replicas[0])) : (int)GPAK_OK;
  });
}

// GP_utils::posteriorMeanVar with the test points sharded over the devices.  The factor is NOT rebuilt: the group
// evaluates logLikelihood() once (distributed; _postVar calls it, GP_Utils.cpp:980), then every device assembles its
// replica's factor from the packed panels it already holds (device-to-device copies, N^2/2 doubles) and predicts a
// contiguous slice with the pooled mean of ALL test points.
int gpak_multi_predict(gpak_multi *g, const double *Xte, long M, int d, double *mean, double *var) {
  if (d != g->d) { g->err = "test points must have as many columns as the training set"; return GPAK_EINVAL; }
  double nlz;
  int rc = gpak_multi_nlz(g, &nlz, nullptr, nullptr, nullptr);
  if (rc) return rc;
  double s2[4] = {0, 0, 0, 0};
  for (int k = 0; k < d; k++)
    for (long i = 0; i < M; i++) s2[k] += Xte[i + (size_t)k * M];
  const long per = ((M + g->P - 1) / g->P + 255) / 256 * 256;
  return g->run([&](int r) {
    const long m0 = std::min(M, r * per), m1 = std::min(M, (r + 1) * per);
    if (m1 <= m0) return (int)GPAK_OK;
    int v = replica_with_factor(g, r);
    if (v) return v;
    gpak_ctx *c = g->replicas[r];
    // the slice as its own column-major (m1-m0) x d array
    std::vector<double> xs((size_t)(m1 - m0) * d);
    for (int k = 0; k < d; k++) memcpy(xs.data() + (size_t)k * (m1 - m0), Xte + (size_t)k * M + m0, sizeof(double) * (m1 - m0));
    v = gpak_predict_impl(c, xs.data(), m1 - m0, mean + m0, var ? var + m0 : nullptr, s2, M);
    return v ? g->fail(r, v, gpak_last_error(c)) : (int)GPAK_OK;
  });
}

// phase times of a group: rank 0's view of the last step and the totals since gpak_set_train, mapped onto
// gpak_phase_times
int gpak_multi_timing(gpak_multi *g, gpak_phase_times *out) {
  memset(out, 0, sizeof(*out));
  gpak_dist_stats st;
  int rc = gpak_dist_get_stats(g->ranks[0], &st);
  if (rc) return rc;
  out->gram_ms = st.fill_ms; out->factor_ms = st.factor_ms; out->solve_ms = st.solve_ms; out->nlz_ms = st.nlz_ms;
  out->trailing_ms = st.bulk_ms; out->trailing_flops = st.bulk_flops; out->trailing_bytes = st.bulk_bytes;
  out->trailing_launches = (int)st.bulk_launches;
  out->kmatvec_ms = st.kmatvec_ms;
  out->n = st.n; out->n_padded = st.n_padded;
  // rank 0 fills its own block columns only: their lower tiles
  for (int b = 0; b < st.n_panels; b += g->P) {
    const double W = std::min(st.nb, st.n_padded - b * st.nb), rows = st.n_padded - b * st.nb;
    out->gram_bytes += 8.0 * (W * rows - W * (W - GPAK_TILE) / 2.0);
  }
  for (int k = 0; k < 4; k++) out->accumulated_ms[k] = g->acc_ms[k];
  out->evaluations = g->evaluations;
  if (g->replicas[0]) {
    gpak_phase_times t;
    if (gpak_timing(g->replicas[0], &t) == GPAK_OK) { out->predict_ms = t.predict_ms; }
  }
  for (int r = 0; r < g->P; r++) out->grad_ms = std::max(out->grad_ms, gpak_dist_grad_ms(g->ranks[r]));
  return GPAK_OK;
}
int gpak_multi_n(gpak_multi *g) { return g->P; }
