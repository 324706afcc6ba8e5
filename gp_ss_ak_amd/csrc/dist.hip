// dist.hip -- the multi-GPU hot path in C++ (include/gpak_dist.h): one rank of the block-column-cyclic
// factorisation / solves / nlZ of GP_utils::ldB2_exact, solve_chol and logLikelihood
// (GP_Utils.cpp:841-845, 872-915, 1138-1162; the reference itself is single-process).
//
// Streams of a rank:
//   bulk  (kind 0)  fill, bulk trailing updates, forward substitution riding along, back substitution, nlZ;
//                   with more than one rank it leaves 8 compute units idle (hipExtStreamCreateWithCUMask) so
//                   that the serial panel chain never queues behind the update's workgroups;
//   panel (kind 1)  high priority: factor a block column 128 columns at a time, in-column updates, packs, and
//                   the updates that bring the NEXT two block columns up to date (look-ahead);
//   comm  (kind 2)  every collective of the factorisation, in one global order (b, sub-panel) on all ranks;
//   aux   (kind 3)  the forward substitution of y/sn2 and the explicit inverses of the diagonal blocks, which ride
//                   along panel by panel (off the bulk stream: 11 short launches per panel would sit between two
//                   bulk updates otherwise).
//
// Column c receives panel b <= c-3 in the bulk update of step b (bulk stream), panel c-2 as one K=nb update and
// panel c-1 sub-panel by sub-panel as the broadcasts land (panel stream, in that order): no two streams ever
// touch a block column at once.  Every rank keeps every packed panel (rows from the diagonal block down) and
// the inverted diagonal blocks, so both triangular solves run locally with no communication.
#include <dlfcn.h>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <thread>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

#include "../../include/gpak_dev.h"
#include "../../include/gpak_dist.h"
#include "gpak_internal.h"

// ------------------------------------------------------------------------------------------------
// small vector kernels of the built-in engine
// ------------------------------------------------------------------------------------------------
__global__ void gpak_vec_sum_f64(int n, const double *__restrict__ in, double *__restrict__ out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {   // n is the number of owned block columns (<= a few hundred)
    double s = 0.0;
    for (int i = 0; i < n; i++) s += in[i];
    out[0] = s;
  }
}

extern "C" int gpak_dev_vec_scale(void *stream, int n, const double *in, double s, double *out) {
  gpak_launch_scale((hipStream_t)stream, n, in, s, out);
  return hipGetLastError() == hipSuccess ? GPAK_OK : GPAK_EHIP;
}
extern "C" int gpak_dev_vec_sum(void *stream, int n, const double *in, double *out) {
  hipLaunchKernelGGL(gpak_vec_sum_f64, dim3(1), dim3(64), 0, (hipStream_t)stream, n, in, out);
  return hipGetLastError() == hipSuccess ? GPAK_OK : GPAK_EHIP;
}

// ------------------------------------------------------------------------------------------------
// built-in HIP engine
// ------------------------------------------------------------------------------------------------
namespace {

struct HipEngineState {
  int device = 0;
  int cu_mask_skip = 8;
  bool mask_failed = false;
};

void *he_alloc(void *, size_t bytes) {
  void *p = nullptr;
  return hipMalloc(&p, bytes ? bytes : 8) == hipSuccess ? p : nullptr;
}
void he_release(void *, void *p) { if (p) hipFree(p); }
int he_upload(void *, void *st, void *dst, const void *src, size_t bytes) {
  // pageable host memory: the copy is staged before the call returns, and ordered in the stream
  return hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t)st) == hipSuccess ? GPAK_OK : GPAK_EHIP;
}
int he_download(void *, void *st, void *dst, const void *src, size_t bytes) {
  if (hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)st) != hipSuccess) return GPAK_EHIP;
  return hipStreamSynchronize((hipStream_t)st) == hipSuccess ? GPAK_OK : GPAK_EHIP;
}
int he_zero(void *, void *st, void *dst, size_t bytes) {
  return hipMemsetAsync(dst, 0, bytes, (hipStream_t)st) == hipSuccess ? GPAK_OK : GPAK_EHIP;
}
int he_copy(void *, void *st, void *dst, const void *src, size_t bytes) {
  return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)st) == hipSuccess ? GPAK_OK : GPAK_EHIP;
}
// set once a bulk stream had to be created WITHOUT a CU mask: its trailing updates may then hold two 210-VGPR waves on
// every SIMD of the chip, and the panel stream's 128 x 128 block kernel must be the build that fits beside them
std::atomic<int> g_he_bulk_unmasked{0};
int he_factor_panel(void *stream, double *blk, long ld, int Np, int J, int W, double *inv, int *info) {
  return gpak_dev_factor_panel_co(stream, blk, ld, Np, J, W, inv, info, g_he_bulk_unmasked.load(std::memory_order_relaxed));
}
void *he_stream_create(void *self, int kind) {
  HipEngineState *s = (HipEngineState *)self;
  hipStream_t st = nullptr;
  int lo = 0, hi = 0;
  hipDeviceGetStreamPriorityRange(&lo, &hi);
  if (kind == 0 && s->cu_mask_skip > 0) {
    void *h = nullptr;
    if (gpak_dev_stream_create(s->cu_mask_skip, &h) == GPAK_OK && h) return h;
    (void)hipGetLastError();
    s->mask_failed = true;   // e.g. a partitioned device: an ordinary stream, only slower
  }
  if (hipStreamCreateWithPriority(&st, hipStreamNonBlocking, kind == 0 ? lo : hi) != hipSuccess) return nullptr;
  if (kind == 0) g_he_bulk_unmasked.store(1, std::memory_order_relaxed);
  return st;
}
void he_stream_destroy(void *, void *st) { if (st) hipStreamDestroy((hipStream_t)st); }
void *he_event_create(void *, int timing) {
  hipEvent_t e = nullptr;
  if (hipEventCreateWithFlags(&e, timing ? hipEventDefault : hipEventDisableTiming) != hipSuccess) return nullptr;
  return e;
}
void he_event_destroy(void *, void *e) { if (e) hipEventDestroy((hipEvent_t)e); }
int he_event_record(void *, void *e, void *st) {
  return hipEventRecord((hipEvent_t)e, (hipStream_t)st) == hipSuccess ? GPAK_OK : GPAK_EHIP;
}
int he_stream_wait_event(void *, void *st, void *e) {
  return hipStreamWaitEvent((hipStream_t)st, (hipEvent_t)e, 0) == hipSuccess ? GPAK_OK : GPAK_EHIP;
}
int he_stream_sync(void *, void *st) {
  return hipStreamSynchronize((hipStream_t)st) == hipSuccess ? GPAK_OK : GPAK_EHIP;
}
int he_event_elapsed(void *, void *e0, void *e1, double *ms) {
  float m = 0.f;
  if (hipEventElapsedTime(&m, (hipEvent_t)e0, (hipEvent_t)e1) != hipSuccess) { (void)hipGetLastError(); *ms = 0.0; return GPAK_EHIP; }
  *ms = m;
  return GPAK_OK;
}

void fill_hip_engine(gpak_dist_engine &e, HipEngineState *st) {
  memset(&e, 0, sizeof(e));
  e.self = st;
  e.alloc = he_alloc; e.release = he_release; e.upload = he_upload; e.download = he_download;
  e.zero = he_zero; e.copy = he_copy;
  e.stream_create = he_stream_create; e.stream_destroy = he_stream_destroy;
  e.event_create = he_event_create; e.event_destroy = he_event_destroy; e.event_record = he_event_record;
  e.stream_wait_event = he_stream_wait_event; e.stream_sync = he_stream_sync; e.event_elapsed_ms = he_event_elapsed;
  e.transform = gpak_dev_transform; e.fill_b = gpak_dev_fill_b; e.factor_panel = he_factor_panel;
  e.update_block = gpak_dev_update_block; e.update_cyclic = gpak_dev_update_cyclic;
  e.trsv_fwd_block = gpak_dev_trsv_fwd_block; e.trsv_bwd_packed = gpak_dev_trsv_bwd_packed;
  e.diag_inverse = gpak_dev_diag_inverse; e.logdiag_block = gpak_dev_logdiag_block; e.kmatvec = gpak_dev_kmatvec;
  e.nlz_terms = gpak_dev_nlz_terms; e.pack = gpak_dev_pack; e.vec_scale = gpak_dev_vec_scale;
  e.vec_sum = gpak_dev_vec_sum;
  e.grad_g_rows = gpak_dev_grad_g_rows; e.grad_binv_rows = gpak_dev_grad_binv_rows;
  e.grad_pairs_rows = gpak_dev_grad_pairs_rows;
  e.fill_rect = gpak_dev_fill_rect; e.solve_rows = gpak_dev_solve_rows; e.update_rect = gpak_dev_update_rect;
  e.gemv_n_add = gpak_dev_gemv_n_add; e.gemv_t = gpak_dev_gemv_t; e.vec_axpy = gpak_dev_vec_axpy;
  e.transform_k = gpak_dev_transform_k;
}

// ------------------------------------------------------------------------------------------------
// built-in RCCL transport (librccl resolved at run time: the library is only needed with > 1 rank)
// ------------------------------------------------------------------------------------------------
typedef struct { char internal[128]; } rccl_uid;
struct Rccl {
  void *lib = nullptr;
  int (*GetUniqueId)(rccl_uid *) = nullptr;
  int (*CommInitRank)(void **, int, rccl_uid, int) = nullptr;
  int (*CommInitAll)(void **, int, const int *) = nullptr;
  int (*CommAbort)(void *) = nullptr;
  int (*CommDestroy)(void *) = nullptr;
  int (*Broadcast)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
  int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
  std::string err;
  bool load() {
    if (lib) return true;
    if (getenv("GPAK_RCCL_DISABLE") && atoi(getenv("GPAK_RCCL_DISABLE"))) { err = "librccl disabled (GPAK_RCCL_DISABLE)"; return false; }
    const char *names[] = {getenv("GPAK_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names) {
      if (!n || !*n) continue;
      lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
      if (lib) break;
    }
    if (!lib) { err = std::string("librccl not found: ") + (dlerror() ? dlerror() : ""); return false; }
    GetUniqueId = (int (*)(rccl_uid *))dlsym(lib, "ncclGetUniqueId");
    CommInitRank = (int (*)(void **, int, rccl_uid, int))dlsym(lib, "ncclCommInitRank");
    CommDestroy = (int (*)(void *))dlsym(lib, "ncclCommDestroy");
    CommInitAll = (int (*)(void **, int, const int *))dlsym(lib, "ncclCommInitAll");
    CommAbort = (int (*)(void *))dlsym(lib, "ncclCommAbort");
    Broadcast = (int (*)(const void *, void *, size_t, int, int, void *, hipStream_t))dlsym(lib, "ncclBroadcast");
    AllReduce = (int (*)(const void *, void *, size_t, int, int, void *, hipStream_t))dlsym(lib, "ncclAllReduce");
    GetErrorString = (const char *(*)(int))dlsym(lib, "ncclGetErrorString");
    if (!GetUniqueId || !CommInitRank || !CommDestroy || !Broadcast || !AllReduce) {
      err = "librccl lacks a required symbol";
      return false;
    }
    return true;
  }
};
Rccl g_rccl;
const int kNcclInt = 2, kNcclDouble = 8, kNcclSum = 0, kNcclMin = 3;

struct RcclTransport {
  void *comm = nullptr;
  int world = 1;
};
int rt_bcast(void *self, void *st, double *buf, size_t count, int root) {
  RcclTransport *t = (RcclTransport *)self;
  if (t->world == 1 && !t->comm) return GPAK_OK;   // (a one-rank communicator exists only when a test asked for it)
  if (!t->comm) return GPAK_ESTATE;
  return g_rccl.Broadcast(buf, buf, count, kNcclDouble, root, t->comm, (hipStream_t)st) == 0 ? GPAK_OK : GPAK_EHIP;
}
int rt_allreduce_sum(void *self, void *st, double *buf, size_t count) {
  RcclTransport *t = (RcclTransport *)self;
  if (t->world == 1 && !t->comm) return GPAK_OK;
  if (!t->comm) return GPAK_ESTATE;
  return g_rccl.AllReduce(buf, buf, count, kNcclDouble, kNcclSum, t->comm, (hipStream_t)st) == 0 ? GPAK_OK : GPAK_EHIP;
}
int rt_allreduce_min_int(void *self, void *st, int *buf, size_t count) {
  RcclTransport *t = (RcclTransport *)self;
  if (t->world == 1 && !t->comm) return GPAK_OK;
  if (!t->comm) return GPAK_ESTATE;
  return g_rccl.AllReduce(buf, buf, count, kNcclInt, kNcclMin, t->comm, (hipStream_t)st) == 0 ? GPAK_OK : GPAK_EHIP;
}

double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// one rank
// ------------------------------------------------------------------------------------------------
struct gpak_dist {
  int rank = 0, P = 1, device = 0;
  gpak_dist_engine E;
  gpak_dist_transport T;
  HipEngineState hip_state;
  RcclTransport rccl_state;
  bool builtin_engine = false, builtin_transport = false;
  std::string err;

  // streams
  void *s_bulk = nullptr, *s_panel = nullptr, *s_comm = nullptr, *s_aux = nullptr;
  bool own_comm_stream = false;
  int flags = 0;
  bool checked = false;

  // problem
  int N = 0, Np = 0, nb = 512, nJ = 0, cap = 0;
  long ld = 0;
  std::vector<int> owned;
  double xsum[4] = {0, 0, 0, 0};
  int d = 3;                          // input columns: 3, or 4 with a rock-type column
  double *x_soa = nullptr, *y = nullptr, *u = nullptr, *local = nullptr, *scratch = nullptr, *small = nullptr;
  double *alpha = nullptr, *fwd_x = nullptr, *fwd_z = nullptr, *rhs = nullptr, *f = nullptr, *bwd_scratch = nullptr;
  double *ld_slots = nullptr;
  int *info = nullptr;
  // gradient workspaces (allocated on the first gpak_dist_grad)
  std::vector<double *> slabs;        // [0] own rows of L^-T (rows_a x Np), [1..2] receive buffers (gpak_dist_grad)
  double *binv = nullptr, *gpart = nullptr, *gred = nullptr;
  double grad_ms = 0;
  std::vector<double *> inv_own;      // per owned block: W/128 x 2 x 128 x 128
  std::vector<double *> panels;       // per block column: packed W x (Np - J), every rank keeps all of them
  std::vector<double *> invs;         // per block column: the inverses as received (owner: alias of inv_own)
  std::vector<double *> rinv;         // per block column: explicit (L_bb^-1)^T, 512 x 512
  std::vector<char> rinv_ok;

  // parameters
  bool have_params = false;
  double expans[8] = {0}, bias = 0, sn2 = 0;
  int mode = GPAK_DIST_DIRECT;
  bool hyb = false;                         // a general composition (gpak_dist_set_kernel): `kern` is what the engine gets
  double kern[GPAK_KERN_SERIAL_MAX] = {0};  // serialized: children, kinds, Sigma_White, parameters (gpak_dev.h)
  double white = 0;

  // results
  bool have_result = false;
  double quad = 0, sumlp = 0, logdet = 0, nlz = 0;
  int failed_col = 0;                 // 1-based failing column of the last factorisation, min-reduced: the same on every rank
  gpak_dist_stats stats;

  // event pools
  std::vector<void *> ev_sync, ev_time;
  size_t sync_used = 0, time_used = 0;
  struct Span { size_t e0, e1; int kind; };   // kind 0 bulk, 1 chain, 2 comm
  std::vector<Span> spans;
  bool profile = true;

  int kmode() const { return mode | (d == 4 ? GPAK_DIST_D4 : 0) | (hyb ? GPAK_DIST_HYB : 0); }   // what the engine calls get
  const double *kpars() const { return hyb ? kern : expans; }
  int width(int b) const { return std::min(nb, Np - b * nb); }
  int start(int b) const { return b * nb; }
  int owner(int b) const { return b % P; }
  double *blk(int b) const { return local + (size_t)(b / P) * nb * ld; }

  void *sync_event() {
    if (sync_used == ev_sync.size()) ev_sync.push_back(E.event_create(E.self, 0));
    return ev_sync[sync_used++];
  }
  size_t time_event(void *stream) {
    if (time_used == ev_time.size()) ev_time.push_back(E.event_create(E.self, 1));
    E.event_record(E.self, ev_time[time_used], stream);
    return time_used++;
  }
};

#define DCHK(call)                                                                    \
  do {                                                                                \
    int rc_ = (call);                                                                 \
    if (rc_ != GPAK_OK) {                                                             \
      h->err = std::string(#call) + " failed with status " + std::to_string(rc_);     \
      return rc_ < 0 ? rc_ : GPAK_EHIP;                                               \
    }                                                                                 \
  } while (0)

static void set_device(gpak_dist *h) {
  if (h->builtin_engine) hipSetDevice(h->device);
}

static void release_problem(gpak_dist *h) {
  gpak_dist_engine &E = h->E;
  auto rel = [&](double *&p) { if (p) E.release(E.self, p); p = nullptr; };
  rel(h->x_soa); rel(h->y); rel(h->u); rel(h->local); rel(h->scratch); rel(h->small); rel(h->alpha);
  rel(h->fwd_x); rel(h->fwd_z); rel(h->rhs); rel(h->f); rel(h->bwd_scratch); rel(h->ld_slots);
  if (h->info) E.release(E.self, h->info);
  h->info = nullptr;
  for (size_t b = 0; b < h->panels.size(); b++) {
    if (h->panels[b]) E.release(E.self, h->panels[b]);
    if (h->invs[b] && h->owner((int)b) != h->rank) E.release(E.self, h->invs[b]);
    if (h->rinv[b]) E.release(E.self, h->rinv[b]);
  }
  for (double *p : h->inv_own) if (p) E.release(E.self, p);
  for (double *p : h->slabs) if (p) E.release(E.self, p);
  h->slabs.clear();
  rel(h->binv); rel(h->gpart); rel(h->gred);
  h->panels.clear(); h->invs.clear(); h->rinv.clear(); h->rinv_ok.clear(); h->inv_own.clear();
  h->owned.clear();
  h->N = h->Np = 0;
  h->have_result = false;
}

static int ensure_streams(gpak_dist *h) {
  if (h->s_bulk) return GPAK_OK;
  gpak_dist_engine &E = h->E;
  if (h->builtin_engine) {
    const char *m = getenv("GPAK_DIST_MASK");
    h->hip_state.cu_mask_skip = (h->P > 1) ? (m ? atoi(m) : 8) : 0;   // one rank: the single-GPU numbers say no mask
    if (h->flags & GPAK_DIST_FLAG_CU_MASK_OFF) h->hip_state.cu_mask_skip = 0;
  }
  h->s_bulk = E.stream_create(E.self, 0);
  h->s_panel = E.stream_create(E.self, 1);
  h->s_aux = E.stream_create(E.self, 3);
  if (h->builtin_engine && h->hip_state.mask_failed) h->flags |= GPAK_DIST_FLAG_CU_MASK_OFF;
  if (h->flags & GPAK_DIST_FLAG_COMM_INLINE) {
    h->s_comm = h->s_panel;
    h->own_comm_stream = false;
  } else {
    h->s_comm = E.stream_create(E.self, 2);
    h->own_comm_stream = true;
  }
  if ((!h->s_bulk || !h->s_panel || !h->s_comm || !h->s_aux) && h->builtin_engine) { h->err = "stream creation failed"; return GPAK_EHIP; }
  return GPAK_OK;
}

static void drop_streams(gpak_dist *h) {
  gpak_dist_engine &E = h->E;
  if (h->own_comm_stream && h->s_comm) E.stream_destroy(E.self, h->s_comm);
  if (h->s_panel) E.stream_destroy(E.self, h->s_panel);
  if (h->s_aux) E.stream_destroy(E.self, h->s_aux);
  if (h->s_bulk) E.stream_destroy(E.self, h->s_bulk);
  h->s_bulk = h->s_panel = h->s_comm = h->s_aux = nullptr;
  h->own_comm_stream = false;
}

extern "C" {

int gpak_dist_create(gpak_dist **out, int rank, int world, int device, const gpak_dist_engine *engine,
                     const gpak_dist_transport *transport) {
  if (!out || world < 1 || rank < 0 || rank >= world) return GPAK_EINVAL;
  *out = nullptr;
  gpak_dist *h = new gpak_dist();
  h->rank = rank; h->P = world; h->device = device;
  memset(&h->stats, 0, sizeof(h->stats));
  if (engine) {
    h->E = *engine;
  } else {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count || hipSetDevice(device) != hipSuccess) {
      delete h;
      return GPAK_EHIP;   // no CPU fallback: the built-in engine needs a gfx950 device
    }
    h->hip_state.device = device;
    fill_hip_engine(h->E, &h->hip_state);
    h->builtin_engine = true;
  }
  if (transport) {
    h->T = *transport;
  } else {
    h->rccl_state.world = world;
    h->T.self = &h->rccl_state;
    h->T.bcast = rt_bcast; h->T.allreduce_sum = rt_allreduce_sum; h->T.allreduce_min_int = rt_allreduce_min_int;
    h->builtin_transport = true;
  }
  if (getenv("GPAK_DIST_PROFILE")) h->profile = atoi(getenv("GPAK_DIST_PROFILE")) != 0;
  if (getenv("GPAK_DIST_PLAIN_STREAMS") && atoi(getenv("GPAK_DIST_PLAIN_STREAMS")))
    h->flags |= GPAK_DIST_FLAG_CU_MASK_OFF | GPAK_DIST_FLAG_COMM_INLINE;
  *out = h;
  return GPAK_OK;
}

void gpak_dist_destroy(gpak_dist *h) {
  if (!h) return;
  set_device(h);
  gpak_dist_engine &E = h->E;
  if (h->s_bulk) { E.stream_sync(E.self, h->s_bulk); E.stream_sync(E.self, h->s_panel); E.stream_sync(E.self, h->s_comm); E.stream_sync(E.self, h->s_aux); }
  release_problem(h);
  for (void *e : h->ev_sync) E.event_destroy(E.self, e);
  for (void *e : h->ev_time) E.event_destroy(E.self, e);
  drop_streams(h);
  if (h->builtin_transport && h->rccl_state.comm) g_rccl.CommDestroy(h->rccl_state.comm);
  delete h;
}

const char *gpak_dist_last_error(const gpak_dist *h) { return h ? h->err.c_str() : "null handle"; }

int gpak_dist_rccl_unique_id(char *id) {
  if (!id) return GPAK_EINVAL;
  if (!g_rccl.load()) return GPAK_EHIP;
  rccl_uid u;
  if (g_rccl.GetUniqueId(&u) != 0) return GPAK_EHIP;
  memcpy(id, u.internal, GPAK_DIST_ID_BYTES);
  return GPAK_OK;
}

int gpak_dist_init_rccl(gpak_dist *h, const char *id) {
  if (!h || !id) return GPAK_EINVAL;
  if (!h->builtin_transport) { h->err = "this handle uses a caller-supplied transport"; return GPAK_ESTATE; }
  if (h->rccl_state.comm) return GPAK_OK;
  if (!g_rccl.load()) { h->err = g_rccl.err; return GPAK_EHIP; }
  set_device(h);
  // ncclCommInitRank is a rendezvous of ALL ranks: when one of them never arrives (its start-up failed, its device is
  // not usable) the others would sit in the bootstrap for ever.  The call therefore runs on a helper thread and this
  // one waits a bounded time (GPAK_RCCL_INIT_TIMEOUT_S, default 90 s); on a time-out the helper is abandoned (it
  // holds only the shared state below) and the caller gets an error it can act on -- every host in this repository
  // then switches ALL ranks to another transport, it never retries RCCL on a subset.
  struct Shared { std::mutex m; std::condition_variable cv; bool done = false; int rc = -1; void *comm = nullptr; };
  auto sh = std::make_shared<Shared>();
  rccl_uid u;
  memcpy(u.internal, id, GPAK_DIST_ID_BYTES);
  const int P = h->P, rank = h->rank, dev = h->device;
  const bool own_dev = h->builtin_engine;
  std::thread([sh, u, P, rank, dev, own_dev]() {
    if (own_dev) hipSetDevice(dev);
    void *c = nullptr;
    const int rc = g_rccl.CommInitRank(&c, P, u, rank);
    std::lock_guard<std::mutex> lk(sh->m);
    sh->rc = rc; sh->comm = c; sh->done = true;
    sh->cv.notify_all();
  }).detach();
  double limit = 90.0;
  if (const char *e = getenv("GPAK_RCCL_INIT_TIMEOUT_S")) limit = atof(e) > 0 ? atof(e) : limit;
  std::unique_lock<std::mutex> lk(sh->m);
  if (!sh->cv.wait_for(lk, std::chrono::duration<double>(limit), [&] { return sh->done; })) {
    h->err = "ncclCommInitRank did not return within " + std::to_string((int)limit) + " s (a rank is missing from the rendezvous)";
    return GPAK_EHIP;
  }
  if (sh->rc != 0) {
    h->err = std::string("ncclCommInitRank: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(sh->rc) : "failed");
    return GPAK_EHIP;
  }
  h->rccl_state.comm = sh->comm;
  return GPAK_OK;
}

// One process, all devices (multi.hip): the communicators of ALL ranks are made by ONE call from ONE thread
// (ncclCommInitAll) -- it either yields n communicators or fails as a whole, there is no rendezvous a rank can
// miss.  comms[r] is then adopted by rank r's handle.
int gpak_dist_rccl_init_all(int n, const int *devices, void **comms, std::string &err) {
  if (!g_rccl.load()) { err = g_rccl.err; return GPAK_EHIP; }
  if (!g_rccl.CommInitAll) { err = "librccl lacks ncclCommInitAll"; return GPAK_EHIP; }
  const int rc = g_rccl.CommInitAll(comms, n, devices);
  if (rc != 0) { err = std::string("ncclCommInitAll: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "failed"); return GPAK_EHIP; }
  return GPAK_OK;
}
void gpak_dist_rccl_destroy(void *comm) { if (comm && g_rccl.CommDestroy) g_rccl.CommDestroy(comm); }
int gpak_dist_adopt_rccl(gpak_dist *h, void *comm) {
  if (!h || !comm) return GPAK_EINVAL;
  if (!h->builtin_transport) { h->err = "this handle uses a caller-supplied transport"; return GPAK_ESTATE; }
  h->rccl_state.comm = comm;
  return GPAK_OK;
}
int gpak_dist_failed_column(const gpak_dist *h) { return h ? h->failed_col : 0; }

// A small rehearsal of exactly what the schedule does: a broadcast from every root and an all-reduce on the
// communication stream while the bulk (CU-masked) stream computes.  Wrong data or an error status on ANY rank
// makes ALL ranks fall back to plain streams with the collectives in line on the panel stream.
int gpak_dist_selfcheck(gpak_dist *h, int *flags_out) {
  if (!h) return GPAK_EINVAL;
  set_device(h);
  gpak_dist_engine &E = h->E;
  gpak_dist_transport &T = h->T;
  if (h->builtin_transport && h->P > 1 && !h->rccl_state.comm) { h->err = "gpak_dist_init_rccl has not been called"; return GPAK_ESTATE; }
  int rc = ensure_streams(h);
  if (rc) return rc;
  const int n = 4096;
  double *a = (double *)E.alloc(E.self, sizeof(double) * n), *b = (double *)E.alloc(E.self, sizeof(double) * n);
  int *flag = (int *)E.alloc(E.self, sizeof(int) * 2);
  if (!a || !b || !flag) { h->err = "self-check allocation failed"; return GPAK_ENOMEM; }
  std::vector<double> host(n), back(n);
  int ok = 1;
  for (int attempt = 0; attempt < 2; attempt++) {
    ok = 1;
    // bulk stream: scale a known vector (checks that kernels run on the masked stream at all)
    for (int i = 0; i < n; i++) host[i] = 1.0 + i;
    if (E.upload(E.self, h->s_bulk, a, host.data(), sizeof(double) * n) != GPAK_OK) ok = 0;
    if (ok && E.vec_scale(h->s_bulk, n, a, 2.0, a) != GPAK_OK) ok = 0;
    // communication stream, concurrently: broadcast from every root, then an all-reduce.  Every rank issues the SAME
    // sequence of collectives whatever it has seen so far (a rank that stopped early would hang the others); only a
    // failing CALL ends the sequence, and then every later call on this rank fails the same way.
    bool call_failed = !ok;
    for (int root = 0; !call_failed && root < h->P; root++) {
      for (int i = 0; i < n; i++) host[i] = (h->rank == root) ? 1000.0 * (root + 1) + i : -1.0;
      if (E.upload(E.self, h->s_comm, b, host.data(), sizeof(double) * n) != GPAK_OK ||
          T.bcast(T.self, h->s_comm, b, n, root) != GPAK_OK ||
          E.download(E.self, h->s_comm, back.data(), b, sizeof(double) * n) != GPAK_OK) { call_failed = true; break; }
      for (int i = 0; i < n; i++) if (back[i] != 1000.0 * (root + 1) + i) { ok = 0; break; }
    }
    if (!call_failed) {
      for (int i = 0; i < n; i++) host[i] = (double)(h->rank + 1);
      if (E.upload(E.self, h->s_comm, b, host.data(), sizeof(double) * n) != GPAK_OK ||
          T.allreduce_sum(T.self, h->s_comm, b, n) != GPAK_OK ||
          E.download(E.self, h->s_comm, back.data(), b, sizeof(double) * n) != GPAK_OK) call_failed = true;
      const double want = 0.5 * h->P * (h->P + 1);
      for (int i = 0; !call_failed && i < n; i++) if (back[i] != want) { ok = 0; break; }
    }
    if (call_failed) ok = 0;
    if (ok) {
      if (E.download(E.self, h->s_bulk, back.data(), a, sizeof(double) * n) != GPAK_OK) ok = 0;
      for (int i = 0; ok && i < n; i++) if (back[i] != 2.0 * (1.0 + i)) ok = 0;
    }
    if (h->builtin_engine) (void)hipGetLastError();
    // same decision on every rank
    int v[2] = {ok, ok};
    bool agreed = E.upload(E.self, h->s_comm, flag, v, sizeof(v)) == GPAK_OK &&
                  T.allreduce_min_int(T.self, h->s_comm, flag, 2) == GPAK_OK &&
                  E.download(E.self, h->s_comm, v, flag, sizeof(v)) == GPAK_OK;
    if (agreed) ok = v[0];
    if (ok || attempt == 1 || (h->flags & (GPAK_DIST_FLAG_CU_MASK_OFF | GPAK_DIST_FLAG_COMM_INLINE)) ==
                                  (GPAK_DIST_FLAG_CU_MASK_OFF | GPAK_DIST_FLAG_COMM_INLINE))
      break;
    // fall back: ordinary streams, collectives in line on the panel stream; then check once more
    E.stream_sync(E.self, h->s_bulk); E.stream_sync(E.self, h->s_panel); E.stream_sync(E.self, h->s_comm);
    drop_streams(h);
    h->flags |= GPAK_DIST_FLAG_CU_MASK_OFF | GPAK_DIST_FLAG_COMM_INLINE;
    rc = ensure_streams(h);
    if (rc) break;
  }
  E.release(E.self, a); E.release(E.self, b); E.release(E.self, flag);
  h->checked = true;
  if (flags_out) *flags_out = h->flags;
  if (rc) return rc;
  if (!ok) { h->err = "distributed self-check failed even on plain streams (collective returned wrong data or an error)"; return GPAK_EHIP; }
  return GPAK_OK;
}

int gpak_dist_set_train(gpak_dist *h, const double *X, const double *y, int N, int d, int nb) {
  if (!h || !X || !y || N <= 0) return GPAK_EINVAL;
  if (d != 3 && d != 4) { h->err = "inputs must have 3 or 4 columns"; return GPAK_ENOTIMPL; }
  if (nb == 0) nb = 512;
  if (nb < GPAK_TILE || nb % GPAK_TILE || nb > 512) { h->err = "nb must be 128, 256, 384 or 512"; return GPAK_EINVAL; }
  set_device(h);
  h->d = d;
  if (!h->checked) {
    int rc = gpak_dist_selfcheck(h, nullptr);
    if (rc) return rc;
  }
  gpak_dist_engine &E = h->E;
  E.stream_sync(E.self, h->s_bulk); E.stream_sync(E.self, h->s_panel); E.stream_sync(E.self, h->s_comm);
  release_problem(h);
  h->N = N;
  h->Np = (N + GPAK_TILE - 1) / GPAK_TILE * GPAK_TILE;
  h->ld = h->Np + (h->Np >= 1024 ? 32 : 0);
  h->nb = nb;
  h->nJ = (h->Np + nb - 1) / nb;
  h->cap = h->Np;
  for (int b = 0; b < h->nJ; b++) if (h->owner(b) == h->rank) h->owned.push_back(b);
  const size_t Np = h->Np;
  auto dalloc = [&](size_t n) { return (double *)E.alloc(E.self, sizeof(double) * (n ? n : 1)); };
  h->x_soa = dalloc(4 * Np); h->y = dalloc(Np);   // four raw columns, the 4th zero for d = 3
  h->u = dalloc(5 * GPAK_MAX_TERMS * Np);          // 5 arrays per child of the composition
  h->local = dalloc(std::max<size_t>(1, h->owned.size()) * nb * h->ld);
  h->scratch = dalloc(64 * Np); h->small = dalloc(16); h->alpha = dalloc(Np);
  h->fwd_x = dalloc(Np); h->fwd_z = dalloc(Np); h->rhs = dalloc(Np); h->f = dalloc(Np);
  h->bwd_scratch = dalloc(24 * 512); h->ld_slots = dalloc(h->owned.size() + 1);
  h->info = (int *)E.alloc(E.self, sizeof(int) * 4);
  bool ok = h->x_soa && h->y && h->u && h->local && h->scratch && h->small && h->alpha && h->fwd_x && h->fwd_z &&
            h->rhs && h->f && h->bwd_scratch && h->ld_slots && h->info;
  h->panels.assign(h->nJ, nullptr); h->invs.assign(h->nJ, nullptr); h->rinv.assign(h->nJ, nullptr);
  h->rinv_ok.assign(h->nJ, 0);
  h->inv_own.assign(h->owned.size(), nullptr);
  for (int b = 0; ok && b < h->nJ; b++) {
    const size_t W = h->width(b), rows = Np - h->start(b);
    const size_t inv_n = W / GPAK_TILE * 2 * GPAK_TILE * GPAK_TILE;
    h->panels[b] = dalloc(W * rows);
    h->rinv[b] = dalloc(512 * 512);
    if (h->owner(b) == h->rank) {
      h->inv_own[b / h->P] = dalloc(inv_n);
      h->invs[b] = h->inv_own[b / h->P];
    } else {
      h->invs[b] = dalloc(inv_n);
    }
    ok = h->panels[b] && h->rinv[b] && h->invs[b];
  }
  if (!ok) { h->err = "device allocation failed for the distributed training set"; release_problem(h); return GPAK_ENOMEM; }
  std::vector<double> xs(4 * Np, 0.0), yp(Np, 0.0);
  h->xsum[3] = 0.0;
  for (int k = 0; k < d; k++) {
    double s = 0.0;
    for (int i = 0; i < N; i++) { xs[k * Np + i] = X[i + (size_t)k * N]; s += X[i + (size_t)k * N]; }
    h->xsum[k] = s;
  }
  for (int i = 0; i < N; i++) yp[i] = y[i];
  DCHK(E.upload(E.self, h->s_bulk, h->x_soa, xs.data(), sizeof(double) * 4 * Np));
  DCHK(E.upload(E.self, h->s_bulk, h->y, yp.data(), sizeof(double) * Np));
  DCHK(E.zero(E.self, h->s_bulk, h->alpha, sizeof(double) * Np));
  DCHK(E.zero(E.self, h->s_bulk, h->small, sizeof(double) * 16));
  DCHK(E.stream_sync(E.self, h->s_bulk));
  memset(&h->stats, 0, sizeof(h->stats));
  h->stats.rank = h->rank; h->stats.world = h->P; h->stats.n = N; h->stats.n_padded = h->Np; h->stats.nb = nb;
  h->stats.n_panels = h->nJ;
  return GPAK_OK;
}

int gpak_dist_set_params(gpak_dist *h, const double *expans, double bias, double sn2, int dist_mode) {
  if (!h || !expans) return GPAK_EINVAL;
  if (dist_mode != GPAK_DIST_EXPANSION && dist_mode != GPAK_DIST_DIRECT) { h->err = "bad dist_mode"; return GPAK_EINVAL; }
  memcpy(h->expans, expans, sizeof(double) * 8);
  h->bias = bias; h->sn2 = sn2; h->mode = dist_mode;
  h->hyb = false; h->white = 0.0;
  h->have_params = true;
  h->have_result = false;   // GP_Utils.cpp:132-133: always invalidates
  return GPAK_OK;
}

int gpak_dist_set_kernel(gpak_dist *h, int nterms, const int *kinds, const double *pars, double bias, double white,
                         double sn2, int dist_mode) {
  if (!h || !kinds || !pars || nterms < 1 || nterms > GPAK_MAX_TERMS) return GPAK_EINVAL;
  if (dist_mode != GPAK_DIST_EXPANSION && dist_mode != GPAK_DIST_DIRECT) { h->err = "bad dist_mode"; return GPAK_EINVAL; }
  if (!h->E.transform_k || !h->E.vec_axpy) { h->err = "the engine has no transform_k / vec_axpy entry"; return GPAK_ENOTIMPL; }
  memset(h->kern, 0, sizeof(h->kern));
  h->kern[0] = nterms; h->kern[4] = white;
  int np = 0;
  for (int t = 0; t < nterms; t++) {
    if (kinds[t] != GPAK_KERN_EXPANS && kinds[t] != GPAK_KERN_EXP && kinds[t] != GPAK_KERN_RBF) { h->err = "unknown kernel kind"; return GPAK_EINVAL; }
    h->kern[1 + t] = kinds[t];
    const int k = kinds[t] == GPAK_KERN_EXPANS ? 8 : kinds[t] == GPAK_KERN_EXP ? 2 : 3;
    if (kinds[t] == GPAK_KERN_EXPANS) memcpy(h->expans, pars + np, sizeof(double) * 8);
    np += k;
  }
  if (5 + np > GPAK_KERN_SERIAL_MAX) return GPAK_EINVAL;
  memcpy(h->kern + 5, pars, sizeof(double) * np);
  h->bias = bias; h->white = white; h->sn2 = sn2; h->mode = dist_mode;
  h->hyb = true;
  h->have_params = true;
  h->have_result = false;
  return GPAK_OK;
}

}  // extern "C"

// factor block column b on its owner 128 columns at a time; broadcast every sub-panel (rows from the diagonal block
// down) as soon as it exists; the owner of b+1 applies it to its column on arrival; the inverted diagonal blocks
// follow in one small broadcast.  `done` is recorded on the communication stream behind the last broadcast.
static int produce(gpak_dist *h, int b, void **done) {
  gpak_dist_engine &E = h->E;
  gpak_dist_transport &T = h->T;
  const int J = h->start(b), W = h->width(b), Np = h->Np;
  const long ld = h->ld;
  const int rows = Np - J;
  const bool own = h->rank == h->owner(b);
  const int nxt = b + 1;
  const bool next_owner = nxt < h->nJ && h->rank == h->owner(nxt);
  double *buf = h->panels[b];
  size_t t0 = 0;
  if (own && h->profile) t0 = h->time_event(h->s_panel);
  for (int s = 0; s < W / GPAK_TILE; s++) {
    double *chunk = buf + (size_t)s * GPAK_TILE * rows;
    if (own) {
      double *sub = h->blk(b) + (size_t)s * GPAK_TILE * ld;
      DCHK(E.factor_panel(h->s_panel, sub, ld, Np, J + s * GPAK_TILE, GPAK_TILE,
                          h->invs[b] + (size_t)s * 2 * GPAK_TILE * GPAK_TILE, h->info));
      // the pack runs on the communication stream, in front of the broadcast it feeds: the sub-panel is final once
      // it is solved, and the panel chain goes straight on with the rest of the owner's own block column
      if (h->s_comm != h->s_panel) {
        void *e = h->sync_event();
        DCHK(E.event_record(E.self, e, h->s_panel));
        DCHK(E.stream_wait_event(E.self, h->s_comm, e));
      }
      DCHK(E.pack(h->s_comm, sub, ld, J, rows, GPAK_TILE, chunk));
      const int rem = W - (s + 1) * GPAK_TILE;
      if (rem > 0)   // the rest of the owner's own block column
        DCHK(E.update_block(h->s_panel, sub, ld, 0, GPAK_TILE, h->blk(b) + (size_t)(s + 1) * GPAK_TILE * ld, ld, Np,
                            J + (s + 1) * GPAK_TILE, rem));
    }
    size_t c0 = 0;
    if (h->profile && h->P > 1) c0 = h->time_event(h->s_comm);
    DCHK(T.bcast(T.self, h->s_comm, chunk, (size_t)GPAK_TILE * rows, h->owner(b)));
    if (h->profile && h->P > 1) h->spans.push_back({c0, h->time_event(h->s_comm), 2});
    h->stats.bytes_broadcast += 8.0 * GPAK_TILE * rows;
    if (next_owner) {
      if (h->s_comm != h->s_panel) {
        void *e = h->sync_event();
        DCHK(E.event_record(E.self, e, h->s_comm));
        DCHK(E.stream_wait_event(E.self, h->s_panel, e));
      }
      DCHK(E.update_block(h->s_panel, chunk, rows, J, GPAK_TILE, h->blk(nxt), ld, Np, h->start(nxt), h->width(nxt)));
    }
  }
  if (own && h->profile) h->spans.push_back({t0, h->time_event(h->s_panel), 1});
  const size_t inv_n = (size_t)W / GPAK_TILE * 2 * GPAK_TILE * GPAK_TILE;
  DCHK(T.bcast(T.self, h->s_comm, h->invs[b], inv_n, h->owner(b)));   // the owner's last factor is already ordered
  h->stats.bytes_broadcast += 8.0 * inv_n;                            // in front of it by the last pack's event
  *done = h->sync_event();
  DCHK(E.event_record(E.self, *done, h->s_comm));
  return GPAK_OK;
}

// Every collective of a rank goes through ONE stream (the communication stream) in one global issue order: RCCL
// serialises the operations of a communicator, and two streams feeding the same communicator concurrently is the
// classic way to deadlock it.  `hop_in` orders the communication stream behind the bulk stream, `hop_out` back.
static int hop_in(gpak_dist *h) {
  if (h->s_comm == h->s_bulk) return GPAK_OK;
  void *e = h->sync_event();
  DCHK(h->E.event_record(h->E.self, e, h->s_bulk));
  DCHK(h->E.stream_wait_event(h->E.self, h->s_comm, e));
  return GPAK_OK;
}
static int hop_out(gpak_dist *h) {
  if (h->s_comm == h->s_bulk) return GPAK_OK;
  void *e = h->sync_event();
  DCHK(h->E.event_record(h->E.self, e, h->s_comm));
  DCHK(h->E.stream_wait_event(h->E.self, h->s_bulk, e));
  return GPAK_OK;
}

static int factor(gpak_dist *h, int *failed_col) {
  gpak_dist_engine &E = h->E;
  gpak_dist_transport &T = h->T;
  const int Np = h->Np, nJ = h->nJ, P = h->P;
  const int init = 0x7fffffff;
  DCHK(E.upload(E.self, h->s_bulk, h->info, &init, sizeof(int)));
  std::fill(h->rinv_ok.begin(), h->rinv_ok.end(), 0);
  // the panel and communication streams start behind the fill -- and behind whatever the bulk stream still reads
  // of the previous step's panels (the receive buffers are reused)
  void *e_fill = h->sync_event();
  DCHK(E.event_record(E.self, e_fill, h->s_bulk));
  DCHK(E.stream_wait_event(E.self, h->s_panel, e_fill));
  if (h->s_comm != h->s_panel) DCHK(E.stream_wait_event(E.self, h->s_comm, e_fill));
  DCHK(E.stream_wait_event(E.self, h->s_aux, e_fill));
  void *done = nullptr, *done_next = nullptr, *e_bulk_prev = nullptr;
  int rc = produce(h, 0, &done);
  if (rc) return rc;
  for (int b = 0; b < nJ; b++) {
    const int J = h->start(b), W = h->width(b), rows = Np - J;
    const int nxt = b + 1, nn = b + 2;
    double *panel = h->panels[b];
    DCHK(E.stream_wait_event(E.self, h->s_bulk, done));            // panel b and its inverses are complete here
    if (nxt < nJ) {
      // look-ahead on the panel stream: column b+2 gets panel b as one K=nb update, then column b+1 is factored
      // (it received panel b sub-panel by sub-panel inside produce(b))
      if (h->s_panel != h->s_comm) DCHK(E.stream_wait_event(E.self, h->s_panel, done));
      if (e_bulk_prev) DCHK(E.stream_wait_event(E.self, h->s_panel, e_bulk_prev));   // bulk update b-1 touched column b+2
      if (nn < nJ && h->rank == h->owner(nn))
        DCHK(E.update_block(h->s_panel, panel, rows, J, W, h->blk(nn), h->ld, Np, h->start(nn), h->width(nn)));
      rc = produce(h, nxt, &done_next);
      if (rc) return rc;
      // bulk update of every owned block column beyond b+2, one launch
      int lb0 = -1;
      for (size_t i = 0; i < h->owned.size(); i++) if (h->owned[i] > nn) { lb0 = (int)i; break; }
      if (lb0 >= 0) {
        size_t t0 = 0;
        if (h->profile) t0 = h->time_event(h->s_bulk);
        DCHK(E.update_cyclic(h->s_bulk, panel, rows, J, W, h->local, h->ld, Np, h->nb, P, h->rank, lb0,
                             (int)h->owned.size(), h->width(h->owned.back())));
        if (h->profile) h->spans.push_back({t0, h->time_event(h->s_bulk), 0});
        for (size_t i = lb0; i < h->owned.size(); i++) {   // algorithmic flops: lower tiles of the owned columns
          const int c = h->owned[i];
          const double wt = h->width(c) / GPAK_TILE, mt = (Np - h->start(c)) / GPAK_TILE;
          h->stats.bulk_flops += (wt * mt - wt * (wt - 1) / 2.0) * 2.0 * GPAK_TILE * GPAK_TILE * W;
          h->stats.bulk_bytes += (wt * mt - wt * (wt - 1) / 2.0) * 2.0 * GPAK_TILE * GPAK_TILE * 8.0;   // C read + written once
        }
        h->stats.bulk_launches += 1;
      }
      e_bulk_prev = h->sync_event();
      DCHK(E.event_record(E.self, e_bulk_prev, h->s_bulk));
    }
    // forward substitution L^-1 (y/sn2) rides along on the aux stream, and the explicit inverse of the diagonal
    // block for the back substitution
    DCHK(E.stream_wait_event(E.self, h->s_aux, done));
    DCHK(E.trsv_fwd_block(h->s_aux, panel - J, rows, Np, J, W, h->invs[b], h->fwd_x, h->fwd_z));
    DCHK(E.diag_inverse(h->s_aux, panel, rows, J, J, W, h->invs[b], h->rinv[b]));
    h->rinv_ok[b] = 1;
    if (nxt >= nJ) break;
    done = done_next;
  }
  // everything of the panel stream has been waited for through `done` except trailing look-ahead work of the last
  // step, which does not exist (nxt >= nJ); the failing column travels as an int min-reduce
  void *e_end = h->sync_event();
  DCHK(E.event_record(E.self, e_end, h->s_panel));
  DCHK(E.stream_wait_event(E.self, h->s_bulk, e_end));
  void *e_aux = h->sync_event();
  DCHK(E.event_record(E.self, e_aux, h->s_aux));
  DCHK(E.stream_wait_event(E.self, h->s_bulk, e_aux));
  DCHK(hop_in(h));
  DCHK(T.allreduce_min_int(T.self, h->s_comm, h->info, 1));
  DCHK(hop_out(h));
  int info = init;
  DCHK(E.download(E.self, h->s_bulk, &info, h->info, sizeof(int)));
  *failed_col = info == init ? 0 : info;
  return GPAK_OK;
}

extern "C" {

int gpak_dist_nlz(gpak_dist *h, double *nlz) {
  if (!h || !nlz) return GPAK_EINVAL;
  *nlz = std::numeric_limits<double>::quiet_NaN();
  if (!h->N) { h->err = "no training set (gpak_dist_set_train)"; return GPAK_ESTATE; }
  if (!h->have_params) { h->err = "no parameters (gpak_dist_set_params)"; return GPAK_ESTATE; }
  if (h->have_result) { *nlz = h->nlz; return GPAK_OK; }
  set_device(h);
  gpak_dist_engine &E = h->E;
  gpak_dist_transport &T = h->T;
  const int N = h->N, Np = h->Np, P = h->P;
  const double t_start = now_ms();
  h->sync_used = 0; h->time_used = 0; h->spans.clear();
  h->stats.bytes_broadcast = 0; h->stats.bulk_flops = 0; h->stats.bulk_bytes = 0; h->stats.bulk_launches = 0;
  h->stats.flags = h->flags;
  size_t tp[6] = {0, 0, 0, 0, 0, 0};
  tp[0] = h->time_event(h->s_bulk);
  // ---- fill: HybKerns::computeK + "(sW sW') % K + I" of ldB2_exact, owned columns only, no communication
  {
    const double n = (double)N;
    double mu[4];
    for (int k = 0; k < 4; k++) {   // pooled mean of X u X exactly as Kernel.cpp:1391-1392 computes it
      const double mX1 = n / (n + n) * h->xsum[k] / n;
      mu[k] = n / (n + n) * h->xsum[k] / n + mX1;
    }
    if (h->hyb) DCHK(E.transform_k(h->s_bulk, h->x_soa, Np, N, h->cap, h->kern, h->kmode(), mu, h->u));
    else DCHK(E.transform(h->s_bulk, h->x_soa, Np, N, h->cap, h->expans, mu, h->u));
    for (int b : h->owned)
      DCHK(E.fill_b(h->s_bulk, h->u, h->cap, N, Np, h->start(b), h->width(b), h->kpars(), h->bias, h->sn2, h->kmode(),
                    h->blk(b), h->ld));
  }
  DCHK(E.vec_scale(h->s_bulk, Np, h->y, 1.0 / h->sn2, h->rhs));            // rhs = y / sn2
  DCHK(E.copy(E.self, h->s_bulk, h->fwd_x, h->rhs, sizeof(double) * Np));
  DCHK(E.zero(E.self, h->s_bulk, h->fwd_z, sizeof(double) * Np));
  tp[1] = h->time_event(h->s_bulk);
  int bad = 0;
  int rc = factor(h, &bad);
  if (rc) return rc;
  h->failed_col = bad;
  tp[2] = h->time_event(h->s_bulk);
  if (bad) {
    h->err = "B = I + K/sn2 is not positive definite";
    DCHK(E.stream_sync(E.self, h->s_bulk));
    return GPAK_ENOTPD;   // Chol_fail -> quiet NaN (GP_Utils.cpp:1145-1158)
  }
  // ---- back substitution: every rank holds every packed panel, no collective (solve_chol, GP_Utils.cpp:841-845)
  DCHK(E.zero(E.self, h->s_bulk, h->alpha, sizeof(double) * Np));
  for (int b = h->nJ - 1; b >= 0; b--) {
    const int J = h->start(b), W = h->width(b);
    DCHK(E.trsv_bwd_packed(h->s_bulk, h->panels[b], Np - J, J, Np, J, W, h->invs[b], h->fwd_z, h->bwd_scratch, h->alpha,
                           h->rinv_ok[b] ? h->rinv[b] : nullptr));
  }
  tp[3] = h->time_event(h->s_bulk);
  // ---- f = K alpha: each rank sums over its slice of source points, then one all-reduce (GP_Utils.cpp:1147)
  int per = (N + P - 1) / P;
  per = (per + 1) / 2 * 2;
  const int i0 = std::min(N, h->rank * per), i1 = std::min(N, (h->rank + 1) * per);
  DCHK(E.zero(E.self, h->s_bulk, h->f, sizeof(double) * Np));
  const size_t tk0 = h->time_event(h->s_bulk);
  if (i1 > i0)
    DCHK(E.kmatvec(h->s_bulk, h->u, h->cap, N, i0, i1, h->alpha, h->kpars(), h->bias, h->kmode(), h->scratch, h->f));
  // Kern_White is the diagonal Sigma_White I (Kernel.cpp:256-263): its share of K alpha, once (rank 0's partial sum)
  if (h->hyb && h->white != 0.0 && h->rank == 0) DCHK(E.vec_axpy(h->s_bulk, N, h->white, h->alpha, h->f));
  const size_t tk1 = h->time_event(h->s_bulk);
  DCHK(hop_in(h));
  DCHK(T.allreduce_sum(T.self, h->s_comm, h->f, (size_t)Np));
  DCHK(hop_out(h));
  for (size_t i = 0; i < h->owned.size(); i++) {
    const int b = h->owned[i];
    DCHK(E.logdiag_block(h->s_bulk, h->blk(b), h->ld, h->start(b), h->width(b), N, h->ld_slots + i));
  }
  DCHK(E.zero(E.self, h->s_bulk, h->small, sizeof(double) * 16));
  if (!h->owned.empty()) DCHK(E.vec_sum(h->s_bulk, (int)h->owned.size(), h->ld_slots, h->small + 2));
  DCHK(hop_in(h));
  DCHK(T.allreduce_sum(T.self, h->s_comm, h->small + 2, 1));
  DCHK(hop_out(h));
  DCHK(E.nlz_terms(h->s_bulk, N, h->y, h->f, h->alpha, h->sn2, h->small));
  tp[4] = h->time_event(h->s_bulk);
  double vals[3];
  DCHK(E.download(E.self, h->s_bulk, vals, h->small, sizeof(vals)));
  h->quad = vals[0]; h->sumlp = vals[1]; h->logdet = vals[2];
  h->nlz = h->quad - h->sumlp + h->logdet;   // GP_Utils.cpp:1159
  h->have_result = true;
  *nlz = h->nlz;
  // ---- per-phase split
  gpak_dist_stats &S = h->stats;
  S.step_ms = now_ms() - t_start;
  auto el = [&](size_t a, size_t b) { double ms = 0; E.event_elapsed_ms(E.self, h->ev_time[a], h->ev_time[b], &ms); return ms; };
  S.fill_ms = el(tp[0], tp[1]); S.factor_ms = el(tp[1], tp[2]); S.solve_ms = el(tp[2], tp[3]); S.nlz_ms = el(tp[3], tp[4]);
  S.kmatvec_ms = el(tk0, tk1);
  S.bulk_ms = S.chain_ms = S.comm_ms = 0;
  if (h->profile) {
    DCHK(E.stream_sync(E.self, h->s_panel));
    DCHK(E.stream_sync(E.self, h->s_comm));
    for (const gpak_dist::Span &sp : h->spans) {
      const double ms = el(sp.e0, sp.e1);
      if (sp.kind == 0) S.bulk_ms += ms; else if (sp.kind == 1) S.chain_ms += ms; else S.comm_ms += ms;
    }
  }
  S.wait_ms = S.factor_ms - S.bulk_ms;
  return GPAK_OK;
}

int gpak_dist_nlz_terms(gpak_dist *h, double *quad, double *sumlp, double *logdet) {
  if (!h) return GPAK_EINVAL;
  double v;
  int rc = gpak_dist_nlz(h, &v);
  if (rc) return rc;
  if (quad) *quad = h->quad;
  if (sumlp) *sumlp = h->sumlp;
  if (logdet) *logdet = h->logdet;
  return GPAK_OK;
}

int gpak_dist_get_alpha(gpak_dist *h, double *alpha_host) {
  if (!h || !alpha_host) return GPAK_EINVAL;
  double v;
  int rc = gpak_dist_nlz(h, &v);
  if (rc) return rc;
  set_device(h);
  DCHK(h->E.download(h->E.self, h->s_bulk, alpha_host, h->alpha, sizeof(double) * h->N));
  return GPAK_OK;
}

int gpak_dist_grad(gpak_dist *h, double *g) {
  if (!h || !g) return GPAK_EINVAL;
  double v;
  if (h->hyb) { h->err = "gpak_dist_grad handles the ExpAns(+Bias) composition"; return GPAK_ENOTIMPL; }
  int rc = gpak_dist_nlz(h, &v);   // GradLL re-enters logLikelihood(): GP_Utils.cpp:1173-1174
  if (rc) return rc;
  set_device(h);
  gpak_dist_engine &E = h->E;
  gpak_dist_transport &T = h->T;
  const int Np = h->Np, P = h->P, Tn = Np / GPAK_TILE;
  auto tiles_of = [&](int q) { return Tn > q ? (Tn - q + P - 1) / P : 0; };
  const int Tmax = tiles_of(0), Ta = tiles_of(h->rank);
  if (h->slabs.empty()) {
    // [0] this rank's rows of L^-T, [1] and [2] the two buffers the other ranks' rows pass through
    h->slabs.assign(P > 1 ? 3 : 1, nullptr);
    bool ok = true;
    for (size_t q = 0; q < h->slabs.size() && ok; q++) {
      h->slabs[q] = (double *)E.alloc(E.self, sizeof(double) * std::max<size_t>(1, (size_t)(q ? Tmax : Ta) * GPAK_TILE * Np));
      ok = h->slabs[q] != nullptr;
    }
    h->binv = (double *)E.alloc(E.self, sizeof(double) * std::max<size_t>(1, (size_t)Ta * GPAK_TILE * P * Tmax * GPAK_TILE));
    h->gpart = (double *)E.alloc(E.self, sizeof(double) * std::max<size_t>(1, (size_t)Ta * (Np / 64) * 16));
    h->gred = (double *)E.alloc(E.self, sizeof(double) * 32);
    if (!ok || !h->binv || !h->gpart || !h->gred) { h->err = "device allocation failed for the gradient workspaces"; return GPAK_ENOMEM; }
  }
  h->sync_used = 0; h->time_used = 0;
  const size_t t0 = h->time_event(h->s_bulk);
  double *own = h->slabs[0];
  DCHK(E.grad_g_rows(h->s_bulk, Np, h->nb, P, h->rank, h->panels.data(), h->invs.data(), own));
  // The slabs go round one at a time: rank q's is broadcast into one of two receive buffers while the product against
  // the previous one runs out of the other, so a rank never holds more than its own slab and two in flight.
  void *e_read[2] = {nullptr, nullptr};   // recorded behind the product that last read the receive buffer
  int slot = 0;
  for (int q = 0; q < P; q++) {
    if (tiles_of(q) == 0) continue;
    double *buf = own;
    if (q == h->rank) {
      DCHK(hop_in(h));                     // the slab is this rank's to send once grad_g_rows has written it
    } else {
      buf = h->slabs[1 + slot];
      if (e_read[slot] && h->s_comm != h->s_bulk) DCHK(E.stream_wait_event(E.self, h->s_comm, e_read[slot]));
    }
    DCHK(T.bcast(T.self, h->s_comm, buf, (size_t)tiles_of(q) * GPAK_TILE * Np, q));
    DCHK(hop_out(h));
    DCHK(E.grad_binv_rows(h->s_bulk, Np, P, h->rank, q, own, buf, h->binv));
    if (q != h->rank) {
      e_read[slot] = h->sync_event();
      DCHK(E.event_record(E.self, e_read[slot], h->s_bulk));
      slot ^= 1;
    }
  }
  DCHK(E.grad_pairs_rows(h->s_bulk, h->u, h->cap, h->x_soa, Np, h->N, Np, h->y, h->f, h->alpha, h->binv, P, h->rank,
                         h->expans, h->bias, h->sn2, h->kmode(), h->gpart, h->gred));
  DCHK(hop_in(h));
  DCHK(T.allreduce_sum(T.self, h->s_comm, h->gred, 16));
  DCHK(hop_out(h));
  const size_t t1 = h->time_event(h->s_bulk);
  double red[17];
  DCHK(E.download(E.self, h->s_bulk, red, h->gred, sizeof(red)));
  E.event_elapsed_ms(E.self, h->ev_time[t0], h->ev_time[t1], &h->grad_ms);
  return gpak_dev_grad_finish_d(h->expans, h->bias, h->sn2, h->N, h->d, red, g);
}

int gpak_dist_get_stats(gpak_dist *h, gpak_dist_stats *out) {
  if (!h || !out) return GPAK_EINVAL;
  *out = h->stats;
  out->flags = h->flags;
  return GPAK_OK;
}

}  // extern "C"

// internal: the memory / stream / event entries of the built-in HIP engine alone (none of them uses `self`), for the
// in-process transport of multi.hip
void gpak_dist_hip_services(gpak_dist_engine *e) {
  memset(e, 0, sizeof(*e));
  e->alloc = he_alloc; e->release = he_release; e->upload = he_upload; e->download = he_download;
  e->zero = he_zero; e->copy = he_copy;
  e->stream_destroy = he_stream_destroy;
  e->event_create = he_event_create; e->event_destroy = he_event_destroy; e->event_record = he_event_record;
  e->stream_wait_event = he_stream_wait_event; e->stream_sync = he_stream_sync; e->event_elapsed_ms = he_event_elapsed;
}

// internal (gpak_internal.h): what multi.hip needs to hand this rank's copy of the factor to a single-GPU context
int gpak_dist_factor_view_get(gpak_dist *h, gpak_dist_factor_view *out) {
  if (!h || !out) return GPAK_EINVAL;
  if (!h->have_result) { h->err = "no current factor (gpak_dist_nlz has not succeeded for these parameters)"; return GPAK_ESTATE; }
  out->N = h->N; out->Np = h->Np; out->nb = h->nb; out->nJ = h->nJ;
  out->panels = h->panels.data(); out->invs = h->invs.data();
  out->alpha = h->alpha; out->f = h->f;
  out->quad = h->quad; out->sumlp = h->sumlp; out->logdet = h->logdet; out->nlz = h->nlz;
  return GPAK_OK;
}
double gpak_dist_grad_ms(const gpak_dist *h) { return h ? h->grad_ms : 0.0; }

#include "grid.inc"
