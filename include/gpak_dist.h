/*
 * gpak_dist.h -- C-ABI of the multi-GPU hot path of libgpak_hip.so: the block-column-cyclic Cholesky /
 * solves / log-marginal-likelihood schedule, in C++ (csrc/dist.hip), one RANK per GPU.
 *
 * The reference is one process on one CPU; this is the build's own distribution of
 * GP_utils::ldB2_exact / solve_chol / logLikelihood (GP_Utils.cpp:841-845, 872-915, 1138-1162):
 *   - outer block column b (width nb) of B = I + K/sn2 lives on rank b % P only;
 *   - fill: every rank fills its own block columns from the replicated coordinates (no communication);
 *   - factor: the owner factors block column b 128 columns at a time and BROADCASTS each sub-panel as soon
 *     as it exists (RCCL ncclBroadcast over xGMI on a dedicated communication stream -- the only bulk
 *     collective); every rank keeps the packed panels, updates its own block columns, and runs both
 *     triangular solves locally;
 *   - f = K alpha: an N-entry all-reduce; log det: a scalar all-reduce; failing column: an int min-reduce.
 *
 * A rank is driven either by its own process (bench.py under torch.distributed.run: gpak_dist_create +
 * gpak_dist_init_rccl with a unique id made on rank 0 and passed through any host-side channel), or by a
 * thread of ONE process that drives all GPUs (gpak_create_multi in gpak.h, used by `gp_ss_ak --gpus n`).
 *
 * The schedule is engine- and transport-agnostic: NULL selects the built-in HIP engine (the gpak_dev_*
 * tile operations of gpak_dev.h) and the built-in RCCL transport; tests pass callback tables instead (a
 * NumPy engine with a gloo transport on CPU boxes; the HIP engine with a host-staged transport to rehearse
 * several ranks on one GPU, which RCCL refuses).
 *
 * Conventions: as gpak.h (status codes, column-major doubles, caller-owned host buffers, synchronous on
 * return).  Every rank calls the same functions in the same order with the same arguments.
 */
#ifndef GPAK_DIST_H
#define GPAK_DIST_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gpak_dist gpak_dist;
struct gpak_ctx;   /* gpak.h */

/* Tile engine: memory / stream / event services + the tile operations of gpak_dev.h (same argument lists).
 * Device buffers are opaque addresses; `stream` / `event` are opaque handles of the engine. */
typedef struct gpak_dist_engine {
  void *self;
  void *(*alloc)(void *self, size_t bytes);
  void (*release)(void *self, void *p);
  int (*upload)(void *self, void *stream, void *dst, const void *host_src, size_t bytes);   /* ordered in `stream` */
  int (*download)(void *self, void *stream, void *host_dst, const void *src, size_t bytes); /* complete on return  */
  int (*zero)(void *self, void *stream, void *dst, size_t bytes);
  int (*copy)(void *self, void *stream, void *dst, const void *src, size_t bytes);          /* device to device     */
  /* kind 0: bulk stream (may leave a few compute units idle), 1: panel chain (high priority), 2: communication,
   * 3: auxiliary (forward substitution and diagonal-block inverses riding along) */
  void *(*stream_create)(void *self, int kind);
  void (*stream_destroy)(void *self, void *stream);
  void *(*event_create)(void *self, int timing);
  void (*event_destroy)(void *self, void *event);
  int (*event_record)(void *self, void *event, void *stream);
  int (*stream_wait_event)(void *self, void *stream, void *event);
  int (*stream_sync)(void *self, void *stream);
  int (*event_elapsed_ms)(void *self, void *e0, void *e1, double *ms);
  /* gpak_dev.h */
  int (*transform)(void *stream, const double *x, int xs, int n, int cap, const double *expans, const double *mu,
                   double *u);
  int (*fill_b)(void *stream, const double *u, int cap, int n, int Np, int J, int W, const double *expans, double bias,
                double sn2, int dist_mode, double *blk, long ld);
  int (*factor_panel)(void *stream, double *blk, long ld, int Np, int J, int W, double *inv, int *info);
  int (*update_block)(void *stream, const double *panel, long ldp, int prow0, int W, double *blk, long ld, int Np,
                      int Jc, int Wc);
  int (*update_cyclic)(void *stream, const double *panel, long ldp, int prow0, int W, double *local, long ld, int Np,
                       int nb, int P, int rank, int lb0, int n_local_blocks, int last_width);
  int (*trsv_fwd_block)(void *stream, const double *blk, long ld, int Np, int J, int W, const double *inv, double *x,
                        double *out);
  int (*trsv_bwd_packed)(void *stream, const double *panel, long ldp, int row0, int Np, int J, int W,
                         const double *inv, const double *z, double *scratch, double *out, const double *rinv);
  int (*diag_inverse)(void *stream, const double *panel, long ldp, int row0, int J, int W, const double *inv,
                      double *rinv);
  int (*logdiag_block)(void *stream, const double *blk, long ld, int J, int W, int N, double *out);
  int (*kmatvec)(void *stream, const double *u, int cap, int n, int i0, int i1, const double *w, const double *expans,
                 double bias, int dist_mode, double *scratch, double *out);
  int (*nlz_terms)(void *stream, int N, const double *y, const double *f, const double *alpha, double sn2, double *out);
  int (*pack)(void *stream, const double *src, long ld, int row0, int nrows, int ncols, double *dst);
  int (*vec_scale)(void *stream, int n, const double *in, double s, double *out);
  int (*vec_sum)(void *stream, int n, const double *in, double *out);   /* out[0] = in[0] + ... + in[n-1], fixed order */
  /* the distributed gradient (gpak_dev.h) */
  int (*grad_g_rows)(void *stream, int Np, int nb, int P, int a, const double *const *panels, const double *const *invs,
                     double *slab);
  int (*grad_binv_rows)(void *stream, int Np, int P, int a, int b, const double *slab_a, const double *slab_b, double *binv);
  int (*grad_pairs_rows)(void *stream, const double *u, int cap, const double *x_soa, int xs, int n, int Np,
                         const double *y, const double *f, const double *alpha, const double *binv, int P, int a,
                         const double *expans, double bias, double sn2, int dist_mode, double *part, double *out);
  /* ---- the row-block x column-block layout (gpak_grid_*): rectangular pieces of a rank's LOCAL storage ---- */
  /* dst[a + c*ld] = (K(row0 + a, col0 + c) / sn2 + [row0 + a == col0 + c]) for a < nrows, c < ncols (global indices into
   * the transformed points u; rows / columns beyond n are zero with a unit diagonal); a piece that holds part of the
   * diagonal (row0 == col0) is filled in its lower 128-tiles only */
  int (*fill_rect)(void *stream, const double *u, int cap, int n, int row0, int nrows, int col0, int ncols,
                   const double *expans, double bias, double sn2, int dist_mode, double *dst, long ld);
  /* P (nrows x W, ld) := P * Lbb^-T, Lbb the W x W lower factor (ldl) with its inverted 128-blocks `inv` */
  int (*solve_rows)(void *stream, double *P, long ld, int nrows, int W, const double *Lbb, long ldl, const double *inv);
  /* C (mrows x ncols, ldc) -= A (mrows x K, lda) * B (ncols x K, ldb)^T; diag_first: the leading ncols x ncols square
   * of C lies on the matrix diagonal -- only its lower 128-tiles are touched */
  int (*update_rect)(void *stream, const double *A, long lda, const double *B, long ldb, int K, double *C, long ldc,
                     int mrows, int ncols, int diag_first);
  int (*gemv_n_add)(void *stream, const double *A, long ld, int nrows, int W, const double *x, double *y); /* y += A x   */
  int (*gemv_t)(void *stream, const double *A, long ld, int nrows, int W, const double *x, double *y);     /* y  = A^T x */
  int (*vec_axpy)(void *stream, int n, double a, const double *x, double *y);                              /* y += a x   */
  /* transform for a serialized HybKerns composition (gpak_dev.h GPAK_DIST_HYB): `kern` in place of expans, the flags in
   * dist_mode; fill_b / kmatvec / fill_rect receive the same array and flags through their expans / dist_mode arguments */
  int (*transform_k)(void *stream, const double *x, int xs, int n, int cap, const double *kern, int dist_mode,
                     const double *mu, double *u);
} gpak_dist_engine;

/* Collectives on device buffers, enqueued on `stream` (an engine stream) in call order; every rank calls them in
 * the same order. */
typedef struct gpak_dist_transport {
  void *self;
  int (*bcast)(void *self, void *stream, double *buf, size_t count, int root);
  int (*allreduce_sum)(void *self, void *stream, double *buf, size_t count);
  int (*allreduce_min_int)(void *self, void *stream, int *buf, size_t count);
  /* Sub-groups of a Pr x Pc process grid, rank = pr + Pr * pc.  group 1: my process ROW (the Pc ranks with my pr;
   * member index = pc), group 2: my process COLUMN (the Pr ranks with my pc; member index = pr); root is a member
   * index.  NULL entries: the transport serves 1-D layouts only (gpak_grid_create refuses it for Pr, Pc > 1). */
  int (*grid_setup)(void *self, int Pr, int Pc);
  int (*bcast_group)(void *self, void *stream, double *buf, size_t count, int root, int group);
  int (*allreduce_sum_group)(void *self, void *stream, double *buf, size_t count, int group);
} gpak_dist_transport;
#define GPAK_GROUP_ROW 1
#define GPAK_GROUP_COL 2

/* rank / world: this rank and the number of ranks; device: HIP ordinal of the built-in engine (ignored with a
 * callback engine).  engine / transport: NULL = built-in HIP engine / built-in RCCL transport (the callback
 * tables are copied).  With the RCCL transport and world > 1, gpak_dist_init_rccl must follow. */
int gpak_dist_create(gpak_dist **out, int rank, int world, int device, const gpak_dist_engine *engine,
                     const gpak_dist_transport *transport);
void gpak_dist_destroy(gpak_dist *h);
const char *gpak_dist_last_error(const gpak_dist *h);

/* RCCL bootstrap: rank 0 makes the 128-byte unique id (ncclGetUniqueId), the host program hands it to every
 * rank (torch.distributed store, a file, MPI ...), then every rank calls gpak_dist_init_rccl (ncclCommInitRank). */
#define GPAK_DIST_ID_BYTES 128
int gpak_dist_rccl_unique_id(char *id);
int gpak_dist_init_rccl(gpak_dist *h, const char *id);

/* Start-up self-check of what the schedule relies on: the CU-masked bulk stream and collectives issued on a side
 * stream while another stream computes.  A failed check switches THIS handle to plain streams (flags report it);
 * the result of the collective test is min-reduced so that all ranks take the same decision.  Called by
 * gpak_dist_set_train when it has not been called yet. */
#define GPAK_DIST_FLAG_CU_MASK_OFF   1   /* bulk stream is an ordinary stream                      */
#define GPAK_DIST_FLAG_COMM_INLINE   2   /* collectives are issued on the panel stream, not their own */
int gpak_dist_selfcheck(gpak_dist *h, int *flags_out);

/* GP_utils ctor: replicated Xinp (N x 3 col-major) and yTarg (N); nb = outer block width (multiple of 128;
 * 0 = 512).  3-D inputs only (the distributed path does not carry the rock-type column). */
int gpak_dist_set_train(gpak_dist *h, const double *X, const double *y, int N, int d, int nb);
/* GP_utils::set_GP_Pars (GP_Utils.cpp:130-157), replicated: as gpak_set_params */
int gpak_dist_set_params(gpak_dist *h, const double *expans, double bias, double sn2, int dist_mode);
/* A general HybKerns composition, replicated: as gpak_set_kernel of gpak.h (kinds / concatenated parameters / Kern_Bias /
 * Kern_White).  logLikelihood and alpha are distributed as for ExpAns(+Bias); gpak_dist_grad is then GPAK_ENOTIMPL (a
 * gpak_create_multi group forms the children's gradients on device 0 from the distributed factor). */
int gpak_dist_set_kernel(gpak_dist *h, int nterms, const int *kinds, const double *pars, double bias, double white,
                         double sn2, int dist_mode);
/* GP_utils::logLikelihood (GP_Utils.cpp:1138-1162): fill + factor + solves + f = K alpha + reductions.
 * Same value on every rank; quiet NaN with GPAK_ENOTPD on Chol_fail. */
int gpak_dist_nlz(gpak_dist *h, double *nlz);
int gpak_dist_nlz_terms(gpak_dist *h, double *quad, double *sumlp, double *logdet);
/* GP_utils::GradLL (GP_Utils.cpp:1171-1262) with the children's getGradients as written (Kernel.cpp:886-1263, 370-377),
 * g[10] = {8 ExpAns, bias, sn2}, same value on every rank.  Distributed by 128-row blocks of L^-T and B^-1 (block g on
 * rank g % P): each rank forms its rows of L^-T from the packed panels it already holds (N^3/(3P) flops, no
 * communication), the row slabs go round one at a time (P broadcasts, N^2 doubles in total, each received into one of
 * two buffers while the product against the previous one runs), each rank forms its rows of B^-1 = L^-T L^-1
 * (N^3/(3P)) and runs the fused pair pass on them; one 16-double all-reduce.  Per-rank workspace: its own slab, two
 * receive buffers and its rows of B^-1, 4 N^2/P doubles in all. */
int gpak_dist_grad(gpak_dist *h, double *g);
/* failing column (1-based, the same on every rank) of the last GPAK_ENOTPD, 0 if none: GP_utils::Chol_fail */
int gpak_dist_failed_column(const gpak_dist *h);
/* alpha = (K + sn2 I)^-1 y of the last gpak_dist_nlz (replicated), N doubles */
int gpak_dist_get_alpha(gpak_dist *h, double *alpha_host);

typedef struct {
  int rank, world, n, n_padded, nb, n_panels, flags;
  double bytes_broadcast;   /* payload of all panel / inverse broadcasts of the last step (per rank view)      */
  double step_ms;           /* whole gpak_dist_nlz, host wall clock                                             */
  double fill_ms;           /* transform + fill of the owned block columns                                      */
  double factor_ms;         /* first panel factor ... last panel complete, on the bulk stream                   */
  double solve_ms;          /* back substitution (the forward one rides along with the factorisation)           */
  double nlz_ms;            /* f = K alpha slice + all-reduces + reductions                                     */
  double bulk_ms;           /* sum of this rank's bulk trailing-update launches (MFMA work)                     */
  double bulk_flops;        /* their algorithmic flops                                                          */
  double chain_ms;          /* sum of this rank's panel-chain segments (factor + in-column updates + packs)     */
  double comm_ms;           /* sum of the broadcast calls' durations on the communication stream (incl. waiting
                               for the root)                                                                    */
  double wait_ms;           /* factor_ms - bulk_ms: what the bulk stream spent not updating (chain / comm wait) */
  double kmatvec_ms;        /* this rank's slice of f = K alpha inside nlz_ms                                   */
  double bulk_bytes;        /* algorithmic HBM bytes of the bulk launches: every owned lower C tile once each way */
  double bulk_launches;     /* how many bulk launches that was                                                  */
} gpak_dist_stats;
int gpak_dist_get_stats(gpak_dist *h, gpak_dist_stats *out);
/* the same for rank `rank` of a multi-GPU context made by gpak_create_multi (gpak.h); GPAK_EINVAL for a one-GPU context */
int gpak_group_rank_stats(struct gpak_ctx *ctx, int rank, gpak_dist_stats *out);

/* ---- row-block x column-block layout (north_star's 2-D sharding): csrc/grid.inc -------------------------------------
 * B = I + K/sn2 in nb x nb blocks, block (i, j), i >= j, on rank (i % Pr) + Pr * (j % Pc) of a Pr x Pc grid.
 * Step b: the owner of the diagonal block factors it (LDS-resident 128-blocks) and broadcasts it with its inverted
 * 128-blocks DOWN its process column; the ranks of that column solve their rows of panel b and broadcast them ALONG
 * their process rows; the row pieces P_j that a rank needs transposed (its own block columns j) come from rank
 * (j % Pr) of its process column: one packed broadcast per root.  A rank therefore receives N^2/2 * 8 B * (1/Pr + 1/Pc)
 * per factorisation instead of the 1-D layout's (P-1)/P * N^2/2 * 8 B, and keeps only its own blocks: N^2 / (2 P)
 * doubles (+ two panel buffers).  The triangular solves are distributed (per block: a group all-reduce of the partial
 * sums, the owner's 512-column solve, a world broadcast of the piece).  logLikelihood / alpha only: the gradient and
 * prediction need a whole factor per rank and stay on the 1-D layout (gpak_dist_*).  Same engine / transport tables;
 * world = Pr * Pc; Pr = 1 is refused (that IS the 1-D layout: use gpak_dist_create). */
typedef struct gpak_grid gpak_grid;
int gpak_grid_create(gpak_grid **out, int rank, int world, int Pr, int Pc, int device, const gpak_dist_engine *engine,
                     const gpak_dist_transport *transport);
void gpak_grid_destroy(gpak_grid *h);
const char *gpak_grid_last_error(const gpak_grid *h);
/* RCCL: world communicator from the unique id, then ncclCommSplit into the row and the column communicator */
int gpak_grid_init_rccl(gpak_grid *h, const char *id);
int gpak_grid_set_train(gpak_grid *h, const double *X, const double *y, int N, int d, int nb);
int gpak_grid_set_params(gpak_grid *h, const double *expans, double bias, double sn2, int dist_mode);
int gpak_grid_nlz(gpak_grid *h, double *nlz);
int gpak_grid_nlz_terms(gpak_grid *h, double *quad, double *sumlp, double *logdet);
int gpak_grid_get_alpha(gpak_grid *h, double *alpha_host);
int gpak_grid_get_stats(gpak_grid *h, gpak_dist_stats *out);   /* bytes_broadcast = what THIS rank received */

/* TEST entry point: a gpak_create_multi group (one host thread per rank, in-process transport, the gpak_ctx surface of
 * logLikelihood / alpha / gradient; no prediction: there are no device replicas) over n_ranks CALLER-SUPPLIED engines.
 * Lets the thread-per-GPU host logic of csrc/multi.hip run on a box without a GPU, e.g. under ThreadSanitizer
 * (tests/test_sanitizers.py with the NumPy engine). */
int gpak_create_multi_with_engines(struct gpak_ctx **out, int n_ranks, const gpak_dist_engine *const *engines);

/* vector helpers of the built-in engine (also in the engine table) */
int gpak_dev_vec_scale(void *stream, int n, const double *in, double s, double *out);
int gpak_dev_vec_sum(void *stream, int n, const double *in, double *out);

#ifdef __cplusplus
}
#endif
#endif /* GPAK_DIST_H */
