/*
 * gpak_dev.h -- device-pointer level C-ABI of libgpak_hip.so: the per-GPU tile operations of
 * the multi-GPU (block-column-cyclic) factorisation and solves.  The reference is a single
 * process with no distribution at all; these entry points are the pieces of
 * GP_utils::ldB2_exact / solve_chol / logLikelihood (GP_Utils.cpp:841-845, 872-915, 1138-1162)
 * that one rank executes on the block columns it owns.  One process per GPU calls them
 * between torch.distributed (RCCL over xGMI) collectives: tests/py_schedule.py (a test harness; the product schedule is csrc/dist.hip).
 *
 * All pointers are DEVICE pointers owned by the caller (e.g. torch tensors); `stream` is a
 * hipStream_t (NULL = the default stream).  Calls only enqueue work; they do not synchronise.
 * Matrices are column-major doubles.  A rank stores a block column [J, J+W) of the N x N
 * matrix as an (Np x W) array with leading dimension ld whose column 0 is global column J;
 * Np = N rounded up to 128, padded rows/columns form an identity block.
 * Every function returns GPAK_OK or a negative HIP status.
 */
#ifndef GPAK_DEV_H
#define GPAK_DEV_H

#ifdef __cplusplus
extern "C" {
#endif

/* Inputs with a 4th ("rock type") column (Kernel.cpp:1411-1424): the raw points x are ALWAYS four SoA
 * columns of stride xs (the 4th all zero for 3-D inputs: its transformed coordinate is then 0 and every result is
 * bit-identical to the 3-column arithmetic), mu has four entries, and the calls that evaluate the kernel take
 * GPAK_DIST_D4 OR-ed into dist_mode to select the 4-coordinate distance. */
#define GPAK_DIST_D4 0x10
/* A general HybKerns composition (Kernel.cpp:140-154: up to three stationary children + Kern_Bias + Kern_White) travels
 * through the same calls as ONE serialized array in place of `expans`, announced by GPAK_DIST_HYB in dist_mode:
 *   kern[0] = number of children (1..3), kern[1..3] = their kinds (GPAK_KERN_* of gpak.h), kern[4] = Sigma_White,
 *   kern[5 ...] = the children's parameter lists concatenated in the reference's order (ExpAns 8, Exp 2, RBF 3 values).
 * `bias` stays its own argument.  The transformed points then hold 5 arrays PER CHILD (u must have 5 * 3 * cap doubles). */
#define GPAK_DIST_HYB 0x20
#define GPAK_KERN_SERIAL_MAX 32

/* u = (x - mu) * sigInv for the N points, SoA: u is 5*cap doubles {u0[cap],u1[cap],u2[cap],|u|^2[cap],u3[cap]}
 * (u3 = transformed 4th input column of the context-level API; zero here, the distributed path is 3-D).
 * x is SoA with stride xs.  mu[3] is the pooled mean (Kernel.cpp:1391-1392), expans[8] as gpak_set_params. */
int gpak_dev_transform(void *stream, const double *x, int xs, int n, int cap, const double *expans,
                       const double *mu, double *u);

/* Fill the lower tiles of B = I + K/sn2 (GP_Utils.cpp:898-902) for global columns [J, J+W)
 * into blk (column 0 of blk = global column J).  u/cap/n as produced by gpak_dev_transform. */
int gpak_dev_fill_b(void *stream, const double *u, int cap, int n, int Np, int J, int W,
                    const double *expans, double bias, double sn2, int dist_mode, double *blk, long ld);

/* Factor block column [J, J+W) in place: its W x W diagonal block (128 columns at a time) and
 * the panel below.  inv receives (W/128) x 2 x 128 x 128 doubles (inverse and inverse-transpose
 * of each 128 x 128 diagonal block).  *info (device int, pre-set to INT_MAX by the caller)
 * gets the minimum failing global column (1-based) if the block is not positive definite. */
int gpak_dev_factor_panel(void *stream, double *blk, long ld, int Np, int J, int W, double *inv, int *info);
/* The same with a placement hint: coresident != 0 selects the 4-wave / 80-VGPR build of the 128 x 128 block kernel,
 * which fits on a compute unit beside two resident trailing-update workgroups; use it when the bulk update runs on a
 * stream that may occupy every compute unit (no CU mask).  Results are identical. */
int gpak_dev_factor_panel_co(void *stream, double *blk, long ld, int Np, int J, int W, double *inv, int *info,
                             int coresident);

/* Trailing update of an owned block column [Jc, Jc+Wc), Jc > J, with the factored panel of
 * [J, J+W) received from its owner.  `panel` holds rows [J+W.., Np) of that block column packed
 * with leading dimension ldp: panel[(r - prow0) + k*ldp] = L[r, J+k], prow0 = first packed row. */
int gpak_dev_update_block(void *stream, const double *panel, long ldp, int prow0, int W, double *blk, long ld,
                          int Np, int Jc, int Wc);

/* The same update for ALL block columns the rank owns from local block lb0 on, in ONE launch.
 * `local` is the rank's storage: its n_local_blocks block columns side by side (each nb columns,
 * leading dimension ld, the last one last_width wide); local block lb is global block lb*P+rank. */
int gpak_dev_update_cyclic(void *stream, const double *panel, long ldp, int prow0, int W, double *local, long ld,
                           int Np, int nb, int P, int rank, int lb0, int n_local_blocks, int last_width);

/* Forward substitution step for block column [J, J+W):  out[J..J+W) = L_bb^-1 x[J..J+W) and
 * x[r] -= L[r, J..J+W) out  for every r >= J+W.  x, out have Np entries. */
int gpak_dev_trsv_fwd_block(void *stream, const double *blk, long ld, int Np, int J, int W, const double *inv,
                            double *x, double *out);
/* s[0..W) = sum_{r >= J+W} L[r, J+c] x[r]   (the owner-local part of the back substitution) */
int gpak_dev_coldot(void *stream, const double *blk, long ld, int Np, int J, int W, const double *x, double *s);
/* out[J..J+W) = L_bb^-T x[J..J+W)  (x[J..J+W) is overwritten with intermediates) */
int gpak_dev_trsv_bwd_block(void *stream, const double *blk, long ld, int J, int W, const double *inv, double *x,
                            double *out);
/* Back substitution step for block column [J, J+W) held as a PACKED panel (element 0 of each column = global row
 * row0, leading dimension ldp):  out[J..J+W) = L_bb^-T ( z[J..J+W) - sum_{r >= J+W} L[r, J..J+W) out[r] ).
 * z is read-only; scratch holds 24*512 doubles; W <= 512.  rinv: the explicit inverse of the diagonal block from
 * gpak_dev_diag_inverse (then: streamed column dots + one matrix-vector product), or NULL (then one workgroup
 * solves the diagonal block in four dependent 128-column phases). */
int gpak_dev_trsv_bwd_packed(void *stream, const double *panel, long ldp, int row0, int Np, int J, int W,
                             const double *inv, const double *z, double *scratch, double *out, const double *rinv);
/* rinv (512 x 512 doubles, leading dimension 512) <- (L_bb^-1)^T of the W x W diagonal block of a packed panel
 * (row0 = global row of element 0 of each column).  Seven small launches; meant to be queued where the stream
 * has slack (right after the panel arrives). */
int gpak_dev_diag_inverse(void *stream, const double *panel, long ldp, int row0, int J, int W, const double *inv,
                          double *rinv);
/* out[0] = sum of log L[c,c] over the valid (c < N) columns of the block column */
int gpak_dev_logdiag_block(void *stream, const double *blk, long ld, int J, int W, int N, double *out);

/* out[j] = sum_{i in [i0,i1)} w[i] k(x_i, x_j), j < n: one rank's share of f = K*alpha (GP_Utils.cpp:1147).
 * scratch must hold 64*cap doubles. */
int gpak_dev_kmatvec(void *stream, const double *u, int cap, int n, int i0, int i1, const double *w,
                     const double *expans, double bias, int dist_mode, double *scratch, double *out);
/* out[0] = Alpha'(0.5 f), out[1] = accu(lp)   (GP_Utils.cpp:810, 1159) */
int gpak_dev_nlz_terms(void *stream, int N, const double *y, const double *f, const double *alpha, double sn2,
                       double *out);

/* dst (nrows x ncols doubles, packed column-major) <- rows [row0, row0+nrows) of ncols columns of src (leading
 * dimension ld): the panel pack in front of a broadcast. */
int gpak_dev_pack(void *stream, const double *src, long ld, int row0, int nrows, int ncols, double *dst);

/* ---- the reference-style gradient (GP_utils::GradLL, GP_Utils.cpp:1171-1262) distributed over P ranks that all hold
 * the factor as packed panels.  128-row block g of G = L^-T and of B^-1 belongs to rank g % P.
 * panels / invs are HOST arrays of device pointers (one per block column).
 * gpak_dev_grad_g_rows: slab (rows_a x Np doubles, leading dimension rows_a = 128 * number of owned row blocks) <- the
 *   owned rows of L^-T by blocked forward substitution on identity rows (N^3/(3P) flops).
 * gpak_dev_grad_binv_rows: binv (rows_a x P*Tmax*128, Tmax = ceil(Np/128 / P)) <- the column groups of source rank b
 *   of the owned rows of B^-1 = G G^T (lower part), from this rank's slab and rank b's (the same pointer when b == a);
 *   the 128-column group of global block g sits at group index (g % P) * Tmax + g / P.  One call per source rank, so a
 *   caller holds its own slab and the one passing through, never all P.
 * gpak_dev_grad_pairs_rows: out[0..16) <- this rank's share of the pair sums (to be all-reduced), out[16] <- the
 *   replicated lp_dhyp sum; part = scratch of rows_a/128 * Np/64 * 16 doubles.
 * gpak_dev_grad_finish (host only): g[10] = {8 ExpAns, bias, sn2} from the all-reduced sums. */
int gpak_dev_grad_g_rows(void *stream, int Np, int nb, int P, int a, const double *const *panels,
                         const double *const *invs, double *slab);
int gpak_dev_grad_binv_rows(void *stream, int Np, int P, int a, int b, const double *slab_a, const double *slab_b,
                            double *binv);
int gpak_dev_grad_pairs_rows(void *stream, const double *u, int cap, const double *x_soa, int xs, int n, int Np,
                             const double *y, const double *f, const double *alpha, const double *binv, int P, int a,
                             const double *expans, double bias, double sn2, int dist_mode, double *part, double *out);
int gpak_dev_grad_finish(const double *expans, double bias, double sn2, int n, const double *red, double *g);
/* the same for d input columns (3 or 4): with a rock-type column g[7] is the InversewidthR slot (Kernel.cpp:1246-1255) */
int gpak_dev_grad_finish_d(const double *expans, double bias, double sn2, int n, int d, const double *red, double *g);
/* host only: the constants of the pair pass, M36 = M_p (6 x {00,01,02,11,12,22}), m2_18 = 2 * column sums of M_p */
int gpak_dev_grad_consts(const double *expans, double *M36, double *m2_18);

/* ---- pieces of the row-block x column-block layout (gpak_grid_*, include/gpak_dist.h): rectangular parts of a rank's
 * LOCAL storage; no global row addressing.  Semantics: the engine entries of the same names in gpak_dist.h.
 * gpak_dev_fill_rect: ExpAns + Bias piece of B = I + K/sn2 (the fill of HybKerns::computeK + ldB2_exact's scaling,
 *   Kernel.cpp:856-882, 362-367, GP_Utils.cpp:898-902); nrows a multiple of 128, ncols of 64.
 * gpak_dev_solve_rows: P := P Lbb^-T in 128-column steps (product with the inverted diagonal block, K = 128 update of
 *   the columns to the right), W <= 512.
 * gpak_dev_update_rect: the trailing update C -= A B^T of one block column of local blocks (MFMA GEMM).
 * gpak_dev_gemv_n_add / gpak_dev_gemv_t: the two matrix-vector products of the distributed triangular solves. */
int gpak_dev_fill_rect(void *stream, const double *u, int cap, int n, int row0, int nrows, int col0, int ncols,
                       const double *expans, double bias, double sn2, int dist_mode, double *dst, long ld);
int gpak_dev_solve_rows(void *stream, double *P, long ld, int nrows, int W, const double *Lbb, long ldl,
                        const double *inv);
int gpak_dev_update_rect(void *stream, const double *A, long lda, const double *B, long ldb, int K, double *C, long ldc,
                         int mrows, int ncols, int diag_first);
int gpak_dev_gemv_n_add(void *stream, const double *A, long ld, int nrows, int W, const double *x, double *y);
int gpak_dev_gemv_t(void *stream, const double *A, long ld, int nrows, int W, const double *x, double *y);
int gpak_dev_vec_axpy(void *stream, int n, double a, const double *x, double *y);
/* gpak_dev_transform for a serialized composition (dist_mode carries GPAK_DIST_HYB [| GPAK_DIST_D4]); without
 * GPAK_DIST_HYB `kern` is the plain 8-value ExpAns list and the call equals gpak_dev_transform */
int gpak_dev_transform_k(void *stream, const double *x, int xs, int n, int cap, const double *kern, int dist_mode,
                         const double *mu, double *u);

/* A HIP stream that may not use the first skip_cus compute units (hipExtStreamCreateWithCUMask): the bulk
 * updates of a rank run there, so that the serial panel chain (potrf128, the small panel products) always finds
 * idle CUs beside them.  skip_cus = 0 gives an ordinary non-blocking stream.  Wrap it for torch with
 * torch.cuda.ExternalStream.  *stream_out is a hipStream_t. */
int gpak_dev_stream_create(int skip_cus, void **stream_out);
int gpak_dev_stream_destroy(void *stream);

#ifdef __cplusplus
}
#endif
#endif
