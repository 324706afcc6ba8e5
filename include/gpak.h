/*
 * gpak.h -- C-ABI of libgpak_hip.so: the MI355X (gfx950) implementation of the
 * GP_SS_AK hot path (ExpAns+Bias Gram build, Cholesky factor/solve,
 * log-marginal-likelihood, predictive mean/variance).
 *
 * The reference has no FFI layer; its seam is the C++ member interface of
 * `Kernels` / `GP_utils`.  Each entry point below names the
 * reference member(s) whose body it replaces.  INTEGRATION.md shows the
 * binding a maintainer of the reference would add (arma::mat::memptr() in,
 * memptr() out).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; no C++/torch types.
 *   - host matrices are column-major doubles (arma::mat layout), caller-owned,
 *     never retained after the call returns.
 *   - every function returns an int status: 0 = GPAK_OK,
 *     GPAK_ENOTPD = the reference's Chol_fail (GP_Utils.cpp:881-888): callers
 *     map it to quiet NaN exactly like GP_Utils.cpp:1145-1158;
 *     negative = HIP runtime failure (text via gpak_last_error()).
 *   - one ctx = one GPU = one host thread; calls are synchronous on return.
 *   - there is NO CPU fallback: if no gfx950 device / code object is usable the
 *     create call fails with GPAK_EHIP.
 */
#ifndef GPAK_H
#define GPAK_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gpak_ctx gpak_ctx;

/* status codes */
#define GPAK_OK       0
#define GPAK_ENOTPD   1   /* B = I + K/sn2 not positive definite (Chol_fail)            */
#define GPAK_EINVAL   2   /* bad argument                                                */
#define GPAK_ESTATE   3   /* call out of order (e.g. no training set yet)                */
#define GPAK_ENOMEM   4   /* device allocation failed                                    */
#define GPAK_ENOTIMPL 5   /* not built (input columns other than 3 or 4, White gradient) */
#define GPAK_EHIP    (-1) /* HIP runtime error                                           */

/* precision (fixed at ctx creation).  GPAK_F64: everything in fp64.  GPAK_F32 (BASELINE.json
 * configs[4]): the M-proportional prediction work -- cross-kernel, forward substitution with M
 * right-hand sides, variance row sums -- runs in fp32 on the fp32 MFMA (twice the fp64 matrix
 * rate); Gram fill, factorisation, alpha, nlZ, gradient and the predictive MEAN stay fp64 (one
 * 0.2 s factorisation is noise beside a 1e6-point prediction). */
#define GPAK_F64 0
#define GPAK_F32 1

/* distance formulation inside MahaDist (Kernel.cpp:1370-1435) */
#define GPAK_DIST_EXPANSION 0 /* |u|^2+|v|^2-2u.v, pooled-mean centred, clamped at 0 (as written) */
#define GPAK_DIST_DIRECT    1 /* |(x_i-x_j)A|^2: same quantity, no cancellation noise (default)    */

/* predictive-variance compatibility flags */
#define GPAK_COMPAT_VARCLAMP 1 /* GP_Utils.cpp:1002-1003: 0/1 mask used as an index list */
#define GPAK_COMPAT_SN2SKIP  2 /* GP_Utils.cpp:1036-1040: "+sn2" skipped when sn2 == 1.0 */

/* ---- lifetime -------------------------------------------------------------------------- */
/* device = HIP device ordinal. */
int  gpak_create(gpak_ctx **out, int device, int precision);
/* ONE process driving n_gpus devices (devices[r] = HIP ordinal of rank r; NULL = 0 .. n_gpus-1): the same gpak_ctx
 * surface -- `gp_ss_ak --gpus n` runs on it.  logLikelihood / alpha run on the block-column-cyclic schedule of
 * gpak_dist.h over all devices (RCCL panel broadcasts; an in-process peer-copy transport when RCCL cannot start or
 * several ranks share a device); the gradient is distributed by row blocks of B^-1 (gpak_dist_grad); prediction is
 * sharded over the test points on the SAME distributed factor (each device assembles it from the packed panels its rank
 * holds: nothing is factored again); solve_chol and the factor copy use it on devices[0]; Gram copies rebuild K there.
 * 3- or 4-column inputs; any composition of gpak_set_kernel (for other children than ExpAns(+Bias) the gradients of
 * gpak_grad_hyb are formed on devices[0] from the distributed factor). */
int  gpak_create_multi(gpak_ctx **out, int n_gpus, const int *devices, int precision);
int  gpak_n_gpus(const gpak_ctx *ctx);
/* how the ranks of a multi-GPU context exchange panels: "rccl", "in-process peer copies" (+ the reason RCCL was
 * not used, if it was tried), or "none" for one GPU */
const char *gpak_transport(const gpak_ctx *ctx);
void gpak_destroy(gpak_ctx *ctx);
const char *gpak_last_error(const gpak_ctx *ctx);
/* library-level error text for failures that happen before a ctx exists */
const char *gpak_global_error(void);

/* ---- model state ------------------------------------------------------------------------ */
/* GP_utils ctor / test-time reload: members Xinp (N x d) and yTarg (N x 1)
 * (GP_Utils.cpp:9-46, gp_ss_ak.cpp:384-395).  d = 3 (x, y, z) or d = 4: a 4th "rock type" column that enters
 * the ExpAns distance with its own inverse width InversewidthR (Kernel.cpp:1411-1424) and the
 * Euclidean distance of Exp / RBF children like any other column (Kernel.cpp:1356-1362). */
int gpak_set_train(gpak_ctx *ctx, const double *X, const double *y, int N, int d);

/* GP_utils::set_GP_Pars (GP_Utils.cpp:130-157): expans[8] in the reference's order
 * {AngleX, inverseWidthx, AngleY, inverseWidthy, AngleZ, inverseWidthz, Sigma, InversewidthR}
 * (Kernel.cpp:737-761), Kern_Bias Sigma_Bias (Kernel.cpp:317-320), hyperlf(0)=sn2 used raw
 * (GP_Utils.cpp:405-406).  Like the reference, every call invalidates K, L and alpha
 * (GP_Utils.cpp:132-133) -- unless `memoise` was enabled with gpak_set_option and the
 * values are bit-identical to the previous call. */
int gpak_set_params(gpak_ctx *ctx, const double *expans, double bias, double sn2, int dist_mode);

/* General additive composition (HybKerns of other children, Kernel.cpp:140-154): up to 3 stationary
 * terms + Kern_Bias + Kern_White.  kinds[t] and the concatenated parameter list in the reference's
 * own order: GPAK_KERN_EXPANS 8 values (Kernel.cpp:737-761), GPAK_KERN_EXP {Hayper_Euc_Exp, Sigma_Exp}
 * (Kernel.cpp:576-600), GPAK_KERN_RBF {Hayper_Euc_RBF, inverseWidth_RBF, Sigma_RBF} (Kernel.cpp:411-428).
 * Gram, factor, alpha, nlZ and prediction work for any composition; gpak_grad only for ExpAns(+Bias). */
#define GPAK_KERN_EXPANS 0
#define GPAK_KERN_EXP    1
#define GPAK_KERN_RBF    2
int gpak_set_kernel(gpak_ctx *ctx, int nterms, const int *kinds, const double *pars, double bias, double white,
                    double sn2, int dist_mode);

#define GPAK_OPT_MEMOISE   1  /* value-based dirty tracking instead of always-invalidate   */
#define GPAK_OPT_NB_OUTER  2  /* outer Cholesky block width (multiple of 128)              */
#define GPAK_OPT_PROFILE   3  /* 1: bracket each trailing-update launch with hip events    */
#define GPAK_OPT_LOOKAHEAD 4  /* 1 (default): factor the next panel beside the bulk update */
/* schedule of the blocked factorisation (defaults = the measured best at N = 32768; the process-wide
 * defaults can also be set for A/B runs with the GPAK_* environment variables read ONCE at the first gpak_create) */
#define GPAK_OPT_NB_WIDE      5  /* panel width while more than NB_WIDE_ROWS rows are left (0: off; default 1024)  */
#define GPAK_OPT_NB_WIDE_ROWS 6  /* default 16384                                                                   */
#define GPAK_OPT_TAIL_ROWS    7  /* rows left from which the bulk updates leave 8 compute units idle (12288)        */
#define GPAK_OPT_FIRST_NARROW 8  /* 1 (default): the very first panel is NB_OUTER wide                              */
#define GPAK_OPT_INV512       9  /* 1 (default): explicit 512-block inverses for the back substitution             */
#define GPAK_OPT_POTRF_CO    10  /* 128 x 128 block kernel: 0 always 8 waves, 2 always the 4-wave build, 1 as needed */
#define GPAK_OPT_PRED_BATCH  11  /* test points per prediction batch (0: 16384 fp64, 65536 fp32)                    */
#define GPAK_OPT_BWD_FUSED   12  /* back substitution (needs INV512): 0 three launches per 512 columns, 1 the far column
                                    dots of the next block under this block's diagonal step, 2 (default) one launch  */
int gpak_set_option(gpak_ctx *ctx, int option, long value);

/* ---- hot path --------------------------------------------------------------------------- */
/* HybKerns::computeK(Xinp, Xinp, K, D2) over {ExpAns, Bias} (Kernel.cpp:140-154, 856-882,
 * 362-367, MahaDist :1370-1435).  K_host (N x N) and D2_host (N x N) may each be NULL. */
int gpak_gram(gpak_ctx *ctx, double *K_host, double *D2_host);

/* Kernels::computeK(X1, X2, K, D2) for arbitrary point sets with the ctx's parameters
 * (the GP_utils::_ComputeK_NewData call, GP_Utils.cpp:943-949, is X1 = Xinp). */
int gpak_compute_k(gpak_ctx *ctx, const double *X1, int n, const double *X2, int m, int d,
                   double *K_host, double *D2_host);

/* GP_utils::ldB2_exact (GP_Utils.cpp:872-915): B = I + (sW sW') % K, chol(B).
 * The factor is kept on the device as the LOWER factor L = R^T of arma::chol's upper R. */
int gpak_factor(gpak_ctx *ctx);
/* copy the factor out as arma::chol would return it in GP_utils::Lchol: R upper, N x N */
int gpak_get_chol_upper(gpak_ctx *ctx, double *R_host);
/* failing column (1-based) of the last GPAK_ENOTPD, 0 if none */
int gpak_failed_column(const gpak_ctx *ctx);

/* GP_utils::updateAlpha / irls (GP_Utils.cpp:191-228, 383-393) for the Gaussian likelihood:
 * alpha = (K + sn2 I)^-1 y, obtained with one factorisation and two triangular solves
 * (the IRLS fixed point).  alpha_host (N) may be NULL. */
int gpak_solve_alpha(gpak_ctx *ctx, double *alpha_host);

/* GP_utils::solve_chol(Lchol, Xr, dB) (GP_Utils.cpp:841-845): X := R^-1 R^-T X, X is N x k. */
int gpak_solve_chol(gpak_ctx *ctx, double *X_host, int k);

/* GP_utils::logLikelihood (GP_Utils.cpp:1138-1162): runs gram/factor/solve if dirty.
 * On GPAK_ENOTPD *nlz is quiet NaN (GP_Utils.cpp:1145-1146, 1155-1158). */
int gpak_nlz(gpak_ctx *ctx, double *nlz);
/* the three terms of GP_Utils.cpp:1159:  quad = Alpha' (0.5 f), sumlp = accu(lp), logdet */
int gpak_nlz_terms(gpak_ctx *ctx, double *quad, double *sumlp, double *logdet);

/* GP_utils::posteriorMeanVar (GP_Utils.cpp:1016-1043) = _ComputeK_NewData + _postMean +
 * _postVar (:943-1004).  Xte is M x d col-major; mean (M) required, var (M) may be NULL.
 * Test points are streamed in batches; the N x M cross-kernel is never materialised. */
int gpak_predict(gpak_ctx *ctx, const double *Xte, long M, int d, double *mean, double *var,
                 int compat_flags);

/* GP_utils::GradLL (GP_Utils.cpp:1171-1262) with Kern_ExpAnisotropic::getGradients
 * (Kernel.cpp:886-1263) and Kern_Bias::getGradients (:370-377), reference formulas as
 * written; g[10] = {8 ExpAns, bias, sn2}. */
int gpak_grad(gpak_ctx *ctx, double *g);

/* The same for any composition set with gpak_set_kernel: g holds the children's blocks in order
 * (ExpAns 8, Exp 2, RBF 3 entries; Kernel.cpp:886-1263, 644-693, 491-540 as written -- the Exp/RBF
 * children use GP_utils' D2 = the SUM of the children's D2), then the Bias entry (trace(QW)), then
 * sn2; ng must be exactly that length.  Compositions with a White child: GPAK_ENOTIMPL (the
 * reference has no gradient for it either). */
int gpak_grad_hyb(gpak_ctx *ctx, double *g, int ng);

/* ---- measurement ------------------------------------------------------------------------ */
typedef struct {
  double gram_ms;      /* fused fill of B = I + K/sn2 (lower tiles)                      */
  double factor_ms;    /* whole blocked Cholesky                                         */
  double solve_ms;     /* two triangular solves                                          */
  double nlz_ms;       /* f = K alpha (fused Gram-matvec), lp, reductions                */
  double predict_ms;   /* last gpak_predict                                              */
  double grad_ms;      /* last gpak_grad                                                 */
  /* trailing-update kernel (the dominant, MFMA-bound launch), last factorisation,
   * measured with hipEvents on the ctx stream when GPAK_OPT_PROFILE is set: */
  double trailing_ms;        /* sum of launch durations                                  */
  double trailing_flops;     /* algorithmic flops of those launches (lower tiles only)   */
  int    trailing_launches;
  /* fused fill kernel: bytes it must write */
  double gram_bytes;
  int    n;                  /* N                                                        */
  int    n_padded;           /* N rounded up to the 128-tile                             */
  /* algorithmic HBM bytes of those trailing-update launches: every lower C tile read once and
   * written once (2 x 128 x 128 x 8 B per tile), whatever the panel width K of the launch     */
  double trailing_bytes;
  double kmatvec_ms;         /* the fused Gram-matvec f = K alpha inside nlz_ms (fp64 VALU bound) */
  double accumulated_ms[4];  /* gram / factor / solve / nlz summed over every evaluation since
                                gpak_set_train, and ...                                          */
  int    evaluations;        /* ... how many evaluations (factorisations) that was               */
} gpak_phase_times;
int gpak_timing(gpak_ctx *ctx, gpak_phase_times *out);

/* re-read the GPAK_* tuning environment into the process-wide defaults (A/B tooling inside one process; contexts made
 * afterwards see it) */
void gpak_reload_tuning(void);

/* fp64-MFMA and HBM-write calibration microbenchmarks */
int gpak_calibrate(gpak_ctx *ctx, double *mfma_f64_tflops, double *hbm_write_gbs);

#ifdef __cplusplus
}
#endif
#endif /* GPAK_H */
