/*
 * gpak_oracle.h -- CPU restatement (fp64, plain C) of the GP_SS_AK hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped
 * product: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load this library, and only as the checker / the timed CPU baseline.
 * The product library (libgpak_hip.so) never links or calls it.
 *
 * PARITY UNPINNED: the reference ships no tests, fixtures or golden vectors
 *  and cannot be compiled here (every translation unit
 * needs <armadillo>, which is absent: SURVEY.md section 8c).  This restatement
 * is therefore pinned only by (i) line-by-line citation of the reference
 * source, (ii) closed-form known-answer tests and (iii) an independent
 * NumPy/SciPy(LAPACK) cross-check in tests/test_oracle.py.
 *
 * All matrices are column-major doubles (arma::mat layout).
 */
#ifndef GPAK_ORACLE_H
#define GPAK_ORACLE_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* distance formulation */
#define ORC_DIST_EXPANSION 0 /* |u|^2+|v|^2-2u.v, pooled-mean centred, clamped: Kernel.cpp:1391-1434 */
#define ORC_DIST_DIRECT    1 /* |(x_i-x_j)A|^2 : same quantity without the cancellation noise      */

/* predictive-variance compatibility flags */
#define ORC_COMPAT_VARCLAMP 1 /* Q3: GP_Utils.cpp:1002-1003 mask-used-as-index "clamp"     */
#define ORC_COMPAT_SN2SKIP  2 /* Q4: GP_Utils.cpp:1036-1040 skip "+sn2" when sn2 == 1.0    */

/* Optionally route Cholesky / triangular solves / GEMV through a LAPACK/BLAS
 * found in a shared object (the SciPy wheel's OpenBLAS: symbols scipy_dpotrf_,
 * scipy_dtrsm_, scipy_dgemv_, scipy_dgemm_), the way Armadillo would issue them
 * (GP_Utils.cpp:881,903 chol; :843-844 solve(trimatl/trimatu); :396 K*alp).
 * Returns 0 when all symbols were bound, non-zero otherwise (in-repo blocked C
 * fallback stays active).  threads<=0 leaves the library default. */
int orc_use_lapack(const char *so_path, int threads);
/* team size of the oracle's own OpenMP loops (oracle.py sets min(usable CPUs, 16) on load) */
void orc_set_threads(int n);
int orc_lapack_active(void);

/* sigInv = Rot * diag(L) * Rot^T  (Kernel.cpp:1399-1425).  par = ParamKer =
 * {alpha, beta, teta, L_alpha, L_beta, L_teta [, L_r]}; A is d x d col-major. */
void orc_siginv(int d, const double *par, double *A);

/* MahaDist (Kernel.cpp:1370-1435).  X1 n x d, X2 m x d, D2 n x m. */
void orc_mahadist(const double *X1, int n, const double *X2, int m, int d,
                  const double *par, int mode, double *D2);

/* HybKerns{ExpAns + Bias}::computeK  (Kernel.cpp:856-882, 362-367, 140-154).
 * expans[8] in the reference's parameter order (Kernel.cpp:737-761):
 * {AngleX, iwx, AngleY, iwy, AngleZ, iwz, Sigma, iwR}.  D2 may be NULL. */
void orc_gram(const double *X1, int n, const double *X2, int m, int d,
              const double *expans, double bias, int mode, double *K, double *D2);

/* HybKerns::computeK over arbitrary children: kinds[t] 0 ExpAns / 1 Exp / 2 RBF with their parameters
 * concatenated in the reference's order, Kern_Bias, Kern_White (Kernel.cpp:140-154, 256-263, 482-488,
 * 1343-1368).  D2sum (sum of the children's D2) may be NULL. */
void orc_gram_hyb(const double *X1, int n, const double *X2, int m, int d, int nterms, const int *kinds,
                  const double *pars, double bias, double white, int mode, double *K, double *D2sum);

/* diag_Compute of the composite kernel (Kernel.cpp:780-783, 328-332, 127-136) */
double orc_kdiag(const double *expans, double bias);

/* In-place lower Cholesky A = L L^T (L = R^T of arma::chol's upper R).
 * Returns 0, or j+1 when the leading minor of order j+1 is not positive
 * definite (LAPACK dpotrf info convention). Upper triangle is left untouched. */
int orc_potrf_lower(int n, double *A, int lda);

/* X := L^-T L^-1 X   (solve_chol, GP_Utils.cpp:841-845), X is n x k col-major */
void orc_solve_chol(int n, const double *L, int ldl, double *X, int k, int ldx);
/* X := L^-1 X only (forward substitution) */
void orc_trsm_lower(int n, const double *L, int ldl, double *X, int k, int ldx);

typedef struct {
  double nlz;        /* GP_Utils.cpp:1159 */
  double logdet;     /* Lchol_db2 = sum log diag chol(B), GP_Utils.cpp:913 */
  double quad;       /* alpha^T (0.5 f) */
  double sumlp;      /* accu(lp) */
  int    chol_fail;  /* Chol_fail flag, GP_Utils.cpp:881-888 */
  int    n_chol;     /* number of Cholesky factorisations performed */
  int    n_gemv;     /* number of K*v products performed */
  int    irls_iters; /* iterations of the while loop GP_Utils.cpp:206-227 */
  double last_step;  /* Brent step length chosen in the last iteration */
} orc_nlz_info;

/* Reference operation sequence of GP_utils::logLikelihood() for the Gaussian
 * likelihood (GP_Utils.cpp:1138-1162 -> updateAlpha/irls :191-228 -> brentmin
 * :229-381 -> PSI :180-190 -> updatelikelihood :398-416 -> ldB2_exact :872-915).
 * K is the N x N Gram (full storage), alpha is in/out (warm start, :69 zero
 * initialised on first use), Lout (N x N, may be NULL) receives the lower
 * factor of B = I + K/sn2 left by the final ldB2_exact(). */
void orc_nlz_refseq(int N, const double *K, const double *y, double sn2,
                    double *alpha, double *Lout, orc_nlz_info *info);

/* Same outputs through one Cholesky + two triangular solves:
 * alpha = (K + sn2 I)^-1 y, f = K alpha, then GP_Utils.cpp:1159 verbatim. */
void orc_nlz_lean(int N, const double *K, const double *y, double sn2,
                  double *alpha, double *Lout, orc_nlz_info *info);

/* posteriorMeanVar (GP_Utils.cpp:943-1043): Xtr N x d, Xte M x d, alpha N,
 * L lower factor of B.  var may be NULL. */
void orc_predict(const double *Xtr, int N, const double *Xte, int M, int d,
                 const double *expans, double bias, double sn2, int mode,
                 const double *alpha, const double *L, int compat_flags,
                 double *mean, double *var);

/* GradLL + dhyp + updateG + updateGlikelihood (GP_Utils.cpp:1164-1284, 846-864)
 * with Kern_ExpAnisotropic::getGradients (Kernel.cpp:886-1263) and
 * Kern_Bias::getGradients (Kernel.cpp:370-377), reference formulas as written
 * (this is NOT the true gradient).  K full N x N, L lower
 * factor of B, g[10] = {8 ExpAns, bias, sn2}. 3-D inputs only. */
void orc_grad_ref(const double *X, int N, const double *y, const double *K,
                  const double *L, const double *alpha, const double *expans,
                  double bias, double sn2, int mode, double *g);

/* the same for d = 3 or 4 input columns (4th = rock type); g[7] is non-zero only for d = 4 */
void orc_grad_ref_d(const double *X, int N, int d, const double *y, const double *K,
                    const double *L, const double *alpha, const double *expans,
                    double bias, double sn2, int mode, double *g);

/* orc_grad_ref_d with Q = B^-1 supplied by the caller (leading dimension ldq) and every other N x N quantity
 * rebuilt slab by slab: one N x N array instead of six.  For the N = 32768 gradient golden. */
void orc_grad_ref_q(const double *X, int N, int d, const double *y, const double *Q, size_t ldq,
                    const double *alpha, const double *expans, double bias, double sn2, int mode, double *g);

/* GradLL for an arbitrary composition; see the .c file.  g: children in order (8 / 2 / 3), bias, sn2 */
void orc_grad_hyb(const double *X, int N, const double *y, const double *K, const double *L, const double *alpha,
                  int nterms, const int *kinds, const double *pars, int has_bias, double sn2, int mode, double *g);

void orc_grad_hyb_d(const double *X, int N, int d, const double *y, const double *K, const double *L,
                    const double *alpha, int nterms, const int *kinds, const double *pars, int has_bias, double sn2,
                    int mode, double *g);

#ifdef __cplusplus
}
#endif
#endif
