/*
 * gpak_oracle.c -- CPU restatement (fp64, plain C) of the GP_SS_AK hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see gpak_oracle.h).  PARITY UNPINNED by the
 * reference (no upstream tests/fixtures; reference unbuildable without
 * Armadillo) -- pinned by citation, known-answer tests and a SciPy cross-check.
 *
 * Every function cites the reference file:line it restates.  Storage is
 * column-major double, like arma::mat.
 */
#include "gpak_oracle.h"

#include <dlfcn.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ------------------------------------------------------------------------- */
/* optional LAPACK/BLAS backend (what Armadillo would call)                  */
/* ------------------------------------------------------------------------- */
typedef void (*dpotrf_fn)(const char *, const int *, double *, const int *, int *);
typedef void (*dtrsm_fn)(const char *, const char *, const char *, const char *, const int *,
                         const int *, const double *, const double *, const int *, double *,
                         const int *);
typedef void (*dgemv_fn)(const char *, const int *, const int *, const double *, const double *,
                         const int *, const double *, const int *, const double *, double *,
                         const int *);
typedef void (*setthr_fn)(int);

static dpotrf_fn g_dpotrf = 0;
static dtrsm_fn g_dtrsm = 0;
static dgemv_fn g_dgemv = 0;

int orc_use_lapack(const char *so_path, int threads) {
  g_dpotrf = 0; g_dtrsm = 0; g_dgemv = 0;
  if (!so_path) return 1;
  void *h = dlopen(so_path, RTLD_NOW | RTLD_LOCAL);
  if (!h) return 2;
  dpotrf_fn p = (dpotrf_fn)dlsym(h, "scipy_dpotrf_");
  dtrsm_fn t = (dtrsm_fn)dlsym(h, "scipy_dtrsm_");
  dgemv_fn v = (dgemv_fn)dlsym(h, "scipy_dgemv_");
  if (!p) p = (dpotrf_fn)dlsym(h, "dpotrf_");
  if (!t) t = (dtrsm_fn)dlsym(h, "dtrsm_");
  if (!v) v = (dgemv_fn)dlsym(h, "dgemv_");
  if (!p || !t || !v) return 3;
  if (threads > 0) {
    setthr_fn s = (setthr_fn)dlsym(h, "scipy_openblas_set_num_threads");
    if (!s) s = (setthr_fn)dlsym(h, "openblas_set_num_threads");
    if (s) s(threads);
  }
  g_dpotrf = p; g_dtrsm = t; g_dgemv = v;
  return 0;
}
int orc_lapack_active(void) { return g_dpotrf != 0; }

/* ------------------------------------------------------------------------- */
/* Kernel part                                                               */
/* ------------------------------------------------------------------------- */

/* Rot as filled at Kernel.cpp:1402-1414 (row r, col c -> Rot[r + c*d]) */
static void rot_fill(int d, double alpha, double beta, double teta, double *Rot) {
  memset(Rot, 0, sizeof(double) * d * d);
  double ca = cos(alpha), sa = sin(alpha), cb = cos(beta), sb = sin(beta);
  double ct = cos(teta), st = sin(teta);
#define RT(r, c) Rot[(r) + (c) * d]
  RT(0, 0) = ca * ct + sa * sb * st;
  RT(0, 1) = -sa * ct + ca * sb * st;
  RT(0, 2) = -cb * st;
  RT(1, 0) = sa * cb;
  RT(1, 1) = ca * cb;
  RT(1, 2) = sb;
  RT(2, 0) = ca * st - sa * sb * ct;
  RT(2, 1) = -sa * st - ca * sb * ct;
  RT(2, 2) = cb * ct;
  if (d == 4) RT(3, 3) = 1.0;
#undef RT
}

/* sigInv = Rot * lambda * Rot.t()   Kernel.cpp:1417-1425 */
void orc_siginv(int d, const double *par, double *A) {
  double Rot[16], lam[4];
  rot_fill(d, par[0], par[1], par[2], Rot);
  lam[0] = par[3]; lam[1] = par[4]; lam[2] = par[5];
  if (d == 4) lam[3] = par[6];
  for (int r = 0; r < d; r++)
    for (int c = 0; c < d; c++) {
      double s = 0.0;
      for (int k = 0; k < d; k++) s += Rot[r + k * d] * lam[k] * Rot[c + k * d];
      A[r + c * d] = s;
    }
}

/* MahaDist  Kernel.cpp:1370-1435 */
void orc_mahadist(const double *X1, int n, const double *X2, int m, int d, const double *par,
                  int mode, double *D2) {
  double A[16], mu[4];
  orc_siginv(d, par, A);
  /* pooled mean, Kernel.cpp:1391-1392: mX1 = n/(n+m)*colmean(X1); mX2 = m/(n+m)*colmean(X2)+mX1 */
  for (int j = 0; j < d; j++) {
    double s1 = 0.0, s2 = 0.0;
    for (int i = 0; i < n; i++) s1 += X1[i + (size_t)j * n];
    for (int i = 0; i < m; i++) s2 += X2[i + (size_t)j * m];
    double mX1 = (double)n / (n + m) * s1 / n;
    mu[j] = (double)m / (n + m) * s2 / m + mX1;
  }
  /* X1 -= mX2; X1 *= sigInv   (:1393-1427) */
  double *U = (double *)malloc(sizeof(double) * (size_t)n * d);
  double *V = (double *)malloc(sizeof(double) * (size_t)m * d);
  for (int i = 0; i < n; i++)
    for (int c = 0; c < d; c++) {
      double s = 0.0;
      for (int k = 0; k < d; k++) s += (X1[i + (size_t)k * n] - mu[k]) * A[k + c * d];
      U[i + (size_t)c * n] = s;
    }
  for (int i = 0; i < m; i++)
    for (int c = 0; c < d; c++) {
      double s = 0.0;
      for (int k = 0; k < d; k++) s += (X2[i + (size_t)k * m] - mu[k]) * A[k + c * d];
      V[i + (size_t)c * m] = s;
    }
  if (mode == ORC_DIST_EXPANSION) {
    /* D2 = sum(X1%X1,1)*1' + 1*sum(X2%X2,1)' - 2*X1*X2' ; clamp (:1431-1434) */
    double *su = (double *)malloc(sizeof(double) * n), *sv = (double *)malloc(sizeof(double) * m);
    for (int i = 0; i < n; i++) {
      double s = 0.0;
      for (int k = 0; k < d; k++) s += U[i + (size_t)k * n] * U[i + (size_t)k * n];
      su[i] = s;
    }
    for (int j = 0; j < m; j++) {
      double s = 0.0;
      for (int k = 0; k < d; k++) s += V[j + (size_t)k * m] * V[j + (size_t)k * m];
      sv[j] = s;
    }
#pragma omp parallel for schedule(static)
    for (int j = 0; j < m; j++)
      for (int i = 0; i < n; i++) {
        double dot = 0.0;
        for (int k = 0; k < d; k++) dot += U[i + (size_t)k * n] * V[j + (size_t)k * m];
        double v = su[i] + sv[j] - 2.0 * dot;
        D2[i + (size_t)j * n] = v < 0 ? 0.0 : v;
      }
    free(su); free(sv);
  } else {
#pragma omp parallel for schedule(static)
    for (int j = 0; j < m; j++)
      for (int i = 0; i < n; i++) {
        double s = 0.0;
        for (int k = 0; k < d; k++) {
          double t = U[i + (size_t)k * n] - V[j + (size_t)k * m];
          s += t * t;
        }
        D2[i + (size_t)j * n] = s;
      }
  }
  free(U); free(V);
}

/* OpenMP team size of this library's own loops (the default is one thread per visible core: on a box whose
 * cgroup grants 16 of 128+ cores that is a heavily oversubscribed team -- seconds per call at N = 1). */
void orc_set_threads(int n) { if (n > 0) omp_set_num_threads(n); }

/* ParamKer packing of Kern_ExpAnisotropic::computeK, Kernel.cpp:864-878 */
static void pack_paramker(const double *e, double *par) {
  par[0] = e[0]; par[1] = e[2]; par[2] = e[4];
  par[3] = e[1]; par[4] = e[3]; par[5] = e[5];
  par[6] = e[7];
}

/* HybKerns::computeK over {ExpAns, Bias}: Kernel.cpp:140-154, 856-882, 362-367 */
void orc_gram(const double *X1, int n, const double *X2, int m, int d, const double *expans,
              double bias, int mode, double *K, double *D2) {
  double par[7];
  pack_paramker(expans, par);
  double *D = D2 ? D2 : (double *)malloc(sizeof(double) * (size_t)n * m);
  orc_mahadist(X1, n, X2, m, d, par, mode, D);
  double var2 = expans[6] * expans[6]; /* :861 */
  size_t tot = (size_t)n * m;
#pragma omp parallel for schedule(static)
  for (size_t t = 0; t < tot; t++)
    K[t] = var2 * exp(-1.0 * sqrt(D[t])) + bias; /* :881 + Kern_Bias fill :366 */
  if (!D2) free(D);
}

double orc_kdiag(const double *expans, double bias) { return expans[6] * expans[6] + bias; }

/* EuclDist, Kernel.cpp:1343-1368 with mlA :1437-1441: AX = X * hyp^-2;
 * D2 = sum(AX1 % X1,1) 1' + 1 sum(AX2 % X2,1)' - 2 X1 AX2', pooled-mean centred, clamped */
void orc_eucldist(const double *X1, int n, const double *X2, int m, int d, double hyp, int mode, double *D2) {
  double mu[4];
  for (int j = 0; j < d; j++) {
    double s1 = 0.0, s2 = 0.0;
    for (int i = 0; i < n; i++) s1 += X1[i + (size_t)j * n];
    for (int i = 0; i < m; i++) s2 += X2[i + (size_t)j * m];
    double mX1 = (double)n / (n + m) * s1 / n;
    mu[j] = (double)m / (n + m) * s2 / m + mX1;
  }
  const double sc = exp(-2.0 * log(hyp));
#pragma omp parallel for schedule(static)
  for (int j = 0; j < m; j++)
    for (int i = 0; i < n; i++) {
      double v;
      if (mode == ORC_DIST_EXPANSION) {
        double a = 0.0, b = 0.0, c = 0.0;
        for (int k = 0; k < d; k++) {
          double x1 = X1[i + (size_t)k * n] - mu[k], x2 = X2[j + (size_t)k * m] - mu[k];
          a += sc * x1 * x1; b += sc * x2 * x2; c += x1 * (sc * x2);
        }
        v = a + b - 2.0 * c;
        if (v < 0) v = 0.0;
      } else {
        v = 0.0;
        for (int k = 0; k < d; k++) {
          double t = (X1[i + (size_t)k * n] - X2[j + (size_t)k * m]) / hyp;
          v += t * t;
        }
      }
      D2[i + (size_t)j * n] = v;
    }
}

/* HybKerns::computeK over arbitrary children (Kernel.cpp:140-154): kinds 0 ExpAns (8 pars,
 * :856-882), 1 Exp (2 pars {hyp, sigma}), 2 RBF (3 pars {hyp, iw, sigma}, :482-488); plus
 * Kern_Bias (:362-367) and Kern_White (:256-263: diagonal only when X1(0)==X2(0) and n==m). */
void orc_gram_hyb(const double *X1, int n, const double *X2, int m, int d, int nterms, const int *kinds,
                  const double *pars, double bias, double white, int mode, double *K, double *D2sum) {
  size_t tot = (size_t)n * m;
  double *D = (double *)malloc(sizeof(double) * tot);
  for (size_t t = 0; t < tot; t++) { K[t] = bias; if (D2sum) D2sum[t] = 0.0; }
  const double *p = pars;
  for (int t = 0; t < nterms; t++) {
    if (kinds[t] == 0) {
      double par[7];
      pack_paramker(p, par);
      orc_mahadist(X1, n, X2, m, d, par, mode, D);
      double v2 = p[6] * p[6];
      for (size_t e = 0; e < tot; e++) K[e] += v2 * exp(-1.0 * sqrt(D[e]));
      p += 8;
    } else if (kinds[t] == 1) {
      orc_eucldist(X1, n, X2, m, d, p[0], mode, D);
      double v2 = p[1] * p[1];
      for (size_t e = 0; e < tot; e++) K[e] += v2 * exp(-1.0 * sqrt(D[e]));
      p += 2;
    } else {
      orc_eucldist(X1, n, X2, m, d, p[0], mode, D);
      double v2 = p[2] * p[2];
      for (size_t e = 0; e < tot; e++) K[e] += exp(-0.5 * p[1] * D[e]) * v2;
      p += 3;
    }
    if (D2sum) for (size_t e = 0; e < tot; e++) D2sum[e] += D[e];
  }
  if (white != 0.0 && X1[0] == X2[0] && n == m)
    for (int i = 0; i < n; i++) K[i + (size_t)i * n] += white;
  free(D);
}

/* ------------------------------------------------------------------------- */
/* Dense linear algebra (in-repo fallback when no LAPACK is bound)           */
/* ------------------------------------------------------------------------- */
#define NB 96
#define MR 8
#define NR 4
typedef double v4d __attribute__((vector_size(32)));

static int potf2_lower(int n, double *A, int lda) {
  for (int j = 0; j < n; j++) {
    double ajj = A[j + (size_t)j * lda];
    for (int k = 0; k < j; k++) ajj -= A[j + (size_t)k * lda] * A[j + (size_t)k * lda];
    if (!(ajj > 0.0)) return j + 1;
    ajj = sqrt(ajj);
    A[j + (size_t)j * lda] = ajj;
    for (int k = 0; k < j; k++) {
      double l = A[j + (size_t)k * lda];
      const double *src = A + (size_t)k * lda;
      double *dst = A + (size_t)j * lda;
      for (int i = j + 1; i < n; i++) dst[i] -= src[i] * l;
    }
    double inv = 1.0 / ajj;
    for (int i = j + 1; i < n; i++) A[i + (size_t)j * lda] *= inv;
  }
  return 0;
}

/* rows [0,m) of P (m x jb, ld lda): P := P * L11^-T, L11 jb x jb lower */
static void trsm_right_lt(int m, int jb, const double *L11, int ldl, double *P, int lda) {
#pragma omp parallel for schedule(static)
  for (int r0 = 0; r0 < m; r0 += 256) {
    int r1 = r0 + 256 < m ? r0 + 256 : m;
    for (int c = 0; c < jb; c++) {
      double *pc = P + (size_t)c * lda;
      for (int k = 0; k < c; k++) {
        double l = L11[c + (size_t)k * ldl];
        const double *pk = P + (size_t)k * lda;
        for (int i = r0; i < r1; i++) pc[i] -= pk[i] * l;
      }
      double inv = 1.0 / L11[c + (size_t)c * ldl];
      for (int i = r0; i < r1; i++) pc[i] *= inv;
    }
  }
}

/* C (m x m lower, ld lda) -= P P^T with P m x jb packed as Pp[tile][k][MR] */
static void syrk_lower_packed(int m, int jb, const double *Pp, double *C, int lda) {
  int ntile = (m + MR - 1) / MR;
#pragma omp parallel for schedule(dynamic, 4)
  for (int it = 0; it < ntile; it++) {
    int i0 = it * MR;
    const double *Ai = Pp + (size_t)it * jb * MR;
    for (int c0 = 0; c0 <= i0 + MR - 1 && c0 < m; c0 += NR) {
      const double *Bj = Pp + (size_t)(c0 / MR) * jb * MR + (c0 % MR);
      v4d acc[NR][2];
      for (int c = 0; c < NR; c++) { acc[c][0] = (v4d){0, 0, 0, 0}; acc[c][1] = (v4d){0, 0, 0, 0}; }
      for (int k = 0; k < jb; k++) {
        v4d a0 = *(const v4d *)(Ai + (size_t)k * MR);
        v4d a1 = *(const v4d *)(Ai + (size_t)k * MR + 4);
        for (int c = 0; c < NR; c++) {
          double b = Bj[(size_t)k * MR + c];
          v4d bb = {b, b, b, b};
          acc[c][0] += a0 * bb;
          acc[c][1] += a1 * bb;
        }
      }
      for (int c = 0; c < NR; c++) {
        int col = c0 + c;
        if (col >= m) break;
        for (int r = 0; r < MR; r++) {
          int row = i0 + r;
          if (row >= m || row < col) continue;
          C[row + (size_t)col * lda] -= (r < 4 ? acc[c][0][r] : acc[c][1][r - 4]);
        }
      }
    }
  }
}

int orc_potrf_lower(int n, double *A, int lda) {
  if (g_dpotrf) {
    int info = 0;
    g_dpotrf("L", &n, A, &lda, &info);
    return info > 0 ? info : 0;
  }
  double *Pp = 0;
  if (posix_memalign((void **)&Pp, 64, sizeof(double) * ((size_t)n + MR) * NB)) return -1;
  for (int j = 0; j < n; j += NB) {
    int jb = n - j < NB ? n - j : NB;
    int info = potf2_lower(jb, A + j + (size_t)j * lda, lda);
    if (info) { free(Pp); return j + info; }
    int m = n - j - jb;
    if (m <= 0) break;
    double *P = A + (j + jb) + (size_t)j * lda;
    trsm_right_lt(m, jb, A + j + (size_t)j * lda, lda, P, lda);
    int ntile = (m + MR - 1) / MR;
#pragma omp parallel for schedule(static)
    for (int it = 0; it < ntile; it++)
      for (int k = 0; k < jb; k++)
        for (int r = 0; r < MR; r++) {
          int row = it * MR + r;
          Pp[(size_t)it * jb * MR + (size_t)k * MR + r] = row < m ? P[row + (size_t)k * lda] : 0.0;
        }
    syrk_lower_packed(m, jb, Pp, A + (j + jb) + (size_t)(j + jb) * lda, lda);
  }
  free(Pp);
  return 0;
}

void orc_trsm_lower(int n, const double *L, int ldl, double *X, int k, int ldx) {
  if (g_dtrsm) {
    double one = 1.0;
    g_dtrsm("L", "L", "N", "N", &n, &k, &one, L, &ldl, X, &ldx);
    return;
  }
#pragma omp parallel for schedule(dynamic, 1) if (k > 1)
  for (int c = 0; c < k; c++) {
    double *x = X + (size_t)c * ldx;
    for (int j = 0; j < n; j++) {
      double xj = x[j] / L[j + (size_t)j * ldl];
      x[j] = xj;
      const double *lj = L + (size_t)j * ldl;
      for (int i = j + 1; i < n; i++) x[i] -= lj[i] * xj;
    }
  }
}

static void trsm_lower_trans(int n, const double *L, int ldl, double *X, int k, int ldx) {
  if (g_dtrsm) {
    double one = 1.0;
    g_dtrsm("L", "L", "T", "N", &n, &k, &one, L, &ldl, X, &ldx);
    return;
  }
#pragma omp parallel for schedule(dynamic, 1) if (k > 1)
  for (int c = 0; c < k; c++) {
    double *x = X + (size_t)c * ldx;
    for (int j = n - 1; j >= 0; j--) {
      const double *lj = L + (size_t)j * ldl;
      double s = x[j];
      for (int i = j + 1; i < n; i++) s -= lj[i] * x[i];
      x[j] = s / lj[j];
    }
  }
}

/* solve_chol, GP_Utils.cpp:841-845:  Xr = solve(trimatl(Lc.t()), dB); Xr = solve(trimatu(Lc), Xr)
 * with Lc the upper factor R; here L = R^T. */
void orc_solve_chol(int n, const double *L, int ldl, double *X, int k, int ldx) {
  orc_trsm_lower(n, L, ldl, X, k, ldx);
  trsm_lower_trans(n, L, ldl, X, k, ldx);
}

/* mvmK_exact, GP_Utils.cpp:394-397 */
static void gemv_full(int N, const double *K, const double *v, double *out) {
  if (g_dgemv) {
    double one = 1.0, zero = 0.0;
    int inc = 1;
    g_dgemv("N", &N, &N, &one, K, &N, v, &inc, &zero, out, &inc);
    return;
  }
#pragma omp parallel for schedule(static)
  for (int r0 = 0; r0 < N; r0 += 512) {
    int r1 = r0 + 512 < N ? r0 + 512 : N;
    for (int i = r0; i < r1; i++) out[i] = 0.0;
    for (int j = 0; j < N; j++) {
      const double *kj = K + (size_t)j * N;
      double vj = v[j];
      for (int i = r0; i < r1; i++) out[i] += kj[i] * vj;
    }
  }
}

/* ------------------------------------------------------------------------- */
/* Inference: reference operation sequence                                   */
/* ------------------------------------------------------------------------- */
typedef struct {
  int N;
  const double *K, *y;
  double sn2;
  double *alpha; /* member Alpha */
  double *mvmK, *lp, *dlp, *d2lp, *Lchol;
  int chol_fail, n_chol, n_gemv;
} gp_state;

/* updatelikelihood(fval), Gaussian case, GP_Utils.cpp:398-416 */
static void lik_update(gp_state *s, const double *fval) {
  double sn2 = s->sn2;
  double c = log(2.0 * M_PI * sn2) / 2.0;
  for (int i = 0; i < s->N; i++) {
    double ymmu = s->y[i] - fval[i];
    s->lp[i] = ymmu * ymmu * (-1.0 / (2.0 * sn2)) - c;
    s->dlp[i] = (1.0 / sn2) * ymmu;
    s->d2lp[i] = 1.0 / sn2;
  }
}

/* PSI, GP_Utils.cpp:180-190 (mf == 0, updateMean :1126-1129) */
static double psi_eval(gp_state *s, const double *alp, double *fval) {
  gemv_full(s->N, s->K, alp, s->mvmK);
  s->n_gemv++;
  memcpy(fval, s->mvmK, sizeof(double) * s->N);
  lik_update(s, fval);
  double q = 0.0, slp = 0.0;
  for (int i = 0; i < s->N; i++) { q += alp[i] * (0.5 * fval[i]); slp += s->lp[i]; }
  return q - slp;
}

/* Lchol = chol((Sw Sw') % K + I), GP_Utils.cpp:874-888 / 896-910 */
static int form_and_factor_B(gp_state *s, const double *W) {
  int N = s->N;
#pragma omp parallel for schedule(static)
  for (int j = 0; j < N; j++) {
    double swj = sqrt(W[j]);
    for (int i = 0; i < N; i++)
      s->Lchol[i + (size_t)j * N] = sqrt(W[i]) * swj * s->K[i + (size_t)j * N] + (i == j ? 1.0 : 0.0);
  }
  s->n_chol++;
  int info = orc_potrf_lower(N, s->Lchol, N);
  s->chol_fail = info != 0;
  return info;
}

/* ldB2_exact(WW, r, QQ), GP_Utils.cpp:872-893 */
static void ldB2_solve(gp_state *s, const double *W, const double *r, double *QQ) {
  if (form_and_factor_B(s, W)) return;
  for (int i = 0; i < s->N; i++) QQ[i] = sqrt(W[i]) * r[i];
  orc_solve_chol(s->N, s->Lchol, s->N, QQ, 1, s->N);
  for (int i = 0; i < s->N; i++) QQ[i] *= sqrt(W[i]);
}

/* sign(), ModelInf.h:14-20 */
static double ref_sign(double v) { return v <= 0 ? -1.0 : 1.0; }

/* brentmin, GP_Utils.cpp:229-381 */
static double brentmin(gp_state *s, double *Fv, const double *dalpha, double *xmin_out) {
  int N = s->N;
  double *Xc = (double *)malloc(sizeof(double) * N);
  const double smin_line = 0.0, smax_line = 2.0;
  const int nmax_line = 10;
  const double thr_line = 1e-4;
  int counters = 0;
  for (int i = 0; i < N; i++) Xc[i] = s->alpha[i] + smin_line * dalpha[i];
  double fa = psi_eval(s, Xc, Fv); counters++;
  for (int i = 0; i < N; i++) Xc[i] = dalpha[i] * smax_line + s->alpha[i];
  double fb = psi_eval(s, Xc, Fv); counters++;
  double seps = sqrt(2.220446049250313e-16);
  double c = 0.5 * (3.0 - sqrt(5.0));
  double a = smin_line, b = smax_line;
  double v = a + c * (b - a), w = v, xf = v, d = 0.0, e = 0.0, x = xf;
  for (int i = 0; i < N; i++) Xc[i] = s->alpha[i] + x * dalpha[i];
  double fc = psi_eval(s, Xc, Fv); counters++;
  double fv = fc, fw = fc;
  double xm = 0.5 * (a + b);
  double tol1 = seps * fabs(xf) + thr_line / 3.0, tol2 = 2.0 * tol1;
  double si, r, q, p, sd, fu;
  int gs;
  while (fabs(xf - xm) > (tol2 - 0.5 * (b - a))) {
    gs = 1;
    if (fabs(e) > tol1) {
      gs = 0;
      r = (xf - w) * (fc - fv);
      q = (xf - v) * (fc - fw);
      p = (xf - v) * q - (xf - w) * r;
      q = 2.0 * (q - r);
      if (q > 0.0) p = -p;
      q = fabs(q);
      r = e; e = d;
      if ((fabs(p) < fabs(0.5 * q * r)) && (p > q * (a - xf)) && (p < q * (b - xf))) {
        d = p / q;
        x = xf + d;
        if (((x - a) < tol2) || ((b - x) < tol2)) {
          si = ref_sign(xm - xf) + ((xm - xf) == 0);
          d = tol1 * si;
        }
      } else {
        gs = 1;
      }
    }
    if (gs == 1) {
      if (xf >= xm) e = a - xf; else e = b - xf;
      d = c * e;
    }
    si = ref_sign(d) + (d == 0);
    sd = (fabs(d) < tol1) ? tol1 : fabs(d);
    x = xf + si * sd;
    for (int i = 0; i < N; i++) Xc[i] = dalpha[i] * x + s->alpha[i];
    fu = psi_eval(s, Xc, Fv); counters++;
    if (fu <= fc) {
      if (x >= xf) a = xf; else b = xf;
      v = w; fv = fw; w = xf; fw = fc; xf = x; fc = fu;
    } else {
      if (x < xf) a = x; else b = x;
      if ((fu <= fw) || (w == xf)) { v = w; fv = fw; w = x; fw = fu; }
      else if ((fu <= fv) || (v == xf) || (v == w)) { v = x; fv = fu; }
    }
    xm = 0.5 * (a + b);
    tol1 = seps * fabs(xf) + thr_line / 3.0;
    tol2 = 2.0 * tol1;
    if (counters >= nmax_line) break;
  }
  if ((fa < fc) && (fa <= fb)) { xf = smin_line; fc = fa; }
  else if (fb < fc) { xf = smax_line; fc = fb; }
  for (int i = 0; i < N; i++) Xc[i] = dalpha[i] * xf + s->alpha[i];
  memcpy(s->alpha, Xc, sizeof(double) * N);
  (void)psi_eval(s, Xc, Fv); /* :380 leaves Fv, lp, dlp, d2lp at the new Alpha */
  free(Xc);
  *xmin_out = xf;
  return fc;
}

static void state_alloc(gp_state *s, int N, const double *K, const double *y, double sn2,
                        double *alpha) {
  memset(s, 0, sizeof(*s));
  s->N = N; s->K = K; s->y = y; s->sn2 = sn2; s->alpha = alpha;
  s->mvmK = (double *)malloc(sizeof(double) * N);
  s->lp = (double *)malloc(sizeof(double) * N);
  s->dlp = (double *)malloc(sizeof(double) * N);
  s->d2lp = (double *)malloc(sizeof(double) * N);
  s->Lchol = (double *)malloc(sizeof(double) * (size_t)N * N);
}
static void state_free(gp_state *s) {
  free(s->mvmK); free(s->lp); free(s->dlp); free(s->d2lp); free(s->Lchol);
}

/* tail of logLikelihood(), GP_Utils.cpp:1147-1160 */
static void nlz_tail(gp_state *s, double *Lout, orc_nlz_info *info) {
  int N = s->N;
  double *yhat = (double *)malloc(sizeof(double) * N);
  gemv_full(N, s->K, s->alpha, s->mvmK); s->n_gemv++;   /* :1147 */
  memcpy(yhat, s->mvmK, sizeof(double) * N);             /* :1148 */
  lik_update(s, yhat);                                    /* :1152 (updatelikelihood() :795-817) */
  form_and_factor_B(s, s->d2lp);                          /* :1154 ldB2_exact() */
  info->chol_fail = s->chol_fail;
  info->n_chol = s->n_chol; info->n_gemv = s->n_gemv;
  if (s->chol_fail) { info->nlz = NAN; free(yhat); return; }
  double ld = 0.0, q = 0.0, slp = 0.0;
  for (int i = 0; i < N; i++) ld += log(s->Lchol[i + (size_t)i * N]); /* :913 */
  for (int i = 0; i < N; i++) { q += s->alpha[i] * (0.5 * yhat[i]); slp += s->lp[i]; }
  info->logdet = ld; info->quad = q; info->sumlp = slp;
  info->nlz = q - slp + ld;                               /* :1159 */
  if (Lout) {
    for (int j = 0; j < N; j++)
      for (int i = 0; i < N; i++)
        Lout[i + (size_t)j * N] = i >= j ? s->Lchol[i + (size_t)j * N] : 0.0;
  }
  free(yhat);
}

void orc_nlz_refseq(int N, const double *K, const double *y, double sn2, double *alpha,
                    double *Lout, orc_nlz_info *info) {
  gp_state s;
  state_alloc(&s, N, K, y, sn2, alpha);
  memset(info, 0, sizeof(*info));
  double *Fv = (double *)calloc(N, sizeof(double));
  double *B = (double *)malloc(sizeof(double) * N);
  double *dalpha = (double *)malloc(sizeof(double) * N);
  double *rhs = (double *)malloc(sizeof(double) * N);
  /* irls, GP_Utils.cpp:191-228 */
  const int maxit = 20;
  const double tol = 1e-6;
  double psi_new = psi_eval(&s, s.alpha, Fv);
  double psi_old = INFINITY;
  int it = 0;
  double step = 0.0;
  while ((psi_old - psi_new) > tol && it < maxit) {
    psi_old = psi_new;
    it++;
    for (int i = 0; i < N; i++) B[i] = Fv[i] * s.d2lp[i] + s.dlp[i]; /* :214-216, mf = 0 */
    gemv_full(N, K, B, s.mvmK); s.n_gemv++;                          /* :217 */
    memcpy(rhs, s.mvmK, sizeof(double) * N);
    ldB2_solve(&s, s.d2lp, rhs, dalpha);                             /* :218 */
    if (s.chol_fail) break;                                          /* :219-222 */
    for (int i = 0; i < N; i++) dalpha[i] = -1.0 * dalpha[i] - s.alpha[i] + B[i]; /* :223 */
    psi_new = brentmin(&s, Fv, dalpha, &step);                       /* :225-226 */
  }
  info->irls_iters = it;
  info->last_step = step;
  if (s.chol_fail) {
    info->chol_fail = 1; info->nlz = NAN; info->n_chol = s.n_chol; info->n_gemv = s.n_gemv;
  } else {
    nlz_tail(&s, Lout, info);
  }
  free(Fv); free(B); free(dalpha); free(rhs);
  state_free(&s);
}

void orc_nlz_lean(int N, const double *K, const double *y, double sn2, double *alpha,
                  double *Lout, orc_nlz_info *info) {
  gp_state s;
  state_alloc(&s, N, K, y, sn2, alpha);
  memset(info, 0, sizeof(*info));
  for (int i = 0; i < N; i++) s.d2lp[i] = 1.0 / sn2;
  form_and_factor_B(&s, s.d2lp);
  info->chol_fail = s.chol_fail; info->n_chol = s.n_chol;
  if (s.chol_fail) { info->nlz = NAN; state_free(&s); return; }
  /* alpha = (K + sn2 I)^-1 y = B^-1 y / sn2 */
  for (int i = 0; i < N; i++) alpha[i] = y[i] / sn2;
  orc_solve_chol(N, s.Lchol, N, alpha, 1, N);
  double *yhat = (double *)malloc(sizeof(double) * N);
  gemv_full(N, K, alpha, yhat); s.n_gemv++;
  lik_update(&s, yhat);
  double ld = 0.0, q = 0.0, slp = 0.0;
  for (int i = 0; i < N; i++) ld += log(s.Lchol[i + (size_t)i * N]);
  for (int i = 0; i < N; i++) { q += alpha[i] * (0.5 * yhat[i]); slp += s.lp[i]; }
  info->logdet = ld; info->quad = q; info->sumlp = slp; info->nlz = q - slp + ld;
  info->n_gemv = s.n_gemv;
  if (Lout)
    for (int j = 0; j < N; j++)
      for (int i = 0; i < N; i++)
        Lout[i + (size_t)j * N] = i >= j ? s.Lchol[i + (size_t)j * N] : 0.0;
  free(yhat);
  state_free(&s);
}

/* ------------------------------------------------------------------------- */
/* Prediction, GP_Utils.cpp:943-1043                                         */
/* ------------------------------------------------------------------------- */
void orc_predict(const double *Xtr, int N, const double *Xte, int M, int d, const double *expans,
                 double bias, double sn2, int mode, const double *alpha, const double *L,
                 int compat_flags, double *mean, double *var) {
  double *kX = (double *)malloc(sizeof(double) * (size_t)N * M);
  orc_gram(Xtr, N, Xte, M, d, expans, bias, mode, kX, 0); /* _ComputeK_NewData :943-949 */
  for (int i = 0; i < M; i++) {                            /* _postMean :958-972 */
    double s = 0.0;
    for (int r = 0; r < N; r++) s += alpha[r] * kX[r + (size_t)i * N];
    mean[i] = s;
  }
  if (var) {
    double kD = orc_kdiag(expans, bias);                   /* _ComputeDiag_NewData :951-956 */
    double Wh = sqrt(1.0 / sn2);                           /* :985 */
    double *LKs = (double *)malloc(sizeof(double) * (size_t)N * M);
    for (size_t t = 0; t < (size_t)N * M; t++) LKs[t] = kX[t] * Wh; /* :986-990 */
    orc_solve_chol(N, L, N, LKs, M, N);                    /* :991 */
    for (int i = 0; i < M; i++) {
      double s = 0.0;
      for (int r = 0; r < N; r++) s += LKs[r + (size_t)i * N] * Wh * kX[r + (size_t)i * N]; /* :993-999 */
      var[i] = kD - s;
    }
    if (compat_flags & ORC_COMPAT_VARCLAMP) {
      /* :1002-1003: the 0/1 comparison mask is used as an index list */
      int any_neg = 0, any_nonneg = 0;
      for (int i = 0; i < M; i++) { if (var[i] < 0) any_neg = 1; else any_nonneg = 1; }
      if (any_nonneg && M > 0) var[0] = 0.0;
      if (any_neg && M > 1) var[1] = 0.0;
    } else {
      for (int i = 0; i < M; i++) if (var[i] < 0) var[i] = 0.0;
    }
    if (!((compat_flags & ORC_COMPAT_SN2SKIP) && sn2 == 1.0)) /* :1036-1040 */
      for (int i = 0; i < M; i++) var[i] += sn2;
    free(LKs);
  }
  free(kX);
}

/* ------------------------------------------------------------------------- */
/* Reference-style "gradient", GP_Utils.cpp:1164-1284 + Kernel.cpp:886-1263  */
/* ------------------------------------------------------------------------- */
static void expans_S_matrices(const double *e, double S[9], double Sp[6][9]) {
  /* Kernel.cpp:955-1166: Rot and its three angle derivatives, S = Rot diag(iw) Rot',
   * S_alpha/S_beta/S_teta "as written" (the (0,0) z-term lacks its factor 2, :1003-1011)
   * and S_Lalpha/S_Lbeta/S_Lteta (column outer products). */
  double al = e[0], be = e[2], te = e[4];
  double iw[3] = {e[1], e[3], e[5]};
  double ca = cos(al), sa = sin(al), cb = cos(be), sb = sin(be), ct = cos(te), st = sin(te);
  double R[3][3], Ra[3][3], Rb[3][3], Rt[3][3];
  R[0][0] = ca * ct + sa * sb * st;   Ra[0][0] = -sa * ct + ca * sb * st;
  Rb[0][0] = sa * cb * st;            Rt[0][0] = -ca * st + sa * sb * ct;
  R[0][1] = -sa * ct + ca * sb * st;  Ra[0][1] = -ca * ct - sa * sb * st;
  Rb[0][1] = ca * cb * st;            Rt[0][1] = sa * st + ca * sb * ct;
  R[0][2] = -cb * st;                 Ra[0][2] = 0.0;
  Rb[0][2] = sb * st;                 Rt[0][2] = -cb * ct;
  R[1][0] = sa * cb;                  Ra[1][0] = ca * cb;
  Rb[1][0] = -sa * sb;                Rt[1][0] = 0.0;
  R[1][1] = ca * cb;                  Ra[1][1] = -sa * cb;
  Rb[1][1] = -ca * sb;                Rt[1][1] = 0.0;
  R[1][2] = sb;                       Ra[1][2] = 0.0;
  Rb[1][2] = cb;                      Rt[1][2] = 0.0;
  R[2][0] = ca * st - sa * sb * ct;   Ra[2][0] = -sa * st - ca * sb * ct;
  Rb[2][0] = -sa * cb * ct;           Rt[2][0] = ca * ct + sa * sb * st;
  R[2][1] = -sa * st - ca * sb * ct;  Ra[2][1] = -ca * st + sa * sb * ct;
  Rb[2][1] = -ca * cb * ct;           Rt[2][1] = -sa * ct + ca * sb * st;
  R[2][2] = cb * ct;                  Ra[2][2] = 0.0;
  Rb[2][2] = -sb * ct;                Rt[2][2] = -cb * st;
  double (*D[3])[3] = {Ra, Rb, Rt};
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) {
      double s = 0.0;
      for (int k = 0; k < 3; k++) s += iw[k] * R[r][k] * R[c][k];
      S[r + 3 * c] = s;
      for (int a = 0; a < 3; a++) {
        double t = 0.0;
        for (int k = 0; k < 3; k++) {
          double term = iw[k] * (D[a][r][k] * R[c][k] + R[r][k] * D[a][c][k]);
          if (r == 0 && c == 0 && k == 2) term *= 0.5; /* :1005,1008,1011 missing "2 *" */
          t += term;
        }
        Sp[2 * a][r + 3 * c] = t;                       /* S_alpha, S_beta, S_teta  -> g0,g2,g4 */
        Sp[2 * a + 1][r + 3 * c] = R[r][a] * R[c][a];   /* S_Lalpha, S_Lbeta, S_Lteta -> g1,g3,g5 */
      }
    }
}

void orc_grad_ref(const double *X, int N, const double *y, const double *K, const double *L,
                  const double *alpha, const double *expans, double bias, double sn2, int mode,
                  double *g) {
  orc_grad_ref_d(X, N, 3, y, K, L, alpha, expans, bias, sn2, mode, g);
}

/* The same for d = 3 or d = 4 input columns.  With a 4th ("rock type") column: the distance carries it
 * (A33 = InversewidthR, Kernel.cpp:1411-1424), S(3,3) = 1 but every S_p(3,3) = 0, so g0..g5 see only the
 * first three columns (:1169-1173), and g7 = -2 * sum(KD2 % Di2_R) / N with Di2_R = 2 (x4_i - x4_j)^2 --
 * the weight there is KD2 = exp(-sqrt(DD2)), NOT R: RColon still holds KD2 from the Sigma block
 * (:1239-1253), reproduced as written. */
void orc_grad_ref_d(const double *X, int N, int d, const double *y, const double *K, const double *L,
                    const double *alpha, const double *expans, double bias, double sn2, int mode,
                    double *g) {
  (void)bias;
  size_t NN = (size_t)N * N;
  double Sw = sqrt(1.0 / sn2);
  /* GradLL :1202-1206:  Q = solve_chol(Lchol, diag(Sw)); Q %= (1/Sw) 1' ; dW = 0.5 sum(Q % K, 1) */
  double *Q = (double *)calloc(NN, sizeof(double));
  for (int i = 0; i < N; i++) Q[i + (size_t)i * N] = Sw;
  orc_solve_chol(N, L, N, Q, N, N);
  for (int j = 0; j < N; j++)
    for (int i = 0; i < N; i++) Q[i + (size_t)j * N] *= (1.0 / Sw);
  double *dW = (double *)calloc(N, sizeof(double));
  for (int j = 0; j < N; j++)
    for (int i = 0; i < N; i++) dW[i] += 0.5 * Q[i + (size_t)j * N] * K[i + (size_t)j * N];
  /* dfhat = dW % d3lp = 0, dahat = 0 (:1210-1219, d3lp.zeros() :414) */
  /* dhyp :1164-1169:  QW = Q % (d2lp 1') - Alpha Alpha' + 2 dlp dahat' */
  double *QW = (double *)malloc(sizeof(double) * NN);
  for (int j = 0; j < N; j++)
    for (int i = 0; i < N; i++)
      QW[i + (size_t)j * N] = Q[i + (size_t)j * N] * (1.0 / sn2) - alpha[i] * alpha[j];
  /* Kern_ExpAnisotropic::getGradients, Kernel.cpp:886-1263 */
  double par[7];
  pack_paramker(expans, par);
  double *DD2 = (double *)malloc(sizeof(double) * NN);
  orc_mahadist(X, N, X, N, d, par, mode, DD2);              /* :925 */
  double var2 = expans[6] * expans[6];
  double S[9], Sp[6][9];
  expans_S_matrices(expans, S, Sp);
  double *R = (double *)malloc(sizeof(double) * NN);
  double gsig = 0.0;
  for (int j = 0; j < N; j++)
    for (int i = 0; i < N; i++) {
      size_t t = i + (size_t)j * N;
      double sd = sqrt(DD2[t]);                             /* :1178 */
      double kd2 = exp(-1.0 * sd);                          /* :1176 */
      double dk = sd == 0 ? 0.0 : kd2 * (-0.5 / sd);        /* :1179-1183 */
      if (i == j) dk = 0.0;                                 /* :1184 */
      R[t] = var2 * QW[t] * dk;                             /* :927,1185 */
      gsig += kd2 * QW[t];                                  /* :1239-1241 */
    }
  for (int p = 0; p < 6; p++) {
    double M[9];
    for (int t = 0; t < 9; t++) M[t] = S[t] * Sp[p][t];     /* S % S_p */
    /* Di2 = sum(2 (X%X) M,1) 1' + 1 sum(2 (X%X) M,1)' - 4 X M X'   (:1192-1194) */
    double *a = (double *)malloc(sizeof(double) * N);
    double *XM = (double *)malloc(sizeof(double) * (size_t)N * 3);
    for (int i = 0; i < N; i++) {
      double s = 0.0;
      for (int c = 0; c < 3; c++) {
        double t = 0.0, u = 0.0;
        for (int k = 0; k < 3; k++) {
          double x = X[i + (size_t)k * N];
          t += 2.0 * x * x * M[k + 3 * c];
          u += x * M[k + 3 * c];
        }
        s += t;
        XM[i + (size_t)c * N] = u;
      }
      a[i] = s;
    }
    double acc = 0.0;
    for (int j = 0; j < N; j++)
      for (int i = 0; i < N; i++) {
        double dot = 0.0;
        for (int c = 0; c < 3; c++) dot += XM[i + (size_t)c * N] * X[j + (size_t)c * N];
        double Di2 = a[i] + a[j] - 4.0 * dot;
        acc += R[i + (size_t)j * N] * Di2;                  /* :1195-1197 */
      }
    g[p] = acc;
    free(a); free(XM);
  }
  g[6] = 2.0 * gsig * expans[6];                            /* :1241-1242 */
  g[7] = 0.0;                                               /* :1256-1257 */
  if (d == 4) {                                             /* :1246-1255 */
    const double *x4 = X + (size_t)3 * N;
    double acc = 0.0;
    for (int j = 0; j < N; j++)
      for (int i = 0; i < N; i++) {
        double Di2 = 2.0 * x4[i] * x4[i] + 2.0 * x4[j] * x4[j] - 4.0 * x4[i] * x4[j];
        acc += exp(-1.0 * sqrt(DD2[i + (size_t)j * N])) * Di2;
      }
    g[7] = -2.0 * acc / N;
  }
  /* Kern_Bias::getGradients, Kernel.cpp:370-377: sum(QW % eye) */
  double tr = 0.0;
  for (int i = 0; i < N; i++) tr += QW[i + (size_t)i * N];
  g[8] = tr;
  /* updateGlikelihood :846-864 and :1222-1235 */
  double *yhat = (double *)malloc(sizeof(double) * N);
  gemv_full(N, K, alpha, yhat);
  double sdW = 0.0, slp = 0.0;
  for (int i = 0; i < N; i++) {
    double ymmu = y[i] - yhat[i];
    sdW += dW[i] * (2.0 / sn2);
    slp += (1.0 / sn2) * ymmu * ymmu - 1.0;
  }
  g[9] = -1.0 * sdW - slp;
  free(yhat); free(R); free(DD2); free(QW); free(dW); free(Q);
}


/* The same sums as orc_grad_ref_d, STREAMED: the caller supplies Q = B^-1 (what GradLL :1202-1206 forms with
 * solve_chol; here typically LAPACK's) and alpha; K, DD2, QW and R are rebuilt one slab of columns at a time, so
 * the only N x N array is Q (N = 32768: 8.6 GB instead of the six of orc_grad_ref_d).  Used by
 * tests/golden/make_golden_grad.py for the gradient golden at the metric's size.  Per slab the distances come
 * from orc_mahadist(X, X[slab]): in DIRECT mode these are the full call's numbers up to the rounding of the
 * centring (the pooled mean is per slab); the EXPANSION form's cancellation noise would differ per slab, so the
 * golden uses DIRECT.  Every formula below is the line of orc_grad_ref_d with the same citation. */
void orc_grad_ref_q(const double *X, int N, int d, const double *y, const double *Q, size_t ldq,
                    const double *alpha, const double *expans, double bias, double sn2, int mode, double *g) {
  enum { W = 256 };
  double par[7];
  pack_paramker(expans, par);
  double var2 = expans[6] * expans[6];
  double S[9], Sp[6][9], M[6][9];
  expans_S_matrices(expans, S, Sp);
  for (int p = 0; p < 6; p++)
    for (int t = 0; t < 9; t++) M[p][t] = S[t] * Sp[p][t];            /* S % S_p */
  /* a_i^(p) = sum(2 (X%X) M_p, 1), XM^(p) = X M_p   (:1192-1194) */
  double *a = (double *)malloc(sizeof(double) * 6 * (size_t)N);
  double *XM = (double *)malloc(sizeof(double) * 18 * (size_t)N);
  for (int p = 0; p < 6; p++)
    for (int i = 0; i < N; i++) {
      double s = 0.0;
      for (int c = 0; c < 3; c++) {
        double t = 0.0, u = 0.0;
        for (int k = 0; k < 3; k++) {
          double x = X[i + (size_t)k * N];
          t += 2.0 * x * x * M[p][k + 3 * c];
          u += x * M[p][k + 3 * c];
        }
        s += t;
        XM[((size_t)p * 3 + c) * N + i] = u;
      }
      a[(size_t)p * N + i] = s;
    }
  double *D = (double *)malloc(sizeof(double) * (size_t)N * W);
  double *Xs = (double *)malloc(sizeof(double) * (size_t)W * d);
  double *yhat = (double *)calloc(N, sizeof(double));
  double gp[6] = {0, 0, 0, 0, 0, 0}, gsig = 0.0, g7 = 0.0, tr = 0.0, sdW = 0.0;
  for (int j0 = 0; j0 < N; j0 += W) {
    int w = N - j0 < W ? N - j0 : W;
    for (int c = 0; c < d; c++)
      for (int j = 0; j < w; j++) Xs[j + (size_t)c * w] = X[j0 + j + (size_t)c * N];
    orc_mahadist(X, N, Xs, w, d, par, mode, D);                       /* :925 */
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0, s5 = 0, ssig = 0, s7 = 0, sw = 0;
#pragma omp parallel for schedule(static) reduction(+ : s0, s1, s2, s3, s4, s5, ssig, s7, sw)
    for (int j = 0; j < w; j++) {
      int gj = j0 + j;
      const double *q = Q + (size_t)gj * ldq;
      for (int i = 0; i < N; i++) {
        double sd = sqrt(D[i + (size_t)j * N]);                       /* :1178 */
        double kd2 = exp(-1.0 * sd);                                  /* :1176 */
        double k = var2 * kd2 + bias;                                 /* :881, :366 */
        double qw = q[i] * (1.0 / sn2) - alpha[i] * alpha[gj];        /* dhyp :1164-1169 */
        double dk = sd == 0 ? 0.0 : kd2 * (-0.5 / sd);                /* :1179-1183 */
        if (i == gj) dk = 0.0;                                        /* :1184 */
        double r = var2 * qw * dk;                                    /* :927, 1185 */
        sw += 0.5 * q[i] * k;                                         /* dW :1206 (summed: only sum(dW) is used) */
        ssig += kd2 * qw;                                             /* :1239-1241 */
        double acc[6];
        for (int p = 0; p < 6; p++) {
          const double *xm = XM + (size_t)p * 3 * N;
          double dot = xm[i] * X[gj] + xm[i + (size_t)N] * X[gj + (size_t)N] +
                       xm[i + 2 * (size_t)N] * X[gj + 2 * (size_t)N];
          acc[p] = r * (a[(size_t)p * N + i] + a[(size_t)p * N + gj] - 4.0 * dot);   /* :1195-1197 */
        }
        s0 += acc[0]; s1 += acc[1]; s2 += acc[2]; s3 += acc[3]; s4 += acc[4]; s5 += acc[5];
        if (d == 4) {                                                 /* :1246-1255, weight KD2 as written */
          const double *x4 = X + (size_t)3 * N;
          s7 += kd2 * (2.0 * x4[i] * x4[i] + 2.0 * x4[gj] * x4[gj] - 4.0 * x4[i] * x4[gj]);
        }
      }
    }
    gp[0] += s0; gp[1] += s1; gp[2] += s2; gp[3] += s3; gp[4] += s4; gp[5] += s5;
    gsig += ssig; g7 += s7; sdW += sw;
    /* yhat += K[:, slab] alpha[slab]   (mvmK_exact, GP_Utils.cpp:394-397) */
#pragma omp parallel for schedule(static)
    for (int i = 0; i < N; i++) {
      double s = 0.0;
      for (int j = 0; j < w; j++) s += (var2 * exp(-1.0 * sqrt(D[i + (size_t)j * N])) + bias) * alpha[j0 + j];
      yhat[i] += s;
    }
  }
  for (int i = 0; i < N; i++) tr += Q[i + (size_t)i * ldq] * (1.0 / sn2) - alpha[i] * alpha[i];
  for (int p = 0; p < 6; p++) g[p] = gp[p];
  g[6] = 2.0 * gsig * expans[6];                                      /* :1241-1242 */
  g[7] = d == 4 ? -2.0 * g7 / N : 0.0;                                /* :1246-1257 */
  g[8] = tr;                                                          /* Kern_Bias::getGradients :370-377 */
  double slp = 0.0;
  for (int i = 0; i < N; i++) {
    double ymmu = y[i] - yhat[i];
    slp += (1.0 / sn2) * ymmu * ymmu - 1.0;                           /* :1226-1234 */
  }
  g[9] = -1.0 * sdW * (2.0 / sn2) - slp;                              /* updateGlikelihood :846-864 */
  free(yhat); free(Xs); free(D); free(XM); free(a);
}

/* GradLL for an arbitrary HybKerns composition (GP_Utils.cpp:1171-1284): the children's
 * getGradients as written -- Kern_ExpAnisotropic (Kernel.cpp:886-1263, recomputes its own
 * MahaDist), Kern_RBF (:491-540) and Kern_Exponential (:644-693), which both work on the D2
 * ARGUMENT, i.e. GP_utils' member D2 = the SUM of the children's D2 (Kernel.cpp:151) --
 * Kern_Bias (:370-377), then the likelihood hyper-parameter (:1222-1235).
 * g: per child in order (8 / 2 / 3 entries), then bias (if has_bias), then sn2. */
void orc_grad_hyb(const double *X, int N, const double *y, const double *K, const double *L, const double *alpha,
                  int nterms, const int *kinds, const double *pars, int has_bias, double sn2, int mode, double *g) {
  orc_grad_hyb_d(X, N, 3, y, K, L, alpha, nterms, kinds, pars, has_bias, sn2, mode, g);
}

/* d = 3 or 4 input columns (the 4th enters every child's distance; ExpAns' g7 as in orc_grad_ref_d) */
void orc_grad_hyb_d(const double *X, int N, int d, const double *y, const double *K, const double *L,
                    const double *alpha, int nterms, const int *kinds, const double *pars, int has_bias, double sn2,
                    int mode, double *g) {
  size_t NN = (size_t)N * N;
  double Sw = sqrt(1.0 / sn2);
  double *Q = (double *)calloc(NN, sizeof(double));
  for (int i = 0; i < N; i++) Q[i + (size_t)i * N] = Sw;
  orc_solve_chol(N, L, N, Q, N, N);
  for (size_t t = 0; t < NN; t++) Q[t] *= (1.0 / Sw);
  double *QW = (double *)malloc(sizeof(double) * NN);
  double sdW = 0.0;
  for (int j = 0; j < N; j++)
    for (int i = 0; i < N; i++) {
      size_t t = i + (size_t)j * N;
      sdW += 0.5 * Q[t] * K[t];
      QW[t] = Q[t] * (1.0 / sn2) - alpha[i] * alpha[j];
    }
  /* member D2 of GP_utils: sum of the children's D2 */
  double *D2s = (double *)calloc(NN, sizeof(double)), *D = (double *)malloc(sizeof(double) * NN);
  const double *p = pars;
  for (int t = 0; t < nterms; t++) {
    if (kinds[t] == 0) { double par[7]; pack_paramker(p, par); orc_mahadist(X, N, X, N, d, par, mode, D); p += 8; }
    else if (kinds[t] == 1) { orc_eucldist(X, N, X, N, d, p[0], mode, D); p += 2; }
    else { orc_eucldist(X, N, X, N, d, p[0], mode, D); p += 3; }
    for (size_t e = 0; e < NN; e++) D2s[e] += D[e];
  }
  int go = 0;
  p = pars;
  for (int t = 0; t < nterms; t++) {
    if (kinds[t] == 0) {
      /* reuse the ExpAns restatement on QW directly */
      double par[7];
      pack_paramker(p, par);
      orc_mahadist(X, N, X, N, d, par, mode, D);
      double var2 = p[6] * p[6], S[9], Sp[6][9];
      expans_S_matrices(p, S, Sp);
      double *R = (double *)malloc(sizeof(double) * NN);
      double gsig = 0.0;
      for (int j = 0; j < N; j++)
        for (int i = 0; i < N; i++) {
          size_t e = i + (size_t)j * N;
          double sd = sqrt(D[e]), kd2 = exp(-1.0 * sd);
          double dk = sd == 0 ? 0.0 : kd2 * (-0.5 / sd);
          if (i == j) dk = 0.0;
          R[e] = var2 * QW[e] * dk;
          gsig += kd2 * QW[e];
        }
      for (int q = 0; q < 6; q++) {
        double M[9];
        for (int e = 0; e < 9; e++) M[e] = S[e] * Sp[q][e];
        double acc = 0.0;
        for (int j = 0; j < N; j++)
          for (int i = 0; i < N; i++) {
            double ai = 0.0, aj = 0.0, dot = 0.0;
            for (int c = 0; c < 3; c++) {
              double u = 0.0;
              for (int k = 0; k < 3; k++) {
                double xi = X[i + (size_t)k * N], xj = X[j + (size_t)k * N];
                ai += 2.0 * xi * xi * M[k + 3 * c];
                aj += 2.0 * xj * xj * M[k + 3 * c];
                u += xi * M[k + 3 * c];
              }
              dot += u * X[j + (size_t)c * N];
            }
            acc += R[i + (size_t)j * N] * (ai + aj - 4.0 * dot);
          }
        g[go + q] = acc;
      }
      g[go + 6] = 2.0 * gsig * p[6];
      g[go + 7] = 0.0;
      if (d == 4) {                                         /* Kernel.cpp:1246-1255, weight KD2 as written */
        const double *x4 = X + (size_t)3 * N;
        double acc = 0.0;
        for (int j = 0; j < N; j++)
          for (int i = 0; i < N; i++) {
            double dx = x4[i] - x4[j];
            acc += exp(-1.0 * sqrt(D[i + (size_t)j * N])) * 2.0 * dx * dx;
          }
        g[go + 7] = -2.0 * acc / N;
      }
      free(R);
      go += 8; p += 8;
    } else if (kinds[t] == 1) {   /* Kern_Exponential::getGradients, Kernel.cpp:644-693 */
      double var2 = p[1] * p[1], g0 = 0.0, g1 = 0.0;
      for (int j = 0; j < N; j++)
        for (int i = 0; i < N; i++) {
          size_t e = i + (size_t)j * N;
          double sd = sqrt(D2s[e]), kd2 = exp(-1.0 * sd);
          double dk = (i == j || sd == 0) ? 0.0 : kd2 * (-0.5 / sd);   /* :665-669 (diag filled with 0) */
          g0 += var2 * QW[e] * dk * D2s[e];                            /* :671-680 */
          g1 += (QW[e] * kd2) * kd2;                                   /* :682-690: (Q % KD2) . KD2 */
        }
      g[go] = g0; g[go + 1] = g1 * p[1];
      go += 2; p += 2;
    } else {                      /* Kern_RBF::getGradients, Kernel.cpp:491-540 */
      double var2 = p[2] * p[2], iw = p[1], s1 = 0.0, s2 = 0.0, s3 = 0.0;
      for (size_t e = 0; e < NN; e++) {
        double kd2 = exp(-0.5 * iw * D2s[e]);
        double dk = exp(-iw / 2 * D2s[e]) * (-iw / 2);
        s1 += var2 * QW[e] * dk * D2s[e];            /* R . D2 */
        s2 += -0.5 * (var2 * QW[e] * kd2) * D2s[e];  /* Qwidth */
        s3 += QW[e] * kd2;
      }
      g[go] = (-2.0 * s1) / 2;                       /* :518, :536 */
      g[go + 1] = s2 / 2;                            /* :521, :537 */
      g[go + 2] = ((s3 * p[2] + s3 * p[2]) * p[2]) / 2;   /* :523-531, :538 */
      go += 3; p += 3;
    }
  }
  if (has_bias) {
    double tr = 0.0;
    for (int i = 0; i < N; i++) tr += QW[i + (size_t)i * N];
    g[go++] = tr;
  }
  double *yhat = (double *)malloc(sizeof(double) * N);
  gemv_full(N, K, alpha, yhat);
  double slp = 0.0;
  for (int i = 0; i < N; i++) { double r = y[i] - yhat[i]; slp += (1.0 / sn2) * r * r - 1.0; }
  g[go] = -1.0 * sdW * (2.0 / sn2) - slp;
  free(yhat); free(D); free(D2s); free(QW); free(Q);
}
