// grad.hip -- the reference-style "gradient" of the negative log marginal likelihood on gfx950.
//
// Replaces GP_utils::GradLL / dhyp / updateG / updateGlikelihood (GP_Utils.cpp:846-864,
// 1164-1284) with Kern_ExpAnisotropic::getGradients (Kernel.cpp:886-1263) and
// Kern_Bias::getGradients (:370-377), FORMULAS AS WRITTEN (this is not the
// true gradient -- the optimiser trajectory of the reference depends on it as it is).
//
// Reference: Q = solve_chol(Lchol, diag(sW)) with N right-hand sides (2N^3 flops), ~15 N x N
// temporaries, six N x 3 * 3 x N GEMMs.  Here, for the Gaussian likelihood (d3lp = 0, so
// dfhat = dahat = 0, GP_Utils.cpp:414, 1210-1219):
//   1. G = L^-T by blocked forward substitution on the identity, touching only the rows that
//      can be non-zero (N^3/3 flops, MFMA);
//   2. B^-1 = G G^T, lower tiles, k-loop started at the row tile (N^3/3 flops, MFMA);
//   3. ONE fused pass over the pairs i >= j that recomputes K_ij from the coordinates and
//      accumulates the nine sums the ten gradient entries are made of.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "../../include/gpak_dev.h"
#include "gpak_internal.h"

#define PB 128
#define GT_ROWS 128
#define GT_COLS 64
#define NSUM 16  // 9 shared/ExpAns sums + 2 per stationary term for the Exp / RBF children + the rock-type sum

struct GradConsts {
  double M[6][6];   // S % S_p, symmetric 3x3 stored as {00,01,02,11,12,22}, p = 0..5
  double m2[6][3];  // 2 * column sums of M_p : a_i^(p) = sum_k x_ik^2 m2[p][k]
  double var2, bias, sn2;
  int mode;
  int te;           // index of the ExpAns term (-1: none)
  int kinds[GPAK_MAX_TERMS];
};

__global__ void gpak_identity_f64(double *W, long ld, int n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t tot = (size_t)n * ld;
  if (i >= tot) return;
  const size_t c = i / ld, r = i - c * ld;
  W[i] = (r == c) ? 1.0 : 0.0;
}

// pair pass over the lower triangle; block partials: part[block][NSUM]
//   [0..5] sum Rm * Di2^(p)   [6] sum QW * exp(-sqrt(D_expans))   [7] sum Q * K   [8] trace(QW)
//   [9+2t], [10+2t]  the two sums of stationary term t when it is an Exp or RBF child; those
//   children work on GP_utils' member D2 = the SUM of the children's D2 (Kernel.cpp:151, 491-540, 644-693)
//   [15] sum exp(-sqrt(D_expans)) * (x4_i - x4_j)^2 for 4-column inputs (Kernel.cpp:1246-1255: the weight is
//        KD2, not R -- RColon still holds KD2 from the Sigma block, reproduced as written)
#define PARRG GPAK_PARR
__global__ __launch_bounds__(256) void gpak_grad_pairs_f64(
    const double *__restrict__ U, int cap, const double *__restrict__ x0, const double *__restrict__ x1,
    const double *__restrict__ x2, const double *__restrict__ x3, const double *__restrict__ alpha,
    const double *__restrict__ Binv, long ld, int N, KernParams kp, GradConsts gc, double *__restrict__ part,
    int rowP, int rowA, int Tmax) {
  // rowP > 0 (distributed gradient): this rank holds the B^-1 ROWS of the 128-row blocks g = t*rowP + rowA as a
  // compact (rows x Np) array whose 128-column groups are ordered by (g % rowP, g / rowP): see gpak_grad_binv_rows
  const int row0 = (rowP ? blockIdx.x * rowP + rowA : blockIdx.x) * GT_ROWS, col0 = blockIdx.y * GT_COLS;
  const int bid = blockIdx.y * gridDim.x + blockIdx.x;
  __shared__ double cq[GPAK_MAX_TERMS][GPAK_PT][GT_COLS];  // transformed column points, per term
  __shared__ double cx3[GT_COLS];                    // raw 4th column of the column points (0 for 3-D)
  __shared__ double cal[GT_COLS];                    // alpha of the column points
  __shared__ double cm[6][3][GT_COLS];               // M_p x_j
  __shared__ double ca[6][GT_COLS];                  // a_j^(p)
  __shared__ double red[4][NSUM];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int nterms = kp.nterms, te = gc.te;
  double acc[NSUM];
#pragma unroll
  for (int k = 0; k < NSUM; k++) acc[k] = 0.0;
  if (row0 + GT_ROWS > col0) {  // tile touches the lower triangle
    if (t < GT_COLS) {
      const int j = col0 + t;
      const bool ok = j < N;
      for (int m = 0; m < nterms; m++)
#pragma unroll
        for (int c = 0; c < GPAK_PT; c++) cq[m][c][t] = ok ? PARRG(U, cap, m, c)[j] : 0.0;
      cal[t] = ok ? alpha[j] : 0.0;
      cx3[t] = (ok && x3) ? x3[j] : 0.0;
      const double a = ok ? x0[j] : 0.0, b = ok ? x1[j] : 0.0, c = ok ? x2[j] : 0.0;
#pragma unroll
      for (int p = 0; p < 6; p++) {
        const double *M = gc.M[p];
        cm[p][0][t] = M[0] * a + M[1] * b + M[2] * c;
        cm[p][1][t] = M[1] * a + M[3] * b + M[4] * c;
        cm[p][2][t] = M[2] * a + M[4] * b + M[5] * c;
        ca[p][t] = a * a * gc.m2[p][0] + b * b * gc.m2[p][1] + c * c * gc.m2[p][2];
      }
    }
    __syncthreads();
    const int r = row0 + 2 * lane;
    // where row r / column j live in Binv
    const long rloc = rowP ? (long)blockIdx.x * GT_ROWS + 2 * lane : (long)r;
    long cgrp = 0;
    if (rowP) {
      const int g = col0 / PB;
      cgrp = ((long)(g % rowP) * Tmax + g / rowP) * PB - (long)g * PB;   // added to j: the permuted column
    }
    double pu[2][GPAK_MAX_TERMS][GPAK_PT], px[2][3], pa[2][6], pal[2], px3[2];
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const int i = r + h;
      const bool ok = i < N;
      for (int m = 0; m < nterms; m++)
#pragma unroll
        for (int c = 0; c < GPAK_PT; c++) pu[h][m][c] = ok ? PARRG(U, cap, m, c)[i] : 0.0;
      px3[h] = (ok && x3) ? x3[i] : 0.0;
      px[h][0] = ok ? x0[i] : 0.0; px[h][1] = ok ? x1[i] : 0.0; px[h][2] = ok ? x2[i] : 0.0;
      pal[h] = ok ? alpha[i] : 0.0;
#pragma unroll
      for (int p = 0; p < 6; p++)
        pa[h][p] = px[h][0] * px[h][0] * gc.m2[p][0] + px[h][1] * px[h][1] * gc.m2[p][1] +
                   px[h][2] * px[h][2] * gc.m2[p][2];
    }
    for (int c = 0; c < GT_COLS / 4; c++) {
      const int jl = w + 4 * c, j = col0 + jl;
      if (j >= N) continue;
      const double2 q2 = *reinterpret_cast<const double2 *>(Binv + rloc + (size_t)(j + cgrp) * ld);
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const int i = r + h;
        if (i >= N || i < j) continue;
        const double q = h ? q2.y : q2.x;
        const double wgt = (i == j) ? 1.0 : 2.0;  // every summand is symmetric in (i, j)
        double d2m[GPAK_MAX_TERMS], d2s = 0.0, kfull = gc.bias;
#pragma unroll
        for (int m = 0; m < GPAK_MAX_TERMS; m++) {
          if (m >= nterms) break;
          double d2;
          if (gc.mode == GPAK_DIST_DIRECT) {
            const double a = pu[h][m][0] - cq[m][0][jl], b = pu[h][m][1] - cq[m][1][jl], cc = pu[h][m][2] - cq[m][2][jl];
            const double e4 = pu[h][m][4] - cq[m][4][jl];
            d2 = a * a + b * b + cc * cc + e4 * e4;
          } else {
            const double dot = pu[h][m][0] * cq[m][0][jl] + pu[h][m][1] * cq[m][1][jl] + pu[h][m][2] * cq[m][2][jl] +
                               pu[h][m][4] * cq[m][4][jl];
            d2 = pu[h][m][3] + cq[m][3][jl] - 2.0 * dot;
            d2 = d2 < 0.0 ? 0.0 : d2;
          }
          d2m[m] = d2;
          d2s += d2;
          kfull += kp.term[m].var2 * gpak_exp_nonpos(kp.term[m].profile == GPAK_PROFILE_RBF ? -0.5 * kp.term[m].iw * d2
                                                                                            : -gpak_sqrt_nonneg(d2));
        }
        const double qw = q * (1.0 / gc.sn2) - pal[h] * cal[jl];   // dhyp, GP_Utils.cpp:1168
        acc[7] = fma(wgt * q, kfull, acc[7]);                       // sum(Q % K), GP_Utils.cpp:1206
        if (i == j) acc[8] += qw;                                   // trace(QW), Kernel.cpp:370-377
        if (te >= 0) {
          const double sd = gpak_sqrt_nonneg(d2m[te]);              // Kernel.cpp:1178
          const double ek = gpak_exp_nonpos(-sd);                   // KD2, :1176
          const double dk = (sd == 0.0 || i == j) ? 0.0 : ek * (-0.5 / sd);  // :1179-1184
          const double rm = gc.var2 * qw * dk;                      // R = Qs % dk, :927, :1185
#pragma unroll
          for (int p = 0; p < 6; p++) {
            const double xmx = px[h][0] * cm[p][0][jl] + px[h][1] * cm[p][1][jl] + px[h][2] * cm[p][2][jl];
            const double di2 = pa[h][p] + ca[p][jl] - 4.0 * xmx;    // Di2, :1192-1194
            acc[p] = fma(wgt * rm, di2, acc[p]);
          }
          acc[6] = fma(wgt * qw, ek, acc[6]);                       // :1239-1241
          const double dx4 = px3[h] - cx3[jl];
          acc[15] = fma(wgt * ek, dx4 * dx4, acc[15]);              // :1246-1253 (Di2_R = 2 dx4^2)
        }
#pragma unroll
        for (int m = 0; m < GPAK_MAX_TERMS; m++) {
          if (m >= nterms) break;
          if (gc.kinds[m] == GPAK_KERN_EXP) {                       // Kernel.cpp:644-693 on the summed D2
            const double sd = gpak_sqrt_nonneg(d2s), kd = gpak_exp_nonpos(-sd);
            const double dk = (sd == 0.0 || i == j) ? 0.0 : kd * (-0.5 / sd);
            acc[9 + 2 * m] = fma(wgt * qw * dk, d2s, acc[9 + 2 * m]);
            acc[10 + 2 * m] = fma(wgt * qw * kd, kd, acc[10 + 2 * m]);
          } else if (gc.kinds[m] == GPAK_KERN_RBF) {                // Kernel.cpp:491-540 on the summed D2
            const double kd = gpak_exp_nonpos(-0.5 * kp.term[m].iw * d2s);
            acc[9 + 2 * m] = fma(wgt * qw * kd, d2s, acc[9 + 2 * m]);
            acc[10 + 2 * m] = fma(wgt * qw, kd, acc[10 + 2 * m]);
          }
        }
      }
    }
  }
  // block reduction (fixed order)
#pragma unroll
  for (int k = 0; k < NSUM; k++) {
    double v = acc[k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    if (lane == 0) red[w][k] = v;
  }
  __syncthreads();
  if (t < NSUM) part[(size_t)bid * NSUM + t] = red[0][t] + red[1][t] + red[2][t] + red[3][t];
}

__global__ __launch_bounds__(256) void gpak_grad_reduce_f64(const double *__restrict__ part, int nblocks,
                                                            double *__restrict__ out) {
  __shared__ double sh[256];
  const int k = blockIdx.x;
  double s = 0.0;
  for (int b = threadIdx.x; b < nblocks; b += 256) s += part[(size_t)b * NSUM + k];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (threadIdx.x < st) sh[threadIdx.x] += sh[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[k] = sh[0];
}

// sum_i ((y_i - f_i)^2 / sn2 - 1)    (lp_dhyp, GP_Utils.cpp:858)
__global__ __launch_bounds__(1024) void gpak_lpdhyp_f64(int N, const double *__restrict__ y,
                                                         const double *__restrict__ f, double sn2, double *out) {
  __shared__ double sh[1024];
  double s = 0.0;
  for (int i = threadIdx.x; i < N; i += 1024) {
    const double d = y[i] - f[i];
    s += (1.0 / sn2) * d * d - 1.0;
  }
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int st = 512; st > 0; st >>= 1) {
    if (threadIdx.x < st) sh[threadIdx.x] += sh[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = sh[0];
}

void gpak_grad_release(gpak_ctx *ctx) {
  if (ctx->dG) hipFree(ctx->dG);
  if (ctx->dBinv) hipFree(ctx->dBinv);
  if (ctx->dGpart) hipFree(ctx->dGpart);
  ctx->dG = ctx->dBinv = ctx->dGpart = nullptr;
  ctx->gpart_elems = 0;
}

// S, S_alpha.. as written at Kernel.cpp:955-1166 (the (0,0) z-term of the angle derivatives lacks
// its factor 2, :1003-1011) and M_p = S % S_p
static void build_grad_consts(const double *e, GradConsts &gc) {
  const double al = e[0], be = e[2], te = e[4];
  const double iw[3] = {e[1], e[3], e[5]};
  const double ca = cos(al), sa = sin(al), cb = cos(be), sb = sin(be), ct = cos(te), st = sin(te);
  double R[3][3], D[3][3][3];
  R[0][0] = ca * ct + sa * sb * st;   D[0][0][0] = -sa * ct + ca * sb * st;
  D[1][0][0] = sa * cb * st;          D[2][0][0] = -ca * st + sa * sb * ct;
  R[0][1] = -sa * ct + ca * sb * st;  D[0][0][1] = -ca * ct - sa * sb * st;
  D[1][0][1] = ca * cb * st;          D[2][0][1] = sa * st + ca * sb * ct;
  R[0][2] = -cb * st;                 D[0][0][2] = 0.0;
  D[1][0][2] = sb * st;               D[2][0][2] = -cb * ct;
  R[1][0] = sa * cb;                  D[0][1][0] = ca * cb;
  D[1][1][0] = -sa * sb;              D[2][1][0] = 0.0;
  R[1][1] = ca * cb;                  D[0][1][1] = -sa * cb;
  D[1][1][1] = -ca * sb;              D[2][1][1] = 0.0;
  R[1][2] = sb;                       D[0][1][2] = 0.0;
  D[1][1][2] = cb;                    D[2][1][2] = 0.0;
  R[2][0] = ca * st - sa * sb * ct;   D[0][2][0] = -sa * st - ca * sb * ct;
  D[1][2][0] = -sa * cb * ct;         D[2][2][0] = ca * ct + sa * sb * st;
  R[2][1] = -sa * st - ca * sb * ct;  D[0][2][1] = -ca * st + sa * sb * ct;
  D[1][2][1] = -ca * cb * ct;         D[2][2][1] = -sa * ct + ca * sb * st;
  R[2][2] = cb * ct;                  D[0][2][2] = 0.0;
  D[1][2][2] = -sb * ct;              D[2][2][2] = -cb * st;
  double S[3][3], Sp[6][3][3];
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) {
      double s = 0.0;
      for (int k = 0; k < 3; k++) s += iw[k] * R[r][k] * R[c][k];
      S[r][c] = s;
      for (int a = 0; a < 3; a++) {
        double t = 0.0;
        for (int k = 0; k < 3; k++) {
          double term = iw[k] * (D[a][r][k] * R[c][k] + R[r][k] * D[a][c][k]);
          if (r == 0 && c == 0 && k == 2) term *= 0.5;
          t += term;
        }
        Sp[2 * a][r][c] = t;
        Sp[2 * a + 1][r][c] = R[r][a] * R[c][a];
      }
    }
  const int ir[6] = {0, 0, 0, 1, 1, 2}, ic[6] = {0, 1, 2, 1, 2, 2};
  for (int p = 0; p < 6; p++) {
    double M[3][3];
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) M[r][c] = S[r][c] * Sp[p][r][c];
    for (int q = 0; q < 6; q++) gc.M[p][q] = M[ir[q]][ic[q]];
    for (int k = 0; k < 3; k++) gc.m2[p][k] = 2.0 * (M[k][0] + M[k][1] + M[k][2]);
  }
}

// the gradient entries from the NSUM pair sums + the lp_dhyp sum (red[NSUM]); children in order, bias, sn2
void gpak_grad_assemble(const KernParams &kp, const int *kinds, const double *expans, int d, int N, double sn2,
                        const double *red, double *g) {
  int go = 0;
  for (int t = 0; t < kp.nterms; t++) {
    const KernTerm &T = kp.term[t];
    const double A = red[9 + 2 * t], B = red[10 + 2 * t];
    if (kinds[t] == GPAK_KERN_EXPANS) {
      for (int p = 0; p < 6; p++) g[go + p] = red[p];            // Kernel.cpp:1195-1233
      g[go + 6] = 2.0 * red[6] * expans[6];                      // :1241-1242
      // :1246-1257: 0 for 3-D inputs; with a rock-type column -2 * sum(KD2 % Di2_R) / N, Di2_R = 2 dx4^2
      g[go + 7] = d == 4 ? -4.0 * red[15] / (double)N : 0.0;
      go += 8;
    } else if (kinds[t] == GPAK_KERN_EXP) {                      // Kernel.cpp:671-690
      g[go] = T.var2 * A;
      g[go + 1] = B * sqrt(T.var2);
      go += 2;
    } else {                                                     // Kernel.cpp:512-538
      g[go] = T.var2 * (T.iw / 2) * A;                           // g1/2 = -(sum R . D2), R = var2 QW KD2 (-iw/2)
      g[go + 1] = -0.25 * T.var2 * A;                            // g2/2
      g[go + 2] = T.var2 * B;                                    // g3/2 = sigma^2 sum QW . KD2
      go += 3;
    }
  }
  g[go++] = red[8];                                              // Kern_Bias::getGradients: trace(QW)
  const double sum_dW = 0.5 * red[7];                            // dW = 0.5 * sum(Q % K, 1)
  g[go] = -1.0 * sum_dW * (2.0 / sn2) - red[NSUM];               // GP_Utils.cpp:1226
}

// kinds of the current composition (set by gpak_set_params / gpak_set_kernel)
int gpak_grad_impl(gpak_ctx *ctx, double *g, int ng) {
  GPAK_HIP(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const int N = ctx->N, Np = ctx->Np;
  const long ld = ctx->ld;
  // expected length: children in order (8 / 2 / 3), bias, sn2
  int need = 2, te = -1;
  for (int t = 0; t < ctx->kp.nterms; t++) {
    need += ctx->kinds[t] == GPAK_KERN_EXPANS ? 8 : ctx->kinds[t] == GPAK_KERN_EXP ? 2 : 3;
    if (ctx->kinds[t] == GPAK_KERN_EXPANS) {
      if (te >= 0) { ctx->err = "gpak_grad: at most one ExpAns child"; return GPAK_ENOTIMPL; }
      te = t;
    }
  }
  if (ng != need) { ctx->err = "gpak_grad: gradient vector has the wrong length"; return GPAK_EINVAL; }
  if (ctx->kp.white != 0.0) {
    // Kern_White has no getGradients upstream (Kernel.h:257-283: the base method calls itself)
    ctx->err = "gpak_grad: compositions with a White child have no gradient in the reference either";
    return GPAK_ENOTIMPL;
  }
  if (!ctx->dG) {
    if (hipMalloc(&ctx->dG, sizeof(double) * (size_t)ld * Np) != hipSuccess ||
        hipMalloc(&ctx->dBinv, sizeof(double) * (size_t)ld * Np) != hipSuccess) {
      ctx->err = "device allocation failed for the gradient workspaces (2 N x N matrices)";
      gpak_grad_release(ctx);
      return GPAK_ENOMEM;
    }
  }
  GPAK_HIP(hipEventRecord(ctx->ev[7], st));
  // 1. G = L^-T: forward substitution on the identity, rows restricted to the non-zero part
  double *G = ctx->dG;
  {
    const size_t tot = (size_t)Np * ld;
    hipLaunchKernelGGL(gpak_identity_f64, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, G, ld, Np);
  }
  // two levels like the Cholesky: 128-column steps inside an outer block of 512 columns, then one
  // K = 512 update of the columns to the right; rows beyond the outer block are still zero
  const int GNB = 512;
  for (int J = 0; J < Np; J += GNB) {
    const int W = (Np - J) < GNB ? (Np - J) : GNB;
    for (int j0 = J; j0 < J + W; j0 += PB) {
      const double *inv = ctx->dInv + (size_t)(j0 / PB) * 2 * PB * PB;
      double *Gj = G + (size_t)j0 * ld;
      const int mt = j0 / PB + 1;  // rows 0 .. j0+127: everything below is zero in L^-T
      gpak_launch_gemm_nt(st, mt, 1, PB, 1.0, Gj, ld, inv, PB, 0.0, Gj, ld, 0, 0, false, false);
      const int nin = (J + W - j0 - PB) / PB;
      if (nin > 0)
        gpak_launch_gemm_nt(st, mt, nin, PB, -1.0, Gj, ld, ctx->dM + (j0 + PB) + (size_t)j0 * ld, ld, 1.0,
                            G + (size_t)(j0 + PB) * ld, ld, 0, 0, false, false);
    }
    const int nrest = (Np - J - W) / PB;
    if (nrest > 0)
      gpak_launch_gemm_nt(st, (J + W) / PB, nrest, W, -1.0, G + (size_t)J * ld, ld,
                          ctx->dM + (J + W) + (size_t)J * ld, ld, 1.0, G + (size_t)(J + W) * ld, ld, 0, 0, false, false);
  }
  // 2. B^-1 = G G^T (lower tiles); G[i,k] = 0 for k < i, so the k-loop starts at the row tile
  gpak_launch_gemm_nt(st, Np / PB, Np / PB, Np, 1.0, G, ld, G, ld, 0.0, ctx->dBinv, ld, 0, 0, true, true, true);
  // 3. fused pair pass
  GradConsts gc;
  memset(&gc, 0, sizeof(gc));
  if (te >= 0) build_grad_consts(ctx->expans, gc);
  gc.var2 = te >= 0 ? ctx->kp.term[te].var2 : 0.0;
  gc.bias = ctx->bias; gc.sn2 = ctx->sn2; gc.mode = ctx->dist_mode; gc.te = te;
  for (int t = 0; t < GPAK_MAX_TERMS; t++) gc.kinds[t] = t < ctx->kp.nterms ? ctx->kinds[t] : -1;
  dim3 grid(Np / GT_ROWS, Np / GT_COLS);
  const size_t nblocks = (size_t)grid.x * grid.y;
  if (ctx->gpart_elems < nblocks * NSUM) {
    if (ctx->dGpart) hipFree(ctx->dGpart);
    ctx->dGpart = nullptr; ctx->gpart_elems = 0;
    if (hipMalloc(&ctx->dGpart, sizeof(double) * nblocks * NSUM) != hipSuccess) {
      ctx->err = "device allocation failed for gradient partial sums";
      return GPAK_ENOMEM;
    }
    ctx->gpart_elems = nblocks * NSUM;
  }
  int rc = gpak_ensure_U(ctx);
  if (rc) return rc;
  hipLaunchKernelGGL(gpak_grad_pairs_f64, grid, dim3(256), 0, st, ctx->U.base, ctx->U.cap, ctx->dX, ctx->dX + Np,
                     ctx->dX + 2 * (size_t)Np, ctx->d == 4 ? ctx->dX + 3 * (size_t)Np : (const double *)nullptr,
                     ctx->dAlpha, ctx->dBinv, ld, N, ctx->kp, gc, ctx->dGpart, 0, 0, 0);
  hipLaunchKernelGGL(gpak_grad_reduce_f64, dim3(NSUM), dim3(256), 0, st, ctx->dGpart, (int)nblocks, ctx->dRed + 8);
  hipLaunchKernelGGL(gpak_lpdhyp_f64, dim3(1), dim3(1024), 0, st, N, ctx->dy, ctx->dF, ctx->sn2, ctx->dRed + 8 + NSUM);
  double red[NSUM + 1];
  GPAK_HIP(hipMemcpyAsync(red, ctx->dRed + 8, sizeof(red), hipMemcpyDeviceToHost, st));
  GPAK_HIP(hipEventRecord(ctx->ev[3], st));
  GPAK_HIP(hipEventSynchronize(ctx->ev[3]));
  float ms = 0;
  GPAK_HIP(hipEventElapsedTime(&ms, ctx->ev[7], ctx->ev[3]));
  ctx->times.grad_ms = ms;
  gpak_grad_assemble(ctx->kp, ctx->kinds, ctx->expans, ctx->d, N, ctx->sn2, red, g);
  return GPAK_OK;
}

// ---------------------------------------------------------------------------------------
// The same gradient distributed over P ranks that all hold the factor as packed panels (csrc/dist.hip).
// G = L^-T is upper triangular and its ROWS are independent right-hand sides of the blocked forward substitution
// (the test-major scheme of predict.hip): rank a computes the rows of the 128-row blocks g = t*P + a into a compact
// slab (rows_a x Np, leading dimension rows_a) -- N^3/(3P) flops, no communication; the slabs are all-gathered (the
// caller's broadcasts); B^-1 rows of the same blocks = sum_k G[I,k] G[J,k] for J <= I against every rank's slab --
// N^3/(3P) flops; then the pair pass on those rows and one all-reduce of the NSUM sums.
// ---------------------------------------------------------------------------------------
__global__ void gpak_slab_identity_f64(double *W, int rows, int Np, int P, int a) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t tot = (size_t)rows * Np;
  if (i >= tot) return;
  const int c = (int)(i / rows), r = (int)(i - (size_t)c * rows);
  const int grow = ((r / PB) * P + a) * PB + (r % PB);   // global row of local row r
  W[i] = (grow == c) ? 1.0 : 0.0;
}

void gpak_build_siginv(const double *e, double *A);   // api.hip

static int my_tiles(int Np, int P, int a) { const int T = Np / PB; return T > a ? (T - a + P - 1) / P : 0; }

extern "C" int gpak_dev_grad_g_rows(void *stream, int Np, int nb, int P, int a, const double *const *panels,
                                    const double *const *invs, double *slab) {
  hipStream_t st = (hipStream_t)stream;
  const int Ta = my_tiles(Np, P, a);
  if (Ta == 0) return GPAK_OK;
  const long rows = (long)Ta * PB;
  const size_t tot = (size_t)rows * Np;
  hipLaunchKernelGGL(gpak_slab_identity_f64, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, slab, (int)rows, Np, P, a);
  auto tiles_upto = [&](int gblock) { return gblock >= a ? std::min(Ta, (gblock - a) / P + 1) : 0; };   // g <= gblock
  for (int b = 0, J = 0; J < Np; b++, J += nb) {
    const int W = std::min(nb, Np - J);
    const long ldp = Np - J;
    const double *panel = panels[b];
    for (int j0 = J; j0 < J + W; j0 += PB) {
      const int mt = tiles_upto(j0 / PB);   // rows below are still zero in L^-T
      if (mt == 0) continue;
      const double *inv = invs[b] + (size_t)((j0 - J) / PB) * 2 * PB * PB;
      double *Wj = slab + (size_t)j0 * rows;
      gpak_launch_gemm_nt(st, mt, 1, PB, 1.0, Wj, rows, inv, PB, 0.0, Wj, rows, 0, 0, false, false);
      const int nin = (J + W - j0 - PB) / PB;
      if (nin > 0)
        gpak_launch_gemm_nt(st, mt, nin, PB, -1.0, Wj, rows, panel + (j0 + PB - J) + (size_t)(j0 - J) * ldp, ldp, 1.0,
                            slab + (size_t)(j0 + PB) * rows, rows, 0, 0, false, false);
    }
    const int nrest = (Np - J - W) / PB, mt2 = tiles_upto((J + W) / PB - 1);
    if (nrest > 0 && mt2 > 0)
      gpak_launch_gemm_nt(st, mt2, nrest, W, -1.0, slab + (size_t)J * rows, rows, panel + W, ldp, 1.0,
                          slab + (size_t)(J + W) * rows, rows, 0, 0, false, false);
  }
  return hipGetLastError() == hipSuccess ? GPAK_OK : GPAK_EHIP;
}

// binv: rows_a x (P * Tmax * 128), leading dimension rows_a; the 128-column group of global block g = u*P + b sits at
// group index b*Tmax + u (so that the launch for source rank b writes a contiguous range of tile columns).  One call
// per source rank: the caller needs only its own slab and the one that is passing through (dist.hip streams them)
extern "C" int gpak_dev_grad_binv_rows(void *stream, int Np, int P, int a, int b, const double *slab_a, const double *slab_b,
                                       double *binv) {
  hipStream_t st = (hipStream_t)stream;
  if (b < 0 || b >= P || a < 0 || a >= P) return GPAK_EINVAL;
  const int Ta = my_tiles(Np, P, a), Tb = my_tiles(Np, P, b), Tmax = my_tiles(Np, P, 0);
  if (Ta == 0 || Tb == 0) return GPAK_OK;
  const long rows = (long)Ta * PB;
  // tile (t, u): global row block t*P + a, global column block u*P + b; needed when u*P + b <= t*P + a, i.e.
  // skipped when t < u + (b > a ? 1 : 0)
  gpak_launch_gemm_nt_k0map(st, Ta, Tb, Np, 1.0, slab_a, rows, slab_b, (long)Tb * PB, binv + (size_t)b * Tmax * PB * rows,
                            rows, b > a ? 1 : 0, P, a);
  return hipGetLastError() == hipSuccess ? GPAK_OK : GPAK_EHIP;
}

// out[0..NSUM) = this rank's share of the pair sums, out[NSUM] = sum_i ((y_i - f_i)^2 / sn2 - 1) (replicated: NOT to
// be all-reduced); part: Ta * (Np / 64) * NSUM doubles of scratch
extern "C" int gpak_dev_grad_pairs_rows(void *stream, const double *u, int cap, const double *x_soa, int xs, int n, int Np,
                                        const double *y, const double *f, const double *alpha, const double *binv, int P,
                                        int a, const double *expans, double bias, double sn2, int dist_mode, double *part,
                                        double *out) {
  hipStream_t st = (hipStream_t)stream;
  const int Ta = my_tiles(Np, P, a), Tmax = my_tiles(Np, P, 0);
  KernParams kp;
  memset(&kp, 0, sizeof(kp));
  kp.nterms = 1;
  gpak_build_siginv(expans, kp.term[0].A);
  kp.term[0].var2 = expans[6] * expans[6];
  kp.term[0].profile = GPAK_PROFILE_EXPSQRT;
  const bool d4 = (dist_mode & GPAK_DIST_D4) != 0;
  dist_mode &= 0xF;
  kp.term[0].a33 = expans[7];
  kp.d = d4 ? 4 : 3; kp.bias = bias; kp.mode = dist_mode;
  GradConsts gc;
  memset(&gc, 0, sizeof(gc));
  build_grad_consts(expans, gc);
  gc.var2 = kp.term[0].var2; gc.bias = bias; gc.sn2 = sn2; gc.mode = dist_mode; gc.te = 0;
  gc.kinds[0] = GPAK_KERN_EXPANS; gc.kinds[1] = gc.kinds[2] = -1;
  hipMemsetAsync(out, 0, sizeof(double) * (NSUM + 1), st);
  if (Ta > 0) {
    dim3 grid(Ta, Np / GT_COLS);
    const int nblocks = (int)(grid.x * grid.y);
    hipLaunchKernelGGL(gpak_grad_pairs_f64, grid, dim3(256), 0, st, u, cap, x_soa, x_soa + xs, x_soa + 2 * (size_t)xs,
                       d4 ? x_soa + 3 * (size_t)xs : (const double *)nullptr, alpha, binv, (long)Ta * PB, n, kp, gc, part, P, a,
                       Tmax);
    hipLaunchKernelGGL(gpak_grad_reduce_f64, dim3(NSUM), dim3(256), 0, st, part, nblocks, out);
  }
  hipLaunchKernelGGL(gpak_lpdhyp_f64, dim3(1), dim3(1024), 0, st, n, y, f, sn2, out + NSUM);
  return hipGetLastError() == hipSuccess ? GPAK_OK : GPAK_EHIP;
}

// the constants of the pair pass (M_p = S % S_p as {00,01,02,11,12,22}, and 2 * its column sums): exported so that a
// test engine can restate the pass without restating Kernel.cpp:955-1166 a third time
extern "C" int gpak_dev_grad_consts(const double *expans, double *M36, double *m2_18) {
  GradConsts gc;
  memset(&gc, 0, sizeof(gc));
  build_grad_consts(expans, gc);
  memcpy(M36, gc.M, sizeof(double) * 36);
  memcpy(m2_18, gc.m2, sizeof(double) * 18);
  return GPAK_OK;
}

// host side of the distributed gradient: g[10] from the all-reduced sums
extern "C" int gpak_dev_grad_finish_d(const double *expans, double bias, double sn2, int n, int d, const double *red,
                                      double *g) {
  KernParams kp;
  memset(&kp, 0, sizeof(kp));
  kp.nterms = 1;
  kp.term[0].var2 = expans[6] * expans[6];
  kp.bias = bias;
  const int kinds[GPAK_MAX_TERMS] = {GPAK_KERN_EXPANS, 0, 0};
  gpak_grad_assemble(kp, kinds, expans, d == 4 ? 4 : 3, n, sn2, red, g);
  return GPAK_OK;
}
extern "C" int gpak_dev_grad_finish(const double *expans, double bias, double sn2, int n, const double *red, double *g) {
  return gpak_dev_grad_finish_d(expans, bias, sn2, n, 3, red, g);
}
