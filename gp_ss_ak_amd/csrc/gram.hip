// gram.hip -- fused evaluation of the covariance function for gfx950.
//
// Replaces MahaDist / EuclDist (Kernel.cpp:1370-1435, 1343-1368) + the children's computeK
// (Kern_ExpAnisotropic :856-882, Kern_Exponential, Kern_RBF :482-488, Kern_Bias :362-367,
// Kern_White :256-263) + HybKerns::computeK (:140-154) + the "(sW sW') % K + I" passes of
// GP_utils::ldB2_exact (GP_Utils.cpp:874-880) with ONE pass that writes each matrix element
// exactly once.  HBM-write bound: 8 bytes per element written, 32 bytes per point and term read.
#include <cstdlib>

#include "gpak_internal.h"

#define PARR GPAK_PARR

// ---------------------------------------------------------------------------------------
// u = (x - mu) A_t  (Kernel.cpp:1393-1427 / :1356-1362), s = |u|^2 ("sum(X1 % X1, 1)", :1431)
// ---------------------------------------------------------------------------------------
__global__ void gpak_transform_f64(const double *__restrict__ x, int xs, int n, int cap, KernParams kp,
                                   double *__restrict__ out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= cap) return;
  double c0 = 0, c1 = 0, c2 = 0, c3 = 0;
  if (i < n) {
    c0 = x[i] - kp.mu[0]; c1 = x[(size_t)xs + i] - kp.mu[1]; c2 = x[2 * (size_t)xs + i] - kp.mu[2];
    if (kp.d == 4) c3 = x[3 * (size_t)xs + i] - kp.mu[3];
  }
  for (int t = 0; t < kp.nterms; t++) {
    const double *A = kp.term[t].A;
    const double a = c0 * A[0] + c1 * A[1] + c2 * A[2];
    const double b = c0 * A[3] + c1 * A[4] + c2 * A[5];
    const double c = c0 * A[6] + c1 * A[7] + c2 * A[8];
    const double e = c3 * kp.term[t].a33;  // Rot(3,3) = 1, lambda(3,3) = L_r (Kernel.cpp:1411-1424); 0 for 3-D
    PARR(out, cap, t, 0)[i] = a; PARR(out, cap, t, 1)[i] = b; PARR(out, cap, t, 2)[i] = c;
    PARR(out, cap, t, 3)[i] = a * a + b * b + c * c + e * e;
    PARR(out, cap, t, 4)[i] = e;
  }
}

void gpak_launch_transform(hipStream_t st, const double *x, int xs, int n, const KernParams &kp,
                           DevPoints &out) {
  int cap = out.cap;
  hipLaunchKernelGGL(gpak_transform_f64, dim3((cap + 255) / 256), dim3(256), 0, st, x, xs, n, cap, kp, out.base);
  out.n = n;
}

// D2 (Kernel.cpp:1431-1434 / :1365-1367, or the cancellation-free equivalent)
__device__ __forceinline__ double gpak_d2(double p0, double p1, double p2, double ps, double p3, double q0,
                                          double q1, double q2, double qs, double q3, int mode) {
  if (mode == GPAK_DIST_DIRECT) {
    double a = p0 - q0, b = p1 - q1, c = p2 - q2, e = p3 - q3;
    return a * a + b * b + c * c + e * e;  // e == 0 for 3-D inputs: the value is unchanged bit for bit
  }
  double dot = p0 * q0 + p1 * q1 + p2 * q2 + p3 * q3;
  double v = ps + qs - 2.0 * dot;
  return v < 0.0 ? 0.0 : v;
}
// var2 * profile(D2): Kernel.cpp:881 (ExpAns / Exp), :487 (RBF)
__device__ __forceinline__ double gpak_profile(double d2, const KernTerm &t) {
  return t.var2 * gpak_exp_nonpos(t.profile == GPAK_PROFILE_RBF ? -0.5 * t.iw * d2 : -gpak_sqrt_nonneg(d2));
}

// ---------------------------------------------------------------------------------------
// Tile fill.  One workgroup = 128 rows x 64 columns; a wave writes 128 consecutive rows
// of one column per store (1 KiB, 16 B per lane).
// ---------------------------------------------------------------------------------------
#define FILL_ROWS 128
#define FILL_COLS 64
template <int NT>
__global__ __launch_bounds__(256) void gpak_fill_f64(const double *__restrict__ P, int capP, int nP,
                                                      const double *__restrict__ Q, int capQ, int nQ,
                                                      KernParams kp, double scale, double diag, double pad_diag,
                                                      int lower_only, double *__restrict__ C, long ld,
                                                      double *__restrict__ D2out, int col_off) {
  const int row0 = blockIdx.x * FILL_ROWS, col0 = blockIdx.y * FILL_COLS;
  // col_off: global index of the first column when only a block column of the matrix is filled
  if (lower_only && row0 + FILL_ROWS <= col0 + col_off) return;
  // NT = 1: the reference's default composition (one ExpAns term) with everything unrolled;
  // NT = 0: any number of terms at run time
  __shared__ double q[NT ? NT : GPAK_MAX_TERMS][GPAK_PT][FILL_COLS];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int nterms = NT ? NT : kp.nterms;
  if (t < FILL_COLS) {
    const int j = col0 + t;
    const bool ok = j < nQ;
    for (int m = 0; m < nterms; m++)
#pragma unroll
      for (int c = 0; c < GPAK_PT; c++) q[m][c][t] = ok ? PARR(Q, capQ, m, c)[j] : 0.0;
  }
  const int r = row0 + 2 * lane;
  double2 a[NT ? NT : GPAK_MAX_TERMS][GPAK_PT];
#pragma unroll
  for (int m = 0; m < (NT ? NT : GPAK_MAX_TERMS); m++)
    if (m < nterms) {
#pragma unroll
      for (int c = 0; c < GPAK_PT; c++) a[m][c] = *reinterpret_cast<const double2 *>(PARR(P, capP, m, c) + r);
    }
  __syncthreads();
#pragma unroll 4
  for (int c = 0; c < FILL_COLS / 4; c++) {
    const int jl = w + 4 * c, j = col0 + jl;
    double k0 = kp.bias, k1 = kp.bias, d0 = 0.0, d1 = 0.0;
#pragma unroll
    for (int m = 0; m < (NT ? NT : GPAK_MAX_TERMS); m++) {
      if (m >= nterms) break;
      const double e0 = gpak_d2(a[m][0].x, a[m][1].x, a[m][2].x, a[m][3].x, a[m][4].x, q[m][0][jl], q[m][1][jl],
                                q[m][2][jl], q[m][3][jl], q[m][4][jl], kp.mode);
      const double e1 = gpak_d2(a[m][0].y, a[m][1].y, a[m][2].y, a[m][3].y, a[m][4].y, q[m][0][jl], q[m][1][jl],
                                q[m][2][jl], q[m][3][jl], q[m][4][jl], kp.mode);
      k0 += gpak_profile(e0, kp.term[m]);
      k1 += gpak_profile(e1, kp.term[m]);
      d0 += e0; d1 += e1;   // HybKerns sums the children's D2 too (Kernel.cpp:151)
    }
    k0 *= scale; k1 *= scale;
    const bool cj = j < nQ;
    if (!(cj && r < nP)) { k0 = 0.0; d0 = 0.0; }
    if (!(cj && r + 1 < nP)) { k1 = 0.0; d1 = 0.0; }
    if (r == j + col_off) k0 += (cj && r < nP) ? diag + kp.white * scale : pad_diag;
    if (r + 1 == j + col_off) k1 += (cj && r + 1 < nP) ? diag + kp.white * scale : pad_diag;
    *reinterpret_cast<double2 *>(C + r + (size_t)j * ld) = make_double2(k0, k1);
    if (D2out) *reinterpret_cast<double2 *>(D2out + r + (size_t)j * ld) = make_double2(d0, d1);
  }
}

// ---------------------------------------------------------------------------------------
// The reference's default composition -- ONE exp(-sqrt(D2)) term + bias (Kern_ExpAnisotropic + Kern_Bias, or
// Kern_Exponential + Kern_Bias) -- without D2 output: the fill of B = I + K/sn2 that every training step starts with.
// Same tile shape as above.  What differs:
//   * exp by table (12 fp64 instructions instead of 19), sqrt without the last correction, `scale` folded into the
//     two constants, the 4th input column only when there is one: 32 fp64 instruction slots per element instead of 47;
//   * interior tiles (all rows and columns valid, not touching the diagonal) take a branch-free path with no masks,
//     two columns (four elements) per trip so that the dependent FMA chains interleave;
//   * a workgroup keeps both stores of a trip in flight while it computes the next trip.
// VALU floor 0.48 ms and store floor 0.78 ms (5.5 TB/s, tools/store_bw.hip) at N=32768 instead of 0.78 + 0.78.
// ---------------------------------------------------------------------------------------
template <int MODE, bool D4>
__device__ __forceinline__ double gpak_k1(double p0, double p1, double p2, double ps, double p3, double q0, double q1,
                                          double q2, double qs, double q3, double sv, double sb, const double *tab) {
  double d;
  if (MODE == GPAK_DIST_DIRECT) {
    const double a = p0 - q0, b = p1 - q1, c = p2 - q2;
    d = a * a + b * b + c * c;
    if (D4) { const double e = p3 - q3; d += e * e; }   // same association as gpak_d2: ((a*a + b*b) + c*c) + e*e
  } else {
    double dot = p0 * q0 + p1 * q1 + p2 * q2;
    if (D4) dot += p3 * q3;
    d = ps + qs - 2.0 * dot;
    d = d < 0.0 ? 0.0 : d;
  }
  return fma(sv, gpak_exp_neg_tab(gpak_sqrt_nonneg_fast(d), tab), sb);
}

template <int MODE, bool D4>
__global__ __launch_bounds__(256) void gpak_fill1_f64(const double *__restrict__ P, int capP, int nP,
                                                       const double *__restrict__ Q, int capQ, int nQ, double sv,
                                                       double sb, double diag, double pad_diag, int lower_only,
                                                       double *__restrict__ C, long ld, int col_off) {
  const int row0 = blockIdx.x * FILL_ROWS, col0 = blockIdx.y * FILL_COLS;
  if (lower_only && row0 + FILL_ROWS <= col0 + col_off) return;
  __shared__ double q[GPAK_PT][FILL_COLS];
  __shared__ double tab[GPAK_EXPTAB_N];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  if (t < FILL_COLS) {
    const int j = col0 + t;
    const bool ok = j < nQ;
#pragma unroll
    for (int c = 0; c < GPAK_PT; c++) q[c][t] = ok ? PARR(Q, capQ, 0, c)[j] : 0.0;
  } else if (t < FILL_COLS + GPAK_EXPTAB_N) {
    tab[t - FILL_COLS] = gpak_exp2_tab[t - FILL_COLS];
  }
  const int r = row0 + 2 * lane;
  double2 a[GPAK_PT];
#pragma unroll
  for (int c = 0; c < GPAK_PT; c++) a[c] = *reinterpret_cast<const double2 *>(PARR(P, capP, 0, c) + r);
  __syncthreads();
  const int gc0 = col0 + col_off;   // global column of the tile's first column (diagonal test)
  const bool interior = row0 + FILL_ROWS <= nP && col0 + FILL_COLS <= nQ &&
                        (row0 >= gc0 + FILL_COLS || row0 + FILL_ROWS <= gc0);
  if (interior) {
#pragma unroll 2
    for (int c = 0; c < FILL_COLS / 8; c++) {
      const int j0 = w + 8 * c, j1 = j0 + 4;
      const double k00 = gpak_k1<MODE, D4>(a[0].x, a[1].x, a[2].x, a[3].x, a[4].x, q[0][j0], q[1][j0], q[2][j0], q[3][j0], q[4][j0], sv, sb, tab);
      const double k01 = gpak_k1<MODE, D4>(a[0].y, a[1].y, a[2].y, a[3].y, a[4].y, q[0][j0], q[1][j0], q[2][j0], q[3][j0], q[4][j0], sv, sb, tab);
      const double k10 = gpak_k1<MODE, D4>(a[0].x, a[1].x, a[2].x, a[3].x, a[4].x, q[0][j1], q[1][j1], q[2][j1], q[3][j1], q[4][j1], sv, sb, tab);
      const double k11 = gpak_k1<MODE, D4>(a[0].y, a[1].y, a[2].y, a[3].y, a[4].y, q[0][j1], q[1][j1], q[2][j1], q[3][j1], q[4][j1], sv, sb, tab);
      *reinterpret_cast<double2 *>(C + r + (size_t)(col0 + j0) * ld) = make_double2(k00, k01);
      *reinterpret_cast<double2 *>(C + r + (size_t)(col0 + j1) * ld) = make_double2(k10, k11);
    }
    return;
  }
  // edge tiles: padding rows / columns and the diagonal
#pragma unroll 2
  for (int c = 0; c < FILL_COLS / 4; c++) {
    const int jl = w + 4 * c, j = col0 + jl;
    double k0 = gpak_k1<MODE, D4>(a[0].x, a[1].x, a[2].x, a[3].x, a[4].x, q[0][jl], q[1][jl], q[2][jl], q[3][jl], q[4][jl], sv, sb, tab);
    double k1 = gpak_k1<MODE, D4>(a[0].y, a[1].y, a[2].y, a[3].y, a[4].y, q[0][jl], q[1][jl], q[2][jl], q[3][jl], q[4][jl], sv, sb, tab);
    const bool cj = j < nQ;
    if (!(cj && r < nP)) k0 = 0.0;
    if (!(cj && r + 1 < nP)) k1 = 0.0;
    if (r == j + col_off) k0 += (cj && r < nP) ? diag : pad_diag;
    if (r + 1 == j + col_off) k1 += (cj && r + 1 < nP) ? diag : pad_diag;
    *reinterpret_cast<double2 *>(C + r + (size_t)j * ld) = make_double2(k0, k1);
  }
}

void gpak_launch_fill(hipStream_t st, const DevPoints &P, const DevPoints &Q, int rows_p, int cols_p,
                      const KernParams &kp, double scale, double diag, double pad_diag, int lower_only,
                      double *C, long ld, double *D2out, int col_off) {
  dim3 grid(rows_p / FILL_ROWS, cols_p / FILL_COLS);
  const bool fast_off = !gpak_tuning().fill_fast;
  if (kp.nterms == 1 && kp.term[0].profile == GPAK_PROFILE_EXPSQRT && !D2out && !fast_off) {
    const double sv = scale * kp.term[0].var2, sb = scale * kp.bias, dg = diag + kp.white * scale;
#define GPAK_FILL1(MODE_, D4_)                                                                                        \
  hipLaunchKernelGGL((gpak_fill1_f64<MODE_, D4_>), grid, dim3(256), 0, st, P.base, P.cap, P.n, Q.base, Q.cap, Q.n, sv, \
                     sb, dg, pad_diag, lower_only, C, ld, col_off)
    const bool d4 = kp.d == 4;
    if (kp.mode == GPAK_DIST_DIRECT) { if (d4) GPAK_FILL1(GPAK_DIST_DIRECT, true); else GPAK_FILL1(GPAK_DIST_DIRECT, false); }
    else { if (d4) GPAK_FILL1(GPAK_DIST_EXPANSION, true); else GPAK_FILL1(GPAK_DIST_EXPANSION, false); }
#undef GPAK_FILL1
    return;
  }
  if (kp.nterms == 1)
    hipLaunchKernelGGL(gpak_fill_f64<1>, grid, dim3(256), 0, st, P.base, P.cap, P.n, Q.base, Q.cap, Q.n, kp, scale,
                       diag, pad_diag, lower_only, C, ld, D2out, col_off);
  else
    hipLaunchKernelGGL(gpak_fill_f64<0>, grid, dim3(256), 0, st, P.base, P.cap, P.n, Q.base, Q.cap, Q.n, kp, scale,
                       diag, pad_diag, lower_only, C, ld, D2out, col_off);
}

// ---------------------------------------------------------------------------------------
// Fused Gram-matvec: out_j = sum_i w_i k(P_i, Q_j).  Serves mvmK_exact (GP_Utils.cpp:394-397,
// f = K*Alpha at :1147) and _postMean (:958-972) without ever storing K / kX.  The white-noise
// diagonal is NOT part of it (callers add white * w_j where the two sets coincide).
// Split over i in `splits` slabs -> part[split][j]; a second kernel sums the slabs in a
// fixed order (deterministic).
// ---------------------------------------------------------------------------------------
#define KMV_CHUNK 256
#define KMV_COLS 2   // target columns per thread: every LDS read of a source point serves two evaluations (4: no faster)
template <int NT>
__global__ __launch_bounds__(256) void gpak_kmatvec_part_f64(const double *__restrict__ P, int capP, int nP, int p_off,
                                                              const double *__restrict__ w, int per_split,
                                                              const double *__restrict__ Q, int capQ, int nQ,
                                                              KernParams kp, double *__restrict__ part,
                                                              int part_ld) {
  constexpr int MT = NT ? NT : GPAK_MAX_TERMS;
  __shared__ double sp[MT * GPAK_PT + 1][KMV_CHUNK];
  const int nterms = NT ? NT : kp.nterms;
  int j[KMV_COLS];
  bool ok[KMV_COLS];
  double b[KMV_COLS][MT][GPAK_PT];
#pragma unroll
  for (int q = 0; q < KMV_COLS; q++) {
    j[q] = (blockIdx.x * KMV_COLS + q) * 256 + threadIdx.x;
    ok[q] = j[q] < nQ;
#pragma unroll
    for (int m = 0; m < MT; m++)
      if (m < nterms) {
#pragma unroll
        for (int c = 0; c < GPAK_PT; c++) b[q][m][c] = ok[q] ? PARR(Q, capQ, m, c)[j[q]] : 0.0;
      }
  }
  const int i_begin = blockIdx.y * per_split;
  const int i_end = min(nP, i_begin + per_split);
  double acc[KMV_COLS];
#pragma unroll
  for (int q = 0; q < KMV_COLS; q++) acc[q] = 0.0;
  for (int i0 = i_begin; i0 < i_end; i0 += KMV_CHUNK) {
    const int i = i0 + threadIdx.x;
    const bool v = i < i_end;
    __syncthreads();
    for (int m = 0; m < nterms; m++)
#pragma unroll
      for (int c = 0; c < GPAK_PT; c++) sp[GPAK_PT * m + c][threadIdx.x] = v ? PARR(P, capP, m, c)[p_off + i] : 0.0;
    sp[MT * GPAK_PT][threadIdx.x] = v ? w[i] : 0.0;  // zero weight masks the tail
    __syncthreads();
#pragma unroll 4
    for (int k = 0; k < KMV_CHUNK; k++) {
      double kv[KMV_COLS];
#pragma unroll
      for (int q = 0; q < KMV_COLS; q++) kv[q] = kp.bias;
#pragma unroll
      for (int m = 0; m < MT; m++) {
        if (m >= nterms) break;
        const double p0 = sp[GPAK_PT * m][k], p1 = sp[GPAK_PT * m + 1][k], p2 = sp[GPAK_PT * m + 2][k],
                     p3 = sp[GPAK_PT * m + 3][k], p4 = sp[GPAK_PT * m + 4][k];
#pragma unroll
        for (int q = 0; q < KMV_COLS; q++)
          kv[q] += gpak_profile(gpak_d2(p0, p1, p2, p3, p4, b[q][m][0], b[q][m][1], b[q][m][2], b[q][m][3], b[q][m][4],
                                        kp.mode), kp.term[m]);
      }
      const double wk = sp[MT * GPAK_PT][k];
#pragma unroll
      for (int q = 0; q < KMV_COLS; q++) acc[q] = fma(wk, kv[q], acc[q]);
    }
  }
#pragma unroll
  for (int q = 0; q < KMV_COLS; q++)
    if (ok[q]) part[(size_t)blockIdx.y * part_ld + j[q]] = acc[q];
}

__global__ void gpak_kmatvec_reduce_f64(const double *__restrict__ part, int part_ld, int splits, int nQ,
                                        double *__restrict__ out) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= nQ) return;
  double s = 0.0;
  for (int k = 0; k < splits; k++) s += part[(size_t)k * part_ld + j];
  out[j] = s;
}

void gpak_launch_sum_splits(hipStream_t st, const double *part, int part_ld, int splits, int n, double *out) {
  hipLaunchKernelGGL(gpak_kmatvec_reduce_f64, dim3((n + 255) / 256), dim3(256), 0, st, part, part_ld, splits, n,
                     out);
}

// scratch must hold splits*Q.cap doubles; splits is chosen by the caller via gpak_kmatvec_splits
int gpak_kmatvec_splits(int nP, int nQ) {
  int wg = (nQ + 256 * KMV_COLS - 1) / (256 * KMV_COLS);
  int s = 2048 / (wg > 0 ? wg : 1);
  if (s < 1) s = 1;
  if (s > 64) s = 64;
  int max_s = (nP + KMV_CHUNK - 1) / KMV_CHUNK;
  if (s > max_s) s = max_s;
  return s < 1 ? 1 : s;
}

// The default composition (one exp(-sqrt(D2)) term + bias) with the table exp of the fill: 32 instead of 47 fp64
// instruction slots per kernel evaluation; fp64-VALU bound (N^2 evaluations, nothing stored).
template <int MODE, bool D4>
__global__ __launch_bounds__(256) void gpak_kmatvec1_part_f64(const double *__restrict__ P, int capP, int nP, int p_off,
                                                               const double *__restrict__ w, int per_split,
                                                               const double *__restrict__ Q, int capQ, int nQ,
                                                               double var2, double bias, double *__restrict__ part,
                                                               int part_ld) {
  __shared__ double sp[GPAK_PT + 1][KMV_CHUNK];
  __shared__ double tab[GPAK_EXPTAB_N];
  if (threadIdx.x < GPAK_EXPTAB_N) tab[threadIdx.x] = gpak_exp2_tab[threadIdx.x];
  int j[KMV_COLS];
  bool ok[KMV_COLS];
  double b[KMV_COLS][GPAK_PT];
#pragma unroll
  for (int q = 0; q < KMV_COLS; q++) {
    j[q] = (blockIdx.x * KMV_COLS + q) * 256 + threadIdx.x;
    ok[q] = j[q] < nQ;
#pragma unroll
    for (int c = 0; c < GPAK_PT; c++) b[q][c] = ok[q] ? PARR(Q, capQ, 0, c)[j[q]] : 0.0;
  }
  const int i_begin = blockIdx.y * per_split;
  const int i_end = min(nP, i_begin + per_split);
  double acc[KMV_COLS], wsum = 0.0;
#pragma unroll
  for (int q = 0; q < KMV_COLS; q++) acc[q] = 0.0;
  for (int i0 = i_begin; i0 < i_end; i0 += KMV_CHUNK) {
    const int i = i0 + threadIdx.x;
    const bool v = i < i_end;
    __syncthreads();
#pragma unroll
    for (int c = 0; c < GPAK_PT; c++) sp[c][threadIdx.x] = v ? PARR(P, capP, 0, c)[p_off + i] : 0.0;
    sp[GPAK_PT][threadIdx.x] = v ? w[i] : 0.0;  // zero weight masks the tail
    __syncthreads();
#pragma unroll 4
    for (int k = 0; k < KMV_CHUNK; k++) {
      const double p0 = sp[0][k], p1 = sp[1][k], p2 = sp[2][k], p3 = sp[3][k], p4 = sp[4][k], wk = sp[GPAK_PT][k];
      wsum += wk;
#pragma unroll
      for (int q = 0; q < KMV_COLS; q++)   // the bias term is added once at the end: sum_k w_k (bias + var2 e_k)
        acc[q] = fma(wk, gpak_k1<MODE, D4>(p0, p1, p2, p3, p4, b[q][0], b[q][1], b[q][2], b[q][3], b[q][4], var2, 0.0, tab),
                     acc[q]);
    }
  }
#pragma unroll
  for (int q = 0; q < KMV_COLS; q++)
    if (ok[q]) part[(size_t)blockIdx.y * part_ld + j[q]] = fma(bias, wsum, acc[q]);
}

// ---------------------------------------------------------------------------------------
// f = K w with BOTH vectors over the same points (the nlZ's f = K alpha): K is symmetric, so every kernel value is
// evaluated once and used twice -- N^2/2 evaluations instead of N^2 of a kernel that is bound by exactly those.
// Macro blocks of KSYM_G * 256 points; one workgroup per pair (mi <= mj).  Thread t owns KSYM_G columns of macro block
// mj (as above: coordinates in registers, one accumulator each); the rows of macro block mi come through LDS 256 at a
// time.  Off the diagonal a value k(i, j) also feeds row i: the thread's w_j k summed over its columns, then summed
// over the wave's 64 lanes -- 16 rows at a time through an LDS transpose (16 writes, 16 reads, 15 adds and two
// quad-permute steps per lane and 16 rows: 3.3 instructions per row; a DPP reduction per row costs 22, a fifth of the
// evaluations it accompanies) -- and after 256 rows the four waves' totals go out.
// part[r][x] receives the contribution of the pair (x's macro block, r) to f[x] -- the column sums of pair (r, m) for
// r <= m, the row sums of pair (m, r) for r > m -- so f = the sum of the nbm rows of part (gpak_kmatvec_reduce_f64),
// in a fixed order.  The diagonal pairs evaluate their full square and keep only the column sums.
// KSYM_G = 2: 2080 workgroups at N = 32768 (4: 528, two waves per SIMD, slower).  Used from 32 macro blocks on
// (N > 15872): below that the grid is too small (N = 8192: 0.45 ms against 0.10 ms for the plain kernel).
// ---------------------------------------------------------------------------------------
#define KSYM_G 2
#define KSYM_MIN_BLOCKS 32
__device__ __forceinline__ double gpak_quad_add(double v, const bool swap2) {
  int lo = __double2loint(v), hi = __double2hiint(v), mlo, mhi;
  if (!swap2) {   // quad_perm:[1,0,3,2]
    mlo = __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xf, 0xf, false); mhi = __builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xf, 0xf, false);
  } else {        // quad_perm:[2,3,0,1]
    mlo = __builtin_amdgcn_update_dpp(0, lo, 0x4E, 0xf, 0xf, false); mhi = __builtin_amdgcn_update_dpp(0, hi, 0x4E, 0xf, 0xf, false);
  }
  return v + __hiloint2double(mhi, mlo);
}

template <int MODE, bool D4>
__global__ __launch_bounds__(256) void gpak_kmatvec1_sym_f64(const double *__restrict__ P, int capP, int n,
                                                              const double *__restrict__ w, double var2, double bias,
                                                              double *__restrict__ part, int part_ld) {
  __shared__ double sp[GPAK_PT + 1][KMV_CHUNK];
  __shared__ double tab[GPAK_EXPTAB_N];
  __shared__ double rw[4][KMV_CHUNK];
  __shared__ double tr[4][16][65];   // one wave's 16 rows x 64 lanes, row stride 65: the transposed read is 2-way at worst
  __shared__ double wred[256];
  const int t = threadIdx.x, lane = t & 63;
  const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
  if (t < GPAK_EXPTAB_N) tab[t] = gpak_exp2_tab[t];
  int mi = blockIdx.x, mj = 0;
  while (mi > mj) { mi -= mj + 1; mj++; }   // pair index -> (mi <= mj), scalar
  const bool offdiag = mi != mj;
  int j[KSYM_G];
  double b[KSYM_G][GPAK_PT], wj[KSYM_G], acc[KSYM_G];
  double wjsum = 0.0;
#pragma unroll
  for (int q = 0; q < KSYM_G; q++) {
    j[q] = (mj * KSYM_G + q) * 256 + t;
    const bool ok = j[q] < n;
#pragma unroll
    for (int c = 0; c < GPAK_PT; c++) b[q][c] = ok ? PARR(P, capP, 0, c)[j[q]] : 0.0;
    wj[q] = ok ? w[j[q]] : 0.0;
    wjsum += wj[q];
    acc[q] = 0.0;
  }
  // W_j = the weights of the whole column macro block (the bias term of the row sums), summed in a fixed order
  wred[t] = wjsum;
  __syncthreads();
  double Wj = 0.0;
  for (int u = 0; u < 256; u++) Wj += wred[u];
  double wsum = 0.0;
  for (int ib = 0; ib < KSYM_G; ib++) {
    const int i0 = (mi * KSYM_G + ib) * 256;
    if (i0 >= n) break;   // uniform
    const int i = i0 + t;
    const bool v = i < n;
    __syncthreads();
#pragma unroll
    for (int c = 0; c < GPAK_PT; c++) sp[c][t] = v ? PARR(P, capP, 0, c)[i] : 0.0;
    sp[GPAK_PT][t] = v ? w[i] : 0.0;  // zero weight masks the tail
    __syncthreads();
    for (int k0 = 0; k0 < KMV_CHUNK; k0 += 16) {
#pragma unroll 4
      for (int kk = 0; kk < 16; kk++) {
        const int k = k0 + kk;
        const double p0 = sp[0][k], p1 = sp[1][k], p2 = sp[2][k], p3 = sp[3][k], p4 = sp[4][k], wk = sp[GPAK_PT][k];
        wsum += wk;
        double r = 0.0;
#pragma unroll
        for (int q = 0; q < KSYM_G; q++) {
          const double kv = gpak_k1<MODE, D4>(p0, p1, p2, p3, p4, b[q][0], b[q][1], b[q][2], b[q][3], b[q][4], var2, 0.0, tab);
          acc[q] = fma(wk, kv, acc[q]);
          r = fma(wj[q], kv, r);
        }
        if (offdiag) tr[wv][kk][lane] = r;
      }
      if (offdiag) {
        // lane l sums a quarter (l & 3) of row l >> 2; the wave's own LDS writes are in order before these reads
        __builtin_amdgcn_wave_barrier();
        const double *src = &tr[wv][lane >> 2][(lane & 3) * 16];
        double s0 = 0.0;
#pragma unroll
        for (int e = 0; e < 16; e++) s0 += src[e];
        s0 = gpak_quad_add(s0, false);
        s0 = gpak_quad_add(s0, true);
        if ((lane & 3) == 0) rw[wv][k0 + (lane >> 2)] = s0;
        __builtin_amdgcn_wave_barrier();
      }
    }
    if (offdiag) {
      __syncthreads();
      if (v) part[(size_t)mj * part_ld + i] = fma(bias, Wj, (rw[0][t] + rw[1][t]) + (rw[2][t] + rw[3][t]));
    }
  }
#pragma unroll
  for (int q = 0; q < KSYM_G; q++)
    if (j[q] < n) part[(size_t)mi * part_ld + j[q]] = fma(bias, wsum, acc[q]);
}

// source points [p_off, p_off + np) of P with weights w[0..np)
void gpak_launch_kmatvec(hipStream_t st, const DevPoints &P, int p_off, int np, const double *w, const DevPoints &Q,
                         const KernParams &kp, double *scratch, int splits, double *out, int scratch_rows) {
  if (scratch_rows < splits) scratch_rows = splits;
  int per = (np + splits - 1) / splits;
  per = (per + KMV_CHUNK - 1) / KMV_CHUNK * KMV_CHUNK;
  dim3 grid((Q.n + 256 * KMV_COLS - 1) / (256 * KMV_COLS), splits);
  const bool fast_off = !gpak_tuning().fill_fast;
  const bool sym_off = !gpak_tuning().kmv_sym;
  // the symmetric kernel: both vectors over the same points, a grid large enough to fill the chip, and its nbm partial
  // rows fit into the scratch (scratch_rows * Q.cap doubles; at least the `splits` rows every caller provides)
  const int nbm = (Q.n + 256 * KSYM_G - 1) / (256 * KSYM_G);
  if (kp.nterms == 1 && kp.term[0].profile == GPAK_PROFILE_EXPSQRT && !fast_off && !sym_off && P.base == Q.base &&
      p_off == 0 && np == Q.n && nbm >= KSYM_MIN_BLOCKS && nbm <= scratch_rows) {
    const dim3 sgrid((unsigned)(nbm * (nbm + 1) / 2));
#define GPAK_KMVS(MODE_, D4_)                                                                                          \
  hipLaunchKernelGGL((gpak_kmatvec1_sym_f64<MODE_, D4_>), sgrid, dim3(256), 0, st, P.base, P.cap, np, w,                 \
                     kp.term[0].var2, kp.bias, scratch, Q.cap)
    const bool d4 = kp.d == 4;
    if (kp.mode == GPAK_DIST_DIRECT) { if (d4) GPAK_KMVS(GPAK_DIST_DIRECT, true); else GPAK_KMVS(GPAK_DIST_DIRECT, false); }
    else { if (d4) GPAK_KMVS(GPAK_DIST_EXPANSION, true); else GPAK_KMVS(GPAK_DIST_EXPANSION, false); }
#undef GPAK_KMVS
    hipLaunchKernelGGL(gpak_kmatvec_reduce_f64, dim3((Q.n + 255) / 256), dim3(256), 0, st, scratch, Q.cap, nbm, Q.n, out);
    return;
  }
  if (kp.nterms == 1 && kp.term[0].profile == GPAK_PROFILE_EXPSQRT && !fast_off) {
#define GPAK_KMV1(MODE_, D4_)                                                                                          \
  hipLaunchKernelGGL((gpak_kmatvec1_part_f64<MODE_, D4_>), grid, dim3(256), 0, st, P.base, P.cap, np, p_off, w, per, Q.base, \
                     Q.cap, Q.n, kp.term[0].var2, kp.bias, scratch, Q.cap)
    const bool d4 = kp.d == 4;
    if (kp.mode == GPAK_DIST_DIRECT) { if (d4) GPAK_KMV1(GPAK_DIST_DIRECT, true); else GPAK_KMV1(GPAK_DIST_DIRECT, false); }
    else { if (d4) GPAK_KMV1(GPAK_DIST_EXPANSION, true); else GPAK_KMV1(GPAK_DIST_EXPANSION, false); }
#undef GPAK_KMV1
  } else if (kp.nterms == 1)
    hipLaunchKernelGGL(gpak_kmatvec_part_f64<1>, grid, dim3(256), 0, st, P.base, P.cap, np, p_off, w, per, Q.base,
                       Q.cap, Q.n, kp, scratch, Q.cap);
  else
    hipLaunchKernelGGL(gpak_kmatvec_part_f64<0>, grid, dim3(256), 0, st, P.base, P.cap, np, p_off, w, per, Q.base,
                       Q.cap, Q.n, kp, scratch, Q.cap);
  hipLaunchKernelGGL(gpak_kmatvec_reduce_f64, dim3((Q.n + 255) / 256), dim3(256), 0, st, scratch, Q.cap, splits, Q.n,
                     out);
}
