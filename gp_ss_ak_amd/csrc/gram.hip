// gram.hip -- fused ExpAns+Bias kernel-function evaluation for gfx950.
//
// Replaces MahaDist (Kernel.cpp:1370-1435) + Kern_ExpAnisotropic::computeK (:856-882) +
// Kern_Bias::computeK (:362-367) + HybKerns::computeK (:140-154) + the "(sW sW') % K + I"
// passes of GP_utils::ldB2_exact (GP_Utils.cpp:874-880) with ONE pass that writes each
// matrix element exactly once.  HBM-write bound: 8 bytes per element written, 48 bytes per
// point read.
#include "gpak_internal.h"

// ---------------------------------------------------------------------------------------
// u = (x - mu) A   (Kernel.cpp:1393-1427), s = |u|^2 (Kernel.cpp:1431 "sum(X1 % X1, 1)")
// ---------------------------------------------------------------------------------------
__global__ void gpak_transform_f64(const double *__restrict__ x, int xs, int n, int cap, KernParams kp,
                                   double *__restrict__ u0, double *__restrict__ u1,
                                   double *__restrict__ u2, double *__restrict__ s) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= cap) return;
  double a = 0, b = 0, c = 0;
  if (i < n) {
    double c0 = x[i] - kp.mu[0], c1 = x[(size_t)xs + i] - kp.mu[1], c2 = x[2 * (size_t)xs + i] - kp.mu[2];
    a = c0 * kp.A[0] + c1 * kp.A[1] + c2 * kp.A[2];
    b = c0 * kp.A[3] + c1 * kp.A[4] + c2 * kp.A[5];
    c = c0 * kp.A[6] + c1 * kp.A[7] + c2 * kp.A[8];
  }
  u0[i] = a; u1[i] = b; u2[i] = c;
  s[i] = a * a + b * b + c * c;
}

void gpak_launch_transform(hipStream_t st, const double *x, int xs, int n, const KernParams &kp,
                           DevPoints &out) {
  int cap = out.cap;
  hipLaunchKernelGGL(gpak_transform_f64, dim3((cap + 255) / 256), dim3(256), 0, st, x, xs, n, cap, kp,
                     out.u0, out.u1, out.u2, out.s);
  out.n = n;
}

// D2 (Kernel.cpp:1431-1434, or its cancellation-free equivalent) and k = var2*exp(-sqrt(D2)) + bias
__device__ __forceinline__ double gpak_d2(double p0, double p1, double p2, double ps, double q0, double q1,
                                          double q2, double qs, int mode) {
  if (mode == GPAK_DIST_DIRECT) {
    double a = p0 - q0, b = p1 - q1, c = p2 - q2;
    return a * a + b * b + c * c;
  }
  double dot = p0 * q0 + p1 * q1 + p2 * q2;
  double v = ps + qs - 2.0 * dot;
  return v < 0.0 ? 0.0 : v;
}
__device__ __forceinline__ double gpak_kfun(double d2, double var2, double bias) {
  return var2 * exp(-1.0 * sqrt(d2)) + bias;
}

// ---------------------------------------------------------------------------------------
// Tile fill.  One workgroup = 128 rows x 64 columns; a wave writes 128 consecutive rows
// of one column per store (1 KiB, 16 B per lane).
// ---------------------------------------------------------------------------------------
#define FILL_ROWS 128
#define FILL_COLS 64
__global__ __launch_bounds__(256) void gpak_fill_f64(
    const double *__restrict__ pu0, const double *__restrict__ pu1, const double *__restrict__ pu2,
    const double *__restrict__ ps, int nP, const double *__restrict__ qu0, const double *__restrict__ qu1,
    const double *__restrict__ qu2, const double *__restrict__ qs, int nQ, double var2, double bias,
    int mode, double scale, double diag, double pad_diag, int lower_only, double *__restrict__ C, long ld,
    double *__restrict__ D2out, int col_off) {
  const int row0 = blockIdx.x * FILL_ROWS, col0 = blockIdx.y * FILL_COLS;
  // col_off: global index of the first column when only a block column of the matrix is filled
  if (lower_only && row0 + FILL_ROWS <= col0 + col_off) return;
  __shared__ double q[4][FILL_COLS];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  if (t < FILL_COLS) {
    int j = col0 + t;
    bool ok = j < nQ;
    q[0][t] = ok ? qu0[j] : 0.0;
    q[1][t] = ok ? qu1[j] : 0.0;
    q[2][t] = ok ? qu2[j] : 0.0;
    q[3][t] = ok ? qs[j] : 0.0;
  }
  const int r = row0 + 2 * lane;
  const double2 a0 = *reinterpret_cast<const double2 *>(pu0 + r);
  const double2 a1 = *reinterpret_cast<const double2 *>(pu1 + r);
  const double2 a2 = *reinterpret_cast<const double2 *>(pu2 + r);
  const double2 as = *reinterpret_cast<const double2 *>(ps + r);
  __syncthreads();
#pragma unroll 4
  for (int c = 0; c < FILL_COLS / 4; c++) {
    const int jl = w + 4 * c, j = col0 + jl;
    const double b0 = q[0][jl], b1 = q[1][jl], b2 = q[2][jl], bs = q[3][jl];
    double d0 = gpak_d2(a0.x, a1.x, a2.x, as.x, b0, b1, b2, bs, mode);
    double d1 = gpak_d2(a0.y, a1.y, a2.y, as.y, b0, b1, b2, bs, mode);
    double k0 = gpak_kfun(d0, var2, bias) * scale;
    double k1 = gpak_kfun(d1, var2, bias) * scale;
    const bool cj = j < nQ;
    if (!(cj && r < nP)) { k0 = 0.0; d0 = 0.0; }
    if (!(cj && r + 1 < nP)) { k1 = 0.0; d1 = 0.0; }
    if (r == j + col_off) k0 += (cj && r < nP) ? diag : pad_diag;
    if (r + 1 == j + col_off) k1 += (cj && r + 1 < nP) ? diag : pad_diag;
    *reinterpret_cast<double2 *>(C + r + (size_t)j * ld) = make_double2(k0, k1);
    if (D2out) *reinterpret_cast<double2 *>(D2out + r + (size_t)j * ld) = make_double2(d0, d1);
  }
}

void gpak_launch_fill(hipStream_t st, const DevPoints &P, const DevPoints &Q, int rows_p, int cols_p,
                      const KernParams &kp, double scale, double diag, double pad_diag, int lower_only,
                      double *C, long ld, double *D2out, int col_off) {
  dim3 grid(rows_p / FILL_ROWS, cols_p / FILL_COLS);
  hipLaunchKernelGGL(gpak_fill_f64, grid, dim3(256), 0, st, P.u0, P.u1, P.u2, P.s, P.n, Q.u0, Q.u1, Q.u2,
                     Q.s, Q.n, kp.var2, kp.bias, kp.mode, scale, diag, pad_diag, lower_only, C, ld, D2out, col_off);
}

// ---------------------------------------------------------------------------------------
// Fused Gram-matvec: out_j = sum_i w_i k(P_i, Q_j).  Serves mvmK_exact (GP_Utils.cpp:394-397,
// f = K*Alpha at :1147) and _postMean (:958-972) without ever storing K / kX.
// Split over i in `splits` slabs -> part[split][j]; a second kernel sums the slabs in a
// fixed order (deterministic).
// ---------------------------------------------------------------------------------------
#define KMV_CHUNK 256
__global__ __launch_bounds__(256) void gpak_kmatvec_part_f64(
    const double *__restrict__ pu0, const double *__restrict__ pu1, const double *__restrict__ pu2,
    const double *__restrict__ ps, const double *__restrict__ w, int nP, int per_split,
    const double *__restrict__ qu0, const double *__restrict__ qu1, const double *__restrict__ qu2,
    const double *__restrict__ qs, int nQ, double var2, double bias, int mode, double *__restrict__ part,
    int part_ld) {
  __shared__ double sp[5][KMV_CHUNK];
  const int j = blockIdx.x * 256 + threadIdx.x;
  const bool ok = j < nQ;
  const double b0 = ok ? qu0[j] : 0.0, b1 = ok ? qu1[j] : 0.0, b2 = ok ? qu2[j] : 0.0, bs = ok ? qs[j] : 0.0;
  const int i_begin = blockIdx.y * per_split;
  const int i_end = min(nP, i_begin + per_split);
  double acc = 0.0;
  for (int i0 = i_begin; i0 < i_end; i0 += KMV_CHUNK) {
    const int i = i0 + threadIdx.x;
    const bool v = i < i_end;
    __syncthreads();
    sp[0][threadIdx.x] = v ? pu0[i] : 0.0;
    sp[1][threadIdx.x] = v ? pu1[i] : 0.0;
    sp[2][threadIdx.x] = v ? pu2[i] : 0.0;
    sp[3][threadIdx.x] = v ? ps[i] : 0.0;
    sp[4][threadIdx.x] = v ? w[i] : 0.0;  // zero weight masks the tail
    __syncthreads();
#pragma unroll 4
    for (int k = 0; k < KMV_CHUNK; k++) {
      double d2 = gpak_d2(sp[0][k], sp[1][k], sp[2][k], sp[3][k], b0, b1, b2, bs, mode);
      acc = fma(sp[4][k], gpak_kfun(d2, var2, bias), acc);
    }
  }
  if (ok) part[(size_t)blockIdx.y * part_ld + j] = acc;
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
  int wg = (nQ + 255) / 256;
  int s = 2048 / (wg > 0 ? wg : 1);
  if (s < 1) s = 1;
  if (s > 64) s = 64;
  int max_s = (nP + KMV_CHUNK - 1) / KMV_CHUNK;
  if (s > max_s) s = max_s;
  return s < 1 ? 1 : s;
}

void gpak_launch_kmatvec(hipStream_t st, const DevPoints &P, const double *w, const DevPoints &Q,
                         const KernParams &kp, double *scratch, int splits, double *out) {
  int per = (P.n + splits - 1) / splits;
  per = (per + KMV_CHUNK - 1) / KMV_CHUNK * KMV_CHUNK;
  dim3 grid((Q.n + 255) / 256, splits);
  hipLaunchKernelGGL(gpak_kmatvec_part_f64, grid, dim3(256), 0, st, P.u0, P.u1, P.u2, P.s, w, P.n, per,
                     Q.u0, Q.u1, Q.u2, Q.s, Q.n, kp.var2, kp.bias, kp.mode, scratch, Q.cap);
  hipLaunchKernelGGL(gpak_kmatvec_reduce_f64, dim3((Q.n + 255) / 256), dim3(256), 0, st, scratch, Q.cap,
                     splits, Q.n, out);
}
