// solve.hip -- triangular solves with one right-hand side, log-determinant and the scalar
// terms of the negative log marginal likelihood, for gfx950.
//
// Replaces GP_utils::solve_chol (GP_Utils.cpp:841-845: two LAPACK dtrtrs), the
// `accu(log(Lchol.diag()))` of ldB2_exact() (:913) and the reductions of
// logLikelihood() (:1159) / updatelikelihood() (:810).  HBM-read bound: each solve reads
// the lower triangle once (8*N^2/2 bytes).
//
// One launch per 128-row block.  Every workgroup first recomputes the block's solution
// z_j = inv(L_jj) * x_j (a 128x128 matvec served from L2) and then applies it to its own
// slice of the remaining right-hand side, so there is no inter-workgroup hand-off.
#include "gpak_internal.h"

#define SB 128

__device__ __forceinline__ double gpak_block_sum(double v, double *sh);

// forward: out_j = inv_j * x_j ;  x[r] -= L[r, jblock] * out_j  for r > jblock
__global__ __launch_bounds__(256) void gpak_trsv_fwd_f64(int Np, int jb, const double *__restrict__ L,
                                                          long ld, const double *__restrict__ inv,
                                                          double *x, double *__restrict__ out) {
  __shared__ double xs[SB], zs[SB], part[256];
  const int t = threadIdx.x;
  const int j0 = jb * SB;
  if (t < SB) xs[t] = x[j0 + t];
  __syncthreads();
  {
    const double *ib = inv + (size_t)jb * 2 * SB * SB;
    const int i = t & (SB - 1), half = t >> 7;
    double s = 0.0;
    for (int k = half * 64; k < half * 64 + 64; k++) s += ib[i + k * SB] * xs[k];
    part[t] = s;
  }
  __syncthreads();
  if (t < SB) {
    double z = part[t] + part[t + SB];
    zs[t] = z;
    if (blockIdx.x == 0) out[j0 + t] = z;
  }
  __syncthreads();
  const int r = j0 + SB + blockIdx.x * 256 + t;
  if (r < Np) {
    const double *Lr = L + r + (size_t)j0 * ld;
    double s0 = 0.0, s1 = 0.0;
#pragma unroll 8
    for (int k = 0; k < SB; k += 2) {
      s0 = fma(Lr[(size_t)k * ld], zs[k], s0);
      s1 = fma(Lr[(size_t)(k + 1) * ld], zs[k + 1], s1);
    }
    x[r] -= (s0 + s1);
  }
}

// backward: out_j = inv_j^T * x_j ;  x[c] -= L[jblock, c]^T * out_j  for c < jblock
__global__ __launch_bounds__(256) void gpak_trsv_bwd_f64(int jb, const double *__restrict__ L, long ld,
                                                          const double *__restrict__ inv, double *x,
                                                          double *__restrict__ out, int cols_per_wg,
                                                          int c_begin) {
  __shared__ double xs[SB], zs[SB], part[256];
  const int t = threadIdx.x;
  const int j0 = jb * SB;
  if (t < SB) xs[t] = x[j0 + t];
  __syncthreads();
  {
    const double *ibT = inv + (size_t)jb * 2 * SB * SB + SB * SB;  // inv(L_jj)^T, column-major
    const int i = t & (SB - 1), half = t >> 7;
    double s = 0.0;
    for (int k = half * 64; k < half * 64 + 64; k++) s += ibT[i + k * SB] * xs[k];
    part[t] = s;
  }
  __syncthreads();
  if (t < SB) {
    double z = part[t] + part[t + SB];
    zs[t] = z;
    if (blockIdx.x == 0) out[j0 + t] = z;
  }
  __syncthreads();
  const int lane = t & 63, w = t >> 6;
  const double z0 = zs[2 * lane], z1 = zs[2 * lane + 1];
  const int cbeg = c_begin + blockIdx.x * cols_per_wg;
  const int cend = min(j0, cbeg + cols_per_wg);
  for (int c = cbeg + w; c < cend; c += 4) {
    const double2 l = *reinterpret_cast<const double2 *>(L + j0 + 2 * lane + (size_t)c * ld);
    double s = l.x * z0 + l.y * z1;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) x[c] -= s;
  }
}

void gpak_launch_trsv_fwd(hipStream_t st, int Np, const double *L, long ld, const double *inv, double *x,
                          double *out) {
  const int T = Np / SB;
  for (int jb = 0; jb < T; jb++) {
    int rest = Np - (jb + 1) * SB;
    int grid = rest > 0 ? (rest + 255) / 256 : 1;
    hipLaunchKernelGGL(gpak_trsv_fwd_f64, dim3(grid), dim3(256), 0, st, Np, jb, L, ld, inv, x, out);
  }
}

void gpak_launch_trsv_bwd(hipStream_t st, int Np, const double *L, long ld, const double *inv, double *x,
                          double *out) {
  const int T = Np / SB;
  for (int jb = T - 1; jb >= 0; jb--) {
    int cols = jb * SB;
    int cpw = 32;
    int grid = cols > 0 ? (cols + cpw - 1) / cpw : 1;
    hipLaunchKernelGGL(gpak_trsv_bwd_f64, dim3(grid), dim3(256), 0, st, jb, L, ld, inv, x, out, cpw, 0);
  }
}

// ---- block-column pieces for a factor that is distributed by block columns ---------------
// forward: the four (W/128) steps of block column [J, J+W); x[r] is updated for ALL r below
void gpak_launch_trsv_fwd_block(hipStream_t st, int Np, int J, int W, const double *L, long ld,
                                const double *inv, double *x, double *out) {
  for (int jb = J / SB; jb < (J + W) / SB; jb++) {
    int rest = Np - (jb + 1) * SB;
    int grid = rest > 0 ? (rest + 255) / 256 : 1;
    hipLaunchKernelGGL(gpak_trsv_fwd_f64, dim3(grid), dim3(256), 0, st, Np, jb, L, ld, inv, x, out);
  }
}
// backward inside the diagonal block of block column [J, J+W): only columns >= J are updated
void gpak_launch_trsv_bwd_block(hipStream_t st, int J, int W, const double *L, long ld, const double *inv,
                                double *x, double *out) {
  for (int jb = (J + W) / SB - 1; jb >= J / SB; jb--) {
    int cols = jb * SB - J;
    int cpw = 32;
    int grid = cols > 0 ? (cols + cpw - 1) / cpw : 1;
    hipLaunchKernelGGL(gpak_trsv_bwd_f64, dim3(grid), dim3(256), 0, st, jb, L, ld, inv, x, out, cpw, J);
  }
}

// s[c - J] = sum_{i >= i0} L[i, c] * x[i]  for the W columns c of a block column (one wave per column)
__global__ __launch_bounds__(256) void gpak_coldot_f64(int Np, int i0, int J, int W, const double *__restrict__ L,
                                                        long ld, const double *__restrict__ x,
                                                        double *__restrict__ s) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = J + blockIdx.x * 4 + w;
  if (c >= J + W) return;
  const double *Lc = L + (size_t)c * ld;
  double a0 = 0.0, a1 = 0.0;
  for (int i = i0 + 2 * lane; i < Np; i += 128) {
    const double2 l = *reinterpret_cast<const double2 *>(Lc + i);
    const double2 v = *reinterpret_cast<const double2 *>(x + i);
    a0 = fma(l.x, v.x, a0);
    a1 = fma(l.y, v.y, a1);
  }
  double a = a0 + a1;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off);
  if (lane == 0) s[c - J] = a;
}
void gpak_launch_coldot(hipStream_t st, int Np, int i0, int J, int W, const double *L, long ld, const double *x,
                        double *s) {
  hipLaunchKernelGGL(gpak_coldot_f64, dim3((W + 3) / 4), dim3(256), 0, st, Np, i0, J, W, L, ld, x, s);
}

// out[0] = sum_{c in [J, min(J+W, N))} log L[c, c]
__global__ __launch_bounds__(256) void gpak_logdiag_block_f64(int J, int W, int N, const double *__restrict__ L,
                                                               long ld, double *out) {
  __shared__ double sh[256];
  double s = 0.0;
  for (int c = J + threadIdx.x; c < J + W && c < N; c += 256) s += log(L[c + (size_t)c * ld]);
  s = gpak_block_sum(s, sh);
  if (threadIdx.x == 0) out[0] = s;
}
void gpak_launch_logdiag_block(hipStream_t st, int J, int W, int N, const double *L, long ld, double *out) {
  hipLaunchKernelGGL(gpak_logdiag_block_f64, dim3(1), dim3(256), 0, st, J, W, N, L, ld, out);
}

// ---------------------------------------------------------------------------------------
// block reductions (single workgroup, fixed order -> deterministic)
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ double gpak_block_sum(double v, double *sh) {
  const int t = threadIdx.x;
  sh[t] = v;
  __syncthreads();
  for (int s = blockDim.x / 2; s > 0; s >>= 1) {
    if (t < s) sh[t] += sh[t + s];
    __syncthreads();
  }
  double r = sh[0];
  __syncthreads();
  return r;
}

// red[0] = sum_{i<N} log L_ii      (GP_Utils.cpp:913)
__global__ __launch_bounds__(1024) void gpak_logdet_f64(int N, const double *__restrict__ L, long ld,
                                                         double *red) {
  __shared__ double sh[1024];
  double s = 0.0;
  for (int i = threadIdx.x; i < N; i += 1024) s += log(L[i + (size_t)i * ld]);
  s = gpak_block_sum(s, sh);
  if (threadIdx.x == 0) red[0] = s;
}

// red[1] = Alpha' * (0.5 f)   red[2] = accu(lp),  lp_i = -(y_i-f_i)^2/(2 sn2) - log(2 pi sn2)/2
// (GP_Utils.cpp:810, 1153, 1159)
__global__ __launch_bounds__(1024) void gpak_nlz_terms_f64(int N, const double *__restrict__ y,
                                                            const double *__restrict__ f,
                                                            const double *__restrict__ alpha, double sn2,
                                                            double *red) {
  __shared__ double sh[1024];
  const double c = log(2.0 * M_PI * sn2) / 2.0;
  double q = 0.0, lp = 0.0;
  for (int i = threadIdx.x; i < N; i += 1024) {
    double ymmu = y[i] - f[i];
    q += alpha[i] * (0.5 * f[i]);
    lp += (-1.0 / (2.0 * sn2)) * ymmu * ymmu - c;
  }
  q = gpak_block_sum(q, sh);
  lp = gpak_block_sum(lp, sh);
  if (threadIdx.x == 0) { red[1] = q; red[2] = lp; }
}

__global__ void gpak_scale_f64(int n, const double *__restrict__ in, double s, double *__restrict__ out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[i] * s;
}

__global__ void gpak_axpy_f64(int n, double a, const double *__restrict__ x, double *__restrict__ y) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = fma(a, x[i], y[i]);
}
void gpak_launch_axpy(hipStream_t st, int n, double a, const double *x, double *y) {
  hipLaunchKernelGGL(gpak_axpy_f64, dim3((n + 255) / 256), dim3(256), 0, st, n, a, x, y);
}

void gpak_launch_logdet(hipStream_t st, int N, const double *L, long ld, double *red) {
  hipLaunchKernelGGL(gpak_logdet_f64, dim3(1), dim3(1024), 0, st, N, L, ld, red);
}
void gpak_launch_nlz_terms(hipStream_t st, int N, const double *y, const double *f, const double *alpha,
                           double sn2, double *red) {
  hipLaunchKernelGGL(gpak_nlz_terms_f64, dim3(1), dim3(1024), 0, st, N, y, f, alpha, sn2, red);
}
void gpak_launch_scale(hipStream_t st, int n, const double *in, double s, double *out) {
  hipLaunchKernelGGL(gpak_scale_f64, dim3((n + 255) / 256), dim3(256), 0, st, n, in, s, out);
}
