// potrf.hip -- blocked right-looking fp64 Cholesky (lower, in place) for gfx950.
//
// Replaces arma::chol inside GP_utils::ldB2_exact (GP_Utils.cpp:881, 903; LAPACK dpotrf).
// The device keeps L = R^T (lower) where the reference keeps the upper R.
//
//   for each outer block column J (width nb_outer, multiple of 128):
//     for each 128-column j inside it:
//       [potrf128]  LDS-resident factorisation of the 128x128 diagonal block + its
//                   inverse (and inverse transpose, for the back substitution)
//       [gemm_nt]   panel solve  P := P * inv(L_jj)^T            (alpha=1, beta=0)
//       [gemm_nt]   update of the remaining columns of the outer block (K = 128)
//     [gemm_nt]     trailing update of everything right of the outer block (K = nb_outer):
//                   the N^3/3 term, MFMA-bound.
#include "gpak_internal.h"

#define PB 128
#define PLD 129  // LDS leading dimension (odd: row and column walks are both conflict-free)

// One workgroup factors the 128x128 block in LDS.
//   A      : block in global memory (column-major, ld); lower triangle is read
//   inv    : 2 x (128x128) doubles: inv(L) then inv(L)^T, column-major ld 128
//   col0   : global column of the block (for the not-positive-definite report)
//   info   : *info = min(*info or INT_MAX, first failing column 1-based)
__global__ __launch_bounds__(256) void gpak_potrf128_f64(double *A, long ld, double *__restrict__ inv,
                                                          int col0, int *info) {
  __shared__ double S[PB * PLD];
  const int t = threadIdx.x;
  // load (full block; the upper part is ignored by the algorithm)
  for (int e = t; e < PB * PB; e += 256) {
    int r = e & (PB - 1), c = e >> 7;
    S[r + c * PLD] = A[r + (size_t)c * ld];
  }
  __syncthreads();

  const int i = t & (PB - 1), half = t >> 7;
  for (int j = 0; j < PB; j++) {
    double piv = S[j + j * PLD];
    if (!(piv > 0.0)) {
      if (t == 0) atomicMin(info, col0 + j + 1);
      piv = 1.0;  // keep going with finite numbers; the host reports the failure
    }
    const double ljj = sqrt(piv);
    const double rinv = 1.0 / ljj;
    __syncthreads();  // everyone has read the pivot
    if (half == 0) {
      if (i == j) S[j + j * PLD] = ljj;
      else if (i > j) S[i + j * PLD] *= rinv;
    }
    __syncthreads();
    // trailing update: S[i][k] -= l_i * l_k for j < k <= i ; thread handles row i, every
    // second column
    if (i > j) {
      const double li = S[i + j * PLD];
      for (int k = j + 1 + half; k <= i; k += 2) S[i + k * PLD] -= li * S[k + j * PLD];
    }
    // (next iteration's first barrier orders these writes before anyone's column scaling;
    //  the pivot read above is of S[j+1][j+1], written only by thread row j+1 -> needs a barrier)
    __syncthreads();
  }

  // write L (upper part of the block zeroed so the stored matrix is cleanly lower)
  for (int e = t; e < PB * PB; e += 256) {
    int r = e & (PB - 1), c = e >> 7;
    A[r + (size_t)c * ld] = r >= c ? S[r + c * PLD] : 0.0;
  }
  __syncthreads();

  // in-place inverse of the lower-triangular block (LAPACK dtrti2 'L' order: last column first)
  for (int j = PB - 1; j >= 0; j--) {
    const double ajj = 1.0 / S[j + j * PLD];
    double y = 0.0;
    if (half == 0 && i > j) {
      for (int k = j + 1; k <= i; k++) y += S[i + k * PLD] * S[k + j * PLD];
    }
    __syncthreads();
    if (half == 0) {
      if (i == j) S[j + j * PLD] = ajj;
      else if (i > j) S[i + j * PLD] = -ajj * y;
    }
    __syncthreads();
  }
  double *invT = inv + PB * PB;
  for (int e = t; e < PB * PB; e += 256) {
    int r = e & (PB - 1), c = e >> 7;
    double v = r >= c ? S[r + c * PLD] : 0.0;
    inv[r + c * PB] = v;
  }
  for (int e = t; e < PB * PB; e += 256) {
    int r = e & (PB - 1), c = e >> 7;  // invT[r][c] = inv[c][r]
    invT[r + c * PB] = c >= r ? S[c + r * PLD] : 0.0;
  }
}

void gpak_launch_potrf128(hipStream_t st, double *A, long ld, double *inv, int col0, int *info) {
  hipLaunchKernelGGL(gpak_potrf128_f64, dim3(1), dim3(256), 0, st, A, ld, inv, col0, info);
}

int gpak_potrf_blocked(gpak_ctx *ctx) {
  const int Np = ctx->Np;
  const long ld = ctx->ld;
  double *M = ctx->dM;
  hipStream_t st = ctx->stream;
  int NB = ctx->nb_outer;
  if (NB < PB) NB = PB;
  NB = NB / PB * PB;
  const int init = 0x7fffffff;
  GPAK_HIP(hipMemcpyAsync(ctx->dInfo, &init, sizeof(int), hipMemcpyHostToDevice, st));

  size_t ev_used = 0;
  double tflops = 0.0;
  int tl = 0;
  for (int J = 0; J < Np; J += NB) {
    const int W = (Np - J) < NB ? (Np - J) : NB;
    for (int j = J; j < J + W; j += PB) {
      double *inv = ctx->dInv + (size_t)(j / PB) * 2 * PB * PB;
      gpak_launch_potrf128(st, M + j + (size_t)j * ld, ld, inv, j, ctx->dInfo);
      const int mt = (Np - j - PB) / PB;
      if (mt > 0) {
        double *P = M + (j + PB) + (size_t)j * ld;
        gpak_launch_gemm_nt(st, mt, 1, PB, 1.0, P, ld, inv, PB, 0.0, P, ld, 0, 0, false, false);
        const int nct = (J + W - (j + PB)) / PB;
        if (nct > 0)
          gpak_launch_gemm_nt(st, mt, nct, PB, -1.0, P, ld, P, ld, 1.0,
                              M + (j + PB) + (size_t)(j + PB) * ld, ld, 0, 0, true, false);
      }
    }
    const int mt = (Np - J - W) / PB;
    if (mt > 0) {
      double *P = M + (J + W) + (size_t)J * ld;
      double *Cc = M + (J + W) + (size_t)(J + W) * ld;
      if (ctx->profile) {
        while (ctx->ev_pool.size() < ev_used + 2) {
          hipEvent_t e;
          GPAK_HIP(hipEventCreate(&e));
          ctx->ev_pool.push_back(e);
        }
        GPAK_HIP(hipEventRecord(ctx->ev_pool[ev_used], st));
      }
      gpak_launch_gemm_nt(st, mt, mt, W, -1.0, P, ld, P, ld, 1.0, Cc, ld, 0, 0, true, true);
      if (ctx->profile) {
        GPAK_HIP(hipEventRecord(ctx->ev_pool[ev_used + 1], st));
        ev_used += 2;
      }
      // algorithmic flops of this launch: lower tiles only, 2*128*128*W each
      tflops += (double)mt * (mt + 1) / 2.0 * 2.0 * PB * PB * W;
      tl++;
    }
  }
  int info = 0;
  GPAK_HIP(hipMemcpyAsync(&info, ctx->dInfo, sizeof(int), hipMemcpyDeviceToHost, st));
  GPAK_HIP(hipStreamSynchronize(st));
  ctx->times.trailing_flops = tflops;
  ctx->times.trailing_launches = tl;
  ctx->times.trailing_ms = 0.0;
  if (ctx->profile) {
    double ms = 0.0;
    for (size_t e = 0; e < ev_used; e += 2) {
      float m = 0;
      GPAK_HIP(hipEventElapsedTime(&m, ctx->ev_pool[e], ctx->ev_pool[e + 1]));
      ms += m;
    }
    ctx->times.trailing_ms = ms;
  }
  if (info != init) {
    // padded columns (>= N) are identity and cannot fail
    ctx->failed_col = info;
    return GPAK_ENOTPD;
  }
  ctx->failed_col = 0;
  return GPAK_OK;
}
