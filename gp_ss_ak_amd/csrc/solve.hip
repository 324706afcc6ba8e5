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

// ---------------------------------------------------------------------------------------
// Two-level back substitution  out = L^-T z  (the second solve of solve_chol, GP_Utils.cpp:844):
// per 512-column block column J, from the last to the first,
//   1. gpak_coldot_split_f64: s[c] = sum_{i >= J+W} L[i, c] out[i] for its W columns -- full columns,
//      streamed once (Np^2/2 * 8 B over the whole solve), rows split over up to 8 workgroups per
//      column group so that ~512 workgroups are in flight;
//   2. gpak_trsv_bwd_diag_f64: ONE workgroup solves the W x W diagonal block: v = z_J - s, then for
//      each 128-column sub-block from the last: out_k = inv(L_kk)^T v_k, v_c -= L[k rows, c]^T out_k for
//      the earlier columns c of the block.
// 2 launches per 512 columns instead of 4 (one per 128 columns) that each re-read inv and
// hand off through global memory: 128 launches for N=32768 instead of 256.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gpak_coldot_split_f64(int Np, int i0, int J, int W, int rows_per_split,
                                                              const double *__restrict__ L, long ld,
                                                              const double *__restrict__ x, double *__restrict__ part,
                                                              int part_ld) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = J + blockIdx.x * 4 + w;
  if (c >= J + W) return;
  const int rb = i0 + blockIdx.y * rows_per_split;
  const int re = min(Np, rb + rows_per_split);
  const double *Lc = L + (size_t)c * ld;
  double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0;
  int i = rb + 2 * lane;
  for (; i + 128 < re; i += 256) {   // two independent 1-KiB pieces per trip
    const double2 l0 = *reinterpret_cast<const double2 *>(Lc + i), v0 = *reinterpret_cast<const double2 *>(x + i);
    const double2 l1 = *reinterpret_cast<const double2 *>(Lc + i + 128), v1 = *reinterpret_cast<const double2 *>(x + i + 128);
    a0 = fma(l0.x, v0.x, a0); a1 = fma(l0.y, v0.y, a1);
    b0 = fma(l1.x, v1.x, b0); b1 = fma(l1.y, v1.y, b1);
  }
  for (; i < re; i += 128) {
    const double2 l0 = *reinterpret_cast<const double2 *>(Lc + i), v0 = *reinterpret_cast<const double2 *>(x + i);
    a0 = fma(l0.x, v0.x, a0); a1 = fma(l0.y, v0.y, a1);
  }
  double a = (a0 + a1) + (b0 + b1);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off);
  if (lane == 0) part[(size_t)blockIdx.y * part_ld + (c - J)] = a;
}

#define BD_MAXW 512
__global__ __launch_bounds__(1024) void gpak_trsv_bwd_diag_f64(int J, int W, const double *__restrict__ L, long ld,
                                                                const double *__restrict__ inv,
                                                                const double *__restrict__ z,
                                                                const double *__restrict__ part, int nsplit,
                                                                int part_ld, double *__restrict__ out) {
  __shared__ double v[BD_MAXW], o[SB], ps[8][SB];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  if (t < W) {
    double s = 0.0;
    for (int r = 0; r < nsplit; r++) s += part[(size_t)r * part_ld + t];  // fixed order
    v[t] = z[J + t] - s;
  }
  __syncthreads();
  for (int k = W / SB - 1; k >= 0; k--) {
    const int j0 = J + k * SB;
    // out_k = inv(L_kk)^T v_k : 128 rows x 8 slices of 16 k'
    {
      const double *ibT = inv + (size_t)(j0 / SB) * 2 * SB * SB + SB * SB;  // inv(L_kk)^T, column-major
      const int i = t & (SB - 1), sl = t >> 7;
      double s = 0.0;
#pragma unroll
      for (int q = 0; q < 16; q++) s = fma(ibT[i + (size_t)(16 * sl + q) * SB], v[k * SB + 16 * sl + q], s);
      ps[sl][i] = s;
    }
    __syncthreads();
    if (t < SB) {
      double s = 0.0;
#pragma unroll
      for (int q = 0; q < 8; q++) s += ps[q][t];
      o[t] = s;
      out[j0 + t] = s;
    }
    __syncthreads();
    // v_c -= sum_r L[j0 + r, J + c] o[r] for the earlier columns of the block: a wave per column
    const double o0 = o[2 * lane], o1 = o[2 * lane + 1];
    // k * 128 columns over 16 waves = 8 k per wave: all of a wave's loads in flight together
    for (int cb = 0; cb < k; cb++) {
      double2 l[8];
#pragma unroll
      for (int u = 0; u < 8; u++)
        l[u] = *reinterpret_cast<const double2 *>(L + j0 + 2 * lane + (size_t)(J + cb * SB + w + 16 * u) * ld);
#pragma unroll
      for (int u = 0; u < 8; u++) {
        double s = l[u].x * o0 + l[u].y * o1;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
        if (lane == 0) v[cb * SB + w + 16 * u] -= s;
      }
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------
// Explicit inverse of a W x W (W <= 512) diagonal block, R = (L_bb^-1)^T, column-major with leading
// dimension 512: forward substitution on an identity whose ROWS are the right-hand sides, 128 columns
// at a time with the pre-inverted 128 x 128 blocks (same two products as every other panel step).
// Seven tiny GEMM launches per block, issued on the side stream that carries the forward substitution,
// i.e. hidden behind the factorisation.  It turns the diagonal step of the back substitution from four
// dependent 128-column phases in one workgroup (46 us) into one matrix-vector product.
// ---------------------------------------------------------------------------------------
__global__ void gpak_identity_w_f64(double *R, int ld, int W) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= W * ld) return;
  const int c = i / ld, r = i - c * ld;
  R[i] = (r == c) ? 1.0 : 0.0;
}
__global__ __launch_bounds__(256) void gpak_transpose_f64(int rows, int cols, const double *__restrict__ src, long lds,
                                                           double *__restrict__ dst, long ldd) {
  __shared__ double tile[32][33];
  const int r0 = blockIdx.x * 32, c0 = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int k = ty; k < 32; k += 8)
    if (r0 + tx < rows && c0 + k < cols) tile[k][tx] = src[(r0 + tx) + (size_t)(c0 + k) * lds];
  __syncthreads();
  for (int k = ty; k < 32; k += 8)
    if (c0 + tx < cols && r0 + k < rows) dst[(c0 + tx) + (size_t)(r0 + k) * ldd] = tile[tx][k];
}

// `extra` more right-hand-side rows ride along below the identity (RL >= W + extra): the rows of
// L[J..J+W, J-extra..J)^T, which the same substitution turns into T = L[b, b-1]^T L_bb^-T, the coupling block of the
// one-launch back-substitution step -- no launch of its own, and the seven products are latency-bound at either height
void gpak_launch_diag_inverse(hipStream_t st, int J, int W, const double *L, long ld, const double *inv, double *R,
                              int RL, int extra) {
  // RL: leading dimension of R = the block width of the caller's back substitution (512 unless told otherwise)
  hipLaunchKernelGGL(gpak_identity_w_f64, dim3((W * RL + 255) / 256), dim3(256), 0, st, R, RL, W);
  if (extra > 0)
    hipLaunchKernelGGL(gpak_transpose_f64, dim3((W + 31) / 32, (extra + 31) / 32), dim3(256), 0, st, W, extra,
                       L + J + (size_t)(J - extra) * ld, ld, R + W, (long)RL);
  const int mt = (W + extra) / SB;
  for (int j0 = 0; j0 < W; j0 += SB) {
    const double *ib = inv + (size_t)((J + j0) / SB) * 2 * SB * SB;
    double *Rj = R + (size_t)j0 * RL;
    gpak_launch_gemm_nt(st, mt, 1, SB, 1.0, Rj, RL, ib, SB, 0.0, Rj, RL, 0, 0, false, false);
    const int nin = (W - j0 - SB) / SB;
    if (nin > 0)
      gpak_launch_gemm_nt(st, mt, nin, SB, -1.0, Rj, RL, L + (J + j0 + SB) + (size_t)(J + j0) * ld, ld, 1.0,
                          R + (size_t)(j0 + SB) * RL, RL, 0, 0, false, false);
  }
}

// part[cg][i] = sum_{c in column group cg} R[i, c] * v[c],  v = z_J - sum of the column-dot partials
#define MV_CG 32
__global__ __launch_bounds__(256) void gpak_bwd_diag_mv_f64(int J, int W, const double *__restrict__ R, int RL,
                                                             const double *__restrict__ z,
                                                             const double *__restrict__ spart, int nsplit,
                                                             int spart_ld, double *__restrict__ part, int part_ld) {
  __shared__ double v[MV_CG];
  const int t = threadIdx.x, c0 = blockIdx.y * MV_CG;
  if (t < MV_CG) {
    double s = 0.0;
    for (int r = 0; r < nsplit; r++) s += spart[(size_t)r * spart_ld + c0 + t];
    v[t] = z[J + c0 + t] - s;
  }
  __syncthreads();
  const int i = blockIdx.x * 256 + t;
  if (i >= W) return;
  double x[MV_CG];
#pragma unroll
  for (int c = 0; c < MV_CG; c++) x[c] = R[i + (size_t)(c0 + c) * RL];   // zero below the diagonal (i > c)
  double a = 0.0;
#pragma unroll
  for (int c = 0; c < MV_CG; c++) a = fma(x[c], v[c], a);
  part[(size_t)blockIdx.y * part_ld + i] = a;
}

// one block column [J, J+W), W <= NB.  scratch: (8 + NB / 32) * NB doubles.  Rinv: the block's explicit inverse
// (gpak_launch_diag_inverse with RL = NB) or nullptr (then the four-phase single-workgroup kernel is used: W <= 512).
void gpak_launch_trsv_bwd_block2(hipStream_t st, int Np, int J, int W, const double *L, long ld, const double *inv,
                                 const double *z, double *out, double *scratch, const double *Rinv, int NB) {
  const int rows = Np - (J + W);
  int R = 0, per = 0;
  if (rows > 0) {
    R = (rows + 4095) / 4096;
    if (R > 8) R = 8;
    per = ((rows + R - 1) / R + 127) / 128 * 128;
    hipLaunchKernelGGL(gpak_coldot_split_f64, dim3(W / 4, R), dim3(256), 0, st, Np, J + W, J, W, per, L, ld, out, scratch,
                       NB);
  }
  if (Rinv) {
    double *mvpart = scratch + 8 * NB;
    hipLaunchKernelGGL(gpak_bwd_diag_mv_f64, dim3((W + 255) / 256, W / MV_CG), dim3(256), 0, st, J, W, Rinv, NB, z, scratch,
                       R, NB, mvpart, NB);
    gpak_launch_sum_splits(st, mvpart, NB, W / MV_CG, W, out + J);
  } else {
    hipLaunchKernelGGL(gpak_trsv_bwd_diag_f64, dim3(1), dim3(1024), 0, st, J, W, L, ld, inv, z, scratch, R, NB, out);
  }
}

// scratch: (8 + NB / 32) * NB doubles.  RinvB: explicit inverses of the NB-column diagonal blocks (NB x NB each, NB =
// bw: 512, 1024 or 2048), or nullptr: then 512-column blocks with the four-phase diagonal kernel
void gpak_launch_trsv_bwd2(hipStream_t st, int Np, const double *L, long ld, const double *inv, const double *z,
                           double *out, double *scratch, const double *RinvB, int bw) {
  const int NB = RinvB ? bw : BD_MAXW;
  const int nJ = (Np + NB - 1) / NB;
  for (int b = nJ - 1; b >= 0; b--) {
    const int J = b * NB, W = min(NB, Np - J);
    gpak_launch_trsv_bwd_block2(st, Np, J, W, L, ld, inv, z, out, scratch, RinvB ? RinvB + (size_t)b * NB * NB : nullptr,
                                NB);
  }
}

// ---------------------------------------------------------------------------------------
// Back substitution with the column dots of the NEXT block column under the diagonal step of this one (round 3).
// The column dots of block b-1 split into the rows of block b ("near": they need out_b, the newest piece of the
// solution) and everything below ("far": known one step earlier, and nearly all of the bytes).  Step b:
//   launch A(b)   workgroups [0, nmv): 16 rows each of R_b times v_b, v_b = z_b - far_b - near_b     -> out_b
//                 the others:          far_{b-1}[c] = sum_{i >= J_b + W_b} L[i, c] out[i]   (rows split as in
//                                      gpak_coldot_split_f64) -- the HBM-bound part, nothing in it waits for out_b
//   launch B(b)   near_{b-1}[c] = sum_{i in block b} L[i, c] out_b[i]                        (2 MB, bwd_fused = 1)
// With the coupling block T_b = L[b, b-1]^T R_b (built beside the factorisation, bwd_fused = 2) near_{b-1} = T_b v_b is
// a function of the same v_b, the rows of T_b join those of R_b in launch A, and B disappears: one launch per step,
// every workgroup of which depends on EARLIER launches only.
// The matrix-vector part reads R_b / T_b as they are built (column-major, rows along the lanes): a workgroup takes 16
// rows, a lane one row and every fourth column of its wave's quarter, with all 32 loads of a 512-column block in flight
// at once (64 rows per workgroup and four loads in flight: 28 us per step instead of 4).
// far / near are double-buffered by the parity of b (a launch reads the set its predecessor wrote).
// ---------------------------------------------------------------------------------------
#define BS_MAXW 2048
#define BS_ROWS 16   // operator rows per workgroup (4 per wave)
#define BS_SPLITS 8  // most row splits of the far column dots
template <int FU>
__global__ __launch_bounds__(256) void gpak_bwd_step_f64(int Np, int J, int W, int Jp, int Wp, const double *__restrict__ R,
                                                          const double *__restrict__ T, int RL, const double *__restrict__ z,
                                                          const double *__restrict__ far_in, int nsplit_in,
                                                          const double *__restrict__ near_in, int part_ld, double *out,
                                                          double *__restrict__ near_out, double *__restrict__ far_out,
                                                          int rows_per_split, int nsplit_out, int nmv,
                                                          const double *__restrict__ L, long ld) {
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  if ((int)blockIdx.x < nmv) {
    __shared__ double v[BS_MAXW];
    for (int c = t; c < W; c += 256) {
      double p[BS_SPLITS];   // all loads issued before the first add (a loop over nsplit_in waits for each in turn)
#pragma unroll
      for (int r = 0; r < BS_SPLITS; r++) p[r] = r < nsplit_in ? far_in[(size_t)r * part_ld + c] : 0.0;
      const double nr = near_in ? near_in[c] : 0.0, zc = z[J + c];
      double s = 0.0;
#pragma unroll
      for (int r = 0; r < BS_SPLITS; r++) s += p[r];   // fixed order
      v[c] = zc - (s + nr);
    }
    __syncthreads();
    __shared__ double ps[4][BS_ROWS];
    const int r0 = blockIdx.x * BS_ROWS, row = lane & 15, ph = lane >> 4;
    const double *M = (r0 < W ? R + r0 : T + (r0 - W)) + row;   // 16 rows of the stacked [R_b ; T_b]
    const int cq = W / 4, cb = w * cq + ph;                    // this wave's quarter of the columns, every fourth one
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    // R_b is upper triangular: a quarter of the columns that lies left of these rows is all zeros
    const int c_end = (r0 < W && (w + 1) * cq <= r0) ? 0 : cq;
    for (int c0 = 0; c0 < c_end; c0 += 128) {
      double m[32];
#pragma unroll
      for (int u = 0; u < 32; u++) m[u] = (c0 + 4 * u < cq) ? M[(size_t)(cb + c0 + 4 * u) * RL] : 0.0;
#pragma unroll
      for (int u = 0; u < 32; u++)
        if (c0 + 4 * u < cq) acc[u & 3] = fma(m[u], v[cb + c0 + 4 * u], acc[u & 3]);
    }
    double a = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    a += __shfl_xor(a, 16);
    a += __shfl_xor(a, 32);
    if (lane < BS_ROWS) ps[w][lane] = a;
    __syncthreads();
    if (t < BS_ROWS) {
      const double r = (ps[0][t] + ps[1][t]) + (ps[2][t] + ps[3][t]);
      if (r0 < W) out[J + r0 + t] = r;
      else near_out[r0 - W + t] = r;
    }
    return;
  }
  const int g = blockIdx.x - nmv, ngrp = Wp / 4;
  const int c = Jp + (g % ngrp) * 4 + w, split = g / ngrp;
  if (split >= nsplit_out) return;
  const int rb = J + W + split * rows_per_split;
  const int re = min(Np, rb + rows_per_split);
  const double *Lc = L + (size_t)c * ld;
  double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0;
  int i = rb + 2 * lane;
  if (FU == 4) {
    double c0 = 0.0, c1 = 0.0, d0 = 0.0, d1 = 0.0;
    for (; i + 384 < re; i += 512) {   // four independent 1-KiB pieces per trip
      const double2 l0 = *reinterpret_cast<const double2 *>(Lc + i), l1 = *reinterpret_cast<const double2 *>(Lc + i + 128);
      const double2 l2 = *reinterpret_cast<const double2 *>(Lc + i + 256), l3 = *reinterpret_cast<const double2 *>(Lc + i + 384);
      const double2 v0 = *reinterpret_cast<const double2 *>(out + i), v1 = *reinterpret_cast<const double2 *>(out + i + 128);
      const double2 v2 = *reinterpret_cast<const double2 *>(out + i + 256), v3 = *reinterpret_cast<const double2 *>(out + i + 384);
      a0 = fma(l0.x, v0.x, a0); a1 = fma(l0.y, v0.y, a1);
      b0 = fma(l1.x, v1.x, b0); b1 = fma(l1.y, v1.y, b1);
      c0 = fma(l2.x, v2.x, c0); c1 = fma(l2.y, v2.y, c1);
      d0 = fma(l3.x, v3.x, d0); d1 = fma(l3.y, v3.y, d1);
    }
    a0 += c0; a1 += c1; b0 += d0; b1 += d1;
  }
  for (; i + 128 < re; i += 256) {   // two independent 1-KiB pieces per trip
    const double2 l0 = *reinterpret_cast<const double2 *>(Lc + i), v0 = *reinterpret_cast<const double2 *>(out + i);
    const double2 l1 = *reinterpret_cast<const double2 *>(Lc + i + 128), v1 = *reinterpret_cast<const double2 *>(out + i + 128);
    a0 = fma(l0.x, v0.x, a0); a1 = fma(l0.y, v0.y, a1);
    b0 = fma(l1.x, v1.x, b0); b1 = fma(l1.y, v1.y, b1);
  }
  for (; i < re; i += 128) {
    const double2 l0 = *reinterpret_cast<const double2 *>(Lc + i), v0 = *reinterpret_cast<const double2 *>(out + i);
    a0 = fma(l0.x, v0.x, a0); a1 = fma(l0.y, v0.y, a1);
  }
  double a = (a0 + a1) + (b0 + b1);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off);
  if (lane == 0) far_out[(size_t)split * part_ld + (c - Jp)] = a;
}

// near[c] = sum_{i < W} L[J + i, Jp + c] out[J + i],  c < Wp: one wave per column
__global__ __launch_bounds__(256) void gpak_bwd_near_f64(int J, int W, int Jp, const double *__restrict__ L, long ld,
                                                          const double *__restrict__ out, double *__restrict__ near_out) {
  const int lane = threadIdx.x & 63, c = blockIdx.x * 4 + (threadIdx.x >> 6);
  const double *Lc = L + J + (size_t)(Jp + c) * ld;
  double a0 = 0.0, a1 = 0.0;
  for (int i = 2 * lane; i < W; i += 128) {
    const double2 l = *reinterpret_cast<const double2 *>(Lc + i), x = *reinterpret_cast<const double2 *>(out + J + i);
    a0 = fma(l.x, x.x, a0); a1 = fma(l.y, x.y, a1);
  }
  double a = a0 + a1;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off);
  if (lane == 0) near_out[c] = a;
}

// scratch: (2 * BS_SPLITS + 2) * NB doubles (two sets of far partials + near).  OP: per block column (stride op_stride,
// leading dimension RL) R_b as built by gpak_launch_diag_inverse and, with_T, T_b in the rows below it (`extra` = NB):
// then one launch per step, else near_{b-1} has a launch of its own
void gpak_launch_trsv_bwd3(hipStream_t st, int Np, const double *L, long ld, const double *z, double *out, double *scratch,
                           const double *OP, size_t op_stride, int RL, bool with_T, int NB) {
  const int nJ = (Np + NB - 1) / NB;
  double *farb[2] = {scratch, scratch + BS_SPLITS * (size_t)NB},
         *nearb[2] = {scratch + 2 * BS_SPLITS * (size_t)NB, scratch + (2 * BS_SPLITS + 1) * (size_t)NB};
  // four 1-KiB pieces in flight per wave and at most 8 row splits of >= 4096 rows: measured against 2 pieces and 4 / 16
  // splits (profiles/r03_bwd_fused.txt: 1.02-1.05 ms against 1.05-1.08, 1.08-1.11 and 1.11-1.13 at N = 32768)
  int nsplit_in = 0;
  for (int b = nJ - 1; b >= 0; b--) {
    const int J = b * NB, W = min(NB, Np - J);
    const int Jp = b > 0 ? J - NB : 0, Wp = b > 0 ? NB : 0;
    const int rows = Np - (J + W);
    int Rs = 0, per = 128;
    if (rows > 0 && b > 0) {
      Rs = min(8, (rows + 4095) / 4096);
      per = ((rows + Rs - 1) / Rs + 127) / 128 * 128;
    }
    const int nmv = (W + (with_T ? Wp : 0)) / BS_ROWS;
    const double *Rb = OP + (size_t)b * op_stride;
    hipLaunchKernelGGL(gpak_bwd_step_f64<4>, dim3(nmv + (Wp / 4) * Rs), dim3(256), 0, st, Np, J, W, Jp, Wp, Rb, Rb + W, RL, z,
                       farb[b & 1], nsplit_in, b < nJ - 1 ? nearb[b & 1] : (const double *)nullptr, NB, out, nearb[(b + 1) & 1],
                       farb[(b + 1) & 1], per, Rs, nmv, L, ld);
    if (!with_T && b > 0)
      hipLaunchKernelGGL(gpak_bwd_near_f64, dim3(Wp / 4), dim3(256), 0, st, J, W, Jp, L, ld, out, nearb[(b + 1) & 1]);
    nsplit_in = Rs;
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
