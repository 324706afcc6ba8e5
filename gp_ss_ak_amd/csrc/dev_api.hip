// dev_api.hip -- device-pointer level C-ABI (include/gpak_dev.h): what one rank of the
// block-column-cyclic multi-GPU factorisation runs on the block columns it owns.
#include <cmath>
#include <cstring>
#include <vector>

#include "../../include/gpak_dev.h"
#include "gpak_internal.h"

void gpak_build_siginv(const double *e, double *A);

// view of points [off, n) of a transformed set: same array stride (cap), shifted base
static DevPoints as_points(const double *u, int cap, int n, int off = 0) {
  DevPoints p;
  p.base = const_cast<double *>(u) + off;
  p.n = n - off;
  p.cap = cap;
  return p;
}
static KernParams make_kp(const double *expans, double bias, int mode, const double *mu) {
  KernParams kp;
  memset(&kp, 0, sizeof(kp));
  if (mode & GPAK_DIST_HYB) {   // a serialized composition (gpak_dev.h)
    const int nterms = (int)expans[0];
    int kinds[GPAK_MAX_TERMS] = {0, 0, 0};
    for (int t = 0; t < GPAK_MAX_TERMS; t++) kinds[t] = (int)expans[1 + t];
    if (gpak_build_kp(nterms, kinds, expans + 5, bias, expans[4], mode & 0xF, &kp, nullptr) != GPAK_OK) kp.nterms = 0;
    for (int k = 0; k < 4; k++) kp.mu[k] = mu ? mu[k] : 0.0;
    kp.d = (mode & GPAK_DIST_D4) ? 4 : 3;
    return kp;
  }
  kp.nterms = 1;
  gpak_build_siginv(expans, kp.term[0].A);
  kp.term[0].var2 = expans[6] * expans[6];
  kp.term[0].profile = GPAK_PROFILE_EXPSQRT;
  for (int k = 0; k < 4; k++) kp.mu[k] = mu ? mu[k] : 0.0;
  kp.term[0].a33 = expans[7];   // InversewidthR: the 4th column's own inverse width (Kernel.cpp:1421-1424)
  kp.d = (mode & GPAK_DIST_D4) ? 4 : 3;
  kp.bias = bias;
  kp.mode = mode & 0xF;
  return kp;
}
static int status() { return hipGetLastError() == hipSuccess ? GPAK_OK : GPAK_EHIP; }

__global__ void gpak_pack_f64(const double *__restrict__ src, long ld, int row0, int nrows, double *__restrict__ dst) {
  const int c = blockIdx.y;
  const int r = 2 * (blockIdx.x * blockDim.x + threadIdx.x);
  if (r < nrows)   // nrows, row0, ld are even: 16-B accesses
    *reinterpret_cast<double2 *>(dst + (size_t)c * nrows + r) =
        *reinterpret_cast<const double2 *>(src + (size_t)c * ld + row0 + r);
}

// y[0..nrows) += A x, A nrows x W column-major: one thread per row (coalesced along the rows), x in LDS
__global__ __launch_bounds__(256) void gpak_gemv_n_add_f64(const double *__restrict__ A, long ld, int nrows, int W,
                                                           const double *__restrict__ x, double *__restrict__ y) {
  __shared__ double xs[512];
  for (int c = threadIdx.x; c < W; c += 256) xs[c] = x[c];
  __syncthreads();
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= nrows) return;
  double s0 = 0.0, s1 = 0.0;
  int c = 0;
  for (; c + 1 < W; c += 2) {
    s0 = fma(A[r + (size_t)c * ld], xs[c], s0);
    s1 = fma(A[r + (size_t)(c + 1) * ld], xs[c + 1], s1);
  }
  if (c < W) s0 = fma(A[r + (size_t)c * ld], xs[c], s0);
  y[r] += s0 + s1;
}
// y[c] = sum_r A[r, c] x[r]: one workgroup per column, fixed-order tree reduction (deterministic)
__global__ __launch_bounds__(256) void gpak_gemv_t_f64(const double *__restrict__ A, long ld, int nrows,
                                                       const double *__restrict__ x, double *__restrict__ y) {
  __shared__ double red[256];
  const double *col = A + (size_t)blockIdx.x * ld;
  double s = 0.0;
  for (int r = threadIdx.x; r < nrows; r += 256) s = fma(col[r], x[r], s);
  red[threadIdx.x] = s;
  __syncthreads();
  for (int h = 128; h > 0; h >>= 1) {
    if (threadIdx.x < h) red[threadIdx.x] += red[threadIdx.x + h];
    __syncthreads();
  }
  if (threadIdx.x == 0) y[blockIdx.x] = red[0];
}

extern "C" {

int gpak_dev_pack(void *stream, const double *src, long ld, int row0, int nrows, int ncols, double *dst) {
  if (nrows <= 0 || ncols <= 0) return GPAK_OK;
  if ((nrows | row0 | ld) & 1) return GPAK_EINVAL;
  hipLaunchKernelGGL(gpak_pack_f64, dim3((nrows / 2 + 255) / 256, ncols), dim3(256), 0, (hipStream_t)stream, src, ld, row0,
                     nrows, dst);
  return status();
}

int gpak_dev_transform(void *stream, const double *x, int xs, int n, int cap, const double *expans,
                       const double *mu, double *u) {
  // always four columns: a 3-D input set carries a zero 4th column (and mu[3] = 0), whose image is 0
  KernParams kp = make_kp(expans, 0.0, GPAK_DIST_DIRECT | GPAK_DIST_D4, mu);
  DevPoints p = as_points(u, cap, n);
  gpak_launch_transform((hipStream_t)stream, x, xs, n, kp, p);
  return status();
}

int gpak_dev_transform_k(void *stream, const double *x, int xs, int n, int cap, const double *kern, int dist_mode,
                         const double *mu, double *u) {
  KernParams kp = make_kp(kern, 0.0, (dist_mode & GPAK_DIST_HYB) | GPAK_DIST_DIRECT | GPAK_DIST_D4, mu);
  if (kp.nterms < 1) return GPAK_EINVAL;
  DevPoints p = as_points(u, cap, n);
  gpak_launch_transform((hipStream_t)stream, x, xs, n, kp, p);
  return status();
}

int gpak_dev_fill_b(void *stream, const double *u, int cap, int n, int Np, int J, int W, const double *expans,
                    double bias, double sn2, int dist_mode, double *blk, long ld) {
  KernParams kp = make_kp(expans, bias, dist_mode, nullptr);
  DevPoints P = as_points(u, cap, n), Q = as_points(u, cap, n, J);
  gpak_launch_fill((hipStream_t)stream, P, Q, Np, W, kp, 1.0 / sn2, 1.0, 1.0, 1, blk, ld, nullptr, J);
  return status();
}

int gpak_dev_factor_panel_co(void *stream, double *blk, long ld, int Np, int J, int W, double *inv, int *info,
                             int coresident) {
  // virtual bases: global (row, column) addressing that only ever touches columns [J, J+W)
  double *Mv = blk - (size_t)J * ld;
  double *invv = inv - (size_t)(J / GPAK_TILE) * 2 * GPAK_TILE * GPAK_TILE;
  gpak_factor_panel((hipStream_t)stream, Mv, ld, Np, J, W, invv, info, true, coresident != 0);  // caller's inv may be uninitialised
  return status();
}
int gpak_dev_factor_panel(void *stream, double *blk, long ld, int Np, int J, int W, double *inv, int *info) {
  return gpak_dev_factor_panel_co(stream, blk, ld, Np, J, W, inv, info, 0);
}

int gpak_dev_update_block(void *stream, const double *panel, long ldp, int prow0, int W, double *blk, long ld,
                          int Np, int Jc, int Wc) {
  const int mt = (Np - Jc) / GPAK_TILE, nt = Wc / GPAK_TILE;
  const double *P = panel + (Jc - prow0);
  // C[rows >= Jc, cols of c] -= P[rows >= Jc] * P[rows of c]^T, lower tiles only
  gpak_launch_gemm_nt((hipStream_t)stream, mt, nt, W, -1.0, P, ldp, P, ldp, 1.0, blk + Jc, ld, 0, 0, true, true);
  return status();
}

int gpak_dev_update_cyclic(void *stream, const double *panel, long ldp, int prow0, int W, double *local, long ld,
                           int Np, int nb, int P, int rank, int lb0, int n_local_blocks, int last_width) {
  // local tile columns [lb0*tpb, total): rows from the first included block's global start
  const int tpb = nb / GPAK_TILE;
  const int lt0 = lb0 * tpb;
  const int total_tiles = (n_local_blocks - 1) * tpb + last_width / GPAK_TILE;
  const int nt = total_tiles - lt0;
  if (nt <= 0) return GPAK_OK;
  const int first_global_block = lb0 * P + rank;
  const int rt0 = first_global_block * tpb;
  const int mt = Np / GPAK_TILE - rt0;
  const double *Pv = panel - prow0;  // virtual base: global row g of the panel is Pv[g + k*ldp]
  // C base: row 0 of local tile column lt0; the kernel adds the GLOBAL row tile (art) itself
  gpak_launch_gemm_cyclic((hipStream_t)stream, mt, nt, W, Pv, ldp, local + (size_t)lt0 * GPAK_TILE * ld, ld, rt0, P,
                          rank, tpb, lt0);
  return status();
}

int gpak_dev_trsv_fwd_block(void *stream, const double *blk, long ld, int Np, int J, int W, const double *inv,
                            double *x, double *out) {
  const double *Lv = blk - (size_t)J * ld;
  const double *invv = inv - (size_t)(J / GPAK_TILE) * 2 * GPAK_TILE * GPAK_TILE;
  gpak_launch_trsv_fwd_block((hipStream_t)stream, Np, J, W, Lv, ld, invv, x, out);
  return status();
}

int gpak_dev_coldot(void *stream, const double *blk, long ld, int Np, int J, int W, const double *x, double *s) {
  const double *Lv = blk - (size_t)J * ld;
  gpak_launch_coldot((hipStream_t)stream, Np, J + W, J, W, Lv, ld, x, s);
  return status();
}

int gpak_dev_trsv_bwd_block(void *stream, const double *blk, long ld, int J, int W, const double *inv, double *x,
                            double *out) {
  const double *Lv = blk - (size_t)J * ld;
  const double *invv = inv - (size_t)(J / GPAK_TILE) * 2 * GPAK_TILE * GPAK_TILE;
  gpak_launch_trsv_bwd_block((hipStream_t)stream, J, W, Lv, ld, invv, x, out);
  return status();
}

int gpak_dev_trsv_bwd_packed(void *stream, const double *panel, long ldp, int row0, int Np, int J, int W,
                             const double *inv, const double *z, double *scratch, double *out, const double *rinv) {
  if (W <= 0 || W > 512 || (W % GPAK_TILE)) return GPAK_EINVAL;
  const double *Lv = panel - row0 - (size_t)J * ldp;   // L[r, c] = Lv[r + c * ldp], global r and c
  const double *invv = inv - (size_t)(J / GPAK_TILE) * 2 * GPAK_TILE * GPAK_TILE;
  gpak_launch_trsv_bwd_block2((hipStream_t)stream, Np, J, W, Lv, ldp, invv, z, out, scratch, rinv);
  return status();
}

int gpak_dev_diag_inverse(void *stream, const double *panel, long ldp, int row0, int J, int W, const double *inv,
                          double *rinv) {
  if (W <= 0 || W > 512 || (W % GPAK_TILE) || !rinv) return GPAK_EINVAL;
  const double *Lv = panel - row0 - (size_t)J * ldp;
  const double *invv = inv - (size_t)(J / GPAK_TILE) * 2 * GPAK_TILE * GPAK_TILE;
  gpak_launch_diag_inverse((hipStream_t)stream, J, W, Lv, ldp, invv, rinv);
  return status();
}

int gpak_dev_logdiag_block(void *stream, const double *blk, long ld, int J, int W, int N, double *out) {
  const double *Lv = blk - (size_t)J * ld;
  gpak_launch_logdiag_block((hipStream_t)stream, J, W, N, Lv, ld, out);
  return status();
}

int gpak_dev_kmatvec(void *stream, const double *u, int cap, int n, int i0, int i1, const double *w,
                     const double *expans, double bias, int dist_mode, double *scratch, double *out) {
  KernParams kp = make_kp(expans, bias, dist_mode, nullptr);
  DevPoints Q = as_points(u, cap, n);
  int splits = gpak_kmatvec_splits(i1 - i0, Q.n);
  gpak_launch_kmatvec((hipStream_t)stream, Q, i0, i1 - i0, w + i0, Q, kp, scratch, splits, out);  // source points [i0, i1)
  return status();
}

int gpak_dev_nlz_terms(void *stream, int N, const double *y, const double *f, const double *alpha, double sn2,
                       double *out) {
  // the launcher writes red[1], red[2]; shift so that they land in out[0], out[1]
  gpak_launch_nlz_terms((hipStream_t)stream, N, y, f, alpha, sn2, out - 1);
  return status();
}

// ---- pieces of the row-block x column-block layout (include/gpak_dev.h) --------------------------------------
int gpak_dev_fill_rect(void *stream, const double *u, int cap, int n, int row0, int nrows, int col0, int ncols,
                       const double *expans, double bias, double sn2, int dist_mode, double *dst, long ld) {
  if (nrows <= 0 || ncols <= 0) return GPAK_OK;
  if (nrows % GPAK_TILE || ncols % 64) return GPAK_EINVAL;
  KernParams kp = make_kp(expans, bias, dist_mode, nullptr);
  DevPoints P = as_points(u, cap, n, row0), Q = as_points(u, cap, n, col0);
  // the diagonal of the matrix is where (row0 + a) == (col0 + c): col_off = col0 - row0 in the fill's own coordinates
  gpak_launch_fill((hipStream_t)stream, P, Q, nrows, ncols, kp, 1.0 / sn2, 1.0, 1.0, row0 == col0 ? 1 : 0, dst, ld, nullptr,
                   col0 - row0);
  return status();
}

int gpak_dev_solve_rows(void *stream, double *P, long ld, int nrows, int W, const double *Lbb, long ldl,
                        const double *inv) {
  if (nrows <= 0) return GPAK_OK;
  if (nrows % GPAK_TILE || W <= 0 || W % GPAK_TILE || W > 512) return GPAK_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int mt = nrows / GPAK_TILE;
  for (int s = 0; s < W / GPAK_TILE; s++) {
    double *Ps = P + (size_t)s * GPAK_TILE * ld;
    // P_s := P_s inv(D_s)^T: an A B^T product with B = the FIRST image of the block's pair (D_s^-1), as in gpak_factor_panel
    gpak_launch_gemm_nt(st, mt, 1, GPAK_TILE, 1.0, Ps, ld, inv + (size_t)s * 2 * GPAK_TILE * GPAK_TILE, GPAK_TILE, 0.0, Ps,
                        ld, 0, 0, false, false);
    const int nct = W / GPAK_TILE - (s + 1);
    if (nct > 0)   // P[:, s+1..] -= P_s Lbb[s+1.., s]^T
      gpak_launch_gemm_nt(st, mt, nct, GPAK_TILE, -1.0, Ps, ld, Lbb + (size_t)(s + 1) * GPAK_TILE + (size_t)s * GPAK_TILE * ldl,
                          ldl, 1.0, P + (size_t)(s + 1) * GPAK_TILE * ld, ld, 0, 0, false, false);
  }
  return status();
}

int gpak_dev_update_rect(void *stream, const double *A, long lda, const double *B, long ldb, int K, double *C, long ldc,
                         int mrows, int ncols, int diag_first) {
  if (mrows <= 0 || ncols <= 0) return GPAK_OK;
  if (mrows % GPAK_TILE || ncols % GPAK_TILE || K % GPAK_TILE) return GPAK_EINVAL;
  // diag_first: tile (ti, tj) of the leading square is skipped when ti < tj (row_block0 = col_block0 = 0); otherwise
  // every tile takes part
  gpak_launch_gemm_nt((hipStream_t)stream, mrows / GPAK_TILE, ncols / GPAK_TILE, K, -1.0, A, lda, B, ldb, 1.0, C, ldc, 0, 0,
                      diag_first != 0, true);
  return status();
}

int gpak_dev_gemv_n_add(void *stream, const double *A, long ld, int nrows, int W, const double *x, double *y) {
  if (nrows <= 0 || W <= 0) return GPAK_OK;
  if (W > 512) return GPAK_EINVAL;
  hipLaunchKernelGGL(gpak_gemv_n_add_f64, dim3((nrows + 255) / 256), dim3(256), 0, (hipStream_t)stream, A, ld, nrows, W, x, y);
  return status();
}

int gpak_dev_gemv_t(void *stream, const double *A, long ld, int nrows, int W, const double *x, double *y) {
  if (W <= 0) return GPAK_OK;
  hipLaunchKernelGGL(gpak_gemv_t_f64, dim3(W), dim3(256), 0, (hipStream_t)stream, A, ld, nrows > 0 ? nrows : 0, x, y);
  return status();
}

int gpak_dev_vec_axpy(void *stream, int n, double a, const double *x, double *y) {
  gpak_launch_axpy((hipStream_t)stream, n, a, x, y);
  return status();
}

int gpak_dev_stream_create(int skip_cus, void **stream_out) {
  if (!stream_out || skip_cus < 0) return GPAK_EINVAL;
  *stream_out = nullptr;
  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return GPAK_EHIP;
  hipStream_t s = nullptr;
  if (skip_cus == 0 || skip_cus >= prop.multiProcessorCount) {
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return GPAK_EHIP;
  } else {
    std::vector<uint32_t> mask((prop.multiProcessorCount + 31) / 32, 0xffffffffu);
    for (int c = 0; c < skip_cus; c++) mask[c / 32] &= ~(1u << (c % 32));
    if (hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()) != hipSuccess) return GPAK_EHIP;
  }
  *stream_out = s;
  return GPAK_OK;
}

int gpak_dev_stream_destroy(void *stream) {
  return (stream && hipStreamDestroy((hipStream_t)stream) == hipSuccess) ? GPAK_OK : GPAK_EHIP;
}

}  // extern "C"
