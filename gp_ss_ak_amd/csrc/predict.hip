// predict.hip -- predictive mean / variance and multi-right-hand-side solve_chol for gfx950.
//
// Replaces GP_utils::_ComputeK_NewData / _postMean / _postVar (GP_Utils.cpp:943-1004).
// The reference materialises kX (N x M) and runs two dtrtrs with M right-hand sides; here
//   mean  = fused Gram-matvec (kX never stored),
//   var_t = kD - (1/sn2) * | L^-1 k*_t |^2       (one forward substitution, not two),
// streamed over test batches.  The cross-kernel is kept TEST-MAJOR (Wt = kX^T, batch x N) so
// that both steps of the blocked forward substitution are the same A*B^T MFMA kernel as the
// Cholesky:   Wt[:, j] := Wt[:, j] * inv(L_jj)^T ;  Wt[:, rest] -= Wt[:, j] * L[rest, j]^T.
#include <algorithm>
#include <cstdlib>

#include "gpak_internal.h"

#define PB 128

// part[split][t] = sum over the split's columns of V[t, c]^2
__global__ __launch_bounds__(256) void gpak_rowsumsq_part_f64(const double *__restrict__ V, long ldv, int rows,
                                                               int cols, int cols_per_split,
                                                               double *__restrict__ part, int part_ld) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= rows) return;
  const int c0 = blockIdx.y * cols_per_split;
  const int c1 = min(cols, c0 + cols_per_split);
  double s0 = 0.0, s1 = 0.0;
  int c = c0;
  for (; c + 1 < c1; c += 2) {
    double a = V[t + (size_t)c * ldv], b = V[t + (size_t)(c + 1) * ldv];
    s0 = fma(a, a, s0);
    s1 = fma(b, b, s1);
  }
  if (c < c1) { double a = V[t + (size_t)c * ldv]; s0 = fma(a, a, s0); }
  part[(size_t)blockIdx.y * part_ld + t] = s0 + s1;
}

void gpak_predict_release(gpak_ctx *ctx) {
  if (ctx->Upred.base) hipFree(ctx->Upred.base);
  if (ctx->Tq.base) hipFree(ctx->Tq.base);
  ctx->Upred = DevPoints(); ctx->Tq = DevPoints();
  if (ctx->dXte) hipFree(ctx->dXte);
  if (ctx->dWt) hipFree(ctx->dWt);
  if (ctx->dPv) hipFree(ctx->dPv);
  if (ctx->dPart) hipFree(ctx->dPart);
  if (ctx->dLf) hipFree(ctx->dLf);
  if (ctx->dInvf) hipFree(ctx->dInvf);
  ctx->dLf = ctx->dInvf = nullptr;
  ctx->lf_ok = false;
  ctx->dXte = ctx->dWt = ctx->dPv = ctx->dPart = nullptr;
  ctx->pred_cap = 0; ctx->wt_elems = 0;
}

static int ensure_predict_bufs(gpak_ctx *ctx, int cap, bool want_var) {
  if (ctx->pred_cap < cap) {
    if (ctx->dXte) hipFree(ctx->dXte);
    if (ctx->dPv) hipFree(ctx->dPv);
    if (ctx->dPart) hipFree(ctx->dPart);
    ctx->dXte = ctx->dPv = ctx->dPart = nullptr;
    if (hipMalloc(&ctx->dXte, sizeof(double) * 4 * (size_t)cap) != hipSuccess ||
        hipMalloc(&ctx->dPv, sizeof(double) * 2 * (size_t)cap) != hipSuccess ||
        hipMalloc(&ctx->dPart, sizeof(double) * 64 * (size_t)cap) != hipSuccess) {
      ctx->err = "device allocation failed for prediction buffers";
      return GPAK_ENOMEM;
    }
    int rc = gpak_alloc_points(ctx, ctx->Tq, cap);
    if (rc) return rc;
    ctx->pred_cap = cap;
  }
  int rc = gpak_alloc_points(ctx, ctx->Upred, ctx->Np);
  if (rc) return rc;
  // (pred_cap + skew) x Np: the batch is stored with a skewed leading dimension (gpak_predict_impl)
  size_t need = want_var ? ((size_t)ctx->pred_cap + 64 * (size_t)std::max(0, ctx->tune.pred_ld_skew)) * ctx->Np : 0;
  if (need > ctx->wt_elems) {
    if (ctx->dWt) hipFree(ctx->dWt);
    ctx->dWt = nullptr; ctx->wt_elems = 0;
    if (hipMalloc(&ctx->dWt, sizeof(double) * need) != hipSuccess) {
      ctx->err = "device allocation failed for the cross-kernel batch";
      return GPAK_ENOMEM;
    }
    ctx->wt_elems = need;
  }
  return GPAK_OK;
}

// blocked forward substitution on the test-major batch: Wt (mbp x Np, ld ldw) := Wt * L^-T.
// A ladder of block widths like the fp32 path below (GpakTuning::fs_levels, 128 / 512 / 2048 / 8192): a block of width
// lv[k] is solved by its children of width lv[k-1], each followed by ONE K = lv[k-1] update of the rest of the block.
// Round 3: the two-level form (128-column steps inside 512 columns, then a K = 512 update of everything to the
// right) left 75 % of the flops in K = 512 products (70.7 TFLOP/s alone); with the ladder they sit in K = 8192 / 2048
// products (75.7): 72.0 -> 73.1 TFLOP/s for the whole fp64 variance pass in a same-box A/B (the pass is within 5 % of
// the package power limit either way); same sums to 13 digits.
static void fs_block_f64(gpak_ctx *ctx, double *Wt, long ldw, int mt, int J0, int W, const std::vector<int> &lv, int k) {
  hipStream_t st = ctx->stream;
  const long ld = ctx->ld;
  if (k == 0) {   // W == 128: product with the explicit inverse of the diagonal block
    const double *inv = ctx->dInv + (size_t)(J0 / PB) * 2 * PB * PB;
    double *Wj = Wt + (size_t)J0 * ldw;
    gpak_launch_gemm_nt(st, mt, 1, PB, 1.0, Wj, ldw, inv, PB, 0.0, Wj, ldw, 0, 0, false, false);
    return;
  }
  const int cw = lv[k - 1];
  for (int j0 = J0; j0 < J0 + W; j0 += cw) {
    const int w = std::min(cw, J0 + W - j0);
    fs_block_f64(ctx, Wt, ldw, mt, j0, w, lv, k - 1);
    const int nrest = (J0 + W - j0 - w) / PB;
    if (nrest > 0)
      gpak_launch_gemm_nt(st, mt, nrest, w, -1.0, Wt + (size_t)j0 * ldw, ldw, ctx->dM + (j0 + w) + (size_t)j0 * ld, ld, 1.0,
                          Wt + (size_t)(j0 + w) * ldw, ldw, 0, 0, false, false);
  }
}
static void forward_subst_batch(gpak_ctx *ctx, double *Wt, long ldw, int mbp) {
  std::vector<int> lv;
  for (int v : ctx->tune.fs_levels) if (v > 0) lv.push_back(v);
  if (lv.empty()) lv = {PB, 512};
  lv.push_back(ctx->Np > lv.back() ? ctx->Np : lv.back() + 1);   // the whole matrix is the top block
  fs_block_f64(ctx, Wt, ldw, mbp / PB, 0, ctx->Np, lv, (int)lv.size() - 1);
}

// fp32 images of the factor for a GPAK_F32 context (rebuilt when the factor changes)
static int ensure_f32_factor(gpak_ctx *ctx) {
  if (ctx->lf_ok) return GPAK_OK;
  const int Np = ctx->Np, T = Np / PB;
  // k-columns 128 KiB apart (Np = 32768 floats) would all fall on the same L2 channel / HBM bank group: skew the
  // leading dimension by 256 B like the fp64 matrix (gpak_set_train)
  ctx->ldLf = Np + 64L * std::max(0, ctx->tune.pred_ld_skew);
  if (!ctx->dLf) {
    if (hipMalloc(&ctx->dLf, sizeof(float) * (size_t)ctx->ldLf * Np) != hipSuccess ||
        hipMalloc(&ctx->dInvf, sizeof(float) * (size_t)T * 2 * PB * PB) != hipSuccess) {
      ctx->err = "device allocation failed for the fp32 factor image";
      return GPAK_ENOMEM;
    }
  }
  gpak_launch_lower_to_f32(ctx->stream, ctx->dM, ctx->ld, Np, ctx->dLf, ctx->ldLf);
  gpak_launch_vec_to_f32(ctx->stream, ctx->dInv, (size_t)T * 2 * PB * PB, ctx->dInvf);
  ctx->lf_ok = true;
  return GPAK_OK;
}

// The forward substitution in fp32 (v_mfma_f32_16x16x4_f32), MULTI-level: a block of width lv[k] is solved child by
// child (width lv[k-1]); after each child ONE product of K = lv[k-1] updates the rest of the block.  Wider products
// move less (each level reads and writes the batch at most 3 times) and keep the MFMA pipe fed longer per launch, but
// with plain fp32 accumulation they cost accuracy: the terms of a product mostly share a sign, a running fp32 sum
// grows ~linearly with K and is rounded at that magnitude K/4 times -- measured at N = 32768, M = 65536 against the
// fp64 context: ladder 128/512 6.8e-6 of the largest variance, 128/512/2048 2.5e-5, 128/512/2048/8192 7e-5
// (profiles/r03_f32_accumulation.txt).  The products therefore run on gpak_gemm_nt_f32_rsw (fp32 chunks of K = 128
// summed in fp64, gemm_f32.hip): 7.4e-7 whatever the ladder, which is then chosen for speed alone.
// GPAK_FS_LEVELS_F32="128,512" restores the two-level scheme (A/B runs).
static void fs_block_f32(gpak_ctx *ctx, float *Wt, long ldw, int mt, int J0, int W, const std::vector<int> &lv, int k) {
  hipStream_t st = ctx->stream;
  if (k == 0) {   // W == 128: product with the explicit inverse of the diagonal block
    const float *inv = ctx->dInvf + (size_t)(J0 / PB) * 2 * PB * PB;
    float *Wj = Wt + (size_t)J0 * ldw;
    gpak_launch_gemm_nt_f32(st, mt, 1, PB, 1.f, Wj, ldw, inv, PB, 0.f, Wj, ldw);
    return;
  }
  const int cw = lv[k - 1];
  for (int j0 = J0; j0 < J0 + W; j0 += cw) {
    const int w = std::min(cw, J0 + W - j0);
    fs_block_f32(ctx, Wt, ldw, mt, j0, w, lv, k - 1);
    const int nrest = (J0 + W - j0 - w) / PB;
    if (nrest > 0)
      gpak_launch_gemm_nt_f32(st, mt, nrest, w, -1.f, Wt + (size_t)j0 * ldw, ldw, ctx->dLf + (j0 + w) + (size_t)j0 * ctx->ldLf, ctx->ldLf,
                              1.f, Wt + (size_t)(j0 + w) * ldw, ldw);
  }
}
static void forward_subst_batch_f32(gpak_ctx *ctx, float *Wt, long ldw, int mbp) {
  std::vector<int> lv;
  for (int v : ctx->tune.fs_levels) if (v > 0) lv.push_back(v);
  if (lv.empty()) lv = {PB, 512};
  lv.push_back(ctx->Np > lv.back() ? ctx->Np : lv.back() + 1);   // the whole matrix is the top block
  // the top block's width need not be a multiple of its children's: fs_block_f32 clips the last child
  fs_block_f32(ctx, Wt, ldw, mbp / PB, 0, ctx->Np, lv, (int)lv.size() - 1);
}

// pool_sum / pool_M: column sums and count of the WHOLE test set when Xte is a slice of it (multi-GPU prediction
// shards the test points; the pooled mean of Kernel.cpp:1391-1392 is over all of them), or nullptr
int gpak_predict_impl(gpak_ctx *ctx, const double *Xte, long M, double *mean, double *var, const double *pool_sum,
                      long pool_M) {
  GPAK_HIP(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const int N = ctx->N, Np = ctx->Np;
  // test points per batch, measured at N=32768: 4096 -> 58 (f64) / 91 (f32) TFLOP/s, 16384 -> 71.6 / 115.4,
  // 32768 -> 71.7 / 118.7, 65536 -> 71.8 / 119.7: the fp32 path takes the larger batch
  const bool f32 = ctx->precision == GPAK_F32;
  int batch = f32 ? 65536 : 16384;
  if (ctx->tune.pred_batch > 0) batch = std::max(2 * PB, ctx->tune.pred_batch / (2 * PB) * (2 * PB));
  // keep the cross-kernel batch under ~8 GiB
  while (batch > 2 * PB && (size_t)batch * Np * (f32 ? sizeof(float) : sizeof(double)) > ((size_t)8 << 30)) batch /= 2;
  batch = batch / (2 * PB) * (2 * PB);
  long Mp = (M + 2 * PB - 1) / (2 * PB) * (2 * PB);   // 256-row granularity: the fp32 GEMM's workgroup tile
  const int cap = (int)std::min<long>(Mp, batch);
  int rc = ensure_predict_bufs(ctx, cap, var != nullptr);
  if (rc) return rc;
  if (f32 && var && (rc = ensure_f32_factor(ctx))) return rc;
  hipEvent_t e0 = ctx->ev[5], e1 = ctx->ev[6];
  GPAK_HIP(hipEventRecord(e0, st));

  // pooled mean over train u (all) test points: Kernel.cpp:1391-1392 with X1 = Xinp, X2 = Xin
  double s2[4] = {0, 0, 0, 0};
  for (int k = 0; k < ctx->d; k++)
    for (long i = 0; i < M; i++) s2[k] += Xte[i + (size_t)k * M];
  KernParams kp = ctx->kp;
  kp.d = ctx->d;
  if (pool_sum) gpak_pooled_mean(ctx->xsum, N, pool_sum, pool_M, kp.mu);
  else gpak_pooled_mean(ctx->xsum, N, s2, M, kp.mu);
  gpak_launch_transform(st, ctx->dX, Np, N, kp, ctx->Upred);

  const double kD = ctx->kdiag;  // diag_Compute of the composition, Kernel.cpp:127-136, 780-783, 328-332
  kp.white = 0.0;                // Kern_White contributes nothing to a train x test block (Kernel.cpp:260-262)
  double *dMean = ctx->dPv, *dSq = ctx->dPv + cap;
  std::vector<double> hsq;
  for (long b0 = 0; b0 < M; b0 += cap) {
    const int mb = (int)std::min<long>(cap, M - b0);
    const int mbp = (mb + 2 * PB - 1) / (2 * PB) * (2 * PB);
    GPAK_HIP(hipMemsetAsync(ctx->dXte, 0, sizeof(double) * 4 * (size_t)cap, st));
    for (int k = 0; k < ctx->d; k++)
      GPAK_HIP(hipMemcpyAsync(ctx->dXte + (size_t)k * cap, Xte + (size_t)k * M + b0, sizeof(double) * mb,
                              hipMemcpyHostToDevice, st));
    gpak_launch_transform(st, ctx->dXte, cap, mb, kp, ctx->Tq);
    // _postMean: mu_t = Alpha . kX(:,t)
    int splits = gpak_kmatvec_splits(N, mb);
    gpak_launch_kmatvec(st, ctx->Upred, 0, N, ctx->dAlpha, ctx->Tq, kp, ctx->dPart, splits, dMean);
    GPAK_HIP(hipMemcpyAsync(mean + b0, dMean, sizeof(double) * mb, hipMemcpyDeviceToHost, st));
    if (var) {
      // test-major batch, k-columns `ldw` apart: a power-of-two stride (cap = 65536 floats = 256 KiB) puts every k-column
      // of a row tile on the same L2 channel -- skewed by 256 B like the fp64 matrix
      const long ldw = cap + (f32 ? 64L : 32L) * std::max(0, ctx->tune.pred_ld_skew);
      int vs = std::max(1, std::min(64, Np / 512));
      int cps = (Np + vs - 1) / vs;
      if (f32) {
        // fp32 arithmetic for the M-proportional work (cross-kernel, substitution, row sums);
        // the cross-kernel uses the direct distance form (cancellation form is meaningless in fp32)
        float *Wf = reinterpret_cast<float *>(ctx->dWt);
        gpak_launch_fill_f32(st, ctx->Tq, ctx->Upred, mbp, Np, kp, Wf, ldw);
        forward_subst_batch_f32(ctx, Wf, ldw, mbp);
        gpak_launch_rowsumsq_f32(st, Wf, ldw, mbp, Np, vs, ctx->dPart, cap);
      } else {
        gpak_launch_fill(st, ctx->Tq, ctx->Upred, mbp, Np, kp, 1.0, 0.0, 0.0, 0, ctx->dWt, ldw, nullptr);
        forward_subst_batch(ctx, ctx->dWt, ldw, mbp);
        hipLaunchKernelGGL(gpak_rowsumsq_part_f64, dim3((mbp + 255) / 256, vs), dim3(256), 0, st, ctx->dWt, ldw,
                           mbp, Np, cps, ctx->dPart, cap);
      }
      gpak_launch_sum_splits(st, ctx->dPart, cap, vs, mb, dSq);
      hsq.resize(mb);
      GPAK_HIP(hipMemcpyAsync(hsq.data(), dSq, sizeof(double) * mb, hipMemcpyDeviceToHost, st));
      GPAK_HIP(hipStreamSynchronize(st));
      // varSigma = kD - sum(LKs % kX) with LKs = sW . B^-1 . sW kX   (GP_Utils.cpp:985-999)
      for (int i = 0; i < mb; i++) var[b0 + i] = kD - hsq[i] / ctx->sn2;
    }
  }
  GPAK_HIP(hipEventRecord(e1, st));
  GPAK_HIP(hipEventSynchronize(e1));
  float ms = 0;
  GPAK_HIP(hipEventElapsedTime(&ms, e0, e1));
  ctx->times.predict_ms = ms;
  return GPAK_OK;
}

// solve_chol with k right-hand sides held on the host (GP_Utils.cpp:841-845)
int gpak_solve_chol_impl(gpak_ctx *ctx, double *X_host, int k) {
  GPAK_HIP(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const int N = ctx->N, Np = ctx->Np;
  double *w0 = ctx->dWork, *w1 = ctx->dWork + Np, *w2 = ctx->dWork + 2 * (size_t)Np;
  ctx->z_ok = false;
  for (int c = 0; c < k; c++) {
    GPAK_HIP(hipMemsetAsync(w0, 0, sizeof(double) * Np, st));
    GPAK_HIP(hipMemcpyAsync(w0, X_host + (size_t)c * N, sizeof(double) * N, hipMemcpyHostToDevice, st));
    gpak_launch_trsv_fwd(st, Np, ctx->dM, ctx->ld, ctx->dInv, w0, w1);
    if (ctx->t512_mode)
      gpak_launch_trsv_bwd3(st, Np, ctx->dM, ctx->ld, w1, w2, ctx->dWork + 3 * (size_t)Np,
                            ctx->t512_mode == 2 ? ctx->dT512 : ctx->dInv512,
                            (size_t)(ctx->t512_mode == 2 ? 2 : 1) * ctx->bwd_bw * ctx->bwd_bw,
                            (ctx->t512_mode == 2 ? 2 : 1) * ctx->bwd_bw, ctx->t512_mode == 2, ctx->bwd_bw);
    else
      gpak_launch_trsv_bwd2(st, Np, ctx->dM, ctx->ld, ctx->dInv, w1, w2, ctx->dWork + 3 * (size_t)Np,
                            ctx->inv512_ok ? ctx->dInv512 : nullptr, ctx->bwd_bw);
    GPAK_HIP(hipMemcpyAsync(X_host + (size_t)c * N, w2, sizeof(double) * N, hipMemcpyDeviceToHost, st));
  }
  GPAK_HIP(hipStreamSynchronize(st));
  return GPAK_OK;
}

