// gemm_f32.hip -- fp32 sibling of gemm.hip for the prediction path (BASELINE.json configs[4]:
// fp32, M = 1e6 test points): C = beta*C + alpha * A * B^T on v_mfma_f32_16x16x4_f32
// (157 TFLOP/s dense on MI355X = twice the fp64 matrix rate).
//
// Same geometry as the fp64 kernel: 128x128 tile, 4 waves (2x2), each 64x64 = 4x4 MFMA tiles,
// LDS-DMA staging of [k][row] images, two buffers, one barrier per stage.  Differences:
//   * a stage is 32 k deep (16 KiB per operand, like 16 k of fp64); one 16-B LDS-DMA instruction
//     covers TWO k-rows (2 x 128 floats), so the image is unpadded (row = 512 B) -- the two
//     k-planes a 32-lane group reads then share banks (2-way), which the fp32 MFMA rate hides;
//   * the f32 16x16x4 result layout is the standard one (col = lane&15, row = 4*(lane>>4)+reg);
//     with the operand roles swapped as in gemm.hip a lane holds row (lane&15), columns
//     4*(lane>>4) + reg of its 16x16 tile.
#include <cstdlib>
#include <cstring>

#include "gpak_internal.h"

typedef float f4 __attribute__((ext_vector_type(4)));

#define TM 128
#define TN 128
#define KB32 32

// ---------------------------------------------------------------------------------------
// Register-streaming variant (the default, see gemm.hip): no LDS, no barriers.  One 16-B load per
// operand and k-step feeds all FOUR of a wave's fragments on that side: lane (l15, l4) fetches rows
// 4*l15 .. 4*l15+3 of k-column 4s + l4 and uses element i as its entry of MFMA tile i, i.e. tile i covers
// the rows 4j + i, j = 0..15, of the wave's 64 (16 lanes = 256 B contiguous); the epilogue undoes the
// permutation with 16-B accesses: of tile (mi, ni), register r, a lane holds C row 4*l15 + mi, C column
// 16*l4 + 4*r + ni.
// ---------------------------------------------------------------------------------------
template <int RS_D>
__global__ __launch_bounds__(256, 3) void gpak_gemm_nt_f32_rs(int K, float alpha, const float *A, long lda,
                                                               const float *B, long ldb, float beta, float *C,
                                                               long ldc, int mt, int nt) {
  int ti, tj;
  {
    const int b = blockIdx.x, q = b >> 3;
    const int slot = q & 63;
    const int ssel = (q >> 6) * 8 + (b & 7);
    const int SR = (mt + 7) >> 3, SC = (nt + 7) >> 3;
    const int sj = ssel / SR, si = ssel - sj * SR;
    if (sj >= SC) return;
    ti = si * 8 + (slot & 7);
    tj = sj * 8 + (slot >> 3);
    if (ti >= mt || tj >= nt) return;
  }
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wr = w & 1, wc = w >> 1;
  const int l15 = lane & 15, l4 = lane >> 4;
  const f4 *Ap = reinterpret_cast<const f4 *>(A + (size_t)ti * TM + wr * 64 + 4 * l15 + (size_t)l4 * lda);
  const f4 *Bp = reinterpret_cast<const f4 *>(B + (size_t)tj * TN + wc * 64 + 4 * l15 + (size_t)l4 * ldb);
  const size_t sa = (size_t)lda, sb = (size_t)ldb;  // 4 k-columns in 16-B units
  f4 acc[4][4];
#pragma unroll
  for (int mi = 0; mi < 4; mi++)
#pragma unroll
    for (int ni = 0; ni < 4; ni++) acc[mi][ni] = (f4){0.f, 0.f, 0.f, 0.f};
  f4 ra[RS_D], rbv[RS_D];
#define RS_LOAD(slot_) \
  ra[slot_] = *Ap;     \
  rbv[slot_] = *Bp;    \
  Ap += sa;            \
  Bp += sb;
#define RS_MFMA(slot_)                                                                                      \
  _Pragma("unroll") for (int mi = 0; mi < 4; mi++) _Pragma("unroll") for (int ni = 0; ni < 4; ni++)         \
      acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(rbv[slot_][ni], ra[slot_][mi], acc[mi][ni], 0, 0, 0);
  const int n = K / 4;  // >= 32 k-steps (K >= 128)
#pragma unroll
  for (int s = 0; s < RS_D; s++) { RS_LOAD(s) }
  int g = 0;
  for (; g + 2 * RS_D <= n; g += RS_D) {
#pragma unroll
    for (int s = 0; s < RS_D; s++) {
      RS_MFMA(s)
      RS_LOAD(s)
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const int r = n - (g + RS_D);
#pragma unroll
  for (int s = 0; s < RS_D; s++) {
    RS_MFMA(s)
    if (s < r) { RS_LOAD(s) }
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int s = 0; s < RS_D; s++)
    if (s < r) { RS_MFMA(s) }
#undef RS_LOAD
#undef RS_MFMA
  if (A == C) __syncthreads();  // in-place panel product: every wave's operand reads before anybody's stores
  float *Cg = C + (size_t)ti * TM + wr * 64 + 4 * l15 + ((size_t)tj * TN + wc * 64 + 16 * l4) * ldc;
#pragma unroll
  for (int ni = 0; ni < 4; ni++)
#pragma unroll
    for (int r4 = 0; r4 < 4; r4++) {
      f4 *p = reinterpret_cast<f4 *>(Cg + (size_t)(4 * r4 + ni) * ldc);
      f4 v = {alpha * acc[0][ni][r4], alpha * acc[1][ni][r4], alpha * acc[2][ni][r4], alpha * acc[3][ni][r4]};
      if (beta != 0.f) {
        const f4 c = *p;
        v.x = fmaf(beta, c.x, v.x);
        v.y = fmaf(beta, c.y, v.y);
        v.z = fmaf(beta, c.z, v.z);
        v.w = fmaf(beta, c.w, v.w);
      }
      *p = v;
    }
}

// ---------------------------------------------------------------------------------------
// The same with a 128 x 64 wave tile (256 x 128 per workgroup): 3 operand loads per 32 MFMAs instead of
// 4 -- the fp32 MFMA runs at twice the fp64 rate, so per clock this path asks the vector L1 / L2 for twice
// the operand bytes of the fp64 kernel, and that, not instruction issue, is what holds it below its peak.
// Needs an even number of 128-row tiles (the prediction batches are padded to 256 rows).
// ---------------------------------------------------------------------------------------
template <int RS_D>
__global__ __launch_bounds__(256, 2) void gpak_gemm_nt_f32_rs2(int K, float alpha, const float *A, long lda,
                                                                const float *B, long ldb, float beta, float *C,
                                                                long ldc, int mt2, int nt) {
  int ti, tj;
  {
    const int b = blockIdx.x, q = b >> 3;
    const int slot = q & 63;
    const int ssel = (q >> 6) * 8 + (b & 7);
    const int SR = (mt2 + 7) >> 3, SC = (nt + 7) >> 3;
    const int sj = ssel / SR, si = ssel - sj * SR;
    if (sj >= SC) return;
    ti = si * 8 + (slot & 7);
    tj = sj * 8 + (slot >> 3);
    if (ti >= mt2 || tj >= nt) return;
  }
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wr = w & 1, wc = w >> 1;
  const int l15 = lane & 15, l4 = lane >> 4;
  const f4 *Ap = reinterpret_cast<const f4 *>(A + (size_t)ti * 256 + wr * 128 + 4 * l15 + (size_t)l4 * lda);
  const f4 *Bp = reinterpret_cast<const f4 *>(B + (size_t)tj * TN + wc * 64 + 4 * l15 + (size_t)l4 * ldb);
  const size_t sa = (size_t)lda, sb = (size_t)ldb;
  f4 acc[8][4];
#pragma unroll
  for (int mi = 0; mi < 8; mi++)
#pragma unroll
    for (int ni = 0; ni < 4; ni++) acc[mi][ni] = (f4){0.f, 0.f, 0.f, 0.f};
  f4 ra[RS_D][2], rbv[RS_D];
#define RS_LOAD(slot_)      \
  ra[slot_][0] = Ap[0];     \
  ra[slot_][1] = Ap[16];    \
  rbv[slot_] = *Bp;         \
  Ap += sa;                 \
  Bp += sb;
#define RS_MFMA(slot_)                                                                                      \
  _Pragma("unroll") for (int mi = 0; mi < 8; mi++) _Pragma("unroll") for (int ni = 0; ni < 4; ni++)         \
      acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(rbv[slot_][ni], ra[slot_][mi >> 2][mi & 3], acc[mi][ni], 0, 0, 0);
  const int n = K / 4;
#pragma unroll
  for (int s = 0; s < RS_D; s++) { RS_LOAD(s) }
  int g = 0;
  for (; g + 2 * RS_D <= n; g += RS_D) {
#pragma unroll
    for (int s = 0; s < RS_D; s++) {
      RS_MFMA(s)
      RS_LOAD(s)
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const int r = n - (g + RS_D);
#pragma unroll
  for (int s = 0; s < RS_D; s++) {
    RS_MFMA(s)
    if (s < r) { RS_LOAD(s) }
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int s = 0; s < RS_D; s++)
    if (s < r) { RS_MFMA(s) }
#undef RS_LOAD
#undef RS_MFMA
  if (A == C) __syncthreads();
  // of tile (mi, ni), register r4: C row 64 (mi >> 2) + 4 l15 + (mi & 3), C column 16 l4 + 4 r4 + ni
  float *Cg = C + (size_t)ti * 256 + wr * 128 + 4 * l15 + ((size_t)tj * TN + wc * 64 + 16 * l4) * ldc;
#pragma unroll
  for (int h = 0; h < 2; h++)
#pragma unroll
    for (int ni = 0; ni < 4; ni++)
#pragma unroll
      for (int r4 = 0; r4 < 4; r4++) {
        f4 *p = reinterpret_cast<f4 *>(Cg + 64 * h + (size_t)(4 * r4 + ni) * ldc);
        f4 v = {alpha * acc[4 * h][ni][r4], alpha * acc[4 * h + 1][ni][r4], alpha * acc[4 * h + 2][ni][r4],
                alpha * acc[4 * h + 3][ni][r4]};
        if (beta != 0.f) {
          const f4 c = *p;
          v.x = fmaf(beta, c.x, v.x);
          v.y = fmaf(beta, c.y, v.y);
          v.z = fmaf(beta, c.z, v.z);
          v.w = fmaf(beta, c.w, v.w);
        }
        *p = v;
      }
}

// ---------------------------------------------------------------------------------------
// WIDE-accumulation variant (64 x 64 wave tile as gpak_gemm_nt_f32_rs).  The fp32 MFMA accumulates a chunk of 32
// k-steps (K = 128) from zero; the chunk sums are then added into fp64 totals on the vector ALU, and the epilogue
// forms alpha * total + beta * C in fp64 and rounds ONCE.  Why: in the forward substitution the terms of a product
// mostly share a sign, so a running fp32 sum grows ~linearly with K and every one of its K/4 roundings happens at that
// growing magnitude -- the variance error of the fp32 prediction was measured to grow like K^1.5 with the panel width
// (two-level 128/512 ladder 6.8e-6 of the largest variance at M = 65536, 128/512/2048: 2.5e-5, .../8192: 7e-5).  With
// chunks of 128 the fp32 part of a K = 512 product carries (128/512)^1.5 = 1/8 of that.  Cost per chunk and wave: 64
// conversions + 64 fp64 adds beside 512 MFMAs.
// ---------------------------------------------------------------------------------------
template <int RS_D, int OCC>
__global__ __launch_bounds__(256, OCC) void gpak_gemm_nt_f32_rsw(int K, float alpha, const float *A, long lda,
                                                                const float *B, long ldb, float beta, float *C,
                                                                long ldc, int mt, int nt) {
  int ti, tj;
  {
    const int b = blockIdx.x, q = b >> 3;
    const int slot = q & 63;
    const int ssel = (q >> 6) * 8 + (b & 7);
    const int SR = (mt + 7) >> 3, SC = (nt + 7) >> 3;
    const int sj = ssel / SR, si = ssel - sj * SR;
    if (sj >= SC) return;
    ti = si * 8 + (slot & 7);
    tj = sj * 8 + (slot >> 3);
    if (ti >= mt || tj >= nt) return;
  }
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wr = w & 1, wc = w >> 1;
  const int l15 = lane & 15, l4 = lane >> 4;
  // operand addresses = wave-uniform bases in scalar registers (advanced on the scalar unit) + a fixed 32-bit lane offset:
  // no vector ALU instruction in the loop but the MFMAs -- each one holds the matrix pipe for 8-16 cycles
  // (tools/mfma_f32_loop.hip: 146 -> 155.6 TFLOP/s for this loop without the two v_lshl_add_u64 per k-step)
  const char *Ac = reinterpret_cast<const char *>(A + (size_t)ti * TM + wr * 64);
  const char *Bc = reinterpret_cast<const char *>(B + (size_t)tj * TN + wc * 64);
  unsigned aoff = (unsigned)((4 * l15 + (size_t)l4 * lda) * sizeof(float));
  unsigned boff = (unsigned)((4 * l15 + (size_t)l4 * ldb) * sizeof(float));
  const size_t sa = 4 * (size_t)lda * sizeof(float), sb = 4 * (size_t)ldb * sizeof(float);  // 4 k-columns, in bytes
#define RSW_LD(p_, o_) (*reinterpret_cast<const f4 *>((p_) + (o_)))
  f4 acc[4][4];
  double tot[4][4][4];
#pragma unroll
  for (int mi = 0; mi < 4; mi++)
#pragma unroll
    for (int ni = 0; ni < 4; ni++)
#pragma unroll
      for (int r = 0; r < 4; r++) tot[mi][ni][r] = 0.0;
  f4 ra[RS_D], rbv[RS_D];
  const int n = K / 4;   // k-steps: a multiple of 32 (K is a multiple of 128)
#pragma unroll
  for (int s = 0; s < RS_D; s++) {
    ra[s] = RSW_LD(Ac, aoff); rbv[s] = RSW_LD(Bc, boff); Ac += sa; Bc += sb;
  }
  for (int c0 = 0; c0 < n; c0 += 32) {
    asm volatile("" : "+v"(aoff), "+v"(boff));   // keeps base + offset from becoming per-lane 64-bit induction variables
#pragma unroll
    for (int s = 0; s < 32; s++) {
      const int slot = s % RS_D;
      if (s == 0) {
#pragma unroll
        for (int mi = 0; mi < 4; mi++)
#pragma unroll
          for (int ni = 0; ni < 4; ni++)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(rbv[slot][ni], ra[slot][mi], (f4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
      } else {
#pragma unroll
        for (int mi = 0; mi < 4; mi++)
#pragma unroll
          for (int ni = 0; ni < 4; ni++)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(rbv[slot][ni], ra[slot][mi], acc[mi][ni], 0, 0, 0);
      }
      // unconditional (a load under a branch falls back to a vector add of base and offset): the bases stop at the last
      // k-step, so that the RS_D loads past the end re-read it into slots nobody consumes
      ra[slot] = RSW_LD(Ac, aoff); rbv[slot] = RSW_LD(Bc, boff);
      { const bool more = c0 + s + RS_D + 1 < n; Ac += more ? sa : 0; Bc += more ? sb : 0; }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int mi = 0; mi < 4; mi++)
#pragma unroll
      for (int ni = 0; ni < 4; ni++)
#pragma unroll
        for (int r = 0; r < 4; r++) tot[mi][ni][r] += (double)acc[mi][ni][r];
  }
  if (A == C) __syncthreads();  // in-place panel product: every wave's operand reads before anybody's stores
  float *Cg = C + (size_t)ti * TM + wr * 64 + 4 * l15 + ((size_t)tj * TN + wc * 64 + 16 * l4) * ldc;
  const double da = (double)alpha, db = (double)beta;
#pragma unroll
  for (int ni = 0; ni < 4; ni++)
#pragma unroll
    for (int r4 = 0; r4 < 4; r4++) {
      f4 *p = reinterpret_cast<f4 *>(Cg + (size_t)(4 * r4 + ni) * ldc);
      double v0 = da * tot[0][ni][r4], v1 = da * tot[1][ni][r4], v2 = da * tot[2][ni][r4], v3 = da * tot[3][ni][r4];
      if (beta != 0.f) {
        const f4 c = *p;
        v0 = fma(db, (double)c.x, v0);
        v1 = fma(db, (double)c.y, v1);
        v2 = fma(db, (double)c.z, v2);
        v3 = fma(db, (double)c.w, v3);
      }
      *p = (f4){(float)v0, (float)v1, (float)v2, (float)v3};
    }
}

void gpak_launch_gemm_nt_f32(hipStream_t st, int mt, int nt, int K, float alpha, const float *A, long lda,
                             const float *B, long ldb, float beta, float *C, long ldc) {
  if (mt <= 0 || nt <= 0) return;
  const int SR = (mt + 7) / 8, SC = (nt + 7) / 8;
  const long nsuper = (long)SR * SC;
  dim3 grid((unsigned)((nsuper + 7) / 8 * 8 * 64)), block(256);
  // Products longer than one chunk take the wide-accumulation kernel (fp32 MFMA chunks of K = 128 summed in fp64):
  // measured at N = 32768, M = 65536 against the fp64 context, variance error 7.4e-7 of the largest variance instead
  // of 6.8e-6, for 118 instead of 120.5 TFLOP/s (profiles/r03_f32_accumulation.txt).  A K = 128 product IS one chunk:
  // the 128 x 64-tile kernel below computes the same sum faster.  GpakTuning::f32_wide = false (GPAK_F32_ACC=plain): the
  // round-2 kernels everywhere.
  const GpakTuning &tn = gpak_tuning();
  const bool wide = tn.f32_wide && K > 128 && K % 128 == 0;
  if (wide) {
    const int v = tn.f32_rsd;
    if (v == 8)        // one workgroup per CU slot pair: all 512 registers of a SIMD lane for one wave
      hipLaunchKernelGGL((gpak_gemm_nt_f32_rsw<8, 1>), grid, block, 0, st, K, alpha, A, lda, B, ldb, beta, C, ldc, mt, nt);
    else if (v == 16)  // the deepest ring one wave per SIMD has registers for (profiles/r03_f32_prediction.txt: 8 / 16 -> 128 / 131.5)
      hipLaunchKernelGGL((gpak_gemm_nt_f32_rsw<16, 1>), grid, block, 0, st, K, alpha, A, lda, B, ldb, beta, C, ldc, mt, nt);
    else if (v == 2)
      hipLaunchKernelGGL((gpak_gemm_nt_f32_rsw<2, 2>), grid, block, 0, st, K, alpha, A, lda, B, ldb, beta, C, ldc, mt, nt);
    else
      hipLaunchKernelGGL((gpak_gemm_nt_f32_rsw<4, 2>), grid, block, 0, st, K, alpha, A, lda, B, ldb, beta, C, ldc, mt, nt);
  } else if (!(mt & 1) && tn.f32_tile != 64) {
    const int mt2 = mt / 2, SR2 = (mt2 + 7) / 8;
    dim3 grid2((unsigned)(((long)SR2 * SC + 7) / 8 * 8 * 64));
    hipLaunchKernelGGL(gpak_gemm_nt_f32_rs2<4>, grid2, block, 0, st, K, alpha, A, lda, B, ldb, beta, C, ldc, mt2, nt);
  } else
    hipLaunchKernelGGL(gpak_gemm_nt_f32_rs<8>, grid, block, 0, st, K, alpha, A, lda, B, ldb, beta, C, ldc, mt, nt);
}

// ---- fp32 images of the fp64 factor ---------------------------------------------------------
__global__ void gpak_lower_to_f32(const double *__restrict__ L, long ld, int Np, float *__restrict__ out, long ldo) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t tot = (size_t)Np * Np;
  if (i >= tot) return;
  const size_t c = i / Np, r = i - c * Np;
  out[r + c * ldo] = r >= c ? (float)L[r + c * ld] : 0.f;
}
__global__ void gpak_vec_to_f32(const double *__restrict__ in, size_t n, float *__restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (float)in[i];
}
void gpak_launch_lower_to_f32(hipStream_t st, const double *L, long ld, int Np, float *out, long ldo) {
  const size_t tot = (size_t)Np * Np;
  hipLaunchKernelGGL(gpak_lower_to_f32, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, L, ld, Np, out, ldo);
}
void gpak_launch_vec_to_f32(hipStream_t st, const double *in, size_t n, float *out) {
  hipLaunchKernelGGL(gpak_vec_to_f32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, in, n, out);
}

// ---- fp32 cross-kernel fill (test-major batch) and row sums of squares ------------------------
// distance in fp64 from the fp64 coordinates (three subtractions), profile sqrt/exp in fp32
// components 0..2 and the 4th input column (array 4 of the point layout; zero for 3-D inputs)
#define PARR32(base, cap, t, c) GPAK_PARR(base, cap, t, (c) < 3 ? (c) : 4)
__global__ __launch_bounds__(256) void gpak_fill_f32(const double *__restrict__ P, int capP, int nP,
                                                      const double *__restrict__ Q, int capQ, int nQ, KernParams kp,
                                                      float *__restrict__ C, long ld) {
  const int row0 = blockIdx.x * 128, col0 = blockIdx.y * 64;
  __shared__ double q[GPAK_MAX_TERMS][4][64];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int nterms = kp.nterms;
  if (t < 64) {
    const int j = col0 + t;
    const bool ok = j < nQ;
    for (int m = 0; m < nterms; m++)
#pragma unroll
      for (int c = 0; c < 4; c++) q[m][c][t] = ok ? PARR32(Q, capQ, m, c)[j] : 0.0;
  }
  const int r = row0 + 2 * lane;
  double2 a[GPAK_MAX_TERMS][4];
  for (int m = 0; m < nterms; m++)
#pragma unroll
    for (int c = 0; c < 4; c++) a[m][c] = *reinterpret_cast<const double2 *>(PARR32(P, capP, m, c) + r);
  __syncthreads();
#pragma unroll 2
  for (int c = 0; c < 16; c++) {
    const int jl = w + 4 * c, j = col0 + jl;
    float k0 = (float)kp.bias, k1 = (float)kp.bias;
    for (int m = 0; m < nterms; m++) {
      const double b0 = q[m][0][jl], b1 = q[m][1][jl], b2 = q[m][2][jl], b3 = q[m][3][jl];
      double dx = a[m][0].x - b0, dy = a[m][1].x - b1, dz = a[m][2].x - b2, dr = a[m][3].x - b3;
      const float d0 = (float)(dx * dx + dy * dy + dz * dz + dr * dr);
      dx = a[m][0].y - b0; dy = a[m][1].y - b1; dz = a[m][2].y - b2; dr = a[m][3].y - b3;
      const float d1 = (float)(dx * dx + dy * dy + dz * dz + dr * dr);
      const float v2 = (float)kp.term[m].var2;
      if (kp.term[m].profile == GPAK_PROFILE_RBF) {
        const float hw = -0.5f * (float)kp.term[m].iw;
        k0 += v2 * __expf(hw * d0); k1 += v2 * __expf(hw * d1);
      } else {
        k0 += v2 * __expf(-sqrtf(d0)); k1 += v2 * __expf(-sqrtf(d1));
      }
    }
    const bool cj = j < nQ;
    if (!(cj && r < nP)) k0 = 0.f;
    if (!(cj && r + 1 < nP)) k1 = 0.f;
    *reinterpret_cast<float2 *>(C + r + (size_t)j * ld) = make_float2(k0, k1);
  }
}
void gpak_launch_fill_f32(hipStream_t st, const DevPoints &P, const DevPoints &Q, int rows_p, int cols_p,
                          const KernParams &kp, float *C, long ld) {
  dim3 grid(rows_p / 128, cols_p / 64);
  hipLaunchKernelGGL(gpak_fill_f32, grid, dim3(256), 0, st, P.base, P.cap, P.n, Q.base, Q.cap, Q.n, kp, C, ld);
}

__global__ __launch_bounds__(256) void gpak_rowsumsq_part_f32(const float *__restrict__ V, long ldv, int rows, int cols,
                                                               int cols_per_split, double *__restrict__ part,
                                                               int part_ld) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= rows) return;
  const int c0 = blockIdx.y * cols_per_split;
  const int c1 = min(cols, c0 + cols_per_split);
  // the squares and their sum in fp64 (the kernel streams fp32 at the HBM rate; 2e9 fp64 FMAs per batch are free):
  // what is subtracted from kD is then exactly the sum of the stored fp32 values' squares
  double s = 0.0, s2 = 0.0;
  int c = c0;
  for (; c + 1 < c1; c += 2) {
    const double a = (double)V[t + (size_t)c * ldv], b = (double)V[t + (size_t)(c + 1) * ldv];
    s = fma(a, a, s);
    s2 = fma(b, b, s2);
  }
  if (c < c1) { const double a = (double)V[t + (size_t)c * ldv]; s = fma(a, a, s); }
  part[(size_t)blockIdx.y * part_ld + t] = s + s2;
}
void gpak_launch_rowsumsq_f32(hipStream_t st, const float *V, long ldv, int rows, int cols, int splits,
                              double *part, int part_ld) {
  int cps = (cols + splits - 1) / splits;
  hipLaunchKernelGGL(gpak_rowsumsq_part_f32, dim3((rows + 255) / 256, splits), dim3(256), 0, st, V, ldv, rows, cols,
                     cps, part, part_ld);
}
