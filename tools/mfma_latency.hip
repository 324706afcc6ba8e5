// Latency / issue-rate probes for one wave on one CU (gfx950): dependent and independent v_mfma_f64_16x16x4_f64 chains,
// an LDS read -> MFMA -> LDS write round trip, and a dependent f64 FMA chain.  hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_latency tools/mfma_latency.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef double d4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ double rdlane(double v, int l) {
  int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
  int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}
// 64-bit DPP move, row_newbcast:l -- lane l of each 16-lane row to the whole row
template <int L>
__device__ __forceinline__ double rowbcast(double v) {
  double r;
  asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v), "n"(L));
  return r;
}
// the VALU 16x16 factor + inverse: lane c < 16 owns column c of the (symmetric) block, lanes 16..31 the identity
template <int MODE>
__global__ void valu_diag(long long *out, const double *Ain, double *Lout, unsigned others) {
  const int lane = threadIdx.x & 63, l15 = lane & 15, l4 = lane >> 4;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (wv > 0) {
    // company for wave 0: the waves named in `others` run dependent f64 MFMAs for about as long as the factorisation
    if (!((others >> wv) & 1)) return;
    d4 acc = {0, 0, 0, 0};
    double a = 1.0 + lane * 1e-9, b = 1.0 - lane * 1e-9;
    for (int i = 0; i < 200; i++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    if (acc[0] == 12345.0) Lout[600] = acc[1];
    return;
  }
  double x[16];
#pragma unroll
  for (int i = 0; i < 16; i++) x[i] = (l4 == 0) ? Ain[i + 16 * l15] : ((l4 == 1 && i == l15) ? 1.0 : 0.0);
  long long t0 = __builtin_readcyclecounter();
  asm volatile("" : "+v"(x[0]), "+v"(x[5]), "+v"(x[15]));
#pragma unroll
  for (int r = 0; r < 16; r++) {
    double d = rdlane(x[r], r);
    d = d > 0.0 ? d : 1.0;
    const double y = __builtin_amdgcn_rsq(d);
    double gg = d * y, h = 0.5 * y;
    double e = fma(-h, gg, 0.5);
    gg = fma(gg, e, gg); h = fma(h, e, h);
    e = fma(-h, gg, 0.5);
    const double g = fma(gg, e, gg); h = fma(h, e, h);
    const double ri = h + h;
    double xr = x[r] * ri;
    xr = (lane == r) ? g : xr;
    x[r] = xr;
    if (MODE == 0) {
#pragma unroll
      for (int i = r + 1; i < 16; i++) { const double ui = rdlane(xr, i); x[i] = fma(-ui, xr, x[i]); }
    } else {
      // u_i comes from the S row of lanes: broadcast inside row 0 by DPP, to the identity lanes by one more step
      // row 1 (the identity lanes) gets row 0's values: v_permlane16_swap (gfx950), one per 32-bit half
      const int xlo = __double2loint(xr), xhi = __double2hiint(xr);
      const double xs = __hiloint2double(__builtin_amdgcn_permlane16_swap(xhi, xhi, false, false)[0],
                                         __builtin_amdgcn_permlane16_swap(xlo, xlo, false, false)[0]);
#define BC(i_) if (i_ > r) { const double ui = rowbcast<i_>(xs); x[i_] = fma(-ui, xr, x[i_]); }
      BC(1) BC(2) BC(3) BC(4) BC(5) BC(6) BC(7) BC(8) BC(9) BC(10) BC(11) BC(12) BC(13) BC(14) BC(15)
#undef BC
    }
  }
  asm volatile("" :: "v"(x[0]), "v"(x[5]), "v"(x[15]), "v"(x[9]));
  long long t1 = __builtin_readcyclecounter();
  if (lane == 0) out[0] = t1 - t0;
#pragma unroll
  for (int i = 0; i < 16; i++) if (l4 < 2) Lout[i + 16 * l15 + 256 * l4] = x[i];
}
__global__ void probe(long long *out, double *sink, int waves_active) {
  __shared__ double L[8 * 256];
  const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);
  for (int i = t; i < 8 * 256; i += blockDim.x) L[i] = 1e-3 * (i & 15);
  __syncthreads();
  if (w >= waves_active) return;
  double a = 1.0 + lane * 1e-9, b = 1.0 - lane * 1e-9;
  d4 acc = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0}, acc3 = {0, 0, 0, 0}, acc4 = {0, 0, 0, 0};
  long long t0, t1;
  // the clock reads are ordered against the asm statements; the asm statements redefine / consume the operands, so the
  // timed arithmetic cannot move out of its window
#define BEGIN() do { t0 = __builtin_readcyclecounter(); asm volatile("" : "+v"(a), "+v"(b), "+v"(acc), "+v"(acc2), "+v"(acc3), "+v"(acc4), "+v"(f)); } while (0)
#define END(x_, slot_) do { double e_ = (x_) + 1.0; asm volatile("" :: "v"(e_)); t1 = __builtin_readcyclecounter(); if (lane == 0) out[w * 8 + slot_] = t1 - t0; } while (0)
  double f = a;
  volatile double *tile = L + (w & 7) * 256;
  // 1: 32 dependent MFMAs
  BEGIN();
#pragma unroll
  for (int i = 0; i < 32; i++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  END(acc[0], 0);
  // 2: 32 MFMAs in 4 independent chains
  BEGIN();
#pragma unroll
  for (int i = 0; i < 8; i++) {
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc2, 0, 0, 0);
    acc3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc3, 0, 0, 0);
    acc4 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc4, 0, 0, 0);
  }
  END(acc[0] + acc2[0] + acc3[0] + acc4[0], 1);
  // 3: 16 x (LDS read 2 operands -> 1 MFMA -> LDS write 4 -> read them back) dependent round trips
  BEGIN();
#pragma unroll
  for (int i = 0; i < 16; i++) {
    double x = tile[lane], y = tile[64 + lane];
    d4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, c, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; r++) tile[64 * r + lane] = c[r] * 1e-3;
  }
  END(tile[lane], 2);
  // 4: 64 dependent f64 FMAs
  BEGIN();
#pragma unroll
  for (int i = 0; i < 64; i++) f = __builtin_fma(f, b, a);
  END(f, 3);
  // 5: 16 dependent LDS round trips (read -> write -> read ...)
  double g = 0;
  BEGIN();
#pragma unroll
  for (int i = 0; i < 16; i++) { g += tile[lane]; tile[lane] = g; }
  END(tile[lane], 4);
  // 6: 16 readlane -> fma chains
  BEGIN();
#pragma unroll
  for (int i = 0; i < 16; i++) { double s = __shfl(f, i, 64); f = __builtin_fma(f, s, a); }
  END(f, 5);
  // 7: 16 x (MFMA -> read accumulator with a VALU op -> feed the next MFMA's A operand): the pivot-loop dependency
  BEGIN();
#pragma unroll
  for (int i = 0; i < 16; i++) { acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0); a = __builtin_fma(acc[0], 1e-9, b); }
  END(a, 6);
  // 8: 16 x (1 MFMA + 8 dependent f64 FMAs that do not touch it): overlap or not?
  BEGIN();
#pragma unroll
  for (int i = 0; i < 16; i++) {
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
#pragma unroll
    for (int j = 0; j < 8; j++) f = __builtin_fma(f, b, a);
  }
  END(acc[0] + f, 7);
  sink[t] = acc[0] + acc2[1] + acc3[2] + acc4[3] + f + g;
}
int main() {
  long long *out; double *sink;
  hipMalloc(&out, 8 * 8 * 8); hipMalloc(&sink, 512 * 8);
  for (int wa : {1, 2, 4, 8}) {
    hipMemset(out, 0, 8 * 8 * 8);
    for (int rep = 0; rep < 3; rep++) hipLaunchKernelGGL(probe, dim3(1), dim3(512), 0, 0, out, sink, wa);
    hipDeviceSynchronize();
    long long h[64]; hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
    for (int w = 0; w < wa; w += (wa > 2 ? wa - 1 : 1))
      printf("waves %d wave %d: 32 dependent MFMA %lld (%.0f each) | 32 MFMA in 4 chains %lld (%.0f each) | 16 LDS->MFMA->LDS %lld (%.0f each) | 64 dep FMA %lld (%.1f each) | 16 LDS rd->wr %lld (%.0f each) | 16 readlane->fma %lld (%.0f each) | 16 MFMA->VALU->MFMA %lld (%.0f each) | 16 x (MFMA + 8 indep. dep-FMA) %lld (%.0f each)\n",
             wa, w, h[w*8], h[w*8]/32.0, h[w*8+1], h[w*8+1]/32.0, h[w*8+2], h[w*8+2]/16.0, h[w*8+3], h[w*8+3]/64.0, h[w*8+4], h[w*8+4]/16.0, h[w*8+5], h[w*8+5]/16.0, h[w*8+6], h[w*8+6]/16.0, h[w*8+7], h[w*8+7]/16.0);
  }
  {
    double hA[256], hL[512], *dA, *dL;
    for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) hA[i + 16 * j] = (i == j ? 20.0 : 0.0) + 1.0 / (1 + i + j);
    hipMalloc(&dA, sizeof hA); hipMalloc(&dL, sizeof hL + 1024); hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice);
    for (int mode = 0; mode < 6; mode++) {
      const unsigned others = mode < 2 ? 0u : mode == 2 ? 0xfeu : mode == 3 ? 0x0eu : mode == 4 ? 0x10u : 0xeeu;
      for (int rep = 0; rep < 3; rep++) {
        if (mode == 1) hipLaunchKernelGGL(valu_diag<1>, dim3(1), dim3(64), 0, 0, out, dA, dL, others);
        else hipLaunchKernelGGL(valu_diag<0>, dim3(1), dim3(mode < 2 ? 64 : 512), 0, 0, out, dA, dL, others);
      }
      hipDeviceSynchronize();
      long long c; hipMemcpy(&c, out, 8, hipMemcpyDeviceToHost); hipMemcpy(hL, dL, sizeof hL, hipMemcpyDeviceToHost);
      // residual of L L^T against A (lane c holds row c of L in x[r], r <= c) and of W L against I
      double res = 0, resi = 0;
      for (int i = 0; i < 16; i++) for (int j = 0; j <= i; j++) {
        double s = 0; for (int k = 0; k <= j; k++) s += hL[k + 16 * i] * hL[k + 16 * j];
        res = fmax(res, fabs(s - hA[i + 16 * j]));
        double w = 0; for (int k = j; k <= i; k++) w += hL[k + 16 * i] * hL[256 + k + 16 * j];   // L[i][k] Linv[k][j]
        resi = fmax(resi, fabs(w - (i == j ? 1.0 : 0.0)));
      }
      printf("VALU 16x16 factor+inverse (%s): %lld cycles (%.0f per pivot), max |LL^T - A| %.2e, max |L Linv - I| %.2e\n",
             mode == 1 ? "DPP row_newbcast + permlane16_swap" : mode == 0 ? "v_readlane" : mode == 2 ? "v_readlane, waves 1-7 on MFMAs" : mode == 3 ? "v_readlane, waves 1-3 (other SIMDs) on MFMAs" : mode == 4 ? "v_readlane, wave 4 (same SIMD) on MFMAs" : "v_readlane, waves 1-3,5-7 on MFMAs", c, c / 16.0, res, resi);
    }
  }
  return 0;
}
