// What does the inner loop of the fp32 register-streamed product cost beyond its MFMAs?  The loop of gpak_gemm_nt_f32_rs
// (64 x 64 wave tile, 16 v_mfma_f32_16x16x4_f32 + two 16-byte operand loads + two 64-bit pointer adds per k-step, RS_D
// slots of look-ahead) on a buffer that stays in L1, whole chip, with pieces switched off:
//   LOADS 0: operands never reloaded (registers only, distinct registers per MFMA)   LOADS 1: as the library
//   SHAPE 16: 16x16x4 (16 per k-step)   SHAPE 32: 32x32x2 (4 per two k... same flops per k-step of 4: 8 MFMAs)
// hipcc --offload-arch=gfx950 -O3 -o tools/bin/mfma_f32_loop tools/mfma_f32_loop.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16 __attribute__((ext_vector_type(16)));

// FLUSH: 0 none; 1 chunk sums added into fp64 totals (v_cvt_f64_f32 + v_add_f64, as gpak_gemm_nt_f32_rsw); 2 into
// two-float totals (TwoSum: 6 fp32 adds per element); 3 into fp32 sums of four chunks that are added into fp32 totals
// (three-level fp32 summation, no fp64 instruction); CH: k-steps per chunk (32 = K 128)
template <int LOADS, int RS_D, int OCC, int FLUSH = 0, int CH = 32>
__global__ __launch_bounds__(256, OCC) void loop16(const float *A, const float *B, long lda, int n, float *sink) {
  const int lane = threadIdx.x & 63, l15 = lane & 15, l4 = lane >> 4, w = threadIdx.x >> 6;
  const f4 *Ap = reinterpret_cast<const f4 *>(A + (w & 1) * 64 + 4 * l15 + (size_t)l4 * lda);
  const f4 *Bp = reinterpret_cast<const f4 *>(B + (w >> 1) * 64 + 4 * l15 + (size_t)l4 * lda);
  const size_t sa = (size_t)lda;
  f4 acc[4][4];
  for (int mi = 0; mi < 4; mi++) for (int ni = 0; ni < 4; ni++) acc[mi][ni] = (f4){0, 0, 0, 0};
  f4 ra[RS_D], rb[RS_D];
  double tot[FLUSH == 1 ? 4 : 1][4][4];
  float thi[FLUSH >= 2 ? 4 : 1][4][4], tlo[FLUSH >= 2 ? 4 : 1][4][4];
  if (FLUSH == 1) for (int mi = 0; mi < 4; mi++) for (int ni = 0; ni < 4; ni++) for (int r = 0; r < 4; r++) tot[mi][ni][r] = 0.0;
  if (FLUSH >= 2) for (int mi = 0; mi < 4; mi++) for (int ni = 0; ni < 4; ni++) for (int r = 0; r < 4; r++) thi[mi][ni][r] = tlo[mi][ni][r] = 0.f;
  for (int s = 0; s < RS_D; s++) { ra[s] = *Ap; rb[s] = *Bp; Ap += sa; Bp += sa; }
  for (int c0 = 0; c0 < n; c0 += CH) {
#pragma unroll
    for (int s = 0; s < CH; s++) {
      const int slot = s % RS_D;
#pragma unroll
      for (int mi = 0; mi < 4; mi++)
#pragma unroll
        for (int ni = 0; ni < 4; ni++)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(rb[slot][ni], ra[slot][mi],
                                                             (FLUSH && s == 0) ? (f4){0.f, 0.f, 0.f, 0.f} : acc[mi][ni], 0, 0, 0);
      if (LOADS) { ra[slot] = *Ap; rb[slot] = *Bp; Ap += sa; Bp += sa; }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (LOADS) { Ap -= CH * sa; Bp -= CH * sa; }   // stay inside the L1 / L2-resident window
    if (FLUSH == 1) {
#pragma unroll
      for (int mi = 0; mi < 4; mi++)
#pragma unroll
        for (int ni = 0; ni < 4; ni++)
#pragma unroll
          for (int r = 0; r < 4; r++) tot[mi][ni][r] += (double)acc[mi][ni][r];
    }
    if (FLUSH == 3) {   // tlo = sum of up to four chunks, thi = total
#pragma unroll
      for (int mi = 0; mi < 4; mi++)
#pragma unroll
        for (int ni = 0; ni < 4; ni++)
#pragma unroll
          for (int r = 0; r < 4; r++) tlo[mi][ni][r] += acc[mi][ni][r];
      if (((c0 / CH) & 3) == 3) {
#pragma unroll
        for (int mi = 0; mi < 4; mi++)
#pragma unroll
          for (int ni = 0; ni < 4; ni++)
#pragma unroll
            for (int r = 0; r < 4; r++) { thi[mi][ni][r] += tlo[mi][ni][r]; tlo[mi][ni][r] = 0.f; }
      }
    }
    if (FLUSH == 2) {
#pragma unroll
      for (int mi = 0; mi < 4; mi++)
#pragma unroll
        for (int ni = 0; ni < 4; ni++)
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const float a = thi[mi][ni][r], b = acc[mi][ni][r];
            const float sm = a + b, bb = sm - a;
            const float err = (a - (sm - bb)) + (b - bb);
            thi[mi][ni][r] = sm;
            tlo[mi][ni][r] += err;
          }
    }
  }
  float s = 0.f;
  for (int mi = 0; mi < 4; mi++) for (int ni = 0; ni < 4; ni++) {
    s += acc[mi][ni][0] + acc[mi][ni][3];
    if (FLUSH == 1) for (int r = 0; r < 4; r++) s += (float)tot[mi][ni][r];
    if (FLUSH >= 2) for (int r = 0; r < 4; r++) s += thi[mi][ni][r] + tlo[mi][ni][r];
  }
  sink[blockIdx.x * 256 + threadIdx.x] = s;
}

// The same loop with buffer loads: resource descriptor and running offset in SCALAR registers (s_add_u32 per step), a
// fixed 32-bit lane offset in one VGPR -- no 64-bit vector add in the loop
typedef unsigned u4 __attribute__((ext_vector_type(4)));
template <int RS_D, int OCC, int FLUSH = 0, int CH = 32, int REBASE = 0>
__global__ __launch_bounds__(256, OCC) void loop16s(const float *A, const float *B, long lda, int n, float *sink) {
  const int lane = threadIdx.x & 63, l15 = lane & 15, l4 = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const __amdgpu_buffer_rsrc_t Ar = __builtin_amdgcn_make_buffer_rsrc((void *)(A + (w & 1) * 64), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t Br = __builtin_amdgcn_make_buffer_rsrc((void *)(B + (w >> 1) * 64), 0, 0x7fffffff, 0x00020000);
  const unsigned off = (unsigned)((4 * l15 + (size_t)l4 * lda) * sizeof(float));
  const unsigned sa = (unsigned)(lda * 4 * sizeof(float));
  unsigned so = 0;
  f4 acc[4][4];
  for (int mi = 0; mi < 4; mi++) for (int ni = 0; ni < 4; ni++) acc[mi][ni] = (f4){0, 0, 0, 0};
  f4 ra[RS_D], rb[RS_D];
  double tot[FLUSH == 1 ? 4 : 1][4][4];
  if (FLUSH == 1) for (int mi = 0; mi < 4; mi++) for (int ni = 0; ni < 4; ni++) for (int r = 0; r < 4; r++) tot[mi][ni][r] = 0.0;
#define BLOAD(rs_) __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rs_, off, so, 0))
  // REBASE: the descriptor is rebuilt from a 64-bit uniform pointer that advances every step (no 4 GiB limit on K * lda)
  const char *Ac = reinterpret_cast<const char *>(A + (w & 1) * 64), *Bc = reinterpret_cast<const char *>(B + (w >> 1) * 64);
#define PLOAD(p_) __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(__builtin_amdgcn_make_buffer_rsrc((void *)(p_), 0, 0x7fffffff, 0x00020000), off, 0, 0))
  for (int s = 0; s < RS_D; s++) { ra[s] = BLOAD(Ar); rb[s] = BLOAD(Br); so += sa; }
  Ac += (size_t)RS_D * sa; Bc += (size_t)RS_D * sa;
  for (int c0 = 0; c0 < n; c0 += CH) {
#pragma unroll
    for (int s = 0; s < CH; s++) {
      const int slot = s % RS_D;
#pragma unroll
      for (int mi = 0; mi < 4; mi++)
#pragma unroll
        for (int ni = 0; ni < 4; ni++)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(rb[slot][ni], ra[slot][mi],
                                                             (FLUSH && s == 0) ? (f4){0.f, 0.f, 0.f, 0.f} : acc[mi][ni], 0, 0, 0);
      if (REBASE) { ra[slot] = PLOAD(Ac); rb[slot] = PLOAD(Bc); Ac += sa; Bc += sa; }
      else { ra[slot] = BLOAD(Ar); rb[slot] = BLOAD(Br); so += sa; }
      __builtin_amdgcn_sched_barrier(0);
    }
    so -= CH * sa; Ac -= (size_t)CH * sa; Bc -= (size_t)CH * sa;
    if (FLUSH == 1) {
#pragma unroll
      for (int mi = 0; mi < 4; mi++)
#pragma unroll
        for (int ni = 0; ni < 4; ni++)
#pragma unroll
          for (int r = 0; r < 4; r++) tot[mi][ni][r] += (double)acc[mi][ni][r];
    }
  }
  float s = 0.f;
  for (int mi = 0; mi < 4; mi++) for (int ni = 0; ni < 4; ni++) {
    s += acc[mi][ni][0] + acc[mi][ni][3];
    if (FLUSH == 1) for (int r = 0; r < 4; r++) s += (float)tot[mi][ni][r];
  }
  sink[blockIdx.x * 256 + threadIdx.x] = s;
}

// The library's global_load loop with the two 64-bit pointer increments done as v_add_co_u32 + v_addc_co_u32 pairs
// (inline asm) instead of the v_lshl_add_u64 the compiler picks on gfx950
template <int RS_D, int OCC>
__global__ __launch_bounds__(256, OCC) void loop16a(const float *A, const float *B, long lda, int n, float *sink) {
  const int lane = threadIdx.x & 63, l15 = lane & 15, l4 = lane >> 4, w = threadIdx.x >> 6;
  typedef const f4 __attribute__((address_space(1))) *gptr;
  gptr Ap = (gptr)(unsigned long long)(A + (w & 1) * 64 + 4 * l15 + (size_t)l4 * lda);
  gptr Bp = (gptr)(unsigned long long)(B + (w >> 1) * 64 + 4 * l15 + (size_t)l4 * lda);
  const unsigned long long sab = (unsigned long long)lda * 16;   // bytes per step
  const unsigned slo = (unsigned)sab, shi = (unsigned)(sab >> 32);
  f4 acc[4][4];
  for (int mi = 0; mi < 4; mi++) for (int ni = 0; ni < 4; ni++) acc[mi][ni] = (f4){0, 0, 0, 0};
  f4 ra[RS_D], rb[RS_D];
#define ADV(p_)                                                                                              \
  {                                                                                                          \
    unsigned long long a_ = (unsigned long long)(p_);                                                        \
    unsigned lo_ = (unsigned)a_, hi_ = (unsigned)(a_ >> 32);                                                 \
    asm volatile("v_add_co_u32 %0, vcc, %2, %0\n\tv_addc_co_u32 %1, vcc, %3, %1, vcc"                        \
                 : "+v"(lo_), "+v"(hi_) : "s"(slo), "v"(shi) : "vcc");                                       \
    p_ = (gptr)(((unsigned long long)hi_ << 32) | lo_);                                \
  }
  for (int s = 0; s < RS_D; s++) { ra[s] = *Ap; rb[s] = *Bp; ADV(Ap) ADV(Bp) }
  for (int c0 = 0; c0 < n; c0 += 32) {
#pragma unroll
    for (int s = 0; s < 32; s++) {
      const int slot = s % RS_D;
#pragma unroll
      for (int mi = 0; mi < 4; mi++)
#pragma unroll
        for (int ni = 0; ni < 4; ni++)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(rb[slot][ni], ra[slot][mi], acc[mi][ni], 0, 0, 0);
      ra[slot] = *Ap; rb[slot] = *Bp; ADV(Ap) ADV(Bp)
      __builtin_amdgcn_sched_barrier(0);
    }
    Ap -= 32 * (size_t)lda; Bp -= 32 * (size_t)lda;
  }
  float s = 0.f;
  for (int mi = 0; mi < 4; mi++) for (int ni = 0; ni < 4; ni++) s += acc[mi][ni][0] + acc[mi][ni][3];
  sink[blockIdx.x * 256 + threadIdx.x] = s;
}

// global_load with a scalar base (advanced on the scalar unit) + a 32-bit lane offset: the SADDR form of the same
// instruction the library uses today; the offset is laundered through an empty asm so that loop-strength reduction does
// not turn base + offset back into a per-lane 64-bit pointer
template <int RS_D, int OCC>
__global__ __launch_bounds__(256, OCC) void loop16g(const float *A, const float *B, long lda, int n, float *sink) {
  const int lane = threadIdx.x & 63, l15 = lane & 15, l4 = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const char *Ab = reinterpret_cast<const char *>(A + (w & 1) * 64), *Bb = reinterpret_cast<const char *>(B + (w >> 1) * 64);
  unsigned off = (unsigned)((4 * l15 + (size_t)l4 * lda) * sizeof(float));
  const size_t sa = (size_t)lda * 4 * sizeof(float);
  f4 acc[4][4];
  for (int mi = 0; mi < 4; mi++) for (int ni = 0; ni < 4; ni++) acc[mi][ni] = (f4){0, 0, 0, 0};
  f4 ra[RS_D], rb[RS_D];
#define GLOAD(b_) (*reinterpret_cast<const f4 *>((b_) + off))
  for (int s = 0; s < RS_D; s++) { ra[s] = GLOAD(Ab); rb[s] = GLOAD(Bb); Ab += sa; Bb += sa; }
  for (int c0 = 0; c0 < n; c0 += 32) {
    asm volatile("" : "+v"(off));   // once per trip: keeps base + offset from becoming a per-lane 64-bit induction variable
#pragma unroll
    for (int s = 0; s < 32; s++) {
      const int slot = s % RS_D;
#pragma unroll
      for (int mi = 0; mi < 4; mi++)
#pragma unroll
        for (int ni = 0; ni < 4; ni++)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(rb[slot][ni], ra[slot][mi], acc[mi][ni], 0, 0, 0);
      ra[slot] = GLOAD(Ab); rb[slot] = GLOAD(Bb); Ab += sa; Bb += sa;
      __builtin_amdgcn_sched_barrier(0);
    }
    Ab -= 32 * sa; Bb -= 32 * sa;
  }
  float s = 0.f;
  for (int mi = 0; mi < 4; mi++) for (int ni = 0; ni < 4; ni++) s += acc[mi][ni][0] + acc[mi][ni][3];
  sink[blockIdx.x * 256 + threadIdx.x] = s;
}

// the same 64 x 64 wave tile with v_mfma_f32_32x32x2_f32: per k-step of 2: A = 2 x 32 rows, B = 2 x 32 columns -> 4 MFMAs;
// operands: one float per lane per 32-row group and k (lane = row + 32 * k): loaded as float2 = rows r, k and k+... here
// simply two k-steps per 16-byte load pair (f4 = 2 row groups x 2 k... ) -- register traffic as the library would have it
template <int LOADS, int RS_D, int OCC>
__global__ __launch_bounds__(256, OCC) void loop32(const float *A, const float *B, long lda, int n, float *sink) {
  const int lane = threadIdx.x & 63, l31 = lane & 31, l2 = lane >> 5, w = threadIdx.x >> 6;
  // lane (row l31, k l2): f4 = rows {l31, l31+32} x k {l2, l2+2}: one load feeds two k-steps of 2 (4 k in all)
  const f4 *Ap = reinterpret_cast<const f4 *>(A + (w & 1) * 64 + 4 * (l31 & 15) + (size_t)(l2 + 2 * (l31 >> 4)) * lda);
  const f4 *Bp = reinterpret_cast<const f4 *>(B + (w >> 1) * 64 + 4 * (l31 & 15) + (size_t)(l2 + 2 * (l31 >> 4)) * lda);
  const size_t sa = (size_t)lda;
  f16 acc[2][2];
  for (int mi = 0; mi < 2; mi++) for (int ni = 0; ni < 2; ni++) for (int j = 0; j < 16; j++) acc[mi][ni][j] = 0.f;
  f4 ra[RS_D], rb[RS_D];
  for (int s = 0; s < RS_D; s++) { ra[s] = *Ap; rb[s] = *Bp; Ap += sa; Bp += sa; }
  for (int c0 = 0; c0 < n; c0 += 32) {
#pragma unroll
    for (int s = 0; s < 32; s++) {
      const int slot = s % RS_D;
#pragma unroll
      for (int kk = 0; kk < 2; kk++)
#pragma unroll
        for (int mi = 0; mi < 2; mi++)
#pragma unroll
          for (int ni = 0; ni < 2; ni++)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(rb[slot][2 * kk + ni], ra[slot][2 * kk + mi], acc[mi][ni], 0, 0, 0);
      if (LOADS) { ra[slot] = *Ap; rb[slot] = *Bp; Ap += sa; Bp += sa; }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (LOADS) { Ap -= 32 * sa; Bp -= 32 * sa; }
  }
  float s = 0.f;
  for (int mi = 0; mi < 2; mi++) for (int ni = 0; ni < 2; ni++) s += acc[mi][ni][0] + acc[mi][ni][15];
  sink[blockIdx.x * 256 + threadIdx.x] = s;
}

template <typename F>
static void run(const char *name, F launch, int occ) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int n = 32 * 2000, nblk = 256 * occ;
  for (int rep = 0; rep < 2; rep++) {
    hipEventRecord(e0, 0);
    launch(nblk, n);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
  }
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  // per wave and k-step of 4: 64 x 64 x 4 x 2 flop
  printf("%-44s %d workgroups per CU: %6.1f TFLOP/s (%.2f ms)\n", name, occ, (double)nblk * 4 * n * 64.0 * 64 * 4 * 2 / (ms * 1e-3) / 1e12, ms);
}

int main() {
  float *A, *B, *sink;
  const long lda = 64;   // 36 steps x 1 KiB per operand and wave: L1 / L2 resident
  hipMalloc(&A, sizeof(float) * lda * 64); hipMalloc(&B, sizeof(float) * lda * 64); hipMalloc(&sink, 4 * 256 * 1024);
  hipMemset(A, 0, sizeof(float) * lda * 64); hipMemset(B, 0, sizeof(float) * lda * 64);
#define RUN(K, L, D, O) run(#K " loads=" #L " depth=" #D, [&](int nb, int n) { hipLaunchKernelGGL((K<L, D, O>), dim3(nb), dim3(256), 0, 0, A, B, lda, n, sink); }, O)
  RUN(loop16, 0, 4, 1); RUN(loop16, 0, 4, 2); RUN(loop16, 0, 8, 3);
  RUN(loop16, 1, 4, 1); RUN(loop16, 1, 4, 2); RUN(loop16, 1, 8, 2); RUN(loop16, 1, 8, 3);
  RUN(loop32, 0, 4, 1); RUN(loop32, 0, 4, 2); RUN(loop32, 0, 8, 3);
  RUN(loop32, 1, 4, 1); RUN(loop32, 1, 4, 2); RUN(loop32, 1, 8, 2); RUN(loop32, 1, 8, 3);
#define RUNF(L, D, O, F, C) run("loop16 loads=" #L " depth=" #D " flush=" #F " chunk=" #C, [&](int nb, int n) { hipLaunchKernelGGL((loop16<L, D, O, F, C>), dim3(nb), dim3(256), 0, 0, A, B, lda, n, sink); }, O)
  RUNF(0, 4, 2, 1, 32); RUNF(1, 4, 2, 1, 32); RUNF(1, 4, 2, 1, 64); RUNF(1, 8, 1, 1, 32); RUNF(1, 16, 1, 1, 32);
  RUNF(0, 4, 2, 2, 32); RUNF(1, 4, 2, 2, 32); RUNF(1, 4, 2, 2, 64); RUNF(1, 8, 1, 2, 32);
  RUNF(0, 4, 2, 3, 32); RUNF(1, 4, 2, 3, 32); RUNF(1, 2, 2, 3, 32); RUNF(1, 8, 1, 3, 32); RUNF(1, 16, 1, 3, 32);
#define RUNS(D, O, F, C) run("loop16 scalar-base loads depth=" #D " flush=" #F " chunk=" #C, [&](int nb, int n) { hipLaunchKernelGGL((loop16s<D, O, F, C>), dim3(nb), dim3(256), 0, 0, A, B, lda, n, sink); }, O)
#define RUNR(D, O, F, C) run("loop16 rebased-descriptor loads depth=" #D " flush=" #F " chunk=" #C, [&](int nb, int n) { hipLaunchKernelGGL((loop16s<D, O, F, C, 1>), dim3(nb), dim3(256), 0, 0, A, B, lda, n, sink); }, O)
  RUNR(4, 2, 0, 32); RUNR(8, 3, 0, 32); RUNR(4, 2, 1, 32);
  RUNS(4, 1, 0, 32); RUNS(4, 2, 0, 32); RUNS(8, 3, 0, 32); RUNS(4, 2, 1, 32); RUNS(4, 2, 1, 64); RUNS(8, 1, 1, 32);
#define RUNA(D, O) run("loop16 global loads, 32-bit add pairs depth=" #D, [&](int nb, int n) { hipLaunchKernelGGL((loop16a<D, O>), dim3(nb), dim3(256), 0, 0, A, B, lda, n, sink); }, O)
  RUNA(4, 1); RUNA(4, 2); RUNA(8, 3);
#define RUNG(D, O) run("loop16 global loads, scalar base + lane offset depth=" #D, [&](int nb, int n) { hipLaunchKernelGGL((loop16g<D, O>), dim3(nb), dim3(256), 0, 0, A, B, lda, n, sink); }, O)
  RUNG(4, 1); RUNG(4, 2); RUNG(8, 3);
  return 0;
}
