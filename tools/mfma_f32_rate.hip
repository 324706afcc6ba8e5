// Issue rate of the fp32 MFMAs (gfx950): v_mfma_f32_16x16x4_f32 vs v_mfma_f32_32x32x2_f32, 1..3 waves per SIMD on the
// whole chip, registers only, timed with events.  hipcc --offload-arch=gfx950 -O3 -o tools/bin/mfma_f32_rate tools/mfma_f32_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16 __attribute__((ext_vector_type(16)));
template <int SHAPE>
__global__ void rate(long long *out, float *sink, int iters) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float a = 1.0f + lane * 1e-6f, b = 1.0f - lane * 1e-6f;
  long long t0, t1;
  float s = 0.f;
  if (SHAPE == 16) {
    f4 acc[8];
    for (int i = 0; i < 8; i++) acc[i] = (f4){0, 0, 0, 0};
    __syncthreads();
    t0 = wall_clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int i = 0; i < 8; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 8; i++) s += acc[i][0];
    t1 = wall_clock64();
  } else {
    f16 acc[4];
    for (int i = 0; i < 4; i++) for (int j = 0; j < 16; j++) acc[i][j] = 0.f;
    __syncthreads();
    t0 = wall_clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int i = 0; i < 4; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 4; i++) s += acc[i][0];
    t1 = wall_clock64();
  }
  if (lane == 0) out[w] = t1 - t0;
  sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
  long long *out; float *sink;
  hipMalloc(&out, 64 * 8); hipMalloc(&sink, 1024 * 4 * 4);
  // 256 x occupancy workgroups of 4 waves
  for (int shape : {16, 32})
    for (int occ : {1, 2, 3}) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      const int iters2 = 20000, nblk = 256 * occ;
      float *sink2; hipMalloc(&sink2, 4 * 256 * nblk + 4096);
      for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0, 0);
        if (shape == 16) hipLaunchKernelGGL(rate<16>, dim3(nblk), dim3(256), 0, 0, out, sink2, iters2);
        else hipLaunchKernelGGL(rate<32>, dim3(nblk), dim3(256), 0, 0, out, sink2, iters2);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
      }
      float ms = 0; hipEventElapsedTime(&ms, e0, e1);
      printf("%dx%d MFMA, whole chip, %d workgroups of 4 waves per CU: %.1f TFLOP/s (%.2f ms)\n", shape, shape, occ,
             (double)nblk * 4 * iters2 * 8 * 2048.0 / (ms * 1e-3) / 1e12, ms);
      hipFree(sink2);
    }
  return 0;
}
