// store_bw.hip -- what store pattern reaches the HBM write roof on MI355X?  (tools only; not part of the library)
//   hipcc -O3 --offload-arch=gfx950 tools/store_bw.hip -o gpurun_out/store_bw && gpurun_out/store_bw
// Variants: contiguous grid-stride 16-B stores (the calibration kernel), the Gram fill's pattern (a wave writes
// 1 KiB of one matrix column, then moves to the next column, ld*8 bytes away), taller tiles (2 or 4 KiB of a column
// per wave before moving on), plain / non-temporal.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <bool NT>
__device__ __forceinline__ void st16(double *p, double a, double b) {
  if (NT) {
    __builtin_nontemporal_store(a, p);
    __builtin_nontemporal_store(b, p + 1);
  } else {
    *reinterpret_cast<double2 *>(p) = make_double2(a, b);
  }
}

template <bool NT, int UNROLL>
__global__ __launch_bounds__(256) void k_contig(double *out, size_t n2) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i + (UNROLL - 1) * stride < n2; i += UNROLL * stride) {
#pragma unroll
    for (int u = 0; u < UNROLL; u++) st16<NT>(out + 2 * (i + u * stride), 1.0, 2.0);
  }
  for (; i < n2; i += stride) st16<NT>(out + 2 * i, 1.0, 2.0);
}

// the fill's pattern: tile = (64 * 2 * RPT) rows x COLS columns per workgroup of 4 waves; wave w writes column
// w, w+4, ...; per column RPT store instructions of 1 KiB each (consecutive 128-row groups)
template <bool NT, int RPT, int COLS>
__global__ __launch_bounds__(256) void k_tile(double *C, long ld, int lower_only) {
  const int rows = 128 * RPT;
  const int row0 = blockIdx.x * rows, col0 = blockIdx.y * COLS;
  if (lower_only && row0 + rows <= col0) return;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int c = w; c < COLS; c += 4) {
    double *p = C + row0 + 2 * lane + (size_t)(col0 + c) * ld;
#pragma unroll
    for (int r = 0; r < RPT; r++) st16<NT>(p + 128 * r, 1.0 + c, 2.0 + r);
  }
}

int main() {
  const int N = 32768;
  const long ld = N + 32;
  double *buf;
  CHECK(hipMalloc(&buf, sizeof(double) * (size_t)ld * N));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const size_t bytes_full = sizeof(double) * (size_t)ld * N;
  auto timeit = [&](const char *name, double bytes, auto launch) {
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    const int reps = 5;
    for (int r = 0; r < reps; r++) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-46s %8.3f ms  %7.0f GB/s\n", name, ms / reps, bytes / (ms / reps * 1e-3) / 1e9);
  };
  const size_t n2 = bytes_full / 16;
  for (int blocks : {1024, 2048, 4096, 8192, 16384, 65536})
    for (int nt = 0; nt < 2; nt++) {
      char name[96];
      snprintf(name, sizeof name, "contig u1 blocks=%d %s", blocks, nt ? "nt" : "plain");
      if (nt) timeit(name, (double)bytes_full, [&] { hipLaunchKernelGGL((k_contig<true, 1>), dim3(blocks), dim3(256), 0, 0, buf, n2); });
      else timeit(name, (double)bytes_full, [&] { hipLaunchKernelGGL((k_contig<false, 1>), dim3(blocks), dim3(256), 0, 0, buf, n2); });
      snprintf(name, sizeof name, "contig u4 blocks=%d %s", blocks, nt ? "nt" : "plain");
      if (nt) timeit(name, (double)bytes_full, [&] { hipLaunchKernelGGL((k_contig<true, 4>), dim3(blocks), dim3(256), 0, 0, buf, n2); });
      else timeit(name, (double)bytes_full, [&] { hipLaunchKernelGGL((k_contig<false, 4>), dim3(blocks), dim3(256), 0, 0, buf, n2); });
    }
  // tile patterns over the lower triangle (what the fill writes): bytes = lower tiles only
  auto lower_bytes = [&](int rows, int cols) {
    double b = 0;
    for (int r0 = 0; r0 < N; r0 += rows)
      for (int c0 = 0; c0 < N; c0 += cols)
        if (!(r0 + rows <= c0)) b += 8.0 * rows * cols;
    return b;
  };
#define TILE_CASE(NT_, RPT_, COLS_)                                                                              \
  {                                                                                                              \
    char name[96];                                                                                               \
    snprintf(name, sizeof name, "tile %dx%d lower %s", 128 * RPT_, COLS_, NT_ ? "nt" : "plain");                \
    timeit(name, lower_bytes(128 * RPT_, COLS_), [&] {                                                           \
      hipLaunchKernelGGL((k_tile<NT_, RPT_, COLS_>), dim3(N / (128 * RPT_), N / COLS_), dim3(256), 0, 0, buf, ld, 1); \
    });                                                                                                          \
    snprintf(name, sizeof name, "tile %dx%d full  %s", 128 * RPT_, COLS_, NT_ ? "nt" : "plain");                \
    timeit(name, 8.0 * N * (double)N, [&] {                                                                      \
      hipLaunchKernelGGL((k_tile<NT_, RPT_, COLS_>), dim3(N / (128 * RPT_), N / COLS_), dim3(256), 0, 0, buf, ld, 0); \
    });                                                                                                          \
  }
  TILE_CASE(false, 1, 64) TILE_CASE(true, 1, 64)
  TILE_CASE(false, 2, 32) TILE_CASE(true, 2, 32)
  TILE_CASE(false, 4, 16) TILE_CASE(true, 4, 16)
  TILE_CASE(false, 4, 32) TILE_CASE(true, 4, 32)
  TILE_CASE(false, 2, 64) TILE_CASE(true, 2, 64)
  TILE_CASE(false, 1, 128) TILE_CASE(true, 1, 128)
  hipFree(buf);
  return 0;
}
