// tools/gemm_lds_f64.hip -- the LDS-staged fp64 GEMM of round 1 (LDS-DMA staging, two buffers, one barrier per stage),
// moved out of the library in round 3: the product uses the register-streaming kernel of csrc/gemm.hip, which beat it
// 71.4 to 66.3 TFLOP/s at N = 32768, K = 512 (DESIGN.md 4.1 has the experiments that explain why).  Kept as the
// reference point of that comparison; not compiled into libgpak_hip.so.  To time it again, paste the kernel back
// into csrc/gemm.hip and launch it from gpak_launch_gemm_nt (git history: gemm.hip before this move).
// ---- LDS-staged kernel (GPAK_GEMM=lds) ---------------------------------------------------
template <bool TRAILING>
__global__ __launch_bounds__(256, 2) void gpak_gemm_nt_f64(int K, double alpha,
                                                            const double *A, long lda, const double *B,
                                                            long ldb,
                                                            double beta, double *C, long ldc, int rb0,
                                                            int cb0, int lower_skip, int mt, int nt, int k0_by_row,
                                                            int cyc_P, int cyc_rank, int cyc_tpb, int cyc_lt0) {
  // Workgroup -> tile map, XCD-aware: the dispatcher deals consecutive workgroup ids round-robin
  // over the 8 XCDs, so id b runs on XCD (b & 7) as that XCD's (b >> 3)-th workgroup.  Each XCD
  // walks 8x8 super-tiles: the 64 workgroups resident on its 32 CUs (2 per CU) cover one
  // super-tile and stream the SAME 8 A-row and 8 B-row panels k-slice by k-slice, so the XCD's
  // 4 MiB L2 serves 7 of every 8 operand reads (placement only affects speed, never results).
  int ti, tj;
  {
    const int b = blockIdx.x, q = b >> 3;
    const int slot = q & 63;
    const int ssel = (q >> 6) * 8 + (b & 7);
    const int SR = (mt + 7) >> 3, SC = (nt + 7) >> 3;
    int si, sj = 0;
    if (lower_skip) {
      int rem = ssel;
      while (sj < SC && rem >= SR - sj) { rem -= SR - sj; sj++; }
      si = sj + rem;
    } else {
      sj = ssel / SR;
      si = ssel - sj * SR;
    }
    if (sj >= SC) return;
    ti = si * 8 + (slot & 7);
    tj = sj * 8 + (slot >> 3);
    if (ti >= mt || tj >= nt) return;
    if (lower_skip && (rb0 + ti) < (cb0 + tj)) return;
  }
  // Block-column-cyclic column map (multi-GPU trailing update): the C columns are the rank's
  // OWN block columns stored side by side; local tile column (cyc_lt0 + tj) belongs to local
  // block lb = ./tpb, i.e. global block lb*P + rank.  A and B are then addressed by GLOBAL row
  // tile (virtual base), and tiles above the global diagonal are skipped.
  int gct = tj;  // tile row of the B operand
  int art = ti;  // tile row of the A operand
  if (cyc_P) {
    const int lt = cyc_lt0 + tj;
    gct = ((lt / cyc_tpb) * cyc_P + cyc_rank) * cyc_tpb + (lt % cyc_tpb);
    art = rb0 + ti;
    if (art < gct) return;
  }
  __shared__ double lds[2][2][KB][LDS_LD];
  const int t = threadIdx.x, lane = t & 63;
  // the wave index as a SCALAR: everything the LDS-DMA needs (LDS row, k-row of the source) is then
  // computed on the scalar unit.  Vector ALU instructions share the issue port with the MFMAs; ~35 of
  // them per stage for address arithmetic cost 6 % of the MFMA rate
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wr = w & 1, wc = w >> 1;
  const int l15 = lane & 15, l4 = lane >> 4;

  // source of a 16-B piece = scalar base (tile, k-row) + 32-bit lane offset: the global_load_lds
  // "saddr + voffset" form, so advancing k is scalar arithmetic
  const char *Au = reinterpret_cast<const char *>(A + (size_t)art * TM);
  const char *Bu = reinterpret_cast<const char *>(B + (size_t)gct * TN);
  const unsigned voff = (unsigned)lane * 16u;
  const size_t lda8 = (size_t)lda * 8, ldb8 = (size_t)ldb * 8;

  d4 acc[4][4];
#pragma unroll
  for (int mi = 0; mi < 4; mi++)
#pragma unroll
    for (int ni = 0; ni < 4; ni++) acc[mi][ni] = (d4){0.0, 0.0, 0.0, 0.0};

  const int nstage = K / KB;
  typedef const __attribute__((address_space(1))) void *gptr_t;
  typedef __attribute__((address_space(3))) void *lptr_t;

  // Direct-to-LDS staging: one global_load_lds_dwordx4 per wave writes one k-row of a tile
  // (64 lanes x 16 B = 128 doubles = 1 KiB contiguous in global memory AND in LDS).
  // Wave w stages k-rows w, w+4, w+8, w+12 of both operands: 8 instructions per stage.
#define GPAK_STAGE(buf_, kbase_)                                                                  \
  _Pragma("unroll") for (int s = 0; s < 4; s++) {                                                 \
    const size_t k_ = (size_t)(kbase_) + w + 4 * s;                                               \
    __builtin_amdgcn_global_load_lds((gptr_t)(Au + k_ * lda8 + voff),                             \
                                     (lptr_t)&lds[buf_][0][w + 4 * s][0], 16, 0, 0);              \
    __builtin_amdgcn_global_load_lds((gptr_t)(Bu + k_ * ldb8 + voff),                             \
                                     (lptr_t)&lds[buf_][1][w + 4 * s][0], 16, 0, 0);              \
  }

  // k0_by_row: A (and B) are upper triangular in (row, k), so tile row ti only has k >= ti*128
  const int st_begin = k0_by_row ? (rb0 + ti) * (TM / KB) : 0;
  GPAK_STAGE(st_begin & 1, (size_t)st_begin * KB)
  __syncthreads();  // emits vmcnt(0) for the in-flight LDS-DMA, then the barrier

#define GPAK_COMPUTE(buf_)                                                                          \
  _Pragma("unroll") for (int kk = 0; kk < KB / 4; kk++) {                                           \
    double a[4], b[4];                                                                              \
    _Pragma("unroll") for (int mi = 0; mi < 4; mi++)                                                \
        a[mi] = lds[buf_][0][kk * 4 + l4][wr * 64 + mi * 16 + l15];                                 \
    _Pragma("unroll") for (int ni = 0; ni < 4; ni++)                                                \
        b[ni] = lds[buf_][1][kk * 4 + l4][wc * 64 + ni * 16 + l15];                                 \
    _Pragma("unroll") for (int mi = 0; mi < 4; mi++)                                                \
        _Pragma("unroll") for (int ni = 0; ni < 4; ni++)                                            \
            acc[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(b[ni], a[mi], acc[mi][ni], 0, 0, 0); \
  }

  // Two stages per trip so that the buffer index is a compile-time constant (LDS offsets become
  // immediates, no per-stage vector address arithmetic); st_begin is even (0 or a multiple of 8).
  int st = st_begin;
  for (; st + 2 < nstage; st += 2) {
    // stage st+1 streams into the other buffer while this stage's MFMAs run
    GPAK_STAGE(1, (size_t)(st + 1) * KB)
    GPAK_COMPUTE(0)
    // keep this stage's MFMAs ABOVE the wait+barrier: without the fence hipcc reads all
    // fragments up front and sinks 61 of the 64 MFMAs below the barrier, so every wave sits
    // out the full LDS-DMA latency before it computes
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    GPAK_STAGE(0, (size_t)(st + 2) * KB)
    GPAK_COMPUTE(1)
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
  }
  if (st + 2 == nstage) {  // an even number of stages: one more full stage before the last
    GPAK_STAGE(1, (size_t)(st + 1) * KB)
    GPAK_COMPUTE(0)
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
  }

  // last stage: no more staging; the C tile (beta != 0) is fetched underneath its MFMAs
  // lane holds rows (.. + l15), columns (.. + l4 + 4*reg)
  double *Cg = C + (size_t)art * TM + wr * 64 + l15 + ((size_t)tj * TN + wc * 64 + l4) * ldc;
  const int lbuf = (nstage - 1) & 1;
  if (beta == 0.0) {
    GPAK_COMPUTE(lbuf)
#pragma unroll
    for (int mi = 0; mi < 4; mi++)
#pragma unroll
      for (int ni = 0; ni < 4; ni++)
#pragma unroll
        for (int r = 0; r < 4; r++)
          Cg[mi * 16 + (size_t)(ni * 16 + 4 * r) * ldc] = alpha * acc[mi][ni][r];
  } else {
    GPAK_COMPUTE(lbuf)
    // read-modify-write of the C tile, 16 rows x 64 columns of the wave's sub-tile at a time
    // (prefetching a row group under the last stage's MFMAs was tried: with the fragment
    // prefetch hipcc does there it overflows 256 VGPRs and spills -- slower, not faster)
#pragma unroll
    for (int mi = 0; mi < 4; mi++) {
      double c[4][4];
#pragma unroll
      for (int ni = 0; ni < 4; ni++)
#pragma unroll
        for (int r = 0; r < 4; r++) c[ni][r] = Cg[mi * 16 + (size_t)(ni * 16 + 4 * r) * ldc];
#pragma unroll
      for (int ni = 0; ni < 4; ni++)
#pragma unroll
        for (int r = 0; r < 4; r++)
          Cg[mi * 16 + (size_t)(ni * 16 + 4 * r) * ldc] = fma(alpha, acc[mi][ni][r], beta * c[ni][r]);
    }
  }
#undef GPAK_STAGE
#undef GPAK_COMPUTE
}


// ---- the fp32 sibling (was csrc/gemm_f32.hip gpak_gemm_nt_f32; needs that file's f4 / TM / TN / KB32 definitions) ----
__global__ __launch_bounds__(256, 2) void gpak_gemm_nt_f32(int K, float alpha, const float *A, long lda,
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
  __shared__ float lds[2][2][KB32][TM];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int wr = w & 1, wc = w >> 1;
  const int l15 = lane & 15, l4 = lane >> 4;
  // LDS-DMA piece of a wave instruction: lanes 0..31 -> k-row 2p, lanes 32..63 -> k-row 2p+1
  const int lrow = (lane & 31) * 4, lk = lane >> 5;
  const float *Ag = A + (size_t)ti * TM + lrow + (size_t)lk * lda;
  const float *Bg = B + (size_t)tj * TN + lrow + (size_t)lk * ldb;

  f4 acc[4][4];
#pragma unroll
  for (int mi = 0; mi < 4; mi++)
#pragma unroll
    for (int ni = 0; ni < 4; ni++) acc[mi][ni] = (f4){0.f, 0.f, 0.f, 0.f};

  const int nstage = K / KB32;
  typedef const __attribute__((address_space(1))) void *gptr_t;
  typedef __attribute__((address_space(3))) void *lptr_t;
  // wave w stages k-row pairs w, w+4, w+8, w+12 (rows 2p, 2p+1) of both operands
#define GPAK_STAGE32(buf_, kbase_)                                                                   \
  _Pragma("unroll") for (int s = 0; s < 4; s++) {                                                    \
    const int p_ = w + 4 * s;                                                                        \
    const size_t k_ = (size_t)(kbase_) + 2 * p_;                                                     \
    __builtin_amdgcn_global_load_lds((gptr_t)(Ag + k_ * lda), (lptr_t)&lds[buf_][0][2 * p_][0], 16, 0, 0); \
    __builtin_amdgcn_global_load_lds((gptr_t)(Bg + k_ * ldb), (lptr_t)&lds[buf_][1][2 * p_][0], 16, 0, 0); \
  }
#define GPAK_COMPUTE32(buf_)                                                                         \
  _Pragma("unroll") for (int kk = 0; kk < KB32 / 4; kk++) {                                          \
    float a[4], b[4];                                                                                \
    _Pragma("unroll") for (int mi = 0; mi < 4; mi++) a[mi] = lds[buf_][0][kk * 4 + l4][wr * 64 + mi * 16 + l15]; \
    _Pragma("unroll") for (int ni = 0; ni < 4; ni++) b[ni] = lds[buf_][1][kk * 4 + l4][wc * 64 + ni * 16 + l15]; \
    _Pragma("unroll") for (int mi = 0; mi < 4; mi++)                                                 \
        _Pragma("unroll") for (int ni = 0; ni < 4; ni++)                                             \
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[ni], a[mi], acc[mi][ni], 0, 0, 0);  \
  }

  GPAK_STAGE32(0, 0)
  __syncthreads();
  for (int st = 0; st + 1 < nstage; st++) {
    const int buf = st & 1;
    GPAK_STAGE32(buf ^ 1, (size_t)(st + 1) * KB32)
    GPAK_COMPUTE32(buf)
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
  }
  const int lbuf = (nstage - 1) & 1;
  GPAK_COMPUTE32(lbuf)
  // lane holds row (.. + l15), columns (.. + 4*l4 + reg)
  float *Cg = C + (size_t)ti * TM + wr * 64 + l15 + ((size_t)tj * TN + wc * 64 + 4 * l4) * ldc;
#pragma unroll
  for (int mi = 0; mi < 4; mi++)
#pragma unroll
    for (int ni = 0; ni < 4; ni++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        float *p = Cg + mi * 16 + (size_t)(ni * 16 + r) * ldc;
        const float v = alpha * acc[mi][ni][r];
        *p = (beta == 0.f) ? v : fmaf(beta, *p, v);
      }
#undef GPAK_STAGE32
#undef GPAK_COMPUTE32
}

