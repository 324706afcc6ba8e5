// tools/panel_su_fused.hip -- NEGATIVE RESULT of round 3, kept as the record of what was tried; not
// built into libgpak_hip.so.  One 128-column step's panel solve + in-panel update fused into one launch with tile-row
// counters in global memory (agent-scope release / acquire).  Wired into factor_panel_128 behind GPAK_SU_MAX_MT it gave
// BIT-IDENTICAL results and never timed out, but was slower at every size (tools/su_ab.py, profiles/r03_su_ab.txt):
//   N=2048  factor 0.91 -> 1.09 ms     N=4096  2.10 -> 2.91 ms     N=8192  6.15 -> 10.0 ms  (about +60 us per step)
// Why: gfx950's eight XCDs have private L2s.  Making one workgroup's stores visible to a workgroup on another XCD is an
// agent-scope release (write back the L2: __threadfence / buffer_wbl2) on the producer and an agent-scope acquire
// (invalidate: buffer_inv) on every poll of the consumer -- on ~500 workgroups per step that costs far more than the
// launch gap and first-load latency it was meant to save.  Intra-launch producer/consumer hand-offs between
// workgroups are therefore the wrong tool for the panel chain on this part; what remains is fewer, cheaper launches
// (DESIGN.md 10.3).
// panel_su.hip -- one 128-column step of the panel factorisation AFTER its diagonal block: panel solve and in-panel
// update fused into ONE launch (round 3; the panel chain of ldB2_exact's chol, GP_Utils.cpp:872-915).
//
// gpak_factor_panel does, per 128-column sub-panel s of a panel [J, J+W):  potrf128 (diagonal tile + its inverse),
// then P := A inv(D_s)^T for all rows below (a K = 128 product, one launch) and A[:, c] -= P P_c^T for the later
// sub-panels c of the panel (a second K = 128 launch).  In the chain-bound part of the factorisation (few row tiles)
// those two launches are ~20 of the ~55 us a step costs, most of it launch latency and the first loads.  Here a
// workgroup owns a 16-row STRIP of the rows below the diagonal tile (four waves, 16 x 32 each -- the shape of
// gpak_gemm_nt_f64_rs32<1, 1>) and does both: it solves its strip (result kept in LDS as the A operand of what
// follows, and stored in place), then updates its rows of the later sub-panels.  The update's B operand is the solved
// tile row c of the SAME launch, produced by the eight strips of that tile row: a counter per tile row in global memory
// (release after the strip's stores, acquire before the first use) orders them.  Those strips have the lowest
// workgroup indices of the launch and never wait for anything themselves before they publish, so the launch cannot
// deadlock however few workgroups are resident; the poll is bounded all the same (a time-out sets the error word and
// the workgroup leaves -- the host then reports GPAK_EHIP instead of hanging the queue).
#include "gpak_internal.h"

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

#define SU_D 8          // operand ring depth (k-steps in flight)
#define SU_STRIP 16     // rows per workgroup
#define SU_SPT (GPAK_TILE / SU_STRIP)   // strips per 128-row tile row

// M: matrix, global (row, column) addressing; j: first column of the sub-panel just factored (its diagonal tile is
// rows [j, j+128)); jend: end of the panel (J + W); inv: D_s^-1 (first image of the block's pair, column-major 128);
// ctr: jend/128 - j/128 - 1 (<= 3... any) counters, zero on entry, ctr[k] counts the solved strips of tile row j/128+1+k;
// err: error word (set to 1 on a time-out).
__global__ __launch_bounds__(256) void gpak_panel_su_f64(double *__restrict__ M, long ld, int j, int jend,
                                                          const double *__restrict__ inv, int *__restrict__ ctr,
                                                          int *__restrict__ err) {
  __shared__ double Ps[GPAK_TILE][SU_STRIP];   // the solved strip, k-major: Ps[k][row]
  __shared__ int bail;
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int l15 = lane & 15, l4 = lane >> 4;
  const int row0 = j + GPAK_TILE + blockIdx.x * SU_STRIP;      // first row of my strip
  const int tr = blockIdx.x / SU_SPT;                          // my tile row, relative to the first one below the diagonal
  __builtin_amdgcn_s_setprio(2);

  // ---- S: P_strip = A_strip inv^T  (K = 128; wave w: columns [32 w, 32 w + 32))
  d4 acc[2];
  {
    const double *Ap = M + row0 + l15 + (size_t)(j + l4) * ld;
    const d2 *Bp = reinterpret_cast<const d2 *>(inv + w * 32 + 2 * l15 + (size_t)l4 * GPAK_TILE);
    const size_t sa = 4 * (size_t)ld, sb = 2 * (size_t)GPAK_TILE;   // 4 k-columns; B in 16-B units
    acc[0] = acc[1] = (d4){0.0, 0.0, 0.0, 0.0};
    double ra[SU_D];
    d2 rb[SU_D];
#pragma unroll
    for (int s = 0; s < SU_D; s++) { ra[s] = *Ap; rb[s] = *Bp; Ap += sa; Bp += sb; }
#pragma unroll
    for (int g = 0; g < 32; g += SU_D) {
#pragma unroll
      for (int s = 0; s < SU_D; s++) {
        acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(rb[s][0], ra[s], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(rb[s][1], ra[s], acc[1], 0, 0, 0);
        if (g + SU_D < 32) { ra[s] = *Ap; rb[s] = *Bp; Ap += sa; Bp += sb; }
      }
    }
  }
  __syncthreads();   // in place: every wave has read the strip's 128 columns before anybody overwrites them
  // lane holds of accumulator ni, register r: row l15, column 32 w + 2 (l4 + 4 r) + ni
#pragma unroll
  for (int ni = 0; ni < 2; ni++)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int c = w * 32 + 2 * (l4 + 4 * r) + ni;
      const double v = acc[ni][r];
      M[row0 + l15 + (size_t)(j + c) * ld] = v;
      Ps[c][l15] = v;
    }
  const int ncol = (jend - j) / GPAK_TILE - 1;   // later sub-panels of the panel
  if (ncol <= 0) return;
  // publish: my strip of tile row tr is solved (only the tile rows that serve as B operands are counted)
  __threadfence();
  __syncthreads();
  if (tr < ncol && t == 0) __hip_atomic_fetch_add(ctr + tr, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);

  // ---- U: A[strip, sub-panel c] -= P_strip P_c^T for c = 1 .. ncol, lower tiles only (tile row >= tile column)
  for (int c = 0; c < ncol && c <= tr; c++) {
    // wait until the eight strips of tile row c have published (bounded); thread 0 polls, the decision is the workgroup's
    if (t == 0) {
      int spins = 0, give_up = 0;
      while (__hip_atomic_load(ctr + c, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < SU_SPT) {
        __builtin_amdgcn_s_sleep(4);
        if (++spins > (1 << 22) || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
          atomicExch(err, 1);
          give_up = 1;
          break;
        }
      }
      bail = give_up;
    }
    __syncthreads();
    if (bail) return;
    __threadfence();
    const int rc = j + GPAK_TILE * (c + 1);                    // rows of tile row c = columns of sub-panel c
    const d2 *Bp = reinterpret_cast<const d2 *>(M + rc + w * 32 + 2 * l15 + (size_t)(j + l4) * ld);
    const size_t sb = 2 * (size_t)ld;
    acc[0] = acc[1] = (d4){0.0, 0.0, 0.0, 0.0};
    d2 rb[SU_D];
#pragma unroll
    for (int s = 0; s < SU_D; s++) { rb[s] = *Bp; Bp += sb; }
#pragma unroll
    for (int g = 0; g < 32; g += SU_D) {
#pragma unroll
      for (int s = 0; s < SU_D; s++) {
        const double a = Ps[4 * (g + s) + l4][l15];
        acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(rb[s][0], a, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(rb[s][1], a, acc[1], 0, 0, 0);
        if (g + SU_D < 32) { rb[s] = *Bp; Bp += sb; }
      }
    }
#pragma unroll
    for (int ni = 0; ni < 2; ni++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        double *p = M + row0 + l15 + (size_t)(rc + w * 32 + 2 * (l4 + 4 * r) + ni) * ld;
        *p = *p - acc[ni][r];
      }
  }
}

// rows below the diagonal tile of sub-panel j: (Np - j - 128) / 16 strips.  ctr must hold (jend - j) / 128 - 1 zeroed ints.
void gpak_launch_panel_su(hipStream_t st, double *M, long ld, int Np, int j, int jend, const double *inv, int *ctr, int *err) {
  const int rows = Np - j - GPAK_TILE;
  if (rows <= 0) return;
  hipLaunchKernelGGL(gpak_panel_su_f64, dim3(rows / SU_STRIP), dim3(256), 0, st, M, ld, j, jend, inv, ctr, err);
}
