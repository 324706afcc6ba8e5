// potrf.hip -- blocked right-looking fp64 Cholesky (lower, in place) for gfx950.
//
// Replaces arma::chol inside GP_utils::ldB2_exact (GP_Utils.cpp:881, 903; LAPACK dpotrf).
// The device keeps L = R^T (lower) where the reference keeps the upper R.
//
//   for each outer block column J (width nb_outer, multiple of 128):
//     for each 128-column j inside it:
//       [potrf128]  LDS-resident factorisation of the 128x128 diagonal block + its
//                   inverse (and inverse transpose, for the back substitution)
//       [gemm_nt]   panel solve  P := P * inv(L_jj)^T            (alpha=1, beta=0)
//       [gemm_nt]   update of the remaining columns of the outer block (K = 128)
//     [gemm_nt]     trailing update of everything right of the outer block (K = nb_outer):
//                   the N^3/3 term, MFMA-bound.
#include <algorithm>
#include <atomic>

#include "gpak_internal.h"

#define PB 128

typedef double d4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double gpak_rdlane(double v, int l) {
  int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
  int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}

// LDS image of the 128x128 block: only the 36 lower 16x16 tiles, each column-major with
// leading dimension 16 (tile (rt,ct), ct<=rt, at slot rt(rt+1)/2+ct).  A 16-double tile column is 32 banks wide,
// so an MFMA fragment read (16 rows x 4 k) is conflict-free.
__device__ __forceinline__ int gpak_tix(int rt, int ct, int i, int k) {
  return ((rt * (rt + 1) / 2 + ct) << 8) + (k << 4) + i;
}

// One workgroup (8 waves, or 4 in the co-resident build) factors the 128x128 block held in LDS, 16 columns at a
// time, and returns L, L^-1 and L^-T (DESIGN.md section 4.2 has the measurements behind every choice below).
//   diagonal 16x16 block (wave 0, vector pipe): lane c < 16 owns column c of the symmetric block, lanes 16..31 the
//     columns of an identity; pivots by v_readlane, 1/sqrt by v_rsq_f64 + a coupled Goldschmidt iteration; the same
//     row operations turn the identity into the block's inverse.
//   panel  P := P * inv(D)^T  and trailing update  C -= P P^T: v_mfma_f64_16x16x4_f64 with operands read from LDS,
//     accumulators kept transposed (conflict-free), several independent tiles in flight per wave.
//   overlap: wave 0 updates the NEXT diagonal tile first and factors it while the helper waves (on the other three
//     SIMDs: f64 MFMAs and f64 vector instructions share the FMA units) finish the trailing update.
//   the 128x128 inverse rides along: block forward substitution on R = I in right-looking form, phase A by the waves
//     without a panel tile, phase B dealt to the helpers, kept transposed in a second LDS array.
//   results are stored as soon as they are final (column kb of L, row kb of the inverse during step kb).
//   A      : block in global memory (column-major, ld); lower triangle is read
//   inv    : 2 x (128x128) doubles: inv(L) then inv(L)^T, column-major ld 128
//   col0   : global column of the block (for the not-positive-definite report)
//   info   : atomicMin of the first failing column (1-based)
#ifdef GPAK_POTRF_TIMING
__device__ long long gpak_potrf_dbg[64];
// stamps go to LDS and leave for memory when the kernel is over: a global store per stamp made the wave wait for that
// store (behind everything else the CU had in flight) the next time its register was reused -- thousands of cycles
// that showed up in whichever phase came next
#define GPAK_TS(i_) do { if (t == 0) gpak_ts_lds[i_] = (long long)__builtin_readcyclecounter(); } while (0)
#define GPAK_TSW(w_, i_) do { if (t == 64 * (w_)) gpak_ts_lds[i_] = (long long)__builtin_readcyclecounter(); } while (0)
// a stamp that first pins the sixteen registers of the diagonal block, so that the arithmetic before it cannot sink below it
#define GPAK_TS_PIN(x_, i_) do { asm volatile("" : "+v"(x_[0]), "+v"(x_[1]), "+v"(x_[2]), "+v"(x_[3]), "+v"(x_[4]), "+v"(x_[5]), "+v"(x_[6]), "+v"(x_[7]), "+v"(x_[8]), "+v"(x_[9]), "+v"(x_[10]), "+v"(x_[11]), "+v"(x_[12]), "+v"(x_[13]), "+v"(x_[14]), "+v"(x_[15])); GPAK_TS(i_); } while (0)
#define GPAK_TS_DECL long long *const gpak_ts_lds = reinterpret_cast<long long *>(SM + GPAK_POTRF_SM_DOUBLES + 28 * 256);
#define GPAK_TS_FLUSH() do { __syncthreads(); if (t < 64) gpak_potrf_dbg[t] = gpak_ts_lds[t]; } while (0)
extern "C" int gpak_dev_potrf_timing(long long *out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(gpak_potrf_dbg), sizeof(long long) * 64) == hipSuccess ? 0 : -1;
}
#else
#define GPAK_TS(i_) do { } while (0)
#define GPAK_TSW(w_, i_) do { } while (0)
#define GPAK_TS_PIN(x_, i_) do { } while (0)
#define GPAK_TS_DECL
#define GPAK_TS_FLUSH() do { } while (0)
#endif
#define GPAK_POTRF_SM_DOUBLES (36 * 256 + 8 * 272 + 256)
#define GPAK_POTRF_LDS_BYTES ((GPAK_POTRF_SM_DOUBLES + 28 * 256 + 64) * 8)
// NW = 8 waves: the fast one (26.5 us).  Its waves need 141 VGPRs, two per SIMD: beside the bulk trailing update (two
// 210-VGPR waves per SIMD on every CU) it cannot start until a CU has lost one of its two bulk workgroups AND the
// dispatcher gives the hole to this kernel -- observed to take up to the whole bulk update (7 ms, tools/chain_trace.sh
// 32768).  NW = 4 waves capped at 80 VGPRs, one per SIMD, fits into the 80 registers two bulk waves leave free: it is
// slower alone but starts at once, and is what the factorisation launches while bulk updates are large.
template <int NW>
__device__ __forceinline__ void gpak_potrf128_body(double *A, long ld, double *__restrict__ inv, int col0, int *info,
                                                    int zero_inv) {
  // one LDS array, so that "which of these does this lane write" is an integer select on an offset and not a choice
  // between pointers (which the compiler turns into a branch tree): T = the 36 lower tiles, then dd = the diagonals
  // of the 8 diagonal-block inverses, then one dump slot per lane
  // (the same goes for reads: "this lane's value or a constant" written as a select on the loaded value comes back
  // from the compiler as a branch around the load, and a taken branch costs more than the load -- so the constants
  // live in LDS too: a 16x16 identity tile, whose off-diagonal entries double as the zero)
  // (dynamic LDS: with a static 132 KiB the compiler concludes that only one workgroup fits on a CU anyway and
  // ignores the register budget the 4-wave build asks for)
  extern __shared__ double gpak_potrf_lds[];
  double *const SM = gpak_potrf_lds;
  double *const T = SM;
  // WI0: the inverses of the 8 diagonal blocks as full column-major 16x16 tiles (their upper halves are the exact zeros
  // the elimination leaves there), leading dimension 17 so that both the column-per-lane write and the MFMA fragment
  // reads are conflict-free; ID0: a 16x16 identity
  constexpr int WI0 = 36 * 256, ID0 = WI0 + 8 * 272;
  GPAK_TS_DECL
  // the 28 strictly-lower 16x16 tiles of the block inverse Y = L^-1, kept TRANSPOSED (tile (i, j), i > j, holds
  // Y[i,j]^T) so that every MFMA operand and accumulator access below is contiguous across lane&15
  double *const Y2 = SM + GPAK_POTRF_SM_DOUBLES;
  // this workgroup is the serial link of the panel chain and shares its CU with two trailing-update
  // waves per SIMD: let its instructions win the issue arbitration
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);   // wave-uniform for the compiler too: scalar branches, no exec-mask loops
  // wave 0 carries the serial chain (the diagonal blocks): it outranks its own helpers, which outrank everybody else
  if (w == 0) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(2);
  const int l15 = lane & 15, l4 = lane >> 4;
  const int lane_ = lane, l15_ = l15, l4_ = l4;
  double *invT = inv + PB * PB;

  // inverse of diagonal block r, element (j,k)
  auto dinv = [&](int r, int j, int k) -> double {
    return SM[WI0 + 272 * r + 17 * k + j];
  };

  // ---- the 128x128 inverse rides along with the factorisation (it used to be a serial 19 k-cycle epilogue).
  // Block forward substitution in right-looking form on R = I: once D_kb is factored, row block kb is finished,
  //   Y[kb,j] = D_kb^-1 R[kb,j]  (phase A, j < kb),   and leaves the rows below,  R[i,j] -= L[i,kb] Y[kb,j]  (phase B).
  // In the transposed storage: YT[j,kb] = RT[j,kb] D_kb^-T and RT[j,i] -= YT[j,kb] L[i,kb]^T -- both A * B^T products
  // of column-major tiles, spread over the waves that are not on the critical path.
  auto y2 = [&](int i, int j) -> double * { return Y2 + ((i * (i - 1) / 2 + j) << 8); };
  auto phase_a = [&](int kb, int j) {
    double *tile = y2(kb, j);
    double xa[4], rb[4];
#pragma unroll
    for (int s4 = 0; s4 < 4; s4++) {
      xa[s4] = dinv(kb, l15, 4 * s4 + l4);
      rb[s4] = tile[((4 * s4 + l4) << 4) + l15];
    }
    d4 acc = (d4){0.0, 0.0, 0.0, 0.0}, acc1 = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s4 = 0; s4 < 4; s4 += 2) {
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[s4], rb[s4], acc, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[s4 + 1], rb[s4 + 1], acc1, 0, 0, 0);
    }
    acc += acc1;
#pragma unroll
    for (int r = 0; r < 4; r++) tile[((l4 + 4 * r) << 4) + l15] = acc[r];
  };
  auto phase_b = [&](int kb, int i, int j) {
    double *dst = y2(i, j);
    double la[4], yb[4];
    d4 acc = (d4){0.0, 0.0, 0.0, 0.0}, acc1 = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s4 = 0; s4 < 4; s4++) {
      la[s4] = -T[gpak_tix(i, kb, l15, 4 * s4 + l4)];
      yb[s4] = (j == kb) ? dinv(kb, 4 * s4 + l4, l15) : y2(kb, j)[((4 * s4 + l4) << 4) + l15];
    }
    if (j != kb) {   // (j == kb: first contribution to R[i,j], which starts at zero)
#pragma unroll
      for (int r = 0; r < 4; r++) acc[r] = dst[((l4 + 4 * r) << 4) + l15];
    }
#pragma unroll
    for (int s4 = 0; s4 < 4; s4 += 2) {
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(la[s4], yb[s4], acc, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(la[s4 + 1], yb[s4 + 1], acc1, 0, 0, 0);
    }
    acc += acc1;
#pragma unroll
    for (int r = 0; r < 4; r++) dst[((l4 + 4 * r) << 4) + l15] = acc[r];
  };

  // ---- results go out as soon as they are final (column kb of L and row kb of the inverse during step kb, by the waves
  // that are off the critical path), 16 B per lane, 128-B runs: the epilogue used to wait for 200 KB of stores to drain
  const int sp2 = 2 * (lane & 7), sc8 = lane >> 3;
  auto store_l_tile = [&](int rt, int ct) {
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const int col = sc8 + 8 * h;
      double2 v = *reinterpret_cast<const double2 *>(&T[gpak_tix(rt, ct, sp2, col)]);
      if (rt == ct) {                  // the diagonal tile's upper half holds leftovers: L goes out cleanly lower
        if (sp2 < col) v.x = 0.0;
        if (sp2 + 1 < col) v.y = 0.0;
      }
      *reinterpret_cast<double2 *>(A + 16 * rt + sp2 + (size_t)(16 * ct + col) * ld) = v;
    }
  };
  auto store_y_tile = [&](int i, int j) {   // tile (i, j), i > j: YT(a, b) = L^-1[16 i + b][16 j + a]
    const double *tile = y2(i, j);
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const int c8 = sc8 + 8 * h;
      *reinterpret_cast<double2 *>(invT + 16 * j + sp2 + (size_t)(16 * i + c8) * PB) =
          *reinterpret_cast<const double2 *>(&tile[(c8 << 4) + sp2]);
      *reinterpret_cast<double2 *>(inv + 16 * i + sp2 + (size_t)(16 * j + c8) * PB) =
          make_double2(tile[(sp2 << 4) + c8], tile[((sp2 + 1) << 4) + c8]);
    }
  };
  auto store_diag_inv = [&](int i) {
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const int c8 = sc8 + 8 * h;
      *reinterpret_cast<double2 *>(inv + 16 * i + sp2 + (size_t)(16 * i + c8) * PB) =
          make_double2(dinv(i, sp2, c8), dinv(i, sp2 + 1, c8));
      *reinterpret_cast<double2 *>(invT + 16 * i + sp2 + (size_t)(16 * i + c8) * PB) =
          make_double2(dinv(i, c8, sp2), dinv(i, c8, sp2 + 1));
    }
  };
  // store jobs of step kb: L tiles (i, kb), i = kb..7; inverse tiles (kb, j), j < kb; the diagonal inverse block kb
  auto store_jobs = [&](int kb, int first, int step) {
    const int njobs = (8 - kb) + kb + 1;
    for (int q = first; q < njobs; q += step) {
      if (q < 8 - kb) store_l_tile(kb + q, kb);
      else if (q < 8) store_y_tile(kb, q - (8 - kb));
      else store_diag_inv(kb);
    }
  };

  GPAK_TS(0);
  // ---- load: straight into the LDS image by LDS-DMA (global_load_lds_dwordx4), no register staging, everything in
  // flight at once.  One wave instruction moves 1 KiB = columns 8h..8h+7 of one 16x16 tile: lane l brings rows
  // 2(l&7), 2(l&7)+1 of column 8h + (l>>3); 72 such pieces.  Wave 0 brings only the two pieces of the first diagonal
  // tile and starts factoring it the moment they land; the other 70 pieces (10 per wave) arrive behind that.
  {
    typedef const __attribute__((address_space(1))) void *gptr_t;
    typedef __attribute__((address_space(3))) void *lptr_t;
    const int li = 2 * (lane & 7), lk = lane >> 3;
    auto piece = [&](int c) {
      const int slot = c >> 1, half = c & 1;
      int rt = 0, rem = slot;
      while (rem > rt) { rem -= rt + 1; rt++; }   // slot -> (rt, ct = rem), scalar
      const double *src = A + (16 * rt + li) + (size_t)(16 * rem + 8 * half + lk) * ld;
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)&T[(slot << 8) + (half << 7)], 16, 0, 0);
    };
    if (w == 0) {
      piece(0);
      piece(1);
#pragma unroll
      for (int q = 0; q < 4; q++) SM[ID0 + lane + 64 * q] = ((lane + 64 * q) >> 4) == (lane & 15) ? 1.0 : 0.0;
    } else {
#pragma unroll
      for (int p = 0; p < (70 + NW - 2) / (NW - 1); p++) {
        const int c = 2 + (w - 1) + (NW - 1) * p;
        if (c < 72) piece(c);
      }
    }
  }
  if (zero_inv && w >= 1) {
    // the caller's inverse buffers are not known to be zero outside the triangles written at the end (the context's
    // own are zeroed once in gpak_set_train)
    const double2 z = make_double2(0.0, 0.0);
#pragma unroll 4
    for (int it = 0; it < (8192 + 64 * (NW - 1) - 1) / (64 * (NW - 1)); it++) {
      const int e = (t - 64) + 64 * (NW - 1) * it, r = (e & 63) * 2, c = e >> 6;
      if (e < 8192) {
        if ((r >> 4) > (c >> 4)) *reinterpret_cast<double2 *>(invT + r + c * PB) = z;        // inv^T is upper
        else if ((r >> 4) < (c >> 4)) *reinterpret_cast<double2 *>(inv + r + c * PB) = z;   // inv is lower
      }
    }
  }

  // diagonal block kb: factor + invert (wave 0 only), on the VECTOR pipe.  Lane c < 16 owns column c of the symmetric
  // 16x16 block (16 registers), lanes 16..31 own the columns of an IDENTITY.  Pivot r: d = S[r][r] comes out of lane r
  // with v_readlane, every lane forms (sqrt d, d^-1/2) itself (v_rsq_f64 + a coupled Goldschmidt iteration, no
  // division), scales its row-r entry and applies x[i] -= u_i x[r] for i > r, u_i = U[r][i] read from lane i into an
  // SGPR pair.  The same row operations turn the identity into L^-1 (Gaussian elimination on [D | I]): the block's
  // inverse costs no extra instruction.  218 cycles per pivot (tools/mfma_latency.hip).
  // Round 2 first moved this onto the matrix pipe (rank-1 updates as 16x16x4 MFMAs, no cross-lane traffic): 430
  // cycles per pivot.  On gfx950 the f64 MFMA runs on the SAME FMA units as f64 VALU work -- a 16x16x4 f64 MFMA holds
  // them for 64 cycles and independent v_fma_f64 of the same wave do not overlap it (probe 8 of that tool: 65 + 8 x
  // 5.2 = 108 cycles, measured 108) -- so a K=4 MFMA that carries a K=1 update is 4x the work of the four v_fma_f64
  // it replaces, and the square roots cannot hide behind it.
  // upd >= 0: the block first receives the update of panel column `upd`, C -= P P^T with P = tile (kb, upd) -- the
  // "tile 0" of the trailing update, done by this wave so that the next block does not wait for the other waves
  auto diag_block = [&](int kb, int upd) {
    // opaque copies of the lane coordinates: without them every lane mask below is hoisted out of the block loop into
    // an SGPR pair of its own, ~60 pairs that spill to VGPR lanes and come back with two v_readlane + s_nop each --
    // more instructions than the compare they save
    int lane = lane_, l15 = l15_, l4 = l4_;
    asm volatile("" : "+v"(lane), "+v"(l15), "+v"(l4));
    double *tile = &T[gpak_tix(kb, kb, 0, 0)];
    {
      // the tile is made fully symmetric on the way (only its lower half is trusted: the upper half is whatever the
      // caller's upper triangle held), so the column read below is one conflict-free access per register
      d4 acc;
#pragma unroll
      for (int r4 = 0; r4 < 4; r4++) {
        const int k = l4 + 4 * r4;
        acc[r4] = tile[k <= l15 ? (k << 4) + l15 : (l15 << 4) + k];
      }
      if (upd >= 0) {
        double pa[4];
#pragma unroll
        for (int s4 = 0; s4 < 4; s4++) pa[s4] = T[gpak_tix(kb, upd, l15, 4 * s4 + l4)];
#pragma unroll
        for (int s4 = 0; s4 < 4; s4++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[s4], -pa[s4], acc, 0, 0, 0);
      }
#pragma unroll
      for (int r4 = 0; r4 < 4; r4++) tile[((l4 + 4 * r4) << 4) + l15] = acc[r4];
    }
    if (kb == 3) GPAK_TS(50);
    double x[16];
#pragma unroll
    for (int i = 0; i < 16; i++) {
      x[i] = SM[(l4 == 0 ? gpak_tix(kb, kb, 0, 0) : ID0) + (i << 4) + l15];   // S[l15][i] = S[i][l15], or the identity
    }
    int bad = 16;   // first non-positive pivot of this block (branch-free: the 16 pivots stay one basic block)
    if (kb == 3) GPAK_TS_PIN(x, 51);
#pragma unroll
    for (int r = 0; r < 16; r++) {
      if (kb == 3 && r == 8) GPAK_TS_PIN(x, 52);
      double dv = gpak_rdlane(x[r], r);
      const bool ok = dv > 0.0;
      bad = min(bad, ok ? 16 : r);
      dv = ok ? dv : 1.0;
      // coupled (Goldschmidt) iteration: g -> sqrt(d), h -> 0.5 / sqrt(d)
      const double y = __builtin_amdgcn_rsq(dv);
      double gg = dv * y, h = 0.5 * y;
      double e = fma(-h, gg, 0.5);
      gg = fma(gg, e, gg);
      h = fma(h, e, h);
      e = fma(-h, gg, 0.5);
      const double g = fma(gg, e, gg);
      h = fma(h, e, h);
      const double ri = h + h;
      double xr = x[r] * ri;
      xr = (lane == r) ? g : xr;
      x[r] = xr;
#pragma unroll
      for (int i = r + 1; i < 16; i++) {
        const double ui = gpak_rdlane(xr, i);   // U[r][i] = L[i][r]
        x[i] = fma(-ui, xr, x[i]);
      }
    }
    if (kb == 3) GPAK_TS_PIN(x, 53);
    if (bad < 16 && lane == 0) atomicMin(info, col0 + 16 * kb + bad + 1);
    // lane c < 16: x[r] = L[c][r] for r <= c (beyond that: leftovers, which land in the tile's unused upper half);
    // lanes 16 + c: x[r] = (L^-1)[r][c], exact zeros for r < c.  One store per register, no per-register lane masks
    if (l4 < 2) {
      double *dst = SM + (l4 == 0 ? gpak_tix(kb, kb, 0, 0) + l15 : WI0 + 272 * kb + 17 * l15);
      const int step = l4 == 0 ? 16 : 1;
#pragma unroll
      for (int r = 0; r < 16; r++) dst[r * step] = x[r];
    }
    if (kb == 3) GPAK_TS(54);
  };

  GPAK_TS(1);
  if (w == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's own two pieces: no barrier before the first block
    diag_block(0, -1);
  }
  GPAK_TS(2);
  __syncthreads();

  for (int kb = 0; kb < 7; kb++) {
    const int nt = 7 - kb;  // 16-row tiles below the diagonal block
    // ---- panel tiles: P := P * inv(D)^T  (a tile depends only on itself: no barrier inside); one tile per wave,
    // all operands requested before the first MFMA
    // seven 16x16x16 products: the 7 - kb panel tiles, then the kb phase-A products (row block kb of the inverse,
    // columns j < kb); one per wave with 8 waves, two rounds with 4
    for (int q = w; q < 7; q += NW) {
      if (q < 7 - kb) {
        const int rt = kb + 1 + q;
        double pa[4], xb[4];
#pragma unroll
        for (int s = 0; s < 4; s++) {
          xb[s] = dinv(kb, l15, 4 * s + l4);   // B[k][j] = inv(D)[j][k]
          pa[s] = T[gpak_tix(rt, kb, l15, 4 * s + l4)];
        }
        d4 acc = (d4){0.0, 0.0, 0.0, 0.0}, acc1 = (d4){0.0, 0.0, 0.0, 0.0};   // two chains of two
#pragma unroll
        // operand roles swapped: the accumulator holds P^T (row of P = lane&15), so its four registers go back to
        // LDS as 16 consecutive doubles per lane group instead of a 128-B stride (8-way bank conflict)
        for (int s = 0; s < 4; s += 2) {
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(xb[s], pa[s], acc, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(xb[s + 1], pa[s + 1], acc1, 0, 0, 0);
        }
        acc += acc1;
#pragma unroll
        for (int r = 0; r < 4; r++) T[gpak_tix(rt, kb, l15, l4 + 4 * r)] = acc[r];
      } else {
        phase_a(kb, q - (7 - kb));
      }
    }
    __syncthreads();
    GPAK_TS(3 + 4 * kb);
    // ---- trailing update of the lower tiles C(ti,tj) -= P_ti P_tj^T.  Tile 0 is the next
    // diagonal tile: wave 0 updates it and goes straight on to factor it; waves 1..3 share the
    // other tiles, four independent tiles in flight at a time.
    const int ntile = nt * (nt + 1) / 2;
    // up to four tiles per round, every LDS read requested before the first MFMA; NU is a compile-time count so that
    // a short last round issues no MFMA for tiles that do not exist (an f64 16x16x4 MFMA holds the SIMD's FMA units
    // for 64 cycles whether its result is wanted or not)
    auto trailing_round = [&](int qb, int qstep, auto nu_tag) {
      constexpr int NU = decltype(nu_tag)::value;
      d4 acc[NU];
      double pa[NU][4], pb[NU][4];
      int rti[NU], rtj[NU];
#pragma unroll
      for (int u = 0; u < NU; u++) {
        int ti = 0, rem = qb + u * qstep;
        while (rem > ti) { rem -= ti + 1; ti++; }  // q -> (ti, tj=rem), tj <= ti
        rti[u] = kb + 1 + ti; rtj[u] = kb + 1 + rem;
#pragma unroll
        for (int s = 0; s < 4; s++) {
          pa[u][s] = -T[gpak_tix(rti[u], kb, l15, 4 * s + l4)];
          pb[u][s] = T[gpak_tix(rtj[u], kb, l15, 4 * s + l4)];
        }
#pragma unroll
        for (int r = 0; r < 4; r++) acc[u][r] = T[gpak_tix(rti[u], rtj[u], l15, l4 + 4 * r)];   // C^T: conflict-free
      }
#pragma unroll
      for (int s = 0; s < 4; s++) {
#pragma unroll
        for (int u = 0; u < NU; u++) acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(pb[u][s], pa[u][s], acc[u], 0, 0, 0);
      }
#pragma unroll
      for (int u = 0; u < NU; u++) {
#pragma unroll
        for (int r = 0; r < 4; r++) T[gpak_tix(rti[u], rtj[u], l15, l4 + 4 * r)] = acc[u][r];
      }
    };
    auto trailing_tiles = [&](int q0, int qstep) {
      constexpr int NUMAX = NW == 8 ? 4 : 2;   // tiles in flight per wave (register budget of the 4-wave build)
      int qb = q0;
      for (; qb + (NUMAX - 1) * qstep < ntile; qb += NUMAX * qstep) trailing_round(qb, qstep, std::integral_constant<int, NUMAX>());
      const int left = qb < ntile ? (ntile - qb + qstep - 1) / qstep : 0;
      if constexpr (NUMAX == 4) {
        if (left == 3) trailing_round(qb, qstep, std::integral_constant<int, 3>());
        else if (left == 2) trailing_round(qb, qstep, std::integral_constant<int, 2>());
      }
      if (left == 1) trailing_round(qb, qstep, std::integral_constant<int, 1>());
    };
    if (w == 0) {
      GPAK_TS(4 + 4 * kb);
      diag_block(kb + 1, kb);         // tile 0 = (kb+1, kb+1): update and factor in registers
      GPAK_TS(5 + 4 * kb);
    } else if (NW == 8 && w == 4) {
      // wave 4 shares its SIMD (and that SIMD's FMA units) with wave 0: it gets the work that needs no MFMA
      store_jobs(kb, 0, 1);           // column kb of L and row kb of the inverse are final
    } else {
      constexpr int NH = NW == 8 ? 6 : 3;   // MFMA helper waves, on the SIMDs wave 0 is not on
      const int h = NW == 8 ? (w < 4 ? w - 1 : w - 2) : w - 1;
      if (kb == 0) GPAK_TSW(1, 44);
      trailing_tiles(h + 1, NH);      // tiles 1.. in steps of NH waves
      if (kb == 0) GPAK_TSW(1, 40);
      // phase B of the inverse: (7 - kb)(kb + 1) tile products, dealt round-robin to the same waves
      const int nprod = (7 - kb) * (kb + 1);
      for (int q = h; q < nprod; q += NH) phase_b(kb, kb + 1 + q / (kb + 1), q % (kb + 1));
      if (kb == 0) GPAK_TSW(1, 41);
      if (NW != 8) store_jobs(kb, h, NH);
    }
    __syncthreads();
    GPAK_TS(6 + 4 * kb);
  }
  GPAK_TS(32);

  GPAK_TS(33);
  // what is left: the last row block of the inverse, then column 7 of L and row 7 of the inverse
  for (int j = w; j < 7; j += NW) phase_a(7, j);
  __syncthreads();
  store_jobs(7, w, NW);
  GPAK_TS(34);
  GPAK_TS_FLUSH();
}

__global__ __launch_bounds__(512) void gpak_potrf128_f64(double *A, long ld, double *__restrict__ inv, int col0, int *info,
                                                          int zero_inv) {
  gpak_potrf128_body<8>(A, ld, inv, col0, info, zero_inv);
}
__global__ __launch_bounds__(256, 6)
void gpak_potrf128_co_f64(double *A, long ld, double *__restrict__ inv, int col0, int *info, int zero_inv) {
  gpak_potrf128_body<4>(A, ld, inv, col0, info, zero_inv);
}

void gpak_launch_potrf128(hipStream_t st, double *A, long ld, double *inv, int col0, int *info, bool zero_inv, bool co,
                          int co_mode) {
  // 132 KiB of dynamic LDS needs the opt-in, once per kernel AND per device (a thread group drives several devices
  // from one process)
  {
    static std::atomic<unsigned long long> done_mask{0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(done_mask.load(std::memory_order_acquire) & bit)) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(gpak_potrf128_f64), hipFuncAttributeMaxDynamicSharedMemorySize,
                                GPAK_POTRF_LDS_BYTES);
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(gpak_potrf128_co_f64),
                                hipFuncAttributeMaxDynamicSharedMemorySize, GPAK_POTRF_LDS_BYTES);
      done_mask.fetch_or(bit, std::memory_order_release);
    }
  }
  // co_mode (GpakTuning::potrf_co): 0 = always the 8-wave build, 2 = always the co-resident 4-wave build, 1 = as the
  // caller asks; -1 = the process-wide setting
  if (co_mode < 0) co_mode = gpak_tuning().potrf_co;
  if (co_mode != 1) co = co_mode == 2;
  if (co) hipLaunchKernelGGL(gpak_potrf128_co_f64, dim3(1), dim3(256), GPAK_POTRF_LDS_BYTES, st, A, ld, inv, col0, info, zero_inv ? 1 : 0);
  else hipLaunchKernelGGL(gpak_potrf128_f64, dim3(1), dim3(512), GPAK_POTRF_LDS_BYTES, st, A, ld, inv, col0, info, zero_inv ? 1 : 0);
}

// Panel factorisation of one outer block column [J, J+W): all rows below it.
// M is addressed with GLOBAL (row, column) indices; only columns [J, J+W) are touched, so a
// rank that stores just this block column passes a virtual base (see dev_api.hip).
// One level: 128-column steps, each followed by the K=128 update of the columns [j+128, J+W).
// `gate`: an event the FIRST in-panel update waits for (the update of this panel's columns [J+128, J+W) by the previous
// panel, when that ran on a side stream: GPAK_NEXT_SPLIT_ROWS)
static void factor_panel_128(hipStream_t st, double *M, long ld, int Np, int J, int W, double *inv_base, int *info,
                             bool zero_inv, bool co, int co_mode, hipEvent_t gate = nullptr) {
  for (int j = J; j < J + W; j += PB) {
    double *inv = inv_base + (size_t)(j / PB) * 2 * PB * PB;
    gpak_launch_potrf128(st, M + j + (size_t)j * ld, ld, inv, j, info, zero_inv, co, co_mode);
    const int mt = (Np - j - PB) / PB;
    if (mt > 0) {
      double *P = M + (j + PB) + (size_t)j * ld;
      gpak_launch_gemm_nt(st, mt, 1, PB, 1.0, P, ld, inv, PB, 0.0, P, ld, 0, 0, false, false);
      const int nct = (J + W - (j + PB)) / PB;
      if (gate && j == J) hipStreamWaitEvent(st, gate, 0);
      if (nct > 0)
        gpak_launch_gemm_nt(st, mt, nct, PB, -1.0, P, ld, P, ld, 1.0, M + (j + PB) + (size_t)(j + PB) * ld,
                            ld, 0, 0, true, false);
    } else if (gate && j == J) {
      hipStreamWaitEvent(st, gate, 0);
    }
  }
}
// Panels wider than GPAK_PANEL_MID columns are factored in GPAK_PANEL_MID-column pieces with a K=MID update
// of the rest of the panel in between (three-level blocking: 128 / MID / W).
#define GPAK_PANEL_MID 512
void gpak_factor_panel(hipStream_t st, double *M, long ld, int Np, int J, int W, double *inv_base, int *info,
                       bool zero_inv, bool co, int co_mode, hipEvent_t gate) {
  for (int j = J; j < J + W; j += GPAK_PANEL_MID) {
    const int w = (J + W - j) < GPAK_PANEL_MID ? (J + W - j) : GPAK_PANEL_MID;
    factor_panel_128(st, M, ld, Np, j, w, inv_base, info, zero_inv, co, co_mode, j == J ? gate : nullptr);
    const int c0 = j + w, nct = (J + W - c0) / PB, mt = (Np - c0) / PB;
    if (nct > 0 && mt > 0) {
      const double *P = M + c0 + (size_t)j * ld;
      gpak_launch_gemm_nt(st, mt, nct, w, -1.0, P, ld, P, ld, 1.0, M + c0 + (size_t)c0 * ld, ld, 0, 0, true, false);
    }
  }
}
static void factor_panel(gpak_ctx *ctx, hipStream_t st, int J, int W, bool co, hipEvent_t gate = nullptr) {
  // ctx->dInv is zeroed once in gpak_set_train and only ever written inside its triangles
  gpak_factor_panel(st, ctx->dM, ctx->ld, ctx->Np, J, W, ctx->dInv, ctx->dInfo, false, co, ctx->tune.potrf_co, gate);
}

// Chain-bound tail: the same panel factorisation (W <= 512), but the update of the NEXT block column [J1, J2) is
// applied sub-panel by sub-panel (K = 128) on a side stream `sx` as soon as each 128-column sub-panel is solved,
// instead of as one K = W product on the panel stream after the whole panel: the next panel's first potrf128 then
// waits for one K = 128 product (~15 us) instead of a K = 512 one (~52 us).  `first_wait` (the previous bulk update,
// which touched [J1, J2)) is waited for once on sx; `done` is recorded on sx behind the last sub-update.
static int factor_panel_tail(gpak_ctx *ctx, hipStream_t sp, hipStream_t sx, int J, int W, int J1, int J2,
                             hipEvent_t first_wait, hipEvent_t *evs, hipEvent_t done) {
  const long ld = ctx->ld;
  double *M = ctx->dM;
  const int Np = ctx->Np;
  if (first_wait) GPAK_HIP(hipStreamWaitEvent(sx, first_wait, 0));
  int k = 0;
  for (int j = J; j < J + W; j += PB, k++) {
    double *inv = ctx->dInv + (size_t)(j / PB) * 2 * PB * PB;
    gpak_launch_potrf128(sp, M + j + (size_t)j * ld, ld, inv, j, ctx->dInfo, false, false, ctx->tune.potrf_co);
    const int mt = (Np - j - PB) / PB;
    if (mt <= 0) continue;
    double *P = M + (j + PB) + (size_t)j * ld;
    gpak_launch_gemm_nt(sp, mt, 1, PB, 1.0, P, ld, inv, PB, 0.0, P, ld, 0, 0, false, false);
    GPAK_HIP(hipEventRecord(evs[k], sp));
    const int nct = (J + W - (j + PB)) / PB;
    if (nct > 0)
      gpak_launch_gemm_nt(sp, mt, nct, PB, -1.0, P, ld, P, ld, 1.0, M + (j + PB) + (size_t)(j + PB) * ld, ld, 0, 0, true,
                          false);
    // next block column: rows >= J1 of sub-panel j against its rows [J1, J2)
    const int mt2 = (Np - J1) / PB, nt2 = (J2 - J1) / PB;
    if (mt2 > 0 && nt2 > 0) {
      GPAK_HIP(hipStreamWaitEvent(sx, evs[k], 0));
      const double *P1 = M + J1 + (size_t)j * ld;
      gpak_launch_gemm_nt(sx, mt2, nt2, PB, -1.0, P1, ld, P1, ld, 1.0, M + J1 + (size_t)J1 * ld, ld, 0, 0, true, false);
    }
  }
  GPAK_HIP(hipEventRecord(done, sx));
  return GPAK_OK;
}

// Update of the columns [c0, c1) (and all rows >= c0) with the factored panel [J, J+W).
static void update_cols(gpak_ctx *ctx, hipStream_t st, int J, int W, int c0, int c1, bool trailing) {
  const long ld = ctx->ld;
  double *M = ctx->dM;
  const int mt = (ctx->Np - c0) / PB, nt = (c1 - c0) / PB;
  if (mt <= 0 || nt <= 0) return;
  const double *P = M + c0 + (size_t)J * ld;
  gpak_launch_gemm_nt(st, mt, nt, W, -1.0, P, ld, P, ld, 1.0, M + c0 + (size_t)c0 * ld, ld, 0, 0, true,
                      trailing);
}

// Right-looking blocked factorisation with one panel of look-ahead:
//   panel stream (high priority):  F(0) | T(0,1) F(1) | T(1,2) F(2) | ...
//   update stream (ctx->stream)  :        T(0,2..)    | T(1,3..)    | ...
// F(b) = factor_panel of outer block b, T(b,c) = update of block column c with panel b.
// T(b,b+1) waits for the previous bulk update (which touched column b+1); the bulk update
// T(b,b+2..) waits for F(b).  While the MFMA-bound bulk update of step b runs, the latency-
// bound panel work of step b+1 proceeds beside it.
int gpak_potrf_blocked(gpak_ctx *ctx) {
  const int Np = ctx->Np;
  // without look-ahead everything is queued on the one main stream, in the classical order
  hipStream_t su = ctx->stream, sp = ctx->lookahead ? ctx->stream_hi : ctx->stream;
  hipStream_t sf = ctx->lookahead ? ctx->stream_fs : ctx->stream;
  int NB = ctx->nb_outer;
  if (NB < PB) NB = PB;
  NB = NB / PB * PB;
  // Panel boundaries.  While the trailing matrix is large the bulk update hides any panel chain, and a wider
  // panel makes it more efficient (K = 1024: 73.7 TFLOP/s in the kernel, K = 512: 71.4); later the narrower
  // panel keeps the chain short (N=32768: 180.8 -> 178.5 ms).  GPAK_NB_WIDE / GPAK_NB_WIDE_ROWS: width and
  // "rows left" threshold; only applies when nb_outer is narrower than the wide width.
  const int nb_wide = ctx->tune.nb_wide / PB * PB;   // 0: off
  const int nb_wide_rows = ctx->tune.nb_wide_rows;
  // a third tier for trailing matrices beyond N = 32768: 2048-column panels while more than 32768 rows are left
  // (N=65536: 1295.0 -> 1285.5 ms with 2048 / 32768 alone, tools/time_sizes.py; nothing changes at N <= 32768)
  const int nb_xwide = ctx->tune.nb_xwide / PB * PB, nb_xwide_rows = ctx->tune.nb_xwide_rows;
  std::vector<int> Js;
  // the very first panel has nothing to hide behind: keep it narrow so that the first bulk update starts early
  // (measured at N=32768, 30-step A/B inside one box: 182.46 -> 181.90 ms)
  const bool first_narrow = ctx->tune.first_narrow;
  for (int J = 0; J < Np;) {
    Js.push_back(J);
    const bool first = first_narrow && J == 0;
    J += (nb_xwide > NB && nb_xwide > nb_wide && Np - J > nb_xwide_rows && !first) ? nb_xwide
         : (nb_wide > NB && Np - J > nb_wide_rows && !first)                        ? nb_wide
                                                                                    : NB;
  }
  const int nJ = (int)Js.size();
  Js.push_back(Np);
  const int init = 0x7fffffff;
  GPAK_HIP(hipMemcpyAsync(ctx->dInfo, &init, sizeof(int), hipMemcpyHostToDevice, su));

  while ((int)ctx->ev_sync.size() < 2 * nJ + 4 + 5) {
    hipEvent_t e;
    GPAK_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    ctx->ev_sync.push_back(e);
  }
  hipEvent_t *EF = ctx->ev_sync.data(), *EU = ctx->ev_sync.data() + nJ;
  hipEvent_t Estart = ctx->ev_sync[2 * nJ], Eend = ctx->ev_sync[2 * nJ + 1], Efs = ctx->ev_sync[2 * nJ + 2];
  hipEvent_t Eorder = ctx->ev_sync[2 * nJ + 3];
  hipEvent_t *EX = ctx->ev_sync.data() + 2 * nJ + 4, EXdone = ctx->ev_sync[2 * nJ + 8];
  // the panel stream starts after everything queued so far on the main stream (the fill)
  GPAK_HIP(hipEventRecord(Estart, su));
  GPAK_HIP(hipStreamWaitEvent(sp, Estart, 0));

  size_t ev_used = 0;
  int done512 = 0;
  double tflops = 0.0, tbytes = 0.0;
  int tl = 0;
  const int tail_rows = ctx->tune.tail_rows;
  // sub-panel updates of the next block column in the tail (factor_panel_tail): measured 183.07 -> 183.73 ms at
  // N=32768 -- the four K=128 products re-read and re-write the column four times and the chain gains nothing
  // measurable; off unless GPAK_SUB_NEXT=1 (kept: the multi-GPU schedule is built the same way and tests compare)
  const bool sub_next = ctx->tune.sub_next;
  bool next_col_done = false;   // the next block column already has this panel's update (applied per sub-panel)
  // GPAK_NEXT_SPLIT_ROWS: in the chain-bound tail only the first 128 columns of the next block column take this panel's
  // K = W update on the panel stream; the others get it on the side stream while the next panel's first block kernel
  // and panel solve run, and that panel's first in-panel update waits for it (`gate`)
  const int next_split_rows = ctx->stream_x && ctx->lookahead ? ctx->tune.next_split_rows : 0;
  hipEvent_t gate = nullptr;
  for (int b = 0; b < nJ; b++) {
    const int J = Js[b], W = Js[b + 1] - J;
    const int J1n = J + W, J2n = J1n < Np ? Js[b + 2] : Np;
    const bool tail_step = sub_next && ctx->lookahead && ctx->stream_x && W <= 512 && J1n < Np && Np - J1n <= tail_rows;
    if (tail_step) {
      if (gate) GPAK_HIP(hipStreamWaitEvent(sp, gate, 0));
      int rc = factor_panel_tail(ctx, sp, ctx->stream_x, J, W, J1n, J2n, b > 0 ? EU[b - 1] : nullptr, EX, EXdone);
      if (rc) return rc;
      next_col_done = true;
    } else {
      // panel b is factored while the bulk update of panel b-1 (rows >= J + W) runs: on the unmasked stream that
      // update holds two 210-VGPR waves on every SIMD of the chip, and only the 4-wave, 80-VGPR potrf128 fits beside it
      const bool beside_bulk = ctx->lookahead && b > 0 && !(ctx->stream_tail && Np - (J + W) <= tail_rows);
      factor_panel(ctx, sp, J, W, beside_bulk, gate);
      next_col_done = false;
    }
    GPAK_HIP(hipEventRecord(EF[b], sp));
    // forward substitution of the right-hand side y/sn2 rides along: block column b of L is final
    // here and the solve touches only the two work vectors.  It has its own stream so that its four
    // small launches (~45 us) are not part of the serial panel chain
    if (ctx->fwd_in_factor) {
      if (sf != sp) GPAK_HIP(hipStreamWaitEvent(sf, EF[b], 0));
      gpak_launch_trsv_fwd_block(sf, Np, J, W, ctx->dM, ctx->ld, ctx->dInv, ctx->dWork, ctx->dWork + Np);
      // explicit inverses of the 512-column diagonal blocks completed by this panel, for the back substitution
      // (same stream: off the panel chain, hidden behind the bulk updates)
      const bool inv512 = ctx->tune.inv512;
      const int bw = ctx->bwd_bw;
      while (inv512 && done512 * bw < Np && (std::min(Np, (done512 + 1) * bw) <= J + W)) {
        const int j5 = done512 * bw;
        const int w5 = std::min(bw, Np - j5);
        // bwd_fused = 2: with the block that couples it to the block column on its left riding along (solve.hip)
        if (ctx->tune.bwd_fused == 2)
          gpak_launch_diag_inverse(sf, j5, w5, ctx->dM, ctx->ld, ctx->dInv, ctx->dT512 + (size_t)done512 * 2 * bw * bw, 2 * bw,
                                   j5 > 0 ? bw : 0);
        else
          gpak_launch_diag_inverse(sf, j5, w5, ctx->dM, ctx->ld, ctx->dInv, ctx->dInv512 + (size_t)done512 * bw * bw, bw);
        done512++;
      }
    }
    const int J1 = J + W;
    if (J1 >= Np) break;
    const int J2 = Js[b + 2];
    // next panel's columns first, on the panel stream (after the previous bulk update)
    gate = nullptr;
    if (next_col_done) {
      GPAK_HIP(hipStreamWaitEvent(sp, EXdone, 0));
    } else if (next_split_rows > 0 && Np - J1 <= next_split_rows && J2 - J1 > PB && J2 - J1 <= GPAK_PANEL_MID) {
      hipStream_t sx = ctx->stream_x;
      GPAK_HIP(hipStreamWaitEvent(sx, EF[b], 0));
      if (b > 0) {
        GPAK_HIP(hipStreamWaitEvent(sx, EU[b - 1], 0));
        GPAK_HIP(hipStreamWaitEvent(sp, EU[b - 1], 0));
      }
      update_cols(ctx, sp, J, W, J1, J1 + PB, false);
      update_cols(ctx, sx, J, W, J1 + PB, J2, false);
      GPAK_HIP(hipEventRecord(EXdone, sx));
      gate = EXdone;
    } else {
      if (b > 0) GPAK_HIP(hipStreamWaitEvent(sp, EU[b - 1], 0));
      update_cols(ctx, sp, J, W, J1, J2, false);
    }
    if (J2 < Np) {
      // chain-bound tail: the bulk update is off the critical path there; on the CU-masked stream it leaves
      // idle compute units to potrf128 and the small panel products
      hipStream_t su_b = (ctx->stream_tail && ctx->lookahead && Np - J2 <= tail_rows) ? ctx->stream_tail
                         : (ctx->stream_bulk && ctx->lookahead) ? ctx->stream_bulk : ctx->stream;
      if (su_b != su) {                       // keep the order of successive bulk updates across the two streams
        GPAK_HIP(hipEventRecord(Eorder, su));
        GPAK_HIP(hipStreamWaitEvent(su_b, Eorder, 0));
        su = su_b;
      }
      GPAK_HIP(hipStreamWaitEvent(su, EF[b], 0));
      if (ctx->profile) {
        while (ctx->ev_pool.size() < ev_used + 2) {
          hipEvent_t e;
          GPAK_HIP(hipEventCreate(&e));
          ctx->ev_pool.push_back(e);
        }
        GPAK_HIP(hipEventRecord(ctx->ev_pool[ev_used], su));
      }
      update_cols(ctx, su, J, W, J2, Np, true);
      if (ctx->profile) {
        GPAK_HIP(hipEventRecord(ctx->ev_pool[ev_used + 1], su));
        ev_used += 2;
      }
      const double mt = (Np - J2) / PB;
      tflops += mt * (mt + 1) / 2.0 * 2.0 * PB * PB * W;  // lower tiles only
      tbytes += mt * (mt + 1) / 2.0 * 2.0 * PB * PB * 8.0; // each C tile read once and written once
      tl++;
    }
    GPAK_HIP(hipEventRecord(EU[b], su));
  }
  int info = 0;
  GPAK_HIP(hipEventRecord(Eend, sp));
  GPAK_HIP(hipStreamWaitEvent(su, Eend, 0));
  if (sf != su) {
    GPAK_HIP(hipEventRecord(Efs, sf));
    GPAK_HIP(hipStreamWaitEvent(su, Efs, 0));
  }
  GPAK_HIP(hipMemcpyAsync(&info, ctx->dInfo, sizeof(int), hipMemcpyDeviceToHost, su));
  GPAK_HIP(hipStreamSynchronize(su));
  ctx->times.trailing_flops = tflops;
  ctx->times.trailing_bytes = tbytes;
  ctx->times.trailing_launches = tl;
  ctx->times.trailing_ms = 0.0;
  if (ctx->profile) {
    double ms = 0.0;
    for (size_t e = 0; e < ev_used; e += 2) {
      float m = 0;
      GPAK_HIP(hipEventElapsedTime(&m, ctx->ev_pool[e], ctx->ev_pool[e + 1]));
      ms += m;
    }
    ctx->times.trailing_ms = ms;
  }
  if (info != init) {
    // padded columns (>= N) are identity and cannot fail
    ctx->failed_col = info;
    return GPAK_ENOTPD;
  }
  ctx->failed_col = 0;
  return GPAK_OK;
}
