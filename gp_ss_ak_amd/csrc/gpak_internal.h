// gpak_internal.h -- shared declarations of libgpak_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "../../include/gpak.h"

#define GPAK_TILE 128  // every matrix dimension on the device is padded to this

// The covariance function as the device sees it: a SUM of up to three stationary terms (the
// children of the reference's HybKerns, Kernel.cpp:140-154), a constant and a white-noise diagonal.
//   term t:  k_t(x, x') = var2 * profile( D2_t ),  D2_t = |(x - x') A_t|^2
//   profile 0: exp(-sqrt(D2))        Kern_ExpAnisotropic (A = Rot diag(L) Rot^T, Kernel.cpp:1399-1425)
//                                    Kern_Exponential    (A = I / Hayper_Euc_Exp, Kernel.cpp:1343-1368, 1437-1441)
//   profile 1: exp(-0.5 * iw * D2)   Kern_RBF            (A = I / Hayper_Euc_RBF, Kernel.cpp:482-488)
#define GPAK_MAX_TERMS 3
#define GPAK_PROFILE_EXPSQRT 0
#define GPAK_PROFILE_RBF 1
struct KernTerm {
  double A[9];   // column-major 3x3 metric factor
  double a33;    // factor of a 4th input column (rock type): InversewidthR for ExpAns
                 // (Kernel.cpp:1411-1424, A(3,3) = L_r), the isotropic scale for Exp / RBF (EuclDist)
  double var2;   // Sigma^2
  double iw;     // inverseWidth_RBF (profile 1 only)
  int profile;
};
struct KernParams {
  KernTerm term[GPAK_MAX_TERMS];
  int nterms;
  double mu[4];  // pooled mean used for centring (Kernel.cpp:1391-1397)
  int d;         // input columns: 3, or 4 (x, y, z, rock type)
  double bias;   // Kern_Bias Sigma_Bias (Kernel.cpp:366), added to every entry
  double white;  // Kern_White Sigma_White (Kernel.cpp:256-263), added where i == j of the same set
  int mode;      // GPAK_DIST_*
};

// A transformed point set on the device: for each term u = (x - mu) A_t, SoA, plus |u|^2:
// array c (0..2 = u0,u1,u2; 3 = |u|^2; 4 = u3, the transformed 4th column, zero for 3-D inputs) of
// term t is base + (GPAK_PT * t + c) * cap.
#define GPAK_PT 5
#define GPAK_PARR(base, cap, t, c) ((base) + (size_t)(GPAK_PT * (t) + (c)) * (cap))
struct DevPoints {
  double *base = nullptr;
  int n = 0;    // valid points
  int cap = 0;  // allocated points (multiple of GPAK_TILE)
};

// ---- tuning ------------------------------------------------------------------------------------------------
// Every schedule / kernel-selection knob of the library in ONE place.  gpak_tuning() is the process-wide set: the
// defaults below (each the measured best at N = 32768 on MI355X, DESIGN.md sections 4.1-4.4), overridden ONCE, at the
// first call, by the GPAK_* environment variables of the same names (A/B tooling; gpak_reload_tuning() re-reads
// them).  A context copies the set at gpak_create; gpak_set_option changes the copy of that context only.
struct GpakTuning {
  // gpak_potrf_blocked
  int nb_outer = 512;          // GPAK_NB_OUTER      outer panel width
  int nb_wide = 1024;          // GPAK_NB_WIDE       panel width while more than nb_wide_rows rows are left (0: off)
  int nb_wide_rows = 16384;    // GPAK_NB_WIDE_ROWS
  int nb_xwide = 2048;         // GPAK_NB_XWIDE      ... and while more than nb_xwide_rows rows are left (0: off)
  int nb_xwide_rows = 32768;   // GPAK_NB_XWIDE_ROWS
  bool first_narrow = true;    // GPAK_FIRST_NARROW  the very first panel is nb_outer wide
  int tail_rows = 12288;       // GPAK_TAIL_ROWS     rows left from which the bulk updates use the CU-masked queue
  bool sub_next = false;       // GPAK_SUB_NEXT      tail: next block column updated sub-panel by sub-panel
  int next_split_rows = 0;     // GPAK_NEXT_SPLIT_ROWS  rows left from which only the first 128 columns of the next block column are
                               //                    updated in the panel chain, the others beside the next panel's first step (0: off)
  bool inv512 = true;          // GPAK_INV512        explicit diagonal-block inverses for the back substitution
  int bwd_fused = 2;           // GPAK_BWD_FUSED     back substitution: 0 three launches per step, 1 far column dots under the diagonal
                               //                    step (two launches), 2 one launch (coupling blocks T_b): solve.hip
  int bwd_block = 512;         // GPAK_BWD_BLOCK     ... of this width: 512, 1024 or 2048 columns per back-substitution step
                               //                    (measured round 3: solve 1.72 / 1.42 / 1.33 ms at N = 32768, but the wider inverses cost
                               //                    the factorisation as much or more: profiles/r03_bwd_block.txt)
  bool lookahead = true;       // GPAK_LOOKAHEAD     0: everything on one stream
  bool fwd_in_factor = true;   // GPAK_FWD_IN_FACTOR forward substitution of y/sn2 rides along with the factorisation
  int potrf_co = 1;            // GPAK_POTRF_CO      0 always the 8-wave block kernel, 2 always the 4-wave one, 1 as asked
  int tail_mask = 8;           // GPAK_TAIL_MASK     compute units the tail's bulk queue leaves idle (0: no such queue)
  int tail_mask_stride = 1;    // GPAK_TAIL_MASK_STRIDE
  bool bulk_queue = true;      // GPAK_BULK_QUEUE    bulk updates on a queue made by hipExtStreamCreateWithCUMask
  long ld_pad = -1;            // GPAK_LD_PAD        leading-dimension skew in doubles (-1: 32 from Np = 1024 on)
  // kernel selection
  int sbase_rows = 0;          // GPAK_SBASE_ROWS      bulk updates of more rows than this use the scalar-base build of the kernel (0: never;
                               //                      measured slower in situ at every threshold: profiles/r03_scalar_base.txt)
  int gemm_small = 160;        // GPAK_GEMM_SMALL      tile grids up to this size take the latency kernel
  int gemm_small_rows = 16;    // GPAK_GEMM_SMALL_ROWS rows per workgroup of that kernel: 16 / 32 / 64
  int super_lr = 3;            // GPAK_SUPER_LR      bulk update: super-tiles of 2^lr x 2^(6-lr) tiles per XCD (3 = 8 x 8)
  bool fill_fast = true;       // GPAK_FILL_FAST     table exp + in-line sqrt fill / Gram-matvec
  bool kmv_sym = true;         // GPAK_KMV_SYM       symmetric Gram-matvec from 32 macro blocks on
  // fp32 prediction (GPAK_F32 contexts)
  bool f32_wide = true;        // GPAK_F32_ACC=plain switches the fp64 accumulation of the fp32 products off
  int f32_rsd = 4;             // GPAK_F32_RSD       operand prefetch depth of the wide-accumulation kernel: 2 / 4 / 8
  int f32_tile = 128;          // GPAK_F32_TILE      wave tile rows of the plain fp32 kernel: 128 / 64
  int fs_levels[8] = {128, 512, 2048, 8192, 0, 0, 0, 0};   // GPAK_FS_LEVELS_F32  ladder of the substitution with many right-hand sides (fp32 and fp64 prediction)
  int pred_batch = 0;          // GPAK_PRED_BATCH    test points per batch (0: 16384 fp64, 65536 fp32)
  int pred_ld_skew = 1;        // GPAK_PRED_LD_SKEW  leading dimensions of the test-major batch and of the fp32 factor image
                               //                    are skewed by this many 256-byte units (0: powers of two, as in round 2)
};
const GpakTuning &gpak_tuning();
int gpak_build_kp(int nterms, const int *kinds, const double *pars, double bias, double white, int dist_mode,
                  KernParams *out, double *kdiag_out);

struct gpak_multi;   // multi.hip: one process driving several GPUs (gpak_create_multi)

struct gpak_ctx {
  gpak_multi *multi = nullptr;   // set: every call is forwarded to the group, the fields below are unused
  int device = 0;
  int precision = GPAK_F64;
  hipStream_t stream = nullptr;     // main stream: fill, bulk trailing updates, solves
  hipStream_t stream_hi = nullptr;  // high-priority stream: panel factorisation (look-ahead)
  hipStream_t stream_tail = nullptr;  // CU-masked copy of the main stream for the tail's bulk updates (optional)
  hipStream_t stream_bulk = nullptr;  // the bulk updates' own queue before the tail: created like stream_tail but with every CU enabled (optional)
  hipStream_t stream_fs = nullptr;  // forward substitution riding along with the factorisation, off the panel chain
  hipStream_t stream_x = nullptr;   // tail: K=128 updates of the next block column, sub-panel by sub-panel
  std::string err;

  // training set
  int N = 0, Np = 0, ld = 0, d = 0;
  double *dX = nullptr;      // raw input columns (3 or 4), SoA with stride Np (4 arrays, the 4th zero for d = 3)
  double *dy = nullptr;      // Np (padded with 0)
  std::vector<double> hX;    // host copy of X (col-major N x d), for pooled means
  double xsum[4] = {0, 0, 0, 0};

  // parameters
  bool have_params = false;
  double expans[8] = {0};   // ExpAns parameters when the kernel is the reference's default composition
  bool expans_only = true;  // kernel == ExpAns(+Bias): the composition of the fixed-length gpak_grad
  int kinds[GPAK_MAX_TERMS] = {0, 0, 0};  // GPAK_KERN_* of each stationary term
  double bias = 0, sn2 = 0;
  int dist_mode = GPAK_DIST_DIRECT;
  KernParams kp;
  double kdiag = 0;          // diag_Compute of the composition (Kernel.cpp:127-136)

  // device state
  DevPoints U;               // transformed training points
  double *dM = nullptr;      // Np x ld matrix buffer: B then L (lower)
  double *dInv = nullptr;    // (Np/128) inverted 128x128 diagonal blocks of L
  double *dInv512 = nullptr; // explicit (L_bb^-1)^T of the bw x bw diagonal blocks (bw = bwd_bw; back substitution)
  int bwd_bw = 512;          // block width dInv512 was sized and is being built for
  double *dT512 = nullptr;   // bwd_fused = 2: per block column the stacked [R_b ; T_b], T_b = L[b, b-1]^T R_b (2 bw x bw: solve.hip)
  bool inv512_ok = false;
  int t512_mode = 0;         // fused back substitution of the current factor: 0 no, 1 from dInv512, 2 from dT512
  double *dAlpha = nullptr;  // Np
  double *dWork = nullptr;   // 4*Np scratch vectors
  double *dRed = nullptr;    // small reduction scratch
  int *dInfo = nullptr;      // first failing column (1-based) or 0
  enum { M_NONE, M_B, M_L } mstate = M_NONE;
  bool alpha_ok = false, nlz_ok = false;
  int failed_col = 0;
  double quad = 0, sumlp = 0, logdet = 0, nlz = 0;

  // prediction buffers (grown on demand, kept across calls)
  DevPoints Upred, Tq;       // train / test-batch points centred on the pooled train+test mean
  double *dXte = nullptr;    // 4 x pred_cap raw test columns (SoA)
  double *dWt = nullptr;     // pred_cap x Np cross-kernel, test-major (transposed kX)
  double *dPv = nullptr;     // 2 x pred_cap: mean, sum of squares
  double *dPart = nullptr;   // 64 x pred_cap partial sums
  int pred_cap = 0;
  size_t wt_elems = 0;
  // fp32 prediction (ctx created with GPAK_F32): fp32 images of L and of the inverse blocks
  float *dLf = nullptr, *dInvf = nullptr;
  long ldLf = 0;             // leading dimension of dLf (Np + skew)
  bool lf_ok = false;

  // gradient buffers (allocated on the first gpak_grad)
  double *dG = nullptr;      // Np x ld: L^-T (upper triangular)
  double *dBinv = nullptr;   // Np x ld: B^-1, lower tiles
  double *dF = nullptr;      // Np: f = K*alpha of the last logLikelihood()
  double *dGpart = nullptr;  // per-workgroup partial sums of the pair pass
  size_t gpart_elems = 0;

  // options
  GpakTuning tune;            // this context's copy of the tuning set (gpak_set_option changes it)
  bool memoise = false;
  int nb_outer = 512;
  bool profile = false;
  bool lookahead = true;
  bool fwd_in_factor = true;  // L^-1 (y/sn2) is computed block column by block column during the factorisation
  bool z_ok = false;          // dWork[Np..2Np) holds L^-1 (y/sn2) of the current factor

  // timing
  gpak_phase_times times;
  hipEvent_t ev[10];
  std::vector<hipEvent_t> ev_pool;   // timing events around the trailing updates
  std::vector<hipEvent_t> ev_sync;   // cross-stream dependencies of the look-ahead pipeline
};

#define GPAK_HIP(call)                                                                  \
  do {                                                                                  \
    hipError_t e_ = (call);                                                             \
    if (e_ != hipSuccess) {                                                             \
      ctx->err = std::string(#call) + ": " + hipGetErrorString(e_);                     \
      return GPAK_EHIP;                                                                 \
    }                                                                                   \
  } while (0)

// ---- device math for the covariance profile ------------------------------------------------
// exp(-sqrt(D2)) costs ~100 instruction slots per matrix element through the device libm (the fill is then
// ALU-bound: 1.57 ms against 0.85 ms for the stores alone); these two cost ~30 and stay within 2 ulp.
#ifdef __HIPCC__
// sqrt(d), d >= 0 finite: v_rsq_f64 + two Goldschmidt steps + one residual correction
__device__ __forceinline__ double gpak_sqrt_nonneg(double d) {
  const double y = __builtin_amdgcn_rsq(d);
  double g = d * y, h = 0.5 * y;
  double r = fma(-h, g, 0.5);
  g = fma(g, r, g);
  h = fma(h, r, h);
  r = fma(-h, g, 0.5);
  g = fma(g, r, g);
  h = fma(h, r, h);
  g = fma(fma(-g, g, d), h, g);
  return d > 0.0 ? g : d;    // d == 0 (rsq(0) = inf) gives 0; a NaN coordinate stays NaN
}
// exp(x), x <= 0 finite: x = n ln2 + r, |r| <= 0.35, Taylor polynomial of degree 13 (remainder 4e-18), ldexp
__device__ __forceinline__ double gpak_exp_nonpos(double x) {
  // n = rint(x log2 e) by the 1.5 * 2^52 trick: the integer sits in the low mantissa bits of t
  const double t = fma(x, 1.4426950408889634074, 6755399441055744.0);
  const double n = t - 6755399441055744.0;
  double r = fma(n, -6.93147180369123816490e-01, x);
  r = fma(n, -1.90821492927058770002e-10, r);
  double p = 1.0 / 6227020800.0;
  p = fma(p, r, 1.0 / 479001600.0);
  p = fma(p, r, 1.0 / 39916800.0);
  p = fma(p, r, 1.0 / 3628800.0);
  p = fma(p, r, 1.0 / 362880.0);
  p = fma(p, r, 1.0 / 40320.0);
  p = fma(p, r, 1.0 / 5040.0);
  p = fma(p, r, 1.0 / 720.0);
  p = fma(p, r, 1.0 / 120.0);
  p = fma(p, r, 1.0 / 24.0);
  p = fma(p, r, 1.0 / 6.0);
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  // 2^n assembled in the exponent field (n clamped at -1022: results below 2^-1022 are not distinguished)
  int ni = __double2loint(t);
  ni = ni < -1022 ? -1022 : ni;
  return p * __hiloint2double((ni + 1023) << 20, 0);
}

// The same function with a 32-entry table of 2^(j/32) (kept in LDS by the caller: 32 doubles span the 64 banks exactly
// once, so any index pattern of a wave is conflict-free): x = (32 m + j) ln2/32 + r, |r| <= ln2/64, Taylor polynomial of
// degree 6 (remainder 3e-18 relative) -- 12 fp64 instructions instead of 19, <= 2 ulp (checked against mpmath on
// 3e4 arguments in [-700, 0]).  Results below 2^-1074 flush to 0 through v_ldexp_f64.
#define GPAK_EXPTAB_N 32
static __constant__ double gpak_exp2_tab[GPAK_EXPTAB_N] = {
    0x1.0000000000000p+0, 0x1.059b0d3158574p+0, 0x1.0b5586cf9890fp+0, 0x1.11301d0125b51p+0, 0x1.172b83c7d517bp+0,
    0x1.1d4873168b9aap+0, 0x1.2387a6e756238p+0, 0x1.29e9df51fdee1p+0, 0x1.306fe0a31b715p+0, 0x1.371a7373aa9cbp+0,
    0x1.3dea64c123422p+0, 0x1.44e086061892dp+0, 0x1.4bfdad5362a27p+0, 0x1.5342b569d4f82p+0, 0x1.5ab07dd485429p+0,
    0x1.6247eb03a5585p+0, 0x1.6a09e667f3bcdp+0, 0x1.71f75e8ec5f74p+0, 0x1.7a11473eb0187p+0, 0x1.82589994cce13p+0,
    0x1.8ace5422aa0dbp+0, 0x1.93737b0cdc5e5p+0, 0x1.9c49182a3f090p+0, 0x1.a5503b23e255dp+0, 0x1.ae89f995ad3adp+0,
    0x1.b7f76f2fb5e47p+0, 0x1.c199bdd85529cp+0, 0x1.cb720dcef9069p+0, 0x1.d5818dcfba487p+0, 0x1.dfc97337b9b5fp+0,
    0x1.ea4afa2a490dap+0, 0x1.f50765b6e4540p+0};
// exp(-s), s >= 0 finite (NaN stays NaN); tab = the table above in LDS
__device__ __forceinline__ double gpak_exp_neg_tab(double s, const double *tab) {
  const double t = fma(s, -0x1.71547652b82fep+5, 6755399441055744.0);   // n = rint(-s * 32/ln2) in the low mantissa bits
  const double n = t - 6755399441055744.0;
  double r = fma(n, -0x1.62e42fefa0000p-6, -s);                         // ln2/32 split so that n * hi is exact
  r = fma(n, -0x1.cf79abc9e3b3ap-45, r);
  double p = 0x1.6c16c16c16c17p-10;                                     // 1/720
  p = fma(p, r, 0x1.1111111111111p-7);
  p = fma(p, r, 0x1.5555555555555p-5);
  p = fma(p, r, 0x1.5555555555555p-3);
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  const int ni = __double2loint(t);
  return ldexp(tab[ni & (GPAK_EXPTAB_N - 1)] * p, ni >> 5);
}
// sqrt(d) for the argument of exp(-sqrt(d)): v_rsq_f64 + two Goldschmidt steps, no final residual correction (an ulp
// or two in s is an absolute 1e-16 * s in the exponent)
// The argument is clamped at 640000 (s = 800: exp(-s) has flushed to 0 long before, at s = 745) so that the table
// exponent of gpak_exp_neg_tab -- rint(-s 32/ln2) read from 32 mantissa bits -- can never wrap (it did for s > 4.65e7
// and gave +inf), and an overflowed D2 = +inf yields K = bias instead of NaN.  Only the HIGH dword is replaced (one
// v_cmp_gt_f64 + one v_cndmask_b32; whatever the low dword holds, the result still flushes); a NaN coordinate fails
// the comparison and stays NaN.
__device__ __forceinline__ double gpak_sqrt_nonneg_fast(double d) {
  d = __hiloint2double(d > 640000.0 ? 0x41238800 : __double2hiint(d), __double2loint(d));   // 640000.0 = 0x4123880000000000
  const double y = __builtin_amdgcn_rsq(d);
  double g = d * y, h = 0.5 * y;
  double r = fma(-h, g, 0.5);
  g = fma(g, r, g);
  h = fma(h, r, h);
  r = fma(-h, g, 0.5);
  g = fma(g, r, g);
  return d > 0.0 ? g : d;
}
#endif

// ---- gram.hip ---------------------------------------------------------------------------
// u = (x - mu) A for n points; x SoA with stride xs.
void gpak_launch_transform(hipStream_t st, const double *x, int xs, int n, const KernParams &kp,
                           DevPoints &out);
// Fill C (rows x cols tile grid, ld) with scale*K(P_i,Q_j) + diag*delta_ij; rows/cols beyond
// the valid counts get delta_ij. lower_only skips tiles strictly above the diagonal.
void gpak_launch_fill(hipStream_t st, const DevPoints &P, const DevPoints &Q, int rows_p, int cols_p,
                      const KernParams &kp, double scale, double diag, double pad_diag, int lower_only,
                      double *C, long ld, double *D2out, int col_off = 0);
// out_j = sum_i w_i K(P_i, Q_j), j < Q.n   (fused Gram-matvec; K never stored).
// scratch holds max(splits, scratch_rows) * Q.cap doubles (scratch_rows: what the caller really has; the symmetric
// kernel for P == Q needs one row per 512 points).
int gpak_kmatvec_splits(int nP, int nQ);
void gpak_launch_kmatvec(hipStream_t st, const DevPoints &P, int p_off, int np, const double *w, const DevPoints &Q,
                         const KernParams &kp, double *scratch, int splits, double *out, int scratch_rows = 0);
void gpak_launch_sum_splits(hipStream_t st, const double *part, int part_ld, int splits, int n, double *out);
int gpak_alloc_points(gpak_ctx *ctx, DevPoints &p, int cap);
void gpak_pooled_mean(const double *s1, long n, const double *s2, long m, double *mu);
int gpak_ensure_U(gpak_ctx *ctx);

// ---- gemm.hip ---------------------------------------------------------------------------
// C[mt x nt tiles of 128] = beta*C + alpha * A (m x K) * B (n x K)^T, all column-major.
// lower_skip: skip tile (ti,tj) when row_block0+ti < col_block0+tj.
// trailing=true selects the instantiation gpak_gemm_nt_f64_rs<4, 2, true> (its own line in profiles).
void gpak_launch_gemm_nt(hipStream_t st, int mt, int nt, int K, double alpha, const double *A, long lda,
                         const double *B, long ldb, double beta, double *C, long ldc, int row_block0,
                         int col_block0, bool lower_skip, bool trailing, bool k0_by_row = false);

void gpak_launch_gemm_nt_k0map(hipStream_t st, int mt, int nt, int K, double alpha, const double *A, long lda,
                               const double *B, long ldb, double *C, long ldc, int skip_shift, int k0_mul, int k0_add);
void gpak_launch_gemm_cyclic(hipStream_t st, int mt, int nt, int K, const double *Pv, long ldp, double *Clocal,
                             long ldc, int rt0, int P, int rank, int tpb, int lt0);

// ---- gemm_f32.hip (fp32 prediction path) ---------------------------------------------------
void gpak_launch_gemm_nt_f32(hipStream_t st, int mt, int nt, int K, float alpha, const float *A, long lda,
                             const float *B, long ldb, float beta, float *C, long ldc);
void gpak_launch_lower_to_f32(hipStream_t st, const double *L, long ld, int Np, float *out, long ldo);
void gpak_launch_vec_to_f32(hipStream_t st, const double *in, size_t n, float *out);
void gpak_launch_fill_f32(hipStream_t st, const DevPoints &P, const DevPoints &Q, int rows_p, int cols_p,
                          const KernParams &kp, float *C, long ld);
void gpak_launch_rowsumsq_f32(hipStream_t st, const float *V, long ldv, int rows, int cols, int splits,
                              double *part, int part_ld);

// ---- potrf.hip --------------------------------------------------------------------------
// Factor the 128x128 block at A (ld) in place (lower), write its inverse to inv (128x128, ld 128).
void gpak_launch_potrf128(hipStream_t st, double *A, long ld, double *inv, int col0, int *info, bool zero_inv,
                          bool co = false, int co_mode = -1);
int gpak_potrf_blocked(gpak_ctx *ctx);
void gpak_factor_panel(hipStream_t st, double *M, long ld, int Np, int J, int W, double *inv_base, int *info,
                       bool zero_inv, bool co = false, int co_mode = -1, hipEvent_t gate = nullptr);

// ---- solve.hip --------------------------------------------------------------------------
// x := L^-1 x ; x := L^-T x  (x has Np entries) using the inverted diagonal blocks.
// x is consumed (overwritten with intermediate values); the solution is written to out.
void gpak_launch_trsv_fwd(hipStream_t st, int Np, const double *L, long ld, const double *inv, double *x,
                          double *out);
// two-level variant: z is read-only, scratch holds 8 * 512 doubles
void gpak_launch_trsv_bwd_block2(hipStream_t st, int Np, int J, int W, const double *L, long ld, const double *inv,
                                 const double *z, double *out, double *scratch, const double *Rinv = nullptr, int NB = 512);
// R = (L_bb^-1)^T of the W x W diagonal block at J (W <= 512), column-major ld 512; seven small GEMM launches
void gpak_launch_diag_inverse(hipStream_t st, int J, int W, const double *L, long ld, const double *inv, double *R,
                              int RL = 512, int extra = 0);
void gpak_launch_trsv_bwd2(hipStream_t st, int Np, const double *L, long ld, const double *inv, const double *z,
                           double *out, double *scratch, const double *RinvB = nullptr, int bw = 512);
// far column dots of the next block column under the diagonal step of this one (solve.hip): scratch 18 * NB doubles,
// OP = per block column R_b (and, with_T, T_b in the rows below it: gpak_launch_diag_inverse with extra = NB)
void gpak_launch_trsv_bwd3(hipStream_t st, int Np, const double *L, long ld, const double *z, double *out, double *scratch,
                           const double *OP, size_t op_stride, int RL, bool with_T, int NB);
void gpak_launch_trsv_bwd(hipStream_t st, int Np, const double *L, long ld, const double *inv, double *x,
                          double *out);
void gpak_launch_trsv_fwd_block(hipStream_t st, int Np, int J, int W, const double *L, long ld,
                                const double *inv, double *x, double *out);
void gpak_launch_trsv_bwd_block(hipStream_t st, int J, int W, const double *L, long ld, const double *inv,
                                double *x, double *out);
void gpak_launch_coldot(hipStream_t st, int Np, int i0, int J, int W, const double *L, long ld, const double *x,
                        double *s);
void gpak_launch_logdiag_block(hipStream_t st, int J, int W, int N, const double *L, long ld, double *out);
// red[0] = sum log L_ii (i < N)
void gpak_launch_logdet(hipStream_t st, int N, const double *L, long ld, double *red);
// red[1] = sum alpha_i * 0.5 f_i ; red[2] = sum lp_i   (GP_Utils.cpp:810, 1159)
void gpak_launch_nlz_terms(hipStream_t st, int N, const double *y, const double *f, const double *alpha,
                           double sn2, double *red);
void gpak_launch_scale(hipStream_t st, int n, const double *in, double s, double *out);
void gpak_launch_axpy(hipStream_t st, int n, double a, const double *x, double *y);  // y += a x

// ---- predict.hip ------------------------------------------------------------------------
void gpak_predict_release(gpak_ctx *ctx);

// ---- the distributed factor as one rank holds it (dist.hip), handed to a single-GPU context of the SAME device
// (multi.hip: group prediction and solve_chol reuse the factor instead of factoring a replica again) ----
struct gpak_dist;
struct gpak_dist_factor_view {
  int N, Np, nb, nJ;
  const double *const *panels;   // per block column: packed W x (Np - J), leading dimension Np - J, rows from the diagonal block down
  const double *const *invs;     // per block column: W/128 x 2 x 128 x 128 inverted diagonal blocks (inverse, inverse transposed)
  const double *alpha, *f;       // Np each (device)
  double quad, sumlp, logdet, nlz;
};
int gpak_dist_factor_view_get(gpak_dist *h, gpak_dist_factor_view *out);
double gpak_dist_grad_ms(const gpak_dist *h);
int gpak_import_factor(gpak_ctx *ctx, const gpak_dist_factor_view *v);

// ---- grad.hip ---------------------------------------------------------------------------
void gpak_grad_release(gpak_ctx *ctx);
void gpak_grad_assemble(const KernParams &kp, const int *kinds, const double *expans, int d, int N, double sn2,
                        const double *red, double *g);
