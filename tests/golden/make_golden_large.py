#!/usr/bin/env python3
"""LAPACK-computed golden scalars at BASELINE.json's full sizes (tests/golden/golden_N<k>.json).

The reference ships no fixtures and cannot be built here (needs Armadillo), so -- like
make_golden.py -- these are NOT reference outputs.  They pin the metric's own size to numbers that
do not come from the HIP path and do not come from the oracle's factorisation either:

  K      : the oracle's Gram (oracle/gpak_oracle.c orc_gram: Kernel.cpp:856-882, 1370-1435), both
           distance formulations (DIRECT = the product default, EXPANSION = the reference as written)
  B      : I + K/sn2                                      (GP_Utils.cpp:898-902)
  chol   : scipy.linalg.cho_factor  (OpenBLAS dpotrf -- the routine arma::chol reaches); beyond N = 16384
           a blocked right-looking factorisation over dpotrf / dtrsm / dgemm on 4096-blocks (see blocked_chol)
  alpha  : L^-T L^-1 (y/sn2) by blocked dtrtrs / dgemv    (the IRLS fixed point, GP_Utils.cpp:191-228)
  f      : K alpha by dgemv on a freshly rebuilt K        (GP_Utils.cpp:1147)
  nlZ    : alpha'(f/2) - sum(lp) + sum(log diag)          (GP_Utils.cpp:1159, 810)
  mean/var at 16 test points: kX' alpha, kD - |L^-1 kX sW|^2 + sn2   (GP_Utils.cpp:958-1004, 1016-1043)

Memory: one N x N double matrix (N=32768: 8.6 GB, N=65536: 34 GB) + slabs.
Run from the repo root:  python tests/golden/make_golden_large.py 32768 [8192 12288 ...]
"""
import json
import math
import os
import sys
import time

import numpy as np
import scipy.linalg as sl

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gp_ss_ak_amd import synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
NB = 4096


# SciPy's dpotrf wrapper crashes (SIGSEGV inside OpenBLAS) for N >= 32768 in this image, so the factorisation
# and the solves are blocked here: LAPACK dpotrf / dtrtrs on 4096 x 4096 diagonal blocks, BLAS dtrsm / dgemm /
# dgemv for everything else.  Checked against the unblocked cho_factor at N = 12288 (same logdet to 1e-15).
def blocked_chol(B):
    """In place: the lower triangle of B (F-ordered) becomes L; the strict upper triangle is left as it was."""
    N = B.shape[0]
    for j in range(0, N, NB):
        j1 = min(N, j + NB)
        Ljj = sl.cholesky(B[j:j1, j:j1], lower=True, check_finite=False)
        B[j:j1, j:j1] = Ljj
        if j1 < N:
            # P = A21 L^-T  <=>  L P' = A21'
            P = sl.solve_triangular(Ljj, B[j1:, j:j1].T, lower=True, check_finite=False).T
            B[j1:, j:j1] = P
            for k in range(j1, N, NB):
                k1 = min(N, k + NB)
                B[k:, k:k1] -= P[k - j1:] @ P[k - j1:k1 - j1].T
    return B


def fwd_solve(L, b):
    z = np.array(b, dtype=np.float64, order="F", copy=True)
    N = L.shape[0]
    for j in range(0, N, NB):
        j1 = min(N, j + NB)
        z[j:j1] = sl.solve_triangular(L[j:j1, j:j1], z[j:j1], lower=True, check_finite=False)
        if j1 < N:
            z[j1:] -= L[j1:, j:j1] @ z[j:j1]
    return z


def bwd_solve(L, z):
    x = np.array(z, dtype=np.float64, order="F", copy=True)
    N = L.shape[0]
    for j in range((N - 1) // NB * NB, -1, -NB):
        j1 = min(N, j + NB)
        if j1 < N:
            x[j:j1] -= L[j1:, j:j1].T @ x[j1:]
        x[j:j1] = sl.solve_triangular(L[j:j1, j:j1], x[j:j1], lower=True, trans="T", check_finite=False)
    return x
E = np.array(synth.DEFAULT_EXPANS)
BIAS, SN2 = synth.DEFAULT_BIAS, synth.DEFAULT_SN2


def one(N, modes=("direct", "expansion")):
    X, y = synth.drillholes(N)
    Xte = synth.test_points(16)
    out = {"N": N, "expans": [float(v) for v in E], "bias": BIAS, "sn2": SN2,
           "data": "gp_ss_ak_amd.synth.drillholes(N), synth.test_points(16)",
           "how": "oracle Gram + scipy.linalg.cho_factor/cho_solve (OpenBLAS LAPACK); see make_golden_large.py"}
    rng = np.random.default_rng(11 + N)
    idx = np.sort(rng.choice(N, 32, replace=False))
    out["alpha_idx"] = [int(i) for i in idx]
    for name in modes:
        mode = orc.DIST_DIRECT if name == "direct" else orc.DIST_EXPANSION
        t0 = time.time()
        if N <= 32768:
            B = orc.gram(X, X, E, BIAS, mode)        # F-ordered N x N
        else:
            # the oracle's Gram keeps a second N x N array (D2): beyond N = 32768 that does not fit beside B in
            # 62 GB, so B is assembled from 4096-column slabs (direct mode: the same numbers up to the rounding of
            # the centring, which differs per slab; the expansion form is not generated at this size)
            B = np.empty((N, N), order="F")
            for j0 in range(0, N, 4096):
                j1 = min(N, j0 + 4096)
                B[:, j0:j1] = orc.gram(X, np.asfortranarray(X[j0:j1]), E, BIAS, mode)
        B *= 1.0 / SN2
        B[np.diag_indices(N)] += 1.0
        t1 = time.time()
        if N <= 16384 and not os.environ.get("GPAK_GOLDEN_BLOCKED"):
            c, _ = sl.cho_factor(B, lower=True, overwrite_a=True, check_finite=False)
        else:
            c = blocked_chol(B)
        t2 = time.time()
        logdet = float(np.log(c.diagonal()).sum())
        alpha = bwd_solve(c, fwd_solve(c, y / SN2))
        kX = orc.gram(X, Xte, E, BIAS, mode)         # N x 16
        mean = kX.T @ alpha
        v = fwd_solve(c, kX / math.sqrt(SN2))
        kD = E[6] ** 2 + BIAS
        var = kD - (v * v).sum(axis=0) + SN2
        del c, B, v
        f = np.empty(N)                              # f = K alpha, K rebuilt in 4096-column slabs (K is symmetric)
        for j0 in range(0, N, 4096):
            j1 = min(N, j0 + 4096)
            f[j0:j1] = orc.gram(X, np.asfortranarray(X[j0:j1]), E, BIAS, mode).T @ alpha
        quad = float(alpha @ (0.5 * f))
        sumlp = float((-(y - f) ** 2 / (2.0 * SN2) - 0.5 * math.log(2.0 * math.pi * SN2)).sum())
        out[name] = {"nlz": quad - sumlp + logdet, "logdet": logdet, "quad": quad, "sumlp": sumlp,
                     "alpha_samples": [float(a) for a in alpha[idx]], "alpha_norm": float(np.linalg.norm(alpha)),
                     "mean": [float(m) for m in mean], "var": [float(s) for s in var],
                     "residual_rel": float(np.abs(y - f - SN2 * alpha).max() / np.abs(y).max())}
        print(f"N={N} {name}: gram {t1 - t0:.1f}s chol {t2 - t1:.1f}s total {time.time() - t0:.1f}s "
              f"nlz={out[name]['nlz']:.15g} logdet={logdet:.15g}", flush=True)
    with open(os.path.join(HERE, f"golden_N{N}.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    return out


if __name__ == "__main__":
    for n in [int(a) for a in sys.argv[1:]] or [8192]:
        one(n, modes=("direct", "expansion") if n <= 32768 else ("direct",))
