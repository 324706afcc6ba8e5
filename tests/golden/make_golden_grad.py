#!/usr/bin/env python3
"""LAPACK-derived golden for the reference-style gradient g[10] at BASELINE configs[2]'s size
(tests/golden/golden_grad_N<k>.json).

What config 3 (the L-BFGS loop at N = 32768) evaluates per step besides nlZ is GradLL + getGradients
(GP_Utils.cpp:1164-1284, Kernel.cpp:886-1263, 370-377).  The oracle's orc_grad_ref_d needs six N x N arrays
(51 GB at N = 32768) -- this generator needs two:

  B      : I + K/sn2 from the oracle's Gram (DIRECT distances), 4096-column slabs
  L      : blocked right-looking Cholesky over OpenBLAS dpotrf / dtrsm / dgemm (make_golden_large.blocked_chol)
  alpha  : L^-T L^-1 (y/sn2)
  Q      : B^-1 = L^-T L^-1 I, 4096 columns of the identity at a time (dtrtrs on the diagonal blocks, dgemm for
           the rest) -- what GradLL forms at :1202-1206 with solve_chol
  g[10]  : oracle/gpak_oracle.c orc_grad_ref_q -- the as-written sums of orc_grad_ref_d (same citations line
           by line; checked against it in tests/test_oracle.py), K / DD2 / QW / R rebuilt slab by slab

Neither the HIP path nor the oracle's own factorisation or inverse takes part; still NOT reference output
(the reference cannot be built here: parity unpinned).  Memory: 2 N x N doubles (17 GB at N = 32768) + slabs.
Run from the repo root:  python tests/golden/make_golden_grad.py 32768 [8192 ...]
"""
import json
import os
import sys
import time

import numpy as np
import scipy.linalg as sl

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gp_ss_ak_amd import synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from make_golden_large import NB, blocked_chol, bwd_solve, fwd_solve  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
E = np.array(synth.DEFAULT_EXPANS)
BIAS, SN2 = synth.DEFAULT_BIAS, synth.DEFAULT_SN2


def inverse_from_factor(L):
    """Q = L^-T L^-1 (full symmetric N x N, F-ordered) from the lower factor in L's lower triangle."""
    N = L.shape[0]
    Q = np.empty((N, N), order="F")
    for s0 in range(0, N, NB):
        s1 = min(N, s0 + NB)
        Y = np.zeros((N, s1 - s0), order="F")
        Y[s0:s1] = np.eye(s1 - s0)
        # forward substitution: rows above s0 stay zero
        for j in range(s0, N, NB):
            j1 = min(N, j + NB)
            Y[j:j1] = sl.solve_triangular(L[j:j1, j:j1], Y[j:j1], lower=True, check_finite=False)
            if j1 < N:
                Y[j1:] -= L[j1:, j:j1] @ Y[j:j1]
        # back substitution
        for j in range((N - 1) // NB * NB, -1, -NB):
            j1 = min(N, j + NB)
            if j1 < N:
                Y[j:j1] -= L[j1:, j:j1].T @ Y[j1:]
            Y[j:j1] = sl.solve_triangular(L[j:j1, j:j1], Y[j:j1], lower=True, trans="T", check_finite=False)
        Q[:, s0:s1] = Y
    return Q


def one(N, params=None, tag=""):
    e, bias, sn2 = (E, BIAS, SN2) if params is None else params
    X, y = synth.drillholes(N)
    t0 = time.time()
    B = np.empty((N, N), order="F")
    for j0 in range(0, N, 4096):
        j1 = min(N, j0 + 4096)
        B[:, j0:j1] = orc.gram(X, np.asfortranarray(X[j0:j1]), e, bias, orc.DIST_DIRECT)
    B *= 1.0 / sn2
    B[np.diag_indices(N)] += 1.0
    t1 = time.time()
    L = blocked_chol(B)
    iu = np.triu_indices(N, 1) if N <= 4096 else None
    if iu is not None:
        L[iu] = 0.0                 # blocked_chol leaves the strict upper triangle of B; nothing below reads it
    t2 = time.time()
    alpha = bwd_solve(L, fwd_solve(L, y / sn2))
    Q = inverse_from_factor(L)
    t3 = time.time()
    del L, B
    sym = float(np.abs(Q[:2048, -2048:] - Q[-2048:, :2048].T).max() / np.abs(Q).max()) if N >= 4096 else 0.0
    g = orc.grad_ref_q(X, y, Q, alpha, e, bias, sn2, orc.DIST_DIRECT)
    t4 = time.time()
    out = {"N": N, "expans": [float(v) for v in e], "bias": float(bias), "sn2": float(sn2),
           "data": "gp_ss_ak_amd.synth.drillholes(N)", "dist_mode": "direct",
           "how": "oracle Gram + OpenBLAS blocked Cholesky + B^-1 by blocked triangular solves + orc_grad_ref_q; "
                  "see make_golden_grad.py",
           "g": [float(v) for v in g], "alpha_norm": float(np.linalg.norm(alpha)),
           "q_trace": float(np.trace(Q)), "q_symmetry_defect": sym}
    print(f"N={N}{tag}: gram {t1 - t0:.1f}s chol {t2 - t1:.1f}s inverse {t3 - t2:.1f}s sums {t4 - t3:.1f}s\n g={g}",
          flush=True)
    with open(os.path.join(HERE, f"golden_grad_N{N}{tag}.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    return out


if __name__ == "__main__":
    for n in [int(a) for a in sys.argv[1:]] or [8192]:
        one(n)
