#!/usr/bin/env python3
"""Generates the golden vectors under tests/golden/ from the CPU oracle (oracle/).

The reference ships no fixtures and cannot run here (needs Armadillo), so these vectors are
NOT reference outputs: they freeze the oracle's restatement (PARITY UNPINNED, see
oracle/gpak_oracle.h) so that (a) the oracle cannot drift silently and (b) the HIP path can be
checked against committed numbers on the GPU box without trusting a freshly built oracle.
Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gp_ss_ak_amd import synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
E = np.array(synth.DEFAULT_EXPANS)
BIAS, SN2 = synth.DEFAULT_BIAS, synth.DEFAULT_SN2


def one(N, d=3):
    orc.use_builtin()
    if d == 3:
        X, y = synth.drillholes(N)
        Xte = synth.test_points(16)
    else:   # x, y, z + rock-type column
        X, y = synth.drillholes4(N)
        Xte = synth.test_points4(16)
    out = {"N": N, "X": X, "y": y, "Xte": Xte, "expans": E, "bias": BIAS, "sn2": SN2}
    for name, mode in (("direct", orc.DIST_DIRECT), ("expansion", orc.DIST_EXPANSION)):
        K = orc.gram(X, X, E, BIAS, mode)
        info, alpha, L = orc.nlz_refseq(K, y, SN2)
        lean, alpha_l, _ = orc.nlz_lean(K, y, SN2)
        mean, var = orc.predict(X, Xte, E, BIAS, SN2, alpha, L, mode, 0)
        mean_c, var_c = orc.predict(X, Xte, E, BIAS, SN2, alpha, L, mode, orc.COMPAT_VARCLAMP | orc.COMPAT_SN2SKIP)
        g = orc.grad_ref(X, y, K, L, alpha, E, BIAS, SN2, mode)
        rng = np.random.default_rng(7 + N)
        ii, jj = rng.integers(0, N, 32), rng.integers(0, N, 32)
        out.update({
            f"{name}_nlz": info.nlz, f"{name}_nlz_lean": lean.nlz, f"{name}_logdet": info.logdet,
            f"{name}_quad": info.quad, f"{name}_sumlp": info.sumlp,
            f"{name}_n_chol": info.n_chol, f"{name}_n_gemv": info.n_gemv, f"{name}_irls": info.irls_iters,
            f"{name}_mean": mean, f"{name}_var": var, f"{name}_var_compat": var_c, f"{name}_grad": g,
            f"{name}_K_sum": K.sum(), f"{name}_K_samples": K[ii, jj], f"{name}_sample_ij": np.stack([ii, jj]),
        })
        if N <= 64:
            out[f"{name}_K"] = K
            out[f"{name}_alpha"] = alpha
            out[f"{name}_L"] = L
    tag = f"golden_N{N}" + ("" if d == 3 else f"_d{d}")
    np.savez_compressed(os.path.join(HERE, tag + ".npz"), **out)
    print("wrote %s.npz  nlz(direct)=%.15g" % (tag, out["direct_nlz"]))


if __name__ == "__main__":
    for N in (8, 64, 512):
        one(N)
    one(64, d=4)
