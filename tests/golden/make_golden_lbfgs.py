#!/usr/bin/env python3
"""Oracle-driven L-BFGS trajectory fixture at BASELINE configs[1]'s size (tests/golden/golden_lbfgs_N<k>.json).

What `gp_ss_ak train -o LBFGS` must reproduce on the GPU: tests/lbfgs_ref.py (the NumPy restatement of
Opt_Algs::LBFGSOptimise + the Potra-Shi search, Opt_pars.cpp:179-332, 543-974) driven by the ORACLE's reference
sequence on the host -- orc_nlz_refseq (PSI / irls / brentmin with three Choleskys, alpha warm-started from the
previous evaluation like the member `Alpha`: GP_Utils.cpp:191-381, 872-915, 1138-1162) and orc_grad_ref (GradLL +
getGradients as written: GP_Utils.cpp:1164-1284, Kernel.cpp:886-1263), Choleskys / triangular solves / GEMVs issued
to the SciPy wheel's OpenBLAS as Armadillo would.  Generated HERE (the build container), not on the GPU box; no
HIP code takes part.  Still not reference output (parity unpinned: the reference cannot be built).

Per iteration: kept objective, cumulative evaluation count, kept point (10 values), all at 17 digits.
The data are synth.drillholes(N) with the extremes set to exactly -1 / +1, so that Control::prep_symmetric is the
identity and the CLI sees bit-identical inputs through its text file (same device as tests/test_host_cpp.py).

Run from the repo root:  python tests/golden/make_golden_lbfgs.py 8192 6
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import lbfgs_ref  # noqa: E402
from gp_ss_ak_amd import synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def prepared(N):
    Xs, ys = synth.drillholes(N)
    Xs = np.asfortranarray(Xs)
    Xs[np.unravel_index(Xs.argmax(), Xs.shape)] = 1.0
    Xs[np.unravel_index(Xs.argmin(), Xs.shape)] = -1.0
    ys[ys.argmax()], ys[ys.argmin()] = 1.0, -1.0
    return Xs, ys


def main(N, maxit):
    assert orc.use_lapack(0), "SciPy's OpenBLAS is part of the image"
    Xs, ys = prepared(N)
    x0 = list(synth.DEFAULT_EXPANS) + [synth.DEFAULT_BIAS, synth.DEFAULT_SN2]
    state = {"alpha": None, "n": 0, "t0": time.time()}

    def fg(x):
        e, bias, sn2 = np.array(x[:8], dtype=float), float(x[8]), float(x[9])
        K = orc.gram(Xs, Xs, e, bias, orc.DIST_DIRECT)
        info, alpha, L = orc.nlz_refseq(K, ys, sn2, alpha0=state["alpha"])
        assert not info.chol_fail
        state["alpha"] = alpha
        g = orc.grad_ref(Xs, ys, K, L, alpha, e, bias, sn2, orc.DIST_DIRECT)
        state["n"] += 1
        print(f"  eval {state['n']:3d}  nlz {info.nlz:.15g}  ({time.time() - state['t0']:.0f} s)", flush=True)
        return info.nlz, g

    trace = []
    lbfgs_ref.lbfgs_optimise(fg, x0, maxit, trace=trace)
    hist = [h for h, _, _ in trace]
    stall = next((k for k in range(1, len(hist)) if hist[k] == hist[k - 1]), None)
    out = {"N": N, "maxit": maxit, "x0": [float(v) for v in x0],
           "data": "synth.drillholes(N), extremes set to exactly -1/+1 (make_golden_lbfgs.prepared)",
           "how": "tests/lbfgs_ref.py over orc_nlz_refseq (warm-started alpha) + orc_grad_ref, OpenBLAS LAPACK, DIRECT distances",
           "first_stall_iteration": None if stall is None else stall + 1,
           "rows": [{"iteration": k + 1, "objective": float(h), "evaluations": int(n), "x": [float(v) for v in xk]}
                    for k, (h, n, xk) in enumerate(trace)]}
    with open(os.path.join(HERE, f"golden_lbfgs_N{N}.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps({k: v for k, v in out.items() if k != "rows"}), [r["objective"] for r in out["rows"]])


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 8192, int(sys.argv[2]) if len(sys.argv) > 2 else 6)
