"""A/B of the back substitution: one launch per 512-column step (coupling blocks T_b, GPAK_OPT_BWD_FUSED = 1, the
default) against the three-launch step (column dots, diagonal matrix-vector product, sum).  solve_ms, step time, alpha
of one against the other, interleaved.  Usage: python tools/bwd_fused_ab.py [N ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gp_ss_ak_amd import gpak, synth  # noqa: E402

E = np.array(synth.DEFAULT_EXPANS)
for N in [int(a) for a in sys.argv[1:]] or [100, 700, 1400, 5000, 8192, 32768]:
    X, y = synth.drillholes(max(N, 8))
    X, y = np.asfortranarray(X[:N]), y[:N].copy()
    g = gpak.Gpak(0)
    g.set_train(X, y)
    ref = None
    for fused in (0, 1, 2, 0, 1, 2):
        g.set_option(gpak.OPT_BWD_FUSED, fused)
        steps, solve, factor = 12, 0.0, 0.0
        for i in range(3):
            g.set_params(E, synth.DEFAULT_BIAS, synth.DEFAULT_SN2 * (1 + 1e-3 * i), gpak.DIST_DIRECT)
            g.logLikelihood()
        t0 = time.perf_counter()
        for i in range(steps):
            g.set_params(E, synth.DEFAULT_BIAS, synth.DEFAULT_SN2 * (1 + 1e-3 * (i % 3)), gpak.DIST_DIRECT)
            g.logLikelihood()
            t = g.timing()
            solve += t["solve_ms"]
            factor += t["factor_ms"]
        ms = (time.perf_counter() - t0) / steps * 1e3
        g.set_params(E, synth.DEFAULT_BIAS, synth.DEFAULT_SN2, gpak.DIST_DIRECT)
        nlz, a = g.logLikelihood(), g.solve_alpha()
        if ref is None:
            ref = (nlz, a)
        print(f"N={N} fused {fused}: step {ms:8.3f} ms  factor {factor / steps:8.3f}  solve {solve / steps:6.3f} ms  "
              f"nlz rel diff {abs(nlz - ref[0]) / abs(ref[0]):.1e}  alpha rel diff {np.abs(a - ref[1]).max() / np.abs(ref[1]).max():.1e}",
              flush=True)
    g.close()
