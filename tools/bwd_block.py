"""A/B of the back substitution's block width (GPAK_BWD_BLOCK: explicit diagonal-block inverses of 512 / 1024 / 2048
columns): solve_ms, step time and alpha against the 512 result.  Usage: python tools/bwd_block.py [N ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gp_ss_ak_amd import gpak, synth  # noqa: E402

E = np.array(synth.DEFAULT_EXPANS)
for N in [int(a) for a in sys.argv[1:]] or [32768]:
    X, y = synth.drillholes(N)
    ref = None
    for bw in (512, 1024, 2048, 512, 1024, 2048):
        os.environ["GPAK_BWD_BLOCK"] = str(bw)
        gpak._lib.load().gpak_reload_tuning()
        g = gpak.Gpak(0)
        g.set_train(X, y)
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
        print(f"N={N} bwd block {bw:5d}: step {ms:8.3f} ms  factor {factor / steps:8.3f}  solve {solve / steps:6.3f} ms  "
              f"nlz rel diff {abs(nlz - ref[0]) / abs(ref[0]):.1e}  alpha rel diff {np.abs(a - ref[1]).max() / np.abs(ref[1]).max():.1e}",
              flush=True)
        g.close()
