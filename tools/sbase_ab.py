"""A/B of GPAK_SBASE_ROWS (gemm.hip): bulk trailing updates of more rows than this use the build of the register-streaming
kernel whose operand addresses are scalar bases + a lane offset (no vector ALU instruction in the loop but the MFMAs).
Interleaved, factor_ms / step / results against the run without it.  Usage: python tools/sbase_ab.py [N ...]"""
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
    for rows in (0, 8192, 12288, 16384, 20480, 24576) * 2:
        os.environ["GPAK_SBASE_ROWS"] = str(rows)
        gpak._lib.load().gpak_reload_tuning()
        g = gpak.Gpak(0)
        g.set_train(X, y)
        steps, factor = 12 if N > 10000 else 40, 0.0
        for i in range(3):
            g.set_params(E, synth.DEFAULT_BIAS, synth.DEFAULT_SN2 * (1 + 1e-3 * i), gpak.DIST_DIRECT)
            g.logLikelihood()
        t0 = time.perf_counter()
        for i in range(steps):
            g.set_params(E, synth.DEFAULT_BIAS, synth.DEFAULT_SN2 * (1 + 1e-3 * (i % 3)), gpak.DIST_DIRECT)
            g.logLikelihood()
            factor += g.timing()["factor_ms"]
        ms = (time.perf_counter() - t0) / steps * 1e3
        g.set_params(E, synth.DEFAULT_BIAS, synth.DEFAULT_SN2, gpak.DIST_DIRECT)
        nlz, a = g.logLikelihood(), g.solve_alpha()
        if ref is None:
            ref = (nlz, a)
        print(f"N={N} sbase_rows {rows:>10d}: step {ms:8.3f} ms  factor {factor / steps:8.3f}  "
              f"nlz rel diff {abs(nlz - ref[0]) / abs(ref[0]):.1e}  alpha rel diff {np.abs(a - ref[1]).max() / np.abs(ref[1]).max():.1e}",
              flush=True)
        g.close()
