"""A/B of the fused panel-step kernel (GPAK_SU_MAX_MT: fused solve + in-panel update while at most that many row tiles
are left; 0 = the two-launch path): step / factor time and the result against the unfused path."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gp_ss_ak_amd import gpak, synth  # noqa: E402

E = np.array(synth.DEFAULT_EXPANS)
for N in [int(a) for a in sys.argv[1:]] or [8192]:
    X, y = synth.drillholes(N)
    ref = None
    for mx in (0, 32, 64, 96, 160, 256, 0, 96):
        os.environ["GPAK_SU_MAX_MT"] = str(mx)
        gpak._lib.load().gpak_reload_tuning()
        g = gpak.Gpak(0)
        g.set_train(X, y)
        steps = 30 if N <= 16384 else 10
        for i in range(3):
            g.set_params(E, synth.DEFAULT_BIAS, synth.DEFAULT_SN2 * (1 + 1e-3 * i), gpak.DIST_DIRECT)
            g.logLikelihood()
        fac = 0.0
        t0 = time.perf_counter()
        for i in range(steps):
            g.set_params(E, synth.DEFAULT_BIAS, synth.DEFAULT_SN2 * (1 + 1e-3 * (i % 3)), gpak.DIST_DIRECT)
            g.logLikelihood()
            fac += g.timing()["factor_ms"]
        ms = (time.perf_counter() - t0) / steps * 1e3
        g.set_params(E, synth.DEFAULT_BIAS, synth.DEFAULT_SN2, gpak.DIST_DIRECT)
        nlz, a = g.logLikelihood(), g.solve_alpha()
        if ref is None:
            ref = (nlz, a)
        print(f"N={N} su_max_mt {mx:4d}: step {ms:8.3f} ms  factor {fac / steps:8.3f} ms  nlz rel diff "
              f"{abs(nlz - ref[0]) / abs(ref[0]):.1e}  alpha rel diff {np.abs(a - ref[1]).max() / np.abs(ref[1]).max():.1e}", flush=True)
        g.close()
