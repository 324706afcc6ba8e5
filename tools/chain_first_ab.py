"""A/B: the tail's first panel step on a queue confined to the CUs the bulk update leaves idle (GPAK_CHAIN_FIRST) for several
widths of that mask (GPAK_TAIL_MASK)."""
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
    for cf, mask in ((0, 8), (1, 8), (1, 16), (1, 32), (1, 48), (1, 64), (0, 32), (0, 8), (1, 32)):
        os.environ["GPAK_CHAIN_FIRST"] = str(cf)
        os.environ["GPAK_TAIL_MASK"] = str(mask)
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
        nlz = g.logLikelihood()
        ref = ref or nlz
        print(f"N={N} chain_first {cf} mask {mask:3d}: step {ms:8.3f} ms  factor {fac / steps:8.3f} ms  nlz rel diff {abs(nlz - ref) / abs(ref):.1e}",
              flush=True)
        g.close()
