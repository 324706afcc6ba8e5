"""A/B: a second CU-masked bulk queue for the MIDDLE phase (GPAK_MID_ROWS / GPAK_MID_MASK): while more than tail_rows but at
most mid_rows rows are left the bulk updates leave mid_mask compute units idle, so that the block kernel's fast 8-wave build
(27 us) is used instead of the co-resident one (~295 us beside a bulk update)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gp_ss_ak_amd import gpak, synth  # noqa: E402

E = np.array(synth.DEFAULT_EXPANS)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
X, y = synth.drillholes(N)
ref = None
for rows, mask in ((0, 2), (20480, 2), (24576, 2), (32768, 2), (20480, 1), (24576, 1), (24576, 4), (16384, 2), (0, 2), (24576, 2)):
    os.environ["GPAK_MID_ROWS"] = str(rows)
    os.environ["GPAK_MID_MASK"] = str(mask)
    gpak._lib.load().gpak_reload_tuning()
    g = gpak.Gpak(0)
    g.set_train(X, y)
    steps = 12
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
    print(f"N={N} mid_rows {rows:6d} mid_mask {mask}: step {ms:8.3f} ms  factor {fac / steps:8.3f} ms  nlz rel diff {abs(nlz - ref) / abs(ref):.1e}",
          flush=True)
    g.close()
