"""A/B: WHICH compute units the tail's bulk queue leaves idle (GPAK_TAIL_MASK_STRIDE: CU c*stride for c < GPAK_TAIL_MASK;
stride 1 = CUs 0..7, stride 32 = one per XCD if CU indices are XCD-major) and from how many rows on (GPAK_TAIL_ROWS)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gp_ss_ak_amd import gpak, synth  # noqa: E402

E = np.array(synth.DEFAULT_EXPANS)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
X, y = synth.drillholes(N)
for stride, rows, mask in ((1, 12288, 8), (32, 12288, 8), (33, 12288, 8), (8, 12288, 8), (32, 16384, 8), (32, 20480, 8), (32, 32768, 8),
                           (1, 32768, 8), (32, 24576, 16), (1, 12288, 8)):
    os.environ["GPAK_TAIL_MASK_STRIDE"] = str(stride)
    os.environ["GPAK_TAIL_ROWS"] = str(rows)
    os.environ["GPAK_TAIL_MASK"] = str(mask)
    gpak._lib.load().gpak_reload_tuning()
    g = gpak.Gpak(0)
    g.set_train(X, y)
    steps = 12 if N > 16384 else 30
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
    print(f"N={N} stride {stride:3d} tail_rows {rows:6d} mask {mask:3d}: step {ms:8.3f} ms  factor {fac / steps:8.3f} ms", flush=True)
    g.close()
