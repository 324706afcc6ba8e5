"""Schedule sweep at the small sizes (BASELINE configs[1]: N=8192): outer panel width, tail threshold, wide panels.
Usage: python tools/sweep_small.py [N ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gp_ss_ak_amd import gpak, synth  # noqa: E402

E = np.array(synth.DEFAULT_EXPANS)


def timed(g, steps=30, warm=5):
    def step(i):
        e = E.copy()
        e[1] += 1e-3 * (i % 7)
        g.set_params(e, synth.DEFAULT_BIAS, synth.DEFAULT_SN2, gpak.DIST_DIRECT)
        return g.logLikelihood()
    for i in range(warm):
        step(i)
    t0 = time.perf_counter()
    f = 0.0
    for i in range(steps):
        step(i)
        f += g.timing()["factor_ms"]
    return (time.perf_counter() - t0) / steps * 1e3, f / steps


for N in [int(a) for a in sys.argv[1:]] or [8192]:
    X, y = synth.drillholes(N)
    g = gpak.Gpak(0)
    g.set_train(X, y)
    base = None
    for name, opts in [("default", {}),
                       ("nb256", {gpak.OPT_NB_OUTER: 256}), ("nb384", {gpak.OPT_NB_OUTER: 384}),
                       ("nb768", {gpak.OPT_NB_OUTER: 768}), ("nb1024", {gpak.OPT_NB_OUTER: 1024}),
                       ("tail0 (bulk never masked)", {gpak.OPT_TAIL_ROWS: 0}),
                       ("tail4096", {gpak.OPT_TAIL_ROWS: 4096}),
                       ("no inv512", {gpak.OPT_INV512: 0}),
                       ("potrf 8-wave always", {gpak.OPT_POTRF_CO: 0}),
                       ("no lookahead", {gpak.OPT_LOOKAHEAD: 0})]:
        defaults = {gpak.OPT_NB_OUTER: 512, gpak.OPT_TAIL_ROWS: 12288, gpak.OPT_INV512: 1, gpak.OPT_POTRF_CO: 1,
                    gpak.OPT_LOOKAHEAD: 1}
        defaults.update(opts)
        for k, v in defaults.items():
            g.set_option(k, v)
        ms, fac = timed(g)
        base = base or ms
        print(f"N={N:6d} {name:28s} step {ms:7.3f} ms  factor {fac:7.3f} ms  ({ms / base:.3f})", flush=True)
    g.close()

# the CU mask of the tail's bulk queue is fixed when a context is made: new context per value
for N in [int(a) for a in sys.argv[1:]] or [8192]:
    X, y = synth.drillholes(N)
    for mask in (8, 16, 32, 64, 96, 128):
        os.environ["GPAK_TAIL_MASK"] = str(mask)
        gpak._lib.load().gpak_reload_tuning()
        g = gpak.Gpak(0)
        g.set_train(X, y)
        ms, fac = timed(g)
        print(f"N={N:6d} tail mask {mask:3d} CUs idle for the chain: step {ms:7.3f} ms  factor {fac:7.3f} ms", flush=True)
        g.close()
