"""Repeated identical train steps (and gradients) must give bit-identical results: every reduction in the library
has a fixed order, and no kernel depends on workgroup placement or timing (diagnostic, GPU only)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gp_ss_ak_amd import gpak, synth
for N in (32768, 5000, 12000):
    X, y = synth.drillholes(N)
    g = gpak.Gpak(0)
    g.set_train(X, y)
    vals, grads = [], []
    for i in range(6 if N == 32768 else 12):
        g.set_params(synth.DEFAULT_EXPANS, synth.DEFAULT_BIAS, synth.DEFAULT_SN2, 1)
        vals.append(g.logLikelihood())
        if i % 3 == 0:
            grads.append(g.GradLL().tobytes())
    a = g.solve_alpha().tobytes()
    print(N, "nlz identical:", len(set(vals)) == 1, vals[0], "grad identical:", len(set(grads)) == 1)
    g.close()
