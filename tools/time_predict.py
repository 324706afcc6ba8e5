"""Diagnostic: predictive mean/variance throughput (GPU only)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gp_ss_ak_amd import gpak, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
M = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
prec = gpak.F32 if (len(sys.argv) > 3 and sys.argv[3] == "f32") else gpak.F64
X, y = synth.drillholes(N)
g = gpak.Gpak(0, prec)
g.set_train(X, y)
g.set_params(np.array(synth.DEFAULT_EXPANS), synth.DEFAULT_BIAS, synth.DEFAULT_SN2)
print("nlz", g.logLikelihood())
Xt = synth.test_points(M)
for want_var in (False, True):
    g.posteriorMeanVar(Xt[:256], want_var=want_var)
    t0 = time.perf_counter()
    mean, var = g.posteriorMeanVar(Xt, want_var=want_var)
    dt = time.perf_counter() - t0
    flops = (N * N * M) if want_var else 0
    print(f"prec={'f32' if prec else 'f64'} N={N} M={M} var={want_var}: {dt*1e3:.1f} ms, {M/dt:.0f} points/s" + (f", {flops/dt/1e12:.1f} TFLOP/s (N^2 M)" if want_var else ""))
print(mean[:3], var[:3])
