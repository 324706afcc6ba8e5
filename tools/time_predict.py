"""configs[4] alone: N = 32768 training points, M test points through a GPAK_F32 context (fp32 MFMA substitution).
Usage: python tools/time_predict.py [label] [M] [f32|f64]   -- prints the wall time and TFLOP/s of the variance pass; under
tools/predict_trace.sh it is the program rocprofv3 traces."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gp_ss_ak_amd import gpak, synth  # noqa: E402

M = int(sys.argv[2]) if len(sys.argv) > 2 else 131072
N = 32768
E = np.array(synth.DEFAULT_EXPANS)
X, y = synth.drillholes(N)
PREC = gpak.F64 if (len(sys.argv) > 3 and sys.argv[3] == "f64") else gpak.F32
g = gpak.Gpak(0, PREC)
g.set_train(X, y)
g.set_params(E, synth.DEFAULT_BIAS, synth.DEFAULT_SN2, gpak.DIST_DIRECT)
g.logLikelihood()
Xte = synth.test_points(M)
g.posteriorMeanVar(Xte[:65536].copy(order="F"))          # buffers, fp32 image of the factor
for rep in range(2):
    t0 = time.perf_counter()
    mean, var = g.posteriorMeanVar(Xte)
    dt = time.perf_counter() - t0
    print(f"{'f64' if PREC == gpak.F64 else 'f32'} M={M}: var[:3] {var[:3]} sum {var.sum():.12e}; {dt * 1e3:.1f} ms, {N * float(N) * M / dt / 1e12:.1f} TFLOP/s (N^2 M flop), {M / dt / 1e3:.1f} k points/s", flush=True)
g.close()
