"""Sustained MFMA-f64 rate: the calibration kernel (registers only) back to back for ~1.5 s, and the same
right after a full N=32768 train step (hot chip).  Shows whether the clock holds under a long MFMA load."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gp_ss_ak_amd import gpak, synth

g = gpak.Gpak(0)
rates = [round(g.calibrate()[0], 1) for _ in range(40)]
print("cold, 40 back-to-back calibrations (TFLOP/s):", rates)
X, y = synth.drillholes(32768)
g.set_train(X, y)
for i in range(3):
    g.set_params(synth.DEFAULT_EXPANS, synth.DEFAULT_BIAS, synth.DEFAULT_SN2 + 1e-3 * i, 1)
    t = time.perf_counter(); g.logLikelihood(); dt = time.perf_counter() - t
    print("step %.1f ms, calibration right after: %.1f TFLOP/s" % (dt * 1e3, g.calibrate()[0]))
