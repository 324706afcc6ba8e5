"""A/B of the fp32 forward substitution's level ladder (GPAK_FS_LEVELS_F32): variance error against the fp64 context
and time per batch, N = 32768.  Usage: python tools/f32_levels.py [M]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gp_ss_ak_amd import gpak, synth  # noqa: E402

N = 32768
M = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
X, y = synth.drillholes(N)
Xte = synth.test_points(M)
E = np.array(synth.DEFAULT_EXPANS)
g64, g32 = gpak.Gpak(0), gpak.Gpak(0, gpak.F32)
for g in (g64, g32):
    g.set_train(X, y)
    g.set_params(E, synth.DEFAULT_BIAS, synth.DEFAULT_SN2, gpak.DIST_DIRECT)
m64, v64 = g64.posteriorMeanVar(Xte)
print(f"fp64 (skew 1): {g64.timing()['predict_ms']:.0f} ms, var range {v64.min():.4g} .. {v64.max():.4g}", flush=True)
os.environ["GPAK_PRED_LD_SKEW"] = "0"
gpak._lib.load().gpak_reload_tuning()
g64b = gpak.Gpak(0)
g64b.set_train(X, y)
g64b.set_params(E, synth.DEFAULT_BIAS, synth.DEFAULT_SN2, gpak.DIST_DIRECT)
g64b.posteriorMeanVar(Xte)
print(f"fp64 (skew 0): {g64b.timing()['predict_ms']:.0f} ms", flush=True)
g64b.close()
del os.environ["GPAK_PRED_LD_SKEW"]
VARIANTS = [
    {},
    {"GPAK_PRED_LD_SKEW": "0"},
    {"GPAK_PRED_LD_SKEW": "2"},
    {"GPAK_F32_ACC": "plain", "GPAK_FS_LEVELS_F32": "128,512"},
    {"GPAK_F32_ACC": "plain", "GPAK_FS_LEVELS_F32": "128,512", "GPAK_PRED_LD_SKEW": "0"},
    {"GPAK_FS_LEVELS_F32": "128,512,2048"},
    {"GPAK_FS_LEVELS_F32": "128,512,4096"},
    {"GPAK_FS_LEVELS_F32": "128,1024,8192"},
    {"GPAK_PRED_BATCH": "32768"},
    {"GPAK_PRED_BATCH": "131072"},
]
KEYS = sorted({k for v in VARIANTS for k in v})
for var in VARIANTS:
    for k in KEYS:
        os.environ.pop(k, None)
    os.environ.update(var)
    gpak._lib.load().gpak_reload_tuning()          # the environment sets the process-wide defaults; a context copies them
    g32.close()
    g32 = gpak.Gpak(0, gpak.F32)
    g32.set_train(X, y)
    g32.set_params(E, synth.DEFAULT_BIAS, synth.DEFAULT_SN2, gpak.DIST_DIRECT)
    g32.posteriorMeanVar(Xte[:512])
    t0 = time.perf_counter()
    m32, v32 = g32.posteriorMeanVar(Xte)
    dt = time.perf_counter() - t0
    dv = np.abs(v32 - v64)
    tf = float(N) * N * M / (g32.timing()["predict_ms"] * 1e-3) / 1e12
    print(f"{str(var):110s} max rel {dv.max() / v64.max():.3e}  rms rel {np.sqrt((dv ** 2).mean()) / v64.max():.3e}  "
          f"predict {g32.timing()['predict_ms']:.0f} ms ({tf:.1f} TFLOP/s)", flush=True)
