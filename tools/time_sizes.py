"""Step time (Gram fill + factorisation + solves + nlZ) at several N on one GPU: python tools/time_sizes.py [N ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gp_ss_ak_amd import gpak, synth
import bench

for N in [int(a) for a in sys.argv[1:]] or [8192, 16384, 32768]:
    X, y = synth.drillholes(N)
    g = gpak.Gpak(0)
    g.set_option(gpak.OPT_PROFILE, int(os.environ.get('PROFILE', '1')))
    if os.environ.get('NB_OUTER'):
        g.set_option(gpak.OPT_NB_OUTER, int(os.environ['NB_OUTER']))
    g.set_train(X, y)
    steps = 20 if N <= 16384 else 8
    wall, nlz, ph, _ = bench._timed_steps(g, gpak.DIST_DIRECT, steps, 3)
    print(f"N={N}: {wall / steps * 1e3:.3f} ms/step  factor {ph['factor_ms'] / steps:.3f}  gram {ph['gram_ms'] / steps:.3f}  "
          f"solve {ph['solve_ms'] / steps:.3f}  kmatvec {ph['kmatvec_ms'] / steps:.3f}  nlZ {nlz:.6f}", flush=True)
    del g
