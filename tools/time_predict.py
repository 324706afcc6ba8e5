"""BASELINE.json configs[4]: N=32768, M=1e6 block-model points, GPAK_F32 prediction (and the fp64 context beside it).
Writes one JSON record (profiles/<tag>_config5_predict_M1e6.json when run as  python tools/time_predict.py <tag>)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gp_ss_ak_amd import gpak, synth
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
N, M = 32768, int(float(sys.argv[2])) if len(sys.argv) > 2 else 1000000
X, y = synth.drillholes(N)
Xt = synth.test_points(M)
rec = {"N": N, "M": M, "test_points": "synth.test_points(1e6): regular 100^3 block model in the standardised cube"}
ref = None
for name, prec in (("f32", gpak.F32), ("f64", gpak.F64)):
    g = gpak.Gpak(0, prec)
    g.set_train(X, y)
    g.set_params(synth.DEFAULT_EXPANS, synth.DEFAULT_BIAS, synth.DEFAULT_SN2, gpak.DIST_DIRECT)
    g.logLikelihood()
    g.posteriorMeanVar(Xt[:65536].copy(order="F"))          # warm-up: buffers, fp32 image of the factor
    t0 = time.perf_counter()
    mean, var = g.posteriorMeanVar(Xt)
    wall = time.perf_counter() - t0
    t0 = time.perf_counter()
    g.posteriorMeanVar(Xt, want_var=False)
    wall_mean = time.perf_counter() - t0
    dev_ms = g.timing()["predict_ms"]
    rec[name] = {"wall_s": wall, "points_per_s": M / wall, "mean_only_wall_s": wall_mean,
                 "variance_tflops": float(N) * N * M / wall / 1e12,
                 "frac_of_mfma_peak": float(N) * N * M / wall / 1e12 / (157.3 if name == "f32" else 78.6)}
    if ref is None:
        ref = (mean, var)
    else:
        rec["f32_vs_f64"] = {"variance_max_rel": float(np.abs(ref[1] - var).max() / var.max()),
                             "mean_max_rel": float(np.abs(ref[0] - mean).max() / np.abs(mean).max())}
    g.close()
print(json.dumps(rec, indent=1))
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", f"{tag}_config5_predict_M{M}.json")
os.makedirs(os.path.dirname(out), exist_ok=True)
json.dump(rec, open(out, "w"), indent=1)
