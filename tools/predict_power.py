"""Package power and shader clock (rocm-smi, 20 samples/s) while the fp32 and the fp64 prediction run (N=32768,
M=262144): is the fp32 GEMM's 77 % of its MFMA peak a power limit like the fp64 factorisation's?
  python tools/predict_power.py"""
import json, os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gp_ss_ak_amd import gpak, synth

N, M = 32768, 262144
X, y = synth.drillholes(N)
Xt = synth.test_points(M)
samples, stop = [], False


def smi():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "-P", "-c", "--json"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                                 timeout=5).stdout.decode()
            c0 = json.loads(out).get("card0", {})
            samples.append((time.perf_counter(), {k: v for k, v in c0.items() if "ower" in k or "sclk" in k}))
        except Exception as e:  # noqa
            samples.append((time.perf_counter(), {"error": str(e)[:80]}))
        time.sleep(0.05)


threading.Thread(target=smi, daemon=True).start()
for name, prec in (("f32", gpak.F32), ("f64", gpak.F64)):
    g = gpak.Gpak(0, prec)
    g.set_train(X, y)
    g.set_params(synth.DEFAULT_EXPANS, synth.DEFAULT_BIAS, synth.DEFAULT_SN2, gpak.DIST_DIRECT)
    g.logLikelihood()
    g.posteriorMeanVar(Xt[:65536].copy(order="F"))
    t0 = time.perf_counter()
    g.posteriorMeanVar(Xt)
    t1 = time.perf_counter()
    busy = [s for t, s in samples if t0 + 0.3 <= t <= t1]
    print(f"{name}: {float(N) * N * M / (t1 - t0) / 1e12:.1f} TFLOP/s over {t1 - t0:.2f} s; rocm-smi during it:", busy[:6], flush=True)
    g.close()
stop = True
