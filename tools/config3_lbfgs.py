"""BASELINE.json configs[2]: N=32768 fp64, full L-BFGS hyper-parameter loop through the CLI
(per-evaluation Gram rebuild + Cholesky; gradient evaluations add the B^-1 build)."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from gp_ss_ak_amd import synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
d = sys.argv[3] if len(sys.argv) > 3 else "/tmp/gpak_cfg3"
os.makedirs(d, exist_ok=True)
Xr, yr = synth.drillholes_raw(N)
with open(os.path.join(d, "train.txt"), "w") as f:
    for r, v in zip(Xr, yr):
        f.write("\t".join(f"{t:.17g}" for t in list(r) + [v]) + "\n")
exe = os.path.join(ROOT, "gp_ss_ak_amd", "host", "gp_ss_ak")
env = dict(os.environ, GPAK_MAX_ITERS=str(iters))
t0 = time.perf_counter()
out = subprocess.check_output([exe, "-v", "1", "-np", "train", "-k", "ExpAns", "-kn", "1", "-o", "LBFGS",
                               os.path.join(d, "train.txt"), os.path.join(d, "model")], env=env, cwd=d).decode()
dt = time.perf_counter() - t0
its = [l for l in out.splitlines() if l.startswith("Iteration")]
print("\n".join(its))
print([l for l in out.splitlines() if "Error" in l or "Log likelihood" in l])
print(f"N={N}: {len(its)} L-BFGS iterations, train verb wall {dt:.1f} s (incl. csv read, final Calc_Out on N points)")
