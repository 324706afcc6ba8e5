"""Reference point from the vendor stack on the same box: torch.linalg.cholesky (fp64, ROCm: hipSOLVER / rocSOLVER dpotrf)
on the SAME matrix B = I + K / sn2 that the library factors, against the library's factorisation time.
Usage: python tools/vendor_potrf.py [--json] [N ...]   (through gpurun; --json: one JSON line per size, what bench.py reads)"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gp_ss_ak_amd import gpak, synth  # noqa: E402

E = np.array(synth.DEFAULT_EXPANS)
JSON = "--json" in sys.argv
for N in [int(a) for a in sys.argv[1:] if a != "--json"] or [2048, 8192, 16384, 32768]:
    X, y = synth.drillholes(N)
    g = gpak.Gpak(0)
    g.set_train(X, y)
    g.set_params(E, synth.DEFAULT_BIAS, synth.DEFAULT_SN2, gpak.DIST_DIRECT)
    K = g.gram()
    K = K[0] if isinstance(K, tuple) else K
    fac = []
    for i in range(6):
        g.set_params(E, synth.DEFAULT_BIAS, synth.DEFAULT_SN2 * (1 + 1e-3 * (i % 3)), gpak.DIST_DIRECT)
        g.logLikelihood()
        fac.append(g.timing()["factor_ms"])
    g.set_params(E, synth.DEFAULT_BIAS, synth.DEFAULT_SN2, gpak.DIST_DIRECT)
    g.logLikelihood()
    ld_lib = g.nlz_terms()[2]
    g.close()
    B = torch.from_numpy(np.ascontiguousarray(K)).cuda()
    B = B / synth.DEFAULT_SN2 + torch.eye(N, dtype=torch.float64, device="cuda")
    ts = []
    for i in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        L = torch.linalg.cholesky(B)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    ld_t = float(torch.log(torch.diagonal(L)).sum())
    fl = N ** 3 / 3.0
    if JSON:
        print(json.dumps({"N": N, "vendor_ms": min(ts[1:]), "vendor_tflops": fl / min(ts[1:]) / 1e9, "library_factor_ms": min(fac[2:]),
                          "library_tflops": fl / min(fac[2:]) / 1e9, "library_speedup": min(ts[1:]) / min(fac[2:]),
                          "sum_log_diag": {"vendor": ld_t, "library": ld_lib}}), flush=True)
        del B, L
        torch.cuda.empty_cache()
        continue
    print(f"N={N}: library factorisation {min(fac[2:]):8.3f} ms = {fl / min(fac[2:]) / 1e9:6.1f} TFLOP/s | torch.linalg.cholesky "
          f"{min(ts[1:]):8.3f} ms = {fl / min(ts[1:]) / 1e9:6.1f} TFLOP/s | ratio {min(ts[1:]) / min(fac[2:]):.2f} | "
          f"sum log diag: library {ld_lib:.9f} torch {ld_t:.9f}", flush=True)
    del B, L
    torch.cuda.empty_cache()
