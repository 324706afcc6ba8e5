"""Worker of test_factor_against_the_vendor_cholesky: torch FIRST (its HIP runtime), then the library; the factor and the
log-determinant of B = I + K / sn2 from both, one JSON line."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gp_ss_ak_amd import gpak, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
E, BIAS, SN2 = np.array(synth.DEFAULT_EXPANS), synth.DEFAULT_BIAS, synth.DEFAULT_SN2
X, y = synth.drillholes(n)
g = gpak.Gpak(0)
try:
    g.set_train(X, y)
    g.set_params(E, BIAS, SN2, gpak.DIST_DIRECT)
    K = g.gram()
    g.logLikelihood()
    R = g.chol_upper()                 # the reference keeps the upper factor: B = R^T R
    ld = g.nlz_terms()[2]
finally:
    g.close()
B = torch.from_numpy(np.ascontiguousarray(K)).cuda() / SN2 + torch.eye(n, dtype=torch.float64, device="cuda")
L = torch.linalg.cholesky(B).cpu().numpy()
print(json.dumps({"n": n, "factor_max_abs_diff": float(np.abs(np.triu(R) - L.T).max()), "factor_max_abs": float(np.abs(L).max()),
                  "logdet_library": float(ld), "logdet_vendor": float(np.log(np.diag(L)).sum())}), flush=True)
