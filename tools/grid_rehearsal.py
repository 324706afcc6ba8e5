"""Rehearsal of the C++ multi-GPU schedules at a size where panels and updates are no longer toy-sized: W ranks on THIS
box's one GPU (HIP engine, gloo transport staged through the host), 1-D block-column-cyclic and Pr x Pc grids, nlZ and
alpha against a single context.  Usage: python tools/grid_rehearsal.py [N] [nb]   (through gpurun; <= 4 ranks)"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gp_ss_ak_amd import gpak, synth  # noqa: E402


def free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
NB = int(sys.argv[2]) if len(sys.argv) > 2 else 512
X, y = synth.drillholes(N)
g = gpak.Gpak(0)
g.set_train(X, y)
g.set_params(np.array(synth.DEFAULT_EXPANS), synth.DEFAULT_BIAS, synth.DEFAULT_SN2, gpak.DIST_DIRECT)
ref_nlz, ref_alpha = g.logLikelihood(), g.solve_alpha()
print(f"N={N}: single context nlz {ref_nlz:.9f}, factor {g.timing()['factor_ms']:.2f} ms", flush=True)
g.close()
CASES = ((4, None), (4, (2, 2)), (4, (4, 1)), (3, None), (2, (2, 1)))
if os.environ.get("REHEARSE_1D"):
    CASES = ((4, None), (4, None), (2, None), (3, None), (4, None))
for world, grid in CASES:
    port = free_port()
    with tempfile.TemporaryDirectory() as d:
        procs = []
        for r in range(world):
            cmd = [sys.executable, os.path.join(ROOT, "tests", "dist_cpp_worker.py"), "--rank", str(r), "--world", str(world),
                   "--port", str(port), "--n", str(N), "--nb", str(NB), "--engine", "hip", "--mode", "1", "--steps", "2",
                   "--grad", "0", "--corrupt", "0", "--out", os.path.join(d, f"r{r}.json")]
            if grid:
                cmd += ["--grid", f"{grid[0]}x{grid[1]}"]
            procs.append(subprocess.Popen(cmd, env=dict(os.environ, OMP_NUM_THREADS="2"), stdout=subprocess.PIPE,
                                          stderr=subprocess.STDOUT))
        outs = [p.communicate(timeout=900)[0].decode() for p in procs]
        for p, o in zip(procs, outs):
            assert p.returncode == 0, o[-2000:]
        res = [json.load(open(os.path.join(d, f"r{r}.json"))) for r in range(world)]
    nl = [r["nlz"] for r in res]
    da = max(np.abs(np.array(r["alpha"]) - ref_alpha).max() for r in res) / np.abs(ref_alpha).max()
    st = res[0]["stats"]
    print(f"world {world} layout {'1-D' if not grid else f'{grid[0]}x{grid[1]}'}: nlz rel diff {max(abs(v - ref_nlz) for v in nl) / abs(ref_nlz):.1e} "
          f"(identical on all ranks: {len(set(nl)) == 1}), alpha {da:.1e}, factor {st.get('factor_ms', 0):.1f} ms, "
          f"broadcast {st.get('bytes_broadcast', 0) / 1e6:.0f} MB; per rank factor_ms {[round(r['stats'].get('factor_ms', 0), 1) for r in res]} "
          f"bulk_ms {[round(r['stats'].get('bulk_ms', 0), 1) for r in res]} comm_ms {[round(r['stats'].get('comm_ms', 0), 1) for r in res]} chain_ms {[round(r['stats'].get('chain_ms', 0), 1) for r in res]}", flush=True)
