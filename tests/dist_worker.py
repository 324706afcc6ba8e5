"""One rank of a multi-process DistGP run (launched by tests/test_multigpu.py)."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rank", type=int, required=True)
    ap.add_argument("--world", type=int, required=True)
    ap.add_argument("--port", type=int, required=True)
    ap.add_argument("--n", type=int, default=300)
    ap.add_argument("--nb", type=int, default=128)
    ap.add_argument("--engine", choices=["numpy", "hip"], default="numpy")
    ap.add_argument("--mode", type=int, default=1)
    ap.add_argument("--sn2", type=float, default=None)
    ap.add_argument("--pipeline", type=int, default=1)
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(a.port)
    dist.init_process_group("gloo", rank=a.rank, world_size=a.world)
    import py_schedule as multigpu
    from gp_ss_ak_amd import synth
    if a.engine == "hip":
        eng = multigpu.HipEngine(0)  # every rank on GPU 0: a schedule test, collectives staged by gloo
    else:
        from np_engine import NumpyEngine
        eng = NumpyEngine()
    X, y = synth.drillholes(a.n)
    gp = multigpu.DistGP(eng, X, y, nb=a.nb, pipeline=bool(a.pipeline))
    sn2 = synth.DEFAULT_SN2 if a.sn2 is None else a.sn2
    gp.set_params(synth.DEFAULT_EXPANS, synth.DEFAULT_BIAS, sn2, a.mode)
    nlz = gp.nlz()
    res = {"rank": a.rank, "nlz": nlz, "owned": gp.owned, "nJ": gp.nJ, "bytes_broadcast": gp.bytes_broadcast}
    if nlz == nlz:
        res.update({"alpha": gp.get_alpha().tolist(), "logdet": gp.logdet, "quad": gp.quad, "sumlp": gp.sumlp})
    json.dump(res, open(a.out, "w"))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
