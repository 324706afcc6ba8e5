"""One rank of a multi-process run of the C++ schedule (csrc/dist.hip), launched by tests/test_dist_cpp.py."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

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
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--grad", type=int, default=0)
    ap.add_argument("--corrupt", type=int, default=0)
    ap.add_argument("--hyb", type=int, default=0, help="1: ExpAns + Exp + RBF + Bias + White through gpak_dist_set_kernel")
    ap.add_argument("--d", type=int, default=3, help="input columns (4: with a rock-type column)")
    ap.add_argument("--grid", default="", help="PrxPc: the row-block x column-block layout (gpak_grid_*)")
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(a.port)
    dist.init_process_group("gloo", rank=a.rank, world_size=a.world)
    from gp_ss_ak_amd import dist as gd, synth
    eng = None
    grid = tuple(int(v) for v in a.grid.split("x")) if a.grid else None
    if grid is not None:
        if a.engine == "hip":
            gp = gd.GridRank(a.rank, a.world, grid[0], grid[1], device=0, transport=gd.StagedTransport())
        else:
            from np_dist_engine import GlooTransport, NumpyDistEngine
            eng, tr = NumpyDistEngine(), GlooTransport()
            gp = gd.GridRank(a.rank, a.world, grid[0], grid[1], engine=eng, transport=tr)
    elif a.engine == "hip":
        # every rank on GPU 0: the built-in HIP engine with the collectives staged through gloo (RCCL refuses two
        # ranks on one device); world 1 uses the built-in RCCL transport (a no-op communicator-free path)
        tr = gd.StagedTransport() if a.world > 1 else None
        gp = gd.DistRank(a.rank, a.world, device=0, transport=tr)
    else:
        from np_dist_engine import GlooTransport, NumpyDistEngine
        eng, tr = NumpyDistEngine(), GlooTransport(corrupt_first=a.corrupt)
        gp = gd.DistRank(a.rank, a.world, engine=eng, transport=tr)
    X, y = synth.drillholes4(max(a.n, 4)) if a.d == 4 else synth.drillholes(max(a.n, 4))
    X, y = X[:a.n].copy(order="F"), y[:a.n].copy()
    gp.set_train(X, y, nb=a.nb)
    sn2 = synth.DEFAULT_SN2 if a.sn2 is None else a.sn2
    res = {"rank": a.rank}
    for s in range(a.steps):          # repeated steps reuse the receive buffers: the second one must not race
        e = list(synth.DEFAULT_EXPANS)
        e[1] += 0.01 * (a.steps - 1 - s)
        if a.hyb:
            gp.set_kernel([(0, e), (1, [0.5, 0.9]), (2, [0.5, 0.9, 0.5])], synth.DEFAULT_BIAS, 0.10, sn2, a.mode)
        else:
            gp.set_params(e, synth.DEFAULT_BIAS, sn2, a.mode)
        nlz = gp.nlz()
    st = gp.stats()
    res.update({"nlz": nlz, "stats": st})
    if nlz == nlz:
        q, s, l = gp.nlz_terms()
        res.update({"alpha": gp.get_alpha().tolist(), "logdet": l, "quad": q, "sumlp": s})
        if a.grad and grid is None:
            res["grad"] = gp.grad().tolist()
    if eng is not None:
        # schedule facts the tests assert: which stream every factor / update was issued on
        res["calls"] = [(op, int(arg), int(st_)) for op, arg, st_ in eng.calls]
    json.dump(res, open(a.out, "w"))
    gp.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
