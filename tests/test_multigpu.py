"""The round-1 Python schedule tests/py_schedule.py DistGP under torch.distributed -- kept as a second, independent
implementation of the multi-GPU schedule (the product path is the C++ one: tests/test_dist_cpp.py).

CPU (-m "not gpu"): world_size 2 and 3 over gloo with the NumPy stand-in engine -- checks the
block-column-cyclic ownership, the look-ahead schedule, the panel broadcast and the solve
collectives against the single-process oracle.
GPU (-m gpu): the same schedule with the real HIP tile engine (gpak_dev_* C-ABI), world_size 1 and
world_size 2 with both ranks on the one GPU of the box (gloo stages the collectives).
"""
import json
import os
import socket
import subprocess
import sys
import tempfile

import numpy as np
import pytest

from gp_ss_ak_amd import synth

HERE = os.path.dirname(os.path.abspath(__file__))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def run_world(world, n, nb, engine="numpy", mode=1, sn2=None, timeout=600, pipeline=1):
    port = free_port()
    with tempfile.TemporaryDirectory() as d:
        procs = []
        for r in range(world):
            cmd = [sys.executable, os.path.join(HERE, "dist_worker.py"), "--rank", str(r), "--world", str(world),
                   "--port", str(port), "--n", str(n), "--nb", str(nb), "--engine", engine, "--mode", str(mode),
                   "--pipeline", str(pipeline), "--out", os.path.join(d, f"r{r}.json")]
            if sn2 is not None:
                cmd += ["--sn2", str(sn2)]
            env = dict(os.environ, OMP_NUM_THREADS="2")
            procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
        outs = [p.communicate(timeout=timeout)[0].decode() for p in procs]
        for p, o in zip(procs, outs):
            assert p.returncode == 0, o[-3000:]
        return [json.load(open(os.path.join(d, f"r{r}.json"))) for r in range(world)]


def oracle_ref(orc, n, mode=1):
    X, y = synth.drillholes(n)
    K = orc.gram(X, X, synth.DEFAULT_EXPANS, synth.DEFAULT_BIAS, mode)
    info, alpha, _ = orc.nlz_refseq(K, y, synth.DEFAULT_SN2)
    return info, alpha


@pytest.mark.parametrize("world,n,nb,pipeline", [(2, 300, 128, 1), (3, 700, 128, 1), (2, 600, 256, 1),
                                                  (3, 1500, 256, 1), (4, 1400, 256, 1), (2, 600, 256, 0),
                                                  (3, 700, 128, 0)])
def test_gloo_cpu_schedule_matches_oracle(orc, world, n, nb, pipeline):
    """pipeline=1: sub-panel broadcasts (the default schedule); 0: one broadcast per outer panel."""
    res = run_world(world, n, nb, pipeline=pipeline)
    info, alpha = oracle_ref(orc, n)
    owned = sorted(b for r in res for b in r["owned"])
    assert owned == list(range(res[0]["nJ"]))                       # every block column has one owner
    for r in res:
        assert r["owned"] == [b for b in range(r["nJ"]) if b % world == r["rank"]]
        assert abs(r["nlz"] - info.nlz) <= 1e-9 * abs(info.nlz)     # same value on every rank
        assert abs(r["logdet"] - info.logdet) <= 1e-10 * abs(info.logdet)
        assert np.abs(np.array(r["alpha"]) - alpha).max() <= 1e-8 * np.abs(alpha).max()
    assert res[0]["bytes_broadcast"] == res[1]["bytes_broadcast"] > 0


def test_gloo_cpu_chol_fail_is_nan_on_every_rank():
    res = run_world(2, 300, 128, sn2=-0.5)
    assert all(r["nlz"] != r["nlz"] for r in res)                   # NaN (GP_Utils.cpp:1145-1146)


def test_single_process_schedule_matches_oracle(orc):
    res = run_world(1, 260, 128)
    info, alpha = oracle_ref(orc, 260)
    assert abs(res[0]["nlz"] - info.nlz) <= 1e-9 * abs(info.nlz)


@pytest.mark.gpu
@pytest.mark.parametrize("world,n,nb,pipeline", [(1, 1000, 256, 1), (2, 1500, 256, 1), (2, 1100, 512, 1),
                                                  (3, 2500, 256, 1), (2, 1500, 256, 0), (2, 5000, 512, 1),
                                                  (4, 6000, 512, 1), (3, 4100, 512, 0)])
def test_hip_engine_schedule_matches_oracle(orc, world, n, nb, pipeline):
    res = run_world(world, n, nb, engine="hip", pipeline=pipeline)
    info, alpha = oracle_ref(orc, n)
    for r in res:
        assert abs(r["nlz"] - info.nlz) <= 1e-9 * abs(info.nlz)
        assert np.abs(np.array(r["alpha"]) - alpha).max() <= 1e-8 * np.abs(alpha).max()


@pytest.mark.gpu
def test_hip_engine_expansion_mode_and_chol_fail(orc):
    res = run_world(2, 900, 256, engine="hip", mode=0)
    info, alpha = oracle_ref(orc, 900, mode=0)
    assert abs(res[0]["nlz"] - info.nlz) <= 1e-6 * abs(info.nlz)    # expansion mode: cancellation noise
    res = run_world(2, 900, 256, engine="hip", sn2=-0.5)
    assert all(r["nlz"] != r["nlz"] for r in res)
