"""The N>1 path in C++: csrc/dist.hip (include/gpak_dist.h) driven one rank per process.

CPU (-m "not gpu"): world_size 1..4 over gloo; the schedule runs in libgpak_hip.so, its tile operations are the
NumPy callback engine of tests/np_dist_engine.py and its collectives a gloo callback transport -- ownership map,
look-ahead order, sub-panel broadcasts, solves on the packed panels and the reductions against the single-process
oracle.
GPU (-m gpu): the same schedule with the built-in HIP engine; world 1 on the built-in RCCL transport, world 2..4 with
all ranks on the one GPU of the box and the collectives staged through gloo (RCCL refuses two ranks per device);
and bench.py --gpus as the driver launches it.
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
ROOT = os.path.dirname(HERE)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def run_world(world, n, nb, engine="numpy", mode=1, sn2=None, steps=1, timeout=900, grad=0, corrupt=0, env=None, grid=None,
              ncols=3, hyb=0):
    port = free_port()
    with tempfile.TemporaryDirectory() as d:
        procs = []
        for r in range(world):
            cmd = [sys.executable, os.path.join(HERE, "dist_cpp_worker.py"), "--rank", str(r), "--world", str(world),
                   "--port", str(port), "--n", str(n), "--nb", str(nb), "--engine", engine, "--mode", str(mode),
                   "--steps", str(steps), "--grad", str(grad), "--corrupt", str(corrupt), "--out",
                   os.path.join(d, f"r{r}.json")]
            if sn2 is not None:
                cmd += ["--sn2", str(sn2)]
            if grid is not None:
                cmd += ["--grid", f"{grid[0]}x{grid[1]}"]
            if ncols != 3:
                cmd += ["--d", str(ncols)]
            if hyb:
                cmd += ["--hyb", "1"]
            penv = dict(os.environ, OMP_NUM_THREADS="2", **(env or {}))
            procs.append(subprocess.Popen(cmd, env=penv, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
        outs = [p.communicate(timeout=timeout)[0].decode() for p in procs]
        for p, o in zip(procs, outs):
            assert p.returncode == 0, o[-3000:]
        return [json.load(open(os.path.join(d, f"r{r}.json"))) for r in range(world)]


def oracle_ref(orc, n, mode=1, e1=None):
    X, y = synth.drillholes(n)
    e = list(synth.DEFAULT_EXPANS)
    K = orc.gram(X, X, e, synth.DEFAULT_BIAS, mode)
    info, alpha, _ = orc.nlz_refseq(K, y, synth.DEFAULT_SN2)
    return info, alpha


@pytest.mark.parametrize("world,n,nb", [(1, 260, 128), (2, 300, 128), (3, 700, 128), (2, 600, 256), (3, 1500, 256),
                                        (4, 1400, 256), (2, 1100, 512), (4, 2100, 512)])
def test_cpp_schedule_over_gloo_matches_oracle(orc, world, n, nb):
    res = run_world(world, n, nb)
    info, alpha = oracle_ref(orc, n)
    nJ = res[0]["stats"]["n_panels"]
    assert nJ == -(-(-(-n // 128) * 128) // nb)
    for r in res:
        assert r["stats"]["world"] == world and r["stats"]["rank"] == r["rank"]
        assert abs(r["nlz"] - info.nlz) <= 1e-9 * abs(info.nlz)     # same value on every rank
        assert abs(r["logdet"] - info.logdet) <= 1e-10 * abs(info.logdet)
        assert np.abs(np.array(r["alpha"]) - alpha).max() <= 1e-8 * np.abs(alpha).max()
        # ownership: block column b is factored on rank b % world only, every 128-column sub-panel once, in order
        fac = [J for op, J, st in r["calls"] if op == "factor"]
        want = [b * nb + s for b in range(nJ) if b % world == r["rank"]
                for s in range(0, min(nb, -(-n // 128) * 128 - b * nb), 128)]
        assert fac == want
        # look-ahead: factor / in-column updates on the panel stream (101), bulk updates on the bulk stream (100)
        assert all(st == 101 for op, J, st in r["calls"] if op in ("factor", "update_block"))
        assert all(st == 100 for op, J, st in r["calls"] if op == "update_cyclic")
    assert len({r["stats"]["bytes_broadcast"] for r in res}) == 1 and res[0]["stats"]["bytes_broadcast"] > 0


@pytest.mark.parametrize("world,n,nb,mode", [(1, 300, 128, 1), (2, 700, 256, 1), (3, 1000, 256, 1), (4, 1400, 512, 0)])
def test_cpp_distributed_gradient_over_gloo_matches_oracle(orc, world, n, nb, mode):
    """gpak_dist_grad: rows of L^-T from the packed panels, all-gather, rows of B^-1, pair pass, 16-double all-reduce
    -- against GradLL + getGradients as written in the oracle (orc_grad_ref)."""
    res = run_world(world, n, nb, grad=1, mode=mode)
    X, y = synth.drillholes(n)
    e = np.array(synth.DEFAULT_EXPANS)
    K = orc.gram(X, X, e, synth.DEFAULT_BIAS, mode)
    info, alpha, L = orc.nlz_lean(K, y, synth.DEFAULT_SN2)
    go = orc.grad_ref(X, y, K, L, alpha, e, synth.DEFAULT_BIAS, synth.DEFAULT_SN2, mode)
    tol = 1e-8 if mode == 1 else 1e-5
    for r in res:
        assert np.abs(np.array(r["grad"]) - go).max() <= tol * np.abs(go).max(), (r["rank"], r["grad"], go.tolist())
    assert all(r["grad"] == res[0]["grad"] for r in res)            # bit-identical on every rank


@pytest.mark.parametrize("world,n,nb", [(4, 300, 128), (3, 130, 128), (4, 500, 512), (2, 1, 128)])
def test_ranks_that_own_nothing(orc, world, n, nb):
    """Fewer block columns (and 128-row blocks) than ranks: the idle ranks still take part in every collective and
    report the same nlZ / alpha / gradient."""
    res = run_world(world, n, nb, grad=1)
    X, y = synth.drillholes(max(n, 4))
    X, y = np.asfortranarray(X[:n]), y[:n].copy()
    e = np.array(synth.DEFAULT_EXPANS)
    K = orc.gram(X, X, e, synth.DEFAULT_BIAS, 1)
    info, alpha, L = orc.nlz_lean(K, y, synth.DEFAULT_SN2)
    go = orc.grad_ref(X, y, K, L, alpha, e, synth.DEFAULT_BIAS, synth.DEFAULT_SN2, 1)
    for r in res:
        assert abs(r["nlz"] - info.nlz) <= 1e-9 * max(1.0, abs(info.nlz))
        assert np.abs(np.array(r["alpha"]) - alpha).max() <= 1e-8 * np.abs(alpha).max()
        assert np.abs(np.array(r["grad"]) - go).max() <= 1e-8 * max(1.0, np.abs(go).max())


def test_cpp_schedule_repeated_steps_and_expansion_mode(orc):
    res = run_world(3, 900, 256, steps=3, mode=0)
    info, alpha = oracle_ref(orc, 900, mode=0)
    for r in res:
        assert abs(r["nlz"] - info.nlz) <= 1e-6 * abs(info.nlz)     # expansion mode: cancellation noise


def test_selfcheck_failure_switches_every_rank_to_plain_streams(orc):
    """A collective that delivers wrong data during the start-up self-check (here: the first broadcast, corrupted on the
    receivers) must be caught BEFORE the first step: every rank -- also the root, which saw nothing wrong -- falls
    back to plain streams with the collectives in line (flags 1|2), re-checks, and then computes the right answer."""
    res = run_world(3, 700, 128, corrupt=1)
    info, alpha = oracle_ref(orc, 700)
    for r in res:
        assert r["stats"]["flags"] == 3, r["stats"]
        assert abs(r["nlz"] - info.nlz) <= 1e-9 * abs(info.nlz)
        # in-line collectives: every factor / update of the panel chain still sits on the panel stream, and there is
        # no separate communication stream any more
        assert all(st == 101 for op, J, st in r["calls"] if op in ("factor", "update_block"))
    ok = run_world(3, 700, 128)
    assert all(r["stats"]["flags"] == 0 for r in ok)


def test_plain_stream_mode_can_be_forced(orc):
    res = run_world(2, 600, 256, env={"GPAK_DIST_PLAIN_STREAMS": "1"}, grad=1)
    info, alpha = oracle_ref(orc, 600)
    for r in res:
        assert r["stats"]["flags"] == 3 and abs(r["nlz"] - info.nlz) <= 1e-9 * abs(info.nlz)


def test_cpp_schedule_chol_fail_is_nan_on_every_rank():
    res = run_world(2, 300, 128, sn2=-0.5)
    assert all(r["nlz"] != r["nlz"] for r in res)                   # NaN (GP_Utils.cpp:1145-1146)


@pytest.mark.gpu
@pytest.mark.parametrize("world,n,nb", [(1, 1000, 256), (2, 1500, 256), (2, 1100, 512), (3, 2500, 256), (2, 5000, 512),
                                        (4, 6000, 512), (3, 4100, 384)])
def test_cpp_schedule_hip_engine_matches_oracle(orc, world, n, nb):
    res = run_world(world, n, nb, engine="hip", steps=2)
    info, alpha = oracle_ref(orc, n)
    for r in res:
        assert abs(r["nlz"] - info.nlz) <= 1e-9 * abs(info.nlz)
        assert np.abs(np.array(r["alpha"]) - alpha).max() <= 1e-8 * np.abs(alpha).max()
        assert r["stats"]["bulk_flops"] >= 0 and r["stats"]["factor_ms"] > 0


@pytest.mark.gpu
@pytest.mark.parametrize("world,n,nb", [(1, 1000, 256), (2, 1500, 256), (3, 2500, 512), (4, 5000, 512), (4, 300, 128)])
def test_cpp_distributed_gradient_hip_engine_matches_oracle(orc, world, n, nb):
    res = run_world(world, n, nb, engine="hip", grad=1)
    X, y = synth.drillholes(n)
    e = np.array(synth.DEFAULT_EXPANS)
    K = orc.gram(X, X, e, synth.DEFAULT_BIAS, 1)
    info, alpha, L = orc.nlz_lean(K, y, synth.DEFAULT_SN2)
    go = orc.grad_ref(X, y, K, L, alpha, e, synth.DEFAULT_BIAS, synth.DEFAULT_SN2, 1)
    for r in res:
        assert np.abs(np.array(r["grad"]) - go).max() <= 1e-8 * np.abs(go).max(), (r["rank"], r["grad"], go.tolist())


@pytest.mark.gpu
def test_cpp_schedule_hip_engine_plain_stream_fallback(orc):
    """The fallback the self-check selects (ordinary bulk stream, collectives in line on the panel stream) with the HIP
    engine: same numbers."""
    res = run_world(3, 2500, 256, engine="hip", env={"GPAK_DIST_PLAIN_STREAMS": "1"}, grad=1, steps=2)
    info, alpha = oracle_ref(orc, 2500)
    for r in res:
        assert r["stats"]["flags"] == 3
        assert abs(r["nlz"] - info.nlz) <= 1e-9 * abs(info.nlz)
        assert np.abs(np.array(r["alpha"]) - alpha).max() <= 1e-8 * np.abs(alpha).max()


def _oracle_d4(orc, n, mode=1):
    X, y = synth.drillholes4(n)
    e = np.array(synth.DEFAULT_EXPANS)
    K = orc.gram(X, X, e, synth.DEFAULT_BIAS, mode)
    info, alpha, L = orc.nlz_lean(K, y, synth.DEFAULT_SN2)
    g = orc.grad_ref(X, y, K, L, alpha, e, synth.DEFAULT_BIAS, synth.DEFAULT_SN2, mode)
    return info, alpha, g


@pytest.mark.parametrize("world,n,nb,grid", [(2, 500, 128, None), (3, 900, 256, None), (4, 700, 128, (2, 2))])
def test_four_column_inputs_on_the_distributed_paths(orc, world, n, nb, grid):
    """SURVEY Q7 on more than one rank: x, y, z + rock type (InversewidthR in the distance, g[7] != 0) through the 1-D
    schedule (nlZ, alpha, gradient) and the 2-D grid (nlZ, alpha), NumPy engine over gloo, against the oracle."""
    res = run_world(world, n, nb, ncols=4, grid=grid, grad=0 if grid else 1)
    info, alpha, g = _oracle_d4(orc, n)
    for r in res:
        assert abs(r["nlz"] - info.nlz) <= 1e-9 * abs(info.nlz)
        assert np.abs(np.array(r["alpha"]) - alpha).max() <= 1e-8 * np.abs(alpha).max()
        if grid is None:
            assert g[7] != 0.0 and np.abs(np.array(r["grad"]) - g).max() <= 1e-8 * np.abs(g).max()


@pytest.mark.gpu
def test_four_column_inputs_on_a_multi_gpu_context(orc):
    """The same through gpak_create_multi on the HIP engine (three ranks on this box's GPU): nlZ, alpha, gradient with
    g[7], and the sharded prediction with a rock-type column in the test points."""
    from gp_ss_ak_amd import gpak
    n = 1500
    X, y = synth.drillholes4(n)
    Xt = synth.test_points4(300)
    e = np.array(synth.DEFAULT_EXPANS)
    info, alpha, go = _oracle_d4(orc, n)
    K = orc.gram(X, X, e, synth.DEFAULT_BIAS, orc.DIST_DIRECT)
    _, _, L = orc.nlz_lean(K, y, synth.DEFAULT_SN2)
    g = gpak.Gpak(devices=[0, 0, 0])
    try:
        g.set_train(X, y)
        g.set_params(e, synth.DEFAULT_BIAS, synth.DEFAULT_SN2, gpak.DIST_DIRECT)
        assert abs(g.logLikelihood() - info.nlz) <= 1e-9 * abs(info.nlz)
        assert np.abs(g.solve_alpha() - alpha).max() <= 1e-8 * np.abs(alpha).max()
        gg = g.GradLL()
        assert gg[7] != 0.0 and np.abs(gg - go).max() <= 1e-8 * np.abs(go).max()
        mean, var = g.posteriorMeanVar(Xt)
        mo, vo = orc.predict(X, Xt, e, synth.DEFAULT_BIAS, synth.DEFAULT_SN2, alpha, L, orc.DIST_DIRECT, 0)
        assert np.abs(mean - mo).max() <= 1e-8 * np.abs(mo).max() and np.abs(var - vo).max() <= 1e-8 * np.abs(vo).max()
    finally:
        g.close()


HYB_TERMS = [(0, list(synth.DEFAULT_EXPANS)), (1, [0.5, 0.9]), (2, [0.5, 0.9, 0.5])]


@pytest.mark.parametrize("world,n,nb,ncols", [(2, 500, 128, 3), (3, 800, 256, 3), (2, 600, 128, 4)])
def test_other_compositions_on_the_distributed_path(orc, world, n, nb, ncols):
    """HybKerns{ExpAns, Exp, RBF} + Bias + White on more than one rank (gpak_dist_set_kernel: the composition travels to the
    engine as one serialized array): nlZ, its terms and alpha against the oracle's gram_hyb; the fixed-length gradient is
    refused on that path."""
    res = run_world(world, n, nb, hyb=1, ncols=ncols)
    X, y = (synth.drillholes4(n) if ncols == 4 else synth.drillholes(n))
    K = orc.gram_hyb(X, X, HYB_TERMS, synth.DEFAULT_BIAS, 0.10, 1)
    info, alpha, _ = orc.nlz_lean(K, y, synth.DEFAULT_SN2)
    for r in res:
        assert abs(r["nlz"] - info.nlz) <= 1e-9 * abs(info.nlz), (r["nlz"], info.nlz)
        assert abs(r["logdet"] - info.logdet) <= 1e-10 * abs(info.logdet)
        assert np.abs(np.array(r["alpha"]) - alpha).max() <= 1e-8 * np.abs(alpha).max()


@pytest.mark.gpu
def test_other_compositions_on_a_multi_gpu_context(orc):
    """The same through gpak_create_multi (three ranks on this box's GPU): gpak_set_kernel on the group, nlZ / alpha
    distributed, the sharded prediction and the children's gradients (device 0) on the distributed factor."""
    import scipy.linalg as sl
    from gp_ss_ak_amd import gpak
    n, M = 1400, 200
    X, y = synth.drillholes(n)
    Xt = synth.test_points(M)
    E = np.array(synth.DEFAULT_EXPANS)
    terms = [(gpak.KERN_EXPANS, E), (gpak.KERN_EXP, [0.5, 0.9]), (gpak.KERN_RBF, [0.5, 0.9, 0.5])]
    g = gpak.Gpak(devices=[0, 0, 0])
    try:
        g.set_train(X, y)
        g.set_kernel(terms, synth.DEFAULT_BIAS, 0.10, synth.DEFAULT_SN2, gpak.DIST_DIRECT)
        Ko = orc.gram_hyb(X, X, terms, synth.DEFAULT_BIAS, 0.10, orc.DIST_DIRECT)
        info, alpha, _ = orc.nlz_lean(Ko, y, synth.DEFAULT_SN2)
        assert abs(g.logLikelihood() - info.nlz) <= 1e-9 * abs(info.nlz)
        assert np.abs(g.solve_alpha() - alpha).max() <= 1e-8 * np.abs(alpha).max()
        mean, var = g.posteriorMeanVar(Xt)
        kX = orc.gram_hyb(X, Xt, terms, synth.DEFAULT_BIAS, 0.10, orc.DIST_DIRECT)
        cf = sl.cho_factor(Ko + synth.DEFAULT_SN2 * np.eye(n), lower=True)
        kD = E[6] ** 2 + 0.9 ** 2 + 0.5 ** 2 + synth.DEFAULT_BIAS + 0.10
        vo = np.maximum(kD - np.einsum("ij,ij->j", kX, sl.cho_solve(cf, kX)), 0) + synth.DEFAULT_SN2
        assert np.abs(mean - kX.T @ alpha).max() <= 1e-8 * np.abs(kX.T @ alpha).max()
        assert np.abs(var - vo).max() <= 1e-8 * np.abs(vo).max()
        with pytest.raises(gpak.GpakError) as ei:
            g.GradLL()
        assert ei.value.status == gpak.ENOTIMPL
        assert g.timing()["evaluations"] == 1
        # without the White child the reference has gradients for every child: on device 0, from the distributed factor
        g.set_kernel(terms, synth.DEFAULT_BIAS, 0.0, synth.DEFAULT_SN2, gpak.DIST_DIRECT)
        gg = g.GradLL_hyb(8 + 2 + 3 + 2)
        K0 = orc.gram_hyb(X, X, terms, synth.DEFAULT_BIAS, 0.0, orc.DIST_DIRECT)
        i0, a0, L0 = orc.nlz_lean(K0, y, synth.DEFAULT_SN2)
        go = orc.grad_hyb(X, y, K0, L0, a0, terms, True, synth.DEFAULT_SN2, orc.DIST_DIRECT)
        assert np.abs(gg - go).max() <= 1e-8 * np.abs(go).max()
        assert g.timing()["evaluations"] == 2
        # and back to the default composition
        g.set_params(E, synth.DEFAULT_BIAS, synth.DEFAULT_SN2, gpak.DIST_DIRECT)
        Kd = orc.gram(X, X, E, synth.DEFAULT_BIAS, orc.DIST_DIRECT)
        idf, _, _ = orc.nlz_lean(Kd, y, synth.DEFAULT_SN2)
        assert abs(g.logLikelihood() - idf.nlz) <= 1e-9 * abs(idf.nlz)
    finally:
        g.close()


# ---- the row-block x column-block layout (gpak_grid_*, csrc/grid.inc) ----------------------------------------------
@pytest.mark.parametrize("grid,n,nb", [((2, 2), 700, 128), ((2, 3), 1300, 128), ((3, 2), 1500, 256), ((2, 2), 600, 256),
                                        ((2, 4), 1200, 128), ((4, 2), 1100, 128), ((2, 1), 500, 128), ((4, 1), 900, 128)])
def test_grid_layout_over_gloo_matches_oracle(orc, grid, n, nb):
    """north_star's 2-D sharding on a Pr x Pc process grid, the C++ schedule driven by the NumPy engine over gloo row /
    column / world groups: nlZ, its three terms and alpha against the oracle, identical on every rank; what a rank
    receives is the (1/Pr + 1/Pc) share of the panels, not all of them."""
    world = grid[0] * grid[1]
    res = run_world(world, n, nb, grid=grid, steps=2)
    info, alpha = oracle_ref(orc, n)
    for r in res:
        assert abs(r["nlz"] - info.nlz) <= 1e-9 * abs(info.nlz), (r["rank"], r["nlz"], info.nlz)
        assert abs(r["logdet"] - info.logdet) <= 1e-10 * abs(info.logdet)
        assert abs(r["quad"] - info.quad) <= 1e-9 * abs(info.quad) and abs(r["sumlp"] - info.sumlp) <= 1e-9 * abs(info.sumlp)
        assert np.abs(np.array(r["alpha"]) - alpha).max() <= 1e-8 * np.abs(alpha).max()
        assert r["nlz"] == res[0]["nlz"]
    # communication volume: a rank receives about (1/Pr + 1/Pc) of the lower triangle (its row pieces + its transposed
    # pieces + the diagonal blocks of its process column), where the 1-D layout sends it (P-1)/P of the triangle --
    # 2 x 2 is therefore WORSE than 1-D (1.0 vs 0.75), 2 x 4 / 4 x 2 better (0.75 vs 0.875): DESIGN.md section 5
    Np = res[0]["stats"]["n_padded"]
    whole = 8.0 * Np * (Np + nb) / 2
    for r in res:
        assert 0 < r["stats"]["bytes_broadcast"] <= 1.35 * whole * (1.0 / grid[0] + 1.0 / grid[1])
    # the bulk updates went to the bulk stream (kind 0 -> 100), the look-ahead column to the panel stream (101)
    streams = {st for op, _, st in res[0]["calls"] if op == "update_rect"}
    assert streams <= {100, 101} and 100 in streams


def test_grid_layout_chol_fail_and_expansion_mode(orc):
    res = run_world(4, 500, 128, grid=(2, 2), sn2=-0.5)
    assert all(r["nlz"] != r["nlz"] for r in res)
    res = run_world(4, 500, 128, grid=(2, 2), mode=0)
    info, alpha = oracle_ref(orc, 500, mode=0)
    assert all(abs(r["nlz"] - info.nlz) <= 1e-6 * abs(info.nlz) for r in res)


@pytest.mark.gpu
# (at most 4 ranks: the box allows 6 processes on its GPU, and the test runner itself holds it too)
@pytest.mark.parametrize("grid,n,nb", [((2, 2), 3000, 512), ((2, 2), 1300, 256), ((4, 1), 5000, 512), ((2, 2), 5000, 384)])
def test_grid_layout_hip_engine_matches_oracle(orc, grid, n, nb):
    """The same on the HIP engine (gpak_dev_fill_rect / solve_rows / update_rect / gemv kernels), Pr x Pc ranks rehearsed on
    this box's one GPU with the collectives staged through gloo."""
    res = run_world(grid[0] * grid[1], n, nb, engine="hip", grid=grid, steps=2)
    info, alpha = oracle_ref(orc, n)
    for r in res:
        assert abs(r["nlz"] - info.nlz) <= 1e-9 * abs(info.nlz), (r["rank"], r["nlz"], info.nlz)
        assert abs(r["logdet"] - info.logdet) <= 1e-10 * abs(info.logdet)
        assert np.abs(np.array(r["alpha"]) - alpha).max() <= 1e-8 * np.abs(alpha).max()


@pytest.mark.gpu
def test_cpp_schedule_hip_engine_chol_fail(orc):
    res = run_world(2, 900, 256, engine="hip", sn2=-0.5)
    assert all(r["nlz"] != r["nlz"] for r in res)


def _bench(world, extra_env, size, timeout=900):
    env = dict(os.environ, **extra_env)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
           "127.0.0.1", "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(world),
           "--steps", "2", "--warmup", "1", "--size", str(size), "--no-cpu", "--no-n65536"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
    assert out.returncode == 0, out.stderr.decode()[-3000:]
    lines = [l for l in out.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1                                   # rank 0 only
    return json.loads(lines[0])


def _single(size):
    single = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--n",
                             str(size), "--no-cpu"], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert single.returncode == 0, single.stderr.decode()[-2000:]
    return json.loads([l for l in single.stdout.decode().splitlines() if l.startswith("{")][-1])


@pytest.mark.gpu
def test_rccl_binding_with_a_one_rank_communicator(orc):
    """The built-in RCCL transport on the one GPU a test box has: a ONE-rank communicator (ncclGetUniqueId,
    ncclCommInitRank through dlopen) carries every broadcast and all-reduce of two steps and a gradient."""
    from gp_ss_ak_amd import dist as gd
    os.environ["GPAK_DIST_RCCL_WORLD1"] = "1"
    try:
        gp = gd.DistRank(0, 1, device=0)
    finally:
        del os.environ["GPAK_DIST_RCCL_WORLD1"]
    n = 1500
    X, y = synth.drillholes(n)
    gp.set_train(X, y, nb=256)
    info, alpha = oracle_ref(orc, n)
    for _ in range(2):
        gp.set_params(synth.DEFAULT_EXPANS, synth.DEFAULT_BIAS, synth.DEFAULT_SN2, 1)
        assert abs(gp.nlz() - info.nlz) <= 1e-9 * abs(info.nlz)
    assert np.abs(gp.get_alpha() - alpha).max() <= 1e-8 * np.abs(alpha).max()
    K = orc.gram(X, X, synth.DEFAULT_EXPANS, synth.DEFAULT_BIAS, 1)
    i2, a2, L2 = orc.nlz_lean(K, y, synth.DEFAULT_SN2)
    go = orc.grad_ref(X, y, K, L2, a2, np.array(synth.DEFAULT_EXPANS), synth.DEFAULT_BIAS, synth.DEFAULT_SN2, 1)
    assert np.abs(gp.grad() - go).max() <= 1e-8 * np.abs(go).max()
    assert gp.stats()["bytes_broadcast"] > 0
    gp.close()


@pytest.mark.gpu
def test_bench_distributed_entry_point_over_rccl_one_rank():
    """bench.py --gpus path exactly as the driver launches it (torch.distributed.run), the C++ schedule with the
    built-in RCCL transport and the one rank this box has; the step must agree with the single-context path."""
    d = _bench(1, {"GPAK_FORCE_DIST": "1"}, 4096)
    assert d["n_gpus"] == 1 and d["unit"] == "steps/s" and d["value"] > 0 and "roofline" in d
    assert d["config"]["transport"] == "RCCL" and "at N=4096" in d["metric"]
    s = _single(4096)
    assert abs(d["nlz"] - s["nlz"]) <= 1e-9 * abs(s["nlz"])


@pytest.mark.gpu
def test_bench_hands_over_to_the_one_process_host_when_rccl_cannot_start():
    """If the per-process RCCL start-up fails (here: librccl made unloadable), every rank learns it through the control
    plane before the first step -- no rank is left inside a rendezvous -- and rank 0 runs the SAME C++ schedule from one
    process (gpak_create_multi); metric, workload and a `fallback` note all say so; same nlZ."""
    d = _bench(1, {"GPAK_FORCE_DIST": "1", "GPAK_DIST_RCCL_WORLD1": "1", "GPAK_RCCL_DISABLE": "1"}, 4096)
    assert "fallback" in d and d["n_gpus"] == 1 and d["value"] > 0
    assert "fallback host" in d["metric"] and "FALLBACK" in d["config"]["workload"]
    s = _single(4096)
    assert abs(d["nlz"] - s["nlz"]) <= 1e-9 * abs(s["nlz"])


def _bench_as_typed(world, size, extra_env, extra_args=()):
    """exactly `python bench.py --gpus N ...` -- no launcher, WORLD_SIZE unset"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(extra_env)
    out = subprocess.run([sys.executable, "bench.py", "--gpus", str(world), "--size", str(size), "--steps", "2",
                          "--warmup", "1", "--no-n65536", "--no-cpu", *extra_args], env=env, cwd=ROOT,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert out.returncode == 0, out.stderr.decode()[-3000:]
    lines = [l for l in out.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_multi_gpu_as_typed_starts_its_own_launcher():
    """`python bench.py --gpus 2 --size 4096` as typed: bench.py has not touched a GPU, starts torch.distributed.run as
    a CHILD and relays the one JSON line (two ranks rehearsed on this box's one GPU, collectives staged through gloo)."""
    d = _bench_as_typed(2, 4096, {"GPAK_DIST_TRANSPORT": "staged", "GPAK_DIST_DEVICE": "0"})
    assert d["n_gpus"] == 2 and d["value"] > 0 and len(d["phases_ms_per_rank"]) == 2
    assert "torch.distributed.run" in d["launcher"]["how"] and d["launcher"]["failed_attempts"] == []
    assert d["roofline"]["traffic"] is None and d["roofline"]["frac"] > 0
    s = _single(4096)
    assert abs(d["nlz"] - s["nlz"]) <= 1e-9 * abs(s["nlz"])


@pytest.mark.gpu
def test_bench_multi_gpu_as_typed_falls_through_to_the_one_process_host():
    """The first start fails (RCCL refuses two ranks on one device: `invalid usage`), the launcher's second attempt --
    ONE process, one host thread per rank, in-process peer copies -- produces the line and names the failed attempt."""
    d = _bench_as_typed(2, 4096, {"GPAK_MULTI_DEVICES": "0,0", "GPAK_DIST_DEVICE": "0", "GPAK_RCCL_INIT_TIMEOUT_S": "40"})
    assert d["n_gpus"] == 2 and d["value"] > 0
    how = d["launcher"]["how"] + " " + d.get("fallback", "")
    assert "gpak_create_multi" in how
    assert "in-process peer copies" in d["config"]["transport"]
    s = _single(4096)
    assert abs(d["nlz"] - s["nlz"]) <= 1e-9 * abs(s["nlz"])


@pytest.mark.gpu
def test_bench_multi_gpu_one_process_host_directly():
    """GPAK_BENCH_MULTI_ORDER=inproc: the launcher goes straight to `bench.py --inproc` (gpak_create_multi)."""
    d = _bench_as_typed(3, 4096, {"GPAK_MULTI_DEVICES": "0,0,0", "GPAK_BENCH_MULTI_ORDER": "inproc"})
    assert d["n_gpus"] == 3 and "--inproc" in d["launcher"]["how"] and len(d["phases_ms_per_rank"]) == 3
    assert d["config"]["host"] == "one process, one thread per GPU" and d["bytes_broadcast_per_step"] > 0
    s = _single(4096)
    assert abs(d["nlz"] - s["nlz"]) <= 1e-9 * abs(s["nlz"])


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_bench_distributed_entry_point_with_several_ranks(world):
    """bench.py --gpus N as the driver launches it, N ranks rehearsed on this box's one GPU (collectives staged
    through gloo instead of RCCL): same nlZ as the single-context path, one JSON line with the per-rank phases."""
    d = _bench(world, {"GPAK_DIST_TRANSPORT": "staged", "GPAK_DIST_DEVICE": "0"}, 4096)
    assert d["n_gpus"] == world and d["scaling"] == "strong" and d["value"] > 0 and d["bytes_broadcast_per_step"] > 0
    assert len(d["phases_ms_per_rank"]) == world
    assert all({"factor_ms", "bulk_ms", "chain_ms", "comm_ms", "wait_ms"} <= set(p) for p in d["phases_ms_per_rank"])
    s = _single(4096)
    assert abs(d["nlz"] - s["nlz"]) <= 1e-9 * abs(s["nlz"])


# ---- ONE process driving several GPUs: gpak_create_multi (csrc/multi.hip) ---------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("ranks,n", [(1, 700), (2, 1500), (3, 2500), (4, 5000)])
def test_multi_context_matches_oracle(orc, ranks, n):
    """The gpak_ctx surface over a group of rank threads (here all on the box's one GPU, collectives through the
    in-process peer-copy transport): nlZ / terms / alpha from the C++ block-column-cyclic schedule, prediction
    sharded over the test points, gradient on the replica of device 0 -- all against the oracle, and repeated
    with new parameters (buffers are reused between steps)."""
    from gp_ss_ak_amd import gpak
    E = np.array(synth.DEFAULT_EXPANS)
    X, y = synth.drillholes(n)
    Xt = synth.test_points(777)
    g = gpak.Gpak(devices=[0] * ranks)
    try:
        g.set_train(X, y)
        for k, sn2 in enumerate((synth.DEFAULT_SN2, 0.05)):
            e = E.copy()
            e[1] += 0.05 * k
            g.set_params(e, synth.DEFAULT_BIAS, sn2, gpak.DIST_DIRECT)
            K = orc.gram(X, X, e, synth.DEFAULT_BIAS, orc.DIST_DIRECT)
            info, alpha, L = orc.nlz_refseq(K, y, sn2)
            nlz = g.logLikelihood()
            assert abs(nlz - info.nlz) <= 1e-9 * abs(info.nlz)
            q, s, l = g.nlz_terms()
            assert abs(l - info.logdet) <= 1e-10 * abs(info.logdet) and abs(q - info.quad) <= 1e-9 * abs(info.quad)
            assert np.abs(g.solve_alpha() - alpha).max() <= 1e-8 * np.abs(alpha).max()
            for compat in (0, 3):
                mean, var = g.posteriorMeanVar(Xt, compat=compat)
                mo, vo = orc.predict(X, Xt, e, synth.DEFAULT_BIAS, sn2, alpha, L, orc.DIST_DIRECT, compat)
                assert np.abs(mean - mo).max() <= 1e-8 * np.abs(mo).max()
                assert np.abs(var - vo).max() <= 1e-8 * np.abs(vo).max()
            go = orc.grad_ref(X, y, K, L, alpha, e, synth.DEFAULT_BIAS, sn2, orc.DIST_DIRECT)
            assert np.abs(g.GradLL() - go).max() <= 1e-8 * np.abs(go).max()
            # prediction, alpha and the gradient all reused the ONE distributed factorisation of these parameters
            # (the replicas take the factor over from the packed panels their rank holds)
            assert g.timing()["evaluations"] == k + 1
            z = np.random.default_rng(k).normal(size=(n, 2))
            xs = g.solve_chol(z)                                     # solve_chol on the imported factor (rank 0's replica)
            assert np.abs(L @ (L.T @ xs) - z).max() <= 1e-8 * np.abs(z).max()
            assert g.timing()["evaluations"] == k + 1
        g.set_params(E, synth.DEFAULT_BIAS, -0.5, gpak.DIST_DIRECT)          # Chol_fail -> NaN on the group too
        assert g.logLikelihood() != g.logLikelihood()
        assert g.failed_column() == 1                                        # ... with the failing column, as one GPU
        assert g.transport() == ("none" if ranks == 1 else "in-process peer copies")
    finally:
        g.close()


@pytest.mark.gpu
def test_multi_context_expansion_mode_prediction_uses_the_pooled_mean_of_all_test_points(orc):
    """Expansion-form distances are centred on the pooled mean of train + ALL test points (Kernel.cpp:1391-1392);
    a sharded prediction must not centre every slice on its own mean."""
    from gp_ss_ak_amd import gpak
    E = np.array(synth.DEFAULT_EXPANS)
    X, y = synth.drillholes(900)
    Xt = synth.test_points(600)
    one, many = gpak.Gpak(0), gpak.Gpak(devices=[0, 0, 0])
    try:
        for g in (one, many):
            g.set_train(X, y)
            g.set_params(E, synth.DEFAULT_BIAS, synth.DEFAULT_SN2, gpak.DIST_EXPANSION)
        m1, v1 = one.posteriorMeanVar(Xt)
        m3, v3 = many.posteriorMeanVar(Xt)
        assert abs(many.logLikelihood() - one.logLikelihood()) <= 1e-9 * abs(one.logLikelihood())
        assert np.abs(m3 - m1).max() <= 1e-9 * np.abs(m1).max() and np.abs(v3 - v1).max() <= 1e-9 * np.abs(v1).max()
    finally:
        one.close()
        many.close()


@pytest.mark.gpu
def test_cli_with_gpus_precision_and_timing(tmp_path):
    """`gp_ss_ak --gpus 2 --precision f32 --timing -`: the device options of SURVEY.md section 5 through the train and
    test verbs (two ranks rehearsed on the box's one GPU: GPAK_MULTI_DEVICES=0,0), same model as one GPU."""
    host = os.path.join(ROOT, "gp_ss_ak_amd", "host")
    subprocess.check_call(["make", "-s", "-C", host])
    Xr, yr = synth.drillholes_raw(640)
    perm = np.random.default_rng(5).permutation(640)
    Xr, yr = Xr[perm], yr[perm]

    def write(path, X, y):
        with open(path, "w") as f:
            for r, v in zip(X, y):
                f.write("\t".join(f"{t:.17g}" for t in list(r) + [v]) + "\n")
    write(tmp_path / "train.txt", Xr[:512], yr[:512])
    write(tmp_path / "test.txt", Xr[512:], yr[512:])
    exe = os.path.join(host, "gp_ss_ak")
    runs = {}
    for tag, opts, env in (("one", [], {}), ("two", ["--gpus", "2", "--precision", "f32"], {"GPAK_MULTI_DEVICES": "0,0"})):
        model = str(tmp_path / f"model_{tag}")
        e = dict(os.environ, GPAK_MAX_ITERS="4", **env)
        out = subprocess.check_output([exe, "-v", "1", "-np"] + opts + ["--timing", "-", "train", "-k", "ExpAns", "-kn", "1",
                                       "-o", "LBFGS", str(tmp_path / "train.txt"), model], env=e, cwd=tmp_path).decode()
        its = [float(line.split("-logL:")[1]) for line in out.splitlines() if line.startswith("Iteration:")]
        tim = json.loads([line for line in out.splitlines() if line.startswith("TIMING ")][0][7:])
        out2 = subprocess.check_output([exe, "-v", "1", "-np"] + opts + ["--timing", str(tmp_path / f"t_{tag}.json"), "test",
                                        str(tmp_path / "test.txt"), model, str(tmp_path / "train.txt")], env=e,
                                       cwd=tmp_path).decode()
        mse = float(out2.split("Mean Square Error of testing:")[1].split()[0])
        tt = json.load(open(tmp_path / f"t_{tag}.json"))
        runs[tag] = (its, tim, mse, tt)
    assert runs["two"][1]["gpus"] == 2 and runs["two"][1]["precision"] == "f32" and runs["one"][1]["gpus"] == 1
    assert runs["one"][1]["evaluations"] > 4 and runs["one"][1]["accumulated"]["factor_ms"] > 0
    assert runs["one"][3]["last"]["predict_ms"] > 0
    assert len(runs["one"][0]) == len(runs["two"][0]) >= 3
    for a, b in zip(runs["one"][0][:2], runs["two"][0][:2]):      # six printed digits, before the optimiser's chaos
        assert abs(a - b) <= 2e-5 * abs(a)
    assert abs(runs["one"][2] - runs["two"][2]) <= 0.05 * runs["one"][2] + 1e-6
