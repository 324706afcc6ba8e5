"""Thread-per-rank harness for the HOST logic of csrc/multi.hip + csrc/dist.hip on a box WITHOUT a GPU (test
infrastructure; run by tests/test_sanitizers.py under ThreadSanitizer / AddressSanitizer + UBSan).

gpak_create_multi_with_engines builds the same group gpak_create_multi builds -- one host thread per rank, the
in-process rendezvous transport, per-rank error slots, the gpak_ctx surface -- over P NumPy engines
(tests/np_dist_engine.py), so every line of the thread-per-GPU logic runs: start-up self-check on all ranks at once,
set_train, two evaluations with new parameters, alpha, the distributed gradient, a Chol_fail step and the teardown.
Results are checked against the oracle; the sanitizer reports through stderr / the exit code."""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
from gp_ss_ak_amd import dist as gd, gpak, synth  # noqa: E402
from np_dist_engine import NumpyDistEngine  # noqa: E402
from oracle import oracle as orc  # noqa: E402


def main(P, n):
    lib = gd._load()
    lib.gpak_create_multi_with_engines.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.POINTER(gd.Engine))]
    engines = [NumpyDistEngine() for _ in range(P)]
    arr = (C.POINTER(gd.Engine) * P)(*[C.pointer(e.table) for e in engines])
    h = C.c_void_p()
    rc = lib.gpak_create_multi_with_engines(C.byref(h), P, arr)
    assert rc == 0, (rc, lib.gpak_global_error())
    g = gpak.Gpak.__new__(gpak.Gpak)
    g._lib, g._h, g.N = lib, h, 0
    try:
        body(g, P, n)
    finally:
        g.close()          # while the interpreter is alive: the engines are Python callbacks
    print(f"san_worker ok: {P} ranks, n={n}")


def body(g, P, n):
    X, y = synth.drillholes(n)
    E = np.array(synth.DEFAULT_EXPANS)
    g.set_train(X, y)
    assert g.transport() == ("none" if P == 1 else "in-process peer copies")
    for k, sn2 in enumerate((synth.DEFAULT_SN2, 0.05)):
        e = E.copy()
        e[1] += 0.05 * k
        g.set_params(e, synth.DEFAULT_BIAS, sn2, gpak.DIST_DIRECT)
        K = orc.gram(X, X, e, synth.DEFAULT_BIAS, orc.DIST_DIRECT)
        info, alpha, L = orc.nlz_lean(K, y, sn2)
        nlz = g.logLikelihood()
        assert abs(nlz - info.nlz) <= 1e-9 * abs(info.nlz), (nlz, info.nlz)
        assert np.abs(g.solve_alpha() - alpha).max() <= 1e-8 * np.abs(alpha).max()
        go = orc.grad_ref(X, y, K, L, alpha, e, synth.DEFAULT_BIAS, sn2, orc.DIST_DIRECT)
        assert np.abs(g.GradLL() - go).max() <= 1e-8 * np.abs(go).max()
        t = g.timing()
        assert t["evaluations"] == k + 1
        for r in range(P):
            assert g.rank_stats(r)["rank"] == r
    try:
        g.posteriorMeanVar(synth.test_points(8))                     # no device replicas behind CPU engines
        raise SystemExit("prediction on a CPU-engine group should be refused")
    except gpak.GpakError as ex:
        assert ex.status == gpak.ENOTIMPL and "rank" in str(ex), str(ex)
    g.set_params(E, synth.DEFAULT_BIAS, -0.5, gpak.DIST_DIRECT)      # Chol_fail on every rank, and the group recovers
    assert g.logLikelihood() != g.logLikelihood() and g.failed_column() == 1
    g.set_params(E, synth.DEFAULT_BIAS, synth.DEFAULT_SN2, gpak.DIST_DIRECT)
    assert np.isfinite(g.logLikelihood())
    print(f"san_worker ok: {P} ranks, n={n}")


if __name__ == "__main__":
    main(int(sys.argv[1]), int(sys.argv[2]))
