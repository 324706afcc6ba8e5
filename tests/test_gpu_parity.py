"""-m gpu: the HIP path (through the C-ABI) against the CPU oracle on the same seeded inputs.

Tolerances (fp64): north_star asks for 1e-5 relative on nlZ / predictive mean / variance
against the Armadillo CPU path.  The tests hold the HIP path to far tighter bounds against
the oracle; each bound is written next to its assert.
"""
import math
import os

import numpy as np
import pytest

from gp_ss_ak_amd import gpak, synth

pytestmark = pytest.mark.gpu

E = np.array(synth.DEFAULT_EXPANS)
BIAS, SN2 = synth.DEFAULT_BIAS, synth.DEFAULT_SN2


def rel(a, b):
    return np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(np.asarray(b)).max(), 1e-300)


@pytest.mark.parametrize("N", [8, 64, 200, 512, 1000])
@pytest.mark.parametrize("mode", [gpak.DIST_DIRECT, gpak.DIST_EXPANSION])
def test_gram_matches_oracle(gp, orc, N, mode):
    X, y = synth.drillholes(N)
    gp.set_train(X, y)
    gp.set_params(E, BIAS, SN2, mode)
    K, D2 = gp.gram(want_d2=True)
    Ko, D2o = orc.gram(X, X, E, BIAS, mode, want_d2=True)
    if mode == gpak.DIST_DIRECT:
        # same arithmetic up to fma contraction and the exp/sqrt implementations
        assert np.abs(D2 - D2o).max() <= 1e-14
        assert rel(K, Ko) <= 1e-13
        assert np.all(K.diagonal() == E[6] ** 2 + BIAS)  # K_ii = sigma^2 + bias exactly
    else:
        # expansion form: cancellation noise ~1e-15 in D2 becomes ~3e-8 in K near D2 = 0
        assert np.abs(D2 - D2o).max() <= 1e-13
        assert rel(K, Ko) <= 2e-7
    assert np.array_equal(K, K.T)


@pytest.mark.parametrize("N", [8, 64, 200, 512, 1000, 2048, 4096, 6000])
def test_factor_alpha_nlz_match_oracle(gp, orc, N):
    X, y = synth.drillholes(N)
    gp.set_train(X, y)
    gp.set_params(E, BIAS, SN2, gpak.DIST_DIRECT)
    Ko = orc.gram(X, X, E, BIAS, orc.DIST_DIRECT)
    info, alpha_o, Lo = orc.nlz_refseq(Ko, y, SN2)  # the reference's IRLS/Brent sequence
    assert gp.factor()
    R = gp.chol_upper()
    assert np.allclose(np.triu(R), R)
    assert rel(R.T, Lo) <= 1e-11                       # factor of B = I + K/sn2
    B = np.eye(N) + Ko / SN2
    assert rel(R.T @ R, B) <= 1e-13                    # R'R = B residual
    alpha = gp.solve_alpha()
    assert rel(alpha, alpha_o) <= 1e-8                 # cond(B) ~ 1e3..1e5 times eps, IRLS tol 1e-6 on psi
    nlz = gp.logLikelihood()
    assert abs(nlz - info.nlz) <= 1e-9 * abs(info.nlz)
    q, slp, ld = gp.nlz_terms()
    assert abs(ld - info.logdet) <= 1e-11 * abs(info.logdet)
    assert abs(q - info.quad) <= 1e-9 * abs(info.quad)
    assert abs(slp - info.sumlp) <= 1e-9 * abs(info.sumlp)


@pytest.mark.parametrize("N,M", [(64, 16), (512, 16), (1000, 300), (4096, 1000)])
def test_predict_matches_oracle(gp, orc, N, M):
    X, y = synth.drillholes(N)
    Xte = synth.test_points(M)
    for mode in (gpak.DIST_DIRECT, gpak.DIST_EXPANSION):
        gp.set_train(X, y)
        gp.set_params(E, BIAS, SN2, mode)
        Ko = orc.gram(X, X, E, BIAS, mode)
        info, alpha_o, Lo = orc.nlz_lean(Ko, y, SN2)
        for compat in (0, gpak.COMPAT_VARCLAMP | gpak.COMPAT_SN2SKIP):
            mean, var = gp.posteriorMeanVar(Xte, compat=compat)
            mo, vo = orc.predict(X, Xte, E, BIAS, SN2, alpha_o, Lo, mode, compat)
            # direct mode: same arithmetic up to rounding.  expansion mode: the 1e-15 cancellation
            # noise of D2 (3e-8 in K near D2=0) is implementation-defined and is amplified by
            # cond(K + sn2 I) in alpha; 1e-5 is north_star's bound.
            tol = 1e-8 if mode == gpak.DIST_DIRECT else 1e-5
            assert rel(mean, mo) <= tol
            assert rel(var, vo) <= tol
            if compat & gpak.COMPAT_VARCLAMP:
                assert var[0] == SN2  # Q3: element 0 is always "clamped"


def test_not_positive_definite_reports_chol_fail(gp):
    X, y = synth.drillholes(64)
    gp.set_train(X, y)
    gp.set_params(E, BIAS, -0.5, gpak.DIST_DIRECT)  # negative noise -> B indefinite
    assert not gp.factor()
    assert gp.failed_column() >= 1
    assert math.isnan(gp.logLikelihood())            # GP_Utils.cpp:1145-1146
    gp.set_params(E, BIAS, SN2, gpak.DIST_DIRECT)    # and the context recovers
    assert gp.factor()


@pytest.mark.parametrize("N", [64, 512, 1000, 2500])
@pytest.mark.parametrize("mode", [gpak.DIST_DIRECT, gpak.DIST_EXPANSION])
def test_reference_style_gradient_matches_oracle(gp, orc, N, mode):
    """GradLL + getGradients as written (not the true gradient: SURVEY.md 8(f-1))."""
    X, y = synth.drillholes(N)
    gp.set_train(X, y)
    gp.set_params(E, BIAS, SN2, mode)
    g = gp.GradLL()
    Ko = orc.gram(X, X, E, BIAS, mode)
    info, alpha_o, Lo = orc.nlz_lean(Ko, y, SN2)
    go = orc.grad_ref(X, y, Ko, Lo, alpha_o, E, BIAS, SN2, mode)
    scale = np.abs(go).max()
    # the sums cancel heavily (QW = B^-1/sn2 - alpha alpha'), so the bound is relative to the
    # largest entry; expansion mode carries the 1/sqrt(D2) amplification of the cancellation noise
    tol = 1e-8 if mode == gpak.DIST_DIRECT else 1e-5
    assert np.abs(g - go).max() <= tol * scale
    assert g[7] == 0.0


def test_fp32_prediction_context(orc):
    """GPAK_F32 (BASELINE.json configs[4]): fp32 cross-kernel / substitution / variance sums, fp64
    everything else.  Tolerance: fp32 arithmetic through a triangular solve of a matrix with
    cond(L) ~ 1e2..1e3 -- 2e-4 relative on the variance, the mean stays at the fp64 bound."""
    N, M = 1500, 700
    X, y = synth.drillholes(N)
    Xte = synth.test_points(M)
    g32 = gpak.Gpak(0, gpak.F32)
    g32.set_train(X, y)
    g32.set_params(E, BIAS, SN2, gpak.DIST_DIRECT)
    Ko = orc.gram(X, X, E, BIAS, orc.DIST_DIRECT)
    info, alpha_o, Lo = orc.nlz_lean(Ko, y, SN2)
    assert abs(g32.logLikelihood() - info.nlz) <= 1e-9 * abs(info.nlz)     # training step is fp64
    mean, var = g32.posteriorMeanVar(Xte)
    mo, vo = orc.predict(X, Xte, E, BIAS, SN2, alpha_o, Lo, orc.DIST_DIRECT, 0)
    assert rel(mean, mo) <= 1e-8
    assert rel(var, vo) <= 2e-4
    g32.close()


@pytest.mark.parametrize("mode", [gpak.DIST_DIRECT, gpak.DIST_EXPANSION])
def test_other_kernel_compositions(gp, orc, mode):
    """f-4: HybKerns{ExpAns, Exp, RBF} + Bias + White through gpak_set_kernel: Gram, nlZ, alpha, prediction.
    (the reference-style gradient exists for ExpAns(+Bias) only)"""
    import scipy.linalg as sl
    N, M = 700, 40
    X, y = synth.drillholes(N)
    Xt = synth.test_points(M)
    white = 0.10
    terms = [(gpak.KERN_EXPANS, E), (gpak.KERN_EXP, [0.5, 0.9]), (gpak.KERN_RBF, [0.5, 0.9, 0.5])]
    gp.set_train(X, y)
    gp.set_kernel(terms, BIAS, white, SN2, mode)
    tolk = 1e-13 if mode == gpak.DIST_DIRECT else 2e-7
    K, D2 = gp.gram(want_d2=True)
    Ko, D2o = orc.gram_hyb(X, X, terms, BIAS, white, mode, want_d2=True)
    assert rel(K, Ko) <= tolk and np.abs(D2 - D2o).max() <= 1e-12
    Kc = gp.compute_k(X, Xt)
    assert rel(Kc, orc.gram_hyb(X, Xt, terms, BIAS, white, mode)) <= tolk
    info, alpha_o, Lo = orc.nlz_lean(Ko, y, SN2)
    tol = 1e-9 if mode == gpak.DIST_DIRECT else 1e-6
    assert abs(gp.logLikelihood() - info.nlz) <= tol * abs(info.nlz)
    assert rel(gp.solve_alpha(), alpha_o) <= (1e-8 if mode == gpak.DIST_DIRECT else 1e-5)
    mean, var = gp.posteriorMeanVar(Xt)
    kX = orc.gram_hyb(X, Xt, terms, BIAS, white, mode)
    cf = sl.cho_factor(Ko + SN2 * np.eye(N), lower=True)
    kD = E[6] ** 2 + 0.9 ** 2 + 0.5 ** 2 + BIAS + white      # HybKerns::diag_Compute
    mo = kX.T @ alpha_o
    vo = np.maximum(kD - np.einsum("ij,ij->j", kX, sl.cho_solve(cf, kX)), 0) + SN2
    tp = 1e-8 if mode == gpak.DIST_DIRECT else 1e-5
    assert rel(mean, mo) <= tp and rel(var, vo) <= tp
    with pytest.raises(gpak.GpakError) as ei:
        gp.GradLL()
    assert ei.value.status == gpak.ENOTIMPL              # fixed-length gradient: ExpAns(+Bias) only
    with pytest.raises(gpak.GpakError) as ei:
        gp.GradLL_hyb(8 + 2 + 3 + 2)
    assert ei.value.status == gpak.ENOTIMPL              # White child: no gradient upstream either
    # without the White child: the children's getGradients as written (Exp / RBF on the summed D2)
    gp.set_kernel(terms, BIAS, 0.0, SN2, mode)
    g = gp.GradLL_hyb(8 + 2 + 3 + 2)
    K0 = orc.gram_hyb(X, X, terms, BIAS, 0.0, mode)
    info0, a0, L0 = orc.nlz_lean(K0, y, SN2)
    go = orc.grad_hyb(X, y, K0, L0, a0, terms, True, SN2, mode)
    assert np.abs(g - go).max() <= (1e-8 if mode == gpak.DIST_DIRECT else 1e-5) * np.abs(go).max()
    # a composition without ExpAns
    t2 = [(gpak.KERN_RBF, [0.7, 1.1, 0.6]), (gpak.KERN_EXP, [0.4, 0.8])]
    gp.set_kernel(t2, BIAS, 0.0, SN2, mode)
    g = gp.GradLL_hyb(3 + 2 + 2)
    K2 = orc.gram_hyb(X, X, t2, BIAS, 0.0, mode)
    info2, a2, L2 = orc.nlz_lean(K2, y, SN2)
    go = orc.grad_hyb(X, y, K2, L2, a2, t2, True, SN2, mode)
    assert abs(gp.logLikelihood() - info2.nlz) <= tol * abs(info2.nlz)
    assert np.abs(g - go).max() <= (1e-8 if mode == gpak.DIST_DIRECT else 1e-5) * np.abs(go).max()
    gp.set_params(E, BIAS, SN2, gpak.DIST_DIRECT)   # back to the default composition for later tests


@pytest.mark.parametrize("mode", [gpak.DIST_DIRECT, gpak.DIST_EXPANSION])
def test_four_column_inputs(gp, orc, mode):
    """x, y, z + rock-type column with its own inverse width (expans[7]); Gram, cross-Gram, nlZ,
    alpha, prediction, the reference-style gradient (g7 != 0 here) and a composition with Exp / RBF children."""
    N, M = 600, 50
    X, y = synth.drillholes4(N)
    Xt = synth.test_points4(M)
    direct = mode == gpak.DIST_DIRECT
    gp.set_train(X, y)
    gp.set_params(E, BIAS, SN2, mode)
    K, D2 = gp.gram(want_d2=True)
    Ko, D2o = orc.gram(X, X, E, BIAS, mode, want_d2=True)
    assert rel(K, Ko) <= (1e-13 if direct else 2e-7) and np.abs(D2 - D2o).max() <= 1e-12
    K3 = orc.gram(X[:, :3].copy(order="F"), X[:, :3].copy(order="F"), E, BIAS, mode)
    assert np.abs(K3 - Ko).max() > 1e-2                  # the rock-type column matters
    assert rel(gp.compute_k(X, Xt), orc.gram(X, Xt, E, BIAS, mode)) <= (1e-13 if direct else 2e-7)
    info, alpha_o, Lo = orc.nlz_refseq(Ko, y, SN2)
    assert abs(gp.logLikelihood() - info.nlz) <= (1e-9 if direct else 1e-6) * abs(info.nlz)
    assert rel(gp.solve_alpha(), alpha_o) <= (1e-8 if direct else 1e-5)
    mean, var = gp.posteriorMeanVar(Xt, compat=3)
    mo, vo = orc.predict(X, Xt, E, BIAS, SN2, alpha_o, Lo, mode, compat=3)
    assert rel(mean, mo) <= (1e-8 if direct else 1e-5) and rel(var, vo) <= (1e-8 if direct else 1e-5)
    g = gp.GradLL()
    go = orc.grad_ref(X, y, Ko, Lo, alpha_o, E, BIAS, SN2, mode)
    assert go[7] != 0.0
    assert np.abs(g - go).max() <= (1e-8 if direct else 1e-5) * np.abs(go).max()
    assert abs(g[7] - go[7]) <= (1e-10 if direct else 1e-6) * abs(go[7])
    with pytest.raises(gpak.GpakError) as ei:
        gp.posteriorMeanVar(Xt[:, :3].copy(order="F"))
    assert ei.value.status == gpak.EINVAL                # column count must match the training set
    terms = [(gpak.KERN_EXPANS, E), (gpak.KERN_RBF, [0.5, 0.9, 0.5]), (gpak.KERN_EXP, [0.6, 0.7])]
    gp.set_kernel(terms, BIAS, 0.0, SN2, mode)
    Kh = orc.gram_hyb(X, X, terms, BIAS, 0.0, mode)
    assert rel(gp.gram(), Kh) <= (1e-13 if direct else 2e-7)
    ih, ah, Lh = orc.nlz_lean(Kh, y, SN2)
    assert abs(gp.logLikelihood() - ih.nlz) <= (1e-9 if direct else 1e-6) * abs(ih.nlz)
    gh = gp.GradLL_hyb(8 + 3 + 2 + 2)
    gho = orc.grad_hyb(X, y, Kh, Lh, ah, terms, True, SN2, mode)
    assert np.abs(gh - gho).max() <= (1e-8 if direct else 1e-5) * np.abs(gho).max()
    gp.set_params(E, BIAS, SN2, gpak.DIST_DIRECT)


@pytest.mark.parametrize("N", [1, 2, 3, 127, 128, 129, 257])
def test_edge_sizes(gp, orc, N):
    """Sizes around the 128-tile padding, down to a single point; one test point; the reference's quirks of the
    predictive variance (Q3: mask used as an index list, Q4: +sn2 skipped when sn2 == 1)."""
    X, y = synth.drillholes(max(N, 4))
    X, y = np.asfortranarray(X[:N]), y[:N].copy()
    gp.set_train(X, y)
    for sn2 in (SN2, 1.0):
        gp.set_params(E, BIAS, sn2, gpak.DIST_DIRECT)
        Ko = orc.gram(X, X, E, BIAS, orc.DIST_DIRECT)
        assert rel(gp.gram(), Ko) <= 1e-13
        info, alpha_o, Lo = orc.nlz_refseq(Ko, y, sn2)
        nlz = gp.logLikelihood()
        assert abs(nlz - info.nlz) <= 1e-10 * max(1.0, abs(info.nlz))
        assert rel(gp.solve_alpha(), alpha_o) <= 1e-9
        Xt = synth.test_points(1)
        for compat in (0, 3):
            mean, var = gp.posteriorMeanVar(Xt, compat=compat)
            mo, vo = orc.predict(X, Xt, E, BIAS, sn2, alpha_o, Lo, orc.DIST_DIRECT, compat=compat)
            assert rel(mean, mo) <= 1e-9 and np.abs(var - vo).max() <= 1e-9 * max(1.0, np.abs(vo).max())
        g = gp.GradLL()
        go = orc.grad_ref(X, y, Ko, Lo, alpha_o, E, BIAS, sn2, orc.DIST_DIRECT)
        assert np.abs(g - go).max() <= 1e-8 * max(1.0, np.abs(go).max())
    gp.set_params(E, BIAS, SN2, gpak.DIST_DIRECT)


def test_coincident_points_and_ragged_batches(gp, orc):
    """Duplicated sample locations (D2 = 0 off the diagonal: the dk = 0 branch of the gradient, Kernel.cpp:1179-1183)
    and a prediction whose size is not a multiple of anything."""
    N = 300
    X, y = synth.drillholes(N)
    X[150:160] = X[10:20]                      # ten coincident pairs
    X = np.asfortranarray(X)
    gp.set_train(X, y)
    gp.set_params(E, BIAS, SN2, gpak.DIST_DIRECT)
    K, D2 = gp.gram(want_d2=True)
    assert np.all(D2[150:160, 10:20].diagonal() == 0.0) and np.all(K[150:160, 10:20].diagonal() == E[6] ** 2 + BIAS)
    Ko = orc.gram(X, X, E, BIAS, orc.DIST_DIRECT)
    info, alpha_o, Lo = orc.nlz_refseq(Ko, y, SN2)
    assert abs(gp.logLikelihood() - info.nlz) <= 1e-9 * abs(info.nlz)
    g = gp.GradLL()
    go = orc.grad_ref(X, y, Ko, Lo, alpha_o, E, BIAS, SN2, orc.DIST_DIRECT)
    assert np.all(np.isfinite(g)) and np.abs(g - go).max() <= 1e-8 * np.abs(go).max()
    Xt = synth.test_points(777)
    mean, var = gp.posteriorMeanVar(Xt)
    mo, vo = orc.predict(X, Xt, E, BIAS, SN2, alpha_o, Lo, orc.DIST_DIRECT)
    assert rel(mean, mo) <= 1e-8 and rel(var, vo) <= 1e-8


def test_alternative_kernels_give_the_same_step():
    """No CU-masked tail stream, no separate bulk queue, everything on the 64x64-per-wave kernel, both block kernels,
    the three workgroup heights of the panel-chain GEMM: same nlZ as the default build of the step (the GPAK_*
    environment sets the process-wide tuning defaults once, at the first gpak_create, hence subprocesses)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    vals = {}
    for name, env in (("default", {}), ("nomask", {"GPAK_TAIL_MASK": "0"}), ("no_bulk_queue", {"GPAK_BULK_QUEUE": "0"}),
                      ("nosmall", {"GPAK_GEMM_SMALL": "0"}),
                      # the two builds of the 128x128 block kernel (8 waves / 4 waves at 80 VGPRs) and the three
                      # workgroup heights of the panel-chain GEMM
                      ("potrf_co_never", {"GPAK_POTRF_CO": "0"}), ("potrf_co_always", {"GPAK_POTRF_CO": "2"}),
                      ("rows64", {"GPAK_GEMM_SMALL_ROWS": "64"}), ("rows32", {"GPAK_GEMM_SMALL_ROWS": "32"}),
                      ("no_lookahead", {"GPAK_LOOKAHEAD": "0"})):
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "1", "--warmup", "0", "--size",
                              "6000", "--no-cpu", "--no-n65536"], env=dict(os.environ, **env), cwd=root, stdout=subprocess.PIPE,
                             stderr=subprocess.PIPE, timeout=600)
        assert out.returncode == 0, out.stderr.decode()[-2000:]
        vals[name] = json.loads([l for l in out.stdout.decode().splitlines() if l.startswith("{")][-1])["nlz"]
    for name, v in vals.items():
        assert abs(v - vals["default"]) <= 1e-11 * abs(vals["default"]), (name, vals)


_BLOCK_KERNEL_SCRIPT = r'''
import hashlib, json, sys
import numpy as np, torch
import os, sys
sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import py_schedule as multigpu
coresident = int(sys.argv[1])
eng = multigpu.HipEngine(0)
rng = np.random.default_rng(3)
G = rng.normal(size=(128, 128))
blocks = {"well": G @ G.T + 128 * np.eye(128),
          "ill": (G * np.logspace(0, -5, 128)) @ (G * np.logspace(0, -5, 128)).T + 1e-9 * np.eye(128)}
digest = hashlib.sha256()
for name, A in blocks.items():
    Ain = np.tril(A) + np.triu(rng.normal(size=(128, 128)) * 1e3, 1)      # the kernel reads the lower triangle only
    blk = eng.from_numpy(np.asfortranarray(Ain).T.ravel().copy())
    inv = eng.empty(2 * 128 * 128)
    info = eng.zeros(4, dtype=torch.int32); info.fill_(0x7fffffff)
    eng.factor_panel_co(blk, 128, 128, 0, 128, inv, info, coresident)
    torch.cuda.synchronize()
    out = blk.cpu().numpy().reshape(128, 128).T
    L = np.tril(out)
    iv = inv.cpu().numpy()
    Li, LiT = iv[:128 * 128].reshape(128, 128).T, iv[128 * 128:].reshape(128, 128).T
    Lref = np.linalg.cholesky(A)
    assert int(info.cpu()[0]) == 0x7fffffff
    assert np.abs(L - Lref).max() <= (1e-12 if name == "well" else 1e-7) * np.abs(Lref).max(), name
    assert np.abs(L @ L.T - A).max() <= 1e-13 * np.abs(A).max(), name
    assert np.abs(Li @ L - np.eye(128)).max() <= (1e-12 if name == "well" else 1e-6), name
    assert np.array_equal(LiT, Li.T)
    tiles = np.kron(np.eye(8), np.ones((16, 16))) > 0                       # the diagonal 16 x 16 tiles
    assert np.array_equal(np.triu(out, 1)[tiles], np.zeros(tiles.sum()))    # ... come back cleanly lower
    above = np.triu(np.ones((128, 128)), 1) > 0
    assert np.array_equal(out[above & ~tiles], Ain[above & ~tiles])         # the tiles above them are not touched
    digest.update(L.tobytes()); digest.update(Li.tobytes())
# not positive definite from column 70 (0-based) of a block that starts at global column 256
A = blocks["well"].copy()
A[70, 70] = -1.0
big = eng.zeros(384 * 128)                                       # block column [256, 384) of a 384-row matrix
big.view(128, 384)[:, 256:] = eng.from_numpy(np.asfortranarray(A).T.ravel().copy()).view(128, 128)
inv = eng.empty(2 * 128 * 128)
info = eng.zeros(4, dtype=torch.int32); info.fill_(0x7fffffff)
eng.factor_panel_co(big, 384, 384, 256, 128, inv, info, coresident)
torch.cuda.synchronize()
assert int(info.cpu()[0]) == 256 + 70 + 1, int(info.cpu()[0])
print(json.dumps({"digest": digest.hexdigest()}))
'''


def test_block_kernel_directly():
    """gpak_potrf128_f64 (8 waves) and gpak_potrf128_co_f64 (4 waves, 80 VGPRs) on single 128 x 128 blocks through the
    device-level C-ABI: L, L^-1 and L^-T against NumPy for a well- and an ill-conditioned block, junk in the upper
    triangle ignored, the first non-positive pivot reported (1-based global column), and both builds bit-identical.
    (Subprocesses: torch holds the device buffers here, and its HIP runtime must come up before the library's.)"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    digests = []
    for coresident in (0, 1):
        r = subprocess.run([sys.executable, "-c", _BLOCK_KERNEL_SCRIPT, str(coresident)], cwd=root,
                           env=dict(os.environ, PYTHONPATH=root), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        assert r.returncode == 0, r.stderr.decode()[-3000:]
        digests.append(json.loads([l for l in r.stdout.decode().splitlines() if l.startswith("{")][-1])["digest"])
    assert digests[0] == digests[1]


@pytest.mark.parametrize("d", [3, 4])
def test_symmetric_gram_matvec_matches_the_plain_one(d):
    """f = K alpha on the symmetric pair kernel (N > 15872: every kernel value used for both of its entries) against
    the plain N^2 kernel (GPAK_KMV_SYM=0), both distance forms, 3 and 4 columns, a ragged last macro block: the three
    terms of the nlZ (quad = alpha' f / 2 and sumlp see every entry of f) agree to rounding.  The environment is read
    once per process, hence subprocesses."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import json, sys; sys.path.insert(0, %r)\n"
        "from gp_ss_ak_amd import gpak, synth\n"
        "import numpy as np\n"
        "N = 16000\n"
        "X, y = (synth.drillholes4(N) if %d == 4 else synth.drillholes(N))\n"
        "E = np.array(list(synth.DEFAULT_EXPANS), dtype=np.float64)\n"
        "g = gpak.Gpak(0); g.set_train(X, y); out = {}\n"
        "for name, mode in (('direct', gpak.DIST_DIRECT), ('expansion', gpak.DIST_EXPANSION)):\n"
        "    g.set_params(E, synth.DEFAULT_BIAS, synth.DEFAULT_SN2, mode)\n"
        "    out[name] = list(g.nlz_terms()) + [g.logLikelihood()]\n"
        "print(json.dumps(out))\n") % (root, d)
    res = {}
    for name, env in (("sym", {}), ("plain", {"GPAK_KMV_SYM": "0"})):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), cwd=root, stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, timeout=600)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        res[name] = json.loads([l for l in r.stdout.decode().splitlines() if l.startswith("{")][-1])
    for mode in ("direct", "expansion"):
        for a, b in zip(res["sym"][mode], res["plain"][mode]):
            assert abs(a - b) <= 1e-11 * abs(b), (mode, res)


def test_nan_input_is_not_hidden(gp):
    """A NaN coordinate must surface (NaN entries in K, Chol_fail -> NaN nlZ as at GP_Utils.cpp:1145-1158), not be
    turned into a finite number by the in-line sqrt/exp."""
    X, y = synth.drillholes(200)
    X = np.asfortranarray(X)
    X[17, 1] = np.nan
    gp.set_train(X, y)
    gp.set_params(E, BIAS, SN2, gpak.DIST_DIRECT)
    K = gp.gram()
    # (the pooled mean of MahaDist is NaN too, so like in the reference every entry is affected)
    assert np.all(np.isnan(K[17, :])) and np.all(np.isnan(K[:, 17]))
    assert math.isnan(gp.logLikelihood())
    gp.set_params(E, BIAS, SN2, gpak.DIST_DIRECT)


def test_far_apart_points_flush_to_the_bias(gp):
    """exp(-sqrt(D2)) at the far end of its range: s = 745 is the last value with a non-zero result, and beyond it the
    covariance must be exactly the bias -- also where the table exponent of the fast exp would have wrapped (s > 4.65e7
    gave +inf before the argument was clamped) and where D2 itself overflows to +inf (coordinates of 1e200)."""
    e = E.copy()
    e[[0, 2, 4]] = 0.0                         # zero angles: sigInv = diag(inverse widths)
    e[[1, 3, 5]] = 1.0                         # ... = I: D2 is the plain squared distance
    X, y = synth.drillholes(256)
    gp.set_train(X, y)
    gp.set_params(e, BIAS, SN2, gpak.DIST_DIRECT)
    X1 = np.zeros((1, 3))
    for far in ([740.0, 745.0, 800.0, 1e5, 5e7], [1e12], [1e200]):   # separate calls: the points are centred on their
        far = np.array(far)                                           # pooled mean, which must not swamp the small ones
        X2 = np.zeros((len(far), 3))
        X2[:, 0] = far
        K = gp.compute_k(X1, X2)[0]
        assert np.all(np.isfinite(K))
        if far[0] == 740.0:
            assert abs(K[0] - (e[6] ** 2 * np.exp(-740.0) + BIAS)) <= 1e-300 and np.all(K[1:] == BIAS)
        else:
            assert np.all(K == BIAS)
    # the same evaluation inside the fused Gram-matvec: the predictive mean of a point far from every training point is
    # bias * sum(alpha), its latent variance the prior's
    alpha = gp.solve_alpha()
    Xt = np.array([[5e7, 0.0, 0.0], [0.1, 0.2, 0.3], [-3e9, 4e9, 1e8]])
    mean, var = gp.posteriorMeanVar(Xt)
    assert np.all(np.isfinite(mean)) and np.all(np.isfinite(var))
    assert abs(mean[0] - BIAS * alpha.sum()) <= 1e-9 * max(1.0, abs(BIAS * alpha.sum()))
    assert abs(mean[2] - BIAS * alpha.sum()) <= 1e-9 * max(1.0, abs(BIAS * alpha.sum()))
    gp.set_params(E, BIAS, SN2, gpak.DIST_DIRECT)


def test_options_and_two_live_contexts(orc):
    """gpak_set_option: every schedule variant gives the same numbers (look-ahead off = the classical order on
    one stream; other outer block sizes); value memoisation skips the rebuild only for bit-identical parameters;
    two contexts do not share state."""
    N = 1500
    X, y = synth.drillholes(N)
    Ko = orc.gram(X, X, E, BIAS, orc.DIST_DIRECT)
    info, alpha_o, _ = orc.nlz_refseq(Ko, y, SN2)
    a, b = gpak.Gpak(0), gpak.Gpak(0)
    try:
        a.set_train(X, y)
        b.set_train(X[:700].copy(order="F"), y[:700])
        a.set_params(E, BIAS, SN2, gpak.DIST_DIRECT)
        b.set_params(E, BIAS, SN2, gpak.DIST_DIRECT)
        ref = a.logLikelihood()
        assert abs(ref - info.nlz) <= 1e-9 * abs(info.nlz)
        ib, _, _ = orc.nlz_refseq(np.ascontiguousarray(Ko[:700, :700]), y[:700], SN2)
        assert abs(b.logLikelihood() - ib.nlz) <= 1e-9 * abs(ib.nlz)       # b is untouched by a's work
        # the schedule knobs that used to be GPAK_* environment variables read inside the factorisation are options of
        # the context (include/gpak.h): wide panels from the start (the context has 3000 rows: NB_WIDE_ROWS below that
        # switches them on), no wide panels, tail threshold, block-kernel build, no 512-block inverses
        for opt, val in ((gpak.OPT_LOOKAHEAD, 0), (gpak.OPT_NB_OUTER, 128), (gpak.OPT_NB_OUTER, 256),
                         (gpak.OPT_NB_OUTER, 1024), (gpak.OPT_NB_OUTER, 512), (gpak.OPT_LOOKAHEAD, 1),
                         (gpak.OPT_NB_WIDE_ROWS, 1024), (gpak.OPT_FIRST_NARROW, 0), (gpak.OPT_NB_WIDE, 0),
                         (gpak.OPT_TAIL_ROWS, 0), (gpak.OPT_TAIL_ROWS, 1 << 30), (gpak.OPT_POTRF_CO, 0),
                         (gpak.OPT_POTRF_CO, 2), (gpak.OPT_POTRF_CO, 1), (gpak.OPT_BWD_FUSED, 0), (gpak.OPT_BWD_FUSED, 1), (gpak.OPT_BWD_FUSED, 2), (gpak.OPT_INV512, 0)):
            a.set_option(opt, val)
            a.set_params(E, BIAS, SN2, gpak.DIST_DIRECT)                    # invalidates like the reference
            v = a.logLikelihood()
            assert abs(v - ref) <= 1e-11 * abs(ref), (opt, val)
            assert rel(a.solve_alpha(), alpha_o) <= 1e-8
        a.set_option(gpak.OPT_LOOKAHEAD, 1)
        a.set_option(gpak.OPT_NB_OUTER, 512)
        a.set_option(gpak.OPT_INV512, 1)
        a.set_option(gpak.OPT_BWD_FUSED, 2)
        for opt, val in ((gpak.OPT_NB_OUTER, 100), (gpak.OPT_NB_WIDE, 700), (gpak.OPT_POTRF_CO, 3), (gpak.OPT_BWD_FUSED, 3),
                         (99, 1)):
            with pytest.raises(gpak.GpakError):
                a.set_option(opt, val)                                      # not a multiple of 128 / out of range / unknown
        a.set_option(gpak.OPT_MEMOISE, 1)
        a.set_params(E, BIAS, SN2, gpak.DIST_DIRECT)
        a.logLikelihood()
        t0 = a.timing()["factor_ms"]
        a.set_params(E, BIAS, SN2, gpak.DIST_DIRECT)                        # identical values: nothing is rebuilt
        assert a.logLikelihood() == ref or abs(a.logLikelihood() - ref) <= 1e-11 * abs(ref)
        assert a.timing()["factor_ms"] == t0
        e2 = E.copy()
        e2[1] *= 1.01
        a.set_params(e2, BIAS, SN2, gpak.DIST_DIRECT)                       # a new value: rebuilt
        K2 = orc.gram(X, X, e2, BIAS, orc.DIST_DIRECT)
        i2, _, _ = orc.nlz_lean(K2, y, SN2)
        assert abs(a.logLikelihood() - i2.nlz) <= 1e-9 * abs(i2.nlz)
    finally:
        a.close()
        b.close()


def test_hip_path_matches_committed_golden_vectors(gp):
    """The committed fixtures (tests/golden/*.npz, frozen oracle output: parity unpinned by the reference) against
    the HIP path, without building or calling the oracle on this box."""
    import glob
    import os
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_N*.npz")))
    assert len(files) == 4
    for f in files:
        z = np.load(f)
        X, y, Xte = np.asfortranarray(z["X"]), z["y"], np.asfortranarray(z["Xte"])
        gp.set_train(X, y)
        for name, mode in (("direct", gpak.DIST_DIRECT), ("expansion", gpak.DIST_EXPANSION)):
            direct = mode == gpak.DIST_DIRECT
            gp.set_params(z["expans"], float(z["bias"]), float(z["sn2"]), mode)
            K = gp.gram()
            ij = z[f"{name}_sample_ij"]
            assert np.abs(K[ij[0], ij[1]] - z[f"{name}_K_samples"]).max() <= (1e-13 if direct else 2e-7)
            nlz = gp.logLikelihood()
            assert abs(nlz - float(z[f"{name}_nlz"])) <= (1e-9 if direct else 1e-6) * abs(nlz)
            q, s, l = gp.nlz_terms()
            assert abs(l - float(z[f"{name}_logdet"])) <= (1e-10 if direct else 1e-6) * abs(l)
            mean, var = gp.posteriorMeanVar(Xte)
            assert rel(mean, z[f"{name}_mean"]) <= (1e-8 if direct else 1e-5)
            assert rel(var, z[f"{name}_var"]) <= (1e-8 if direct else 1e-5)
            mean, var = gp.posteriorMeanVar(Xte, compat=3)
            assert rel(var, z[f"{name}_var_compat"]) <= (1e-8 if direct else 1e-5)
            g = gp.GradLL()
            go = z[f"{name}_grad"]
            assert np.abs(g - go).max() <= (1e-8 if direct else 1e-5) * np.abs(go).max()
            if f"{name}_alpha" in z:
                assert rel(gp.solve_alpha(), z[f"{name}_alpha"]) <= (1e-8 if direct else 1e-5)
    gp.set_params(E, BIAS, SN2, gpak.DIST_DIRECT)


@pytest.mark.parametrize("N", [32768, 65536])
def test_full_size_properties(gp, N):
    """BASELINE.json's full sizes (N=32768: the metric; N=65536: configs[3], 34 GB on one GPU), fp64: no CPU
    oracle finishes here in seconds, so the
    step is checked through identities that hold at any size:
      (K + sn2 I) alpha = y   =>   y - f = sn2 alpha  with f = K alpha computed by the device;
      quad = alpha'(f/2),  sumlp = -|y-f|^2/(2 sn2) - N/2 log(2 pi sn2);
      solve_chol is linear and inverts B = I + K/sn2 (checked on alpha itself);
      the step is deterministic (bit-identical when repeated)."""
    X, y = synth.drillholes(N)
    gp.set_train(X, y)
    gp.set_params(E, BIAS, SN2, gpak.DIST_DIRECT)
    nlz1 = gp.logLikelihood()
    alpha = gp.solve_alpha()
    quad, sumlp, logdet = gp.nlz_terms()
    assert np.isfinite(nlz1) and nlz1 == quad - sumlp + logdet
    # |y - f|^2 from sumlp must equal sn2^2 |alpha|^2
    res2 = -2.0 * SN2 * (sumlp + 0.5 * N * math.log(2 * math.pi * SN2))
    assert abs(res2 - SN2 ** 2 * (alpha @ alpha)) <= 1e-7 * res2
    # quad = 1/2 alpha'(y - sn2 alpha)
    assert abs(quad - 0.5 * alpha @ (y - SN2 * alpha)) <= 1e-8 * abs(quad)
    # B^-1 (y/sn2) = alpha, and linearity of the two triangular solves
    r = np.column_stack([y / SN2, np.ones(N), 2.0 * y / SN2 - 3.0 * np.ones(N)])
    S = gp.solve_chol(r)
    assert rel(S[:, 0], alpha) <= 1e-12
    assert rel(S[:, 2], 2.0 * S[:, 0] - 3.0 * S[:, 1]) <= 1e-10
    # log det B lies between N log(1) and N log(1 + kdiag*N/sn2) and the diagonal of L is positive
    assert 0.0 < logdet < 0.5 * N * math.log(1.0 + (E[6] ** 2 + BIAS) * N / SN2)
    # deterministic
    gp.set_params(E, BIAS, SN2, gpak.DIST_DIRECT)
    assert gp.logLikelihood() == nlz1 and np.array_equal(gp.solve_alpha(), alpha)
    # prediction at training points: the latent variance is below the prior and above zero, the mean
    # reproduces y - sn2 alpha (f at the training points)
    idx = np.arange(0, N, 997)
    mean, var = gp.posteriorMeanVar(X[idx])
    assert rel(mean, (y - SN2 * alpha)[idx]) <= 1e-7
    assert np.all(var - SN2 >= 0) and np.all(var - SN2 < E[6] ** 2 + BIAS)
    gp.set_train(X[:64], y[:64])   # release the 8.6 GB buffer for the following tests


@pytest.mark.parametrize("n,sn2", [(3000, 0.016), (3000, 1e-4), (2500, 1e-6)])
def test_back_substitution_modes_by_residual(orc, n, sn2):
    """The three back substitutions (GPAK_OPT_BWD_FUSED 0 / 1 / 2: column dots + diagonal product + sum; far dots under
    the diagonal step; one launch per step with the coupling blocks T_b = L[b,b-1]^T R_b) on well- and ill-conditioned
    systems (cond(B) ~ 1/sn2): each mode's residual |(K + sn2 I) alpha - y| stays within a small factor of what SciPy's
    LAPACK solve leaves on the same matrix -- explicit inverses and their products must not cost accuracy."""
    import scipy.linalg as sla
    X, y = synth.drillholes(n)
    K = orc.gram(X, X, E, BIAS, gpak.DIST_DIRECT)
    A = K + sn2 * np.eye(n)
    a_ref = sla.cho_solve(sla.cho_factor(A, lower=True), y)
    r_ref = np.abs(A @ a_ref - y).max()
    g = gpak.Gpak(0)
    try:
        g.set_train(X, y)
        out = []
        for mode in (0, 1, 2):
            g.set_option(gpak.OPT_BWD_FUSED, mode)
            g.set_params(E, BIAS, sn2, gpak.DIST_DIRECT)
            a = g.solve_alpha()
            r = np.abs(A @ a - y).max()
            out.append((mode, r, np.abs(a - a_ref).max() / np.abs(a_ref).max()))
    finally:
        g.close()
    print(f"\nn={n} sn2={sn2:g}: LAPACK residual {r_ref:.2e}; " + "; ".join(f"mode {m}: residual {r:.2e}, alpha vs LAPACK {d:.1e}" for m, r, d in out))
    for m, r, d in out:
        assert r <= 20 * r_ref + 1e-13 * np.abs(y).max(), (m, r, r_ref)


def test_factor_against_the_vendor_cholesky():
    """An oracle-independent check on the device itself: torch.linalg.cholesky (hipSOLVER / rocSOLVER dpotrf) of the same
    B = I + K / sn2 gives the same factor and the same log-determinant (tools/vendor_potrf.py times the two).  In a
    process of its own: torch brings its own HIP runtime, which must come up BEFORE the library's and never share a
    process that initialised the library first (a double free at interpreter exit otherwise)."""
    import json
    import os
    import subprocess
    import sys
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "vendor_worker.py")
    r = subprocess.run([sys.executable, worker, "3000"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = r.stdout.decode()
    assert r.returncode == 0, out[-2000:]
    res = json.loads([ln for ln in out.splitlines() if ln.startswith("{")][-1])
    assert res["factor_max_abs_diff"] <= 1e-11 * res["factor_max_abs"]
    assert abs(res["logdet_library"] - res["logdet_vendor"]) <= 1e-12 * abs(res["logdet_vendor"])


def test_tuning_environment_matrix():
    """Every schedule / kernel-selection knob that is only reachable through the GPAK_* environment (read once per process,
    re-read by gpak_reload_tuning) in combinations that push the code through its less-travelled branches at small sizes:
    three panel tiers inside a 5000-row matrix, wide back-substitution blocks with the fused step, the scalar-base build
    forced on, split and sub-panel updates of the next block column, a partial last block, other ladders.  nlZ, alpha,
    the gradient and a prediction equal the default build's to rounding."""
    from gp_ss_ak_amd import _lib
    lib = _lib.load()
    Xte = synth.test_points(300)
    cases = [
        {},
        {"GPAK_NB_XWIDE": "1024", "GPAK_NB_XWIDE_ROWS": "3000", "GPAK_NB_WIDE": "768", "GPAK_NB_WIDE_ROWS": "1500", "GPAK_NB_OUTER": "256"},
        {"GPAK_NB_XWIDE": "0", "GPAK_NB_WIDE": "0", "GPAK_TAIL_ROWS": "100000"},
        {"GPAK_BWD_BLOCK": "1024", "GPAK_BWD_FUSED": "2"}, {"GPAK_BWD_BLOCK": "2048", "GPAK_BWD_FUSED": "1"},
        {"GPAK_BWD_BLOCK": "1024", "GPAK_BWD_FUSED": "0"}, {"GPAK_INV512": "0"},
        {"GPAK_SBASE_ROWS": "1"}, {"GPAK_NEXT_SPLIT_ROWS": "100000"}, {"GPAK_SUB_NEXT": "1", "GPAK_TAIL_ROWS": "100000"},
        {"GPAK_NEXT_SPLIT_ROWS": "100000", "GPAK_SUB_NEXT": "1", "GPAK_NB_OUTER": "384"},
        {"GPAK_FS_LEVELS_F32": "128,512"}, {"GPAK_FS_LEVELS_F32": "128,256,1024"}, {"GPAK_GEMM_SMALL": "0"},
        {"GPAK_GEMM_SMALL": "100000", "GPAK_GEMM_SMALL_ROWS": "64"}, {"GPAK_POTRF_CO": "2", "GPAK_LOOKAHEAD": "0"},
        {"GPAK_FWD_IN_FACTOR": "0"}, {"GPAK_TAIL_MASK": "0", "GPAK_BULK_QUEUE": "0"}, {"GPAK_LD_PAD": "0"},
        {"GPAK_SUPER_LR": "2"}, {"GPAK_SUPER_LR": "4", "GPAK_SBASE_ROWS": "1"}, {"GPAK_FILL_FAST": "0", "GPAK_KMV_SYM": "0"},
    ]
    keys = sorted({k for c in cases for k in c})
    saved = {k: os.environ.get(k) for k in keys}
    ref = {}
    try:
        for n in (5000, 1300):
            X, y = synth.drillholes(n)
            for c in cases:
                for k in keys:
                    os.environ.pop(k, None)
                os.environ.update(c)
                lib.gpak_reload_tuning()
                g = gpak.Gpak(0)
                try:
                    g.set_train(X, y)
                    g.set_params(E, BIAS, SN2, gpak.DIST_DIRECT)
                    got = (g.logLikelihood(), g.solve_alpha(), g.GradLL(), g.posteriorMeanVar(Xte))
                finally:
                    g.close()
                if not c:
                    ref[n] = got
                    continue
                r = ref[n]
                assert abs(got[0] - r[0]) <= 1e-11 * abs(r[0]), (n, c)
                assert rel(got[1], r[1]) <= 1e-9, (n, c)
                assert rel(got[2], r[2]) <= 1e-8, (n, c)
                assert rel(got[3][0], r[3][0]) <= 1e-9 and rel(got[3][1], r[3][1]) <= 1e-9, (n, c)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        lib.gpak_reload_tuning()


def test_fp32_kernel_builds_and_ladders():
    """Every build of the fp32 wide-accumulation product the launcher can pick (GPAK_F32_RSD 2 / 4 at two waves per SIMD,
    8 / 16 at one), the plain kernels (GPAK_F32_ACC=plain, both tiles) and other ladders: the GPAK_F32 variance against the
    fp64 context's on the same points, at a size where the long products (K up to 4096) run."""
    from gp_ss_ak_amd import _lib
    lib = _lib.load()
    n, M = 5000, 700
    X, y = synth.drillholes(n)
    Xte = synth.test_points(M)
    g = gpak.Gpak(0)
    try:
        g.set_train(X, y)
        g.set_params(E, BIAS, SN2, gpak.DIST_DIRECT)
        m64, v64 = g.posteriorMeanVar(Xte)
    finally:
        g.close()
    cases = [{}, {"GPAK_F32_RSD": "2"}, {"GPAK_F32_RSD": "8"}, {"GPAK_F32_RSD": "16"}, {"GPAK_F32_ACC": "plain"},
             {"GPAK_F32_ACC": "plain", "GPAK_F32_TILE": "64"}, {"GPAK_FS_LEVELS_F32": "128,512"},
             {"GPAK_FS_LEVELS_F32": "128,256,1024,2048"}, {"GPAK_PRED_BATCH": "512"}, {"GPAK_PRED_LD_SKEW": "0"}]
    keys = sorted({k for c in cases for k in c})
    saved = {k: os.environ.get(k) for k in keys}
    try:
        for c in cases:
            for k in keys:
                os.environ.pop(k, None)
            os.environ.update(c)
            lib.gpak_reload_tuning()
            g32 = gpak.Gpak(0, gpak.F32)
            try:
                g32.set_train(X, y)
                g32.set_params(E, BIAS, SN2, gpak.DIST_DIRECT)
                m32, v32 = g32.posteriorMeanVar(Xte)
            finally:
                g32.close()
            assert rel(m32, m64) <= 1e-9, c
            assert np.abs(v32 - v64).max() <= (2e-4 if c.get("GPAK_F32_ACC") == "plain" else 2e-5) * v64.max(), (c, np.abs(v32 - v64).max() / v64.max())
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        lib.gpak_reload_tuning()


def test_seeded_random_cases_against_lapack(orc):
    """Twenty-four seeded random problems -- sizes from 1 to 3000 (most not multiples of 128 or 512), noise variances from
    1e-5 to 1, length scales, angles, signal variance and bias drawn over two decades, both distance forms -- against
    SciPy's LAPACK on the oracle's Gram: R^T R = B, the residual of alpha at LAPACK's level, nlZ, mean and variance."""
    import scipy.linalg as sla
    rng = np.random.default_rng(20260305)
    g = gpak.Gpak(0)
    try:
        for case in range(24):
            n = int(rng.choice([1, 2, 7, 127, 128, 129, 300, 511, 513, 1000, 1537, 2049, 3000]))
            sn2 = float(10 ** rng.uniform(-5, 0))
            e = np.array([rng.uniform(0, np.pi), 10 ** rng.uniform(-1, 1), rng.uniform(0, np.pi), 10 ** rng.uniform(-1, 1),
                          rng.uniform(0, np.pi), 10 ** rng.uniform(-1, 1), 10 ** rng.uniform(-1, 0.5), 0.6])
            bias = float(10 ** rng.uniform(-3, 0))
            mode = gpak.DIST_DIRECT if case % 3 else gpak.DIST_EXPANSION
            X, y = synth.drillholes(max(n, 8))
            X, y = np.asfortranarray(X[:n]), y[:n].copy()
            Xte = synth.test_points(40)
            g.set_train(X, y)
            g.set_params(e, bias, sn2, mode)
            nlz = g.logLikelihood()
            q, slp, ld = g.nlz_terms()
            alpha = g.solve_alpha()
            R = g.chol_upper()
            mean, var = g.posteriorMeanVar(Xte)
            Ko = orc.gram(X, X, e, bias, mode)
            tag = (case, n, sn2, mode)
            K = g.gram()                  # the linear algebra below is held against the matrix the device itself built ...
            g.set_params(e, bias, sn2, mode)
            assert rel(K, Ko) <= (1e-12 if mode == gpak.DIST_DIRECT else 2e-7), tag   # ... and that against the oracle's
            A = K + sn2 * np.eye(n)
            B = np.eye(n) + K / sn2
            cf = sla.cho_factor(A, lower=True)
            a_ref = sla.cho_solve(cf, y)
            assert rel(R.T @ R, B) <= 1e-12, tag
            assert np.abs(A @ alpha - y).max() <= 50 * np.abs(A @ a_ref - y).max() + 1e-12 * np.abs(y).max(), tag
            ld_ref = np.log(np.diag(sla.cholesky(B, lower=True))).sum()
            assert abs(ld - ld_ref) <= 1e-10 * max(1.0, abs(ld_ref)), tag
            Ks = orc.gram(X, Xte, e, bias, mode)
            m_ref = Ks.T @ a_ref
            # the mean inherits cond(A) ~ 1/sn2 times the rounding of alpha (and, in expansion form, of K): north_star's 1e-5
            assert rel(mean, m_ref) <= 1e-5, tag
            assert np.isfinite(nlz) and np.isfinite(var).all() and (var > -1e-8 * (e[6] ** 2 + bias)).all(), tag
    finally:
        g.close()
