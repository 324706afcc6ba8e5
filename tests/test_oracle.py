"""CPU tests (-m "not gpu") of the oracle: known-answer tests, an independent
NumPy/SciPy(LAPACK) cross-check, and the committed golden vectors.

The reference has no tests or fixtures and cannot be built here, so parity of the
oracle itself is UNPINNED by the reference; these tests are what pins it.
"""
import glob
import os

import numpy as np
import pytest
import scipy.linalg as sl

from gp_ss_ak_amd import synth

E = np.array(synth.DEFAULT_EXPANS)
BIAS, SN2 = synth.DEFAULT_BIAS, synth.DEFAULT_SN2
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def np_rot(al, be, te):
    """R(alpha,beta,teta) as filled at Kernel.cpp:1402-1410."""
    ca, sa, cb, sb, ct, st = np.cos(al), np.sin(al), np.cos(be), np.sin(be), np.cos(te), np.sin(te)
    return np.array([[ca * ct + sa * sb * st, -sa * ct + ca * sb * st, -cb * st],
                     [sa * cb, ca * cb, sb],
                     [ca * st - sa * sb * ct, -sa * st - ca * sb * ct, cb * ct]])


def np_gram(X1, X2, e, bias):
    """Independent NumPy statement of SURVEY.md section 0 (direct distance)."""
    R = np_rot(e[0], e[2], e[4])
    A = R @ np.diag([e[1], e[3], e[5]]) @ R.T
    if X1.shape[1] == 4:                       # rock-type column: A33 = InversewidthR (Kernel.cpp:1411-1424)
        A = np.block([[A, np.zeros((3, 1))], [np.zeros((1, 3)), np.array([[e[7]]])]])
    U, V = X1 @ A, X2 @ A
    D2 = ((U[:, None, :] - V[None, :, :]) ** 2).sum(-1)
    return e[6] ** 2 * np.exp(-np.sqrt(D2)) + bias, D2


def test_siginv_matches_numpy_and_is_symmetric(orc):
    par = [E[0], E[2], E[4], E[1], E[3], E[5]]
    A = orc.siginv(par)
    R = np_rot(E[0], E[2], E[4])
    assert np.allclose(A, R @ np.diag([E[1], E[3], E[5]]) @ R.T, rtol=0, atol=1e-15)
    assert np.allclose(A, A.T, rtol=0, atol=1e-16)
    assert np.allclose(R @ R.T, np.eye(3), atol=1e-15)  # Rot is a rotation


@pytest.mark.parametrize("N", [1, 2, 8, 65, 300])
def test_gram_matches_numpy_and_diag(orc, N):
    X, y = synth.drillholes(max(N, 2))
    X = X[:N]
    Kn, D2n = np_gram(X, X, E, BIAS)
    Kd, D2d = orc.gram(X, X, E, BIAS, orc.DIST_DIRECT, want_d2=True)
    Ke, D2e = orc.gram(X, X, E, BIAS, orc.DIST_EXPANSION, want_d2=True)
    assert np.abs(D2d - D2n).max() <= 1e-14
    assert np.abs(Kd - Kn).max() <= 1e-14
    assert np.all(Kd.diagonal() == E[6] ** 2 + BIAS)           # K_ii = sigma^2 + bias, exactly
    assert np.abs(Ke.diagonal() - (E[6] ** 2 + BIAS)).max() <= 1e-7  # expansion: cancellation noise
    assert np.abs(D2e - D2n).max() <= 1e-13
    assert (D2e >= 0).all()                                   # clamp, Kernel.cpp:1433-1434
    assert orc.kdiag(E, BIAS) == E[6] ** 2 + BIAS             # bias un-squared, sigma squared (Q1)


def test_zero_angles_give_axis_aligned_metric(orc):
    X, _ = synth.drillholes(40)
    e = E.copy()
    e[[0, 2, 4]] = 0.0
    _, D2 = orc.gram(X, X, e, BIAS, orc.DIST_DIRECT, want_d2=True)
    d = X[:, None, :] - X[None, :, :]
    # inverse widths act SQUARED (Q2): D2 = sum_k L_k^2 delta_k^2
    ref = (e[1] * d[..., 0]) ** 2 + (e[3] * d[..., 1]) ** 2 + (e[5] * d[..., 2]) ** 2
    assert np.abs(D2 - ref).max() <= 1e-14


def test_alpha_invariance_when_lx_equals_ly_and_translation_invariance(orc):
    X, _ = synth.drillholes(50)
    e1, e2 = E.copy(), E.copy()
    e1[1] = e1[3] = 1.4
    e2[1] = e2[3] = 1.4
    e2[0] = 0.123  # AngleX rotates columns 0,1 of Rot into each other
    K1 = orc.gram(X, X, e1, BIAS, orc.DIST_DIRECT)
    K2 = orc.gram(X, X, e2, BIAS, orc.DIST_DIRECT)
    assert np.abs(K1 - K2).max() <= 1e-14
    for mode in (orc.DIST_DIRECT, orc.DIST_EXPANSION):
        Ka = orc.gram(X, X, E, BIAS, mode)
        Kb = orc.gram(X + np.array([3.0, -2.0, 0.5]), X + np.array([3.0, -2.0, 0.5]), E, BIAS, mode)
        assert np.abs(Ka - Kb).max() <= (1e-13 if mode == orc.DIST_DIRECT else 5e-7)


def test_cross_gram_uses_pooled_mean_but_value_is_translation_free(orc):
    X, _ = synth.drillholes(30)
    Xt = synth.test_points(7)
    Kn, _ = np_gram(X, Xt, E, BIAS)
    assert np.abs(orc.gram(X, Xt, E, BIAS, orc.DIST_DIRECT) - Kn).max() <= 1e-14
    assert np.abs(orc.gram(X, Xt, E, BIAS, orc.DIST_EXPANSION) - Kn).max() <= 1e-7


@pytest.mark.parametrize("N", [1, 2])
def test_closed_form_nlz_tiny(orc, N):
    X, y = synth.drillholes(2)
    X, y = X[:N], y[:N]
    K = orc.gram(X, X, E, BIAS, orc.DIST_DIRECT)
    C = K + SN2 * np.eye(N)
    want = 0.5 * y @ np.linalg.solve(C, y) + 0.5 * np.log(np.linalg.det(C)) + 0.5 * N * np.log(2 * np.pi)
    info, alpha, L = orc.nlz_refseq(K, y, SN2)
    assert abs(info.nlz - want) <= 1e-12 * max(1, abs(want))
    info2, _, _ = orc.nlz_lean(K, y, SN2)
    assert abs(info2.nlz - want) <= 1e-12 * max(1, abs(want))


@pytest.mark.parametrize("N", [8, 64, 400])
def test_refseq_equals_closed_form_and_scipy(orc, N):
    X, y = synth.drillholes(N)
    K = orc.gram(X, X, E, BIAS, orc.DIST_DIRECT)
    info, alpha, L = orc.nlz_refseq(K, y, SN2)
    # operation counts of the reference sequence: 2 IRLS Choleskys + 1 in ldB2_exact()
    assert info.n_chol == 3 and info.irls_iters == 2 and not info.chol_fail
    assert 20 <= info.n_gemv <= 27
    C = K + SN2 * np.eye(N)
    cf = sl.cho_factor(C, lower=True)
    a_s = sl.cho_solve(cf, y)
    want = 0.5 * y @ a_s + np.log(np.diag(cf[0])).sum() + 0.5 * N * np.log(2 * np.pi)
    assert np.abs(alpha - a_s).max() <= 1e-9 * np.abs(a_s).max()      # IRLS fixed point == closed form
    assert abs(info.nlz - want) <= 1e-11 * abs(want)
    lean, a_l, L2 = orc.nlz_lean(K, y, SN2)
    assert abs(lean.nlz - info.nlz) <= 1e-12 * abs(info.nlz)
    assert np.abs(a_l - a_s).max() <= 1e-10 * np.abs(a_s).max()
    B = np.eye(N) + K / SN2
    assert np.abs(L @ L.T - B).max() <= 1e-12 * np.abs(B).max()       # R'R = B
    assert np.abs(L - sl.cholesky(B, lower=True)).max() <= 1e-11 * np.abs(L).max()
    assert abs(info.logdet - np.log(np.diag(L)).sum()) <= 1e-12


def test_warm_start_takes_one_iteration(orc):
    X, y = synth.drillholes(64)
    K = orc.gram(X, X, E, BIAS, orc.DIST_DIRECT)
    info, alpha, _ = orc.nlz_refseq(K, y, SN2)
    info2, alpha2, _ = orc.nlz_refseq(K, y, SN2, alpha0=alpha)
    assert info2.irls_iters == 1 and info2.n_chol == 2
    assert abs(info2.nlz - info.nlz) <= 1e-10 * abs(info.nlz)


def test_chol_fail_gives_nan(orc):
    X, y = synth.drillholes(16)
    K = orc.gram(X, X, E, BIAS, orc.DIST_DIRECT)
    info, _, _ = orc.nlz_refseq(K, y, -0.5)
    assert info.chol_fail and np.isnan(info.nlz)                     # GP_Utils.cpp:1145-1146
    L, status = orc.potrf_lower(-np.eye(4))
    assert status == 1


@pytest.mark.parametrize("N", [130, 777])
def test_blocked_cholesky_builtin_vs_lapack(orc, N):
    rng = np.random.default_rng(N)
    G = rng.normal(size=(N, N))
    A = G @ G.T + N * np.eye(N)
    orc.use_builtin()
    L1, s1 = orc.potrf_lower(A)
    assert s1 == 0 and np.abs(L1 @ L1.T - A).max() <= 1e-12 * np.abs(A).max()
    rhs = rng.normal(size=(N, 3))
    x1 = orc.solve_chol(L1, rhs)
    assert np.abs(A @ x1 - rhs).max() <= 1e-10
    if orc.use_lapack(2):
        L2, s2 = orc.potrf_lower(A)
        assert s2 == 0 and np.abs(L1 - L2).max() <= 1e-11 * np.abs(L1).max()
        x2 = orc.solve_chol(L2, rhs)
        assert np.abs(x1 - x2).max() <= 1e-11
        orc.use_builtin()


def test_predict_against_numpy(orc):
    N, M = 120, 9
    X, y = synth.drillholes(N)
    Xt = synth.test_points(M)
    K = orc.gram(X, X, E, BIAS, orc.DIST_DIRECT)
    info, alpha, L = orc.nlz_lean(K, y, SN2)
    kX, _ = np_gram(X, Xt, E, BIAS)
    C = K + SN2 * np.eye(N)
    mean_ref = kX.T @ np.linalg.solve(C, y)
    var_ref = (E[6] ** 2 + BIAS) - np.einsum("ij,ij->j", kX, np.linalg.solve(C, kX)) + SN2
    mean, var = orc.predict(X, Xt, E, BIAS, SN2, alpha, L, orc.DIST_DIRECT, 0)
    assert np.abs(mean - mean_ref).max() <= 1e-9
    assert np.abs(var - var_ref).max() <= 1e-9
    # Q3: the reference's "clamp" zeroes element 0 of the latent variance -> reported var = sn2
    _, var_c = orc.predict(X, Xt, E, BIAS, SN2, alpha, L, orc.DIST_DIRECT, orc.COMPAT_VARCLAMP)
    assert var_c[0] == SN2 and np.allclose(var_c[1:], var[1:], rtol=0, atol=0)
    # Q4: "+ sn2" is skipped when sn2 == 1.0 exactly
    info1, alpha1, L1 = orc.nlz_lean(K, y, 1.0)
    _, v_skip = orc.predict(X, Xt, E, BIAS, 1.0, alpha1, L1, orc.DIST_DIRECT, orc.COMPAT_SN2SKIP)
    _, v_add = orc.predict(X, Xt, E, BIAS, 1.0, alpha1, L1, orc.DIST_DIRECT, 0)
    assert np.allclose(v_add - v_skip, 1.0)
    # prediction AT the training points: latent variance -> sn2*(...) small, mean -> close to y
    m_tr, v_tr = orc.predict(X, X[:5], E, BIAS, SN2, alpha, L, orc.DIST_DIRECT, 0)
    assert np.all(v_tr - SN2 < SN2) and np.all(v_tr - SN2 >= 0)


def np_grad_as_written(X, y, e, bias, sn2):
    """NumPy restatement of GradLL/getGradients with full N x N temporaries, mirroring the
    Armadillo expressions (Kernel.cpp:886-1263, GP_Utils.cpp:1164-1262) -- independent of the C."""
    N = len(y)
    K, D2 = np_gram(X, X, e, bias)
    C = K + sn2 * np.eye(N)
    alpha = np.linalg.solve(C, y)
    Binv = np.linalg.inv(np.eye(N) + K / sn2)
    Q = Binv                                   # (B^-1 diag(sW)) % ((1/sW) 1')
    dW = 0.5 * (Q * K).sum(1)
    QW = Q / sn2 - np.outer(alpha, alpha)
    al, be, te = e[0], e[2], e[4]
    iw = np.array([e[1], e[3], e[5]])
    R = np_rot(al, be, te)
    h = 1e-6

    def drot(i):  # analytic derivatives as written are standard d/d(angle); verify by central difference
        a = [al, be, te]
        ap, am = list(a), list(a)
        ap[i] += h
        am[i] -= h
        return (np_rot(*ap) - np_rot(*am)) / (2 * h)

    S = R @ np.diag(iw) @ R.T
    g = np.zeros(10)
    SD = np.sqrt(D2)
    with np.errstate(divide="ignore", invalid="ignore"):
        dk = np.where(SD == 0, 0.0, np.exp(-SD) * (-0.5 / SD))
    np.fill_diagonal(dk, 0.0)
    Rm = e[6] ** 2 * QW * dk
    for a in range(3):
        dR = drot(a)
        Sa = dR @ np.diag(iw) @ R.T + R @ np.diag(iw) @ dR.T
        Sa[0, 0] -= iw[2] * R[0, 2] * dR[0, 2]          # the missing "2 *" on the z-term of S_x(0,0)
        SL = np.outer(R[:, a], R[:, a])
        for slot, Sp in ((2 * a, Sa), (2 * a + 1, SL)):
            Mp = S * Sp
            X3 = X[:, :3]                                # S_p(3,3) = 0: the 4th column drops out (:1169-1173)
            av = (2 * X3 * X3) @ Mp
            Di2 = av.sum(1)[:, None] + av.sum(1)[None, :] - 4 * X3 @ Mp @ X3.T
            g[slot] = (Rm * Di2).sum()
    g[6] = 2 * (np.exp(-SD) * QW).sum() * e[6]
    g[7] = 0.0
    if X.shape[1] == 4:                                  # Kernel.cpp:1246-1255, weight = KD2 as written
        x4 = X[:, 3]
        Di2 = 2 * x4[:, None] ** 2 + 2 * x4[None, :] ** 2 - 4 * np.outer(x4, x4)
        g[7] = -2 * (np.exp(-SD) * Di2).sum() / N
    g[8] = np.trace(QW)
    f = K @ alpha
    g[9] = -(2 / sn2) * dW.sum() - (((y - f) ** 2) / sn2 - 1).sum()
    return g


def test_reference_style_gradient_against_numpy(orc):
    N = 60
    X, y = synth.drillholes(N)
    K = orc.gram(X, X, E, BIAS, orc.DIST_DIRECT)
    info, alpha, L = orc.nlz_lean(K, y, SN2)
    g = orc.grad_ref(X, y, K, L, alpha, E, BIAS, SN2, orc.DIST_DIRECT)
    gn = np_grad_as_written(X, y, E, BIAS, SN2)
    assert g[7] == 0.0
    scale = np.abs(gn).max()
    assert np.abs(g - gn).max() <= 2e-6 * scale   # angle slots use a central difference of Rot (h=1e-6)
    assert np.abs(g[[1, 3, 5, 6, 8, 9]] - gn[[1, 3, 5, 6, 8, 9]]).max() <= 1e-9 * scale


@pytest.mark.parametrize("N,d", [(300, 3), (900, 3), (500, 4)])
def test_streamed_gradient_equals_the_full_restatement(orc, N, d):
    """orc_grad_ref_q (Q = B^-1 supplied by the caller, K / DD2 / QW / R rebuilt slab by slab -- what
    tests/golden/make_golden_grad.py uses at N = 32768, where the six N x N arrays of orc_grad_ref_d do not fit) against
    orc_grad_ref_d itself, with Q from NumPy's inverse: same sums, different association."""
    X, y = (synth.drillholes4(N) if d == 4 else synth.drillholes(N))
    K = orc.gram(X, X, E, BIAS, orc.DIST_DIRECT)
    info, alpha, L = orc.nlz_lean(K, y, SN2)
    g = orc.grad_ref(X, y, K, L, alpha, E, BIAS, SN2, orc.DIST_DIRECT)
    Q = np.asfortranarray(np.linalg.inv(np.eye(N) + K / SN2))
    gq = orc.grad_ref_q(X, y, Q, alpha, E, BIAS, SN2, orc.DIST_DIRECT)
    assert np.abs(g - gq).max() <= 1e-11 * np.abs(g).max()
    assert (gq[7] != 0.0) == (d == 4)
    # a strided view (leading dimension > N) is accepted too
    Qbig = np.zeros((N + 8, N), order="F")
    Qbig[:N] = Q
    assert np.abs(orc.grad_ref_q(X, y, Qbig[:N], alpha, E, BIAS, SN2, orc.DIST_DIRECT) - gq).max() <= 1e-13 * np.abs(g).max()


def test_round3_goldens_are_consistent_with_the_earlier_ones():
    """The gradient goldens were generated from their own factorisation: alpha must be the alpha of the nlZ goldens; the
    L-BFGS fixture starts from the default parameters and never increases its kept objective."""
    import json
    for N in (8192, 32768):
        a = json.load(open(os.path.join(GOLD, f"golden_N{N}.json")))["direct"]
        g = json.load(open(os.path.join(GOLD, f"golden_grad_N{N}.json")))
        assert abs(a["alpha_norm"] - g["alpha_norm"]) <= 1e-10 * a["alpha_norm"]
        assert g["expans"] == [float(v) for v in E] and g["bias"] == BIAS and g["sn2"] == SN2 and len(g["g"]) == 10
        assert g["g"][7] == 0.0 and g["q_symmetry_defect"] <= 1e-12
    z = json.load(open(os.path.join(GOLD, "golden_lbfgs_N8192.json")))
    obj = [r["objective"] for r in z["rows"]]
    assert z["x0"] == [float(v) for v in E] + [BIAS, SN2] and all(b <= a for a, b in zip(obj, obj[1:]))
    assert [r["evaluations"] for r in z["rows"]] == sorted(r["evaluations"] for r in z["rows"])
    assert z["first_stall_iteration"] == 2 and len(z["rows"]) == z["maxit"] == 6


def test_four_column_inputs_gram_gradient_prediction(orc):
    """a 4th (rock-type) input column with its own inverse width."""
    N = 70
    X, y = synth.drillholes4(N)
    assert X.shape == (N, 4) and abs(X[:, 3]).max() == 1.0 and len(np.unique(X[:, 3])) == 4
    Kn, D2n = np_gram(X, X, E, BIAS)
    for mode, tol in ((orc.DIST_DIRECT, 1e-13), (orc.DIST_EXPANSION, 1e-6)):
        K, D2 = orc.gram(X, X, E, BIAS, mode, want_d2=True)
        assert np.abs(K - Kn).max() <= tol
    # the 4th column matters: same coordinates, different rock type => smaller covariance
    X2 = X.copy()
    X2[:, 3] = -X2[:, 3]
    assert np.abs(orc.gram(X, X2, E, BIAS, orc.DIST_DIRECT) - Kn).max() > 1e-3
    K = orc.gram(X, X, E, BIAS, orc.DIST_DIRECT)
    info, alpha, L = orc.nlz_lean(K, y, SN2)
    g = orc.grad_ref(X, y, K, L, alpha, E, BIAS, SN2, orc.DIST_DIRECT)
    gn = np_grad_as_written(X, y, E, BIAS, SN2)
    scale = np.abs(gn).max()
    assert g[7] != 0.0 and abs(g[7] - gn[7]) <= 1e-10 * abs(gn[7])
    assert np.abs(g - gn).max() <= 2e-6 * scale
    Xt = synth.test_points4(25)
    mean, var = orc.predict(X, Xt, E, BIAS, SN2, alpha, L, orc.DIST_DIRECT)
    kX, _ = np_gram(X, Xt, E, BIAS)
    assert np.abs(mean - kX.T @ alpha).max() <= 1e-10
    C = K + SN2 * np.eye(N)
    vn = (E[6] ** 2 + BIAS) - np.einsum("ij,ij->j", kX, np.linalg.solve(C, kX)) + SN2
    assert np.abs(var - vn).max() <= 1e-9


def test_golden_vectors_still_match_the_oracle(orc):
    files = sorted(glob.glob(os.path.join(GOLD, "golden_N*.npz")))
    assert len(files) == 4   # N = 8, 64, 512 (3-D) and N = 64 with a rock-type column
    orc.use_builtin()
    for f in files:
        z = np.load(f)  # allow_pickle=False by default
        X, y, Xte = z["X"], z["y"], z["Xte"]
        for name, mode in (("direct", orc.DIST_DIRECT), ("expansion", orc.DIST_EXPANSION)):
            K = orc.gram(X, X, z["expans"], float(z["bias"]), mode)
            ij = z[f"{name}_sample_ij"]
            assert np.abs(K[ij[0], ij[1]] - z[f"{name}_K_samples"]).max() <= 1e-15
            assert abs(K.sum() - float(z[f"{name}_K_sum"])) <= 1e-12 * abs(K.sum())
            info, alpha, L = orc.nlz_refseq(K, y, float(z["sn2"]))
            assert abs(info.nlz - float(z[f"{name}_nlz"])) <= 1e-12 * abs(info.nlz)
            mean, var = orc.predict(X, Xte, z["expans"], float(z["bias"]), float(z["sn2"]), alpha, L, mode, 0)
            assert np.abs(mean - z[f"{name}_mean"]).max() <= 1e-11
            assert np.abs(var - z[f"{name}_var"]).max() <= 1e-11
            if f"{name}_K" in z:
                assert np.abs(K - z[f"{name}_K"]).max() <= 1e-15
                assert np.abs(alpha - z[f"{name}_alpha"]).max() <= 1e-10 * np.abs(alpha).max()


def test_synthetic_generator_is_deterministic_and_standardised():
    X1, y1 = synth.drillholes(300)
    X2, y2 = synth.drillholes(300)
    assert np.array_equal(X1, X2) and np.array_equal(y1, y2)
    assert abs(X1.max() - 1) < 1e-12 and abs(X1.min() + 1) < 1e-12   # one common centre/half-range
    assert abs(y1.max() - 1) < 1e-12 and abs(y1.min() + 1) < 1e-12
    Xr, yr = synth.drillholes_raw(300)
    Xs, ys, params = synth.symmetric_standardise(Xr, yr)
    # round trip of Control::postData (Control.cpp:213-218)
    assert np.abs(Xs * params[1:, 1] + params[1:, 0] - Xr).max() <= 1e-9
    assert np.abs(ys * params[0, 1] + params[0, 0] - yr).max() <= 1e-12
    assert len({params[1, 0], params[2, 0], params[3, 0]}) == 1      # geometry preserved


def test_other_kernels_match_numpy(orc):
    """f-4: Kern_Exponential / Kern_RBF over EuclDist, Kern_White, and their sum with ExpAns + Bias."""
    X, _ = synth.drillholes(90)
    Xt = synth.test_points(11)
    hyp_e, sig_e, hyp_r, iw_r, sig_r, white = 0.5, 0.9, 0.5, 0.9, 0.5, 0.10   # Kernel.cpp:586-590, 424-429, 214-217
    terms = [(0, E), (1, [hyp_e, sig_e]), (2, [hyp_r, iw_r, sig_r])]
    d = X[:, None, :] - X[None, :, :]
    r2 = (d ** 2).sum(-1)
    Kx, _ = np_gram(X, X, E, 0.0)
    want = Kx + sig_e ** 2 * np.exp(-np.sqrt(r2 / hyp_e ** 2)) + sig_r ** 2 * np.exp(-0.5 * iw_r * r2 / hyp_r ** 2) \
        + BIAS + white * np.eye(90)
    for mode, tol in ((orc.DIST_DIRECT, 1e-13), (orc.DIST_EXPANSION, 5e-7)):
        K = orc.gram_hyb(X, X, terms, BIAS, white, mode)
        assert np.abs(K - want).max() <= tol
    # cross block: no white noise (X1(0) != X2(0)), Kernel.cpp:260-262
    dc = X[:, None, :] - Xt[None, :, :]
    rc = (dc ** 2).sum(-1)
    Kc, _ = np_gram(X, Xt, E, 0.0)
    wantc = Kc + sig_e ** 2 * np.exp(-np.sqrt(rc / hyp_e ** 2)) + sig_r ** 2 * np.exp(-0.5 * iw_r * rc / hyp_r ** 2) + BIAS
    assert np.abs(orc.gram_hyb(X, Xt, terms, BIAS, white, orc.DIST_DIRECT) - wantc).max() <= 1e-13
