"""ctypes binding of the CPU oracle (oracle/libgpak_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by the product package gp_ss_ak_amd.
PARITY UNPINNED (see gpak_oracle.h): the reference has no fixtures and cannot be
built here; the oracle is pinned by citation, known-answer tests and SciPy.
"""
import ctypes as C
import glob
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

DIST_EXPANSION = 0
DIST_DIRECT = 1
COMPAT_VARCLAMP = 1
COMPAT_SN2SKIP = 2

_dp = C.POINTER(C.c_double)


class NlzInfo(C.Structure):
    _fields_ = [("nlz", C.c_double), ("logdet", C.c_double), ("quad", C.c_double),
                ("sumlp", C.c_double), ("chol_fail", C.c_int), ("n_chol", C.c_int),
                ("n_gemv", C.c_int), ("irls_iters", C.c_int), ("last_step", C.c_double)]


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libgpak_oracle.so")
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
        _LIB.orc_use_lapack.argtypes = [C.c_char_p, C.c_int]
        _LIB.orc_use_lapack.restype = C.c_int
        _LIB.orc_potrf_lower.restype = C.c_int
        _LIB.orc_kdiag.restype = C.c_double
        # a team per visible core oversubscribes a cgroup-limited box by 8x and makes every call take seconds
        try:
            usable = len(os.sched_getaffinity(0))
        except AttributeError:
            usable = os.cpu_count() or 1
        _LIB.orc_set_threads(C.c_int(max(1, min(usable, int(os.environ.get("GPAK_ORACLE_THREADS", "16"))))))
    return _LIB


def _f(a):
    return np.asfortranarray(a, dtype=np.float64)


def _p(a):
    return a.ctypes.data_as(_dp)


def use_lapack(threads=0):
    """Bind the SciPy wheel's OpenBLAS (what Armadillo would call). Returns True on success."""
    try:
        import scipy
    except ImportError:
        return False
    root = os.path.join(os.path.dirname(os.path.dirname(scipy.__file__)), "scipy.libs")
    cands = sorted(glob.glob(os.path.join(root, "libscipy_openblas*.so")))
    if not cands:
        return False
    return lib().orc_use_lapack(cands[0].encode(), int(threads)) == 0


def use_builtin():
    lib().orc_use_lapack(None, 0)


def lapack_active():
    return bool(lib().orc_lapack_active())


def siginv(par, d=3):
    par = np.ascontiguousarray(par, dtype=np.float64)
    A = np.zeros((d, d), order="F")
    lib().orc_siginv(C.c_int(d), _p(par), _p(A))
    return A


def mahadist(X1, X2, par, mode=DIST_EXPANSION):
    X1, X2 = _f(X1), _f(X2)
    n, d = X1.shape
    m = X2.shape[0]
    par = np.ascontiguousarray(par, dtype=np.float64)
    D2 = np.zeros((n, m), order="F")
    lib().orc_mahadist(_p(X1), C.c_int(n), _p(X2), C.c_int(m), C.c_int(d), _p(par), C.c_int(mode), _p(D2))
    return D2


def gram(X1, X2, expans, bias, mode=DIST_EXPANSION, want_d2=False):
    X1, X2 = _f(X1), _f(X2)
    n, d = X1.shape
    m = X2.shape[0]
    e = np.ascontiguousarray(expans, dtype=np.float64)
    K = np.zeros((n, m), order="F")
    D2 = np.zeros((n, m), order="F") if want_d2 else None
    lib().orc_gram(_p(X1), C.c_int(n), _p(X2), C.c_int(m), C.c_int(d), _p(e), C.c_double(bias),
                   C.c_int(mode), _p(K), _p(D2) if want_d2 else None)
    return (K, D2) if want_d2 else K


def gram_hyb(X1, X2, terms, bias, white, mode=DIST_EXPANSION, want_d2=False):
    """terms = [(kind, params)], kind 0 ExpAns / 1 Exp / 2 RBF."""
    X1, X2 = _f(X1), _f(X2)
    n, d = X1.shape
    m = X2.shape[0]
    kinds = (C.c_int * len(terms))(*[int(k) for k, _ in terms])
    pars = np.ascontiguousarray(np.concatenate([np.asarray(p, dtype=np.float64) for _, p in terms]))
    K = np.zeros((n, m), order="F")
    D2 = np.zeros((n, m), order="F") if want_d2 else None
    lib().orc_gram_hyb(_p(X1), C.c_int(n), _p(X2), C.c_int(m), C.c_int(d), C.c_int(len(terms)), kinds, _p(pars),
                       C.c_double(bias), C.c_double(white), C.c_int(mode), _p(K), _p(D2) if want_d2 else None)
    return (K, D2) if want_d2 else K


def kdiag(expans, bias):
    e = np.ascontiguousarray(expans, dtype=np.float64)
    return lib().orc_kdiag(_p(e), C.c_double(bias))


def potrf_lower(A):
    """Returns (L, info); L lower with zeroed upper."""
    A = np.array(A, dtype=np.float64, order="F", copy=True)
    n = A.shape[0]
    info = lib().orc_potrf_lower(C.c_int(n), _p(A), C.c_int(n))
    return np.tril(A), info


def solve_chol(L, X):
    L = _f(L)
    X = np.array(X, dtype=np.float64, order="F", copy=True)
    X2 = X.reshape(L.shape[0], -1, order="F")
    lib().orc_solve_chol(C.c_int(L.shape[0]), _p(L), C.c_int(L.shape[0]), _p(X2), C.c_int(X2.shape[1]),
                         C.c_int(L.shape[0]))
    return X2.reshape(X.shape, order="F")


def _nlz(fn, K, y, sn2, alpha0, want_L):
    K = _f(K)
    N = K.shape[0]
    y = np.ascontiguousarray(y, dtype=np.float64).reshape(N)
    alpha = np.zeros(N) if alpha0 is None else np.array(alpha0, dtype=np.float64).reshape(N).copy()
    L = np.zeros((N, N), order="F") if want_L else None
    info = NlzInfo()
    fn(C.c_int(N), _p(K), _p(y), C.c_double(sn2), _p(alpha), _p(L) if want_L else None, C.byref(info))
    return info, alpha, L


def nlz_refseq(K, y, sn2, alpha0=None, want_L=True):
    """Reference operation sequence (IRLS + Brent + 3 Choleskys)."""
    return _nlz(lib().orc_nlz_refseq, K, y, sn2, alpha0, want_L)


def nlz_lean(K, y, sn2, want_L=True):
    """One Cholesky + two triangular solves."""
    return _nlz(lib().orc_nlz_lean, K, y, sn2, None, want_L)


def predict(Xtr, Xte, expans, bias, sn2, alpha, L, mode=DIST_EXPANSION, compat=0, want_var=True):
    Xtr, Xte, L = _f(Xtr), _f(Xte), _f(L)
    N, d = Xtr.shape
    M = Xte.shape[0]
    e = np.ascontiguousarray(expans, dtype=np.float64)
    alpha = np.ascontiguousarray(alpha, dtype=np.float64)
    mean = np.zeros(M)
    var = np.zeros(M) if want_var else None
    lib().orc_predict(_p(Xtr), C.c_int(N), _p(Xte), C.c_int(M), C.c_int(d), _p(e), C.c_double(bias),
                      C.c_double(sn2), C.c_int(mode), _p(alpha), _p(L), C.c_int(compat), _p(mean),
                      _p(var) if want_var else None)
    return mean, var


def grad_ref(X, y, K, L, alpha, expans, bias, sn2, mode=DIST_EXPANSION):
    X, K, L = _f(X), _f(K), _f(L)
    N = X.shape[0]
    y = np.ascontiguousarray(y, dtype=np.float64)
    alpha = np.ascontiguousarray(alpha, dtype=np.float64)
    e = np.ascontiguousarray(expans, dtype=np.float64)
    g = np.zeros(10)
    lib().orc_grad_ref_d(_p(X), C.c_int(N), C.c_int(X.shape[1]), _p(y), _p(K), _p(L), _p(alpha), _p(e),
                         C.c_double(bias), C.c_double(sn2), C.c_int(mode), _p(g))
    return g


def grad_ref_q(X, y, Q, alpha, expans, bias, sn2, mode=DIST_DIRECT):
    """orc_grad_ref_q: the as-written gradient from a caller-supplied Q = B^-1 (F-ordered, may be a view with a
    leading dimension), everything else rebuilt slab by slab."""
    X = _f(X)
    N = X.shape[0]
    assert Q.dtype == np.float64 and Q.flags.f_contiguous or Q.strides[0] == 8
    ldq = Q.strides[1] // 8
    y = np.ascontiguousarray(y, dtype=np.float64)
    alpha = np.ascontiguousarray(alpha, dtype=np.float64)
    e = np.ascontiguousarray(expans, dtype=np.float64)
    g = np.zeros(10)
    lib().orc_grad_ref_q(_p(X), C.c_int(N), C.c_int(X.shape[1]), _p(y), _p(Q), C.c_size_t(ldq), _p(alpha), _p(e),
                         C.c_double(bias), C.c_double(sn2), C.c_int(mode), _p(g))
    return g


def grad_hyb(X, y, K, L, alpha, terms, has_bias, sn2, mode=DIST_EXPANSION):
    X, K, L = _f(X), _f(K), _f(L)
    N = X.shape[0]
    y = np.ascontiguousarray(y, dtype=np.float64)
    alpha = np.ascontiguousarray(alpha, dtype=np.float64)
    kinds = (C.c_int * len(terms))(*[int(k) for k, _ in terms])
    pars = np.ascontiguousarray(np.concatenate([np.asarray(p, dtype=np.float64) for _, p in terms]))
    ng = sum({0: 8, 1: 2, 2: 3}[int(k)] for k, _ in terms) + (1 if has_bias else 0) + 1
    g = np.zeros(ng)
    lib().orc_grad_hyb_d(_p(X), C.c_int(N), C.c_int(X.shape[1]), _p(y), _p(K), _p(L), _p(alpha), C.c_int(len(terms)),
                         kinds, _p(pars), C.c_int(1 if has_bias else 0), C.c_double(sn2), C.c_int(mode), _p(g))
    return g
