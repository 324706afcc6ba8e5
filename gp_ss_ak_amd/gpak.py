"""Python face of the C-ABI (include/gpak.h), used by tests/, bench.py and the smoke test.

Method names follow the reference members they stand in for (GP_utils::logLikelihood,
GP_utils::posteriorMeanVar, Kernels::computeK, ...); every call goes through
libgpak_hip.so -- nothing is computed in Python.
"""
import ctypes as C
import math

import numpy as np

from . import _lib

OK, ENOTPD, EINVAL, ESTATE, ENOMEM, ENOTIMPL, EHIP = 0, 1, 2, 3, 4, 5, -1
F64, F32 = 0, 1
DIST_EXPANSION, DIST_DIRECT = 0, 1
COMPAT_VARCLAMP, COMPAT_SN2SKIP = 1, 2
OPT_MEMOISE, OPT_NB_OUTER, OPT_PROFILE, OPT_LOOKAHEAD = 1, 2, 3, 4
OPT_NB_WIDE, OPT_NB_WIDE_ROWS, OPT_TAIL_ROWS, OPT_FIRST_NARROW, OPT_INV512, OPT_POTRF_CO, OPT_PRED_BATCH = 5, 6, 7, 8, 9, 10, 11
OPT_BWD_FUSED = 12
KERN_EXPANS, KERN_EXP, KERN_RBF = 0, 1, 2

_dp = C.POINTER(C.c_double)


class GpakError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__(f"gpak status {status}: {msg}")
        self.status = status


def _p(a):
    return a.ctypes.data_as(_dp) if a is not None else None


def _f(a):
    return np.asfortranarray(a, dtype=np.float64)


class Gpak:
    """One context = one GPU. Mirrors the slice of GP_utils that sits on the hot path."""

    def __init__(self, device=0, precision=F64, devices=None):
        """devices: a list of HIP ordinals -> gpak_create_multi (one process driving several GPUs; an ordinal may
        repeat on a test box, the ranks then share that GPU)."""
        self._lib = _lib.load()
        h = C.c_void_p()
        if devices is not None:
            arr = (C.c_int * len(devices))(*[int(v) for v in devices])
            rc = self._lib.gpak_create_multi(C.byref(h), len(devices), arr, precision)
        else:
            rc = self._lib.gpak_create(C.byref(h), device, precision)
        if rc != OK:
            raise GpakError(rc, self._lib.gpak_global_error().decode())
        self._h = h
        self.N = 0

    def close(self):
        if getattr(self, "_h", None):
            self._lib.gpak_destroy(self._h)
            self._h = None

    __del__ = close

    def _check(self, rc, allow=()):
        if rc != OK and rc not in allow:
            raise GpakError(rc, self._lib.gpak_last_error(self._h).decode())
        return rc

    # -- state ---------------------------------------------------------------------------
    def set_option(self, opt, value):
        self._check(self._lib.gpak_set_option(self._h, opt, int(value)))

    def set_train(self, X, y):
        X = _f(X)
        y = np.ascontiguousarray(y, dtype=np.float64).ravel()
        self.N, self.d = X.shape
        self._check(self._lib.gpak_set_train(self._h, _p(X), _p(y), self.N, self.d))

    def set_params(self, expans, bias, sn2, dist_mode=DIST_DIRECT):
        e = np.ascontiguousarray(expans, dtype=np.float64)
        assert e.size == 8
        self._check(self._lib.gpak_set_params(self._h, _p(e), float(bias), float(sn2), int(dist_mode)))

    def set_kernel(self, terms, bias, white, sn2, dist_mode=DIST_DIRECT):
        """General HybKerns composition: terms = [(KERN_*, [parameters in the reference's order]), ...]."""
        kinds = (C.c_int * len(terms))(*[int(k) for k, _ in terms])
        pars = np.ascontiguousarray(np.concatenate([np.asarray(p, dtype=np.float64) for _, p in terms]))
        self._check(self._lib.gpak_set_kernel(self._h, len(terms), kinds, _p(pars), float(bias), float(white),
                                              float(sn2), int(dist_mode)))

    # -- hot path ------------------------------------------------------------------------
    def gram(self, want_d2=False):
        K = np.zeros((self.N, self.N), order="F")
        D2 = np.zeros((self.N, self.N), order="F") if want_d2 else None
        self._check(self._lib.gpak_gram(self._h, _p(K), _p(D2)))
        return (K, D2) if want_d2 else K

    def compute_k(self, X1, X2, want_d2=False):
        X1, X2 = _f(X1), _f(X2)
        n, d = X1.shape
        m = X2.shape[0]
        K = np.zeros((n, m), order="F")
        D2 = np.zeros((n, m), order="F") if want_d2 else None
        self._check(self._lib.gpak_compute_k(self._h, _p(X1), n, _p(X2), m, d, _p(K), _p(D2)))
        return (K, D2) if want_d2 else K

    def factor(self):
        """Returns True on success, False on Chol_fail (not positive definite)."""
        return self._check(self._lib.gpak_factor(self._h), allow=(ENOTPD,)) == OK

    def failed_column(self):
        return self._lib.gpak_failed_column(self._h)

    def chol_upper(self):
        R = np.zeros((self.N, self.N), order="F")
        self._check(self._lib.gpak_get_chol_upper(self._h, _p(R)))
        return R

    def solve_alpha(self):
        a = np.zeros(self.N)
        self._check(self._lib.gpak_solve_alpha(self._h, _p(a)))
        return a

    def solve_chol(self, X):
        X = np.array(X, dtype=np.float64, order="F", copy=True)
        k = 1 if X.ndim == 1 else X.shape[1]
        self._check(self._lib.gpak_solve_chol(self._h, _p(X), k))
        return X

    def logLikelihood(self):
        """GP_utils::logLikelihood(): the negative log marginal likelihood, NaN on Chol_fail."""
        v = C.c_double()
        rc = self._check(self._lib.gpak_nlz(self._h, C.byref(v)), allow=(ENOTPD,))
        return v.value if rc == OK else math.nan

    def nlz_terms(self):
        q, s, l = C.c_double(), C.c_double(), C.c_double()
        self._check(self._lib.gpak_nlz_terms(self._h, C.byref(q), C.byref(s), C.byref(l)))
        return q.value, s.value, l.value

    def posteriorMeanVar(self, Xte, want_var=True, compat=0):
        Xte = _f(Xte)
        M, d = Xte.shape
        mean = np.zeros(M)
        var = np.zeros(M) if want_var else None
        self._check(self._lib.gpak_predict(self._h, _p(Xte), M, d, _p(mean), _p(var), compat))
        return mean, var

    def GradLL(self):
        g = np.zeros(10)
        self._check(self._lib.gpak_grad(self._h, _p(g)))
        return g

    def GradLL_hyb(self, ng):
        """Gradient of a general composition: children's blocks in order, bias, sn2."""
        g = np.zeros(int(ng))
        self._check(self._lib.gpak_grad_hyb(self._h, _p(g), int(ng)))
        return g

    # -- measurement ---------------------------------------------------------------------
    def timing(self):
        t = _lib.PhaseTimes()
        self._check(self._lib.gpak_timing(self._h, C.byref(t)))
        out = {name: getattr(t, name) for name, _ in t._fields_}
        out["accumulated_ms"] = list(t.accumulated_ms)
        return out

    def transport(self):
        return self._lib.gpak_transport(self._h).decode()

    def rank_stats(self, rank):
        """gpak_dist_stats of one rank of a multi-GPU context (last step)."""
        from . import dist as _dist
        _dist._load()
        st = _dist.Stats()
        self._check(self._lib.gpak_group_rank_stats(self._h, int(rank), C.byref(st)))
        return {name: getattr(st, name) for name, _ in st._fields_}

    def calibrate(self):
        a, b = C.c_double(), C.c_double()
        self._check(self._lib.gpak_calibrate(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value
