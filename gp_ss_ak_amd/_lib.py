"""Loader of libgpak_hip.so (the C-ABI declared in include/gpak.h).

There is no fallback of any kind: if the shared object is missing, cannot be
loaded, or lacks a declared symbol, importing the binding raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# GPAK_LIB_PATH: tests load a sanitizer build of the same sources (csrc/Makefile targets tsan / asan); never a fallback
LIB_PATH = os.environ.get("GPAK_LIB_PATH") or os.path.join(_HERE, "libgpak_hip.so")

# every symbol include/gpak.h declares (tests check the header and this list agree)
SYMBOLS = [
    "gpak_create", "gpak_create_multi", "gpak_n_gpus", "gpak_transport", "gpak_destroy", "gpak_last_error", "gpak_global_error", "gpak_set_train",
    "gpak_set_params", "gpak_set_kernel", "gpak_set_option", "gpak_gram", "gpak_compute_k", "gpak_factor",
    "gpak_get_chol_upper", "gpak_failed_column", "gpak_solve_alpha", "gpak_solve_chol", "gpak_nlz",
    "gpak_nlz_terms", "gpak_predict", "gpak_grad", "gpak_grad_hyb", "gpak_timing", "gpak_calibrate", "gpak_reload_tuning",
]


class PhaseTimes(C.Structure):
    _fields_ = [("gram_ms", C.c_double), ("factor_ms", C.c_double), ("solve_ms", C.c_double),
                ("nlz_ms", C.c_double), ("predict_ms", C.c_double), ("grad_ms", C.c_double),
                ("trailing_ms", C.c_double), ("trailing_flops", C.c_double),
                ("trailing_launches", C.c_int), ("gram_bytes", C.c_double), ("n", C.c_int),
                ("n_padded", C.c_int), ("trailing_bytes", C.c_double), ("kmatvec_ms", C.c_double),
                ("accumulated_ms", C.c_double * 4), ("evaluations", C.c_int)]


_lib = None


def load():
    """Returns the ctypes handle; raises RuntimeError if the HIP library is unusable."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). gp_ss_ak_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    missing = [s for s in SYMBOLS if not hasattr(lib, s)]
    if missing:
        raise RuntimeError(f"libgpak_hip.so lacks symbols declared in include/gpak.h: {missing}")
    dp = C.POINTER(C.c_double)
    vp = C.c_void_p
    lib.gpak_create.argtypes = [C.POINTER(vp), C.c_int, C.c_int]
    lib.gpak_create_multi.argtypes = [C.POINTER(vp), C.c_int, C.POINTER(C.c_int), C.c_int]
    lib.gpak_n_gpus.argtypes = [vp]
    lib.gpak_transport.argtypes = [vp]
    lib.gpak_transport.restype = C.c_char_p
    lib.gpak_destroy.argtypes = [vp]
    lib.gpak_destroy.restype = None
    lib.gpak_last_error.argtypes = [vp]
    lib.gpak_last_error.restype = C.c_char_p
    lib.gpak_global_error.restype = C.c_char_p
    lib.gpak_set_train.argtypes = [vp, dp, dp, C.c_int, C.c_int]
    lib.gpak_set_params.argtypes = [vp, dp, C.c_double, C.c_double, C.c_int]
    lib.gpak_set_kernel.argtypes = [vp, C.c_int, C.POINTER(C.c_int), dp, C.c_double, C.c_double, C.c_double, C.c_int]
    lib.gpak_set_option.argtypes = [vp, C.c_int, C.c_long]
    lib.gpak_gram.argtypes = [vp, dp, dp]
    lib.gpak_compute_k.argtypes = [vp, dp, C.c_int, dp, C.c_int, C.c_int, dp, dp]
    lib.gpak_factor.argtypes = [vp]
    lib.gpak_get_chol_upper.argtypes = [vp, dp]
    lib.gpak_failed_column.argtypes = [vp]
    lib.gpak_solve_alpha.argtypes = [vp, dp]
    lib.gpak_solve_chol.argtypes = [vp, dp, C.c_int]
    lib.gpak_nlz.argtypes = [vp, dp]
    lib.gpak_nlz_terms.argtypes = [vp, dp, dp, dp]
    lib.gpak_predict.argtypes = [vp, dp, C.c_long, C.c_int, dp, dp, C.c_int]
    lib.gpak_grad.argtypes = [vp, dp]
    lib.gpak_grad_hyb.argtypes = [vp, dp, C.c_int]
    lib.gpak_timing.argtypes = [vp, C.POINTER(PhaseTimes)]
    lib.gpak_calibrate.argtypes = [vp, dp, dp]
    lib.gpak_reload_tuning.restype = None
    _lib = lib
    return lib
