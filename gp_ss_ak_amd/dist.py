"""Python face of the C++ multi-GPU schedule (include/gpak_dist.h, csrc/dist.hip): one rank per process.

Nothing is scheduled or computed here: `DistRank` forwards to gpak_dist_* of libgpak_hip.so, whose built-in HIP
engine and RCCL transport are the product path.  Python only (a) carries the 128-byte RCCL unique id from rank 0 to
the other ranks through torch.distributed's host-side store, and (b) for tests, supplies callback tables: a
host-staged transport over a torch.distributed (gloo) group -- RCCL refuses two ranks on one GPU, which is all a
test box has -- and, in tests/np_dist_engine.py, a NumPy engine for boxes with no GPU at all.

`bench(args)` is what `bench.py --gpus N` runs under torch.distributed.run (bench.py starts that launcher itself as a
child when it was typed without one).
"""
import ctypes as C
import math
import os
import time

import numpy as np

from . import _lib

_dp = C.POINTER(C.c_double)
_vp = C.c_void_p
_F = C.CFUNCTYPE

# field order == struct gpak_dist_engine (include/gpak_dist.h)
ENGINE_FIELDS = [
    ("self", _vp),
    ("alloc", _F(_vp, _vp, C.c_size_t)),
    ("release", _F(None, _vp, _vp)),
    ("upload", _F(C.c_int, _vp, _vp, _vp, _vp, C.c_size_t)),
    ("download", _F(C.c_int, _vp, _vp, _vp, _vp, C.c_size_t)),
    ("zero", _F(C.c_int, _vp, _vp, _vp, C.c_size_t)),
    ("copy", _F(C.c_int, _vp, _vp, _vp, _vp, C.c_size_t)),
    ("stream_create", _F(_vp, _vp, C.c_int)),
    ("stream_destroy", _F(None, _vp, _vp)),
    ("event_create", _F(_vp, _vp, C.c_int)),
    ("event_destroy", _F(None, _vp, _vp)),
    ("event_record", _F(C.c_int, _vp, _vp, _vp)),
    ("stream_wait_event", _F(C.c_int, _vp, _vp, _vp)),
    ("stream_sync", _F(C.c_int, _vp, _vp)),
    ("event_elapsed_ms", _F(C.c_int, _vp, _vp, _vp, _dp)),
    ("transform", _F(C.c_int, _vp, _vp, C.c_int, C.c_int, C.c_int, _dp, _dp, _vp)),
    ("fill_b", _F(C.c_int, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _dp, C.c_double, C.c_double, C.c_int,
                  _vp, C.c_long)),
    ("factor_panel", _F(C.c_int, _vp, _vp, C.c_long, C.c_int, C.c_int, C.c_int, _vp, _vp)),
    ("update_block", _F(C.c_int, _vp, _vp, C.c_long, C.c_int, C.c_int, _vp, C.c_long, C.c_int, C.c_int, C.c_int)),
    ("update_cyclic", _F(C.c_int, _vp, _vp, C.c_long, C.c_int, C.c_int, _vp, C.c_long, C.c_int, C.c_int, C.c_int,
                         C.c_int, C.c_int, C.c_int, C.c_int)),
    ("trsv_fwd_block", _F(C.c_int, _vp, _vp, C.c_long, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp)),
    ("trsv_bwd_packed", _F(C.c_int, _vp, _vp, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp)),
    ("diag_inverse", _F(C.c_int, _vp, _vp, C.c_long, C.c_int, C.c_int, C.c_int, _vp, _vp)),
    ("logdiag_block", _F(C.c_int, _vp, _vp, C.c_long, C.c_int, C.c_int, C.c_int, _vp)),
    ("kmatvec", _F(C.c_int, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _dp, C.c_double, C.c_int, _vp, _vp)),
    ("nlz_terms", _F(C.c_int, _vp, C.c_int, _vp, _vp, _vp, C.c_double, _vp)),
    ("pack", _F(C.c_int, _vp, _vp, C.c_long, C.c_int, C.c_int, C.c_int, _vp)),
    ("vec_scale", _F(C.c_int, _vp, C.c_int, _vp, C.c_double, _vp)),
    ("vec_sum", _F(C.c_int, _vp, C.c_int, _vp, _vp)),
    ("grad_g_rows", _F(C.c_int, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_vp), C.POINTER(_vp), _vp)),
    ("grad_binv_rows", _F(C.c_int, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp)),
    ("grad_pairs_rows", _F(C.c_int, _vp, _vp, C.c_int, _vp, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, C.c_int, C.c_int,
                           _dp, C.c_double, C.c_double, C.c_int, _vp, _vp)),
    # the row-block x column-block layout (gpak_grid_*)
    ("fill_rect", _F(C.c_int, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _dp, C.c_double, C.c_double,
                     C.c_int, _vp, C.c_long)),
    ("solve_rows", _F(C.c_int, _vp, _vp, C.c_long, C.c_int, C.c_int, _vp, C.c_long, _vp)),
    ("update_rect", _F(C.c_int, _vp, _vp, C.c_long, _vp, C.c_long, C.c_int, _vp, C.c_long, C.c_int, C.c_int, C.c_int)),
    ("gemv_n_add", _F(C.c_int, _vp, _vp, C.c_long, C.c_int, C.c_int, _vp, _vp)),
    ("gemv_t", _F(C.c_int, _vp, _vp, C.c_long, C.c_int, C.c_int, _vp, _vp)),
    ("vec_axpy", _F(C.c_int, _vp, C.c_int, C.c_double, _vp, _vp)),
    # a serialized HybKerns composition (gpak_dev.h GPAK_DIST_HYB)
    ("transform_k", _F(C.c_int, _vp, _vp, C.c_int, C.c_int, C.c_int, _dp, C.c_int, _dp, _vp)),
]


class Engine(C.Structure):
    _fields_ = ENGINE_FIELDS


class Transport(C.Structure):
    _fields_ = [
        ("self", _vp),
        ("bcast", _F(C.c_int, _vp, _vp, _vp, C.c_size_t, C.c_int)),
        ("allreduce_sum", _F(C.c_int, _vp, _vp, _vp, C.c_size_t)),
        ("allreduce_min_int", _F(C.c_int, _vp, _vp, _vp, C.c_size_t)),
        # sub-groups of a Pr x Pc grid (NULL: 1-D layouts only)
        ("grid_setup", _F(C.c_int, _vp, C.c_int, C.c_int)),
        ("bcast_group", _F(C.c_int, _vp, _vp, _vp, C.c_size_t, C.c_int, C.c_int)),
        ("allreduce_sum_group", _F(C.c_int, _vp, _vp, _vp, C.c_size_t, C.c_int)),
    ]


class Stats(C.Structure):
    _fields_ = [("rank", C.c_int), ("world", C.c_int), ("n", C.c_int), ("n_padded", C.c_int), ("nb", C.c_int),
                ("n_panels", C.c_int), ("flags", C.c_int), ("bytes_broadcast", C.c_double), ("step_ms", C.c_double),
                ("fill_ms", C.c_double), ("factor_ms", C.c_double), ("solve_ms", C.c_double), ("nlz_ms", C.c_double),
                ("bulk_ms", C.c_double), ("bulk_flops", C.c_double), ("chain_ms", C.c_double), ("comm_ms", C.c_double),
                ("wait_ms", C.c_double), ("kmatvec_ms", C.c_double), ("bulk_bytes", C.c_double),
                ("bulk_launches", C.c_double)]


DIST_SYMBOLS = ["gpak_dist_create", "gpak_dist_destroy", "gpak_dist_last_error", "gpak_dist_rccl_unique_id",
                "gpak_dist_init_rccl", "gpak_dist_selfcheck", "gpak_dist_set_train", "gpak_dist_set_params", "gpak_dist_set_kernel",
                "gpak_dist_nlz", "gpak_dist_nlz_terms", "gpak_dist_grad", "gpak_dist_get_alpha", "gpak_dist_get_stats",
                "gpak_dist_failed_column", "gpak_group_rank_stats", "gpak_create_multi_with_engines",
                "gpak_grid_create", "gpak_grid_destroy", "gpak_grid_last_error", "gpak_grid_init_rccl", "gpak_grid_set_train",
                "gpak_grid_set_params", "gpak_grid_nlz", "gpak_grid_nlz_terms", "gpak_grid_get_alpha", "gpak_grid_get_stats",
                "gpak_dev_vec_scale", "gpak_dev_vec_sum"]


def _load():
    lib = _lib.load()
    missing = [s for s in DIST_SYMBOLS if not hasattr(lib, s)]
    if missing:
        raise RuntimeError(f"libgpak_hip.so lacks symbols declared in include/gpak_dist.h: {missing}")
    lib.gpak_dist_create.argtypes = [C.POINTER(_vp), C.c_int, C.c_int, C.c_int, C.POINTER(Engine), C.POINTER(Transport)]
    lib.gpak_dist_destroy.argtypes = [_vp]
    lib.gpak_dist_destroy.restype = None
    lib.gpak_dist_last_error.argtypes = [_vp]
    lib.gpak_dist_last_error.restype = C.c_char_p
    lib.gpak_dist_rccl_unique_id.argtypes = [C.c_char_p]
    lib.gpak_dist_init_rccl.argtypes = [_vp, C.c_char_p]
    lib.gpak_dist_selfcheck.argtypes = [_vp, C.POINTER(C.c_int)]
    lib.gpak_dist_set_train.argtypes = [_vp, _dp, _dp, C.c_int, C.c_int, C.c_int]
    lib.gpak_dist_set_params.argtypes = [_vp, _dp, C.c_double, C.c_double, C.c_int]
    lib.gpak_dist_set_kernel.argtypes = [_vp, C.c_int, C.POINTER(C.c_int), _dp, C.c_double, C.c_double, C.c_double, C.c_int]
    lib.gpak_dist_nlz.argtypes = [_vp, _dp]
    lib.gpak_dist_nlz_terms.argtypes = [_vp, _dp, _dp, _dp]
    lib.gpak_dist_get_alpha.argtypes = [_vp, _dp]
    lib.gpak_dist_grad.argtypes = [_vp, _dp]
    lib.gpak_dist_get_stats.argtypes = [_vp, C.POINTER(Stats)]
    lib.gpak_dist_failed_column.argtypes = [_vp]
    lib.gpak_group_rank_stats.argtypes = [_vp, C.c_int, C.POINTER(Stats)]
    lib.gpak_grid_create.argtypes = [C.POINTER(_vp), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(Engine),
                                     C.POINTER(Transport)]
    lib.gpak_grid_destroy.argtypes = [_vp]
    lib.gpak_grid_destroy.restype = None
    lib.gpak_grid_last_error.argtypes = [_vp]
    lib.gpak_grid_last_error.restype = C.c_char_p
    lib.gpak_grid_init_rccl.argtypes = [_vp, C.c_char_p]
    lib.gpak_grid_set_train.argtypes = [_vp, _dp, _dp, C.c_int, C.c_int, C.c_int]
    lib.gpak_grid_set_params.argtypes = [_vp, _dp, C.c_double, C.c_double, C.c_int]
    lib.gpak_grid_nlz.argtypes = [_vp, _dp]
    lib.gpak_grid_nlz_terms.argtypes = [_vp, _dp, _dp, _dp]
    lib.gpak_grid_get_alpha.argtypes = [_vp, _dp]
    lib.gpak_grid_get_stats.argtypes = [_vp, C.POINTER(Stats)]
    return lib


class DistError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__(f"gpak_dist status {status}: {msg}")
        self.status = status


class StagedTransport:
    """TEST transport: collectives of a torch.distributed group (gloo) staged through host memory, for the built-in
    HIP engine (several ranks rehearsed on ONE GPU).  The product transport is RCCL inside the library."""

    def __init__(self, group=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        hip = C.CDLL("libamdhip64.so")
        self.hip = hip
        hip.hipMemcpy.argtypes = [_vp, _vp, C.c_size_t, C.c_int]
        hip.hipStreamSynchronize.argtypes = [_vp]
        self.rank = dist.get_rank(group)

        def stage(st, buf, count, dtype, op):
            t = torch.empty(int(count), dtype=dtype)
            nbytes = t.numel() * t.element_size()
            hip.hipStreamSynchronize(st)
            if hip.hipMemcpy(t.data_ptr(), buf, nbytes, 2) != 0:     # device -> host
                return -1
            op(t)
            return 0 if hip.hipMemcpy(buf, t.data_ptr(), nbytes, 1) == 0 else -1   # host -> device

        def bcast(_self, st, buf, count, root):
            return stage(st, buf, count, torch.float64, lambda t: dist.broadcast(t, src=root, group=group))

        def ar_sum(_self, st, buf, count):
            return stage(st, buf, count, torch.float64, lambda t: dist.all_reduce(t, group=group))

        def ar_min(_self, st, buf, count):
            return stage(st, buf, count, torch.int32, lambda t: dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group))

        # sub-groups of a Pr x Pc grid (rank = pr + Pr * pc): torch.distributed groups made on demand, the same way on
        # every rank (new_group is collective over the world)
        self.groups = {}

        def grid_setup(_self, Pr, Pc):
            world = dist.get_world_size(group)
            rank = dist.get_rank(group)
            if Pr * Pc != world:
                return 2
            pr, pc = rank % Pr, rank // Pr
            for r in range(Pr):
                members = [r + Pr * c for c in range(Pc)]
                g = dist.new_group(members)
                if r == pr:
                    self.groups[1] = (g, members)
            for c in range(Pc):
                members = [r + Pr * c for r in range(Pr)]
                g = dist.new_group(members)
                if c == pc:
                    self.groups[2] = (g, members)
            return 0

        def bcast_g(_self, st, buf, count, root, grp):
            g, members = self.groups[grp]
            return stage(st, buf, count, torch.float64, lambda t: dist.broadcast(t, src=members[root], group=g))

        def ar_sum_g(_self, st, buf, count, grp):
            g, _ = self.groups[grp]
            return stage(st, buf, count, torch.float64, lambda t: dist.all_reduce(t, group=g))

        F = Transport._fields_
        self._keep = (F[1][1](bcast), F[2][1](ar_sum), F[3][1](ar_min), F[4][1](grid_setup), F[5][1](bcast_g), F[6][1](ar_sum_g))
        self.table = Transport(None, *self._keep)


class DistRank:
    """One rank of the C++ schedule.  engine / transport: None = built-in HIP engine / built-in RCCL transport."""

    def __init__(self, rank, world, device=0, engine=None, transport=None, rccl_id=None):
        self._lib = _load()
        self._engine, self._transport = engine, transport       # keep the callback objects alive
        h = _vp()
        rc = self._lib.gpak_dist_create(C.byref(h), rank, world, device,
                                        C.byref(engine.table) if engine is not None else None,
                                        C.byref(transport.table) if transport is not None else None)
        if rc != 0:
            raise DistError(rc, "gpak_dist_create failed (no gfx950 device? there is no CPU fallback)")
        self._h = h
        self.rank, self.world = rank, world
        # GPAK_DIST_RCCL_WORLD1=1 (tests): a ONE-rank RCCL communicator, so that a one-GPU box exercises the library's
        # RCCL binding (dlopen, ncclCommInitRank, ncclBroadcast / ncclAllReduce on the communication stream) at all
        if transport is None and world == 1 and os.environ.get("GPAK_DIST_RCCL_WORLD1") and rccl_id is None:
            rccl_id = self.rccl_unique_id()
        if transport is None and (world > 1 or rccl_id is not None):
            if rccl_id is None:
                raise ValueError("the RCCL transport needs the unique id made on rank 0 (rccl_unique_id())")
            self._check(self._lib.gpak_dist_init_rccl(self._h, rccl_id))

    @staticmethod
    def rccl_unique_id():
        buf = C.create_string_buffer(128)
        rc = _load().gpak_dist_rccl_unique_id(buf)
        if rc != 0:
            raise DistError(rc, "ncclGetUniqueId failed")
        return buf.raw

    def _check(self, rc, allow=()):
        if rc != 0 and rc not in allow:
            raise DistError(rc, self._lib.gpak_dist_last_error(self._h).decode())
        return rc

    def close(self):
        if getattr(self, "_h", None):
            self._lib.gpak_dist_destroy(self._h)
            self._h = None

    __del__ = close

    def selfcheck(self):
        flags = C.c_int()
        self._check(self._lib.gpak_dist_selfcheck(self._h, C.byref(flags)))
        return flags.value

    def set_train(self, X, y, nb=512):
        X = np.asfortranarray(X, dtype=np.float64)
        y = np.ascontiguousarray(y, dtype=np.float64).ravel()
        self.N = X.shape[0]
        self._check(self._lib.gpak_dist_set_train(self._h, X.ctypes.data_as(_dp), y.ctypes.data_as(_dp), X.shape[0],
                                                  X.shape[1], int(nb)))

    def set_params(self, expans, bias, sn2, dist_mode=1):
        e = np.ascontiguousarray(expans, dtype=np.float64)
        self._check(self._lib.gpak_dist_set_params(self._h, e.ctypes.data_as(_dp), float(bias), float(sn2), int(dist_mode)))

    def set_kernel(self, terms, bias, white, sn2, dist_mode=1):
        """General HybKerns composition: terms = [(kind, [parameters in the reference's order]), ...] (gpak.KERN_*)."""
        kinds = (C.c_int * len(terms))(*[int(k) for k, _ in terms])
        pars = np.ascontiguousarray(np.concatenate([np.asarray(p, dtype=np.float64) for _, p in terms]))
        self._check(self._lib.gpak_dist_set_kernel(self._h, len(terms), kinds, pars.ctypes.data_as(_dp), float(bias),
                                                   float(white), float(sn2), int(dist_mode)))

    def nlz(self):
        v = C.c_double()
        rc = self._check(self._lib.gpak_dist_nlz(self._h, C.byref(v)), allow=(1,))
        return v.value if rc == 0 else math.nan

    def nlz_terms(self):
        q, s, l = C.c_double(), C.c_double(), C.c_double()
        self._check(self._lib.gpak_dist_nlz_terms(self._h, C.byref(q), C.byref(s), C.byref(l)))
        return q.value, s.value, l.value

    def grad(self):
        """GradLL as written, g[10] = {8 ExpAns, bias, sn2}; distributed by row blocks of B^-1."""
        g = np.zeros(10)
        self._check(self._lib.gpak_dist_grad(self._h, g.ctypes.data_as(_dp)))
        return g

    def get_alpha(self):
        a = np.zeros(self.N)
        self._check(self._lib.gpak_dist_get_alpha(self._h, a.ctypes.data_as(_dp)))
        return a

    def stats(self):
        s = Stats()
        self._check(self._lib.gpak_dist_get_stats(self._h, C.byref(s)))
        return {name: getattr(s, name) for name, _ in s._fields_}


class GridRank:
    """One rank of the row-block x column-block layout (gpak_grid_* of include/gpak_dist.h, csrc/grid.inc) on a Pr x Pc
    process grid, rank = pr + Pr * pc.  engine / transport: None = built-in HIP engine / built-in RCCL transport (world
    communicator + ncclCommSplit row and column communicators)."""

    def __init__(self, rank, world, Pr, Pc, device=0, engine=None, transport=None, rccl_id=None):
        self._lib = _load()
        self._engine, self._transport = engine, transport
        h = _vp()
        rc = self._lib.gpak_grid_create(C.byref(h), rank, world, Pr, Pc, device,
                                        C.byref(engine.table) if engine is not None else None,
                                        C.byref(transport.table) if transport is not None else None)
        if rc != 0:
            raise DistError(rc, "gpak_grid_create failed (Pr * Pc != world, Pr == 1, an engine / transport without the "
                                "2-D entries, or no gfx950 device: there is no CPU fallback)")
        self._h = h
        self.rank, self.world, self.Pr, self.Pc = rank, world, Pr, Pc
        if transport is None:
            if rccl_id is None:
                raise ValueError("the RCCL transport needs the unique id made on rank 0 (DistRank.rccl_unique_id())")
            self._check(self._lib.gpak_grid_init_rccl(self._h, rccl_id))

    def _check(self, rc, allow=()):
        if rc != 0 and rc not in allow:
            raise DistError(rc, self._lib.gpak_grid_last_error(self._h).decode())
        return rc

    def close(self):
        if getattr(self, "_h", None):
            self._lib.gpak_grid_destroy(self._h)
            self._h = None

    __del__ = close

    def set_train(self, X, y, nb=512):
        X = np.asfortranarray(X, dtype=np.float64)
        y = np.ascontiguousarray(y, dtype=np.float64).ravel()
        self.N = X.shape[0]
        self._check(self._lib.gpak_grid_set_train(self._h, X.ctypes.data_as(_dp), y.ctypes.data_as(_dp), X.shape[0],
                                                  X.shape[1], int(nb)))

    def set_params(self, expans, bias, sn2, dist_mode=1):
        e = np.ascontiguousarray(expans, dtype=np.float64)
        self._check(self._lib.gpak_grid_set_params(self._h, e.ctypes.data_as(_dp), float(bias), float(sn2), int(dist_mode)))

    def nlz(self):
        v = C.c_double()
        rc = self._check(self._lib.gpak_grid_nlz(self._h, C.byref(v)), allow=(1,))
        return v.value if rc == 0 else math.nan

    def nlz_terms(self):
        q, s, l = C.c_double(), C.c_double(), C.c_double()
        self._check(self._lib.gpak_grid_nlz_terms(self._h, C.byref(q), C.byref(s), C.byref(l)))
        return q.value, s.value, l.value

    def get_alpha(self):
        a = np.zeros(self.N)
        self._check(self._lib.gpak_grid_get_alpha(self._h, a.ctypes.data_as(_dp)))
        return a

    def stats(self):
        s = Stats()
        self._check(self._lib.gpak_grid_get_stats(self._h, C.byref(s)))
        return {name: getattr(s, name) for name, _ in s._fields_}


# ------------------------------------------------------------------------------------------------
# bench.py --gpus N entry point (launched by torch.distributed.run, one rank per GPU)
# ------------------------------------------------------------------------------------------------
def _one_size(args, N, dist, rank, world, local, make_rank, steps, warmup, bench_mod):
    from . import synth
    X, y = synth.drillholes(N)
    gp = make_rank()
    gp.set_train(X, y, nb=args.nb_outer or 512)
    mode = 1 if args.dist == "direct" else 0

    def step(i):
        e, bias, sn2 = bench_mod.params_for_step(i)
        gp.set_params(e, bias, sn2, mode)
        return gp.nlz()

    for i in range(warmup):
        step(i)
    dist.barrier()
    t0 = time.perf_counter()
    nlz = None
    acc = {}
    for i in range(steps):
        nlz = step(warmup + i)          # synchronous: returns after this rank's device finished
        for k, v in gp.stats().items():
            if k.endswith("_ms") or k in ("bulk_flops", "bytes_broadcast"):
                acc[k] = acc.get(k, 0.0) + v
    dist.barrier()
    wall = time.perf_counter() - t0
    import torch
    w = torch.tensor([wall], dtype=torch.float64)
    dist.all_reduce(w, op=dist.ReduceOp.MAX)
    wall = float(w.item())
    st = gp.stats()
    mine = {k: v / steps for k, v in acc.items()}
    mine["rank"] = rank
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    gp.close()
    return {"N": N, "wall": wall, "nlz": nlz, "stats": st, "per_rank": gathered}


def bench(args):
    import datetime

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs WORLD_SIZE={args.gpus} (launch with torch.distributed.run); "
                         f"got WORLD_SIZE={world}")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    # one node: the control plane talks over loopback (a container's hostname need not resolve)
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    # host-side control plane only (unique id, barriers, max over ranks): gloo.  The data path -- every panel
    # broadcast and all-reduce -- is RCCL inside libgpak_hip.so.
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=300))
    # GPAK_DIST_TRANSPORT=staged / GPAK_DIST_DEVICE exist for the tests: they rehearse this exact entry point with
    # several ranks on the ONE GPU of a test box (RCCL refuses two ranks per GPU; gloo stages the collectives)
    staged = os.environ.get("GPAK_DIST_TRANSPORT", "rccl") == "staged"
    if "GPAK_DIST_DEVICE" in os.environ:
        local = int(os.environ["GPAK_DIST_DEVICE"])
    import bench as bench_mod

    def make_rank():
        if staged:
            return DistRank(rank, world, device=local, transport=StagedTransport())
        # whether RCCL is usable at all is settled BEFORE any rank enters the ncclCommInitRank rendezvous: the id is
        # made on rank 0 (dlopen + ncclGetUniqueId) and a failure there travels to every rank instead of leaving them
        # in a broadcast; a rank whose device is missing says so through the same channel
        ids = [None]
        if rank == 0 and world > 1:
            try:
                ids = [DistRank.rccl_unique_id()]
            except Exception as e:   # noqa: BLE001
                ids = [("error", f"{type(e).__name__}: {e}")]
        dist.broadcast_object_list(ids, src=0)
        if isinstance(ids[0], tuple):
            raise RuntimeError(f"rank 0 could not make the RCCL unique id: {ids[0][1]}")
        have = torch.tensor([1 if local < torch.cuda.device_count() else 0], dtype=torch.int32)
        dist.all_reduce(have, op=dist.ReduceOp.MIN)
        if int(have.item()) == 0:
            raise RuntimeError(f"a rank has no device of its own (rank {rank}: ordinal {local}, "
                               f"{torch.cuda.device_count()} visible)")
        return DistRank(rank, world, device=local, rccl_id=ids[0])

    # start-up probe: create a rank (dlopen librccl, ncclCommInitRank) and run the library's self-check once.  If that
    # fails -- on ANY rank: the verdict is min-reduced over the control plane -- the round-1 Python schedule over
    # torch's own NCCL process group takes over, and the line says so; a failure later than this is a failure.
    ok, why = 1, ""
    try:
        probe = make_rank()
        probe.selfcheck()
        probe.close()
    except Exception as e:   # noqa: BLE001 -- anything at start-up selects the fallback
        ok, why = 0, f"{type(e).__name__}: {e}"
    flag = torch.tensor([ok], dtype=torch.int32)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag.item()) == 0 and not staged:
        # The per-process RCCL start-up failed somewhere (the verdict is the same on every rank).  The run is handed to
        # the OTHER host of the same C++ schedule: rank 0 drives all GPUs from one process (gpak_create_multi: RCCL
        # communicators from one ncclCommInitAll, or the in-process peer-copy transport); the other ranks wait.  The
        # line says so in `metric`, `config.workload` and `fallback` -- it is a different host, not a silent swap.
        import sys
        print(f"[gpak rank {rank}] per-process RCCL start-up failed ({why or 'on another rank'}); rank 0 runs the same "
              f"C++ schedule from one process (gpak_create_multi)", file=sys.stderr, flush=True)
        whys = [None] * world
        dist.all_gather_object(whys, why)
        out = None
        if rank == 0:
            out = bench_mod.run_inproc(args)
            reason = next((w for w in whys if w), "unknown")
            out["fallback"] = ("one process per GPU could not start RCCL (" + reason + "); measured with ONE process "
                               "driving all GPUs instead (gpak_create_multi)")
            out["metric"] += " [fallback host: one process, thread per GPU]"
            out["config"]["workload"] += " -- FALLBACK from the process-per-GPU host: " + reason
        dist.barrier()
        dist.destroy_process_group()
        return out
    if int(flag.item()) == 0:
        raise RuntimeError(f"distributed start-up failed: {why}")
    res = _one_size(args, args.n, dist, rank, world, local, make_rank, args.steps, args.warmup, bench_mod)
    def build(res, extra, grids):
        def line(r, steps):
            Np = r["stats"]["n_padded"]
            flops = Np ** 3 / 3.0
            per_step = r["wall"] / steps
            bulk_ms = max(p["bulk_ms"] for p in r["per_rank"])
            bulk_tf = [p["bulk_flops"] / (p["bulk_ms"] * 1e-3) / 1e12 if p["bulk_ms"] > 0 else None for p in r["per_rank"]]
            return {"N": r["N"], "steps_per_s": steps / r["wall"], "ms_per_step": per_step * 1e3, "nlz": r["nlz"],
                    "whole_step_tflops_per_gpu": flops / per_step / 1e12 / world,
                    "whole_step_frac_of_mfma_peak": flops / per_step / 1e12 / world / bench_mod.PEAK_F64_MFMA_TFLOPS,
                    "bulk_update_tflops_per_rank": bulk_tf, "bytes_broadcast_per_step": r["per_rank"][0]["bytes_broadcast"],
                    "phases_ms_per_rank": [{k: round(v, 3) for k, v in p.items()} for p in r["per_rank"]],
                    "max_bulk_ms": bulk_ms, "stream_flags": r["stats"]["flags"]}
        main = line(res, args.steps)
        out = {
            "metric": f"GP train step/sec (Gram+Cholesky+logML) at N={args.n} fp64",
            "value": main["steps_per_s"], "unit": "steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": main["ms_per_step"], "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"N={args.n} fp64 ExpAns+Bias Gram + block-column-cyclic Cholesky + solves + logML, "
                                   f"C++ schedule (csrc/dist.hip), sub-panel broadcast over RCCL",
                       "N": args.n, "dist_mode": args.dist, "nb_outer": res["stats"]["nb"],
                       "parallelism": f"block-column-cyclic x{world}",
                       "transport": "host-staged gloo (test rehearsal)" if staged else "RCCL"},
            "nlz": main["nlz"],
            "roofline": {"kernel": "gpak_gemm_nt_f64_rs<4,2,true> (bulk trailing update of the owned block columns)",
                         "bound": "mfma",
                         "achieved": min(v for v in main["bulk_update_tflops_per_rank"] if v) if any(main["bulk_update_tflops_per_rank"]) else None,
                         "peak": bench_mod.PEAK_F64_MFMA_TFLOPS, "unit": "TFLOP/s", "traffic": None,
                         "note": "slowest rank's bulk launches: algorithmic flops / summed launch durations (hipEvents)"},
            "bytes_broadcast_per_step": main["bytes_broadcast_per_step"],
            "phases_ms_per_rank": main["phases_ms_per_rank"],
            "whole_step_frac_of_mfma_peak_per_gpu": main["whole_step_frac_of_mfma_peak"],
            "stream_flags": main["stream_flags"],
        }
        a = out["roofline"]["achieved"]
        out["roofline"]["frac"] = a / bench_mod.PEAK_F64_MFMA_TFLOPS if a else None
        if extra is not None:
            out["n65536"] = line(extra, max(1, min(args.steps, 3)))
        if grids:
            out["grid_layouts"] = []
            for (gr, gc), g_res, why in grids:
                if g_res is None:
                    out["grid_layouts"].append({"grid": f"{gr}x{gc}", "error": why or "failed on another rank"})
                    continue
                gl = line(g_res, max(1, min(args.steps, 3)))
                out["grid_layouts"].append({
                    "grid": f"{gr}x{gc}", "layout": "row-block x column-block, block (i, j) on rank (i % Pr) + Pr * (j % Pc)",
                    "steps_per_s": gl["steps_per_s"], "ms_per_step": gl["ms_per_step"], "nlz": gl["nlz"],
                    "bytes_received_per_rank_per_step": [p["bytes_broadcast"] for p in gl["phases_ms_per_rank"]],
                    "model_bytes_per_rank": 8.0 * g_res["stats"]["n_padded"] ** 2 / 2 * (1.0 / gr + 1.0 / gc),
                    "phases_ms_per_rank": gl["phases_ms_per_rank"]})
            out["grid_layouts_note"] = ("1-D block-column-cyclic (the headline value above) receives (P-1)/P * N^2/2 * 8 B per "
                                        "rank and step: " + str(8.0 * res["stats"]["n_padded"] ** 2 / 2 * (world - 1) / world))
        return out

    # The headline measurement is complete here.  What follows is optional (the N=65536 point, the grid layouts -- code
    # paths no multi-GPU node has run yet): a watchdog guarantees that a sub-run that hangs cannot take the headline
    # line with it -- at the deadline rank 0 prints what it has and every rank leaves.
    import json
    import sys
    import threading
    state = {"extra": None, "grids": [], "phase": "start"}
    deadline = float(os.environ.get("GPAK_BENCH_EXTRAS_TIMEOUT_S", "300"))

    def on_timeout():
        if rank == 0:
            o = build(res, state["extra"], state["grids"])
            o["extras_note"] = (f"optional sub-runs stopped by the {deadline:.0f} s watchdog (in: {state['phase']}); the "
                                f"headline numbers are complete")
            print(json.dumps(o), flush=True)
        sys.stdout.flush()
        os._exit(0)
    timer = threading.Timer(deadline, on_timeout)
    timer.daemon = True
    timer.start()
    extra = None
    if args.n != 65536 and not args.no_n65536:
        # north_star's scaling curve is quoted at N=65536 (BASELINE.json configs[3]): a short run of that size rides
        # along as a sub-object; it needs 32 GiB / P + 17 GB per GPU
        state["phase"] = "N=65536"
        extra = _one_size(args, 65536, dist, rank, world, local, make_rank, max(1, min(args.steps, 3)), 1, bench_mod)
    # the row-block x column-block layouts of the same world size (north_star's 2-D sharding, csrc/grid.inc) ride along:
    # a short run per grid, so that one line holds both layouts' step time and bytes received per rank
    state["extra"] = extra
    grids = state["grids"]
    want = getattr(args, "grid", "auto")
    if want != "none":
        shapes = ([tuple(int(v) for v in want.split("x"))] if want not in ("auto", "") else
                  [(pr, world // pr) for pr in (2, 4, 8) if world % pr == 0 and pr <= world])
        for (gr, gc) in shapes:
            if gr * gc != world or gr < 2:
                continue

            def make_grid(gr=gr, gc=gc):
                if staged:
                    return GridRank(rank, world, gr, gc, device=local, transport=StagedTransport())
                ids = [DistRank.rccl_unique_id() if rank == 0 else None]
                dist.broadcast_object_list(ids, src=0)
                return GridRank(rank, world, gr, gc, device=local, rccl_id=ids[0])
            ok, why = 1, ""
            state["phase"] = f"grid {gr}x{gc}"
            try:
                g_res = _one_size(args, args.n, dist, rank, world, local, make_grid, max(1, min(args.steps, 3)), 1, bench_mod)
            except Exception as e:   # noqa: BLE001 -- a layout that cannot start is reported, not fatal
                ok, why, g_res = 0, f"{type(e).__name__}: {e}", None
            flag = torch.tensor([ok], dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            grids.append(((gr, gc), g_res if int(flag.item()) else None, why))
    timer.cancel()
    out = build(res, extra, grids) if rank == 0 else None
    dist.destroy_process_group()
    return out
