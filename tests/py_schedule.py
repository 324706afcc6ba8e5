"""TEST INFRASTRUCTURE (moved out of the product package in round 3): the round-1 PYTHON schedule of the multi-GPU hot
path -- block-column-cyclic Cholesky / solves / nlZ, one process per GPU -- kept as a second, independent implementation
that tests/test_multigpu.py holds against the oracle, and `HipEngine`, a thin ctypes face of the gpak_dev_* tile
operations that tests and tools/ use to time single kernels.  The PRODUCT schedule is C++ (csrc/dist.hip).

The reference is a single process; this is the build's own distribution of
GP_utils::ldB2_exact / solve_chol / logLikelihood (GP_Utils.cpp:841-845, 872-915, 1138-1162):

  * outer block column b (width nb) of B = I + K/sn2 lives on rank b % P, nowhere else
    (N=65536: 32 GiB / P per GPU);
  * fill: every rank fills its own block columns from the replicated coordinates -- no
    communication;
  * factor: the owner factors block column b and BROADCASTS the panel below its diagonal block
    (the only bulk collective: RCCL broadcast over xGMI); every rank then updates the block
    columns it owns.  One panel of look-ahead: the owner of b+1 updates and factors its column
    first and its broadcast is in flight while all ranks run the bulk update with panel b;
  * solves: forward substitution needs one nb-entry all-reduce per block column, backward one
    nb-entry broadcast; f = K*alpha is an N-entry all-reduce of per-rank partial sums; log-det
    a scalar all-reduce.

torch is plumbing here: device memory (tensors), streams and torch.distributed ("nccl" = RCCL on
ROCm; "gloo" on CPU for the tests).  All arithmetic goes through the engine: `HipEngine` calls
the gpak_dev_* C-ABI of libgpak_hip.so (include/gpak_dev.h).  The schedule below is engine- and
backend-agnostic so that the world_size-2 gloo tests exercise exactly this code.
"""
import ctypes as C
import math
import os
import time

import numpy as np

TILE = 128
INT_MAX = 0x7FFFFFFF


def _torch():
    import torch
    return torch


class HipEngine:
    """gpak_dev_* over torch CUDA(=HIP) tensors.  Fails loudly if the library is missing."""

    def __init__(self, device_index):
        torch = _torch()
        from gp_ss_ak_amd import _lib
        if not torch.cuda.is_available():
            raise RuntimeError("HipEngine needs a GPU (there is no CPU fallback)")
        torch.cuda.set_device(device_index)
        self.device = torch.device("cuda", device_index)
        self.lib = _lib.load()
        for name in ("gpak_dev_transform", "gpak_dev_fill_b", "gpak_dev_factor_panel", "gpak_dev_update_block",
                     "gpak_dev_update_cyclic", "gpak_dev_trsv_fwd_block", "gpak_dev_coldot", "gpak_dev_trsv_bwd_block",
                     "gpak_dev_logdiag_block", "gpak_dev_kmatvec", "gpak_dev_nlz_terms", "gpak_dev_stream_create",
                     "gpak_dev_stream_destroy", "gpak_dev_pack", "gpak_dev_trsv_bwd_packed", "gpak_dev_diag_inverse"):
            if not hasattr(self.lib, name):
                raise RuntimeError(f"libgpak_hip.so lacks {name}")
            getattr(self.lib, name).restype = C.c_int
        self._host_e = None
        self._cur = None   # cuda_stream handle of the stream the engine launches on (None: ask torch)

    # -- memory ------------------------------------------------------------------------------
    def empty(self, n, dtype=None):
        torch = _torch()
        return torch.empty(int(n), dtype=dtype or torch.float64, device=self.device)

    def zeros(self, n, dtype=None):
        torch = _torch()
        return torch.zeros(int(n), dtype=dtype or torch.float64, device=self.device)

    def from_numpy(self, a):
        torch = _torch()
        return torch.from_numpy(np.ascontiguousarray(a)).to(self.device)

    def sync(self):
        _torch().cuda.synchronize()

    has_streams = True

    def panel_stream(self):
        """High-priority side stream for the look-ahead panel work (created once)."""
        torch = _torch()
        if not hasattr(self, "_ps"):
            self._ps = torch.cuda.Stream(device=self.device, priority=-1)
        return self._ps

    def bulk_stream(self, skip_cus=8):
        """A stream that leaves `skip_cus` compute units free (gpak_dev_stream_create): a rank's whole step runs
        on it except the panel chain, which then always finds idle CUs (created once)."""
        torch = _torch()
        if not hasattr(self, "_bulk"):
            h = C.c_void_p()
            rc = self.lib.gpak_dev_stream_create(int(skip_cus), C.byref(h))
            if rc == 0 and h.value:
                self._bulk_handle = h
                self._bulk = torch.cuda.ExternalStream(h.value, device=self.device)
            else:   # CU masking unavailable (e.g. a partitioned device): an ordinary stream, only slower
                self._bulk = torch.cuda.Stream(device=self.device)
        return self._bulk

    def _st(self):
        # torch.cuda.current_stream() costs ~7 us of host time per call; the schedule makes ~800 engine calls per
        # step, so the handle of the stream last entered through `on()` is cached
        if self._cur is None:
            self._cur = _torch().cuda.current_stream().cuda_stream
        return C.c_void_p(self._cur)

    def on(self, stream):
        """Context manager: make `stream` torch's current stream AND the stream of the engine's launches."""
        eng, torch = self, _torch()

        class _ctx:
            def __enter__(c):
                c.prev = eng._cur
                c.t = torch.cuda.stream(stream)
                c.t.__enter__()
                eng._cur = stream.cuda_stream

            def __exit__(c, *a):
                c.t.__exit__(*a)
                eng._cur = c.prev
        return _ctx()

    def pack(self, src, ld, row0, nrows, ncols, dst):
        """dst (nrows x ncols, packed) <- rows [row0, row0+nrows) of the ncols columns of src (leading dimension ld)."""
        self._chk(self.lib.gpak_dev_pack(self._st(), self._p(src), C.c_long(ld), int(row0), int(nrows), int(ncols),
                                         self._p(dst)), "gpak_dev_pack")

    @staticmethod
    def _p(t):
        return C.c_void_p(t.data_ptr())

    def _e(self, expans):
        self._host_e = (C.c_double * 8)(*[float(v) for v in expans])
        return self._host_e

    def _chk(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what} failed with status {rc}")

    # -- tile operations (include/gpak_dev.h) ---------------------------------------------------
    def transform(self, x_soa, xs, n, cap, expans, mu, u):
        m = (C.c_double * 4)(*([float(v) for v in mu] + [0.0] * (4 - len(mu))))
        self._chk(self.lib.gpak_dev_transform(self._st(), self._p(x_soa), xs, n, cap, self._e(expans), m,
                                              self._p(u)), "gpak_dev_transform")

    def fill_b(self, u, cap, n, Np, J, W, expans, bias, sn2, mode, blk, ld):
        self._chk(self.lib.gpak_dev_fill_b(self._st(), self._p(u), cap, n, Np, J, W, self._e(expans),
                                           C.c_double(bias), C.c_double(sn2), mode, self._p(blk), C.c_long(ld)),
                  "gpak_dev_fill_b")

    def factor_panel(self, blk, ld, Np, J, W, inv, info):
        self._chk(self.lib.gpak_dev_factor_panel(self._st(), self._p(blk), C.c_long(ld), Np, J, W, self._p(inv),
                                                 self._p(info)), "gpak_dev_factor_panel")

    def factor_panel_co(self, blk, ld, Np, J, W, inv, info, coresident):
        """gpak_dev_factor_panel with the placement hint (coresident != 0: the 4-wave / 80-VGPR block kernel)."""
        self.lib.gpak_dev_factor_panel_co.restype = C.c_int
        self._chk(self.lib.gpak_dev_factor_panel_co(self._st(), self._p(blk), C.c_long(ld), Np, J, W, self._p(inv),
                                                    self._p(info), int(coresident)), "gpak_dev_factor_panel_co")

    def update_block(self, panel, ldp, prow0, W, blk, ld, Np, Jc, Wc):
        self._chk(self.lib.gpak_dev_update_block(self._st(), self._p(panel), C.c_long(ldp), prow0, W, self._p(blk),
                                                 C.c_long(ld), Np, Jc, Wc), "gpak_dev_update_block")

    def update_cyclic(self, panel, ldp, prow0, W, local, ld, Np, nb, P, rank, lb0, n_local, last_width):
        self._chk(self.lib.gpak_dev_update_cyclic(self._st(), self._p(panel), C.c_long(ldp), prow0, W,
                                                  self._p(local), C.c_long(ld), Np, nb, P, rank, lb0, n_local,
                                                  last_width), "gpak_dev_update_cyclic")

    # row0: global row held in element 0 of each column of `blk` (0 = a full block column of the matrix; J for a
    # packed panel that starts at its diagonal block).  The C-ABI addresses L[r, J+k] as blk[r + k*ld], so a packed
    # panel is passed as a pointer shifted back by row0 elements (never dereferenced below the panel itself).
    @staticmethod
    def _pshift(t, row0):
        return C.c_void_p(t.data_ptr() - 8 * int(row0))

    def trsv_fwd_block(self, blk, ld, Np, J, W, inv, x, out, row0=0):
        self._chk(self.lib.gpak_dev_trsv_fwd_block(self._st(), self._pshift(blk, row0), C.c_long(ld), Np, J, W,
                                                   self._p(inv), self._p(x), self._p(out)), "gpak_dev_trsv_fwd_block")

    def coldot(self, blk, ld, Np, J, W, x, s, row0=0):
        self._chk(self.lib.gpak_dev_coldot(self._st(), self._pshift(blk, row0), C.c_long(ld), Np, J, W, self._p(x),
                                           self._p(s)), "gpak_dev_coldot")

    def trsv_bwd_block(self, blk, ld, J, W, inv, x, out, row0=0):
        self._chk(self.lib.gpak_dev_trsv_bwd_block(self._st(), self._pshift(blk, row0), C.c_long(ld), J, W,
                                                   self._p(inv), self._p(x), self._p(out)), "gpak_dev_trsv_bwd_block")

    def trsv_bwd_packed(self, panel, ldp, row0, Np, J, W, inv, z, scratch, out, rinv=None):
        self._chk(self.lib.gpak_dev_trsv_bwd_packed(self._st(), self._p(panel), C.c_long(ldp), int(row0), Np, J, W,
                                                    self._p(inv), self._p(z), self._p(scratch), self._p(out),
                                                    self._p(rinv) if rinv is not None else C.c_void_p(None)),
                  "gpak_dev_trsv_bwd_packed")

    def diag_inverse(self, panel, ldp, row0, J, W, inv, rinv):
        self._chk(self.lib.gpak_dev_diag_inverse(self._st(), self._p(panel), C.c_long(ldp), int(row0), J, W,
                                                 self._p(inv), self._p(rinv)), "gpak_dev_diag_inverse")

    def logdiag_block(self, blk, ld, J, W, N, out):
        self._chk(self.lib.gpak_dev_logdiag_block(self._st(), self._p(blk), C.c_long(ld), J, W, N, self._p(out)),
                  "gpak_dev_logdiag_block")

    def kmatvec(self, u, cap, n, i0, i1, w, expans, bias, mode, scratch, out):
        self._chk(self.lib.gpak_dev_kmatvec(self._st(), self._p(u), cap, n, i0, i1, self._p(w), self._e(expans),
                                            C.c_double(bias), mode, self._p(scratch), self._p(out)),
                  "gpak_dev_kmatvec")

    def nlz_terms(self, N, y, f, alpha, sn2, out):
        self._chk(self.lib.gpak_dev_nlz_terms(self._st(), N, self._p(y), self._p(f), self._p(alpha),
                                              C.c_double(sn2), self._p(out)), "gpak_dev_nlz_terms")


class DistGP:
    """The hot path on P ranks.  Every rank calls the same methods in the same order."""

    def __init__(self, engine, X, y, nb=512, group=None, pipeline=None):
        torch = _torch()
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.eng = engine
        self.P = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        X = np.asfortranarray(X, dtype=np.float64)
        y = np.ascontiguousarray(y, dtype=np.float64).ravel()
        self.N = X.shape[0]
        assert X.shape[1] == 3, "HIP path handles 3-D inputs"
        self.Np = (self.N + TILE - 1) // TILE * TILE
        self.ld = self.Np + (32 if self.Np >= 1024 else 0)
        self.nb = max(TILE, nb // TILE * TILE)
        self.nJ = (self.Np + self.nb - 1) // self.nb
        self.cap = self.Np
        self.owned = [b for b in range(self.nJ) if b % self.P == self.rank]
        xs = np.zeros((4, self.Np))          # four raw columns (include/gpak_dev.h), the 4th zero for 3-D inputs
        xs[:3, :self.N] = X.T
        self.x_soa = engine.from_numpy(xs.ravel())
        yp = np.zeros(self.Np)
        yp[:self.N] = y
        self.y = engine.from_numpy(yp)
        self.xsum = X.sum(axis=0)
        self.u = engine.empty(5 * self.cap)   # u0, u1, u2, |u|^2, u3 (include/gpak_dev.h)
        # the rank's block columns side by side in ONE array (nb columns each, leading dimension ld), so
        # that the trailing update of all of them is a single launch (gpak_dev_update_cyclic)
        self.local = engine.empty(max(1, len(self.owned)) * self.nb * self.ld)
        self.blk = {b: self.local[i * self.nb * self.ld: i * self.nb * self.ld + self.width(b) * self.ld]
                    for i, b in enumerate(self.owned)}
        self.inv = {b: engine.empty(self.width(b) // TILE * 2 * TILE * TILE) for b in self.owned}
        self.info = engine.zeros(4, dtype=torch.int32)
        self.scratch = engine.empty(64 * self.cap)
        self.small = engine.zeros(8)
        self.alpha = engine.zeros(self.Np)
        self.params = None
        self.bytes_broadcast = 0
        self.panels, self.invs = {}, {}   # pipelined schedule: every rank keeps every packed panel + inverses
        self.rinv, self.rinv_ok = {}, set()   # explicit inverses of the diagonal blocks (valid for the current factor)
        # sub-panel broadcasts pay off when there IS a transfer to hide; one rank keeps whole panels
        # (235.6 vs 231.0 ms at N=32768 on one GPU)
        self.pipeline = ((os.environ.get("GPAK_DIST_PIPELINE", "1") != "0" and self.P > 1) if pipeline is None
                         else bool(pipeline))

    def start(self, b):
        return b * self.nb

    def width(self, b):
        return min(self.nb, self.Np - b * self.nb)

    def owner(self, b):
        return b % self.P

    # GP_utils::set_GP_Pars (GP_Utils.cpp:130-157): parameters are replicated, 10 doubles
    def set_params(self, expans, bias, sn2, dist_mode=1):
        self.params = (np.array(expans, dtype=np.float64), float(bias), float(sn2), int(dist_mode))

    def _bcast(self, t, src, async_op=False):
        if self.P == 1:
            return None
        return self.dist.broadcast(t, src=src, group=self.group, async_op=async_op)

    def _allreduce(self, t):
        if self.P > 1:
            self.dist.all_reduce(t, group=self.group)

    # ---- fill: HybKerns::computeK + the "(sW sW') % K + I" of ldB2_exact, owned columns only ----
    def fill(self):
        e, bias, sn2, mode = self.params
        n = float(self.N)
        # pooled mean of X u X (Kernel.cpp:1391-1392)
        mX1 = n / (2 * n) * self.xsum / n
        mu = n / (2 * n) * self.xsum / n + mX1
        self.eng.transform(self.x_soa, self.Np, self.N, self.cap, e, mu, self.u)
        for b in self.owned:
            self.eng.fill_b(self.u, self.cap, self.N, self.Np, self.start(b), self.width(b), e, bias, sn2, mode,
                            self.blk[b], self.ld)

    # ---- factor: right-looking, one panel of look-ahead, panel broadcast is the only bulk collective
    def _factor_and_pack(self, b):
        J, W = self.start(b), self.width(b)
        rows = self.Np - (J + W)
        if self.rank == self.owner(b):
            self.eng.factor_panel(self.blk[b], self.ld, self.Np, J, W, self.inv[b], self.info)
            if rows > 0:
                return self.blk[b].view(W, self.ld)[:, J + W:self.Np].contiguous().view(-1)
            return None
        return self.eng.empty(W * rows) if rows > 0 else None

    def factor(self, rhs=None):
        """Returns 0, or the first failing column (1-based) like LAPACK dpotrf's info.  With `rhs` (pipelined
        schedule) the forward substitution L^-1 rhs rides along: each rank applies panel b to its own copy right
        after its bulk update, in the slack the serial panel chain leaves on the main stream."""
        self._fwd = None
        self.rinv_ok = set()
        if self.pipeline:
            if rhs is not None:
                self._fwd = (rhs.clone(), self.eng.zeros(self.Np))
            return self._factor_pipelined()
        return self._factor_whole_panels()

    def _finish_info(self):
        torch = _torch()
        info = self.info[:1].to(torch.int64)
        if self.P > 1:
            self.dist.all_reduce(info, op=self.dist.ReduceOp.MIN, group=self.group)
        v = int(info.item())
        return 0 if v == INT_MAX else v

    def _factor_whole_panels(self):
        """One broadcast per outer panel (the simple schedule; kept for comparison and tests)."""
        torch = _torch()
        self.info.fill_(INT_MAX)
        self.bytes_broadcast = 0
        streams = getattr(self.eng, "has_streams", False)
        main = torch.cuda.current_stream() if streams else None
        ps = self.eng.panel_stream() if streams else None
        panel = self._factor_and_pack(0)
        if panel is not None:
            self._bcast(panel, self.owner(0))
            self.bytes_broadcast += panel.numel() * 8
        for b in range(self.nJ):
            J, W = self.start(b), self.width(b)
            rows = self.Np - (J + W)
            if rows <= 0:
                break
            nxt = b + 1
            handle, panel_next = None, None
            if nxt < self.nJ:
                # look-ahead: the next panel's owner updates + factors its column on the side stream
                # (beside the bulk update below) and the broadcast of panel b+1 is posted by every rank
                # before its bulk update, so the transfer overlaps the MFMA work of step b
                if streams:
                    ps.wait_stream(main)  # panel b is complete, earlier bulk updates of blk[nxt] are queued
                    ctx = self.eng.on(ps)
                    ctx.__enter__()
                try:
                    if self.rank == self.owner(nxt):
                        self.eng.update_block(panel, rows, J + W, W, self.blk[nxt], self.ld, self.Np,
                                              self.start(nxt), self.width(nxt))
                    panel_next = self._factor_and_pack(nxt)
                    if panel_next is not None:
                        handle = self._bcast(panel_next, self.owner(nxt), async_op=True)
                        self.bytes_broadcast += panel_next.numel() * 8
                finally:
                    if streams:
                        ctx.__exit__(None, None, None)
            lb0 = next((i for i, c in enumerate(self.owned) if c > nxt), None)
            if lb0 is not None:
                self.eng.update_cyclic(panel, rows, J + W, W, self.local, self.ld, self.Np, self.nb, self.P,
                                       self.rank, lb0, len(self.owned), self.width(self.owned[-1]))
            if handle is not None:
                handle.wait()
            if streams and nxt < self.nJ:
                main.wait_stream(ps)
                if panel_next is not None:
                    panel_next.record_stream(main)
                panel.record_stream(ps)
            panel = panel_next
        return self._finish_info()

    # The serial chain of the distributed factorisation is  factor(b) -> transfer -> update of column b+1
    # -> factor(b+1) ...; per outer panel that is 4 x ~150 us of factor work plus the transfer of up to
    # (N x nb) doubles.  Sending the panel in 128-column sub-panels as they are finished hides three
    # quarters of the transfer under the owner's remaining factor work, and the next owner applies each
    # sub-panel to its column as it lands, so only the last quarter is exposed.
    def _produce(self, b):
        """Factor block column b on its owner, 128 columns at a time; broadcast each sub-panel -- rows from the
        diagonal block down -- as soon as it exists; the owner of b+1 applies it to its column on arrival.
        The inverted diagonal blocks follow in one small broadcast.  Every rank KEEPS the packed panel and the
        inverses (N^2/2 doubles in total: 4.3 GB at N=32768), so that the two triangular solves afterwards need
        no communication at all.  Runs on the side stream.  Returns (packed panel, broadcast handles)."""
        eng, ld, Np = self.eng, self.ld, self.Np
        J, W = self.start(b), self.width(b)
        rows = Np - J                     # packed rows: global rows [J, Np), leading dimension rows
        own = self.rank == self.owner(b)
        nxt = b + 1
        buf = eng.empty(W * rows)
        handles = []
        for s in range(W // TILE):
            if own:
                sub = self.blk[b][s * TILE * ld:(s + 1) * TILE * ld]
                eng.factor_panel(sub, ld, Np, J + s * TILE, TILE,
                                 self.inv[b][s * 2 * TILE * TILE:(s + 1) * 2 * TILE * TILE], self.info)
                rem = W - (s + 1) * TILE
                if rem > 0:  # the rest of the owner's own block column
                    eng.update_block(sub, ld, 0, TILE, self.blk[b][(s + 1) * TILE * ld:], ld, Np,
                                     J + (s + 1) * TILE, rem)
                eng.pack(sub, ld, J, rows, TILE, buf[s * TILE * rows:(s + 1) * TILE * rows])
            chunk = buf[s * TILE * rows:(s + 1) * TILE * rows]
            h = self._bcast(chunk, self.owner(b), async_op=True)
            self.bytes_broadcast += chunk.numel() * 8
            if h is not None:
                handles.append(h)
            if nxt < self.nJ and self.rank == self.owner(nxt):
                if h is not None:
                    h.wait()
                eng.update_block(chunk, rows, J, TILE, self.blk[nxt], ld, Np, self.start(nxt), self.width(nxt))
        inv = self.inv[b] if own else eng.empty(W // TILE * 2 * TILE * TILE)
        h = self._bcast(inv, self.owner(b), async_op=True)
        self.bytes_broadcast += inv.numel() * 8
        if h is not None:
            handles.append(h)
        self.panels[b], self.invs[b] = buf, inv
        return buf, handles

    def _diag_inverse(self, b, panel, rows):
        """Explicit inverse of block column b's diagonal block (width <= 512), queued in the main stream's slack; the
        back substitution then needs one matrix-vector product per block instead of four dependent phases."""
        J, W = self.start(b), self.width(b)
        if W > 512 or not hasattr(self.eng, "diag_inverse"):
            return
        if b not in self.rinv:
            self.rinv[b] = self.eng.empty(512 * 512)
        self.eng.diag_inverse(panel, rows, J, J, W, self.invs[b], self.rinv[b])
        self.rinv_ok.add(b)

    def _factor_pipelined(self):
        """Column c receives panel b <= c-3 in the bulk update of step b (main stream), panel c-2 as one
        K=nb update and panel c-1 sub-panel by sub-panel (both on the side stream, in that order)."""
        torch = _torch()
        self.info.fill_(INT_MAX)
        self.bytes_broadcast = 0
        streams = getattr(self.eng, "has_streams", False)
        main = torch.cuda.current_stream() if streams else None
        ps = self.eng.panel_stream() if streams else None

        class _side:  # `with side:` = queue on the panel stream when the engine has streams
            def __enter__(s2):
                if streams:
                    s2.c = self.eng.on(ps)
                    s2.c.__enter__()

            def __exit__(s2, *a):
                if streams:
                    s2.c.__exit__(*a)
        side = _side()
        if streams:
            ps.wait_stream(main)  # the fill
        with side:
            panel, handles = self._produce(0)
        for b in range(self.nJ):
            J, W = self.start(b), self.width(b)
            rows = self.Np - J            # the packed panel starts at its diagonal block (prow0 = J)
            nxt, nn = b + 1, b + 2
            # panel b is complete on this rank once its broadcasts (non-owners) / packs (owner) are done
            for h in handles:
                h.wait()
            if streams:
                main.wait_stream(ps)
                panel.record_stream(main)
                self.invs[b].record_stream(main)
            if nxt >= self.nJ:
                if self._fwd is not None:
                    self.eng.trsv_fwd_block(panel, rows, self.Np, J, W, self.invs[b], self._fwd[0], self._fwd[1], row0=J)
                    self._diag_inverse(b, panel, rows)
                break
            panel_next, handles_next = None, []
            if streams:
                ps.wait_stream(main)  # bulk update b-1 has finished with block column b+2
            with side:
                if nn < self.nJ and self.rank == self.owner(nn):
                    self.eng.update_block(panel, rows, J, W, self.blk[nn], self.ld, self.Np,
                                          self.start(nn), self.width(nn))
                panel_next, handles_next = self._produce(nxt)
            lb0 = next((i for i, c in enumerate(self.owned) if c > nn), None)
            if lb0 is not None:
                self.eng.update_cyclic(panel, rows, J, W, self.local, self.ld, self.Np, self.nb, self.P,
                                       self.rank, lb0, len(self.owned), self.width(self.owned[-1]))
            if self._fwd is not None:
                self.eng.trsv_fwd_block(panel, rows, self.Np, J, W, self.invs[b], self._fwd[0], self._fwd[1], row0=J)
                self._diag_inverse(b, panel, rows)
            panel, handles = panel_next, handles_next
        if streams:
            main.wait_stream(ps)
        return self._finish_info()

    # ---- solve_chol (GP_Utils.cpp:841-845) with the factor distributed by block columns -------
    def solve(self, rhs):
        """Returns B^-1 rhs (Np entries, replicated). rhs: replicated Np-vector (not modified)."""
        eng = self.eng
        if self.pipeline:
            # every rank holds every packed panel and inverse: both solves run locally, identically on all
            # ranks, with no collective (solve_chol, GP_Utils.cpp:841-845)
            if getattr(self, "_fwd", None) is not None:
                z = self._fwd[1]          # L^-1 rhs was computed during the factorisation
                self._fwd = None
            else:
                xw, z = rhs.clone(), eng.zeros(self.Np)
                for b in range(self.nJ):
                    J, W = self.start(b), self.width(b)
                    eng.trsv_fwd_block(self.panels[b], self.Np - J, self.Np, J, W, self.invs[b], xw, z, row0=J)
            x = eng.zeros(self.Np)
            scratch = eng.empty(24 * 512)
            for b in range(self.nJ - 1, -1, -1):
                J, W = self.start(b), self.width(b)
                eng.trsv_bwd_packed(self.panels[b], self.Np - J, J, self.Np, J, W, self.invs[b], z, scratch, x,
                                    rinv=self.rinv[b] if b in self.rinv_ok else None)
            return x
        xw = rhs.clone() if self.rank == 0 else eng.zeros(self.Np)
        z = eng.zeros(self.Np)
        for b in range(self.nJ):
            J, W = self.start(b), self.width(b)
            self._allreduce(xw[J:J + W])
            if self.rank == self.owner(b):
                eng.trsv_fwd_block(self.blk[b], self.ld, self.Np, J, W, self.inv[b], xw, z)
        self._allreduce(z)
        x = eng.zeros(self.Np)
        s = eng.empty(self.nb)
        for b in range(self.nJ - 1, -1, -1):
            J, W = self.start(b), self.width(b)
            if self.rank == self.owner(b):
                if J + W < self.Np:
                    eng.coldot(self.blk[b], self.ld, self.Np, J, W, x, s)
                    z[J:J + W] -= s[:W]
                eng.trsv_bwd_block(self.blk[b], self.ld, J, W, self.inv[b], z, x)
            self._bcast(x[J:J + W], self.owner(b))
        return x

    # ---- GP_utils::logLikelihood (GP_Utils.cpp:1138-1162) ----------------------------------------
    def nlz(self):
        # with several ranks the serial panel chain decides the step time: everything but the chain runs on a
        # stream that leaves 8 CUs idle, so that potrf128 and the small panel products never queue behind the
        # bulk update's workgroups (single GPU: 253 us instead of 78 us for a contended potrf128)
        if self.P > 1 and getattr(self.eng, "has_streams", False) and os.environ.get("GPAK_DIST_MASK", "8") != "0":
            torch = _torch()
            bulk = self.eng.bulk_stream(int(os.environ.get("GPAK_DIST_MASK", "8")))
            bulk.wait_stream(torch.cuda.current_stream())
            with self.eng.on(bulk):
                v = self._nlz()
            torch.cuda.current_stream().wait_stream(bulk)
            return v
        return self._nlz()

    def _nlz(self):
        torch = _torch()
        e, bias, sn2, mode = self.params
        self.fill()
        rhs = self.y / sn2
        bad = self.factor(rhs)
        if bad:
            return math.nan  # Chol_fail -> quiet NaN, GP_Utils.cpp:1145-1158
        self.alpha = self.solve(rhs)  # alpha = (K + sn2 I)^-1 y
        # f = K*alpha: each rank sums over its slice of source points, then one all-reduce
        per = (self.N + self.P - 1) // self.P
        per = (per + 1) // 2 * 2
        i0, i1 = min(self.N, self.rank * per), min(self.N, (self.rank + 1) * per)
        f = self.eng.zeros(self.Np)
        if i1 > i0:
            self.eng.kmatvec(self.u, self.cap, self.N, i0, i1, self.alpha, e, bias, mode, self.scratch, f)
        self._allreduce(f)
        ld_local = self.eng.zeros(1)
        for b in self.owned:
            self.eng.logdiag_block(self.blk[b], self.ld, self.start(b), self.width(b), self.N, self.small[4:5])
            ld_local += self.small[4:5]
        self._allreduce(ld_local)
        self.eng.nlz_terms(self.N, self.y, f, self.alpha, sn2, self.small[0:2])
        vals = torch.cat([self.small[0:2], ld_local]).cpu().numpy()
        self.quad, self.sumlp, self.logdet = float(vals[0]), float(vals[1]), float(vals[2])
        return self.quad - self.sumlp + self.logdet  # GP_Utils.cpp:1159

    def get_alpha(self):
        return self.alpha[:self.N].cpu().numpy()


# ------------------------------------------------------------------------------------------------
# bench.py --gpus N entry point (launched by torch.distributed.run, one rank per GPU)
# ------------------------------------------------------------------------------------------------
