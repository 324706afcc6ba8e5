"""CPU stand-in for HipEngine, TEST INFRASTRUCTURE ONLY (uses numpy/scipy and the oracle's sigInv).

It lets the world_size>1 gloo tests drive tests/py_schedule.py's DistGP -- the ownership map, the
look-ahead schedule, the panel pack/broadcast and the solve collectives -- on a machine without
a GPU.  It mirrors the semantics of include/gpak_dev.h function by function.
"""
import numpy as np
import scipy.linalg as sl
import torch

TILE = 128


class NumpyEngine:
    def empty(self, n, dtype=None):
        return torch.zeros(int(n), dtype=dtype or torch.float64)

    zeros = empty

    def from_numpy(self, a):
        return torch.from_numpy(np.ascontiguousarray(a).copy())

    def sync(self):
        pass

    @staticmethod
    def _A(expans):
        from oracle import oracle as orc
        par = [expans[0], expans[2], expans[4], expans[1], expans[3], expans[5]]
        return orc.siginv(par)

    def transform(self, x_soa, xs, n, cap, expans, mu, u):
        # FOUR raw columns (include/gpak_dev.h): the 4th is zero for 3-D inputs, its image u3 = InversewidthR * (x3 - mu3)
        X4 = x_soa.numpy().reshape(4, xs)[:, :n].T
        mu = list(mu) + [0.0] * (4 - len(mu))
        U = (X4[:, :3] - np.asarray(mu[:3])[None, :]) @ self._A(expans)
        u3 = expans[7] * (X4[:, 3] - mu[3])
        un = self._un(u, cap)
        un[:5] = 0
        un[:3, :n] = U.T
        un[4, :n] = u3
        un[3, :n] = (U * U).sum(1) + u3 * u3

    @staticmethod
    def decode_kern(kern):
        """gpak_dev.h GPAK_DIST_HYB serialization -> (terms [(kind, params)], white)."""
        n = int(kern[0])
        kinds = [int(kern[1 + t]) for t in range(n)]
        terms, p = [], 5
        for k in kinds:
            m = {0: 8, 1: 2, 2: 3}[k]
            terms.append((k, [float(kern[p + i]) for i in range(m)]))
            p += m
        return terms, float(kern[4])

    def transform_k(self, x_soa, xs, n, cap, kern, mode, mu, u):
        """5 arrays per child: ExpAns -> (x - mu) sigInv and InversewidthR * x3; Exp / RBF -> every column / Hayper."""
        if not mode & 0x20:
            return self.transform(x_soa, xs, n, cap, kern, mu, u)
        terms, _ = self.decode_kern(kern)
        X4 = x_soa.numpy().reshape(4, xs)[:, :n].T
        mu = np.asarray(list(mu) + [0.0] * (4 - len(mu)))
        un = self._un(u, cap)[:15].reshape(3, 5, cap)
        un[:] = 0
        for t, (kind, p) in enumerate(terms):
            if kind == 0:
                U3 = (X4[:, :3] - mu[None, :3]) @ self._A(p)
                u3 = p[7] * (X4[:, 3] - mu[3])
            else:
                U3 = (X4[:, :3] - mu[None, :3]) / p[0]
                u3 = (X4[:, 3] - mu[3]) / p[0]
            un[t, :3, :n] = U3.T
            un[t, 4, :n] = u3
            un[t, 3, :n] = (U3 * U3).sum(1) + u3 * u3

    @classmethod
    def _kfun(cls, un, rows, cols, expans, bias, mode):
        if mode & 0x20:                   # GPAK_DIST_HYB: `expans` is the serialized composition, un has 5 arrays per child
            terms, _white = cls.decode_kern(expans)
            cap = un.shape[1]
            unt = un[:15].reshape(3, 5, cap)
            K = np.full((len(range(*rows.indices(cap))), len(range(*cols.indices(cap)))), float(bias))
            for t, (kind, p) in enumerate(terms):
                sel = [0, 1, 2, 4] if mode & 0x10 else [0, 1, 2]
                P, Q = unt[t][sel][:, rows].T, unt[t][sel][:, cols].T
                if (mode & 0xF) == 1:
                    D2 = ((P[:, None, :] - Q[None, :, :]) ** 2).sum(-1)
                else:
                    D2 = unt[t][3, rows][:, None] + unt[t][3, cols][None, :] - 2 * P @ Q.T
                    D2[D2 < 0] = 0
                if kind == 2:
                    K += p[2] ** 2 * np.exp(-0.5 * p[1] * D2)
                else:
                    K += (p[6] if kind == 0 else p[1]) ** 2 * np.exp(-np.sqrt(D2))
            return K
        d4 = bool(mode & 0x10)            # GPAK_DIST_D4: the transformed 4th column takes part in the distance
        mode &= 0xF
        sel = [0, 1, 2, 4] if d4 else [0, 1, 2]
        P, Q = un[sel][:, rows].T, un[sel][:, cols].T
        if mode == 1:
            D2 = ((P[:, None, :] - Q[None, :, :]) ** 2).sum(-1)
        else:
            D2 = un[3, rows][:, None] + un[3, cols][None, :] - 2 * P @ Q.T
            D2[D2 < 0] = 0
        return expans[6] ** 2 * np.exp(-np.sqrt(D2)) + bias

    @staticmethod
    def _un(u, cap):
        """the transformed points as (arrays x cap): 5 arrays, or 15 when the buffer holds a composition's three children"""
        a = u.numpy()
        rows = a.size // cap
        return a[:rows * cap].reshape(rows, cap)

    def fill_b(self, u, cap, n, Np, J, W, expans, bias, sn2, mode, blk, ld):
        un = self._un(u, cap)
        M = blk.numpy().reshape(W, ld).T  # (ld x W) column-major view
        M[:Np, :] = 0
        nc = max(0, min(W, n - J))
        if nc > 0:
            M[:n, :nc] = self._kfun(un, slice(0, n), slice(J, J + nc), expans, bias, mode) / sn2
        white = self.decode_kern(expans)[1] if mode & 0x20 else 0.0     # Kern_White: Sigma_White on the diagonal
        for c in range(W):
            M[J + c, c] += 1.0 + (white / sn2 if J + c < n else 0.0)

    def factor_panel(self, blk, ld, Np, J, W, inv, info):
        M = blk.numpy().reshape(W, ld).T
        D = np.tril(M[J:J + W, :])
        D = D + np.tril(D, -1).T
        try:
            L = sl.cholesky(D, lower=True)
        except sl.LinAlgError:
            # leading minor that fails, 1-based global column
            k = next(i for i in range(1, W + 1) if np.linalg.eigvalsh(D[:i, :i]).min() <= 0)
            info[0] = min(int(info[0]), J + k)
            L = np.eye(W)
        M[J:J + W, :] = L
        if J + W < Np:
            M[J + W:Np, :] = sl.solve_triangular(L, M[J + W:Np, :].T, lower=True).T
        iv = inv.numpy().reshape(W // TILE, 2, TILE, TILE)
        for k in range(W // TILE):
            X = np.linalg.inv(L[k * TILE:(k + 1) * TILE, k * TILE:(k + 1) * TILE])
            iv[k, 0] = X.T   # stored column-major: memory [c][r] = X[r][c]
            iv[k, 1] = X     # inverse transpose, column-major

    def pack(self, src, ld, row0, nrows, ncols, dst):
        dst.view(ncols, nrows).copy_(src.view(ncols, ld)[:, row0:row0 + nrows])

    def update_block(self, panel, ldp, prow0, W, blk, ld, Np, Jc, Wc):
        P = panel.numpy().reshape(W, ldp).T
        C = blk.numpy().reshape(Wc, ld).T
        a = P[Jc - prow0:Np - prow0, :]
        C[Jc:Np, :] -= a @ a[:Wc, :].T

    def update_cyclic(self, panel, ldp, prow0, W, local, ld, Np, nb, P, rank, lb0, n_local, last_width):
        for lb in range(lb0, n_local):
            Wc = last_width if lb == n_local - 1 else nb
            Jc = (lb * P + rank) * nb
            blk = local[lb * nb * ld: lb * nb * ld + Wc * ld]
            self.update_block(panel, ldp, prow0, W, blk, ld, Np, Jc, Wc)

    # row0: global row held in element 0 of each column (0: full block column; J: packed panel from its diagonal)
    def trsv_fwd_block(self, blk, ld, Np, J, W, inv, x, out, row0=0):
        M = self._rows_from(blk, W, ld, row0, Np)
        iv = inv.numpy().reshape(W // TILE, 2, TILE, TILE)
        xn, on = x.numpy(), out.numpy()
        for k in range(W // TILE):
            j0 = J + k * TILE
            z = iv[k, 0].T @ xn[j0:j0 + TILE]
            on[j0:j0 + TILE] = z
            xn[j0 + TILE:Np] -= M[j0 + TILE:Np, k * TILE:(k + 1) * TILE] @ z

    @staticmethod
    def _rows_from(blk, W, ld, row0, Np):
        """(Np x W) view-like array addressed by GLOBAL row; rows above row0 are never read by the callers."""
        M = blk.numpy().reshape(W, ld).T
        if row0 == 0:
            return M
        full = np.zeros((Np, W))
        full[row0:Np] = M[:Np - row0]
        return full

    def coldot(self, blk, ld, Np, J, W, x, s, row0=0):
        M = self._rows_from(blk, W, ld, row0, Np)
        s.numpy()[:W] = M[J + W:Np, :].T @ x.numpy()[J + W:Np]

    def trsv_bwd_block(self, blk, ld, J, W, inv, x, out, row0=0):
        M = self._rows_from(blk, W, ld, row0, J + W)
        iv = inv.numpy().reshape(W // TILE, 2, TILE, TILE)
        xn, on = x.numpy(), out.numpy()
        for k in range(W // TILE - 1, -1, -1):
            j0 = J + k * TILE
            w = iv[k, 1].T @ xn[j0:j0 + TILE]      # inv(L_jj)^T x_j
            on[j0:j0 + TILE] = w
            if k > 0:
                xn[J:j0] -= M[j0:j0 + TILE, :k * TILE].T @ w

    def trsv_bwd_packed(self, panel, ldp, row0, Np, J, W, inv, z, scratch, out, rinv=None):
        s = torch.zeros(W, dtype=torch.float64)
        if J + W < Np:
            self.coldot(panel, ldp, Np, J, W, out, s, row0=row0)
        v = z.clone()
        v[J:J + W] -= s
        self.trsv_bwd_block(panel, ldp, J, W, inv, v, out, row0=row0)

    def logdiag_block(self, blk, ld, J, W, N, out):
        M = blk.numpy().reshape(W, ld).T
        nc = max(0, min(W, N - J))
        out[0] = float(np.log(np.diag(M[J:J + nc, :nc])).sum()) if nc else 0.0

    def kmatvec(self, u, cap, n, i0, i1, w, expans, bias, mode, scratch, out):
        un = self._un(u, cap)
        K = self._kfun(un, slice(i0, i1), slice(0, n), expans, bias, mode)
        out.numpy()[:n] = w.numpy()[i0:i1] @ K

    def nlz_terms(self, N, y, f, alpha, sn2, out):
        yn, fn, an = y.numpy()[:N], f.numpy()[:N], alpha.numpy()[:N]
        out[0] = float((an * 0.5 * fn).sum())
        out[1] = float((-(yn - fn) ** 2 / (2 * sn2) - np.log(2 * np.pi * sn2) / 2).sum())
