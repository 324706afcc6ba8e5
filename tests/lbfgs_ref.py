"""NumPy restatement of the reference's L-BFGS driver, written from Opt_pars.cpp (:11-332,
:543-974) independently of gp_ss_ak_amd/host/opt_algs.cpp.  TEST INFRASTRUCTURE: the two
restatements are compared with each other on analytic objectives (the reference itself cannot
run here, so its trajectory is unpinned)."""
import numpy as np

EPS = np.finfo(float).eps


# Every reduction below is a left-to-right loop in plain Python floats: the algorithm is full of
# exact comparisons (minf == y0, t[i] == mt, fnew < fx ...), so the two restatements are compared
# bit for bit and must round identically.
def sdot(a, b):
    s = 0.0
    for x, y in zip(a, b):
        s += float(x) * float(y)
    return s


def smatvec(A, x):
    return np.array([sdot(A[i, :], x) for i in range(A.shape[0])])


def smatTvec(A, x):
    out = [0.0] * A.shape[1]
    for i in range(A.shape[0]):
        for j in range(A.shape[1]):
            out[j] += float(A[i, j]) * float(x[i])
    return np.array(out)


def snorm(a):
    return float(np.sqrt(sdot(a, a)))


def sinv(A):
    """Gauss-Jordan with partial pivoting (what `inv` amounts to for these <= 12 x 12 matrices)."""
    n = A.shape[0]
    M = [[float(A[i, j]) for j in range(n)] for i in range(n)]
    Iv = [[1.0 if i == j else 0.0 for j in range(n)] for i in range(n)]
    for col in range(n):
        piv = col
        for i in range(col + 1, n):
            if abs(M[i][col]) > abs(M[piv][col]):
                piv = i
        if piv != col:
            M[col], M[piv] = M[piv], M[col]
            Iv[col], Iv[piv] = Iv[piv], Iv[col]
        d = M[col][col]
        for j in range(n):
            M[col][j] /= d
            Iv[col][j] /= d
        for i in range(n):
            if i != col:
                f = M[i][col]
                if f != 0.0:
                    for j in range(n):
                        M[i][j] -= f * M[col][j]
                        Iv[i][j] -= f * Iv[col][j]
    return np.array(Iv)


class Problem:
    """callbacks of Opt_pars.h:51-55"""

    def __init__(self, f_and_g, x0):
        self.fg = f_and_g
        self.x = np.array(x0, dtype=float)
        self.nfev = 0

    def set(self, x):
        self.x = np.array(x, dtype=float)

    def grad_values(self):
        self.nfev += 1
        return self.fg(self.x)

    def objval(self):
        self.nfev += 1
        return self.fg(self.x)[0]


def chk_bnd_stat(A, lb, ub):
    return bool(np.any(A < lb) or np.any(A > ub))


def chk_bnd(A, lb, ub):
    A = A.copy()
    lo = A < lb
    A[lo] = lb[lo]
    hi = A > ub
    A[hi] = lb[hi]          # Opt_pars.h:96-97 writes lb here
    return A


def build_mk(Dk, Sk, Yk, theta):
    nc = len(Dk)
    M = np.zeros((2 * nc, 2 * nc))
    for i in range(nc):
        M[i, i] = -Dk[i]
    for i in range(nc):
        for j in range(nc):
            sy, ss = sdot(Sk[:, i], Yk[:, j]), sdot(Sk[:, i], Sk[:, j])
            if i >= j:                     # Lk = trimatl(Sk'Yk) keeps the diagonal
                M[nc + i, j] = sy
                M[j, nc + i] = sy
            M[nc + i, nc + j] = theta * ss
    return sinv(M)


def cauchy_point(g, X, Wk, Mk, theta, mnc, lb, ub):
    epsi = 1e-100
    n = len(X)
    c = np.zeros(2 * mnc)
    xcp = X.copy()
    d = -g.copy()
    t = np.zeros(n)
    for j in range(n):
        if g[j] < 0:
            t[j] = (X[j] - ub[j]) / g[j]
        elif g[j] > 0:
            t[j] = (X[j] - lb[j]) / g[j]
        else:
            t[j] = np.finfo(float).max
        if -epsi < t[j] < epsi:
            d[j] = 0.0
    F = [t[j] for j in range(n) if t[j] > 0.0]
    p = smatTvec(Wk, d)
    fprime = -sdot(d, d)
    fsec = np.float64(-theta * fprime - sdot(p, smatvec(Mk, p)))
    with np.errstate(all="ignore"):
        dt_min = np.float64(-fprime) / fsec
    t_old = 0.0
    if not F:
        return c, xcp, [0.0]
    b = int(np.argmin(F))
    mt = F.pop(b)
    index_r = [float(b)]
    dt = mt - t_old
    while dt_min >= dt and len(F) > 0:
        if d[b] > 0:
            xcp[b] = ub[b]
        elif d[b] < 0:
            xcp[b] = lb[b]
        zb = xcp[b] - X[b]
        c = c + dt * p
        wb = Wk[b, :]
        fprime += dt * fsec + g[b] * g[b] + theta * g[b] * zb - g[b] * sdot(wb, smatvec(Mk, c))
        fsec += -theta * g[b] * g[b] - 2.0 * g[b] * sdot(wb, smatvec(Mk, p)) - g[b] * g[b] * sdot(wb, smatvec(Mk, wb))
        p = p + g[b] * wb
        d[b] = 0.0
        with np.errstate(all="ignore"):
            dt_min = np.float64(-fprime) / np.float64(fsec)
        t_old = mt
        b = int(np.argmin(F))
        mt = F.pop(b)
        index_r.append(float(b))
        dt = mt - t_old
    dt_min = max(dt_min, 0.0)
    t_old += dt_min
    for i in range(n):
        if t[i] >= mt:
            xcp[i] = X[i] + t_old * d[i]
    i = 0
    while i < len(F):
        if t[i] == mt:
            F.pop(i)
            index_r.append(float(i))
        i += 1
    C = c + dt_min * p
    return C, xcp, index_r


def primal_cg(index_r, xcp, X, Wk, Mk, C, g, theta, lb, ub):
    n = len(X)
    direction = np.zeros(n)
    if n - len(index_r) == 0:
        return xcp - X
    Z = np.ones(n)
    for i in range(n):
        if any(ir == i for ir in index_r):
            Z[i] = 0.0
    rc = Z * ((g + theta * (xcp - X)) - smatvec(Wk, smatvec(Mk, C)))
    r = rc.copy()
    p = -r
    rho2, it = sdot(r, r), 0
    while snorm(r) >= min(0.1, np.sqrt(snorm(rc))) * snorm(rc):
        if it > 50:
            break
        it += 1
        alpha1 = -np.inf
        for i in range(n):
            if p[i] < 0:
                alpha1 = max(alpha1, (lb[i] - xcp[i] - direction[i]) / p[i])
            elif p[i] > 0:
                alpha1 = max(alpha1, (ub[i] - xcp[i] - direction[i]) / p[i])
        q = theta * p - smatvec(Wk, smatvec(Mk, smatTvec(Wk, p)))
        with np.errstate(all="ignore"):
            alpha2 = np.float64(rho2) / np.float64(sdot(p, q))
        if alpha2 > alpha1:
            direction = direction + alpha1 * p
            break
        direction = direction + alpha2 * p
        r = r + alpha2 * q
        rho1, rho2 = rho2, sdot(r, r)
        with np.errstate(all="ignore"):
            p = -r + (np.float64(rho2) / np.float64(rho1)) * p
    return direction


class LineSearch:
    def __init__(self):
        self.fail_pre_bfgs = False   # uninitialised in the reference (Opt_pars.h:218)

    def run(self, prob, fxk, X, gk, sk, final_steplength, lb, ub):
        with np.errstate(all="ignore"):
            return self._run(prob, fxk, X, gk, sk, final_steplength, lb, ub)

    def _run(self, prob, fxk, X, gk, sk, final_steplength, lb, ub):
        # np.float64 scalars: division by zero must give inf/nan as in C++, not raise
        rho, sig, J, tau1, tau2, tau3 = (np.float64(v) for v in (1e-14, 0.99, 2.0, 1e-14, 0.49, 2.1))
        maxls = 4
        steplength, a, b = np.float64(1.0), np.float64(0.0), np.float64(1.0)
        fxk = np.float64(fxk)
        returnflg = False
        f0 = fxk
        gv = [f0, final_steplength]    # global_val, final_steplength

        def better(f, s):
            if f < gv[0]:
                gv[0], gv[1] = f, s

        def shrink(s, div):
            Xn = X + s * sk
            while chk_bnd_stat(Xn, lb, ub):
                s /= div
                Xn = X + s * sk
                if s < EPS:
                    return np.float64(0.0), X.copy()
            return s, Xn

        def obj(Xn):
            prob.set(Xn)
            return prob.objval()

        def grd(Xn):
            prob.set(Xn)
            return prob.grad_values()

        fprim0 = np.float64(sdot(gk, sk))
        if self.fail_pre_bfgs:
            steplength = np.float64(-1.0)
        steplength, Xn = shrink(steplength, 1.2)
        f1, gnew = grd(Xn)
        if f1 < gv[0]:
            gv[0], gv[1] = f1, 1.0
        fa = fb = 0.0
        if f1 > f0 + rho * fprim0:
            a, b = 0.0, steplength
            fa = obj(X + a * sk)
            better(fa, a)
            b, Xn = shrink(b, 1.2)
            fb = obj(Xn)
            better(fb, b)
        else:
            if f1 >= f0 + sig * fprim0:     # sig > 0.5 branch
                gv[1] = 1.0
                returnflg = True
            if returnflg:
                self.fail_pre_bfgs = not (gv[0] <= f0)
                return gv[1]
            an, bn = np.float64(1.0), J
            an, Xn = shrink(an, 1.2)
            fa = obj(Xn)
            better(fa, an)
            bn, Xn = shrink(bn, 1.2)
            fb = obj(Xn)
            better(fb, bn)
            while True:
                if fb > fa + (bn - an) * rho * fprim0:
                    a, b = an, bn
                    break
                elif fb >= fa + (bn - an) * sig * fprim0:
                    gv[1] = bn
                    returnflg = True
                    break
                else:
                    an = bn
                    bn = J * bn
                    Xn = X + an * sk
                    while chk_bnd_stat(Xn, lb, ub):
                        an /= 1.2
                        Xn = X + an * sk
                        viol = chk_bnd_stat(Xn, lb, ub)
                        an /= 2.0
                        if an < EPS:
                            Xn, an = X.copy(), np.float64(0.0)
                            break
                        if not viol:
                            break
                    if fa != fa or fb != fb:
                        returnflg = True
                        break
                    fa = obj(Xn)
                    better(fa, an)
                    bn, Xn = shrink(bn, 1.2)
                    fb = obj(Xn)
                    better(fb, bn)
        if returnflg:
            self.fail_pre_bfgs = not (gv[0] <= f0)
            return gv[1]
        an, bn, cn, deltan, it = a, b, a, np.float64(0.0), 0
        while it < maxls:
            it += 1
            lowv = an + tau1 * (bn - an)
            highv = an + tau2 * (bn - an)
            Xn = X + lowv * sk
            while chk_bnd_stat(Xn, lb, ub):
                tau1 /= 1.2
                lowv = an + tau1 * (bn - an)
                Xn = X + lowv * sk
                if lowv < EPS and chk_bnd_stat(Xn, lb, ub) is not None:
                    if lowv < EPS:
                        Xn, lowv = X.copy(), np.float64(0.0)
                        break
            flow, glow = grd(Xn)
            better(flow, lowv)
            Xn = X + highv * sk
            while chk_bnd_stat(Xn, lb, ub):
                tau2 /= 1.1
                highv = an + tau2 * (bn - an)
                Xn = X + highv * sk
                if tau2 >= tau1:
                    break
                if highv < EPS:
                    Xn, highv = X.copy(), np.float64(0.0)
                    break
            fhigh, ghigh = grd(Xn)
            better(fhigh, highv)
            fpl, fph = np.float64(sdot(glow, sk)), np.float64(sdot(ghigh, sk))
            flow, fhigh = np.float64(flow), np.float64(fhigh)

            def interp(x):
                return ((flow + (x - lowv) * fpl) * (highv - x) / (highv - lowv) +
                        (fhigh + (x - highv) * fph) * (x - lowv) / (highv - lowv))
            xs = [0.25 * (lowv + highv), 0.5 * (lowv + highv), 0.75 * (lowv + highv)]
            with np.errstate(all="ignore"):
                ys = [interp(x) for x in xs]
            minf = min(min(ys[0], ys[1]), ys[2])
            if minf == ys[0]:
                cn = xs[0]
            elif minf == ys[1]:
                cn = xs[1]
            elif minf == ys[2]:
                cn = xs[2]
            cn, Xn = shrink(cn, 1.1)
            fcn = np.float64(obj(Xn))
            better(fcn, cn)
            fa, fb = np.float64(fa), np.float64(fb)
            with np.errstate(all="ignore"):
                dl = abs(((fb - fcn) / (bn - cn) - (fcn - fa) / (cn - an)) / (bn - an))
            if it == 1:
                deltan = dl
            if fcn <= fa + (cn - an) * rho * fprim0 and fcn >= fa + (cn - an) * sig * fprim0:
                gv[1] = cn
                returnflg = True
                break
            else:
                deltan = dl
            if fcn <= fa + (cn - an) * rho * fprim0:
                if (rho - sig) * fprim0 >= tau3 * (bn - an) * deltan:
                    steplength = cn
                else:
                    an = cn
                    fa = obj(chk_bnd(X + an * sk, lb, ub))
                    better(fa, an)
            else:
                if (rho - sig) * fprim0 >= tau3 * (bn - an) * deltan and an > 0:
                    gv[1] = an
                    returnflg = True
                    break
                else:
                    bn = cn
                    fb = obj(chk_bnd(X + bn * sk, lb, ub))
                    if fcn < gv[0]:
                        gv[1], gv[0] = bn, fb
        self.fail_pre_bfgs = not (gv[0] < f0)
        if not returnflg:
            gv[1] = steplength
        return gv[1]


def lbfgs_optimise(f_and_g, x0, maxit, trace=None):
    """Returns (x_best, [fx after each iteration], number of function evaluations).
    trace: optional list that receives (fx, evaluations so far, kept point) after every iteration."""
    prob = Problem(f_and_g, x0)
    n = len(x0)
    lb, ub = np.full(n, 1e-4), np.full(n, 6.0)
    nc, mnc, theta = 1, 6, 0.9
    X0 = np.array(x0, dtype=float)
    prob.set(X0)
    fx, g = prob.grad_values()
    Dk = [sdot(X0, g)]
    Yk, Sk = g.reshape(n, 1).copy(), X0.reshape(n, 1).copy()
    Wk = np.column_stack([g, theta * X0])
    Mk = build_mk(Dk, Sk, Yk, theta)
    gnew, Xnew = g.copy(), X0.copy()
    ls = LineSearch()
    final = 1.0
    hist, it = [], 0
    while True:
        it += 1
        gold, Xold = gnew.copy(), Xnew.copy()
        C, xcp, index_r = cauchy_point(gold, X0, Wk, Mk, theta, nc, lb, ub)
        sd = primal_cg(index_r, xcp, X0, Wk, Mk, C, gold, theta, lb, ub)
        final = ls.run(prob, fx, X0, gold, sd, final, lb, ub)
        Xnew = X0 + final * sd
        while chk_bnd_stat(Xnew, lb, ub):
            final /= 1.2
            Xnew = X0 + final * sd
            if final < EPS:
                Xnew, final = X0.copy(), 0.0
                break
        prob.set(Xnew)
        fnew, gnew = prob.grad_values()
        if fnew < fx:
            X0, fx = Xnew.copy(), fnew
        yk, sk = gnew - gold, Xnew - Xold
        if sdot(sk, yk) <= EPS * sdot(yk, yk):
            hist.append(fx)
            if trace is not None:
                trace.append((fx, prob.nfev, X0.copy()))
            if it >= maxit:
                break
            continue
        if nc < mnc:
            nc += 1
            Dk.append(sdot(sk, yk))
            Yk = np.column_stack([Yk, yk])
            Sk = np.column_stack([Sk, sk])
            Wk = np.column_stack([Yk, theta * Sk])
            Mk = build_mk(Dk, Sk, Yk, theta)
        else:
            Dk[0] = sdot(sk, yk)
            Yk[:, 0], Sk[:, 0] = yk, sk
            Wk[:, 0], Wk[:, mnc] = g, theta * X0
            Mk = build_mk(Dk, Sk, Yk, theta)
        with np.errstate(all="ignore"):
            theta = np.float64(sdot(yk, yk)) / np.float64(sdot(yk, sk))
        hist.append(fx)
        if trace is not None:
            trace.append((fx, prob.nfev, X0.copy()))
        if it >= maxit:
            break
    return X0, hist, prob.nfev
