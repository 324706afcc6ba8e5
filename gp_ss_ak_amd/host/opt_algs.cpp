// opt_algs.cpp -- the reference's bound-constrained L-BFGS driver, restated from the source:
//   Opt_Algs::cauchy_point           Opt_pars.cpp:11-105
//   Opt_Algs::Primal_Conjugate_grad  Opt_pars.cpp:108-174
//   Opt_Algs::LBFGSOptimise          Opt_pars.cpp:179-332
//   Opt_Algs::Efficient_line_search  Opt_pars.cpp:543-974  (Potra-Shi style bracketing)
// It is the CALLER of the hot path: each Grad_Values()/ObjVal() is one device evaluation, and the
// number of them per iteration (3 to ~19) is decided here.  The restatement keeps the reference's
// behaviour as written, including what looks unintended -- they all shape the trajectory:
//   * the Cauchy search takes the index of the smallest breakpoint inside the COMPRESSED list of
//     positive breakpoints and uses it as a coordinate index (Opt_pars.cpp:51-54, 72-74);
//   * the subspace step keeps the LARGEST feasible step ratio (max, :150-156) and the returned
//     direction does not include xcp - X unless every variable is fixed (:119-123);
//   * Lk = trimatl(Sk'Yk) keeps the diagonal (:223, 307, 316);
//   * once the memory is full only column 0 of Yk/Sk is replaced and Wk gets the INITIAL
//     gradient and the current best point (:312-323);
//   * gold is the gradient at the last evaluated point even when that point was rejected (:240-243);
//   * ChkBnd sets entries above the upper bound to the LOWER bound (Opt_pars.h:92-98);
//   * fail_pre_bfgs is read before it is ever written in the reference (Opt_pars.h:218,
//     Opt_pars.cpp:577): it is defined false here;
//   * a line search that finds a better point reports steplength 1.0 for it (:593-597).
// PARITY UNPINNED: the reference cannot run here; tests compare this file with an independent
// NumPy restatement of the same source lines (tests/lbfgs_ref.py) on analytic objectives.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <limits>
#include <vector>

#include "gp_utils.hpp"

namespace {
typedef std::vector<double> Vec;
const double epsilon = std::numeric_limits<double>::epsilon();

struct Dm {  // small dense matrix, row-major
  int r = 0, c = 0;
  Vec v;
  Dm() {}
  Dm(int r_, int c_) : r(r_), c(c_), v((size_t)r_ * c_, 0.0) {}
  double &operator()(int i, int j) { return v[(size_t)i * c + j]; }
  double operator()(int i, int j) const { return v[(size_t)i * c + j]; }
};
double dot(const Vec &a, const Vec &b) { double s = 0; for (size_t i = 0; i < a.size(); i++) s += a[i] * b[i]; return s; }
double norm2(const Vec &a) { return std::sqrt(dot(a, a)); }
Vec matvec(const Dm &A, const Vec &x) { Vec y(A.r, 0.0); for (int i = 0; i < A.r; i++) for (int j = 0; j < A.c; j++) y[i] += A(i, j) * x[j]; return y; }
Vec matTvec(const Dm &A, const Vec &x) { Vec y(A.c, 0.0); for (int i = 0; i < A.r; i++) for (int j = 0; j < A.c; j++) y[j] += A(i, j) * x[i]; return y; }
Vec rowof(const Dm &A, int i) { Vec y(A.c); for (int j = 0; j < A.c; j++) y[j] = A(i, j); return y; }
// general inverse (arma::inv): Gauss-Jordan with partial pivoting
Dm inverse(const Dm &A) {
  int n = A.r;
  Dm M = A, I(n, n);
  for (int i = 0; i < n; i++) I(i, i) = 1.0;
  for (int col = 0; col < n; col++) {
    int piv = col;
    for (int i = col + 1; i < n; i++) if (std::fabs(M(i, col)) > std::fabs(M(piv, col))) piv = i;
    if (piv != col) for (int j = 0; j < n; j++) { std::swap(M(col, j), M(piv, j)); std::swap(I(col, j), I(piv, j)); }
    double d = M(col, col);
    for (int j = 0; j < n; j++) { M(col, j) /= d; I(col, j) /= d; }
    for (int i = 0; i < n; i++) if (i != col) {
      double f = M(i, col);
      if (f != 0.0) for (int j = 0; j < n; j++) { M(i, j) -= f * M(col, j); I(i, j) -= f * I(col, j); }
    }
  }
  return I;
}
// Mk = inv([[-Dk, Lk'],[Lk, theta Sk'Sk]]), Lk = trimatl(Sk'Yk)   (Opt_pars.cpp:223-229 and twins)
Dm build_Mk(const Vec &Dk, const Dm &Sk, const Dm &Yk, double theta, int nc) {
  int n = Sk.r;
  Dm M(2 * nc, 2 * nc);
  for (int i = 0; i < nc; i++) M(i, i) = -Dk[i];
  for (int i = 0; i < nc; i++)
    for (int j = 0; j < nc; j++) {
      double sy = 0, ss = 0;
      for (int k = 0; k < n; k++) { sy += Sk(k, i) * Yk(k, j); ss += Sk(k, i) * Sk(k, j); }
      if (i >= j) { M(nc + i, j) = sy; M(j, nc + i) = sy; }   // Lk (lower incl. diagonal) and Lk'
      M(nc + i, nc + j) = theta * ss;
    }
  return inverse(M);
}
}  // namespace

struct LbfgsState {  // the members of Opt_Algs this algorithm uses
  Vec lb, ub;
  bool fail_pre_bfgs = false;
};

static bool ChkBndStat(const Vec &A, const LbfgsState &s) {
  for (size_t i = 0; i < A.size(); i++) if (A[i] < s.lb[i] || A[i] > s.ub[i]) return true;
  return false;
}
static void ChkBnd(Vec &A, const LbfgsState &s) {  // Opt_pars.h:92-98, as written
  for (size_t i = 0; i < A.size(); i++) if (A[i] < s.lb[i]) A[i] = s.lb[i];
  for (size_t i = 0; i < A.size(); i++) if (A[i] > s.ub[i]) A[i] = s.lb[i];
}

// Opt_pars.cpp:11-105
static void cauchy_point(const LbfgsState &st, const Vec &g, const Vec &X, const Dm &Wk, const Dm &Mk, Vec &C,
                         Vec &xcp, Vec &index_r, double theta, int mnc) {
  const double epsi = 1e-100;
  const int n = (int)X.size();
  Vec c(2 * mnc, 0.0);
  index_r.assign(1, 0.0);
  xcp = X;
  Vec t(n), d(n);
  for (int j = 0; j < n; j++) d[j] = -g[j];
  for (int j = 0; j < n; j++) {
    if (g[j] < 0) t[j] = (X[j] - st.ub[j]) / g[j];
    else if (g[j] > 0) t[j] = (X[j] - st.lb[j]) / g[j];
    else t[j] = std::numeric_limits<double>::max();
    if (t[j] > -epsi && t[j] < epsi) d[j] = 0.0;
  }
  Vec F;
  for (int j = 0; j < n; j++) if (t[j] > 0.0) F.push_back(t[j]);
  Vec p = matTvec(Wk, d);
  double fprime = -dot(d, d);
  double fsec = -theta * fprime - dot(p, matvec(Mk, p));
  double dt_min = -fprime / fsec;
  double t_old = 0.0;
  auto take_min = [&](double &mt, int &b) {  // F.min(), F.index_min(), F.shed_row(b)
    b = 0;
    for (size_t i = 1; i < F.size(); i++) if (F[i] < F[b]) b = (int)i;
    mt = F[b];
    F.erase(F.begin() + b);
  };
  double mt = 0;
  int b = 0;
  if (F.empty()) { C = c; return; }  // arma would throw on min() of an empty vector
  take_min(mt, b);
  index_r[0] = b;
  double dt = mt - t_old;
  while (dt_min >= dt && !F.empty()) {
    if (d[b] > 0) xcp[b] = st.ub[b];
    else if (d[b] < 0) xcp[b] = st.lb[b];
    double zb = xcp[b] - X[b];
    for (size_t i = 0; i < c.size(); i++) c[i] += dt * p[i];
    Vec wb = rowof(Wk, b);
    fprime += dt * fsec + g[b] * g[b] + theta * g[b] * zb - g[b] * dot(wb, matvec(Mk, c));
    fsec += -theta * g[b] * g[b] - 2.0 * g[b] * dot(wb, matvec(Mk, p)) - g[b] * g[b] * dot(wb, matvec(Mk, wb));
    for (size_t i = 0; i < p.size(); i++) p[i] += g[b] * wb[i];
    d[b] = 0.0;
    dt_min = -fprime / fsec;
    t_old = mt;
    take_min(mt, b);
    index_r.push_back(b);
    dt = mt - t_old;
  }
  dt_min = std::max(dt_min, 0.0);
  t_old += dt_min;
  for (int i = 0; i < n; i++) if (t[i] >= mt) xcp[i] = X[i] + t_old * d[i];
  for (size_t i = 0; i < F.size(); i++) {
    if (t[i] == mt) { F.erase(F.begin() + i); index_r.push_back((double)i); }
  }
  C.assign(c.size(), 0.0);
  for (size_t i = 0; i < c.size(); i++) C[i] = c[i] + dt_min * p[i];
}

// Opt_pars.cpp:108-174
static void Primal_Conjugate_grad(const LbfgsState &st, const Vec &index_r, const Vec &xcp, const Vec &X,
                                  const Dm &Wk, const Dm &Mk, const Vec &C, const Vec &g, double theta,
                                  Vec &direction) {
  const int maxit = 50, n = (int)X.size();
  std::fill(direction.begin(), direction.end(), 0.0);
  if (n - (int)index_r.size() == 0) {
    for (int i = 0; i < n; i++) direction[i] = xcp[i] - X[i];
    return;
  }
  Vec free_mask(n, 1.0);
  for (int i = 0; i < n; i++)
    for (size_t j = 0; j < index_r.size(); j++) if (index_r[j] == i) { free_mask[i] = 0.0; break; }
  Vec WMC = matvec(Wk, matvec(Mk, C));
  Vec rc(n);
  for (int i = 0; i < n; i++) rc[i] = free_mask[i] * ((g[i] + theta * (xcp[i] - X[i])) - WMC[i]);
  Vec r = rc, p(n);
  for (int i = 0; i < n; i++) p[i] = -r[i];
  double Rho2 = dot(r, r), Rho1 = 0.0;
  int it = 0;
  while (norm2(r) >= std::min(0.1, std::sqrt(norm2(rc))) * norm2(rc)) {
    if (it > maxit) break;
    it++;
    double alpha1 = -std::numeric_limits<double>::infinity();
    for (int i = 0; i < n; i++) {
      if (p[i] < 0) alpha1 = std::max(alpha1, (st.lb[i] - xcp[i] - direction[i]) / p[i]);
      else if (p[i] > 0) alpha1 = std::max(alpha1, (st.ub[i] - xcp[i] - direction[i]) / p[i]);
    }
    // q = (theta I - Wk Mk Wk') p
    Vec q = matvec(Wk, matvec(Mk, matTvec(Wk, p)));
    for (int i = 0; i < n; i++) q[i] = theta * p[i] - q[i];
    double alpha2 = Rho2 / dot(p, q);
    if (alpha2 > alpha1) {
      for (int i = 0; i < n; i++) direction[i] += alpha1 * p[i];
      break;
    } else {
      for (int i = 0; i < n; i++) { direction[i] += alpha2 * p[i]; r[i] += alpha2 * q[i]; }
      Rho1 = Rho2;
      Rho2 = dot(r, r);
      double beta = Rho2 / Rho1;
      for (int i = 0; i < n; i++) p[i] = -r[i] + beta * p[i];
    }
  }
}

// Opt_pars.cpp:543-974
static void Efficient_line_search(Opt_Algs *self, LbfgsState &st, const double fxk, const Vec &X, const Vec &gk,
                                  const Vec &sk, double &final_steplength) {
  double rho = 1e-14, sig = 0.99, J = 2.0, tau1 = 1e-14, tau2 = 0.49, tau3 = 2.1;
  const int maxls = 4, n = (int)X.size();
  double steplength = 1.0, a = 0.0, b = steplength;
  bool returnflg = false;
  const double f0 = fxk;
  double global_val = f0;
  const double fprim0 = dot(gk, sk);
  Vec gnew = gk, Xnew(n);
  mat P(1, n), G(1, n);
  auto at = [&](double s) { for (int i = 0; i < n; i++) Xnew[i] = X[i] + s * sk[i]; };
  auto set = [&]() { for (int i = 0; i < n; i++) P(i) = Xnew[i]; self->set_GP_Pars(P); };
  auto objval = [&]() { set(); self->numFuncEval++; return self->ObjVal(); };
  auto gradval = [&](Vec &gout) { set(); self->numFuncEval++; double f = self->Grad_Values(G); for (int i = 0; i < n; i++) gout[i] = G(i); return f; };
  // "while (violate) { s /= div; ...; if (s < epsilon) { Xnew = X; s = 0; break; } }"
  auto shrink = [&](double &s, double div) {
    at(s);
    bool violate = ChkBndStat(Xnew, st);
    while (violate) {
      s /= div;
      at(s);
      violate = ChkBndStat(Xnew, st);
      if (s < epsilon) { Xnew = X; s = 0.0; break; }
    }
  };
  auto better = [&](double f, double s) { if (f < global_val) { final_steplength = s; global_val = f; } };

  if (st.fail_pre_bfgs) steplength = -1.0;
  shrink(steplength, 1.2);
  const double f1 = gradval(gnew);
  if (f1 < global_val) { final_steplength = 1.0; global_val = f1; }
  double fa = 0, fb = 0;
  if (f1 > f0 + rho * fprim0) {
    a = 0.0; b = steplength;
    at(a);
    fa = objval();
    better(fa, a);
    shrink(b, 1.2);
    fb = objval();
    better(fb, b);
  } else {
    if (sig > 0.5) {
      if (f1 >= f0 + sig * fprim0) { final_steplength = 1.0; returnflg = true; }
    } else {
      double fprim1 = dot(gnew, sk);
      if (fprim1 >= sig * fprim0) { final_steplength = 1.0; returnflg = true; }
    }
    if (returnflg) { st.fail_pre_bfgs = !(global_val <= f0); return; }
    double an = 1.0, bn = J;
    shrink(an, 1.2);
    fa = objval();
    better(fa, an);
    shrink(bn, 1.2);
    fb = objval();
    better(fb, bn);
    while (true) {
      if (fb > fa + (bn - an) * rho * fprim0) { a = an; b = bn; break; }
      else if (fb >= fa + (bn - an) * sig * fprim0) { final_steplength = bn; returnflg = true; break; }
      else {
        an = bn;
        bn = J * bn;
        at(an);
        bool violate = ChkBndStat(Xnew, st);
        while (violate) {  // :744-756: divides by 1.2 AND by 2 each round
          an /= 1.2;
          at(an);
          violate = ChkBndStat(Xnew, st);
          an /= 2.0;
          if (an < epsilon) { Xnew = X; an = 0.0; break; }
        }
        if (fa != fa || fb != fb) { returnflg = true; break; }
        fa = objval();
        better(fa, an);
        shrink(bn, 1.2);
        fb = objval();
        better(fb, bn);
      }
    }
  }
  if (returnflg) { st.fail_pre_bfgs = !(global_val <= f0); return; }

  // step 3
  double an = a, bn = b, cn = an, deltan = 0.0;
  int it = 0;
  Vec glow(n), ghigh(n);
  while (it < maxls) {
    it++;
    double lowv = an + tau1 * (bn - an), highv = an + tau2 * (bn - an);
    at(lowv);
    bool violate = ChkBndStat(Xnew, st);
    while (violate) {
      tau1 /= 1.2;
      lowv = an + tau1 * (bn - an);
      at(lowv);
      violate = ChkBndStat(Xnew, st);
      if (lowv < epsilon) { Xnew = X; lowv = 0.0; break; }
    }
    glow = gk; ghigh = gk;
    double flow = gradval(glow);
    better(flow, lowv);
    at(highv);
    violate = ChkBndStat(Xnew, st);
    while (violate) {
      tau2 /= 1.1;
      highv = an + tau2 * (bn - an);
      at(highv);
      violate = ChkBndStat(Xnew, st);
      if (tau2 >= tau1) break;
      if (highv < epsilon) { Xnew = X; highv = 0.0; break; }
    }
    double fhigh = gradval(ghigh);
    better(fhigh, highv);
    double fprimlow = dot(glow, sk), fprimhigh = dot(ghigh, sk);
    auto interp = [&](double x) {
      return (flow + (x - lowv) * fprimlow) * (highv - x) / (highv - lowv) +
             (fhigh + (x - highv) * fprimhigh) * (x - lowv) / (highv - lowv);
    };
    double x0 = 0.25 * (lowv + highv), x1 = 0.5 * (lowv + highv), x2 = 0.75 * (lowv + highv);
    double y0 = interp(x0), y1 = interp(x1), y2 = interp(x2);
    double minf = std::min(std::min(y0, y1), y2);
    if (minf == y0) cn = x0;
    else if (minf == y1) cn = x1;
    else if (minf == y2) cn = x2;
    shrink(cn, 1.1);
    double fcn = objval();
    better(fcn, cn);
    if (it == 1) deltan = std::fabs(((fb - fcn) / (bn - cn) - (fcn - fa) / (cn - an)) / (bn - an));
    if (fcn <= fa + (cn - an) * rho * fprim0 && fcn >= fa + (cn - an) * sig * fprim0) {
      final_steplength = cn;
      returnflg = true;
      break;
    } else {
      deltan = std::fabs(((fb - fcn) / (bn - cn) - (fcn - fa) / (cn - an)) / (bn - an));
    }
    if (fcn <= fa + (cn - an) * rho * fprim0) {
      if ((rho - sig) * fprim0 >= tau3 * (bn - an) * deltan) {
        steplength = cn;
      } else {
        an = cn;
        at(an);
        ChkBnd(Xnew, st);
        fa = objval();
        better(fa, an);
      }
    } else {
      if ((rho - sig) * fprim0 >= tau3 * (bn - an) * deltan && an > 0) {
        final_steplength = an;
        returnflg = true;
        break;
      } else {
        bn = cn;
        at(bn);
        ChkBnd(Xnew, st);
        fb = objval();
        if (fcn < global_val) { final_steplength = bn; global_val = fb; }  // as written (:951-955)
      }
    }
  }
  st.fail_pre_bfgs = !(global_val < f0);
  if (!returnflg) final_steplength = steplength;
}

// GPAK_OPT_TRACE=<file>: one line per iteration with the kept objective, the evaluation count so far and the
// kept point at 17 significant digits (stdout carries 6, like the reference's `cout << fx`); test hook only.
static void trace_iter(int iter, double fx, unsigned nfev, const Vec &x) {
  const char *path = getenv("GPAK_OPT_TRACE");
  if (!path) return;
  FILE *f = fopen(path, "a");
  if (!f) return;
  fprintf(f, "%d %.17g %u", iter, fx, nfev);
  for (double v : x) fprintf(f, " %.17g", v);
  fprintf(f, "\n");
  fclose(f);
}

// Opt_pars.cpp:179-332
void Opt_Algs::LBFGSOptimise() {
  const int n = (int)getNumPars();
  LbfgsState st;
  st.lb.assign(n, 1e-4);
  st.ub.assign(n, 6.0);
  int nc = 1;
  const int mnc = 6;
  double theta = 0.9;
  Vec C(2 * mnc, 0.0), index_r(1, 0.0);
  mat P(1, n), G(1, n);
  get_GP_Pars(P);
  set_GP_Pars(P);
  Vec X0(n), g(n);
  for (int i = 0; i < n; i++) X0[i] = P(i);
  double fx = Grad_Values(G);
  numFuncEval++;
  for (int i = 0; i < n; i++) g[i] = G(i);
  Vec xcp = X0, search_direction(n, 0.0);
  Vec Dk(1, dot(X0, g));
  Dm Yk(n, 1), Sk(n, 1), Wk(n, 2);
  for (int i = 0; i < n; i++) { Yk(i, 0) = g[i]; Sk(i, 0) = X0[i]; Wk(i, 0) = g[i]; Wk(i, 1) = theta * X0[i]; }
  Dm Mk = build_Mk(Dk, Sk, Yk, theta, nc);
  const int Maxit = (int)getMaxIters();
  int iter = 0;
  Vec gnew = g, Xnew = X0;
  double final_steplength = 1;
  while (true) {
    iter++;
    Vec gold = gnew, Xold = Xnew;
    cauchy_point(st, gold, X0, Wk, Mk, C, xcp, index_r, theta, nc);
    Primal_Conjugate_grad(st, index_r, xcp, X0, Wk, Mk, C, gold, theta, search_direction);
    Efficient_line_search(this, st, fx, X0, gold, search_direction, final_steplength);
    for (int i = 0; i < n; i++) Xnew[i] = X0[i] + final_steplength * search_direction[i];
    bool violate = ChkBndStat(Xnew, st);
    while (violate) {
      final_steplength /= 1.2;
      for (int i = 0; i < n; i++) Xnew[i] = X0[i] + final_steplength * search_direction[i];
      violate = ChkBndStat(Xnew, st);
      if (final_steplength < epsilon) { Xnew = X0; final_steplength = 0.0; break; }
    }
    for (int i = 0; i < n; i++) P(i) = Xnew[i];
    set_GP_Pars(P);
    double fnew = Grad_Values(G);
    numFuncEval++;
    for (int i = 0; i < n; i++) gnew[i] = G(i);
    if (fnew < fx) { X0 = Xnew; fx = fnew; }
    Vec yk(n), sk(n);
    for (int i = 0; i < n; i++) { yk[i] = gnew[i] - gold[i]; sk[i] = Xnew[i] - Xold[i]; }
    if (dot(sk, yk) <= epsilon * dot(yk, yk)) {
      if (getVerbose() > 0) std::cout << "Iteration: " << iter << " -logL: " << fx << std::endl;
      trace_iter(iter, fx, numFuncEval, X0);
      if (iter >= Maxit) break;
      continue;
    }
    if (nc < mnc) {
      nc++;
      Dk.push_back(dot(sk, yk));
      Dm Y2(n, nc), S2(n, nc);
      for (int i = 0; i < n; i++) {
        for (int j = 0; j < nc - 1; j++) { Y2(i, j) = Yk(i, j); S2(i, j) = Sk(i, j); }
        Y2(i, nc - 1) = yk[i]; S2(i, nc - 1) = sk[i];
      }
      Yk = Y2; Sk = S2;
      Wk = Dm(n, 2 * nc);
      for (int i = 0; i < n; i++)
        for (int j = 0; j < nc; j++) { Wk(i, j) = Yk(i, j); Wk(i, nc + j) = theta * Sk(i, j); }
      Mk = build_Mk(Dk, Sk, Yk, theta, nc);
    } else {
      Dk[0] = dot(sk, yk);
      for (int i = 0; i < n; i++) { Yk(i, 0) = yk[i]; Sk(i, 0) = sk[i]; Wk(i, 0) = g[i]; Wk(i, mnc) = theta * X0[i]; }
      Mk = build_Mk(Dk, Sk, Yk, theta, mnc);
    }
    theta = dot(yk, yk) / dot(yk, sk);
    trace_iter(iter, fx, numFuncEval, X0);
    if (iter >= Maxit) break;
    if (getVerbose() > 0) std::cout << "Iteration: " << iter << " -logL: " << fx << std::endl;
  }
  for (int i = 0; i < n; i++) P(i) = X0[i];
  set_GP_Pars(P);
}
