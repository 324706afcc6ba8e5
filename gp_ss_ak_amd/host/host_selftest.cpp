// host_selftest.cpp -- exercises the class surface the way the reference's own call sites do
// (gp_ss_ak.cpp:139-303, 376-412) and prints one JSON object; tests/test_host_cpp.py compares
// it with the CPU checker.  Usage: host_selftest train.csv test.csv
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <fstream>
#include <iostream>

#include "control.hpp"
#include "gp_utils.hpp"

static void print_vec(const char *name, const mat &v, bool last = false) {
  printf("\"%s\": [", name);
  for (size_t i = 0; i < v.n_elem; i++) printf("%s%.17g", i ? ", " : "", v[i]);
  printf("]%s\n", last ? "" : ",");
}

// CPU-only part: argument walker, data reader, the three standardisations and their inverse,
// kernel parameter tables and the text model format -- nothing here touches the device.
static int logic_only(const char *train_csv) {
  char a0[] = "gp_ss_ak", a1[] = "-v", a2[] = "2", a3[] = "-pm", a4[] = "1", a5[] = "train";
  char *cargv[] = {a0, a1, a2, a3, a4, a5, nullptr};
  Control io(6, cargv);
  io.setMode("train");
  int ds[2];
  mat X, y;
  io.readDataSize(train_csv, ds);
  io.readDataFile(X, y, ds, train_csv);
  mat X0 = X, y0 = y;
  io.prepareData(X, y, true, "/tmp/gpak_logic_model");
  printf("{\n\"verb\": \"%s\", \"verbose\": %d, \"n\": %d, \"d\": %d,\n", io.getArg().c_str(), io.getVerbose(), ds[0], ds[1]);
  printf("\"xmin\": %.17g, \"xmax\": %.17g, \"ymin\": %.17g, \"ymax\": %.17g,\n", X.min(), X.max(), y.min(), y.max());
  mat Xb = X, yb = y;
  io.postData(Xb, yb, true, "/tmp/gpak_logic_model");
  double ex = 0, ey = 0;
  for (size_t i = 0; i < Xb.n_elem; i++) ex = std::max(ex, std::fabs(Xb[i] - X0[i]));
  for (size_t i = 0; i < yb.n_elem; i++) ey = std::max(ey, std::fabs(yb[i] - y0[i]));
  printf("\"roundtrip_x\": %.3g, \"roundtrip_y\": %.3g,\n", ex, ey);
  HybKerns Kerns(X);
  Kern_ExpAnisotropic ea(X);
  Kern_Bias kb(X);
  Kerns.addNewKernel(&ea);
  Kerns.addNewKernel(&kb);
  Kerns.setParam(1.23456789, 1);
  Kerns.setParam(2.0, 3);
  std::ofstream out("/tmp/gpak_logic_kernel");
  Kerns.StrmOut(out);
  out.close();
  std::ifstream in("/tmp/gpak_logic_kernel");
  Kernels *k2 = ReadKerFromFile(in);
  printf("\"npars\": %u, \"name8\": \"%s\", \"p1_reloaded\": %.17g, \"p3_reloaded\": %.17g, \"p8\": %.17g\n}\n",
         k2->getNPars(), k2->getParamName(8).c_str(), k2->getParam(1), k2->getParam(3), k2->getParam(8));
  delete k2;
  return 0;
}

// The optimiser on an analytic objective (no device): f(x) = sum_i w_i (x_i - c_i)^2 +
// 0.05 sum_i x_i x_{i+1} + 0.01 sum_i x_i^4, some minimisers outside the box [1e-4, 6].
class QuadModel : public Opt_Algs {
 public:
  mutable std::vector<double> x{1.5, 0.9, 2.5, 0.3, 4.0, 1.1};
  std::vector<double> w{1.0, 3.0, 0.5, 2.0, 0.25, 1.5}, c{2.0, 7.0, -1.0, 0.5, 3.0, 5.5};
  explicit QuadModel(int variant) {
    if (variant == 1) { x = {0.5, 0.5, 0.5, 0.5, 0.5, 0.5}; w = {0.2, 0.4, 0.6, 0.8, 1.0, 1.2}; c = {1.0, 2.0, 3.0, 4.0, 5.0, 5.9}; }
    if (variant == 2) { x = {5.5, 0.01, 3.0, 3.0, 0.2, 2.2}; w = {2.0, 0.1, 1.0, 1.0, 3.0, 0.7}; c = {8.0, -2.0, 3.1, 2.9, 0.1, 2.0}; }
  }
  unsigned int getNumPars() const override { return 6; }
  void get_GP_Pars(mat &p) const override { for (int i = 0; i < 6; i++) p(i) = x[i]; }
  void set_GP_Pars(mat &p) const override { for (int i = 0; i < 6; i++) x[i] = p(i); }
  double eval(double *g) const {
    double f = 0;
    for (int i = 0; i < 6; i++) {
      f += w[i] * (x[i] - c[i]) * (x[i] - c[i]) + 0.01 * x[i] * x[i] * x[i] * x[i];
      if (g) g[i] = 2 * w[i] * (x[i] - c[i]) + 0.04 * x[i] * x[i] * x[i];
    }
    for (int i = 0; i < 5; i++) {
      f += 0.05 * x[i] * x[i + 1];
      if (g) { g[i] += 0.05 * x[i + 1]; g[i + 1] += 0.05 * x[i]; }
    }
    return f;
  }
  double Grad_Values(mat &g) const override { double gg[6]; double f = eval(gg); for (int i = 0; i < 6; i++) g(i) = gg[i]; return f; }
  double ObjVal() const override { return eval(nullptr); }
};

static int opt_only(int maxit, int variant) {
  QuadModel m(variant);
  m.setVerbose(1);
  m.setMaxIters(maxit);
  std::cout.precision(17);
  m.LBFGSOptimise();
  printf("FINAL");
  for (double v : m.x) printf(" %.17g", v);
  printf("\nNFEV %u\n", m.numFuncEval);
  return 0;
}

int main(int argc, char **argv) {
  if (argc >= 3 && std::string(argv[1]) == "--logic") return logic_only(argv[2]);
  if (argc >= 3 && std::string(argv[1]) == "--opt") return opt_only(atoi(argv[2]), argc >= 4 ? atoi(argv[3]) : 0);
  if (argc < 3) { fprintf(stderr, "usage: host_selftest train.csv test.csv\n"); return 2; }
  char *cargv[] = {argv[0], nullptr};
  Control io(1, cargv);
  int ds[2];
  mat X, y, Xt, yt;
  io.readDataSize(argv[1], ds);
  io.readDataFile(X, y, ds, argv[1]);
  io.readDataSize(argv[2], ds);
  io.readDataFile(Xt, yt, ds, argv[2]);

  HybKerns Kerns(X);
  Kern_ExpAnisotropic ea(X);
  Kern_Bias kb(X);
  Kerns.addNewKernel(&ea);
  Kerns.addNewKernel(&kb);
  GP_utils gp(&Kerns, X, y, GP_utils::inf_laplace, GP_utils::likeL_Gaussian, GP_utils::mean_zero, 8, 1, 0, 0);

  printf("{\n\"npars\": %u,\n", gp.getNumPars());
  mat p(1, gp.getNumPars());
  gp.get_GP_Pars(p);
  print_vec("params", p);
  printf("\"param_names\": [");
  for (unsigned i = 0; i < Kerns.getNPars(); i++) printf("%s\"%s\"", i ? ", " : "", Kerns.getParamName(i).c_str());
  printf("],\n");
  // Kernels::computeK on host matrices (Kernel.h:54)
  mat K, D2;
  Kerns.computeK(X, X, K, D2);
  double ksum = 0;
  for (size_t i = 0; i < K.n_elem; i++) ksum += K[i];
  printf("\"K_sum\": %.17g,\n\"K_00\": %.17g,\n", ksum, K(0, 0));
  mat kd(X.n_rows, 1);
  Kerns.diag_Compute(kd, X);
  printf("\"kdiag0\": %.17g,\n", kd(0));
  printf("\"nlz\": %.17g,\n", gp.logLikelihood());
  mat g(1, gp.getNumPars());
  double f = gp.GradLL(g);
  printf("\"nlz_from_grad\": %.17g,\n", f);
  print_vec("grad", g);
  mat mu(Xt.n_rows, 1), var(Xt.n_rows, 1);
  gp.Calc_Out(mu, var, Xt);
  print_vec("mean", mu);
  print_vec("var", var);
  // set_GP_Pars invalidates and a changed sn2 changes the value
  p(9) = 0.05;
  gp.set_GP_Pars(p);
  printf("\"nlz_sn2_005\": %.17g,\n", gp.logLikelihood());
  p(9) = -0.5;  // Chol_fail -> NaN
  gp.set_GP_Pars(p);
  double bad = gp.logLikelihood();
  printf("\"chol_fail_is_nan\": %s,\n", bad != bad ? "true" : "false");
  // f-4: another composition through the same classes: ExpAns + RBF + Exp + Bias + White
  {
    HybKerns H2(X);
    Kern_RBF rbf(X); Kern_Exponential ex(X); Kern_White wh(X);
    H2.addNewKernel(&ea); H2.addNewKernel(&rbf); H2.addNewKernel(&ex); H2.addNewKernel(&kb); H2.addNewKernel(&wh);
    mat K2, D22;
    H2.computeK(X, X, K2, D22);
    double s2 = 0;
    for (size_t i = 0; i < K2.n_elem; i++) s2 += K2[i];
    GP_utils gp2(&H2, X, y, GP_utils::inf_laplace, GP_utils::likeL_Gaussian, GP_utils::mean_zero, 8, 1, 0, 0);
    printf("\"hyb5_npars\": %u,\n\"hyb5_K_sum\": %.17g,\n\"hyb5_K00\": %.17g,\n\"hyb5_nlz\": %.17g,\n", gp2.getNumPars(), s2,
           K2(0, 0), gp2.logLikelihood());
    // gradient of a composition in CHILD order: Bias first, then RBF, then ExpAns
    HybKerns H3(X);
    H3.addNewKernel(&kb); H3.addNewKernel(&rbf); H3.addNewKernel(&ea);
    GP_utils gp3(&H3, X, y, GP_utils::inf_laplace, GP_utils::likeL_Gaussian, GP_utils::mean_zero, 8, 1, 0, 0);
    mat g3(1, gp3.getNumPars());
    gp3.GradLL(g3);
    print_vec("hyb3_grad", g3);
  }
  // model file round trip
  p(9) = 0.016;
  gp.set_GP_Pars(p);
  writeGPFile(gp, "/tmp/gpak_selftest_model", "# GP_SS_AK Model File ");
  GP_utils *m2 = readGpFromFile("/tmp/gpak_selftest_model", 0);
  mat p2(1, m2->getNumPars());
  m2->get_GP_Pars(p2);
  print_vec("params_reloaded", p2, true);
  printf("}\n");
  delete m2;
  return 0;
}
