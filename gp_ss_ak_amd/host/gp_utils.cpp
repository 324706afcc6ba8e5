// gp_utils.cpp -- see gp_utils.hpp.
#include "gp_utils.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <vector>

#include "../../include/gpak.h"

int GP_utils::default_precision = GPAK_F64;
int GP_utils::default_gpus = 1;

void GP_utils::create_ctx() {
  int rc;
  if (default_gpus > 1) rc = gpak_create_multi(&ctx, default_gpus, nullptr, default_precision);
  else rc = gpak_create(&ctx, getenv("GPAK_DEVICE") ? atoi(getenv("GPAK_DEVICE")) : 0, default_precision);
  if (rc != GPAK_OK) gpak_host_fatal(default_gpus > 1 ? "gpak_create_multi" : "gpak_create", nullptr);
}

std::string GP_utils::timingJson() const {
  gpak_phase_times t;
  if (gpak_timing(ctx, &t) != GPAK_OK) return "{}";
  char buf[1024];
  snprintf(buf, sizeof buf,
           "{\"n\": %d, \"n_padded\": %d, \"evaluations\": %d, \"last\": {\"gram_ms\": %.6g, \"factor_ms\": %.6g, "
           "\"solve_ms\": %.6g, \"nlz_ms\": %.6g, \"kmatvec_ms\": %.6g, \"predict_ms\": %.6g, \"grad_ms\": %.6g}, "
           "\"accumulated\": {\"gram_ms\": %.6g, \"factor_ms\": %.6g, \"solve_ms\": %.6g, \"nlz_ms\": %.6g}, "
           "\"precision\": \"%s\", \"gpus\": %d}",
           t.n, t.n_padded, t.evaluations, t.gram_ms, t.factor_ms, t.solve_ms, t.nlz_ms, t.kmatvec_ms, t.predict_ms,
           t.grad_ms, t.accumulated_ms[0], t.accumulated_ms[1], t.accumulated_ms[2], t.accumulated_ms[3],
           default_precision == GPAK_F32 ? "f32" : "f64", default_gpus);
  return buf;
}

GP_utils::GP_utils() { create_ctx(); }

GP_utils::GP_utils(Kernels *kernel, mat Xin, mat Yin, int, int likeLtype, int, unsigned int numhyper,
                   unsigned int numlik_par, unsigned int numMF_par, int verbos)
    : Xinp(Xin), yTarg(Yin), KerenlW(kernel) {
  create_ctx();
  setNumMFpar(numMF_par);
  setNumlikfpar(numlik_par);
  setNumCovpar(numhyper);
  setLikelihoodType(likeLtype);
  setVerbose(verbos);
  setOutDim((unsigned)yTarg.n_cols);
  setInpDim((unsigned)Xin.n_cols);
  setNumData((unsigned)yTarg.n_rows);
  if (numlik_par > 0) {
    hyperlf.resize(numlik_par, 1);
    for (unsigned i = 0; i < numlik_par; i++) hyperlf(i) = 0.016;  // GP_Utils.cpp:43
  }
  initialize_vars();
}

GP_utils::~GP_utils() {
  if (ctx) gpak_destroy(ctx);
  if (owns_kernel) delete KerenlW;
}

void GP_utils::initialize_vars() {
  if (Xinp.n_rows == 0) return;
  if (gpak_set_train(ctx, Xinp.memptr(), yTarg.memptr(), (int)Xinp.n_rows, (int)Xinp.n_cols) != GPAK_OK)
    gpak_host_fatal("gpak_set_train", ctx);
  dirty = true;
}

unsigned int GP_utils::getNumPars() const { return KerenlW->getNPars() + getNumMFpar() + getNumlikfpar(); }

void GP_utils::get_GP_Pars(mat &param) const {  // GP_Utils.cpp:101-128
  unsigned c = 0;
  for (unsigned i = 0; i < KerenlW->getNPars(); i++) param(c++) = KerenlW->getParam(i);
  for (unsigned i = 0; i < getNumlikfpar(); i++) param(c++) = hyperlf(i);
}

void GP_utils::set_GP_Pars(mat &param) const {  // GP_Utils.cpp:130-157: always invalidates
  dirty = true;
  unsigned c = 0;
  for (unsigned i = 0; i < KerenlW->getNPars(); i++) KerenlW->setParam(param(c++), i);
  for (unsigned i = 0; i < getNumlikfpar(); i++) hyperlf(i) = param(c++);
}

void GP_utils::sync_params() const {
  if (!dirty) return;
  double e[8], bias, white;
  if (gpak_extract_expans_bias(KerenlW, e, &bias)) {
    if (gpak_set_params(ctx, e, bias, hyperlf(0), gpak_host_dist_mode()) != GPAK_OK) gpak_host_fatal("gpak_set_params", ctx);
  } else {
    std::vector<int> kinds;
    std::vector<double> pars;
    if (!gpak_extract_composition(KerenlW, kinds, pars, &bias, &white) || kinds.empty()) {
      std::cerr << "GP_utils: this kernel composition is not on the HIP path." << std::endl;
      exit(1);
    }
    if (gpak_set_kernel(ctx, (int)kinds.size(), kinds.data(), pars.data(), bias, white, hyperlf(0), gpak_host_dist_mode()) != GPAK_OK)
      gpak_host_fatal("gpak_set_kernel", ctx);
  }
  dirty = false;
}

void GP_utils::updateKernel() const { sync_params(); }

double GP_utils::logLikelihood() const {
  sync_params();
  double nlz = std::numeric_limits<double>::quiet_NaN();
  int rc = gpak_nlz(ctx, &nlz);
  Chol_fail = (rc == GPAK_ENOTPD);
  if (rc != GPAK_OK && rc != GPAK_ENOTPD) gpak_host_fatal("gpak_nlz", ctx);
  return nlz;
}

double GP_utils::GradLL(mat &g) const {  // GP_Utils.cpp:1171-1262
  double nlz = logLikelihood();
  if (Chol_fail) return std::numeric_limits<double>::quiet_NaN();
  // one device call evaluates GradLL + every child's getGradients; the result comes back as the
  // stationary children's blocks in order, then the Bias entry, then the likelihood entry
  std::vector<const Kernels *> leaves;
  if (const HybKerns *h = dynamic_cast<const HybKerns *>(KerenlW)) {
    for (unsigned i = 0; i < h->getNumKerns(); i++) leaves.push_back(h->getKern(i));
  } else {
    leaves.push_back(KerenlW);
  }
  int ng = 2;
  for (const Kernels *c : leaves) {
    if (dynamic_cast<const Kern_Bias *>(c)) continue;
    if (dynamic_cast<const Kern_White *>(c)) {
      // upstream, Kern_White inherits a getGradients that calls itself (Kernel.h:56-60, 257-283)
      std::cerr << "GradLL: a White child has no gradient (the reference recurses without end here)." << std::endl;
      exit(1);
    }
    ng += (int)c->getNPars();
  }
  std::vector<double> gd(ng);
  if (gpak_grad_hyb(ctx, gd.data(), ng) != GPAK_OK) gpak_host_fatal("gpak_grad_hyb", ctx);
  unsigned out = 0;
  int in = 0;
  for (const Kernels *c : leaves) {
    if (dynamic_cast<const Kern_Bias *>(c)) { g(out++) = gd[ng - 2]; continue; }
    for (unsigned i = 0; i < c->getNPars(); i++) g(out++) = gd[in++];
  }
  g(out) = gd[ng - 1];
  return nlz;
}

void GP_utils::posteriorMeanVar(mat &mu, mat &varSigma, const mat &Xin) const {
  logLikelihood();
  if (Chol_fail) { mu.fill(std::numeric_limits<double>::quiet_NaN()); varSigma.fill(std::numeric_limits<double>::quiet_NaN()); return; }
  if (mu.n_elem != Xin.n_rows) mu.resize(Xin.n_rows, 1);
  if (varSigma.n_elem != Xin.n_rows) varSigma.resize(Xin.n_rows, 1);
  if (gpak_predict(ctx, Xin.memptr(), (long)Xin.n_rows, (int)Xin.n_cols, mu.memptr(), varSigma.memptr(), compat) != GPAK_OK)
    gpak_host_fatal("gpak_predict", ctx);
}

void GP_utils::Calc_Out(mat &yPred, mat &yVar, const mat &Xin) const { posteriorMeanVar(yPred, yVar, Xin); }

void GP_utils::OptimisePars(unsigned int iters) {  // GP_Utils.cpp:1288-1301
  if (getVerbose() > 2) { std::cout << "Initial model:" << std::endl; ShowKernelPars(std::cout); }
  // the reference only honours `iters` at verbosity > 2 (the `if` swallowed the call, :1295-1296)
  if (getVerbose() > 2 && getNumPars() < 40) setMaxIters(iters);
  if (getenv("GPAK_MAX_ITERS")) setMaxIters((unsigned)atoi(getenv("GPAK_MAX_ITERS")));
  Optimise();
  if (getVerbose() > 0) ShowKernelPars(std::cout);
}

std::ostream &GP_utils::ShowKernelPars(std::ostream &os) const {  // GP_Utils.cpp:1303-1322
  std::cout << "Standard GP Model: " << std::endl;
  std::cout << "Optimiser: " << getDefaultOptimiserStr() << std::endl;
  std::cout << "Inference: " << getInf() << std::endl;
  std::cout << "Data Set Size: " << getNumData() << std::endl;
  std::cout << "Kernel Type: " << std::endl;
  KerenlW->ShowKernelPars(os);
  for (unsigned i = 0; i < getNumlikfpar(); i++) std::cout << "likelihood hyperparmeters : " << hyperlf(i) << std::endl;
  if (getVerbose()) std::cout << "Log likelihood: " << logLikelihood() << std::endl;
  return os;
}

void GP_utils::ToFile_GP_Params(std::ostream &out) const {  // GP_Utils.cpp:1360-1390
  out << "Inference=" << getInf() << std::endl;
  out << "likelihood=" << getLikelihoodType() << std::endl;
  out << "MeanFunction=" << getMean() << std::endl;
  out << "numData=" << getNumData() << std::endl;
  out << "outputDim=" << getOutDim() << std::endl;
  out << "inputDim=" << getInpDim() << std::endl;
  out << "NumHyperKernel=" << KerenlW->getNPars() << std::endl;
  out << "NumHyperLik=" << getNumlikfpar() << std::endl;
  out << "NumHyperMean=" << getNumMFpar() << std::endl;
  KerenlW->StrmOut(out);
  for (unsigned i = 0; i < getNumlikfpar(); i++) out << "Hyperparams_likelihood=" << hyperlf(i) << std::endl;
}

void GP_utils::FromFile_GP_Params(std::istream &in) {  // GP_Utils.cpp:1324-1358
  setInf(ReadStrStrm(in, "Inference"));
  setLikelihoodType(ReadIntStrm(in, "likelihood"));
  setMean(ReadStrStrm(in, "MeanFunction"));
  setNumData(ReadIntStrm(in, "numData"));
  setOutDim(ReadIntStrm(in, "outputDim"));
  setInpDim(ReadIntStrm(in, "inputDim"));
  setNumCovpar(ReadIntStrm(in, "NumHyperKernel"));
  setNumlikfpar(ReadIntStrm(in, "NumHyperLik"));
  setNumMFpar(ReadIntStrm(in, "NumHyperMean"));
  KerenlW = ReadKerFromFile(in);
  owns_kernel = true;
  if (getNumlikfpar() > 0) {
    hyperlf.resize(getNumlikfpar(), 1);
    for (unsigned i = 0; i < getNumlikfpar(); i++) hyperlf(i) = ReadDoubleStrm(in, "Hyperparams_likelihood");
  }
  dirty = true;
}

void writeGPFile(const GP_utils &model, const std::string &modelFileName, const std::string &comment) {
  model.WFile(modelFileName, comment);
}

GP_utils *readGpFromFile(const std::string &modelFileName, int verbosity) {
  if (verbosity > 0) std::cout << "Loading model file." << std::endl;
  std::ifstream in(modelFileName.c_str());
  if (!in.is_open()) { std::cout << "Error in reading file name. \n"; exit(1); }
  GP_utils *m = new GP_utils();
  // skip the comment line, then the key=value body
  m->StrmIn(in);
  if (verbosity > 0) std::cout << "Model Info has been read.\n";
  m->setVerbose(verbosity);
  return m;
}

// ---------------------------------------------------------------------------------------------
// A plain projected L-BFGS with backtracking (NOT the reference's algorithm; opt_algs.cpp holds
// the restatement of Opt_pars.cpp).  Kept as an alternative driver: GPAK_OPT=simple.
// ---------------------------------------------------------------------------------------------
void Opt_Algs::SimpleLBFGSOptimise() {
  const unsigned n = getNumPars();
  const double lb = 1e-4, ub = 6.0;  // Opt_pars.cpp:184-188
  const unsigned mem = 6;
  mat x(1, n), g(1, n), xn(1, n), gn(1, n);
  get_GP_Pars(x);
  for (unsigned i = 0; i < n; i++) x(i) = std::min(ub, std::max(lb, x(i)));
  set_GP_Pars(x);
  double fx = Grad_Values(g);
  numFuncEval++;
  std::vector<std::vector<double>> S, Y;
  std::vector<double> rho;
  for (unsigned iter = 1; iter <= getMaxIters(); iter++) {
    // two-loop recursion on the free variables
    std::vector<double> d(n), q(n), al(S.size());
    for (unsigned i = 0; i < n; i++) q[i] = g(i);
    for (int k = (int)S.size() - 1; k >= 0; k--) {
      double a = 0; for (unsigned i = 0; i < n; i++) a += S[k][i] * q[i];
      al[k] = a * rho[k];
      for (unsigned i = 0; i < n; i++) q[i] -= al[k] * Y[k][i];
    }
    double gam = 1.0;
    if (!S.empty()) {
      double sy = 0, yy = 0;
      for (unsigned i = 0; i < n; i++) { sy += S.back()[i] * Y.back()[i]; yy += Y.back()[i] * Y.back()[i]; }
      if (yy > 0) gam = sy / yy;
    } else {
      double gn2 = 0; for (unsigned i = 0; i < n; i++) gn2 += g(i) * g(i);
      gam = gn2 > 0 ? 0.1 / std::sqrt(gn2) : 1.0;
    }
    for (unsigned i = 0; i < n; i++) q[i] *= gam;
    for (size_t k = 0; k < S.size(); k++) {
      double b = 0; for (unsigned i = 0; i < n; i++) b += Y[k][i] * q[i];
      b *= rho[k];
      for (unsigned i = 0; i < n; i++) q[i] += S[k][i] * (al[k] - b);
    }
    for (unsigned i = 0; i < n; i++) {
      d[i] = -q[i];
      if ((x(i) <= lb && d[i] < 0) || (x(i) >= ub && d[i] > 0)) d[i] = 0;  // active bounds
    }
    // backtracking on the objective; a step is kept only if it decreases nlZ (Opt_pars.cpp:268)
    double step = 1.0, fnew = fx;
    bool ok = false;
    for (int ls = 0; ls < 12; ls++) {
      for (unsigned i = 0; i < n; i++) xn(i) = std::min(ub, std::max(lb, x(i) + step * d[i]));
      set_GP_Pars(xn);
      fnew = Grad_Values(gn);
      numFuncEval++;
      if (fnew == fnew && fnew < fx) { ok = true; break; }
      step *= 0.5;
    }
    if (!ok) {
      set_GP_Pars(x);
      if (S.empty()) break;      // steepest-descent step failed too: stop
      S.clear(); Y.clear(); rho.clear();  // drop the curvature pairs and retry along -g
      if (getVerbose() > 0) std::cout << "Iteration: " << iter << " -logL: " << fx << std::endl;
      continue;
    }
    std::vector<double> s(n), yv(n);
    double sy = 0;
    for (unsigned i = 0; i < n; i++) { s[i] = xn(i) - x(i); yv[i] = gn(i) - g(i); sy += s[i] * yv[i]; }
    if (sy > 1e-12) {
      if (S.size() == mem) { S.erase(S.begin()); Y.erase(Y.begin()); rho.erase(rho.begin()); }
      S.push_back(s); Y.push_back(yv); rho.push_back(1.0 / sy);
    }
    double dx = 0; for (unsigned i = 0; i < n; i++) dx = std::max(dx, std::fabs(s[i]));
    x = xn; g = gn;
    double df = fx - fnew;
    fx = fnew;
    if (getVerbose() > 0) std::cout << "Iteration: " << iter << " -logL: " << fx << std::endl;
    if (dx < 1e-7 || df < 1e-9 * std::fabs(fx)) break;
  }
  set_GP_Pars(x);
}
