// gp_utils.hpp -- host-side mirror of the reference's model classes on the hot path:
// ModelInfo / Modeling (ModelInf.h:22-179), Opt_Algs (Opt_pars.h:21-315, callbacks only) and
// GP_utils (GP_Utils.h:23-389).  Same member names and call sequence; every O(N^2)/O(N^3)
// member forwards to the C-ABI (include/gpak.h) instead of Armadillo.
#pragma once
#include <cstdlib>
#include <limits>
#include <string>

#include "kernels.hpp"

struct gpak_ctx;

class ModelInfo {
 public:
  virtual ~ModelInfo() {}
  virtual unsigned int getNumData() const { return numData; }
  virtual void setNumData(unsigned int v) { numData = v; }

 private:
  unsigned int numData = 0;
};

class Modeling : public ModelInfo {
 public:
  void setInpDim(unsigned int d) { inputDim = d; }
  unsigned int getInpDim() const { return inputDim; }
  void setOutDim(unsigned int d) { outputDim = d; }
  unsigned int getOutDim() const { return outputDim; }
  void setNumMFpar(unsigned int v) { NumMF = v; }
  unsigned int getNumMFpar() const { return NumMF; }
  void setNumlikfpar(unsigned int v) { Numlikf = v; }
  unsigned int getNumlikfpar() const { return Numlikf; }
  void setNumCovpar(unsigned int v) { NumCov = v; }
  unsigned int getNumCovpar() const { return NumCov; }
  virtual void Calc_Out(mat &yPred, mat &yVar, const mat &X) const = 0;
  virtual unsigned int getNumPars() const = 0;

 private:
  unsigned int outputDim = 1, inputDim = 0, NumMF = 0, Numlikf = 0, NumCov = 0;
};

// the optimiser's view of a model (Opt_pars.h:51-55) and the driver loop.
class Opt_Algs {
 public:
  enum { SCG, BFGS, LBFGS };
  virtual ~Opt_Algs() {}
  virtual unsigned int getNumPars() const = 0;
  virtual void get_GP_Pars(mat &param) const = 0;
  virtual void set_GP_Pars(mat &param) const = 0;
  virtual double Grad_Values(mat &g) const = 0;
  virtual double ObjVal() const = 0;
  void setVerbose(int v) const { verbose = v; }
  int getVerbose() const { return verbose; }
  void setOptimiser(int v) { defaultOptimiser = v; }
  int getOptimiser() const { return defaultOptimiser; }
  std::string getDefaultOptimiserStr() const { return defaultOptimiser == SCG ? "SCG" : defaultOptimiser == BFGS ? "BFGS" : "LBFGS"; }
  void setMaxIters(unsigned int v) { maxIters = v; }
  unsigned int getMaxIters() const { return maxIters; }
  // Opt_pars.cpp:179-332 with cauchy_point (:11-105), Primal_Conjugate_grad (:108-174) and
  // Efficient_line_search (:543-974), restated as written in opt_algs.cpp: box [1e-4, 6] on every
  // parameter, memory 6, progress lines "Iteration: k -logL: v".
  void LBFGSOptimise();
  // plain projected L-BFGS with backtracking (not the reference's trajectory); GPAK_OPT=simple
  void SimpleLBFGSOptimise();
  void Optimise() {
    const char *e = getenv("GPAK_OPT");
    if (e && std::string(e) == "simple") SimpleLBFGSOptimise();
    else LBFGSOptimise();
  }
  unsigned int numFuncEval = 0;

 private:
  mutable int verbose = 0;
  int defaultOptimiser = LBFGS;
  unsigned int maxIters = 100;
};

class GP_utils : public Modeling, public Opt_Algs, public StreamIntfce {
 public:
  enum likelihoodTypeE { likeL_Gaussian, likeL_WarpGauss };
  enum InferenceTypeE { inf_laplace, inf_EP };
  enum MeanTypeE { mean_zero, mean_sum };

  GP_utils();
  GP_utils(Kernels *kernel, mat Xin, mat Yin, int Inf_type = inf_laplace, int likeLtype = likeL_Gaussian,
           int mean_type = mean_zero, unsigned int numhyper = 1, unsigned int numlik_par = 1,
           unsigned int numMF_par = 0, int verbos = 2);
  ~GP_utils();

  // uploads Xinp / yTarg to the device (the reference allocates its N x N members here,
  // GP_Utils.cpp:60-86; none of them exists on the host any more)
  void initialize_vars();

  // Modeling
  void Calc_Out(mat &yPred, mat &yVar, const mat &Xin) const override;
  void posteriorMeanVar(mat &mu, mat &varSigma, const mat &Xin) const;

  // Opt_Algs callbacks (Opt_pars.h:236-253)
  unsigned int getNumPars() const override;
  void get_GP_Pars(mat &param) const override;
  void set_GP_Pars(mat &param) const override;
  double Grad_Values(mat &g) const override { return GradLL(g); }
  double ObjVal() const override { return logLikelihood(); }

  double logLikelihood() const;   // NaN on Chol_fail (GP_Utils.cpp:1145-1158)
  double GradLL(mat &g) const;    // g is 1 x getNumPars()
  void OptimisePars(unsigned int iters);
  void updateKernel() const;
  std::ostream &ShowKernelPars(std::ostream &os) const;

  double getHyperlfVal(unsigned int i) const { return hyperlf(i); }
  void setHyperlfVal(double v, unsigned int i) const { hyperlf(i) = v; dirty = true; }
  void setInf(const std::string &s) { InfS = s; }
  std::string getInf() const { return InfS; }
  void setMean(const std::string &s) { MeanS = s; }
  std::string getMean() const { return MeanS; }
  void setLikelihoodType(int v) { likelihoodType = v; }
  int getLikelihoodType() const { return likelihoodType; }
  void setCompatFlags(int f) { compat = f; }
  // device-side options of every GP_utils constructed afterwards (the CLI's --precision / --gpus)
  static void setDeviceOptions(int precision, int gpus) { default_precision = precision; default_gpus = gpus; }
  // gpak_phase_times of this model's context as one JSON object (the CLI's --timing)
  std::string timingJson() const;
  bool Chol_failed() const { return Chol_fail; }

  void ToFile_GP_Params(std::ostream &out) const override;
  void FromFile_GP_Params(std::istream &in) override;

  mat Xinp;
  mat yTarg;
  Kernels *KerenlW = nullptr;
  mutable mat hyperlf;

 private:
  void sync_params() const;
  gpak_ctx *ctx = nullptr;
  static int default_precision, default_gpus;
  void create_ctx();
  bool owns_kernel = false;
  mutable bool dirty = true;     // KUpdateStat / AlphaUpStatus (GP_Utils.h:257-279)
  mutable bool Chol_fail = false;
  int likelihoodType = likeL_Gaussian;
  int compat = 3;                // reproduce Q3/Q4 of SURVEY.md 8(c) like the reference
  std::string InfS = "Lapalce", MeanS = "Zero";
};

void writeGPFile(const GP_utils &model, const std::string &modelFileName, const std::string &comment = "");
GP_utils *readGpFromFile(const std::string &modelFileName, int verbosity = 2);
