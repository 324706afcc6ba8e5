// kernels.hpp -- host-side mirror of the reference's kernel classes (Kernel.h:27-501) for the
// hot path: Kernels (abstract), Kern_ExpAnisotropic, Kern_Bias, HybKerns.  Same names, same
// parameter order/names/defaults, same text serialisation; the arithmetic of computeK /
// diag_Compute runs on the MI355X through the C-ABI (include/gpak.h).
#pragma once
#include <iostream>
#include <string>
#include <vector>

#include "gpak_mat.hpp"

using gpak_host::mat;

// key=value text streams (StreamInt.h:46-124)
class StreamIntfce {
 public:
  virtual ~StreamIntfce() {}
  virtual void StrmOut(std::ostream &out) const { ToFile_GP_Params(out); }
  virtual void StrmIn(std::istream &in) { FromFile_GP_Params(in); }
  static std::string ReadStrStrm(std::istream &in, const std::string &fieldName);
  static int ReadIntStrm(std::istream &in, const std::string &fieldName);
  static double ReadDoubleStrm(std::istream &in, const std::string &fieldName);
  virtual void ToFile_GP_Params(std::ostream &out) const = 0;
  virtual void FromFile_GP_Params(std::istream &in) = 0;
  void WFile(const std::string &fileName, const std::string &comment = "") const;
  void RFile(const std::string &fileName);
};

class Kernels : public StreamIntfce {
 public:
  Kernels() {}
  virtual ~Kernels() {}
  virtual Kernels *clone() const = 0;
  virtual void setInitPars() = 0;
  virtual double Diag_Kernel(const mat &X, unsigned int index) const = 0;
  virtual void diag_Compute(mat &d, const mat &X) const;
  virtual void setParam(double, unsigned int) = 0;
  virtual double getParam(unsigned int) const = 0;
  // K and D2 are caller-allocated n x m, like the reference (Kernel.h:54)
  virtual void computeK(const mat &X1, const mat &X2, mat &K, mat &D2) const = 0;
  virtual void getGradients(mat &g, const mat &X, const mat &X2, const mat &D2, const mat &QW) const;
  void GetGrads(mat &g, const mat &X, const mat &X2, const mat &D2, const mat &QW) const { getGradients(g, X, X2, D2, QW); }
  virtual unsigned int addNewKernel(const Kernels *) { std::cerr << "Error in adding new kernel." << std::endl; return 0; }
  void setParams(const mat &v) { for (unsigned i = 0; i < nParams; i++) setParam(v(i), i); }
  void getParams(mat &v) const { for (unsigned i = 0; i < nParams; i++) v(i) = getParam(i); }
  std::string getKerName() const { return kernName; }
  void setKerName(const std::string &n) { kernName = n; }
  void setInputDim(unsigned int d) { inputDim = d; }
  unsigned getInputDim() const { return inputDim; }
  unsigned int getNPars() const { return nParams; }
  void setParamName(const std::string &name, unsigned int index);
  virtual std::string getParamName(unsigned int index) const { return paramNames[index]; }
  void ToFile_GP_Params(std::ostream &out) const override;
  void FromFile_GP_Params(std::istream &in) override;
  virtual std::ostream &ShowKernelPars(std::ostream &os) const;

 protected:
  unsigned int nParams = 0;
  std::string kernName;
  std::vector<std::string> paramNames;

 private:
  unsigned int inputDim = 0;
};

// 8-parameter rotated-anisotropic exponential kernel (Kernel.h:418-501, Kernel.cpp:704-882)
class Kern_ExpAnisotropic : public Kernels {
 public:
  Kern_ExpAnisotropic() { _init(); }
  explicit Kern_ExpAnisotropic(unsigned int inDim) { _init(); setInputDim(inDim); }
  explicit Kern_ExpAnisotropic(const mat &X) { _init(); setInputDim((unsigned)X.n_cols); }
  Kern_ExpAnisotropic *clone() const override { return new Kern_ExpAnisotropic(*this); }
  void setInitPars() override;
  double Diag_Kernel(const mat &, unsigned int) const override { return Sigma_ExpAns * Sigma_ExpAns; }
  void diag_Compute(mat &d, const mat &X) const override;
  void setParam(double val, unsigned int paramNo) override;
  double getParam(unsigned int paramNo) const override;
  void computeK(const mat &X1, const mat &X2, mat &K, mat &D2) const override;

 private:
  void _init();
  double AngleX_ExpAns, inverseWidthx_ExpAns, AngleY_ExpAns, inverseWidthy_ExpAns, AngleZ_ExpAns,
      inverseWidthz_ExpAns, Sigma_ExpAns, InversewidthR_ExpAns;
};

// constant kernel (Kernel.h:285-314, Kernel.cpp:283-377); Sigma_Bias is NOT squared
class Kern_Bias : public Kernels {
 public:
  Kern_Bias() { _init(); }
  explicit Kern_Bias(unsigned int inDim) { _init(); setInputDim(inDim); }
  explicit Kern_Bias(const mat &X) { _init(); setInputDim((unsigned)X.n_cols); }
  Kern_Bias *clone() const override { return new Kern_Bias(*this); }
  void setInitPars() override { Sigma_Bias = 0.2; }
  double Diag_Kernel(const mat &, unsigned int) const override { return Sigma_Bias; }
  void setParam(double val, unsigned int paramNo) override;
  double getParam(unsigned int paramNo) const override;
  void computeK(const mat &X1, const mat &X2, mat &K, mat &D2) const override;

 private:
  void _init();
  double Sigma_Bias;
};

// isotropic exponential kernel over EuclDist (Kernel.cpp:550-700, 1343-1368): {Hayper_Euc_Exp, Sigma_Exp}
class Kern_Exponential : public Kernels {
 public:
  Kern_Exponential() { _init(); }
  explicit Kern_Exponential(unsigned int inDim) { _init(); setInputDim(inDim); }
  explicit Kern_Exponential(const mat &X) { _init(); setInputDim((unsigned)X.n_cols); }
  Kern_Exponential *clone() const override { return new Kern_Exponential(*this); }
  void setInitPars() override { Hayper_Euc_Exp = 0.5; Sigma_Exp = 0.9; }   // Kernel.cpp:586-590
  double Diag_Kernel(const mat &, unsigned int) const override { return Sigma_Exp * Sigma_Exp; }
  void setParam(double val, unsigned int paramNo) override;
  double getParam(unsigned int paramNo) const override;
  void computeK(const mat &X1, const mat &X2, mat &K, mat &D2) const override;

 private:
  void _init();
  double Hayper_Euc_Exp, Sigma_Exp;
};

// isotropic squared-exponential kernel over EuclDist (Kernel.cpp:384-548):
// {Hayper_Euc_RBF, inverseWidth_RBF, Sigma_RBF}
class Kern_RBF : public Kernels {
 public:
  Kern_RBF() { _init(); }
  explicit Kern_RBF(unsigned int inDim) { _init(); setInputDim(inDim); }
  explicit Kern_RBF(const mat &X) { _init(); setInputDim((unsigned)X.n_cols); }
  Kern_RBF *clone() const override { return new Kern_RBF(*this); }
  void setInitPars() override { Hayper_Euc_RBF = 0.5; inverseWidth_RBF = 0.9; Sigma_RBF = 0.5; }  // Kernel.cpp:424-429
  double Diag_Kernel(const mat &, unsigned int) const override { return Sigma_RBF * Sigma_RBF; }
  void setParam(double val, unsigned int paramNo) override;
  double getParam(unsigned int paramNo) const override;
  void computeK(const mat &X1, const mat &X2, mat &K, mat &D2) const override;

 private:
  void _init();
  double Hayper_Euc_RBF, inverseWidth_RBF, Sigma_RBF;
};

// white noise on the diagonal of a set with itself (Kernel.cpp:180-270); Sigma_White is NOT squared
class Kern_White : public Kernels {
 public:
  Kern_White() { _init(); }
  explicit Kern_White(unsigned int inDim) { _init(); setInputDim(inDim); }
  explicit Kern_White(const mat &X) { _init(); setInputDim((unsigned)X.n_cols); }
  Kern_White *clone() const override { return new Kern_White(*this); }
  void setInitPars() override { Sigma_White = 0.10; }                        // Kernel.cpp:214-217
  double Diag_Kernel(const mat &, unsigned int) const override { return Sigma_White; }
  void setParam(double val, unsigned int paramNo) override;
  double getParam(unsigned int paramNo) const override;
  void computeK(const mat &X1, const mat &X2, mat &K, mat &D2) const override;

 private:
  void _init();
  double Sigma_White;
};

// additive container with flat parameter indexing (Kernel.h:158-253, Kernel.cpp:55-169)
class HybKerns : public Kernels {
 public:
  HybKerns() { _init(); }
  explicit HybKerns(unsigned int inDim) { _init(); setInputDim(inDim); }
  explicit HybKerns(const mat &X) { _init(); setInputDim((unsigned)X.n_cols); }
  HybKerns(const HybKerns &o);
  ~HybKerns();
  HybKerns *clone() const override { return new HybKerns(*this); }
  void setInitPars() override {}
  unsigned int addNewKernel(const Kernels *kern) override;
  unsigned int getNumKerns() const { return (unsigned)MainKEl.size(); }
  const Kernels *getKern(unsigned i) const { return MainKEl[i]; }
  void setParam(double val, unsigned int paramNo) override;
  double getParam(unsigned int paramNo) const override;
  std::string getParamName(unsigned int paramNo) const override;
  double Diag_Kernel(const mat &X, unsigned int index) const override;
  void diag_Compute(mat &d, const mat &X) const override;
  void computeK(const mat &X1, const mat &X2, mat &K, mat &D2) const override;
  void ToFile_GP_Params(std::ostream &out) const override;
  void FromFile_GP_Params(std::istream &in) override;
  std::ostream &ShowKernelPars(std::ostream &os) const override;

 private:
  void _init();
  std::vector<Kernels *> MainKEl;
};

Kernels *ReadKerFromFile(std::istream &in);

// What the hot path needs from a kernel object: the {ExpAns, Bias} parameters.  Returns false
// when the kernel is not an ExpAns(+Bias) composition (those run on the reference's CPU path).
bool gpak_extract_expans_bias(const Kernels *k, double expans[8], double *bias);
// General form: the stationary children (kinds GPAK_KERN_*, parameters concatenated), Bias and White.
bool gpak_extract_composition(const Kernels *k, std::vector<int> &kinds, std::vector<double> &pars, double *bias,
                              double *white);

// process-wide scratch device context for kernels evaluated outside a GP_utils
struct gpak_ctx;
gpak_ctx *gpak_host_scratch_ctx();
void gpak_host_fatal(const std::string &what, gpak_ctx *ctx);
int gpak_host_dist_mode();
