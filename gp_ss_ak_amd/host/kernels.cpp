// kernels.cpp -- see kernels.hpp.  Parameter tables: Kernel.cpp:737-773 (ExpAns), :305-320 (Bias).
#include "kernels.hpp"

#include <cstdlib>
#include <cstring>
#include <fstream>

#include "../../include/gpak.h"

// ---------------------------------------------------------------------------------------------
// device access
// ---------------------------------------------------------------------------------------------
void gpak_host_fatal(const std::string &what, gpak_ctx *ctx) {
  // the reference's fatal errors print and exit(1) (ModelInf.h:84-88, Control.cpp:331-337)
  std::cerr << what << ": " << (ctx ? gpak_last_error(ctx) : gpak_global_error()) << std::endl;
  exit(1);
}
int gpak_host_dist_mode() {
  const char *e = getenv("GPAK_DIST_MODE");
  if (e && std::string(e) == "expansion") return GPAK_DIST_EXPANSION;
  return GPAK_DIST_DIRECT;
}
gpak_ctx *gpak_host_scratch_ctx() {
  static gpak_ctx *ctx = nullptr;
  if (!ctx) {
    const char *d = getenv("GPAK_DEVICE");
    if (gpak_create(&ctx, d ? atoi(d) : 0, GPAK_F64) != GPAK_OK) gpak_host_fatal("gpak_create", nullptr);
  }
  return ctx;
}

// ---------------------------------------------------------------------------------------------
// StreamIntfce
// ---------------------------------------------------------------------------------------------
std::string StreamIntfce::ReadStrStrm(std::istream &in, const std::string &) {
  std::string line;
  std::getline(in, line);
  while (!line.empty() && line[0] == '#') std::getline(in, line);
  size_t pos = line.find("=");
  return line.substr(pos + 1);
}
int StreamIntfce::ReadIntStrm(std::istream &in, const std::string &f) { return (int)atol(ReadStrStrm(in, f).c_str()); }
double StreamIntfce::ReadDoubleStrm(std::istream &in, const std::string &f) { return atof(ReadStrStrm(in, f).c_str()); }
void StreamIntfce::WFile(const std::string &fileName, const std::string &comment) const {
  std::ofstream out(fileName.c_str());
  if (!out) { std::cout << "The file " << fileName << " is open.\n"; exit(1); }
  out << comment << std::endl;
  StrmOut(out);
}
void StreamIntfce::RFile(const std::string &fileName) {
  std::ifstream in(fileName.c_str());
  if (!in.is_open()) { std::cout << "The file could not be read. \n"; exit(1); }
  StrmIn(in);
}

// ---------------------------------------------------------------------------------------------
// Kernels base
// ---------------------------------------------------------------------------------------------
void Kernels::diag_Compute(mat &d, const mat &X) const {
  for (unsigned int i = 0; i < X.n_rows; i++) d(i) = Diag_Kernel(X, i);
}
void Kernels::setParamName(const std::string &name, unsigned int index) {
  if (paramNames.size() <= index) paramNames.resize(index + 1, "no name");
  paramNames[index] = name;
}
void Kernels::getGradients(mat &, const mat &, const mat &, const mat &, const mat &) const {
  // The per-kernel gradient of the reference takes the N x N matrix QW from the host
  // (Kernel.h:56-60).  On the HIP path QW never leaves the device: GP_utils::GradLL calls
  // gpak_grad, which evaluates GradLL + every child's getGradients in one fused pass.
  std::cerr << "Kernels::getGradients: use GP_utils::GradLL (gpak_grad) on the HIP path." << std::endl;
  exit(1);
}
// text form: Kernel.cpp:20-40 -- default ostream precision (6 significant digits), whole numbers
// printed as integers
void Kernels::ToFile_GP_Params(std::ostream &out) const {
  out << "KernelName=" << getKerName() << std::endl;
  out << "inputDim=" << getInputDim() << std::endl;
  out << "numParams=" << getNPars() << std::endl;
  for (unsigned int j = 0; j < getNPars(); j++) {
    double val = getParam(j);
    if ((val - (int)val) == 0.0) out << (int)val << " ";
    else out << val << " ";
  }
  out << std::endl;
}
void Kernels::FromFile_GP_Params(std::istream &in) {
  setInputDim(ReadIntStrm(in, "inputDim"));
  unsigned int nPars = ReadIntStrm(in, "numParams");
  std::string lineC;
  if (!std::getline(in, lineC)) { std::cout << "Can not read " << getKerName() << " kernel parameters. \n"; exit(1); }
  for (unsigned int i = 0; i < nPars; i++) {
    if (lineC.empty()) {
      std::cout << "The nember of Hyper-parameters of " << getKerName() << " are not sufficient. \n";
      exit(1);
    }
    size_t pos = lineC.find(" ");
    std::string val = lineC.substr(0, pos == std::string::npos ? lineC.size() : pos + 1);
    lineC.erase(0, pos == std::string::npos ? lineC.size() : pos + 1);
    setParam(atof(val.c_str()), i);
  }
}
std::ostream &Kernels::ShowKernelPars(std::ostream &os) const {
  os << getKerName() << " kernel:" << std::endl;
  for (unsigned int i = 0; i < nParams; i++) os << getParamName(i) << ": " << getParam(i) << std::endl;
  return os;
}

// ---------------------------------------------------------------------------------------------
// Kern_ExpAnisotropic
// ---------------------------------------------------------------------------------------------
void Kern_ExpAnisotropic::_init() {
  nParams = 8;
  setKerName("ExpAns");
  const char *names[8] = {"AngleX_ExpAns", "inverseWidthx_ExpAns", "AngleY_ExpAns", "inverseWidthy_ExpAns",
                          "AngleZ_ExpAns", "inverseWidthz_ExpAns", "Sigma_ExpAns", "InversewidthR_ExpAns"};
  for (unsigned i = 0; i < 8; i++) setParamName(names[i], i);
  setInitPars();
}
void Kern_ExpAnisotropic::setInitPars() {  // Kernel.cpp:763-773
  AngleX_ExpAns = M_PI / 3.1; inverseWidthx_ExpAns = 1.5;
  AngleY_ExpAns = M_PI / 3.1; inverseWidthy_ExpAns = 1.5;
  AngleZ_ExpAns = M_PI / 3.1; inverseWidthz_ExpAns = 1.3;
  Sigma_ExpAns = 0.9; InversewidthR_ExpAns = 0.6;
}
void Kern_ExpAnisotropic::diag_Compute(mat &d, const mat &) const { d.fill(Sigma_ExpAns * Sigma_ExpAns); }
void Kern_ExpAnisotropic::setParam(double val, unsigned int paramNo) {
  double *slots[8] = {&AngleX_ExpAns, &inverseWidthx_ExpAns, &AngleY_ExpAns, &inverseWidthy_ExpAns,
                      &AngleZ_ExpAns, &inverseWidthz_ExpAns, &Sigma_ExpAns, &InversewidthR_ExpAns};
  if (paramNo >= 8) { std::cout << "Requested parameter doesn't exist.\n"; exit(1); }
  *slots[paramNo] = val;
}
double Kern_ExpAnisotropic::getParam(unsigned int paramNo) const {
  const double vals[8] = {AngleX_ExpAns, inverseWidthx_ExpAns, AngleY_ExpAns, inverseWidthy_ExpAns,
                          AngleZ_ExpAns, inverseWidthz_ExpAns, Sigma_ExpAns, InversewidthR_ExpAns};
  if (paramNo >= 8) { std::cout << "Requested parameter doesn't exist.\n"; exit(1); }
  return vals[paramNo];
}

static void device_compute_k(const double e[8], double bias, const mat &X1, const mat &X2, mat &K, mat &D2) {
  gpak_ctx *ctx = gpak_host_scratch_ctx();
  if (gpak_set_params(ctx, e, bias, 1.0, gpak_host_dist_mode()) != GPAK_OK) gpak_host_fatal("gpak_set_params", ctx);
  if (K.n_rows != X1.n_rows || K.n_cols != X2.n_rows) K.resize(X1.n_rows, X2.n_rows);
  if (D2.n_rows != X1.n_rows || D2.n_cols != X2.n_rows) D2.resize(X1.n_rows, X2.n_rows);
  if (gpak_compute_k(ctx, X1.memptr(), (int)X1.n_rows, X2.memptr(), (int)X2.n_rows, (int)X1.n_cols, K.memptr(),
                     D2.memptr()) != GPAK_OK)
    gpak_host_fatal("gpak_compute_k", ctx);
}
void Kern_ExpAnisotropic::computeK(const mat &X1, const mat &X2, mat &K, mat &D2) const {
  double e[8];
  for (unsigned i = 0; i < 8; i++) e[i] = getParam(i);
  device_compute_k(e, 0.0, X1, X2, K, D2);
}

// ---------------------------------------------------------------------------------------------
// Kern_Bias
// ---------------------------------------------------------------------------------------------
void Kern_Bias::_init() {
  nParams = 1;
  setKerName("Bias");
  setParamName("Sigma_Bias", 0);
  setInitPars();
}
void Kern_Bias::setParam(double val, unsigned int paramNo) {
  if (paramNo != 0) { std::cout << "Requested parameter doesn't exist.\n"; exit(1); }
  Sigma_Bias = val;
}
double Kern_Bias::getParam(unsigned int paramNo) const {
  if (paramNo != 0) { std::cout << "Requested parameter doesn't exist.\n"; exit(1); }
  return Sigma_Bias;
}
void Kern_Bias::computeK(const mat &X1, const mat &X2, mat &K, mat &D2) const {  // Kernel.cpp:362-367
  if (K.n_rows != X1.n_rows || K.n_cols != X2.n_rows) K.resize(X1.n_rows, X2.n_rows);
  if (D2.n_rows != X1.n_rows || D2.n_cols != X2.n_rows) D2.resize(X1.n_rows, X2.n_rows);
  D2.zeros();
  K.fill(Sigma_Bias);
}

// ---------------------------------------------------------------------------------------------
// Kern_Exponential / Kern_RBF / Kern_White
// ---------------------------------------------------------------------------------------------
static void device_compute_composition(const Kernels *k, const mat &X1, const mat &X2, mat &K, mat &D2) {
  std::vector<int> kinds;
  std::vector<double> pars;
  double bias, white;
  if (!gpak_extract_composition(k, kinds, pars, &bias, &white) || kinds.empty()) {
    std::cerr << "computeK: composition not on the HIP path" << std::endl;
    exit(1);
  }
  gpak_ctx *ctx = gpak_host_scratch_ctx();
  if (gpak_set_kernel(ctx, (int)kinds.size(), kinds.data(), pars.data(), bias, white, 1.0, gpak_host_dist_mode()) != GPAK_OK)
    gpak_host_fatal("gpak_set_kernel", ctx);
  if (K.n_rows != X1.n_rows || K.n_cols != X2.n_rows) K.resize(X1.n_rows, X2.n_rows);
  if (D2.n_rows != X1.n_rows || D2.n_cols != X2.n_rows) D2.resize(X1.n_rows, X2.n_rows);
  if (gpak_compute_k(ctx, X1.memptr(), (int)X1.n_rows, X2.memptr(), (int)X2.n_rows, (int)X1.n_cols, K.memptr(),
                     D2.memptr()) != GPAK_OK)
    gpak_host_fatal("gpak_compute_k", ctx);
}
void Kern_Exponential::_init() {
  nParams = 2;
  setKerName("Exp");
  setParamName("Hayper_Euc_Exp", 0);
  setParamName("Sigma_Exp", 1);
  setInitPars();
}
void Kern_Exponential::setParam(double val, unsigned int paramNo) {
  if (paramNo == 0) Hayper_Euc_Exp = val;
  else if (paramNo == 1) Sigma_Exp = val;
  else { std::cout << "Requested parameter doesn't exist.\n"; exit(1); }
}
double Kern_Exponential::getParam(unsigned int paramNo) const {
  if (paramNo == 0) return Hayper_Euc_Exp;
  if (paramNo == 1) return Sigma_Exp;
  std::cout << "Requested parameter doesn't exist.\n"; exit(1);
}
void Kern_Exponential::computeK(const mat &X1, const mat &X2, mat &K, mat &D2) const { device_compute_composition(this, X1, X2, K, D2); }

void Kern_RBF::_init() {
  nParams = 3;
  setKerName("RBF");
  setParamName("Hayper_Euc_RBF", 0);
  setParamName("inverseWidth_RBF", 1);
  setParamName("Sigma_RBF", 2);
  setInitPars();
}
void Kern_RBF::setParam(double val, unsigned int paramNo) {
  if (paramNo == 0) Hayper_Euc_RBF = val;
  else if (paramNo == 1) inverseWidth_RBF = val;
  else if (paramNo == 2) Sigma_RBF = val;
  else { std::cout << "Requested parameter doesn't exist.\n"; exit(1); }
}
double Kern_RBF::getParam(unsigned int paramNo) const {
  if (paramNo == 0) return Hayper_Euc_RBF;
  if (paramNo == 1) return inverseWidth_RBF;
  if (paramNo == 2) return Sigma_RBF;
  std::cout << "Requested parameter doesn't exist.\n"; exit(1);
}
void Kern_RBF::computeK(const mat &X1, const mat &X2, mat &K, mat &D2) const { device_compute_composition(this, X1, X2, K, D2); }

void Kern_White::_init() {
  nParams = 1;
  setKerName("white");
  setParamName("Sigma_White", 0);
  setInitPars();
}
void Kern_White::setParam(double val, unsigned int paramNo) {
  if (paramNo != 0) { std::cout << "Requested parameter doesn't exist.\n"; exit(1); }
  Sigma_White = val;
}
double Kern_White::getParam(unsigned int paramNo) const {
  if (paramNo != 0) { std::cout << "Requested parameter doesn't exist.\n"; exit(1); }
  return Sigma_White;
}
void Kern_White::computeK(const mat &X1, const mat &X2, mat &K, mat &D2) const {  // Kernel.cpp:256-263
  if (K.n_rows != X1.n_rows || K.n_cols != X2.n_rows) K.resize(X1.n_rows, X2.n_rows);
  if (D2.n_rows != X1.n_rows || D2.n_cols != X2.n_rows) D2.resize(X1.n_rows, X2.n_rows);
  D2.zeros();
  K.zeros();
  if (X1(0) == X2(0) && X1.n_rows == X2.n_rows)
    for (size_t i = 0; i < X1.n_rows; i++) K(i, i) = Sigma_White;
}

// ---------------------------------------------------------------------------------------------
// HybKerns
// ---------------------------------------------------------------------------------------------
void HybKerns::_init() { nParams = 0; setKerName("Hyb"); }
HybKerns::HybKerns(const HybKerns &o) : Kernels(o) {
  for (auto *k : o.MainKEl) MainKEl.push_back(k->clone());
}
HybKerns::~HybKerns() { for (auto *k : MainKEl) delete k; }
unsigned int HybKerns::addNewKernel(const Kernels *kern) {
  MainKEl.push_back(kern->clone());
  nParams += kern->getNPars();
  return (unsigned)MainKEl.size() - 1;
}
void HybKerns::setParam(double val, unsigned int paramNo) {  // flat index, Kernel.h:172-186
  unsigned int start = 0;
  for (auto *k : MainKEl) {
    if (paramNo < start + k->getNPars()) { k->setParam(val, paramNo - start); return; }
    start += k->getNPars();
  }
}
double HybKerns::getParam(unsigned int paramNo) const {
  unsigned int start = 0;
  for (auto *k : MainKEl) {
    if (paramNo < start + k->getNPars()) return k->getParam(paramNo - start);
    start += k->getNPars();
  }
  return -1;
}
std::string HybKerns::getParamName(unsigned int paramNo) const {
  unsigned int start = 0;
  for (auto *k : MainKEl) {
    if (paramNo < start + k->getNPars()) return k->getParamName(paramNo - start);
    start += k->getNPars();
  }
  return "";
}
double HybKerns::Diag_Kernel(const mat &X, unsigned int index) const {
  double y = 0.0;
  for (auto *k : MainKEl) y += k->Diag_Kernel(X, index);
  return y;
}
void HybKerns::diag_Compute(mat &d, const mat &X) const {  // Kernel.cpp:127-136
  d.zeros();
  mat s(d.n_rows, d.n_cols);
  for (auto *k : MainKEl) {
    k->diag_Compute(s, X);
    for (size_t i = 0; i < d.n_elem; i++) d[i] += s[i];
  }
}
void HybKerns::computeK(const mat &X1, const mat &X2, mat &K, mat &D2) const {  // Kernel.cpp:140-154
  double e[8], bias;
  if (gpak_extract_expans_bias(this, e, &bias)) {
    device_compute_k(e, bias, X1, X2, K, D2);  // ExpAns + Bias fused into one fill
    return;
  }
  std::vector<int> kinds;
  std::vector<double> pars;
  double white;
  if (gpak_extract_composition(this, kinds, pars, &bias, &white) && !kinds.empty()) {
    device_compute_composition(this, X1, X2, K, D2);  // every child in the same fused fill
    return;
  }
  // anything else: sum the children as the reference does
  K.resize(X1.n_rows, X2.n_rows);
  D2.resize(X1.n_rows, X2.n_rows);
  mat Kt(X1.n_rows, X2.n_rows), Dt(X1.n_rows, X2.n_rows);
  for (auto *k : MainKEl) {
    k->computeK(X1, X2, Kt, Dt);
    for (size_t i = 0; i < K.n_elem; i++) { K[i] += Kt[i]; D2[i] += Dt[i]; }
  }
}
void HybKerns::ToFile_GP_Params(std::ostream &out) const {  // Kernel.cpp:64-75
  out << "KernelName=" << getKerName() << std::endl;
  out << "NumberOfKernels=" << getNumKerns() << std::endl;
  for (auto *k : MainKEl) k->StrmOut(out);
}
void HybKerns::FromFile_GP_Params(std::istream &in) {  // Kernel.cpp:55-62
  unsigned int n = ReadIntStrm(in, "NumberOfKernels");
  for (unsigned int i = 0; i < n; i++) {
    Kernels *k = ReadKerFromFile(in);
    addNewKernel(k);
    delete k;
  }
}
std::ostream &HybKerns::ShowKernelPars(std::ostream &os) const {
  for (auto *k : MainKEl) k->ShowKernelPars(os);
  return os;
}

Kernels *ReadKerFromFile(std::istream &in) {  // Kernel.cpp:1281-1307
  std::string line;
  std::getline(in, line);
  std::string name = line.substr(line.find("=") + 1);
  Kernels *k = nullptr;
  if (name == "Bias") k = new Kern_Bias();
  else if (name == "white") k = new Kern_White();
  else if (name == "RBF") k = new Kern_RBF();
  else if (name == "Exp") k = new Kern_Exponential();
  else if (name == "ExpAns") k = new Kern_ExpAnisotropic();
  else if (name == "Hyb") k = new HybKerns();
  else { std::cout << "Unknown kernel type \n"; exit(1); }
  k->FromFile_GP_Params(in);
  return k;
}

bool gpak_extract_expans_bias(const Kernels *k, double expans[8], double *bias) {
  *bias = 0.0;
  const Kern_ExpAnisotropic *ea = dynamic_cast<const Kern_ExpAnisotropic *>(k);
  if (ea) { for (unsigned i = 0; i < 8; i++) expans[i] = ea->getParam(i); return true; }
  const HybKerns *h = dynamic_cast<const HybKerns *>(k);
  if (!h || h->getNumKerns() < 1 || h->getNumKerns() > 2) return false;
  ea = dynamic_cast<const Kern_ExpAnisotropic *>(h->getKern(0));
  if (!ea) return false;
  for (unsigned i = 0; i < 8; i++) expans[i] = ea->getParam(i);
  if (h->getNumKerns() == 2) {
    const Kern_Bias *b = dynamic_cast<const Kern_Bias *>(h->getKern(1));
    if (!b) return false;
    *bias = b->getParam(0);
  }
  return true;
}

bool gpak_extract_composition(const Kernels *k, std::vector<int> &kinds, std::vector<double> &pars, double *bias,
                              double *white) {
  kinds.clear(); pars.clear();
  *bias = 0.0; *white = 0.0;
  std::vector<const Kernels *> leaves;
  if (const HybKerns *h = dynamic_cast<const HybKerns *>(k)) {
    for (unsigned i = 0; i < h->getNumKerns(); i++) leaves.push_back(h->getKern(i));
  } else {
    leaves.push_back(k);
  }
  for (const Kernels *c : leaves) {
    int kind = -1;
    if (dynamic_cast<const Kern_ExpAnisotropic *>(c)) kind = GPAK_KERN_EXPANS;
    else if (dynamic_cast<const Kern_Exponential *>(c)) kind = GPAK_KERN_EXP;
    else if (dynamic_cast<const Kern_RBF *>(c)) kind = GPAK_KERN_RBF;
    else if (dynamic_cast<const Kern_Bias *>(c)) { *bias += c->getParam(0); continue; }
    else if (dynamic_cast<const Kern_White *>(c)) { *white += c->getParam(0); continue; }
    else return false;
    if (kinds.size() == 3) return false;  // the device composition holds three stationary terms
    kinds.push_back(kind);
    for (unsigned i = 0; i < c->getNPars(); i++) pars.push_back(c->getParam(i));
  }
  return true;
}
