// gp_ss_ak.cpp -- the reference's command line (gp_ss_ak.cpp:14-557) over the HIP hot path:
//   gp_ss_ak [-v n] [-pm m] [-np] [device options] train [-k ExpAns] [-kn 1] [-o LBFGS] [-# iters] train.txt [model]
//   gp_ss_ak [-v n] [-pm m]       [device options] test  test.txt model train.txt [out_file]
// device options (not reference flags): --gpus n (multi-GPU context), --precision f64|f32
// (fp32 prediction work), --timing file|- (JSON of the context's phase times after the verb).
// Same verbs, flags and files (<model>, <model>_Statistics.txt, <model>_predict.txt,
// <model>_gnu.plt); -np/--no-prompt skips the two interactive stdin questions of `train`
// (gp_ss_ak.cpp:235-285) and the gnuplot call of `test` (:503-505).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <numeric>

#include "control.hpp"
#include "gp_utils.hpp"

class GP_Cntrl : public Control {
 public:
  GP_Cntrl(int argc, char **argv) : Control(argc, argv) { GP_utils::setDeviceOptions(precision, gpus); }
  void writeTiming(const GP_utils &m) const {
    if (timing_file.empty()) return;
    const std::string j = m.timingJson();
    if (timing_file == "-") { std::cout << "TIMING " << j << std::endl; return; }
    std::ofstream out(timing_file.c_str());
    out << j << "\n";
  }
  void train();
  void test();
  void Help() const;
};

void GP_Cntrl::Help() const {
  std::cout << "\nGP_SS_AK hot path on MI355X\nCommand:\n \t ./gp_ss_ak [options] Command [Comnd-options] TrainDataFile.txt modelName\n"
            << "Commands:\ntrain :\n \t To find hyperparameter by maxmizing likelihood.\n"
            << "test :\n \t To estimate test data set and plot the results.\n";
}

void GP_Cntrl::train() {
  incArg();
  setMode("train");
  bool yscale = true, Knoise = true;
  std::string optimiser = "LBFGS", modelName = "gp_model";
  std::vector<std::string> KernT;
  int iters = 100;
  while (isFlgs()) {
    if (isArgFlg()) {
      if (isArg("-h", "--help")) { Help(); exit(0); }
      else if (isArg("-mf", "--meanfunction")) { incArg(); if (getArg() != "mean_zero") ErrorTermination("Unrecognised mean function"); }
      else if (isArg("-lf", "--likefunction")) { incArg(); if (getArg() != "Gauss") ErrorTermination("Only the Gauss likelihood is reachable (gp_ss_ak.cpp:192-196)"); }
      else if (isArg("-k", "--kernel")) { incArg(); KernT.push_back(getArg()); }
      else if (isArg("-o", "--optimiser")) { incArg(); optimiser = getArg(); }
      else if (isArg("-#", "--iterations")) { incArg(); iters = getIntArg(); }
      else if (isArg("-kn", "--Knoise")) { incArg(); Knoise = getIntArg() != 0; }
      else if (isArg("-np", "--no-prompt")) { no_prompt = true; }
      else UnkFlg();
      incArg();
    } else setFlgs(false);
  }
  if (getArgNo() >= argc) ErrorTermination("There are not enough input parameters.");
  std::string trainFileName = getArg();
  if (getArgNo() + 1 < argc) modelName = argv[getArgNo() + 1];
  int ds[2];
  readDataSize(trainFileName, ds);
  mat X, y;
  readDataFile(X, y, ds, trainFileName);
  prepareData(X, y, yscale, modelName);

  HybKerns Kerns(X);
  for (auto &k : KernT) {
    if (k == "ExpAns") { Kern_ExpAnisotropic e(X); Kerns.addNewKernel(&e); }
    else if (k == "Bias") { Kern_Bias b(X); Kerns.addNewKernel(&b); }
    else if (k == "RBF") { Kern_RBF r(X); Kerns.addNewKernel(&r); }
    else if (k == "Exp") { Kern_Exponential e(X); Kerns.addNewKernel(&e); }
    else if (k == "White") { Kern_White w(X); Kerns.addNewKernel(&w); }
    else ErrorTermination("Unknown covariance function: " + k);
  }
  if (Kerns.getNumKerns() == 0) { Kern_ExpAnisotropic e(X); Kerns.addNewKernel(&e); }
  if (Knoise) { Kern_Bias b(X); Kerns.addNewKernel(&b); }

  GP_utils *GPModel = new GP_utils(&Kerns, X, y, GP_utils::inf_laplace, GP_utils::likeL_Gaussian, GP_utils::mean_zero,
                                   8, 1, 0, getVerbose());
  std::cout << "The inital value of the kernel parameters are as follows :" << std::endl;
  std::cout << "There are " << GPModel->KerenlW->getNPars() << " parameters to be optimized" << std::endl;
  for (unsigned i = 0; i < GPModel->KerenlW->getNPars(); i++)
    std::cout << GPModel->KerenlW->getParamName(i) << " : " << GPModel->KerenlW->getParam(i) << std::endl;
  if (!no_prompt) {
    std::cout << "Do you want to change the defult kernel parameters (Yes|Y|y or press any key)?" << std::endl;
    std::string res = "No";
    std::cin >> res;
    if (res == "Yes" || res == "Y" || res == "y") {
      for (unsigned i = 0; i < GPModel->KerenlW->getNPars(); i++) {
        if (GPModel->KerenlW->getParamName(i) == "InversewidthR_ExpAns" && X.n_cols == 3) continue;
        std::cout << " Please input an initial value for " << GPModel->KerenlW->getParamName(i) << " (Default value was "
                  << GPModel->KerenlW->getParam(i) << ") : " << std::endl;
        double d = GPModel->KerenlW->getParam(i);
        std::cin >> d;
        GPModel->KerenlW->setParam(d, i);
      }
    }
  }
  std::cout << "The inital value of the likelihood function are as follows :" << std::endl;
  std::cout << "likelihood hyperparameter : " << GPModel->getHyperlfVal(0) << std::endl;
  if (!no_prompt) {
    std::cout << "Do you want to change the defult likelihood function parameters (Yes|Y|y or press any key)?" << std::endl;
    std::string res = "No";
    std::cin >> res;
    if (res == "Yes" || res == "Y" || res == "y") {
      std::cout << "Please input an initial value for Gauss likelihood function : " << std::endl;
      double d = GPModel->getHyperlfVal(0);
      std::cin >> d;
      GPModel->setHyperlfVal(d, 0);
    }
  }
  if (optimiser == "LBFGS") GPModel->setOptimiser(GP_utils::LBFGS);
  else if (optimiser == "BFGS" || optimiser == "SCG") {
    std::cout << "Optimiser " << optimiser << " is not built on the HIP path; using LBFGS." << std::endl;
    GPModel->setOptimiser(GP_utils::LBFGS);
  } else ErrorTermination("Unrecognised optimiser type: " + optimiser);
  // gp_ss_ak.cpp:295: the iteration count travels only through OptimisePars, which applies it when verbosity > 2
  // (GP_Utils.cpp:1295-1296); otherwise the constructor's 100 (Opt_pars.h:35) stays.  Reproduced as written.
  GPModel->OptimisePars(iters);

  writeGPFile(*GPModel, modelName, "# GP_SS_AK Model File ");
  // the reference evaluates the model on its own training set (gp_ss_ak.cpp:301-325)
  mat EstVals(X.n_rows, 1), EstVals_Var(X.n_rows, 1);
  GPModel->Calc_Out(EstVals, EstVals_Var, X);
  postData(X, EstVals, yscale, modelName);
  postData_var(EstVals_Var, yscale, modelName);
  postData(y, yscale, modelName);
  double mse = 0, ym = 0, vy = 0;
  for (size_t i = 0; i < y.n_elem; i++) { mse += (y[i] - EstVals[i]) * (y[i] - EstVals[i]); ym += y[i]; }
  mse /= X.n_rows; ym /= y.n_elem;
  for (size_t i = 0; i < y.n_elem; i++) vy += (y[i] - ym) * (y[i] - ym);
  vy /= y.n_elem;
  if (getVerbose() > 0) { std::cout << "Mean Square Error of training: " << mse << "\n"; std::cout << "Var MSE Train: " << vy << "\n"; }
  else { std::cout << mse << "\n" << vy << "\n"; }
  writeTiming(*GPModel);
  delete GPModel;
  exit(0);
}

void GP_Cntrl::test() {
  incArg();
  setMode("test");
  bool yscale = true;
  std::string modelName = "model", trFile;
  while (isFlgs()) {
    if (isArgFlg()) {
      if (isArg("-?", "--?") || isArg("-h", "--help")) { Help(); exit(0); }
      else if (isArg("-np", "--no-prompt")) { no_prompt = true; }
      else UnkFlg();
      incArg();
    } else setFlgs(false);
  }
  if (getArgNo() >= argc) ErrorTermination("There are not enough input parameters.");
  std::string teFile = getArg();
  if (getArgNo() + 1 < argc) modelName = argv[getArgNo() + 1];
  if (getArgNo() + 2 < argc) trFile = argv[getArgNo() + 2];
  else ErrorTermination("Please provide training data");
  std::string PredictOut = modelName + "_predict.txt";
  if (getArgNo() + 3 < argc) PredictOut = argv[getArgNo() + 3];
  int ds[2];
  readDataSize(teFile, ds);
  mat X, y;
  readDataFile(X, y, ds, teFile);
  prepareData(X, y, yscale, modelName);
  GP_utils *GPModel = readGpFromFile(modelName, getVerbose());  // parameters at 6 significant digits (Q5)
  readDataSize(trFile, ds);
  mat Xtr, ytr;
  readDataFile(Xtr, ytr, ds, trFile);
  prepareData(Xtr, ytr, yscale, modelName);
  GPModel->yTarg = ytr;
  GPModel->Xinp = Xtr;
  GPModel->setNumData((unsigned)Xtr.n_rows);
  GPModel->initialize_vars();
  GPModel->logLikelihood();
  if (X.n_cols != GPModel->getInpDim()) ErrorTermination("Incorrect dimension of input data.");
  mat EstVals(y.n_rows, 1), EstVals_Var(y.n_rows, 1);
  GPModel->Calc_Out(EstVals, EstVals_Var, X);
  postData(X, EstVals, yscale, modelName);
  postData_var(EstVals_Var, yscale, modelName);
  postData(y, yscale, modelName);
  double mse = 0, ym = 0, vy = 0;
  for (size_t i = 0; i < y.n_elem; i++) { mse += (y[i] - EstVals[i]) * (y[i] - EstVals[i]); ym += y[i]; }
  mse /= X.n_rows; ym /= y.n_elem;
  for (size_t i = 0; i < y.n_elem; i++) vy += (y[i] - ym) * (y[i] - ym);
  vy /= y.n_elem;
  if (getVerbose() > 0) { std::cout << "Mean Square Error of testing: " << mse << "\n"; std::cout << "Var MSE Test: " << vy << "\n"; }
  else { std::cout << mse << "\n" << vy << "\n"; }
  // rows sorted by ascending y (gp_ss_ak.cpp:434-481)
  std::vector<size_t> idx(y.n_elem);
  std::iota(idx.begin(), idx.end(), 0);
  std::stable_sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return y[a] < y[b]; });
  std::ofstream out(PredictOut.c_str());
  out << "# SampleNo, Y,  Yh, StdYh, Inputs" << "\n";
  for (size_t r = 0; r < idx.size(); r++) {
    size_t i = idx[r];
    out << (r + 1) << "\t" << y[i] << "\t" << EstVals[i] << "\t" << EstVals_Var[i] << "\t";
    for (size_t j = 0; j < X.n_cols; j++) out << X(i, j) << "\t";
    out << "\n";
  }
  out.close();
  std::ofstream gnu((modelName + "_gnu.plt").c_str());
  gnu << "#gnuplot -persist output.plt\n set term pdf transparent enhanced \n set output '" << modelName << "_predict.pdf'  \n"
      << "set title \"Observed vs Estimated\"\n set xlabel \"Sample\"\n"
      << "plot \"" << PredictOut << "\" using 1:($3 + $4):($3 - $4) with filledcurve fc rgb \"green\" title '95% CI', "
      << "\"\" using 1:3 with lines lc rgb \"red\" t \"Estimated\", \"\" using 1:2 lc rgb \"blue\" t \"Observed\" with lines \n";
  gnu.close();
  if (!no_prompt && system("command -v gnuplot > /dev/null 2>&1") == 0) {
    std::string cmd = "gnuplot -persist " + modelName + "_gnu.plt";
    if (system(cmd.c_str()) != 0) std::cerr << "gnuplot failed" << std::endl;
  }
  writeTiming(*GPModel);
  delete GPModel;
  exit(0);
}

int main(int argc, char **argv) {
  GP_Cntrl ctl(argc, argv);
  if (ctl.getArgNo() >= argc) { ctl.Help(); return 1; }
  std::string verb = ctl.getArg();
  if (verb == "train") ctl.train();
  else if (verb == "test") ctl.test();
  else if (verb == "-h" || verb == "--help" || verb == "-?") { ctl.Help(); return 0; }
  else ctl.ErrorTermination("Invalid command provided.");
  return 0;
}
