// control.cpp -- see control.hpp
#include "control.hpp"

#include <cmath>
#include <cstdlib>
#include <fstream>
#include <iostream>

Control::Control(int arc, char **arv) : argc(arc), argv(arv) {
  // global options before the verb: -v/--verboseL n, -pm/--prepMethod m (gp_ss_ak.cpp:14-63)
  while (argNo < argc && argv[argNo][0] == '-') {
    if (isArg("-v", "--verboseL")) { incArg(); verbose = getIntArg(); }
    else if (isArg("-pm", "--prepMethod")) { incArg(); prepareM = getIntArg(); }
    else if (isArg("-np", "--no-prompt")) { no_prompt = true; }
    // SURVEY.md section 5 "Config / flags": device-side options of the HIP path (not reference flags)
    else if (isArg("-g", "--gpus")) { incArg(); gpus = getIntArg(); if (gpus < 1) ErrorTermination("--gpus needs a positive count"); }
    else if (isArg("-p", "--precision")) {
      incArg();
      if (getArg() == "f64") precision = 0;
      else if (getArg() == "f32") precision = 1;
      else ErrorTermination("--precision takes f64 or f32");
    }
    else if (isArg("-t", "--timing")) { incArg(); timing_file = getArg(); }   // "-" = stdout
    else break;
    incArg();
  }
}
bool Control::isArg(const std::string &s, const std::string &l) const { return getArg() == s || getArg() == l; }
void Control::UnkFlg() const { ErrorTermination("Unknown flag: " + getArg() + " provided."); }
void Control::ErrorTermination(const std::string &error) const {
  std::cerr << error << std::endl << std::endl;
  std::cout << "To get more information use help command." << std::endl;
  exit(1);
}

static void split_line(const std::string &line, std::vector<std::string> &tok) {
  tok.clear();
  std::string cur;
  for (char ch : line) {
    if (ch == '\t' || ch == ',') { if (!cur.empty()) tok.push_back(cur); cur.clear(); }
    else if (ch != '\r') cur += ch;
  }
  if (!cur.empty()) tok.push_back(cur);
}

void Control::readDataSize(const std::string &file, int data_size[2]) const {
  std::ifstream in(file.c_str());
  if (!in.is_open()) ErrorTermination("File is " + file + " not readable");
  std::string line;
  std::vector<std::string> tok;
  int n = 0, maxd = 0;
  while (std::getline(in, line)) {
    if (!line.empty() && line[0] == '#') continue;
    split_line(line, tok);
    n++;
    if ((int)tok.size() - 1 > maxd) maxd = (int)tok.size() - 1;
  }
  data_size[0] = n; data_size[1] = maxd;
  if (verbose > 0) {
    std::cout << "Number of features in the input file are: " << maxd << std::endl;
    std::cout << "Number of readable data are: " << n << std::endl;
  }
}

void Control::readDataFile(mat &X, mat &y, const int data_size[2], const std::string &file) const {
  std::ifstream in(file.c_str());
  if (!in.is_open()) ErrorTermination("File is " + file + " not readable");
  X.resize(data_size[0], data_size[1]);
  y.resize(data_size[0], 1);
  std::string line;
  std::vector<std::string> tok;
  int p = 0;
  while (std::getline(in, line)) {
    if (!line.empty() && line[0] == '#') continue;
    split_line(line, tok);
    for (int j = 0; j < (int)tok.size(); j++) {
      if (j < data_size[1]) X(p, j) = atof(tok[j].c_str());
      else y(p) = atof(tok[j].c_str());
    }
    p++;
  }
}

void Control::loadStatistics(const std::string &ModelN, size_t d) {
  mat S;
  if (!S.load_csv(ModelN + "_Statistics.txt")) ErrorTermination("File is " + ModelN + "_Statistics.txt not readable");
  params.resize(d + 1, 2); MinData.resize(d + 1, 1); MaxData.resize(d + 1, 1); MeanData.resize(d + 1, 1); StData.resize(d + 1, 1);
  for (size_t i = 0; i < d + 1; i++) {
    params(i, 0) = S(i, 0); params(i, 1) = S(i, 1);
    MinData(i) = S(i, 2); MaxData(i) = S(i, 3); MeanData(i) = S(i, 4); StData(i) = S(i, 5);
  }
}

void Control::prepareData(mat &X, mat &y, bool yscale, const std::string &ModelN) {
  const size_t d = X.n_cols, n = X.n_rows;
  if (getMode() == "train") {
    // StatisticsCalc (Control.h:43-75): row 0 = y, rows 1..d = the input columns
    params.resize(d + 1, 2); MinData.resize(d + 1, 1); MaxData.resize(d + 1, 1); MeanData.resize(d + 1, 1); StData.resize(d + 1, 1);
    MaxTotalin = X.max(); MinTotalin = X.min(); MaxTotalo = y.max(); MinTotalo = y.min();
    for (size_t i = 0; i <= d; i++) {
      double s = 0, lo, hi;
      if (i == 0) { lo = MinTotalo; hi = MaxTotalo; for (size_t r = 0; r < n; r++) s += y(r); }
      else { lo = X.colmin(i - 1); hi = X.colmax(i - 1); for (size_t r = 0; r < n; r++) s += X(r, i - 1); }
      double mean = s / n, ss = 0;
      for (size_t r = 0; r < n; r++) { double v = (i == 0 ? y(r) : X(r, i - 1)) - mean; ss += v * v; }
      MinData(i) = lo; MaxData(i) = hi; MeanData(i) = mean; StData(i) = std::sqrt(ss / (n - 1));
    }
    if (prepareM == 0) {
      for (size_t i = 0; i <= d; i++) { params(i, 0) = MeanData(i); params(i, 1) = StData(i); }
    } else if (prepareM == 1) {  // prep_symmetric, Control.cpp:299-316
      params(0, 0) = 0.5 * (MaxTotalo + MinTotalo); params(0, 1) = 0.5 * (MaxTotalo - MinTotalo);
      for (size_t j = 0; j < 3 && j < d; j++) { params(j + 1, 0) = 0.5 * (MaxTotalin + MinTotalin); params(j + 1, 1) = 0.5 * (MaxTotalin - MinTotalin); }
      for (size_t j = 3; j < d; j++) { params(j + 1, 0) = 0.5 * (MaxData(j + 1) + MinData(j + 1)); params(j + 1, 1) = 0.5 * (MaxData(j + 1) - MinData(j + 1)); }
    } else if (prepareM == 2) {  // zeroandone, Control.cpp:278-286 (offset is 0.5*min, as written)
      for (size_t i = 0; i <= d; i++) { params(i, 0) = 0.5 * MinData(i); params(i, 1) = 0.5 * (MaxData(i) - MinData(i)); }
    } else ErrorTermination("Unrecognised preparation method.");
  } else {
    loadStatistics(ModelN, d);
  }
  for (size_t j = 0; j < d; j++)
    for (size_t r = 0; r < n; r++) X(r, j) = (X(r, j) - params(j + 1, 0)) / params(j + 1, 1);
  if (yscale)
    for (size_t r = 0; r < n; r++) y(r) = (y(r) - params(0, 0)) / params(0, 1);
  if (verbose > 0) std::cout << "Preparation method is " << (prepareM == 0 ? "between mean and standardDev" : prepareM == 1 ? "symmetric" : "between 0 and 1")
                             << " and y scale is " << yscale << std::endl;
  if (getMode() == "train") {  // Control.cpp:187-194: (d+1) x 6 csv: offset, scale, min, max, mean, std
    mat S(d + 1, 6);
    for (size_t i = 0; i <= d; i++) { S(i, 0) = params(i, 0); S(i, 1) = params(i, 1); S(i, 2) = MinData(i); S(i, 3) = MaxData(i); S(i, 4) = MeanData(i); S(i, 5) = StData(i); }
    S.save_csv(ModelN + "_Statistics.txt");
  }
}

void Control::postData(mat &X, mat &y, bool yscale, const std::string &ModelN) {
  if (getMode() == "test") loadStatistics(ModelN, X.n_cols);
  for (size_t j = 0; j < X.n_cols; j++)
    for (size_t r = 0; r < X.n_rows; r++) X(r, j) = X(r, j) * params(j + 1, 1) + params(j + 1, 0);
  if (yscale) for (size_t r = 0; r < y.n_elem; r++) y[r] = y[r] * params(0, 1) + params(0, 0);
}
void Control::postData(mat &y, bool, const std::string &) {
  for (size_t r = 0; r < y.n_elem; r++) y[r] = y[r] * params(0, 1) + params(0, 0);
}
void Control::postData_var(mat &v, bool yscale, const std::string &) {  // Control.cpp:238-255: returns a std-dev
  if (yscale) for (size_t r = 0; r < v.n_elem; r++) v[r] = std::sqrt(v[r] * params(0, 1) * params(0, 1));
}
