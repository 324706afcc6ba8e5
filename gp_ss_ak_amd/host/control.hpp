// control.hpp -- argument walker, data reader and (de)standardisation of the reference's
// Control class (Control.h:23-169, Control.cpp:5-337): host glue around the hot path.
#pragma once
#include <string>

#include "gpak_mat.hpp"

using gpak_host::mat;

class Control {
 public:
  Control(int argc, char **argv);
  virtual ~Control() {}
  bool isArg(const std::string &shortName, const std::string &longName) const;
  bool isArgFlg() const { return argNo < argc && argv[argNo][0] == '-'; }
  bool isFlgs() const { return flgs && argNo < argc; }
  void setFlgs(bool v) { flgs = v; }
  void incArg() { argNo++; }
  int getArgNo() const { return argNo; }
  std::string getArg() const { return argNo < argc ? argv[argNo] : ""; }
  int getIntArg() const { return atoi(getArg().c_str()); }
  size_t getArgLen() const { return getArg().size(); }
  void UnkFlg() const;
  void setMode(const std::string &m) { mode = m; }
  std::string getMode() const { return mode; }
  int getVerbose() const { return verbose; }

  // last column = y, separators tab or comma, '#' comment lines (Control.cpp:27-141)
  void readDataSize(const std::string &file, int data_size[2]) const;
  void readDataFile(mat &X, mat &y, const int data_size[2], const std::string &file) const;
  // prepareM: 0 mean/std, 1 symmetric (default), 2 zero-and-one (Control.cpp:142-324)
  void prepareData(mat &X, mat &y, bool yscale, const std::string &ModelN);
  void postData(mat &X, mat &y, bool yscale, const std::string &ModelN);
  void postData(mat &y, bool yscale, const std::string &ModelN);
  void postData_var(mat &v, bool yscale, const std::string &ModelN);
  void ErrorTermination(const std::string &error) const;

  int argc;
  char **argv;

 protected:
  void loadStatistics(const std::string &ModelN, size_t d);
  int argNo = 1;
  bool flgs = true;
  int verbose = 0;
  int prepareM = 1;
  bool no_prompt = false;
  int gpus = 1;                  // --gpus n: block-column-cyclic multi-GPU context (gpak_create_multi)
  int precision = 0;             // --precision f64|f32: GPAK_F64 / GPAK_F32 (fp32 prediction work)
  std::string timing_file;       // --timing file|-: JSON of gpak_phase_times after the verb
  std::string mode = "gp";
  mat params, MinData, MaxData, MeanData, StData;
  double MaxTotalin = 0, MinTotalin = 0, MaxTotalo = 0, MinTotalo = 0;
};
