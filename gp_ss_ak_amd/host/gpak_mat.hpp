// gpak_mat.hpp -- the small column-major double matrix the host classes use where the
// reference uses arma::mat (Armadillo is neither available nor needed: all O(N^2)/O(N^3)
// work happens behind the C-ABI).  Same storage order and accessors as arma::mat
// (memptr(), n_rows, n_cols, (i,j), (i)), so a maintainer can swap the typedef back.
#pragma once
#include <cmath>
#include <cstddef>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

namespace gpak_host {

class mat {
 public:
  size_t n_rows = 0, n_cols = 0, n_elem = 0;
  mat() {}
  mat(size_t r, size_t c) { resize(r, c); }
  void resize(size_t r, size_t c) {
    n_rows = r; n_cols = c; n_elem = r * c;
    v_.assign(n_elem, 0.0);
  }
  void zeros() { fill(0.0); }
  void zeros(size_t r, size_t c) { resize(r, c); }
  void ones() { fill(1.0); }
  void fill(double x) { for (auto &e : v_) e = x; }
  double *memptr() { return v_.data(); }
  const double *memptr() const { return v_.data(); }
  double &operator()(size_t i, size_t j) { return v_[i + j * n_rows]; }
  double operator()(size_t i, size_t j) const { return v_[i + j * n_rows]; }
  double &operator()(size_t i) { return v_[i]; }
  double operator()(size_t i) const { return v_[i]; }
  double &operator[](size_t i) { return v_[i]; }
  double operator[](size_t i) const { return v_[i]; }
  double min() const { double m = v_.empty() ? 0 : v_[0]; for (double e : v_) m = e < m ? e : m; return m; }
  double max() const { double m = v_.empty() ? 0 : v_[0]; for (double e : v_) m = e > m ? e : m; return m; }
  double colmin(size_t j) const { double m = (*this)(0, j); for (size_t i = 1; i < n_rows; i++) m = (*this)(i, j) < m ? (*this)(i, j) : m; return m; }
  double colmax(size_t j) const { double m = (*this)(0, j); for (size_t i = 1; i < n_rows; i++) m = (*this)(i, j) > m ? (*this)(i, j) : m; return m; }
  bool has_nan() const { for (double e : v_) if (e != e) return true; return false; }

  // csv_ascii like arma::mat::save / load
  bool save_csv(const std::string &name) const {
    std::ofstream out(name.c_str());
    if (!out) return false;
    out.precision(14);
    out << std::scientific;
    for (size_t i = 0; i < n_rows; i++) {
      for (size_t j = 0; j < n_cols; j++) out << (*this)(i, j) << (j + 1 < n_cols ? "," : "");
      out << "\n";
    }
    return true;
  }
  bool load_csv(const std::string &name) {
    std::ifstream in(name.c_str());
    if (!in.is_open()) return false;
    std::vector<std::vector<double>> rows;
    std::string line;
    while (std::getline(in, line)) {
      if (line.empty()) continue;
      std::vector<double> r;
      std::stringstream ss(line);
      std::string tok;
      while (std::getline(ss, tok, ',')) r.push_back(atof(tok.c_str()));
      rows.push_back(r);
    }
    if (rows.empty()) return false;
    resize(rows.size(), rows[0].size());
    for (size_t i = 0; i < n_rows; i++)
      for (size_t j = 0; j < n_cols && j < rows[i].size(); j++) (*this)(i, j) = rows[i][j];
    return true;
  }

 private:
  std::vector<double> v_;
};

}  // namespace gpak_host
