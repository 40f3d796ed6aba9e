// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/ad.hpp header).
//
// Just enough modelling sugar to restate the reference's benchmark/test problems
// with the SAME graph-construction order as the reference DSL:
//   include/sleipnir/autodiff/variable.hpp        (:143-151 value, operators,
//                                                  :716-778 make_constraints,
//                                                  :1008-1013 bounds)
//   include/sleipnir/autodiff/variable_matrix.hpp (:505-521 matmul with the
//                                                  `sum{0} += a*b` pattern,
//                                                  :609-640 scalar*matrix puts the
//                                                  MATRIX element on the left,
//                                                  :1516-1541 solve 1x1/2x2)
// Not API-compatible with slp:: (g++-11 has no multi-arg operator[]); only the
// produced expression graphs matter.
#pragma once

#include <cassert>
#include <initializer_list>
#include <vector>

#include "ad.hpp"

namespace orc {

struct Var {
  Expr* e = nullptr;
  Var() : e(decision_variable()) {}  // variable.hpp:289-290 default = decision variable
  Var(double v) : e(constant(v)) {}  // NOLINT
  Var(int v) : e(constant(static_cast<double>(v))) {}  // NOLINT
  explicit Var(Expr* x) : e(x) {}
  static Var null() { return Var(static_cast<Expr*>(nullptr)); }

  void set_value(double v) { e->val = v; }
  // variable.hpp:143-151
  double value() {
    Graph g = topological_sort(e);
    update_values(g);
    return e->val;
  }
  Type type() const { return e->type; }
};

inline Var operator*(const Var& a, const Var& b) { return Var(mul(a.e, b.e)); }
inline Var operator/(const Var& a, const Var& b) { return Var(div(a.e, b.e)); }
inline Var operator+(const Var& a, const Var& b) { return Var(add(a.e, b.e)); }
inline Var operator-(const Var& a, const Var& b) { return Var(sub(a.e, b.e)); }
inline Var operator-(const Var& a) { return Var(neg(a.e)); }
inline Var& operator+=(Var& a, const Var& b) { return a = a + b; }
inline Var& operator-=(Var& a, const Var& b) { return a = a - b; }
inline Var& operator*=(Var& a, const Var& b) { return a = a * b; }
inline Var& operator/=(Var& a, const Var& b) { return a = a / b; }

inline Var sin(const Var& x) { return Var(sin(x.e)); }
inline Var cos(const Var& x) { return Var(cos(x.e)); }
inline Var pow(const Var& b, const Var& p) { return Var(pow(b.e, p.e)); }
inline Var sqrt(const Var& x) { return Var(sqrt(x.e)); }
inline Var exp(const Var& x) { return Var(exp(x.e)); }
inline Var log(const Var& x) { return Var(log(x.e)); }
inline Var abs(const Var& x) { return Var(abs(x.e)); }
inline Var hypot(const Var& x, const Var& y) { return Var(hypot(x.e, y.e)); }

// Row-major matrix of Var handles (variable_matrix.hpp:310-313)
struct VarMat {
  int rows = 0, cols = 0;
  std::vector<Var> s;

  VarMat() = default;
  // detail::empty constructor (:60-67): null handles
  VarMat(int r, int c) : rows(r), cols(c), s(static_cast<size_t>(r) * c, Var::null()) {}
  VarMat(std::initializer_list<std::initializer_list<Var>> list) {
    rows = static_cast<int>(list.size());
    cols = rows ? static_cast<int>(list.begin()->size()) : 0;
    for (auto& row : list)
      for (auto& v : row) s.push_back(v);
  }
  VarMat(const Var& v) : rows(1), cols(1), s{v} {}  // NOLINT

  Var& operator()(int r, int c) { return s[static_cast<size_t>(r) * cols + c]; }
  const Var& operator()(int r, int c) const { return s[static_cast<size_t>(r) * cols + c]; }
  Var& operator()(int i) { return s[i]; }
  const Var& operator()(int i) const { return s[i]; }

  VarMat block(int r0, int c0, int nr, int nc) const {
    VarMat m(nr, nc);
    for (int r = 0; r < nr; ++r)
      for (int c = 0; c < nc; ++c) m(r, c) = (*this)(r0 + r, c0 + c);
    return m;
  }
  void set_block(int r0, int c0, const VarMat& b) {
    for (int r = 0; r < b.rows; ++r)
      for (int c = 0; c < b.cols; ++c) (*this)(r0 + r, c0 + c) = b(r, c);
  }
  VarMat col(int c) const { return block(0, c, rows, 1); }
  VarMat row(int r) const { return block(r, 0, 1, cols); }
  VarMat segment(int off, int len) const {  // column-vector segment
    assert(cols == 1);
    return block(off, 0, len, 1);
  }
  VarMat T() const {
    VarMat m(cols, rows);
    for (int r = 0; r < rows; ++r)
      for (int c = 0; c < cols; ++c) m(c, r) = (*this)(r, c);
    return m;
  }
};

// Dense constant matrix (stands in for Eigen::Matrix<double,...> operands)
struct Mat {
  int rows = 0, cols = 0;
  std::vector<double> d;
  Mat() = default;
  Mat(int r, int c) : rows(r), cols(c), d(static_cast<size_t>(r) * c, 0.0) {}
  Mat(std::initializer_list<std::initializer_list<double>> list) {
    rows = static_cast<int>(list.size());
    cols = rows ? static_cast<int>(list.begin()->size()) : 0;
    for (auto& row : list)
      for (double v : row) d.push_back(v);
  }
  double& operator()(int r, int c) { return d[static_cast<size_t>(r) * cols + c]; }
  double operator()(int r, int c) const { return d[static_cast<size_t>(r) * cols + c]; }
};

// variable_matrix.hpp:505-557 (all three matmul overloads share this loop)
template <typename L, typename R>
VarMat matmul(const L& lhs, const R& rhs) {
  assert(lhs.cols == rhs.rows);
  VarMat result(lhs.rows, rhs.cols);
  for (int i = 0; i < lhs.rows; ++i)
    for (int j = 0; j < rhs.cols; ++j) {
      Var sum{0.0};
      for (int k = 0; k < lhs.cols; ++k) sum += Var(lhs(i, k)) * Var(rhs(k, j));
      result(i, j) = sum;
    }
  return result;
}
inline VarMat operator*(const VarMat& a, const VarMat& b) { return matmul(a, b); }
inline VarMat operator*(const Mat& a, const VarMat& b) { return matmul(a, b); }
inline VarMat operator*(const VarMat& a, const Mat& b) { return matmul(a, b); }

// :592-640: matrix (x) scalar always builds `element * scalar`
inline VarMat operator*(const VarMat& a, const Var& s) {
  VarMat r(a.rows, a.cols);
  for (size_t i = 0; i < a.s.size(); ++i) r.s[i] = a.s[i] * s;
  return r;
}
inline VarMat operator*(const Var& s, const VarMat& a) { return a * s; }
inline VarMat operator*(double s, const VarMat& a) { return a * Var(s); }
inline VarMat operator*(const VarMat& a, double s) { return a * Var(s); }
inline VarMat operator/(const VarMat& a, const Var& s) {
  VarMat r(a.rows, a.cols);
  for (size_t i = 0; i < a.s.size(); ++i) r.s[i] = a.s[i] / s;
  return r;
}
inline VarMat operator+(const VarMat& a, const VarMat& b) {
  assert(a.rows == b.rows && a.cols == b.cols);
  VarMat r(a.rows, a.cols);
  for (size_t i = 0; i < a.s.size(); ++i) r.s[i] = a.s[i] + b.s[i];
  return r;
}
inline VarMat operator-(const VarMat& a, const VarMat& b) {
  assert(a.rows == b.rows && a.cols == b.cols);
  VarMat r(a.rows, a.cols);
  for (size_t i = 0; i < a.s.size(); ++i) r.s[i] = a.s[i] - b.s[i];
  return r;
}
inline VarMat operator-(const VarMat& a) {
  VarMat r(a.rows, a.cols);
  for (size_t i = 0; i < a.s.size(); ++i) r.s[i] = -a.s[i];
  return r;
}

// variable_matrix.hpp:1516-1541 (1x1 and 2x2 closed forms)
inline VarMat solve(const VarMat& A, const VarMat& B) {
  assert(A.rows == B.rows);
  if (A.rows == 1 && A.cols == 1) {
    return VarMat(B(0, 0) / A(0, 0));
  }
  assert(A.rows == 2 && A.cols == 2);
  const Var& a = A(0, 0);
  const Var& b = A(0, 1);
  const Var& c = A(1, 0);
  const Var& d = A(1, 1);
  VarMat adj_A{{d, -b}, {-c, a}};
  Var det_A = a * d - b * c;
  return adj_A / det_A * B;
}

// variable.hpp:716-778 make_constraints: rows are lhs - rhs, row-major
inline std::vector<Var> make_constraints(const VarMat& lhs, const VarMat& rhs) {
  std::vector<Var> out;
  if (lhs.rows == 1 && lhs.cols == 1 && !(rhs.rows == 1 && rhs.cols == 1)) {
    for (auto& r : rhs.s) out.push_back(lhs.s[0] - r);
  } else if (rhs.rows == 1 && rhs.cols == 1 && !(lhs.rows == 1 && lhs.cols == 1)) {
    for (auto& l : lhs.s) out.push_back(l - rhs.s[0]);
  } else {
    assert(lhs.rows == rhs.rows && lhs.cols == rhs.cols);
    for (size_t i = 0; i < lhs.s.size(); ++i) out.push_back(lhs.s[i] - rhs.s[i]);
  }
  return out;
}
inline VarMat to_varmat(const Mat& m) {
  VarMat r(m.rows, m.cols);
  for (size_t i = 0; i < m.d.size(); ++i) r.s[i] = Var(m.d[i]);
  return r;
}
inline VarMat col_vector(std::initializer_list<double> v) {
  VarMat r(static_cast<int>(v.size()), 1);
  int i = 0;
  for (double x : v) r.s[i++] = Var(x);
  return r;
}

// lhs == rhs  -> equality rows; lhs >= rhs -> inequality rows (c(x) >= 0)
inline std::vector<Var> eq(const VarMat& lhs, const VarMat& rhs) { return make_constraints(lhs, rhs); }
inline std::vector<Var> ge(const VarMat& lhs, const VarMat& rhs) { return make_constraints(lhs, rhs); }
inline std::vector<Var> le(const VarMat& lhs, const VarMat& rhs) { return make_constraints(rhs, lhs); }
// variable.hpp:1008-1013: bounds(l, x, u) = {l <= x, x <= u} = {x - l ..., u - x ...}
inline std::vector<Var> bounds(const VarMat& l, const VarMat& x, const VarMat& u) {
  std::vector<Var> out = le(l, x);
  std::vector<Var> hi = le(x, u);
  out.insert(out.end(), hi.begin(), hi.end());
  return out;
}

}  // namespace orc
