// slp::Variable / slp::VariableMatrix / slp::VariableBlock — the modelling surface
// of the reference kept source-compatible for the path's callers
// (include/sleipnir/autodiff/variable.hpp, variable_matrix.hpp, variable_block.hpp),
// but recording into the flat SoA arena of graph.hpp instead of a pointer graph.
//
// Kept semantics (cited where they matter for graph identity):
//   * default-constructed Variable = decision variable with value 0 (variable.hpp:289-290)
//   * scalar (x) matrix builds `element * scalar` (variable_matrix.hpp:592-640)
//   * matmul accumulates `sum{0}; sum += a*b` (variable_matrix.hpp:505-557)
//   * lhs ? rhs builds rows `lhs - rhs`, row-major (variable.hpp:716-778)
//   * bounds(l, x, u) = {l <= x, x <= u} (variable.hpp:1008-1013)
//   * closed-form solve() for 1x1..3x3 (variable_matrix.hpp:1516-1620)
// Eigen is not available in this toolchain; slp::DenseMatrix stands in for the
// Eigen::Matrix<double,...> constants that appear in problem definitions.
#pragma once

#include <cassert>
#include <cmath>
#include <concepts>
#include <initializer_list>
#include <type_traits>
#include <utility>
#include <vector>

#include "../graph.hpp"

namespace slp {

using slpx::ExprType;
using slpx::NodeId;

namespace detail {
inline slpx::Graph& G() { return slpx::graph(); }
struct empty_t {};
inline constexpr empty_t empty{};
}  // namespace detail

enum class ExpressionType : uint8_t { NONE = 0, CONSTANT, LINEAR, QUADRATIC, NONLINEAR };

class VariableMatrix;
class VariableBlock;

class Variable {
 public:
  using Scalar = double;

  Variable() : expr{detail::G().variable(0.0)} {}
  explicit constexpr Variable(std::nullptr_t) : expr{slpx::kNull} {}
  Variable(double value) : expr{detail::G().constant(value)} {}  // NOLINT
  Variable(std::integral auto value) : expr{detail::G().constant(static_cast<double>(value))} {}  // NOLINT
  Variable(const VariableMatrix& value);  // NOLINT (1x1 only)
  Variable(const VariableBlock& value);   // NOLINT (1x1 only)
  struct from_node_t {};
  Variable(from_node_t, NodeId n) : expr{n} {}

  Variable& operator=(double value) {
    expr = detail::G().constant(value);
    return *this;
  }

  // variable.hpp:125-138
  void set_value(double value) { detail::G().val[expr] = value; }
  // variable.hpp:143-151
  double value() const { return detail::G().value(expr); }
  ExpressionType type() const { return static_cast<ExpressionType>(detail::G().type[expr]); }

  friend Variable operator*(const Variable& l, const Variable& r) { return wrap(detail::G().mul(l.expr, r.expr)); }
  friend Variable operator/(const Variable& l, const Variable& r) { return wrap(detail::G().div(l.expr, r.expr)); }
  friend Variable operator+(const Variable& l, const Variable& r) { return wrap(detail::G().add(l.expr, r.expr)); }
  friend Variable operator-(const Variable& l, const Variable& r) { return wrap(detail::G().sub(l.expr, r.expr)); }
  friend Variable operator-(const Variable& l) { return wrap(detail::G().neg(l.expr)); }
  friend Variable operator+(const Variable& l) { return l; }
  // arithmetic-with-scalar overloads: exact matches, so `0.5 * x` never competes
  // with the Variable<->VariableMatrix conversions (variable.hpp:157-167)
  friend Variable operator*(double l, const Variable& r) { return Variable{l} * r; }
  friend Variable operator*(const Variable& l, double r) { return l * Variable{r}; }
  friend Variable operator/(double l, const Variable& r) { return Variable{l} / r; }
  friend Variable operator/(const Variable& l, double r) { return l / Variable{r}; }
  friend Variable operator+(double l, const Variable& r) { return Variable{l} + r; }
  friend Variable operator+(const Variable& l, double r) { return l + Variable{r}; }
  friend Variable operator-(double l, const Variable& r) { return Variable{l} - r; }
  friend Variable operator-(const Variable& l, double r) { return l - Variable{r}; }
  Variable& operator*=(const Variable& r) { return *this = *this * r; }
  Variable& operator/=(const Variable& r) { return *this = *this / r; }
  Variable& operator+=(const Variable& r) { return *this = *this + r; }
  Variable& operator-=(const Variable& r) { return *this = *this - r; }

  static Variable wrap(NodeId n) { return Variable{from_node_t{}, n}; }

  NodeId expr;
};

#define SLP_UNARY(name, OP) \
  inline Variable name(const Variable& x) { return Variable::wrap(detail::G().unary(slpx::OP, x.expr)); }
SLP_UNARY(abs, OP_ABS)
SLP_UNARY(acos, OP_ACOS)
SLP_UNARY(asin, OP_ASIN)
SLP_UNARY(atan, OP_ATAN)
SLP_UNARY(cbrt, OP_CBRT)
SLP_UNARY(cos, OP_COS)
SLP_UNARY(cosh, OP_COSH)
SLP_UNARY(erf, OP_ERF)
SLP_UNARY(exp, OP_EXP)
SLP_UNARY(log, OP_LOG)
SLP_UNARY(log10, OP_LOG10)
SLP_UNARY(sign, OP_SIGN)
SLP_UNARY(sin, OP_SIN)
SLP_UNARY(sinh, OP_SINH)
SLP_UNARY(sqrt, OP_SQRT)
SLP_UNARY(tan, OP_TAN)
SLP_UNARY(tanh, OP_TANH)
#undef SLP_UNARY
#define SLP_BINARY(name, OP)                                     \
  inline Variable name(const Variable& a, const Variable& b) {   \
    return Variable::wrap(detail::G().binary(slpx::OP, a.expr, b.expr)); \
  }
SLP_BINARY(atan2, OP_ATAN2)
SLP_BINARY(hypot, OP_HYPOT)
SLP_BINARY(max, OP_MAX)
SLP_BINARY(min, OP_MIN)
SLP_BINARY(pow, OP_POW)
#undef SLP_BINARY
// variable.hpp:711-714
inline Variable hypot(const Variable& x, const Variable& y, const Variable& z) {
  return sqrt(pow(x, 2) + pow(y, 2) + pow(z, 2));
}

// Dense constant matrix (row-major), stand-in for Eigen::Matrix in model code
class DenseMatrix {
 public:
  DenseMatrix() = default;
  DenseMatrix(int rows, int cols) : m_rows{rows}, m_cols{cols}, m_d(static_cast<size_t>(rows) * cols, 0.0) {}
  DenseMatrix(std::initializer_list<std::initializer_list<double>> list) {
    m_rows = static_cast<int>(list.size());
    m_cols = m_rows ? static_cast<int>(list.begin()->size()) : 0;
    for (auto& row : list)
      for (double v : row) m_d.push_back(v);
  }
  static DenseMatrix vector(std::initializer_list<double> v) {
    DenseMatrix m(static_cast<int>(v.size()), 1);
    int i = 0;
    for (double x : v) m.m_d[i++] = x;
    return m;
  }
  int rows() const { return m_rows; }
  int cols() const { return m_cols; }
  double& operator[](int r, int c) { return m_d[static_cast<size_t>(r) * m_cols + c]; }
  double operator[](int r, int c) const { return m_d[static_cast<size_t>(r) * m_cols + c]; }
  double& operator[](int i) { return m_d[i]; }
  double operator[](int i) const { return m_d[i]; }
  double& operator()(int r, int c) { return (*this)[r, c]; }
  double operator()(int r, int c) const { return (*this)[r, c]; }

 private:
  int m_rows = 0, m_cols = 0;
  std::vector<double> m_d;
};

class VariableMatrix {
 public:
  using Scalar = double;
  VariableMatrix() = default;
  explicit VariableMatrix(int rows) : VariableMatrix(rows, 1) {}
  // variable_matrix.hpp:47-58: filled with default (decision-variable) handles
  VariableMatrix(int rows, int cols) : m_rows{rows}, m_cols{cols} {
    m_storage.reserve(static_cast<size_t>(rows) * cols);
    for (int i = 0; i < rows * cols; ++i) m_storage.emplace_back();
  }
  VariableMatrix(detail::empty_t, int rows, int cols)
      : m_rows{rows}, m_cols{cols}, m_storage(static_cast<size_t>(rows) * cols, Variable{nullptr}) {}
  VariableMatrix(std::initializer_list<std::initializer_list<Variable>> list) {
    m_rows = static_cast<int>(list.size());
    m_cols = m_rows ? static_cast<int>(list.begin()->size()) : 0;
    for (auto& row : list) {
      assert(static_cast<int>(row.size()) == m_cols);
      for (auto& v : row) m_storage.push_back(v);
    }
  }
  VariableMatrix(const DenseMatrix& values)  // NOLINT
      : m_rows{values.rows()}, m_cols{values.cols()} {
    for (int r = 0; r < m_rows; ++r)
      for (int c = 0; c < m_cols; ++c) m_storage.emplace_back(values[r, c]);
  }
  VariableMatrix(const Variable& v) : m_rows{1}, m_cols{1}, m_storage{v} {}  // NOLINT
  VariableMatrix(const VariableBlock& b);                                     // NOLINT
  explicit VariableMatrix(const std::vector<Variable>& values)
      : m_rows{static_cast<int>(values.size())}, m_cols{1}, m_storage{values} {}

  Variable& operator[](int row, int col) {
    assert(row >= 0 && row < m_rows && col >= 0 && col < m_cols);
    return m_storage[static_cast<size_t>(row) * m_cols + col];
  }
  const Variable& operator[](int row, int col) const {
    assert(row >= 0 && row < m_rows && col >= 0 && col < m_cols);
    return m_storage[static_cast<size_t>(row) * m_cols + col];
  }
  Variable& operator[](int index) { return m_storage[index]; }
  const Variable& operator[](int index) const { return m_storage[index]; }
  Variable& operator()(int row, int col) { return (*this)[row, col]; }
  const Variable& operator()(int row, int col) const { return (*this)[row, col]; }

  VariableBlock block(int row_offset, int col_offset, int block_rows, int block_cols);
  VariableMatrix block(int row_offset, int col_offset, int block_rows, int block_cols) const;
  VariableBlock segment(int offset, int length);
  VariableMatrix segment(int offset, int length) const;
  VariableBlock row(int row);
  VariableMatrix row(int row) const;
  VariableBlock col(int col);
  VariableMatrix col(int col) const;

  VariableMatrix T() const {
    VariableMatrix result{detail::empty, m_cols, m_rows};
    for (int r = 0; r < m_rows; ++r)
      for (int c = 0; c < m_cols; ++c) result[c, r] = (*this)[r, c];
    return result;
  }

  int rows() const { return m_rows; }
  int cols() const { return m_cols; }
  int size() const { return m_rows * m_cols; }

  double value(int row, int col) const { return (*this)[row, col].value(); }
  double value(int index) const { return (*this)[index].value(); }
  DenseMatrix value() const {
    DenseMatrix result{m_rows, m_cols};
    for (int r = 0; r < m_rows; ++r)
      for (int c = 0; c < m_cols; ++c) result[r, c] = value(r, c);
    return result;
  }
  void set_value(const DenseMatrix& values) {
    assert(values.rows() == m_rows && values.cols() == m_cols);
    for (int r = 0; r < m_rows; ++r)
      for (int c = 0; c < m_cols; ++c) (*this)[r, c].set_value(values[r, c]);
  }

  auto begin() { return m_storage.begin(); }
  auto end() { return m_storage.end(); }
  auto begin() const { return m_storage.begin(); }
  auto end() const { return m_storage.end(); }

  VariableMatrix& operator+=(const VariableMatrix& rhs);
  VariableMatrix& operator-=(const VariableMatrix& rhs);
  VariableMatrix& operator*=(const Variable& rhs) {
    for (auto& v : m_storage) v *= rhs;
    return *this;
  }
  VariableMatrix& operator/=(const Variable& rhs) {
    for (auto& v : m_storage) v /= rhs;
    return *this;
  }

 private:
  int m_rows = 0, m_cols = 0;
  std::vector<Variable> m_storage;
};

// Mutable view into a VariableMatrix (variable_block.hpp:27).  Assigning a matrix
// re-points the viewed handles, as in the reference.
class VariableBlock {
 public:
  VariableBlock(VariableMatrix& mat, int row_offset, int col_offset, int rows, int cols)
      : m_mat{&mat}, m_r0{row_offset}, m_c0{col_offset}, m_rows{rows}, m_cols{cols} {}

  VariableBlock& operator=(const VariableMatrix& values) {
    assert(values.rows() == m_rows && values.cols() == m_cols);
    for (int r = 0; r < m_rows; ++r)
      for (int c = 0; c < m_cols; ++c) (*this)[r, c] = values[r, c];
    return *this;
  }
  VariableBlock& operator=(const VariableBlock& values) {
    if (this == &values) return *this;
    return *this = VariableMatrix{values};
  }
  VariableBlock(const VariableBlock&) = default;
  VariableBlock& operator=(const DenseMatrix& values) { return *this = VariableMatrix{values}; }
  VariableBlock& operator=(double value) {
    assert(m_rows == 1 && m_cols == 1);
    (*this)[0, 0] = Variable{value};
    return *this;
  }

  Variable& operator[](int row, int col) const { return (*m_mat)[m_r0 + row, m_c0 + col]; }
  Variable& operator[](int index) const { return (*this)[index / m_cols, index % m_cols]; }
  Variable& operator()(int row, int col) const { return (*this)[row, col]; }
  int rows() const { return m_rows; }
  int cols() const { return m_cols; }

  VariableBlock block(int r0, int c0, int rows, int cols) const {
    return VariableBlock{*m_mat, m_r0 + r0, m_c0 + c0, rows, cols};
  }
  VariableBlock segment(int offset, int length) const {
    return m_cols == 1 ? block(offset, 0, length, 1) : block(0, offset, 1, length);
  }
  VariableBlock row(int r) const { return block(r, 0, 1, m_cols); }
  VariableBlock col(int c) const { return block(0, c, m_rows, 1); }
  VariableMatrix T() const { return VariableMatrix{*this}.T(); }

  double value(int row, int col) const { return (*this)[row, col].value(); }
  double value(int index) const { return (*this)[index].value(); }
  DenseMatrix value() const { return VariableMatrix{*this}.value(); }
  void set_value(const DenseMatrix& values) {
    for (int r = 0; r < m_rows; ++r)
      for (int c = 0; c < m_cols; ++c) (*this)[r, c].set_value(values[r, c]);
  }
  void set_value(double value) {
    assert(m_rows == 1 && m_cols == 1);
    (*this)[0, 0].set_value(value);
  }

 private:
  VariableMatrix* m_mat;
  int m_r0, m_c0, m_rows, m_cols;
};

inline Variable::Variable(const VariableMatrix& value) : expr{value[0, 0].expr} {
  assert(value.rows() == 1 && value.cols() == 1);
}
inline Variable::Variable(const VariableBlock& value) : expr{value[0, 0].expr} {
  assert(value.rows() == 1 && value.cols() == 1);
}
inline VariableMatrix::VariableMatrix(const VariableBlock& b) : m_rows{b.rows()}, m_cols{b.cols()} {
  m_storage.reserve(static_cast<size_t>(m_rows) * m_cols);
  for (int r = 0; r < m_rows; ++r)
    for (int c = 0; c < m_cols; ++c) m_storage.push_back(b[r, c]);
}
inline VariableBlock VariableMatrix::block(int r0, int c0, int rows, int cols) {
  return VariableBlock{*this, r0, c0, rows, cols};
}
inline VariableMatrix VariableMatrix::block(int r0, int c0, int rows, int cols) const {
  VariableMatrix m{detail::empty, rows, cols};
  for (int r = 0; r < rows; ++r)
    for (int c = 0; c < cols; ++c) m[r, c] = (*this)[r0 + r, c0 + c];
  return m;
}
inline VariableBlock VariableMatrix::segment(int offset, int length) {
  return m_cols == 1 ? block(offset, 0, length, 1) : block(0, offset, 1, length);
}
inline VariableMatrix VariableMatrix::segment(int offset, int length) const {
  return m_cols == 1 ? block(offset, 0, length, 1) : block(0, offset, 1, length);
}
inline VariableBlock VariableMatrix::row(int r) { return block(r, 0, 1, m_cols); }
inline VariableMatrix VariableMatrix::row(int r) const { return block(r, 0, 1, m_cols); }
inline VariableBlock VariableMatrix::col(int c) { return block(0, c, m_rows, 1); }
inline VariableMatrix VariableMatrix::col(int c) const { return block(0, c, m_rows, 1); }

// ---- matrix arithmetic ---------------------------------------------------------
namespace detail {
template <typename L, typename R>
VariableMatrix matmul(const L& lhs, const R& rhs) {
  assert(lhs.cols() == rhs.rows());
  VariableMatrix result{empty, lhs.rows(), rhs.cols()};
  for (int i = 0; i < lhs.rows(); ++i)
    for (int j = 0; j < rhs.cols(); ++j) {
      Variable sum{0.0};
      for (int k = 0; k < lhs.cols(); ++k) sum += Variable{lhs[i, k]} * Variable{rhs[k, j]};
      result[i, j] = sum;
    }
  return result;
}
template <typename L, typename R, typename F>
VariableMatrix cwise(const L& lhs, const R& rhs, F&& f) {
  assert(lhs.rows() == rhs.rows() && lhs.cols() == rhs.cols());
  VariableMatrix result{empty, lhs.rows(), lhs.cols()};
  for (int r = 0; r < lhs.rows(); ++r)
    for (int c = 0; c < lhs.cols(); ++c) result[r, c] = f(Variable{lhs[r, c]}, Variable{rhs[r, c]});
  return result;
}
template <typename M, typename F>
VariableMatrix cwise1(const M& m, F&& f) {
  VariableMatrix result{empty, m.rows(), m.cols()};
  for (int r = 0; r < m.rows(); ++r)
    for (int c = 0; c < m.cols(); ++c) result[r, c] = f(Variable{m[r, c]});
  return result;
}
}  // namespace detail

inline VariableMatrix operator*(const VariableMatrix& l, const VariableMatrix& r) { return detail::matmul(l, r); }
inline VariableMatrix operator*(const DenseMatrix& l, const VariableMatrix& r) { return detail::matmul(l, r); }
inline VariableMatrix operator*(const VariableMatrix& l, const DenseMatrix& r) { return detail::matmul(l, r); }
// matrix (x) scalar: element on the LEFT in both argument orders (variable_matrix.hpp:592-640)
inline VariableMatrix operator*(const VariableMatrix& l, const Variable& r) {
  return detail::cwise1(l, [&](const Variable& e) { return e * r; });
}
inline VariableMatrix operator*(const Variable& l, const VariableMatrix& r) {
  return detail::cwise1(r, [&](const Variable& e) { return e * l; });
}
inline VariableMatrix operator*(const VariableMatrix& l, double r) { return l * Variable{r}; }
inline VariableMatrix operator*(double l, const VariableMatrix& r) { return Variable{l} * r; }
inline VariableMatrix operator*(const DenseMatrix& l, const Variable& r) {
  return detail::cwise1(l, [&](const Variable& e) { return e * r; });
}
inline VariableMatrix operator*(const Variable& l, const DenseMatrix& r) {
  return detail::cwise1(r, [&](const Variable& e) { return e * l; });
}
inline VariableMatrix operator/(const VariableMatrix& l, const Variable& r) {
  return detail::cwise1(l, [&](const Variable& e) { return e / r; });
}
inline VariableMatrix operator/(const VariableMatrix& l, double r) { return l / Variable{r}; }
inline VariableMatrix operator+(const VariableMatrix& l, const VariableMatrix& r) {
  return detail::cwise(l, r, [](const Variable& a, const Variable& b) { return a + b; });
}
inline VariableMatrix operator+(const DenseMatrix& l, const VariableMatrix& r) {
  return detail::cwise(l, r, [](const Variable& a, const Variable& b) { return a + b; });
}
inline VariableMatrix operator+(const VariableMatrix& l, const DenseMatrix& r) {
  return detail::cwise(l, r, [](const Variable& a, const Variable& b) { return a + b; });
}
inline VariableMatrix operator-(const VariableMatrix& l, const VariableMatrix& r) {
  return detail::cwise(l, r, [](const Variable& a, const Variable& b) { return a - b; });
}
inline VariableMatrix operator-(const DenseMatrix& l, const VariableMatrix& r) {
  return detail::cwise(l, r, [](const Variable& a, const Variable& b) { return a - b; });
}
inline VariableMatrix operator-(const VariableMatrix& l, const DenseMatrix& r) {
  return detail::cwise(l, r, [](const Variable& a, const Variable& b) { return a - b; });
}
inline VariableMatrix operator-(const VariableMatrix& m) {
  return detail::cwise1(m, [](const Variable& e) { return -e; });
}
inline VariableMatrix& VariableMatrix::operator+=(const VariableMatrix& rhs) {
  assert(m_rows == rhs.rows() && m_cols == rhs.cols());
  for (int i = 0; i < size(); ++i) m_storage[i] += rhs[i];
  return *this;
}
inline VariableMatrix& VariableMatrix::operator-=(const VariableMatrix& rhs) {
  assert(m_rows == rhs.rows() && m_cols == rhs.cols());
  for (int i = 0; i < size(); ++i) m_storage[i] -= rhs[i];
  return *this;
}
// VariableBlock operands convert to VariableMatrix
inline VariableMatrix operator*(const VariableBlock& l, const VariableBlock& r) { return VariableMatrix{l} * VariableMatrix{r}; }
inline VariableMatrix operator*(const VariableMatrix& l, const VariableBlock& r) { return l * VariableMatrix{r}; }
inline VariableMatrix operator*(const VariableBlock& l, const VariableMatrix& r) { return VariableMatrix{l} * r; }
inline VariableMatrix operator*(const DenseMatrix& l, const VariableBlock& r) { return l * VariableMatrix{r}; }
inline VariableMatrix operator*(double l, const VariableBlock& r) { return l * VariableMatrix{r}; }
inline VariableMatrix operator*(const VariableBlock& l, double r) { return VariableMatrix{l} * r; }
inline VariableMatrix operator+(const VariableBlock& l, const VariableMatrix& r) { return VariableMatrix{l} + r; }
inline VariableMatrix operator+(const VariableMatrix& l, const VariableBlock& r) { return l + VariableMatrix{r}; }
inline VariableMatrix operator+(const VariableBlock& l, const VariableBlock& r) { return VariableMatrix{l} + VariableMatrix{r}; }
inline VariableMatrix operator-(const VariableBlock& l, const VariableMatrix& r) { return VariableMatrix{l} - r; }
inline VariableMatrix operator-(const VariableMatrix& l, const VariableBlock& r) { return l - VariableMatrix{r}; }
inline VariableMatrix operator-(const VariableBlock& l, const VariableBlock& r) { return VariableMatrix{l} - VariableMatrix{r}; }
inline VariableMatrix operator-(const DenseMatrix& l, const VariableBlock& r) { return l - VariableMatrix{r}; }
inline VariableMatrix operator-(const VariableBlock& l, const DenseMatrix& r) { return VariableMatrix{l} - r; }
inline VariableMatrix operator-(const VariableBlock& m) { return -VariableMatrix{m}; }

// variable_matrix.hpp:1516-1620
inline VariableMatrix solve(const VariableMatrix& A, const VariableMatrix& B) {
  assert(A.rows() == B.rows());
  if (A.rows() == 1 && A.cols() == 1) {
    return VariableMatrix{B[0, 0] / A[0, 0]};
  } else if (A.rows() == 2 && A.cols() == 2) {
    const auto& a = A[0, 0];
    const auto& b = A[0, 1];
    const auto& c = A[1, 0];
    const auto& d = A[1, 1];
    VariableMatrix adj_A{{d, -b}, {-c, a}};
    auto det_A = a * d - b * c;
    return adj_A / det_A * B;
  } else if (A.rows() == 3 && A.cols() == 3) {
    const auto& a = A[0, 0];
    const auto& b = A[0, 1];
    const auto& c = A[0, 2];
    const auto& d = A[1, 0];
    const auto& e = A[1, 1];
    const auto& f = A[1, 2];
    const auto& g = A[2, 0];
    const auto& h = A[2, 1];
    const auto& i = A[2, 2];
    auto ae = a * e; auto af = a * f; auto ah = a * h; auto ai = a * i;
    auto bd = b * d; auto bf = b * f; auto bg = b * g; auto bi = b * i;
    auto cd = c * d; auto ce = c * e; auto cg = c * g; auto ch = c * h;
    auto dh = d * h; auto di = d * i; auto eg = e * g; auto ei = e * i;
    auto fg = f * g; auto fh = f * h;
    auto adj_A00 = ei - fh;
    auto adj_A10 = fg - di;
    auto adj_A20 = dh - eg;
    VariableMatrix adj_A{{adj_A00, ch - bi, bf - ce}, {adj_A10, ai - cg, cd - af}, {adj_A20, bg - ah, ae - bd}};
    auto det_A = a * adj_A00 + b * adj_A10 + c * adj_A20;
    return adj_A / det_A * B;
  }
  assert(false && "solve(): only 1x1, 2x2 and 3x3 systems are built in this round");
  return {};
}

// ---- constraints (variable.hpp:716-1013) ------------------------------------------
namespace detail {
template <typename T>
concept ScalarOperand = std::is_arithmetic_v<std::decay_t<T>> || std::same_as<std::decay_t<T>, Variable>;
template <typename T>
concept MatrixOperand = std::same_as<std::decay_t<T>, VariableMatrix> || std::same_as<std::decay_t<T>, VariableBlock> ||
                        std::same_as<std::decay_t<T>, DenseMatrix>;
template <typename T>
concept SleipnirOperand = std::same_as<std::decay_t<T>, Variable> || std::same_as<std::decay_t<T>, VariableMatrix> ||
                          std::same_as<std::decay_t<T>, VariableBlock>;

template <typename L, typename R>
std::vector<Variable> make_constraints(const L& lhs, const R& rhs) {
  std::vector<Variable> out;
  if constexpr (ScalarOperand<L> && ScalarOperand<R>) {
    out.push_back(Variable{lhs} - Variable{rhs});
  } else if constexpr (ScalarOperand<L>) {
    for (int r = 0; r < rhs.rows(); ++r)
      for (int c = 0; c < rhs.cols(); ++c) out.push_back(Variable{lhs} - Variable{rhs[r, c]});
  } else if constexpr (ScalarOperand<R>) {
    for (int r = 0; r < lhs.rows(); ++r)
      for (int c = 0; c < lhs.cols(); ++c) out.push_back(Variable{lhs[r, c]} - Variable{rhs});
  } else {
    assert(lhs.rows() == rhs.rows() && lhs.cols() == rhs.cols());
    for (int r = 0; r < lhs.rows(); ++r)
      for (int c = 0; c < lhs.cols(); ++c) out.push_back(Variable{lhs[r, c]} - Variable{rhs[r, c]});
  }
  return out;
}
}  // namespace detail

struct EqualityConstraints {
  std::vector<Variable> constraints;
  EqualityConstraints() = default;
  EqualityConstraints(std::initializer_list<EqualityConstraints> list) {
    for (auto& e : list) constraints.insert(constraints.end(), e.constraints.begin(), e.constraints.end());
  }
  explicit EqualityConstraints(std::vector<Variable> c) : constraints{std::move(c)} {}
  explicit operator bool() const {
    for (auto& c : constraints)
      if (c.value() != 0.0) return false;
    return true;
  }
};
struct InequalityConstraints {
  std::vector<Variable> constraints;
  InequalityConstraints() = default;
  InequalityConstraints(std::initializer_list<InequalityConstraints> list) {
    for (auto& e : list) constraints.insert(constraints.end(), e.constraints.begin(), e.constraints.end());
  }
  explicit InequalityConstraints(std::vector<Variable> c) : constraints{std::move(c)} {}
  explicit operator bool() const {
    for (auto& c : constraints)
      if (!(c.value() >= 0.0)) return false;
    return true;
  }
};

template <typename L, typename R>
  requires(detail::ScalarOperand<L> || detail::MatrixOperand<L>) && (detail::ScalarOperand<R> || detail::MatrixOperand<R>) &&
          (detail::SleipnirOperand<L> || detail::SleipnirOperand<R>)
EqualityConstraints operator==(const L& lhs, const R& rhs) {
  return EqualityConstraints{detail::make_constraints(lhs, rhs)};
}
template <typename L, typename R>
  requires(detail::ScalarOperand<L> || detail::MatrixOperand<L>) && (detail::ScalarOperand<R> || detail::MatrixOperand<R>) &&
          (detail::SleipnirOperand<L> || detail::SleipnirOperand<R>)
InequalityConstraints operator>=(const L& lhs, const R& rhs) {
  return InequalityConstraints{detail::make_constraints(lhs, rhs)};
}
template <typename L, typename R>
  requires(detail::ScalarOperand<L> || detail::MatrixOperand<L>) && (detail::ScalarOperand<R> || detail::MatrixOperand<R>) &&
          (detail::SleipnirOperand<L> || detail::SleipnirOperand<R>)
InequalityConstraints operator<=(const L& lhs, const R& rhs) {
  return rhs >= lhs;
}
template <typename L, typename R>
  requires(detail::ScalarOperand<L> || detail::MatrixOperand<L>) && (detail::ScalarOperand<R> || detail::MatrixOperand<R>) &&
          (detail::SleipnirOperand<L> || detail::SleipnirOperand<R>)
InequalityConstraints operator>(const L& lhs, const R& rhs) {
  return lhs >= rhs;
}
template <typename L, typename R>
  requires(detail::ScalarOperand<L> || detail::MatrixOperand<L>) && (detail::ScalarOperand<R> || detail::MatrixOperand<R>) &&
          (detail::SleipnirOperand<L> || detail::SleipnirOperand<R>)
InequalityConstraints operator<(const L& lhs, const R& rhs) {
  return rhs >= lhs;
}

template <typename L, typename X, typename U>
InequalityConstraints bounds(const L& l, const X& x, const U& u) {
  return InequalityConstraints{l <= x, x <= u};
}

}  // namespace slp
