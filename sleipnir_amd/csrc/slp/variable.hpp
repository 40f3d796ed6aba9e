// slp::VariableF64 / slp::VariableMatrixF64 / slp::VariableBlockF64 — the modelling surface
// of the reference kept source-compatible for the path's callers
// (include/sleipnir/autodiff/variable.hpp, variable_matrix.hpp, variable_block.hpp),
// but recording into the flat SoA arena of graph.hpp instead of a pointer graph.
//
// Kept semantics (cited where they matter for graph identity):
//   * default-constructed VariableF64 = decision variable with value 0 (variable.hpp:289-290)
//   * scalar (x) matrix builds `element * scalar` (variable_matrix.hpp:592-640)
//   * matmul accumulates `sum{0}; sum += a*b` (variable_matrix.hpp:505-557)
//   * lhs ? rhs builds rows `lhs - rhs`, row-major (variable.hpp:716-778)
//   * bounds(l, x, u) = {l <= x, x <= u} (variable.hpp:1008-1013)
//   * closed-form solve() for 1x1..3x3 (variable_matrix.hpp:1516-1620)
// Eigen is not available in this toolchain; slp::DenseMatrix stands in for the
// Eigen::Matrix<double,...> constants that appear in problem definitions.
#pragma once

#include <algorithm>
#include <cassert>
#include <cmath>
#include <concepts>
#include <functional>
#include <initializer_list>
#include <iterator>
#include <limits>
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

class VariableMatrixF64;
class VariableBlockF64;

class VariableF64 {
 public:
  using Scalar = double;

  VariableF64() : expr{detail::G().variable(0.0)} {}
  explicit constexpr VariableF64(std::nullptr_t) : expr{slpx::kNull} {}
  VariableF64(double value) : expr{detail::G().constant(value)} {}  // NOLINT
  VariableF64(std::integral auto value) : expr{detail::G().constant(static_cast<double>(value))} {}  // NOLINT
  VariableF64(const VariableMatrixF64& value);  // NOLINT (1x1 only)
  VariableF64(const VariableBlockF64& value);   // NOLINT (1x1 only)
  struct from_node_t {};
  VariableF64(from_node_t, NodeId n) : expr{n} {}

  VariableF64& operator=(double value) {
    expr = detail::G().constant(value);
    return *this;
  }

  // variable.hpp:125-138
  void set_value(double value) { detail::G().val[expr] = value; }
  // variable.hpp:143-151
  double value() const { return detail::G().value(expr); }
  ExpressionType type() const { return static_cast<ExpressionType>(detail::G().type[expr]); }

  friend VariableF64 operator*(const VariableF64& l, const VariableF64& r) { return wrap(detail::G().mul(l.expr, r.expr)); }
  friend VariableF64 operator/(const VariableF64& l, const VariableF64& r) { return wrap(detail::G().div(l.expr, r.expr)); }
  friend VariableF64 operator+(const VariableF64& l, const VariableF64& r) { return wrap(detail::G().add(l.expr, r.expr)); }
  friend VariableF64 operator-(const VariableF64& l, const VariableF64& r) { return wrap(detail::G().sub(l.expr, r.expr)); }
  friend VariableF64 operator-(const VariableF64& l) { return wrap(detail::G().neg(l.expr)); }
  friend VariableF64 operator+(const VariableF64& l) { return l; }
  // arithmetic-with-scalar overloads: exact matches, so `0.5 * x` never competes
  // with the VariableF64<->VariableMatrixF64 conversions (variable.hpp:157-167)
  friend VariableF64 operator*(double l, const VariableF64& r) { return VariableF64{l} * r; }
  friend VariableF64 operator*(const VariableF64& l, double r) { return l * VariableF64{r}; }
  friend VariableF64 operator/(double l, const VariableF64& r) { return VariableF64{l} / r; }
  friend VariableF64 operator/(const VariableF64& l, double r) { return l / VariableF64{r}; }
  friend VariableF64 operator+(double l, const VariableF64& r) { return VariableF64{l} + r; }
  friend VariableF64 operator+(const VariableF64& l, double r) { return l + VariableF64{r}; }
  friend VariableF64 operator-(double l, const VariableF64& r) { return VariableF64{l} - r; }
  friend VariableF64 operator-(const VariableF64& l, double r) { return l - VariableF64{r}; }
  VariableF64& operator*=(const VariableF64& r) { return *this = *this * r; }
  VariableF64& operator/=(const VariableF64& r) { return *this = *this / r; }
  VariableF64& operator+=(const VariableF64& r) { return *this = *this + r; }
  VariableF64& operator-=(const VariableF64& r) { return *this = *this - r; }

  static VariableF64 wrap(NodeId n) { return VariableF64{from_node_t{}, n}; }

  NodeId expr;
};

#define SLP_UNARY(name, OP) \
  inline VariableF64 name(const VariableF64& x) { return VariableF64::wrap(detail::G().unary(slpx::OP, x.expr)); }
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
  inline VariableF64 name(const VariableF64& a, const VariableF64& b) {   \
    return VariableF64::wrap(detail::G().binary(slpx::OP, a.expr, b.expr)); \
  }
SLP_BINARY(atan2, OP_ATAN2)
SLP_BINARY(hypot, OP_HYPOT)
SLP_BINARY(max, OP_MAX)
SLP_BINARY(min, OP_MIN)
SLP_BINARY(pow, OP_POW)
#undef SLP_BINARY
// variable.hpp:711-714
inline VariableF64 hypot(const VariableF64& x, const VariableF64& y, const VariableF64& z) {
  return sqrt(pow(x, 2) + pow(y, 2) + pow(z, 2));
}

// slice.hpp:14-160: Python's slice — start, stop, step, any of them `_` (slicing::none_t) —
// with adjust(length) doing what slice.indices() does and returning the number of elements.
namespace slicing {
struct none_t {};
inline constexpr none_t _;
}  // namespace slicing

class Slice {
 public:
  int start = 0, stop = 0, step = 1;

  constexpr Slice() = default;
  constexpr Slice(slicing::none_t) : start{0}, stop{kMax}, step{1} {}  // NOLINT: everything
  // one element; -1 is the last one, whose successor is not 0 but the end
  constexpr Slice(int index) : start{index}, stop{index == -1 ? kMax : index + 1}, step{1} {}  // NOLINT

  template <typename Start, typename Stop>
    requires(std::same_as<Start, slicing::none_t> || std::convertible_to<Start, int>) &&
            (std::same_as<Stop, slicing::none_t> || std::convertible_to<Stop, int>)
  constexpr Slice(Start from, Stop to) : Slice(from, to, 1) {}

  template <typename Start, typename Stop, typename Step>
    requires(std::same_as<Start, slicing::none_t> || std::convertible_to<Start, int>) &&
            (std::same_as<Stop, slicing::none_t> || std::convertible_to<Stop, int>) &&
            (std::same_as<Step, slicing::none_t> || std::convertible_to<Step, int>)
  constexpr Slice(Start from, Stop to, Step by) {
    if constexpr (!std::same_as<Step, slicing::none_t>) {
      assert(by != 0);
      step = by == std::numeric_limits<int>::min() ? -kMax : static_cast<int>(by);  // -step must exist
    }
    // an open end is "as far as it goes" in the direction of travel
    if constexpr (std::same_as<Start, slicing::none_t>) start = step < 0 ? kMax : 0;
    else start = from;
    if constexpr (std::same_as<Stop, slicing::none_t>) stop = step < 0 ? std::numeric_limits<int>::min() : kMax;
    else stop = to;
  }

  // clamp to a sequence of `length` elements (negative indices count from the end); returns how
  // many elements the slice selects
  constexpr int adjust(int length) {
    assert(step != 0);
    auto clamp = [&](int i) {
      if (i < 0) {
        i += length;
        return i < 0 ? (step < 0 ? -1 : 0) : i;
      }
      return i >= length ? (step < 0 ? length - 1 : length) : i;
    };
    start = clamp(start);
    stop = clamp(stop);
    if (step < 0) return stop < start ? (start - stop - 1) / -step + 1 : 0;
    return start < stop ? (stop - start - 1) / step + 1 : 0;
  }

 private:
  static constexpr int kMax = std::numeric_limits<int>::max();
};

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
  friend bool operator==(const DenseMatrix& a, const DenseMatrix& b) {
    return a.m_rows == b.m_rows && a.m_cols == b.m_cols && a.m_d == b.m_d;
  }

 private:
  int m_rows = 0, m_cols = 0;
  std::vector<double> m_d;
};

class VariableMatrixF64 {
 public:
  using Scalar = double;
  VariableMatrixF64() = default;
  explicit VariableMatrixF64(int rows) : VariableMatrixF64(rows, 1) {}
  // variable_matrix.hpp:47-58: filled with default (decision-variable) handles
  VariableMatrixF64(int rows, int cols) : m_rows{rows}, m_cols{cols} {
    m_storage.reserve(static_cast<size_t>(rows) * cols);
    for (int i = 0; i < rows * cols; ++i) m_storage.emplace_back();
  }
  VariableMatrixF64(detail::empty_t, int rows, int cols)
      : m_rows{rows}, m_cols{cols}, m_storage(static_cast<size_t>(rows) * cols, VariableF64{nullptr}) {}
  VariableMatrixF64(std::initializer_list<std::initializer_list<VariableF64>> list) {
    m_rows = static_cast<int>(list.size());
    m_cols = m_rows ? static_cast<int>(list.begin()->size()) : 0;
    for (auto& row : list) {
      assert(static_cast<int>(row.size()) == m_cols);
      for (auto& v : row) m_storage.push_back(v);
    }
  }
  VariableMatrixF64(const DenseMatrix& values)  // NOLINT
      : m_rows{values.rows()}, m_cols{values.cols()} {
    for (int r = 0; r < m_rows; ++r)
      for (int c = 0; c < m_cols; ++c) m_storage.emplace_back(values[r, c]);
  }
  VariableMatrixF64(const VariableF64& v) : m_rows{1}, m_cols{1}, m_storage{v} {}  // NOLINT
  VariableMatrixF64(const VariableBlockF64& b);                                     // NOLINT
  explicit VariableMatrixF64(const std::vector<VariableF64>& values)
      : m_rows{static_cast<int>(values.size())}, m_cols{1}, m_storage{values} {}

  VariableF64& operator[](int row, int col) {
    assert(row >= 0 && row < m_rows && col >= 0 && col < m_cols);
    return m_storage[static_cast<size_t>(row) * m_cols + col];
  }
  const VariableF64& operator[](int row, int col) const {
    assert(row >= 0 && row < m_rows && col >= 0 && col < m_cols);
    return m_storage[static_cast<size_t>(row) * m_cols + col];
  }
  VariableF64& operator[](int index) { return m_storage[index]; }
  const VariableF64& operator[](int index) const { return m_storage[index]; }
  VariableF64& operator()(int row, int col) { return (*this)[row, col]; }
  const VariableF64& operator()(int row, int col) const { return (*this)[row, col]; }
  // variable_matrix.hpp:330-400: a strided view / copy
  VariableBlockF64 operator[](Slice row_slice, Slice col_slice);
  VariableMatrixF64 operator[](Slice row_slice, Slice col_slice) const;

  VariableBlockF64 block(int row_offset, int col_offset, int block_rows, int block_cols);
  VariableMatrixF64 block(int row_offset, int col_offset, int block_rows, int block_cols) const;
  VariableBlockF64 segment(int offset, int length);
  VariableMatrixF64 segment(int offset, int length) const;
  VariableBlockF64 row(int row);
  VariableMatrixF64 row(int row) const;
  VariableBlockF64 col(int col);
  VariableMatrixF64 col(int col) const;

  VariableMatrixF64 T() const {
    VariableMatrixF64 result{detail::empty, m_cols, m_rows};
    for (int r = 0; r < m_rows; ++r)
      for (int c = 0; c < m_cols; ++c) result[c, r] = (*this)[r, c];
    return result;
  }

  // variable_matrix.hpp:1044-1098: the matrix exponential, (13, 13) Padé approximant
  VariableMatrixF64 exp() const;

  // variable_matrix.hpp:1027-1039
  VariableMatrixF64 cwise_transform(const std::function<VariableF64(const VariableF64&)>& unary_op) const {
    VariableMatrixF64 result{detail::empty, m_rows, m_cols};
    for (int r = 0; r < m_rows; ++r)
      for (int c = 0; c < m_cols; ++c) result[r, c] = unary_op((*this)[r, c]);
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
  auto cbegin() const { return m_storage.cbegin(); }
  auto cend() const { return m_storage.cend(); }
  auto rbegin() { return m_storage.rbegin(); }
  auto rend() { return m_storage.rend(); }
  auto rbegin() const { return m_storage.rbegin(); }
  auto rend() const { return m_storage.rend(); }
  auto crbegin() const { return m_storage.crbegin(); }
  auto crend() const { return m_storage.crend(); }

  // variable_matrix.hpp:1100-1160
  static VariableMatrixF64 constant(int rows, int cols, double value) {
    VariableMatrixF64 m{detail::empty, rows, cols};
    for (auto& e : m.m_storage) e = VariableF64{value};
    return m;
  }
  static VariableMatrixF64 zero(int rows, int cols) { return constant(rows, cols, 0.0); }
  static VariableMatrixF64 one(int rows, int cols) { return constant(rows, cols, 1.0); }
  static VariableMatrixF64 identity(int rows) {
    VariableMatrixF64 m{detail::empty, rows, rows};
    for (int r = 0; r < rows; ++r)
      for (int c = 0; c < rows; ++c) m[r, c] = VariableF64{r == c ? 1.0 : 0.0};
    return m;
  }

  VariableMatrixF64& operator+=(const VariableMatrixF64& rhs);
  VariableMatrixF64& operator-=(const VariableMatrixF64& rhs);
  VariableMatrixF64& operator*=(const VariableMatrixF64& rhs);  // matrix product (variable_matrix.hpp:700-720)
  VariableMatrixF64& operator+=(const VariableF64& rhs) {
    for (auto& v : m_storage) v += rhs;
    return *this;
  }
  VariableMatrixF64& operator-=(const VariableF64& rhs) {
    for (auto& v : m_storage) v -= rhs;
    return *this;
  }
  VariableMatrixF64& operator*=(const VariableF64& rhs) {
    for (auto& v : m_storage) v *= rhs;
    return *this;
  }
  VariableMatrixF64& operator/=(const VariableF64& rhs) {
    for (auto& v : m_storage) v /= rhs;
    return *this;
  }

 private:
  int m_rows = 0, m_cols = 0;
  std::vector<VariableF64> m_storage;
};

// Mutable view into a VariableMatrixF64 (variable_block.hpp:27).  Assigning a matrix
// re-points the viewed handles, as in the reference.
class VariableBlockF64 {
 public:
  VariableBlockF64(VariableMatrixF64& mat, int row_offset, int col_offset, int rows, int cols, int row_step = 1,
                   int col_step = 1)
      : m_mat{&mat}, m_r0{row_offset}, m_c0{col_offset}, m_rows{rows}, m_cols{cols}, m_rstep{row_step}, m_cstep{col_step} {}

  VariableBlockF64& operator=(const VariableMatrixF64& values) {
    assert(values.rows() == m_rows && values.cols() == m_cols);
    for (int r = 0; r < m_rows; ++r)
      for (int c = 0; c < m_cols; ++c) (*this)[r, c] = values[r, c];
    return *this;
  }
  VariableBlockF64& operator=(const VariableBlockF64& values) {
    if (this == &values) return *this;
    return *this = VariableMatrixF64{values};
  }
  VariableBlockF64(const VariableBlockF64&) = default;
  VariableBlockF64& operator=(const DenseMatrix& values) { return *this = VariableMatrixF64{values}; }
  VariableBlockF64& operator=(double value) {
    assert(m_rows == 1 && m_cols == 1);
    (*this)[0, 0] = VariableF64{value};
    return *this;
  }

  VariableF64& operator[](int row, int col) const { return (*m_mat)[m_r0 + row * m_rstep, m_c0 + col * m_cstep]; }
  VariableF64& operator[](int index) const { return (*this)[index / m_cols, index % m_cols]; }
  // a slice of a slice: offsets add, steps multiply (variable_block.hpp)
  VariableBlockF64 operator[](Slice row_slice, Slice col_slice) const {
    const int rows = row_slice.adjust(m_rows), cols = col_slice.adjust(m_cols);
    return VariableBlockF64{*m_mat, m_r0 + row_slice.start * m_rstep, m_c0 + col_slice.start * m_cstep, rows, cols,
                            m_rstep * row_slice.step, m_cstep * col_slice.step};
  }
  VariableF64& operator()(int row, int col) const { return (*this)[row, col]; }
  int rows() const { return m_rows; }
  int cols() const { return m_cols; }

  VariableBlockF64 block(int r0, int c0, int rows, int cols) const {
    return VariableBlockF64{*m_mat, m_r0 + r0 * m_rstep, m_c0 + c0 * m_cstep, rows, cols, m_rstep, m_cstep};
  }
  VariableBlockF64 segment(int offset, int length) const {
    return (m_rows == 1 && m_cols != 1) ? block(0, offset, 1, length) : block(offset, 0, length, 1);
  }
  VariableBlockF64 row(int r) const { return block(r, 0, 1, m_cols); }
  VariableBlockF64 col(int c) const { return block(0, c, m_rows, 1); }
  VariableMatrixF64 T() const { return VariableMatrixF64{*this}.T(); }
  // variable_block.hpp: cwise_transform, exp
  VariableMatrixF64 cwise_transform(const std::function<VariableF64(const VariableF64&)>& unary_op) const;
  VariableMatrixF64 exp() const;

  double value(int row, int col) const { return (*this)[row, col].value(); }
  double value(int index) const { return (*this)[index].value(); }
  DenseMatrix value() const { return VariableMatrixF64{*this}.value(); }
  void set_value(const DenseMatrix& values) {
    for (int r = 0; r < m_rows; ++r)
      for (int c = 0; c < m_cols; ++c) (*this)[r, c].set_value(values[r, c]);
  }

  // variable_block.hpp: compound assignment writes through to the viewed matrix
  VariableBlockF64& operator+=(const VariableMatrixF64& rhs);
  VariableBlockF64& operator-=(const VariableMatrixF64& rhs);
  VariableBlockF64& operator*=(const VariableMatrixF64& rhs);  // matrix product
  VariableBlockF64& operator+=(const VariableF64& rhs);        // 1x1 blocks
  VariableBlockF64& operator-=(const VariableF64& rhs);
  VariableBlockF64& operator*=(const VariableF64& rhs) {
    for (int r = 0; r < m_rows; ++r)
      for (int c = 0; c < m_cols; ++c) (*this)[r, c] *= rhs;
    return *this;
  }
  VariableBlockF64& operator/=(const VariableF64& rhs) {
    for (int r = 0; r < m_rows; ++r)
      for (int c = 0; c < m_cols; ++c) (*this)[r, c] /= rhs;
    return *this;
  }

  // elements in row-major order of the view
  class iterator {
   public:
    using iterator_category = std::bidirectional_iterator_tag;
    using value_type = VariableF64;
    using difference_type = std::ptrdiff_t;
    using pointer = VariableF64*;
    using reference = VariableF64&;
    iterator() = default;
    iterator(const VariableBlockF64* block, int index) : m_block{block}, m_index{index} {}
    reference operator*() const { return (*m_block)[m_index]; }
    pointer operator->() const { return &(*m_block)[m_index]; }
    iterator& operator++() { ++m_index; return *this; }
    iterator operator++(int) { iterator old = *this; ++m_index; return old; }
    iterator& operator--() { --m_index; return *this; }
    iterator operator--(int) { iterator old = *this; --m_index; return old; }
    friend bool operator==(const iterator& a, const iterator& b) { return a.m_index == b.m_index; }

   private:
    const VariableBlockF64* m_block = nullptr;
    int m_index = 0;
  };
  iterator begin() const { return iterator{this, 0}; }
  iterator end() const { return iterator{this, m_rows * m_cols}; }
  iterator cbegin() const { return begin(); }
  iterator cend() const { return end(); }
  std::reverse_iterator<iterator> rbegin() const { return std::reverse_iterator<iterator>{end()}; }
  std::reverse_iterator<iterator> rend() const { return std::reverse_iterator<iterator>{begin()}; }
  std::reverse_iterator<iterator> crbegin() const { return rbegin(); }
  std::reverse_iterator<iterator> crend() const { return rend(); }
  void set_value(double value) {
    assert(m_rows == 1 && m_cols == 1);
    (*this)[0, 0].set_value(value);
  }

 private:
  VariableMatrixF64* m_mat;
  int m_r0, m_c0, m_rows, m_cols, m_rstep = 1, m_cstep = 1;
};

inline VariableF64::VariableF64(const VariableMatrixF64& value) : expr{value[0, 0].expr} {
  assert(value.rows() == 1 && value.cols() == 1);
}
inline VariableF64::VariableF64(const VariableBlockF64& value) : expr{value[0, 0].expr} {
  assert(value.rows() == 1 && value.cols() == 1);
}
inline VariableMatrixF64::VariableMatrixF64(const VariableBlockF64& b) : m_rows{b.rows()}, m_cols{b.cols()} {
  m_storage.reserve(static_cast<size_t>(m_rows) * m_cols);
  for (int r = 0; r < m_rows; ++r)
    for (int c = 0; c < m_cols; ++c) m_storage.push_back(b[r, c]);
}
inline VariableMatrixF64 VariableBlockF64::cwise_transform(
    const std::function<VariableF64(const VariableF64&)>& unary_op) const {
  return VariableMatrixF64{*this}.cwise_transform(unary_op);
}
// variable_matrix.hpp:1378-1395
inline VariableMatrixF64 cwise_reduce(const VariableMatrixF64& lhs, const VariableMatrixF64& rhs,
                                      const std::function<VariableF64(const VariableF64&, const VariableF64&)>& binary_op) {
  assert(lhs.rows() == rhs.rows() && lhs.cols() == rhs.cols());
  VariableMatrixF64 result{detail::empty, lhs.rows(), lhs.cols()};
  for (int r = 0; r < lhs.rows(); ++r)
    for (int c = 0; c < lhs.cols(); ++c) result[r, c] = binary_op(lhs[r, c], rhs[r, c]);
  return result;
}
inline VariableBlockF64 VariableMatrixF64::operator[](Slice row_slice, Slice col_slice) {
  const int rows = row_slice.adjust(m_rows), cols = col_slice.adjust(m_cols);
  return VariableBlockF64{*this, row_slice.start, col_slice.start, rows, cols, row_slice.step, col_slice.step};
}
inline VariableMatrixF64 VariableMatrixF64::operator[](Slice row_slice, Slice col_slice) const {
  const int rows = row_slice.adjust(m_rows), cols = col_slice.adjust(m_cols);
  VariableMatrixF64 m{detail::empty, rows, cols};
  for (int r = 0; r < rows; ++r)
    for (int c = 0; c < cols; ++c) m[r, c] = (*this)[row_slice.start + r * row_slice.step, col_slice.start + c * col_slice.step];
  return m;
}
inline VariableBlockF64 VariableMatrixF64::block(int r0, int c0, int rows, int cols) {
  return VariableBlockF64{*this, r0, c0, rows, cols};
}
inline VariableMatrixF64 VariableMatrixF64::block(int r0, int c0, int rows, int cols) const {
  VariableMatrixF64 m{detail::empty, rows, cols};
  for (int r = 0; r < rows; ++r)
    for (int c = 0; c < cols; ++c) m[r, c] = (*this)[r0 + r, c0 + c];
  return m;
}
inline VariableBlockF64 VariableMatrixF64::segment(int offset, int length) {
  return (m_rows == 1 && m_cols != 1) ? block(0, offset, 1, length) : block(offset, 0, length, 1);
}
inline VariableMatrixF64 VariableMatrixF64::segment(int offset, int length) const {
  return (m_rows == 1 && m_cols != 1) ? block(0, offset, 1, length) : block(offset, 0, length, 1);
}
inline VariableBlockF64 VariableMatrixF64::row(int r) { return block(r, 0, 1, m_cols); }
inline VariableMatrixF64 VariableMatrixF64::row(int r) const { return block(r, 0, 1, m_cols); }
inline VariableBlockF64 VariableMatrixF64::col(int c) { return block(0, c, m_rows, 1); }
inline VariableMatrixF64 VariableMatrixF64::col(int c) const { return block(0, c, m_rows, 1); }

// ---- matrix arithmetic ---------------------------------------------------------
namespace detail {
template <typename L, typename R>
VariableMatrixF64 matmul(const L& lhs, const R& rhs) {
  assert(lhs.cols() == rhs.rows());
  VariableMatrixF64 result{empty, lhs.rows(), rhs.cols()};
  for (int i = 0; i < lhs.rows(); ++i)
    for (int j = 0; j < rhs.cols(); ++j) {
      VariableF64 sum{0.0};
      for (int k = 0; k < lhs.cols(); ++k) sum += VariableF64{lhs[i, k]} * VariableF64{rhs[k, j]};
      result[i, j] = sum;
    }
  return result;
}
template <typename L, typename R, typename F>
VariableMatrixF64 cwise(const L& lhs, const R& rhs, F&& f) {
  assert(lhs.rows() == rhs.rows() && lhs.cols() == rhs.cols());
  VariableMatrixF64 result{empty, lhs.rows(), lhs.cols()};
  for (int r = 0; r < lhs.rows(); ++r)
    for (int c = 0; c < lhs.cols(); ++c) result[r, c] = f(VariableF64{lhs[r, c]}, VariableF64{rhs[r, c]});
  return result;
}
template <typename M, typename F>
VariableMatrixF64 cwise1(const M& m, F&& f) {
  VariableMatrixF64 result{empty, m.rows(), m.cols()};
  for (int r = 0; r < m.rows(); ++r)
    for (int c = 0; c < m.cols(); ++c) result[r, c] = f(VariableF64{m[r, c]});
  return result;
}
}  // namespace detail

inline VariableMatrixF64 operator*(const VariableMatrixF64& l, const VariableMatrixF64& r) { return detail::matmul(l, r); }
inline VariableMatrixF64 operator*(const DenseMatrix& l, const VariableMatrixF64& r) { return detail::matmul(l, r); }
inline VariableMatrixF64 operator*(const VariableMatrixF64& l, const DenseMatrix& r) { return detail::matmul(l, r); }
// matrix (x) scalar: element on the LEFT in both argument orders (variable_matrix.hpp:592-640)
inline VariableMatrixF64 operator*(const VariableMatrixF64& l, const VariableF64& r) {
  return detail::cwise1(l, [&](const VariableF64& e) { return e * r; });
}
inline VariableMatrixF64 operator*(const VariableF64& l, const VariableMatrixF64& r) {
  return detail::cwise1(r, [&](const VariableF64& e) { return e * l; });
}
inline VariableMatrixF64 operator*(const VariableMatrixF64& l, double r) { return l * VariableF64{r}; }
inline VariableMatrixF64 operator*(double l, const VariableMatrixF64& r) { return VariableF64{l} * r; }
inline VariableMatrixF64 operator*(const DenseMatrix& l, const VariableF64& r) {
  return detail::cwise1(l, [&](const VariableF64& e) { return e * r; });
}
inline VariableMatrixF64 operator*(const VariableF64& l, const DenseMatrix& r) {
  return detail::cwise1(r, [&](const VariableF64& e) { return e * l; });
}
inline VariableMatrixF64 operator/(const VariableMatrixF64& l, const VariableF64& r) {
  return detail::cwise1(l, [&](const VariableF64& e) { return e / r; });
}
inline VariableMatrixF64 operator/(const VariableMatrixF64& l, double r) { return l / VariableF64{r}; }
inline VariableMatrixF64 operator+(const VariableMatrixF64& l, const VariableMatrixF64& r) {
  return detail::cwise(l, r, [](const VariableF64& a, const VariableF64& b) { return a + b; });
}
inline VariableMatrixF64 operator+(const DenseMatrix& l, const VariableMatrixF64& r) {
  return detail::cwise(l, r, [](const VariableF64& a, const VariableF64& b) { return a + b; });
}
inline VariableMatrixF64 operator+(const VariableMatrixF64& l, const DenseMatrix& r) {
  return detail::cwise(l, r, [](const VariableF64& a, const VariableF64& b) { return a + b; });
}
inline VariableMatrixF64 operator-(const VariableMatrixF64& l, const VariableMatrixF64& r) {
  return detail::cwise(l, r, [](const VariableF64& a, const VariableF64& b) { return a - b; });
}
inline VariableMatrixF64 operator-(const DenseMatrix& l, const VariableMatrixF64& r) {
  return detail::cwise(l, r, [](const VariableF64& a, const VariableF64& b) { return a - b; });
}
inline VariableMatrixF64 operator-(const VariableMatrixF64& l, const DenseMatrix& r) {
  return detail::cwise(l, r, [](const VariableF64& a, const VariableF64& b) { return a - b; });
}
inline VariableMatrixF64 operator-(const VariableMatrixF64& m) {
  return detail::cwise1(m, [](const VariableF64& e) { return -e; });
}
inline VariableMatrixF64& VariableMatrixF64::operator+=(const VariableMatrixF64& rhs) {
  assert(m_rows == rhs.rows() && m_cols == rhs.cols());
  for (int i = 0; i < size(); ++i) m_storage[i] += rhs[i];
  return *this;
}
inline VariableMatrixF64& VariableMatrixF64::operator-=(const VariableMatrixF64& rhs) {
  assert(m_rows == rhs.rows() && m_cols == rhs.cols());
  for (int i = 0; i < size(); ++i) m_storage[i] -= rhs[i];
  return *this;
}
inline VariableMatrixF64& VariableMatrixF64::operator*=(const VariableMatrixF64& rhs) {
  return *this = detail::matmul(*this, rhs);
}
inline VariableBlockF64& VariableBlockF64::operator+=(const VariableMatrixF64& rhs) {
  assert(m_rows == rhs.rows() && m_cols == rhs.cols());
  for (int r = 0; r < m_rows; ++r)
    for (int c = 0; c < m_cols; ++c) (*this)[r, c] += rhs[r, c];
  return *this;
}
inline VariableBlockF64& VariableBlockF64::operator-=(const VariableMatrixF64& rhs) {
  assert(m_rows == rhs.rows() && m_cols == rhs.cols());
  for (int r = 0; r < m_rows; ++r)
    for (int c = 0; c < m_cols; ++c) (*this)[r, c] -= rhs[r, c];
  return *this;
}
inline VariableBlockF64& VariableBlockF64::operator*=(const VariableMatrixF64& rhs) {
  return *this = detail::matmul(VariableMatrixF64{*this}, rhs);
}
inline VariableBlockF64& VariableBlockF64::operator+=(const VariableF64& rhs) {
  for (int r = 0; r < m_rows; ++r)
    for (int c = 0; c < m_cols; ++c) (*this)[r, c] += rhs;
  return *this;
}
inline VariableBlockF64& VariableBlockF64::operator-=(const VariableF64& rhs) {
  for (int r = 0; r < m_rows; ++r)
    for (int c = 0; c < m_cols; ++c) (*this)[r, c] -= rhs;
  return *this;
}
// variable_matrix.hpp:1397-1470: [[A, B], [C]] -> one matrix
inline VariableMatrixF64 block(std::initializer_list<std::initializer_list<VariableMatrixF64>> list) {
  int rows = 0, cols = -1;
  for (const auto& row : list) {
    int h = -1, w = 0;
    for (const auto& m : row) {
      assert(h < 0 || h == m.rows());  // blocks of one row: same height
      h = m.rows();
      w += m.cols();
    }
    assert(cols < 0 || cols == w);  // block rows: same width
    cols = w;
    rows += std::max(h, 0);
  }
  VariableMatrixF64 result{detail::empty, rows, std::max(cols, 0)};
  int r0 = 0;
  for (const auto& row : list) {
    int c0 = 0, h = 0;
    for (const auto& m : row) {
      for (int r = 0; r < m.rows(); ++r)
        for (int c = 0; c < m.cols(); ++c) result[r0 + r, c0 + c] = m[r, c];
      c0 += m.cols();
      h = m.rows();
    }
    r0 += h;
  }
  return result;
}
// VariableBlockF64 operands convert to VariableMatrixF64
inline VariableMatrixF64 operator*(const VariableBlockF64& l, const VariableBlockF64& r) { return VariableMatrixF64{l} * VariableMatrixF64{r}; }
inline VariableMatrixF64 operator*(const VariableMatrixF64& l, const VariableBlockF64& r) { return l * VariableMatrixF64{r}; }
inline VariableMatrixF64 operator*(const VariableBlockF64& l, const VariableMatrixF64& r) { return VariableMatrixF64{l} * r; }
inline VariableMatrixF64 operator*(const DenseMatrix& l, const VariableBlockF64& r) { return l * VariableMatrixF64{r}; }
inline VariableMatrixF64 operator*(double l, const VariableBlockF64& r) { return l * VariableMatrixF64{r}; }
inline VariableMatrixF64 operator*(const VariableBlockF64& l, double r) { return VariableMatrixF64{l} * r; }
inline VariableMatrixF64 operator+(const VariableBlockF64& l, const VariableMatrixF64& r) { return VariableMatrixF64{l} + r; }
inline VariableMatrixF64 operator+(const VariableMatrixF64& l, const VariableBlockF64& r) { return l + VariableMatrixF64{r}; }
inline VariableMatrixF64 operator+(const VariableBlockF64& l, const VariableBlockF64& r) { return VariableMatrixF64{l} + VariableMatrixF64{r}; }
inline VariableMatrixF64 operator-(const VariableBlockF64& l, const VariableMatrixF64& r) { return VariableMatrixF64{l} - r; }
inline VariableMatrixF64 operator-(const VariableMatrixF64& l, const VariableBlockF64& r) { return l - VariableMatrixF64{r}; }
inline VariableMatrixF64 operator-(const VariableBlockF64& l, const VariableBlockF64& r) { return VariableMatrixF64{l} - VariableMatrixF64{r}; }
inline VariableMatrixF64 operator-(const DenseMatrix& l, const VariableBlockF64& r) { return l - VariableMatrixF64{r}; }
inline VariableMatrixF64 operator-(const VariableBlockF64& l, const DenseMatrix& r) { return VariableMatrixF64{l} - r; }
inline VariableMatrixF64 operator-(const VariableBlockF64& m) { return -VariableMatrixF64{m}; }

// variable_matrix.hpp:1516-1620
inline VariableMatrixF64 solve(const VariableMatrixF64& A, const VariableMatrixF64& B) {
  assert(A.rows() == B.rows());
  if (A.rows() == 1 && A.cols() == 1) {
    return VariableMatrixF64{B[0, 0] / A[0, 0]};
  } else if (A.rows() == 2 && A.cols() == 2) {
    const auto& a = A[0, 0];
    const auto& b = A[0, 1];
    const auto& c = A[1, 0];
    const auto& d = A[1, 1];
    VariableMatrixF64 adj_A{{d, -b}, {-c, a}};
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
    VariableMatrixF64 adj_A{{adj_A00, ch - bi, bf - ce}, {adj_A10, ai - cg, cd - af}, {adj_A20, bg - ah, ae - bd}};
    auto det_A = a * adj_A00 + b * adj_A10 + c * adj_A20;
    return adj_A / det_A * B;
  }
  // beyond 3x3: Gaussian elimination on the expressions, rows pivoted by the CURRENT values (the
  // reference goes through Eigen's decompositions of Variables there, variable_matrix.hpp:1588-1620)
  const int n = A.rows(), k = B.cols();
  assert(A.cols() == n);
  VariableMatrixF64 M{detail::empty, n, n + k};
  for (int r = 0; r < n; ++r) {
    for (int c = 0; c < n; ++c) M[r, c] = A[r, c];
    for (int c = 0; c < k; ++c) M[r, n + c] = B[r, c];
  }
  for (int col = 0; col < n; ++col) {
    int piv = col;
    for (int r = col + 1; r < n; ++r)
      if (std::fabs(M[r, col].value()) > std::fabs(M[piv, col].value())) piv = r;
    if (piv != col)
      for (int c = col; c < n + k; ++c) std::swap(M[piv, c], M[col, c]);
    for (int r = col + 1; r < n; ++r) {
      const VariableF64 f = M[r, col] / M[col, col];
      for (int c = col + 1; c < n + k; ++c) M[r, c] = M[r, c] - f * M[col, c];
    }
  }
  VariableMatrixF64 X{detail::empty, n, k};
  for (int c = 0; c < k; ++c)
    for (int r = n - 1; r >= 0; --r) {
      VariableF64 acc = M[r, n + c];
      for (int j = r + 1; j < n; ++j) acc = acc - M[r, j] * X[j, c];
      X[r, c] = acc / M[r, r];
    }
  return X;
}

// exp(A) ~ q(A)^-1 p(A) with p the degree-13 diagonal Padé numerator, q(A) = p(-A); the
// coefficients follow c_0 = 1, c_k = c_{k-1} (m - k + 1) / (k (2m - k + 1)), m = 13.  Horner in A.
inline VariableMatrixF64 VariableMatrixF64::exp() const {
  assert(m_rows == m_cols);
  constexpr int m = 13;
  double c[m + 1];
  c[0] = 1.0;
  for (int k = 1; k <= m; ++k) c[k] = c[k - 1] * static_cast<double>(m - k + 1) / static_cast<double>(k * (2 * m - k + 1));
  const VariableMatrixF64 I = identity(m_rows);
  VariableMatrixF64 P = I * c[m], Q = I * -c[m];  // (-1)^13 = -1
  for (int k = m - 1; k >= 0; --k) {
    P = P * *this + I * c[k];
    Q = Q * *this + I * ((k & 1) ? -c[k] : c[k]);
  }
  return solve(Q, P);
}
inline VariableMatrixF64 VariableBlockF64::exp() const { return VariableMatrixF64{*this}.exp(); }

// ---- constraints (variable.hpp:716-1013) ------------------------------------------
namespace detail {
// (derived_from: the Variable<double> / VariableMatrix<double> spellings are classes derived
// from the fp64 ones)
template <typename T>
concept ScalarOperand = std::is_arithmetic_v<std::decay_t<T>> || std::derived_from<std::decay_t<T>, VariableF64>;
template <typename T>
concept MatrixOperand = std::derived_from<std::decay_t<T>, VariableMatrixF64> || std::derived_from<std::decay_t<T>, VariableBlockF64> ||
                        std::same_as<std::decay_t<T>, DenseMatrix>;
template <typename T>
concept SleipnirOperand = std::derived_from<std::decay_t<T>, VariableF64> || std::derived_from<std::decay_t<T>, VariableMatrixF64> ||
                          std::derived_from<std::decay_t<T>, VariableBlockF64>;

template <typename L, typename R>
std::vector<VariableF64> make_constraints(const L& lhs, const R& rhs) {
  std::vector<VariableF64> out;
  if constexpr (ScalarOperand<L> && ScalarOperand<R>) {
    out.push_back(VariableF64{lhs} - VariableF64{rhs});
  } else if constexpr (ScalarOperand<L>) {
    for (int r = 0; r < rhs.rows(); ++r)
      for (int c = 0; c < rhs.cols(); ++c) out.push_back(VariableF64{lhs} - VariableF64{rhs[r, c]});
  } else if constexpr (ScalarOperand<R>) {
    for (int r = 0; r < lhs.rows(); ++r)
      for (int c = 0; c < lhs.cols(); ++c) out.push_back(VariableF64{lhs[r, c]} - VariableF64{rhs});
  } else {
    assert(lhs.rows() == rhs.rows() && lhs.cols() == rhs.cols());
    for (int r = 0; r < lhs.rows(); ++r)
      for (int c = 0; c < lhs.cols(); ++c) out.push_back(VariableF64{lhs[r, c]} - VariableF64{rhs[r, c]});
  }
  return out;
}
}  // namespace detail

struct EqualityConstraintsF64 {
  std::vector<VariableF64> constraints;
  EqualityConstraintsF64() = default;
  EqualityConstraintsF64(std::initializer_list<EqualityConstraintsF64> list) {
    for (auto& e : list) constraints.insert(constraints.end(), e.constraints.begin(), e.constraints.end());
  }
  explicit EqualityConstraintsF64(std::vector<VariableF64> c) : constraints{std::move(c)} {}
  explicit operator bool() const {
    for (auto& c : constraints)
      if (c.value() != 0.0) return false;
    return true;
  }
};
struct InequalityConstraintsF64 {
  std::vector<VariableF64> constraints;
  InequalityConstraintsF64() = default;
  InequalityConstraintsF64(std::initializer_list<InequalityConstraintsF64> list) {
    for (auto& e : list) constraints.insert(constraints.end(), e.constraints.begin(), e.constraints.end());
  }
  explicit InequalityConstraintsF64(std::vector<VariableF64> c) : constraints{std::move(c)} {}
  explicit operator bool() const {
    for (auto& c : constraints)
      if (!(c.value() >= 0.0)) return false;
    return true;
  }
};

template <typename L, typename R>
  requires(detail::ScalarOperand<L> || detail::MatrixOperand<L>) && (detail::ScalarOperand<R> || detail::MatrixOperand<R>) &&
          (detail::SleipnirOperand<L> || detail::SleipnirOperand<R>)
EqualityConstraintsF64 operator==(const L& lhs, const R& rhs) {
  return EqualityConstraintsF64{detail::make_constraints(lhs, rhs)};
}
template <typename L, typename R>
  requires(detail::ScalarOperand<L> || detail::MatrixOperand<L>) && (detail::ScalarOperand<R> || detail::MatrixOperand<R>) &&
          (detail::SleipnirOperand<L> || detail::SleipnirOperand<R>)
InequalityConstraintsF64 operator>=(const L& lhs, const R& rhs) {
  return InequalityConstraintsF64{detail::make_constraints(lhs, rhs)};
}
template <typename L, typename R>
  requires(detail::ScalarOperand<L> || detail::MatrixOperand<L>) && (detail::ScalarOperand<R> || detail::MatrixOperand<R>) &&
          (detail::SleipnirOperand<L> || detail::SleipnirOperand<R>)
InequalityConstraintsF64 operator<=(const L& lhs, const R& rhs) {
  return rhs >= lhs;
}
template <typename L, typename R>
  requires(detail::ScalarOperand<L> || detail::MatrixOperand<L>) && (detail::ScalarOperand<R> || detail::MatrixOperand<R>) &&
          (detail::SleipnirOperand<L> || detail::SleipnirOperand<R>)
InequalityConstraintsF64 operator>(const L& lhs, const R& rhs) {
  return lhs >= rhs;
}
template <typename L, typename R>
  requires(detail::ScalarOperand<L> || detail::MatrixOperand<L>) && (detail::ScalarOperand<R> || detail::MatrixOperand<R>) &&
          (detail::SleipnirOperand<L> || detail::SleipnirOperand<R>)
InequalityConstraintsF64 operator<(const L& lhs, const R& rhs) {
  return rhs >= lhs;
}

template <typename L, typename X, typename U>
InequalityConstraintsF64 bounds(const L& l, const X& x, const U& u) {
  return InequalityConstraintsF64{l <= x, x <= u};
}


// ---------------------------------------------------------------------------
// The reference's spellings: slp::Variable<Scalar>, VariableMatrix<Scalar>, VariableBlock<...>,
// EqualityConstraints<Scalar>, InequalityConstraints<Scalar>
// (include/sleipnir/autodiff/variable.hpp:52, variable_matrix.hpp:43, variable.hpp:832,877).
// Only Scalar = double exists: the expression graph, the tapes and every kernel behind it are
// fp64.  The specializations ARE the fp64 classes above (derived, constructors inherited), so
// every operator and function of this header serves them, `slp::Variable J = 0.0;` deduces
// Variable<double> like the reference's deduction guides do (variable.hpp:283-295), and
// `auto X = problem.decision_variable(4, N + 1)` is assignable to a VariableMatrix<double>.
// ---------------------------------------------------------------------------
template <typename Scalar>
class Variable;
template <typename Scalar>
class VariableMatrix;
template <typename Scalar>
struct EqualityConstraints;
template <typename Scalar>
struct InequalityConstraints;

template <>
class Variable<double> : public VariableF64 {
 public:
  using VariableF64::VariableF64;
  using VariableF64::operator=;
  Variable() = default;
  Variable(const VariableF64& v) : VariableF64{v} {}  // NOLINT
};
Variable() -> Variable<double>;
Variable(double) -> Variable<double>;
Variable(std::integral auto) -> Variable<double>;
Variable(const VariableF64&) -> Variable<double>;
Variable(const VariableMatrixF64&) -> Variable<double>;
Variable(const VariableBlockF64&) -> Variable<double>;

template <>
class VariableMatrix<double> : public VariableMatrixF64 {
 public:
  using VariableMatrixF64::VariableMatrixF64;
  using VariableMatrixF64::operator=;
  VariableMatrix() = default;
  VariableMatrix(const VariableMatrixF64& m) : VariableMatrixF64{m} {}   // NOLINT
  VariableMatrix(VariableMatrixF64&& m) : VariableMatrixF64{std::move(m)} {}  // NOLINT
  VariableMatrix(const VariableBlockF64& b) : VariableMatrixF64{b} {}    // NOLINT
};

// slp::sin<Scalar> and friends, as template-ids (`arm.row(0).cwise_transform(slp::sin<T>)`,
// arm_on_elevator_problem_test.cpp:97; variable.hpp:340-830 declares every function as a template
// over Scalar).  Plain calls keep resolving to the fp64 functions above.
#define SLP_UNARY_T(name) \
  template <typename Scalar>                                                              \
  Variable<Scalar> name(const Variable<Scalar>& x) { return name(static_cast<const VariableF64&>(x)); }
SLP_UNARY_T(abs)
SLP_UNARY_T(acos)
SLP_UNARY_T(asin)
SLP_UNARY_T(atan)
SLP_UNARY_T(cbrt)
SLP_UNARY_T(cos)
SLP_UNARY_T(cosh)
SLP_UNARY_T(erf)
SLP_UNARY_T(exp)
SLP_UNARY_T(log)
SLP_UNARY_T(log10)
SLP_UNARY_T(sign)
SLP_UNARY_T(sin)
SLP_UNARY_T(sinh)
SLP_UNARY_T(sqrt)
SLP_UNARY_T(tan)
SLP_UNARY_T(tanh)
#undef SLP_UNARY_T
#define SLP_BINARY_T(name) \
  template <typename Scalar>                                                              \
  Variable<Scalar> name(const Variable<Scalar>& a, const Variable<Scalar>& b) {           \
    return name(static_cast<const VariableF64&>(a), static_cast<const VariableF64&>(b));  \
  }
SLP_BINARY_T(atan2)
SLP_BINARY_T(hypot)
SLP_BINARY_T(max)
SLP_BINARY_T(min)
SLP_BINARY_T(pow)
#undef SLP_BINARY_T

template <>
struct EqualityConstraints<double> : public EqualityConstraintsF64 {
  using EqualityConstraintsF64::EqualityConstraintsF64;
  EqualityConstraints(const EqualityConstraintsF64& c) : EqualityConstraintsF64{c} {}  // NOLINT
};
template <>
struct InequalityConstraints<double> : public InequalityConstraintsF64 {
  using InequalityConstraintsF64::InequalityConstraintsF64;
  InequalityConstraints(const InequalityConstraintsF64& c) : InequalityConstraintsF64{c} {}  // NOLINT
};
// slp::cwise_reduce<T>(A, B, std::multiplies<>{}) (variable_matrix_test.cpp:522-532)
template <typename Scalar, typename F>
VariableMatrix<Scalar> cwise_reduce(const VariableMatrix<Scalar>& lhs, const VariableMatrix<Scalar>& rhs, F&& binary_op) {
  return cwise_reduce(static_cast<const VariableMatrixF64&>(lhs), static_cast<const VariableMatrixF64&>(rhs),
                      [&](const VariableF64& x, const VariableF64& y) -> VariableF64 { return binary_op(x, y); });
}
VariableMatrix(const VariableMatrixF64&) -> VariableMatrix<double>;
VariableMatrix(const VariableBlockF64&) -> VariableMatrix<double>;
VariableMatrix(std::initializer_list<std::initializer_list<VariableF64>>) -> VariableMatrix<double>;

// `EqualityConstraints eq = x == y;`, `EqualityConstraints eqs{eq1, eq2};` (constraints_test.cpp:247-276)
EqualityConstraints(const EqualityConstraintsF64&) -> EqualityConstraints<double>;
EqualityConstraints(std::initializer_list<EqualityConstraintsF64>) -> EqualityConstraints<double>;
InequalityConstraints(const InequalityConstraintsF64&) -> InequalityConstraints<double>;
InequalityConstraints(std::initializer_list<InequalityConstraintsF64>) -> InequalityConstraints<double>;

}  // namespace slp
