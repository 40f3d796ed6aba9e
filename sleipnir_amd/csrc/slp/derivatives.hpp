// slp::Gradient, slp::Jacobian, slp::Hessian — the reference's derivative evaluators
// (include/sleipnir/autodiff/gradient.hpp:24-77, jacobian.hpp:30-170, hessian.hpp:33-170) on the
// device path: get() is the symbolic gradient tree, value() runs the COMPILED TAPE on the GPU at
// the variables' current values and reads g / A_e / H_f out of the value vector (no CPU
// fallback: value() throws without a HIP device).
//
// The reference returns Eigen::SparseMatrix / SparseVector; Eigen is not in this toolchain, so
// value() returns slp::SparseMatrix, a column-compressed matrix with the members the reference's
// tests use (coeff, toDense, rows, cols, nonZeros).
//
// How value() gets there: the expression(s) become the cost / the equality rows of a throw-away
// slp::Problem whose decision variables are `wrt`; structure and tape are compiled once, every
// value() is one sweep launch (the path of a Newton step's AD refresh).
#pragma once

#include <algorithm>
#include <memory>
#include <utility>
#include <vector>

#include "problem.hpp"

namespace slp {

// Eigen's UpLo constants by value, for Hessian's second template argument
inline constexpr int Lower = 1, Upper = 2;

class SparseMatrix {
 public:
  SparseMatrix() = default;
  SparseMatrix(int rows, int cols) : m_rows{rows}, m_cols{cols}, m_colptr(static_cast<size_t>(cols) + 1, 0) {}
  int rows() const { return m_rows; }
  int cols() const { return m_cols; }
  int nonZeros() const { return static_cast<int>(m_values.size()); }
  double coeff(int row, int col) const {
    double v = 0.0;
    for (int q = m_colptr[col]; q < m_colptr[col + 1]; ++q)
      if (m_rowidx[q] == row) v += m_values[q];
    return v;
  }
  double coeff(int index) const { return m_cols == 1 ? coeff(index, 0) : coeff(0, index); }  // a vector
  DenseMatrix toDense() const {
    DenseMatrix d{m_rows, m_cols};
    for (int c = 0; c < m_cols; ++c)
      for (int q = m_colptr[c]; q < m_colptr[c + 1]; ++q) d[m_rowidx[q], c] += m_values[q];
    return d;
  }
  SparseMatrix transpose() const {
    std::vector<std::vector<std::pair<int, double>>> cols(m_rows);
    for (int c = 0; c < m_cols; ++c)
      for (int q = m_colptr[c]; q < m_colptr[c + 1]; ++q) cols[m_rowidx[q]].emplace_back(c, m_values[q]);
    SparseMatrix t{m_cols, m_rows};
    for (int c = 0; c < m_rows; ++c) {
      for (auto& [r, v] : cols[c]) t.push(r, v);
      t.end_column(c);
    }
    return t;
  }
  const std::vector<int>& outerIndex() const { return m_colptr; }
  const std::vector<int>& innerIndex() const { return m_rowidx; }
  const std::vector<double>& values() const { return m_values; }

  // column by column: push the entries of column c, then end_column(c)
  void push(int row, double value) {
    m_rowidx.push_back(row);
    m_values.push_back(value);
  }
  void end_column(int col) { m_colptr[col + 1] = static_cast<int>(m_values.size()); }

 private:
  int m_rows = 0, m_cols = 0;
  std::vector<int> m_colptr, m_rowidx;
  std::vector<double> m_values;
};

namespace detail {

inline std::vector<VariableF64> elements(const VariableMatrixF64& m) { return {m.begin(), m.end()}; }
// `wrt` as the reference takes it: one Variable, or anything matrix-like (matrix, block, slice)
template <typename W>
std::vector<VariableF64> wrt_elements(const W& wrt) {
  if constexpr (std::derived_from<W, VariableF64>) return {wrt};
  else return elements(VariableMatrixF64{wrt});
}

// symbolic row of derivatives of `f` with respect to `wrt` (structural zeros are the constant 0)
inline std::vector<VariableF64> gradient_row(const VariableF64& f, const std::vector<VariableF64>& wrt) {
  auto& g = G();
  std::vector<NodeId> w;
  for (auto& v : wrt) w.push_back(v.expr);
  const std::vector<NodeId> grad = g.gradient_tree(g.topological_sort(f.expr), w);
  std::vector<VariableF64> out;
  for (NodeId n : grad) out.push_back(n == slpx::kNull ? VariableF64{0.0} : VariableF64::wrap(n));
  return out;
}

class DerivativeEvaluator {
 public:
  DerivativeEvaluator(const VariableF64* cost, const std::vector<VariableF64>& rows, std::vector<VariableF64> wrt)
      : m_wrt{std::move(wrt)}, m_rows{static_cast<int>(rows.size())}, m_problem{std::make_unique<ProblemF64>()} {
    for (auto& w : m_wrt) m_problem->adopt_decision_variable(w);
    if (cost) m_problem->minimize(*cost);
    for (auto& r : rows) m_problem->subject_to(EqualityConstraintsF64{std::vector<VariableF64>{r}});
  }
  // one full sweep on the device at the variables' current values
  const std::vector<double>& sweep() {
    slpx::NewtonSystem& sys = m_problem->compile();  // once; throws without a HIP device
    const auto& st = sys.structure();
    auto& dev = sys.device();
    std::vector<double> x(m_wrt.size());
    for (size_t i = 0; i < x.size(); ++i) x[i] = m_wrt[i].value();
    const std::vector<double> zeros(std::max(1, m_rows), 0.0), ones(1, 1.0);
    dev.set_scaling(std::vector<double>(st.n_scales(), 1.0));
    dev.upload_x(x.data());
    dev.upload_duals(ones.data(), zeros.data(), ones.data());
    dev.sweep_full();
    m_V.resize(st.nV);
    dev.download_V(m_V.data());
    return m_V;
  }
  const slpx::NlpStructure& structure() { return m_problem->compile().structure(); }
  int n() const { return static_cast<int>(m_wrt.size()); }

 private:
  std::vector<VariableF64> m_wrt;
  int m_rows;
  std::unique_ptr<ProblemF64> m_problem;
  std::vector<double> m_V;
};

inline SparseMatrix from_pattern(const slpx::CscPattern& pat, const double* values, int rows, int cols) {
  SparseMatrix m{rows, cols};
  for (int c = 0; c < cols; ++c) {
    for (int q = pat.colptr[c]; q < pat.colptr[c + 1]; ++q) m.push(pat.rowidx[q], values[q]);
    m.end_column(c);
  }
  return m;
}

}  // namespace detail

template <typename Scalar>
class Jacobian;
template <>
class Jacobian<double> {
 public:
  // (variable | variables, wrt | wrt matrix): jacobian.hpp:37-60
  template <typename V, typename W>
  Jacobian(const V& variables, const W& wrt) : m_rows{detail::wrt_elements(variables)}, m_wrt{detail::wrt_elements(wrt)} {}

  // jacobian.hpp:113-130: row r = the gradient tree of variables[r]
  VariableMatrix<double> get() const {
    VariableMatrixF64 J{detail::empty, static_cast<int>(m_rows.size()), static_cast<int>(m_wrt.size())};
    for (size_t r = 0; r < m_rows.size(); ++r) {
      const auto row = detail::gradient_row(m_rows[r], m_wrt);
      for (size_t c = 0; c < m_wrt.size(); ++c) J[static_cast<int>(r), static_cast<int>(c)] = row[c];
    }
    return J;
  }

  // jacobian.hpp:134-156
  const SparseMatrix& value() {
    if (!m_eval) m_eval = std::make_shared<detail::DerivativeEvaluator>(nullptr, m_rows, m_wrt);
    const auto& V = m_eval->sweep();
    const auto& st = m_eval->structure();
    m_J = detail::from_pattern(st.Ae, V.data() + st.off_Ae, static_cast<int>(m_rows.size()), m_eval->n());
    return m_J;
  }

 private:
  std::vector<VariableF64> m_rows, m_wrt;
  std::shared_ptr<detail::DerivativeEvaluator> m_eval;
  SparseMatrix m_J;
};
template <typename V, typename W>
Jacobian(const V&, const W&) -> Jacobian<double>;

template <typename Scalar>
class Gradient;
template <>
class Gradient<double> {
 public:
  template <typename W>
  Gradient(const VariableF64& variable, const W& wrt) : m_f{variable}, m_wrt{detail::wrt_elements(wrt)} {}

  // gradient.hpp:53-57: a column
  VariableMatrix<double> get() const { return VariableMatrixF64{detail::gradient_row(m_f, m_wrt)}; }

  // the cost's gradient block g of the value vector
  const SparseMatrix& value() {
    if (!m_eval) m_eval = std::make_shared<detail::DerivativeEvaluator>(&m_f, std::vector<VariableF64>{}, m_wrt);
    const auto& V = m_eval->sweep();
    const auto& st = m_eval->structure();
    // g_pat is 1 x n: one (possibly empty) column per variable
    m_g = detail::from_pattern(st.g_pat, V.data() + st.off_g, 1, m_eval->n()).transpose();
    return m_g;
  }

 private:
  VariableF64 m_f;
  std::vector<VariableF64> m_wrt;
  std::shared_ptr<detail::DerivativeEvaluator> m_eval;
  SparseMatrix m_g;
};
template <typename W>
Gradient(const VariableF64&, const W&) -> Gradient<double>;

template <typename Scalar, int UpLo = (Lower | Upper)>
class Hessian;
template <int UpLo>
class Hessian<double, UpLo> {
  static_assert(UpLo == Lower || UpLo == (Lower | Upper), "Hessian: Lower or Lower | Upper (hessian.hpp:33-38)");

 public:
  template <typename W>
  Hessian(const VariableF64& variable, const W& wrt) : m_f{variable}, m_wrt{detail::wrt_elements(wrt)} {}

  // hessian.hpp:111-128: the Jacobian of the gradient tree
  VariableMatrix<double> get() const {
    const auto g = detail::gradient_row(m_f, m_wrt);
    const int n = static_cast<int>(m_wrt.size());
    VariableMatrixF64 H{detail::empty, n, n};
    for (int r = 0; r < n; ++r) {
      const auto row = detail::gradient_row(g[r], m_wrt);
      for (int c = 0; c < n; ++c) H[r, c] = (UpLo == Lower && c > r) ? VariableF64{0.0} : row[c];
    }
    return H;
  }

  // the cost's Hessian block H_f (lower triangle on the device), mirrored when both are asked for
  const SparseMatrix& value() {
    if (!m_eval) m_eval = std::make_shared<detail::DerivativeEvaluator>(&m_f, std::vector<VariableF64>{}, m_wrt);
    const auto& V = m_eval->sweep();
    const auto& st = m_eval->structure();
    const int n = m_eval->n();
    const SparseMatrix lower = detail::from_pattern(st.Hf, V.data() + st.off_Hf, n, n);
    if constexpr (UpLo == Lower) {
      m_H = lower;
    } else {
      const SparseMatrix upper = lower.transpose();
      m_H = SparseMatrix{n, n};
      for (int c = 0; c < n; ++c) {
        for (int q = upper.outerIndex()[c]; q < upper.outerIndex()[c + 1]; ++q)
          if (upper.innerIndex()[q] < c) m_H.push(upper.innerIndex()[q], upper.values()[q]);
        for (int q = lower.outerIndex()[c]; q < lower.outerIndex()[c + 1]; ++q) m_H.push(lower.innerIndex()[q], lower.values()[q]);
        m_H.end_column(c);
      }
    }
    return m_H;
  }

 private:
  VariableF64 m_f;
  std::vector<VariableF64> m_wrt;
  std::shared_ptr<detail::DerivativeEvaluator> m_eval;
  SparseMatrix m_H;
};
template <typename W>
Hessian(const VariableF64&, const W&) -> Hessian<double, (Lower | Upper)>;

}  // namespace slp
