// slp::Problem<double> — the user surface of the reference
// (include/sleipnir/optimization/problem.hpp:67-744) kept for the path's callers
// (ProblemF64 is the implementation, Problem<Scalar> at the end of this header the spelling):
// decision_variable(), minimize()/maximize(), subject_to(), solve(), add_callback().
// solve() compiles the model for the GPU (NewtonSystem) and runs the interior-point
// iteration around the device Newton step.
//
// Like the reference, solve() routes unconstrained problems to newton() and equality-only
// problems to sqp() (problem.hpp:335,403), everything else to interior_point() — all three on
// the same device Newton step (csrc/ipm.cpp).
#pragma once

#include <algorithm>
#include <cstdint>
#include <fstream>
#include <functional>
#include <map>
#include <memory>
#include <optional>
#include <string>
#include <utility>
#include <vector>

#include "../ipm.hpp"
#include "../newton.hpp"
#include "variable.hpp"

namespace slp {

using ExitStatus = slpx::ExitStatus;
using Options = slpx::Options;
// slp::IterationInfo<Scalar> (solver/iteration_info.hpp:13): only fp64 exists
template <typename Scalar = double>
using IterationInfo = slpx::IterationInfo;
using SolveReport = slpx::SolveReport;

// util/spy.hpp:20-80: the file tools/spy.py plots — title, row label, column label (each a
// 32-bit little-endian length and the bytes), rows, columns, then per iteration the number of
// coordinates and (row, column, '+' | '-' | '0') for every stored entry, column by column.
class Spy {
 public:
  Spy(const std::string& filename, const std::string& title, const std::string& row_label,
      const std::string& col_label, int rows, int cols)
      : m_file{filename, std::ios::binary} {
    for (const std::string* text : {&title, &row_label, &col_label}) {
      write32le(static_cast<int32_t>(text->size()));
      m_file.write(text->data(), static_cast<std::streamsize>(text->size()));
    }
    write32le(rows);
    write32le(cols);
  }
  // the entries of `patterns` (same shape; values at V + offset, summed where two patterns meet)
  void add(std::initializer_list<std::pair<const slpx::CscPattern*, const double*>> patterns, int cols) {
    std::vector<std::map<int, double>> columns(static_cast<size_t>(cols));
    int32_t count = 0;
    for (const auto& [pat, values] : patterns)
      for (int c = 0; c < cols && c + 1 < static_cast<int>(pat->colptr.size()); ++c)
        for (int q = pat->colptr[c]; q < pat->colptr[c + 1]; ++q) {
          auto [it, fresh] = columns[c].try_emplace(pat->rowidx[q], 0.0);
          it->second += values[q];
          count += fresh;
        }
    write32le(count);
    for (int c = 0; c < cols; ++c)
      for (const auto& [row, value] : columns[c]) {
        write32le(row);
        write32le(c);
        m_file << (value > 0.0 ? '+' : value < 0.0 ? '-' : '0');
      }
    m_file.flush();
  }

 private:
  std::ofstream m_file;
  void write32le(int32_t num) {
    const unsigned char b[4] = {static_cast<unsigned char>(num), static_cast<unsigned char>(num >> 8),
                                static_cast<unsigned char>(num >> 16), static_cast<unsigned char>(num >> 24)};
    m_file.write(reinterpret_cast<const char*>(b), 4);
  }
};

class ProblemF64 {
 public:
  ProblemF64() noexcept = default;

  [[nodiscard]] VariableF64 decision_variable() {
    invalidate();
    m_decision_variables.emplace_back();
    return m_decision_variables.back();
  }

  // problem.hpp:91-104
  [[nodiscard]] VariableMatrixF64 decision_variable(int rows, int cols = 1) {
    invalidate();
    VariableMatrixF64 vars{detail::empty, rows, cols};
    for (int row = 0; row < rows; ++row)
      for (int col = 0; col < cols; ++col) {
        m_decision_variables.emplace_back();
        vars[row, col] = m_decision_variables.back();
      }
    return vars;
  }

  // FFI helper (no reference counterpart): make an already-created free VariableF64 a
  // decision variable of this problem.
  void adopt_decision_variable(const VariableF64& v) {
    invalidate();
    m_decision_variables.push_back(v);
  }

  // problem.hpp:118-140
  [[nodiscard]] VariableMatrixF64 symmetric_decision_variable(int rows) {
    invalidate();
    VariableMatrixF64 vars{detail::empty, rows, rows};
    for (int row = 0; row < rows; ++row)
      for (int col = 0; col <= row; ++col) {
        m_decision_variables.emplace_back();
        vars[row, col] = m_decision_variables.back();
        vars[col, row] = m_decision_variables.back();
      }
    return vars;
  }

  // The compiled system belongs to the model as it was when compile() ran: every change of
  // the model drops it (the reference rebuilds its evaluators in every solve(), problem.hpp:517-660).
  void minimize(const VariableF64& cost) {
    invalidate();
    m_f = cost;
  }
  void maximize(const VariableF64& objective) {
    invalidate();
    m_f = -objective;
  }
  void subject_to(const EqualityConstraintsF64& constraint) {
    invalidate();
    m_equality_constraints.insert(m_equality_constraints.end(), constraint.constraints.begin(),
                                  constraint.constraints.end());
  }
  void subject_to(const InequalityConstraintsF64& constraint) {
    invalidate();
    m_inequality_constraints.insert(m_inequality_constraints.end(),
                                    constraint.constraints.begin(), constraint.constraints.end());
  }

  ExpressionType cost_function_type() const { return m_f ? m_f->type() : ExpressionType::NONE; }
  ExpressionType equality_constraint_type() const { return max_type(m_equality_constraints); }
  ExpressionType inequality_constraint_type() const { return max_type(m_inequality_constraints); }

  template <typename F>
    requires requires(F cb, const slpx::IterationInfo& info) { { cb(info) } -> std::same_as<void>; }
  void add_callback(F&& callback) {
    m_iteration_callbacks.emplace_back([cb = std::forward<F>(callback)](const slpx::IterationInfo& info) {
      cb(info);
      return false;
    });
  }
  template <typename F>
    requires requires(F cb, const slpx::IterationInfo& info) { { cb(info) } -> std::same_as<bool>; }
  void add_callback(F&& callback) {
    m_iteration_callbacks.emplace_back(std::forward<F>(callback));
  }
  void clear_callbacks() { m_iteration_callbacks.clear(); }
  // problem.hpp:714-735: callbacks that survive clear_callbacks() and run after the others
  template <typename F>
    requires requires(F cb, const slpx::IterationInfo& info) { { cb(info) } -> std::same_as<void>; }
  void add_persistent_callback(F&& callback) {
    m_persistent_iteration_callbacks.emplace_back([cb = std::forward<F>(callback)](const slpx::IterationInfo& info) {
      cb(info);
      return false;
    });
  }
  template <typename F>
    requires requires(F cb, const slpx::IterationInfo& info) { { cb(info) } -> std::same_as<bool>; }
  void add_persistent_callback(F&& callback) {
    m_persistent_iteration_callbacks.emplace_back(std::forward<F>(callback));
  }

  // problem.hpp:281-679
  ExitStatus solve(const Options& options = Options{}, bool spy = false) {
    auto& g = detail::G();
    std::vector<double> x(m_decision_variables.size());
    for (size_t i = 0; i < x.size(); ++i) x[i] = m_decision_variables[i].value();

    // problem.hpp:304-313
    if (cost_function_type() <= ExpressionType::CONSTANT &&
        equality_constraint_type() <= ExpressionType::CONSTANT &&
        inequality_constraint_type() <= ExpressionType::CONSTANT) {
      return ExitStatus::SUCCESS;
    }

    compile();

    // get_bounds conflict test (util/bounds.hpp:55-190, problem.hpp:597-606)
    std::vector<double> V(m_sys->structure().nV);
    auto& dev = m_sys->device();
    dev.set_scaling(std::vector<double>(m_sys->structure().n_scales(), 1.0));
    std::vector<double> zeros_e(std::max<size_t>(1, m_equality_constraints.size()), 0.0);
    std::vector<double> ones_i(std::max<size_t>(1, m_inequality_constraints.size()), 1.0);
    dev.upload_x(x.data());
    dev.upload_duals(ones_i.data(), zeros_e.data(), ones_i.data());
    dev.sweep_full();
    dev.download_V(V.data());
    if (has_conflicting_bounds(V)) return ExitStatus::GLOBALLY_INFEASIBLE;

    // problem.hpp:615-616
    m_scales = slpx::compute_problem_scaling(m_sys->structure(), V);
    dev.set_scaling(m_scales);

    // problem.hpp:365-375, 453-470, 571-596: sparsity files of H (lower triangle of the Lagrangian's
    // Hessian), A_e, A_i — one record per iteration, written from a callback like the reference's
    std::vector<slpx::IterationCallback> callbacks = m_iteration_callbacks;
    callbacks.insert(callbacks.end(), m_persistent_iteration_callbacks.begin(), m_persistent_iteration_callbacks.end());
    std::unique_ptr<Spy> H_spy, A_e_spy, A_i_spy;
    if (spy) {
      const int n = static_cast<int>(x.size()), m_e = static_cast<int>(m_equality_constraints.size()),
                m_i = static_cast<int>(m_inequality_constraints.size());
      H_spy = std::make_unique<Spy>("H.spy", "Hessian", "Decision variables", "Decision variables", n, n);
      if (m_e || m_i) A_e_spy = std::make_unique<Spy>("A_e.spy", "Equality constraint Jacobian", "Constraints", "Decision variables", m_e, n);
      if (m_i) A_i_spy = std::make_unique<Spy>("A_i.spy", "Inequality constraint Jacobian", "Constraints", "Decision variables", m_i, n);
      callbacks.emplace_back([&](const slpx::IterationInfo& info) {
        const slpx::NlpStructure& st = info.structure ? *info.structure : m_sys->structure();
        const double* V = info.V.data();
        H_spy->add({{&st.Hf, V + st.off_Hf}, {&st.Hc, V + st.off_Hc}}, st.n);
        if (A_e_spy) A_e_spy->add({{&st.Ae, V + st.off_Ae}}, st.n);
        if (A_i_spy) A_i_spy->add({{&st.Ai, V + st.off_Ai}}, st.n);
        return false;
      });
    }

    // problem.hpp:335, 403, 512: the solver follows the kinds of constraints present
    ExitStatus status;
    if (m_equality_constraints.empty() && m_inequality_constraints.empty()) {
      m_s.clear();
      m_y.clear();
      m_z.clear();
      status = slpx::newton(*m_sys, m_scales, callbacks, options, x, &m_report);
    } else if (m_inequality_constraints.empty()) {
      m_s.clear();
      m_z.clear();
      status = slpx::sqp(*m_sys, m_scales, callbacks, options, x, &m_y, &m_report);
    } else {
      status = slpx::interior_point(*m_sys, m_scales, callbacks, options, x, &m_s, &m_y, &m_z,
                                    &m_report);
    }
    // problem.hpp:676
    for (size_t i = 0; i < x.size(); ++i) g.val[m_decision_variables[i].expr] = x[i];
    return status;
  }

  // feasibility_restoration (util/feasibility_restoration.hpp:347-628) from a caller-given
  // iterate, `steps` iterations of it, with the scaling solve() would use (at the variables'
  // current values): the seam the restoration parity tests compare with the oracle at.
  ExitStatus restoration_steps(const Options& options, std::vector<double>& x, std::vector<double>& s,
                               std::vector<double>& y, std::vector<double>& z, double mu, int steps) {
    auto& g = detail::G();
    compile();
    std::vector<double> x0(m_decision_variables.size());
    for (size_t i = 0; i < x0.size(); ++i) x0[i] = g.val[m_decision_variables[i].expr];
    std::vector<double> V(m_sys->structure().nV);
    auto& dev = m_sys->device();
    dev.set_scaling(std::vector<double>(m_sys->structure().n_scales(), 1.0));
    std::vector<double> zeros_e(std::max<size_t>(1, m_equality_constraints.size()), 0.0);
    std::vector<double> ones_i(std::max<size_t>(1, m_inequality_constraints.size()), 1.0);
    dev.upload_x(x0.data());
    dev.upload_duals(ones_i.data(), zeros_e.data(), ones_i.data());
    dev.sweep_full();
    dev.download_V(V.data());
    m_scales = slpx::compute_problem_scaling(m_sys->structure(), V);
    dev.set_scaling(m_scales);
    return slpx::feasibility_restoration_steps(*m_sys, m_scales, options, x, s, y, z, mu, steps, &m_report);
  }

  // Compiles (once) the NLP for the device; exposed so harnesses can time setup
  // separately and drive the Newton step directly.
  slpx::NewtonSystem& compile(const slpx::NewtonOptions& opt = {}) {
    // Values of free Variables that are not decision variables ("parameters") are folded into
    // the compiled constants and the cached linear rows: a changed one means a new system.
    if (m_sys) {
      auto& g = detail::G();
      for (const auto& [node, value] : m_param_snapshot)
        if (g.val[node] != value) {
          invalidate();
          break;
        }
    }
    if (!m_sys) {
      std::vector<NodeId> xs, ce, ci;
      for (auto& v : m_decision_variables) xs.push_back(v.expr);
      for (auto& v : m_equality_constraints) ce.push_back(v.expr);
      for (auto& v : m_inequality_constraints) ci.push_back(v.expr);
      m_sys = std::make_unique<slpx::NewtonSystem>(detail::G(), xs, m_f ? m_f->expr : slpx::kNull, ce,
                                                   ci, opt);
      m_param_snapshot.clear();
      for (const slpx::TapeProgram* prog : {&m_sys->structure().full, &m_sys->structure().values})
        for (const auto& [node, slot] : prog->params) m_param_snapshot.emplace_back(node, detail::G().val[node]);
    }
    return *m_sys;
  }
  // true when the next compile() / solve() will build a new system (handles from
  // slpx_problem_system taken before that are stale)
  bool needs_compile() const { return !m_sys; }

  const SolveReport& report() const { return m_report; }
  const VariableF64& cost() const { return *m_f; }
  const std::vector<VariableF64>& decision_variables() const { return m_decision_variables; }
  const std::vector<VariableF64>& equality_constraints() const { return m_equality_constraints; }
  const std::vector<VariableF64>& inequality_constraints() const { return m_inequality_constraints; }
  const std::vector<double>& scales() const { return m_scales; }
  const std::vector<double>& slack() const { return m_s; }
  const std::vector<double>& equality_duals() const { return m_y; }
  const std::vector<double>& inequality_duals() const { return m_z; }

 private:
  void invalidate() {
    m_sys.reset();
    m_param_snapshot.clear();
  }

  static ExpressionType max_type(const std::vector<VariableF64>& v) {
    ExpressionType t = ExpressionType::NONE;
    for (auto& e : v) t = std::max(t, e.type());
    return t;
  }

  bool has_conflicting_bounds(const std::vector<double>& V) {
    const auto& st = m_sys->structure();
    auto& g = detail::G();
    const double inf = std::numeric_limits<double>::infinity();
    std::vector<std::pair<double, double>> b(st.n, {-inf, inf});
    // row -> (count, col, value) from the CSC pattern of A_i
    std::vector<int> cnt(st.m_i, 0), col(st.m_i, -1);
    std::vector<double> coef(st.m_i, 0.0);
    for (int c = 0; c < st.n; ++c)
      for (int p = st.Ai.colptr[c]; p < st.Ai.colptr[c + 1]; ++p) {
        const int r = st.Ai.rowidx[p];
        ++cnt[r];
        col[r] = c;
        coef[r] = V[st.off_Ai + p];
      }
    bool conflict = false;
    for (int r = 0; r < st.m_i; ++r) {
      if (m_inequality_constraints[r].type() != ExpressionType::LINEAR || cnt[r] != 1) continue;
      const NodeId var = m_decision_variables[col[r]].expr;
      const double saved = g.val[var];
      g.val[var] = 0.0;
      const double constant_term = m_inequality_constraints[r].value();
      g.val[var] = saved;
      const double detected = -constant_term / coef[r];
      auto& [lo, hi] = b[col[r]];
      if (coef[r] < 0.0 && detected < hi) hi = detected;
      else if (coef[r] > 0.0 && detected > lo) lo = detected;
      if (lo > hi) conflict = true;
    }
    return conflict;
  }

  std::vector<VariableF64> m_decision_variables;
  std::optional<VariableF64> m_f;
  std::vector<VariableF64> m_equality_constraints;
  std::vector<VariableF64> m_inequality_constraints;
  std::vector<slpx::IterationCallback> m_iteration_callbacks, m_persistent_iteration_callbacks;
  std::unique_ptr<slpx::NewtonSystem> m_sys;
  std::vector<std::pair<NodeId, double>> m_param_snapshot;  // (parameter node, value at compile time)
  std::vector<double> m_scales, m_s, m_y, m_z;
  SolveReport m_report;
};


// slp::Problem<Scalar> (include/sleipnir/optimization/problem.hpp:66-67): only double exists.
template <typename Scalar>
class Problem;
template <>
class Problem<double> : public ProblemF64 {
 public:
  using ProblemF64::ProblemF64;
};

}  // namespace slp
