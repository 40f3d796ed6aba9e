// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/ad.hpp header).
//
// Restates include/sleipnir/optimization/problem.hpp (:78-104 decision_variable,
// :151-234 minimize/subject_to, :281-679 solve — interior-point branch :512-668)
// and util/bounds.hpp:55-190 (conflicting-bound detection only; bound projection
// is compiled out by default in the reference, CMakeLists.txt:34-38).
//
// Deviation (documented): the reference dispatches unconstrained problems to
// newton() and equality-only problems to sqp() (problem.hpp:335,403); every
// BASELINE.json configuration has inequalities and takes the IPM branch, so the
// oracle always runs the IPM branch (with m_e and/or m_i possibly zero).
#pragma once

#include <memory>
#include <optional>
#include <vector>

#include "dsl.hpp"
#include "ipm.hpp"

namespace orc {

class Problem {
 public:
  Var decision_variable() {
    m_decision_variables.emplace_back();
    return m_decision_variables.back();
  }
  // problem.hpp:91-104: row-major creation order
  VarMat decision_variable(int rows, int cols = 1) {
    VarMat vars(rows, cols);
    for (int r = 0; r < rows; ++r)
      for (int c = 0; c < cols; ++c) {
        m_decision_variables.emplace_back();
        vars(r, c) = m_decision_variables.back();
      }
    return vars;
  }
  void minimize(const Var& cost) { m_f = cost; }
  void maximize(const Var& objective) { m_f = -objective; }
  void subject_to_eq(const std::vector<Var>& c) {
    m_equality_constraints.insert(m_equality_constraints.end(), c.begin(), c.end());
  }
  void subject_to_ineq(const std::vector<Var>& c) {
    m_inequality_constraints.insert(m_inequality_constraints.end(), c.begin(), c.end());
  }

  Type cost_function_type() const { return m_f ? m_f->type() : NONE; }
  Type equality_constraint_type() const { return max_type(m_equality_constraints); }
  Type inequality_constraint_type() const { return max_type(m_inequality_constraints); }

  int num_decision_variables() const { return static_cast<int>(m_decision_variables.size()); }
  int num_equality_constraints() const { return static_cast<int>(m_equality_constraints.size()); }
  int num_inequality_constraints() const {
    return static_cast<int>(m_inequality_constraints.size());
  }

  void add_callback(IterationCallback cb) { m_iteration_callbacks.push_back(std::move(cb)); }
  void clear_callbacks() { m_iteration_callbacks.clear(); }  // problem.hpp:712

  // Everything Problem::solve builds before calling interior_point (problem.hpp:517-660)
  struct Evaluators {
    std::vector<Expr*> x_ad, y_ad, z_ad, c_e_ad, c_i_ad;
    Expr* f = nullptr;
    Gradient g;
    Hessian H_f, H_c;
    Jacobian A_e, A_i;
    std::vector<Graph> c_e_graphs, c_i_graphs;
    Graph f_graph;
    ProblemScaling scaling;
    MatrixCallbacks callbacks;
  };

  // Builds evaluators + scaling at the current variable values.  Returns
  // GLOBALLY_INFEASIBLE on conflicting bounds (problem.hpp:597-606), else SUCCESS.
  ExitStatus setup(Evaluators& ev) {
    for (auto& v : m_decision_variables) ev.x_ad.push_back(v.e);
    ev.f = m_f ? m_f->e : constant(0.0);
    for (auto& v : m_equality_constraints) ev.c_e_ad.push_back(v.e);
    for (auto& v : m_inequality_constraints) ev.c_i_ad.push_back(v.e);
    VarMat y_ad(num_equality_constraints(), 1), z_ad(num_inequality_constraints(), 1);
    for (auto& v : y_ad.s) v = Var();
    for (auto& v : z_ad.s) v = Var();
    for (auto& v : y_ad.s) ev.y_ad.push_back(v.e);
    for (auto& v : z_ad.s) ev.z_ad.push_back(v.e);

    ev.g = Gradient(ev.f, ev.x_ad);                 // :535
    ev.H_f = Hessian(ev.f, ev.x_ad, true);          // :542
    // :547-548  -y_adᵀ c_e_ad - z_adᵀ c_i_ad
    VarMat c_e_m(num_equality_constraints(), 1), c_i_m(num_inequality_constraints(), 1);
    for (int i = 0; i < c_e_m.rows; ++i) c_e_m(i, 0) = m_equality_constraints[i];
    for (int i = 0; i < c_i_m.rows; ++i) c_i_m(i, 0) = m_inequality_constraints[i];
    VarMat lag = (-y_ad.T()) * c_e_m - z_ad.T() * c_i_m;
    ev.H_c = Hessian(lag(0, 0).e, ev.x_ad, true);
    ev.A_e = Jacobian(ev.c_e_ad, ev.x_ad);          // :555
    ev.A_i = Jacobian(ev.c_i_ad, ev.x_ad);          // :560

    ev.f_graph = topological_sort(ev.f);
    for (Expr* c : ev.c_e_ad) ev.c_e_graphs.push_back(topological_sort(c));
    for (Expr* c : ev.c_i_ad) ev.c_i_graphs.push_back(topological_sort(c));

    // get_bounds (bounds.hpp:55-190): only the conflict check affects the solve
    if (has_conflicting_bounds(ev.A_i.value())) return ExitStatus::GLOBALLY_INFEASIBLE;

    // :615-616 scaling at x0
    ev.scaling = ProblemScaling(sparse_row_to_dense(ev.g.value()), ev.A_e.value(), ev.A_i.value());

    make_callbacks(ev);
    return ExitStatus::SUCCESS;
  }

  ExitStatus solve(const Options& options = Options{}, SolveStats* stats = nullptr,
                   const std::vector<int>* user_perm = nullptr) {
    Vec x(m_decision_variables.size());
    for (size_t i = 0; i < x.size(); ++i) x[i] = m_decision_variables[i].value();

    Type f_type = cost_function_type(), ce_type = equality_constraint_type(),
         ci_type = inequality_constraint_type();
    if (f_type <= CONSTANT && ce_type <= CONSTANT && ci_type <= CONSTANT)
      return ExitStatus::SUCCESS;  // :304-313

    m_ev = std::make_unique<Evaluators>();
    ExitStatus st = setup(*m_ev);
    if (st != ExitStatus::SUCCESS) return st;

    std::vector<IterationCallback> callbacks = m_iteration_callbacks;
    auto t0 = std::chrono::steady_clock::now();
    // problem.hpp:335, 403, 512: the solver follows the kinds of constraints present
    if (m_equality_constraints.empty() && m_inequality_constraints.empty()) {
      int iterations = 0;
      st = newton(m_ev->callbacks, callbacks, options, x, iterations, stats);
      if (stats) stats->iterations = iterations;
      last_s.clear();
      last_y.clear();
      last_z.clear();
    } else if (m_inequality_constraints.empty()) {
      int iterations = 0;
      Vec y(m_equality_constraints.size(), 0.0);  // problem.hpp:503-504
      st = sqp(m_ev->callbacks, callbacks, options, x, y, iterations, stats, user_perm);
      if (stats) stats->iterations = iterations;
      last_s.clear();
      last_y = y;
      last_z.clear();
    } else {
      st = interior_point(m_ev->callbacks, callbacks, options, x, stats, user_perm, &last_s, &last_y,
                          &last_z);
    }
    if (stats)
      stats->t_total = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    for (size_t i = 0; i < x.size(); ++i) m_decision_variables[i].set_value(x[i]);  // :676
    return st;
  }

  Evaluators* evaluators() { return m_ev.get(); }
  void ensure_setup() {
    if (!m_ev) {
      m_ev = std::make_unique<Evaluators>();
      setup(*m_ev);
    }
  }
  std::vector<Var>& decision_variables() { return m_decision_variables; }
  Vec last_s, last_y, last_z;

 private:
  static Type max_type(const std::vector<Var>& v) {
    if (v.empty()) return NONE;
    Type t = NONE;
    for (auto& e : v) t = tmax(t, e.type());
    return t;
  }

  void set_values(const std::vector<Expr*>& nodes, const Vec& v) {
    for (size_t i = 0; i < nodes.size(); ++i) nodes[i]->val = v[i];
  }

  // problem.hpp:618-660
  void make_callbacks(Evaluators& ev) {
    MatrixCallbacks& cb = ev.callbacks;
    cb.num_decision_variables = num_decision_variables();
    cb.num_equality_constraints = num_equality_constraints();
    cb.num_inequality_constraints = num_inequality_constraints();
    cb.scaling = ev.scaling;
    Evaluators* e = &ev;
    cb.f = [this, e](const Vec& x) {
      set_values(e->x_ad, x);
      update_values(e->f_graph);
      return e->scaling.f * e->f->val;
    };
    cb.g = [this, e](const Vec& x) {
      set_values(e->x_ad, x);
      Vec g = sparse_row_to_dense(e->g.value());
      for (auto& v : g) v *= e->scaling.f;
      return g;
    };
    auto set_duals = [this, e](const Vec& y, const Vec& z) {
      for (size_t i = 0; i < e->y_ad.size(); ++i) e->y_ad[i]->val = e->scaling.c_e[i] * y[i];
      for (size_t i = 0; i < e->z_ad.size(); ++i) e->z_ad[i]->val = e->scaling.c_i[i] * z[i];
    };
    cb.H = [this, e, set_duals](const Vec& x, const Vec& y, const Vec& z) {
      set_values(e->x_ad, x);
      set_duals(y, z);
      return add(e->H_f.value(), e->H_c.value(), e->scaling.f, 1.0);
    };
    cb.H_c = [this, e, set_duals](const Vec& x, const Vec& y, const Vec& z) {
      set_values(e->x_ad, x);
      set_duals(y, z);
      return e->H_c.value();
    };
    cb.c_e = [this, e](const Vec& x) {
      set_values(e->x_ad, x);
      Vec c(e->c_e_ad.size());
      // variable_matrix.hpp:993-1004: each element's graph evaluated separately
      for (size_t i = 0; i < c.size(); ++i) {
        update_values(e->c_e_graphs[i]);
        c[i] = e->scaling.c_e[i] * e->c_e_ad[i]->val;
      }
      return c;
    };
    cb.A_e = [this, e](const Vec& x) {
      set_values(e->x_ad, x);
      return row_scaled(e->scaling.c_e, e->A_e.value());
    };
    cb.c_i = [this, e](const Vec& x) {
      set_values(e->x_ad, x);
      Vec c(e->c_i_ad.size());
      for (size_t i = 0; i < c.size(); ++i) {
        update_values(e->c_i_graphs[i]);
        c[i] = e->scaling.c_i[i] * e->c_i_ad[i]->val;
      }
      return c;
    };
    cb.A_i = [this, e](const Vec& x) {
      set_values(e->x_ad, x);
      return row_scaled(e->scaling.c_i, e->A_i.value());
    };
  }

  // bounds.hpp:55-190, reduced to the conflict test
  bool has_conflicting_bounds(const CSC& A_i) {
    const int n = num_decision_variables();
    const double inf = std::numeric_limits<double>::infinity();
    std::vector<std::pair<double, double>> b(n, {-inf, inf});
    CSC At = transpose(A_i);  // column r of At = row r of A_i
    bool conflict = false;
    for (int r = 0; r < static_cast<int>(m_inequality_constraints.size()); ++r) {
      if (m_inequality_constraints[r].type() != LINEAR) continue;
      int nz = At.colptr[r + 1] - At.colptr[r];
      if (nz != 1) continue;
      double coeff = At.val[At.colptr[r]];
      int var = At.rowidx[At.colptr[r]];
      double saved = m_decision_variables[var].e->val;
      double constant_term;
      if (saved != 0.0) {
        m_decision_variables[var].set_value(0.0);
        constant_term = m_inequality_constraints[r].value();
        m_decision_variables[var].set_value(saved);
      } else {
        constant_term = m_inequality_constraints[r].value();
      }
      double detected = -constant_term / coeff;
      auto& [lo, hi] = b[var];
      if (coeff < 0.0 && detected < hi) hi = detected;
      else if (coeff > 0.0 && detected > lo) lo = detected;
      if (lo > hi) conflict = true;
    }
    return conflict;
  }

  std::vector<Var> m_decision_variables;
  std::optional<Var> m_f;
  std::vector<Var> m_equality_constraints;
  std::vector<Var> m_inequality_constraints;
  std::vector<IterationCallback> m_iteration_callbacks;
  std::unique_ptr<Evaluators> m_ev;
};

}  // namespace orc
